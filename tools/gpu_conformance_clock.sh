#!/bin/bash
# Round-4 first measurement set (one gpurun call): conformance of the bench's default precision on the reference's own configuration
# (ViT-S/14 @ 224) and on cfg2 with the near-tie-guard gap analysis; clock / MFMA-utilisation counters of the north-star kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04
mkdir -p $O/clock
cd $R
python tools/conformance.py --config cfg1 --out $O/conformance_cfg1_fp16_mixed.json > $O/conf_cfg1.log 2>&1; tail -n 2 $O/conf_cfg1.log | cut -c1-300
python tools/conformance.py --config cfg2 --out $O/conformance_fp16_mixed.json > $O/conf_cfg2.log 2>&1; tail -n 2 $O/conf_cfg2.log | cut -c1-300
python tools/conformance.py --config cfg1 --backbone bf16x3 --head bf16x3 --out $O/conformance_cfg1_bf16x3.json > $O/conf_cfg1_x3.log 2>&1
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_power_probe $R/tools/mfma_power_probe.hip
CTR="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES"
rocprofv3 --pmc $CTR --kernel-trace -d $O/clock/probe -o r -- /tmp/mfma_power_probe > $O/clock/probe.txt 2>&1
export SHAPES=qkv:20800:2304:768,sq4096:4096:4096:4096,fc1:20800:3072:768 VARIANTS=0 ITERS=20 REPS=1
rocprofv3 --pmc $CTR --kernel-trace -d $O/clock/g8_random -o r -- python $R/tools/g8_lab.py > $O/clock/g8_random.txt 2>&1
ZERO=1 rocprofv3 --pmc $CTR --kernel-trace -d $O/clock/g8_zero -o r -- python $R/tools/g8_lab.py > $O/clock/g8_zero.txt 2>&1
cd $R
for d in probe g8_random g8_zero; do
  DB=$(ls $O/clock/$d/*/*results.db $O/clock/$d/*results.db 2>/dev/null | head -1)
  python tools/clock_study.py $DB > $O/clock/$d.csv
done
rm -rf $O/clock/probe $O/clock/g8_random $O/clock/g8_zero
head -n 30 $O/clock/*.csv
