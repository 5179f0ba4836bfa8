#!/bin/bash
# Schedule sweep of the 8-phase GEMM + PMC of the picked configurations (one gpurun call).  usage: bash tools/gpu_g8s.sh [outdir]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/${1:-gpurun_out/g8s}; mkdir -p $O
cd $R
python tools/g8_sched.py sweep --out $O > $O/sweep.log 2>&1; tail -n 60 $O/sweep.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $O/tcc_counters.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/p_fetch -o r -- python $R/tools/g8_sched.py replay --manifest $O/manifest.json --iters 3 > $O/pmc.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/p_hit -o r -- python $R/tools/g8_sched.py replay --manifest $O/manifest.json --iters 3 >> $O/pmc.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/p_wr -o r -- python $R/tools/g8_sched.py replay --manifest $O/manifest.json --iters 3 >> $O/pmc.log 2>&1
cd $R
python tools/g8_sched_pmc.py --manifest $O/manifest.json --iters 3 $(ls $O/p_*/*results.db $O/p_*/*/*results.db 2>/dev/null) --out $O/pmc.csv 2>> $O/pmc.log
cat $O/pmc.csv; tail -n 5 $O/pmc.log
rm -rf $O/p_fetch $O/p_hit $O/p_wr
