#!/bin/bash
# A/B of the backbone attention kernels on one MI355X (run through gpurun): op-level tests, then the per-kernel average of
# tools/attn_bench.py under rocprofv3 for the pipelined (default) and the unpipelined (EC_ATTN_PIPE=0) kernel.
TAG=${1:-attn_ab}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 200 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention" > $OUT/tests.log 2>&1; tail -n 5 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
for v in 0; do
  for prec in 3; do
    EC_ATTN_PIPE=$v PREC=$prec ITERS=30 timeout 90 rocprofv3 --kernel-trace --stats -d $OUT/p${v}_$prec -o r -- python $R/tools/attn_bench.py > /dev/null 2> $OUT/p${v}_$prec.err
    DB=$(ls $OUT/p${v}_$prec/*/*results.db $OUT/p${v}_$prec/*results.db 2>/dev/null | head -1)
    python $R/tools/rocpd_stats.py $DB $OUT/stats_p${v}_$prec.csv
    echo "pipe=$v prec=$prec"; grep "attn" $OUT/stats_p${v}_$prec.csv | cut -c1-200
  done
done
