#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-quick}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_precision_modes.py tests/test_gpu_next_rows.py -m gpu -q -x 2>&1 | tail -n 3
EC_TIMELINE=1 timeout 120 python tools/timeline_probe.py 2>&1 | grep timeline | tail -3 | cut -c1-330
for r in 1 2 3; do
  timeout 200 python bench.py --no-cpu-baseline --no-episode --no-alt --steps 20 > $O/bench${r}.json 2>/dev/null; python tools/bench_line.py < $O/bench${r}.json | cut -c1-60
done
