#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-kptchain}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_precision_modes.py tests/test_gpu_next_rows.py -m gpu -q -x 2>&1 | tail -n 4
for v in 1 0; do
  echo "EC_KPT_CHAIN=$v"
  EC_KPT_CHAIN=$v EC_TIMELINE=1 timeout 120 python tools/timeline_probe.py 2>&1 | grep timeline | tail -2 | cut -c1-330
done
for r in 1 2 3; do for v in 1 0; do
  EC_KPT_CHAIN=$v timeout 200 python bench.py --no-cpu-baseline --no-episode --no-alt --steps 20 > $O/bench${r}_$v.json 2>/dev/null; echo -n "kpt_chain=$v "; python tools/bench_line.py < $O/bench${r}_$v.json | cut -c1-60
done; done
