#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/encchain3; mkdir -p $O
for cfg in "1 1" "1 0" "0 1" "0 0"; do set -- $cfg
  echo "EC_ENC_CHAIN=$1 SIDE_PRIO=$2"
  if [ $2 = 1 ]; then export EC_SIDE_PRIO=1; else unset EC_SIDE_PRIO; fi
  EC_ENC_CHAIN=$1 EC_TIMELINE=1 timeout 120 python tools/timeline_probe.py 2>&1 | grep timeline | tail -1 | cut -c1-330
  for r in 1 2; do EC_ENC_CHAIN=$1 timeout 200 python bench.py --no-cpu-baseline --no-episode --no-alt --steps 20 > $O/b.json 2>/dev/null; python tools/bench_line.py < $O/b.json | cut -c1-40; done
done
