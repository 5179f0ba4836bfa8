#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 1 0; do
  echo "EC_ENC_CHAIN=$v EC_OVERLAP=0"
  EC_OVERLAP=0 EC_ENC_CHAIN=$v EC_TIMELINE=1 timeout 120 python tools/timeline_probe.py 2>&1 | grep timeline | tail -2 | cut -c1-330
done
