#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04h
mkdir -p $O
cd $R
python tools/dual_backbone_probe.py 2>/dev/null > $O/dual_backbone.txt; cat $O/dual_backbone.txt
bash tools/gpu_round_end.sh r04h --no-tests
