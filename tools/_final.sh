#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/gpu_round_end.sh r04n
