#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04k
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "nonsquare or golden or backbone" -s > $O/nonsq.log 2>&1; grep -E "^\(|^head_|passed|failed|Error|assert" $O/nonsq.log | cut -c1-260 | tail -30
python -m pytest tests/test_gpu_errors.py -m gpu -q > $O/err.log 2>&1; tail -n 3 $O/err.log
