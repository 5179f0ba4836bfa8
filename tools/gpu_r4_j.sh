#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04j
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "dynamic_keypoint" -s > $O/dynk.log 2>&1; grep -E "^K |passed|failed|Error" $O/dynk.log | cut -c1-200
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -n 6 $O/tests.log
