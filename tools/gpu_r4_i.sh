#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04i
mkdir -p $O
cd $R
python tools/dual_pipelined_probe.py 2>/dev/null > $O/dual_pipelined.txt; cat $O/dual_pipelined.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_mfma_probe tools/valu_mfma_probe.hip 2>/dev/null && /tmp/valu_mfma_probe > $O/valu_mfma_probe.txt 2>&1; tail -n 12 $O/valu_mfma_probe.txt
