#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/headgemm; mkdir -p $O
S1="encin:13568:768:256,encout:13568:256:256,ffn2:13568:256:768,kvdec:10368:3072:256,kvskel:10368:1024:256,inproj:10368:256:768,fold:10368:256:512"
S3="encin:13568:768:768,encout:13568:256:768,ffn2:13568:256:2304,kvdec:10368:3072:768,kvskel:10368:1024:768,inproj:10368:256:2304,fold:10368:256:1536"
NOCHECK=1 SHAPES=$S1 timeout 120 python tools/gemm_bench.py bf16x3 2>&1 | grep -v amdgpu | tee $O/x3.txt
NOCHECK=1 SHAPES=$S3 timeout 120 python tools/gemm_bench.py bf16 2>&1 | grep -v amdgpu | tee $O/g8_k3.txt
NOCHECK=1 SHAPES=$S1 timeout 120 python tools/gemm_bench.py bf16 2>&1 | grep -v amdgpu | tee $O/g8_k1.txt
