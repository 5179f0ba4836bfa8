#!/bin/bash
# conformance records of the CURRENT library for every benched configuration, copied into profiles/ as r05_conformance_*, then the headline
# bench line once more (it quotes the records only for the library hash they carry)   usage: bash tools/gpu_conf_then_bench.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/gpu_conformance_all.sh r05conf
for f in gpurun_out/r05conf/conformance_*.json; do cp $f profiles/r05_$(basename $f); done
mkdir -p gpurun_out/r05b
python bench.py > gpurun_out/r05b/bench.json 2> gpurun_out/r05b/bench.err; python tools/bench_line.py < gpurun_out/r05b/bench.json | cut -c1-300
python bench.py --shots 5 --batch 16 --no-alt --steps 10 --cpu-batches 16 --cpu-runs 3 > gpurun_out/r05b/cfg4_5shot_b16.json 2>> gpurun_out/r05b/bench.err
cp profiles/r05_conformance_*.json gpurun_out/r05b/
