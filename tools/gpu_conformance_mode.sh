#!/bin/bash
# Conformance records of the CURRENT library for ONE precision mode on every benched configuration + cfg2 with planted outliers
# (tools/conformance.py; the oracle's answers are cached under /tmp and shared by the runs of one box).
#   usage: bash tools/gpu_conformance_mode.sh <tag> <backbone precision> <head precision>   -> gpurun_out/<tag>/conformance_*.json
export TAG=${1:-r06conf}
BB=${2:-bf16x3}
HD=${3:-bf16x3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python tools/conformance.py --config cfg2 --backbone $BB --head $HD --out $O/conformance_cfg2_${BB}_${HD}.json > $O/cfg2.log 2>&1
python tools/conformance.py --config cfg4 --batches 16 --backbone $BB --head $HD --out $O/conformance_cfg4_${BB}_${HD}.json > $O/cfg4.log 2>&1
python tools/conformance.py --config cfg5 --batches 16 --backbone $BB --head $HD --out $O/conformance_cfg5_${BB}_${HD}.json > $O/cfg5.log 2>&1
python tools/conformance.py --config cfg1 --backbone $BB --head $HD --out $O/conformance_cfg1_${BB}_${HD}.json > $O/cfg1.log 2>&1
python tools/conformance.py --config cfg2 --outliers --backbone $BB --head $HD --out $O/conformance_cfg2_outliers_${BB}_${HD}.json > $O/cfg2o.log 2>&1
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", os.environ["TAG"], "conformance_*.json"))):
    p = json.load(open(f))["pooled"]
    print(os.path.basename(f), {k: p[k] for k in ("pairs", "n_valid", "flips", "max_clean", "p99", "median", "frac_gt_1e3", "clean_samples", "pck_vs_oracle")})
PY
