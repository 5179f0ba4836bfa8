#!/bin/bash
# Conformance records of the CURRENT library for every benched configuration (tools/conformance.py: disjoint pairs x 2 weight seeds against
# the CPU oracle, with the near-tie-guard analysis): fp16 / mixed and bf16x3 / bf16x3 on cfg1 (ViT-S/14 @ 224), cfg2, cfg4, cfg5 (the oracle's
# answers are cached under /tmp and shared by the two precisions); both on cfg2 with planted activation outliers.
#   usage: bash tools/gpu_conformance_all.sh <tag>   -> gpurun_out/<tag>/conformance_*.json   (copy into profiles/ as r<NN>_conformance_*)
export TAG=${1:-r05conf}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python tools/conformance.py --config cfg2 --out $O/conformance_fp16_mixed.json > $O/cfg2.log 2>&1
python tools/conformance.py --config cfg1 --out $O/conformance_cfg1_fp16_mixed.json > $O/cfg1.log 2>&1
python tools/conformance.py --config cfg4 --batches 16 --out $O/conformance_cfg4_fp16_mixed.json > $O/cfg4.log 2>&1
python tools/conformance.py --config cfg5 --batches 16 --out $O/conformance_cfg5_fp16_mixed.json > $O/cfg5.log 2>&1
python tools/conformance.py --config cfg2 --backbone bf16x3 --head bf16x3 --out $O/conformance_bf16x3_bf16x3.json > $O/cfg2x3.log 2>&1
python tools/conformance.py --config cfg1 --backbone bf16x3 --head bf16x3 --out $O/conformance_cfg1_bf16x3_bf16x3.json > $O/cfg1x3.log 2>&1
python tools/conformance.py --config cfg4 --batches 16 --backbone bf16x3 --head bf16x3 --out $O/conformance_cfg4_bf16x3_bf16x3.json > $O/cfg4x3.log 2>&1
python tools/conformance.py --config cfg5 --batches 16 --backbone bf16x3 --head bf16x3 --out $O/conformance_cfg5_bf16x3_bf16x3.json > $O/cfg5x3.log 2>&1
python tools/conformance.py --config cfg2 --outliers --out $O/conformance_cfg2_outliers_fp16_mixed.json > $O/cfg2o.log 2>&1
python tools/conformance.py --config cfg2 --outliers --backbone bf16x3 --head bf16x3 --out $O/conformance_cfg2_outliers_bf16x3_bf16x3.json > $O/cfg2ox3.log 2>&1
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", os.environ.get("TAG", "r04conf"), "conformance_*.json"))):
    p = json.load(open(f))["pooled"]
    print(os.path.basename(f), {k: p[k] for k in ("pairs", "n_valid", "flips", "max_clean", "p99", "median", "frac_gt_1e3", "clean_samples", "pck_vs_oracle")})
PY
