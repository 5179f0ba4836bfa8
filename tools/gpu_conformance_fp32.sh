#!/bin/bash
# The floor of the flip count: the EXACT mode (fp32 / fp32: fp32 MFMAs, 6e-7 against the oracle) on the configurations where the
# conforming mode shows one flip - does an fp32 GPU implementation flip the same near-tie against the fp32 CPU oracle?
#   usage: bash tools/gpu_conformance_fp32.sh <tag>   -> gpurun_out/<tag>/conformance_*fp32_fp32.json
export TAG=${1:-r05fp32}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python tools/conformance.py --config cfg2 --backbone fp32 --head fp32 --out $O/conformance_fp32_fp32.json > $O/cfg2.log 2>&1
python tools/conformance.py --config cfg2 --backbone bf16x3 --head bf16x3 --out $O/conformance_bf16x3_bf16x3.json > $O/cfg2x3.log 2>&1
python tools/conformance.py --config cfg4 --batches 16 --backbone fp32 --head fp32 --out $O/conformance_cfg4_fp32_fp32.json > $O/cfg4.log 2>&1
python tools/conformance.py --config cfg5 --batches 16 --backbone fp32 --head fp32 --out $O/conformance_cfg5_fp32_fp32.json > $O/cfg5.log 2>&1
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", os.environ["TAG"], "conformance_*.json"))):
    d = json.load(open(f)); p = d["pooled"]
    print(os.path.basename(f), {k: p[k] for k in ("pairs", "n_valid", "flips", "max_all", "max_clean", "p99", "frac_gt_1e3", "pck_vs_oracle")})
PY
