#!/bin/bash
# Which side makes the conforming mode's two flips above the fp32 floor (cfg4, cfg5)?  Conformance at scale with ONE side exact:
# bf16x3 backbone + fp32 head, and fp32 backbone + bf16x3 head (the oracle's answers are computed once per configuration and shared).
export TAG=${1:-r05src}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for c in cfg4 cfg5; do
  python tools/conformance.py --config $c --batches 16 --backbone bf16x3 --head fp32 --out $O/conformance_${c}_bf16x3_fp32.json > $O/${c}_bb.log 2>&1
  python tools/conformance.py --config $c --batches 16 --backbone fp32 --head bf16x3 --out $O/conformance_${c}_fp32_bf16x3.json > $O/${c}_hd.log 2>&1
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", os.environ["TAG"], "conformance_*.json"))):
    d = json.load(open(f)); p = d["pooled"]
    print(os.path.basename(f), {k: p[k] for k in ("pairs", "n_valid", "flips", "max_clean", "p99", "frac_gt_1e3")})
PY
