#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/gates2
mkdir -p $O
cd $R
bash tools/gpu_r5_e.sh
for cb in cfg4:2 cfg5:2; do
  c=${cb%%:*}; nb=${cb##*:}
  python tools/conformance.py --config $c --batches $nb --out $O/gate_$c.json > $O/$c.log 2>&1
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "gates2", "gate_*.json"))):
    d = json.load(open(f)); p = d["pooled"]
    print(os.path.basename(f), {k: p[k] for k in ("pairs", "n_valid", "flips", "max_clean", "p99", "median", "frac_gt_1e3", "clean_samples", "pck_vs_oracle")},
          "seed flips", [s["flips"] for s in d["per_weight_seed"]], "seed max_clean", [s["max_clean"] for s in d["per_weight_seed"]])
PY
