#!/bin/bash
# the observed conformance numbers the at-scale GATES of tests/test_gpu_precision_modes.py are set from (4 disjoint batches x 2 weight seeds
# per configuration: what test_headline_conformance_at_scale runs)   -> gpurun_out/gates/
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/gates
mkdir -p $O
cd $R
for cb in cfg1:4 cfg2:4 cfg4:3 cfg5:3; do
  c=${cb%%:*}; nb=${cb##*:}
  T0=$(date +%s)
  python tools/conformance.py --config $c --batches $nb --out $O/gate_$c.json > $O/$c.log 2>&1
  echo "$c seconds: $(( $(date +%s) - T0 ))"
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "gates", "gate_*.json"))):
    d = json.load(open(f)); p = d["pooled"]
    print(os.path.basename(f), {k: p[k] for k in ("pairs", "n_valid", "flips", "max_clean", "p99", "median", "frac_gt_1e3", "clean_samples", "pck_vs_oracle")},
          "seed flips", [s["flips"] for s in d["per_weight_seed"]], "seed max_clean", [s["max_clean"] for s in d["per_weight_seed"]])
PY
