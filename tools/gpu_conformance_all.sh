#!/bin/bash
# Conformance records of the CURRENT library for every benched configuration (tools/conformance.py: disjoint pairs x 2 weight seeds against
# the CPU oracle, with the near-tie-guard analysis): the headline precision (fp16 / mixed) and the conforming one (fp16x2 / bf16x3, round 6)
# on cfg1 (ViT-S/14 @ 224), cfg2, cfg4, cfg5 (the oracle's answers are cached under /tmp and shared by the precisions); both on cfg2 with
# planted activation outliers; and both through the reference's evaluation PROTOCOL (ec_forward_episodes, 17 episodes x 15 queries per
# weight seed in calls sized like bench.py's episode leg) on cfg2 and cfg4.
#   usage: bash tools/gpu_conformance_all.sh <tag> [modes]  -> gpurun_out/<tag>/conformance_*.json   (copy into profiles/ as r<NN>_conformance_*)
#          modes: space-separated "backbone/head" pairs, default "fp16/mixed fp16x2/bf16x3"
export TAG=${1:-r06conf}
MODES=${2:-"fp16/mixed fp16x2/bf16x3"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for m in $MODES; do
  BB=${m%/*}; HD=${m#*/}
  python tools/conformance.py --config cfg2 --backbone $BB --head $HD --out $O/conformance_${BB}_${HD}.json > $O/cfg2_$BB.log 2>&1
  python tools/conformance.py --config cfg1 --backbone $BB --head $HD --out $O/conformance_cfg1_${BB}_${HD}.json > $O/cfg1_$BB.log 2>&1
  python tools/conformance.py --config cfg4 --batches 16 --backbone $BB --head $HD --out $O/conformance_cfg4_${BB}_${HD}.json > $O/cfg4_$BB.log 2>&1
  python tools/conformance.py --config cfg5 --batches 16 --backbone $BB --head $HD --out $O/conformance_cfg5_${BB}_${HD}.json > $O/cfg5_$BB.log 2>&1
  python tools/conformance.py --config cfg2 --outliers --backbone $BB --head $HD --out $O/conformance_cfg2_outliers_${BB}_${HD}.json > $O/cfg2o_$BB.log 2>&1
  python tools/conformance.py --config cfg2 --episodes 17 --backbone $BB --head $HD --out $O/conformance_episodes_cfg2_${BB}_${HD}.json > $O/ep2_$BB.log 2>&1
  python tools/conformance.py --config cfg4 --episodes 17 --backbone $BB --head $HD --out $O/conformance_episodes_cfg4_${BB}_${HD}.json > $O/ep4_$BB.log 2>&1
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", os.environ.get("TAG", "r06conf"), "conformance_*.json"))):
    p = json.load(open(f))["pooled"]
    print(os.path.basename(f), {k: p[k] for k in ("pairs", "n_valid", "flips", "max_clean", "p99", "median", "frac_gt_1e3", "clean_samples", "pck_vs_oracle")})
PY
