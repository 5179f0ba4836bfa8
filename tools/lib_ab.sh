#!/bin/bash
# Same-box A/B of whole LIBRARY VERSIONS (the driver-timed figures drifted down three rounds running - r03 5529, r04 5473,
# r05 5412 images/s pipelined - each step inside box-to-box noise).  Two halves:
#   bash tools/lib_ab.sh build <commit> [<commit> ...]   HERE (build container, has .git): every commit's tree is exported into
#        ab_libs/<commit>/ and its library cross-compiled there (git-ignored, NOT gpurun-ignored: the trees travel to the GPU box)
#   bash tools/lib_ab.sh run [rounds]                    on the GPU box (through gpurun): every tree's OWN bench.py, pipelined and
#        --no-pipeline, interleaved `rounds` times (default 6) plus the working tree itself as HEAD -> gpurun_out/lib_ab/summary.txt
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
if [ "$1" = "build" ]; then
  shift
  mkdir -p ab_libs
  for c in "$@"; do
    rm -rf ab_libs/$c && mkdir -p ab_libs/$c
    git archive $c | tar -x -C ab_libs/$c
    rm -rf ab_libs/$c/profiles ab_libs/$c/gpurun_out ab_libs/$c/tests/golden          # (not needed by bench.py; keeps the snapshot small)
    (cd ab_libs/$c && python -m edgecape_amd.build > build.log 2>&1 && echo "built ab_libs/$c: $(ls -la edgecape_amd/libedgecape_hip.so | awk '{print $5}') bytes")
  done
  exit 0
fi
ROUNDS=${2:-6}
O=$R/gpurun_out/lib_ab
mkdir -p $O
TREES="$(ls -d ab_libs/*/ 2>/dev/null | sed 's#/$##') ."
for r in $(seq 1 $ROUNDS); do
  for t in $TREES; do
    n=$(basename $t); [ "$t" = "." ] && n=HEAD
    extra=""; grep -q "sustained-seconds" $t/bench.py && extra="--sustained-seconds 0"
    (cd $t && python bench.py --no-cpu-baseline --no-episode --no-alt $extra --steps 24 --warmup 3 2>/dev/null | tail -1 > $O/${n}_pipe_$r.json)
    (cd $t && python bench.py --no-cpu-baseline --no-episode --no-alt $extra --no-pipeline --steps 24 --warmup 3 2>/dev/null | tail -1 > $O/${n}_plain_$r.json)
  done
done
python - <<'PY' | tee $O/summary.txt
import glob, json, os, statistics as st
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "lib_ab")
names = sorted({os.path.basename(f).rsplit("_", 2)[0] for f in glob.glob(O + "/*_pipe_*.json")})
print("library versions on ONE box, interleaved rounds; images/s (cfg2, fp16 / mixed, 24 timed steps): median [min .. max] over the rounds")
for n in names:
    row = []
    for leg in ("pipe", "plain"):
        v = []
        for f in sorted(glob.glob(f"{O}/{n}_{leg}_*.json")):
            try:
                v.append(json.load(open(f))["value"])
            except Exception:
                pass
        row.append(f"{leg}: {st.median(v):7.1f} [{min(v):7.1f} .. {max(v):7.1f}] n={len(v)}" if v else f"{leg}: no data")
    print(f"{n:10s} " + "   ".join(row))
PY
