R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/chain_alone; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/prof -o r1 -- python -m pytest $R/tests/test_gpu_ops.py -m gpu -q -x -k "row_chain" -p no:cacheprovider > $OUT/log.txt 2>&1
tail -n 2 $OUT/log.txt
cd $R
DB=$(ls $OUT/prof/*/*results.db $OUT/prof/*results.db 2>/dev/null | head -1)
python - <<PY
import sqlite3
db=sqlite3.connect("$DB"); cur=db.cursor()
cols=[r[1] for r in cur.execute("pragma table_info(kernels)")]
nc="name" if "name" in cols else "kernel_name"
for n,s,e in cur.execute(f"select {nc}, start, end from kernels order by start"):
    if "chain" in n: print("chain_kernel", (e-s)/1e3, "us")
PY
