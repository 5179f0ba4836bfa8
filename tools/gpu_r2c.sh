set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
python tools/gemm_bench.py bf16 2>&1 | tee $O/gemm.txt
python -m pytest tests/test_gpu_ops.py -m gpu -q > $O/ops.log 2>&1; echo "ops rc $?"
tail -n 12 $O/ops.log
python - <<'PY' > $O/edge.txt 2>&1
import torch, ctypes as C
from edgecape_amd import _lib
lib=_lib.load()
K=128
def run(vals, bias=False, M=1024):
    vals = torch.tensor(vals)
    A = vals.repeat(M * K // vals.numel()).reshape(M, K).contiguous()
    W = torch.eye(K).repeat(2, 1)
    Cd = torch.empty(M, 256, device="cuda"); Ad, Wd = A.cuda(), W.cuda()
    b = torch.zeros(256, device="cuda")
    rc = lib.ec_op_linear(Ad.data_ptr(), Wd.data_ptr(), b.data_ptr() if bias else None, None, None, Cd.data_ptr(), M, 256, K, 0, 3, None)
    torch.cuda.synchronize()
    c = Cd.cpu()
    print(rc, "nan", int(torch.isnan(c).sum()), "of", c.numel(), "eq", bool(torch.equal(c[:, :K], A.half().float())), c[0, :16].tolist())
full=[0.0, 1.0, -1.0, 65504.0, 65519.0, 6.1e-5, 5.96e-8, 2.98e-8, 3.1e-8, 1.0 + 2 ** -11, 1.0 + 3 * 2 ** -11, 0.1, -0.3333333, 1e-3, 123.456, 2049.0]
run(full); run(full, bias=True)
run([0.0, 1.0, -1.0, 2.0]*4)
run([0.0, 1.0, -1.0, 65504.0]*4)
run([0.0, 1.0, -1.0, 65519.0]*4)
run([0.0, 1.0, 6.1e-5, 5.96e-8]*4)
run([0.0, 1.0, 2.98e-8, 3.1e-8]*4)
PY
cat $O/edge.txt
python -m pytest tests/test_gpu_precision_modes.py -m gpu -q -s > $O/modes.log 2>&1; echo "modes rc $?"
tail -n 12 $O/modes.log | cut -c1-400
for p in bf16 fp16; do
  python bench.py --precision $p --no-cpu-baseline --no-episode --steps 20 > $O/bench_$p.json 2> $O/bench_$p.err; echo "bench $p rc $?"
done
cat $O/bench_*.json | cut -c1-1500
