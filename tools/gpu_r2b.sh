set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b; mkdir -p $O
python -m pytest tests -m gpu -q > $O/all.log 2>&1; echo "all rc $?"
tail -n 15 $O/all.log
python - <<'PY' > $O/edge.txt 2>&1
import torch, ctypes as C
from edgecape_amd import _lib
lib=_lib.load()
K=128
vals = torch.tensor([0.0, 1.0, -1.0, 65504.0, 65519.0, 6.1e-5, 5.96e-8, 2.98e-8, 3.1e-8, 1.0 + 2 ** -11, 1.0 + 3 * 2 ** -11, 0.1, -0.3333333, 1e-3, 123.456, 2049.0])
A = vals.repeat(1024 * K // vals.numel()).reshape(1024, K).contiguous()
W = torch.eye(K).repeat(2, 1)
Cd = torch.empty(1024, 256, device="cuda"); Ad, Wd = A.cuda(), W.cuda()
rc = lib.ec_op_linear(Ad.data_ptr(), Wd.data_ptr(), None, None, None, Cd.data_ptr(), 1024, 256, K, 0, 3, None)
torch.cuda.synchronize()
print(rc); print(Cd.cpu()[0,:16].tolist()); print(A.half().float()[0,:16].tolist())
PY
cat $O/edge.txt
for p in bf16 fp16; do
  python bench.py --precision $p --no-cpu-baseline --no-episode --steps 20 > $O/bench_$p.json 2> $O/bench_$p.err; echo "bench $p rc $?"
done
cat $O/bench_*.json
python tools/gemm_bench.py bf16 2>&1 | tee $O/gemm.txt
