import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ctypes as C, torch
from edgecape_amd import _lib
lib = _lib.load()
M, N, K = 2048, 512, 256
g = torch.Generator().manual_seed(6)
A = torch.randn(M, K, generator=g); A[77, 5] = float("nan"); A[1500, 200] = float("nan")
W = torch.randn(N, K, generator=g); b = torch.randn(N, generator=g)
for kind, act in (("bias", 0), ("gelu", 2)):
    buf = torch.empty(M * N, device="cuda", dtype=torch.float16)
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()
    rc = lib.ec_op_linear_h16(C.c_void_p(Ad.data_ptr()), C.c_void_p(Wd.data_ptr()), C.c_void_p(bd.data_ptr()), None, C.c_void_p(buf.data_ptr()), M, N, K, act, 3, 1, None)
    torch.cuda.synchronize()
    got = buf.view(M, N)
    nn = torch.isnan(got).sum(dim=1)
    print(kind, "rc", rc, "rows with NaN:", nn.nonzero().flatten().tolist()[:10], "counts", nn[nn > 0].tolist()[:10])
    print("  row 77 first 8:", got[77, :8].tolist(), " row 76:", got[76, :4].tolist())
    x = got[77].float()
    print("  row 77: nan", int(torch.isnan(x).sum()), "== 65504:", int((x.abs() == 65504).sum()), "inf", int(torch.isinf(x).sum()))
