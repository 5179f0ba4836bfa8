"""-m gpu: each HIP kernel exported through the C ABI vs a plain PyTorch fp32 reference of the same op."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from edgecape_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return _lib.load()


def _chk(lib, rc):
    assert rc == 0, lib.ec_last_error().decode()


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 256, 256), (1000, 384, 640), (325 * 2, 1152, 384), (77, 128, 128)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_fp32(lib, M, N, K, act):
    g = torch.Generator().manual_seed(M + N + K + act)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5   # asymmetric operands: catches transposed C writes
    b = torch.randn(N, generator=g)
    gam = torch.rand(N, generator=g) + 0.5
    R = torch.randn(M, N, generator=g)
    ref = A.double() @ W.double().T + b.double()
    ref = {0: ref, 1: ref.relu(), 2: torch.nn.functional.gelu(ref)}[act]
    ref = (ref * gam.double() + R.double()).float()
    Ad, Wd, bd, gd, Rd = (x.cuda() for x in (A, W, b, gam, R))
    Cd = torch.empty(M, N, device="cuda")
    _chk(lib, lib.ec_op_linear(_p(Ad), _p(Wd), _p(bd), _p(gd), _p(Rd), _p(Cd), M, N, K, act, 0, None))
    torch.cuda.synchronize()
    err = (Cd.cpu() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 256, 256), (3200, 256, 768), (10368, 1024, 256), (650, 1152, 384), (77, 128, 128)])
@pytest.mark.parametrize("act", [0, 2])
def test_linear_bf16x3(lib, M, N, K, act):
    """Split-bf16 (hi+lo) GEMM of the head's throughput mode: fp32-class accuracy (~2^-17 relative per operand)."""
    g = torch.Generator().manual_seed(M + N + K + act)
    A = torch.randn(M, K, generator=g) * 3
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    gam = torch.rand(N, generator=g) + 0.5
    R = torch.randn(M, N, generator=g)
    ref = A.double() @ W.double().T + b.double()
    ref = {0: ref, 2: torch.nn.functional.gelu(ref)}[act]
    ref = (ref * gam.double() + R.double()).float()
    Ad, Wd, bd, gd, Rd = (x.cuda() for x in (A, W, b, gam, R))
    Cd = torch.empty(M, N, device="cuda")
    _chk(lib, lib.ec_op_linear(_p(Ad), _p(Wd), _p(bd), _p(gd), _p(Rd), _p(Cd), M, N, K, act, 2, None))
    torch.cuda.synchronize()
    err = (Cd.cpu() - ref).abs().max().item()
    assert err < 1e-4 * max(1.0, ref.abs().max().item()), err      # bf16 alone would be ~1e-2 here


def test_linear_identity_layout(lib):
    """A = I with an asymmetric W: C must equal W^T exactly (transpose-detecting)."""
    N = K = 128
    A = torch.eye(K).cuda()
    W = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251).cuda()
    Cd = torch.empty(K, N, device="cuda")
    _chk(lib, lib.ec_op_linear(_p(A), _p(W), None, None, None, _p(Cd), K, N, K, 0, 0, None))
    torch.cuda.synchronize()
    assert torch.equal(Cd, W.T)


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (650, 1152, 384), (1000, 768, 3072),
                                   # >= 1024 rows: the 8-phase 256x256x64 kernel (ragged M, N not a multiple of 256, long K)
                                   (1300, 768, 768), (2600, 1152, 384), (4099, 2304, 768), (1024, 256, 3072), (5000, 1536, 128)])
def test_linear_bf16(lib, M, N, K):
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ref = (A.bfloat16().double() @ W.bfloat16().double().T + b.double()).float()  # exact products of bf16 operands
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()
    Cd = torch.empty(M, N, device="cuda")
    _chk(lib, lib.ec_op_linear(_p(Ad), _p(Wd), _p(bd), None, None, _p(Cd), M, N, K, 0, 1, None))
    torch.cuda.synchronize()
    err = (Cd.cpu() - ref).abs().max().item()
    assert err < 2e-4, err   # fp32 accumulation-order noise only


@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_bf16_epilogue(lib, act):
    """bias + activation + LayerScale + residual epilogue of the 8-phase bf16 kernel (fp32 output)."""
    M, N, K = 2500, 768, 256
    g = torch.Generator().manual_seed(7 + act)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    gam = torch.rand(N, generator=g) + 0.5
    R = torch.randn(M, N, generator=g)
    ref = A.bfloat16().double() @ W.bfloat16().double().T + b.double()
    ref = {0: ref, 1: ref.relu(), 2: torch.nn.functional.gelu(ref)}[act]
    ref = (ref * gam.double() + R.double()).float()
    Ad, Wd, bd, gd, Rd = (x.cuda() for x in (A, W, b, gam, R))
    Cd = torch.empty(M, N, device="cuda")
    _chk(lib, lib.ec_op_linear(_p(Ad), _p(Wd), _p(bd), _p(gd), _p(Rd), _p(Cd), M, N, K, act, 1, None))
    torch.cuda.synchronize()
    err = (Cd.cpu() - ref).abs().max().item()
    assert err < 2e-4, err


def test_linear_bf16_identity_layout(lib):
    """A = [I; I; ...] with an asymmetric W through the 8-phase kernel: C rows must equal W^T rows exactly."""
    N, K, reps = 512, 256, 5
    A = torch.eye(K).repeat(reps, 1).cuda()                      # M = 1280
    W = ((torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 127) - 63).cuda()   # exactly representable in bf16
    Cd = torch.empty(reps * K, N, device="cuda")
    _chk(lib, lib.ec_op_linear(_p(A), _p(W), None, None, None, _p(Cd), reps * K, N, K, 0, 1, None))
    torch.cuda.synchronize()
    assert torch.equal(Cd, W.T.repeat(reps, 1))


@pytest.mark.parametrize("rows,cols,eps", [(100, 256, 1e-5), (650, 384, 1e-6), (37, 768, 1e-6), (9, 1024, 1e-6)])
def test_layernorm(lib, rows, cols, eps):
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, cols, generator=g) * 3 + 1
    w, b = torch.randn(cols, generator=g), torch.randn(cols, generator=g)
    ref = torch.nn.functional.layer_norm(x, (cols,), w, b, eps)
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    yd = torch.empty_like(xd)
    _chk(lib, lib.ec_op_layernorm(_p(xd), _p(wd), _p(bd), _p(yd), rows, cols, eps, None))
    torch.cuda.synchronize()
    assert (yd.cpu() - ref).abs().max().item() < 2e-5


def _attn_ref(q, k, v, H, hd, kmask, bias):
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    qh = q.double().reshape(B, Lq, H, hd).transpose(1, 2) * hd ** -0.5
    kh = k.double().reshape(B, Lk, H, hd).transpose(1, 2)
    vh = v.double().reshape(B, Lk, H, hd).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2)
    if bias is not None:
        s = s + bias.double()
    if kmask is not None:
        s = s.masked_fill(kmask.bool()[:, None, None, :], float("-inf"))
    return (s.softmax(-1) @ vh).transpose(1, 2).reshape(B, Lq, H * hd).float()


@pytest.mark.parametrize("B,H,Lq,Lk,hd,masked,biased", [
    (2, 6, 257, 257, 64, False, False),    # ViT-S backbone block
    (2, 8, 356, 356, 32, True, False),     # head encoder (image + keypoint tokens, key padding)
    (3, 8, 100, 100, 32, True, True),      # decoder biased self-attention
    (2, 8, 100, 324, 64, False, False),    # token -> image cross attention
    (2, 8, 324, 100, 64, False, False),    # image -> token (two-way)
    (1, 12, 130, 70, 64, True, False),     # ragged tile edges
])
def test_attention_fp32(lib, B, H, Lq, Lk, hd, masked, biased):
    g = torch.Generator().manual_seed(B * 1000 + Lq + Lk)
    q = torch.randn(B, Lq, H * hd, generator=g)
    k = torch.randn(B, Lk, H * hd, generator=g)
    v = torch.randn(B, Lk, H * hd, generator=g)
    kmask = None
    if masked:
        kmask = (torch.rand(B, Lk, generator=g) < 0.4).to(torch.uint8)
        kmask[:, 0] = 0
        kmask[0, 1:] = 1 if Lk <= 128 else kmask[0, 1:]   # one sample with a single visible key
    bias = torch.randn(B, H, Lq, Lk, generator=g) if biased else None
    ref = _attn_ref(q, k, v, H, hd, kmask, bias)
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    md = kmask.cuda() if masked else None
    bd = bias.cuda() if biased else None
    od = torch.empty(B, Lq, H * hd, device="cuda")
    _chk(lib, lib.ec_op_attention(_p(qd), _p(kd), _p(vd), _p(md), _p(bd), _p(od), B, H, Lq, Lk, hd, 0, None))
    torch.cuda.synchronize()
    err = (od.cpu() - ref).abs().max().item()
    assert err < 2e-5, err


@pytest.mark.parametrize("B,H,Lq,Lk,hd,masked,biased", [
    (2, 8, 100, 100, 32, True, False), (2, 8, 100, 100, 32, True, True), (2, 8, 424, 424, 32, True, False),
    (2, 8, 100, 324, 64, False, False), (2, 8, 324, 100, 64, False, False), (1, 12, 130, 70, 64, True, False),
    (1, 8, 37, 53, 32, False, True), (3, 8, 829, 829, 32, True, False)])
def test_attention_bf16x3(lib, B, H, Lq, Lk, hd, masked, biased):
    """Split-bf16 attention of the head's throughput mode vs fp64 math: fp32-class accuracy."""
    g = torch.Generator().manual_seed(B * 1000 + Lq + Lk + hd)
    q = torch.randn(B, Lq, H * hd, generator=g) * 1.5
    k = torch.randn(B, Lk, H * hd, generator=g) * 1.5
    v = torch.randn(B, Lk, H * hd, generator=g)
    kmask = None
    if masked:
        kmask = (torch.rand(B, Lk, generator=g) < 0.4).to(torch.uint8)
        kmask[:, 0] = 0
    bias = torch.randn(B, H, Lq, Lk, generator=g) if biased else None
    ref = _attn_ref(q, k, v, H, hd, kmask, bias)
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    md = kmask.cuda() if masked else None
    bd = bias.cuda() if biased else None
    od = torch.empty(B, Lq, H * hd, device="cuda")
    _chk(lib, lib.ec_op_attention(_p(qd), _p(kd), _p(vd), _p(md), _p(bd), _p(od), B, H, Lq, Lk, hd, 2, None))
    torch.cuda.synchronize()
    err = (od.cpu() - ref).abs().max().item()
    assert err < 1e-4, err          # plain bf16 operands give ~1e-2 here


@pytest.mark.parametrize("B,H,L", [(2, 6, 257), (2, 12, 325), (1, 16, 730), (1, 6, 64), (2, 6, 300), (2, 6, 96), (1, 6, 33), (1, 6, 161),
                                   (1, 6, 20), (1, 6, 32), (1, 6, 128), (1, 6, 129), (2, 6, 65), (1, 6, 193),
                                   (40, 12, 325), (70, 12, 130), (150, 6, 64)])   # > 768 work items: several per persistent workgroup
def test_attention_bf16(lib, B, H, L):
    """bf16 MFMA attention (backbone shape) vs fp64 math on the bf16-rounded operands."""
    hd = 64
    g = torch.Generator().manual_seed(L)
    q = torch.randn(B, L, H * hd, generator=g)
    k = torch.randn(B, L, H * hd, generator=g)
    v = torch.randn(B, L, H * hd, generator=g)
    r16 = lambda x: x.bfloat16().float()
    ref = _attn_ref(r16(q), r16(k), r16(v), H, hd, None, None)
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    od = torch.empty(B, L, H * hd, device="cuda")
    _chk(lib, lib.ec_op_attention(_p(qd), _p(kd), _p(vd), None, None, _p(od), B, H, L, L, hd, 1, None))
    torch.cuda.synchronize()
    err = (od.cpu() - ref).abs().max().item()
    # P and O are rounded to bf16 (8 mantissa bits): |O| <~ 1 -> a few 1e-3 absolute
    assert err < 2e-2, err
    assert (od.cpu() - ref).abs().mean().item() < 2e-3


@pytest.mark.parametrize("M,N,K,prec", [(20800, 2304, 768, 1), (5000, 768, 3072, 1), (4100, 1152, 384, 1), (4500, 384, 384, 0),
                                        (6000, 512, 256, 0), (300, 256, 256, 1)])
def test_linear_tile_configs(lib, M, N, K, prec):
    """Exercise every tile configuration of the persistent GEMM (256x256, 256x128, 128x128) incl. ragged M edges."""
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g)
    if prec == 1:
        ref = (A.bfloat16().double() @ W.bfloat16().double().T + b.double() + R.double()).float()
        tol = 3e-4
    else:
        ref = (A.double() @ W.double().T + b.double() + R.double()).float()
        tol = 3e-5
    Ad, Wd, bd, Rd = A.cuda(), W.cuda(), b.cuda(), R.cuda()
    Cd = torch.empty(M, N, device="cuda")
    _chk(lib, lib.ec_op_linear(_p(Ad), _p(Wd), _p(bd), None, _p(Rd), _p(Cd), M, N, K, 0, prec, None))
    torch.cuda.synchronize()
    err = (Cd.cpu() - ref).abs().max().item()
    assert err < tol * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("M,N,K", [(20800, 2304, 768), (4099, 768, 3072), (2600, 1152, 384)])
def test_gemm8_race_screen(lib, M, N, K):
    """The 8-phase kernel orders LDS-DMA writes and ds_reads only by counted vmcnt + barriers (ec_gemm8.hip header): an
    early read shows up as rare wrong tiles that depend on timing.  Run the same problem repeatedly, alone and beside a
    bandwidth-heavy kernel on another stream, and require bit-identical results that also match the reference."""
    g = torch.Generator().manual_seed(M)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()
    ref = (Ad.bfloat16().double() @ Wd.bfloat16().double().T + bd.double()).float()
    outs = []
    noise = torch.empty(64 * 1024 * 1024, device="cuda")
    side = torch.cuda.Stream()
    for it in range(12):
        Cd = torch.empty(M, N, device="cuda")
        if it >= 4:                                   # perturb timing: stream 256 MB through HBM concurrently
            with torch.cuda.stream(side):
                noise.add_(1.0)
        _chk(lib, lib.ec_op_linear(_p(Ad), _p(Wd), _p(bd), None, None, _p(Cd), M, N, K, 0, 1, None))
        torch.cuda.synchronize()
        outs.append(Cd)
    assert (outs[0] - ref).abs().max().item() < 3e-4
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("kind", ["bias", "gelu", "scale"])
def test_linear_h16_fp16_saturates(lib, kind):
    """fp16 outputs of the 8-phase kernel saturate at +-65504 (ec_common.h pack4_h / f2h, round 3) instead of overflowing to inf: an
    activation outlier of a real checkpoint (DINOv2 has outlier channels in the MLP hidden layer) must not turn into NaN in the next
    LayerNorm / softmax.  Operands scaled so that a sizeable share of the results lies beyond the fp16 range."""
    M, N, K = 2048, 512, 256
    g = torch.Generator().manual_seed(5)
    A = (torch.randn(M, K, generator=g) * 3000.0)
    W = torch.randn(N, K, generator=g)
    b = torch.randn(N, generator=g)
    gam = (torch.rand(N, generator=g) + 0.5) if kind == "scale" else None
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()
    ref = Ad.half().double() @ Wd.half().double().T + bd.double()
    if kind == "gelu":
        ref = torch.nn.functional.gelu(ref)
    if kind == "scale":
        ref = ref * gam.cuda().double()
    buf = torch.empty(M * N, device="cuda", dtype=torch.float16)
    _chk(lib, lib.ec_op_linear_h16(_p(Ad), _p(Wd), _p(bd), _p(gam.cuda()) if gam is not None else None, _p(buf), M, N, K,
                                   2 if kind == "gelu" else 0, 3, 1, None))
    torch.cuda.synchronize()
    got = buf.view(M, N).double()
    assert torch.isfinite(got).all()
    big = ref.abs() > 65504.0
    assert big.float().mean().item() > 0.002                                   # the case does exercise the overflow
    assert torch.all(got[big].abs() == 65504.0) and torch.all(torch.sign(got[big]) == torch.sign(ref[big]))
    ok = ref.abs() < 60000.0
    assert ((got[ok] - ref[ok]).abs() <= ref[ok].abs() * 2.0 ** -10 + 0.05).all()


@pytest.mark.parametrize("kind", ["bias", "gelu", "scale"])
def test_linear_h16_fp16_nan_in_nan_out(lib, kind):
    """The fp16 saturation keeps NaN (v_med3_f32 alone turned a NaN into -65504 - a NaN activation then left the backbone
    as a finite value and the caller never saw it; ec_common.h sat_h16 adds v * 0 back).  A NaN in one activation row must come out
    as NaN in every column of that row and nowhere else; the saturated values of test_linear_h16_fp16_saturates are unchanged."""
    M, N, K = 2048, 512, 256
    g = torch.Generator().manual_seed(6)
    A = torch.randn(M, K, generator=g)
    A[77, 5] = float("nan")
    A[1500, 200] = float("nan")
    A[300, 7] = float("inf")          #  an infinite activation: one infinite product per output, the accumulator is +-inf
    W = torch.randn(N, K, generator=g)
    b = torch.randn(N, generator=g)
    gam = (torch.rand(N, generator=g) + 0.5) if kind == "scale" else None
    buf = torch.empty(M * N, device="cuda", dtype=torch.float16)
    _chk(lib, lib.ec_op_linear_h16(_p(A.cuda()), _p(W.cuda()), _p(b.cuda()), _p(gam.cuda()) if gam is not None else None, _p(buf), M, N, K,
                                   2 if kind == "gelu" else 0, 3, 1, None))
    torch.cuda.synchronize()
    got = buf.view(M, N)
    nan_rows = torch.isnan(got).all(dim=1).nonzero().flatten().tolist()
    # an infinite activation becomes NaN where the operands are rounded to fp16 (sat_h16: finite -> clamp, NaN / +-inf -> NaN, ec_common.h),
    # so its row is NaN in every output kind; an accumulator that overflows to +-inf inside the GEMM keeps the infinity through the
    # bias / LayerScale epilogues and becomes NaN in the GELU one (gelu_fast8: -|x| * 2^(-inf) = inf * 0)
    assert nan_rows == [77, 300, 1500], nan_rows
    assert int(torch.isnan(got).any(dim=1).sum().item()) == 3
    assert bool(torch.isfinite(got[[0, 299, 301, M - 1]]).all())


@pytest.mark.parametrize("prec", [1, 3])
@pytest.mark.parametrize("M,N,K,kind", [(20800, 2304, 768, "bias"), (20800, 768, 768, "scale"), (4099, 3072, 768, "gelu"), (4099, 768, 3072, "scale"),
                                        (2600, 1152, 384, "bias"), (2600, 384, 1536, "scale"), (1300, 1536, 384, "gelu"), (5840, 1024, 1024, "bias")])
def test_linear_h16_output_kinds(lib, M, N, K, kind, prec):
    """The 16-bit-output epilogues of the 8-phase kernel (qkv / proj: + bias; fc2: + bias, * LayerScale; fc1: GELU(+ bias)) as the
    backbone runs them, at ViT-B / ViT-S / ViT-L shapes with ragged M and N not a multiple of 256: values vs fp64 math of the rounded
    operands (half an ulp of the 16-bit result + accumulation noise), nothing stored past M rows (guard region), and bit-identical
    results over repeated back-to-back launches beside a bandwidth-heavy kernel (the tile seam's counted vmcnt waits)."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    gam = (torch.rand(N, generator=g) + 0.5) if kind == "scale" else None
    h = (lambda t: t.bfloat16()) if prec == 1 else (lambda t: t.half())
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()
    ref = h(Ad).double() @ h(Wd).double().T + bd.double()
    if kind == "gelu":
        ref = torch.nn.functional.gelu(ref)
    if kind == "scale":
        ref = ref * gam.cuda().double()
    dt = torch.bfloat16 if prec == 1 else torch.float16
    guard_rows = 300
    noise = torch.empty(64 * 1024 * 1024, device="cuda")
    side = torch.cuda.Stream()
    outs = []
    for it in range(4):
        buf = torch.full(((M + guard_rows) * N,), 7.0, device="cuda", dtype=dt)
        if it >= 2:
            with torch.cuda.stream(side):
                noise.add_(1.0)
        _chk(lib, lib.ec_op_linear_h16(_p(Ad), _p(Wd), _p(bd), _p(gam.cuda()) if gam is not None else None, _p(buf), M, N, K,
                                       2 if kind == "gelu" else 0, prec, 1 if it == 0 else 6, None))
        torch.cuda.synchronize()
        assert torch.all(buf[M * N:] == 7.0), "stored past the last row"
        outs.append(buf[:M * N].view(M, N))
    got = outs[0].double()
    ulp = 2.0 ** -7 if prec == 1 else 2.0 ** -10
    tol = ref.abs() * (ulp * 0.5 + 2e-6) + 3e-4 + (3e-5 if kind == "gelu" else 0.0)
    bad = ((got - ref).abs() > tol)
    assert not bad.any(), (int(bad.sum()), (got - ref).abs().max().item())
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("batch,M,N,K,transB", [(3, 100, 768, 324, 0), (3, 100, 100, 100, 0), (2, 100, 100, 256, 1), (2, 100, 324, 256, 1),
                                                (2, 17, 384, 256, 0), (2, 17, 17, 256, 1), (2, 17, 17, 17, 0), (1, 5, 256, 256, 1),
                                                (2, 33, 65, 12, 0), (1, 1, 384, 256, 0)])
def test_bgemm(lib, batch, M, N, K, transB):
    """Small batched fp32 contractions of the head: MFMA kernel (4-float aligned rows) and the VALU fallback (odd K)."""
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(batch, M, K, generator=g)
    B = torch.randn(batch, N, K, generator=g) if transB else torch.randn(batch, K, N, generator=g)
    ref = (A.double() @ (B.double().transpose(1, 2) if transB else B.double())).float()
    Ad, Bd = A.cuda(), B.cuda()
    Cd = torch.full((batch, M, N), float("nan"), device="cuda")
    _chk(lib, lib.ec_op_bgemm(_p(Ad), _p(Bd), _p(Cd), batch, M, N, K, transB, None))
    torch.cuda.synchronize()
    err = (Cd.cpu() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("M,N,K", [(3200, 256, 256), (10368, 1024, 512), (650, 384, 768), (3200, 768, 256), (100, 512, 128)])
@pytest.mark.parametrize("act", [0, 1])
def test_linear_fp16x1(lib, M, N, K, act):
    """Single-pass fp16 form of the head GEMM (GM_SPLIT1: fp32 A rounded to fp16 in registers, fp16 weights, fp32 accumulate) vs exact
    products of the fp16-rounded operands."""
    g = torch.Generator().manual_seed(11 + M)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g)
    ref = A.half().double() @ W.half().double().T + b.double()
    ref = ((ref.relu() if act else ref) + R.double()).float()
    Ad, Wd, bd, Rd = (x.cuda() for x in (A, W, b, R))
    Cd = torch.empty(M, N, device="cuda")
    _chk(lib, lib.ec_op_linear(_p(Ad), _p(Wd), _p(bd), None, _p(Rd), _p(Cd), M, N, K, act, 4, None))
    torch.cuda.synchronize()
    err = (Cd.cpu() - ref).abs().max().item()
    assert err < 2e-4, err


# ---- IEEE fp16 operand format (backbone_precision = "fp16": 11 significand bits at the bf16 MFMA rate) --------------------
@pytest.mark.parametrize("M,N,K", [(650, 1152, 384), (1300, 768, 768), (4099, 2304, 768), (1024, 256, 3072), (5000, 1536, 128)])
def test_linear_fp16(lib, M, N, K):
    """fp16-operand GEMM (v_mfma_f32_16x16x32_f16 in the 8-phase kernel, 32x32x16 in the small-shape kernels) vs exact
    products of the fp16-rounded operands."""
    g = torch.Generator().manual_seed(2)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ref = (A.half().double() @ W.half().double().T + b.double()).float()
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()
    Cd = torch.empty(M, N, device="cuda")
    _chk(lib, lib.ec_op_linear(_p(Ad), _p(Wd), _p(bd), None, None, _p(Cd), M, N, K, 0, 3, None))
    torch.cuda.synchronize()
    err = (Cd.cpu() - ref).abs().max().item()
    assert err < 2e-4, err   # fp32 accumulation-order noise only


@pytest.mark.parametrize("act", [0, 2])
def test_linear_fp16_epilogue(lib, act):
    M, N, K = 2500, 768, 256
    g = torch.Generator().manual_seed(17 + act)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    gam = torch.rand(N, generator=g) + 0.5
    R = torch.randn(M, N, generator=g)
    ref = A.half().double() @ W.half().double().T + b.double()
    ref = {0: ref, 2: torch.nn.functional.gelu(ref)}[act]
    ref = (ref * gam.double() + R.double()).float()
    Ad, Wd, bd, gd, Rd = (x.cuda() for x in (A, W, b, gam, R))
    Cd = torch.empty(M, N, device="cuda")
    _chk(lib, lib.ec_op_linear(_p(Ad), _p(Wd), _p(bd), _p(gd), _p(Rd), _p(Cd), M, N, K, act, 3, None))
    torch.cuda.synchronize()
    err = (Cd.cpu() - ref).abs().max().item()
    assert err < 2e-4, err


def test_linear_fp16_conversion_edge_values(lib):
    """Device conversion (v_cvt_pk_f16_f32, RNE) on values around the fp16 subnormal / overflow boundaries, through a GEMM with
    an identity weight: C = fp16(A) exactly."""
    K = 128
    vals = torch.tensor([0.0, 1.0, -1.0, 65504.0, 65519.0, 6.1e-5, 5.96e-8, 2.98e-8, 3.1e-8, 1.0 + 2 ** -11, 1.0 + 3 * 2 ** -11,
                         0.1, -0.3333333, 1e-3, 123.456, 2049.0])
    A = vals.repeat(1024 * K // vals.numel()).reshape(1024, K).contiguous()
    W = torch.eye(K).repeat(2, 1)                                       # N = 256 rows: [I; I]
    Cd = torch.empty(1024, 256, device="cuda")
    Ad, Wd = A.cuda(), W.cuda()                                          # keep the device tensors alive across the call
    _chk(lib, lib.ec_op_linear(_p(Ad), _p(Wd), None, None, None, _p(Cd), 1024, 256, K, 0, 3, None))
    torch.cuda.synchronize()
    assert torch.equal(Cd.cpu()[:, :K], A.half().float())
    assert torch.equal(Cd.cpu()[:, K:], A.half().float())


@pytest.mark.parametrize("B,H,L", [(2, 12, 325), (1, 16, 730), (2, 6, 257), (1, 6, 33)])
def test_attention_fp16(lib, B, H, L):
    """fp16 MFMA attention (backbone shape) vs fp64 math on the fp16-rounded operands: ~8x tighter than the bf16 form."""
    hd = 64
    g = torch.Generator().manual_seed(L + 1)
    q = torch.randn(B, L, H * hd, generator=g)
    k = torch.randn(B, L, H * hd, generator=g)
    v = torch.randn(B, L, H * hd, generator=g)
    r16 = lambda x: x.half().float()
    ref = _attn_ref(r16(q), r16(k), r16(v), H, hd, None, None)
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    od = torch.empty(B, L, H * hd, device="cuda")
    _chk(lib, lib.ec_op_attention(_p(qd), _p(kd), _p(vd), None, None, _p(od), B, H, L, L, hd, 3, None))
    torch.cuda.synchronize()
    err = (od.cpu() - ref).abs().max().item()
    assert err < 3e-3, err            # P and O rounded to fp16 (11 significand bits)
    assert (od.cpu() - ref).abs().mean().item() < 3e-4


def test_attention_fp16_large_logits(lib):
    """Scores far apart (|q.k| up to ~60 after scaling): the lazy running maximum keeps P <= 2^8, inside the fp16 range, and the
    rescale branch is exercised by a key row that dominates only in a late tile."""
    B, H, L, hd = 1, 6, 325, 64
    g = torch.Generator().manual_seed(9)
    q = torch.randn(B, L, H * hd, generator=g) * 2.0
    k = torch.randn(B, L, H * hd, generator=g) * 2.0
    v = torch.randn(B, L, H * hd, generator=g)
    k[:, 300] = q[:, 5] * 1.5             # key 300 (last tile) dominates query 5 -> forces a late rescale
    r16 = lambda x: x.half().float()
    ref = _attn_ref(r16(q), r16(k), r16(v), H, hd, None, None)
    od = torch.empty(B, L, H * hd, device="cuda")
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    _chk(lib, lib.ec_op_attention(_p(qd), _p(kd), _p(vd), None, None, _p(od), B, H, L, L, hd, 3, None))
    torch.cuda.synchronize()
    assert torch.isfinite(od).all()
    assert (od.cpu() - ref).abs().max().item() < 5e-3


@pytest.mark.parametrize("h1", [0, 1])
@pytest.mark.parametrize("alias", [True, False])
@pytest.mark.parametrize("rows,K1,Kcat,N2,act2,period,third", [
    (3200, 256, 256, 512, 0, 0, False),     # decoder: out_proj + norm1 -> q_proj([x | qpe])
    (3200, 512, 0, 768, 0, 0, False),       # decoder: fold + norm2 -> ffn1
    (3200, 768, 0, 768, 0, 0, False),       # skeleton: ffn2 + norm3 -> next in_proj
    (1000, 256, 0, 384, 1, 0, True),        # encoder: out_proj + norm1 -> linear1 + ReLU -> linear2 + norm2
    (77, 384, 128, 1536, 2, 0, False),      # ragged slab, GELU, wide second stage (six passes)
    (648, 512, 0, 1024, 0, 324, False),     # image lane: fold + norm4 -> K|V projection + positional table
    (31, 256, 0, 256, 0, 0, True),          # fewer rows than one slab
])
def test_row_chain(lib, rows, K1, Kcat, N2, act2, period, third, alias, h1):
    """Row-chain kernel (ec_chain.hip; bf16x3, or h1: the single-pass fp16 form of EC_MIXED) vs fp64 math of the same residual blocks.  alias: the residual is updated in place
    (one workgroup per slab); otherwise it is a separate buffer and two workgroups per slab share the work (ChainP::split)."""
    g = torch.Generator().manual_seed(rows + K1 + N2)
    X = torch.randn(rows, K1, generator=g)
    W1 = torch.randn(256, K1, generator=g) / K1 ** 0.5
    b1 = torch.randn(256, generator=g)
    R = torch.randn(rows, 256, generator=g)
    g1, be1 = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g)
    cat = torch.randn(rows, Kcat, generator=g) if Kcat else None
    K2 = 256 + Kcat
    W2 = torch.randn(N2, K2, generator=g) / K2 ** 0.5
    b2 = torch.randn(N2, generator=g)
    table = torch.randn(period, N2, generator=g) if period else None
    W3 = torch.randn(256, N2, generator=g) / N2 ** 0.5 if third else None
    b3 = torch.randn(256, generator=g) if third else None
    g3, be3 = (torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g)) if third else (None, None)

    ln = lambda v, w, b: torch.nn.functional.layer_norm(v, (256,), w.double(), b.double(), 1e-5)
    rq = (lambda t: t.half().double()) if h1 else (lambda t: t.double())   # single-pass fp16: every MFMA operand is rounded to fp16
    x1 = ln(R.double() + rq(X) @ rq(W1).T + b1.double(), g1, be1)
    x2in = torch.cat([rq(x1.float()), rq(cat)], 1) if Kcat else rq(x1.float())
    o2 = x2in @ rq(W2).T + b2.double()
    if period:
        o2 = o2 + table.double()[torch.arange(rows) % period]
    o2 = {0: o2, 1: o2.relu(), 2: torch.nn.functional.gelu(o2)}[act2]
    x3 = ln(x1 + rq(o2.float()) @ rq(W3).T + b3.double(), g3, be3) if third else None

    dev = lambda t: t.cuda() if t is not None else None
    Xd, W1d, b1d, Rd, g1d, be1d, catd, W2d, b2d, td, W3d, b3d, g3d, be3d = map(dev, (X, W1, b1, R, g1, be1, cat, W2, b2, table, W3, b3, g3, be3))
    x1d = Rd if alias else torch.full((rows, 256), float("nan"), device="cuda")
    Rd = x1d if alias else Rd
    o2d = torch.full((rows, N2), float("nan"), device="cuda")
    x3d = torch.full((rows, 256), float("nan"), device="cuda") if third else None
    _chk(lib, lib.ec_op_chain(_p(Xd), K1, _p(W1d), _p(b1d), _p(Rd), _p(g1d), _p(be1d), _p(x1d), _p(catd), Kcat, _p(W2d), _p(b2d), N2,
                              act2, _p(td), period, _p(o2d), _p(W3d), _p(b3d), _p(g3d), _p(be3d), _p(x3d), rows, 4 if h1 else 2, None))
    torch.cuda.synchronize()
    e1 = (x1d.cpu().double() - x1).abs().max().item()
    e2 = (o2d.cpu().double() - o2).abs().max().item()
    # bf16x3: ~2^-17 relative per operand, |values| of a few units.  h1 (reference with the same fp16-rounded operands): an intermediate
    # next to an fp16 rounding boundary may round the other way in the kernel - one fp16 ulp of one operand of a later stage
    tol = 6e-4 if h1 else 1e-4
    assert e1 < 5e-5 and e2 < tol, (e1, e2)
    if third:
        e3 = (x3d.cpu().double() - x3).abs().max().item()
        assert e3 < tol, e3


# ---- fp16x2 operands (EC_F16X2 backbone, round 6): a_hi W_hi in fp16 MFMAs + both correction terms in ONE block-scaled FP8 pass --------
def _x2_emulate(A, W):
    """The fp16x2 product as ec_common.h split4_x2 / ec_model.hip pack_x2_weights define it, in fp64 on the rounded operands:
    activations fp16 + e5m2((a - hi) * 2^11) + e5m2(a) with fixed scales, weights fp16 + e4m3 planes with one power-of-two scale each."""
    q5 = lambda t: t.clamp(-57344.0, 57344.0).to(torch.float8_e5m2).double()
    q4 = lambda t: t.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).double()
    sat = lambda t: t.clamp(-65504.0, 65504.0)
    ah = sat(A).half().float()
    al8 = q5((A - ah) * 2048.0) / 2048.0
    ah8 = q5(A)
    wh = W.half().float()
    wl = W - wh
    planes = []
    for t in (W, wl):
        amax = float(t.abs().max())
        s = int(np.floor(np.log2(448.0 / amax))) if amax > 0 else 0
        planes.append(q4(t * 2.0 ** s) * 2.0 ** -s)
    return ah.double() @ wh.double().T + al8 @ planes[0].T + ah8 @ planes[1].T


@pytest.mark.parametrize("M,N,K,kind", [(20800, 2304, 768, "bias"), (20800, 768, 768, "res"), (4099, 3072, 768, "gelu"), (4099, 768, 3072, "res"),
                                        (2600, 1152, 384, "bias"), (2600, 384, 1536, "res"), (1300, 1536, 384, "gelu"), (5840, 1024, 1024, "bias"),
                                        (325, 1152, 384, "bias"), (77, 1536, 384, "gelu"), (1, 384, 1536, "res")])
def test_linear_x2_output_kinds(lib, M, N, K, kind):
    """The block GEMMs of the EC_F16X2 backbone on the 8-phase kernel (ec_gemm8.hip, X2 instantiations) at ViT-S / -B / -L shapes with
    ragged and tiny M (this mode has no other GEMM: one image = 325 rows, one row): (a) against the fp64 emulation of the SCHEME on
    the rounded operands - only fp32 accumulation noise is allowed, so a wrong K mapping of the FP8 fragments, a wrong scale byte or a
    wrong plane order fails by orders of magnitude; (b) the scheme itself against the exact product: ~2^-14 relative to the row's
    |a| . |w| (two MFMA units per product; bf16x3 spends three for ~2^-17); nothing stored past M rows; repeated launches bit-identical."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))      # rows of different magnitude
    A[:, 3] *= 60.0                                                                      # an outlier channel
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    gam = (torch.rand(N, generator=g) + 0.5) if kind == "res" else None
    R = torch.randn(M, N, generator=g) if kind == "res" else None
    emu = _x2_emulate(A, W) + b.double()
    exact = A.double() @ W.double().T + b.double()
    scale = (A.double().abs() @ W.double().abs().T)                                      # |a| . |w| per output: what the rounding errors scale with
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()
    guard = 40
    outs = []
    for it in range(3):
        if kind == "gelu":
            buf = torch.full(((M + guard) * 4 * N,), 0x55, device="cuda", dtype=torch.uint8)
            _chk(lib, lib.ec_op_linear_x2(_p(Ad), _p(Wd), _p(bd), None, None, _p(buf), M, N, K, 2, 1 if it == 0 else 4, None, None))
        else:
            buf = torch.full(((M + guard) * N,), 7.0, device="cuda")
            if kind == "res":
                buf[:M * N] = R.cuda().flatten()
                reps = 1                                                                  # (in place: every launch adds again)
            else:
                reps = 1 if it == 0 else 4
            _chk(lib, lib.ec_op_linear_x2(_p(Ad), _p(Wd), _p(bd), _p(gam.cuda()) if gam is not None else None, _p(buf), None, M, N, K, 0, reps, None, None))
        torch.cuda.synchronize()
        if kind == "gelu":
            assert torch.all(buf[M * 4 * N:] == 0x55), "stored past the last row"
            outs.append(buf[:M * 4 * N].view(M, 4 * N).cpu())
        else:
            assert torch.all(buf[M * N:] == 7.0), "stored past the last row"
            outs.append(buf[:M * N].view(M, N).cpu())
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    if kind == "gelu":
        rows = outs[0]
        hi = rows[:, :2 * N].contiguous().view(torch.float16).double()
        lo8 = rows[:, 2 * N:3 * N].contiguous().view(torch.float8_e5m2).double() / 2048.0
        hi8 = rows[:, 3 * N:].contiguous().view(torch.float8_e5m2).double()
        ref = torch.nn.functional.gelu(emu)
        tol = scale * 3e-6 + 2e-6 + ref.abs() * 2.0 ** -13                                # accumulation + GELU form (6.4e-7) + the e5m2 rounding of lo (2^-11 * 2^-3)
        bad = (hi + lo8 - ref).abs() > tol
        assert not bad.any(), (int(bad.sum()), float((hi + lo8 - ref).abs().max()))
        assert ((hi - ref).abs() <= ref.abs() * 2.0 ** -11 + tol).all()                   # the fp16 plane alone: half an ulp
        assert ((hi8 - ref).abs() <= ref.abs() * 2.0 ** -3 + tol + 2.0 ** -17).all()      # the e5m2 plane: half an ulp of 2 mantissa bits
        return
    got = outs[0].double()
    if kind == "res":
        emu, exact = emu * gam.double() + R.double(), exact * gam.double() + R.double()
        scale = scale * gam.double()
    err_emu = (got - emu).abs()
    assert (err_emu <= scale * 3e-6 + 2e-6).all(), float((err_emu / (scale * 3e-6 + 2e-6)).max())
    rel = ((got - exact).abs() / scale)
    print(f"fp16x2 {M}x{N}x{K} {kind}: |got - exact| / (|a|.|w|) max {float(rel.max()):.2e} rms {float((rel ** 2).mean().sqrt()):.2e}; vs emulation max {float((err_emu / scale).max()):.2e}")
    assert float(rel.max()) < 2.0 ** -13 and float((rel ** 2).mean().sqrt()) < 2.0 ** -16


def test_linear_x2_nan_inf_and_saturation(lib):
    """fp16x2 operands keep the fp16 mode's contract for non-finite and out-of-range activations: a NaN activation makes its output row
    NaN (the fp16 plane carries it; the e5m2 planes are clamped finite), an infinite one makes it non-finite, a finite activation beyond
    the fp16 range saturates at +-65504 in the fp16 plane (and at +-57344 / +-28 in the e5m2 planes) instead of becoming inf - every
    other row is untouched."""
    M, N, K = 1300, 384, 384
    g = torch.Generator().manual_seed(11)
    A = torch.randn(M, K, generator=g)
    A[7, 5] = float("nan")
    A[300, 9] = float("inf")
    A[900, 3] = 1.0e6                                             # saturates: the row stays finite
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    C_ = torch.empty(M, N, device="cuda")
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()                     # (kept alive: a temporary's block goes back to the caching allocator at once)
    _chk(lib, lib.ec_op_linear_x2(_p(Ad), _p(Wd), _p(bd), None, _p(C_), None, M, N, K, 0, 1, None, None))
    torch.cuda.synchronize()
    got = C_.cpu()
    assert torch.isnan(got[7]).all()
    assert not torch.isfinite(got[300]).any()
    assert torch.isfinite(got[900]).all()
    As = A.clone(); As[900, 3] = 65504.0
    ok = torch.ones(M, dtype=torch.bool); ok[[7, 300]] = False
    ref = (As[ok].double() @ W.double().T + b.double())
    err = (got[ok].double() - ref).abs()
    scale = As[ok].double().abs() @ W.double().abs().T
    assert (err <= scale * 2.0 ** -9).all()                       # (the saturated value's e5m2 copies are clamped: its correction terms are coarse)
    assert torch.isfinite(got[ok]).all()
