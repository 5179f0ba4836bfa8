"""-m gpu: the throughput precisions of the hot path, gated against the CPU oracle with explicit error-distribution gates.

north_star tolerance: keypoint coordinates within 1e-3 abs of the reference fp32 forward, PCK@0.2 within +-0.1.  The path has
one hard discontinuity, the proposal generator's argmax over the similarity map (encoder_decoder.py:91-110): when two cells of a
keypoint's similarity map are within rounding distance, ANY perturbation moves the proposal by a grid cell, and through the
decoder's self-attention / GCN every keypoint of that sample follows.  So every mode is gated on
  (i)   argmax flips among valid keypoints (count, as a fraction of the valid keypoints),
  (ii)  max |d output_kpts| over the samples WITHOUT a flip  (the continuous part of the error),
  (iii) quantiles of |d| over everything, and the fraction above 1e-3,
  (iv)  PCK@0.2 of the HIP predictions against the ORACLE's predictions as ground truth (1.0 = same answers; the synthetic
        weights are random, so PCK against the synthetic GT is chance level and says nothing).
Modes (backbone / head):
  bf16x3 / bf16x3  "parity mode": every MFMA operand split hi+lo bf16 - fp32-class; must meet 1e-3 outright, no flips.
  fp16   / bf16x3  headline throughput mode: IEEE fp16 operands at the bf16 MFMA rate; continuous error ~1e-4.
  bf16   / bf16x3  the north-star's literal bf16 tiles: continuous error ~1e-3, reported and bounded, not parity-grade.
CPU emulation of the operand rounding on 256 pairs (oracle/precision_study.py) predicts: bf16 95 flips, fp16 11, bf16x3 0.
"""
import functools

import numpy as np
import pytest
import torch

from edgecape_amd import synth

pytestmark = pytest.mark.gpu

CFG = {
    "cfg2": dict(arch="dinov2_vitb14", H=256, bs=32, S=1, wseed=0, iseed=1000),    # BASELINE configs[1]
    "cfg4": dict(arch="dinov2_vitb14", H=256, bs=16, S=5, wseed=0, iseed=2000),    # configs[3]
    "cfg5": dict(arch="dinov2_vitl14", H=384, bs=8, S=1, wseed=0, iseed=3000),     # configs[4]
}


@functools.lru_cache(maxsize=None)
def _weights(arch, seed):
    return synth.make_weights(arch, seed=seed)


@functools.lru_cache(maxsize=None)
def _case(name):
    """(batch, mask, oracle outputs) of a BASELINE config at its FULL batch size (the CPU oracle needs seconds per config)."""
    from oracle import edgecape_oracle as orc   # the checker
    c = CFG[name]
    batch = synth.make_pairs(c["bs"], c["S"], c["H"], seed=c["iseed"], fixed_n_kp=False)
    mask = batch["target_weight_s"][0].copy()
    for tw in batch["target_weight_s"]:
        mask = mask * tw
    torch.set_num_threads(min(32, torch.get_num_threads()))
    _, out = orc.forward_test(_weights(c["arch"], c["wseed"]), batch, synth.ARCHS[c["arch"]]["heads"])
    ref = {k: out[k].numpy() for k in ("output_kpts", "similarity_map", "adj")}
    return batch, mask, ref


def _run(name, backbone, head):
    from edgecape_amd.engine import HipEngine
    c = CFG[name]
    batch, mask, ref = _case(name)
    eng = HipEngine(_weights(c["arch"], c["wseed"]), arch=c["arch"], image_size=c["H"], max_batch=c["bs"], max_shots=c["S"],
                    backbone_precision=backbone, head_precision=head)
    o = eng.forward(batch["img_q"], batch["img_s"], batch["target_s"], mask, [m["sample_skeleton"][0] for m in batch["img_metas"]])
    torch.cuda.synchronize()
    got = {k: o[k].cpu().numpy() for k in ("output_kpts", "similarity_map", "adj")}
    del eng
    return stats(got, ref, mask[:, :, 0] > 0, c["H"])


def stats(got, ref, valid, H):
    bs = valid.shape[0]
    d = np.abs(got["output_kpts"] - ref["output_kpts"])                          # [layers, bs, K, 2]
    am_g = got["similarity_map"].reshape(bs, valid.shape[1], -1).argmax(-1)
    am_r = ref["similarity_map"].reshape(bs, valid.shape[1], -1).argmax(-1)
    flip = (am_g != am_r) & valid
    clean = ~flip.any(1)                                                         # samples without any flipped valid keypoint
    dv = d[:, valid]
    dclean = d[:, clean][:, valid[clean]] if clean.any() else np.zeros(1)
    dist = np.linalg.norm((got["output_kpts"][-1] - ref["output_kpts"][-1]), axis=-1)   # normalised units, thr 0.2 of the bbox side
    pck_vs_oracle = float((dist[valid] < 0.2).mean())
    return dict(n_valid=int(valid.sum()), flips=int(flip.sum()), flip_frac=float(flip.sum() / max(valid.sum(), 1)),
                clean_samples=int(clean.sum()), max_clean=float(dclean.max()), max_all=float(dv.max()),
                median=float(np.median(dv)), p99=float(np.quantile(dv, 0.99)), frac_gt_1e3=float((dv > 1e-3).mean()),
                pck_vs_oracle=pck_vs_oracle, adj_err=float(np.abs(got["adj"] - ref["adj"]).max()))


@pytest.mark.parametrize("name", ["cfg2", "cfg4", "cfg5"])
def test_parity_mode_bf16x3(name):
    """bf16x3 backbone + bf16x3 head: the tolerance-conforming fast mode.  Full batch of the BASELINE config vs the oracle."""
    s = _run(name, "bf16x3", "bf16x3")
    print(name, "bf16x3/bf16x3", s)
    assert s["flips"] == 0
    assert s["max_all"] < 1e-3, s
    assert s["p99"] < 1e-4 and s["adj_err"] < 1e-4
    assert s["pck_vs_oracle"] == 1.0


@pytest.mark.parametrize("name", ["cfg2", "cfg4", "cfg5"])
def test_headline_mode_fp16(name):
    """fp16 backbone + bf16x3 head (bench.py's headline precision).  Continuous error an order of magnitude inside the gate;
    argmax near-ties may flip (emulation: 0.15 % of the valid keypoints with random weights)."""
    s = _run(name, "fp16", "bf16x3")
    print(name, "fp16/bf16x3", s)
    assert s["max_clean"] < 1e-3, s                       # every sample without an argmax flip is inside the north-star tolerance
    assert s["p99"] < 5e-4 and s["median"] < 2e-5
    assert s["flip_frac"] <= 0.01
    assert s["clean_samples"] >= 0.75 * CFG[name]["bs"]
    assert s["pck_vs_oracle"] >= 0.98                     # north star: PCK@0.2 within +-0.1 (here against the oracle's own answers)


@pytest.mark.parametrize("name", ["cfg2", "cfg4", "cfg5"])
def test_headline_mode_fp16_mixed_head(name):
    """fp16 backbone + MIXED head (bench.py's headline precision, round 2): bf16x3 wherever the proposal generator's argmax depends on
    it, single-pass fp16 MFMAs in the Linear layers of the skeleton head and the decoder layers (EC_MIXED).  Same gates as the
    fp16 / bf16x3 mode - the head's share of the error is below the backbone's (oracle/head_precision_study.py) - plus: no more argmax
    flips than that mode, and the refined adjacency (the skeleton head's product) still at 1e-3."""
    s = _run(name, "fp16", "mixed")
    s0 = _run(name, "fp16", "bf16x3")
    print(name, "fp16/mixed", s, "\n     fp16/bf16x3", s0)
    assert s["max_clean"] < 1e-3, s
    assert s["p99"] < 5e-4 and s["median"] < 2e-5
    assert s["flips"] <= s0["flips"]                      # the encoder / proposal path is bit-identical to the bf16x3 head's
    assert s["clean_samples"] >= 0.75 * CFG[name]["bs"]
    assert s["pck_vs_oracle"] >= 0.98
    assert s["adj_err"] < 1e-3


def test_bf16_mode_cfg2_bounded():
    """bf16 backbone + bf16x3 head (the north-star's literal bf16 MFMA tiles): NOT parity-grade - 8 significand bits put the
    continuous error at the 1e-3 gate and flip ~1.3 % of the argmaxes with random weights.  Bounded here so a kernel bug cannot
    hide behind 'bf16 is inexact': distribution gates an order of magnitude tighter than any indexing / synchronisation bug."""
    s = _run("cfg2", "bf16", "bf16x3")
    print("cfg2 bf16/bf16x3", s)
    assert s["median"] < 1e-4 and s["p99"] < 0.5
    assert s["flip_frac"] <= 0.05
    assert s["max_clean"] < 2e-2
    assert s["pck_vs_oracle"] >= 0.9                      # north star: PCK within +-0.1
