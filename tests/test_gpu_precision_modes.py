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
CPU emulation of the operand rounding on 256 pairs (oracle/precision_study.py, ViT-S) predicted: bf16 95 flips, fp16 11, bf16x3 0.
MEASURED on MI355X at cfg2, 256 DISJOINT pairs x 2 weight seeds (round 3, test_headline_conformance_at_scale, record
profiles/r03_conformance_fp16_mixed.json): fp16 backbone + mixed head 13 flips of 20 293 valid keypoints (6.4e-4), max |d| 1.65e-4 on the
flip-free samples, 5e-4 of the keypoints (flipped samples) above 1e-3, PCK@0.2 vs the oracle 0.9997; bf16x3 / bf16x3: 1 flip, max
1.1e-5 (profiles/r03_conformance_bf16x3.json).  (An earlier record with 0 flips had drawn 39 distinct pairs per weight seed.)
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
    # observed on all three configs: 0 flips, max |d| 1.4-1.6e-4 (round 2 and 3); at scale: test_headline_conformance_at_scale
    assert s["max_clean"] < 5e-4, s                       # every sample without an argmax flip: 2x inside the north-star tolerance
    assert s["p99"] < 2e-4 and s["median"] < 2e-5
    assert s["flips"] <= 1, s                             # one near-tie may flip on another box; it moves ONE sample
    assert s["clean_samples"] >= CFG[name]["bs"] - 1
    if s["flips"] == 0:
        assert s["max_all"] < 5e-4 and s["frac_gt_1e3"] == 0.0
    assert s["pck_vs_oracle"] >= 0.99                     # north star: PCK@0.2 within +-0.1 (here against the oracle's own answers)


@pytest.mark.parametrize("name", ["cfg2", "cfg4", "cfg5"])
def test_headline_mode_fp16_mixed_head(name):
    """fp16 backbone + MIXED head (bench.py's headline precision, round 2): bf16x3 wherever the proposal generator's argmax depends on
    it, single-pass fp16 MFMAs in the Linear layers of the skeleton head and the decoder layers (EC_MIXED).  Same gates as the
    fp16 / bf16x3 mode - the head's share of the error is below the backbone's (oracle/head_precision_study.py) - plus: no more argmax
    flips than that mode, and the refined adjacency (the skeleton head's product) still at 1e-3."""
    s = _run(name, "fp16", "mixed")
    s0 = _run(name, "fp16", "bf16x3")
    print(name, "fp16/mixed", s, "\n     fp16/bf16x3", s0)
    assert s["max_clean"] < 5e-4, s
    assert s["p99"] < 2e-4 and s["median"] < 2e-5
    assert s["flips"] <= s0["flips"] and s["flips"] <= 1  # the encoder / proposal path is bit-identical to the bf16x3 head's
    assert s["clean_samples"] >= CFG[name]["bs"] - 1
    if s["flips"] == 0:
        assert s["max_all"] < 5e-4 and s["frac_gt_1e3"] == 0.0
    assert s["pck_vs_oracle"] >= 0.99
    assert s["adj_err"] < 1e-4


def conformance_at_scale(n_batches=8, wseeds=(0, 1), backbone="fp16", head="mixed", name="cfg2"):
    """The headline precision against the oracle on n_batches x bs pairs per weight seed (cfg2: 8 x 32 = 256 pairs, 2 seeds): the
    MEASURED rate of argmax flips and of keypoints outside 1e-3, instead of one lucky 32-pair sample.  Returns one stats dict per
    weight seed plus the pooled one.  (Also run by tools/conformance.py, which writes the record under profiles/.)"""
    from edgecape_amd.engine import HipEngine
    from oracle import edgecape_oracle as orc   # the checker
    c = CFG[name]
    torch.set_num_threads(min(32, torch.get_num_threads()))
    per_seed, pooled_got, pooled_ref, pooled_valid = [], [], [], []
    for ws in wseeds:
        w = synth.make_weights(c["arch"], seed=ws)
        eng = HipEngine(w, arch=c["arch"], image_size=c["H"], max_batch=c["bs"], max_shots=c["S"], backbone_precision=backbone, head_precision=head)
        gots, refs, valids = [], [], []
        for b in range(n_batches):
            # DISJOINT pairs: pair i of synth.make_pairs is seeded by seed + first_index + i, so batch b takes the indices b*bs .. b*bs+bs-1
            # of the weight seed's own range (the first round-3 record used seed + b and so saw the same 39 pairs eight times over)
            batch = synth.make_pairs(c["bs"], c["S"], c["H"], seed=c["iseed"] + 100000 * (1 + ws), first_index=b * c["bs"], fixed_n_kp=False)
            mask = batch["target_weight_s"][0].copy()
            for tw in batch["target_weight_s"]:
                mask = mask * tw
            _, out = orc.forward_test(w, batch, synth.ARCHS[c["arch"]]["heads"])
            o = eng.forward(batch["img_q"], batch["img_s"], batch["target_s"], mask, [m["sample_skeleton"][0] for m in batch["img_metas"]])
            torch.cuda.synchronize()
            gots.append({k: o[k].cpu().numpy() for k in ("output_kpts", "similarity_map", "adj")})
            refs.append({k: out[k].numpy() for k in ("output_kpts", "similarity_map", "adj")})
            valids.append(mask[:, :, 0] > 0)
        del eng
        cat = lambda L, k, ax: np.concatenate([x[k] for x in L], ax)
        got = dict(output_kpts=cat(gots, "output_kpts", 1), similarity_map=cat(gots, "similarity_map", 0), adj=cat(gots, "adj", 0))
        ref = dict(output_kpts=cat(refs, "output_kpts", 1), similarity_map=cat(refs, "similarity_map", 0), adj=cat(refs, "adj", 0))
        valid = np.concatenate(valids, 0)
        st = stats(got, ref, valid, c["H"])
        st["weight_seed"], st["pairs"] = ws, int(valid.shape[0])
        per_seed.append(st)
        pooled_got.append(got); pooled_ref.append(ref); pooled_valid.append(valid)
    cat = lambda L, k, ax: np.concatenate([x[k] for x in L], ax)
    pooled = stats(dict(output_kpts=cat(pooled_got, "output_kpts", 1), similarity_map=cat(pooled_got, "similarity_map", 0), adj=cat(pooled_got, "adj", 0)),
                   dict(output_kpts=cat(pooled_ref, "output_kpts", 1), similarity_map=cat(pooled_ref, "similarity_map", 0), adj=cat(pooled_ref, "adj", 0)),
                   np.concatenate(pooled_valid, 0), c["H"])
    pooled["pairs"] = int(sum(v.shape[0] for v in pooled_valid))
    return per_seed, pooled


def test_headline_conformance_at_scale():
    """cfg2, fp16 backbone + mixed head (the bench default), 256 DISJOINT pairs x 2 weight seeds vs the oracle.  Gates = the observed
    rates with head-room (profiles/r03_conformance_fp16_mixed.json): the share of valid keypoints whose proposal argmax flips and the
    share outside 1e-3 are MEASURED quantities of this mode; every flip-free sample must be inside the tolerance outright."""
    per_seed, pooled = conformance_at_scale()
    print("conformance", per_seed, pooled)
    # observed (MI355X, round 3, 512 disjoint pairs, split-precision patch embedding): 13 argmax flips of 20 293 valid keypoints (6.4e-4;
    # 9 / 4 per weight seed), max |d| 1.65e-4 on the flip-free samples, p99 7.0e-5, median 3.3e-6, 5.2e-4 of the keypoints outside 1e-3
    # (all in flipped samples), PCK@0.2 against the oracle's answers 0.9997 (with single fp16 operands in the patch embedding: 22 flips,
    # 2.8e-4, 8.7e-5, 9.9e-4, 0.9993).  (The first round-3 record - 0 flips - had drawn the same 39 pairs per
    # weight seed eight times over: overlapping seeds.)  BASELINE.md section 4 gates a reduced-precision mode by its PCK@0.2 delta
    # (<= 0.1) and reports the flip count; the 1e-3 gate is the parity modes' (fp32, bf16x3).
    assert pooled["pairs"] >= 512
    assert pooled["max_clean"] < 5e-4, pooled                       # continuous part of the error: 2x inside the tolerance
    assert pooled["p99"] < 2e-4 and pooled["median"] < 1e-5
    assert pooled["flip_frac"] <= 2.5e-3, pooled                    # observed 1.08e-3: an indexing / synchronisation bug flips percents
    assert pooled["frac_gt_1e3"] <= 2.5e-3, pooled
    assert pooled["clean_samples"] >= 0.93 * pooled["pairs"], pooled   # observed 491 / 512
    assert pooled["pck_vs_oracle"] >= 0.998                         # observed 0.9993; north star: PCK@0.2 within +-0.1
    for st in per_seed:
        assert st["max_clean"] < 5e-4 and st["flip_frac"] <= 4e-3, st


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
