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
MEASURED on MI355X against the oracle, disjoint pairs x 2 weight seeds per configuration, on the round's FINAL library
(tools/gpu_conformance_all.sh, records profiles/r04_conformance_*.json, re-measured unchanged as r05_conformance_*.json; the at-scale gates below are these observations + <= 50 %):
  fp16 / mixed   cfg1 (ViT-S/14 @ 224, the reference's shipped config; 512 pairs)  11 flips of 19 288 valid keypoints (5.7e-4), 5.4e-4
                      outside 1e-3, max |d| 2.39e-4 on the 501 flip-free samples, PCK@0.2 vs the oracle's answers 0.9995
                 cfg2 (512 pairs) 12 of 20 293 (5.9e-4), 5.4e-4, 2.02e-4 on 501 samples, 0.9996
                 cfg4 (512 pairs) 17 of 19 699 (8.6e-4), 6.6e-4, 1.62e-4 on 496 samples, 0.9995
                 cfg5 (256 pairs) 14 of  9 645 (1.45e-3), 1.45e-3, 1.56e-4 on 242 samples, 0.9991
  bf16x3 / bf16x3  cfg1 0 flips, max 9.1e-6 over all 512 pairs; cfg2 1 flip (a 4.9e-5 near-tie), 1.1e-5 on the other 511.
                   Round 5 (profiles/r05_conformance_*: the backbone's GEMMs K-concatenated on the 8-phase kernel, same three products per
                   multiply): cfg1 0 flips (max 1.04e-5), cfg2 the same 1 flip (8.3e-6 elsewhere), cfg4 1 of 19 699 (8.5e-6), cfg5 1 of
                   9 645 (5.8e-6).  The floor: the EXACT mode (fp32 / fp32) flips that cfg2 near-tie too and nothing on cfg4 / cfg5.
(At the start of round 4, before the GELU / accumulator changes of the GEMM epilogue: cfg1 13 flips, cfg2 13 - the same rates.)
A device-side near-tie guard is NOT viable (near_tie_guard in the records): the similarity map (scale ~ 60) is up to 2.8e-2 off in
fp16 (3.2e-2 on the final library), and 54-70 % of the samples hold a valid keypoint whose top-2 gap is below twice that (2.4-3.9 % of
the keypoints).
"""
import functools
import os

import numpy as np
import pytest
import torch

from edgecape_amd import synth

pytestmark = pytest.mark.gpu

CFG = {
    # the reference's own shipped configuration (configs/test/1shot_split1.py:37,74; EdgeCape.py:33): ViT-S/14 @ 224, at the bench's batch
    "cfg1": dict(arch="dinov2_vits14", H=224, bs=32, S=1, wseed=0, iseed=4000),
    "cfg2": dict(arch="dinov2_vitb14", H=256, bs=32, S=1, wseed=0, iseed=1000),    # BASELINE configs[1]
    "cfg4": dict(arch="dinov2_vitb14", H=256, bs=16, S=5, wseed=0, iseed=2000),    # configs[3]
    "cfg5": dict(arch="dinov2_vitl14", H=384, bs=8, S=1, wseed=0, iseed=3000),     # configs[4]
}


@functools.lru_cache(maxsize=None)
def _weights(arch, seed):
    return synth.make_weights(arch, seed=seed)


# ---- the oracle's answers, computed once per box: the CPU oracle is the expensive half of every test here (5 pairs/s at cfg2), the same
# batches are asked for by several tests and by every child process of test_runtime_switch_matrix, and the at-scale cases need hundreds
# of pairs.  Results are cached under /tmp keyed by everything that determines them; missing ones are computed by a
# pool of worker processes side by side (the host has far more cores than one eager CPU forward can use).
ORACLE_KEYS = ("output_kpts", "similarity_map", "adj")


def _oracle_cache_path(name, wseed, seed, first_index, outliers):
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in (os.path.join(root, "oracle", "edgecape_oracle.py"), os.path.join(root, "edgecape_amd", "synth.py")):
        with open(f, "rb") as fh:
            h.update(fh.read())
    c = CFG[name]
    h.update(repr((c["arch"], c["H"], c["bs"], c["S"], wseed, seed, first_index, bool(outliers))).encode())
    d = os.path.join(os.environ.get("EC_ORACLE_CACHE", "/tmp/ec_oracle_cache"))
    os.makedirs(d, exist_ok=True)
    return os.path.join(d, h.hexdigest()[:24] + ".npz")


def _oracle_task(task):
    """One batch through the CPU oracle (worker process or in-line); returns the cache path it wrote."""
    name, wseed, seed, first_index, outliers, threads = task
    path = _oracle_cache_path(name, wseed, seed, first_index, outliers)
    if os.path.exists(path):
        return path
    from oracle import edgecape_oracle as orc   # the checker
    c = CFG[name]
    torch.set_num_threads(threads)
    w = synth.make_weights(c["arch"], seed=wseed, outliers=outliers)
    batch = synth.make_pairs(c["bs"], c["S"], c["H"], seed=seed, first_index=first_index, fixed_n_kp=False)
    with torch.no_grad():
        _, out = orc.forward_test(w, batch, synth.ARCHS[c["arch"]]["heads"])
    tmp = path + ".%d.tmp.npz" % os.getpid()
    np.savez(tmp, **{k: out[k].numpy() for k in ORACLE_KEYS})
    os.replace(tmp, path)
    return path


# ---- the reference's evaluation protocol (test_dataset.py:86-99): episodes of one support set and `qpe` consecutive queries ------------
def episode_set(name, wseed, n_ep, qpe=15):
    """n_ep synthetic episodes of a configuration: support sets from one generator stream, n_ep * qpe query images from another (disjoint
    from the pairwise conformance sets' seeds).  Returns (sup batch, query images, mask [n_ep,K,1], skeletons, episode_of_pair, query metas)."""
    c = CFG[name]
    sup = synth.make_pairs(n_ep, c["S"], c["H"], seed=c["iseed"] + 300000 * (1 + wseed), fixed_n_kp=False)
    qry = synth.make_pairs(n_ep * qpe, 1, c["H"], seed=c["iseed"] + 400000 * (1 + wseed))
    mask = sup["target_weight_s"][0].copy()
    for tw in sup["target_weight_s"]:
        mask = mask * tw
    skels = [m["sample_skeleton"][0] for m in sup["img_metas"]]
    return sup, qry["img_q"], mask, skels, np.repeat(np.arange(n_ep, dtype=np.int32), qpe), qry["img_metas"]


def expanded_pairs(sup, img_q, skels, ep, metas_q, idx):
    """the (support set, query) pairs `idx` of an episode set as the batch dict the reference's forward_test takes (what the
    reference evaluates: every pair on its own, support side recomputed per pair)"""
    e = ep[idx]
    S = len(sup["img_s"])
    metas = []
    for i, ei in zip(idx, e):
        m = dict(metas_q[i])
        m["sample_skeleton"] = [skels[ei] for _ in range(S)]
        m["sample_image_file"] = sup["img_metas"][ei]["sample_image_file"]
        metas.append(m)
    return dict(img_q=img_q[idx], img_s=[x[e] for x in sup["img_s"]], target_s=[x[e] for x in sup["target_s"]],
                target_weight_s=[x[e] for x in sup["target_weight_s"]], img_metas=metas)


def _oracle_episode_path(name, wseed, n_ep, qpe, chunk, chunk_pairs, outliers=False):
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in (os.path.join(root, "oracle", "edgecape_oracle.py"), os.path.join(root, "edgecape_amd", "synth.py")):
        with open(f, "rb") as fh:
            h.update(fh.read())
    c = CFG[name]
    h.update(repr(("episodes", c["arch"], c["H"], c["S"], c["iseed"], wseed, n_ep, qpe, chunk, chunk_pairs, bool(outliers))).encode())
    d = os.path.join(os.environ.get("EC_ORACLE_CACHE", "/tmp/ec_oracle_cache"))
    os.makedirs(d, exist_ok=True)
    return os.path.join(d, "ep_" + h.hexdigest()[:24] + ".npz")


def _oracle_episode_task(task):
    name, wseed, n_ep, qpe, chunk, chunk_pairs, outliers, threads = task
    path = _oracle_episode_path(name, wseed, n_ep, qpe, chunk, chunk_pairs, outliers)
    if os.path.exists(path):
        return path
    from oracle import edgecape_oracle as orc   # the checker
    c = CFG[name]
    torch.set_num_threads(threads)
    w = synth.make_weights(c["arch"], seed=wseed, outliers=outliers)
    sup, img_q, mask, skels, ep, metas_q = episode_set(name, wseed, n_ep, qpe)
    idx = np.arange(chunk * chunk_pairs, min((chunk + 1) * chunk_pairs, len(ep)))
    with torch.no_grad():
        _, out = orc.forward_test(w, expanded_pairs(sup, img_q, skels, ep, metas_q, idx), synth.ARCHS[c["arch"]]["heads"])
    tmp = path + ".%d.tmp.npz" % os.getpid()
    np.savez(tmp, **{k: out[k].numpy() for k in ORACLE_KEYS})
    os.replace(tmp, path)
    return path


def oracle_episode_outputs(name, wseed, n_ep, qpe=15, chunk_pairs=30, outliers=False):
    """The oracle's answers for all n_ep * qpe expanded pairs of an episode set, in pair order (cached per chunk of chunk_pairs pairs)."""
    n_chunks = (n_ep * qpe + chunk_pairs - 1) // chunk_pairs
    ncpu = effective_cpus()
    tasks = [(name, wseed, n_ep, qpe, ch, chunk_pairs, outliers) for ch in range(n_chunks)]
    missing = [t for t in tasks if not os.path.exists(_oracle_episode_path(*t))]
    workers = max(1, min(len(missing), 8, ncpu // 16))
    threads = max(4, min(32, ncpu // max(workers, 1)))
    if len(missing) > 1 and workers > 1:
        import multiprocessing as mp
        with mp.get_context("spawn").Pool(workers) as pool:
            pool.map(_oracle_episode_task, [t + (threads,) for t in missing], chunksize=1)
    else:
        for t in missing:
            _oracle_episode_task(t + (threads,))
    outs = []
    for t in tasks:
        with np.load(_oracle_episode_path(*t)) as z:
            outs.append({k: z[k] for k in ORACLE_KEYS})
    return dict(output_kpts=np.concatenate([o["output_kpts"] for o in outs], 1), similarity_map=np.concatenate([o["similarity_map"] for o in outs], 0),
                adj=np.concatenate([o["adj"] for o in outs], 0))


def episode_call_size(name, qpe=15):
    """queries per ec_forward_episodes call as bench.py's episode leg sizes them: a call's backbone pass holds about as many images as a
    pairwise step of the configuration ((1 + S) * bs): 60 + 4 at cfg2, 72 + 20..25 at cfg4."""
    c = CFG[name]
    return max(1, (1 + c["S"]) * c["bs"] * qpe // (qpe + c["S"]))


def conformance_episodes(n_ep=17, wseeds=(0, 1), backbone="fp16", head="mixed", name="cfg2", qpe=15, pipelined=True, outliers=False, also_pairwise=False):
    """A precision mode against the oracle through ec_forward_episodes - the entry point of the reference's evaluation protocol (15 queries
    per support set, test_dataset.py:86-99) - at the call size bench.py's `episode_cached` leg uses: n_ep episodes per weight seed
    (17 x 15 = 255 pairs) streamed in calls of episode_call_size() queries, the support sets of the episodes that start in a call riding
    in its backbone pass, slot cache, pipelined heads.  The oracle evaluates every expanded (support set, query) pair on its own, as
    the reference does.  Same statistics as conformance_at_scale."""
    from edgecape_amd.engine import HipEngine
    from edgecape_amd.episodes import stream_schedule
    c = CFG[name]
    q = episode_call_size(name, qpe)
    cap = (q + qpe - 1) // qpe + 2
    per_seed, pg, pr, pv = [], [], [], []
    for ws in wseeds:
        ref = oracle_episode_outputs(name, ws, n_ep, qpe, outliers=outliers)
        sup, img_q, mask, skels, ep, _ = episode_set(name, ws, n_ep, qpe)
        w = synth.make_weights(c["arch"], seed=ws, outliers=outliers)
        eng = HipEngine(w, arch=c["arch"], image_size=c["H"], max_batch=q, max_shots=c["S"], backbone_precision=backbone, head_precision=head)
        cache = eng.support_cache(cap)
        res = []
        for call in stream_schedule(ep, q, cap):
            new = None
            if len(call["new_episodes"]):
                e = np.asarray(call["new_episodes"])
                new = dict(img_s=[x[e] for x in sup["img_s"]], target_s=[x[e] for x in sup["target_s"]], mask_s=mask[e],
                           skeletons=[skels[i] for i in e], slots=call["new_slots"])
            res.append(eng.forward_episodes(cache, img_q[call["queries"]], call["slot_of_query"], new=new, pipelined=pipelined))
        if pipelined:
            eng.pipeline_flush()
        torch.cuda.synchronize()
        got = dict(output_kpts=np.concatenate([o["output_kpts"].cpu().numpy() for o in res], 1),
                   similarity_map=np.concatenate([o["similarity_map"].cpu().numpy() for o in res], 0),
                   adj=np.concatenate([o["adj"].cpu().numpy() for o in res], 0))
        del cache, res
        valid = (mask[:, :, 0] > 0)[ep]
        st = stats(got, ref, valid, c["H"])
        st["weight_seed"], st["pairs"], st["episodes"] = ws, int(valid.shape[0]), n_ep
        if also_pairwise:
            # the SAME expanded pairs through the pairwise entry point (ec_forward, batches of q): is a flip the data's (a near-tie of this
            # pair) or the entry point's?
            outs = []
            for i0 in range(0, len(ep), q):
                idx = np.arange(i0, min(i0 + q, len(ep)))
                e = ep[idx]
                o = eng.forward(img_q[idx], [x[e] for x in sup["img_s"]], [x[e] for x in sup["target_s"]], mask[e], [skels[j] for j in e])
                torch.cuda.synchronize()
                outs.append({k: o[k].cpu().numpy() for k in ORACLE_KEYS})
            gp = dict(output_kpts=np.concatenate([o["output_kpts"] for o in outs], 1), similarity_map=np.concatenate([o["similarity_map"] for o in outs], 0),
                      adj=np.concatenate([o["adj"] for o in outs], 0))
            sp = stats(gp, ref, valid, c["H"])
            am = lambda t: t["similarity_map"].reshape(valid.shape[0], valid.shape[1], -1).argmax(-1)
            fe, fp_ = (am(got) != am(ref)) & valid, (am(gp) != am(ref)) & valid
            st["pairwise_same_pairs"] = dict(flips=sp["flips"], max_clean=sp["max_clean"], p99=sp["p99"], flips_in_both=int((fe & fp_).sum()),
                                             flips_only_episodes=int((fe & ~fp_).sum()), flips_only_pairwise=int((~fe & fp_).sum()))
        del eng
        per_seed.append(st)
        pg.append(got); pr.append(ref); pv.append(valid)
    cat = lambda L, k, ax: np.concatenate([x[k] for x in L], ax)
    pooled = stats(dict(output_kpts=cat(pg, "output_kpts", 1), similarity_map=cat(pg, "similarity_map", 0), adj=cat(pg, "adj", 0)),
                   dict(output_kpts=cat(pr, "output_kpts", 1), similarity_map=cat(pr, "similarity_map", 0), adj=cat(pr, "adj", 0)),
                   np.concatenate(pv, 0), c["H"])
    pooled["pairs"] = int(sum(v.shape[0] for v in pv))
    pooled["queries_per_call"], pooled["queries_per_episode"], pooled["cache_slots"] = q, qpe, cap
    if also_pairwise:
        pooled["pairwise_same_pairs"] = {k: (max if k in ("max_clean", "p99") else sum)(st["pairwise_same_pairs"][k] for st in per_seed)
                                         for k in per_seed[0]["pairwise_same_pairs"]}
    return per_seed, pooled


def effective_cpus():
    """CPUs this process can really use: the affinity mask, cut down to the cgroup's CPU quota (a GPU box of the pool shows 256 logical
    CPUs and grants a fraction of them: eight 32-thread workers under such a quota ran an order of magnitude SLOWER than one)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def oracle_outputs(tasks):
    """tasks: (config name, weight seed, pair seed, first_index, outliers) -> list of dicts of the oracle's outputs, from the cache or
    computed now - several at once in worker processes when more than one is missing."""
    ncpu = effective_cpus()
    missing = [t for t in tasks if not os.path.exists(_oracle_cache_path(*t))]
    workers = max(1, min(len(missing), 8, ncpu // 16))
    threads = max(4, min(32, ncpu // max(workers, 1)))
    if len(missing) > 1 and workers > 1:
        import multiprocessing as mp
        with mp.get_context("spawn").Pool(workers) as pool:
            pool.map(_oracle_task, [t + (threads,) for t in missing], chunksize=1)
    else:
        for t in missing:
            _oracle_task(t + (threads,))
    out = []
    for t in tasks:
        with np.load(_oracle_cache_path(*t)) as z:
            out.append({k: z[k] for k in ORACLE_KEYS})
    return out


@functools.lru_cache(maxsize=None)
def _case(name):
    """(batch, mask, oracle outputs) of a BASELINE config at its FULL batch size (the CPU oracle needs seconds per config)."""
    c = CFG[name]
    batch = synth.make_pairs(c["bs"], c["S"], c["H"], seed=c["iseed"], fixed_n_kp=False)
    mask = batch["target_weight_s"][0].copy()
    for tw in batch["target_weight_s"]:
        mask = mask * tw
    ref = oracle_outputs([(name, c["wseed"], c["iseed"], 0, False)])[0]
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


def gap_analysis(got_sim, ref_sim, valid, flip=None):
    """What a device-side near-tie guard would see: per valid keypoint the top-2 gap of the HIP similarity map
    and the map's error against the oracle.  A guard that re-runs (through the conforming precision) every SAMPLE with a valid
    keypoint whose gap is below `thr` is safe when thr >= 2 x the map error (the oracle's argmax cell is then the HIP one); reported
    for thr = 2 x {p99, p99.9, max} of the per-keypoint maximum map error: the share of keypoints / samples it would re-run and
    how many of the observed flips it would have caught."""
    bs, K = valid.shape
    g = got_sim.reshape(bs, K, -1).astype(np.float64)
    r = ref_sim.reshape(bs, K, -1).astype(np.float64)
    top2 = np.partition(g, -2, axis=-1)[:, :, -2:]
    gap = (top2[:, :, 1] - top2[:, :, 0])                       # [bs,K] >= 0
    err = np.abs(g - r).max(-1)                                 # per keypoint: max over the map
    if flip is None:
        flip = (g.argmax(-1) != r.argmax(-1)) & valid
    gv, ev = gap[valid], err[valid]
    edges = [0.0, 1e-3, 2e-3, 5e-3, 1e-2, 2e-2, 5e-2, 1e-1, 2e-1, 5e-1, 1.0, 2.0, 5.0, 1e9]
    hist = np.histogram(gv, bins=edges)[0]
    out = dict(n_valid=int(valid.sum()), samples=int(bs), map_scale=float(np.abs(r[valid]).max()),
               map_err=dict(median=float(np.median(ev)), p99=float(np.quantile(ev, 0.99)), p999=float(np.quantile(ev, 0.999)), max=float(ev.max())),
               gap_hist=dict(edges=edges[:-1] + ["inf"], counts=[int(x) for x in hist]),
               gap_of_flipped=[float(x) for x in np.sort(gap[flip])], guards={})
    for tag, e in (("2x_p99", out["map_err"]["p99"]), ("2x_p999", out["map_err"]["p999"]), ("2x_max", out["map_err"]["max"])):
        thr = 2.0 * e
        near = (gap < thr) & valid
        out["guards"][tag] = dict(thr=float(thr), kpt_frac=float(near.sum() / max(valid.sum(), 1)), sample_frac=float(near.any(1).mean()),
                                  flips_caught=int((near & flip).sum()), flips=int(flip.sum()),
                                  flipped_samples_caught=int((near.any(1) & flip.any(1)).sum()), flipped_samples=int(flip.any(1).sum()))
    return out


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg4", "cfg5"])
def test_parity_mode_bf16x3(name):
    """bf16x3 backbone + bf16x3 head: the tolerance-conforming fast mode.  Full batch of the BASELINE config vs the oracle."""
    s = _run(name, "bf16x3", "bf16x3")
    print(name, "bf16x3/bf16x3", s)
    assert s["flips"] == 0
    assert s["max_all"] < 1e-3, s
    assert s["p99"] < 1e-4 and s["adj_err"] < 1e-4
    assert s["pck_vs_oracle"] == 1.0


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg4", "cfg5"])
def test_parity_mode_fp16x2(name):
    """fp16x2 backbone (fp16 MFMAs + both correction terms in one block-scaled FP8 pass: two MFMA units per product) + bf16x3 head: the
    tolerance-conforming mode at two thirds of the bf16x3 backbone's matrix work.  Full batch of the BASELINE config vs the oracle,
    the same gates as the bf16x3 parity mode."""
    s = _run(name, "fp16x2", "bf16x3")
    print(name, "fp16x2/bf16x3", s)
    assert s["flips"] == 0
    assert s["max_all"] < 1e-3, s
    assert s["p99"] < 1e-4 and s["adj_err"] < 1e-4
    assert s["pck_vs_oracle"] == 1.0


def _headline_gates(s, name):
    """One full batch of a configuration in the bench's default backbone precision.  Observed on every configuration (rounds 2-4):
    0 flips on these batches, max |d| 1.4-1.9e-4, p99 <= 8e-5, median <= 6e-6; gates = that + <= 50 %.  A near-tie may flip on another
    box (a flip moves ONE sample by up to a grid cell): at most one, and then only that sample may leave the tolerance."""
    assert s["max_clean"] < 3.0e-4, s                     # every sample without an argmax flip: > 3x inside the north-star tolerance
    assert s["p99"] < 1.2e-4 and s["median"] < 9e-6, s
    assert s["flips"] <= 1, s
    assert s["clean_samples"] >= CFG[name]["bs"] - 1
    if s["flips"] == 0:
        assert s["max_all"] < 3.0e-4 and s["frac_gt_1e3"] == 0.0
    assert s["pck_vs_oracle"] >= 0.99                     # north star: PCK@0.2 within +-0.1 (here against the oracle's own answers)


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg4", "cfg5"])
def test_headline_mode_fp16(name):
    """fp16 backbone + bf16x3 head.  Continuous error most of an order of magnitude inside the tolerance; argmax near-ties may flip
    (measured rate at scale: 6-15e-4 of the valid keypoints, module docstring)."""
    s = _run(name, "fp16", "bf16x3")
    print(name, "fp16/bf16x3", s)
    _headline_gates(s, name)


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg4", "cfg5"])
def test_headline_mode_fp16_mixed_head(name):
    """fp16 backbone + MIXED head (bench.py's default precision): bf16x3 wherever the proposal generator's argmax depends on it,
    single-pass fp16 MFMAs in the Linear layers and attentions of the skeleton head and the decoder layers (EC_MIXED).  Same gates as the
    fp16 / bf16x3 mode - the head's share of the error is below the backbone's (oracle/head_precision_study.py) - plus: no more argmax
    flips than that mode, and the refined adjacency (the skeleton head's product) at 1e-4."""
    s = _run(name, "fp16", "mixed")
    s0 = _run(name, "fp16", "bf16x3")
    print(name, "fp16/mixed", s, "\n     fp16/bf16x3", s0)
    _headline_gates(s, name)
    assert s["flips"] <= s0["flips"]                      # the encoder / proposal path is bit-identical to the bf16x3 head's
    assert s["adj_err"] < 1e-4


def conformance_at_scale(n_batches=8, wseeds=(0, 1), backbone="fp16", head="mixed", name="cfg2", outliers=False):
    """The headline precision against the oracle on n_batches x bs pairs per weight seed (cfg2: 8 x 32 = 256 pairs, 2 seeds): the
    MEASURED rate of argmax flips and of keypoints outside 1e-3, instead of one lucky 32-pair sample.  Returns one stats dict per
    weight seed plus the pooled one.  (Also run by tools/conformance.py, which writes the record under profiles/.)"""
    from edgecape_amd.engine import HipEngine
    c = CFG[name]
    # every oracle answer first (cached / computed side by side), then the HIP forwards
    # DISJOINT pairs: pair i of synth.make_pairs is seeded by seed + first_index + i, so batch b takes the indices b*bs .. b*bs+bs-1
    # of the weight seed's own range (the first round-3 record used seed + b and so saw the same 39 pairs eight times over)
    tasks = [(name, ws, c["iseed"] + 100000 * (1 + ws), b * c["bs"], outliers) for ws in wseeds for b in range(n_batches)]
    oracle = dict(zip(tasks, oracle_outputs(tasks)))
    per_seed, pooled_got, pooled_ref, pooled_valid = [], [], [], []
    for ws in wseeds:
        w = synth.make_weights(c["arch"], seed=ws, outliers=outliers)   # outliers: planted DINOv2-like activation statistics (synth.add_activation_outliers)
        eng = HipEngine(w, arch=c["arch"], image_size=c["H"], max_batch=c["bs"], max_shots=c["S"], backbone_precision=backbone, head_precision=head)
        gots, refs, valids = [], [], []
        for b in range(n_batches):
            batch = synth.make_pairs(c["bs"], c["S"], c["H"], seed=c["iseed"] + 100000 * (1 + ws), first_index=b * c["bs"], fixed_n_kp=False)
            mask = batch["target_weight_s"][0].copy()
            for tw in batch["target_weight_s"]:
                mask = mask * tw
            o = eng.forward(batch["img_q"], batch["img_s"], batch["target_s"], mask, [m["sample_skeleton"][0] for m in batch["img_metas"]])
            torch.cuda.synchronize()
            gots.append({k: o[k].cpu().numpy() for k in ORACLE_KEYS})
            refs.append(oracle[(name, ws, c["iseed"] + 100000 * (1 + ws), b * c["bs"], outliers)])
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
    pooled["near_tie_guard"] = gap_analysis(cat(pooled_got, "similarity_map", 0), cat(pooled_ref, "similarity_map", 0), np.concatenate(pooled_valid, 0))
    return per_seed, pooled


# Observed at scale on the round's library (gpurun_out/gates, tools/gpu_gate_numbers.sh: the first n_batches disjoint batches of each of the two
# weight seeds, fp16 / mixed): the gates are these + <= 50 % (flip counts: + 50 % or + 3, whichever is larger - a handful of near-ties).
# The GPU boxes of the pool grant 16 CPU cores (cgroup quota; 256 are visible), so the CPU oracle - 5 pairs/s at cfg2, 0.8 at cfg5 - is
# what these tests cost: 4 batches per seed for the 1-shot ViT-S / ViT-B configurations (25 + 65 s), 2 for the 5-shot and the ViT-L ones
# (3 took 71 + 87 s) - too few pairs for a flip RATE there, the gates that bite are the continuous ones (max on flip-free samples, p99,
# median).  The full-scale records (512 / 512 / 512 / 256 pairs; README) are tools/conformance.py's, under profiles/.
AT_SCALE = {
    "cfg1": dict(n_batches=4, pairs=256, flips=5, n_valid=9538, frac_gt_1e3=5.3e-4, max_clean=2.39e-4, p99=6.7e-5, median=5.0e-6, flipped_samples=5, pck=0.9995, seed_flips=3),
    "cfg2": dict(n_batches=4, pairs=256, flips=7, n_valid=9842, frac_gt_1e3=7.2e-4, max_clean=2.02e-4, p99=7.1e-5, median=3.3e-6, flipped_samples=6, pck=0.9996, seed_flips=4),
    "cfg4": dict(n_batches=2, pairs=64, flips=1, n_valid=2309, frac_gt_1e3=3.0e-4, max_clean=1.23e-4, p99=5.5e-5, median=3.0e-6, flipped_samples=1, pck=0.9990, seed_flips=1),
    "cfg5": dict(n_batches=2, pairs=32, flips=1, n_valid=1239, frac_gt_1e3=8.1e-4, max_clean=1.39e-4, p99=7.7e-5, median=4.9e-6, flipped_samples=1, pck=0.9992, seed_flips=1),
}


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg4", "cfg5"])
def test_headline_conformance_at_scale(name):
    """fp16 backbone + mixed head (the bench default) on 256 DISJOINT pairs x 2 weight seeds vs the oracle, for the reference's own
    shipped configuration (cfg1: ViT-S/14 @ 224, configs/test/1shot_split1.py:37,74) and the benched one (cfg2).  The share of valid
    keypoints whose proposal argmax flips and the share outside 1e-3 are MEASURED rates of this mode; every flip-free sample must be
    inside the tolerance outright.  BASELINE.md section 4 gates a reduced-precision mode by its PCK@0.2 delta (<= 0.1) and reports
    the flip count; the 1e-3 coordinate gate on EVERY keypoint is the parity modes' (fp32, bf16x3: test_parity_mode_bf16x3)."""
    o = AT_SCALE[name]
    per_seed, pooled = conformance_at_scale(n_batches=o["n_batches"], name=name)
    print("conformance", name, per_seed, {k: v for k, v in pooled.items() if k != "near_tie_guard"})
    assert pooled["pairs"] == o["pairs"] and abs(pooled["n_valid"] - o["n_valid"]) <= 0.01 * o["n_valid"]      # the same pairs as the record
    more = lambda n: max(1.5 * n, n + 3)                                   # flip counts of a few: + 50 % or + 3
    assert pooled["flips"] <= more(o["flips"]), pooled                     # a regression that doubles the flip rate fails
    assert pooled["frac_gt_1e3"] <= max(1.5 * o["frac_gt_1e3"], more(o["flips"]) * 45 / o["n_valid"]), pooled   # (a flipped sample: ~40 keypoints)
    assert pooled["max_clean"] < 1.5 * o["max_clean"], pooled              # continuous part of the error: > 3.5x inside the tolerance
    assert pooled["p99"] < 1.5 * o["p99"] and pooled["median"] < 1.5 * o["median"], pooled
    assert pooled["pairs"] - pooled["clean_samples"] <= more(o["flipped_samples"]), pooled
    assert pooled["pck_vs_oracle"] >= o["pck"] - 3.0 * 45 / o["n_valid"], pooled      # north star: PCK@0.2 within +-0.1 (three more flipped samples)
    for st in per_seed:
        assert st["max_clean"] < 1.5 * o["max_clean"] and st["flips"] <= more(o["seed_flips"]), st
    g = pooled["near_tie_guard"]["guards"]["2x_max"]                       # the guard decision's evidence stays measurable
    assert g["flips_caught"] == g["flips"] and g["sample_frac"] > 0.25, g


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg4", "cfg5"])
def test_conforming_mode_at_scale(name):
    """fp16x2 backbone + bf16x3 head (the conforming mode, round 6) on the SAME disjoint pairs as test_headline_conformance_at_scale (the
    oracle's answers come from the cache that test filled).  Full-scale records (profiles/r06_conformance_*fp16x2_bf16x3.json, 512 / 512 /
    512 / 256 pairs): 0 / 2 / 1 / 1 argmax flips of 19 288 / 20 293 / 19 699 / 9 645 valid keypoints - near-ties with top-2 gaps of ~2e-5 in a
    map of scale ~45, of which the EXACT fp32-MFMA mode flips one as well - and max |d kpt| 1.0e-5 .. 1.4e-5 on the flip-free samples.
    Gates: every flip-free sample 20 x inside the tolerance; at most the handful of near-tie flips the records show."""
    o = AT_SCALE[name]
    per_seed, pooled = conformance_at_scale(n_batches=o["n_batches"], backbone="fp16x2", head="bf16x3", name=name)
    print("conformance fp16x2/bf16x3", name, {k: v for k, v in pooled.items() if k != "near_tie_guard"})
    assert pooled["pairs"] == o["pairs"]
    assert pooled["flips"] <= 2, pooled
    assert pooled["max_clean"] < 5e-5 and pooled["p99"] < 2e-5 and pooled["median"] < 1e-6, pooled
    assert pooled["pairs"] - pooled["clean_samples"] <= 2
    assert pooled["pck_vs_oracle"] >= 1.0 - 2.0 * 45 / o["n_valid"], pooled


def test_fp16_backbone_on_outlier_activation_statistics():
    """Model-level evidence that the fp16 backbone survives the activation statistics of released DINOv2 checkpoints (VERDICT r3
    missing item 5; the checkpoints are unreachable offline, so the statistics are planted into random-init weights by
    synth.add_activation_outliers: massive residual-stream channels from block 2 on, outlier neurons in every MLP hidden layer,
    LayerNorm gains of 10-15).  ViT-S/14 @ 224, 16 pairs against the oracle on the SAME weights: the planted statistics must really be
    there (oracle features / hidden layer, recorded in the printout), every output finite, and the continuous error of the same
    order as on ordinary weights.  bf16x3 on the same weights stays inside the tolerance outright."""
    from edgecape_amd.engine import HipEngine
    from oracle import edgecape_oracle as orc   # the checker
    arch, H, bs = "dinov2_vits14", 224, 16
    w = synth.make_weights(arch, seed=5, outliers=True)
    batch = synth.make_pairs(bs, 1, H, seed=5000, fixed_n_kp=False)
    mask = batch["target_weight_s"][0].copy()
    torch.set_num_threads(min(32, torch.get_num_threads()))
    _, out = orc.forward_test(w, batch, synth.ARCHS[arch]["heads"])
    ref = {k: out[k].numpy() for k in ("output_kpts", "similarity_map", "adj")}
    taps = {}
    feat = orc.dinov2_features(w, torch.from_numpy(batch["img_q"]), synth.ARCHS[arch]["heads"], taps=taps).numpy()   # [B, C, g, g]
    print("planted statistics (oracle): |feature| max", float(np.abs(feat).max()), "median", float(np.median(np.abs(feat))),
          {k: v for k, v in taps.items() if isinstance(v, float)})
    assert taps["resid_absmax"] > 50.0 * taps["resid_absmedian"] and taps["hidden_absmax"] > 100.0      # the statistics are there
    res = {}
    for bb, hd in (("fp16", "mixed"), ("bf16x3", "bf16x3")):
        eng = HipEngine(w, arch=arch, image_size=H, max_batch=bs, max_shots=1, backbone_precision=bb, head_precision=hd)
        o = eng.forward(batch["img_q"], batch["img_s"], batch["target_s"], mask, [m["sample_skeleton"][0] for m in batch["img_metas"]])
        torch.cuda.synchronize()
        got = {k: o[k].cpu().numpy() for k in ("output_kpts", "similarity_map", "adj")}
        assert all(np.isfinite(v).all() for v in got.values()), (bb, hd)
        res[bb] = stats(got, ref, mask[:, :, 0] > 0, H)
        f = eng.backbone(batch["img_q"]).cpu().numpy()
        res[bb]["feat_err_over_max"] = float(np.abs(f - feat).max() / np.abs(feat).max())
        del eng
    print("outlier statistics: fp16/mixed", res["fp16"], "\n                    bf16x3", res["bf16x3"])
    assert res["bf16x3"]["flips"] == 0 and res["bf16x3"]["max_all"] < 1e-3
    s = res["fp16"]
    # observed (MI355X, round 4): residual-stream maximum 516 against a median of 1.0, hidden-layer maximum 199; fp16 / mixed: 1 flip of 520,
    # max 3.3e-4 on the flip-free samples (1.9e-4 on ordinary weights), p99 2.2e-4; bf16x3: 6.3e-6
    assert s["flips"] <= 2 and s["max_clean"] < 5e-4 and s["p99"] < 3.3e-4 and s["pck_vs_oracle"] >= 0.98, s


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
