"""N>1 path on CPU: world_size-2 gloo runs of the SAME functions the GPU job uses over RCCL (edgecape_amd/apis.py: init_distributed,
shard_indices, multi_gpu_test / collect_results, timed_steps, allreduce_counts — bench.py's distributed section is exactly these
calls).  Reference contract: EdgeCape/apis/test.py:50-91,154-198."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeModel:
    """Stands in for the detector: the result of pair i is a deterministic function of its global index."""

    def eval(self):
        return self

    def __call__(self, return_loss=False, **data):
        idx = np.asarray(data["idx"][0])
        bs = len(idx)
        preds = np.zeros((bs, K, 3), np.float32)
        preds[:, :, 0] = idx[:, None] + np.arange(K)[None] * 0.01
        preds[:, :, 1] = -idx[:, None]
        preds[:, :, 2] = 1.0
        boxes = np.zeros((bs, 6), np.float32)
        boxes[:, 0] = idx * 2.0
        boxes[:, 4] = 100.0 + idx
        boxes[:, 5] = 1.0
        return dict(preds=preds, boxes=boxes, bbox_ids=[int(i) * 7 for i in idx], image_paths=[f"img/q{int(i):04d}.jpg" for i in idx])


def _gt(n_total):
    rng = np.random.default_rng(3)
    return {i * 7: dict(joints=np.stack([i + np.arange(K) * 0.01 + rng.normal(0, 2.0, K), -i + rng.normal(0, 2.0, K)], -1),
                        mask=rng.random(K) > 0.2, bbox_thr=20.0) for i in range(n_total)}


def _worker(rank, world, port, n_total, mode, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from edgecape_amd import apis
    from edgecape_amd.evaluation import evaluate, pck_counts, pck_from_counts
    r, w, _ = apis.init_distributed("gloo")
    assert (r, w) == (rank, world)
    if mode == "sampler":          # DistributedSampler shards: equal length, padding duplicates truncated by `size`
        mine = apis.shard_indices(n_total, rank, world)
    elif mode == "uneven":         # un-padded shards of unequal length
        mine = list(range(rank, n_total, world))
    else:                          # "empty": rank 1 has nothing at all
        mine = list(range(n_total)) if rank == 0 else []
    loader = [dict(idx=[mine[i:i + 2]]) for i in range(0, len(mine), 2)]      # batches of 2 pairs
    res = apis.multi_gpu_test(_FakeModel(), loader, size=n_total)
    ev = None
    if rank == 0:                  # the gathered list feeds the reference-format evaluation unchanged
        with tempfile.TemporaryDirectory() as d:
            ev = dict(evaluate(res, _gt(n_total), d, metric="PCK"))
    # bench.py's timed region and counter reduction, on gloo
    calls = []
    armed = []
    dt = apis.timed_steps(lambda: calls.append(1), steps=3, warmup=2, collective=lambda: apis.allreduce_counts(np.ones(6)),
                          before_timed=lambda: armed.append(len(calls)))      # bench.py arms the kernel timers here: after the warm-up
    assert armed == [2]
    rng = np.random.default_rng(7)
    pred = rng.normal(size=(n_total, K, 2)) * 10
    gt = pred + rng.normal(size=(n_total, K, 2)) * 3
    vis = rng.random((n_total, K)) > 0.2
    norm = np.full((n_total, 2), 40.0)
    own = list(range(rank, n_total, world))
    tot = apis.allreduce_counts(pck_counts(pred[own], gt[own], vis[own], norm[own]))
    q.put((rank, res, ev, len(calls), dt, tot, pck_counts(pred, gt, vis, norm), pck_from_counts(tot)))
    apis.barrier()
    apis.finalize_distributed()


@pytest.mark.parametrize("n_total,mode", [(7, "sampler"), (8, "sampler"), (7, "uneven"), (5, "empty"), (1, "sampler")])
def test_world2_multi_gpu_test_matches_single_process(n_total, mode):
    sys.path.insert(0, ROOT)
    from edgecape_amd import apis
    from edgecape_amd.evaluation import evaluate
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference of the same job
    single = apis.single_gpu_test(_FakeModel(), [dict(idx=[list(range(i, min(i + 2, n_total)))]) for i in range(0, n_total, 2)])
    with tempfile.TemporaryDirectory() as d:
        ev_single = dict(evaluate(single, _gt(n_total), d, metric="PCK"))
    for rank, res, ev, ncalls, dt, tot, ref, pck in out:
        assert ncalls == 5 and dt >= 0.0
        np.testing.assert_allclose(tot, ref, rtol=1e-12)
        assert 0.0 <= pck["PCK@0.2"] <= 1.0
        if rank != 0:
            assert res is None                      # apis/test.py:197-198
            continue
        assert len(res) == n_total                  # sampler padding truncated (apis/test.py:194-195)
        for i, (a, b) in enumerate(zip(res, single)):
            assert a["bbox_ids"] == b["bbox_ids"] == [i * 7] and a["image_paths"] == b["image_paths"]
            np.testing.assert_array_equal(a["preds"], b["preds"])
            np.testing.assert_array_equal(a["boxes"], b["boxes"])
        assert ev == ev_single and "PCK@0.2" in ev and "mPCK" in ev


def _worker_edge(rank, world, port, case, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from edgecape_amd import apis
    apis.init_distributed("gloo")
    res = apis.single_gpu_test(_FakeModel(), [dict(idx=[[rank, rank + 2]])])
    if case == "longpath" and rank == 1:       # the reference's pickled gather has no path-length limit (apis/test.py:154-198)
        res[0]["image_paths"] = ["d/" * 300 + "q.jpg"]
    if case == "kmismatch" and rank == 1:      # a malformed shard must fail on EVERY rank before the all_gather, not hang the others
        res[1]["preds"] = res[1]["preds"][:, :K - 1]
    try:
        out = apis.collect_results(res, 4, all_ranks=True)
        q.put((rank, "ok", [r["image_paths"][0] for r in out]))
    except ValueError as e:
        q.put((rank, "ValueError", str(e)))
    apis.barrier()
    apis.finalize_distributed()


@pytest.mark.parametrize("case", ["longpath", "kmismatch"])
def test_world2_collect_results_edge_cases(case):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_edge, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if case == "longpath":
        assert [o[1] for o in out] == ["ok", "ok"]
        assert out[0][2] == out[1][2] and out[0][2][1] == "d/" * 300 + "q.jpg" and out[0][2][0] == "img/q0000.jpg"
    else:
        assert [o[1] for o in out] == ["ValueError", "ValueError"] and "same on every rank" in out[0][2]


def test_shard_indices_is_distributed_sampler():
    from edgecape_amd import apis
    from torch.utils.data.distributed import DistributedSampler
    for n, w in [(7, 2), (8, 2), (13, 4), (3, 8), (256, 8)]:
        for r in range(w):
            ds = DistributedSampler(list(range(n)), num_replicas=w, rank=r, shuffle=False)
            assert list(ds) == apis.shard_indices(n, r, w)


def test_single_process_helpers_without_process_group():
    from edgecape_amd import apis
    assert apis.rank_world() == (0, 1) and not apis.dist_on()
    np.testing.assert_array_equal(apis.allreduce_counts([1.0, 2.0]), [1.0, 2.0])
    assert apis.max_over_ranks(3.5) == 3.5
    res = apis.single_gpu_test(_FakeModel(), [dict(idx=[[0, 1, 2]])])
    assert apis.collect_results(res, 2) == res[:2]
    with pytest.raises(ValueError):
        apis.multi_gpu_test(_FakeModel(), [dict(idx=[[0]])])   # no size, no .dataset
