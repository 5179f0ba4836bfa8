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
    elif mode == "episodes":       # whole episodes (groups of 3 consecutive pairs) per rank: what the episode cache needs (round 5)
        mine = apis.shard_indices(n_total, rank, world, group=3)
        assert all(mine[i] + 1 == mine[i + 1] for i in range(0, len(mine) - 1) if mine[i] % 3 != 2 and mine[i] + 1 < n_total)
    else:                          # "empty": rank 1 has nothing at all
        mine = list(range(n_total)) if rank == 0 else []
    loader = [dict(idx=[mine[i:i + 2]]) for i in range(0, len(mine), 2)]      # batches of 2 pairs
    res = apis.multi_gpu_test(_FakeModel(), loader, size=n_total, group=3 if mode == "episodes" else 1)
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


@pytest.mark.parametrize("n_total,mode", [(7, "sampler"), (8, "sampler"), (7, "uneven"), (5, "empty"), (1, "sampler"), (12, "episodes"), (10, "episodes"),
                                          (2, "episodes")])
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


class _FakeLib:
    """ec_profile / ec_profile_read of the C ABI (the QKV launch timers bench.py arms, mode 2: one sampled launch of 1 ms per step)."""

    def __init__(self):
        self.armed = 0

    def ec_profile(self, h, on, n):
        self.armed = n if on else 0
        return 0

    def ec_profile_read(self, h, tot, nl):
        tot._obj.value, nl._obj.value = 1.0 * self.armed, self.armed
        return 0


class _FakeEngine:
    """Stands in for edgecape_amd.engine.HipEngine in bench.main() on CPU: same constructor keys and the methods bench.py calls;
    the 'keypoints' of pair i are a deterministic function of the query image, so every rank's shard gives different outputs."""
    K, dec_layers = 100, 3

    def __init__(self, sd, arch, image_size, max_batch, max_shots, backbone_precision, head_precision):
        self.lib, self.h, self.calls = _FakeLib(), None, 0

    @staticmethod
    def _edges(skeletons, bs):
        return np.zeros((0, 2), np.int32), np.zeros(bs + 1, np.int32)

    def _outputs(self, bs):
        import torch
        return dict(output_kpts=torch.zeros(self.dec_layers, bs, self.K, 2)), None

    def forward_resident(self, iq, is_, ts, ms, edges, off, outputs):
        self.calls += 1
        m = iq.reshape(iq.shape[0], -1).mean(1)
        outputs[0]["output_kpts"][:] = (0.5 + 0.1 * m)[None, :, None, None]
        return outputs[0]

    forward_pipelined = forward_resident          # (the stand-in has no decoder to defer)

    def pipeline_flush(self, stream=None):
        self.flushed = getattr(self, "flushed", 0) + 1

    # episode-cache entry points (bench.episode_mode)
    def support_cache(self, max_episodes):
        return dict(cap=max_episodes)

    def prepare_episode_call(self, img_q, slot_of_query, new=None):
        assert new is None or len(new["slots"]) == new["img_s"][0].shape[0]
        return dict(bs=img_q.shape[0], new=0 if new is None else len(new["slots"]))

    def forward_episodes(self, cache, prepared=None, outputs=None, pipelined=False):
        assert pipelined and outputs is not None and prepared["new"] <= cache["cap"]
        self.episode_calls = getattr(self, "episode_calls", 0) + 1
        return outputs[0]


def _worker_bench(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import io
    import json
    from contextlib import redirect_stdout
    import edgecape_amd.engine as engine
    engine.HipEngine = _FakeEngine
    import bench
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "3", "--warmup", "1", "--batch", "4", "--image-size", "56", "--arch", "dinov2_vits14",
                "--sustained-seconds", "0.05"]
    buf = io.StringIO()
    with redirect_stdout(buf):
        res = bench.main()
    lines = [ln for ln in buf.getvalue().splitlines() if ln.strip()]
    q.put((rank, res, [json.loads(ln) for ln in lines]))


def test_world2_bench_main_runs_its_distributed_branches():
    """bench.py main() itself with WORLD_SIZE = 2 on gloo and a stand-in engine: the N > 1 branches (per-rank shard by global pair
    index, timed region with the counter all-reduce, max over ranks, result assembled and printed by rank 0 only, CPU legs skipped,
    the `sustained` and episode-protocol legs with their own timed regions on every rank)
    execute here before an 8-GPU driver run does it for the first time.  Contract: apis/test.py:154-198 + the bench contract."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bench, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, res0, printed0), (r1, res1, printed1) = out
    assert res1 is None and printed1 == []                      # only rank 0 reports
    assert len(printed0) == 1 and printed0[0] == res0           # ONE JSON line
    assert res0["n_gpus"] == 2 and res0["steps"] == 3 and res0["warmup"] == 1 and res0["scaling"] == "weak"
    assert res0["config"]["global_batch"] == 8 and res0["config"]["parallelism"].startswith("dp2")
    assert res0["value"] > 0 and abs(res0["value"] - 2 * 4 * 3 / (res0["ms_per_step"] * 3e-3)) / res0["value"] < 0.01   # whole-job aggregate
    assert res0["roofline"]["launches_timed"] == 3 and res0["roofline"]["avg_launch_ms"] == 1.0     # sampled: one QKV launch per step
    assert "cpu_baseline" not in res0 and "bf16_mode" not in res0 and "conforming_mode" not in res0   # rank-0-only legs are N = 1 only
    # the legs every rank takes part in (their own timed regions with the same barriers / max over ranks): whole-job aggregates
    su, epi = res0["sustained"], res0["episode_cached"]
    assert su["steps"] >= 3 and su["value"] > 0 and abs(su["value"] - 2 * 4 * su["steps"] / su["seconds"]) / su["value"] < 0.01
    assert epi["pairs"] == 32 * 15 and epi["queries_per_call"] == 7 and epi["calls_per_pass"] == -(-480 // 7)
    assert abs(epi["value"] - 2 * 480 * epi["passes_timed"] / epi["seconds"]) / epi["value"] < 0.01
    assert "parity_sample" not in epi                           # the episode leg's oracle sample (round 6) is an N = 1, rank-0-only leg as well
    assert res0["pipelined"] is True and "unpipelined" not in res0
    assert set(res0["pck_vs_synthetic_gt"]) >= {"PCK@0.2"}


def test_shard_indices_is_distributed_sampler():
    from edgecape_amd import apis
    from torch.utils.data.distributed import DistributedSampler
    for n, w in [(7, 2), (8, 2), (13, 4), (3, 8), (256, 8)]:
        for r in range(w):
            ds = DistributedSampler(list(range(n)), num_replicas=w, rank=r, shuffle=False)
            assert list(ds) == apis.shard_indices(n, r, w)
    # group-wise shards (round 5): every rank holds WHOLE groups, all ranks equally many, every item covered
    for n, w, g in [(480, 8, 15), (45, 2, 15), (47, 4, 15), (7, 3, 15), (30, 4, 1)]:
        shards = [apis.shard_indices(n, r, w, group=g) for r in range(w)]
        n_groups = -(-n // g)
        assert set(i for s in shards for i in s) == set(range(n))
        assert len(set(len(set(i // g for i in s)) for s in shards)) == 1 or n_groups < w           # the same number of groups each
        for s in shards:
            for gi in set(i // g for i in s):
                assert [i for i in s if i // g == gi][:min(g, n - gi * g)] == list(range(gi * g, min((gi + 1) * g, n)))


def test_single_process_helpers_without_process_group():
    from edgecape_amd import apis
    assert apis.rank_world() == (0, 1) and not apis.dist_on()
    np.testing.assert_array_equal(apis.allreduce_counts([1.0, 2.0]), [1.0, 2.0])
    assert apis.max_over_ranks(3.5) == 3.5
    res = apis.single_gpu_test(_FakeModel(), [dict(idx=[[0, 1, 2]])])
    assert apis.collect_results(res, 2) == res[:2]
    with pytest.raises(ValueError):
        apis.multi_gpu_test(_FakeModel(), [dict(idx=[[0]])])   # no size, no .dataset


def test_init_distributed_nccl_passes_device_id(monkeypatch):
    """the RCCL process group is created with device_id = this rank's GPU (eager communicator creation on the right
    device - a lazily created communicator would be built inside the first collective, i.e. possibly inside bench.py's timed region)."""
    import torch
    import torch.distributed as dist
    from edgecape_amd import apis
    calls = {}
    monkeypatch.setenv("RANK", "1"); monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: calls.setdefault("set_device", d))
    monkeypatch.setattr(dist, "is_initialized", lambda: False)
    monkeypatch.setattr(dist, "init_process_group", lambda backend, **kw: calls.update(backend=backend, **kw))
    assert apis.init_distributed("nccl") == (1, 2, 1)
    assert calls["backend"] == "nccl" and calls["set_device"] == 1
    assert calls["device_id"] == torch.device("cuda", 1) and calls["rank"] == 1 and calls["world_size"] == 2
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" or os.environ["MASTER_ADDR"]


def test_timed_steps_warms_collectives_before_the_clock(monkeypatch):
    """The order of bench.py's timed region: warm-up steps, an all-reduce and a barrier (communicators exist), the arming hook, a
    barrier, THEN the clock; K steps, the job's collective and the closing barrier inside; max over ranks after the clock."""
    import time
    from edgecape_amd import apis
    log = []
    monkeypatch.setattr(apis, "barrier", lambda: log.append("barrier"))
    monkeypatch.setattr(apis, "max_over_ranks", lambda v: (log.append("allreduce_max"), float(v))[1])
    ticks = iter(range(100))
    monkeypatch.setattr(time, "perf_counter", lambda: (log.append("clock"), float(next(ticks)))[1])
    dt = apis.timed_steps(lambda: log.append("step"), 3, 2, collective=lambda: log.append("collective"), before_timed=lambda: log.append("arm"))
    assert log == ["step", "step", "allreduce_max", "barrier", "arm", "barrier", "clock", "step", "step", "step", "collective", "barrier", "clock",
                   "allreduce_max"]
    assert dt == 1.0
