"""N>1 path on CPU: world_size-2 gloo run of the sharding + gather code the GPU job uses over RCCL
(edgecape_amd/apis.py; reference contract EdgeCape/apis/test.py:50-91,154-198)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeModel:
    """Stands in for the detector: prediction of pair i is a deterministic function of its global index."""

    def eval(self):
        return self

    def __call__(self, return_loss=False, **data):
        idx = np.asarray(data["idx"][0])
        bs = len(idx)
        preds = np.zeros((bs, 5, 3), np.float32)
        preds[:, :, 0] = idx[:, None] + np.arange(5)[None] * 0.01
        preds[:, :, 1] = -idx[:, None]
        preds[:, :, 2] = 1.0
        return dict(preds=preds, boxes=np.zeros((bs, 6), np.float32), bbox_ids=list(idx), image_paths=[f"q{i}" for i in idx])


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from edgecape_amd import apis
    from edgecape_amd.evaluation import pck_counts, pck_from_counts
    mine = apis.shard_indices(n_total, rank, world)
    loader = [dict(idx=[mine[i:i + 2]]) for i in range(0, len(mine), 2)]      # batches of 2 pairs
    allp = apis.multi_gpu_test(_FakeModel(), loader, n_total=n_total)
    # PCK counters: every rank contributes its shard, the sum must equal the single-process result
    rng = np.random.default_rng(7)
    pred = rng.normal(size=(n_total, 5, 2)) * 10
    gt = pred + rng.normal(size=(n_total, 5, 2)) * 3
    vis = rng.random((n_total, 5)) > 0.2
    norm = np.full((n_total, 2), 40.0)
    own = sorted(set(i for i in mine if True))[: len(mine)]
    own = list(range(rank, n_total, world))                                     # un-padded shard
    c = pck_counts(pred[own], gt[own], vis[own], norm[own])
    tot = apis.allreduce_counts(c)
    ref = pck_counts(pred, gt, vis, norm)
    q.put((rank, allp, tot, ref, pck_from_counts(tot)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [7, 8])
def test_world2_gather_matches_single_process(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, allp, tot, ref, pck in res:
        assert allp.shape == (n_total, 5, 3)
        # dataset order restored, sampler padding truncated (apis/test.py:187-196)
        np.testing.assert_allclose(allp[:, 0, 0], np.arange(n_total), atol=1e-6)
        np.testing.assert_allclose(allp[:, 0, 1], -np.arange(n_total), atol=1e-6)
        np.testing.assert_allclose(tot, ref, rtol=1e-12)
        assert 0.0 <= pck["PCK@0.2"] <= 1.0


def test_shard_indices_is_distributed_sampler():
    from edgecape_amd import apis
    from torch.utils.data.distributed import DistributedSampler
    for n, w in [(7, 2), (8, 2), (13, 4), (3, 8)]:
        for r in range(w):
            ds = DistributedSampler(list(range(n)), num_replicas=w, rank=r, shuffle=False)
            assert list(ds) == apis.shard_indices(n, r, w)
