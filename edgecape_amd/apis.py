"""Evaluation loops with the reference's contract (EdgeCape/apis/test.py) and the one-process-per-GPU plumbing around them.

`single_gpu_test`   apis/test.py:14-47: run the loader, split every batch result into per-sample result dicts.
`multi_gpu_test`    apis/test.py:50-91 + collect_results_gpu (:154-198): every rank runs its shard, rank 0 gets the per-sample
                    result dicts of the whole dataset in dataset order (sampler padding truncated), other ranks get None.
                    The reference pickles python objects and all_gathers the bytes; here every sample is ONE fixed-size
                    byte record (preds [K,3] f32 | box [6] f32 | bbox_id i64 | path) and the exchange is one all_gather of a
                    [n, record] uint8 tensor - RCCL over xGMI with backend "nccl", the same code on CPU tensors with gloo.
`init_distributed / timed_steps / allreduce_counts / max_over_ranks`
                    the rank / barrier / timing / counter-reduction helpers `bench.py` runs on the GPUs; the world-size-2 gloo
                    tests (tests/test_dist_gloo.py) call exactly these functions.
Pairs are independent: nothing but results and PCK counters ever crosses GPUs.
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist



# ---- process group --------------------------------------------------------------------------------------------------
def dist_on():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def comm_device(device=None):
    if device is not None:
        return torch.device(device)
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def init_distributed(backend=None):
    """One process per GPU, launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment).
    backend None -> "nccl" (= RCCL on ROCm) when a GPU is visible, else "gloo".  Returns (rank, world, local_rank)."""
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = dict(device_id=torch.device("cuda", local_rank)) if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def finalize_distributed():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def _device_sync():
    if torch.cuda.is_available() and (not dist.is_initialized() or dist.get_backend() == "nccl"):
        torch.cuda.synchronize()


def barrier():
    """Rendezvous of all ranks with the local device drained on both sides."""
    _device_sync()
    if dist_on():
        dist.barrier()
    _device_sync()


def max_over_ranks(value):
    if not dist_on():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=comm_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_counts(counts, device=None):
    """Sum a small float64 vector (the PCK hit / pair counters of evaluation.pck_counts) over the ranks."""
    c = np.asarray(counts, np.float64)
    if not dist_on():
        return c
    t = torch.from_numpy(c.copy()).to(comm_device(device))
    dist.all_reduce(t)
    return t.cpu().numpy()


def timed_steps(step, steps, warmup, collective=None, before_timed=None):
    """bench.py's timed region: `warmup` untimed steps, then EXACTLY `steps` steps between two barriers (device drained on both
    sides); `collective` (the job's only cross-rank exchange) runs inside the region; `before_timed` (e.g. arming the kernel
    timers) runs between the warm-up and the first barrier; returns the MAX over ranks of the wall time."""
    for _ in range(warmup):
        step()
    # One all-reduce and one barrier BEFORE the clock starts, whatever the arguments: RCCL creates communicators / channels lazily on
    # the first collective of a kind, and that one-off cost (hundreds of ms on an 8-GPU node) must not land inside the timed region,
    # where `collective` and the closing barrier run.  (init_distributed also passes device_id, which makes the process group
    # create its communicator eagerly.)
    max_over_ranks(0.0)
    if before_timed is not None:
        barrier()
        before_timed()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if collective is not None:
        collective()
    barrier()
    return max_over_ranks(time.perf_counter() - t0)


# ---- sharding -------------------------------------------------------------------------------------------------------
def shard_indices(n_total, rank, world_size, group=1):
    """DistributedSampler(shuffle=False) semantics: pad to equal length by wrapping around, rank r takes r, r+W, ...
    group > 1 (round 5): the same over GROUPS of `group` consecutive items - rank r takes groups r, r+W, ... whole.  The reference's
    pair order is 15 consecutive pairs per support set (test_dataset.py:86-99): with group=15 a rank sees every one of its support sets
    with all its queries, which is what the support-side episode cache needs (item-wise round-robin would hand every rank 15 / W
    queries of EVERY support set and have all ranks encode all of them)."""
    if group <= 1:
        per = -(-n_total // world_size)
        idx = list(range(n_total))
        pad = per * world_size - n_total
        if pad > 0 and n_total > 0:
            idx += (idx * (-(-pad // n_total)))[:pad]   # wraps more than once when n_total < world_size
        return idx[rank::world_size]
    n_groups = -(-n_total // group)
    out = []
    for g in shard_indices(n_groups, rank, world_size):
        out.extend(range(g * group, min((g + 1) * group, n_total)))
    return out


# ---- evaluation loops -----------------------------------------------------------------------------------------------
def _per_sample(result):
    """Split one batch result of `model(return_loss=False, ...)` into the reference's per-sample result dicts."""
    if "preds" not in result:
        return
    preds, boxes = np.asarray(result["preds"]), np.asarray(result["boxes"])
    for i, (bid, path) in enumerate(zip(result["bbox_ids"], result["image_paths"])):
        yield {"preds": preds[i:i + 1], "boxes": boxes[i:i + 1], "bbox_ids": [bid], "image_paths": [path]}


def single_gpu_test(model, data_loader, pipelined=False):
    """apis/test.py:14-47.  Returns the list of per-sample result dicts in loader order.

    pipelined=True (models with submit() / collect(), i.e. edgecape_amd.detector.EdgeCape): batch i+1 is submitted before batch i is
    collected, so batch i's decoder phase and host decode overlap batch i+1's backbone (ec_forward_pipelined); same results, same
    order - the reference's loop (apis/test.py:31-33) only needs them in order."""
    model.eval()
    out = []
    if pipelined and hasattr(model, "submit"):
        pending = None
        for data in data_loader:
            data = {k: v for k, v in data.items() if k != "return_loss"}
            ticket = model.submit(**data)
            if pending is not None:
                out.extend(_per_sample(model.collect(pending)))
            pending = ticket
        if pending is not None:
            out.extend(_per_sample(model.collect(pending)))
        return out
    for data in data_loader:
        out.extend(_per_sample(model(return_loss=False, **data)))
    return out


def _pack(results, n_rows, K, path_bytes):
    """per-sample result dicts -> uint8 [n_rows, record]; rows past len(results) are marked invalid.  Inputs were validated by
    _local_meta on every rank before any collective."""
    rec = 1 + K * 12 + 24 + 8 + 4 + path_bytes
    buf = np.zeros((n_rows, rec), np.uint8)
    for i, r in enumerate(results):
        p = np.ascontiguousarray(r["preds"], np.float32).reshape(-1)
        path = str(r["image_paths"][0]).encode()
        row, o = buf[i], 1
        row[0] = 1
        row[o:o + K * 12] = p.view(np.uint8); o += K * 12
        row[o:o + 24] = np.ascontiguousarray(r["boxes"], np.float32).reshape(6).view(np.uint8); o += 24
        row[o:o + 8] = np.array([int(r["bbox_ids"][0])], np.int64).view(np.uint8); o += 8
        row[o:o + 4] = np.array([len(path)], np.uint32).view(np.uint8); o += 4
        row[o:o + len(path)] = np.frombuffer(path, np.uint8)
    return buf


def _unpack(row, K):
    o = 1
    preds = row[o:o + K * 12].copy().view(np.float32).reshape(1, K, 3); o += K * 12
    boxes = row[o:o + 24].copy().view(np.float32).reshape(1, 6); o += 24
    bid = int(row[o:o + 8].copy().view(np.int64)[0]); o += 8
    n = int(row[o:o + 4].copy().view(np.uint32)[0]); o += 4
    return {"preds": preds, "boxes": boxes, "bbox_ids": [bid], "image_paths": [bytes(row[o:o + n]).decode()]}


def _local_meta(local_results):
    """(rows, K_max, -K_min, longest path, error flag) of this rank's results: reduced with MAX over the ranks BEFORE anything is
    packed, so that a malformed result raises on every rank together instead of leaving the others blocked in the all_gather."""
    ks, plen, bad = [], 0, 0
    for r in local_results:
        try:
            p = np.asarray(r["preds"], np.float32)
            if p.ndim < 2 or p.shape[-1] != 3 or np.asarray(r["boxes"]).size != 6:
                bad = 1
                continue
            ks.append(int(p.size // 3))
            plen = max(plen, len(str(r["image_paths"][0]).encode()))
            int(r["bbox_ids"][0])
        except Exception:  # noqa: BLE001
            bad = 1
    kmax = max(ks) if ks else 0
    kmin = min(ks) if ks else 1 << 30
    return [len(local_results), kmax, -kmin, plen, bad]


def collect_results(local_results, size, device=None, all_ranks=False, group=1):
    """collect_results_gpu (apis/test.py:154-198) on fixed-size records.  `local_results`: this rank's per-sample dicts in its
    shard order (rank r holds dataset items r, r+W, ...; with group > 1 the items of shard_indices(size, r, W, group)); `size` =
    len(dataset).  Rank 0 (every rank with all_ranks=True) gets the `size` result dicts in dataset order, the others None.  Ranks may
    hold unequal (even zero) numbers of samples."""
    rank, world = rank_world()
    if world == 1:
        return list(local_results)[:size]
    dev = comm_device(device)
    meta = torch.tensor(_local_meta(local_results), dtype=torch.int64, device=dev)
    dist.all_reduce(meta, op=dist.ReduceOp.MAX)          # rows per rank, K, path length, errors: agreed without assuming equal shards
    n_rows, K, neg_kmin, path_bytes, bad = (int(v) for v in meta.tolist())
    if bad:
        raise ValueError("collect_results: a rank holds a malformed result dict (preds [.., K, 3], boxes [6], bbox_ids, image_paths)")
    if -neg_kmin < K:                                     # (a rank without results reports K_min = 2^30 and does not lower it)
        raise ValueError(f"collect_results: results with {-neg_kmin} and {K} keypoints: K must be the same on every rank")
    mine = torch.from_numpy(_pack(local_results, n_rows, K, path_bytes)).to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    if rank != 0 and not all_ranks:
        return None
    if group > 1:                                         # group-wise shards: put every record back at its dataset index
        ordered = [None] * size
        for r in range(world):
            rows_r = parts[r].cpu().numpy()
            for i, gi in enumerate(shard_indices(size, r, world, group)):
                if i < n_rows and rows_r[i][0] and ordered[gi] is None:    # (wrapped-around padding repeats earlier items)
                    ordered[gi] = _unpack(rows_r[i], K)
        missing = [i for i, o in enumerate(ordered) if o is None]
        if missing:
            raise ValueError(f"collect_results: no rank delivered dataset items {missing[:8]}{'...' if len(missing) > 8 else ''}")
        return ordered
    rows = torch.stack(parts, 1).reshape(n_rows * world, -1).cpu().numpy()   # row i*W + r  <-  rank r, local i
    ordered = [_unpack(r, K) for r in rows if r[0]]
    return ordered[:size]                                 # "the dataloader may pad some samples"


def multi_gpu_test(model, data_loader, size=None, tmpdir=None, gpu_collect=True, all_ranks=False, pipelined=False, group=1):
    """apis/test.py:50-91.  `data_loader` yields this rank's shard; `size` defaults to len(data_loader.dataset).  group: the shard
    granularity the loader used (shard_indices(size, rank, world, group); 15 = whole episodes per rank, for the episode cache)."""
    if size is None:
        ds = getattr(data_loader, "dataset", None)
        if ds is None:
            raise ValueError("multi_gpu_test needs `size` (len(dataset)) when the loader has no .dataset")
        size = len(ds)
    return collect_results(single_gpu_test(model, data_loader, pipelined=pipelined), size, all_ranks=all_ranks, group=group)


def gather_predictions(local_preds, n_total, device=None):
    """all_gather of [n_local, K, 3] float32 predictions only; returns [n_total, K, 3] in dataset order on every rank."""
    local_preds = np.asarray(local_preds, np.float32)
    if not dist_on():
        return local_preds[:n_total]
    res = [{"preds": p[None], "boxes": np.zeros((1, 6), np.float32), "bbox_ids": [0], "image_paths": [""]} for p in local_preds]
    out = collect_results(res, n_total, device=device, all_ranks=True)
    return np.concatenate([r["preds"] for r in out], 0) if out else np.zeros((0,) + local_preds.shape[1:], np.float32)
