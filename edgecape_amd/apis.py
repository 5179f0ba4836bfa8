"""Evaluation loops with the reference's contract (EdgeCape/apis/test.py).

`single_gpu_test` mirrors apis/test.py:14-47 (per-sample split of the batch result).  `multi_gpu_test`
replaces the reference's pickled-results all_gather (apis/test.py:154-198) by ONE fixed-size
all_gather of float32 predictions over RCCL (backend "nccl" on ROCm) — pairs are independent, nothing
else crosses GPUs.  Works with gloo on CPU tensors as well (used by the CPU tests).
"""
import numpy as np
import torch
import torch.distributed as dist


def _batch_size(data):
    return len(next(iter(data.values()))[0])


def single_gpu_test(model, data_loader):
    model.eval()
    results = []
    for data in data_loader:
        result = model(return_loss=False, **data)
        batch_size = _batch_size(data)
        if "preds" in result:
            for i in range(batch_size):
                results.append({
                    "preds": result["preds"][i][None],
                    "boxes": result["boxes"][i][None],
                    "bbox_ids": [result["bbox_ids"][i]],
                    "image_paths": [result["image_paths"][i]],
                })
    return results


def shard_indices(n_total, rank, world_size):
    """DistributedSampler(shuffle=False) semantics: pad to equal length, rank r takes r, r+W, ..."""
    per = (n_total + world_size - 1) // world_size
    idx = list(range(n_total))
    pad = per * world_size - n_total
    if pad > 0 and n_total > 0:
        idx += (idx * ((pad + n_total - 1) // n_total))[:pad]   # wraps more than once when n_total < world_size
    return idx[rank::world_size]


def gather_predictions(local_preds, n_total, device=None):
    """all_gather of [n_local, K, 3] float32 predictions; returns [n_total, K, 3] in dataset order
    (interleave by rank, truncate the sampler padding — apis/test.py:187-196)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return np.asarray(local_preds)[:n_total]
    W = dist.get_world_size()
    t = torch.as_tensor(np.ascontiguousarray(local_preds), dtype=torch.float32)
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = t.to(device)
    parts = [torch.empty_like(t) for _ in range(W)]
    dist.all_gather(parts, t)
    stacked = torch.stack(parts, 1).reshape(-1, *t.shape[1:])   # index i*W + r  <-  rank r, local i
    return stacked[:n_total].cpu().numpy()


def multi_gpu_test(model, data_loader, n_total=None, gpu_collect=True):
    """Each rank runs its shard (data_loader yields only its pairs); rank-ordered results are gathered."""
    local = single_gpu_test(model, data_loader)
    if not local:
        preds = np.zeros((0, 1, 3), np.float32)
    else:
        preds = np.concatenate([r["preds"] for r in local], 0)
    if n_total is None:
        n_total = len(local) * (dist.get_world_size() if dist.is_initialized() else 1)
    return gather_predictions(preds, n_total)


def allreduce_counts(counts, device=None):
    """Sum a small float64 vector (PCK hit/count) over ranks."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return np.asarray(counts, np.float64)
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.as_tensor(np.asarray(counts, np.float64)).to(device)
    dist.all_reduce(t)
    return t.cpu().numpy()
