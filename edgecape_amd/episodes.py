"""Episode construction of the reference's evaluation (EdgeCape/datasets/datasets/mp100/test_dataset.py:86-99).

The MP-100 test protocol draws, per category and episode, `num_shots` support annotations and `num_queries` query annotations and
evaluates every (support set, query) pair: `num_queries` consecutive pairs share ONE support set.  That is what the support-side
cache of the HIP path exploits (`ec_support_encode` once per episode, `ec_forward_cached` for its queries).

`make_paired_samples` reproduces the reference's pair list exactly (same RNG protocol: `random.seed(1)`, one `random.sample` per
category and episode); `group_episodes` turns a pair list into (unique support sets, episode index of every pair).
"""
import random

import numpy as np


def make_paired_samples(cat2obj, valid_class_ids, num_shots=1, num_queries=15, num_episodes=100, seed=1):
    """cat2obj: {category id: [annotation ids]}.  Returns int array [n_pairs, num_shots + 1]: support ids then the query id."""
    rng = random.Random(seed)          # the reference seeds the module-level generator; a private one draws the same sequence
    pairs = []
    for cls in valid_class_ids:
        for _ in range(num_episodes):
            drawn = rng.sample(cat2obj[cls], num_shots + num_queries)
            support, queries = drawn[:num_shots], drawn[num_shots:]
            pairs.extend(support + [q] for q in queries)
    return np.array(pairs)


def group_episodes(paired_samples):
    """pairs [n, S + 1] -> (support_sets [n_ep, S], episode_of_pair [n] int32): consecutive pairs with the same support ids
    form one episode (the order the reference's sequential sampler visits them in)."""
    ps = np.asarray(paired_samples)
    if ps.size == 0:
        return ps.reshape(0, max(ps.shape[-1] - 1, 0)), np.zeros(0, np.int32)
    sup = ps[:, :-1]
    new = np.ones(len(ps), bool)
    new[1:] = (sup[1:] != sup[:-1]).any(axis=1)
    ep = np.cumsum(new).astype(np.int32) - 1
    return sup[new], ep


def stream_schedule(episode_of_pair, batch, capacity):
    """Cut the reference's pair order (episode_of_pair from group_episodes: non-decreasing episode ids) into the calls of
    `ec_forward_episodes` (include/edgecape_hip.h): every call takes the next `batch` pairs' queries and ENCODES the episodes whose
    first query lies in it; an episode occupies a cache slot from that call until the call holding its last query has been issued.
    Returns a list of dicts: `queries` (pair indices of the call), `slot_of_query` (int32), `new_episodes` (episode ids to encode in
    this call), `new_slots` (int32, their cache slots).  `capacity` slots must cover the episodes alive across one call boundary
    (ValueError otherwise): ceil(batch / queries-per-episode) + 1 always does."""
    ep = np.asarray(episode_of_pair, np.int64)
    if ep.size and (np.diff(ep) < 0).any():
        raise ValueError("episode_of_pair must be non-decreasing (the reference's sequential pair order)")
    last = {}
    for i, e in enumerate(ep):
        last[int(e)] = i
    free = list(range(capacity - 1, -1, -1))
    slot_of, calls = {}, []
    for q0 in range(0, len(ep), batch):
        idx = np.arange(q0, min(q0 + batch, len(ep)))
        new = []
        for e in dict.fromkeys(int(x) for x in ep[idx]):
            if e not in slot_of:
                if not free:
                    raise ValueError(f"capacity {capacity} too small for batch {batch}: more episodes alive than cache slots")
                slot_of[e] = free.pop()
                new.append(e)
        calls.append(dict(queries=idx, slot_of_query=np.array([slot_of[int(e)] for e in ep[idx]], np.int32),
                          new_episodes=new, new_slots=np.array([slot_of[e] for e in new], np.int32)))
        for e in [e for e in slot_of if last[e] < q0 + batch]:      # their last query is in this call: the slot is free for the next one
            free.append(slot_of.pop(e))
    return calls
