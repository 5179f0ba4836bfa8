#!/usr/bin/env python
"""Measured conformance of a precision mode against the CPU oracle at scale (cfg2: 256 pairs x 2 weight seeds): argmax flips,
share of keypoints outside 1e-3, error quantiles.  Writes the record that README / DESIGN quote and the gates of
tests/test_gpu_precision_modes.py::test_headline_conformance_at_scale are set from.
    python tools/conformance.py [--backbone fp16 --head mixed --batches 8 --seeds 0,1] --out profiles/r03_conformance_fp16_mixed.json"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_precision_modes as T  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="fp16")
ap.add_argument("--head", default="mixed")
ap.add_argument("--batches", type=int, default=8)
ap.add_argument("--seeds", default="0,1")
ap.add_argument("--config", default="cfg2", help="cfg1 | cfg2 | cfg4 | cfg5 (tests/test_gpu_precision_modes.py CFG)")
ap.add_argument("--outliers", action="store_true", help="weights with planted DINOv2-like activation outliers (synth.add_activation_outliers)")
ap.add_argument("--episodes", type=int, default=0,
                help="N > 0: the reference's evaluation PROTOCOL instead of pairwise batches - N episodes of 15 queries per weight seed streamed through "
                     "ec_forward_episodes in calls sized like bench.py's episode leg (tests/test_gpu_precision_modes.py conformance_episodes)")
ap.add_argument("--also-pairwise", action="store_true", help="with --episodes: the same expanded pairs through ec_forward as well (whose flips are the data's?)")
ap.add_argument("--out", default=None)
a = ap.parse_args()
c = T.CFG[a.config]
if a.episodes:
    per_seed, pooled = T.conformance_episodes(a.episodes, tuple(int(x) for x in a.seeds.split(",")), a.backbone, a.head, a.config, outliers=a.outliers, also_pairwise=a.also_pairwise)
    what = (f"{a.config}: {c['S']}-shot, {c['H']}x{c['H']}, {c['arch']}; {a.episodes} episodes x 15 queries per weight seed through ec_forward_episodes (pipelined), "
            f"{pooled['queries_per_call']} queries per call + the support images of the episodes that start in it")
else:
    per_seed, pooled = T.conformance_at_scale(a.batches, tuple(int(x) for x in a.seeds.split(",")), a.backbone, a.head, a.config, outliers=a.outliers)
    what = f"{a.config}: {c['S']}-shot, batch {c['bs']}, {c['H']}x{c['H']}, {c['arch']}; {a.batches} disjoint batches per weight seed"
rec = dict(config=what + ("; weights with planted activation outliers" if a.outliers else ""), backbone=a.backbone, head=a.head, oracle="oracle/edgecape_oracle.py (fp32 CPU)",
           tolerance="1e-3 abs on output_kpts of valid keypoints", per_weight_seed=per_seed, pooled=pooled,
           library_source_hash=__import__("edgecape_amd.build", fromlist=["source_hash"]).source_hash())   # bench.py quotes the record only for this library
print(json.dumps(rec, indent=1))
if a.out:
    with open(a.out, "w") as f:
        json.dump(rec, f, indent=1)
