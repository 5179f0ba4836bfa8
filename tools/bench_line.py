#!/usr/bin/env python
"""Print the key fields of a bench.py JSON line read from stdin."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d["roofline"]
    print(sys.argv[1] if len(sys.argv) > 1 else "", d["value"], "img/s", d["ms_per_step"], "ms/step  qkv", r["achieved"], "TF", r["frac"],
          " episode:", (d.get("episode_cached") or {}).get("value"), " parity:", d.get("parity_sample"))
