#!/usr/bin/env python3
"""Two-level search against the direct scan on embedding-shaped corpora (bench.embedding_variant) at a few sizes /
metrics: `python tools/embedding_probe.py [rows ...]` (default 1 M and 4 M), one JSON line per case."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import probing_rag_amd as pra
import bench

sizes = [int(a) for a in sys.argv[1:]] or [1 << 20, 1 << 22]
for n in sizes:
    for metric, store in (("cos", "f16"), ("l2", "f16"), ("l2", "f32")):
        for B in (64, 1):
            for kw in ({}, {"n_outlier": 4, "outlier_ratio": (10.0, 14.0)}, {"n_outlier": 0}):
                rec = bench.embedding_variant(torch, pra, bench.D_EMB, 10, n, metric, store, B=B, **kw)
                rec["structure"] = kw or "default"
                print(json.dumps(rec), flush=True)
