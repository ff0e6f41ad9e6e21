#!/usr/bin/env python3
"""The contiguous-cluster corpus of bench.py's `clustered_...` variant at several batch sizes (diagnostic)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import probing_rag_amd as pra
for B in (64, 32, 1, 128):
    r = bench.clustered_variant(torch, pra, 768, 10, B=B)
    print(f"B={B}: direct {r['rows_scanned_directly']['ms_per_search']:.3f} ms | two-level {r['two_level']['ms_per_search']:.3f} ms, "
          f"fallbacks {r['two_level']['exact_fallbacks_last_search']}, ids identical {r['ids_identical']}", flush=True)
