#!/usr/bin/env python3
"""Timing-only ablations of the int8 tiles (make -C probing-rag_amd/csrc diag; PRAG_LIB=.../libprag_diag.so,
PRAG_MM_ABLATE=0|8|1|9|2|4: full | no filter | no MFMAs | neither | no LDS-DMA | no fragment reads): duration of the
LARGEST segment launch of a 1000-query search (results of ablated runs are wrong and every query then goes through
the fallbacks - only the profiled launch is of interest).  python tools/mm8_ablate.py [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd.synth import synth_rows
N = int(sys.argv[1]) if len(sys.argv) > 1 else 21_000_000
ix = pra.HipFlatIndex(768, "cos", "f16", capacity=N)
ix.set_shadow(2)
ix.add_synthetic(42, 0, N)
Q = torch.from_numpy(synth_rows(7, 0, 1000, 768)).cuda()
ix.profile(64)
for _ in range(2):
    ix.search(Q, 10)
torch.cuda.synchronize()
seg = np.asarray(ix.profile_read())
print(f"PRAG_MM_ABLATE={os.environ.get('PRAG_MM_ABLATE', '0')} N={N}: profiled launches (largest segment of each tiled pass) {np.round(seg, 3).tolist()} ms; "
      f"int8 tier failed {ix.last_tiled8()}", flush=True)
