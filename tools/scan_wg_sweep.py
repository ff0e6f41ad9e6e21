"""Two-level search of 64 queries over N x 768 fp16 rows with the scan on fewer workgroups than CUs (set_scan_workgroups):
what the scan loses when CUs are set aside for concurrent work.  usage: scan_wg_sweep.py [rows]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import probing_rag_amd as pra  # noqa: E402
from probing_rag_amd.synth import synth_rows  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 21_000_000
STORE = os.environ.get("SCAN_STORE", "f16")
ix = pra.HipFlatIndex(768, os.environ.get("SCAN_METRIC", "cos"), STORE, capacity=N)
ix.add_synthetic(42, 0, N)
ix.set_shadow(int(os.environ.get("SCAN_SHADOW", 1)))
ix.prepare()
NQ = int(os.environ.get("SCAN_Q", 64))
q = torch.from_numpy(synth_rows(7, 0, NQ, 768)).cuda()
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
for rep in range(int(os.environ.get("SCAN_REPS", 2))):
    for wg in [int(x) for x in os.environ.get("SCAN_WGS", "0,248,240,224,0").split(",")]:
        ix.set_scan_workgroups(wg)
        for _ in range(3):
            ix.search(q, 10)
        torch.cuda.synchronize()
        ix.profile(64)
        for _ in range(40):
            ix.search(q, 10)
        torch.cuda.synchronize()
        ms = np.asarray(ix.profile_read())
        ix.profile(0)
        print(f"{STORE} shadow {os.environ.get('SCAN_SHADOW', 1)} rows {N} queries {NQ} scan workgroups {wg or n_cu:4d}: scan kernel(s) {ms.sum() / 40:.4f} ms", flush=True)
