#!/usr/bin/env python3
"""Diagnostic behind exact_mfma_kernel's design (round 6): 37 queries, each the exact copy of a row that occurs 40 times, k = 40,
tagged ids (23 bits of the float32 residual of every float64 score).  With the float64 MFMA's sum taken AS the score, the copies
of a row came back in two groups - rows 12-15 of a 16-row tile (the accumulator's fourth register) carry a sum that differs in
the last bit - so ties broke by position, not by id.  The kernel therefore only selects with the MFMA value and re-scores what
may enter a list in the one-query kernel's order; this script now shows every copy in id order."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from oracle import oracle_c, oracle_np as onp
import probing_rag_amd as pra
os.environ["PRAG_EXACT_GROUP"] = "1"
for d, store, metric in ((128, "f32", 1), (256, "f16", 1), (512, "f16", 1)):
    N, B, k = 20_011, 37, 40
    X = onp.synth_rows(71, 0, N, d)
    rng = np.random.default_rng(d)
    dup_rows = []
    for i in range(B):
        dups = np.sort(rng.choice(N, 40, replace=False))
        X[dups] = X[dups[0]]
        dup_rows.append(dups)
    Q = np.stack([X[dr[0]] for dr in dup_rows]).astype(np.float32)
    ix = pra.HipFlatIndex(d, metric, store)
    ix.add(X)
    D, I = ix.search(torch.from_numpy(Q).cuda(), k, tagged=True)
    I = I.cpu().numpy(); D = D.cpu().numpy()
    ids = I & ((1 << 40) - 1); res = I >> 40
    for b in (0, 1, 2):
        got = ids[b].tolist(); want = dup_rows[b].tolist()
        pos = [got.index(w) if w in got else -1 for w in want]
        print("   dup ids", want[:14]); print("   got ids", got[:14]); print("   positions of the dups in the result", pos)
        print(f"d={d} {store} q{b}: rows%128 {(ids[b] % 128)[:12].tolist()} residual tags {res[b][:12].tolist()} distinct tags {len(set(res[b].tolist()))}")
    ix.close()
