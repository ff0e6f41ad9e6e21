#!/usr/bin/env python3
"""Differential fuzz of the fused prober ensemble + gate against eager PyTorch float32 on the GPU (diagnostic):
random batch sizes (tile remainders, the small-batch path), layer counts, activation dtypes, both weight modes,
activation scales.  f32-parity weights: logits within 1e-4 of the float32 modules (the contract); fp16 weights:
within 2e-2 (weight rounding) and decisions equal away from the threshold.  python tools/fuzz_prober.py [s] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.nn as nn
import probing_rag_amd as pra
from probing_rag_amd.synth import random_prober_state


class Ref(nn.Module):                      # utils.py:29-57: Linear -> SiLU -> LayerNorm, eval mode
    def __init__(s, d, h=512, c=2):
        super().__init__()
        s.layer_norm_input = nn.LayerNorm(d); s.fc1 = nn.Linear(d, h); s.layer_norm1 = nn.LayerNorm(h)
        s.fc2 = nn.Linear(h, h); s.layer_norm2 = nn.LayerNorm(h); s.fc3 = nn.Linear(h, c); s.silu = nn.SiLU()

    def forward(s, x):
        x = s.layer_norm_input(x)
        x = s.layer_norm1(s.silu(s.fc1(x)))
        x = s.layer_norm2(s.silu(s.fc2(x)))
        return s.fc3(x)


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
d = 2048
pool = []
for L in (1, 3, 6):
    states = [random_prober_state(int(rng.integers(1 << 20)), d) for _ in range(L)]
    refs = []
    for st in states:
        m = Ref(d).cuda().eval()
        m.load_state_dict({k: torch.as_tensor(v) for k, v in st.items()})
        refs.append(m)
    ens = {}
    for wm in ("f32", "f16"):
        e = pra.HipProberEnsemble(L, d, 2, weights=wm)
        for l, st in enumerate(states):
            e.load_layer(l, st)
        ens[wm] = e
    pool.append((L, refs, ens))
t0, n_case, worst = time.time(), 0, {"f32": 0.0, "f16": 0.0}
torch.backends.cuda.matmul.allow_tf32 = False
while time.time() - t0 < budget:
    L, refs, ens = pool[int(rng.integers(len(pool)))]
    B = int(rng.choice([1, 2, 3, 4, 5, 31, 32, 33, 64, 65, 96, 127, 128, 129, 500, 1000, 4095, 4096, 4097]))
    dt = [torch.float32, torch.float16, torch.bfloat16][int(rng.integers(3))]
    scale = float(rng.choice([0.05, 1.0, 30.0]))
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    x = (torch.randn((L, B, d), generator=g, device="cuda") * scale + float(rng.normal()) * scale).to(dt)
    with torch.no_grad():
        want = torch.stack([refs[l](x[l].float()) for l in range(L)])
        p = torch.softmax(want, -1).sum(0)
    theta = float(rng.choice([-1.0, 0.0, 0.5]))
    for wm in ("f32", "f16"):
        logits, ps, dec = ens[wm].gate(x, ablation=0, threshold=theta)
        err = float((logits - want).abs().max())
        worst[wm] = max(worst[wm], err)
        tol = 1e-4 if wm == "f32" and dt == torch.float32 else 2e-2
        margin = (p[:, 0] + theta - p[:, 1]).abs()
        want_dec = (~(p[:, 0] + theta < p[:, 1])).int()
        dec_ok = bool(((dec == want_dec) | (margin < 4 * tol)).all())
        if not (err <= tol and dec_ok and torch.isfinite(logits).all()):
            print(f"MISMATCH L={L} B={B} {dt} scale={scale} weights={wm}: max |dlogit| {err:.3e} (tol {tol}), decisions ok {dec_ok}", flush=True)
            sys.exit(1)
    n_case += 1
print(f"fuzz ok: {n_case} cases in {time.time() - t0:.0f} s; worst |dlogit| f32-weights {worst['f32']:.2e}, f16-weights {worst['f16']:.2e}", flush=True)
