#!/usr/bin/env python3
"""Latency of one prober training step (train.py:210-220): HIP trainer vs PyTorch-ROCm eager (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from probing_rag_amd.synth import synth_rows
from tests.golden import cases


def timeit(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

# ---- prober training step (train.py:210-220): HIP trainer vs plain PyTorch on the same GPU ----
F = torch.nn.functional
for B in (8, 64):
    d = 2048
    st = cases.synth_state(900, d)
    x = torch.from_numpy(synth_rows(901, 0, B, d)).cuda()
    y = torch.from_numpy(np.arange(B) % 2).cuda()
    tr = pra.HipProberTrainer(d, 2, seed=1).load_state_dict(st)
    t_hip = timeit(lambda: tr.step(x, y), n=200)

    class Probe(torch.nn.Module):          # structure of utils.py:29-57
        def __init__(self):
            super().__init__()
            self.layer_norm_input = torch.nn.LayerNorm(d); self.fc1 = torch.nn.Linear(d, 512)
            self.fc2 = torch.nn.Linear(512, 512); self.fc3 = torch.nn.Linear(512, 2)
            self.layer_norm1 = torch.nn.LayerNorm(512); self.layer_norm2 = torch.nn.LayerNorm(512)
            self.dropout = torch.nn.Dropout(0.1)

        def forward(self, v):
            v = self.layer_norm_input(v)
            v = self.dropout(self.layer_norm1(F.silu(self.fc1(v))))
            v = self.dropout(self.layer_norm2(F.silu(self.fc2(v))))
            return self.fc3(v)

    m = Probe().cuda().train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
    sch = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.995)

    def torch_step():
        loss = F.cross_entropy(torch.softmax(m(x), dim=-1), y)
        loss.backward(); opt.step(); sch.step(); opt.zero_grad()

    t_torch = timeit(torch_step, n=100)
    print(f"train step B={B}: HIP trainer {t_hip:7.1f} us | PyTorch-ROCm eager (same structure) {t_torch:7.1f} us", flush=True)
