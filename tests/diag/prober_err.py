#!/usr/bin/env python3
"""Print max |logit error| of every prober mode against the float64 oracle on
identical operands (diagnostic; GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probing_rag_amd as pra
from oracle import oracle_np as onp
from tests.golden import cases

for case in cases.PROBER_CASES + [dict(name="c2s", d=2048, L=6, B=512, sigma=1.0, wseed=100, xseed=999)]:
    x = cases.case_x(case)
    for weights in ("f32", "f16"):
        ens = pra.HipProberEnsemble(case["L"], case["d"], 2, weights=weights)
        for l in range(case["L"]):
            ens.load_layer(l, cases.synth_state(case["wseed"] + l, case["d"]))
        eff = [ens.effective_state_dict(l) for l in range(case["L"])]
        for xd in ("f32", "f16"):
            if xd == "f16" and np.abs(x).max() > 6e4:
                continue
            xt = torch.from_numpy(x).cuda()
            if xd == "f16":
                xt = xt.half()
            seen = xt.float().cpu().numpy()
            got = ens.forward(xt).cpu().numpy()
            want = np.stack([onp.prober_forward(eff[l], seen[l]) for l in range(case["L"])])
            err = np.abs(got - want)
            print(f"{case['name']:14s} w={weights} x={xd}  max={err.max():.2e}  mean={err.mean():.2e}")
