#!/usr/bin/env python3
"""The one-launch form of the small-batch gate (`small_fused_kernel`, prober_small.hip) is in the `make diag` build only
since round 5 (it is measured slower than the three launches: profiles/r04c_latency.txt).  This checker is the pytest
case that covered it while it shipped; run it on a GPU box against the diag library:

    make -C probing-rag_amd/csrc diag
    PRAG_LIB=probing-rag_amd/lib/libprag_diag.so python tests/diag/one_launch_gate.py

It must equal the default three launches bit for bit, call after call (the arrival counters return to zero inside the
launch), per layer, and replayed from a captured graph.  Reference call shape: exp_rag.py:381-389, 406-415."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_np as onp          # noqa: E402
from tests.golden import cases                # noqa: E402

TOL = 1e-4


def _ensemble(case, weights):
    import probing_rag_amd as pra
    states = [cases.synth_state(case["wseed"] + l, case["d"]) for l in range(case["L"])]
    ens = pra.HipProberEnsemble(case["L"], case["d"], 2, weights=weights)
    for l, st in enumerate(states):
        ens.load_layer(l, st)
    return ens, states


def check(weights, B):
    import torch
    case = dict(cases.PROBER_CASES[1], B=B)
    x = torch.from_numpy(cases.case_x(case)).cuda()
    os.environ.pop("PRAG_PROBER_SMALL", None)
    ens3, _ = _ensemble(case, weights)
    os.environ["PRAG_PROBER_SMALL"] = "3"        # read when a handle is created
    ens, _ = _ensemble(case, weights)
    os.environ.pop("PRAG_PROBER_SMALL")
    want = [t.cpu().numpy() for t in ens3.gate(x, 1, 0.5)]
    for _ in range(5):
        got = [t.cpu().numpy() for t in ens.gate(x, 1, 0.5)]
        for g, w_ in zip(got, want):
            assert np.array_equal(g, w_)
    eff = np.stack([onp.prober_forward(ens.effective_state_dict(l), cases.case_x(case)[l]) for l in range(ens.n_layers)])
    np.testing.assert_allclose(got[0], eff, atol=TOL, rtol=0)
    one = ens.probers[2](x[2]).cpu().numpy()
    assert np.array_equal(one, want[0][2])
    out = (torch.empty((case["L"], B, 2), device="cuda"), torch.empty((B, 2), device="cuda"),
           torch.empty((B,), dtype=torch.int32, device="cuda"))
    ens.gate(x, 1, 0.5, out=out)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ens.gate(x, 1, 0.5, out=out)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            ens.gate(x, 1, 0.5, out=out)
    for _ in range(3):
        for t in out:
            t.zero_()
        g.replay()
        torch.cuda.synchronize()
        for t, w_ in zip(out, want):
            assert np.array_equal(t.cpu().numpy(), w_)


if __name__ == "__main__":
    if "diag" not in os.environ.get("PRAG_LIB", ""):
        raise SystemExit("set PRAG_LIB to libprag_diag.so: the one-launch kernel is not in libprag.so")
    for weights, B in (("f32", 1), ("f32", 4), ("f16", 2), ("f32", 3)):
        check(weights, B)
        print("one-launch gate == three launches:", weights, B)
