#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own code.

Run in the build container only (``/root/reference`` does not exist on the
GPU box):

    python tests/golden/gen_golden.py

It imports ``/root/reference/utils.py`` (with ``faiss`` / ``spacy`` stubbed in
``sys.modules`` — they are only needed by code that is not on the hot path,
SURVEY.md §8c) and evaluates the genuine ``ImprovedProbe`` (utils.py:29-57),
``return_prober_logit_gemma_2b`` (utils.py:389-390), ``_method_2_util``
(utils.py:181-189), ``return_acc`` (utils.py:158-170) and ``method_2_train``
(utils.py:191-197, with torch.optim.AdamW / ExponentialLR as in train.py:131-135)
on deterministic inputs, and the retrieval wrappers ``batch_topk_sim`` / ``find_topk_sim``
(utils.py:374-380) against a recording index (what they hand to ``index.search``: array type, dtype,
shape, ``k`` as a keyword) -> ``topk_golden.npz`` (``--only topk`` regenerates just that file).  Weights and inputs come from the counter-based generator
``oracle_np.synth_rows`` so that the fixtures only need to store seeds,
shapes and the reference's OUTPUTS (a few KB), never weights or source.

The gate expressions of exp_rag.py:407-415 live inside ``main()`` and cannot
be imported; they are evaluated here with the same torch calls
(``torch.nn.Softmax(dim=1)``, float32 accumulation in layer order, ``.item()``
compare) on the reference's logits.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import oracle_np as onp  # noqa: E402
from tests.golden import cases  # noqa: E402


def import_reference_utils():
    ref = "/root/reference"
    if not os.path.isdir(ref):
        raise SystemExit("reference tree not present; golden vectors can only be "
                         "generated in the build container")
    sys.modules.setdefault("faiss", types.ModuleType("faiss"))
    sp = types.ModuleType("spacy")
    sp.load = lambda name: None
    sys.modules.setdefault("spacy", sp)
    sys.path.insert(0, ref)
    import utils as ref_utils  # the reference module
    return ref_utils


def gen_topk(ru):
    """utils.py:374-380 run as they are: the reference's own `batch_topk_sim` and `find_topk_sim` call a
    recording index (over the oracle's flat search - faiss is not in the image) with a stub
    encoder.  Stored: the call descriptions and the D / I they return."""
    case = cases.TOPK_CASE
    X, Q = cases.topk_inputs(case)
    out = {}
    enc = cases.StubEncoder(Q)
    rec = cases.RecordingIndex(cases.OracleIndex(X))
    D, I = ru.batch_topk_sim(enc, ["q%d" % i for i in range(case["B"])], rec, case["k"])   # exp_rag.py:432 call shape
    out["batch/D"], out["batch/I"] = np.asarray(D, np.float32), np.asarray(I, np.int64)
    D1, I1 = ru.find_topk_sim(enc, "one question", rec, case["k"])
    out["find/D"], out["find/I"] = np.asarray(D1, np.float32), np.asarray(I1, np.int64)
    out["index_calls"] = np.array(rec.calls)
    out["encode_calls"] = np.array(enc.calls)
    path = os.path.join(HERE, "topk_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", rec.calls, enc.calls)


def main():
    import torch
    torch.set_num_threads(4)
    ru = import_reference_utils()
    gen_topk(ru)
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "topk":
        return
    out = {}

    # ---- known-answer: parameter count (exp_parameter_check.py:52) ---------
    p = ru.ImprovedProbe(input_size=2048, output_size=2)
    out["param_count"] = np.int64(sum(t.numel() for t in p.parameters()))
    out["state_keys"] = np.array(list(p.state_dict().keys()))

    softmax_f = torch.nn.Softmax(dim=1)  # exp_rag.py:393

    for case in cases.PROBER_CASES:
        name, d, L, B, sigma = case["name"], case["d"], case["L"], case["B"], case["sigma"]
        states = [cases.synth_state(case["wseed"] + l, d) for l in range(L)]
        x = cases.case_x(case)
        probers = []
        for st in states:
            m = ru.ImprovedProbe(input_size=d, output_size=2)
            m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()})
            m.eval()  # utils.py:329
            probers.append(m)

        # utils.py:389-390 : list of per-layer [B,2] CPU tensors
        cfgs = list(range(L))
        with torch.no_grad():
            logits = ru.return_prober_logit_gemma_2b(
                lambda cfg, prober: prober(torch.from_numpy(x[cfg])), cfgs, probers)
        lg = torch.stack(logits).numpy()
        out[f"{name}/logits"] = lg.astype(np.float32)

        # exp_rag.py:407-415 for every (theta, ablation)
        for ab in cases.ABLATIONS:
            if ab >= L:
                continue
            acc = torch.zeros_like(logits[0])
            for num in range(ab, len(logits)):
                acc += softmax_f(logits[num])
            out[f"{name}/probsum_ab{ab}"] = acc.numpy().astype(np.float32)
            for th in cases.THETAS:
                dec = np.array([0 if acc[b, 0].item() + th < acc[b, 1].item() else 1
                                for b in range(B)], dtype=np.int32)
                out[f"{name}/decision_ab{ab}_th{th}"] = dec

    # ---- train/eval-time forward: ragged mean pool + double-softmax CE -----
    for case in cases.POOL_CASES:
        name, d, B, T = case["name"], case["d"], case["B"], case["T"]
        st = cases.synth_state(case["wseed"], d)
        m = ru.ImprovedProbe(input_size=d, output_size=2)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()})
        m.eval()
        acts, pred_lens, labels = cases.synth_pool_inputs(case)
        args = types.SimpleNamespace(device="cpu")
        # utils.py:456 (`from scipy.special import softmax`, baseline-RAG code)
        # shadows the nn.Softmax(dim=-1) bound at utils.py:15 that make_loss
        # (utils.py:130) was written against; train.py:141-151 — the copy that
        # actually runs — uses nn.Softmax(dim=-1).  Restore that binding.
        ru.softmax = torch.nn.Softmax(dim=-1)
        with torch.no_grad():
            loss, probs = ru._method_2_util(m, torch.from_numpy(acts), torch.from_numpy(labels),
                                            torch.from_numpy(pred_lens), args)
        out[f"{name}/probs"] = probs.numpy().astype(np.float32)
        out[f"{name}/loss"] = np.float32(loss.item())
        out[f"{name}/acc"] = np.float64(ru.return_acc(probs, torch.from_numpy(labels)))
        # the other two eval methods on the same inputs: method_1_eval (`each_token`, utils.py:175-179) and
        # method_3_eval (`last_token`, utils.py:222-226) - the reference functions themselves
        with torch.no_grad():
            a1, n1, l1 = ru.method_1_eval(m, torch.from_numpy(acts), torch.from_numpy(labels),
                                          torch.from_numpy(pred_lens), args)
            _, p1 = ru.make_loss(m, *ru._input_tensor_method1(torch.from_numpy(acts), torch.from_numpy(labels),
                                                              torch.from_numpy(pred_lens), args))
            a3, n3, l3 = ru.method_3_eval(m, torch.from_numpy(acts), torch.from_numpy(labels),
                                          torch.from_numpy(pred_lens), args)
            _, p3 = ru._method_3_util(m, torch.from_numpy(acts), torch.from_numpy(labels),
                                      torch.from_numpy(pred_lens), args)
        out[f"{name}/m1_acc"], out[f"{name}/m1_n"], out[f"{name}/m1_loss"] = np.float64(a1), np.int64(n1), np.float32(l1.item())
        out[f"{name}/m1_probs"] = p1.numpy().astype(np.float32)
        out[f"{name}/m3_acc"], out[f"{name}/m3_n"], out[f"{name}/m3_loss"] = np.float64(a3), np.int64(n3), np.float32(l3.item())
        out[f"{name}/m3_probs"] = p3.numpy().astype(np.float32)
        # inference-time sum pool of the same tokens (exp_rag.py:385-387)
        with torch.no_grad():
            sums = torch.stack([torch.from_numpy(acts[i, T - int(n):, :]).sum(dim=0)
                                for i, n in enumerate(pred_lens)])
            out[f"{name}/sum_logits"] = m(sums).numpy().astype(np.float32)

    # ---- training step: the reference's method_2_train (utils.py:191-197) with the optimiser and
    #      scheduler of train.py:131-135 (AdamW(lr), ExponentialLR(gamma=0.995)).  The only change
    #      to the reference module is the SOURCE of the dropout masks (torch's global RNG stream
    #      cannot be reproduced by another implementation): `dropout` is replaced by a module that
    #      applies oracle_np.dropout_keep masks with torch's own scaling x * keep / (1 - p).
    from torch.optim import AdamW
    from torch.optim.lr_scheduler import ExponentialLR

    class MaskDropout(torch.nn.Module):
        def __init__(self, seed, p):
            super().__init__()
            self.seed, self.p, self.step, self.site = seed, p, 1, 0

        def forward(self, x):
            if self.p == 0.0:
                return x
            keep = onp.dropout_keep(self.seed, self.step, self.site, x.shape[0], x.shape[1], self.p)
            self.site += 1
            return x * torch.from_numpy(keep.astype(np.float32)) * np.float32(1.0 / (1.0 - self.p))

    for case in cases.TRAIN_CASES + cases.TRAIN_METHOD_CASES:
        name, d = case["name"], case["d"]
        method = case.get("method", "tokens_mean")
        # train.py:268-279: each_token -> method_1_train, tokens_mean -> method_2_train, last_token -> method_3_train
        train_fn = {"each_token": ru.method_1_train, "tokens_mean": ru.method_2_train, "last_token": ru.method_3_train}[method]

        def raw_loss(m_, a_, l_, p_, args_):
            if method == "each_token":
                return ru.make_loss(m_, *ru._input_tensor_method1(a_, l_, p_, args_))[0]
            if method == "last_token":
                return ru._method_3_util(m_, a_, l_, p_, args_)[0]
            return ru._method_2_util(m_, a_, l_, p_, args_)[0]
        pdrop = case.get("dropout_p", 0.1)
        st = cases.synth_state(case["wseed"], d)
        m = ru.ImprovedProbe(input_size=d, output_size=2)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()})
        assert m.dropout.p == 0.1                                   # utils.py:39
        m.dropout = MaskDropout(case["seed"], pdrop)
        m.train()                                                   # train.py:248
        optim = AdamW(m.parameters(), lr=onp.ADAMW_DEFAULTS["lr"])  # train.py:131 (--lr 1e-4 in train_prober.sh)
        sched = ExponentialLR(optim, gamma=0.995)                   # train.py:134
        args = types.SimpleNamespace(device="cpu")
        ru.softmax = torch.nn.Softmax(dim=-1)
        losses, lrs = [], []
        for t in range(1, case["steps"] + 1):
            acts, pred_lens, labels = cases.synth_train_batch(case, t)
            m.dropout.step, m.dropout.site = t, 0
            lr_used = optim.param_groups[0]["lr"]
            # method_2_train returns (round(loss, 4), lr after scheduler.step()); recompute the raw
            # loss of the same forward for a tighter comparison
            m.dropout.site = 0
            with torch.no_grad():
                raw = raw_loss(m, torch.from_numpy(acts), torch.from_numpy(labels), torch.from_numpy(pred_lens), args)
            m.dropout.site = 0
            lrnd, lr_next = train_fn(m, optim, sched, torch.from_numpy(acts), torch.from_numpy(labels),
                                     torch.from_numpy(pred_lens), args)
            assert abs(lrnd - round(raw.item(), 4)) < 1e-9
            losses.append(raw.item())
            lrs.append(lr_used)
        out[f"{name}/losses"] = np.array(losses, dtype=np.float64)
        out[f"{name}/lrs"] = np.array(lrs, dtype=np.float64)
        sd = {k: v.detach().numpy() for k, v in m.state_dict().items()}
        for k, v in sd.items():
            if v.size > 5000:
                out[f"{name}/final/{k}/sample"] = v.reshape(-1)[::cases.TRAIN_SAMPLE_STRIDE].astype(np.float32)
                out[f"{name}/final/{k}/sum"] = np.float64(v.astype(np.float64).sum())
                out[f"{name}/final/{k}/delta_l2"] = np.float64(np.sqrt(((v.astype(np.float64) - st[k]) ** 2).sum()))
            else:
                out[f"{name}/final/{k}"] = v.astype(np.float32)

    path = os.path.join(HERE, "prober_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
