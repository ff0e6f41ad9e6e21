"""GPU parity of the HIP prober training step (through the C ABI) against
  * the golden vectors of the reference's own method_2_train (utils.py:191-197) with
    torch.optim.AdamW + ExponentialLR (train.py:131-135),
  * the float64 oracle on other shapes,
  * plain PyTorch fp32 autograd + torch.optim.AdamW on the same masks (the torch reference of a
    floating-point kernel).
Tolerances: losses 2e-6; parameters 5e-6 absolute (updates are ~lr = 1e-4 per step; Adam's
m / sqrt(v) amplifies fp32 rounding where |grad| ~ eps)."""
import numpy as np
import pytest

from oracle import oracle_np as onp
from tests.golden import cases

pytestmark = pytest.mark.gpu


def _params_close(got, want):
    """5e-6 everywhere except a vanishing fraction of elements whose gradient is ~eps (1e-8): there
    Adam's m / (sqrt(v) + eps) turns fp32 rounding of the gradient into up to ~1e-5 of the update."""
    err = np.abs(np.asarray(got, dtype=np.float64) - np.asarray(want, dtype=np.float64))
    assert err.max() < 3e-5, err.max()
    assert (err > 5e-6).mean() < 1e-5, (err > 5e-6).mean()


def _batches(case):
    out = []
    for t in range(1, case["steps"] + 1):
        acts, pred_lens, labels = cases.synth_train_batch(case, t)
        out.append((acts, pred_lens, labels))
    return out


@pytest.mark.parametrize("case", cases.TRAIN_CASES + cases.TRAIN_METHOD_CASES, ids=lambda c: c["name"])
def test_training_steps_match_reference_golden(golden, case):
    import torch
    import probing_rag_amd as pra
    name = case["name"]
    st = cases.synth_state(case["wseed"], case["d"])
    tr = pra.HipProberTrainer(case["d"], 2, dropout_p=case.get("dropout_p", 0.1), seed=case["seed"])
    tr.load_state_dict(st)
    losses, lrs = [], []
    for acts, pred_lens, labels in _batches(case):
        lrs.append(tr.lr)
        # the reference's call: method_2_train(model, optim, scheduler, activations, labels, pred_lens, args)
        method = case.get("method", "tokens_mean")
        step_labels = torch.from_numpy(labels)
        if method == "each_token":       # method_1_train (utils.py:164-173): every trailing token is a row
            pooled, step_labels = pra.pool_each_token(torch.from_numpy(acts).cuda(), pred_lens, labels)
            want_rows, want_labels = onp.pool_each_token(acts, pred_lens, labels)
            assert np.array_equal(pooled.cpu().numpy(), want_rows) and np.array_equal(step_labels.cpu().numpy(), want_labels)
        elif method == "last_token":     # method_3_train (utils.py:213-220)
            pooled = pra.pool_last_token(torch.from_numpy(acts).cuda())
            assert np.array_equal(pooled.cpu().numpy(), acts[:, -1, :])
        else:
            pooled = pra.pool_ragged(torch.from_numpy(acts).cuda(), pred_lens, mean=True)
        loss, probs = tr.step(pooled, step_labels)
        losses.append(loss.item())
        assert probs.shape == (pooled.shape[0], 2) and abs(probs.sum().item() - pooled.shape[0]) < 1e-4
    np.testing.assert_allclose(losses, golden[f"{name}/losses"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(lrs, golden[f"{name}/lrs"], rtol=1e-12)
    assert tr.steps == case["steps"]
    sd = tr.state_dict()
    for k in onp.STATE_KEYS:
        got = sd[k].numpy()
        if f"{name}/final/{k}" in golden:
            np.testing.assert_allclose(got, golden[f"{name}/final/{k}"], atol=5e-6, rtol=0)
        else:
            np.testing.assert_allclose(got.reshape(-1)[::cases.TRAIN_SAMPLE_STRIDE],
                                       golden[f"{name}/final/{k}/sample"], atol=5e-6, rtol=0)
            assert abs(got.astype(np.float64).sum() - float(golden[f"{name}/final/{k}/sum"])) < 1e-2


def test_method_2_train_signature_and_checkpoint_roundtrip():
    """Drop-in call shape of utils.py:191-197 and a state_dict that the inference prober loads."""
    import torch
    import probing_rag_amd as pra
    case = cases.TRAIN_CASES[0]
    st = cases.synth_state(case["wseed"], case["d"])
    tr = pra.HipProberTrainer(case["d"], 2, seed=case["seed"]).load_state_dict(st)
    acts, pred_lens, labels = cases.synth_train_batch(case, 1)
    loss4, lr = pra.method_2_train(tr, None, None, torch.from_numpy(acts).cuda(), torch.from_numpy(labels),
                                   torch.from_numpy(pred_lens), None)
    assert loss4 == round(loss4, 4) and abs(lr - 1e-4 * 0.995) < 1e-15
    sd = tr.state_dict()
    pr = pra.HipProber(case["d"], 2, weights="f32")
    pr.load_state_dict(sd)
    x = torch.from_numpy(onp.synth_rows(5, 0, 4, case["d"])).cuda()
    want = onp.prober_forward({k: v.numpy() for k, v in sd.items()}, x.cpu().numpy())
    np.testing.assert_allclose(pr(x).cpu().numpy(), want, atol=1e-4, rtol=0)


@pytest.mark.parametrize("B,d", [(1, 2048), (19, 2048), (64, 4096)])
def test_training_step_matches_oracle_and_torch_autograd(B, d):
    """Other batch sizes / the d_model=4096 branch: two steps against the float64 oracle and against
    torch fp32 autograd + torch.optim.AdamW + ExponentialLR with the same dropout masks."""
    import torch
    import probing_rag_amd as pra
    seed, p = 99, 0.1
    st = cases.synth_state(600 + B, d)
    xs = [onp.synth_rows(700 + t, 0, B, d) * np.float32(1.5) + np.float32(0.2) for t in (1, 2)]
    labs = [np.array([(3 * i + t) % 2 for i in range(B)], dtype=np.int64) for t in (1, 2)]
    tr = pra.HipProberTrainer(d, 2, seed=seed).load_state_dict(st)
    got_losses = [tr.step(torch.from_numpy(x).cuda(), torch.from_numpy(y))[0].item() for x, y in zip(xs, labs)]
    final, losses, _ = onp.train_steps(st, list(zip(xs, labs)), seed, {"dropout_p": p})
    np.testing.assert_allclose(got_losses, losses, atol=2e-6, rtol=0)
    sd = tr.state_dict()
    for k in onp.STATE_KEYS:
        _params_close(sd[k].numpy(), final[k])

    # plain PyTorch fp32 (autograd + AdamW + ExponentialLR) on the GPU with the same masks
    dev = torch.device("cuda")
    W = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in st.items()}
    opt = torch.optim.AdamW(list(W.values()), lr=1e-4)
    sch = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.995)
    F = torch.nn.functional
    for t, (x, y) in enumerate(zip(xs, labs), start=1):
        k1 = torch.from_numpy(onp.dropout_keep(seed, t, 0, B, 512, p)).to(dev).float() * np.float32(1 / (1 - p))
        k2 = torch.from_numpy(onp.dropout_keep(seed, t, 1, B, 512, p)).to(dev).float() * np.float32(1 / (1 - p))
        h = F.layer_norm(torch.from_numpy(x).to(dev), (d,), W["layer_norm_input.weight"], W["layer_norm_input.bias"])
        h = F.layer_norm(F.silu(F.linear(h, W["fc1.weight"], W["fc1.bias"])), (512,), W["layer_norm1.weight"],
                         W["layer_norm1.bias"]) * k1
        h = F.layer_norm(F.silu(F.linear(h, W["fc2.weight"], W["fc2.bias"])), (512,), W["layer_norm2.weight"],
                         W["layer_norm2.bias"]) * k2
        probs = torch.softmax(F.linear(h, W["fc3.weight"], W["fc3.bias"]), dim=-1)
        loss = F.cross_entropy(probs, torch.from_numpy(y).to(dev))      # train.py:149-150
        assert abs(loss.item() - got_losses[t - 1]) < 2e-6
        loss.backward()
        opt.step()
        sch.step()
        opt.zero_grad()
    for k in onp.STATE_KEYS:
        _params_close(sd[k].numpy(), W[k].detach().cpu().numpy())


def test_labels_are_validated_and_eval_mode_disables_dropout():
    """torch's CrossEntropyLoss raises for a class outside [0, C) (train.py:149-150); `probe.eval()`
    makes dropout the identity (train.py:299)."""
    import torch
    import probing_rag_amd as pra
    d, B = 2048, 6
    st = cases.synth_state(77, d)
    x = torch.from_numpy(onp.synth_rows(78, 0, B, d)).cuda()
    tr = pra.HipProberTrainer(d, 2, seed=5).load_state_dict(st)
    for bad in ([0, 1, 2, 0, 1, 0], [0, -1, 1, 0, 1, 0], [0, 1, -100, 0, 1, 0]):
        with pytest.raises(IndexError, match="out of bounds"):
            tr.step(x, torch.tensor(bad))
    with pytest.raises(ValueError):
        tr.step(x, torch.tensor([0, 1]))
    assert tr.steps == 0
    labels = torch.tensor([0, 1, 1, 0, 1, 0])
    # eval-mode step: forward == the inference prober's probabilities (no dropout mask)
    tr.eval()
    _, probs = tr.step(x, labels)
    want = onp.prober_forward(st, x.cpu().numpy()).astype(np.float64)
    want = np.exp(want - want.max(1, keepdims=True))
    want /= want.sum(1, keepdims=True)
    np.testing.assert_allclose(probs.cpu().numpy(), want, atol=2e-5, rtol=0)
    # ... and it is a forward pass ONLY (the reference validates under `probe.eval()`, train.py:299): no
    # optimiser / scheduler step, parameters bit-identical
    assert tr.steps == 0 and tr.lr == pra.HipProberTrainer(d, 2, seed=5).lr
    sd = tr.state_dict()
    for k in onp.STATE_KEYS:
        assert np.array_equal(sd[k].numpy(), np.asarray(st[k], np.float32)), k
    # back in train mode the masks are applied again (p = 0.1 changes the forward)
    tr2 = pra.HipProberTrainer(d, 2, seed=5).load_state_dict(st)
    _, p_train = tr2.train().step(x, labels)
    assert np.abs(p_train.cpu().numpy() - want).max() > 1e-4


def test_each_token_and_last_token_methods_keep_the_reference_signatures(golden):
    """method_1_train / method_3_train (utils.py:164-173, 213-220) and method_1_eval / method_3_eval (utils.py:175-179,
    222-226) by their reference signatures: eval outputs against the golden vectors of the reference functions
    themselves; the training wrappers against the explicit pool + step; half-precision activations pool exactly."""
    import torch
    import probing_rag_amd as pra
    case = cases.POOL_CASES[0]
    name = case["name"]
    st = cases.synth_state(case["wseed"], case["d"])
    acts, pred_lens, labels = cases.synth_pool_inputs(case)
    prober = pra.HipProber(case["d"], 2)
    prober.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    a = torch.from_numpy(acts).cuda()
    for tag, fn in (("m1", pra.method_1_eval), ("m3", pra.method_3_eval)):
        acc, n, loss, probs = fn(prober, a, torch.from_numpy(labels), torch.from_numpy(pred_lens), None, return_probs=True)
        assert len(fn(prober, a, torch.from_numpy(labels), torch.from_numpy(pred_lens), None)) == 3   # the reference's call
        np.testing.assert_allclose(probs.cpu().numpy(), golden[f"{name}/{tag}_probs"], atol=1e-4, rtol=0)
        assert abs(loss.item() - float(golden[f"{name}/{tag}_loss"])) < 1e-4
        assert acc == float(golden[f"{name}/{tag}_acc"]) and n == int(golden[f"{name}/{tag}_n"])
    # fp16 / bf16 activations (a half-precision LM): the gather converts, nothing else
    for dt in (torch.float16, torch.bfloat16):
        ah = a.to(dt)
        rows, lab = pra.pool_each_token(ah, pred_lens, labels)
        want, wl = onp.pool_each_token(ah.float().cpu().numpy(), pred_lens, labels)
        assert np.array_equal(rows.cpu().numpy(), want) and np.array_equal(lab.cpu().numpy(), wl)
        assert np.array_equal(pra.pool_last_token(ah).cpu().numpy(), ah[:, -1, :].float().cpu().numpy())
    # empty and full-length samples
    lens = np.array([0, case["T"], 1, 0, 3, case["T"], 2, 0], dtype=np.int64)
    rows, lab = pra.pool_each_token(a, lens, labels)
    want, wl = onp.pool_each_token(acts, lens, labels)
    assert rows.shape[0] == int(lens.sum()) and np.array_equal(rows.cpu().numpy(), want) and np.array_equal(lab.cpu().numpy(), wl)
    # ... which the reference's method_1_* cannot digest (`-0:` takes every position, labels repeat 0 times): they raise
    for bad in (lens, np.full_like(lens, case["T"] + 1)):
        with pytest.raises(ValueError):
            pra.method_1_eval(prober, a, torch.from_numpy(labels), torch.from_numpy(bad), None)
    # the training wrappers
    for train_fn, pool in ((pra.method_1_train, lambda: pra.pool_each_token(a, pred_lens, labels)),
                           (pra.method_3_train, lambda: (pra.pool_last_token(a), torch.from_numpy(labels)))):
        t1 = pra.HipProberTrainer(case["d"], 2, seed=5).load_state_dict(st)
        t2 = pra.HipProberTrainer(case["d"], 2, seed=5).load_state_dict(st)
        lrnd, lr = train_fn(t1, None, None, a, torch.from_numpy(labels), torch.from_numpy(pred_lens), None)
        x, lab = pool()
        loss, _ = t2.step(x, lab)
        assert lrnd == round(loss.item(), 4) and lr == t2.lr == pytest.approx(1e-4 * 0.995)
        for k in onp.STATE_KEYS:
            assert torch.equal(t1.state_dict()[k], t2.state_dict()[k])
