"""GPU parity: HIP prober ensemble + gate (through the C ABI) vs the oracle and
the golden vectors of the reference's ImprovedProbe.  Tolerance: 1e-4 on logits
(BASELINE.json north_star); decisions exact outside 1e-4 of the threshold."""
import numpy as np
import pytest

from oracle import oracle_c, oracle_np as onp
from tests.golden import cases

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _ensemble(case, weights):
    import probing_rag_amd as pra
    states = [cases.synth_state(case["wseed"] + l, case["d"]) for l in range(case["L"])]
    ens = pra.HipProberEnsemble(case["L"], case["d"], 2, weights=weights)
    for l, st in enumerate(states):
        ens.load_layer(l, st)
    return ens, states


def _oracle_effective(ens, x_np):
    return np.stack([onp.prober_forward(ens.effective_state_dict(l), x_np[l]) for l in range(ens.n_layers)])


@pytest.mark.parametrize("case", cases.PROBER_CASES, ids=lambda c: c["name"])
def test_fp32_parity_against_reference_golden(torch_cuda, golden, case):
    """The reference's own call shape: fp32 activations, fp32 weights."""
    torch = torch_cuda
    ens, states = _ensemble(case, "f32")
    x = cases.case_x(case)
    xd = torch.from_numpy(x).cuda()
    logits = ens.forward(xd).cpu().numpy()
    ref = golden[f"{case['name']}/logits"]
    np.testing.assert_allclose(logits, ref, atol=TOL, rtol=0)
    np.testing.assert_allclose(logits, onp.ensemble_forward(states, x), atol=TOL, rtol=0)
    # per-layer `prober(input)` (exp_rag.py:387) == ensemble launch
    for l, prober in enumerate(ens.probers):
        one = prober(xd[l]).to("cpu").numpy()
        np.testing.assert_allclose(one, logits[l], atol=1e-6, rtol=0)
    # gate for every (ablation, theta) of the paper's sweep
    for ab in cases.ABLATIONS:
        if ab >= case["L"]:
            continue
        for th in cases.THETAS:
            lg, ps, dec = ens.gate(xd, ablation=ab, threshold=th)
            ps, dec = ps.cpu().numpy(), dec.cpu().numpy()
            ref_ps = golden[f"{case['name']}/probsum_ab{ab}"]
            np.testing.assert_allclose(ps, ref_ps, atol=TOL, rtol=0)
            ref_dec = golden[f"{case['name']}/decision_ab{ab}_th{th}"]
            margin = np.abs(ref_ps[:, 0] + np.float32(th) - ref_ps[:, 1])
            assert not ((dec != ref_dec) & (margin > TOL)).any()
            # bit-for-bit against the oracle's gate on OUR logits
            ops, odec = oracle_c.gate(lg.cpu().numpy(), ab, th)
            np.testing.assert_allclose(ps, ops, atol=1e-6, rtol=0)
            m2 = np.abs(ops[:, 0] + np.float32(th) - ops[:, 1])
            assert not ((dec != odec) & (m2 > 1e-6)).any()


@pytest.mark.parametrize("weights,xdtype", [("f16", "f16"), ("f16", "f32"), ("f32", "f16")])
@pytest.mark.parametrize("case", cases.PROBER_CASES, ids=lambda c: c["name"])
def test_reduced_precision_modes_against_oracle_on_same_operands(torch_cuda, golden, case, weights, xdtype):
    """fp16 weights / activations: parity is defined on identical numeric inputs —
    the oracle consumes the fp16-rounded activations and the weights the kernel
    actually holds (effective_state_dict), in float64."""
    torch = torch_cuda
    ens, _ = _ensemble(case, weights)
    x = cases.case_x(case)
    if xdtype == "f16":
        if np.abs(x).max() > 6e4:
            pytest.skip("fp16 overflow")
        xd = torch.from_numpy(x).cuda().half()
        x_seen = xd.float().cpu().numpy()
    else:
        xd = torch.from_numpy(x).cuda()
        x_seen = x
    logits = ens.forward(xd).cpu().numpy()
    np.testing.assert_allclose(logits, _oracle_effective(ens, x_seen), atol=TOL, rtol=0)
    # and it stays close to the full-precision reference (weight rounding only)
    np.testing.assert_allclose(logits, golden[f"{case['name']}/logits"], atol=2e-2, rtol=0)


@pytest.mark.parametrize("B", [5, 70, 200, 600])
def test_fp16_weight_mode_lo_term_under_stress(torch_cuda, B):
    """The fp16-weight mode carries the low bits of fc2's input as fp8 against an fp8 copy of W2.  Weights
    with a wide dynamic range (a few rows and columns 100x the rest, many tiny entries), large fc1 outputs
    (|SiLU| up to ~50: the lo term reaches its clamp region only far beyond that) and every tile shape
    (B = 5 ... 600 takes the 32-, 64- and 128-row kernels) stay inside the 1e-4 contract against the
    float64 oracle on the same operands."""
    torch = torch_cuda
    import probing_rag_amd as pra
    d = 1024
    rng = np.random.default_rng(B)
    st = cases.synth_state(77, d)
    w2 = st["fc2.weight"].copy()
    w2[rng.integers(0, 512, 6)] *= 100.0
    w2[:, rng.integers(0, 512, 6)] *= 100.0
    w2[rng.random(w2.shape) < 0.3] *= 1e-3
    st["fc2.weight"] = w2.astype(np.float32)
    st["fc1.weight"] = (st["fc1.weight"] * 8.0).astype(np.float32)      # large pre-activations
    ens = pra.HipProberEnsemble(1, d, 2, weights="f16")
    ens.load_layer(0, st)
    x = (rng.standard_normal((1, B, d)) * 3.0).astype(np.float32)
    xd = torch.from_numpy(x).cuda().half()
    got = ens.forward(xd).cpu().numpy()
    want = _oracle_effective(ens, xd.float().cpu().numpy())
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, want, atol=TOL, rtol=0)


@pytest.mark.parametrize("d", [128, 192, 320, 1600, 2304, 3584, 5120])
def test_odd_model_widths_all_tile_shapes(torch_cuda, d):
    """d_model only has to be a multiple of 64: two K steps (128), odd numbers of K steps (192, 320, 1600),
    Gemma-2 / Qwen / Llama widths - every tile shape (B = 3 ... 300), both weight modes, fp16 and fp32
    activations, against the float64 oracle on the weights the kernel holds."""
    torch = torch_cuda
    import probing_rag_amd as pra
    rng = np.random.default_rng(d)
    for weights in ("f16", "f32"):
        ens = pra.HipProberEnsemble(2, d, 2, weights=weights)
        for l in range(2):
            ens.load_layer(l, cases.synth_state(300 + l, d))
        for B in (3, 40, 130, 300):
            x = rng.standard_normal((2, B, d)).astype(np.float32)
            for xd in (torch.from_numpy(x).cuda().half(), torch.from_numpy(x).cuda()):
                got = ens.forward(xd).cpu().numpy()
                want = _oracle_effective(ens, xd.float().cpu().numpy())
                np.testing.assert_allclose(got, want, atol=TOL, rtol=0, err_msg=f"d={d} B={B} w={weights} x={xd.dtype}")


def test_state_dict_contract(torch_cuda):
    import probing_rag_amd as pra
    p = pra.HipProber(2048, 2)
    st = cases.synth_state(1, 2048)
    bad = dict(st)
    bad.pop("fc3.bias")
    with pytest.raises(RuntimeError, match="missing"):
        p.load_state_dict(bad)
    bad = dict(st)
    bad["fc1.weight"] = bad["fc1.weight"][:, :100]
    with pytest.raises(RuntimeError, match="size mismatch"):
        p.load_state_dict(bad)
    with pytest.raises(pra.PragError, match="PRAG_ESTATE"):
        p(torch_cuda.zeros(1, 2048, device="cuda"))
    p.load_state_dict({k: torch_cuda.from_numpy(v) for k, v in st.items()}).eval().to("cuda")
    with pytest.raises(RuntimeError, match="same device"):
        p(torch_cuda.zeros(1, 2048))
    out = p(torch_cuda.zeros(0, 2048, device="cuda"))
    assert tuple(out.shape) == (0, 2)


def test_c2_full_size_properties(torch_cuda):
    """BASELINE config 2: B=4096 x 6 layers x d=2048 fp16.  The oracle checks a
    256-row sample; size-independent properties cover the rest: a row's logits do
    not depend on which tile/batch it sits in, nor on a power-of-two rescale of
    the activations (LayerNorm-first)."""
    torch = torch_cuda
    case = dict(name="c2", d=2048, L=6, B=4096, sigma=1.0, wseed=100, xseed=999)
    ens, _ = _ensemble(case, "f16")
    x = cases.case_x(case)
    xd = torch.from_numpy(x).cuda().half()
    logits, probsum, dec = ens.gate(xd, 0, 0.0)
    logits = logits.cpu().numpy()
    assert np.isfinite(logits).all()
    rows = np.arange(0, 4096, 16)
    want = _oracle_effective(ens, xd[:, rows].float().cpu().numpy())
    np.testing.assert_allclose(logits[:, rows], want, atol=TOL, rtol=0)
    # batch-composition independence (different tile shapes are used for B=37)
    sub = np.array([3, 4095, 1027, 64, 65, 2048] + list(range(500, 531)))
    small = ens.forward(xd[:, sub].contiguous()).cpu().numpy()
    np.testing.assert_allclose(small, logits[:, sub], atol=2e-6, rtol=0)
    # scale invariance: x*4 is exact in fp16
    scaled = ens.forward(xd * 4).cpu().numpy()
    np.testing.assert_allclose(scaled, logits, atol=TOL, rtol=0)
    # gate consistency with the oracle on these logits
    ops, odec = oracle_c.gate(logits, 0, 0.0)
    np.testing.assert_allclose(probsum.cpu().numpy(), ops, atol=1e-6, rtol=0)
    m = np.abs(ops[:, 0] - ops[:, 1])
    assert not ((dec.cpu().numpy() != odec) & (m > 1e-6)).any()


def test_pooling_kernels(torch_cuda, golden):
    """exp_rag.py:385-386 (sum over decode steps, prompt pass skipped) and
    train.py:153-162/202-205 (ragged mean) on device."""
    torch = torch_cuda
    import probing_rag_amd as pra
    case = cases.POOL_CASES[0]
    acts, pred_lens, labels = cases.synth_pool_inputs(case)
    ad = torch.from_numpy(acts).cuda()
    mean = pra.pool_ragged(ad, pred_lens, mean=True).cpu().numpy()
    np.testing.assert_allclose(mean, onp.pool_ragged_mean(acts, pred_lens), atol=1e-6, rtol=1e-6)
    st = cases.synth_state(case["wseed"], case["d"])
    p = pra.HipProber(case["d"], 2)
    p.load_state_dict(st)
    probs = torch.softmax(p(torch.from_numpy(mean).cuda()), dim=-1).cpu().numpy()
    np.testing.assert_allclose(probs, golden[f"{case['name']}/probs"], atol=TOL, rtol=0)
    sums = pra.pool_ragged(ad, pred_lens, mean=False)
    np.testing.assert_allclose(p(sums).cpu().numpy(), golden[f"{case['name']}/sum_logits"], atol=TOL, rtol=0)
    # hook accumulator: pass 0 = prompt (skipped), then T decode steps
    pool = pra.HiddenStatePool(2, 64, batch=1)
    rng = np.random.default_rng(0)
    passes = [rng.standard_normal((1, 9, 64)).astype(np.float32)] + \
             [rng.standard_normal((1, 1, 64)).astype(np.float32) for _ in range(7)]
    with pytest.raises(RuntimeError):
        pool.pooled()
    for layer in range(2):
        for a in passes:
            pool.observe(layer, torch.from_numpy(a * (layer + 1)).cuda())
    got = pool.pooled().cpu().numpy()
    want = onp.pool_sum_decode_steps(passes)
    np.testing.assert_allclose(got[0], want, atol=1e-5)
    np.testing.assert_allclose(got[1], 2 * want, atol=1e-5)
    pool.reset()
    pool.observe(0, torch.from_numpy(passes[0]).cuda())
    pool.observe(0, torch.from_numpy(passes[0]).cuda())   # a multi-position later pass
    pool.observe(1, torch.from_numpy(passes[0]).cuda())
    pool.observe(1, torch.from_numpy(passes[1]).cuda())
    np.testing.assert_allclose(pool.pooled()[0].cpu().numpy(), passes[0].sum(axis=1), atol=1e-5)


def test_bf16_activations_and_masked_mean_pool(torch_cuda):
    """bf16 hidden states (the usual HF dtype) go through the fp32 pre-normalise path; the
    encoder-side masked mean pool matches its definition."""
    torch = torch_cuda
    import probing_rag_amd as pra
    case = cases.PROBER_CASES[2]
    ens, _ = _ensemble(case, "f32")
    x = cases.case_x(case)
    xb = torch.from_numpy(x).cuda().bfloat16()
    got = ens.forward(xb).cpu().numpy()
    want = _oracle_effective(ens, xb.float().cpu().numpy())
    np.testing.assert_allclose(got, want, atol=TOL, rtol=0)
    rng = np.random.default_rng(3)
    h = rng.standard_normal((5, 17, 768)).astype(np.float32)
    lens = [17, 1, 9, 4, 12]
    mask = np.array([[1] * n + [0] * (17 - n) for n in lens], np.int64)
    want = (h * mask[:, :, None]).sum(1) / mask.sum(1, keepdims=True)
    for dt, tol in ((torch.float32, 1e-6), (torch.float16, 2e-3), (torch.bfloat16, 2e-2)):
        got = pra.masked_mean_pool(torch.from_numpy(h).cuda().to(dt), torch.from_numpy(mask).cuda()).cpu().numpy()
        np.testing.assert_allclose(got, want, atol=tol, rtol=0)


def test_bf16_hidden_states_pool_without_a_conversion_pass(torch_cuda):
    """Gemma's hidden states are bf16: the hook accumulator and the ragged pool read them as they are
    (bf16 -> f32 is exact, sums are f32 in position order), so the result equals the pool of the
    same values held in float32."""
    torch = torch_cuda
    import probing_rag_amd as pra
    rng = np.random.default_rng(11)
    acts = torch.from_numpy(rng.standard_normal((3, 13, 256)).astype(np.float32)).cuda().bfloat16()
    lens = [13, 4, 1]
    for mean in (True, False):
        got = pra.pool_ragged(acts, lens, mean=mean).cpu().numpy()
        want = pra.pool_ragged(acts.float(), lens, mean=mean).cpu().numpy()
        np.testing.assert_array_equal(got, want)
    pool16, pool32 = pra.HiddenStatePool(1, 256, batch=3), pra.HiddenStatePool(1, 256, batch=3)
    pool16.observe(0, acts)            # prompt pass (skipped)
    pool32.observe(0, acts.float())
    for t in range(5):
        step = acts[:, t:t + 1].contiguous()
        pool16.observe(0, step)
        pool32.observe(0, step.float())
    pool16.observe(0, acts)            # a later pass without KV cache: all of its positions
    pool32.observe(0, acts.float())
    np.testing.assert_array_equal(pool16.pooled().cpu().numpy(), pool32.pooled().cpu().numpy())


def test_method_2_eval_matches_reference_golden(torch_cuda, golden):
    """train.py's evaluation forward (ragged mean pool -> prober -> double-softmax CE -> acc)."""
    torch = torch_cuda
    import probing_rag_amd as pra
    case = cases.POOL_CASES[0]
    acts, pred_lens, labels = cases.synth_pool_inputs(case)
    p = pra.HipProber(case["d"], 2)
    p.load_state_dict(cases.synth_state(case["wseed"], case["d"]))
    acc, n, loss, probs = pra.method_2_eval(p, torch.from_numpy(acts).cuda(), labels, pred_lens, return_probs=True)
    np.testing.assert_allclose(probs.cpu().numpy(), golden[f"{case['name']}/probs"], atol=TOL, rtol=0)
    assert abs(float(loss) - float(golden[f"{case['name']}/loss"])) < 1e-4
    assert acc == round(float(golden[f"{case['name']}/acc"]), 4) and n == case["B"]


def _model_with_cfg(model_id="google/gemma-2b"):
    class Cfg:
        d_model, tokenizer_name = 2048, model_id

    class Model:
        cfg = Cfg()
    return Model()


@pytest.mark.parametrize("ds", [3, 25, 12345])
def test_load_prober_models_with_the_reference_call(torch_cuda, golden, tmp_path, monkeypatch, ds):
    """exp_rag.py:311-312 verbatim: ``cfg_list = load_prober_cfg_gemma_2b(model, Config_Maker, position,
    device, 6, 17, 2); probers = load_prober_models(_ds, cfg_list)`` with the --ds INTEGER.  The
    checkpoints (not shipped with the reference) are written under the names utils.py:303-326 builds,
    relative to the working directory, then the reference's own per-layer loop runs on the result."""
    import probing_rag_amd as pra
    torch = torch_cuda
    case = cases.PROBER_CASES[1]
    model = _model_with_cfg()
    cfg_list = pra.load_prober_cfg_gemma_2b(model, pra.Config_Maker, "resid_post", "cuda", 6, 17, 2)
    states = [cases.synth_state(case["wseed"] + l, case["d"]) for l in range(case["L"])]
    monkeypatch.chdir(tmp_path)
    for cfg, st in zip(cfg_list, states):
        path = tmp_path / pra.prober_checkpoint_path(ds, cfg)
        path.parent.mkdir(parents=True, exist_ok=True)
        torch.save({k: torch.from_numpy(v) for k, v in st.items()}, str(path))    # train.py:344-345
    probers = pra.load_prober_models(ds, cfg_list)
    assert len(probers) == 6
    x = torch.from_numpy(cases.case_x(case)).cuda()
    logits = pra.return_prober_logit_gemma_2b(lambda cfg, prober: prober(x[cfg_list.index(cfg)]), cfg_list, probers)
    got = np.stack([t.numpy() for t in logits])
    np.testing.assert_allclose(got, golden[f"{case['name']}/logits"], atol=TOL, rtol=0)
    # one prober at a time, as utils.py:291-330 does it
    single = pra.load_prober(ds, cfg_list[2])
    np.testing.assert_allclose(single(x[2]).cpu().numpy(), got[2], atol=1e-6, rtol=0)
    with pytest.raises(RuntimeError, match="eval-mode"):
        single.train()                                         # points at HipProberTrainer
    assert single.train(False) is single
    # a missing checkpoint is the reference's FileNotFoundError
    with pytest.raises(FileNotFoundError):
        pra.load_prober_models(777, cfg_list)
    # a model id without a branch in utils.py:303-326: nothing is loaded (no-op assert), first call raises
    other = pra.load_prober_cfg_gemma_2b(_model_with_cfg("meta-llama/Llama-2-7b"), pra.Config_Maker,
                                         "resid_post", "cuda", 6, 9, 2)
    unloaded = pra.load_prober_models(ds, other)
    with pytest.raises(pra.PragError, match="PRAG_ESTATE"):
        unloaded[0](x[0])


def test_reloading_a_layer_replaces_its_weights(torch_cuda):
    """load_state_dict per epoch / checkpoint swap: the layer's device buffers are replaced (and the
    old ones freed), other layers keep theirs."""
    import probing_rag_amd as pra
    torch = torch_cuda
    case = cases.PROBER_CASES[1]
    ens, states = _ensemble(case, "f32")
    x = cases.case_x(case)
    xd = torch.from_numpy(x).cuda()
    before = ens.forward(xd).cpu().numpy()
    new3 = cases.synth_state(999, case["d"])
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(20):
        ens.load_layer(3, new3)
    assert torch.cuda.mem_get_info()[0] >= free0 - (8 << 20)            # 20 reloads do not pile up buffers
    after = ens.forward(xd).cpu().numpy()
    np.testing.assert_allclose(after[3], onp.prober_forward(new3, x[3]), atol=TOL, rtol=0)
    keep = [l for l in range(case["L"]) if l != 3]
    assert np.array_equal(after[keep], before[keep])


def test_gate_threshold_is_compared_in_double_like_python(torch_cuda):
    """exp_rag.py:414 compares Python floats: float32 sums widened to double, theta a double.  A
    threshold that only differs from the knife edge below float32 resolution must still decide."""
    import probing_rag_amd as pra
    torch = torch_cuda
    lg = np.zeros((1, 4, 2), np.float32)
    lg[0, :, 1] = [0.0, 1.0, -1.0, 0.3]
    logits = torch.from_numpy(lg).cuda()
    ps, _ = pra.gate_from_logits(logits, 0, 0.0)
    ps = ps.cpu().numpy().astype(np.float64)
    edge = ps[:, 1] - ps[:, 0]                                  # s0 + theta < s1  <=>  theta < edge
    for b in range(4):
        for theta in (np.nextafter(edge[b], -np.inf), edge[b], np.nextafter(edge[b], np.inf)):
            _, dec = pra.gate_from_logits(logits, 0, float(theta))
            want = 0 if ps[b, 0] + float(theta) < ps[b, 1] else 1
            assert int(dec[b]) == want
            _, odec = onp.gate(lg, 0, float(theta))
            assert int(odec[b]) == want


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_decode_step_of_all_layers_pools_in_one_launch(torch_cuda, dtype):
    """exp_rag.py:317-329 hooks six layers: the accumulator adds a decode step of all of them in one launch
    (prag_pool_accumulate_layers); same sums as one launch per layer, also when a layer fires twice before the
    others, when a multi-position pass interleaves, and when pooled() is read mid-step."""
    torch = torch_cuda
    import probing_rag_amd as pra
    L, Bt, d = 6, 2, 256
    rng = np.random.default_rng(3)
    dt = getattr(torch, dtype)
    a, b = pra.HiddenStatePool(L, d, batch=Bt, defer=True), pra.HiddenStatePool(L, d, batch=Bt)
    assert a.defer and not b.defer       # immediate adds are the default
    steps = [torch.from_numpy(rng.standard_normal((L, Bt, 9 if t in (0, 4) else 1, d)).astype(np.float32)).cuda().to(dt)
             for t in range(8)]
    for t, st in enumerate(steps):
        order = list(range(L)) if t != 3 else [0, 0, 1, 2, 3, 4, 5]       # layer 0 observed twice in step 3
        for l in order:
            a.observe(l, st[l])
            b.observe(l, st[l])
        if t == 5:
            assert torch.equal(a.pooled(), b.pooled())
    assert torch.equal(a.pooled(), b.pooled())
    want = sum(st.float().sum(dim=2) for st in steps[1:]) + steps[3][0].float().sum(dim=1)[None] * \
        torch.tensor([1.0] + [0.0] * (L - 1), device="cuda")[:, None, None]
    np.testing.assert_allclose(a.pooled().cpu().numpy(), want.cpu().numpy(), atol=1e-4)
    # a noted tensor that is modified in place before the flush is an error, not a wrong sum
    c = pra.HiddenStatePool(2, d, batch=Bt, defer=True)
    h = [torch.ones((Bt, 1, d), device="cuda") for _ in range(2)]
    for l in range(2):
        c.observe(l, h[l])                  # prompt pass (skipped)
    c.observe(0, h[0])                      # noted, waiting for layer 1
    h[0].add_(1.0)                          # e.g. an in-place residual add into a static buffer
    with pytest.raises(RuntimeError, match="modified in place"):
        c.observe(1, h[1])


@pytest.mark.parametrize("d", [2048, 320])
def test_fp16_mode_on_both_mfma_shapes(torch_cuda, monkeypatch, d):
    """The fp16 x fp16 mode runs on 16 x 16 MFMA tiles (prober16.hip); PRAG_PROBER_SHAPE=32 keeps the 32 x 32 kernel.
    Same weights, same folding, same fp8 lo term: the two agree to accumulation-order noise at every tile height
    (32-, 64- and 128-row tiles, ragged last tiles), and both meet the oracle on the weights the kernel holds."""
    torch = torch_cuda
    import probing_rag_amd as pra
    L = 3
    states = [cases.synth_state(700 + l, d) for l in range(L)]
    ens16 = pra.HipProberEnsemble(L, d, 2, weights="f16")
    monkeypatch.setenv("PRAG_PROBER_SHAPE", "32")
    ens32 = pra.HipProberEnsemble(L, d, 2, weights="f16")
    monkeypatch.delenv("PRAG_PROBER_SHAPE")
    for l, st in enumerate(states):
        ens16.load_layer(l, st)
        ens32.load_layer(l, st)
    rng = np.random.default_rng(d)
    for B in (3, 33, 64, 100, 130, 1000, 2100):
        x = torch.from_numpy((rng.standard_normal((L, B, d)) * 2.0 + 0.3).astype(np.float32)).cuda().half()
        a = ens16.forward(x).cpu().numpy()
        b = ens32.forward(x).cpu().numpy()
        np.testing.assert_allclose(a, b, atol=3e-5, rtol=0, err_msg=f"B={B}")
        if B <= 130:
            want = _oracle_effective(ens16, x.float().cpu().numpy())
            np.testing.assert_allclose(a, want, atol=TOL, rtol=0, err_msg=f"B={B}")
    # a row's logits do not depend on the tile it sits in (tile heights 32 / 64 / 128 share the kernel template)
    x = torch.from_numpy(rng.standard_normal((L, 3000, d)).astype(np.float32)).cuda().half()
    big = ens16.forward(x).cpu().numpy()
    sub = np.array([0, 1, 31, 32, 63, 64, 100, 127, 128, 1500, 2999])
    small = ens16.forward(x[:, sub].contiguous()).cpu().numpy()
    np.testing.assert_allclose(small, big[:, sub], atol=2e-6, rtol=0)


@pytest.mark.parametrize("weights,B", [("f32", 1), ("f32", 3), ("f16", 2), ("f16", 40), ("f32", 300)])
def test_decide_returns_the_gate_decisions_in_host_memory(torch_cuda, golden, monkeypatch, weights, B):
    """exp_rag.py:393, 406-415 end in a host branch: `ens.decide` (prag_gate_decide) is gate + copy-out + wait in one
    call and must give exactly what `ens.gate` writes on the device - on the small-batch path (B = 1, the reference's
    call shape), the tiled path, beyond the 256 rows the kernels write into host memory themselves, with the spin
    wait and with the plain stream wait, with and without the sums, and call after call on one handle."""
    torch = torch_cuda
    import probing_rag_amd as pra
    case = dict(cases.PROBER_CASES[1], B=B)
    x_np = cases.synth_x(case["xseed"], case["L"], B, case["d"], case["sigma"])
    x = torch.from_numpy(x_np).cuda()
    for spin in ("1", "0"):
        monkeypatch.setenv("PRAG_DECIDE_SPIN", spin)
        ens, _ = _ensemble(case, weights)
        monkeypatch.delenv("PRAG_DECIDE_SPIN")
        for ab, th in ((0, 0.0), (2, -1.0), (5, 1.5)):
            _, ps_dev, dec_dev = ens.gate(x, ab, th)
            for _ in range(3):
                dec = ens.decide(x, ab, th)
                assert dec.dtype == np.int32 and dec.shape == (B,)
                assert np.array_equal(dec, dec_dev.cpu().numpy())
            dec2, ps = ens.decide(x, ab, th, with_probsum=True)
            assert np.array_equal(dec2, dec) and np.array_equal(ps, ps_dev.cpu().numpy())
        # a smaller batch after a larger one reuses the handle's buffers
        d1 = ens.decide(x[:, :1].contiguous(), 0, 0.0)
        assert np.array_equal(d1, ens.gate(x[:, :1].contiguous(), 0, 0.0)[2].cpu().numpy())
        ens.close()


@pytest.mark.parametrize("B", [40, 129, 512, 1400])
def test_gate_folded_into_the_prober_launch_equals_gate_kernel(torch_cuda, monkeypatch, B):
    """Round 5 (PRAG_GATE_FOLD=1): the last of the six workgroups of a row tile runs exp_rag.py:407-415 for the tile's rows
    inside prober16_kernel (a ticket per tile; one launch less per batch of decisions).  Same arithmetic, same order as
    gate_kernel: logits, sums and decisions are IDENTICAL to the two-launch form (PRAG_GATE_FOLD=0), for every ablation,
    call after call (the tickets return to zero inside the launch) and replayed from a captured graph."""
    torch = torch_cuda
    case = dict(cases.PROBER_CASES[1], B=B)
    x = torch.from_numpy(cases.synth_x(case["xseed"], case["L"], B, case["d"], 1.0)).cuda().half()
    monkeypatch.setenv("PRAG_GATE_FOLD", "0")
    ens0, _ = _ensemble(case, "f16")
    monkeypatch.setenv("PRAG_GATE_FOLD", "1")         # (measured slower, off by default: prober.hip gate_fold)
    ens1, _ = _ensemble(case, "f16")
    monkeypatch.delenv("PRAG_GATE_FOLD")
    for ab, th in ((0, 0.0), (3, -0.5), (5, 1.0), (6, 0.0)):
        want = [t.cpu().numpy() for t in ens0.gate(x, ab, th)]
        for _ in range(3):
            got = [t.cpu().numpy() for t in ens1.gate(x, ab, th)]
            for g, w_ in zip(got, want):
                assert np.array_equal(g, w_), (ab, th)
    out = tuple(torch.empty_like(t) for t in ens1.gate(x, 1, 0.25))
    want = [t.clone() for t in ens0.gate(x, 1, 0.25)]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ens1.gate(x, 1, 0.25, out=out)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            ens1.gate(x, 1, 0.25, out=out)
    for _ in range(3):
        for t in out:
            t.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(out, want))


@pytest.mark.parametrize("weights,Bt,dtype", [("f32", 1, "float32"), ("f32", 1, "float16"), ("f32", 4, "bfloat16"),
                                              ("f16", 2, "float16"), ("f32", 5, "float32"), ("f16", 3, "float32")])
def test_decode_step_launch_also_decides(torch_cuda, weights, Bt, dtype):
    """Round 6: with `attach_gate` the launch that adds a decode step of all hooked layers (exp_rag.py:317-329) also runs
    the gate on the sums it has just formed (prag_pool_step_gate), and `pool.decide()` - exp_rag.py:393, 406-415's host
    branch - reads that result.  Sums, decisions and the two softmax sums must be bit-identical to a plain pool
    followed by `ens.decide(pool.pooled())`: after every step, across a reset, when a multi-position pass (no KV
    cache) invalidates the step's decision, and for a batch the small-batch gate does not serve (falls back to exactly
    that call)."""
    torch = torch_cuda
    import probing_rag_amd as pra
    case = cases.PROBER_CASES[1]
    L, d = case["L"], case["d"]
    ens, _ = _ensemble(case, weights)
    dt = getattr(torch, dtype)
    rng = np.random.default_rng(11)
    a = pra.HiddenStatePool(L, d, batch=Bt, defer=True).attach_gate(ens, 1, -0.25)
    b = pra.HiddenStatePool(L, d, batch=Bt, defer=True)
    stepped = weights == "f32" and Bt <= 4 or weights == "f16" and Bt <= 2
    for episode in range(3):
        a.reset()
        b.reset()
        n_steps = 6 if episode < 2 else 3
        for t in range(n_steps):
            T = 7 if t == 0 or (episode == 1 and t == 3) else 1          # prompt pass; one pass without KV cache
            st = torch.from_numpy(rng.standard_normal((L, Bt, T, d)).astype(np.float32) * 3.0).cuda().to(dt)
            for l in range(L):
                a.observe(l, st[l])
                b.observe(l, st[l])
            if t == 0:
                continue
            assert (a._step_tag is not None) == (stepped and T == 1)
            if t % 2 == 1 or t == n_steps - 1:
                want_dec, want_ps = ens.decide(b.pooled(), 1, -0.25, with_probsum=True)
                got_dec, got_ps = a.decide(with_probsum=True)
                assert torch.equal(a.pooled(), b.pooled())
                assert np.array_equal(got_dec, want_dec) and np.array_equal(got_ps, want_ps)
                assert np.array_equal(a.decide(), want_dec)              # asking twice is fine
    # an older step's tag is refused by the library (the block holds the most recent step only) ...
    if stepped:
        import ctypes
        from probing_rag_amd import _lib
        dec = np.empty((8,), np.int32)
        rc = _lib.lib().prag_gate_step_result(ens._h, ctypes.c_uint64(a._step_tag.value - 1), Bt, dec.ctypes.data, None, None)
        assert rc == -5
    # ... and the pool never asks for one: a detached pool decides with the plain call
    with pytest.raises(RuntimeError, match="attach_gate"):
        b.decide()
    ens.close()


def test_decode_step_launch_under_contention(torch_cuda):
    """The step launch hands data between its workgroups inside the launch (arrival counters, bounded waits).  With another
    stream keeping every CU busy - large matrix products back to back - the waits are longer; the decision must still be
    `ens.decide`'s on the same sums (a hand-off that gives up voids the step and `pool.decide()` falls back to that call)."""
    torch = torch_cuda
    import probing_rag_amd as pra
    case = cases.PROBER_CASES[1]
    L, d = case["L"], case["d"]
    ens, _ = _ensemble(case, "f32")
    a = pra.HiddenStatePool(L, d, batch=1, defer=True).attach_gate(ens, 0, 0.0)
    b = pra.HiddenStatePool(L, d, batch=1, defer=True)
    side = torch.cuda.Stream()
    A = torch.randn((4096, 4096), device="cuda", dtype=torch.float16)
    rng = np.random.default_rng(5)
    for episode in range(6):
        a.reset()
        b.reset()
        with torch.cuda.stream(side):
            for _ in range(12):                      # ~1 ms of chip-filling work per episode on the other stream
                A = (A @ A).clamp_(-1, 1)
        for t in range(5):
            st = torch.from_numpy(rng.standard_normal((L, 1, 7 if t == 0 else 1, d)).astype(np.float32)).cuda()
            for l in range(L):
                a.observe(l, st[l])
                b.observe(l, st[l])
        got, got_ps = a.decide(with_probsum=True)
        want, want_ps = ens.decide(b.pooled(), 0, 0.0, with_probsum=True)
        assert np.array_equal(got, want) and np.array_equal(got_ps, want_ps) and torch.equal(a.pooled(), b.pooled())
    torch.cuda.synchronize()
    ens.close()
