#!/usr/bin/env python3
"""BASELINE config 5 in miniature, on one GPU: the retrieve-decide loop of exp_rag.py:396-474 around
a Gemma-2B-SHAPED decoder (18 layers, d_model 2048, random weights - the checkpoint is not in this
image; PyTorch-ROCm, plumbing) with the HIP gate and the HIP flat index in the loop.

What it shows: the hot path drops into a real-size generation loop through forward hooks
(exp_rag.py:317-329 -> HiddenStatePool), and what share of a query the gate and the retrieval
take next to generation.  Token ids, query embeddings and passages are synthetic.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=2_625_000, help="rows of the flat index (one 8-GPU shard by default)")
    ap.add_argument("--queries", type=int, default=6)
    ap.add_argument("--prompt-len", type=int, default=64)
    ap.add_argument("--new-tokens", type=int, default=32)
    ap.add_argument("--theta", type=float, default=0.0)
    args = ap.parse_args()
    import probing_rag_amd as pra
    from probing_rag_amd.synth import synth_rows
    from tests.golden import cases
    from transformers import GemmaConfig, GemmaForCausalLM

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    cfg = GemmaConfig(vocab_size=256000, hidden_size=2048, intermediate_size=16384, num_hidden_layers=18,
                      num_attention_heads=8, num_key_value_heads=1, head_dim=256, max_position_embeddings=8192)
    t0 = time.perf_counter()
    with torch.device(dev):
        lm = GemmaForCausalLM(cfg)
    lm = lm.half().eval()
    print(f"Gemma-2B-shaped decoder: {sum(p.numel() for p in lm.parameters())/1e9:.2f} B parameters, random init "
          f"({time.perf_counter()-t0:.1f} s)", flush=True)

    layers = list(range(6, 17, 2))                                  # exp_rag.py:311
    pool = pra.HiddenStatePool(len(layers), 2048, defer=True)
    for slot, l in enumerate(layers):                               # 'blocks.{l}.hook_resid_post' = layer output
        lm.model.layers[l].register_forward_hook(
            lambda mod, inp, out, slot=slot: pool.observe(slot, out[0] if isinstance(out, tuple) else out))
    ens = pra.HipProberEnsemble(len(layers), 2048, 2, weights="f32")
    for slot in range(len(layers)):
        ens.load_layer(slot, cases.synth_state(100 + slot, 2048))
    index = pra.HipFlatIndex(768, "l2", "f16", capacity=args.docs)
    index.add_synthetic(42, 0, args.docs)

    ev = {k: [] for k in ("generate", "gate", "retrieve")}

    def timed(kind, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        ev[kind].append((a, b))
        return out

    rng = np.random.default_rng(1)
    counts = []
    for qi in range(args.queries):
        first = torch.from_numpy(rng.integers(5, 250000, size=(1, args.prompt_len))).to(dev)

        def generate(ids):
            return timed("generate", lambda: lm.generate(ids, max_new_tokens=args.new_tokens, do_sample=False, use_cache=True,
                                                         pad_token_id=0))

        def gate():
            _, _, dec = timed("gate", lambda: ens.gate(pool.pooled(), ablation=0, threshold=args.theta))
            return int(dec[0])

        def retrieve(text, k):
            q = torch.from_numpy(synth_rows(900 + qi, len(ev["retrieve"]), 1, 768)).to(dev)
            return timed("retrieve", lambda: index.search(q, k))

        pred, rc = pra.retrieve_decide(
            "question", first, generate=generate, gate=gate, retrieve=retrieve,
            lookup=lambda ids: [f"passage {i}" for i in ids],
            make_prompt=lambda q, evid: evid,
            tokenize=lambda s: torch.cat([first, torch.from_numpy(rng.integers(5, 250000, size=(1, 5 * 100))).to(dev)], 1),
            to_string=lambda out: ["decoded text"], reset=pool.reset, k=5)
        counts.append(rc)
    torch.cuda.synchronize()
    ms = {k: np.array([a.elapsed_time(b) for a, b in v]) for k, v in ev.items()}
    print(f"{args.queries} queries, retrieval rounds per query: {counts}")
    for k in ("generate", "gate", "retrieve"):
        if len(ms[k]):
            print(f"  {k:9s}: {len(ms[k]):3d} calls, median {np.median(ms[k]):9.3f} ms, total {ms[k].sum():9.1f} ms")
    tot = sum(m.sum() for m in ms.values())
    print(f"  gate + retrieval = {100 * (ms['gate'].sum() + ms['retrieve'].sum()) / tot:.2f} % of the loop's GPU time "
          f"(index: {args.docs} x 768 fp16 rows; generation: {args.prompt_len}(+500) prompt tokens, {args.new_tokens} new tokens, "
          "HF generate, eager PyTorch-ROCm)")


if __name__ == "__main__":
    main()
