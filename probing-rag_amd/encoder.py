"""Device-resident query encoder wrapper.

Reference: ``model_retr = SentenceTransformer('facebook/contriever-msmarco')`` (exp_rag.py:246-247)
and ``encode_query(model_retr, query) = model_retr.encode(query)`` (utils.py:365-366): a BERT-base
encoder followed by attention-masked MEAN pooling, no normalisation; the result goes to host
NumPy and into ``index.search``.

``MeanPoolEncoder`` keeps that ``.encode(list_of_str)`` surface but stays on the GPU: the
transformer runs in PyTorch-ROCm (plumbing - any module returning ``last_hidden_state``), the
pooling is libprag's ``prag_pool_masked_mean`` kernel, and the float32 ``[B,d]`` embeddings are
handed to ``HipFlatIndex.search`` as a CUDA tensor, so ``batch_topk_sim(model_retr, query, index,
k)`` never leaves the device.  Weights of contriever-msmarco are not in this image: the encoder
and tokenizer are whatever the caller passes in.
"""
from .loop import masked_mean_pool


class MeanPoolEncoder:
    def __init__(self, model, tokenizer, device="cuda", max_length: int = 512):
        self.model = model.to(device).eval()
        self.tokenizer = tokenizer
        self.device = device
        self.max_length = max_length

    def encode(self, sentences, convert_to_numpy: bool = False):
        """list[str] -> float32 [B,d] embeddings on the GPU; ONE str -> [d], as ``SentenceTransformer.encode`` returns
        it (``find_topk_sim``, utils.py:374-376, unsqueezes that itself).  ``convert_to_numpy=True`` gives
        SentenceTransformer.encode's host array."""
        import torch
        single = isinstance(sentences, str)
        if single:
            sentences = [sentences]
        enc = self.tokenizer(sentences, padding=True, truncation=True, max_length=self.max_length, return_tensors="pt")
        ids = enc["input_ids"].to(self.device)
        mask = enc["attention_mask"].to(self.device)
        with torch.no_grad():
            out = self.model(input_ids=ids, attention_mask=mask)
        hidden = out.last_hidden_state if hasattr(out, "last_hidden_state") else out[0]
        emb = masked_mean_pool(hidden, mask)
        if single:
            emb = emb[0]
        return emb.cpu().numpy() if convert_to_numpy else emb
