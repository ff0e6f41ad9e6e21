"""probing_rag_amd — MI355X (gfx950) implementation of Probing-RAG's
retrieval-gating hot path: fused prober ensemble + gate, and the flat dense
index (scan + top-k), behind the reference's own call signatures.

Everything computes through libprag.so (HIP kernels, C ABI in include/prag.h);
importing this package does not need a GPU, using it does.
"""
from ._lib import PragError, build, lib  # noqa: F401
from .prober import (Config_Maker, HipProber, HipProberEnsemble, gate_from_logits,  # noqa: F401
                     load_prober, load_prober_cfg_gemma_2b, load_prober_models, prober_checkpoint_path,
                     return_prober_logit_gemma_2b)
from .index import (HipFlatIndex, IndexFlatIP, IndexFlatL2, batch_topk_sim, encode_query,  # noqa: F401
                    find_topk_sim, merge_topk, plan_search, read_index, read_index_header, write_index)
from .docstore import Docstore, lookup_passages, read_docstore, write_docstore  # noqa: F401
from .sharded import ShardedFlatIndex, partition_rows, search_shards_on_one_gpu  # noqa: F401
from .trainer import HipProberTrainer, method_1_train, method_2_train, method_3_train  # noqa: F401
from .encoder import MeanPoolEncoder  # noqa: F401
from .loop import (HiddenStatePool, masked_mean_pool, method_1_eval, method_2_eval, method_3_eval, pool_each_token,  # noqa: F401
                   pool_last_token, pool_ragged, retrieve_decide, return_evidences, search_and_gate)

ImprovedProbe = HipProber  # utils.py:29
__version__ = "0.1.0"
