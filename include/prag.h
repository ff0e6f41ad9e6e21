/*
 * prag.h — C ABI of libprag.so, the MI355X (gfx950) implementation of the
 * Probing-RAG retrieval-gating hot path.
 *
 * The reference (baekingeol/Probing-RAG, paths relative to its root) is pure
 * Python with no FFI of its own; the boundary it offers is a handful of
 * duck-typed call sites.  Each entry point below names the reference call
 * site it replaces.  The Python host side (probing-rag_amd/) binds these
 * with ctypes and re-creates the reference's objects (`prober(x)`,
 * `index.search(q, k)`, `batch_topk_sim(...)`) on top of them; INTEGRATION.md
 * shows the binding a reference maintainer would add.
 *
 * Conventions
 *  - every call returns PRAG_OK (0) or a negative PRAG_E* code and never
 *    throws; prag_last_error() returns a thread-local message for the last
 *    failure on the calling thread;
 *  - `*_dev` pointers are device pointers BORROWED for the duration of the
 *    stream work (tensor.data_ptr()); the caller keeps them alive;
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream()
 *    .cuda_stream); all kernels are launched on it and no call synchronises
 *    the device unless its comment says so;
 *  - handles own their device buffers; a handle is not thread-safe (the
 *    reference loop is single-threaded, exp_rag.py:396).
 */
#ifndef PRAG_H
#define PRAG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRAG_VERSION 100 /* 0.1.0 */

/* status codes */
#define PRAG_OK 0
#define PRAG_EINVAL (-1)       /* bad argument / shape */
#define PRAG_EHIP (-2)         /* a HIP runtime call failed */
#define PRAG_ENOMEM (-3)
#define PRAG_EUNSUPPORTED (-4) /* valid request the kernels do not cover */
#define PRAG_ESTATE (-5)       /* e.g. forward before all layers are loaded */

/* element types of caller-provided activations / stored corpus rows */
#define PRAG_F32 0
#define PRAG_F16 1
#define PRAG_BF16 2   /* activations only (hidden states of a bf16 LM); handled like PRAG_F32 */

/* weight precision of a prober handle */
#define PRAG_W_F16 1   /* weights rounded to 11 significant bits (one fp16 MFMA term); fc2's low
                        * activation bits meet an fp8 copy of W2: logits within ~2e-5 of float64 on the
                        * same operands (PRAG_W_F32: ~2e-6)                                             */
#define PRAG_W_F32 0   /* weights kept to ~22 bits as hi+lo fp16 terms (fp32 parity)    */

/* similarity metrics of the flat index */
#define PRAG_METRIC_L2 0   /* squared L2, ascending  == faiss.IndexFlatL2 (make_indexer.py:450) */
#define PRAG_METRIC_IP 1   /* inner product, descending == faiss.IndexFlatIP                 */
#define PRAG_METRIC_COS 2  /* rows and queries L2-normalised, then IP (BASELINE config 3)    */

int prag_version(void);
const char* prag_last_error(void);

/* ------------------------------------------------------------------------
 * Prober ensemble + gate
 * ---------------------------------------------------------------------- */
typedef struct prag_prober prag_prober_t;

/* Replaces `ImprovedProbe(input_size=cfg.d_model, output_size=cfg.num_classes)`
 * x n_layers (utils.py:302, 385-387).  d_hidden must be 512 and n_classes 2
 * (utils.py:30, 289); d_model a multiple of 64. */
int prag_prober_create(prag_prober_t** out, int n_layers, int d_model, int d_hidden,
                       int n_classes, int weight_mode);

/* Replaces `prober.load_state_dict(torch.load(path))` + `.eval()`
 * (utils.py:303-329) for one layer.  Host float32 arrays in the reference's
 * state-dict layout ([out,in] row-major weights).  The data is copied (and
 * LayerNorm affines are folded into the following Linear) before return;
 * synchronises the device. */
int prag_prober_load_layer(prag_prober_t* p, int layer_idx,
                           const float* ln0_w, const float* ln0_b,
                           const float* W1, const float* b1,
                           const float* ln1_w, const float* ln1_b,
                           const float* W2, const float* b2,
                           const float* ln2_w, const float* ln2_b,
                           const float* W3, const float* b3);

/* Replaces `logit = prober(input)` (exp_rag.py:387, utils.py:45-57) for
 * `n_run` consecutive layers starting at `layer0`, in ONE launch.
 *   x_dev: activations, element type x_dtype (PRAG_F32 | PRAG_F16 | PRAG_BF16); row b of
 *          layer l starts at element (l - layer0) * x_layer_stride + b * d_model
 *   logits_dev: float32 [n_run, B, 2]                                        */
int prag_prober_forward(prag_prober_t* p, const void* x_dev, int x_dtype,
                        int64_t x_layer_stride, int layer0, int n_run, int B,
                        float* logits_dev, void* stream);

/* Replaces return_prober_logit_gemma_2b + the softmax / sum / threshold of
 * exp_rag.py:406-415 for a whole batch: all layers' probers, then
 * probsum[b] = sum_{n>=ablation} softmax(logits[n,b]) (float32, layer order)
 * and decision[b] = ((double)probsum[b,0] + theta < (double)probsum[b,1]) ? 0 : 1
 * (1 = retrieve; the reference compares Python floats, exp_rag.py:414).
 * logits_dev [L,B,2], probsum_dev [B,2], decision_dev [B]. */
int prag_gate(prag_prober_t* p, const void* x_dev, int x_dtype, int64_t x_layer_stride,
              int B, int ablation, double theta, float* logits_dev, float* probsum_dev,
              int32_t* decision_dev, void* stream);

/* The gate as the retrieve-decide loop consumes it: exp_rag.py:393, 406-415 end with a HOST branch on the two sums
 * (`if logit_sum[0] + theta < logit_sum[1]`), once per generation, on ONE pooled state per layer.  prag_gate followed
 * by a device->host copy of the decision and a stream wait in one call: the logits stay in a buffer of the handle,
 * decision_host int32 [B] (plain host memory, 1 = retrieve) is filled before the call returns; probsum_host float32
 * [B,2] is optional (the sums the reference prints, exp_rag.py:420).  Nothing is allocated per call; for B <= 256 the
 * gate's last kernel writes into pinned host memory itself and the call waits for those words only (the stream is
 * not otherwise synchronised; with probsum_host the call waits for the stream). */
int prag_gate_decide(prag_prober_t* p, const void* x_dev, int x_dtype, int64_t x_layer_stride, int B, int ablation,
                     double theta, int32_t* decision_host, float* probsum_host, void* stream);

/* The gate arithmetic alone (exp_rag.py:407-415) on existing logits. */
int prag_gate_from_logits(const float* logits_dev, int L, int B, int ablation, double theta,
                          float* probsum_dev, int32_t* decision_dev, void* stream);

/* Test/inspection hook: the weights the kernels actually compute with, as a
 * state dict whose LayerNorm affines are identity (folded into W/b).  Host
 * float32 outputs: W1 [512,d], b1 [512], W2 [512,512], b2 [512], W3 [2,512],
 * b3 [2].  Any pointer may be NULL. */
int prag_prober_effective_weights(prag_prober_t* p, int layer_idx, float* W1, float* b1,
                                  float* W2, float* b2, float* W3, float* b3);

/* Pre-size the internal activation workspace for batches up to max_B rows so
 * that later forward calls never allocate (stream-capture safe). */
int prag_prober_reserve(prag_prober_t* p, int max_B);

/* Measurement hook (no reference counterpart): record a HIP-event pair on the
 * launch stream around every fused prober kernel (slots > 0 enables and sizes
 * the ring, 0 disables).  prag_prober_profile_read waits for the recorded
 * events, writes up to `cap` durations in ms, returns their count in *n_out and
 * rewinds the ring. */
int prag_prober_profile(prag_prober_t* p, int slots);
int prag_prober_profile_read(prag_prober_t* p, float* ms, int cap, int* n_out);

void prag_prober_destroy(prag_prober_t* p);

/* Replaces `torch.sum(torch.concat(cache[name][1:], dim=1), dim=1)`
 * (exp_rag.py:385-386): acc[l, b, :] (+)= h[l, b, :] for one decode step, on
 * device, so hooks need no D2H copy.
 * acc_dev float32 [L,B,d]; h_dev element type h_dtype (PRAG_F32 / PRAG_F16 / PRAG_BF16), same layout; if
 * `assign` != 0 the accumulator is overwritten instead of added to. */
int prag_pool_accumulate(float* acc_dev, const void* h_dev, int h_dtype, int64_t n_elems,
                         int assign, void* stream);

/* The same for every probed layer of one decode step in ONE launch (exp_rag.py:317-329 registers one hook per
 * layer of range(6,17,2): six `activations.detach().cpu()` per generated token in the reference, six launches per
 * token with the call above): acc_dev float32 [n_layers, n_elems]; h_dev_ptrs is a HOST array of n_layers device
 * pointers (wherever the model left each layer's activations; n_elems elements of type h_dtype each).
 * n_layers <= 64, n_elems a multiple of 4. */
int prag_pool_accumulate_layers(float* acc_dev, const void* const* h_dev_ptrs, int n_layers, int h_dtype,
                                int64_t n_elems, int assign, void* stream);

/* One decode step of the hooked layers AND the gate on the sums so far, in ONE launch (round 6).  The reference's hooks
 * copy each probed layer's activations to the host on every forward pass (exp_rag.py:317-329) and the gate - cat / sum,
 * six probers, softmax / sum / threshold, exp_rag.py:381-389, 406-415 - only starts after `generate` has returned.
 * Here the launch that adds a decode step's activations to the running sums (prag_pool_accumulate_layers' one launch
 * per token) also runs the B <= 4 gate on the sums it has just formed and leaves decision, sums and a tag in pinned host
 * memory: the decision of the last decode step is on the host by the time `generate` returns, computed in the shadow of
 * the LM's remaining layers; the loop's `if` (exp_rag.py:414) reads it with prag_gate_step_result.  The decision is
 * bit-identical to prag_gate_decide on acc_out (same kernels' arithmetic, same order).
 *   acc_in_dev / acc_out_dev  float32 [L][B][d_model]: sums before / after this step - two DIFFERENT buffers (the
 *                             caller alternates them); acc_in is not read when assign != 0 (first step after a reset)
 *   h_dev_ptrs                HOST array of L device pointers: each layer's [B][d_model] activations, type h_dtype
 *   *tag_out                  names this step for prag_gate_step_result
 * Work on `stream`; the handle's workspaces are shared with prag_gate*, so use one stream per handle.
 * PRAG_EUNSUPPORTED when prag_gate would not take the small-batch kernels for this B (B > 4, or > 2 with PRAG_W_F16),
 * d_model > 4096 or L > 16: the caller uses prag_pool_accumulate_layers + prag_gate_decide. */
int prag_pool_step_gate(prag_prober_t* p, const float* acc_in_dev, float* acc_out_dev, const void* const* h_dev_ptrs,
                        int h_dtype, int B, int assign, int ablation, double theta, uint64_t* tag_out, void* stream);

/* exp_rag.py:393, 406-415's host branch for the step `tag` (the handle's most recent prag_pool_step_gate; anything
 * else: PRAG_ESTATE): decision_host int32 [B] (1 = retrieve), probsum_host float32 [B,2] or NULL.  Polls the tag word
 * the launch writes last (~200 us), then waits for `stream`.  PRAG_ESTATE also when a hand-off inside the launch gave
 * up after ~50 ms (another kernel held the chip): the sums are intact, decide on them with prag_gate_decide. */
int prag_gate_step_result(prag_prober_t* p, uint64_t tag, int B, int32_t* decision_host, float* probsum_host,
                          void* stream);

/* Replaces input_tensor_method1 + per-sample mean (train.py:153-162, 202-205;
 * utils.py:134-143, 184-186): out[b,:] = mean over the last pred_lens[b]
 * positions of acts[b] ([B,T,d], element type dtype: PRAG_F32 / PRAG_F16 / PRAG_BF16).  out float32 [B,d];
 * pred_lens_dev int64 [B].  scale_mean=0 gives the inference-time sum pool. */
int prag_pool_ragged(const void* acts_dev, int dtype, int B, int T, int d,
                     const int64_t* pred_lens_dev, int scale_mean, float* out_dev, void* stream);

/* Replaces `_input_tensor_method1` (train.py:153-162, utils.py:134-143) - the input of the `each_token` training
 * method, train.py:354's default (method_1_train / method_1_eval, train.py:182-197): the last pred_lens[b] positions
 * of every sample concatenated in sample order, out float32 [n_rows, d], and `torch.repeat_interleave(labels,
 * pred_lens)` -> labels_out int32 [n_rows] (labels_dev int32 [B]; both may be NULL).  row_offsets_dev int64 [B+1] =
 * exclusive prefix sums of pred_lens (clamped to [0, T]), row_offsets[B] = n_rows: the host knows them, it sizes
 * `out`.  acts [B,T,d] of element type dtype. */
int prag_pool_each_token(const void* acts_dev, int dtype, int B, int T, int d, const int64_t* row_offsets_dev,
                         int64_t n_rows, const int32_t* labels_dev, float* out_dev, int32_t* labels_out_dev,
                         void* stream);

/* Replaces the attention-masked mean pooling at the end of
 * `SentenceTransformer('facebook/contriever-msmarco').encode` (utils.py:365-366,
 * make_indexer.py:447-455; third-party code, restated from its published
 * definition): out[b,:] = sum_t mask[b,t] * h[b,t,:] / max(sum_t mask[b,t], 1).
 * hidden_dev [B,T,d] (PRAG_F32 | PRAG_F16 | PRAG_BF16), mask_dev int64 [B,T],
 * out_dev float32 [B,d] - ready to hand to prag_index_search on the device. */
int prag_pool_masked_mean(const void* hidden_dev, int dtype, const int64_t* mask_dev, int B, int T, int d,
                          float* out_dev, void* stream);

/* ------------------------------------------------------------------------
 * Prober training step
 * ---------------------------------------------------------------------- */
typedef struct prag_trainer prag_trainer_t;

/* One ImprovedProbe (utils.py:29-57) with its optimiser state.  Replaces
 *   probe = ImprovedProbe(d_model, num_classes); optimizer = AdamW(probe.parameters(), lr=lr);
 *   scheduler = ExponentialLR(optimizer, gamma=0.995)              (train.py:126-135)
 * torch.optim.AdamW defaults are beta1 0.9, beta2 0.999, eps 1e-8, weight_decay 0.01;
 * dropout_p is 0.1 in the reference (utils.py:39).  `seed` keys the dropout masks
 * (counter hash of (seed, step, site, row, unit): reproducible, not torch's RNG stream). */
int prag_trainer_create(prag_trainer_t** out, int d_model, int d_hidden, int n_classes, double lr, double beta1,
                        double beta2, double eps, double weight_decay, double gamma, double dropout_p,
                        uint32_t seed);

/* Initial parameters: host fp32 tensors in state-dict order (as prag_prober_load_layer).
 * Copies them and resets the optimiser state and the step counter. */
int prag_trainer_load(prag_trainer_t* t, const float* ln0_w, const float* ln0_b, const float* W1, const float* b1,
                      const float* ln1_w, const float* ln1_b, const float* W2, const float* b2, const float* ln2_w,
                      const float* ln2_b, const float* W3, const float* b3);

/* Replaces method_2_train's `loss.backward(); optim.step(); scheduler.step(); optim.zero_grad()`
 * (train.py:210-220, utils.py:191-197) including the forward of _method_2_util after pooling:
 * x_dev float32 [B,d_model] (per-sample mean of the last pred_len hidden states, see
 * prag_pool_ragged), labels_dev int32 [B].  Optional device outputs of the forward pass:
 * loss_dev (1 float: CrossEntropyLoss applied to the softmax probabilities, train.py:149-150)
 * and probs_dev float32 [B,n_classes].  All work is enqueued on `stream`.
 * In eval mode (prag_trainer_set_training(t, 0)) the call is the forward pass only: parameters,
 * optimiser state, learning rate and step counter are left untouched. */
int prag_trainer_step(prag_trainer_t* t, const float* x_dev, const int32_t* labels_dev, int B, float* loss_dev,
                      float* probs_dev, void* stream);

/* Current parameters to host fp32 tensors (state-dict order): `probe.state_dict()`. */
int prag_trainer_export(prag_trainer_t* t, float* ln0_w, float* ln0_b, float* W1, float* b1, float* ln1_w,
                        float* ln1_b, float* W2, float* b2, float* ln2_w, float* ln2_b, float* W3, float* b3);

/* `probe.train()` / `probe.eval()` (train.py:255-257, 299): with training == 0 the dropout of the
 * following steps is the identity and prag_trainer_step stops after the forward pass.  Handles start in
 * training mode. */
int prag_trainer_set_training(prag_trainer_t* t, int training);

/* Learning rate the next step will use (`optim.param_groups[0]['lr']`), steps taken so far. */
double prag_trainer_lr(const prag_trainer_t* t);
int64_t prag_trainer_steps(const prag_trainer_t* t);
void prag_trainer_destroy(prag_trainer_t* t);

/* ------------------------------------------------------------------------
 * Flat (exact, brute-force) index
 * ---------------------------------------------------------------------- */
typedef struct prag_index prag_index_t;

/* Replaces `faiss.IndexFlatL2(768)` (make_indexer.py:449-450).  store_dtype:
 * PRAG_F32 keeps rows as given (reference parity), PRAG_F16 rounds them to
 * fp16 (half the scan bytes).  `capacity_rows` > 0 pre-allocates. */
int prag_index_create(prag_index_t** out, int d, int metric, int store_dtype,
                      int64_t capacity_rows);

/* Replaces `index.add(model.encode(texts[...]))` (make_indexer.py:455): append
 * n rows of float32 [n,d] (host pointer if src_is_device == 0).  Row ids are
 * insertion order, 0-based.  The copy / conversion runs on `stream` (device rows may
 * still be in flight there, e.g. an encoder's output) and the call returns after
 * synchronising that stream. */
int prag_index_add(prag_index_t* ix, const float* x, int64_t n, int src_is_device, void* stream);

/* Append n synthetic rows generated on the device by the counter-based
 * generator shared with the oracle (oracle_np.synth_rows): row ids
 * [row0, row0+n) of stream `seed`.  Bench / test helper. */
int prag_index_add_synthetic(prag_index_t* ix, uint32_t seed, int64_t row0, int64_t n);

int64_t prag_index_ntotal(const prag_index_t* ix);
int prag_index_d(const prag_index_t* ix);

/* Replaces `D, I = index.search(x, k)` (utils.py:379; exp_rag.py:432-436).
 *   q: float32 [B,d] (device pointer if io_is_device, else host);
 *   D: float32 [B,k] squared-L2 ascending (L2) or score descending (IP/COS);
 *   I: int64 [B,k] row ids + id_offset, -1 padded when ntotal < k.
 * k <= 911.  The result is the exact brute-force answer on the stored rows, always (what
 * faiss.IndexFlat guarantees): the matrix-core scan only proposes candidates, the float64 rerank
 * scores them and CERTIFIES per query that no other row can reach the top k (error bound of the
 * selection vs the gap to the last candidate); a query that cannot be certified - dense
 * near-duplicates, squared-L2 cancellation, a deep-list overflow - is recomputed by an exact
 * float64 scan of every row inside the same call (one more pass over the shard for that query).
 * No host synchronisation on the device-io path: every fallback - the exact scan, the second tier of the
 * 8-bit tiled selection (prag_index_last_tiled8 below) - is enqueued unconditionally and switched by a word
 * on the device, so a search whose workspaces exist (any earlier search of the same shape made them) can be
 * captured into a HIP graph and replayed, and every rank of a lockstep retrieval issues the same COLLECTIVES (one
 * all-gather per sharded search).  The kernel sequence itself may differ between ranks and between runs of one input:
 * whether a > 128-query search takes the int8 tiles is per-process host state fed by an asynchronous copy of the
 * previous searches' failed counts (prag_index_last_tiled8), and the workgroup count of a two-level scan of <= 64
 * queries (7/8 of the CUs or all of them) is timed on each index's own first searches (PRAG_SCAN_WG_TUNE=0|1 pins it;
 * prag_index_last_plan reports what ran) - results are the definition's either way.
 * With io_is_device == 0 the call copies in/out and synchronises `stream`. */
int prag_index_search(prag_index_t* ix, const float* q, int B, int k, int64_t id_offset,
                      float* D, int64_t* I, int io_is_device, void* stream);

/* The exchange step of the row-sharded index: merge `n_parts` per-shard
 * results (D_parts/I_parts laid out [n_parts, B, k], e.g. straight out of an
 * RCCL all-gather) by (score, id) into the global top-k.  Device pointers. */
int prag_merge_topk(const float* D_parts_dev, const int64_t* I_parts_dev, int n_parts, int B,
                    int k, int metric, float* D_dev, int64_t* I_dev, void* stream);

/* Same merge for results that were exchanged as ONE buffer per shard: part p
 * starts at parts_dev + p * part_stride_bytes and holds D float32 [B,k] followed
 * (at the next multiple of 8 bytes) by I int64 [B,k] - what prag_index_search
 * writes when D/I point into such a buffer - so the exchange is a single
 * all-gather. */
int prag_merge_topk_packed(const void* parts_dev, int64_t part_stride_bytes, int n_parts, int B, int k,
                           int metric, float* D_dev, int64_t* I_dev, void* stream);

/* The same call and the same merge for results that cross shards ("tagged" ids).  Only float32 scores
 * travel between shards, so two rows of different shards whose float64 scores round to the same float32
 * would be ordered by id in the merge while an unsharded search (like the float64 definition the oracle
 * states) orders them by the float64 value - at the k-th place that can even change
 * which row is returned.  prag_index_search_tagged writes, above the low 40 bits of each global row id, 23
 * order-preserving bits of the float32 residual score64 - (double)D; prag_merge_topk_packed_tagged
 * compares (D, residual, id) and returns plain ids.  Global ids must be < 2^40.  Tagged ids are an
 * exchange format: nothing else reads them.  (The reference has no counterpart: faiss-cpu, one process,
 * exp_rag.py:248, 432.) */
int prag_index_search_tagged(prag_index_t* ix, const float* q, int B, int k, int64_t id_offset, float* D, int64_t* I,
                             int io_is_device, void* stream);
int prag_merge_topk_packed_tagged(const void* parts_dev, int64_t part_stride_bytes, int n_parts, int B, int k,
                                  int metric, float* D_dev, int64_t* I_dev, void* stream);

/* The exchange in C: a C / ctypes host shards the index without torch.distributed.
 * prag_index_set_comm hands the index the caller's RCCL communicator (an ncclComm_t, borrowed; NULL with world 1 =
 * single GPU), this process's rank and the world size.  prag_index_search_sharded is then the whole sharded search
 * of utils.py:378-380's call in one entry point: local search with tagged ids into this rank's packed block,
 * ONE ncclAllGather of world x (B*k*12 bytes, padded to 16) on `stream`, the (score, residual, id) merge; q
 * replicated on every rank, identical D / I on every rank, device pointers, no host synchronisation.  id_offset =
 * global id of this shard's first row.  RCCL is bound at run time (dlopen): hosts that cannot reach a communicator
 * of their own (Python: torch.distributed does not expose its ncclComm_t) create one with the three helpers -
 * prag_rccl_unique_id on one rank, the 128 bytes broadcast by any means, prag_rccl_comm_init_rank on every rank
 * (collective, on the current device). */
int prag_rccl_unique_id(void* id_out_128);
int prag_rccl_comm_init_rank(void** comm_out, int world, int rank, const void* id_128);
/* ncclAllGather of `bytes_per_rank` device bytes per rank on `stream` (the exchange step of the sharded search by
 * itself: ShardedFlatIndex checks a new communicator with it before any search relies on it). */
int prag_rccl_all_gather(void* comm, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream);
int prag_rccl_comm_destroy(void* comm);
int prag_index_set_comm(prag_index_t* ix, void* nccl_comm, int rank, int world);
int prag_index_search_sharded(prag_index_t* ix, const float* q_dev, int B, int k, int64_t id_offset,
                              float* D_dev, int64_t* I_dev, void* stream);

/* Read back stored rows [row0,row0+n) as float32 (what the scan sees). */
int prag_index_reconstruct(prag_index_t* ix, int64_t row0, int64_t n, float* out_host);

/* Performance knob only (results are exact at every depth): the fused scan keeps KC >= k candidates
 * per query; a deeper list certifies more queries on near-duplicate-dense corpora without the exact
 * fallback pass.  depth = 0 picks the default (8 for k<=5, 16 for k<=12, 32 for k<=26); 16 or 32
 * force a deeper candidate list. */
int prag_index_set_candidate_depth(prag_index_t* ix, int depth);

/* Measurement hook: how many queries of the most recent search on this handle could not be
 * certified and went through the exact float64 scan.  Synchronises `stream`. */
int prag_index_last_fallbacks(prag_index_t* ix, void* stream, int* n_out);

/* One pass of the hot path as one call: `D, I = index.search(q, k)` (utils.py:379; exp_rag.py:432-436) AND the gate over
 * the NEXT batch of pooled states (exp_rag.py:406-415) - two independent pieces of work of a loop that keeps batches
 * in flight.  Arguments = those of prag_index_search (device i/o) followed by those of prag_gate.  When the search is
 * a two-level search the prober's workgroups are carried by a launch of the search: behind the scan's own workgroups
 * when the gate fits under the scan (an HBM-bound scan runs on 7/8 of the CUs; workgroups of one launch are placed in
 * index order, so the scan's settle first and the prober's take the CUs left), else beside the bound kernel - what
 * follows the scan leaves 3/4 of the chip idle; otherwise the call equals prag_index_search followed by prag_gate.
 * (PRAG_SCAN_GATE=0|1 forces the second / first form; prag_index_last_plan names the launch.)  Results are
 * those of the two calls in every case; Bg = 0 skips the gate.  flags bit 0: ids in the exchange format of
 * prag_index_search_tagged (a row shard of a sharded index: the caller all-gathers and merges). */
int prag_search_and_gate(prag_index_t* ix, const float* q_dev, int B, int k, int64_t id_offset, float* D_dev, int64_t* I_dev,
                         prag_prober_t* p, const void* x_dev, int x_dtype, int64_t x_layer_stride, int Bg, int ablation,
                         double theta, float* logits_dev, float* probsum_dev, int32_t* decision_dev, int flags,
                         void* stream);

/* Pipelining hook: make `other_stream` wait until the CORPUS SCAN of the most recent search enqueued on this handle
 * has finished - not the whole search.  What follows the scan (the exact bound / rerank of the survivors, the fallback
 * probes) runs on a few dozen workgroups; independent work of the caller - exp_rag.py's gate for the NEXT batch of
 * generations (exp_rag.py:406-415) - launched on `other_stream` after this call runs beside that tail instead of behind
 * it.  The first call on a handle only switches the recording on (searches enqueued afterwards record an event behind
 * their scan) and waits for nothing.  Capturable: inside a stream capture the wait becomes a fork edge. */
int prag_index_stream_wait_scan(prag_index_t* ix, void* other_stream);

/* Measurement hook: how many rows the proof-carrying filter of the most recent two-level search let through to the
 * exact rerank (the last query tile of that search: total over its queries, the largest count of one query, the
 * number of queries counted; 0 queries: that search scanned the rows directly).  Synchronises `stream`.  (No reference
 * counterpart: faiss.IndexFlat scores every row, make_indexer.py:449-450.) */
int prag_index_last_survivors(prag_index_t* ix, void* stream, int64_t* total_out, int* max_per_query_out,
                              int* n_queries_out);

/* The search plan - which kernel family makes the corpus pass, query-tile height, candidate depth, grid, corpus
 * passes per search, algorithmic bytes per pass, workspace bytes - is a pure function of the request shape and the
 * index state.  prag_plan_search describes the plan of a hypothetical index (no GPU needed: d, metric, store_dtype,
 * rows, queries, k, whether it keeps an up-to-date 8-bit shadow (2 = shadow kept at any size), compute units);
 * prag_index_last_plan returns the plan the most recent search on this handle executed.  Both write a
 * NUL-terminated line of `key=value` fields; bench.py prices its roofline with it. */
int prag_plan_search(int d, int metric, int store_dtype, int64_t ntotal, int B, int k, int shadow_ready, int n_cu,
                     char* out, int cap);
int prag_index_last_plan(prag_index_t* ix, char* out, int cap);

/* Measurement hook: batches of > 128 queries on an index that keeps the 8-bit shadow (below) select their
 * candidates on int8 matrix tiles over the shadow first (256 candidates per query, the shadow's error bound
 * in the certificate); queries that fail that certificate are searched again - up to 64 of them (and at most
 * a quarter of the batch) as a compact batch of their own, more than that by repeating the whole batch on the
 * fp16 tiles.  Both continuations are always enqueued and the failed count, on the device, opens one of them:
 * the search never reads it back.  It travels to the host asynchronously; this hook waits for it.
 * *n_failed_out = queries of the most recent search that failed the 8-bit certificate (0: the first tier
 * answered); -1: that search did not take the 8-bit tiles (fewer than 2 Mi rows in shadow mode 1, no shadow,
 * <= 128 queries); -2: it skipped them because two searches in a row had to repeat their whole batch (they are
 * probed again after 64, 128, ... 4096 eligible searches, or at once when rows are added or
 * prag_index_set_shadow is called); -3: unknown - the search was captured into a graph.  Reference call:
 * utils.py:378-380 (batch_topk_sim -> IndexFlat.search); results are the definition's either way. */
int prag_index_last_tiled8(prag_index_t* ix, int* n_failed_out);

/* Two-level exact search: keep an 8-bit shadow of the stored rows (+d+12 bytes per row, built on the
 * device as rows are added - `prag_index_add*` extend it before they return, so no search pays for
 * the build) and scan IT for batches of <= 128 queries (d a multiple of 128, <= 1024,
 * k <= 26; larger batches: prag_index_last_tiled8 above): half the bytes of fp16 storage, a quarter of float32.  Results do not change: a
 * proof-carrying filter (Cauchy-Schwarz bound on the quantisation error of every row) keeps each row
 * that can still belong to the top k, the survivors are scored in float64 from the stored rows, and a
 * query whose candidate store overflows is recomputed by the exact scan.
 * mode 0 = off (the stored rows are scanned directly), 1 = shards of >= 2^20 rows when the device has
 * room for the shadow next to the rows (the default), 2 = any size. */
int prag_index_set_shadow(prag_index_t* ix, int mode);

/* Bring every derived structure of the index up to date with its rows on `stream` (the 8-bit shadow
 * when the current mode and shard size call for one, max ||x||^2 for the exactness bounds), so that
 * the first search after a mode change does not pay for it.  `prag_index_add*` already do this; the
 * call is needed only after `prag_index_set_shadow` switched the shadow on for rows that were added
 * while it was off.  (faiss has no counterpart: IndexFlat::add, make_indexer.py:455, is the whole
 * build.) */
int prag_index_prepare(prag_index_t* ix, void* stream);

/* Allocate every per-search workspace a search of up to B queries and this k needs, now (throw-away searches on
 * `stream`, waited for).  The index otherwise grows its workspaces inside the first search of a
 * larger shape - a hipFree / hipMalloc, i.e. a device synchronisation, and not capturable into a graph.  After this
 * call searches of that shape allocate nothing.  The sizing searches use a fixed pseudo-random query pattern, leave the
 * statistics of the int8-tile heuristic untouched and cover both of its settings.  (No reference counterpart; faiss
 * allocates per call.) */
int prag_index_reserve(prag_index_t* ix, int B, int k, void* stream);

/* The deterministic plan (round 6).  By default a handle adapts to its own history: the workgroup count of a two-level
 * scan of <= 64 queries is timed on the index's first searches, the retry tier and the sliced gather are armed by what
 * earlier searches flagged, the int8 tiles switch themselves off after two whole-batch repeats - all result-safe, but the
 * launch sequence then differs from run to run and from rank to rank.  on = 0 (or PRAG_ADAPTIVE=0 in the environment
 * when the handle is created) freezes all of it: 7/8 of the CUs for an HBM-bound two-level scan, flagged queries of a
 * device-io search go straight to the float64 scan (a host-io search still runs the retry tier on its OWN flag count),
 * the gather is always enqueued, the int8 tiles stay on.  A search then issues the same launches on every rank and in
 * every run of one input (prag_index_last_plan says `adaptive=0`).  Results are the definition's either way.
 * (No reference counterpart: faiss.IndexFlat has one code path, make_indexer.py:449-450.) */
int prag_index_set_adaptive(prag_index_t* ix, int on);

/* Cap the number of workgroups (= CUs) the scan kernels occupy; 0 = all CUs.  The scan is
 * HBM-bound, so leaving a few CUs free lets an independent kernel on another stream (e.g. the
 * prober gate of the next batch) run concurrently instead of queueing behind it. */
int prag_index_set_scan_workgroups(prag_index_t* ix, int n_workgroups);

/* Measurement hook: as prag_prober_profile, around every scan_topk launch. */
int prag_index_profile(prag_index_t* ix, int slots);
int prag_index_profile_read(prag_index_t* ix, float* ms, int cap, int* n_out);
/* The same ring around the ncclAllGather of every prag_index_search_sharded (enabled by prag_index_profile on an
 * index that holds a communicator): the exchange step of the sharded search by itself, in ms per call. */
int prag_index_profile_read_exchange(prag_index_t* ix, float* ms, int cap, int* n_out);

void prag_index_destroy(prag_index_t* ix);

#ifdef __cplusplus
}
#endif
#endif /* PRAG_H */
