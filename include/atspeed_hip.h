/*
 * atspeed_hip.h — C-ABI of libatspeed_hip.so, the MI355X (gfx950) engine behind the
 * AtSpeed beam-speculative-decoding hot path.
 *
 * The reference (/root/reference, Linxyhaha/AtSpeed) has NO native layer and no FFI: its
 * "plugin surface" for this path is four Python callables.  Each entry point below names
 * the reference interface it replaces (file:line under /root/reference/code).  A
 * maintainer binds these with ctypes; atspeed_amd/_lib.py is that
 * binding.
 *
 * Conventions
 *   - every pointer named *_dev is DEVICE memory owned by the caller (PyTorch-ROCm
 *     tensors' data_ptr()); the library never allocates caller-visible memory.  Handles
 *     own only their private workspace / KV arena (create/destroy pairs).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are
 *     stream-ordered and never synchronise the device unless documented.
 *   - return value: 0 = ok, negative = atspeed_status; atspeed_last_error() gives the text
 *     (thread-local).
 *   - dtype: ATSPEED_F32 (fp32 weights/activations, exact-fp32 MFMA; parity mode),
 *     ATSPEED_BF16 (bf16 weights/activations/KV, fp32 accumulate, fp32 logits) or
 *     ATSPEED_F16 (the same engine on IEEE half: the type the reference loads both models in, code/inference.py:75-100
 *     `torch_dtype=torch.float16` -- a checkpoint's fp16 weights are used bit for bit; same kernels, v_mfma_f32_*_f16).
 */
#ifndef ATSPEED_HIP_H
#define ATSPEED_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum atspeed_status {
  ATSPEED_OK = 0,
  ATSPEED_ERR_INVALID = -1,      /* bad argument / shape the kernels do not support   */
  ATSPEED_ERR_HIP = -2,          /* a HIP runtime call failed                          */
  ATSPEED_ERR_CAPACITY = -3,     /* prompt/beam/slot count exceeds the handle's limits */
  ATSPEED_ERR_CONSTRAINT = -4,   /* a beam reached a node with no allowed token (HF raises ValueError there) */
  ATSPEED_ERR_NO_DEVICE = -5,
  ATSPEED_ERR_FILTERED = -6      /* every beam of a step fell to the post-top-k id filter of beamSD.py:80-86 (the reference then dies on a shape mismatch) */
} atspeed_status;

typedef enum atspeed_dtype { ATSPEED_F32 = 0, ATSPEED_BF16 = 1, ATSPEED_F16 = 2 } atspeed_dtype;

#define ATSPEED_MAX_BEAMS 64      /* K, DK <= 64 (one wavefront holds a beam list)       */
#define ATSPEED_MAX_NEW_TOKENS 16 /* generated-suffix capacity per beam                  */
#define ATSPEED_MAX_GAMMA 8

const char* atspeed_version(void);
const char* atspeed_last_error(void);
/* number of HIP devices visible (0 on a CPU-only host; never initialises a context) */
int atspeed_device_count(void);

/* ------------------------------------------------------------------ synthetic fill
 * value(i) = float(int(s_i) - 131070) * scale, s_i = sum of four 16-bit hash pieces of
 * (i + offset, seed) — bit-identical to atspeed_amd.synth.hash_normal.  `add` is added
 * after scaling (norm weights: 1 + jitter).  No reference counterpart (test/bench input). */
int atspeed_fill_hash_normal(void* dst_dev, size_t n, uint32_t seed, float scale, float add,
                             int dtype, uint64_t offset, void* stream);

/* ------------------------------------------------------------------ constraint automaton
 * Device form of the reference's mask functions — Trie.get() (generation_trie.py:27-70),
 * prefix_allowed_tokens_fn (generation_trie.py:92-98), the position-set function
 * (data.py:84-104) — which the reference calls once per beam per step on the host through
 * HF's PrefixConstrainedLogitsProcessor (beamSD.py:62-64,288-291).
 * CSR: node n allows tok[row_ptr[n]..row_ptr[n+1]) (strictly ascending), edge e leads to
 * node nxt[e].  Arrays are HOST pointers, copied to the device. */
typedef struct atspeed_fsm atspeed_fsm;
int atspeed_fsm_create(const int32_t* row_ptr, const int32_t* tok, const int32_t* nxt,
                       int32_t n_nodes, int32_t n_edges, int32_t vocab_size, atspeed_fsm** out);
void atspeed_fsm_destroy(atspeed_fsm* fsm);
/* The "automaton" of a call WITHOUT a mask: BSSD(..., logits_processor=None, prefix_allowed_tokens_fn=None) is legal in the reference
 * (beamSD.py:460-481: both optional; with an empty processor list :60-64 is the identity and the id filter :80-86 is skipped).  Every
 * token of the vocabulary is then a candidate of every beam; the decoder first reduces each logit row to its k best tokens
 * (atspeed_row_topk) and expands those.  Greedy mode only. */
int atspeed_fsm_create_free(int32_t vocab_size, atspeed_fsm** out);
/* one_step_beam_search drops, AFTER the top-k, every pick whose token is `< 32000 and != 2` (beamSD.py:80-86, hard-coded for the
 * Llama-2 vocabulary + item codes).  An automaton is created with those two numbers; other vocabularies set their own here
 * (min_item_token <= 0: keep every pick).  If a step loses ALL its beams to the filter the generate call returns ATSPEED_ERR_FILTERED. */
int atspeed_fsm_set_id_filter(atspeed_fsm* fsm, int32_t min_item_token, int32_t eos_token);
/* host-side flattening of token sequences into the CSR above (breadth-first node ids,
 * children ascending): native counterpart of Trie.__init__/_add_to_trie
 * (generation_trie.py:8-14,40-44).  Two-call protocol: pass NULL outputs to get the sizes. */
int atspeed_trie_flatten(const int32_t* seq_tokens, const int32_t* seq_offsets, int32_t n_seqs,
                         int32_t* row_ptr_out, int32_t* tok_out, int32_t* nxt_out,
                         int32_t* n_nodes_out, int32_t* n_edges_out);

/* ------------------------------------------------------------------ model
 * Replaces the model object the path calls as model(**inputs) (beamSD.py:52,221):
 * a Llama decoder whose weights live in HBM.  Weight pointers are device pointers in the
 * handle's dtype, row-major [out, in] like HF nn.Linear:
 *   wqkv  [3*hidden, hidden]   rows = q_proj | k_proj | v_proj
 *   wo    [hidden, hidden]
 *   wgu   [2*ffn, hidden]      gate/up interleaved in 16-row groups: rows 32b..32b+15 =
 *                              gate[16b..], rows 32b+16..32b+31 = up[16b..]
 *   wd    [hidden, ffn]
 *   norm weights [hidden]; embed [vocab, hidden]; lm_head [vocab, hidden]           */
/* ATSPEED_WEIGHTS_ROW_MAJOR: [out, in] row-major as HF stores them.  ATSPEED_WEIGHTS_PACKED (bf16 only; hidden and ffn multiples of 32):
 * each matrix run through atspeed_pack_rows -- two consecutive rows share one 128-byte line per 64-byte block of the in dimension, which
 * is what lets every LDS-DMA piece of the projection GEMMs fetch FULL cache lines (83 against 55 GB/s per CU from L2, measured).  With
 * packed weights the engine keeps the activations a projection reads in the same layout internally; results are identical. */
typedef enum atspeed_weight_layout { ATSPEED_WEIGHTS_ROW_MAJOR = 0, ATSPEED_WEIGHTS_PACKED = 1 } atspeed_weight_layout;
/* row-major [rows][row_bytes] (row_bytes % 64 == 0) -> packed operand layout, out of place: byte b of row r goes to
 * ((r >> 1) * (row_bytes / 64) + (b >> 6)) * 128 + (r & 1) * 64 + (b & 63); dst holds rows rounded up to even (the pad row is zeroed).
 * atspeed_unpack_rows is the inverse.  Both 16-byte aligned device buffers. */
int atspeed_pack_rows(const void* src_dev, void* dst_dev, int32_t rows, int32_t row_bytes, void* stream);
int atspeed_unpack_rows(const void* src_dev, void* dst_dev, int32_t rows, int32_t row_bytes, void* stream);

typedef struct atspeed_llama_layer_weights {
  const void* input_norm;
  const void* wqkv;
  const void* wo;
  const void* post_norm;
  const void* wgu;
  const void* wd;
} atspeed_llama_layer_weights;

typedef struct atspeed_llama_config {
  int32_t vocab_size, hidden, n_layers, n_heads, ffn;
  float rope_theta, rms_eps;
  int32_t dtype;          /* atspeed_dtype */
  int32_t max_slots;      /* KV capacity (multiple of 64) */
  int32_t max_tokens;     /* max tokens per forward */
  int32_t max_logit_rows; /* max rows the lm_head is applied to */
  int32_t weight_layout;  /* atspeed_weight_layout of the projection weights and lm_head (embed and norms are always plain) */
} atspeed_llama_config;

typedef struct atspeed_llama atspeed_llama;
int atspeed_llama_create(const atspeed_llama_config* cfg, const void* embed_dev, const void* final_norm_dev,
                         const void* lm_head_dev, const atspeed_llama_layer_weights* layers /* host array */,
                         atspeed_llama** out);
void atspeed_llama_destroy(atspeed_llama* m);

/* One forward (beamSD.py:52 / :221): T tokens, written to KV slots `slots`, each row seeing
 * the slots whose bit is set in vis_bits[t][0..max_slots/64).  The 4-D additive mask of the
 * reference (beamSD.py:89,204-209) is never materialised.  Logits (fp32, row stride
 * atspeed_llama_logits_ld()) of the LAST n_logit_rows tokens go to the handle's logits
 * buffer (atspeed_llama_logits()) or to logits_out_dev when not NULL. */
int atspeed_llama_forward(atspeed_llama* m, const int32_t* ids_dev, const int32_t* pos_dev,
                          const int32_t* slots_dev, const uint64_t* vis_bits_dev, int32_t n_tokens,
                          int32_t n_slots_visible, int32_t n_logit_rows, float* logits_out_dev, void* stream);
/* n independent sequences as ONE forward (generate_teacher_data.py:225-232 scores the label and the K beams of every sample;
 * the reference runs one sample at a time).  Host arrays of n device pointers / counts; sequence i gets KV arena i of a pool the
 * handle owns (grown on demand), so slot numbers are private to a sequence.  The logits of the last n_logit_rows[i] rows of
 * sequence 0, 1, ... follow each other in logits_out_dev (row stride atspeed_llama_logits_ld()).  n <= 256. */
int atspeed_llama_forward_batch(atspeed_llama* m, int32_t n, const int32_t* const* ids_dev, const int32_t* const* pos_dev,
                                const int32_t* const* slots_dev, const uint64_t* const* vis_bits_dev, const int32_t* n_tokens,
                                const int32_t* n_slots_visible, const int32_t* n_logit_rows, float* logits_out_dev, void* stream);
float* atspeed_llama_logits(atspeed_llama* m);
/* hipEvent brackets around the forward's five GEMM kinds (0 qkv, 1 o_proj, 2 gate_up+SwiGLU, 3 down,
 * 4 lm_head), recorded on the launch stream.  Returns the sums since the last reset in ms_out[5] /
 * count_out[5] / rows_out[5] (sum of M); enable = 1/0 switches the brackets on/off and resets,
 * enable < 0 only reads.  Measurement hook for bench.py's roofline (no reference counterpart). */
int atspeed_llama_profile(atspeed_llama* m, int32_t enable, double* ms_out, int64_t* count_out, int64_t* rows_out);
/* The same accumulators restricted to launches of >= 1024 tokens (the 256x256 ring kernel's launches): read BEFORE the
 * atspeed_llama_profile call that resets them.  bench.py's roofline uses these so that its average launch time is
 * that of one kernel (gemm_ring_kernel<EPI, 8, false>) and can be checked against the rocprofv3 kernel summary. */
int atspeed_llama_profile_big(atspeed_llama* m, double* ms_out, int64_t* count_out, int64_t* rows_out);
/* Token counts of the forwards this model ran since logging was switched on: enable = 1 starts (and clears) the log, 0 stops it, < 0
 * only reads.  Writes up to `max_pairs` (tokens, logit rows) pairs, oldest first, to pairs_out (may be NULL) and returns how many forwards
 * the log holds (it keeps the first 4096).  bench.py prices every forward of a decode against max(weight stream, MFMA time) with these:
 * the packed verification of beamSD.py:203-221 makes the token count of a forward data dependent.  No reference counterpart. */
int32_t atspeed_llama_forward_log(atspeed_llama* m, int32_t enable, int32_t* pairs_out, int32_t max_pairs);
/* Measured peaks for the roofline report (nominal peaks are re-measured on the box).  No reference counterpart.
 * atspeed_probe_mfma_bf16: register-only loop of v_mfma_f32_16x16x32_bf16 on random operands, `iters` trips of 32 MFMAs per wave,
 * 8 waves x 1024 workgroups; scratch_dev >= 2 MiB.  atspeed_probe_hbm_read: `reps` read-only passes over buf_dev (use a buffer far
 * larger than the 256 MB Infinity Cache); scratch_dev >= 4 bytes.  Both time themselves with HIP events on `stream` and synchronise. */
int atspeed_probe_mfma_bf16(int32_t iters, void* scratch_dev, size_t scratch_bytes, void* stream, double* tflops_out);
int atspeed_probe_hbm_read(const void* buf_dev, size_t bytes, int32_t reps, void* scratch_dev, void* stream, double* gbs_out);
int32_t atspeed_llama_logits_ld(const atspeed_llama* m);
/* BASELINE config 5 (fp8 target verification): build OCP-e4m3 copies of the layer projections (per-output-row scales,
 * library-owned) from the bf16 weights.  From then on the batched forwards (M >= 512 tokens, shapes that fill the chip)
 * quantise activations per token and run W8A8 MFMA GEMMs (v_mfma_f32_16x16x32_fp8_fp8, fp32 accumulate); smaller
 * forwards, the lm_head, norms, attention and the KV cache stay bf16.  No reference counterpart (its target is int8
 * weights via bitsandbytes, inference.py:88). */
int atspeed_llama_enable_fp8(atspeed_llama* m, void* stream);
/* how many launches of each layer projection (0 qkv, 1 o_proj, 2 gate_up, 3 down; one per layer per forward) ran as an fp8 GEMM
 * (fp8_out[4]) and how many as a bf16 / fp32 GEMM (other_out[4]) since the last reset: lets a test or a bench state that config 5
 * really ran its projections in fp8 rather than fell back by shape.  Either output may be NULL.  No reference counterpart. */
int atspeed_llama_fp8_counters(atspeed_llama* m, int64_t* fp8_out, int64_t* other_out, int32_t reset);
/* how many qkv projections (one per layer per forward) ran with the rotary embedding and the KV-cache scatter in the GEMM's epilogue
 * (batched bf16 forwards, head_dim 128, hidden % 256 == 0: the reference's apply_rotary_pos_emb + cache update, modeling_llama.py as
 * called from beamSD.py:120, without re-reading the projection) rather than as the separate pass, since the last reset; -1 on a NULL
 * model.  The switch "fuse_qkv_rope" (atspeed_set_switch; initial value ATSPEED_FUSE_QKV_ROPE, default 1) = 0 selects the separate pass:
 * results are bit-identical.  Since round 6 ONE user's W8A8 qkv projection (1-256 tokens, 150-256 unsplit tiles of 64 weight rows: Llama-7B's
 * 192) carries the same epilogue in the weight-streaming kernel. */
int64_t atspeed_llama_rope_fused_launches(atspeed_llama* m, int32_t reset);
/* bytes of the split-K arena this model owns (0: none yet -- it is allocated in front of the model's first forward of >= 257 tokens, so a
 * draft or a one-user target never holds one; thin ring-kernel grids of a model without one take the device's shared arena or the plain grid) */
int64_t atspeed_llama_sk_arena_bytes(const atspeed_llama* m);

/* lm_head fused with the full-vocabulary normaliser of beamSD.py:58,285 (log_softmax over ALL columns, before masking): bf16
 * x [rows, hidden] times w [vocab, hidden]^T -> fp32 logits (row stride ld) and lse[row] = log sum_v exp(logits[row][v]).  On the
 * batched path (rows >= 257 and a tile grid that fills the chip) the (max, sum exp) partials come out of the GEMM epilogue, the
 * logits are never re-read, and with `fsm` given only the 256-column tiles that hold a token of the automaton are written (the
 * others are never read by a step: Beauty 5 of 129 tiles) -- *fused_out = 1; otherwise the plain GEMM + atspeed_lse_rows (0).
 * workspace: at least rows * ceil(vocab / 256) * 8 bytes for the fused path (+ the split-K slabs of the small path).  This is what
 * the decoder's forwards run; standalone for tests and benches. */
int atspeed_lmhead_lse(const void* x_dev, const void* w_dev, float* logits_dev, float* lse_dev, int32_t rows, int32_t vocab,
                       int32_t hidden, int32_t ld, const atspeed_fsm* fsm /* may be NULL */, void* workspace_dev, size_t workspace_bytes,
                       int32_t* fused_out /* may be NULL */, void* stream);

/* ------------------------------------------------------------------ scan kernels
 * log-softmax normaliser over the FULL vocabulary, before masking (beamSD.py:58,285):
 * lse[r] = log(sum_v exp(logits[r][v])).  Rows are `ld` floats apart. */
int atspeed_lse_rows(const float* logits_dev, int32_t n_rows, int32_t vocab, int32_t ld,
                     float* lse_out_dev, void* stream);

/* out[r][c] = logits[r][c] - lse[r] for c < vocab: the log-softmax rows (beamSD.py:58) as a tensor, for callers that must hand them to
 * host-side logits processors (beamSD.py:62-64 with a non-empty LogitsProcessorList); the decoder itself never materialises them. */
int atspeed_log_softmax_rows(const float* logits_dev, int32_t ld, const float* lse_dev, int32_t n_rows, int32_t vocab, float* out_dev,
                             int32_t ld_out, void* stream);

/* Fused constraint mask + beam expand + prune (beamSD.py:60-87):
 *   cand(r, t) = logits[r][t] - lse[r] + beam_score[r]   for t allowed at node[r]
 *   top-k by (score desc, flat id r*vocab+t asc)  ->  out_*[0..k)
 * -inf / missing candidates are flagged out_flat = -1 (never a beam).
 * Standalone form used by tests; the decoder uses the same device code inside its step
 * kernels together with the mask/position bookkeeping (beamSD.py:88-91). */
int atspeed_beam_expand_prune(const float* logits_dev, int32_t ld, const float* lse_dev,
                              const float* beam_score_dev, const int32_t* beam_node_dev, int32_t n_rows,
                              const atspeed_fsm* fsm, int32_t k,
                              float* out_score_dev, int32_t* out_parent_dev, int32_t* out_token_dev,
                              int32_t* out_node_dev, int32_t* out_flat_dev, void* stream);

/* The same without a mask (beamSD.py:58,69-78 with an empty processor list; also the expand of rows that a host-side logits
 * processor has already rewritten: pass lse = 0): candidates are ALL `vocab` columns of each row; -inf entries are never picked.
 * atspeed_row_topk: out_tokens[r][ATSPEED_MAX_BEAMS] = the k best columns of row r (value desc, column asc; -1 = fewer finite ones) --
 * the k best (row, token) pairs lie among them.  row_cand_ws_dev: n_rows * ATSPEED_MAX_BEAMS int32 of scratch. */
int atspeed_row_topk(const float* scores_dev, int32_t n_rows, int32_t vocab, int32_t ld, int32_t k, int32_t* out_tokens_dev, void* stream);
int atspeed_beam_expand_prune_free(const float* logits_dev, int32_t ld, const float* lse_dev, const float* beam_score_dev, int32_t n_rows,
                                   int32_t vocab, int32_t k, int32_t* row_cand_ws_dev, float* out_score_dev, int32_t* out_parent_dev,
                                   int32_t* out_token_dev, int32_t* out_flat_dev, void* stream);

/* Top-K-aligned acceptance test (beamSD.py:371-380): accept iff every target id is among
 * the draft ids.  hit[r] = r-th smallest draft position that was hit, score_by_hit[r] = the
 * target score of that entry (beamSD.py:295-296,373-376).  out_accept = 1/0. */
int atspeed_accept(const int32_t* target_flat_dev, const float* target_score_dev, int32_t k,
                   const int32_t* draft_flat_dev, int32_t dk,
                   int32_t* hit_out_dev, float* score_by_hit_out_dev, int32_t* accept_out_dev, void* stream);

/* ------------------------------------------------------------------ decoder
 * Replaces BSSD() (beamSD.py:458-542) and target_generate() (beamSD.py:544-595) for one
 * user stream: draft steps, the packed target verification forward, verify(), and all
 * bookkeeping (masks as bitsets, positions, KV slots, beam suffixes) stay on the device;
 * the host reads back one small mailbox per round. */
typedef struct atspeed_decoder atspeed_decoder;
int atspeed_decoder_create(atspeed_llama* target, atspeed_llama* draft /* may be NULL */,
                           int32_t max_prompt, atspeed_decoder** out);
void atspeed_decoder_destroy(atspeed_decoder* d);

/* Sampling mode of a decoder (generation_config.do_sample / temperature; beamSD.py:65-75 draft and final steps,
 * :293-321,332-369 verification, :529-531 final sort).  Off by default: every BASELINE config is greedy.  Draws are
 * counter-based functions of (seed, round, step, candidate), so a call is reproducible and the CPU restatement
 * (oracle/beamsd_sample_ref.py, HashRng) makes the same decisions; the law is that of the reference's torch.multinomial /
 * rand / randperm draws (statistical parity).  Applies to the bssd and target_generate calls made with this decoder. */
int atspeed_decoder_set_sampling(atspeed_decoder* d, int32_t do_sample, float temperature, uint32_t seed);

typedef struct atspeed_gen_stats {
  int32_t n_run;               /* verification rounds (beamSD.py:527)                 */
  int32_t total_accept_steps;  /* sum of n_matches (beamSD.py:528)                    */
  int32_t accept_steps[ATSPEED_MAX_NEW_TOKENS];
  int32_t n_valid;             /* beams with a finite score                           */
  int32_t n_target_forwards, n_draft_forwards;
  float draft_ms, target_ms, verify_ms, total_ms;   /* hipEvent stage times (Timer, beamSD.py:12-37) */
  int32_t status;              /* batched calls: ATSPEED_OK, or ATSPEED_ERR_FILTERED for a user whose step lost every beam to the id
                                  filter of beamSD.py:80-86 -- that user ends with n_valid = 0, the rest of the batch finishes (the
                                  one-user calls return the error instead)                */
} atspeed_gen_stats;

/* prompt_ids_dev: [prompt_len] int32.  out_tokens_dev: [k][max_new_tokens] int32 generated
 * suffixes ordered by score desc; out_scores_dev: [k] fp32.  Synchronises `stream` once per
 * round (to read n_matches) and at exit. */
int atspeed_bssd_generate(atspeed_decoder* d, const int32_t* prompt_ids_dev, int32_t prompt_len,
                          const atspeed_fsm* fsm, int32_t start_node, int32_t gamma, int32_t max_new_tokens,
                          int32_t k, int32_t dk, int32_t* out_tokens_dev, float* out_scores_dev,
                          atspeed_gen_stats* stats_host, void* stream);

/* n independent users, one decoder each, interleaved on the decoders' private streams (the reference runs users
 * strictly one after another, inference.py:162-176): hides the per-round mailbox sync and the launch latency of
 * the small forwards, and lets forwards of different users overlap.  Results equal n sequential calls. */
int atspeed_bssd_generate_batch(atspeed_decoder** decoders, int32_t n, const int32_t* const* prompt_ids_dev,
                                const int32_t* prompt_lens, const atspeed_fsm* fsm, const int32_t* start_nodes,
                                int32_t gamma, int32_t max_new_tokens, int32_t k, int32_t dk,
                                int32_t* const* out_tokens_dev, float* const* out_scores_dev,
                                atspeed_gen_stats* stats_host /* [n] */, void* stream);

int atspeed_target_generate(atspeed_decoder* d, const int32_t* prompt_ids_dev, int32_t prompt_len,
                            const atspeed_fsm* fsm, int32_t start_node, int32_t max_new_tokens, int32_t k,
                            int32_t* out_tokens_dev, float* out_scores_dev, atspeed_gen_stats* stats_host,
                            void* stream);

/* target_generate for n users in lock step (one forward + one beam-step launch per generated position for all of
 * them): the constrained beam search of generate_teacher_data.py:211-244 at scale.  Results equal n single calls. */
int atspeed_target_generate_batch(atspeed_decoder** decoders, int32_t n, const int32_t* const* prompt_ids_dev,
                                  const int32_t* prompt_lens, const atspeed_fsm* fsm, const int32_t* start_nodes,
                                  int32_t max_new_tokens, int32_t k, int32_t* const* out_tokens_dev,
                                  float* const* out_scores_dev, atspeed_gen_stats* stats_host /* [n] */, void* stream);

/* Result tensors of a whole batch in one launch (the reference builds `beam_sequence` per beam per step with torch.cat,
 * beamSD.py:87,383): user u's [k][P_u + new_tokens] int64 rows = its prompt followed by the generated suffix of beam j, written at
 * out_dev + k * (prompt_off[u] + u * new_tokens).  prompts_flat_dev: the prompts one after the other (int32); prompt_off_host[n + 1]:
 * their offsets (HOST array); toks_dev [n][k][new_tokens] int32 (the out_tokens of atspeed_bssd_generate_batch laid out contiguously). */
int atspeed_assemble_sequences(const int32_t* prompts_flat_dev, const int64_t* prompt_off_host, const int32_t* toks_dev, int32_t n, int32_t k,
                               int32_t new_tokens, int64_t* out_dev, void* stream);

/* per-round trace of the last atspeed_bssd_generate call (host memory, for parity tests):
 * for round r, step i: the draft's flat ids (dk entries, -1 = not a beam).  Returns the
 * number of ints written. */
int atspeed_decoder_trace(atspeed_decoder* d, int32_t* rounds_out, int32_t cap);

/* Decision trace (parity tests at batch sizes the per-round trace above skips; greedy mode; no reference counterpart -- the reference
 * keeps these as Python locals of verify(), beamSD.py:277-380).  level 1: after every round of the bssd calls made with this decoder
 * the beam blocks and the verify walk's picks are copied to the host (about 60 KB per user and round); level 0 (default): off.
 * atspeed_decoder_decisions returns the number of int32 words of the last call's trace and copies up to cap_words of them.  One
 * record per round: header {kind (0 verify round, 1 final single step), nb, dl, n_matches, tokens generated before the round, k, dk,
 * n_blocks}; n_blocks beam-set images of 5 * 64 + 64 * 16 words each {score bits[64], node[64], parent[64], tok[64], flat[64],
 * seq[64][16]} -- kind 0: the round's beams, draft blocks 1..dl, the new round beams; kind 1: the step's parents, its result --; for
 * kind 0 then (ATSPEED_MAX_GAMMA + 1) * 3 * 64 words: the target's picks of verify step i as {score bits[64], parent[64] (index into
 * block i), token[64] (-1 = no pick)}, valid for steps 0..n_matches. */
int atspeed_decoder_set_trace(atspeed_decoder* d, int32_t level);
int64_t atspeed_decoder_decisions(atspeed_decoder* d, int32_t* out, int64_t cap_words);

/* ------------------------------------------------------------------ low-level ops (tests, benches)
 * C[M,N] = A[M,K] * W[N,K]^T on MFMA; epilogue: 0 store (dtype), 1 fp32 store, 2 residual add into
 * C (dtype), 3 SwiGLU over interleaved gate/up column groups (C is [M, N/2]). */
int atspeed_gemm(const void* a_dev, const void* w_dev, void* c_dev, int32_t m, int32_t n, int32_t k,
                 int32_t lda, int32_t ldc, int32_t dtype, int32_t epilogue, void* workspace_dev,
                 size_t workspace_bytes, void* stream);
/* Which kernel family the library's GEMM launches took since the last reset (dispatch is a fitted cost model: tests assert the path they
 * mean to exercise).  out[i], i < n: 0 ring kernel, 1 ring kernel with its split-K tail, 2 weight-streaming kernel, 3 the same split in K,
 * 4 ring kernel in split-K mode, 5 LDS-tiled kernel, 6 fp8 ring kernel, 7 / 8 fp8 weight-streaming kernel / split, 9 / 10 panel kernel /
 * split, 11 fp8 ring kernel cut in K.  Returns the number of counters the library keeps. */
int atspeed_gemm_path_counters(int64_t* out, int32_t n, int32_t reset);
/* Process-wide tuning / test switches.  Each is an int read ONCE from its environment variable when the library first needs one (the
 * variable is the way to set it for a whole run) and changeable afterwards only through this call (tests and sweeps that compare two
 * settings in one process) -- the dispatch path never calls getenv.  Names: "gemm_sk" (ATSPEED_GEMM_SK, 1), "gemm_sk_g" (no variable; test
 * hook, 0), "gemm_panel" (ATSPEED_GEMM_PANEL, 1), "gemm_force_mt" (ATSPEED_GEMM_FORCE_MT, 0), "graphs" (ATSPEED_GRAPHS, 0),
 * "fuse_qkv_rope" (ATSPEED_FUSE_QKV_ROPE, 1), "fuse_qkv_reduce" (ATSPEED_FUSE_QKV_REDUCE, 1), "gemm_kcut" (ATSPEED_GEMM_KCUT, 2); meanings in INTEGRATION.md.  Unknown name: ATSPEED_ERR_INVALID. */
int atspeed_set_switch(const char* name, int32_t value);
int atspeed_get_switch(const char* name, int32_t* value_out);
/* atspeed_gemm / atspeed_gemm_fp8 on operands in the packed layout (a / xq and w / wq through atspeed_pack_rows; K % 32 == 0, for fp8 K % 64 == 0):
 * what the bf16 / fp8 engine runs.  The SwiGLU epilogue's output (ldc % 32 == 0) is packed as well -- it is the down projection's operand --,
 * every other output is row-major.  Same arithmetic as the row-major calls: results are bit-identical. */
int atspeed_gemm_packed(const void* a_dev, const void* w_dev, void* c_dev, int32_t m, int32_t n, int32_t k, int32_t ldc, int32_t epilogue,
                        void* workspace_dev, size_t workspace_bytes, void* stream);
int atspeed_gemm_fp8_packed(const void* xq_dev, const float* sx_dev, const void* wq_dev, const float* sw_dev, void* c_dev, int32_t m,
                            int32_t n, int32_t k, int32_t ldc, int32_t epilogue, void* workspace_dev, size_t workspace_bytes, void* stream);
/* per-row e4m3 quantisation q = e4m3(x / scale[r]), scale[r] = max|x[r]| / 448, and the W8A8 GEMM over such operands */
int atspeed_quant_rows_fp8(const void* x_bf16_dev, int32_t rows, int32_t cols, void* q_dev, float* scale_dev, void* stream);
/* the same on operands in the packed layout (x through atspeed_pack_rows with row_bytes = 2 cols, q comes out as atspeed_pack_rows with
 * row_bytes = cols would lay it out; cols % 64 == 0; both buffers hold an even number of rows): what the engine runs between a bf16
 * producer (attention, SwiGLU) and the fp8 projection that consumes it -- one workgroup per row PAIR, whole 128-byte lines in and out */
int atspeed_quant_rows_fp8_packed(const void* x_bf16_packed_dev, int32_t rows, int32_t cols, void* q_packed_dev, float* scale_dev, void* stream);
/* m >= 257: the block-scaled MFMA ring kernel (K % 256 == 0, lock-step batches).  m <= 256 (round 5: one user's forwards, the reference's
 * own regime -- code/inference.py:86-91 loads its target 8-bit for every batch-1 forward): the weight-streaming kernel on e4m3 rows
 * (K % 128 == 0, K >= 512).  There a narrow N is cut in K and finished by a reduce pass over fp32 partial sums in `workspace_dev`
 * (parts x m x n x 4 bytes with parts = min(256 / tiles, K / 512), tiles = N / 64 up to N = 16384, else N / 128 -- up to 256 parts for a very
 * small N; 64 MB always suffices); with too little workspace the launch runs one part per tile; the residual epilogue (2) then takes the ring
 * kernel when K % 256 == 0 and returns ATSPEED_ERR_CAPACITY otherwise.
 * SIGNATURE NOTE: (workspace_dev, workspace_bytes) were inserted before `stream` in round 5; atspeed_version() reports 0.2 since round 6 so
 * that a caller built against the 0.1 header (stream in the workspace position) can tell. */
int atspeed_gemm_fp8(const void* xq_dev, const float* sx_dev, const void* wq_dev, const float* sw_dev, void* c_dev, int32_t m,
                     int32_t n, int32_t k, int32_t ldc, int32_t epilogue, void* workspace_dev, size_t workspace_bytes, void* stream);
int atspeed_rmsnorm(const void* x_dev, const void* w_dev, void* y_dev, int32_t rows, int32_t hidden,
                    float eps, int32_t dtype, void* stream);
/* bf16 RMSNorm fused with the per-token e4m3 quantisation of its output (what the fp8 forward runs in front of the qkv and
 * gate_up projections); y_dev may be NULL; q / scale equal atspeed_quant_rows_fp8 of the bf16 norm output bit for bit */
int atspeed_rmsnorm_quant_fp8(const void* x_dev, const void* w_dev, void* y_dev, void* q_dev, float* scale_dev, int32_t rows,
                              int32_t hidden, float eps, void* stream);
/* tree attention over a slot-addressed KV cache ([max_slots][hidden] per K and V) */
int atspeed_tree_attention(const void* q_dev, int32_t ldq, const void* kcache_dev, const void* vcache_dev,
                           const uint64_t* vis_bits_dev, int32_t vis_words, void* out_dev, int32_t n_tokens,
                           int32_t n_slots, int32_t n_heads, int32_t head_dim, int32_t dtype, void* stream);
/* the same with the tiling chosen by the caller (parity tests reach every kernel form): qtile_rows 0 (auto) / 64 / 128 / 256 query rows
 * per workgroup; rows_per_wave 0 (auto) / 16 (v_mfma 16x16x32, register-staged tiles: small grids) / 32 (v_mfma 32x32x16, LDS-DMA
 * double buffer: the lock-step batches).  bf16 with head_dim 64 / 128 only; other inputs take the scalar kernel whatever is asked. */
int atspeed_tree_attention_tiled(const void* q_dev, int32_t ldq, const void* kcache_dev, const void* vcache_dev,
                                 const uint64_t* vis_bits_dev, int32_t vis_words, void* out_dev, int32_t n_tokens,
                                 int32_t n_slots, int32_t n_heads, int32_t head_dim, int32_t dtype, int32_t qtile_rows,
                                 int32_t rows_per_wave, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ATSPEED_HIP_H */
