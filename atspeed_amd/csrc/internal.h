// Internal host-side interfaces between the translation units of libatspeed_hip.
#pragma once
#include <atomic>

#include "common.h"

// ---- segment table: one forward over the tokens of several users ----------------------
// Each user (decoder) contributes a contiguous run of rows; its KV cache, visibility bitsets, positions
// and slots stay private.  The host builds the table, stages it to device memory (ats_stage) and kernels read it
// through a pointer (256 users x 72 B do not fit the 4 KB kernel-argument block).
constexpr int ATS_MAX_SEGS = 256;           // users per lock-step batch (qtile_seg is a byte: <= 256)
constexpr int ATS_MAX_QTILES = ATS_MAX_SEGS * 8;
struct Seg {
  const int32_t* ids; const int32_t* pos; const int32_t* slot; const uint64_t* vis;   // this user's per-token arrays
  void* kc; void* vc;          // this user's KV cache (layer 0 base)
  int row0, n_tok;             // rows [row0, row0 + n_tok) of the batched activations
  int n_slots;                 // slots visible to this user's rows
  int logit_row0, n_logit;     // the last n_logit tokens get logits, at rows [logit_row0, ..) of the logits buffer
};
struct SegTable {
  int n, total_tok, total_logit;
  int n_qtiles;                                   // query tiles of all segments (attention grid)
  int qtile_rows;                                 // rows per query tile: 64 (4 waves) or 128 (8 waves, halves the K/V re-reads)
  Seg seg[ATS_MAX_SEGS];
  unsigned char qtile_seg[ATS_MAX_QTILES], qtile_idx[ATS_MAX_QTILES];
};
// per batched row, resolved once per forward (ats_row_info): where the row's K / V go and which rotation it takes
struct RowInfo { void* kc; void* vc; int pos; int slot; };        // caches' layer-0 bases of the row's user; pos clamped to the RoPE table
struct RopeEpi { const RowInfo* rows = nullptr; const float* cos_tab = nullptr; const float* sin_tab = nullptr; size_t layer_off = 0; int hidden = 0; };
// copy a host object to device memory, stream ordered (pinned staging ring); returns the device address
int ats_stage(const void* host_obj, size_t bytes, const void** dev_out, hipStream_t st);
int ats_stage_to(const void* host_obj, size_t bytes, void* dev_dst, hipStream_t st);   // same, to a fixed device address
void ats_stage_reset();                           // after a stream synchronisation: every staged copy has landed

// Arena of the ring kernel's split-K tail (gemm.hip, gemm_ring_kernel<..., SK = true>): 512 slots of 256 KB of fp32 partials + one arrival
// counter per tail tile (zero between launches: the finishing part leaves its counter at zero).  Launches that share an arena must be
// ordered: a model's forwards own one (ActCtx, allocated with the activations, so a captured graph contains no allocation and no
// cross-stream wait); callers without one (the C ABI's plain atspeed_gemm) share a per-device arena under a mutex.
struct SkArena { float* ws = nullptr; int* cnt = nullptr; };
constexpr size_t ATS_SK_ARENA_BYTES = (size_t)512 * 256 * 1024;
constexpr int ATS_SK_ARENA_COUNTERS = 256;

// which kernel family a GEMM launch took (atspeed_gemm_path_counters: tests assert the path they mean to test, dispatch being a fitted model)
enum { ATS_PATH_RING = 0, ATS_PATH_RING_SK = 1, ATS_PATH_WDMA = 2, ATS_PATH_WDMA_SPLIT = 3, ATS_PATH_RING_SPLIT = 4, ATS_PATH_TILED = 5,
       ATS_PATH_FP8_RING = 6, ATS_PATH_FP8_WDMA = 7, ATS_PATH_FP8_WDMA_SPLIT = 8, ATS_PATH_PANEL = 9, ATS_PATH_PANEL_SPLIT = 10, ATS_PATH_FP8_RING_SPLIT = 11,
       ATS_N_PATHS = 16 };
extern std::atomic<long long> g_ats_path_cnt[ATS_N_PATHS];          // engine.hip
inline void ats_count_path(int p) { g_ats_path_cnt[p].fetch_add(1, std::memory_order_relaxed); }

// Process-wide tuning / test switches (engine.hip): each is an int initialised ONCE from its environment variable and changeable afterwards
// only through the C ABI (atspeed_set_switch) -- no getenv on the dispatch path, no race with a host thread that edits the environment.
enum { ATS_SW_GEMM_SK = 0,        // "gemm_sk"       ATSPEED_GEMM_SK (1): ring kernel's split-K tail -- 0 off, 1 cost model, 2-4 that many parts wherever they fit
       ATS_SW_GEMM_SK_G,          // "gemm_sk_g"     (no variable: test hook) an UNALIGNED deal of the tail's k-units over G workgroups
       ATS_SW_GEMM_PANEL,         // "gemm_panel"    ATSPEED_GEMM_PANEL (1): panel form for 257-384 tokens -- 0 off, 1 where it won, 2 every shape it can take
       ATS_SW_GEMM_FORCE_MT,      // "gemm_force_mt" ATSPEED_GEMM_FORCE_MT (0): 8 / 4 = always 256- / 128-row token tiles
       ATS_SW_GRAPHS,             // "graphs"        ATSPEED_GRAPHS (0): replay recurring forwards as hipGraphs
       ATS_SW_FUSE_QKV_ROPE,      // "fuse_qkv_rope" ATSPEED_FUSE_QKV_ROPE (1): RoPE + KV scatter in the qkv projection's epilogue
       ATS_SW_GEMM_KCUT,          // "gemm_kcut"     ATSPEED_GEMM_KCUT (1): bf16 N <= 4096 projections at 257-1100 tokens cut in K over the whole chip
       ATS_SW_FUSE_QKV_REDUCE,    // "fuse_qkv_reduce" ATSPEED_FUSE_QKV_REDUCE (1): one user's qkv split-K slabs are summed by the RoPE kernel instead of a reduce launch
       ATS_N_SW };
int ats_switch(int id);

enum { EPI_STORE = 0, EPI_F32 = 1, EPI_RESID = 2, EPI_SWIGLU = 3, EPI_F32_LSE = 4 /* ring kernel only: fp32 store + per-tile (max, sum exp) */,
       EPI_QKV_ROPE = 5 /* ring kernel only: qkv projection with RoPE + KV-cache scatter in the epilogue (ats_gemm_qkv_rope) */ };
// the kernel translation units' functions, once per flavour (declarations: kernels_decl.inc)
namespace ats_bf16 {
#include "kernels_decl.inc"
}
namespace ats_f16 {
#include "kernels_decl.inc"
}

// ---- scan.hip -------------------------------------------------------------------------
struct FsmDev {
  const int32_t* row_ptr;
  const int32_t* tok;
  const int32_t* nxt;
  int32_t n_nodes, n_edges, vocab;     // n_nodes == 0: no mask at all (atspeed_fsm_create_free): candidates come from row_cand lists
  // one_step_beam_search's post-top-k id filter (beamSD.py:80-86 hard-codes `tok >= 32000 | tok == 2`): picks with a token below
  // filter_min that is not filter_eos are dropped; filter_min <= 0 keeps everything (atspeed_fsm_set_id_filter)
  int32_t filter_min, filter_eos;
};
struct atspeed_fsm {
  FsmDev dev;
  int32_t *d_row_ptr, *d_tok, *d_nxt;
  unsigned char* d_tile_store;   // [ceil(vocab / 256)] device bytes: 1 = some edge token lies in columns [256 t, 256 t + 256) (the logit tiles a step can read)
};

struct TokBuf {       // inputs of a forward: one row per token
  int32_t* ids;
  int32_t* pos;
  int32_t* slot;
  uint64_t* vis;      // [rows][W]
};
struct BeamSet {      // a block of beams (round beams or one draft step's output)
  float* score;       // [MAXB]
  int32_t* node;      // [MAXB] FSM node after the beam's last token
  int32_t* seq;       // [MAXB][ATSPEED_MAX_NEW_TOKENS] generated suffix
  int32_t* parent;    // [MAXB] index into the previous block
  int32_t* tok;       // [MAXB]
  int32_t* flat;      // [MAXB] parent*V + tok, -1 = not a beam
};
struct Mailbox {      // device -> host, one per round
  int32_t n_matches;
  int32_t status;     // 0 ok, ATSPEED_ERR_*
  int32_t n_valid;
  int32_t pad;
};

int ats_lse_rows(const float* logits, int n_rows, int vocab, int ld, float* lse, hipStream_t st);
// out[row][ATSPEED_MAX_BEAMS]: the kk best columns of each row (value desc, column asc; -inf never; -1 = none)
int ats_row_topk(const float* logits, int n_rows, int vocab, int ld, int kk, int32_t* out, hipStream_t st);

struct BeamStepArgs {
  BeamSet src;  int n_src;  int gen_len;     // beams being expanded; tokens generated so far
  const float* logits;  int ld;  const float* lse;
  FsmDev fsm;
  int k;
  BeamSet dst;
  int emit;                                   // write next-step inputs?
  int filter_ids;                             // one_step_beam_search's post-top-k id filter (beamSD.py:80-86); off for verify-style expands
  TokBuf in;   int in_row0;                   // rows of the src beams (for parent vis/pos)
  TokBuf out;  int out_row0;  int out_slot0;  int vis_words;
  Mailbox* mail;                              // status only
  const int32_t* row_cand;  int n_row_cand;   // mask-free search (fsm.n_nodes == 0): [logit row][ATSPEED_MAX_BEAMS] best tokens per row (ats_row_topk)
  // sampling (generation_config.do_sample, beamSD.py:65-75): draw k of the candidates without replacement with probability
  // softmax(score / temperature); tab_* (optional) keep the draft's whole candidate distribution for the verification
  int sample;  float temperature;  uint32_t rng_sub;
  float* tab_score;  int32_t* tab_off;  float* tab_lse;   // [candidates in expand order], [n_src + 1], [1]
};
int ats_beam_step(const BeamStepArgs& a, hipStream_t st);
// one workgroup per user; `dev_args` is a DEVICE array of n argument blocks
int ats_beam_step_multi(const BeamStepArgs* dev_args, int n, hipStream_t st);

struct VerifyArgs {
  BeamSet blk[ATSPEED_MAX_GAMMA + 1];         // blk[0] round beams, blk[i] draft step i
  int nb, dl, k, dk, gen_len0;
  const float* logits;  int ld;  const float* lse;   // rows: block 0 (nb), then dl blocks of dk
  FsmDev fsm;
  TokBuf cur;  int n0;                        // packed target inputs of this round
  TokBuf next;                                // next round's target inputs (k rows)
  TokBuf dnext;                               // draft re-ingest inputs (dk + k rows) when everything was accepted
  int vis_words;
  BeamSet res;                                // new round beams (k)
  Mailbox* mail;
  const int32_t* row_cand;  int n_row_cand;   // mask-free search: as in BeamStepArgs, rows aligned with `logits`
  // sampling verification (beamSD.py:293-321,332-369)
  int sample;  float temperature;  uint32_t seed;  int round;
  const float* dtab_score[ATSPEED_MAX_GAMMA];  const int32_t* dtab_off[ATSPEED_MAX_GAMMA];  const float* dtab_lse[ATSPEED_MAX_GAMMA];
  // decision trace (atspeed_decoder_set_trace level 1, greedy walk only): step i's k picks as [i][3][ATSPEED_MAX_BEAMS] words
  // (score bits, parent = index into blk[i], token; flat < 0 picks have token -1); NULL = off
  int32_t* vtrace;
};
int ats_verify_walk(const VerifyArgs& a, hipStream_t st);
int ats_verify_walk_multi(const VerifyArgs* dev_args, int n, hipStream_t st);

int ats_init_prompt(TokBuf tb, const int32_t* prompt, int prompt_len, int vis_words, BeamSet beams, int start_node,
                    int vocab, Mailbox* mail, hipStream_t st);
// out_tokens[k][max_new] / out_scores[k] from a beam set
int ats_export_beams(BeamSet b, int k, int max_new, int32_t* out_tokens, float* out_scores, hipStream_t st, bool sort_desc = false);
// the same for every user of a lock-step batch in one launch each; `dev_args` = DEVICE arrays of n argument blocks (ats_stage)
struct InitPromptArgs { TokBuf tb; const int32_t* prompt; int prompt_len; int start_node; BeamSet beams; Mailbox* mail; };
struct ExportBeamsArgs { BeamSet b; int32_t* out_tokens; float* out_scores; int sort_desc; };
int ats_init_prompt_multi(const InitPromptArgs* dev_args, int n, int max_prompt_len, int vis_words, int vocab, hipStream_t st);
int ats_export_beams_multi(const ExportBeamsArgs* dev_args, int n, int k, int max_new, hipStream_t st);
constexpr int ATS_MAX_CAND = 16384;           // candidate capacity of one expand (scan.hip kMaxCand)

int ats_accept(const int32_t* target_flat, const float* target_score, int k, const int32_t* draft_flat, int dk,
               int32_t* hit, float* score_by_hit, int32_t* accept, hipStream_t st);
