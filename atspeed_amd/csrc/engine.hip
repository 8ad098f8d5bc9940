// Host-side engine: model handle (weights in HBM, KV arena, forward sequencing) and the
// decoder that runs BSSD / target_generate for one user stream with all state on the device.
// Reference: beamSD.py:458-542 (BSSD), :544-595 (target_generate), :108-179, :190-232, :242-456.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <vector>

#include "internal.h"

// ---------------------------------------------------------------------------- errors / misc
static thread_local std::string g_last_error;

void atspeed_set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

extern "C" const char* atspeed_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* atspeed_version(void) { return "atspeed_hip 0.1 (gfx950)"; }
extern "C" int atspeed_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ---------------------------------------------------------------------------- FSM
extern "C" int atspeed_fsm_create(const int32_t* row_ptr, const int32_t* tok, const int32_t* nxt, int32_t n_nodes,
                                  int32_t n_edges, int32_t vocab_size, atspeed_fsm** out) {
  ATS_REQUIRE(row_ptr && out && n_nodes >= 1 && n_edges >= 0 && vocab_size > 0, ATSPEED_ERR_INVALID, "fsm_create: bad arguments");
  ATS_REQUIRE(row_ptr[0] == 0 && row_ptr[n_nodes] == n_edges, ATSPEED_ERR_INVALID, "fsm_create: row_ptr must span [0, n_edges]");
  for (int n = 0; n < n_nodes; ++n) {
    ATS_REQUIRE(row_ptr[n + 1] >= row_ptr[n], ATSPEED_ERR_INVALID, "fsm_create: row_ptr not monotone at node %d", n);
    for (int e = row_ptr[n]; e < row_ptr[n + 1]; ++e) {
      ATS_REQUIRE(tok[e] >= 0 && tok[e] < vocab_size, ATSPEED_ERR_INVALID, "fsm_create: token %d out of vocab at edge %d", tok[e], e);
      ATS_REQUIRE(nxt[e] >= 0 && nxt[e] < n_nodes, ATSPEED_ERR_INVALID, "fsm_create: next node %d out of range at edge %d", nxt[e], e);
      ATS_REQUIRE(e == row_ptr[n] || tok[e] > tok[e - 1], ATSPEED_ERR_INVALID, "fsm_create: children of node %d not strictly ascending", n);
    }
  }
  ATS_REQUIRE((int64_t)ATSPEED_MAX_BEAMS * vocab_size < (int64_t)0x7fffffff, ATSPEED_ERR_CAPACITY, "fsm_create: vocab too large for 32-bit flat ids");
  atspeed_fsm* f = new atspeed_fsm();
  memset(f, 0, sizeof(*f));
  size_t ne = (size_t)std::max(n_edges, 1);
  ATS_HIP(hipMalloc(&f->d_row_ptr, (size_t)(n_nodes + 1) * sizeof(int32_t)));
  ATS_HIP(hipMalloc(&f->d_tok, ne * sizeof(int32_t)));
  ATS_HIP(hipMalloc(&f->d_nxt, ne * sizeof(int32_t)));
  ATS_HIP(hipMemcpy(f->d_row_ptr, row_ptr, (size_t)(n_nodes + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
  if (n_edges > 0) {
    ATS_HIP(hipMemcpy(f->d_tok, tok, (size_t)n_edges * sizeof(int32_t), hipMemcpyHostToDevice));
    ATS_HIP(hipMemcpy(f->d_nxt, nxt, (size_t)n_edges * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  f->dev = FsmDev{f->d_row_ptr, f->d_tok, f->d_nxt, n_nodes, n_edges, vocab_size};
  *out = f;
  return ATSPEED_OK;
}

extern "C" void atspeed_fsm_destroy(atspeed_fsm* f) {
  if (!f) return;
  hipFree(f->d_row_ptr); hipFree(f->d_tok); hipFree(f->d_nxt);
  delete f;
}

extern "C" int atspeed_trie_flatten(const int32_t* seq_tokens, const int32_t* seq_offsets, int32_t n_seqs,
                                    int32_t* row_ptr_out, int32_t* tok_out, int32_t* nxt_out, int32_t* n_nodes_out,
                                    int32_t* n_edges_out) {
  ATS_REQUIRE(seq_offsets && n_nodes_out && n_edges_out && n_seqs >= 0, ATSPEED_ERR_INVALID, "trie_flatten: bad arguments");
  // build with ordered child maps, then renumber breadth-first so a level's nodes are contiguous
  std::vector<std::map<int32_t, int32_t>> kids(1);
  for (int s = 0; s < n_seqs; ++s) {
    int32_t node = 0;
    for (int32_t i = seq_offsets[s]; i < seq_offsets[s + 1]; ++i) {
      int32_t t = seq_tokens[i];
      auto it = kids[node].find(t);
      if (it == kids[node].end()) {
        int32_t nn = (int32_t)kids.size();
        kids[node][t] = nn;
        kids.emplace_back();
        node = nn;
      } else {
        node = it->second;
      }
    }
  }
  std::vector<int32_t> order{0}, newid(kids.size(), -1);
  newid[0] = 0;
  for (size_t i = 0; i < order.size(); ++i)
    for (auto& kv : kids[order[i]]) { newid[kv.second] = (int32_t)order.size(); order.push_back(kv.second); }
  int32_t n_nodes = (int32_t)order.size(), n_edges = n_nodes - 1;
  *n_nodes_out = n_nodes;
  *n_edges_out = n_edges;
  if (!row_ptr_out || !tok_out || !nxt_out) return ATSPEED_OK;
  int32_t e = 0;
  for (int32_t i = 0; i < n_nodes; ++i) {
    row_ptr_out[i] = e;
    for (auto& kv : kids[order[i]]) { tok_out[e] = kv.first; nxt_out[e] = newid[kv.second]; ++e; }
  }
  row_ptr_out[n_nodes] = e;
  return ATSPEED_OK;
}

// ---------------------------------------------------------------------------- model
// Weights/config/RoPE tables are shared and read-only; everything a forward WRITES lives in a LlamaCtx, so several
// user streams can run the same model concurrently (one context per decoder).
struct LlamaCtx {
  void *kcache = nullptr, *vcache = nullptr;     // [n_layers][max_slots][hidden]
  void *h = nullptr, *xn = nullptr, *qkv = nullptr, *att = nullptr, *act = nullptr;   // activations
  float* logits = nullptr;                       // [max_logit_rows][logits_ld]
  void* ws = nullptr; size_t ws_bytes = 0;       // split-K slabs
  // optional per-GEMM hipEvent brackets (bench.py roofline): 0 qkv, 1 o_proj, 2 gate_up, 3 down, 4 lm_head
  bool prof_pending = false;
  std::vector<hipEvent_t> prof_ev;               // 2 events per bracket
  std::vector<int> prof_kind, prof_m;            // kind / M per bracket of the pending forward
};

struct atspeed_llama {
  atspeed_llama_config cfg;
  const void *embed, *final_norm, *lm_head;
  std::vector<atspeed_llama_layer_weights> layers;
  int esz, head_dim, vis_words, logits_ld;
  float *cos_tab, *sin_tab;                      // [max_slots][head_dim/2]
  LlamaCtx* ctx0;                                // context of the plain atspeed_llama_forward API
  std::vector<LlamaCtx*> ctxs;                   // every live context (profiling harvest)
  bool prof_on = false;
  double prof_ms[5] = {0, 0, 0, 0, 0};
  long prof_cnt[5] = {0, 0, 0, 0, 0};
  long prof_rows[5] = {0, 0, 0, 0, 0};           // sum of M over the bracketed launches
};

static void prof_harvest(atspeed_llama* m, LlamaCtx* cx) {
  if (!cx->prof_pending) return;
  for (size_t b = 0; b < cx->prof_kind.size(); ++b) {
    float ms = 0.f;
    if (hipEventSynchronize(cx->prof_ev[2 * b + 1]) == hipSuccess &&
        hipEventElapsedTime(&ms, cx->prof_ev[2 * b], cx->prof_ev[2 * b + 1]) == hipSuccess) {
      int kd = cx->prof_kind[b];
      m->prof_ms[kd] += ms; m->prof_cnt[kd] += 1; m->prof_rows[kd] += cx->prof_m[b];
    }
  }
  cx->prof_kind.clear(); cx->prof_m.clear();
  cx->prof_pending = false;
}

struct ProfBracket {
  LlamaCtx* cx; hipStream_t st; bool on; size_t idx;
  ProfBracket(atspeed_llama* m, LlamaCtx* cx_, int kind, int rows, hipStream_t st_) : cx(cx_), st(st_), on(m->prof_on), idx(0) {
    if (!on) return;
    idx = cx->prof_kind.size();
    while (cx->prof_ev.size() < 2 * (idx + 1)) { hipEvent_t e; hipEventCreate(&e); cx->prof_ev.push_back(e); }
    cx->prof_kind.push_back(kind); cx->prof_m.push_back(rows);
    hipEventRecord(cx->prof_ev[2 * idx], st);
  }
  ~ProfBracket() { if (on) { hipEventRecord(cx->prof_ev[2 * idx + 1], st); cx->prof_pending = true; } }
};

static size_t gemm_ws_for(const atspeed_llama_config& c) {
  size_t best = 0;
  for (int m = 1; m <= c.max_tokens; ++m) {      // the split-K plan depends on the exact M: take the true maximum
    best = std::max(best, ats_gemm_workspace_bytes(m, 3 * c.hidden, c.hidden, c.dtype));
    best = std::max(best, ats_gemm_workspace_bytes(m, c.hidden, c.hidden, c.dtype));
    best = std::max(best, ats_gemm_workspace_bytes(m, 2 * c.ffn, c.hidden, c.dtype));
    best = std::max(best, ats_gemm_workspace_bytes(m, c.hidden, c.ffn, c.dtype));
    int lm = std::min(m, c.max_logit_rows);
    best = std::max(best, ats_gemm_workspace_bytes(lm, c.vocab_size, c.hidden, c.dtype));
  }
  return best + (1 << 20);
}

static int ctx_create(atspeed_llama* m, LlamaCtx** out) {
  const atspeed_llama_config& c = m->cfg;
  LlamaCtx* cx = new LlamaCtx();
  size_t T = c.max_tokens, H = c.hidden, e = m->esz;
  size_t kv = (size_t)c.n_layers * c.max_slots * H * e;
  ATS_HIP(hipMalloc(&cx->kcache, kv));
  ATS_HIP(hipMalloc(&cx->vcache, kv));
  ATS_HIP(hipMemset(cx->kcache, 0, kv));
  ATS_HIP(hipMemset(cx->vcache, 0, kv));
  ATS_HIP(hipMalloc(&cx->h, T * H * e));
  ATS_HIP(hipMalloc(&cx->xn, T * H * e));
  ATS_HIP(hipMalloc(&cx->qkv, T * 3 * H * e));
  ATS_HIP(hipMalloc(&cx->att, T * H * e));
  ATS_HIP(hipMalloc(&cx->act, T * (size_t)c.ffn * e));
  ATS_HIP(hipMalloc((void**)&cx->logits, (size_t)c.max_logit_rows * m->logits_ld * sizeof(float)));
  cx->ws_bytes = gemm_ws_for(c);
  ATS_HIP(hipMalloc(&cx->ws, cx->ws_bytes));
  m->ctxs.push_back(cx);
  *out = cx;
  return ATSPEED_OK;
}

static void ctx_destroy(atspeed_llama* m, LlamaCtx* cx) {
  if (!cx) return;
  hipFree(cx->kcache); hipFree(cx->vcache); hipFree(cx->h); hipFree(cx->xn); hipFree(cx->qkv); hipFree(cx->att);
  hipFree(cx->act); hipFree(cx->logits); hipFree(cx->ws);
  for (hipEvent_t e : cx->prof_ev) hipEventDestroy(e);
  if (m) m->ctxs.erase(std::remove(m->ctxs.begin(), m->ctxs.end(), cx), m->ctxs.end());
  delete cx;
}

extern "C" int atspeed_llama_create(const atspeed_llama_config* cfg, const void* embed, const void* final_norm,
                                    const void* lm_head, const atspeed_llama_layer_weights* layers, atspeed_llama** out) {
  ATS_REQUIRE(cfg && embed && final_norm && lm_head && layers && out, ATSPEED_ERR_INVALID, "llama_create: null argument");
  ATS_REQUIRE(cfg->dtype == ATSPEED_F32 || cfg->dtype == ATSPEED_BF16, ATSPEED_ERR_INVALID, "llama_create: bad dtype");
  ATS_REQUIRE(cfg->n_heads > 0 && cfg->hidden % cfg->n_heads == 0, ATSPEED_ERR_INVALID, "llama_create: hidden %% n_heads != 0");
  int hd = cfg->hidden / cfg->n_heads;
  ATS_REQUIRE(hd % 8 == 0 && hd <= 256, ATSPEED_ERR_INVALID, "llama_create: head_dim %d unsupported", hd);
  ATS_REQUIRE(cfg->hidden % 8 == 0 && cfg->ffn % 16 == 0, ATSPEED_ERR_INVALID, "llama_create: hidden %% 8 / ffn %% 16 required");
  ATS_REQUIRE(cfg->max_slots > 0 && cfg->max_slots % 64 == 0 && cfg->max_slots <= 2048, ATSPEED_ERR_INVALID,
              "llama_create: max_slots must be a multiple of 64, <= 2048");
  ATS_REQUIRE(cfg->max_tokens > 0 && cfg->max_logit_rows > 0 && cfg->max_logit_rows <= cfg->max_tokens, ATSPEED_ERR_INVALID,
              "llama_create: bad token limits");
  atspeed_llama* m = new atspeed_llama();
  m->cfg = *cfg;
  m->embed = embed; m->final_norm = final_norm; m->lm_head = lm_head;
  m->layers.assign(layers, layers + cfg->n_layers);
  m->esz = cfg->dtype == ATSPEED_F32 ? 4 : 2;
  m->head_dim = hd;
  m->vis_words = cfg->max_slots / 64;
  m->logits_ld = (cfg->vocab_size + 63) / 64 * 64;
  ATS_TRY(ctx_create(m, &m->ctx0));
  // RoPE tables: HF computes inv_freq and the angle in fp32; cos/sin of that angle in double, rounded once
  int half = hd / 2;
  std::vector<float> ct((size_t)cfg->max_slots * half), stv((size_t)cfg->max_slots * half);
  for (int p = 0; p < cfg->max_slots; ++p)
    for (int i = 0; i < half; ++i) {
      float inv = 1.0f / powf(cfg->rope_theta, (float)(2 * i) / (float)hd);
      float ang = (float)p * inv;
      ct[(size_t)p * half + i] = (float)cos((double)ang);
      stv[(size_t)p * half + i] = (float)sin((double)ang);
    }
  ATS_HIP(hipMalloc((void**)&m->cos_tab, ct.size() * sizeof(float)));
  ATS_HIP(hipMalloc((void**)&m->sin_tab, stv.size() * sizeof(float)));
  ATS_HIP(hipMemcpy(m->cos_tab, ct.data(), ct.size() * sizeof(float), hipMemcpyHostToDevice));
  ATS_HIP(hipMemcpy(m->sin_tab, stv.data(), stv.size() * sizeof(float), hipMemcpyHostToDevice));
  *out = m;
  return ATSPEED_OK;
}

extern "C" void atspeed_llama_destroy(atspeed_llama* m) {
  if (!m) return;
  while (!m->ctxs.empty()) ctx_destroy(m, m->ctxs.back());
  hipFree(m->cos_tab); hipFree(m->sin_tab);
  delete m;
}

extern "C" int atspeed_llama_profile(atspeed_llama* m, int32_t enable, double* ms_out, int64_t* count_out, int64_t* rows_out) {
  ATS_REQUIRE(m, ATSPEED_ERR_INVALID, "profile: null model");
  for (LlamaCtx* cx : m->ctxs) prof_harvest(m, cx);
  for (int i = 0; i < 5; ++i) {
    if (ms_out) ms_out[i] = m->prof_ms[i];
    if (count_out) count_out[i] = m->prof_cnt[i];
    if (rows_out) rows_out[i] = m->prof_rows[i];
  }
  if (enable >= 0) {
    m->prof_on = enable != 0;
    for (int i = 0; i < 5; ++i) { m->prof_ms[i] = 0; m->prof_cnt[i] = 0; m->prof_rows[i] = 0; }
  }
  return ATSPEED_OK;
}

extern "C" float* atspeed_llama_logits(atspeed_llama* m) { return m ? m->ctx0->logits : nullptr; }
extern "C" int32_t atspeed_llama_logits_ld(const atspeed_llama* m) { return m ? m->logits_ld : 0; }

static int llama_forward(atspeed_llama* m, LlamaCtx* cx, const int32_t* ids, const int32_t* pos, const int32_t* slots,
                         const uint64_t* vis, int T, int S, int n_logit_rows, float* logits_out, hipStream_t st) {
  const atspeed_llama_config& c = m->cfg;
  ATS_REQUIRE(T >= 1 && T <= c.max_tokens, ATSPEED_ERR_CAPACITY, "forward: %d tokens exceed max_tokens %d", T, c.max_tokens);
  ATS_REQUIRE(S >= 1 && S <= c.max_slots, ATSPEED_ERR_CAPACITY, "forward: %d slots exceed max_slots %d", S, c.max_slots);
  ATS_REQUIRE(n_logit_rows >= 0 && n_logit_rows <= T && n_logit_rows <= c.max_logit_rows, ATSPEED_ERR_CAPACITY,
              "forward: %d logit rows exceed the limit %d", n_logit_rows, c.max_logit_rows);
  const int H = c.hidden, dt = c.dtype;
  const size_t e = m->esz;
  if (m->prof_on) prof_harvest(m, cx);
  const size_t layer_kv = (size_t)c.max_slots * H * e;
  ATS_TRY(ats_embed(m->embed, ids, cx->h, T, H, c.vocab_size, dt, st));
  ATS_TRY(ats_rmsnorm(cx->h, m->layers[0].input_norm, cx->xn, T, H, c.rms_eps, dt, st));
  for (int l = 0; l < c.n_layers; ++l) {
    const atspeed_llama_layer_weights& w = m->layers[l];
    char* kc = (char*)cx->kcache + l * layer_kv;
    char* vc = (char*)cx->vcache + l * layer_kv;
    // cx->xn holds rmsnorm(h) * input_norm here (from the embed above or the previous layer's fused down_proj epilogue)
    { ProfBracket pb(m, cx, 0, T, st);
      ATS_TRY(ats_gemm(cx->xn, w.wqkv, cx->qkv, T, 3 * H, H, H, 3 * H, dt, EPI_STORE, cx->ws, cx->ws_bytes, st)); }
    ATS_TRY(ats_rope_kv(cx->qkv, pos, slots, m->cos_tab, m->sin_tab, kc, vc, T, c.n_heads, m->head_dim, c.max_slots, dt, st));
    ATS_TRY(ats_tree_attention(cx->qkv, 3 * H, kc, vc, vis, m->vis_words, cx->att, H, T, S, c.n_heads, m->head_dim, dt, st));
    { ProfBracket pb(m, cx, 1, T, st);     // h += att Wo^T ; xn = rmsnorm(h) * post_norm
      ATS_TRY(ats_gemm_resid_norm(cx->att, w.wo, cx->h, T, H, H, H, H, dt, w.post_norm, cx->xn, c.rms_eps, cx->ws, cx->ws_bytes, st)); }
    { ProfBracket pb(m, cx, 2, T, st);
      ATS_TRY(ats_gemm(cx->xn, w.wgu, cx->act, T, 2 * c.ffn, H, H, c.ffn, dt, EPI_SWIGLU, cx->ws, cx->ws_bytes, st)); }
    { ProfBracket pb(m, cx, 3, T, st);     // h += act Wd^T ; xn = rmsnorm(h) * next layer's input_norm
      if (l + 1 < c.n_layers) {
        ATS_TRY(ats_gemm_resid_norm(cx->act, w.wd, cx->h, T, H, c.ffn, c.ffn, H, dt, m->layers[l + 1].input_norm, cx->xn, c.rms_eps,
                                    cx->ws, cx->ws_bytes, st));
      } else {
        ATS_TRY(ats_gemm(cx->act, w.wd, cx->h, T, H, c.ffn, c.ffn, H, dt, EPI_RESID, cx->ws, cx->ws_bytes, st));
      } }
  }
  if (n_logit_rows > 0) {
    char* hrows = (char*)cx->h + (size_t)(T - n_logit_rows) * H * e;
    ATS_TRY(ats_rmsnorm(hrows, m->final_norm, cx->xn, n_logit_rows, H, c.rms_eps, dt, st));
    float* lo = logits_out ? logits_out : cx->logits;
    ProfBracket pb(m, cx, 4, n_logit_rows, st);
    ATS_TRY(ats_gemm(cx->xn, m->lm_head, lo, n_logit_rows, c.vocab_size, H, H, m->logits_ld, dt, EPI_F32, cx->ws, cx->ws_bytes, st));
  }
  return ATSPEED_OK;
}

extern "C" int atspeed_llama_forward(atspeed_llama* m, const int32_t* ids, const int32_t* pos, const int32_t* slots,
                                     const uint64_t* vis, int32_t n_tokens, int32_t n_slots_visible, int32_t n_logit_rows,
                                     float* logits_out, void* stream) {
  ATS_REQUIRE(m && ids && pos && slots && vis, ATSPEED_ERR_INVALID, "forward: null argument");
  return llama_forward(m, m->ctx0, ids, pos, slots, vis, n_tokens, n_slots_visible, n_logit_rows, logits_out, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------- decoder
namespace {
constexpr int MAXB = ATSPEED_MAX_BEAMS;
constexpr int LMAX = ATSPEED_MAX_NEW_TOKENS;
constexpr int NBLK = ATSPEED_MAX_GAMMA + 1;
constexpr int kMaxEvents = 4 * (ATSPEED_MAX_NEW_TOKENS + 2);
}  // namespace

// One user stream: private forward contexts (KV + activations) for target and draft, device-side beam state, a
// pinned mailbox, and the bookkeeping of the call in flight (so several decoders can be interleaved, see
// atspeed_bssd_generate_batch).
struct atspeed_decoder {
  atspeed_llama *target, *draft;
  LlamaCtx *tctx, *dctx;
  int max_prompt, tok_cap, W;
  char* arena;
  TokBuf tin[2], dround;
  BeamSet round_beams[2], blk[NBLK];
  float* lse;
  Mailbox* mail_dev;
  Mailbox* mail_host;       // pinned
  int32_t* trace_host;      // pinned: per round [dl][MAXB] draft flat ids
  std::vector<int32_t> trace;   // rounds: {dl, n_matches, nb, flat ids...}
  hipEvent_t ev[kMaxEvents];
  hipStream_t own_stream;   // used by the batch API
  // ---- state of the generate call in flight
  struct Run {
    const atspeed_fsm* fsm; int gamma, max_new, k, dk;
    int32_t* out_tokens; float* out_scores; atspeed_gen_stats* stats_out;
    hipStream_t st;
    atspeed_gen_stats s;
    int cur, gen, base, n0, nb, dl, nev, e_begin, e_end;
    bool reingest, final_step, exported, done;
    std::vector<int> ev_marks;
  } run;
};

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static void carve_tokbuf(char*& p, TokBuf& tb, int rows, int W) {
  tb.ids = (int32_t*)p;  p += align_up((size_t)rows * 4, 256);
  tb.pos = (int32_t*)p;  p += align_up((size_t)rows * 4, 256);
  tb.slot = (int32_t*)p; p += align_up((size_t)rows * 4, 256);
  tb.vis = (uint64_t*)p; p += align_up((size_t)rows * W * 8, 256);
}
static void carve_beams(char*& p, BeamSet& b) {
  b.score = (float*)p;    p += 256;
  b.node = (int32_t*)p;   p += 256;
  b.parent = (int32_t*)p; p += 256;
  b.tok = (int32_t*)p;    p += 256;
  b.flat = (int32_t*)p;   p += 256;
  b.seq = (int32_t*)p;    p += align_up((size_t)MAXB * LMAX * 4, 256);
}

extern "C" int atspeed_decoder_create(atspeed_llama* target, atspeed_llama* draft, int32_t max_prompt, atspeed_decoder** out) {
  ATS_REQUIRE(target && out && max_prompt >= 1, ATSPEED_ERR_INVALID, "decoder_create: bad arguments");
  if (draft) {
    ATS_REQUIRE(draft->cfg.vocab_size == target->cfg.vocab_size, ATSPEED_ERR_INVALID, "decoder_create: draft/target vocab differ");
    ATS_REQUIRE(draft->cfg.max_slots == target->cfg.max_slots, ATSPEED_ERR_INVALID, "decoder_create: draft/target max_slots differ");
  }
  atspeed_decoder* d = new atspeed_decoder();
  d->target = target; d->draft = draft;
  d->tctx = nullptr; d->dctx = nullptr;
  ATS_TRY(ctx_create(target, &d->tctx));
  if (draft) ATS_TRY(ctx_create(draft, &d->dctx));
  d->max_prompt = max_prompt;
  d->W = target->vis_words;
  d->tok_cap = max_prompt + ATSPEED_MAX_GAMMA * MAXB + MAXB;
  size_t tb_bytes = 3 * align_up((size_t)d->tok_cap * 4, 256) + align_up((size_t)d->tok_cap * d->W * 8, 256);
  size_t bs_bytes = 5 * 256 + align_up((size_t)MAXB * LMAX * 4, 256);
  size_t total = 3 * tb_bytes + (2 + NBLK) * bs_bytes + align_up((size_t)target->cfg.max_logit_rows * 4, 256) + 256;
  ATS_HIP(hipMalloc((void**)&d->arena, total));
  ATS_HIP(hipMemset(d->arena, 0, total));
  char* p = d->arena;
  carve_tokbuf(p, d->tin[0], d->tok_cap, d->W);
  carve_tokbuf(p, d->tin[1], d->tok_cap, d->W);
  carve_tokbuf(p, d->dround, d->tok_cap, d->W);
  carve_beams(p, d->round_beams[0]);
  carve_beams(p, d->round_beams[1]);
  for (int i = 0; i < NBLK; ++i) carve_beams(p, d->blk[i]);
  d->lse = (float*)p; p += align_up((size_t)target->cfg.max_logit_rows * 4, 256);
  d->mail_dev = (Mailbox*)p; p += 256;
  ATS_HIP(hipHostMalloc((void**)&d->mail_host, sizeof(Mailbox)));
  ATS_HIP(hipHostMalloc((void**)&d->trace_host, sizeof(int32_t) * ATSPEED_MAX_GAMMA * MAXB));
  for (int i = 0; i < kMaxEvents; ++i) ATS_HIP(hipEventCreate(&d->ev[i]));
  ATS_HIP(hipStreamCreateWithFlags(&d->own_stream, hipStreamNonBlocking));
  d->run.done = true;
  *out = d;
  return ATSPEED_OK;
}

extern "C" void atspeed_decoder_destroy(atspeed_decoder* d) {
  if (!d) return;
  hipStreamSynchronize(d->own_stream);
  hipStreamDestroy(d->own_stream);
  ctx_destroy(d->target, d->tctx);
  if (d->draft) ctx_destroy(d->draft, d->dctx);
  hipFree(d->arena);
  hipHostFree(d->mail_host);
  hipHostFree(d->trace_host);
  for (int i = 0; i < kMaxEvents; ++i) hipEventDestroy(d->ev[i]);
  delete d;
}

static TokBuf tb_offset(const TokBuf& t, int row, int W) {
  TokBuf o;
  o.ids = t.ids + row; o.pos = t.pos + row; o.slot = t.slot + row; o.vis = t.vis + (size_t)row * W;
  return o;
}

static int check_common(atspeed_decoder* d, const int32_t* prompt, int P, const atspeed_fsm* fsm, int start_node,
                        int max_new, int k, const int32_t* out_tokens, const float* out_scores) {
  ATS_REQUIRE(d && prompt && fsm && out_tokens && out_scores, ATSPEED_ERR_INVALID, "generate: null argument");
  ATS_REQUIRE(P >= 1 && P <= d->max_prompt, ATSPEED_ERR_CAPACITY, "generate: prompt length %d exceeds max_prompt %d", P, d->max_prompt);
  ATS_REQUIRE(max_new >= 1 && max_new <= LMAX, ATSPEED_ERR_CAPACITY, "generate: max_new_tokens %d out of [1,%d]", max_new, LMAX);
  ATS_REQUIRE(k >= 1 && k <= MAXB, ATSPEED_ERR_CAPACITY, "generate: beam size %d out of [1,%d]", k, MAXB);
  ATS_REQUIRE(fsm->dev.vocab == d->target->cfg.vocab_size, ATSPEED_ERR_INVALID, "generate: constraint vocab %d != model vocab %d",
              fsm->dev.vocab, d->target->cfg.vocab_size);
  ATS_REQUIRE(start_node >= 0 && start_node < fsm->dev.n_nodes, ATSPEED_ERR_INVALID, "generate: start node out of range");
  return ATSPEED_OK;
}

static int read_mailbox(atspeed_decoder* d, hipStream_t st) {
  ATS_HIP(hipStreamSynchronize(st));
  if (d->mail_host->status == ATSPEED_ERR_CONSTRAINT) {
    atspeed_set_error("`prefix_allowed_tokens_fn` returned an empty list for batch ID 0. This means that the constraint is unsatisfiable.");
    return ATSPEED_ERR_CONSTRAINT;
  }
  if (d->mail_host->status != 0) {
    atspeed_set_error("decoder: device status %d (candidate capacity exceeded?)", d->mail_host->status);
    return d->mail_host->status;
  }
  return ATSPEED_OK;
}

static int run_mark(atspeed_decoder* d) {
  atspeed_decoder::Run& r = d->run;
  if (r.nev < kMaxEvents) hipEventRecord(d->ev[r.nev], r.st);
  return r.nev++;
}

// -- BSSD as a resumable state machine: begin -> (enqueue_round -> finish_round)* ------------------------------
static int bssd_begin(atspeed_decoder* d, const int32_t* prompt, int P, const atspeed_fsm* fsm, int start_node, int gamma,
                      int max_new, int k, int dk, int32_t* out_tokens, float* out_scores, atspeed_gen_stats* stats,
                      hipStream_t st) {
  ATS_TRY(check_common(d, prompt, P, fsm, start_node, max_new, k, out_tokens, out_scores));
  ATS_REQUIRE(d->draft, ATSPEED_ERR_INVALID, "bssd: decoder was created without a draft model");
  ATS_REQUIRE(dk >= k && dk <= MAXB, ATSPEED_ERR_CAPACITY, "bssd: draft beam size %d must be in [k=%d, %d]", dk, k, MAXB);
  ATS_REQUIRE(gamma >= 1 && gamma <= ATSPEED_MAX_GAMMA, ATSPEED_ERR_CAPACITY, "bssd: gamma %d out of [1,%d]", gamma, ATSPEED_MAX_GAMMA);
  atspeed_decoder::Run& r = d->run;
  r.fsm = fsm; r.gamma = gamma; r.max_new = max_new; r.k = k; r.dk = dk;
  r.out_tokens = out_tokens; r.out_scores = out_scores; r.stats_out = stats; r.st = st;
  memset(&r.s, 0, sizeof(r.s));
  r.cur = 0; r.gen = 0; r.base = 0; r.n0 = P; r.nb = 1; r.dl = 0; r.nev = 0;
  r.reingest = false; r.final_step = false; r.exported = false; r.done = false;
  r.ev_marks.clear();
  d->trace.clear();
  ATS_TRY(ats_init_prompt(d->tin[0], prompt, P, d->W, d->round_beams[0], start_node, d->target->cfg.vocab_size, d->mail_dev, st));
  r.e_begin = run_mark(d);
  return ATSPEED_OK;
}

// enqueue everything up to the next point where the host must look at the device (no synchronisation here)
static int bssd_enqueue_round(atspeed_decoder* d) {
  atspeed_decoder::Run& r = d->run;
  hipStream_t st = r.st;
  atspeed_llama *T = d->target, *D = d->draft;
  const int W = d->W, V = T->cfg.vocab_size, k = r.k, dk = r.dk;
  if (r.gen >= r.max_new) {                                                          // nothing left: export
    r.final_step = true;
  } else {
    r.dl = std::min(r.gamma, r.max_new - r.gen - 1);                                 // beamSD.py:504
    TokBuf& tin = d->tin[r.cur];
    BeamSet& beams = d->round_beams[r.cur];
    if (r.dl == 0) {                                                                 // :505-509
      ATS_TRY(llama_forward(T, d->tctx, tin.ids, tin.pos, tin.slot, tin.vis, r.n0, r.base + r.n0, r.nb, nullptr, st));
      r.s.n_target_forwards++;
      ATS_TRY(ats_lse_rows(d->tctx->logits, r.nb, V, T->logits_ld, d->lse, st));
      BeamStepArgs a{};
      a.src = beams; a.n_src = r.nb; a.gen_len = r.gen;
      a.logits = d->tctx->logits; a.ld = T->logits_ld; a.lse = d->lse; a.fsm = r.fsm->dev; a.k = k;
      a.dst = d->round_beams[r.cur ^ 1]; a.emit = 0; a.mail = d->mail_dev; a.vis_words = W;
      ATS_TRY(ats_beam_step(a, st));
      r.cur ^= 1;
      r.gen += 1;
      r.final_step = true;
    } else {
      const int n0 = r.n0, nb = r.nb, dl = r.dl, base = r.base;
      ATS_REQUIRE(n0 + dl * dk <= d->tok_cap && n0 + dl * dk <= T->cfg.max_tokens, ATSPEED_ERR_CAPACITY, "bssd: packed target input too long");
      ATS_REQUIRE(base + n0 + dl * dk <= T->cfg.max_slots, ATSPEED_ERR_CAPACITY, "bssd: KV slots exhausted (%d needed, %d available)",
                  base + n0 + dl * dk, T->cfg.max_slots);
      ATS_REQUIRE(nb + dl * dk <= T->cfg.max_logit_rows, ATSPEED_ERR_CAPACITY, "bssd: %d logit rows exceed max_logit_rows", nb + dl * dk);
      r.ev_marks.push_back(run_mark(d));
      // ---- 1. draft: dl steps of one_step_beam_search (:108-179)
      for (int i = 0; i < dl; ++i) {
        int n_src;
        if (i == 0) {
          n_src = nb;
          if (r.reingest) {
            ATS_TRY(llama_forward(D, d->dctx, d->dround.ids, d->dround.pos, d->dround.slot, d->dround.vis, dk + k, base + n0, nb, nullptr, st));
          } else {
            ATS_TRY(llama_forward(D, d->dctx, tin.ids, tin.pos, tin.slot, tin.vis, n0, base + n0, nb, nullptr, st));
          }
        } else {
          n_src = dk;
          TokBuf b = tb_offset(tin, n0 + (i - 1) * dk, W);
          ATS_TRY(llama_forward(D, d->dctx, b.ids, b.pos, b.slot, b.vis, dk, base + n0 + i * dk, dk, nullptr, st));
        }
        r.s.n_draft_forwards++;
        ATS_TRY(ats_lse_rows(d->dctx->logits, n_src, V, D->logits_ld, d->lse, st));
        BeamStepArgs a{};
        a.src = i == 0 ? beams : d->blk[i]; a.n_src = n_src; a.gen_len = r.gen + i;
        a.logits = d->dctx->logits; a.ld = D->logits_ld; a.lse = d->lse; a.fsm = r.fsm->dev; a.k = dk;
        a.dst = d->blk[i + 1]; a.emit = 1;
        a.in = tin; a.in_row0 = i == 0 ? n0 - nb : n0 + (i - 1) * dk;
        a.out = tin; a.out_row0 = n0 + i * dk; a.out_slot0 = base + n0 + i * dk; a.vis_words = W;
        a.mail = d->mail_dev;
        ATS_TRY(ats_beam_step(a, st));
      }
      r.ev_marks.push_back(run_mark(d));
      // ---- 2. target: ONE forward over round inputs ++ all draft blocks (:190-232)
      const int Tn = n0 + dl * dk, rows = nb + dl * dk;
      ATS_TRY(llama_forward(T, d->tctx, tin.ids, tin.pos, tin.slot, tin.vis, Tn, base + Tn, rows, nullptr, st));
      r.s.n_target_forwards++;
      r.ev_marks.push_back(run_mark(d));
      // ---- 3. verify (:242-456)
      ATS_TRY(ats_lse_rows(d->tctx->logits, rows, V, T->logits_ld, d->lse, st));
      VerifyArgs va{};
      va.blk[0] = beams;
      for (int i = 1; i <= dl; ++i) va.blk[i] = d->blk[i];
      va.nb = nb; va.dl = dl; va.k = k; va.dk = dk; va.gen_len0 = r.gen;
      va.logits = d->tctx->logits; va.ld = T->logits_ld; va.lse = d->lse; va.fsm = r.fsm->dev;
      va.cur = tin; va.n0 = n0; va.next = d->tin[r.cur ^ 1]; va.dnext = d->dround; va.vis_words = W;
      va.res = d->round_beams[r.cur ^ 1]; va.mail = d->mail_dev;
      ATS_TRY(ats_verify_walk(va, st));
      r.ev_marks.push_back(run_mark(d));
      for (int i = 1; i <= dl; ++i)   // trace of the draft's flat ids for parity tests (tiny copies, same stream)
        ATS_HIP(hipMemcpyAsync(d->trace_host + (i - 1) * MAXB, d->blk[i].flat, sizeof(int32_t) * dk, hipMemcpyDeviceToHost, st));
    }
  }
  if (r.final_step && !r.exported) {
    ATS_TRY(ats_export_beams(d->round_beams[r.cur], k, r.max_new, r.out_tokens, r.out_scores, st));
    r.e_end = run_mark(d);
    r.exported = true;
  }
  ATS_HIP(hipMemcpyAsync(d->mail_host, d->mail_dev, sizeof(Mailbox), hipMemcpyDeviceToHost, st));
  return ATSPEED_OK;
}

// wait for the enqueued work, read the mailbox, advance the state; sets run.done after the final step
static int bssd_finish_round(atspeed_decoder* d) {
  atspeed_decoder::Run& r = d->run;
  ATS_TRY(read_mailbox(d, r.st));                                                   // the round's only sync
  if (r.final_step) {
    r.s.n_valid = d->mail_host->n_valid;
    if (r.nev <= kMaxEvents) {
      float ms = 0.f;
      hipEventElapsedTime(&ms, d->ev[r.e_begin], d->ev[r.e_end]); r.s.total_ms = ms;
      for (size_t i = 0; i + 3 < r.ev_marks.size(); i += 4) {
        hipEventElapsedTime(&ms, d->ev[r.ev_marks[i]], d->ev[r.ev_marks[i + 1]]); r.s.draft_ms += ms;
        hipEventElapsedTime(&ms, d->ev[r.ev_marks[i + 1]], d->ev[r.ev_marks[i + 2]]); r.s.target_ms += ms;
        hipEventElapsedTime(&ms, d->ev[r.ev_marks[i + 2]], d->ev[r.ev_marks[i + 3]]); r.s.verify_ms += ms;
      }
    }
    if (r.stats_out) *r.stats_out = r.s;
    r.done = true;
    return ATSPEED_OK;
  }
  const int nm = d->mail_host->n_matches, dl = r.dl, dk = r.dk;
  d->trace.push_back(dl); d->trace.push_back(nm); d->trace.push_back(r.nb);
  for (int i = 0; i < dl; ++i) for (int j = 0; j < dk; ++j) d->trace.push_back(d->trace_host[i * MAXB + j]);
  if (r.s.n_run < ATSPEED_MAX_NEW_TOKENS) r.s.accept_steps[r.s.n_run] = nm;
  r.s.n_run++;
  r.s.total_accept_steps += nm;
  r.base += r.n0 + nm * dk;             // compact: keep up to the end of block nm
  r.n0 = r.k; r.nb = r.k;
  r.gen += nm + 1;                      // :522
  r.reingest = (nm == dl);
  r.cur ^= 1;
  return ATSPEED_OK;
}

extern "C" int atspeed_bssd_generate(atspeed_decoder* d, const int32_t* prompt, int32_t P, const atspeed_fsm* fsm,
                                     int32_t start_node, int32_t gamma, int32_t max_new, int32_t k, int32_t dk,
                                     int32_t* out_tokens, float* out_scores, atspeed_gen_stats* stats, void* stream) {
  ATS_TRY(bssd_begin(d, prompt, P, fsm, start_node, gamma, max_new, k, dk, out_tokens, out_scores, stats, (hipStream_t)stream));
  while (!d->run.done) {
    ATS_TRY(bssd_enqueue_round(d));
    ATS_TRY(bssd_finish_round(d));
  }
  return ATSPEED_OK;
}

// Several independent users interleaved on the decoders' own streams: while the host waits for user A's mailbox,
// users B, C, ... already have their rounds queued, so neither the per-round sync nor the launch latency of the
// small-M forwards leaves the GPU idle, and forwards of different users overlap on the chip.  Results are
// identical to n sequential atspeed_bssd_generate calls.  `stream` is the caller's stream: the call begins after
// its pending work and ends synchronised with it.
extern "C" int atspeed_bssd_generate_batch(atspeed_decoder** decs, int32_t n, const int32_t* const* prompts,
                                           const int32_t* prompt_lens, const atspeed_fsm* fsm, const int32_t* start_nodes,
                                           int32_t gamma, int32_t max_new, int32_t k, int32_t dk, int32_t* const* out_tokens,
                                           float* const* out_scores, atspeed_gen_stats* stats, void* stream) {
  ATS_REQUIRE(decs && prompts && prompt_lens && start_nodes && out_tokens && out_scores && n >= 1, ATSPEED_ERR_INVALID,
              "bssd_batch: bad arguments");
  hipStream_t caller = (hipStream_t)stream;
  hipEvent_t ready;
  ATS_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
  ATS_HIP(hipEventRecord(ready, caller));
  int rc = ATSPEED_OK;
  for (int i = 0; i < n && rc == ATSPEED_OK; ++i) {
    for (int j = 0; j < i; ++j) ATS_REQUIRE(decs[i] != decs[j], ATSPEED_ERR_INVALID, "bssd_batch: decoder %d used twice", i);
    ATS_HIP(hipStreamWaitEvent(decs[i]->own_stream, ready, 0));
    rc = bssd_begin(decs[i], prompts[i], prompt_lens[i], fsm, start_nodes[i], gamma, max_new, k, dk, out_tokens[i], out_scores[i],
                    stats ? &stats[i] : nullptr, decs[i]->own_stream);
    if (rc == ATSPEED_OK) rc = bssd_enqueue_round(decs[i]);
  }
  int active = n;
  while (rc == ATSPEED_OK && active > 0) {
    active = 0;
    for (int i = 0; i < n && rc == ATSPEED_OK; ++i) {
      atspeed_decoder* d = decs[i];
      if (d->run.done) continue;
      rc = bssd_finish_round(d);
      if (rc == ATSPEED_OK && !d->run.done) { rc = bssd_enqueue_round(d); ++active; }
    }
  }
  for (int i = 0; i < n; ++i) hipStreamSynchronize(decs[i]->own_stream);
  hipEventDestroy(ready);
  if (rc == ATSPEED_OK) ATS_HIP(hipStreamSynchronize(caller));
  return rc;
}

extern "C" int atspeed_target_generate(atspeed_decoder* d, const int32_t* prompt, int32_t P, const atspeed_fsm* fsm,
                                       int32_t start_node, int32_t max_new, int32_t k, int32_t* out_tokens,
                                       float* out_scores, atspeed_gen_stats* stats, void* stream) {
  ATS_TRY(check_common(d, prompt, P, fsm, start_node, max_new, k, out_tokens, out_scores));
  hipStream_t st = (hipStream_t)stream;
  atspeed_llama* T = d->target;
  const int W = d->W, V = T->cfg.vocab_size;
  atspeed_gen_stats s;
  memset(&s, 0, sizeof(s));
  ATS_REQUIRE(P + max_new * k <= T->cfg.max_slots, ATSPEED_ERR_CAPACITY, "target_generate: KV slots exhausted");
  ATS_REQUIRE(P + max_new * k <= d->tok_cap && P <= T->cfg.max_tokens, ATSPEED_ERR_CAPACITY, "target_generate: token buffer too small");
  TokBuf& tin = d->tin[0];
  ATS_TRY(ats_init_prompt(tin, prompt, P, W, d->round_beams[0], start_node, V, d->mail_dev, st));
  hipEventRecord(d->ev[0], st);
  int cur = 0, row0 = 0, n_in = P, nb = 1, base = 0;
  for (int g = 0; g < max_new; ++g) {                                                // beamSD.py:579-588
    TokBuf b = tb_offset(tin, row0, W);
    ATS_TRY(llama_forward(T, d->tctx, b.ids, b.pos, b.slot, b.vis, n_in, base + n_in, nb, nullptr, st));
    s.n_target_forwards++;
    ATS_TRY(ats_lse_rows(d->tctx->logits, nb, V, T->logits_ld, d->lse, st));
    BeamStepArgs a{};
    a.src = d->round_beams[cur]; a.n_src = nb; a.gen_len = g;
    a.logits = d->tctx->logits; a.ld = T->logits_ld; a.lse = d->lse; a.fsm = fsm->dev; a.k = k;
    a.dst = d->round_beams[cur ^ 1]; a.emit = 1;
    a.in = tin; a.in_row0 = row0 + n_in - nb;
    a.out = tin; a.out_row0 = row0 + n_in; a.out_slot0 = base + n_in; a.vis_words = W;
    a.mail = d->mail_dev;
    ATS_TRY(ats_beam_step(a, st));
    row0 += n_in; base += n_in; n_in = k; nb = k; cur ^= 1;
  }
  ATS_TRY(ats_export_beams(d->round_beams[cur], k, max_new, out_tokens, out_scores, st));
  hipEventRecord(d->ev[1], st);
  ATS_HIP(hipMemcpyAsync(d->mail_host, d->mail_dev, sizeof(Mailbox), hipMemcpyDeviceToHost, st));
  ATS_TRY(read_mailbox(d, st));
  s.n_valid = d->mail_host->n_valid;
  hipEventElapsedTime(&s.total_ms, d->ev[0], d->ev[1]);
  s.target_ms = s.total_ms;
  if (stats) *stats = s;
  return ATSPEED_OK;
}

extern "C" int atspeed_decoder_trace(atspeed_decoder* d, int32_t* rounds_out, int32_t cap) {
  if (!d) return 0;
  int n = (int)std::min<size_t>(d->trace.size(), (size_t)std::max(cap, 0));
  if (rounds_out && n > 0) memcpy(rounds_out, d->trace.data(), sizeof(int32_t) * n);
  return (int)d->trace.size();
}
