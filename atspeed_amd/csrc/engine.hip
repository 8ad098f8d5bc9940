// Host-side engine: model handle (weights in HBM, KV arena, forward sequencing) and the
// decoder that runs BSSD / target_generate for one user stream with all state on the device.
// Reference: beamSD.py:458-542 (BSSD), :544-595 (target_generate), :108-179, :190-232, :242-456.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
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
// 0.2 (round 6): atspeed_gemm_fp8 / atspeed_gemm_fp8_packed take (workspace, bytes) before the stream (changed in round 5 without a bump);
// atspeed_set_switch / atspeed_get_switch; atspeed_llama_enable_fp8 on fp16 models
extern "C" const char* atspeed_version(void) { return "atspeed_hip 0.2 (gfx950)"; }

// ---- process-wide switches (internal.h: ATS_SW_*)
namespace {
struct SwDef { const char* name; const char* env; int dflt; };
const SwDef kSwitches[ATS_N_SW] = {{"gemm_sk", "ATSPEED_GEMM_SK", 1},           {"gemm_sk_g", nullptr, 0},
                                   {"gemm_panel", "ATSPEED_GEMM_PANEL", 1},     {"gemm_force_mt", "ATSPEED_GEMM_FORCE_MT", 0},
                                   {"graphs", "ATSPEED_GRAPHS", 0},             {"fuse_qkv_rope", "ATSPEED_FUSE_QKV_ROPE", 1},
                                   {"gemm_kcut", "ATSPEED_GEMM_KCUT", 2},       {"fuse_qkv_reduce", "ATSPEED_FUSE_QKV_REDUCE", 1}};
std::atomic<int> g_switch[ATS_N_SW];
std::once_flag g_switch_once;
void switches_init() {
  std::call_once(g_switch_once, [] {
    for (int i = 0; i < ATS_N_SW; ++i) {
      const char* v = kSwitches[i].env ? getenv(kSwitches[i].env) : nullptr;
      g_switch[i].store(v ? atoi(v) : kSwitches[i].dflt, std::memory_order_relaxed);
    }
  });
}
}  // namespace
int ats_switch(int id) { switches_init(); return g_switch[id].load(std::memory_order_relaxed); }
extern "C" int atspeed_set_switch(const char* name, int32_t value) {
  ATS_REQUIRE(name, ATSPEED_ERR_INVALID, "set_switch: null name");
  switches_init();
  for (int i = 0; i < ATS_N_SW; ++i)
    if (!strcmp(name, kSwitches[i].name)) { g_switch[i].store(value, std::memory_order_relaxed); return ATSPEED_OK; }
  atspeed_set_error("set_switch: unknown switch '%s'", name);
  return ATSPEED_ERR_INVALID;
}
extern "C" int atspeed_get_switch(const char* name, int32_t* value_out) {
  ATS_REQUIRE(name && value_out, ATSPEED_ERR_INVALID, "get_switch: null argument");
  switches_init();
  for (int i = 0; i < ATS_N_SW; ++i)
    if (!strcmp(name, kSwitches[i].name)) { *value_out = g_switch[i].load(std::memory_order_relaxed); return ATSPEED_OK; }
  atspeed_set_error("get_switch: unknown switch '%s'", name);
  return ATSPEED_ERR_INVALID;
}
std::atomic<long long> g_ats_path_cnt[ATS_N_PATHS];
extern "C" int atspeed_gemm_path_counters(int64_t* out, int32_t n, int32_t reset) {
  for (int i = 0; i < ATS_N_PATHS; ++i) {
    const long long v = reset ? g_ats_path_cnt[i].exchange(0) : g_ats_path_cnt[i].load();
    if (out && i < n) out[i] = v;
  }
  return ATS_N_PATHS;
}
extern "C" int atspeed_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ---------------------------------------------------------------------------- host->device staging ring
// Small host-built tables (segment tables, per-user argument blocks) travel through one pinned buffer and one
// device buffer; a slot is reused only after ats_stage_reset() (called after a stream synchronisation) or, if the
// ring fills up, after an explicit synchronisation of the stream.
namespace {
struct StageRing {
  char *host = nullptr, *dev = nullptr;
  size_t cap = 0, used = 0;
};
thread_local StageRing g_stage_dev[ATS_MAX_DEVICES];      // one ring per device (ats_cur_device)
constexpr size_t kStageBytes = 8u << 20;
}  // namespace

int ats_stage(const void* host_obj, size_t bytes, const void** dev_out, hipStream_t st) {
  const int dev = ats_cur_device();
  ATS_REQUIRE(dev >= 0, ATSPEED_ERR_NO_DEVICE, "staging: no current HIP device, or its id is >= %d", ATS_MAX_DEVICES);
  StageRing& r = g_stage_dev[dev];
  if (!r.host) {
    ATS_HIP(hipHostMalloc((void**)&r.host, kStageBytes));
    ATS_HIP(hipMalloc((void**)&r.dev, kStageBytes));
    r.cap = kStageBytes; r.used = 0;
  }
  size_t need = (bytes + 255) / 256 * 256;
  ATS_REQUIRE(need <= r.cap, ATSPEED_ERR_CAPACITY, "staging: object of %zu bytes too large", bytes);
  if (r.used + need > r.cap) { ATS_HIP(hipStreamSynchronize(st)); r.used = 0; }
  memcpy(r.host + r.used, host_obj, bytes);
  ATS_HIP(hipMemcpyAsync(r.dev + r.used, r.host + r.used, bytes, hipMemcpyHostToDevice, st));
  *dev_out = r.dev + r.used;
  r.used += need;
  return ATSPEED_OK;
}
// the same through the pinned ring, but to a device address of the caller's choice
int ats_stage_to(const void* host_obj, size_t bytes, void* dev_dst, hipStream_t st) {
  const int dev = ats_cur_device();
  ATS_REQUIRE(dev >= 0, ATSPEED_ERR_NO_DEVICE, "staging: no current HIP device, or its id is >= %d", ATS_MAX_DEVICES);
  StageRing& r = g_stage_dev[dev];
  if (!r.host) {
    ATS_HIP(hipHostMalloc((void**)&r.host, kStageBytes));
    ATS_HIP(hipMalloc((void**)&r.dev, kStageBytes));
    r.cap = kStageBytes; r.used = 0;
  }
  size_t need = (bytes + 255) / 256 * 256;
  ATS_REQUIRE(need <= r.cap, ATSPEED_ERR_CAPACITY, "staging: object of %zu bytes too large", bytes);
  if (r.used + need > r.cap) { ATS_HIP(hipStreamSynchronize(st)); r.used = 0; }
  memcpy(r.host + r.used, host_obj, bytes);
  ATS_HIP(hipMemcpyAsync(dev_dst, r.host + r.used, bytes, hipMemcpyHostToDevice, st));
  r.used += need;
  return ATSPEED_OK;
}
void ats_stage_reset() { const int dev = ats_cur_device(); if (dev >= 0) g_stage_dev[dev].used = 0; }

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
  f->dev = FsmDev{f->d_row_ptr, f->d_tok, f->d_nxt, n_nodes, n_edges, vocab_size, 32000, 2};     // id filter of beamSD.py:80-86
  // which 256-column tiles of a logit row a step can ever read: the fused lm_head epilogue stores only those (gemm.hip EPI_F32_LSE)
  // one byte per 256-column tile of THIS vocabulary (the epilogue indexes it with every tile of the lm_head's grid)
  std::vector<unsigned char> tiles((size_t)(vocab_size + 255) / 256, 0);
  for (int e = 0; e < n_edges; ++e) tiles[tok[e] / 256] = 1;
  ATS_HIP(hipMalloc((void**)&f->d_tile_store, tiles.size()));
  ATS_HIP(hipMemcpy(f->d_tile_store, tiles.data(), tiles.size(), hipMemcpyHostToDevice));
  *out = f;
  return ATSPEED_OK;
}

// "no mask": BSSD(..., prefix_allowed_tokens_fn=None) with an empty processor list (beamSD.py:469-478 builds no processor, :60-64 is
// the identity, :80-86 is skipped): every token of the vocabulary is a candidate of every beam
extern "C" int atspeed_fsm_create_free(int32_t vocab_size, atspeed_fsm** out) {
  ATS_REQUIRE(out && vocab_size > 0, ATSPEED_ERR_INVALID, "fsm_create_free: bad arguments");
  ATS_REQUIRE((int64_t)ATSPEED_MAX_BEAMS * vocab_size < (int64_t)0x7fffffff, ATSPEED_ERR_CAPACITY, "fsm_create_free: vocab too large for 32-bit flat ids");
  atspeed_fsm* f = new atspeed_fsm();
  memset(f, 0, sizeof(*f));
  f->dev = FsmDev{nullptr, nullptr, nullptr, 0, 0, vocab_size, 0, -1};
  std::vector<unsigned char> tiles((size_t)(vocab_size + 255) / 256, 1);      // every logit tile can be read
  ATS_HIP(hipMalloc((void**)&f->d_tile_store, tiles.size()));
  ATS_HIP(hipMemcpy(f->d_tile_store, tiles.data(), tiles.size(), hipMemcpyHostToDevice));
  *out = f;
  return ATSPEED_OK;
}

extern "C" int atspeed_fsm_set_id_filter(atspeed_fsm* f, int32_t min_item_token, int32_t eos_token) {
  ATS_REQUIRE(f, ATSPEED_ERR_INVALID, "fsm_set_id_filter: null automaton");
  ATS_REQUIRE(f->dev.n_nodes > 0, ATSPEED_ERR_INVALID, "fsm_set_id_filter: a mask-free automaton has no id filter (beamSD.py:80 applies it only with a processor)");
  f->dev.filter_min = min_item_token;
  f->dev.filter_eos = eos_token;
  return ATSPEED_OK;
}

extern "C" void atspeed_fsm_destroy(atspeed_fsm* f) {
  if (!f) return;
  hipFree(f->d_row_ptr); hipFree(f->d_tok); hipFree(f->d_nxt); hipFree(f->d_tile_store);
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
// Weights / config / RoPE tables are shared and read-only.  What a forward WRITES is split in two:
//   KvCache — per user (decoder): the slot-addressed K/V arena;
//   ActCtx  — per forward batch: activations, logits, split-K slabs, sized for the tokens of ALL users of the batch.
struct KvCache { void* k = nullptr; void* v = nullptr; };

struct ActCtx {
  int cap_tok = 0, cap_rows = 0;
  void *h = nullptr, *xn = nullptr, *qkv = nullptr, *att = nullptr, *act = nullptr, *gath = nullptr;
  float* logits = nullptr;                       // [cap_rows][logits_ld]
  float* lse = nullptr;                          // [cap_rows]
  float* lse_part = nullptr; size_t lse_part_bytes = 0;   // [cap_rows][vocab tiles] (max, sum exp) partials of the fused lm_head epilogue
  int32_t* row_cand = nullptr;                   // [cap_rows][ATSPEED_MAX_BEAMS] best tokens per logit row (mask-free search only)
  RowInfo* rowinfo = nullptr;                    // [cap_tok] cache / slot / rotation of each batched row (qkv projection's fused epilogue)
  void* xq = nullptr; float* sx = nullptr;       // fp8 activations [cap_tok][max(hidden, ffn)] + per-token scales
  void* ws = nullptr; size_t ws_bytes = 0;       // split-K slabs
  SkArena sk;                                    // the ring kernel's split-K tail (16-bit models: internal.h SkArena): allocated in front of the first forward of
  bool sk_tried = false;                         // >= 257 tokens that is not being captured (sk_arena_lazy), never inside the launch sequence
  // forwards of a recurring shape are replayed as hipGraphs (one launch instead of ~9 per layer: a user's later rounds are
  // 20-140 tokens and launch-bound); the segment table lives at a fixed device address so that it is data, not a kernel argument
  SegTable* segtab_dev = nullptr;
  hipStream_t cap_stream = nullptr;
  typedef std::tuple<int, int, int, int, int, int> GraphKey;      // tokens, logit rows, segments, query tiles, tile rows, fp8
  std::map<GraphKey, hipGraphExec_t> graphs;
  std::map<GraphKey, int> graph_seen;
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
  int pk;                                        // 1: weights and GEMM-operand activations in the packed operand layout (common.h), bf16 only
  size_t layer_kv_bytes;
  float *cos_tab, *sin_tab;                      // [max_slots][head_dim/2]
  // optional fp8 (e4m3, per-output-row scales) copies of the layer projections, library-owned (atspeed_llama_enable_fp8)
  struct Fp8Layer { void *wqkv, *wo, *wgu, *wd; float *sqkv, *so, *sgu, *sd; };
  std::vector<Fp8Layer> fp8;
  KvCache kv0;                                   // cache of the plain atspeed_llama_forward API
  std::vector<KvCache> kv_pool;                  // one more cache per segment of atspeed_llama_forward_batch (grown on demand)
  ActCtx* act;                                   // grown on demand (ensure_act)
  bool prof_on = false;
  double prof_ms[5] = {0, 0, 0, 0, 0};
  long prof_cnt[5] = {0, 0, 0, 0, 0};
  long prof_rows[5] = {0, 0, 0, 0, 0};           // sum of M over the bracketed launches
  // the same, restricted to launches of at least ATS_PROF_BIG_ROWS tokens (these take the 256x256 ring kernel)
  double prof_big_ms[5] = {0, 0, 0, 0, 0};
  long prof_big_cnt[5] = {0, 0, 0, 0, 0};
  long prof_big_rows[5] = {0, 0, 0, 0, 0};
  // how often each layer projection (0 qkv, 1 o_proj, 2 gate_up, 3 down) ran as an fp8 / as a bf16 (fp32) GEMM (atspeed_llama_fp8_counters)
  long fp8_cnt[4] = {0, 0, 0, 0}, other_cnt[4] = {0, 0, 0, 0};
  long rope_fused_cnt = 0;                       // qkv projections that carried RoPE + the KV scatter in their epilogue (atspeed_llama_rope_fused_launches)
  bool fwd_log_on = false;                       // (tokens, logit rows) of every forward while on (atspeed_llama_forward_log)
  std::vector<int32_t> fwd_log;
};
constexpr int ATS_PROF_BIG_ROWS = 1024;

static void prof_harvest(atspeed_llama* m) {
  ActCtx* cx = m->act;
  if (!cx || !cx->prof_pending) return;
  for (size_t b = 0; b < cx->prof_kind.size(); ++b) {
    float ms = 0.f;
    if (hipEventSynchronize(cx->prof_ev[2 * b + 1]) == hipSuccess &&
        hipEventElapsedTime(&ms, cx->prof_ev[2 * b], cx->prof_ev[2 * b + 1]) == hipSuccess) {
      int kd = cx->prof_kind[b];
      m->prof_ms[kd] += ms; m->prof_cnt[kd] += 1; m->prof_rows[kd] += cx->prof_m[b];
      if (cx->prof_m[b] >= ATS_PROF_BIG_ROWS) { m->prof_big_ms[kd] += ms; m->prof_big_cnt[kd] += 1; m->prof_big_rows[kd] += cx->prof_m[b]; }
    }
  }
  cx->prof_kind.clear(); cx->prof_m.clear();
  cx->prof_pending = false;
}

struct ProfBracket {
  ActCtx* cx; hipStream_t st; bool on; size_t idx;
  ProfBracket(atspeed_llama* m, int kind, int rows, hipStream_t st_) : cx(m->act), st(st_), on(m->prof_on), idx(0) {
    if (!on) return;
    idx = cx->prof_kind.size();
    while (cx->prof_ev.size() < 2 * (idx + 1)) { hipEvent_t e; hipEventCreate(&e); cx->prof_ev.push_back(e); }
    cx->prof_kind.push_back(kind); cx->prof_m.push_back(rows);
    hipEventRecord(cx->prof_ev[2 * idx], st);
  }
  ~ProfBracket() { if (on) { hipEventRecord(cx->prof_ev[2 * idx + 1], st); cx->prof_pending = true; } }
};

static size_t gemm_ws_for(const atspeed_llama_config& c, int max_tok, int max_rows) {
  size_t best = 0;
  for (int m = 1; m <= max_tok; ++m) {          // the split-K plan depends on the exact M: take the true maximum
    best = std::max(best, ATS_KD(c.dtype, ats_gemm_workspace_bytes(m, 3 * c.hidden, c.hidden, c.dtype)));
    best = std::max(best, ATS_KD(c.dtype, ats_gemm_workspace_bytes(m, c.hidden, c.hidden, c.dtype)));
    best = std::max(best, ATS_KD(c.dtype, ats_gemm_workspace_bytes(m, 2 * c.ffn, c.hidden, c.dtype)));
    best = std::max(best, ATS_KD(c.dtype, ats_gemm_workspace_bytes(m, c.hidden, c.ffn, c.dtype)));
    if (m <= max_rows) best = std::max(best, ATS_KD(c.dtype, ats_gemm_workspace_bytes(m, c.vocab_size, c.hidden, c.dtype)));
    if (c.dtype != ATSPEED_F32) {                  // the W8A8 copies (atspeed_llama_enable_fp8): the split plans of one user's projections and of thin ring grids
      best = std::max(best, ATS_KD(c.dtype, ats_gemm_fp8_workspace_bytes(m, 3 * c.hidden, c.hidden)));
      best = std::max(best, ATS_KD(c.dtype, ats_gemm_fp8_workspace_bytes(m, c.hidden, c.hidden)));
      best = std::max(best, ATS_KD(c.dtype, ats_gemm_fp8_workspace_bytes(m, c.hidden, c.ffn)));
    }
  }
  return best + (1 << 20);
}

static void act_free(ActCtx* cx) {
  if (!cx) return;
  hipFree(cx->h); hipFree(cx->xn); hipFree(cx->qkv); hipFree(cx->att); hipFree(cx->act); hipFree(cx->gath);
  hipFree(cx->logits); hipFree(cx->lse); hipFree(cx->lse_part); hipFree(cx->ws); hipFree(cx->xq); hipFree(cx->sx); hipFree(cx->rowinfo);
  hipFree(cx->row_cand); hipFree(cx->sk.ws); hipFree(cx->sk.cnt);
  for (hipEvent_t e : cx->prof_ev) hipEventDestroy(e);
  for (auto& g : cx->graphs) hipGraphExecDestroy(g.second);
  if (cx->cap_stream) hipStreamDestroy(cx->cap_stream);
  hipFree(cx->segtab_dev);
  delete cx;
}

// make sure the model's activation context can hold `tok` tokens and `rows` logit rows (device must be idle on it)
static int ensure_act(atspeed_llama* m, int tok, int rows) {
  if (m->act && m->act->cap_tok >= tok && m->act->cap_rows >= rows) return ATSPEED_OK;
  if (m->act) { prof_harvest(m); ATS_HIP(hipDeviceSynchronize()); act_free(m->act); m->act = nullptr; }
  const atspeed_llama_config& c = m->cfg;
  ActCtx* cx = new ActCtx();
  // 25 % headroom: the next batch's token count differs by a few prompt tokens, and growing costs a device synchronisation plus
  // the re-allocation of every activation buffer (seen as 30-40 ms hiccups inside timed regions)
  // even counts: the packed operand layout stores row pairs, a last row with an even index still owns a full 128-byte line pair
  cx->cap_tok = (std::max((tok + tok / 4 + 255) / 256 * 256, c.max_tokens) + 1) & ~1;
  cx->cap_rows = (std::max((rows + rows / 4 + 63) / 64 * 64, c.max_logit_rows) + 1) & ~1;
  size_t T = cx->cap_tok, H = c.hidden, e = m->esz;
  ATS_HIP(hipMalloc(&cx->h, T * H * e));
  ATS_HIP(hipMalloc(&cx->xn, T * H * e));
  ATS_HIP(hipMalloc(&cx->qkv, T * 3 * H * e));
  ATS_HIP(hipMalloc(&cx->att, T * H * e));
  ATS_HIP(hipMalloc(&cx->act, T * (size_t)c.ffn * e));
  ATS_HIP(hipMalloc(&cx->gath, (size_t)cx->cap_rows * H * e));
  ATS_HIP(hipMalloc((void**)&cx->logits, (size_t)cx->cap_rows * m->logits_ld * sizeof(float)));
  ATS_HIP(hipMalloc((void**)&cx->lse, (size_t)cx->cap_rows * sizeof(float)));
  ATS_HIP(hipMalloc((void**)&cx->row_cand, (size_t)cx->cap_rows * ATSPEED_MAX_BEAMS * sizeof(int32_t)));
  cx->lse_part_bytes = ats_bf16::ats_lmhead_lse_part_bytes(cx->cap_rows, c.vocab_size);
  ATS_HIP(hipMalloc((void**)&cx->lse_part, cx->lse_part_bytes));
  ATS_HIP(hipMalloc(&cx->xq, T * (size_t)std::max(c.hidden, c.ffn)));
  ATS_HIP(hipMalloc((void**)&cx->sx, T * sizeof(float)));
  ATS_HIP(hipMalloc((void**)&cx->rowinfo, T * sizeof(RowInfo)));
  cx->ws_bytes = gemm_ws_for(c, cx->cap_tok, cx->cap_rows);
  ATS_HIP(hipMalloc(&cx->ws, cx->ws_bytes));
  ATS_HIP(hipMalloc((void**)&cx->segtab_dev, sizeof(SegTable)));
  m->act = cx;
  return ATSPEED_OK;
}

// The split-K tail's arena (128 MB + counters) belongs to the model, but only a model that really runs forwards of >= 257 tokens gets one
// (every model used to own one from creation -- the 68M draft and every rank's one-user target included, whose forwards never reach
// the ring kernel).  Allocated HERE, in front of the launch sequence of the first such forward, never inside it; a forward whose stream is
// being captured by the caller allocates nothing (its thin grids run the plain kernel, as before).  Optional: without room for it the model
// still runs (thin / partly filled grids then take the device's shared arena or the plain grid).  atspeed_llama_sk_arena_bytes reports it.
static void sk_arena_lazy(atspeed_llama* m, int T, hipStream_t st) {
  ActCtx* cx = m->act;
  const atspeed_llama_config& c = m->cfg;
  if (cx->sk.ws || cx->sk_tried || T < 257 || c.dtype == ATSPEED_F32 || c.hidden % 128 != 0) return;   // shapes the ring kernel can take at all (big_kernel_applies)
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return; }
  if (cs != hipStreamCaptureStatusNone) return;
  cx->sk_tried = true;
  if (hipMalloc((void**)&cx->sk.ws, ATS_SK_ARENA_BYTES) != hipSuccess || hipMalloc((void**)&cx->sk.cnt, ATS_SK_ARENA_COUNTERS * sizeof(int)) != hipSuccess ||
      hipMemsetAsync(cx->sk.cnt, 0, ATS_SK_ARENA_COUNTERS * sizeof(int), st) != hipSuccess) {
    (void)hipGetLastError();
    hipFree(cx->sk.ws); hipFree(cx->sk.cnt);
    cx->sk = SkArena{};
  }
}
extern "C" int64_t atspeed_llama_sk_arena_bytes(const atspeed_llama* m) { return m && m->act && m->act->sk.ws ? (int64_t)ATS_SK_ARENA_BYTES : 0; }

static int kv_create(atspeed_llama* m, KvCache* kv) {
  size_t bytes = (size_t)m->cfg.n_layers * m->layer_kv_bytes;
  ATS_HIP(hipMalloc(&kv->k, bytes));
  ATS_HIP(hipMalloc(&kv->v, bytes));
  ATS_HIP(hipMemset(kv->k, 0, bytes));
  ATS_HIP(hipMemset(kv->v, 0, bytes));
  return ATSPEED_OK;
}
static void kv_free(KvCache* kv) { hipFree(kv->k); hipFree(kv->v); kv->k = kv->v = nullptr; }

extern "C" int atspeed_llama_create(const atspeed_llama_config* cfg, const void* embed, const void* final_norm,
                                    const void* lm_head, const atspeed_llama_layer_weights* layers, atspeed_llama** out) {
  ATS_REQUIRE(cfg && embed && final_norm && lm_head && layers && out, ATSPEED_ERR_INVALID, "llama_create: null argument");
  ATS_REQUIRE(cfg->dtype == ATSPEED_F32 || cfg->dtype == ATSPEED_BF16 || cfg->dtype == ATSPEED_F16, ATSPEED_ERR_INVALID, "llama_create: bad dtype");
  ATS_REQUIRE(cfg->n_heads > 0 && cfg->hidden % cfg->n_heads == 0, ATSPEED_ERR_INVALID, "llama_create: hidden %% n_heads != 0");
  int hd = cfg->hidden / cfg->n_heads;
  ATS_REQUIRE(hd % 8 == 0 && hd <= 256, ATSPEED_ERR_INVALID, "llama_create: head_dim %d unsupported", hd);
  ATS_REQUIRE(cfg->hidden % 8 == 0 && cfg->ffn % 16 == 0, ATSPEED_ERR_INVALID, "llama_create: hidden %% 8 / ffn %% 16 required");
  ATS_REQUIRE(cfg->max_slots > 0 && cfg->max_slots % 64 == 0 && cfg->max_slots <= 2048, ATSPEED_ERR_INVALID,
              "llama_create: max_slots must be a multiple of 64, <= 2048");
  ATS_REQUIRE(cfg->max_tokens > 0 && cfg->max_logit_rows > 0 && cfg->max_logit_rows <= cfg->max_tokens, ATSPEED_ERR_INVALID,
              "llama_create: bad token limits");
  ATS_REQUIRE(cfg->weight_layout == ATSPEED_WEIGHTS_ROW_MAJOR || cfg->weight_layout == ATSPEED_WEIGHTS_PACKED, ATSPEED_ERR_INVALID,
              "llama_create: weight_layout must be ATSPEED_WEIGHTS_ROW_MAJOR (0) or ATSPEED_WEIGHTS_PACKED (1)");
  atspeed_llama* m = new atspeed_llama();
  m->cfg = *cfg;
  m->embed = embed; m->final_norm = final_norm; m->lm_head = lm_head;
  m->layers.assign(layers, layers + cfg->n_layers);
  m->esz = cfg->dtype == ATSPEED_F32 ? 4 : 2;
  m->head_dim = hd;
  m->vis_words = cfg->max_slots / 64;
  m->logits_ld = (cfg->vocab_size + 63) / 64 * 64;
  m->layer_kv_bytes = (size_t)cfg->max_slots * cfg->hidden * m->esz;
  m->pk = cfg->weight_layout == ATSPEED_WEIGHTS_PACKED ? 1 : 0;
  if (m->pk && !((cfg->dtype == ATSPEED_BF16 || cfg->dtype == ATSPEED_F16) && cfg->hidden % 32 == 0 && cfg->ffn % 32 == 0)) {
    atspeed_set_error("llama_create: packed weights need bf16 / fp16 and hidden / ffn multiples of 32 (hidden %d, ffn %d)", cfg->hidden, cfg->ffn);
    delete m;
    return ATSPEED_ERR_INVALID;
  }
  m->act = nullptr;
  ATS_TRY(kv_create(m, &m->kv0));
  ATS_TRY(ensure_act(m, cfg->max_tokens, cfg->max_logit_rows));
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
  kv_free(&m->kv0);
  for (KvCache& kv : m->kv_pool) kv_free(&kv);
  act_free(m->act);
  for (auto& f : m->fp8) { hipFree(f.wqkv); hipFree(f.wo); hipFree(f.wgu); hipFree(f.wd); hipFree(f.sqkv); hipFree(f.so); hipFree(f.sgu); hipFree(f.sd); }
  hipFree(m->cos_tab); hipFree(m->sin_tab);
  delete m;
}

extern "C" int atspeed_llama_profile(atspeed_llama* m, int32_t enable, double* ms_out, int64_t* count_out, int64_t* rows_out) {
  ATS_REQUIRE(m, ATSPEED_ERR_INVALID, "profile: null model");
  prof_harvest(m);
  for (int i = 0; i < 5; ++i) {
    if (ms_out) ms_out[i] = m->prof_ms[i];
    if (count_out) count_out[i] = m->prof_cnt[i];
    if (rows_out) rows_out[i] = m->prof_rows[i];
  }
  if (enable >= 0) {
    m->prof_on = enable != 0;
    for (int i = 0; i < 5; ++i) {
      m->prof_ms[i] = 0; m->prof_cnt[i] = 0; m->prof_rows[i] = 0;
      m->prof_big_ms[i] = 0; m->prof_big_cnt[i] = 0; m->prof_big_rows[i] = 0;
    }
  }
  return ATSPEED_OK;
}

extern "C" int32_t atspeed_llama_forward_log(atspeed_llama* m, int32_t enable, int32_t* pairs_out, int32_t max_pairs) {
  if (!m) return -1;
  const int32_t n = (int32_t)(m->fwd_log.size() / 2);
  if (pairs_out) for (int32_t i = 0; i < n && i < max_pairs; ++i) { pairs_out[2 * i] = m->fwd_log[2 * i]; pairs_out[2 * i + 1] = m->fwd_log[2 * i + 1]; }
  if (enable >= 0) { m->fwd_log_on = enable != 0; if (enable) m->fwd_log.clear(); }
  return n;
}

extern "C" int atspeed_llama_profile_big(atspeed_llama* m, double* ms_out, int64_t* count_out, int64_t* rows_out) {
  ATS_REQUIRE(m && ms_out && count_out && rows_out, ATSPEED_ERR_INVALID, "profile_big: null argument");
  prof_harvest(m);
  for (int i = 0; i < 5; ++i) { ms_out[i] = m->prof_big_ms[i]; count_out[i] = m->prof_big_cnt[i]; rows_out[i] = m->prof_big_rows[i]; }
  return ATSPEED_OK;
}

extern "C" int atspeed_lmhead_lse(const void* x, const void* w, float* logits, float* lse, int32_t rows, int32_t vocab, int32_t hidden, int32_t ld,
                                  const atspeed_fsm* fsm, void* workspace, size_t workspace_bytes, int32_t* fused_out, void* stream) {
  ATS_REQUIRE(x && w && logits && lse && rows >= 0 && vocab > 0 && hidden > 0 && ld >= vocab, ATSPEED_ERR_INVALID, "lmhead_lse: bad arguments");
  const size_t pb = ats_bf16::ats_lmhead_lse_part_bytes(rows, vocab);
  const bool have = workspace && workspace_bytes >= pb && ((uintptr_t)workspace & 15) == 0;
  char* rest = have ? (char*)workspace + (pb + 255) / 256 * 256 : (char*)workspace;
  const size_t rest_bytes = have ? (workspace_bytes > (pb + 255) / 256 * 256 ? workspace_bytes - (pb + 255) / 256 * 256 : 0) : workspace_bytes;
  int fused = 0;
  const int rc = ats_bf16::ats_lmhead_lse(x, w, logits, rows, vocab, hidden, hidden, ld, ATSPEED_BF16, fsm ? fsm->d_tile_store : nullptr, have ? (float*)workspace : nullptr,
                                have ? pb : 0, lse, rest, rest_bytes, (hipStream_t)stream, &fused);
  if (fused_out) *fused_out = fused;
  return rc;
}

extern "C" int atspeed_llama_fp8_counters(atspeed_llama* m, int64_t* fp8_out, int64_t* other_out, int32_t reset) {
  ATS_REQUIRE(m, ATSPEED_ERR_INVALID, "fp8_counters: null model");
  for (int i = 0; i < 4; ++i) {
    if (fp8_out) fp8_out[i] = m->fp8_cnt[i];
    if (other_out) other_out[i] = m->other_cnt[i];
    if (reset) { m->fp8_cnt[i] = 0; m->other_cnt[i] = 0; }
  }
  return ATSPEED_OK;
}

extern "C" int64_t atspeed_llama_rope_fused_launches(atspeed_llama* m, int32_t reset) {
  if (!m) return -1;
  const int64_t n = m->rope_fused_cnt;
  if (reset) m->rope_fused_cnt = 0;
  return n;
}

extern "C" float* atspeed_llama_logits(atspeed_llama* m) { return m && m->act ? m->act->logits : nullptr; }
extern "C" int32_t atspeed_llama_logits_ld(const atspeed_llama* m) { return m ? m->logits_ld : 0; }

extern "C" int atspeed_llama_enable_fp8(atspeed_llama* m, void* stream) {
  ATS_REQUIRE(m, ATSPEED_ERR_INVALID, "enable_fp8: null model");
  // bf16 or fp16 weights (round 6: the reference loads fp16 checkpoints and runs the target 8-bit, code/inference.py:75-91): the e4m3 copies are
  // made from the model's own 16-bit values by the quantisation kernel of its flavour, the activations between the W8A8 projections stay in that type
  ATS_REQUIRE(m->cfg.dtype == ATSPEED_BF16 || m->cfg.dtype == ATSPEED_F16, ATSPEED_ERR_INVALID, "enable_fp8: the model must hold bf16 or fp16 weights");
  ATS_REQUIRE(m->cfg.hidden % 256 == 0 && m->cfg.ffn % 256 == 0, ATSPEED_ERR_INVALID, "enable_fp8: hidden and ffn must be multiples of 256");
  if (!m->fp8.empty()) return ATSPEED_OK;
  hipStream_t st = (hipStream_t)stream;
  const int H = m->cfg.hidden, F = m->cfg.ffn;
  auto quant = [&](const void* w, int rows, int cols, void** q, float** sc) -> int {     // packed 16-bit rows -> packed e4m3 rows (or row-major both)
    ATS_HIP(hipMalloc(q, (size_t)((rows + 1) & ~1) * cols));
    ATS_HIP(hipMalloc((void**)sc, (size_t)rows * sizeof(float)));
    return ATS_KD(m->cfg.dtype, ats_quant_rows_fp8(w, rows, cols, cols, *q, *sc, st, m->pk));
  };
  m->fp8.resize(m->cfg.n_layers);
  for (int l = 0; l < m->cfg.n_layers; ++l) {
    const atspeed_llama_layer_weights& w = m->layers[l];
    atspeed_llama::Fp8Layer& f = m->fp8[l];
    ATS_TRY(quant(w.wqkv, 3 * H, H, &f.wqkv, &f.sqkv));
    ATS_TRY(quant(w.wo, H, H, &f.wo, &f.so));
    ATS_TRY(quant(w.wgu, 2 * F, H, &f.wgu, &f.sgu));
    ATS_TRY(quant(w.wd, H, F, &f.wd, &f.sd));
  }
  ATS_HIP(hipStreamSynchronize(st));
  return ATSPEED_OK;
}

// One forward over the tokens of every segment (user) of the table.  Logits of each segment's last n_logit rows land
// in act->logits rows [logit_row0, ..) (or in logits_out), their log-sum-exp in act->lse.
static int llama_forward_body(atspeed_llama* m, const SegTable& t, const SegTable* dtab, float* logits_out, hipStream_t st,
                              const unsigned char* tile_store);

// tile_store (device bytes, one per 256 vocabulary columns; NULL = every column): which logit tiles the caller will read.  The decoder
// passes its automaton's (atspeed_fsm::d_tile_store): rows then get their full-vocabulary normaliser from the lm_head epilogue and
// only those tiles are written to the logits buffer.
static int llama_forward_segs(atspeed_llama* m, const SegTable& t, float* logits_out, hipStream_t st, const unsigned char* tile_store = nullptr) {
  const atspeed_llama_config& c = m->cfg;
  ActCtx* cx = m->act;
  const int T = t.total_tok;
  ATS_REQUIRE(T >= 1 && T <= cx->cap_tok, ATSPEED_ERR_CAPACITY, "forward: %d tokens exceed the activation capacity %d", T, cx->cap_tok);
  ATS_REQUIRE(t.total_logit >= 0 && t.total_logit <= cx->cap_rows, ATSPEED_ERR_CAPACITY, "forward: %d logit rows exceed the capacity %d",
              t.total_logit, cx->cap_rows);
  for (int i = 0; i < t.n; ++i) {
    ATS_REQUIRE(t.seg[i].n_tok >= 1 && t.seg[i].n_slots >= 1 && t.seg[i].n_slots <= c.max_slots, ATSPEED_ERR_CAPACITY,
                "forward: segment %d has %d tokens / %d slots (max_slots %d)", i, t.seg[i].n_tok, t.seg[i].n_slots, c.max_slots);
    ATS_REQUIRE(t.seg[i].n_logit >= 0 && t.seg[i].n_logit <= t.seg[i].n_tok, ATSPEED_ERR_INVALID, "forward: bad logit row count");
  }
  if (m->prof_on) prof_harvest(m);
  if (m->fwd_log_on && m->fwd_log.size() < 2 * 4096) { m->fwd_log.push_back(T); m->fwd_log.push_back(t.total_logit); }
  sk_arena_lazy(m, T, st);
  // the table travels through the pinned ring to its fixed device address (stream ordered behind the previous forward)
  ATS_TRY(ats_stage_to(&t, sizeof(t), cx->segtab_dev, st));
  const SegTable* dtab = cx->segtab_dev;
  // opt-in (ATSPEED_GRAPHS=1): measured on MI355X, replaying a 100-token forward as one hipGraph does not shorten it (676 vs 679
  // items/s in the one-user-at-a-time loop) -- the ~10 us between dependent kernels is the GPU's own barrier / cache-flush latency,
  // not host launch cost, and a graph keeps every node boundary
  const int use_graphs = ats_switch(ATS_SW_GRAPHS);                // (atspeed_set_switch("graphs", 1): the tests compare both modes in one process)
  constexpr int graph_max_tok = 512;
  if (use_graphs && !m->prof_on && logits_out == nullptr && T <= graph_max_tok) {
    const ActCtx::GraphKey key(T, t.total_logit, t.n, t.n_qtiles, t.qtile_rows, m->fp8.empty() ? 0 : 1);
    auto it = cx->graphs.find(key);
    if (it == cx->graphs.end() && cx->graphs.size() < 64 && ++cx->graph_seen[key] >= 2) {   // a shape seen twice recurs (K + dl*DK tokens)
      if (!cx->cap_stream) ATS_HIP(hipStreamCreateWithFlags(&cx->cap_stream, hipStreamNonBlocking));
      ATS_HIP(hipStreamBeginCapture(cx->cap_stream, hipStreamCaptureModeThreadLocal));
      const int rc = llama_forward_body(m, t, dtab, nullptr, cx->cap_stream, nullptr);
      hipGraph_t g = nullptr;
      hipError_t e = hipStreamEndCapture(cx->cap_stream, &g);
      if (rc != ATSPEED_OK) { if (g) hipGraphDestroy(g); return rc; }
      ATS_HIP(e);
      hipGraphExec_t ex = nullptr;
      e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
      hipGraphDestroy(g);
      ATS_HIP(e);
      it = cx->graphs.emplace(key, ex).first;
    }
    if (it != cx->graphs.end()) {
      ATS_HIP(hipGraphLaunch(it->second, st));
      return ATSPEED_OK;
    }
  }
  return llama_forward_body(m, t, dtab, logits_out, st, logits_out ? nullptr : tile_store);
}

namespace fwd_bf16 {
using namespace ats_bf16;
#include "engine_forward.inc"
}
namespace fwd_f16 {
using namespace ats_f16;
#include "engine_forward.inc"
}
static int llama_forward_body(atspeed_llama* m, const SegTable& t, const SegTable* dtab, float* logits_out, hipStream_t st,
                              const unsigned char* tile_store) {
  return m->cfg.dtype == ATSPEED_F16 ? fwd_f16::llama_forward_body(m, t, dtab, logits_out, st, tile_store)
                                     : fwd_bf16::llama_forward_body(m, t, dtab, logits_out, st, tile_store);
}

extern "C" int atspeed_llama_forward(atspeed_llama* m, const int32_t* ids, const int32_t* pos, const int32_t* slots,
                                     const uint64_t* vis, int32_t n_tokens, int32_t n_slots_visible, int32_t n_logit_rows,
                                     float* logits_out, void* stream) {
  ATS_REQUIRE(m && ids && pos && slots && vis, ATSPEED_ERR_INVALID, "forward: null argument");
  ATS_REQUIRE(n_tokens >= 1 && n_tokens <= m->cfg.max_tokens, ATSPEED_ERR_CAPACITY, "forward: %d tokens exceed max_tokens %d", n_tokens, m->cfg.max_tokens);
  ATS_REQUIRE(n_logit_rows >= 0 && n_logit_rows <= n_tokens && n_logit_rows <= m->cfg.max_logit_rows, ATSPEED_ERR_CAPACITY,
              "forward: %d logit rows exceed the limit %d", n_logit_rows, m->cfg.max_logit_rows);
  SegTable t{};
  t.n = 1; t.total_tok = n_tokens; t.total_logit = n_logit_rows;
  Seg& s = t.seg[0];
  s.ids = ids; s.pos = pos; s.slot = slots; s.vis = vis; s.kc = m->kv0.k; s.vc = m->kv0.v;
  s.row0 = 0; s.n_tok = n_tokens; s.n_slots = n_slots_visible; s.logit_row0 = 0; s.n_logit = n_logit_rows;
  t.n_qtiles = 0; t.qtile_rows = n_tokens > 96 ? 128 : 64;
  for (int j = 0; j * t.qtile_rows < n_tokens; ++j) { t.qtile_seg[t.n_qtiles] = 0; t.qtile_idx[t.n_qtiles++] = (unsigned char)j; }
  return llama_forward_segs(m, t, logits_out, (hipStream_t)stream);
}

static int seg_finish(SegTable& t);

// n independent forwards as ONE batched forward (segment table): sequence i has its own token / position / slot / visibility
// arrays and a KV arena of its own from the model's pool; logits of its last n_logit[i] rows follow each other in logits_out.
// What the teacher-data job needs to score the label and the K beams of many samples at once (generate_teacher_data.py:225-232).
extern "C" int atspeed_llama_forward_batch(atspeed_llama* m, int32_t n, const int32_t* const* ids, const int32_t* const* pos,
                                           const int32_t* const* slots, const uint64_t* const* vis, const int32_t* n_tokens,
                                           const int32_t* n_slots_visible, const int32_t* n_logit_rows, float* logits_out,
                                           void* stream) {
  ATS_REQUIRE(m && ids && pos && slots && vis && n_tokens && n_slots_visible && n_logit_rows && logits_out, ATSPEED_ERR_INVALID,
              "forward_batch: null argument");
  ATS_REQUIRE(n >= 1 && n <= ATS_MAX_SEGS, ATSPEED_ERR_CAPACITY, "forward_batch: %d sequences per call (max %d)", n, ATS_MAX_SEGS);
  while ((int)m->kv_pool.size() < n) {
    KvCache kv;
    ATS_TRY(kv_create(m, &kv));
    m->kv_pool.push_back(kv);
  }
  SegTable t{};
  int tot = 0, rows = 0;
  for (int i = 0; i < n; ++i) {
    ATS_REQUIRE(ids[i] && pos[i] && slots[i] && vis[i], ATSPEED_ERR_INVALID, "forward_batch: null array for sequence %d", i);
    ATS_REQUIRE(n_tokens[i] >= 1 && n_tokens[i] <= m->cfg.max_tokens, ATSPEED_ERR_CAPACITY, "forward_batch: %d tokens exceed max_tokens %d",
                n_tokens[i], m->cfg.max_tokens);
    ATS_REQUIRE(n_logit_rows[i] >= 0 && n_logit_rows[i] <= n_tokens[i], ATSPEED_ERR_INVALID, "forward_batch: bad logit row count");
    Seg& sg = t.seg[t.n++];
    sg.ids = ids[i]; sg.pos = pos[i]; sg.slot = slots[i]; sg.vis = vis[i]; sg.kc = m->kv_pool[i].k; sg.vc = m->kv_pool[i].v;
    sg.n_tok = n_tokens[i]; sg.n_slots = n_slots_visible[i]; sg.n_logit = n_logit_rows[i];
    tot += n_tokens[i]; rows += n_logit_rows[i];
  }
  ATS_TRY(seg_finish(t));
  ATS_TRY(ensure_act(m, tot, rows));
  return llama_forward_segs(m, t, logits_out, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------- decoder
namespace {
constexpr int MAXB = ATSPEED_MAX_BEAMS;
constexpr int LMAX = ATSPEED_MAX_NEW_TOKENS;
constexpr int NBLK = ATSPEED_MAX_GAMMA + 1;
}  // namespace

// One user: private KV caches for target and draft, device-side beam state, pinned mailbox and argument staging,
// plus the bookkeeping of the generate call in flight.  Forwards of several users are batched (SegTable).
struct atspeed_decoder {
  atspeed_llama *target, *draft;
  KvCache tkv, dkv;
  int max_prompt, tok_cap, W;
  char* arena;
  TokBuf tin[2], dround;
  BeamSet round_beams[2], blk[NBLK];
  Mailbox* mail_dev;
  Mailbox* mail_host;       // pinned
  int32_t* trace_host;      // pinned: per round [dl][MAXB] draft flat ids
  std::vector<int32_t> trace;   // rounds: {dl, n_matches, nb, flat ids...}
  // decision trace (atspeed_decoder_set_trace level 1): after every round the whole beam area (round beams, draft blocks) and the
  // verify walk's picks travel to the host; one record per round in `decisions` (format: include/atspeed_hip.h)
  int trace_level = 0;
  char* beam_area = nullptr; size_t beam_area_bytes = 0, bs_bytes = 0;
  int32_t* vtrace_dev = nullptr;
  char* dump_host = nullptr;    // pinned, beam_area_bytes (allocated on first use)
  std::vector<int32_t> decisions;
  // sampling mode (atspeed_decoder_set_sampling): off by default = the greedy path of every BASELINE config
  bool sample = false; float temperature = 1.f; uint32_t seed = 0;
  float* tab_score = nullptr;   // [ATSPEED_MAX_GAMMA][ATS_MAX_CAND] the draft's candidate scores per step (allocated on first use)
  int32_t* tab_off = nullptr;   // [ATSPEED_MAX_GAMMA][MAXB + 1]
  float* tab_lse = nullptr;     // [ATSPEED_MAX_GAMMA]
  struct Run {
    const atspeed_fsm* fsm; int gamma, max_new, k, dk;
    int32_t* out_tokens; float* out_scores; atspeed_gen_stats* stats_out;
    atspeed_gen_stats s;
    int cur, gen, base, n0, nb, dl;
    bool reingest, final_step, export_only, done;
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
  ATS_TRY(kv_create(target, &d->tkv));
  if (draft) ATS_TRY(kv_create(draft, &d->dkv));
  d->max_prompt = max_prompt;
  d->W = target->vis_words;
  d->tok_cap = max_prompt + ATSPEED_MAX_GAMMA * MAXB + MAXB;
  size_t tb_bytes = 3 * align_up((size_t)d->tok_cap * 4, 256) + align_up((size_t)d->tok_cap * d->W * 8, 256);
  size_t bs_bytes = 5 * 256 + align_up((size_t)MAXB * LMAX * 4, 256);
  size_t vt_bytes = align_up((size_t)NBLK * 3 * MAXB * sizeof(int32_t), 256);
  size_t total = 3 * tb_bytes + (2 + NBLK) * bs_bytes + vt_bytes + 256;
  ATS_HIP(hipMalloc((void**)&d->arena, total));
  ATS_HIP(hipMemset(d->arena, 0, total));
  char* p = d->arena;
  carve_tokbuf(p, d->tin[0], d->tok_cap, d->W);
  carve_tokbuf(p, d->tin[1], d->tok_cap, d->W);
  carve_tokbuf(p, d->dround, d->tok_cap, d->W);
  d->beam_area = p; d->bs_bytes = bs_bytes; d->beam_area_bytes = (2 + NBLK) * bs_bytes + vt_bytes;
  carve_beams(p, d->round_beams[0]);
  carve_beams(p, d->round_beams[1]);
  for (int i = 0; i < NBLK; ++i) carve_beams(p, d->blk[i]);
  d->vtrace_dev = (int32_t*)p; p += vt_bytes;
  p += 256;
  // the mailbox is pinned host memory the kernels write directly (16 bytes per round over PCIe): the host reads it after the
  // round's stream synchronisation, no copy to enqueue per user
  ATS_HIP(hipHostMalloc((void**)&d->mail_host, sizeof(Mailbox), hipHostMallocMapped));
  ATS_HIP(hipHostGetDevicePointer((void**)&d->mail_dev, d->mail_host, 0));
  ATS_HIP(hipHostMalloc((void**)&d->trace_host, sizeof(int32_t) * ATSPEED_MAX_GAMMA * MAXB));
  d->run.done = true;
  *out = d;
  return ATSPEED_OK;
}

extern "C" void atspeed_decoder_destroy(atspeed_decoder* d) {
  if (!d) return;
  hipFree(d->tab_score); hipFree(d->tab_off); hipFree(d->tab_lse);
  hipDeviceSynchronize();
  kv_free(&d->tkv);
  if (d->draft) kv_free(&d->dkv);
  hipFree(d->arena);
  hipHostFree(d->mail_host);
  hipHostFree(d->trace_host);
  if (d->dump_host) hipHostFree(d->dump_host);
  delete d;
}

extern "C" int atspeed_decoder_set_sampling(atspeed_decoder* d, int32_t do_sample, float temperature, uint32_t seed) {
  ATS_REQUIRE(d, ATSPEED_ERR_INVALID, "set_sampling: null decoder");
  ATS_REQUIRE(!do_sample || (temperature > 0.f && temperature == temperature), ATSPEED_ERR_INVALID, "set_sampling: temperature must be positive");
  d->sample = do_sample != 0; d->temperature = do_sample ? temperature : 1.f; d->seed = seed;
  if (d->sample && !d->tab_score) {
    ATS_HIP(hipMalloc((void**)&d->tab_score, sizeof(float) * ATSPEED_MAX_GAMMA * ATS_MAX_CAND));
    ATS_HIP(hipMalloc((void**)&d->tab_off, sizeof(int32_t) * ATSPEED_MAX_GAMMA * (MAXB + 1)));
    ATS_HIP(hipMalloc((void**)&d->tab_lse, sizeof(float) * ATSPEED_MAX_GAMMA));
  }
  return ATSPEED_OK;
}

extern "C" int atspeed_decoder_set_trace(atspeed_decoder* d, int32_t level) {
  ATS_REQUIRE(d && (level == 0 || level == 1), ATSPEED_ERR_INVALID, "set_trace: level must be 0 or 1");
  if (level == 1 && !d->dump_host) ATS_HIP(hipHostMalloc((void**)&d->dump_host, d->beam_area_bytes));
  d->trace_level = level;
  return ATSPEED_OK;
}

extern "C" int64_t atspeed_decoder_decisions(atspeed_decoder* d, int32_t* out, int64_t cap_words) {
  if (!d) return 0;
  const int64_t n = (int64_t)d->decisions.size();
  if (out && cap_words > 0) memcpy(out, d->decisions.data(), sizeof(int32_t) * (size_t)std::min(n, cap_words));
  return n;
}

// one record of the decision trace from the host copy of the beam area: header {kind, nb, dl, nm, gen0, k, dk, n_blocks}, the blocks
// (beam-set images in arena order: score, node, parent, tok, flat [MAXB] each, seq [MAXB][LMAX]) and, for a verify round, the walk's picks
static void decisions_append(atspeed_decoder* d, int kind, int nb, int dl, int nm, int gen0, int k, int dk, const std::vector<int>& blocks) {
  std::vector<int32_t>& v = d->decisions;
  const int32_t hdr[8] = {kind, nb, dl, nm, gen0, k, dk, (int32_t)blocks.size()};
  v.insert(v.end(), hdr, hdr + 8);
  const size_t bw = d->bs_bytes / 4;
  for (int b : blocks) {
    const int32_t* src = (const int32_t*)(d->dump_host + (size_t)b * d->bs_bytes);
    v.insert(v.end(), src, src + bw);
  }
  if (kind == 0) {
    const int32_t* vt = (const int32_t*)(d->dump_host + (size_t)(2 + NBLK) * d->bs_bytes);
    v.insert(v.end(), vt, vt + (size_t)NBLK * 3 * MAXB);
  }
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
  ATS_REQUIRE(start_node >= 0 && start_node < std::max(fsm->dev.n_nodes, 1), ATSPEED_ERR_INVALID, "generate: start node out of range");
  ATS_REQUIRE(!(d->sample && fsm->dev.n_nodes == 0), ATSPEED_ERR_INVALID, "generate: sampling needs a constraint automaton (mask-free search is greedy only)");
  return ATSPEED_OK;
}

// a user that ends with NO valid beam (per-user ERR_FILTERED inside a lock-step batch): its result block must not keep whatever the caller's
// buffer held before (torch.empty hands back the previous batch's block): scores = -inf, tokens = 0, stream ordered
static int blank_outputs(int32_t* out_tokens, float* out_scores, int k, int max_new, hipStream_t st) {
  ATS_HIP(hipMemsetAsync(out_tokens, 0, (size_t)k * max_new * sizeof(int32_t), st));
  ATS_HIP(hipMemsetD32Async((hipDeviceptr_t)out_scores, (int)0xff800000u, (size_t)k, st));      // -inf
  return ATSPEED_OK;
}

static int mailbox_status(atspeed_decoder* d) {
  if (d->mail_host->status == ATSPEED_ERR_CONSTRAINT) {
    atspeed_set_error("`prefix_allowed_tokens_fn` returned an empty list for batch ID 0. This means that the constraint is unsatisfiable.");
    return ATSPEED_ERR_CONSTRAINT;
  }
  if (d->mail_host->status == ATSPEED_ERR_FILTERED) {
    atspeed_set_error("every beam of a step was dropped by the post-top-k id filter (tokens below %d other than %d, beamSD.py:80-86): "
                      "set the automaton's thresholds with atspeed_fsm_set_id_filter", d->run.fsm ? d->run.fsm->dev.filter_min : 32000,
                      d->run.fsm ? d->run.fsm->dev.filter_eos : 2);
    return ATSPEED_ERR_FILTERED;
  }
  if (d->mail_host->status != 0) {
    atspeed_set_error("decoder: device status %d (candidate capacity exceeded?)", d->mail_host->status);
    return d->mail_host->status;
  }
  return ATSPEED_OK;
}

// ---- a group of users decoded in lock step ----------------------------------------------------------------------
struct StageEvents { hipEvent_t ev[8]; bool init = false; };
static thread_local StageEvents g_ev_dev[ATS_MAX_DEVICES];    // stage-time events, per device
static int stage_events(hipEvent_t** out) {
  const int dev = ats_cur_device();
  ATS_REQUIRE(dev >= 0, ATSPEED_ERR_NO_DEVICE, "stage events: no current HIP device, or its id is >= %d", ATS_MAX_DEVICES);
  StageEvents& s = g_ev_dev[dev];
  if (!s.init) { for (auto& e : s.ev) ATS_HIP(hipEventCreate(&e)); s.init = true; }
  *out = s.ev;
  return ATSPEED_OK;
}

template <typename A>
static int stage_args(const std::vector<A>& v, const A** dev_out, hipStream_t st) {
  const void* p = nullptr;
  ATS_TRY(ats_stage(v.data(), v.size() * sizeof(A), &p, st));
  *dev_out = (const A*)p;
  return ATSPEED_OK;
}

static Seg make_seg(const TokBuf& tb, int n_tok, int n_slots, int n_logit, const KvCache& kv) {
  Seg s{};
  s.ids = tb.ids; s.pos = tb.pos; s.slot = tb.slot; s.vis = tb.vis; s.kc = kv.k; s.vc = kv.v;
  s.n_tok = n_tok; s.n_slots = n_slots; s.n_logit = n_logit;
  return s;
}
static int seg_finish(SegTable& t) {
  t.total_tok = t.total_logit = t.n_qtiles = 0;
  int long_segs = 0;
  for (int i = 0; i < t.n; ++i) long_segs += t.seg[i].n_tok > 96 ? 1 : 0;
  // 128-row tiles (8 waves) halve the K/V re-reads of long segments; 256-row tiles (16 waves, kept in attn.hip) measured slower:
  // 393 vs ~210 us per launch at 64 users x 220 tokens
  t.qtile_rows = (2 * long_segs >= t.n) ? 128 : 64;
  // lock-step batches (thousands of workgroups) run the 32-rows-per-wave kernel: its 4-wave 128-row tile also wins on short segments
  // (the idle waves still carry a quarter of the tile's DMA): 249 vs 286 us at 256 users
  if (t.n >= 16) t.qtile_rows = 128;
  for (int i = 0; i < t.n; ++i) {
    t.seg[i].row0 = t.total_tok; t.total_tok += t.seg[i].n_tok;
    t.seg[i].logit_row0 = t.total_logit; t.total_logit += t.seg[i].n_logit;
    for (int j = 0; j * t.qtile_rows < t.seg[i].n_tok; ++j) {
      ATS_REQUIRE(t.n_qtiles < ATS_MAX_QTILES, ATSPEED_ERR_CAPACITY, "forward: too many query tiles in one batch");
      t.qtile_seg[t.n_qtiles] = (unsigned char)i; t.qtile_idx[t.n_qtiles++] = (unsigned char)j;
    }
  }
  return ATSPEED_OK;
}

// prompts of every user of a batch -> token buffers, start beams and mailboxes, one launch
static int init_prompts_multi(atspeed_decoder** decs, int n, const int32_t* const* prompts, const int32_t* prompt_lens, const int32_t* start_nodes,
                              hipStream_t st) {
  std::vector<InitPromptArgs> ia(n);
  int max_p = 1;
  for (int u = 0; u < n; ++u) {
    atspeed_decoder* d = decs[u];
    ia[u] = InitPromptArgs{d->tin[0], prompts[u], prompt_lens[u], start_nodes[u], d->round_beams[0], d->mail_dev};
    max_p = std::max(max_p, (int)prompt_lens[u]);
  }
  const InitPromptArgs* dev = nullptr;
  ATS_TRY(stage_args(ia, &dev, st));
  return ats_init_prompt_multi(dev, n, max_p, decs[0]->W, decs[0]->target->cfg.vocab_size, st);
}

static int bssd_begin(atspeed_decoder* d, const int32_t* prompt, int P, const atspeed_fsm* fsm, int start_node, int gamma,
                      int max_new, int k, int dk, int32_t* out_tokens, float* out_scores, atspeed_gen_stats* stats,
                      hipStream_t st, bool init_prompt = true) {
  ATS_TRY(check_common(d, prompt, P, fsm, start_node, max_new, k, out_tokens, out_scores));
  ATS_REQUIRE(d->draft, ATSPEED_ERR_INVALID, "bssd: decoder was created without a draft model");
  ATS_REQUIRE(dk >= k && dk <= MAXB, ATSPEED_ERR_CAPACITY, "bssd: draft beam size %d must be in [k=%d, %d]", dk, k, MAXB);
  ATS_REQUIRE(gamma >= 1 && gamma <= ATSPEED_MAX_GAMMA, ATSPEED_ERR_CAPACITY, "bssd: gamma %d out of [1,%d]", gamma, ATSPEED_MAX_GAMMA);
  atspeed_decoder::Run& r = d->run;
  r.fsm = fsm; r.gamma = gamma; r.max_new = max_new; r.k = k; r.dk = dk;
  r.out_tokens = out_tokens; r.out_scores = out_scores; r.stats_out = stats;
  memset(&r.s, 0, sizeof(r.s));
  r.cur = 0; r.gen = 0; r.base = 0; r.n0 = P; r.nb = 1; r.dl = 0;
  r.reingest = false; r.final_step = false; r.export_only = false; r.done = false;
  d->trace.clear();
  d->decisions.clear();
  if (init_prompt) ATS_TRY(ats_init_prompt(d->tin[0], prompt, P, d->W, d->round_beams[0], start_node, d->target->cfg.vocab_size, d->mail_dev, st));
  return ATSPEED_OK;
}

// BSSD (beamSD.py:458-542) for n users in lock step: every draft step and every target verification of the round
// is ONE forward over the tokens of all users that need it (weights are streamed once per forward, not per user).
static int bssd_group_run(atspeed_decoder** decs, int n, hipStream_t st) {
  atspeed_llama *T = decs[0]->target, *D = decs[0]->draft;
  const int W = decs[0]->W;
  hipEvent_t* g_ev = nullptr;
  ATS_TRY(stage_events(&g_ev));
  // capacity for the largest possible batched forward of this group
  int cap_t = 0, cap_r = 0, cap_d = 0;
  for (int u = 0; u < n; ++u) {
    const atspeed_decoder::Run& r = decs[u]->run;
    cap_t += std::max(r.n0, r.k) + r.gamma * r.dk;
    cap_r += MAXB + r.gamma * r.dk;
    cap_d += std::max(std::max(r.n0, r.dk + r.k), r.dk);
  }
  ATS_TRY(ensure_act(T, cap_t, cap_r));
  ATS_TRY(ensure_act(D, cap_d, n * MAXB));
  bool any = true;
  while (any) {
    std::vector<atspeed_decoder*> ver, fin;          // users doing a verify round / the final single step this round
    int max_dl = 0;
    for (int u = 0; u < n; ++u) {
      atspeed_decoder* d = decs[u];
      atspeed_decoder::Run& r = d->run;
      if (r.done) continue;
      r.final_step = r.export_only = false;
      if (r.gen >= r.max_new) { r.final_step = r.export_only = true; continue; }
      r.dl = std::min(r.gamma, r.max_new - r.gen - 1);                               // beamSD.py:504
      if (r.dl == 0) { r.final_step = true; fin.push_back(d); continue; }            // :505-509
      ATS_REQUIRE(r.n0 + r.dl * r.dk <= d->tok_cap, ATSPEED_ERR_CAPACITY, "bssd: packed target input too long");
      ATS_REQUIRE(r.base + r.n0 + r.dl * r.dk <= T->cfg.max_slots, ATSPEED_ERR_CAPACITY,
                  "bssd: KV slots exhausted (%d needed, %d available)", r.base + r.n0 + r.dl * r.dk, T->cfg.max_slots);
      ver.push_back(d);
      max_dl = std::max(max_dl, r.dl);
    }
    hipEventRecord(g_ev[0], st);
    // ---- 1. draft: step i of one_step_beam_search for every user that still drafts (:108-179)
    for (int i = 0; i < max_dl; ++i) {
      SegTable t{};
      std::vector<BeamStepArgs> args;
      std::vector<atspeed_decoder*> us;
      for (atspeed_decoder* d : ver) if (d->run.dl > i) us.push_back(d);
      for (atspeed_decoder* d : us) {
        atspeed_decoder::Run& r = d->run;
        TokBuf& tin = d->tin[r.cur];
        Seg sg;
        if (i == 0) sg = r.reingest ? make_seg(d->dround, r.dk + r.k, r.base + r.n0, r.nb, d->dkv)
                                    : make_seg(tin, r.n0, r.base + r.n0, r.nb, d->dkv);
        else        sg = make_seg(tb_offset(tin, r.n0 + (i - 1) * r.dk, W), r.dk, r.base + r.n0 + i * r.dk, r.dk, d->dkv);
        t.seg[t.n++] = sg;
        r.s.n_draft_forwards++;
      }
      ATS_TRY(seg_finish(t));
      ATS_TRY(llama_forward_segs(D, t, nullptr, st, decs[0]->run.fsm->d_tile_store));
      const bool free_fsm = decs[0]->run.fsm->dev.n_nodes == 0;
      if (free_fsm) ATS_TRY(ats_row_topk(D->act->logits, t.total_logit, D->cfg.vocab_size, D->logits_ld, decs[0]->run.dk, D->act->row_cand, st));
      for (size_t j = 0; j < us.size(); ++j) {
        atspeed_decoder* d = us[j];
        atspeed_decoder::Run& r = d->run;
        TokBuf& tin = d->tin[r.cur];
        BeamStepArgs a{};
        a.src = i == 0 ? d->round_beams[r.cur] : d->blk[i]; a.n_src = t.seg[j].n_logit; a.gen_len = r.gen + i;
        a.logits = D->act->logits + (size_t)t.seg[j].logit_row0 * D->logits_ld; a.ld = D->logits_ld;
        a.lse = D->act->lse + t.seg[j].logit_row0; a.fsm = r.fsm->dev; a.k = r.dk;
        a.dst = d->blk[i + 1]; a.emit = 1; a.filter_ids = free_fsm ? 0 : 1;
        a.row_cand = D->act->row_cand + (size_t)t.seg[j].logit_row0 * ATSPEED_MAX_BEAMS; a.n_row_cand = r.dk;
        a.in = tin; a.in_row0 = i == 0 ? r.n0 - r.nb : r.n0 + (i - 1) * r.dk;
        a.out = tin; a.out_row0 = r.n0 + i * r.dk; a.out_slot0 = r.base + r.n0 + i * r.dk; a.vis_words = W;
        a.mail = d->mail_dev;
        if (d->sample) {
          a.sample = 1; a.temperature = d->temperature; a.rng_sub = ats_rng_sub(d->seed, ATS_RNG_STEP, r.s.n_run, i, 1);
          a.tab_score = d->tab_score + (size_t)i * ATS_MAX_CAND; a.tab_off = d->tab_off + (size_t)i * (MAXB + 1); a.tab_lse = d->tab_lse + i;
        }
        args.push_back(a);
      }
      const BeamStepArgs* dev_args = nullptr;
      ATS_TRY(stage_args(args, &dev_args, st));
      ATS_TRY(ats_beam_step_multi(dev_args, (int)args.size(), st));
    }
    hipEventRecord(g_ev[1], st);
    // ---- 2. target: ONE forward over (round inputs ++ draft blocks) of every verifying user and the inputs of
    //         every user on its final step (:190-232, :505-509)
    if (!ver.empty() || !fin.empty()) {
      SegTable t{};
      for (atspeed_decoder* d : ver) {
        atspeed_decoder::Run& r = d->run;
        int Tn = r.n0 + r.dl * r.dk;
        t.seg[t.n++] = make_seg(d->tin[r.cur], Tn, r.base + Tn, r.nb + r.dl * r.dk, d->tkv);
        r.s.n_target_forwards++;
      }
      for (atspeed_decoder* d : fin) {
        atspeed_decoder::Run& r = d->run;
        t.seg[t.n++] = make_seg(d->tin[r.cur], r.n0, r.base + r.n0, r.nb, d->tkv);
        r.s.n_target_forwards++;
      }
      ATS_TRY(seg_finish(t));
      ATS_TRY(llama_forward_segs(T, t, nullptr, st, decs[0]->run.fsm->d_tile_store));
      const bool free_fsm = decs[0]->run.fsm->dev.n_nodes == 0;
      if (free_fsm) ATS_TRY(ats_row_topk(T->act->logits, t.total_logit, T->cfg.vocab_size, T->logits_ld, decs[0]->run.k, T->act->row_cand, st));
      hipEventRecord(g_ev[2], st);
      // ---- 3. verify (:242-456) for the verifying users, one workgroup each
      std::vector<VerifyArgs> vargs;
      for (size_t j = 0; j < ver.size(); ++j) {
        atspeed_decoder* d = ver[j];
        atspeed_decoder::Run& r = d->run;
        VerifyArgs va{};
        va.blk[0] = d->round_beams[r.cur];
        for (int i = 1; i <= r.dl; ++i) va.blk[i] = d->blk[i];
        va.nb = r.nb; va.dl = r.dl; va.k = r.k; va.dk = r.dk; va.gen_len0 = r.gen;
        va.logits = T->act->logits + (size_t)t.seg[j].logit_row0 * T->logits_ld; va.ld = T->logits_ld;
        va.lse = T->act->lse + t.seg[j].logit_row0; va.fsm = r.fsm->dev;
        va.row_cand = T->act->row_cand + (size_t)t.seg[j].logit_row0 * ATSPEED_MAX_BEAMS; va.n_row_cand = r.k;
        va.cur = d->tin[r.cur]; va.n0 = r.n0; va.next = d->tin[r.cur ^ 1]; va.dnext = d->dround; va.vis_words = W;
        va.res = d->round_beams[r.cur ^ 1]; va.mail = d->mail_dev;
        va.vtrace = (d->trace_level >= 1 && !d->sample) ? d->vtrace_dev : nullptr;
        if (d->sample) {
          va.sample = 1; va.temperature = d->temperature; va.seed = d->seed; va.round = r.s.n_run;
          for (int i = 0; i < r.dl; ++i) {
            va.dtab_score[i] = d->tab_score + (size_t)i * ATS_MAX_CAND; va.dtab_off[i] = d->tab_off + (size_t)i * (MAXB + 1);
            va.dtab_lse[i] = d->tab_lse + i;
          }
        }
        vargs.push_back(va);
      }
      if (!vargs.empty()) {
        const VerifyArgs* dv = nullptr;
        ATS_TRY(stage_args(vargs, &dv, st));
        ATS_TRY(ats_verify_walk_multi(dv, (int)vargs.size(), st));
      }
      std::vector<BeamStepArgs> fargs;
      for (size_t j = 0; j < fin.size(); ++j) {
        atspeed_decoder* d = fin[j];
        atspeed_decoder::Run& r = d->run;
        const Seg& sg = t.seg[ver.size() + j];
        BeamStepArgs a{};
        a.src = d->round_beams[r.cur]; a.n_src = r.nb; a.gen_len = r.gen;
        a.logits = T->act->logits + (size_t)sg.logit_row0 * T->logits_ld; a.ld = T->logits_ld;
        a.lse = T->act->lse + sg.logit_row0; a.fsm = r.fsm->dev; a.k = r.k;
        a.dst = d->round_beams[r.cur ^ 1]; a.emit = 0; a.filter_ids = free_fsm ? 0 : 1; a.mail = d->mail_dev; a.vis_words = W;
        a.row_cand = T->act->row_cand + (size_t)sg.logit_row0 * ATSPEED_MAX_BEAMS; a.n_row_cand = r.k;
        if (d->sample) { a.sample = 1; a.temperature = d->temperature; a.rng_sub = ats_rng_sub(d->seed, ATS_RNG_STEP, r.s.n_run, 0, 0); }
        fargs.push_back(a);
      }
      if (!fargs.empty()) {
        const BeamStepArgs* df = nullptr;
        ATS_TRY(stage_args(fargs, &df, st));
        ATS_TRY(ats_beam_step_multi(df, (int)fargs.size(), st));
      }
      for (atspeed_decoder* d : fin) { d->run.cur ^= 1; d->run.gen += 1; }
    } else {
      hipEventRecord(g_ev[2], st);
    }
    hipEventRecord(g_ev[3], st);
    // ---- 4. outputs of finished users, mailboxes, the round's single synchronisation
    std::vector<ExportBeamsArgs> ex;                  // users finishing this round: one export launch for all of them
    for (int u = 0; u < n; ++u) {
      atspeed_decoder* d = decs[u];
      atspeed_decoder::Run& r = d->run;
      if (!r.done && r.final_step) ex.push_back(ExportBeamsArgs{d->round_beams[r.cur], r.out_tokens, r.out_scores, d->sample ? 1 : 0});
    }
    if (ex.size() == 1) ATS_TRY(ats_export_beams(ex[0].b, decs[0]->run.k, decs[0]->run.max_new, ex[0].out_tokens, ex[0].out_scores, st, ex[0].sort_desc != 0));
    else if (!ex.empty()) {
      const ExportBeamsArgs* de = nullptr;
      ATS_TRY(stage_args(ex, &de, st));
      ATS_TRY(ats_export_beams_multi(de, (int)ex.size(), decs[0]->run.k, decs[0]->run.max_new, st));
    }
    for (int u = 0; u < n; ++u) {
      atspeed_decoder* d = decs[u];
      atspeed_decoder::Run& r = d->run;
      if (r.done) continue;
      if (r.final_step) {}
      else if (n <= 4)                      // trace of the draft's flat ids (parity tests drive one user at a time; batches skip the copies)
        for (int i = 1; i <= r.dl; ++i)
          ATS_HIP(hipMemcpyAsync(d->trace_host + (i - 1) * MAXB, d->blk[i].flat, sizeof(int32_t) * r.dk, hipMemcpyDeviceToHost, st));
      if (d->trace_level >= 1 && !r.export_only)
        ATS_HIP(hipMemcpyAsync(d->dump_host, d->beam_area, d->beam_area_bytes, hipMemcpyDeviceToHost, st));
    }
    ATS_HIP(hipStreamSynchronize(st));
    ats_stage_reset();
    float ms_d = 0.f, ms_t = 0.f, ms_v = 0.f;
    hipEventElapsedTime(&ms_d, g_ev[0], g_ev[1]);
    hipEventElapsedTime(&ms_t, g_ev[1], g_ev[2]);
    hipEventElapsedTime(&ms_v, g_ev[2], g_ev[3]);
    int n_act = 0;
    for (int u = 0; u < n; ++u) if (!decs[u]->run.done) ++n_act;
    any = false;
    for (int u = 0; u < n; ++u) {
      atspeed_decoder* d = decs[u];
      atspeed_decoder::Run& r = d->run;
      if (r.done) continue;
      if (n > 1 && d->mail_host->status == ATSPEED_ERR_FILTERED) {
        // a lock-step batch does not die with one user: this user lost every beam of a step to the id filter (beamSD.py:80-86; the
        // reference dies on a shape mismatch there) and ends with no valid beam, the others go on (the one-user call returns the error)
        r.s.n_valid = 0; r.s.status = ATSPEED_ERR_FILTERED;
        ATS_TRY(blank_outputs(r.out_tokens, r.out_scores, r.k, r.max_new, st));
        r.s.total_ms = r.s.draft_ms + r.s.target_ms + r.s.verify_ms;
        if (r.stats_out) *r.stats_out = r.s;
        r.done = true;
        continue;
      }
      ATS_TRY(mailbox_status(d));
      // stage times: the group's stage time shared equally by the users that were active in the round
      r.s.draft_ms += ms_d / n_act; r.s.target_ms += ms_t / n_act; r.s.verify_ms += ms_v / n_act;
      if (r.final_step) {
        // `cur` was flipped when the step was launched: the parents are round_beams[cur ^ 1], the result round_beams[cur]
        if (d->trace_level >= 1 && !r.export_only) decisions_append(d, 1, r.nb, 0, 0, r.gen - 1, r.k, r.dk, {r.cur ^ 1, r.cur});
        r.s.n_valid = d->mail_host->n_valid;
        r.s.total_ms = r.s.draft_ms + r.s.target_ms + r.s.verify_ms;
        if (r.stats_out) *r.stats_out = r.s;
        r.done = true;
        continue;
      }
      const int nm = d->mail_host->n_matches, dl = r.dl, dk = r.dk;
      if (n <= 4) {
        d->trace.push_back(dl); d->trace.push_back(nm); d->trace.push_back(r.nb);
        for (int i = 0; i < dl; ++i) for (int j = 0; j < dk; ++j) d->trace.push_back(d->trace_host[i * MAXB + j]);
      }
      if (d->trace_level >= 1) {
        std::vector<int> blocks{r.cur};
        for (int i = 1; i <= dl; ++i) blocks.push_back(2 + i);
        blocks.push_back(r.cur ^ 1);
        decisions_append(d, 0, r.nb, dl, nm, r.gen, r.k, dk, blocks);
      }
      if (r.s.n_run < ATSPEED_MAX_NEW_TOKENS) r.s.accept_steps[r.s.n_run] = nm;
      r.s.n_run++;
      r.s.total_accept_steps += nm;
      r.base += r.n0 + nm * dk;             // compact: keep up to the end of block nm
      r.n0 = r.k; r.nb = r.k;
      r.gen += nm + 1;                      // :522
      r.reingest = (nm == dl);
      r.cur ^= 1;
      any = true;
    }
  }
  return ATSPEED_OK;
}

extern "C" int atspeed_bssd_generate_batch(atspeed_decoder** decs, int32_t n, const int32_t* const* prompts,
                                           const int32_t* prompt_lens, const atspeed_fsm* fsm, const int32_t* start_nodes,
                                           int32_t gamma, int32_t max_new, int32_t k, int32_t dk, int32_t* const* out_tokens,
                                           float* const* out_scores, atspeed_gen_stats* stats, void* stream) {
  ATS_REQUIRE(decs && prompts && prompt_lens && start_nodes && out_tokens && out_scores, ATSPEED_ERR_INVALID, "bssd_batch: null argument");
  ATS_REQUIRE(n >= 1 && n <= ATS_MAX_SEGS, ATSPEED_ERR_CAPACITY, "bssd_batch: %d users per call (max %d)", n, ATS_MAX_SEGS);
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < n; ++i) {
    ATS_REQUIRE(decs[i] && decs[i]->target == decs[0]->target && decs[i]->draft == decs[0]->draft, ATSPEED_ERR_INVALID,
                "bssd_batch: decoders must share one target/draft pair");
    for (int j = 0; j < i; ++j) ATS_REQUIRE(decs[i] != decs[j], ATSPEED_ERR_INVALID, "bssd_batch: decoder %d used twice", i);
    ATS_TRY(bssd_begin(decs[i], prompts[i], prompt_lens[i], fsm, start_nodes[i], gamma, max_new, k, dk, out_tokens[i], out_scores[i],
                       stats ? &stats[i] : nullptr, st, false));
  }
  ATS_TRY(init_prompts_multi(decs, n, prompts, prompt_lens, start_nodes, st));
  return bssd_group_run(decs, n, st);
}

extern "C" int atspeed_bssd_generate(atspeed_decoder* d, const int32_t* prompt, int32_t P, const atspeed_fsm* fsm,
                                     int32_t start_node, int32_t gamma, int32_t max_new, int32_t k, int32_t dk,
                                     int32_t* out_tokens, float* out_scores, atspeed_gen_stats* stats, void* stream) {
  ATS_TRY(bssd_begin(d, prompt, P, fsm, start_node, gamma, max_new, k, dk, out_tokens, out_scores, stats, (hipStream_t)stream));
  return bssd_group_run(&d, 1, (hipStream_t)stream);
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
  ATS_REQUIRE(P + max_new * k <= d->tok_cap, ATSPEED_ERR_CAPACITY, "target_generate: token buffer too small");
  ATS_TRY(ensure_act(T, std::max(P, k), MAXB));
  hipEvent_t* g_ev = nullptr;
  ATS_TRY(stage_events(&g_ev));
  TokBuf& tin = d->tin[0];
  ATS_TRY(ats_init_prompt(tin, prompt, P, W, d->round_beams[0], start_node, T->cfg.vocab_size, d->mail_dev, st));
  hipEventRecord(g_ev[0], st);
  int cur = 0, row0 = 0, n_in = P, nb = 1, base = 0;
  for (int g = 0; g < max_new; ++g) {                                                // beamSD.py:579-588
    SegTable t{};
    t.n = 1;
    t.seg[0] = make_seg(tb_offset(tin, row0, W), n_in, base + n_in, nb, d->tkv);
    ATS_TRY(seg_finish(t));
    ATS_TRY(llama_forward_segs(T, t, nullptr, st, fsm->d_tile_store));
    const bool free_fsm = fsm->dev.n_nodes == 0;
    if (free_fsm) ATS_TRY(ats_row_topk(T->act->logits, t.total_logit, V, T->logits_ld, k, T->act->row_cand, st));
    s.n_target_forwards++;
    BeamStepArgs a{};
    a.src = d->round_beams[cur]; a.n_src = nb; a.gen_len = g;
    a.logits = T->act->logits; a.ld = T->logits_ld; a.lse = T->act->lse; a.fsm = fsm->dev; a.k = k;
    a.dst = d->round_beams[cur ^ 1]; a.emit = 1; a.filter_ids = free_fsm ? 0 : 1;
    a.row_cand = T->act->row_cand; a.n_row_cand = k;
    a.in = tin; a.in_row0 = row0 + n_in - nb;
    a.out = tin; a.out_row0 = row0 + n_in; a.out_slot0 = base + n_in; a.vis_words = W;
    a.mail = d->mail_dev;
    if (d->sample) { a.sample = 1; a.temperature = d->temperature; a.rng_sub = ats_rng_sub(d->seed, ATS_RNG_STEP, g, 0, 0); }
    ATS_TRY(ats_beam_step(a, st));
    row0 += n_in; base += n_in; n_in = k; nb = k; cur ^= 1;
  }
  ATS_TRY(ats_export_beams(d->round_beams[cur], k, max_new, out_tokens, out_scores, st, d->sample));
  hipEventRecord(g_ev[1], st);
  ATS_HIP(hipStreamSynchronize(st));
  ATS_TRY(mailbox_status(d));
  s.n_valid = d->mail_host->n_valid;
  hipEventElapsedTime(&s.total_ms, g_ev[0], g_ev[1]);
  s.target_ms = s.total_ms;
  if (stats) *stats = s;
  return ATSPEED_OK;
}

// target_generate (beamSD.py:544-595) for n users in lock step: step g of every user is ONE forward + one beam-step launch
// (the teacher-data job of generate_teacher_data.py:211-244 is exactly this loop over a whole training set).
extern "C" int atspeed_target_generate_batch(atspeed_decoder** decs, int32_t n, const int32_t* const* prompts,
                                             const int32_t* prompt_lens, const atspeed_fsm* fsm, const int32_t* start_nodes,
                                             int32_t max_new, int32_t k, int32_t* const* out_tokens, float* const* out_scores,
                                             atspeed_gen_stats* stats, void* stream) {
  ATS_REQUIRE(decs && prompts && prompt_lens && start_nodes && out_tokens && out_scores, ATSPEED_ERR_INVALID, "target_generate_batch: null argument");
  ATS_REQUIRE(n >= 1 && n <= ATS_MAX_SEGS, ATSPEED_ERR_CAPACITY, "target_generate_batch: %d users per call (max %d)", n, ATS_MAX_SEGS);
  hipStream_t st = (hipStream_t)stream;
  atspeed_llama* T = decs[0]->target;
  const int W = decs[0]->W;
  int cap_t = 0;
  for (int u = 0; u < n; ++u) {
    atspeed_decoder* d = decs[u];
    ATS_REQUIRE(d && d->target == T, ATSPEED_ERR_INVALID, "target_generate_batch: decoders must share one target model");
    for (int j = 0; j < u; ++j) ATS_REQUIRE(decs[u] != decs[j], ATSPEED_ERR_INVALID, "target_generate_batch: decoder %d used twice", u);
    ATS_TRY(check_common(d, prompts[u], prompt_lens[u], fsm, start_nodes[u], max_new, k, out_tokens[u], out_scores[u]));
    ATS_REQUIRE(prompt_lens[u] + max_new * k <= T->cfg.max_slots, ATSPEED_ERR_CAPACITY, "target_generate: KV slots exhausted");
    ATS_REQUIRE(prompt_lens[u] + max_new * k <= d->tok_cap, ATSPEED_ERR_CAPACITY, "target_generate: token buffer too small");
    cap_t += std::max(prompt_lens[u], k);
  }
  ATS_TRY(ensure_act(T, cap_t, n * MAXB));
  hipEvent_t* g_ev = nullptr;
  ATS_TRY(stage_events(&g_ev));
  ATS_TRY(init_prompts_multi(decs, n, prompts, prompt_lens, start_nodes, st));
  hipEventRecord(g_ev[0], st);
  struct St { int row0, n_in, nb, base, cur; };
  std::vector<St> s(n);
  for (int u = 0; u < n; ++u) s[u] = St{0, prompt_lens[u], 1, 0, 0};
  for (int g = 0; g < max_new; ++g) {                                                // beamSD.py:579-588
    SegTable t{};
    for (int u = 0; u < n; ++u)
      t.seg[t.n++] = make_seg(tb_offset(decs[u]->tin[0], s[u].row0, W), s[u].n_in, s[u].base + s[u].n_in, s[u].nb, decs[u]->tkv);
    ATS_TRY(seg_finish(t));
    ATS_TRY(llama_forward_segs(T, t, nullptr, st, fsm->d_tile_store));
    const bool free_fsm = fsm->dev.n_nodes == 0;
    if (free_fsm) ATS_TRY(ats_row_topk(T->act->logits, t.total_logit, T->cfg.vocab_size, T->logits_ld, k, T->act->row_cand, st));
    std::vector<BeamStepArgs> args;
    for (int u = 0; u < n; ++u) {
      atspeed_decoder* d = decs[u];
      BeamStepArgs a{};
      a.src = d->round_beams[s[u].cur]; a.n_src = s[u].nb; a.gen_len = g;
      a.logits = T->act->logits + (size_t)t.seg[u].logit_row0 * T->logits_ld; a.ld = T->logits_ld;
      a.lse = T->act->lse + t.seg[u].logit_row0; a.fsm = fsm->dev; a.k = k;
      a.dst = d->round_beams[s[u].cur ^ 1]; a.emit = 1; a.filter_ids = free_fsm ? 0 : 1;
      a.row_cand = T->act->row_cand + (size_t)t.seg[u].logit_row0 * ATSPEED_MAX_BEAMS; a.n_row_cand = k;
      a.in = d->tin[0]; a.in_row0 = s[u].row0 + s[u].n_in - s[u].nb;
      a.out = d->tin[0]; a.out_row0 = s[u].row0 + s[u].n_in; a.out_slot0 = s[u].base + s[u].n_in; a.vis_words = W;
      a.mail = d->mail_dev;
      if (d->sample) { a.sample = 1; a.temperature = d->temperature; a.rng_sub = ats_rng_sub(d->seed, ATS_RNG_STEP, g, 0, 0); }
      args.push_back(a);
      s[u].row0 += s[u].n_in; s[u].base += s[u].n_in; s[u].n_in = k; s[u].nb = k; s[u].cur ^= 1;
    }
    const BeamStepArgs* dev_args = nullptr;
    ATS_TRY(stage_args(args, &dev_args, st));
    ATS_TRY(ats_beam_step_multi(dev_args, n, st));
  }
  { std::vector<ExportBeamsArgs> ex;
    for (int u = 0; u < n; ++u) ex.push_back(ExportBeamsArgs{decs[u]->round_beams[s[u].cur], out_tokens[u], out_scores[u], decs[u]->sample ? 1 : 0});
    const ExportBeamsArgs* de = nullptr;
    ATS_TRY(stage_args(ex, &de, st));
    ATS_TRY(ats_export_beams_multi(de, n, k, max_new, st)); }
  hipEventRecord(g_ev[1], st);
  ATS_HIP(hipStreamSynchronize(st));
  ats_stage_reset();
  float ms = 0.f;
  hipEventElapsedTime(&ms, g_ev[0], g_ev[1]);
  bool any_filtered = false;
  for (int u = 0; u < n; ++u)
    if (n > 1 && decs[u]->mail_host->status == ATSPEED_ERR_FILTERED) { ATS_TRY(blank_outputs(out_tokens[u], out_scores[u], k, max_new, st)); any_filtered = true; }
  if (any_filtered) ATS_HIP(hipStreamSynchronize(st));
  for (int u = 0; u < n; ++u) {
    const bool filtered = n > 1 && decs[u]->mail_host->status == ATSPEED_ERR_FILTERED;    // per user, as in the beam-SD batch loop
    if (!filtered) ATS_TRY(mailbox_status(decs[u]));
    if (stats) {
      atspeed_gen_stats gs;
      memset(&gs, 0, sizeof(gs));
      gs.n_target_forwards = max_new; gs.n_valid = filtered ? 0 : decs[u]->mail_host->n_valid;
      gs.status = filtered ? ATSPEED_ERR_FILTERED : ATSPEED_OK;
      gs.total_ms = gs.target_ms = ms / n;                      // the group's time shared equally by its users
      stats[u] = gs;
    }
  }
  return ATSPEED_OK;
}

extern "C" int atspeed_decoder_trace(atspeed_decoder* d, int32_t* rounds_out, int32_t cap) {
  if (!d) return 0;
  int n = (int)std::min<size_t>(d->trace.size(), (size_t)std::max(cap, 0));
  if (rounds_out && n > 0) memcpy(rounds_out, d->trace.data(), sizeof(int32_t) * n);
  return (int)d->trace.size();
}
