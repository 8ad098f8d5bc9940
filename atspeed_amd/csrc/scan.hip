// Scan kernels of beam speculative decoding: full-vocabulary log-sum-exp, fused
// constraint-mask + beam expand + top-K prune, the top-K-aligned acceptance test, and the
// on-device bookkeeping (visibility bitsets, positions, KV slots, beam suffixes) that the
// reference does with host round trips (beamSD.py:58-91, :278-380, :383-416).
//
// Candidate keys are 64-bit: (monotone(score) << 32) | ~flat_id, so an unsigned max gives
// "highest score first, lowest flat id on ties" — the tie-break this build defines
// (torch.topk leaves it unspecified).  Key 0 = no candidate.
#include "internal.h"
#include <cstdlib>

namespace {

constexpr int kScanThreads = 1024;
constexpr int kScanWaves = kScanThreads / 64;
constexpr int kCPT = 16;                               // candidate slots per thread
constexpr int kMaxCand = kScanThreads * kCPT;          // 16384 >= 64 beams * 256 children
constexpr int MAXB = ATSPEED_MAX_BEAMS;
constexpr int LMAX = ATSPEED_MAX_NEW_TOKENS;
constexpr uint32_t kOrdNegInf = 0x007fffffu;           // ford(-inf)

// ---------------------------------------------------------------------------- LSE
// one workgroup per row, 16-byte non-temporal streaming reads (the row is read once and never again), online (max, sum) per lane.
// Chunks of 8 independent loads per thread, double-buffered: chunk c+1 is in flight while chunk c is exponentiated (with one chunk per
// 1024-thread workgroup the loads and the exponentials of a row alternated and only other workgroups overlapped them: 6.5 TB/s; the
// read-only stream peak measured by atspeed_probe_hbm_read is 7.05 TB/s).  One running-max rescale per chunk instead of per element.
template <int NT>
__global__ __launch_bounds__(NT) void lse_rows_kernel(const float* __restrict__ logits, int vocab, int ld, float* __restrict__ lse) {
  constexpr int U = 8, NWV = NT / 64;
  __shared__ float smax[NWV], ssum[NWV];
  const float* row = logits + (size_t)blockIdx.x * ld;
  const f32x4_t* row4 = reinterpret_cast<const f32x4_t*>(row);
  float m = -INFINITY, s = 0.f;
  const int nv4 = vocab >> 2;
  const int n_full = nv4 / (NT * U);                      // chunks in which every thread has all U loads
  auto load = [&](int c, f32x4_t (&v)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(row4 + (size_t)c * NT * U + u * NT + threadIdx.x);
  };
  auto consume = [&](const f32x4_t (&v)[U]) {
    float cm = -INFINITY;
#pragma unroll
    for (int u = 0; u < U; ++u) cm = fmaxf(cm, fmaxf(fmaxf(v[u][0], v[u][1]), fmaxf(v[u][2], v[u][3])));
    if (cm > m) { s *= __expf(m - cm); m = cm; }
    if (m != -INFINITY) {
#pragma unroll
      for (int u = 0; u < U; ++u) s += __expf(v[u][0] - m) + __expf(v[u][1] - m) + __expf(v[u][2] - m) + __expf(v[u][3] - m);
    }
  };
  f32x4_t a[U], b[U];
  if (n_full > 0) load(0, a);
  for (int c = 0; c < n_full; c += 2) {
    if (c + 1 < n_full) load(c + 1, b);
    consume(a);
    if (c + 1 >= n_full) break;
    if (c + 2 < n_full) load(c + 2, a);
    consume(b);
  }
  for (int i = n_full * NT * U + threadIdx.x; i < nv4; i += NT) {           // the last partial chunk: at most U - 1 loads per thread
    const f32x4_t v = __builtin_nontemporal_load(row4 + i);
    const float cm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    if (cm > m) { s *= __expf(m - cm); m = cm; }
    if (m != -INFINITY) s += __expf(v[0] - m) + __expf(v[1] - m) + __expf(v[2] - m) + __expf(v[3] - m);
  }
  for (int i = (nv4 << 2) + threadIdx.x; i < vocab; i += NT) {
    float v = row[i];
    if (v > m) { s *= __expf(m - v); m = v; }
    s += __expf(v - m);
  }
  float wm = wave_max_f32(m);
  s = (m == -INFINITY) ? 0.f : s * expf(m - wm);
  s = wave_sum_f32(s);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { smax[wave] = wm; ssum[wave] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float gm = smax[0];
    for (int w = 1; w < NWV; ++w) gm = fmaxf(gm, smax[w]);
    float gs = 0.f;
    for (int w = 0; w < NWV; ++w) gs += (smax[w] == -INFINITY) ? 0.f : ssum[w] * expf(smax[w] - gm);
    lse[blockIdx.x] = gm + logf(gs);
  }
}

// ---------------------------------------------------------------------------- block top-k
struct TopkShared {
  unsigned long long wtop[kScanWaves][MAXB];
  unsigned long long sel[MAXB];
  unsigned long long bound;                      // block_topk: lower bound of the k-th largest key after the lead rounds
};

// rounds j0 .. j1-1 of wave-argmax: the owner lane retires its key, out[j] = the j-th largest key of this wave.  Stops early -- zero-filling
// out[j .. k) -- when the wave has nothing left or its best remaining key is below `floor_key`; returns true when it stopped.
template <int NK>
__device__ __forceinline__ bool wave_topk_rounds(unsigned long long (&keys)[NK], int j0, int j1, int k, unsigned long long floor_key,
                                                 unsigned long long* out, int lane) {
  for (int j = j0; j < j1; ++j) {
    unsigned long long m = 0;
#pragma unroll
    for (int c = 0; c < NK; ++c) m = keys[c] > m ? keys[c] : m;
    const unsigned long long M = wave_max_u64(m);
    if (M == 0 || M < floor_key) {                  // wave-uniform: nothing left that can be among the block's k best
      for (int jj = j + lane; jj < k; jj += 64) out[jj] = 0;
      return true;
    }
    if (m == M) {
#pragma unroll
      for (int c = 0; c < NK; ++c) if (keys[c] == M) keys[c] = 0;
    }
    if (lane == 0) out[j] = M;
  }
  return false;
}

// every thread passes its NK candidate keys (distinct, 0 = none); afterwards sh.sel[0..k) holds the k largest, descending.
// Two phases: every wave extracts its kTopkLead best keys; the k-th largest of those 16 x kTopkLead keys is a lower bound of the block's k-th
// largest key, so a wave goes on only while its best remaining key reaches that bound -- 16 x 4 + about k wave-rounds instead of 16 x k
// (the 16 waves of a workgroup share four SIMDs: at k = 40 the full rounds were 50 of the 85 us of one user's beam step).  The selected set
// and its order are the same: every key at or above the bound is extracted by its wave, and the merge takes the k best of their union.
constexpr int kTopkLead = 4;
static_assert(kScanWaves * kTopkLead <= 64, "the bound is computed by one wave, one lead key per lane");
template <int NK>
__device__ void block_topk(unsigned long long (&keys)[NK], int k, TopkShared& sh) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lead = k < kTopkLead ? k : kTopkLead;
  const bool done = wave_topk_rounds<NK>(keys, 0, lead, k, 0ull, sh.wtop[wave], lane);
  __syncthreads();
  if (wave == 0) {
    const unsigned long long x = lane < kScanWaves * lead ? sh.wtop[lane / lead][lane % lead] : 0ull;
    const unsigned xlo = (unsigned)(x & 0xffffffffull), xhi = (unsigned)(x >> 32);
    int rank = 0;                                    // lead keys above mine
#pragma unroll
    for (int l = 0; l < 64; ++l) {
      const unsigned long long y = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)xhi, l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)xlo, l);
      rank += y > x ? 1 : 0;
    }
    // the lead key with k - 1 keys above it (fewer than k lead keys: no such lane, or only lanes holding 0 -> no bound)
    const unsigned long long hit = __ballot(rank == k - 1);
    unsigned long long bound = 0;
    if (hit) {
      const int src = __ffsll((long long)hit) - 1;
      bound = ((unsigned long long)(unsigned)__shfl((int)xhi, src, 64) << 32) | (unsigned)__shfl((int)xlo, src, 64);
    }
    if (lane == 0) sh.bound = bound;
  }
  __syncthreads();
  if (!done) wave_topk_rounds<NK>(keys, lead, k, k, sh.bound, sh.wtop[wave], lane);
  __syncthreads();
  if (wave == 0) {
    unsigned long long k2[kScanWaves];
#pragma unroll
    for (int c = 0; c < kScanWaves; ++c) {
      int e = lane + 64 * c;                         // < 16*64
      k2[c] = (e < kScanWaves * k) ? sh.wtop[e / k][e % k] : 0ull;
    }
    wave_topk_rounds<kScanWaves>(k2, 0, k, k, 0ull, sh.sel, lane);
  }
  __syncthreads();
}

__device__ __forceinline__ int find_edge(const FsmDev& f, int node, int tok) {
  int lo = f.row_ptr[node], hi = f.row_ptr[node + 1];
  while (lo < hi) {                                   // children strictly ascending
    int mid = (lo + hi) >> 1;
    int t = f.tok[mid];
    if (t == tok) return mid;
    if (t < tok) lo = mid + 1; else hi = mid;
  }
  return -1;
}

struct ExpandShared {
  TopkShared topk;
  int off[MAXB + 1];
  int node[MAXB];
  int brow[MAXB];       // row index inside the block (parent id space)
  int lrow[MAXB];       // row index inside the logits buffer
  float bscore[MAXB];
  float lse[MAXB];
  int rp[MAXB];         // first edge of the row's node (fsm.row_ptr[node]) and its number of children
  int deg[MAXB];
  int status;
  float red_m[kScanWaves], red_s[kScanWaves];   // block reductions of the sampling paths
  float red_out;
};

// Expand n_rows parent rows (sh.node/brow/lrow/bscore filled by the caller, first n_rows entries)
// through the FSM and leave the k best candidates in sh.topk.sel.  beamSD.py:58-78.
__device__ void expand_and_select(ExpandShared& sh, int n_rows, const float* __restrict__ logits, int ld,
                                  const float* __restrict__ lse, const FsmDev& fsm, int k,
                                  const int32_t* __restrict__ row_cand = nullptr, int n_row_cand = 0) {
  const int tid = threadIdx.x;
  const bool free_mode = fsm.n_nodes == 0;            // no mask (prefix_allowed_tokens_fn = None): a row's candidates = its row_cand list
  if (tid < n_rows) {
    // every row fetches its own edge range (one thread walking 40 rows paid 40 dependent global round trips: 35 of the 87 us a step of ONE
    // user took); the range's start stays in LDS for the candidates below
    sh.lse[tid] = lse[sh.lrow[tid]];
    const int nd = sh.node[tid];
    const int e0 = free_mode ? 0 : fsm.row_ptr[nd];
    sh.rp[tid] = e0;
    sh.deg[tid] = free_mode ? n_row_cand : fsm.row_ptr[nd + 1] - e0;
  }
  __syncthreads();
  if (tid == 0) {
    int tot = 0;
    for (int r = 0; r < n_rows; ++r) {
      sh.off[r] = tot;
      const int deg = sh.deg[r];
      bool live = sh.bscore[r] > -INFINITY;
      if (live && deg == 0) sh.status = ATSPEED_ERR_CONSTRAINT;   // HF: "returned an empty list" ValueError
      tot += live ? deg : 0;
    }
    sh.off[n_rows] = tot;
    if (tot > kMaxCand) sh.status = ATSPEED_ERR_CAPACITY;
  }
  __syncthreads();
  const int total = min(sh.off[n_rows], kMaxCand);
  unsigned long long keys[kCPT];
#pragma unroll
  for (int i = 0; i < kCPT; ++i) {
    int c = tid + i * kScanThreads;
    unsigned long long key = 0;
    if (c < total) {
      int lo = 0, hi = n_rows;                        // last r with off[r] <= c
      while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (sh.off[mid] <= c) lo = mid; else hi = mid; }
      int r = lo;
      int tok = free_mode ? row_cand[(size_t)sh.lrow[r] * MAXB + (c - sh.off[r])] : fsm.tok[sh.rp[r] + (c - sh.off[r])];
      if (tok >= 0) {
        float sc = (logits[(size_t)sh.lrow[r] * ld + tok] - sh.lse[r]) + sh.bscore[r];
        uint32_t o = ford(sc);
        uint32_t flat = (uint32_t)sh.brow[r] * (uint32_t)fsm.vocab + (uint32_t)tok;
        if (o > kOrdNegInf || sc != sc) key = ((unsigned long long)o << 32) | (unsigned long long)(~flat);
      }
    }
    keys[i] = key;
  }
  block_topk<kCPT>(keys, k, sh.topk);
}

// The k best columns of one logits row by (value desc, column asc), -inf excluded: with no mask the candidates of a beam are all V
// tokens, and the k best (row, token) pairs of an expand lie among each row's k best tokens (within a row the score is the logit plus a
// constant).  One workgroup per row; chunks of kMaxCand columns, the running best list rides along as a 17th key per thread.
__global__ __launch_bounds__(kScanThreads) void row_topk_kernel(const float* __restrict__ logits, int vocab, int ld, int kk, int32_t* __restrict__ out) {
  __shared__ TopkShared sh;
  __shared__ unsigned long long best[MAXB];
  const int tid = threadIdx.x;
  const float* row = logits + (size_t)blockIdx.x * ld;
  if (tid < MAXB) best[tid] = 0;
  __syncthreads();
  for (int c0 = 0; c0 < vocab; c0 += kMaxCand) {
    unsigned long long keys[kCPT + 1];
#pragma unroll
    for (int i = 0; i < kCPT; ++i) {
      const int col = c0 + tid + i * kScanThreads;
      unsigned long long key = 0;
      if (col < vocab) {
        const float v = row[col];
        const uint32_t o = ford(v);
        if (o > kOrdNegInf || v != v) key = ((unsigned long long)o << 32) | (unsigned long long)(~(uint32_t)col);
      }
      keys[i] = key;
    }
    keys[kCPT] = tid < kk ? best[tid] : 0;
    block_topk<kCPT + 1>(keys, kk, sh);
    if (tid < kk) best[tid] = sh.sel[tid];
    __syncthreads();
  }
  if (tid < MAXB) out[(size_t)blockIdx.x * MAXB + tid] = (tid < kk && best[tid]) ? (int32_t)(~(uint32_t)(best[tid] & 0xffffffffull)) : -1;
}

struct Pick { float score; int parent; int tok; int node; int flat; };

__device__ __forceinline__ Pick decode_pick(unsigned long long key, const FsmDev& fsm, const int* rows_brow,
                                            const int* rows_node, int n_rows) {
  Pick p;
  if (key == 0) { p.score = -INFINITY; p.parent = 0; p.tok = 0; p.node = 0; p.flat = -1; return p; }
  uint32_t flat = ~(uint32_t)(key & 0xffffffffull);
  p.flat = (int)flat;
  p.parent = (int)(flat / (uint32_t)fsm.vocab);
  p.tok = (int)(flat % (uint32_t)fsm.vocab);
  p.score = ford_inv((uint32_t)(key >> 32));
  if (fsm.n_nodes == 0) { p.node = 0; return p; }       // no mask: there is no automaton to walk
  int nd = 0;
  for (int r = 0; r < n_rows; ++r) if (rows_brow[r] == p.parent) { nd = rows_node[r]; break; }
  int e = find_edge(fsm, nd, p.tok);
  p.node = e >= 0 ? fsm.nxt[e] : 0;
  return p;
}

// ---------------------------------------------------------------------------- sampling helpers (do_sample)
// Candidates of an expand kept in registers: slot i of thread t is candidate c = t + i * kScanThreads in expand order
// (rows in order, children of a row's automaton node ascending).  sc = log-softmax / temperature + beam score.
struct CandRegs { int flat[kCPT]; float sc[kCPT]; };

__device__ void expand_candidates(ExpandShared& sh, int n_rows, const float* __restrict__ logits, int ld,
                                  const float* __restrict__ lse, const FsmDev& fsm, float temperature, CandRegs& cr) {
  const int tid = threadIdx.x;
  if (tid < n_rows) {                                 // as in expand_and_select: every row fetches its own edge range
    sh.lse[tid] = lse[sh.lrow[tid]];
    const int nd = sh.node[tid], e0 = fsm.row_ptr[nd];
    sh.rp[tid] = e0;
    sh.deg[tid] = fsm.row_ptr[nd + 1] - e0;
  }
  __syncthreads();
  if (tid == 0) {
    int tot = 0;
    for (int r = 0; r < n_rows; ++r) {
      sh.off[r] = tot;
      const int deg = sh.deg[r];
      bool live = sh.bscore[r] > -INFINITY;
      if (live && deg == 0) sh.status = ATSPEED_ERR_CONSTRAINT;
      tot += live ? deg : 0;
    }
    sh.off[n_rows] = tot;
    if (tot > kMaxCand) sh.status = ATSPEED_ERR_CAPACITY;
  }
  __syncthreads();
  const int total = min(sh.off[n_rows], kMaxCand);
#pragma unroll
  for (int i = 0; i < kCPT; ++i) {
    int c = tid + i * kScanThreads;
    cr.flat[i] = -1; cr.sc[i] = -INFINITY;
    if (c < total) {
      int lo = 0, hi = n_rows;
      while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (sh.off[mid] <= c) lo = mid; else hi = mid; }
      int r = lo;
      int tok = fsm.tok[sh.rp[r] + (c - sh.off[r])];
      cr.sc[i] = (logits[(size_t)sh.lrow[r] * ld + tok] - sh.lse[r]) / temperature + sh.bscore[r];
      cr.flat[i] = sh.brow[r] * fsm.vocab + tok;
    }
  }
}

// log sum exp over every thread's kCPT values (block-wide); -inf entries contribute nothing
__device__ float block_lse(ExpandShared& sh, const float (&v)[kCPT]) {
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < kCPT; ++i) m = fmaxf(m, v[i]);
  float s = 0.f;
  if (m > -INFINITY) {
#pragma unroll
    for (int i = 0; i < kCPT; ++i) s += (v[i] > -INFINITY) ? expf(v[i] - m) : 0.f;
  }
  float wm = wave_max_f32(m);
  s = (m > -INFINITY) ? s * expf(m - wm) : 0.f;
  s = wave_sum_f32(s);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) { sh.red_m[wave] = wm; sh.red_s[wave] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float gm = -INFINITY;
    for (int w = 0; w < kScanWaves; ++w) gm = fmaxf(gm, sh.red_m[w]);
    float gs = 0.f;
    for (int w = 0; w < kScanWaves; ++w) gs += (sh.red_m[w] > -INFINITY) ? sh.red_s[w] * expf(sh.red_m[w] - gm) : 0.f;
    sh.red_out = gm > -INFINITY ? gm + logf(gs) : -INFINITY;
  }
  __syncthreads();
  return sh.red_out;
}
__device__ float block_sum(ExpandShared& sh, float v) {
  v = wave_sum_f32(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sh.red_s[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) { float t = 0.f; for (int w = 0; w < kScanWaves; ++w) t += sh.red_s[w]; sh.red_out = t; }
  __syncthreads();
  return sh.red_out;
}

// n draws without replacement with probability proportional to exp(logw): the n largest logw + Gumbel(hash(flat))
// (Plackett-Luce, the law of torch.multinomial's sequential draws); result in sh.topk.sel, key 0 = none
__device__ void sample_topn(ExpandShared& sh, const CandRegs& cr, const float (&logw)[kCPT], int n, uint32_t sub) {
  unsigned long long keys[kCPT];
#pragma unroll
  for (int i = 0; i < kCPT; ++i) {
    unsigned long long key = 0;
    if (cr.flat[i] >= 0 && logw[i] > -INFINITY) {
      float kf = logw[i] + ats_gumbel(ats_hash_u32((uint32_t)cr.flat[i], sub));
      key = ((unsigned long long)ford(kf) << 32) | (unsigned long long)(~(uint32_t)cr.flat[i]);
    }
    keys[i] = key;
  }
  block_topk<kCPT>(keys, n, sh.topk);
}

// a pick by flat id with the TRUE tempered score (sampling keys carry noise); rows describe the expand that produced it
__device__ Pick pick_of_flat(int flat, const FsmDev& fsm, const ExpandShared& sh, int n_rows, const float* __restrict__ logits,
                             int ld, float temperature) {
  Pick p;
  p.flat = flat;
  if (flat < 0) { p.score = -INFINITY; p.parent = 0; p.tok = 0; p.node = 0; return p; }
  p.parent = flat / fsm.vocab;
  p.tok = flat % fsm.vocab;
  int r = -1;
  for (int q = 0; q < n_rows; ++q) if (sh.brow[q] == p.parent) { r = q; break; }
  if (r < 0) { p.score = -INFINITY; p.node = 0; return p; }
  p.score = (logits[(size_t)sh.lrow[r] * ld + p.tok] - sh.lse[r]) / temperature + sh.bscore[r];
  int e = find_edge(fsm, sh.node[r], p.tok);
  p.node = e >= 0 ? fsm.nxt[e] : 0;
  if (e < 0) p.score = -INFINITY;
  return p;
}

// write a block of k new beams + (optionally) the forward inputs that feed them next
__device__ void emit_block(const Pick& pk, int j, int k, const BeamSet& src, int gen_len, const BeamSet& dst, bool emit,
                           const TokBuf& in, int in_row0, const TokBuf& out, int out_row0, int out_slot0, int W) {
  if (j < k) {
    dst.score[j] = pk.score; dst.parent[j] = pk.parent; dst.tok[j] = pk.tok; dst.flat[j] = pk.flat;
    if (dst.node) dst.node[j] = pk.node;
    if (dst.seq) {
      for (int g = 0; g < gen_len; ++g) dst.seq[j * LMAX + g] = src.seq ? src.seq[pk.parent * LMAX + g] : 0;
      if (gen_len < LMAX) dst.seq[j * LMAX + gen_len] = pk.tok;
    }
    if (emit) {
      out.ids[out_row0 + j] = pk.tok;
      out.pos[out_row0 + j] = in.pos[in_row0 + pk.parent] + 1;
      out.slot[out_row0 + j] = out_slot0 + j;
    }
  }
}
__device__ void emit_vis(const int* parents /*LDS*/, int k, const TokBuf& in, int in_row0, const TokBuf& out,
                         int out_row0, int out_slot0, int W) {
  for (int idx = threadIdx.x; idx < k * W; idx += blockDim.x) {
    int j = idx / W, w = idx - j * W;
    uint64_t bits = in.vis[(size_t)(in_row0 + parents[j]) * W + w];
    int s = out_slot0 + j;
    if ((s >> 6) == w) bits |= 1ull << (s & 63);
    out.vis[(size_t)(out_row0 + j) * W + w] = bits;
  }
}

// ---------------------------------------------------------------------------- beam step
__device__ void beam_step_body(const BeamStepArgs& a);

__global__ __launch_bounds__(kScanThreads) void beam_step_kernel(BeamStepArgs a) { beam_step_body(a); }
__global__ __launch_bounds__(kScanThreads) void beam_step_multi_kernel(const BeamStepArgs* __restrict__ args) {
  __shared__ BeamStepArgs a;                       // one workgroup per user: copy its argument block first
  const int words = sizeof(BeamStepArgs) / 4;
  for (int i = threadIdx.x; i < words; i += blockDim.x)
    reinterpret_cast<uint32_t*>(&a)[i] = reinterpret_cast<const uint32_t*>(args + blockIdx.x)[i];
  __syncthreads();
  beam_step_body(a);
}

__device__ void beam_step_body(const BeamStepArgs& a) {
  __shared__ ExpandShared sh;
  __shared__ int parents[MAXB];
  const int tid = threadIdx.x;
  if (tid == 0) sh.status = 0;
  if (tid < a.n_src) {
    sh.node[tid] = a.src.node ? a.src.node[tid] : 0;
    sh.brow[tid] = tid;
    sh.lrow[tid] = tid;
    sh.bscore[tid] = a.src.score[tid];
  }
  __syncthreads();
  Pick pk;
  if (a.sample) {                                                        // beamSD.py:65-75
    CandRegs cr;
    expand_candidates(sh, a.n_src, a.logits, a.ld, a.lse, a.fsm, a.temperature, cr);
    if (a.tab_score) {                                                   // the draft's whole distribution, for verify
      const float L = block_lse(sh, cr.sc);
      const int total = min(sh.off[a.n_src], kMaxCand);
#pragma unroll
      for (int i = 0; i < kCPT; ++i) { int c = tid + i * kScanThreads; if (c < total) a.tab_score[c] = cr.sc[i]; }
      if (tid <= a.n_src) a.tab_off[tid] = sh.off[tid];
      if (tid == 0) *a.tab_lse = L;
    }
    sample_topn(sh, cr, cr.sc, a.k, a.rng_sub);
    if (tid < a.k) {
      unsigned long long key = sh.topk.sel[tid];
      pk = pick_of_flat(key ? (int)(~(uint32_t)(key & 0xffffffffull)) : -1, a.fsm, sh, a.n_src, a.logits, a.ld, a.temperature);
      parents[tid] = pk.parent;
    }
  } else {
    expand_and_select(sh, a.n_src, a.logits, a.ld, a.lse, a.fsm, a.k, a.row_cand, a.n_row_cand);
    if (tid < a.k) {
      pk = decode_pick(sh.topk.sel[tid], a.fsm, sh.brow, sh.node, a.n_src);
      parents[tid] = pk.parent;
    }
  }
  __syncthreads();
  int slot = tid;
  if (a.filter_ids) {
    // beamSD.py:80-86: picks whose token is neither an item code (>= 32000) nor EOS (2) are dropped AFTER the top-k, so the
    // beam set shrinks instead of lower-ranked candidates moving up; the survivors keep their order (stable compaction),
    // the freed slots hold "no beam" (flat = -1) at the end of the block
    __shared__ int keep[MAXB];
    const bool had = tid < a.k && pk.flat >= 0;
    const bool ok = had && (pk.tok >= a.fsm.filter_min || pk.tok == a.fsm.filter_eos);     // reference: tok >= 32000 | tok == 2
    if (tid < MAXB) keep[tid] = ok ? 1 : 0;
    const int n_had = __syncthreads_count(had);
    if (tid == 0) {
      int total = 0;
      for (int j = 0; j < a.k; ++j) total += keep[j];
      // every pick dropped: the reference goes on with an empty beam set and fails in the next forward (quirk 6); say what happened
      if (n_had > 0 && total == 0) sh.status = ATSPEED_ERR_FILTERED;
    }
    if (tid < a.k) {
      int before = 0, total = 0;
      for (int j = 0; j < a.k; ++j) { total += keep[j]; before += (j < tid) ? keep[j] : 0; }
      slot = ok ? before : total + (tid - before);
      if (!ok) { pk.score = -INFINITY; pk.parent = 0; pk.tok = 0; pk.node = 0; pk.flat = -1; }
    }
    __syncthreads();
    if (tid < a.k) parents[slot] = pk.parent;
    __syncthreads();
  }
  emit_block(pk, slot, a.k, a.src, a.gen_len, a.dst, a.emit != 0, a.in, a.in_row0, a.out, a.out_row0, a.out_slot0, a.vis_words);
  if (a.emit) emit_vis(parents, a.k, a.in, a.in_row0, a.out, a.out_row0, a.out_slot0, a.vis_words);
  if (a.mail) {
    int valid = __syncthreads_count(tid < a.k && pk.flat >= 0);
    if (tid == 0) {
      if (sh.status != 0) a.mail->status = sh.status;
      a.mail->n_valid = valid;
    }
  }
}

// ---------------------------------------------------------------------------- verify
__device__ void verify_walk_body(const VerifyArgs& a);

__device__ void verify_sample_body(const VerifyArgs& a);
__global__ __launch_bounds__(kScanThreads) void verify_walk_kernel(VerifyArgs a) {
  if (a.sample) verify_sample_body(a); else verify_walk_body(a);
}
__global__ __launch_bounds__(kScanThreads) void verify_walk_multi_kernel(const VerifyArgs* __restrict__ args) {
  __shared__ VerifyArgs a;
  const int words = sizeof(VerifyArgs) / 4;
  for (int i = threadIdx.x; i < words; i += blockDim.x)
    reinterpret_cast<uint32_t*>(&a)[i] = reinterpret_cast<const uint32_t*>(args + blockIdx.x)[i];
  __syncthreads();
  if (a.sample) verify_sample_body(a); else verify_walk_body(a);
}

// Sampling verification (beamSD.py:293-321 distributions, :332-369 accept / resample, :303-309 bonus draw), one launch for
// all steps.  Per step i: p = softmax over (hit beams x allowed tokens) of the target's tempered cumulative scores, q the
// draft's (kept by its step kernel: tab_*); draft candidate j is accepted iff u_j q_j <= p_j; with >= K accepted, K of them
// (the K smallest hashes) become the next beams in ascending flat-id order and the walk goes on; otherwise the missing
// beams are drawn from max(p - q, 0) without the accepted ones and the walk stops.  All draws are counter-based
// (ats_rng_sub), so the CPU restatement (oracle/beamsd_sample_ref.py, HashRng) makes the same decisions.
__device__ void verify_sample_body(const VerifyArgs& a) {
  __shared__ ExpandShared sh;
  __shared__ int t_parent[MAXB], hit[MAXB], d_flat[MAXB], d_acc[MAXB], fin_flat[MAXB];
  __shared__ uint32_t d_h[MAXB];
  __shared__ float sbh[MAXB], d_sc[MAXB];
  const int tid = threadIdx.x;
  const int k = a.k, dk = a.dk;
  if (tid == 0) sh.status = 0;
  int nm = 0, n_rows_last = a.nb;
  Pick pk;
  pk.flat = -1; pk.score = -INFINITY; pk.parent = 0; pk.tok = 0; pk.node = 0;
  for (int i = 0; i <= a.dl; ++i) {
    const int n_rows = i == 0 ? a.nb : k;
    n_rows_last = n_rows;
    __syncthreads();
    if (tid < n_rows) {
      int br = i == 0 ? tid : hit[tid];
      sh.brow[tid] = br;
      sh.node[tid] = a.blk[i].node[br];
      sh.lrow[tid] = i == 0 ? tid : a.nb + (i - 1) * dk + br;
      sh.bscore[tid] = i == 0 ? a.blk[0].score[tid] : sbh[tid];
    }
    __syncthreads();
    CandRegs cr;
    expand_candidates(sh, n_rows, a.logits, a.ld, a.lse, a.fsm, a.temperature, cr);
    if (i == a.dl) {                                                     // bonus draw from the target (:303-309)
      sample_topn(sh, cr, cr.sc, k, ats_rng_sub(a.seed, ATS_RNG_BONUS, a.round, i, 0));
      if (tid < k) {
        unsigned long long key = sh.topk.sel[tid];
        fin_flat[tid] = key ? (int)(~(uint32_t)(key & 0xffffffffull)) : -1;
      }
      __syncthreads();
      break;
    }
    const float Lt = block_lse(sh, cr.sc);
    const float Ld = *a.dtab_lse[i];
    // ---- accept test of the draft's candidates (:334-339)
    if (tid < MAXB) { d_flat[tid] = -1; d_acc[tid] = 0; d_sc[tid] = -INFINITY; d_h[tid] = 0; }
    __syncthreads();
    int acc = 0;
    if (tid < dk) {
      const int fl = a.blk[i + 1].flat[tid];
      d_flat[tid] = fl;
      if (fl >= 0) {
        Pick t = pick_of_flat(fl, a.fsm, sh, n_rows, a.logits, a.ld, a.temperature);
        const float p = t.score > -INFINITY ? expf(t.score - Lt) : 0.f;
        const float q = expf(a.blk[i + 1].score[tid] - Ld);
        const float u = ats_u01(ats_hash_u32((uint32_t)tid, ats_rng_sub(a.seed, ATS_RNG_ACCEPT, a.round, i, 0)));
        acc = (u * q <= p) ? 1 : 0;
        d_sc[tid] = t.score;
        d_h[tid] = ats_hash_u32((uint32_t)tid, ats_rng_sub(a.seed, ATS_RNG_PERM, a.round, i, 0));
      }
      d_acc[tid] = acc;
    }
    const int n_acc = __syncthreads_count(acc);
    if (n_acc >= k) {                                                    // :341-350
      int chosen = 0;
      if (tid < dk && acc) {
        int rank = 0;
        for (int j = 0; j < dk; ++j) rank += (d_acc[j] && (d_h[j] < d_h[tid] || (d_h[j] == d_h[tid] && j < tid))) ? 1 : 0;
        chosen = rank < k;
      }
      __syncthreads();
      if (tid < dk) d_acc[tid] = chosen;
      __syncthreads();
      if (tid < dk && chosen) {
        int rf = 0;
        for (int j = 0; j < dk; ++j) rf += (d_acc[j] && d_flat[j] < d_flat[tid]) ? 1 : 0;
        hit[rf] = tid;                 // position in the draft's block i+1
        sbh[rf] = d_sc[tid];
        fin_flat[rf] = d_flat[tid];
      }
      __syncthreads();
      nm += 1;
      continue;
    }
    // ---- rejected: the missing k - n_acc beams from the residual max(p - q, 0) without the accepted ones (:351-369)
    float logw[kCPT], pr[kCPT];
    float part = 0.f;
    const float* dts = a.dtab_score[i];
    const int* dto = a.dtab_off[i];
#pragma unroll
    for (int s2 = 0; s2 < kCPT; ++s2) {
      logw[s2] = -INFINITY; pr[s2] = 0.f;
      const int fl = cr.flat[s2];
      if (fl >= 0 && cr.sc[s2] > -INFINITY) {
        const int c = tid + s2 * kScanThreads;
        int lo = 0, hi = n_rows;
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (sh.off[mid] <= c) lo = mid; else hi = mid; }
        const float q = expf(dts[dto[sh.brow[lo]] + (c - sh.off[lo])] - Ld);   // same node -> same children order in the draft's table
        const float p = expf(cr.sc[s2] - Lt);
        bool taken = false;
        for (int j = 0; j < dk; ++j) taken |= (d_acc[j] && d_flat[j] == fl);
        if (!taken) {
          pr[s2] = p;
          const float rsd = fmaxf(p - q, 0.f);
          if (rsd > 0.f) { logw[s2] = logf(rsd); part += rsd; }
        }
      }
    }
    const float tot = block_sum(sh, part);
    if (tot == 0.f) {
#pragma unroll
      for (int s2 = 0; s2 < kCPT; ++s2) logw[s2] = pr[s2] > 0.f ? logf(pr[s2]) : -INFINITY;
    }
    sample_topn(sh, cr, logw, k - n_acc, ats_rng_sub(a.seed, ATS_RNG_RESID, a.round, i, 0));
    // final set = accepted ++ drawn, ascending flat id
    if (tid < dk && acc) {
      int r0 = 0;
      for (int j = 0; j < tid; ++j) r0 += d_acc[j];
      hit[r0] = d_flat[tid];                                           // scratch: unsorted flats
    }
    if (tid < k - n_acc) {
      unsigned long long key = sh.topk.sel[tid];
      hit[n_acc + tid] = key ? (int)(~(uint32_t)(key & 0xffffffffull)) : 0x7fffffff;
    }
    __syncthreads();
    if (tid < k) {
      const int fl = hit[tid];
      int rf = 0;
      for (int j = 0; j < k; ++j) rf += (hit[j] < fl || (hit[j] == fl && j < tid)) ? 1 : 0;
      fin_flat[rf] = fl == 0x7fffffff ? -1 : fl;
    }
    __syncthreads();
    break;
  }
  // picks of the final set, in the row space of block nm (sh.* still describe the expand of the last step)
  if (tid < k) {
    pk = pick_of_flat(fin_flat[tid], a.fsm, sh, n_rows_last, a.logits, a.ld, a.temperature);
    t_parent[tid] = pk.parent;
  }
  __syncthreads();
  const int cnt = nm == 0 ? a.nb : dk;
  const int rowbase = nm == 0 ? a.n0 - a.nb : a.n0 + (nm - 1) * dk;
  const int base_next = a.cur.slot[rowbase] + cnt;
  const int W = a.vis_words;
  emit_block(pk, tid, k, a.blk[nm], a.gen_len0 + nm, a.res, true, a.cur, rowbase, a.next, 0, base_next, W);
  emit_vis(t_parent, k, a.cur, rowbase, a.next, 0, base_next, W);
  if (nm == a.dl && a.dl > 0) {
    for (int j = tid; j < dk; j += blockDim.x) {
      a.dnext.ids[j] = a.cur.ids[rowbase + j];
      a.dnext.pos[j] = a.cur.pos[rowbase + j];
      a.dnext.slot[j] = a.cur.slot[rowbase + j];
    }
    for (int idx = tid; idx < dk * W; idx += blockDim.x) a.dnext.vis[idx] = a.cur.vis[(size_t)rowbase * W + idx];
    emit_block(pk, tid, k, a.blk[nm], a.gen_len0 + nm, a.res, true, a.cur, rowbase, a.dnext, dk, base_next, W);
    emit_vis(t_parent, k, a.cur, rowbase, a.dnext, dk, base_next, W);
  }
  int valid = __syncthreads_count(tid < k && pk.flat >= 0);
  if (tid == 0) {
    a.mail->n_matches = nm;
    a.mail->n_valid = valid;
    if (sh.status != 0) a.mail->status = sh.status;
  }
}

__device__ void verify_walk_body(const VerifyArgs& a) {
  __shared__ ExpandShared sh;
  __shared__ int t_parent[MAXB], t_pos[MAXB], hit[MAXB];
  __shared__ float sbh[MAXB];
  __shared__ int reject;
  const int tid = threadIdx.x;
  const int k = a.k, dk = a.dk;
  if (tid == 0) { sh.status = 0; reject = 0; }
  int nm = 0;
  Pick pk;
  pk.flat = -1; pk.score = -INFINITY; pk.parent = 0; pk.tok = 0; pk.node = 0;
  for (int i = 0; i <= a.dl; ++i) {
    const int n_rows = i == 0 ? a.nb : k;
    __syncthreads();
    if (tid < n_rows) {                                                  // beamSD.py:279-296
      int br = i == 0 ? tid : hit[tid];
      sh.brow[tid] = br;
      sh.node[tid] = a.blk[i].node[br];
      sh.lrow[tid] = i == 0 ? tid : a.nb + (i - 1) * dk + br;
      sh.bscore[tid] = i == 0 ? a.blk[0].score[tid] : sbh[tid];
    }
    __syncthreads();
    expand_and_select(sh, n_rows, a.logits, a.ld, a.lse, a.fsm, k, a.row_cand, a.n_row_cand);     // :297-298,323-328
    if (tid < k) {
      pk = decode_pick(sh.topk.sel[tid], a.fsm, sh.brow, sh.node, n_rows);
      t_parent[tid] = pk.parent;
      if (a.vtrace) {                                                    // decision trace: what the target chose at step i
        int32_t* vt = a.vtrace + (size_t)i * 3 * MAXB;
        vt[tid] = (int32_t)__float_as_uint(pk.score); vt[MAXB + tid] = pk.parent; vt[2 * MAXB + tid] = pk.flat >= 0 ? pk.tok : -1;
      }
    }
    if (i == a.dl) break;                                                // :329-330 (uniform)
    // acceptance: every target id must be among the draft's ids of step i+1 (:371-380)
    int pos = -1;
    if (tid < k && pk.flat >= 0) {
      const int* df = a.blk[i + 1].flat;
      for (int d = 0; d < dk; ++d) if (df[d] == pk.flat) { pos = d; break; }
    }
    if (tid < k) { t_pos[tid] = pos; if (pos < 0) atomicOr(&reject, 1); }
    __syncthreads();
    if (reject) break;                                                   // uniform
    if (tid < k) {                                                       // :373-376: order by draft position
      int rank = 0;
      for (int j = 0; j < k; ++j) rank += (t_pos[j] < pos) ? 1 : 0;
      hit[rank] = pos;
      sbh[rank] = pk.score;
    }
    nm += 1;
  }
  __syncthreads();
  // ---- outputs: new round beams, next-round inputs (:383-416)
  const int cnt = nm == 0 ? a.nb : dk;
  const int rowbase = nm == 0 ? a.n0 - a.nb : a.n0 + (nm - 1) * dk;
  const int base_next = a.cur.slot[rowbase] + cnt;
  const int W = a.vis_words;
  emit_block(pk, tid, k, a.blk[nm], a.gen_len0 + nm, a.res, true, a.cur, rowbase, a.next, 0, base_next, W);
  emit_vis(t_parent, k, a.cur, rowbase, a.next, 0, base_next, W);
  if (nm == a.dl && a.dl > 0) {
    // the draft has not seen its own last block: next draft input = that block ++ the new beams (:402-416)
    for (int j = tid; j < dk; j += blockDim.x) {
      a.dnext.ids[j] = a.cur.ids[rowbase + j];
      a.dnext.pos[j] = a.cur.pos[rowbase + j];
      a.dnext.slot[j] = a.cur.slot[rowbase + j];
    }
    for (int idx = tid; idx < dk * W; idx += blockDim.x) a.dnext.vis[idx] = a.cur.vis[(size_t)rowbase * W + idx];
    emit_block(pk, tid, k, a.blk[nm], a.gen_len0 + nm, a.res, true, a.cur, rowbase, a.dnext, dk, base_next, W);
    emit_vis(t_parent, k, a.cur, rowbase, a.dnext, dk, base_next, W);
  }
  int valid = __syncthreads_count(tid < k && pk.flat >= 0);
  if (tid == 0) {
    a.mail->n_matches = nm;
    a.mail->n_valid = valid;
    if (sh.status != 0) a.mail->status = sh.status;
  }
}

// ---------------------------------------------------------------------------- prompt init / export / accept
__device__ __forceinline__ void init_prompt_row(const TokBuf& tb, const int32_t* __restrict__ prompt, int t, int W, int vocab) {
  int id = prompt[t];
  tb.ids[t] = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  tb.pos[t] = t;
  tb.slot[t] = t;
  for (int w = 0; w < W; ++w) {                         // causal prefix: bits [0, t]
    int lo = w * 64;
    uint64_t bits = (t >= lo + 63) ? ~0ull : (t < lo ? 0ull : ((~0ull) >> (63 - (t - lo))));
    tb.vis[(size_t)t * W + w] = bits;
  }
}
__global__ void init_prompt_kernel(TokBuf tb, const int32_t* __restrict__ prompt, int P, int W, BeamSet beams,
                                   int start_node, int vocab, Mailbox* mail) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < P) init_prompt_row(tb, prompt, t, W, vocab);
  if (t == 0) {
    beams.score[0] = 0.f; beams.node[0] = start_node; beams.parent[0] = 0; beams.tok[0] = 0; beams.flat[0] = 0;
    mail->n_matches = 0; mail->status = 0; mail->n_valid = 1; mail->pad = 0;
  }
}

// every user of a lock-step batch in ONE launch (grid.y = user): 256 launches per batch less than the per-user form
__global__ void init_prompt_multi_kernel(const InitPromptArgs* __restrict__ args, int W, int vocab) {
  const InitPromptArgs a = args[blockIdx.y];
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < a.prompt_len) init_prompt_row(a.tb, a.prompt, t, W, vocab);
  if (t == 0) {
    a.beams.score[0] = 0.f; a.beams.node[0] = a.start_node; a.beams.parent[0] = 0; a.beams.tok[0] = 0; a.beams.flat[0] = 0;
    a.mail->n_matches = 0; a.mail->status = 0; a.mail->n_valid = 1; a.mail->pad = 0;
  }
}

__device__ __forceinline__ void export_beams_row(const BeamSet& b, int j, int k, int max_new, int32_t* out_tokens, float* out_scores, int sort_desc) {
  int dst = j;
  if (sort_desc) {                                     // beamSD.py:529-531: stable sort by score, best first
    const float sj = b.score[j];
    dst = 0;
    for (int i = 0; i < k; ++i) { const float si = b.score[i]; dst += (si > sj || (si == sj && i < j)) ? 1 : 0; }
  }
  out_scores[dst] = b.score[j];
  for (int g = 0; g < max_new; ++g) out_tokens[dst * max_new + g] = b.seq[j * LMAX + g];
}
__global__ void export_beams_multi_kernel(const ExportBeamsArgs* __restrict__ args, int k, int max_new) {
  const ExportBeamsArgs a = args[blockIdx.x];
  if ((int)threadIdx.x < k) export_beams_row(a.b, threadIdx.x, k, max_new, a.out_tokens, a.out_scores, a.sort_desc);
}

// beam_sequence of every user of a batch (beamSD.py:87,383: prompt ++ generated suffix per beam, int64) in one launch:
// block (beam j, user u) writes row j of user u
__global__ void assemble_sequences_kernel(const int32_t* __restrict__ prompts, const int64_t* __restrict__ off, const int32_t* __restrict__ toks,
                                          int k, int L, int64_t* __restrict__ out) {
  const int u = blockIdx.y, j = blockIdx.x;
  const int64_t p0 = off[u];
  const int P = (int)(off[u + 1] - p0);
  int64_t* row = out + (int64_t)k * (p0 + (int64_t)u * L) + (int64_t)j * (P + L);
  for (int t = threadIdx.x; t < P; t += blockDim.x) row[t] = prompts[p0 + t];
  for (int t = threadIdx.x; t < L; t += blockDim.x) row[P + t] = toks[((size_t)u * k + j) * L + t];
}

__global__ void export_beams_kernel(BeamSet b, int k, int max_new, int32_t* out_tokens, float* out_scores, int sort_desc) {
  int j = threadIdx.x;
  if (j < k) export_beams_row(b, j, k, max_new, out_tokens, out_scores, sort_desc);
}

// out[r][c] = logits[r][c] - lse[r]: the log-softmax rows a host-side logits processor receives (beamSD.py:58 -> :62-64)
__global__ void log_softmax_rows_kernel(const float* __restrict__ logits, int ld, const float* __restrict__ lse, int vocab, float* __restrict__ out, int ld_out) {
  const float l = lse[blockIdx.y];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < vocab) out[(size_t)blockIdx.y * ld_out + c] = logits[(size_t)blockIdx.y * ld + c] - l;
}

__global__ void accept_kernel(const int32_t* __restrict__ tflat, const float* __restrict__ tscore, int k,
                              const int32_t* __restrict__ dflat, int dk, int32_t* hit, float* sbh, int32_t* accept) {
  __shared__ int t_pos[MAXB];
  int j = threadIdx.x;
  int pos = -1;
  if (j < k && tflat[j] >= 0)
    for (int d = 0; d < dk; ++d) if (dflat[d] == tflat[j]) { pos = d; break; }
  if (j < k) t_pos[j] = pos;
  int bad = __syncthreads_count(j < k && pos < 0);
  if (j == 0) *accept = bad == 0;
  if (j < k) { hit[j] = -1; sbh[j] = -INFINITY; }
  __syncthreads();
  if (bad == 0 && j < k) {
    int rank = 0;
    for (int i = 0; i < k; ++i) rank += t_pos[i] < pos ? 1 : 0;
    hit[rank] = pos;
    sbh[rank] = tscore[j];
  }
}

}  // namespace

int ats_lse_rows(const float* logits, int n_rows, int vocab, int ld, float* lse, hipStream_t st) {
  if (n_rows <= 0) return ATSPEED_OK;
  ATS_REQUIRE((ld & 3) == 0 && ((uintptr_t)logits & 15) == 0, ATSPEED_ERR_INVALID, "lse: rows must be 16-byte aligned (ld %d)", ld);
  // measured (tools/lse_bench.py, 30976 rows of 32859): 1024 threads 6.2 TB/s, 512: 6.5, 256 (four double-buffered chunks per row, eight
  // workgroups per CU): 6.9 TB/s = 98 % of the measured read peak; a single user's 121 rows are launch-bound (6.0 us with 512 threads, 6.6 with 256)
  const int nt = n_rows >= 1024 ? 256 : 512;
  if (nt == 256) lse_rows_kernel<256><<<n_rows, 256, 0, st>>>(logits, vocab, ld, lse);
  else           lse_rows_kernel<512><<<n_rows, 512, 0, st>>>(logits, vocab, ld, lse);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

int ats_row_topk(const float* logits, int n_rows, int vocab, int ld, int kk, int32_t* out, hipStream_t st) {
  if (n_rows <= 0) return ATSPEED_OK;
  ATS_REQUIRE(kk >= 1 && kk <= MAXB, ATSPEED_ERR_CAPACITY, "row_topk: k=%d out of [1,%d]", kk, MAXB);
  row_topk_kernel<<<n_rows, kScanThreads, 0, st>>>(logits, vocab, ld, kk, out);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

int ats_beam_step(const BeamStepArgs& a, hipStream_t st) {
  ATS_REQUIRE(a.k >= 1 && a.k <= MAXB && a.n_src >= 1 && a.n_src <= MAXB, ATSPEED_ERR_CAPACITY,
              "beam step: k=%d / rows=%d exceed %d", a.k, a.n_src, MAXB);
  beam_step_kernel<<<1, kScanThreads, 0, st>>>(a);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

int ats_beam_step_multi(const BeamStepArgs* dev_args, int n, hipStream_t st) {
  if (n <= 0) return ATSPEED_OK;
  beam_step_multi_kernel<<<n, kScanThreads, 0, st>>>(dev_args);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

int ats_verify_walk_multi(const VerifyArgs* dev_args, int n, hipStream_t st) {
  if (n <= 0) return ATSPEED_OK;
  verify_walk_multi_kernel<<<n, kScanThreads, 0, st>>>(dev_args);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

int ats_verify_walk(const VerifyArgs& a, hipStream_t st) {
  ATS_REQUIRE(a.k >= 1 && a.k <= MAXB && a.dk >= a.k && a.dk <= MAXB && a.dl >= 1 && a.dl <= ATSPEED_MAX_GAMMA,
              ATSPEED_ERR_CAPACITY, "verify: k=%d dk=%d dl=%d out of range", a.k, a.dk, a.dl);
  verify_walk_kernel<<<1, kScanThreads, 0, st>>>(a);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

int ats_init_prompt(TokBuf tb, const int32_t* prompt, int prompt_len, int vis_words, BeamSet beams, int start_node,
                    int vocab, Mailbox* mail, hipStream_t st) {
  init_prompt_kernel<<<(prompt_len + 255) / 256, 256, 0, st>>>(tb, prompt, prompt_len, vis_words, beams, start_node, vocab, mail);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

int ats_init_prompt_multi(const InitPromptArgs* dev_args, int n, int max_prompt_len, int vis_words, int vocab, hipStream_t st) {
  if (n <= 0) return ATSPEED_OK;
  init_prompt_multi_kernel<<<dim3((max_prompt_len + 255) / 256, n), 256, 0, st>>>(dev_args, vis_words, vocab);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

int ats_export_beams_multi(const ExportBeamsArgs* dev_args, int n, int k, int max_new, hipStream_t st) {
  if (n <= 0) return ATSPEED_OK;
  export_beams_multi_kernel<<<n, 64, 0, st>>>(dev_args, k, max_new);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

extern "C" int atspeed_assemble_sequences(const int32_t* prompts_flat, const int64_t* prompt_off_host, const int32_t* toks, int32_t n, int32_t k,
                                          int32_t new_tokens, int64_t* out, void* stream) {
  ATS_REQUIRE(prompts_flat && prompt_off_host && toks && out && n >= 1 && k >= 1 && k <= 65535 && new_tokens >= 0, ATSPEED_ERR_INVALID,
              "assemble_sequences: bad arguments");
  ATS_REQUIRE(n <= 65535, ATSPEED_ERR_CAPACITY, "assemble_sequences: %d users per call (max 65535)", n);
  for (int u = 0; u < n; ++u)
    ATS_REQUIRE(prompt_off_host[u + 1] >= prompt_off_host[u], ATSPEED_ERR_INVALID, "assemble_sequences: offsets not monotone at user %d", u);
  const void* off_dev = nullptr;
  ATS_TRY(ats_stage(prompt_off_host, (size_t)(n + 1) * sizeof(int64_t), &off_dev, (hipStream_t)stream));
  assemble_sequences_kernel<<<dim3(k, n), 128, 0, (hipStream_t)stream>>>(prompts_flat, (const int64_t*)off_dev, toks, k, new_tokens, out);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

int ats_export_beams(BeamSet b, int k, int max_new, int32_t* out_tokens, float* out_scores, hipStream_t st, bool sort_desc) {
  export_beams_kernel<<<1, 64, 0, st>>>(b, k, max_new, out_tokens, out_scores, sort_desc ? 1 : 0);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

int ats_accept(const int32_t* target_flat, const float* target_score, int k, const int32_t* draft_flat, int dk,
               int32_t* hit, float* score_by_hit, int32_t* accept, hipStream_t st) {
  ATS_REQUIRE(k >= 1 && k <= MAXB && dk >= 1 && dk <= MAXB, ATSPEED_ERR_CAPACITY, "accept: k=%d dk=%d out of range", k, dk);
  accept_kernel<<<1, 64, 0, st>>>(target_flat, target_score, k, draft_flat, dk, hit, score_by_hit, accept);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

extern "C" int atspeed_lse_rows(const float* logits, int32_t n_rows, int32_t vocab, int32_t ld, float* lse, void* stream) {
  ATS_REQUIRE(logits && lse && vocab > 0 && ld >= vocab, ATSPEED_ERR_INVALID, "lse: bad arguments");
  return ats_lse_rows(logits, n_rows, vocab, ld, lse, (hipStream_t)stream);
}

extern "C" int atspeed_log_softmax_rows(const float* logits, int32_t ld, const float* lse, int32_t n_rows, int32_t vocab, float* out, int32_t ld_out,
                                        void* stream) {
  ATS_REQUIRE(logits && lse && out && vocab > 0 && ld >= vocab && ld_out >= vocab && n_rows >= 0 && n_rows <= 65535, ATSPEED_ERR_INVALID,
              "log_softmax_rows: bad arguments");
  if (n_rows == 0) return ATSPEED_OK;
  log_softmax_rows_kernel<<<dim3((vocab + 255) / 256, n_rows), 256, 0, (hipStream_t)stream>>>(logits, ld, lse, vocab, out, ld_out);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

extern "C" int atspeed_accept(const int32_t* target_flat, const float* target_score, int32_t k, const int32_t* draft_flat,
                              int32_t dk, int32_t* hit, float* score_by_hit, int32_t* accept, void* stream) {
  ATS_REQUIRE(target_flat && target_score && draft_flat && hit && score_by_hit && accept, ATSPEED_ERR_INVALID, "accept: null argument");
  return ats_accept(target_flat, target_score, k, draft_flat, dk, hit, score_by_hit, accept, (hipStream_t)stream);
}

extern "C" int atspeed_beam_expand_prune(const float* logits, int32_t ld, const float* lse, const float* beam_score,
                                         const int32_t* beam_node, int32_t n_rows, const atspeed_fsm* fsm, int32_t k,
                                         float* out_score, int32_t* out_parent, int32_t* out_token, int32_t* out_node,
                                         int32_t* out_flat, void* stream) {
  ATS_REQUIRE(logits && lse && beam_score && beam_node && fsm && out_score && out_parent && out_token && out_node && out_flat,
              ATSPEED_ERR_INVALID, "beam_expand_prune: null argument");
  BeamStepArgs a{};
  a.src.score = const_cast<float*>(beam_score);
  a.src.node = const_cast<int32_t*>(beam_node);
  a.src.seq = nullptr;
  a.n_src = n_rows; a.gen_len = 0;
  a.logits = logits; a.ld = ld; a.lse = lse;
  a.fsm = fsm->dev;
  a.k = k;
  a.dst.score = out_score; a.dst.parent = out_parent; a.dst.tok = out_token; a.dst.node = out_node; a.dst.flat = out_flat;
  a.dst.seq = nullptr;
  a.emit = 0; a.mail = nullptr; a.vis_words = 0;
  ATS_REQUIRE(fsm->dev.n_nodes > 0, ATSPEED_ERR_INVALID, "beam_expand_prune: a mask-free automaton needs atspeed_beam_expand_prune_free");
  return ats_beam_step(a, (hipStream_t)stream);
}

extern "C" int atspeed_row_topk(const float* scores, int32_t n_rows, int32_t vocab, int32_t ld, int32_t k, int32_t* out_tokens, void* stream) {
  ATS_REQUIRE(scores && out_tokens && vocab > 0 && ld >= vocab && n_rows >= 0, ATSPEED_ERR_INVALID, "row_topk: bad arguments");
  return ats_row_topk(scores, n_rows, vocab, ld, k, out_tokens, (hipStream_t)stream);
}

extern "C" int atspeed_beam_expand_prune_free(const float* logits, int32_t ld, const float* lse, const float* beam_score, int32_t n_rows,
                                              int32_t vocab, int32_t k, int32_t* row_cand_ws, float* out_score, int32_t* out_parent,
                                              int32_t* out_token, int32_t* out_flat, void* stream) {
  ATS_REQUIRE(logits && lse && beam_score && row_cand_ws && out_score && out_parent && out_token && out_flat, ATSPEED_ERR_INVALID,
              "beam_expand_prune_free: null argument");
  ATS_REQUIRE(vocab > 0 && ld >= vocab && (int64_t)MAXB * vocab < (int64_t)0x7fffffff, ATSPEED_ERR_INVALID, "beam_expand_prune_free: bad vocabulary");
  ATS_REQUIRE(n_rows >= 1 && n_rows <= MAXB && k >= 1 && k <= MAXB, ATSPEED_ERR_CAPACITY, "beam_expand_prune_free: rows=%d / k=%d out of [1,%d] (row_cand_ws holds rows x %d ids)",
              n_rows, k, MAXB, MAXB);
  ATS_TRY(ats_row_topk(logits, n_rows, vocab, ld, k, row_cand_ws, (hipStream_t)stream));
  BeamStepArgs a{};
  a.src.score = const_cast<float*>(beam_score);
  a.src.node = nullptr; a.src.seq = nullptr;
  a.n_src = n_rows; a.gen_len = 0;
  a.logits = logits; a.ld = ld; a.lse = lse;
  a.fsm = FsmDev{nullptr, nullptr, nullptr, 0, 0, vocab, 0, -1};
  a.k = k;
  a.dst.score = out_score; a.dst.parent = out_parent; a.dst.tok = out_token; a.dst.node = nullptr; a.dst.flat = out_flat;
  a.dst.seq = nullptr;
  a.row_cand = row_cand_ws; a.n_row_cand = k;
  a.emit = 0; a.mail = nullptr; a.vis_words = 0;
  return ats_beam_step(a, (hipStream_t)stream);
}
