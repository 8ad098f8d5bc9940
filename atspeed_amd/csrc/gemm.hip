// Projection GEMMs of the draft / target forward on the CDNA4 matrix cores.
//
//   C[M,N] (+)= A[M,K] * W[N,K]^T      A, W row-major with K contiguous (HF nn.Linear layout)
//
// Two regimes, two kernels (dispatch in ats_gemm / big_kernel_applies):
//   * gemm_kernel<T,...>  — M = one user's tokens (tens to a few hundred): every launch is one pass over W and is
//     bound by that HBM stream.  256 threads = 4 waves, tile BM x {64,128} x (128 bytes of K), LDS double buffered,
//     register-staged 16-byte loads one tile ahead; 16-byte chunk c of LDS row r lives at chunk c ^ (r & 7);
//     bf16: v_mfma_f32_16x16x32_bf16, fp32 parity mode: v_mfma_f32_16x16x4_f32 with a k permutation that lets a
//     lane take its four k-steps from one chunk; split-K into fp32 slabs sized for >= 2-3 workgroups per CU, reduced
//     by a second kernel that also applies the epilogue (and, for the residual projections, the next RMSNorm).
//   * gemm_ring_kernel<EPI,MT2,FP8> — M = tokens of all users of a lock-step batch (thousands): MFMA-bound.  256 x
//     {256,128} tile, 8 waves, both operands by LDS-DMA into a four-stage ring of 32-deep k-steps, hand-placed inner
//     loop (see the kernel's comment); the same kernel on e4m3 operands with per-row scales (W8A8).
// Epilogues: store (dtype), fp32 store (logits), residual add, SwiGLU over interleaved gate/up 16-row groups.
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <set>

#include "internal.h"

namespace ATS_NS {

namespace {

static int env_int(const char* name, int dflt);
// share of the last wave of 256 CUs a grid of `tiles` workgroups keeps busy, in percent
static int big_fill_pct(int tiles) { return tiles * 100 / (((tiles + 255) / 256) * 256); }
// Tile height of the 256-wide kernels from a cost model fitted to tools/gemm_ab.py sweeps (MI355X, ring kernel): time ~
// whole rounds of 256 workgroups + a partial round that costs 55 % of a round plus 45 % of its fill (the emptier chip
// clocks higher), and a 128-row tile costs 0.65 of a 256-row one (half the flops, 1.5x the operand bytes per flop).
static float big_rounds(int tiles) {
  const int full = tiles / 256, rem = tiles % 256;
  return (float)full + (rem ? 0.55f + 0.45f * (float)rem / 256.f : 0.f);
}
static bool big_use_256_rows(int t256, int t128) { return big_rounds(t256) <= 0.65f * big_rounds(t128); }

// ---- split-K tail of the ring kernel (gemm_ring_kernel<..., SK = true>): plan and cost model, host side.
// The kernel accepts any even deal of the tail's k-units over G workgroups; the plans made here are the ALIGNED ones, G = R x S: each of the
// R tail tiles is cut into S equal k-ranges, one workgroup each, one seam per tile.  Unaligned deals (a workgroup finishing one tile and
// starting the next) were measured first and lost: every part publishes a whole fp32 tile (256 / 128 KB) whatever its share of K, so the
// seam traffic -- R x S slots written, R x (S - 1) read back -- must stay small against the k-loop it shortens (gate_up at 912 tokens, 88 tail
// tiles over 256 workgroups in 3-4 parts each: 160 MB of slots next to 200 MB of operands, no gain; down at 2400 tokens 189 -> 293 us).
// Cost model, fitted to tools/sk_sweep.py on MI355X (profiles/r04_sk_sweep.txt): a tile's time grows with the number of workgroups running
// beside it (clock, L2, fabric) -- 256 x 256: k x (14.5 + 7.5 c / 256) ns, 256 x 128: k x (7 + 7.5 c / 256) ns for c concurrent workgroups
// (82 us for a 256-row tile at K = 4096 and 172 tiles, 94 at 240; 42 / 50 / 64 us for a 128-row one at 160 / 192 / 240) -- and a seam costs
// 22 / 38 / 46 us (256 KB slots) or 14 / 24 / 27 us (128 KB slots) for 2 / 3 / 4 parts: second prologue, dump, ticket, the finisher's reads.
struct SkPlan { bool on; int n_dp, G, U, TU, parts; };
static SkPlan sk_plan_s(int tiles, int k, int S) {
  SkPlan p{false, 0, 0, 0, 0, 1};
  p.U = k / 128;
  p.n_dp = tiles / 256 * 256;
  const int R = tiles - p.n_dp;
  if (S < 2 || R == 0 || R * S > 256 || p.U < 4 * S) return p;    // >= 4 units (512 k) per part
  p.TU = R * p.U;
  p.G = R * S;
  p.parts = S;
  p.on = true;
  // test hook (tests/test_kernels_gpu.py, atspeed_set_switch("gemm_sk_g", G); no environment variable): an UNALIGNED deal of the tail's k-units over G
  // workgroups -- a workgroup then ends one tile and starts the next, the kernel's segment loop and second slot, which no plan of the launcher uses
  // (they lost to the aligned plans, see above) but the kernel keeps
  const int g_env = ats_switch(ATS_SW_GEMM_SK_G);
  if (g_env >= R && g_env <= 256 && g_env <= p.TU) p.G = g_env;     // (G >= R: a workgroup's range never spans more than two tiles)
  return p;
}
static float ring_tile_us(int k, bool rows256, int concurrent) {
  return 1e-3f * (float)k * ((rows256 ? 14.5f : 7.f) + 7.5f * (float)concurrent / 256.f);
}
static float sk_cost_us(int tiles, int k, bool rows256, SkPlan* plan_out) {
  const int mode = ats_switch(ATS_SW_GEMM_SK);                     // 0: off, 1: cost model, 2-4: that many parts wherever they fit
  const int full = tiles / 256, R = tiles % 256;
  const float whole = (float)full * ring_tile_us(k, rows256, 256);
  if (plan_out) *plan_out = SkPlan{false, 0, 0, 0, 0, 1};
  if (R == 0) return whole;
  float best = whole + ring_tile_us(k, rows256, R);               // the partly filled last round as it is
  if (mode == 0) return best;
  static const float seam256[5] = {0.f, 0.f, 22.f, 38.f, 46.f}, seam128[5] = {0.f, 0.f, 14.f, 24.f, 27.f};
  for (int S = 2; S <= 4; ++S) {
    if (mode >= 2 && S != mode) continue;
    const SkPlan p = sk_plan_s(tiles, k, S);
    if (!p.on) continue;
    const float c = whole + ring_tile_us(k, rows256, p.G) / (float)S + (rows256 ? seam256[S] : seam128[S]);
    if (c < best || mode >= 2) { best = c; if (plan_out) *plan_out = p; }
  }
  return best;
}
// tile height and split decision of a ring-kernel launch
struct BigChoice { bool rows256; SkPlan sk; };
static BigChoice big_choose(int t256, int t128, int k) {
  const int force_mt = ats_switch(ATS_SW_GEMM_FORCE_MT);           // tuning: 8 / 4 = always 256- / 128-row token tiles
  BigChoice c;
  SkPlan p256, p128;
  const float c256 = sk_cost_us(t256, k, true, &p256), c128 = sk_cost_us(t128, k, false, &p128);
  c.rows256 = force_mt ? force_mt == 8 : c256 <= c128;
  if (!force_mt && ats_switch(ATS_SW_GEMM_SK) >= 2 && p256.on != p128.on) c.rows256 = p256.on;   // forced tail (tests, sweeps): the height that has a plan
  c.sk = c.rows256 ? p256 : p128;
  return c;
}
static bool sk_any_plan(int tiles, int k) { SkPlan p; sk_cost_us(tiles, k, false, &p); return p.on; }

constexpr int kThreads = 256;
constexpr int kRowBytes = 128;      // bytes of K per LDS row
constexpr int kChunks = 8;          // 16-byte chunks per row

template <typename T> struct GemmTraits;
template <> struct GemmTraits<bf16_t> { static constexpr int BK = 64; static constexpr int EPC = 8; };
template <> struct GemmTraits<float>  { static constexpr int BK = 32; static constexpr int EPC = 4; };

__device__ __forceinline__ int swz(int row, int chunk) { return (row * kChunks + (chunk ^ (row & 7))) * 16; }
template <int CH> __device__ __forceinline__ int swz(int row, int chunk) { return (row * CH + (chunk ^ (row & (CH - 1)))) * 16; }

template <typename T, int BM, int BN, int WM, int WN>
struct TileCfg {
  static constexpr int MI = BM / WM / 16;
  static constexpr int NI = BN / WN / 16;
  static constexpr int A_CHUNKS = BM * kChunks;
  static constexpr int W_CHUNKS = BN * kChunks;
  static constexpr int A_PER_THREAD = (A_CHUNKS + kThreads - 1) / kThreads;
  static constexpr int W_PER_THREAD = (W_CHUNKS + kThreads - 1) / kThreads;
  static constexpr int STAGE_BYTES = (BM + BN) * kRowBytes;
};

template <typename T, int BM, int BN, int WM, int WN, int EPI, bool SPLIT>
__global__ __launch_bounds__(kThreads) void gemm_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                        void* __restrict__ Cv, int M, int N, int K, int lda, int ldc,
                                                        int k_per_split, float* __restrict__ partial, int pk) {
  // pk: A, W (and the SwiGLU output, itself the next projection's operand) are in the packed operand layout (common.h)
  using Cfg = TileCfg<T, BM, BN, WM, WN>;
  constexpr int BK = GemmTraits<T>::BK;
  constexpr int EPC = GemmTraits<T>::EPC;
  constexpr int MI = Cfg::MI, NI = Cfg::NI;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kz0 = blockIdx.z * k_per_split;
  const int kz1 = min(K, kz0 + k_per_split);
  const int n_ktiles = (kz1 - kz0 + BK - 1) / BK;

  uint4 ra[Cfg::A_PER_THREAD], rw[Cfg::W_PER_THREAD];

  // per-thread source offsets (bytes) of its 16-byte chunks at k-tile 0, computed once: row-major  row * ld * esz + chunk * 16;
  // packed  (row >> 1) * 2 * ld * esz + (row & 1) * 64 + (chunk >> 2) * 128 + (chunk & 3) * 16  (a 128-byte k-tile is two 64-byte blocks,
  // each in the line its row pair shares).  A k-tile later is 128 bytes further on row-major rows, 256 on packed ones.
  constexpr int ESZ = (int)sizeof(T);
  const size_t kstep = pk ? 256 : 128;
  size_t aoff[Cfg::A_PER_THREAD], woff[Cfg::W_PER_THREAD];
  const size_t k0b = (size_t)kz0 * ESZ;                       // multiple of 128 (k_per_split is a multiple of BK)
#pragma unroll
  for (int i = 0; i < Cfg::A_PER_THREAD; ++i) {
    const int q = tid + i * kThreads, r = q >> 3, c = q & 7;
    const size_t gm = (size_t)min(m0 + r, M - 1);
    aoff[i] = pk ? (gm >> 1) * ((size_t)lda * ESZ * 2) + (gm & 1) * 64 + (k0b * 2) + (size_t)(c >> 2) * 128 + (c & 3) * 16
                 : gm * ((size_t)lda * ESZ) + k0b + (size_t)c * 16;
  }
#pragma unroll
  for (int i = 0; i < Cfg::W_PER_THREAD; ++i) {
    const int q = tid + i * kThreads, r = q >> 3, c = q & 7;
    const size_t gn = (size_t)min(n0 + r, N - 1);
    woff[i] = pk ? (gn >> 1) * ((size_t)K * ESZ * 2) + (gn & 1) * 64 + (k0b * 2) + (size_t)(c >> 2) * 128 + (c & 3) * 16
                 : gn * ((size_t)K * ESZ) + k0b + (size_t)c * 16;
  }
  const unsigned char* Ab = reinterpret_cast<const unsigned char*>(A);
  const unsigned char* Wb = reinterpret_cast<const unsigned char*>(W);

  auto load_tile = [&](int kt) {
    const int kbase = kz0 + kt * BK;
#pragma unroll
    for (int i = 0; i < Cfg::A_PER_THREAD; ++i) {
      int q = tid + i * kThreads;
      if (q < Cfg::A_CHUNKS) {
        int gk = kbase + (q & 7) * EPC;
        ra[i] = (gk < kz1) ? *reinterpret_cast<const uint4*>(Ab + aoff[i] + (size_t)kt * kstep) : make_uint4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < Cfg::W_PER_THREAD; ++i) {
      int q = tid + i * kThreads;
      if (q < Cfg::W_CHUNKS) {
        int gk = kbase + (q & 7) * EPC;
        rw[i] = (gk < kz1) ? *reinterpret_cast<const uint4*>(Wb + woff[i] + (size_t)kt * kstep) : make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto store_tile = [&](int buf) {
    unsigned char* sa = smem + buf * Cfg::STAGE_BYTES;
    unsigned char* sw = sa + BM * kRowBytes;
#pragma unroll
    for (int i = 0; i < Cfg::A_PER_THREAD; ++i) {
      int q = tid + i * kThreads;
      if (q < Cfg::A_CHUNKS) *reinterpret_cast<uint4*>(sa + swz(q >> 3, q & 7)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < Cfg::W_PER_THREAD; ++i) {
      int q = tid + i * kThreads;
      if (q < Cfg::W_CHUNKS) *reinterpret_cast<uint4*>(sw + swz(q >> 3, q & 7)) = rw[i];
    }
  };

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  if (n_ktiles > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();

  const int frow = lane & 15, fk = lane >> 4;
  for (int kt = 0; kt < n_ktiles; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < n_ktiles) load_tile(kt + 1);        // global loads in flight under the MFMAs
    const unsigned char* sa = smem + buf * Cfg::STAGE_BYTES + (wm * (BM / WM)) * kRowBytes;
    const unsigned char* sw = smem + buf * Cfg::STAGE_BYTES + BM * kRowBytes + (wn * (BN / WN)) * kRowBytes;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {                // two k-steps of 32
        s16x8_t af[MI], bfr[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const s16x8_t*>(sa + swz(i * 16 + frow, ks * 4 + fk));
#pragma unroll
        for (int j = 0; j < NI; ++j) bfr[j] = *reinterpret_cast<const s16x8_t*>(sw + swz(j * 16 + frow, ks * 4 + fk));
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = ATS_MFMA_16x16x32(__builtin_bit_cast(bf16x8_t, af[i]),
                                                                __builtin_bit_cast(bf16x8_t, bfr[j]), acc[i][j]);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {                // two groups of 16 k: lane group g owns chunk ks*4+g
        f32x4_t af[MI], bfr[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const f32x4_t*>(sa + swz(i * 16 + frow, ks * 4 + fk));
#pragma unroll
        for (int j = 0; j < NI; ++j) bfr[j] = *reinterpret_cast<const f32x4_t*>(sw + swz(j * 16 + frow, ks * 4 + fk));
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bfr[j][e], acc[i][j], 0, 0, 0);
      }
    }
    if (kt + 1 < n_ktiles) store_tile(buf ^ 1);       // other buffer: nobody reads it this iteration
    __syncthreads();
  }

  // ------------------------------------------------------------------ epilogue
  // C/D fragment: col = lane & 15, row = (lane >> 4) * 4 + reg
  const int crow = (lane >> 4) * 4, ccol = lane & 15;
  const int wrow0 = m0 + wm * (BM / WM), wcol0 = n0 + wn * (BN / WN);
  if constexpr (SPLIT) {
    float* P = partial + (size_t)blockIdx.z * M * N;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int gm = wrow0 + i * 16 + crow + r, gn = wcol0 + j * 16 + ccol;
          if (gm < M && gn < N) P[(size_t)gm * N + gn] = acc[i][j][r];
        }
  } else if constexpr (EPI == EPI_SWIGLU) {
    T* C = reinterpret_cast<T*>(Cv);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; j += 2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int gm = wrow0 + i * 16 + crow + r, gn = wcol0 + j * 16;      // gate group start (multiple of 32)
          if (gm < M && gn < N) {
            float g = acc[i][j][r], u = acc[i][j + 1][r];
            if constexpr (sizeof(T) == 2) { g = bf2f(f2bf(g)); u = bf2f(f2bf(u)); }
            float s = ats_silu<sizeof(T) == 4>(g);
            Elt<T>::store(C + ats_opnd_idx<sizeof(T)>(pk, gm, (gn >> 1) + ccol, ldc), s * u);
          }
        }
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int gm = wrow0 + i * 16 + crow + r, gn = wcol0 + j * 16 + ccol;
          if (gm < M && gn < N) {
            float v = acc[i][j][r];
            if constexpr (EPI == EPI_F32) {
              reinterpret_cast<float*>(Cv)[(size_t)gm * ldc + gn] = v;
            } else if constexpr (EPI == EPI_RESID) {
              T* C = reinterpret_cast<T*>(Cv);
              if constexpr (sizeof(T) == 2) v = bf2f(f2bf(v));           // HF: o_proj output is bf16 before the add
              Elt<T>::store(C + (size_t)gm * ldc + gn, Elt<T>::load(C + (size_t)gm * ldc + gn) + v);
            } else {
              Elt<T>::store(reinterpret_cast<T*>(Cv) + (size_t)gm * ldc + gn, v);
            }
          }
        }
  }
}

// V consecutive floats of every slab, summed in slab order.  The loads of four slabs are issued before their adds: with a runtime
// slab count the plain loop waited out one memory round trip per slab (8 slabs = 8 x ~2 us; seen as 19 us reduce kernels behind
// 15 us GEMMs in the one-user trace).
template <int V>
__device__ __forceinline__ void sum_slabs(const float* __restrict__ p, size_t slab_stride, int splits, float (&acc)[V]) {
  typedef float vf __attribute__((ext_vector_type(V)));
#pragma unroll
  for (int i = 0; i < V; ++i) acc[i] = 0.f;
  int z = 0;
  for (; z + 4 <= splits; z += 4) {
    vf q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const vf*>(p + (size_t)(z + u) * slab_stride);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < V; ++i) acc[i] += q[u][i];
    }
  }
  for (; z < splits; ++z) {
    vf q = *reinterpret_cast<const vf*>(p + (size_t)z * slab_stride);
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] += q[i];
  }
}

// sum the split-K slabs and apply the epilogue; one thread per V consecutive output elements (pairs for SwiGLU).
// V = 4 needs N % 4 == 0 and ldc % 4 == 0 (N % 32 == 0 holds for SwiGLU).
template <typename T, int EPI, int V>
__global__ void splitk_reduce_kernel(const float* __restrict__ partial, void* __restrict__ Cv, int M, int N, int ldc,
                                     int splits, int pk) {
  const size_t mn = (size_t)M * N;
  if constexpr (EPI == EPI_SWIGLU) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V;   // over M * N/2 outputs
    size_t tot = (size_t)M * (N / 2);
    if (i >= tot) return;
    int m = (int)(i / (N / 2)), o = (int)(i % (N / 2));
    int grp = o >> 4, c = o & 15;
    size_t gi = (size_t)m * N + grp * 32 + c, ui = gi + 16;
    float g[V], u[V];
    sum_slabs<V>(partial + gi, mn, splits, g);
    sum_slabs<V>(partial + ui, mn, splits, u);
    T* out = reinterpret_cast<T*>(Cv) + ats_opnd_idx<sizeof(T)>(pk, m, o, ldc);     // the SwiGLU output is the down projection's operand (V <= 4 stays inside a k-block)
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float gj = g[j], uj = u[j];
      if constexpr (sizeof(T) == 2) { gj = bf2f(f2bf(gj)); uj = bf2f(f2bf(uj)); }
      float sj = ats_silu<sizeof(T) == 4>(gj);
      Elt<T>::store(out + j, sj * uj);
    }
  } else {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (i >= mn) return;
    int m = (int)(i / N), n = (int)(i % N);
    float v[V];
    sum_slabs<V>(partial + i, mn, splits, v);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      if constexpr (EPI == EPI_F32) {
        reinterpret_cast<float*>(Cv)[(size_t)m * ldc + n + j] = v[j];
      } else if constexpr (EPI == EPI_RESID) {
        T* C = reinterpret_cast<T*>(Cv);
        float x = v[j];
        if constexpr (sizeof(T) == 2) x = bf2f(f2bf(x));
        Elt<T>::store(C + (size_t)m * ldc + n + j, Elt<T>::load(C + (size_t)m * ldc + n + j) + x);
      } else {
        Elt<T>::store(reinterpret_cast<T*>(Cv) + (size_t)m * ldc + n + j, v[j]);
      }
    }
  }
}


// Order of the weight rows inside each 64-row quarter of the ring kernel's LDS image: which weight row the DMA puts at LDS row rho.
// The MFMA leaves lane (lq, g) with FOUR consecutive positions p = g*4 + r of every 16-row tile i.  In weight order those are 4 output
// columns per tile: eight-byte stores, each instruction touching 16 token rows x 32 bytes.  Since the DMA's per-lane source address is
// free, the quarter's rows are laid into LDS in an order that makes a lane's positions of tiles i = 0..3 ADJACENT output columns:
//   ROWS_LANE16  (store / residual / qkv+RoPE): LDS row i*16 + p holds weight row (p>>2)*16 + i*4 + (p&3): lane (., g) owns columns
//                g*16 .. g*16+15 of the quarter -- two 16-byte stores per token row, a full 128-byte line per (row, wave);
//   ROWS_SWIGLU8 (gate_up, weights interleaved in 16-row gate / up groups): tiles (0, 1) and (2, 3) are (gate, up) of outputs
//                g*8 + r and g*8 + 4 + r: one 16-byte store per token row.
// No cost in the main loop (fragment reads and swizzle see LDS rows only; rows still come in groups of four, so the packed layout's
// full 128-byte lines per DMA piece are kept).  Timing-only forms of the store pattern on the Llama-7B projections at 26 k tokens
// (profiles/README.md): no stores at all +9-12 % (qkv, gate_up; down +4 %), the same bytes fully coalesced +4 %, this form +2.4-2.8 %.
// 16-byte epilogue store (a non-temporal form was tried at 26 k tokens: the store-only GEMM gained 0.4-2.4 %, the bench nothing -- every
// output is the next kernel's input).
__device__ __forceinline__ void epi_store16(bf16_t* p, const uint4& v) {
  *reinterpret_cast<uint4*>(p) = v;
}
enum { ROWS_IDENTITY = 0, ROWS_LANE16 = 1, ROWS_SWIGLU8 = 2 };
template <int EPI> constexpr int ring_row_order() {
  return (EPI == EPI_STORE || EPI == EPI_RESID || EPI == EPI_QKV_ROPE) ? ROWS_LANE16 : EPI == EPI_SWIGLU ? ROWS_SWIGLU8 : ROWS_IDENTITY;
}
template <int ORD> __device__ __forceinline__ int ring_src_row(int rho) {        // LDS row -> weight row, both 0..63 inside the quarter
  const int i = rho >> 4, p = rho & 15;
  if constexpr (ORD == ROWS_LANE16) return (p >> 2) * 16 + i * 4 + (p & 3);
  if constexpr (ORD == ROWS_SWIGLU8) { const int o = (p >> 2) * 8 + (i >> 1) * 4 + (p & 3); return (o >> 4) * 32 + (i & 1) * 16 + (o & 15); }
  return rho;
}

// Epilogue of the 256-wide kernels: wave (wn, wm) holds NA x MT2 accumulator tiles; token row m = m0 + wm*MT2*16 + j*16 + lq; the
// weight row of acc[i][j][r] is n0 + wn*NA*16 + (i>>2)*64 + ring_src_row<order of EPI>((i&3)*16 + g*4 + r).
template <int EPI, int NA, int MT2>
__device__ __forceinline__ void big_epilogue(f32x4_t (&acc)[NA][MT2], void* __restrict__ Cv, int M, int N, int ldc, int m0, int n0,
                                             int wn, int wm, int lq, int g, int pk = 0) {
  const int nw = n0 + wn * (NA * 16);                             // the wave's first weight row
  if constexpr (EPI == EPI_STORE || EPI == EPI_RESID) {
    bf16_t* Cb = reinterpret_cast<bf16_t*>(Cv);
    if ((ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(Cv) & 15) == 0 && nw + NA * 16 <= N) {
      // lane (., g) owns columns nw + q*64 + g*16 + [0, 16): tiles 4q + (0, 1) are the first eight, 4q + (2, 3) the second eight
      auto pack8 = [&](int i0, int j) {
        return make_uint4(f2bf_pk(acc[i0][j][0], acc[i0][j][1]), f2bf_pk(acc[i0][j][2], acc[i0][j][3]),
                          f2bf_pk(acc[i0 + 1][j][0], acc[i0 + 1][j][1]), f2bf_pk(acc[i0 + 1][j][2], acc[i0 + 1][j][3]));
      };
      auto col8 = [&](int i0) { return nw + (i0 >> 2) * 64 + g * 16 + ((i0 & 3) >> 1) * 8; };
      if constexpr (EPI == EPI_RESID) {
        // read-modify-write of h: the loads of a row group must not wait behind the previous group's stores (same pointer: the compiler
        // keeps them in order, one memory round trip per store), so all residuals of a 64-column quarter are fetched first
#pragma unroll
        for (int q = 0; q < NA / 4; ++q) {                          // (NA = 8: 64 residual registers at a time, not 128)
          uint4 rs[2][MT2];
#pragma unroll
          for (int j = 0; j < MT2; ++j) {
            const int gm = min(m0 + wm * (MT2 * 16) + j * 16 + lq, M - 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) rs[h][j] = *reinterpret_cast<const uint4*>(Cb + (size_t)gm * ldc + col8(4 * q + 2 * h));
          }
#pragma unroll
          for (int j = 0; j < MT2; ++j) {
            const int gm = m0 + wm * (MT2 * 16) + j * 16 + lq;
            if (gm >= M) continue;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const uint4 p = pack8(4 * q + 2 * h, j), r = rs[h][j];
              uint4 o;
              o.x = f2bf_pk(bf_lo(r.x) + bf_lo(p.x), bf_hi(r.x) + bf_hi(p.x));
              o.y = f2bf_pk(bf_lo(r.y) + bf_lo(p.y), bf_hi(r.y) + bf_hi(p.y));
              o.z = f2bf_pk(bf_lo(r.z) + bf_lo(p.z), bf_hi(r.z) + bf_hi(p.z));
              o.w = f2bf_pk(bf_lo(r.w) + bf_lo(p.w), bf_hi(r.w) + bf_hi(p.w));
              epi_store16(Cb + (size_t)gm * ldc + col8(4 * q + 2 * h), o);
            }
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < MT2; ++j) {
          const int gm = m0 + wm * (MT2 * 16) + j * 16 + lq;
          if (gm >= M) continue;
#pragma unroll
          for (int h = 0; h < NA / 2; ++h) epi_store16(Cb + (size_t)gm * ldc + col8(2 * h), pack8(2 * h, j));
        }
      }
      return;
    }
    // tiles that straddle N, odd strides: element by element
#pragma unroll
    for (int j = 0; j < MT2; ++j) {
      const int gm = m0 + wm * (MT2 * 16) + j * 16 + lq;
      if (gm >= M) continue;
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gn = nw + (i >> 2) * 64 + g * 16 + (i & 3) * 4 + r;
          if (gn >= N) continue;
          bf16_t* C = Cb + (size_t)gm * ldc + gn;
          float v = acc[i][j][r];
          if constexpr (EPI == EPI_RESID) v = bf2f(*C) + bf2f(f2bf(v));
          *C = f2bf(v);
        }
    }
  } else if constexpr (EPI == EPI_SWIGLU) {
    // lane (., g) owns outputs q*32 + g*8 + [0, 8) of the wave's NA*8: gate in tiles 4q and 4q+2, up in 4q+1 and 4q+3.  gate and up are rounded to
    // bf16 first (the reference's two projections are bf16 tensors), packed conversions throughout
    bf16_t* C = reinterpret_cast<bf16_t*>(Cv);
    const bool vec16 = (ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(Cv) & 15) == 0;
    auto silu_mul = [&](int ig, int j, int r) {
      const uint32_t gp = f2bf_pk(acc[ig][j][r], acc[ig][j][r + 1]), upk = f2bf_pk(acc[ig + 1][j][r], acc[ig + 1][j][r + 1]);
      return f2bf_pk(ats_silu<false>(bf_lo(gp)) * bf_lo(upk), ats_silu<false>(bf_hi(gp)) * bf_hi(upk));
    };
#pragma unroll
    for (int j = 0; j < MT2; ++j) {
      const int gm = m0 + wm * (MT2 * 16) + j * 16 + lq;
      if (gm >= M) continue;
#pragma unroll
      for (int q = 0; q < NA / 4; ++q) {
        if (nw + q * 64 + (g >> 1) * 32 >= N) continue;            // N % 32 == 0: a (gate, up) group of 16 outputs is inside N or not at all
        const int oc = (nw >> 1) + q * 32 + g * 8;
        const uint4 o = make_uint4(silu_mul(4 * q, j, 0), silu_mul(4 * q, j, 2), silu_mul(4 * q + 2, j, 0), silu_mul(4 * q + 2, j, 2));
        bf16_t* dst = C + ats_opnd_idx<2>(pk, gm, oc, ldc);        // the down projection's operand: packed when pk (8 outputs stay inside a 64-byte block)
        if (vec16) epi_store16(dst, o);
        else { *reinterpret_cast<uint2*>(dst) = make_uint2(o.x, o.y); *reinterpret_cast<uint2*>(dst + 4) = make_uint2(o.z, o.w); }
      }
    }
  } else {
    static_assert(EPI == EPI_F32, "fp32 store");
    const bool vec = (ldc & 3) == 0;
#pragma unroll
    for (int j = 0; j < MT2; ++j) {
      const int gm = m0 + wm * (MT2 * 16) + j * 16 + lq;
      if (gm >= M) continue;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int gn = nw + i * 16 + g * 4;
        if (gn >= N) continue;
        float* C = reinterpret_cast<float*>(Cv) + (size_t)gm * ldc + gn;
        if (gn + 3 < N && vec) *reinterpret_cast<float4*>(C) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        else
#pragma unroll
          for (int r = 0; r < 4; ++r) if (gn + r < N) C[r] = acc[i][j][r];
      }
    }
  }
}

// Epilogue of the batched qkv projection (EPI_QKV_ROPE): RoPE and the KV-cache scatter on the accumulators, so that the projection's
// output is never re-read (the separate pass moved 3 x tokens x hidden x 2 x 2 bytes per layer).  hidden % 256 == 0: a 256-column tile
// lies inside one of q / k / v and holds two heads of head_dim 128; wave (wn, wm) accumulated the half (wn & 1) of head (wn >> 1) for the
// rows of wm.  v tiles go to the cache as they are.  q / k tiles pass through the (drained) ring's LDS as bf16 and are re-divided by
// ROWS: a wave then rotates both halves of both heads for a quarter of wm's rows, so each (cos, sin) is fetched once per row and pair
// index instead of once per wave that holds the column (4 x fewer table bytes through the L1 -- the first form's cost).  Numerics are
// those of ats_gemm + rope_kv_segs_vec_kernel: the projection rounded to bf16, the rotation in fp32 on those values, one more rounding.
template <int NA, int MT2>
__device__ __forceinline__ void qkv_rope_epilogue(f32x4_t (&acc)[NA][MT2], bf16_t* __restrict__ qkv, int M, int ldc, int m0, int n0,
                                                  int wave, int lane, const RopeEpi& rp, unsigned char* smem) {
  static_assert(NA == 4 && MT2 % 4 == 0, "eight-wave tiling: a wave holds 64 columns");
  // weight rows in ROWS_LANE16 order: lane (lq, g) holds columns wn*64 + g*16 + i*4 + r of the tile, i.e. 16 adjacent columns
  const int wn = wave >> 1, wm = wave & 1, lq = lane & 15, g = lane >> 4;
  const int H = rp.hidden;
  const int sec = n0 / H;                                         // 0 q, 1 k, 2 v (uniform over the workgroup)
  const int fsec = n0 - sec * H;                                  // the tile's first column inside q / k / v
  auto pack4 = [&](int i, int j) { return make_uint2(f2bf_pk(acc[i][j][0], acc[i][j][1]), f2bf_pk(acc[i][j][2], acc[i][j][3])); };
  if (sec == 2) {
#pragma unroll
    for (int j = 0; j < MT2; ++j) {
      const int row = m0 + wm * (MT2 * 16) + j * 16 + lq;
      if (row >= M) continue;
      const RowInfo ri = rp.rows[row];
      bf16_t* dst = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(ri.vc) + rp.layer_off) + (size_t)ri.slot * H + fsec + wn * 64 + g * 16;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint2 a = pack4(2 * h, j), b = pack4(2 * h + 1, j);
        epi_store16(dst + h * 8, make_uint4(a.x, a.y, b.x, b.y));
      }
    }
    return;
  }
  uint2* ex = reinterpret_cast<uint2*>(smem);                     // [wave][j][i][lane]: 8 x MT2 x 4 x 64 x 8 B = 128 / 64 KB, within the ring's own size
  __syncthreads();                                                // slower waves may still be reading the ring's last stages
#pragma unroll
  for (int j = 0; j < MT2; ++j)
#pragma unroll
    for (int i = 0; i < NA; ++i) ex[((wave * MT2 + j) * NA + i) * 64 + lane] = pack4(i, j);
  __syncthreads();
  constexpr int JW = MT2 / 4;                                     // 16-row groups per wave after the re-division
#pragma unroll
  for (int jj = 0; jj < JW; ++jj) {
    const int j = wn * JW + jj;
    const int row = m0 + wm * (MT2 * 16) + j * 16 + lq;
    if (row >= M) continue;
    const RowInfo ri = rp.rows[row];
    bf16_t* dst = (sec == 0 ? qkv + (size_t)row * ldc
                            : reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(ri.kc) + rp.layer_off) + (size_t)ri.slot * H) + fsec + g * 16;
    const float* cp = rp.cos_tab + (size_t)ri.pos * 64 + g * 16;  // pair index inside the head: g * 16 + i * 4 + r
    const float* sp = rp.sin_tab + (size_t)ri.pos * 64 + g * 16;
#pragma unroll
    for (int h = 0; h < 2; ++h) {                                 // eight adjacent pair indices per 16-byte store
      float4 c[2], s[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) { c[e] = *reinterpret_cast<const float4*>(cp + (2 * h + e) * 4); s[e] = *reinterpret_cast<const float4*>(sp + (2 * h + e) * 4); }
#pragma unroll
      for (int hd = 0; hd < 2; ++hd) {
        uint32_t o0[4], o1[4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int i = 2 * h + e;
          const uint2 x = ex[((((hd * 2 + 0) * 2 + wm) * MT2 + j) * NA + i) * 64 + lane];      // x[d]      (wave wn = 2 hd)
          const uint2 y = ex[((((hd * 2 + 1) * 2 + wm) * MT2 + j) * NA + i) * 64 + lane];      // x[d + 64] (wave wn = 2 hd + 1)
          o0[2 * e] = f2bf_pk(rope_first(bf_lo(x.x), bf_lo(y.x), c[e].x, s[e].x), rope_first(bf_hi(x.x), bf_hi(y.x), c[e].y, s[e].y));
          o0[2 * e + 1] = f2bf_pk(rope_first(bf_lo(x.y), bf_lo(y.y), c[e].z, s[e].z), rope_first(bf_hi(x.y), bf_hi(y.y), c[e].w, s[e].w));
          o1[2 * e] = f2bf_pk(rope_second(bf_lo(x.x), bf_lo(y.x), c[e].x, s[e].x), rope_second(bf_hi(x.x), bf_hi(y.x), c[e].y, s[e].y));
          o1[2 * e + 1] = f2bf_pk(rope_second(bf_lo(x.y), bf_lo(y.y), c[e].z, s[e].z), rope_second(bf_hi(x.y), bf_hi(y.y), c[e].w, s[e].w));
        }
        epi_store16(dst + hd * 128 + h * 8, make_uint4(o0[0], o0[1], o0[2], o0[3]));
        epi_store16(dst + hd * 128 + 64 + h * 8, make_uint4(o1[0], o1[1], o1[2], o1[3]));
      }
    }
  }
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}
// inline asm: hipcc puts s_waitcnt vmcnt(0) in front of any ds_read it emits itself while an LDS-DMA is pending
#define ATS_DS_READ_B128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))

// =====================================================================================
// Large-M GEMM for the lock-step multi-user forwards (M = tokens of all users, 0.5-8 k rows): MFMA-bound.
//
//   C^T[n][m] = sum_k W[n][k] X[m][k]      A operand = W rows, B operand = X rows, both K-contiguous in HBM
//   workgroup tile 256 (n) x 256 or 128 (m), 8 waves (4 x 2), wave tile 64 x 128 (64) = 4 x 8 (4) MFMA 16x16x32 tiles.
//
// How it got here (profiles/README.md has the numbers): a two-stage 256x256x64 LDS-DMA loop ran the MFMA pipe 52 % busy
// (SQ counters), 1.3-1.6 PF without its DMA and unchanged when the vmcnt wait was dropped: what cost was the burst of
// 8 LDS-DMA pieces per wave right after the barrier (100-185 cycles of issue each while the VMEM queue is full) and the
// lock-step burst of 24 fragment reads, not the landing latency.  hipcc sinks and serialises the LDS-DMA builtin when
// asked to interleave it with MFMAs, so the loop body is inline asm (volatile asm statements keep their order):
//   * a stage is ONE 32-deep k-step (64-byte LDS rows, 32 KB for a 256x256 tile), four stages form a ring, and the DMA
//     of k-step s+4 is issued during the MFMAs of k-step s: three k-steps (96 KB per CU) are always in flight;
//   * the 12 fragment reads of k-step s+1 (register double buffering) and the 4 DMA pieces are spread between the
//     32 MFMAs of k-step s, so neither is a burst;
//   * one barrier per k-step both publishes k-step s+2 and retires the reads of k-step s+1.
// Both operands arrive by LDS-DMA (global_load_lds_dwordx4, saddr + 32-bit lane offset, M0 = destination): no VGPR round
// trip.  LDS image of a stage: W rows then X rows, 64 B each; the 16-byte chunk c of row r sits at position
// c ^ f((r>>2)&3), f = {2,0,1,3}, applied on the per-lane SOURCE address (the DMA destination is lane-linear) and on the
// fragment reads: conflict-free for ds_read_b128's lane groups ({0-3,12-15,20-27},...) with 64-byte rows (PMC: 0 conflicts).
// Workgroup ids are remapped so that each XCD (private 4 MB L2) walks a contiguous run of tiles, inside it bands of GM
// tile rows, W-panel-major: the ~32 tiles an XCD runs at once share GM X panels and 8 W panels.
// Result on MI355X: MFMA pipe 74 % busy at ~1.55 GHz (the chip lowers its clock under this load), 1.15-1.25 PF on the
// Llama-7B projections at 2-7 k tokens; hipBLASLt's stream-K 256x256x64 kernel reaches 1.05-1.39 PF on the same shapes.
#define ATS_MFMA_BF16(c, a, b) asm volatile(ATS_MFMA_16x16x32_NAME " %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define ATS_MFMA_BF16_A(c, a, b) asm volatile(ATS_MFMA_16x16x32_NAME " %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
#define ATS_MFMA_FP8(c, a, b) asm volatile("v_mfma_f32_16x16x32_fp8_fp8 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
// (m0 is written here without an "m0" clobber on purpose: m0 is a RESERVED register for LLVM's AMDGPU backend -- it never keeps a value
// live in it across instructions, it re-materialises m0 glued to each of its own m0 readers -- and hipcc rejects the clobber with
// -Winline-asm "clobber list contains reserved registers ... may lead to undefined behaviour".)
#define ATS_DMA16(voff, sbase, m0v) \
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(m0v) : "memory")

// FP8: the operands are e4m3 bytes with per-row scales (W8A8); a 64-byte row then holds 64 k, a lane's 16-byte fragment
// chunk feeds two v_mfma_f32_16x16x32_fp8_fp8 (low / high 8 bytes: the k order inside a 64-k group is permuted
// identically on both operands), so a stage carries twice the flops for the same DMA and LDS bytes.
// SPLITK (one user's tokens, M <= 256: every launch is one pass over W and HBM-bound): the grid is tiles x n_split, part z
// accumulates the k-steps of its share of the 128-k units and stores fp32 partials to slab z of Cv ([z][M][N]); the deep
// LDS-DMA ring (three k-steps = 72-96 KB per CU in flight, no register staging) is what pulls the weight stream.
// NA = 16-row weight tiles per wave: 4 -> 8 waves (4 x 2) of 64 x (MT2*16); 8 -> 4 waves (2 x 2) of 128 x 128 with the 256
// accumulator registers in AGPRs: one wave per SIMD and a third fewer LDS fragment bytes per flop (the kernel is power-limited).
// SK (stream-K tail, bf16 batched forwards in the 4-64-user band): the tile grid of such a forward is 0.3-3 rounds of 256 workgroups and a
// tile lasts 50-85 us, so the last, partly filled round costs a whole tile time (gate_up at 912 tokens: 344 tiles = 1.34 rounds, paid as 2).
// With SK the first n_dp = 256 * floor(tiles / 256) tiles run as before (one workgroup each, whole K, the epilogue of EPI), and the k-steps
// of the R remaining tiles are dealt EVENLY to G <= 256 more workgroups in units of 128 k: workgroup g takes units [g TU / G, (g + 1) TU / G)
// of the R x U unit space (U = K / 128 units per tile), i.e. the end of one tile and / or the start of the next.  A tile that ends up in
// several parts is finished in-launch by the part that arrives LAST (cdna_hip_programming.md, "In-launch split-K reduction", the sc1 form
// of its hand-off table's first row): every part writes its fp32 accumulators lane-linearly to its slot of the workspace with sc1
// (write-through) 16-byte stores, every wave drains vmcnt, workgroup barrier, one lane adds to the tile's counter (agent scope); the part
// whose add returns parts - 1 sums ALL parts in part order -- its own from registers at its place in that order, the others by sc1 loads:
// the sum does not depend on who arrived last -- and runs the ordinary epilogue.  Nobody waits for anybody: no spin, no co-residency
// assumption.  Consecutive ranges are given to workgroup ids 8 apart (same XCD under round-robin dispatch: a speed choice only).
struct SkTail { int n_dp = 0; int G = 0; int U = 0; int TU = 0; float* ws = nullptr; int* cnt = nullptr; };

// PANEL (WN = 2, WM = 4; round 5: one launch's 257-512 tokens -- 4-16 users in lock step, a long prompt's first verification): the same
// eight waves arranged 2 (n) x 4 (m): a workgroup owns a 128-row weight panel and ALL token rows (XR = 4 x MT2 x 16 = 384 or 512), so W
// is streamed once, no token tile is 38 % padding (320 tokens in two 256-row tiles) and a wide projection is one round of N / 128
// workgroups.  Stage size, pieces per wave and the loop are those of the 256 x 256 tile (128 + 384 rows = 32 KB, 1 W + 3 X pieces per wave).
template <int EPI, int MT2, bool FP8, bool SPLITK = false, int NA = 4, bool SK = false, int WN = 256 / (NA * 16), int WM = 2>
__global__ __launch_bounds__(WN * WM * 64, 1) void gemm_ring_kernel(const void* __restrict__ X, const void* __restrict__ W,
                                                           const float* __restrict__ sx, const float* __restrict__ sw,
                                                           void* __restrict__ Cv, int M, int N, int K, int ldx, int ldc,
                                                           int tiles_n, int tiles_m, int GM, int n_split = 1,
                                                           float* __restrict__ lse_part = nullptr, const unsigned char* __restrict__ tile_store = nullptr,
                                                           int pk = 0, RopeEpi rope = RopeEpi{}, SkTail sk = SkTail{}) {
  static_assert(!(SK && (SPLITK || FP8 || NA != 4 || WN != 4 || WM != 2)), "the stream-K tail is built for the bf16 eight-wave 4 x 2 form");
  static_assert((EPI != EPI_F32_LSE && EPI != EPI_QKV_ROPE) || (WN == 4 && WM == 2), "the LSE / RoPE epilogues are written for the 4 x 2 wave grid");
  // pk: X and W (and the SwiGLU output) are in the packed operand layout -- every 1 KB DMA piece is then eight FULL 128-byte lines
  // (two rows x 64 bytes each) instead of sixteen half lines: 83 against 55 GB/s per CU from L2 (tools/probe/dma_depth.hip)
  constexpr int BT = WN * NA * 16, RB = 64, ESZ = FP8 ? 1 : 2;   // weight rows per workgroup: 256 (4 x 2 waves) or 128 (panel form, 2 x 4)
  constexpr int BK = RB / ESZ;                                   // k per stage: 32 (bf16) or 64 (fp8)
  constexpr int XR = WM * MT2 * 16;                              // token rows per workgroup (256 or 128; panel: 384 or 512)
  constexpr int NWV = WN * WM;                                   // waves per workgroup: 8 or 4
  constexpr int WP = (BT / 16) / NWV;                            // W DMA pieces (16 rows each) per wave per k-step
  static_assert((BT / 16) % NWV == 0 && XR % (16 * NWV) == 0, "whole DMA pieces per wave");
  constexpr int XP = XR / (16 * NWV);                            // X DMA pieces per wave per k-step
  constexpr int NP = WP + XP;                                    // DMA pieces per wave per k-step
  constexpr int STAGE = (BT + XR) * RB;                          // 32 or 24 KB
  constexpr int NR = NA + MT2;                                   // fragment reads per wave per k-step
  constexpr int NMF = NA * MT2;                                  // MFMAs per wave per k-step
  constexpr int RG = (NMF * 3 / 4) / NR;                         // one read every RG MFMAs, from the segment's start (RG = 1: +0.0-0.6 %, not kept)
  constexpr int DG = (NMF - NR * RG) / NP;                       // then one DMA piece every DG MFMAs
  static_assert(RG >= 1 && DG >= 1, "segment too short for its reads and DMA pieces");
  constexpr int ORD = SPLITK ? ROWS_IDENTITY : ring_row_order<EPI>();   // order of the weight rows in the LDS image (see big_epilogue)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 15, g = lane >> 4;
  const int nwg = tiles_n * tiles_m;
  int bid = blockIdx.x, zpart = 0;
  if constexpr (SPLITK) { zpart = bid / nwg; bid -= zpart * nwg; }
  // stream-K tail: this workgroup's range [sk_u, sk_ue) of the tail's k-units, walked tile by tile by the segment loop below
  int sk_g = 0, sk_u = 0, sk_ue = 0, sk_seg = 0;
  bool sk_tail = false;
  if constexpr (SK) {
    if (bid >= sk.n_dp) {
      const int j = bid - sk.n_dp;
      sk_g = (sk.G & 7) ? j : (j & 7) * (sk.G >> 3) + (j >> 3);    // G % 8 == 0: consecutive ranges on ids 8 apart (one XCD)
      sk_u = (int)((long long)sk_g * sk.TU / sk.G);
      sk_ue = (int)((long long)(sk_g + 1) * sk.TU / sk.G);
      sk_tail = true;
    }
  }
  const int wn = wave / WM, wm = wave % WM;
  auto swz = [](int row) { return (0xD2 >> (((row >> 2) & 3) * 2)) & 3; };   // f = {2,0,1,3} packed in 0b11010010
  const unsigned lbase = lds_addr(smem);
  int m0w[WP], m0x[XP];
#pragma unroll
  for (int j = 0; j < WP; ++j) m0w[j] = __builtin_amdgcn_readfirstlane((int)lbase + (wave * WP + j) * 1024);
#pragma unroll
  for (int j = 0; j < XP; ++j) m0x[j] = __builtin_amdgcn_readfirstlane((int)lbase + BT * RB + (wave * XP + j) * 1024);
  bool sk_more;
  do {                                                             // one pass unless SK: a tail workgroup's range may end one tile and start the next
  sk_more = false;
  int sk_tt = 0;                                                   // tail tile of this segment, whole = it covers the tile's whole K
  bool sk_whole = true;
  int ks0 = 0, nks = K / BK;                                     // launcher: K % (4 BK) == 0; nks = END of this part's k-steps
  if constexpr (SK) {
    if (sk_tail) {
      sk_tt = sk_u / sk.U;
      const int k0 = sk_u - sk_tt * sk.U, k1 = min(sk.U, sk_ue - sk_tt * sk.U);
      ks0 = 4 * k0; nks = 4 * k1;
      sk_whole = k0 == 0 && k1 == sk.U;
      sk_u = sk_tt * sk.U + k1;
      sk_more = sk_u < sk_ue;
      bid = sk.n_dp + sk_tt;                                       // the tail's tiles are the last positions of the tile order
    } else {
      bid = (bid & 7) * (sk.n_dp >> 3) + (bid >> 3);               // n_dp % 256 == 0: XCD x walks positions [x n_dp / 8, (x + 1) n_dp / 8)
    }
  } else {
    const int q = nwg / 8, r = nwg % 8, x = bid % 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
  }
  const int band = bid / (GM * tiles_n), rem = bid % (GM * tiles_n);
  const int band_rows = min(GM, tiles_m - band * GM);
  const int tn = rem / band_rows, tm = band * GM + rem % band_rows;
  const int n0 = tn * BT, m0 = tm * XR;
  if constexpr (SPLITK) {
    const int units = nks >> 2;                                  // launcher: n_split <= units
    ks0 = 4 * (int)((long long)zpart * units / n_split);
    nks = 4 * (int)((long long)(zpart + 1) * units / n_split);
  }

  // per-lane DMA source offsets (bytes): piece = 16 rows x 64 B, lane l -> row l>>2, stored position l&3
  unsigned woff[WP], xoff[XP];
#pragma unroll
  for (int j = 0; j < WP; ++j) {
    const int row = (wave * WP + j) * 16 + (lane >> 2);          // LDS row; its weight row follows the epilogue's row order (ring_src_row)
    const int gr = min(n0 + (row & ~63) + ring_src_row<ORD>(row & 63), N - 1);
    woff[j] = (pk ? (unsigned)(gr >> 1) * (unsigned)(K * ESZ * 2) + (gr & 1) * 64 : (unsigned)gr * (unsigned)(K * ESZ)) + (((lane & 3) ^ swz(row)) * 16);
  }
#pragma unroll
  for (int j = 0; j < XP; ++j) {
    const int row = (wave * XP + j) * 16 + (lane >> 2), gr = min(m0 + row, M - 1);
    xoff[j] = (pk ? (unsigned)(gr >> 1) * (unsigned)(ldx * ESZ * 2) + (gr & 1) * 64 : (unsigned)gr * (unsigned)(ldx * ESZ)) + (((lane & 3) ^ swz(row)) * 16);
  }
  const unsigned long long wb = (unsigned long long)W, xb = (unsigned long long)X;
  const unsigned long long kadv = pk ? 2 * RB : RB;               // bytes from one k-step's 64-byte block of a row to the next
  // fragment addresses: lane (lq, g) reads chunk g of row lq of each 16-row tile
  const unsigned lp = lq * RB + ((g ^ swz(lq)) * 16);
  unsigned aA[2], aB[2];                                          // stages {0,1} and {2,3}
  aA[0] = lbase + (wn * NA * 16) * RB + lp;
  aB[0] = lbase + BT * RB + (wm * MT2 * 16) * RB + lp;
  aA[1] = aA[0] + 2 * STAGE;
  aB[1] = aB[0] + 2 * STAGE;

  f32x4_t acc[NA][MT2];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < MT2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  u32x4_t fa[2][NA], fb[2][MT2];

  auto dma_piece = [&](int q, int ks, int d) {                    // piece d of k-step ks into stage q (q, d compile-time after unrolling)
    if (d < WP) ATS_DMA16(woff[d % WP], wb + (unsigned long long)ks * kadv, m0w[d % WP] + q * STAGE);
    else        ATS_DMA16(xoff[(d - WP) % XP], xb + (unsigned long long)ks * kadv, m0x[(d - WP) % XP] + q * STAGE);
  };
  auto read_one = [&](int q, int buf, int r) {                    // fragment read r of stage q into register buffer buf
    const unsigned a = aA[q >> 1], b = aB[q >> 1];
    const int so = (q & 1) * STAGE;
#define ATS_RD(dst, base, t) case t: ATS_DS_READ_B128(dst, base, so + (t) * 1024); break;
    if (r < NA) {
      switch (r) {
        ATS_RD(fa[buf][0], a, 0) ATS_RD(fa[buf][1], a, 1) ATS_RD(fa[buf][2], a, 2) ATS_RD(fa[buf][3], a, 3)
        case 4: if constexpr (NA == 8) ATS_DS_READ_B128(fa[buf][NA - 4], a, so + 4096); break;
        case 5: if constexpr (NA == 8) ATS_DS_READ_B128(fa[buf][NA - 3], a, so + 5120); break;
        case 6: if constexpr (NA == 8) ATS_DS_READ_B128(fa[buf][NA - 2], a, so + 6144); break;
        case 7: if constexpr (NA == 8) ATS_DS_READ_B128(fa[buf][NA - 1], a, so + 7168); break;
        default: break;
      }
    } else {
      switch (r - NA) {
        ATS_RD(fb[buf][0], b, 0) ATS_RD(fb[buf][1], b, 1) ATS_RD(fb[buf][2], b, 2) ATS_RD(fb[buf][3], b, 3)
        case 4: if constexpr (MT2 > 4) ATS_DS_READ_B128(fb[buf][MT2 > 4 ? 4 : 0], b, so + 4096); break;
        case 5: if constexpr (MT2 > 5) ATS_DS_READ_B128(fb[buf][MT2 > 5 ? 5 : 0], b, so + 5120); break;
        case 6: if constexpr (MT2 > 6) ATS_DS_READ_B128(fb[buf][MT2 > 6 ? 6 : 0], b, so + 6144); break;
        case 7: if constexpr (MT2 > 7) ATS_DS_READ_B128(fb[buf][MT2 > 7 ? 7 : 0], b, so + 7168); break;
        default: break;
      }
    }
#undef ATS_RD
  };
  // one k-step: MFMAs of stage Q from register buffer Q&1, reads of stage Q+1 into the other buffer, DMA of k-step
  // ks+4 into stage Q; VM = vmcnt to wait for before the closing barrier (-1: no wait, no barrier)
#define ATS_RING_SEGMENT(Q, DMA, RD, VM, ks)                                                            \
  {                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) _Pragma("unroll") for (int j = 0; j < MT2; ++j) {     \
      const int idx = i * MT2 + j;                                                                       \
      if (RD && idx % RG == 0 && idx / RG < NR) read_one(((Q) + 1) & 3, ((Q) + 1) & 1, idx / RG);       \
      if (DMA && idx >= NR * RG && (idx - NR * RG) % DG == 0 && (idx - NR * RG) / DG < NP)              \
        dma_piece((Q), (ks) + 4, (idx - NR * RG) / DG);                                                  \
      if constexpr (FP8) {                                                                               \
        ATS_MFMA_FP8(acc[i][j], __builtin_shufflevector(fa[(Q) & 1][i], fa[(Q) & 1][i], 0, 1),           \
                     __builtin_shufflevector(fb[(Q) & 1][j], fb[(Q) & 1][j], 0, 1));                     \
        ATS_MFMA_FP8(acc[i][j], __builtin_shufflevector(fa[(Q) & 1][i], fa[(Q) & 1][i], 2, 3),           \
                     __builtin_shufflevector(fb[(Q) & 1][j], fb[(Q) & 1][j], 2, 3));                     \
      } else if constexpr (NA == 8) {                                                                    \
        ATS_MFMA_BF16_A(acc[i][j], fa[(Q) & 1][i], fb[(Q) & 1][j]);                                      \
      } else {                                                                                           \
        ATS_MFMA_BF16(acc[i][j], fa[(Q) & 1][i], fb[(Q) & 1][j]);                                        \
      }                                                                                                  \
    }                                                                                                    \
    if (RD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                           \
    if ((VM) >= 0) {                                                                                     \
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((VM) < 0 ? 0 : (VM)) : "memory");                         \
      asm volatile("s_barrier" ::: "memory");                                                            \
    }                                                                                                    \
  }

  // prologue: k-steps 0..3 into stages 0..3; fragments of k-step 0
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int d = 0; d < NP; ++d) dma_piece(q, ks0 + q, d);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NP) : "memory");
  asm volatile("s_barrier" ::: "memory");
#pragma unroll
  for (int r = 0; r < NR; ++r) read_one(0, 0, r);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
  asm volatile("s_barrier" ::: "memory");                          // k-step 1 published, stage 0 read by everyone

  int ks = ks0;
  for (; ks + 4 < nks; ks += 4) {
    ATS_RING_SEGMENT(0, true, true, 2 * NP, ks);
    ATS_RING_SEGMENT(1, true, true, 2 * NP, ks + 1);
    ATS_RING_SEGMENT(2, true, true, 2 * NP, ks + 2);
    ATS_RING_SEGMENT(3, true, true, 2 * NP, ks + 3);
  }
  ATS_RING_SEGMENT(0, false, true, NP, ks);
  ATS_RING_SEGMENT(1, false, true, 0, ks + 1);
  ATS_RING_SEGMENT(2, false, true, -1, ks + 2);
  ATS_RING_SEGMENT(3, false, false, -1, ks + 3);
#undef ATS_RING_SEGMENT
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");               // MFMA results -> VALU reads (the compiler cannot see the asm MFMAs)

  bool sk_epilogue = true;
  if constexpr (SK) {
    if (sk_tail && !sk_whole) {
      // a PART of tail tile sk_tt: publish the accumulators, draw a ticket; the last part to arrive sums all parts and goes on
      constexpr int NT = NWV * 64, SLOT = NA * MT2 * NT;           // threads, 16-byte items per slot (256 / 128 KB)
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(sk.ws, 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < MT2; ++j) {
          u32x4_t v;
          __builtin_memcpy(&v, &acc[i][j], 16);
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, (((sk_g * 2 + sk_seg) * SLOT) + (i * MT2 + j) * NT + tid) * 16, 0, 16);   // aux 16 = sc1
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // every storing wave, before the barrier the ticket lane joins
      __syncthreads();                                             // (also: every wave is done with the ring, smem[0..3] is free)
      const int u_lo = sk_tt * sk.U, u_hi = u_lo + sk.U - 1;       // the part that holds unit u: ceil((u + 1) G / TU) - 1
      const int g_first = (int)(((long long)(u_lo + 1) * sk.G + sk.TU - 1) / sk.TU) - 1;
      const int g_last = (int)(((long long)(u_hi + 1) * sk.G + sk.TU - 1) / sk.TU) - 1;
      const int parts = g_last - g_first + 1;
      volatile int* flag = reinterpret_cast<volatile int*>(smem);
      if (tid == 0) {
        const int old = __hip_atomic_fetch_add(sk.cnt + sk_tt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == parts - 1) __hip_atomic_store(sk.cnt + sk_tt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next launch
        *flag = old;
      }
      __syncthreads();
      sk_epilogue = *flag == parts - 1;
      if (sk_epilogue) {
        auto slot_of = [&](int gp) { return (gp * 2 + (((int)((long long)gp * sk.TU / sk.G)) / sk.U == sk_tt ? 0 : 1)) * SLOT; };
        auto part_load = [&](int slot, int idx) {
          const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (slot + idx * NT + tid) * 16, 0, 16);
          f32x4_t f;
          __builtin_memcpy(&f, &v, 16);
          return f;
        };
        // sum in part order, the own part (still in registers) at its place: bit-identical whoever arrived last.  CH loads in flight per
        // lane and part (the RoPE epilogue's registers leave room for 4: with 8 that instantiation spilled)
        constexpr int CH = EPI == EPI_QKV_ROPE ? 4 : 8;
#pragma unroll
        for (int c0 = 0; c0 < NA * MT2; c0 += CH) {
          f32x4_t pre[CH];
#pragma unroll
          for (int e = 0; e < CH; ++e) pre[e] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          for (int gp = g_first; gp < sk_g; ++gp) {
            const int so = slot_of(gp);
            f32x4_t v[CH];
#pragma unroll
            for (int e = 0; e < CH; ++e) v[e] = part_load(so, c0 + e);
#pragma unroll
            for (int e = 0; e < CH; ++e) pre[e] += v[e];
          }
          if (sk_g > g_first) {
#pragma unroll
            for (int e = 0; e < CH; ++e) acc[(c0 + e) / MT2][(c0 + e) % MT2] = pre[e] + acc[(c0 + e) / MT2][(c0 + e) % MT2];
          }
          for (int gp = sk_g + 1; gp <= g_last; ++gp) {
            const int so = slot_of(gp);
            f32x4_t v[CH];
#pragma unroll
            for (int e = 0; e < CH; ++e) v[e] = part_load(so, c0 + e);
#pragma unroll
            for (int e = 0; e < CH; ++e) acc[(c0 + e) / MT2][(c0 + e) % MT2] += v[e];
          }
        }
      }
      sk_seg = 1;
    }
  }
  if (sk_epilogue) {
  if constexpr (FP8) {                                             // per-row scales: acc[i][j][r] *= sx[m] * sw[n]
#pragma unroll
    for (int j = 0; j < MT2; ++j) {
      const float fx = sx[min(m0 + wm * (MT2 * 16) + j * 16 + lq, M - 1)];
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int gn = n0 + wn * (NA * 16) + (i >> 2) * 64 + ring_src_row<ORD>((i & 3) * 16 + g * 4);   // weight row of acc[i][j][0]; r: the next three
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] *= fx * sw[min(gn + r, N - 1)];
      }
    }
  }
  if constexpr (EPI == EPI_F32_LSE) {
    // lm_head fused with the full-vocabulary normaliser (beamSD.py:58,285: log-softmax over ALL columns before masking): this tile's
    // (max, sum exp) per token row goes to lse_part[row][tile_n]; the fp32 logits themselves are stored only when some column of the
    // tile can ever be asked for (tile_store[tn]: tiles holding a token of the constraint automaton).  A wave holds 64 of the tile's
    // 256 columns for its rows: lane-local over the 16 registers, across the four column groups g by two shuffles, across the four
    // waves wn through 8 KB of LDS behind the ring (the ring itself may still be read by slower waves).
    float* red = reinterpret_cast<float*>(smem + 4 * STAGE);      // [XR rows][4 wn][2]
#pragma unroll
    for (int j = 0; j < MT2; ++j) {
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int gn = n0 + wn * (NA * 16) + i * 16 + g * 4 + r; if (gn < N) mx = fmaxf(mx, acc[i][j][r]); }
      float sm = 0.f;
      if (mx > -INFINITY) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const int gn = n0 + wn * (NA * 16) + i * 16 + g * 4 + r; if (gn < N) sm += __expf(acc[i][j][r] - mx); }
      }
#pragma unroll
      for (int o = 16; o <= 32; o <<= 1) {
        const float om = __shfl_xor(mx, o, 64), os = __shfl_xor(sm, o, 64);
        const float nm = fmaxf(mx, om);
        sm = (nm > -INFINITY) ? sm * __expf(mx - nm) + os * __expf(om - nm) : 0.f;
        mx = nm;
      }
      if (g == 0) { float2* dst = reinterpret_cast<float2*>(red) + (size_t)(wm * (MT2 * 16) + j * 16 + lq) * 4 + wn; *dst = make_float2(mx, sm); }
    }
    __syncthreads();
    if (tid < XR && m0 + tid < M) {
      const float2* src = reinterpret_cast<const float2*>(red) + (size_t)tid * 4;
      float mx = src[0].x, sm = src[0].y;
#pragma unroll
      for (int q = 1; q < 4; ++q) {
        const float nm = fmaxf(mx, src[q].x);
        sm = (nm > -INFINITY) ? sm * __expf(mx - nm) + src[q].y * __expf(src[q].x - nm) : 0.f;
        mx = nm;
      }
      reinterpret_cast<float2*>(lse_part)[(size_t)(m0 + tid) * tiles_n + tn] = make_float2(mx, sm);
    }
    if (tile_store == nullptr || tile_store[tn]) big_epilogue<EPI_F32, NA, MT2>(acc, Cv, M, N, ldc, m0, n0, wn, wm, lq, g);
  } else if constexpr (EPI == EPI_QKV_ROPE) qkv_rope_epilogue<NA, MT2>(acc, reinterpret_cast<bf16_t*>(Cv), M, ldc, m0, n0, wave, lane, rope, smem);
  else if constexpr (SPLITK) big_epilogue<EPI_F32, NA, MT2>(acc, reinterpret_cast<float*>(Cv) + (size_t)zpart * M * N, M, N, N, m0, n0, wn, wm, lq, g);
  else                  big_epilogue<EPI, NA, MT2>(acc, Cv, M, N, ldc, m0, n0, wn, wm, lq, g, pk);
  }
  if constexpr (SK) { if (sk_more) { sk_seg = 1; __syncthreads(); } }   // the next segment's DMA overwrites LDS an epilogue may still be reading
  } while (sk_more);
}

// =====================================================================================
// The same ring on the block-scaled fp8 MFMA (BASELINE config 5): v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands and unit
// block scales (E8M0 0x7f) issues 2x the flops per cycle of the bf16 / non-scaled fp8 forms (tools/probe/mx_probe.hip on MI355X:
// 4.6 PF register-only against 2.4 PF for v_mfma_f32_16x16x32_fp8_fp8; lane map checked there with exact integer data: lane l
// holds row l&31 and the 32 bytes k = 32*(l>>5) .. +31; C/D as the bf16 32x32 forms).  The per-row / per-token fp32 scales of
// the W8A8 scheme are applied to the accumulators as before, so the numerics equal the non-scaled kernel's (exact products,
// fp32 accumulation; only the summation order inside a 64-k group differs).
// Same LDS image, DMA pieces and four-stage ring as gemm_ring_kernel<.., FP8 = true> (a stage = 64 k of e4m3 in 64-byte rows);
// what changes is the consumer: a wave's 64 x (MT2*16) tile is 2 x MT2/2 tiles of 32x32, a lane's operand for one 32x32x64
// MFMA is the two 16-byte chunks 2h, 2h+1 of its row (two ds_read_b128 into the halves of an 8-register tuple), and a k-step
// is 8 (4) MFMAs of 64 cycles with the 12 (8) fragment reads of the next k-step and the 4 (3) DMA pieces of k-step s+4 spread
// evenly between them.  Register budget as before: 128 accumulators + 96 fragment registers (double buffered).
typedef unsigned int u32x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
#define ATS_MFMA_MX(c, a, b, s) \
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(c) : "v"(a), "v"(b), "v"(s))
// accumulators in AGPRs (the one-wave-per-SIMD form: 256 accumulator registers do not fit the 256 architectural VGPRs beside the fragments)
// (the host pass of hipcc checks asm constraints against x86, where "a" is rax: a 64-byte operand fails template substitution there and the
// kernel's host stub silently disappears, so the statement exists in the device pass only)
#if defined(__HIP_DEVICE_COMPILE__)
#define ATS_MFMA_MX_A(c, a, b, s) \
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+a"(c) : "v"(a), "v"(b), "v"(s))
#else
#define ATS_MFMA_MX_A(c, a, b, s) ((void)0)
#endif

// accumulator tile (it, jt) of a wave: m = m0w + jt*32 + (lane&31); registers 4q..4q+3 hold n = n0w + it*32 + 8q + 4(lane>>5) + r
// The per-row scales of the W8A8 scheme (acc *= sx[m] * sw[n]) are applied tile by tile right where a tile is consumed: scaling all
// accumulators first kept 256 of them live in VGPRs in the one-wave-per-SIMD form (they sit in AGPRs during the loop) and spilled.
template <int TA, int TB>
__device__ __forceinline__ void mx_scale_tile(f32x16_t& t, const float* __restrict__ sx, const float* __restrict__ sw, int gm, int gn0, int h, int M, int N) {
  const float fx = sx[min(gm, M - 1)];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) t[4 * q + r] *= fx * sw[min(gn0 + 8 * q + 4 * h + r, N - 1)];
}

template <int EPI, int TA, int TB>
__device__ __forceinline__ void mx_epilogue(f32x16_t (&acc)[TA][TB], void* __restrict__ Cv, int M, int N, int ldc, int m0w, int n0w, int lane,
                                            const float* __restrict__ sx, const float* __restrict__ sw, int pk) {
  const int r32 = lane & 31, h = lane >> 5;
  const bool vec = (ldc & 3) == 0;
  if constexpr (EPI == EPI_RESID) {
    if (vec && n0w + TA * 32 <= N) {       // residuals of the whole wave tile fetched before the first store (see big_epilogue)
      bf16_t* Cb = reinterpret_cast<bf16_t*>(Cv);
      uint2 rs[TA][TB][4];
#pragma unroll
      for (int jt = 0; jt < TB; ++jt) {
        const int gm = min(m0w + jt * 32 + r32, M - 1);
#pragma unroll
        for (int it = 0; it < TA; ++it)
#pragma unroll
          for (int q = 0; q < 4; ++q) rs[it][jt][q] = *reinterpret_cast<const uint2*>(Cb + (size_t)gm * ldc + n0w + it * 32 + 8 * q + 4 * h);
      }
#pragma unroll
      for (int jt = 0; jt < TB; ++jt) {
        const int gm = m0w + jt * 32 + r32;
        if (gm >= M) continue;
#pragma unroll
        for (int it = 0; it < TA; ++it) {
          mx_scale_tile<TA, TB>(acc[it][jt], sx, sw, gm, n0w + it * 32, h, M, N);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t p0 = f2bf_pk(acc[it][jt][4 * q], acc[it][jt][4 * q + 1]), p1 = f2bf_pk(acc[it][jt][4 * q + 2], acc[it][jt][4 * q + 3]);
            uint2 o;
            o.x = f2bf_pk(bf_lo(rs[it][jt][q].x) + bf_lo(p0), bf_hi(rs[it][jt][q].x) + bf_hi(p0));
            o.y = f2bf_pk(bf_lo(rs[it][jt][q].y) + bf_lo(p1), bf_hi(rs[it][jt][q].y) + bf_hi(p1));
            *reinterpret_cast<uint2*>(Cb + (size_t)gm * ldc + n0w + it * 32 + 8 * q + 4 * h) = o;
          }
        }
      }
      return;
    }
  }
#pragma unroll
  for (int jt = 0; jt < TB; ++jt) {
    const int gm = m0w + jt * 32 + r32;
    if (gm >= M) continue;
#pragma unroll
    for (int it = 0; it < TA; ++it) {
      mx_scale_tile<TA, TB>(acc[it][jt], sx, sw, gm, n0w + it * 32, h, M, N);
      if constexpr (EPI == EPI_SWIGLU) {
        // rows 0-15 of the 32-row weight tile are a gate group (q = 0, 1), rows 16-31 its up group (q = 2, 3): the pair sits in one lane
        bf16_t* C = reinterpret_cast<bf16_t*>(Cv);
        const int gn0 = n0w + it * 32;
        if (gn0 >= N) continue;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          uint2 o;
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const uint32_t gp = f2bf_pk(acc[it][jt][4 * q + r], acc[it][jt][4 * q + r + 1]);
            const uint32_t upk = f2bf_pk(acc[it][jt][4 * (q + 2) + r], acc[it][jt][4 * (q + 2) + r + 1]);
            const float g0 = bf_lo(gp), g1 = bf_hi(gp);
            const uint32_t res = f2bf_pk(ats_silu<false>(g0) * bf_lo(upk), ats_silu<false>(g1) * bf_hi(upk));
            if (r == 0) o.x = res; else o.y = res;
          }
          *reinterpret_cast<uint2*>(C + ats_opnd_idx<2>(pk, gm, (gn0 >> 1) + 8 * q + 4 * h, ldc)) = o;      // the down projection's operand
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int gn = n0w + it * 32 + 8 * q + 4 * h;
          if (gn >= N) continue;
          const float v0 = acc[it][jt][4 * q], v1 = acc[it][jt][4 * q + 1], v2 = acc[it][jt][4 * q + 2], v3 = acc[it][jt][4 * q + 3];
          if constexpr (EPI == EPI_F32) {
            float* C = reinterpret_cast<float*>(Cv) + (size_t)gm * ldc + gn;
            if (gn + 3 < N && vec) *reinterpret_cast<float4*>(C) = make_float4(v0, v1, v2, v3);
            else { const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
              for (int r = 0; r < 4; ++r) if (gn + r < N) C[r] = vv[r]; }
          } else {
            bf16_t* C = reinterpret_cast<bf16_t*>(Cv) + (size_t)gm * ldc + gn;
            if (gn + 3 < N && vec) {
              uint2 o;
              const uint32_t p0 = f2bf_pk(v0, v1), p1 = f2bf_pk(v2, v3);
              if constexpr (EPI == EPI_RESID) {
                const uint2 rs = *reinterpret_cast<const uint2*>(C);
                o.x = f2bf_pk(bf_lo(rs.x) + bf_lo(p0), bf_hi(rs.x) + bf_hi(p0));
                o.y = f2bf_pk(bf_lo(rs.y) + bf_lo(p1), bf_hi(rs.y) + bf_hi(p1));
              } else { o.x = p0; o.y = p1; }
              *reinterpret_cast<uint2*>(C) = o;
            } else {
              const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (gn + r < N) {
                  float v = vv[r];
                  if constexpr (EPI == EPI_RESID) v = bf2f(C[r]) + bf2f(f2bf(v));
                  C[r] = f2bf(v);
                }
            }
          }
        }
      }
    }
  }
}

// EPI_QKV_ROPE on the block-scaled kernel's accumulator layout (eight waves: TA = 2): the same scheme as qkv_rope_epilogue -- v tiles
// straight to the cache; q / k tiles through the drained ring's LDS as bf16, then re-divided so that a wave rotates both halves of both
// heads for a quarter of wm's (32-row group, 32-column group) pairs and fetches each (cos, sin) once.
template <int TA, int TB>
__device__ __forceinline__ void mx_qkv_rope_epilogue(f32x16_t (&acc)[TA][TB], bf16_t* __restrict__ qkv, int M, int N, int ldc, int m0, int n0, int wave,
                                                     int lane, const float* __restrict__ sx, const float* __restrict__ sw, const RopeEpi& rp,
                                                     unsigned char* smem) {
  static_assert(TA == 2 && TB % 2 == 0, "eight-wave tiling: a wave holds 64 columns");
  const int wn = wave >> 1, wm = wave & 1, r32 = lane & 31, h = lane >> 5;
  const int H = rp.hidden;
  const int sec = n0 / H, fsec = n0 - sec * H;                    // 0 q, 1 k, 2 v; the tile's first column inside it
  const int m0w = m0 + wm * (TB * 32), n0w = n0 + wn * 64;
  if (sec == 2) {
#pragma unroll
    for (int jt = 0; jt < TB; ++jt) {
      const int row = m0w + jt * 32 + r32;
      if (row >= M) continue;
      const RowInfo ri = rp.rows[row];
      bf16_t* dst = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(ri.vc) + rp.layer_off) + (size_t)ri.slot * H + fsec + wn * 64 + 4 * h;
#pragma unroll
      for (int it = 0; it < TA; ++it) {
        mx_scale_tile<TA, TB>(acc[it][jt], sx, sw, row, n0w + it * 32, h, M, N);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<uint2*>(dst + it * 32 + 8 * q) = make_uint2(f2bf_pk(acc[it][jt][4 * q], acc[it][jt][4 * q + 1]), f2bf_pk(acc[it][jt][4 * q + 2], acc[it][jt][4 * q + 3]));
      }
    }
    return;
  }
  uint2* ex = reinterpret_cast<uint2*>(smem);                     // [wave][jt][it][q][lane]: 8 x TB x 2 x 4 x 64 x 8 B = 128 / 64 KB
  __syncthreads();                                                // slower waves may still be reading the ring's last stages
#pragma unroll
  for (int jt = 0; jt < TB; ++jt) {
    const int row = m0w + jt * 32 + r32;
#pragma unroll
    for (int it = 0; it < TA; ++it) {
      mx_scale_tile<TA, TB>(acc[it][jt], sx, sw, row, n0w + it * 32, h, M, N);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        ex[(((wave * TB + jt) * TA + it) * 4 + q) * 64 + lane] = make_uint2(f2bf_pk(acc[it][jt][4 * q], acc[it][jt][4 * q + 1]), f2bf_pk(acc[it][jt][4 * q + 2], acc[it][jt][4 * q + 3]));
    }
  }
  __syncthreads();
  constexpr int PW = TB * TA / 4;                                 // (jt, it) pairs per wave after the re-division
#pragma unroll
  for (int pp = 0; pp < PW; ++pp) {
    const int p = wn * PW + pp, jt = p / TA, it = p % TA;
    const int row = m0w + jt * 32 + r32;
    if (row >= M) continue;
    const RowInfo ri = rp.rows[row];
    bf16_t* dst = (sec == 0 ? qkv + (size_t)row * ldc
                            : reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(ri.kc) + rp.layer_off) + (size_t)ri.slot * H) + fsec + it * 32 + 4 * h;
    const float* cp = rp.cos_tab + (size_t)ri.pos * 64 + it * 32 + 4 * h;   // pair index inside the head: it * 32 + 8 q + 4 h + r
    const float* sp = rp.sin_tab + (size_t)ri.pos * 64 + it * 32 + 4 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 c = *reinterpret_cast<const float4*>(cp + 8 * q), s = *reinterpret_cast<const float4*>(sp + 8 * q);
#pragma unroll
      for (int hd = 0; hd < 2; ++hd) {
        const uint2 x = ex[(((((hd * 2 + 0) * 2 + wm) * TB + jt) * TA + it) * 4 + q) * 64 + lane];      // x[d]      (wave wn = 2 hd)
        const uint2 y = ex[(((((hd * 2 + 1) * 2 + wm) * TB + jt) * TA + it) * 4 + q) * 64 + lane];      // x[d + 64] (wave wn = 2 hd + 1)
        uint2 o0, o1;
        o0.x = f2bf_pk(rope_first(bf_lo(x.x), bf_lo(y.x), c.x, s.x), rope_first(bf_hi(x.x), bf_hi(y.x), c.y, s.y));
        o0.y = f2bf_pk(rope_first(bf_lo(x.y), bf_lo(y.y), c.z, s.z), rope_first(bf_hi(x.y), bf_hi(y.y), c.w, s.w));
        o1.x = f2bf_pk(rope_second(bf_lo(x.x), bf_lo(y.x), c.x, s.x), rope_second(bf_hi(x.x), bf_hi(y.x), c.y, s.y));
        o1.y = f2bf_pk(rope_second(bf_lo(x.y), bf_lo(y.y), c.z, s.z), rope_second(bf_hi(x.y), bf_hi(y.y), c.w, s.w));
        *reinterpret_cast<uint2*>(dst + hd * 128 + 8 * q) = o0;
        *reinterpret_cast<uint2*>(dst + hd * 128 + 64 + 8 * q) = o1;
      }
    }
  }
}

// NWV = 8: waves 4 (n) x 2 (m), wave tile 64 x MT2*16 (2 x MT2/2 tiles of 32x32), two waves per SIMD.  NWV = 4: waves 2 x 2, wave tile
// 128 x MT2*16 (4 x MT2/2 tiles), ONE wave per SIMD with up to 512 registers: a third fewer LDS fragment bytes per flop and no two waves
// contending for a SIMD's matrix pipe and issue slots (measured 2-5 % slower; the launcher instantiates NWV = 8 only).
// SPLITK (round 5: the N = 4096 projections of 257-2000-token forwards in fp8, whose tile grids fill a fifth to a half of a round): the grid is
// tiles x n_split, part z accumulates its share of the 256-k units and stores SCALED fp32 partial sums to slab z of Cv ([z][M][N]); the reduce
// kernels of every other split form finish the job (residual + RMSNorm + the next projection's e4m3 rows).
template <int EPI, int MT2, int NWV = 8, bool SPLITK = false>
__global__ __launch_bounds__(NWV * 64, 1) void gemm_ring_mx_kernel(const void* __restrict__ X, const void* __restrict__ W, const float* __restrict__ sx,
                                                              const float* __restrict__ sw, void* __restrict__ Cv, int M, int N, int K, int ldc,
                                                              int tiles_n, int tiles_m, int GM, int pk, RopeEpi rope = RopeEpi{}, int n_split = 1) {
  static_assert(!SPLITK || EPI == EPI_F32, "split-K parts leave fp32 slabs");
  constexpr int BT = 256, RB = 64;                                // 64-byte LDS rows = 64 k of e4m3 per stage
  constexpr int XR = 2 * MT2 * 16;                               // token rows per workgroup (256 or 128)
  constexpr int WP = 16 / NWV, XP = XR / (16 * NWV), NP = WP + XP;   // DMA pieces (16 rows x 64 B) per wave per k-step
  constexpr int STAGE = (BT + XR) * RB;
  constexpr int TA = 16 / NWV, TB = MT2 / 2;                     // 32x32 tiles per wave: weight rows x token rows (NWV/2 waves along n, 2 along m)
  static_assert(XP >= 1, "token tile too small for this wave count");
  constexpr int NR = 2 * (TA + TB);                              // ds_read_b128 per wave per k-step
  constexpr int NMF = TA * TB;                                   // MFMAs per wave per k-step
  constexpr int NIT = NR + NP;                                   // reads + DMA pieces placed between the MFMAs of a k-step
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, h = lane >> 5;
  const int nwg = tiles_n * tiles_m;
  int bid = blockIdx.x, zpart = 0;
  if constexpr (SPLITK) { zpart = bid / nwg; bid -= zpart * nwg; }
  {
    const int q = nwg / 8, r = nwg % 8, x = bid % 8;             // XCD-contiguous runs of tiles (see gemm_ring_kernel)
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
  }
  const int band = bid / (GM * tiles_n), rem = bid % (GM * tiles_n);
  const int band_rows = min(GM, tiles_m - band * GM);
  const int tn = rem / band_rows, tm = band * GM + rem % band_rows;
  const int n0 = tn * BT, m0 = tm * XR;
  const int wn = wave >> 1, wm = wave & 1;                        // NWV/2 waves along the weight rows, 2 along the token rows
  int ks0 = 0, nks = K / 64;                                      // launcher: K % 256 == 0; nks = END of this part's k-steps
  if constexpr (SPLITK) {
    const int units = nks >> 2;                                   // launcher: n_split <= units
    ks0 = 4 * (int)((long long)zpart * units / n_split);
    nks = 4 * (int)((long long)(zpart + 1) * units / n_split);
  }

  auto swz = [](int row) { return (0xD2 >> (((row >> 2) & 3) * 2)) & 3; };   // f = {2,0,1,3}
  unsigned woff[WP], xoff[XP];
  int m0w_[WP], m0x_[XP];
  const unsigned lbase = lds_addr(smem);
#pragma unroll
  for (int j = 0; j < WP; ++j) {
    const int row = (wave * WP + j) * 16 + (lane >> 2), gr = min(n0 + row, N - 1);
    woff[j] = (pk ? (unsigned)(gr >> 1) * (unsigned)(K * 2) + (gr & 1) * 64 : (unsigned)gr * (unsigned)K) + (((lane & 3) ^ swz(row)) * 16);
    m0w_[j] = __builtin_amdgcn_readfirstlane((int)lbase + (wave * WP + j) * 1024);
  }
#pragma unroll
  for (int j = 0; j < XP; ++j) {
    const int row = (wave * XP + j) * 16 + (lane >> 2), gr = min(m0 + row, M - 1);
    xoff[j] = (pk ? (unsigned)(gr >> 1) * (unsigned)(K * 2) + (gr & 1) * 64 : (unsigned)gr * (unsigned)K) + (((lane & 3) ^ swz(row)) * 16);
    m0x_[j] = __builtin_amdgcn_readfirstlane((int)lbase + BT * RB + (wave * XP + j) * 1024);
  }
  const unsigned long long wb = (unsigned long long)W, xb = (unsigned long long)X;
  const unsigned long long kadv = pk ? 2 * RB : RB;               // packed operands: full 128-byte lines per DMA piece (see gemm_ring_kernel)
  // fragment addresses: lane (r32, h) reads chunks 2h and 2h+1 of row r32 of each 32-row tile
  const unsigned lp_lo = r32 * RB + (((2 * h) ^ swz(r32)) * 16), lp_hi = r32 * RB + (((2 * h + 1) ^ swz(r32)) * 16);
  unsigned aAl[2], aAh[2], aBl[2], aBh[2];                        // stages {0,1} and {2,3}
  aAl[0] = lbase + (wn * TA * 32) * RB + lp_lo;  aAh[0] = lbase + (wn * TA * 32) * RB + lp_hi;
  aBl[0] = lbase + BT * RB + (wm * TB * 32) * RB + lp_lo;  aBh[0] = lbase + BT * RB + (wm * TB * 32) * RB + lp_hi;
  aAl[1] = aAl[0] + 2 * STAGE; aAh[1] = aAh[0] + 2 * STAGE; aBl[1] = aBl[0] + 2 * STAGE; aBh[1] = aBh[0] + 2 * STAGE;

  f32x16_t acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  u32x4_t fal[2][TA], fah[2][TA], fbl[2][TB], fbh[2][TB];
  const unsigned unit_scale = 0x7f7f7f7fu;                        // E8M0 1.0 for every 32-k block of both operands

  auto dma_piece = [&](int q, int ks, int d) {
    if (d < WP) ATS_DMA16(woff[d % WP], wb + (unsigned long long)ks * kadv, m0w_[d % WP] + q * STAGE);
    else        ATS_DMA16(xoff[(d - WP) % XP], xb + (unsigned long long)ks * kadv, m0x_[(d - WP) % XP] + q * STAGE);
  };
  auto read_one = [&](int q, int buf, int r) {                    // read r of stage q into register buffer buf (all compile-time after unrolling)
    const int so = (q & 1) * STAGE, tile = r >> 1;
    if (r < 2 * TA) {
      if (r & 1) ATS_DS_READ_B128(fah[buf][tile], aAh[q >> 1], so + tile * 2048);
      else       ATS_DS_READ_B128(fal[buf][tile], aAl[q >> 1], so + tile * 2048);
    } else {
      const int tb = tile - TA;
      if (r & 1) ATS_DS_READ_B128(fbh[buf][tb], aBh[q >> 1], so + tb * 2048);
      else       ATS_DS_READ_B128(fbl[buf][tb], aBl[q >> 1], so + tb * 2048);
    }
  };
#define ATS_CAT8(lo, hi) __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7)
  // one k-step: MFMAs of stage Q from register buffer Q&1; item t (reads of stage Q+1, then the DMA pieces of k-step ks+4 into
  // stage Q) is issued in front of MFMA t*NMF/NIT; VM = vmcnt to wait for before the closing barrier (-1: no wait, no barrier)
#define ATS_MX_SEGMENT(Q, DMA, RD, VM, ks)                                                               \
  {                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < TA; ++i) _Pragma("unroll") for (int j = 0; j < TB; ++j) {      \
      const int idx = i * TB + j;                                                                        \
      _Pragma("unroll") for (int t = 0; t < NIT; ++t) {                                                  \
        if (t * NMF / NIT == idx) {                                                                      \
          if (t < NR) { if (RD) read_one(((Q) + 1) & 3, ((Q) + 1) & 1, t); }                             \
          else if (DMA) dma_piece((Q), (ks) + 4, t - NR);                                                \
        }                                                                                                \
      }                                                                                                  \
      if constexpr (NWV == 4)                                                                            \
        ATS_MFMA_MX_A(acc[i][j], ATS_CAT8(fal[(Q) & 1][i], fah[(Q) & 1][i]), ATS_CAT8(fbl[(Q) & 1][j], fbh[(Q) & 1][j]), unit_scale); \
      else                                                                                               \
        ATS_MFMA_MX(acc[i][j], ATS_CAT8(fal[(Q) & 1][i], fah[(Q) & 1][i]), ATS_CAT8(fbl[(Q) & 1][j], fbh[(Q) & 1][j]), unit_scale); \
    }                                                                                                    \
    if (RD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                           \
    if ((VM) >= 0) {                                                                                     \
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((VM) < 0 ? 0 : (VM)) : "memory");                         \
      asm volatile("s_barrier" ::: "memory");                                                            \
    }                                                                                                    \
  }

  // prologue: k-steps 0..3 into stages 0..3; fragments of k-step 0
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int d = 0; d < NP; ++d) dma_piece(q, ks0 + q, d);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NP) : "memory");
  asm volatile("s_barrier" ::: "memory");
#pragma unroll
  for (int r = 0; r < NR; ++r) read_one(0, 0, r);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
  asm volatile("s_barrier" ::: "memory");

  int ks = ks0;
  for (; ks + 4 < nks; ks += 4) {
    ATS_MX_SEGMENT(0, true, true, 2 * NP, ks);
    ATS_MX_SEGMENT(1, true, true, 2 * NP, ks + 1);
    ATS_MX_SEGMENT(2, true, true, 2 * NP, ks + 2);
    ATS_MX_SEGMENT(3, true, true, 2 * NP, ks + 3);
  }
  ATS_MX_SEGMENT(0, false, true, NP, ks);
  ATS_MX_SEGMENT(1, false, true, 0, ks + 1);
  ATS_MX_SEGMENT(2, false, true, -1, ks + 2);
  ATS_MX_SEGMENT(3, false, false, -1, ks + 3);
#undef ATS_MX_SEGMENT
#undef ATS_CAT8
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // 16-pass MFMA results -> VALU reads (the compiler cannot see the asm MFMAs)

  if constexpr (EPI == EPI_QKV_ROPE) {
    mx_qkv_rope_epilogue<TA, TB>(acc, reinterpret_cast<bf16_t*>(Cv), M, N, ldc, m0, n0, wave, lane, sx, sw, rope, smem);
  } else {
    const int m0w = m0 + wm * (TB * 32), n0w = n0 + wn * (TA * 32);
    if constexpr (SPLITK) mx_epilogue<EPI_F32, TA, TB>(acc, reinterpret_cast<float*>(Cv) + (size_t)zpart * M * N, M, N, N, m0w, n0w, lane, sx, sw, 0);
    else                  mx_epilogue<EPI, TA, TB>(acc, Cv, M, N, ldc, m0w, n0w, lane, sx, sw, pk);
  }
}

// Arena of the split-K tail for callers that bring none (the C ABI's plain atspeed_gemm; a model's forwards pass their own, internal.h
// SkArena): per device, shared by all host threads.  Launches that use it must not overlap: the caller holds `g_sk_mu` from here until its
// kernel is enqueued, and a launch on ANOTHER stream than the previous one first waits for that one (event).  Never used while `st` is
// capturing (launch_big falls back to the plain grid): the cross-stream wait and the first-use allocation are illegal inside a capture, and
// a replayed graph would bypass the mutex.
struct SkShared { SkArena a; hipStream_t last = nullptr; hipEvent_t ev = nullptr; bool have_last = false; bool failed = false; };
static std::mutex g_sk_mu;
static SkShared g_sk[ATS_MAX_DEVICES];
static const SkArena* sk_shared_arena(hipStream_t st) {              // g_sk_mu held by the caller; nullptr = no arena (launch without the tail)
  const int d = ats_cur_device();
  if (d < 0) return nullptr;
  SkShared& w = g_sk[d];
  if (w.failed) return nullptr;
  if (!w.a.ws) {
    float* ws = nullptr; int* cnt = nullptr; hipEvent_t ev = nullptr;
    if (hipMalloc((void**)&ws, ATS_SK_ARENA_BYTES) != hipSuccess || hipMalloc((void**)&cnt, ATS_SK_ARENA_COUNTERS * sizeof(int)) != hipSuccess ||
        hipMemset(cnt, 0, ATS_SK_ARENA_COUNTERS * sizeof(int)) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();                                     // no room for the arena: this device runs its thin grids without the tail
      if (ws) hipFree(ws);
      if (cnt) hipFree(cnt);
      w.failed = true;
      return nullptr;
    }
    w.a.ws = ws; w.a.cnt = cnt; w.ev = ev;
  }
  if (w.have_last && w.last != st) {
    const hipStream_t prev = w.last;
    w.last = st;                                                   // before anything that can fail: a destroyed `prev` must not be recorded on again
    if (hipEventRecord(w.ev, prev) == hipSuccess) { if (hipStreamWaitEvent(st, w.ev, 0) != hipSuccess) (void)hipGetLastError(); }
    else (void)hipGetLastError();                                  // the previous stream is gone: so is its work
  }
  w.last = st; w.have_last = true;
  return &w.a;
}
static bool stream_is_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
  return cs != hipStreamCaptureStatusNone;
}

template <int EPI>
int launch_big(const bf16_t* x, const bf16_t* w, void* c, int m, int n, int k, int ldx, int ldc, hipStream_t st, int pk, const RopeEpi& rope = RopeEpi{},
               float* lse_part = nullptr, const unsigned char* tile_store = nullptr, const SkArena* arena = nullptr) {
  // height of the tile bands an XCD walks (its ~32 concurrent tiles are gm token-tile rows x 32 / gm weight panels).  4 is the measured minimum of
  // the L2-miss traffic (2 / 4 / 8: 12.45 / 9.57 / 11.96 GB per gate_up launch, round 2; 6 -- the 5.7 x 5.7 optimum of the panel count -- in round 6,
  // profiles/r06_ring_raster_ab.txt).  -DATS_RING_GM=n builds a probe library for that A/B (tools/r06_call.sh raster); not a run-time switch.
#ifndef ATS_RING_GM
#define ATS_RING_GM 4
#endif
  constexpr int gm = ATS_RING_GM;
  const int tiles_n = (n + 255) / 256;
  constexpr int LDS8 = EPI == EPI_F32_LSE ? 136 * 1024 : 128 * 1024, LDS4 = EPI == EPI_F32_LSE ? 100 * 1024 : 96 * 1024;
  static thread_local AtsPerDeviceFlag attr_flag;
  bool& attr_done = attr_flag.cur();
  if (!attr_done) {
    ATS_HIP(hipFuncSetAttribute((const void*)gemm_ring_kernel<EPI, 8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS8));
    ATS_HIP(hipFuncSetAttribute((const void*)gemm_ring_kernel<EPI, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS4));
    ATS_HIP(hipFuncSetAttribute((const void*)gemm_ring_kernel<EPI, 8, false, false, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS8));
    ATS_HIP(hipFuncSetAttribute((const void*)gemm_ring_kernel<EPI, 4, false, false, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS4));
    attr_done = true;
  }
  const int t256 = tiles_n * ((m + 255) / 256), t128 = tiles_n * ((m + 127) / 128);
  const BigChoice ch = big_choose(t256, t128, k);
  const bool use256 = ch.rows256;
  const float* none = nullptr;
  if (ch.sk.on && (arena != nullptr || !stream_is_capturing(st))) {
    // split-K tail: n_dp whole tiles + G workgroups that share the k-steps of the remaining ones evenly.  With the caller's arena plain
    // stream order is all there is; the shared one needs bookkeeping + enqueue as one step (see SkShared)
    std::unique_lock<std::mutex> lk(g_sk_mu, std::defer_lock);
    if (!arena) { lk.lock(); arena = sk_shared_arena(st); }
    if (arena) {
      const SkTail tail{ch.sk.n_dp, ch.sk.G, ch.sk.U, ch.sk.TU, arena->ws, arena->cnt};
      const int grid = ch.sk.n_dp + ch.sk.G;
      if (use256) hipLaunchKernelGGL((gemm_ring_kernel<EPI, 8, false, false, 4, true>), dim3(grid), dim3(512), LDS8, st, (const void*)x, (const void*)w, none, none, c, m, n, k, ldx, ldc, tiles_n, (m + 255) / 256, gm, 1, lse_part, tile_store, pk, rope, tail);
      else        hipLaunchKernelGGL((gemm_ring_kernel<EPI, 4, false, false, 4, true>), dim3(grid), dim3(512), LDS4, st, (const void*)x, (const void*)w, none, none, c, m, n, k, ldx, ldc, tiles_n, (m + 127) / 128, gm, 1, lse_part, tile_store, pk, rope, tail);
      ATS_LAUNCH_CHECK();
      ats_count_path(ATS_PATH_RING_SK);
      return ATSPEED_OK;
    }
  }
  // (a four-wave form, one wave per SIMD with 128 x 128 per wave and the accumulators in AGPRs, was 3 % slower on every projection and is
  // not instantiated any more: profiles/README.md)
  if (use256) hipLaunchKernelGGL((gemm_ring_kernel<EPI, 8, false>), dim3(t256), dim3(512), LDS8, st, (const void*)x, (const void*)w, none, none, c, m, n, k, ldx, ldc, tiles_n, (m + 255) / 256, gm, 1, lse_part, tile_store, pk, rope);
  else        hipLaunchKernelGGL((gemm_ring_kernel<EPI, 4, false>), dim3(t128), dim3(512), LDS4, st, (const void*)x, (const void*)w, none, none, c, m, n, k, ldx, ldc, tiles_n, (m + 127) / 128, gm, 1, lse_part, tile_store, pk, rope);
  ATS_LAUNCH_CHECK();
  ats_count_path(ATS_PATH_RING);
  return ATSPEED_OK;
}

// lse[row] = log sum over the row's per-tile (max, sum exp) partials of the fused lm_head epilogue; one wave per row
__global__ __launch_bounds__(256) void lse_combine_kernel(const float2* __restrict__ part, int rows, int tiles_n, float* __restrict__ lse) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float mx = -INFINITY, sm = 0.f;
  for (int t = lane; t < tiles_n; t += 64) {
    const float2 p = part[(size_t)row * tiles_n + t];
    const float nm = fmaxf(mx, p.x);
    sm = (nm > -INFINITY) ? sm * __expf(mx - nm) + p.y * __expf(p.x - nm) : 0.f;
    mx = nm;
  }
  const float gm = wave_max_f32(mx);
  sm = (mx > -INFINITY) ? sm * expf(mx - gm) : 0.f;
  sm = wave_sum_f32(sm);
  if (lane == 0) lse[row] = gm + logf(sm);
}

// lm_head over the batched rows with the normaliser fused into the epilogue (gemm_ring_kernel<EPI_F32_LSE>)
int launch_big_lse(const bf16_t* x, const bf16_t* w, float* c, int m, int n, int k, int ldx, int ldc, float* part, const unsigned char* tile_store,
                   float* lse, hipStream_t st, int pk, const SkArena* arena) {
  const int tiles_n = (n + 255) / 256;
  ATS_TRY(launch_big<EPI_F32_LSE>(x, w, (void*)c, m, n, k, ldx, ldc, st, pk, RopeEpi{}, part, tile_store, arena));
  lse_combine_kernel<<<(m + 3) / 4, 256, 0, st>>>(reinterpret_cast<const float2*>(part), m, tiles_n, lse);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

// split-K reduce + residual add + RMSNorm of the updated row, one workgroup per token row:
//   h[m][:] += sum_z partial[z][m][:]        (the o_proj / down_proj epilogue)
//   xn[m][:] = w * (h[m][:] * rsqrt(mean(h^2) + eps))   (the NEXT op's input norm)
// Saves one launch and one read of h per projection; numerics identical to the unfused pair
// (statistics are taken from the stored, dtype-rounded h).
// QUANT (the consumer is a W8A8 projection, one user's fp8 forwards): the normalised row also leaves as OCP e4m3 with its per-token scale --
// what ats_rmsnorm_quant_fp8 makes of the same h, bit for bit (amax over the bf16-rounded outputs, scale = amax / 448) --; xn may then be null.
template <typename T, int NPT, int V, bool QUANT = false>
__global__ __launch_bounds__(1024) void splitk_resid_rmsnorm_kernel(const float* __restrict__ partial, T* __restrict__ h,
                                                                    const T* __restrict__ norm_w, T* __restrict__ xn, int M, int N,
                                                                    int ldh, int splits, float eps, int pk,
                                                                    unsigned char* __restrict__ q = nullptr, float* __restrict__ qscale = nullptr) {
  // thread t owns columns (t + i*1024) * V .. + V-1, i < NPT / V   (V = 4: 16-byte slab loads; needs N % 4 == 0)
  __shared__ float red[16];
  const int m = blockIdx.x;
  const size_t mn = (size_t)M * N;
  float vals[NPT];
  float ss = 0.f;
  const float* prow = partial + (size_t)m * N;
#pragma unroll
  for (int i = 0; i < NPT / V; ++i) {
    const int n = (threadIdx.x + i * 1024) * V;
    float acc[V];
    if (n < N) sum_slabs<V>(prow + n, mn, splits, acc);
#pragma unroll
    for (int j = 0; j < V; ++j) vals[i * V + j] = n < N ? acc[j] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < NPT / V; ++i) {
    const int n = (threadIdx.x + i * 1024) * V;
    if (n < N) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float v = vals[i * V + j];
        if constexpr (sizeof(T) == 2) v = bf2f(f2bf(v));
        v = Elt<T>::load(h + (size_t)m * ldh + n + j) + v;
        Elt<T>::store(h + (size_t)m * ldh + n + j, v);
        if constexpr (sizeof(T) == 2) v = bf2f(f2bf(v));
        ss += v * v;
        vals[i * V + j] = v;
      }
    }
  }
  ss = wave_sum_f32(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) tot += red[w];
  const float rs = rsqrtf(tot / (float)N + eps);
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NPT / V; ++i) {
    const int n = (threadIdx.x + i * 1024) * V;
    if (n < N) {
      T* xo = xn + ats_opnd_idx<sizeof(T)>(pk, m, n, N);      // xn: the next projection's operand (V <= 4 consecutive columns stay inside one 64-byte block)
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float v = vals[i * V + j] * rs;
        if constexpr (sizeof(T) == 2) v = bf2f(f2bf(v));
        v = Elt<T>::load(norm_w + n + j) * v;
        if constexpr (QUANT) {
          v = bf2f(f2bf(v));                                  // the stored, 16-bit-rounded output is what gets quantised
          vals[i * V + j] = v;
          amax = fmaxf(amax, fabsf(v));
          if (xn) Elt<T>::store(xo + j, v);
        } else Elt<T>::store(xo + j, v);
      }
    }
  }
  if constexpr (QUANT) {
    static_assert(V == 4, "four e4m3 bytes per store");
    amax = wave_max_f32(amax);
    __syncthreads();                                          // red[] was read by everyone above
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
    __syncthreads();
    amax = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) amax = fmaxf(amax, red[w]);
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (threadIdx.x == 0) qscale[m] = sc;
#pragma unroll
    for (int i = 0; i < NPT / V; ++i) {
      const int n = (threadIdx.x + i * 1024) * V;
      if (n < N) {
        float f[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = fminf(fmaxf(vals[i * V + j] * inv, -448.f), 448.f);
        int w4 = 0;
        w4 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w4, false);
        w4 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w4, true);
        *reinterpret_cast<int*>(q + ats_opnd_idx<1>(pk, m, n, N)) = w4;
      }
    }
  }
}

template <int EPI>
int launch_big_fp8(const unsigned char* x, const float* sx, const unsigned char* w, const float* sw, void* c, int m, int n, int k,
                   int ldc, hipStream_t st, int pk, const RopeEpi& rope = RopeEpi{}) {
  constexpr int gm = 4;
  const int tiles_n = (n + 255) / 256;
  static thread_local AtsPerDeviceFlag attr_flag;
  bool& attr_done = attr_flag.cur();
  if (!attr_done) {
    ATS_HIP(hipFuncSetAttribute((const void*)gemm_ring_kernel<EPI, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    ATS_HIP(hipFuncSetAttribute((const void*)gemm_ring_kernel<EPI, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    attr_done = true;
  }
  const int t256 = tiles_n * ((m + 255) / 256), t128 = tiles_n * ((m + 127) / 128);
  // block-scaled MFMA form (2x the flops per cycle); ATSPEED_FP8_MX=0 keeps the non-scaled v_mfma_f32_16x16x32_fp8_fp8 kernel for A/B runs
  static const int use_mx = env_int("ATSPEED_FP8_MX", 1);
  if (use_mx) {
    static thread_local AtsPerDeviceFlag mx_flag;
    bool& mx_done = mx_flag.cur();
    if (!mx_done) {
      ATS_HIP(hipFuncSetAttribute((const void*)gemm_ring_mx_kernel<EPI, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
      ATS_HIP(hipFuncSetAttribute((const void*)gemm_ring_mx_kernel<EPI, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      mx_done = true;
    }
    // (the four-wave form of this kernel, NWV = 4, was 2-5 % slower and is not instantiated any more)
    if (big_use_256_rows(t256, t128))
      hipLaunchKernelGGL((gemm_ring_mx_kernel<EPI, 8>), dim3(t256), dim3(512), 128 * 1024, st, (const void*)x, (const void*)w, sx, sw, c, m, n, k, ldc, tiles_n, (m + 255) / 256, gm, pk, rope);
    else
      hipLaunchKernelGGL((gemm_ring_mx_kernel<EPI, 4>), dim3(t128), dim3(512), 96 * 1024, st, (const void*)x, (const void*)w, sx, sw, c, m, n, k, ldc, tiles_n, (m + 127) / 128, gm, pk, rope);
    ATS_LAUNCH_CHECK();
    ats_count_path(ATS_PATH_FP8_RING);
    return ATSPEED_OK;
  }
  if (big_use_256_rows(t256, t128))
    hipLaunchKernelGGL((gemm_ring_kernel<EPI, 8, true>), dim3(t256), dim3(512), 128 * 1024, st, (const void*)x, (const void*)w, sx, sw, c, m, n, k, k, ldc, tiles_n, (m + 255) / 256, gm, 1, (float*)nullptr, (const unsigned char*)nullptr, pk, rope);
  else
    hipLaunchKernelGGL((gemm_ring_kernel<EPI, 4, true>), dim3(t128), dim3(512), 96 * 1024, st, (const void*)x, (const void*)w, sx, sw, c, m, n, k, k, ldc, tiles_n, (m + 127) / 128, gm, 1, (float*)nullptr, (const unsigned char*)nullptr, pk, rope);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}


// =====================================================================================
// One user's WIDE projections (qkv, gate_up, lm_head) at 33-128 tokens, without split-K: the weight-streaming regime.
//
//   C^T[n][m] = sum_k W[n][k] X[m][k]: a workgroup owns BN weight rows and the WHOLE K for all (<= BM) token rows, so its epilogue
//   (store / SwiGLU / fp32) applies directly: no fp32 slabs, no reduce launch, one kernel per projection.  Used for gate_up and the
//   lm_head (wdma_applies); hipBLASLt's kernels for these shapes have the same form (MT128x32x128 / MT96x128x128, one launch).
//
// What makes this stream (measured round 3, profiles/r03_one_user_*.txt): a workgroup must keep several k-tiles of loads in flight, or
// every tile costs a memory round trip (~1.5 us).  Register prefetch cannot do it on this compiler (hipcc drains vmcnt across loop
// iterations), so the tiles travel by LDS-DMA issued from INLINE ASM -- the compiler does not know these
// vector-memory operations exist, hence inserts no waits for them -- into a ring of NST stages of (BM + BN) rows x 128 bytes; the wait
// for "tile kt has landed" is a hand-counted s_waitcnt vmcnt((NST-2) * pieces per wave) followed by s_barrier.  The MFMA work of a stage
// is ordinary compiler-scheduled code (fragment reads from LDS that is stable between two barriers): at 100 tokens it is a third of the
// time the stage's bytes need, so nothing has to be interleaved by hand.  LDS image of a stage: X rows then W rows, chunk c of row r at
// position c ^ (r & 7) (conflict-free for the 16-row fragment reads), applied on the per-lane SOURCE address since the DMA's destination is
// lane-linear (1 KB piece = 8 rows x 128 bytes, lane l -> row l >> 3, position l & 7).
// SPLIT: the grid's y dimension divides K (64-k tiles z * n / splits .. (z+1) * n / splits); part z stores its fp32 partial sums to slab z of
// Cv ([z][M][N]) and the caller's reduce kernel finishes the job (qkv: the RoPE kernel sums the slabs; down: reduce + residual + RMSNorm).
// WM: waves along the token rows (2 x WM waves per workgroup).  2 x 2 up to 128 tokens, where the MFMAs are free (ablation builds,
// profiles/r03_one_user_stream_ceiling.txt); 2 x 4 for the 256-row tile, whose 64 MFMAs per k-tile and wave at one wave per SIMD showed
// (73.8 us against 60 without them at 225 tokens): two waves per SIMD overlap one's fragment reads with the other's MFMAs.
// F8 (round 5; BASELINE config 5 in the reference's own regime, one user per call, inference.py:86-91 `load_in_8bit`): the same ring on e4m3
// operands with per-row scales (W8A8: W rows scaled per output row, X rows per token).  A 128-byte LDS row then holds 128 k, the DMA pieces,
// the LDS image and the waits are unchanged (row bytes = K instead of 2 K), and a stage's MFMA work is ONE block-scaled
// v_mfma_scale_f32_16x16x128_f8f6f4 per 16 x 16 tile with unit E8M0 scales (tools/probe/mx16_probe.hip: lane l holds row l & 15; twice the
// flops per cycle of the bf16 form, so a stage costs the same cycles for twice the k): the weight bytes per launch halve.  A lane's 32 bytes are
// chunks g and 4 + g of its row -- not 2g, 2g + 1: with the c ^ (r & 7) swizzle the lanes {0-3,12-15,20-27} of a ds_read_b128 group would
// meet two to a bank -- the same k permutation on both operands, so the product is unchanged.  The fp32 scales multiply the accumulators
// (acc * sx[m] * sw[n]) before the epilogue, also on the fp32 split-K partials (the sum is linear), so every consumer is the bf16 form's.
typedef int i32x8_t __attribute__((ext_vector_type(8)));
template <int BM, int BN, int NST, int EPI, bool SPLIT = false, int WM = 2, bool F8 = false>
__global__ __launch_bounds__(128 * WM) void gemm_wdma_kernel(const void* __restrict__ X, const void* __restrict__ W, void* __restrict__ Cv,
                                                             int M, int N, int K, int ldx, int ldc, int pk, int n_split,
                                                             const float* __restrict__ sx = nullptr, const float* __restrict__ sw = nullptr,
                                                             RopeEpi rope = RopeEpi{}) {
  // EPI_QKV_ROPE (round 6; one user's W8A8 qkv projection at head_dim 128, 64-row weight tiles, no split): RoPE and the KV-cache scatter in
  // the epilogue, as in the ring kernels' EPI_QKV_ROPE.  A rotation pairs columns d and d + 64 of a head, and a 64-row tile cannot hold a whole
  // head; since the DMA's per-lane source address is free, tile t (0 / 1) of a head takes the weight rows {32 t + 16 wn + q, 64 + 32 t + 16 wn + q}
  // (wn = the wave's half of the tile, q < 16) laid out so that a lane's accumulators acc[0][j][r] and acc[1][j][r] ARE the pair (d, d + 64):
  // no exchange between lanes or waves.  Numerics = the plain store + rope_kv_segs_vec_kernel: the projection rounded to the 16-bit type,
  // the rotation in fp32 on those values (rope_first / rope_second), one more rounding -- bit-identical (tests/test_closures_gpu.py).
  static_assert(EPI != EPI_QKV_ROPE || (F8 && !SPLIT && BN == 64), "RoPE epilogue: the W8A8 no-split form with 64-row weight tiles");
  constexpr int RB = 128, STAGE = (BM + BN) * RB, NWAVE = 2 * WM;
  constexpr int ESZ = F8 ? 1 : 2, BK = RB / ESZ;                       // k per stage: 64 (16-bit) or 128 (e4m3)
  constexpr int NPIECE = (BM + BN) / 8, NP = NPIECE / NWAVE;           // 1 KB pieces per stage; per wave
  constexpr int NI = BN / 2 / 16, MI = BM / WM / 16;                   // wave tile (2 x WM waves): BN/2 weight rows x BM/WM token rows
  static_assert(NPIECE % NWAVE == 0 && (NST - 2) * NP <= 63 && BM % (16 * WM) == 0, "pieces per wave / vmcnt range");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave / WM, wm = wave % WM, lq = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * BN;
  int kt0 = 0, n_kt = K / BK;                                          // launcher: K % BK == 0; this part's tiles are kt0 .. kt0 + n_kt - 1
  if constexpr (SPLIT) {
    const int all = n_kt, z = blockIdx.y;
    kt0 = (int)((long long)z * all / n_split);
    n_kt = (int)((long long)(z + 1) * all / n_split) - kt0;
  }
  const unsigned lbase = lds_addr(smem);

  // this wave's pieces: piece index p = wave * NP + j covers LDS rows 8p .. 8p+7 of a stage (rows < BM: X, then W)
  unsigned voff[NP];
  int m0p[NP];
  bool is_w[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int piece = wave * NP + j, row = piece * 8 + (lane >> 3), pos = lane & 7, c = pos ^ (row & 7);
    const bool w = row >= BM;                                          // uniform per piece (BM % 8 == 0)
    int wr = n0 + row - BM;                                            // weight row of this LDS row
    if constexpr (EPI == EPI_QKV_ROPE) {
      const int rw = row - BM;                                         // LDS row (wn, i, q) = (rw >> 5, (rw >> 4) & 1, rw & 15) -> column i * 64 + 32 t + 16 wn + q of the head
      wr = (int)(blockIdx.x >> 1) * 128 + ((rw >> 4) & 1) * 64 + (int)(blockIdx.x & 1) * 32 + (rw >> 5) * 16 + (rw & 15);
    }
    const int gr = w ? min(wr, N - 1) : min(row, M - 1);
    const unsigned ldb = (w ? (unsigned)K : (unsigned)ldx) * ESZ;      // row bytes
    voff[j] = pk ? (unsigned)(gr >> 1) * (ldb * 2u) + (gr & 1) * 64 + (unsigned)(c >> 2) * 128 + (c & 3) * 16 : (unsigned)gr * ldb + c * 16;
    m0p[j] = __builtin_amdgcn_readfirstlane((int)lbase + piece * 1024);
    is_w[j] = __builtin_amdgcn_readfirstlane(piece * 8 >= BM ? 1 : 0) != 0;
  }
  const unsigned long long wb = (unsigned long long)W, xb = (unsigned long long)X;
  const unsigned long long kstep = pk ? 256 : 128;                     // bytes from one 128-byte tile of a row to the next
  auto issue = [&](int kt) {                                           // kt: tile index inside this part
    const int so = (kt % NST) * STAGE;
#pragma unroll
    for (int j = 0; j < NP; ++j) ATS_DMA16(voff[j], (is_w[j] ? wb : xb) + (unsigned long long)(kt0 + kt) * kstep, m0p[j] + so);
  };

  f32x4_t acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int t = 0; t < NST - 1; ++t) if (t < n_kt) issue(t);           // (a part shorter than the ring only ever waits vmcnt(0) below)
  for (int kt = 0; kt < n_kt; ++kt) {
    if (kt + NST - 2 < n_kt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * NP) : "memory");   // tiles kt+1 .. kt+NST-2 may still fly
    else                     asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");                            // everyone's pieces of tile kt; and stage (kt-1) % NST is read out
    if (kt + NST - 1 < n_kt) issue(kt + NST - 1);
    const unsigned char* sxl = smem + (kt % NST) * STAGE + (wm * (BM / WM)) * RB;
    const unsigned char* swl = smem + (kt % NST) * STAGE + BM * RB + (wn * (BN / 2)) * RB;
    if constexpr (F8) {
      i32x8_t wf[NI], xf[MI];                                          // chunks g (low 16 bytes) and 4 + g (high) of row lq of each 16-row tile
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(swl + swz<8>(i * 16 + lq, g)), hi = *reinterpret_cast<const u32x4_t*>(swl + swz<8>(i * 16 + lq, 4 + g));
        wf[i] = i32x8_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
      }
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(sxl + swz<8>(j * 16 + lq, g)), hi = *reinterpret_cast<const u32x4_t*>(sxl + swz<8>(j * 16 + lq, 4 + g));
        xf[j] = i32x8_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[i], xf[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {                                 // two k-steps of 32 per 128-byte row
        s16x8_t wf[NI], xf[MI];
#pragma unroll
        for (int i = 0; i < NI; ++i) wf[i] = *reinterpret_cast<const s16x8_t*>(swl + swz<8>(i * 16 + lq, ks * 4 + g));
#pragma unroll
        for (int j = 0; j < MI; ++j) xf[j] = *reinterpret_cast<const s16x8_t*>(sxl + swz<8>(j * 16 + lq, ks * 4 + g));
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < MI; ++j)
            acc[i][j] = ATS_MFMA_16x16x32(__builtin_bit_cast(bf16x8_t, wf[i]), __builtin_bit_cast(bf16x8_t, xf[j]), acc[i][j]);
      }
    }
  }

  // epilogue: acc[i][j][r] = C[m][n] with m = wm*BM/WM + j*16 + lq (token row), n = n0 + wn*BN/2 + i*16 + g*4 + r (weight row): a lane
  // holds four adjacent output columns of one token row
  const int nw = n0 + wn * (BN / 2);
  // weight row (= output column) of acc[i][.][r]
  auto wcol = [&](int i, int r) {
    if constexpr (EPI == EPI_QKV_ROPE) return (int)(blockIdx.x >> 1) * 128 + i * 64 + (int)(blockIdx.x & 1) * 32 + wn * 16 + g * 4 + r;
    else return nw + i * 16 + g * 4 + r;
  };
  if constexpr (F8) {                                                  // W8A8: per-token x per-output-row scales on the accumulators
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const float fx = sx[min(wm * (BM / WM) + j * 16 + lq, M - 1)];
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] *= fx * sw[min(wcol(i, r), N - 1)];
    }
  }
  if constexpr (EPI == EPI_QKV_ROPE) {
    static_assert(NI == 2, "a lane holds the pair (d, d + 64) in its two weight tiles");
    bf16_t* qkv = reinterpret_cast<bf16_t*>(Cv);
    const int H = rope.hidden;
    const int hb = (int)(blockIdx.x >> 1) * 128;                       // the head's first column in [q | k | v]
    const int sec = hb / H, fsec = hb - sec * H;                       // 0 q, 1 k, 2 v (uniform over the workgroup); the head's first column inside it
    const int d0 = (int)(blockIdx.x & 1) * 32 + wn * 16 + g * 4;       // pair index of r = 0: this lane owns d0 .. d0 + 3 and their partners 64 further
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const int gm = wm * (BM / WM) + j * 16 + lq;
      if (gm >= M) continue;
      const RowInfo ri = rope.rows[gm];
      const uint32_t a01 = f2bf_pk(acc[0][j][0], acc[0][j][1]), a23 = f2bf_pk(acc[0][j][2], acc[0][j][3]);     // x[d]: the projection's 16-bit outputs
      const uint32_t b01 = f2bf_pk(acc[1][j][0], acc[1][j][1]), b23 = f2bf_pk(acc[1][j][2], acc[1][j][3]);     // x[d + 64]
      if (sec == 2) {
        bf16_t* dst = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(ri.vc) + rope.layer_off) + (size_t)ri.slot * H + fsec + d0;
        *reinterpret_cast<uint2*>(dst) = make_uint2(a01, a23);
        *reinterpret_cast<uint2*>(dst + 64) = make_uint2(b01, b23);
      } else {
        const float4 c = *reinterpret_cast<const float4*>(rope.cos_tab + (size_t)ri.pos * 64 + d0);
        const float4 sn = *reinterpret_cast<const float4*>(rope.sin_tab + (size_t)ri.pos * 64 + d0);
        uint2 o0, o1;
        o0.x = f2bf_pk(rope_first(bf_lo(a01), bf_lo(b01), c.x, sn.x), rope_first(bf_hi(a01), bf_hi(b01), c.y, sn.y));
        o0.y = f2bf_pk(rope_first(bf_lo(a23), bf_lo(b23), c.z, sn.z), rope_first(bf_hi(a23), bf_hi(b23), c.w, sn.w));
        o1.x = f2bf_pk(rope_second(bf_lo(a01), bf_lo(b01), c.x, sn.x), rope_second(bf_hi(a01), bf_hi(b01), c.y, sn.y));
        o1.y = f2bf_pk(rope_second(bf_lo(a23), bf_lo(b23), c.z, sn.z), rope_second(bf_hi(a23), bf_hi(b23), c.w, sn.w));
        bf16_t* dst = (sec == 0 ? qkv + (size_t)gm * ldc
                                : reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(ri.kc) + rope.layer_off) + (size_t)ri.slot * H) + fsec + d0;
        *reinterpret_cast<uint2*>(dst) = o0;
        *reinterpret_cast<uint2*>(dst + 64) = o1;
      }
    }
    return;
  }
  if constexpr (SPLIT) {
    float* P = reinterpret_cast<float*>(Cv) + (size_t)blockIdx.y * M * N;
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const int gm = wm * (BM / WM) + j * 16 + lq;
      if (gm >= M) continue;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int gn = nw + i * 16 + g * 4;
        if (gn >= N) continue;
        float* C = P + (size_t)gm * N + gn;
        if (gn + 3 < N && (N & 3) == 0) *reinterpret_cast<float4*>(C) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        else
#pragma unroll
          for (int r = 0; r < 4; ++r) if (gn + r < N) C[r] = acc[i][j][r];
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int gm = wm * (BM / WM) + j * 16 + lq;
    if (gm >= M) continue;
    if constexpr (EPI == EPI_SWIGLU) {                                 // gate rows 32b .. 32b+15, up rows 32b+16 .. 32b+31: tiles (2q, 2q+1)
      bf16_t* C = reinterpret_cast<bf16_t*>(Cv);
#pragma unroll
      for (int q = 0; q < NI / 2; ++q) {
        const int gn = nw + q * 32 + g * 4;                            // gate row of r = 0
        if (gn + 16 >= N) continue;                                    // N % 32 == 0: the whole (gate, up) group is inside N or not at all
        uint32_t o[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const uint32_t gp = f2bf_pk(acc[2 * q][j][2 * h], acc[2 * q][j][2 * h + 1]), up = f2bf_pk(acc[2 * q + 1][j][2 * h], acc[2 * q + 1][j][2 * h + 1]);
          o[h] = f2bf_pk(ats_silu<false>(bf_lo(gp)) * bf_lo(up), ats_silu<false>(bf_hi(gp)) * bf_hi(up));
        }
        *reinterpret_cast<uint2*>(C + ats_opnd_idx<2>(pk, gm, (nw >> 1) + q * 16 + g * 4, ldc)) = make_uint2(o[0], o[1]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int gn = nw + i * 16 + g * 4;
        if (gn >= N) continue;
        if constexpr (EPI == EPI_F32) {
          float* C = reinterpret_cast<float*>(Cv) + (size_t)gm * ldc + gn;
          if (gn + 3 < N && (ldc & 3) == 0) *reinterpret_cast<float4*>(C) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
          else
#pragma unroll
            for (int r = 0; r < 4; ++r) if (gn + r < N) C[r] = acc[i][j][r];
        } else {
          static_assert(EPI == EPI_STORE || EPI == EPI_SWIGLU || EPI == EPI_F32 || EPI == EPI_QKV_ROPE, "store / fp32 / SwiGLU (the RoPE epilogue returned above)");
          bf16_t* C = reinterpret_cast<bf16_t*>(Cv) + (size_t)gm * ldc + gn;
          if (gn + 3 < N && (ldc & 3) == 0) *reinterpret_cast<uint2*>(C) = make_uint2(f2bf_pk(acc[i][j][0], acc[i][j][1]), f2bf_pk(acc[i][j][2], acc[i][j][3]));
          else
#pragma unroll
            for (int r = 0; r < 4; ++r) if (gn + r < N) C[r] = f2bf(acc[i][j][r]);
        }
      }
    }
  }
}

struct Plan { int bm; int bn; int splits; int k_per_split; };

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

// Occupancy first: tools/stream_bench.hip shows streamed bandwidth ~ resident streaming waves (up to ~770), and a
// workgroup keeps only one k-tile of loads in flight, so the plan aims at >= ~2 workgroups per CU (split-K when the
// M x N tiling alone is too coarse) before anything else.
template <typename T>
Plan make_plan(int m, int n, int k) {
  constexpr int BK = GemmTraits<T>::BK;
  Plan p;
  p.bm = m <= 16 ? 16 : (m <= 32 ? 32 : (m <= 64 ? 64 : 128));
  // measured on MI355X (tools/sweep_gemm.sh): ~3 workgroups per CU for the pure streaming shapes (M <= 64),
  // ~2 per CU above; the widest GEMM (gate_up) prefers 64-column tiles over split-K slabs at M > 128
  const int target_wgs = m <= 64 ? 768 : 512;       // (constants since round 6: ATSPEED_GEMM_TARGET_WGS / ATSPEED_GEMM_BN were tools/sweep_gemm.sh's, round 1)
  p.bn = (sizeof(T) == 2 && p.bm == 128 && m > 128 && n >= 16384 && n < 32000) ? 64 : 128;
  int tiles = ((m + p.bm - 1) / p.bm) * ((n + p.bn - 1) / p.bn);
  int ktiles = (k + BK - 1) / BK;
  int splits = 1;
  if (tiles * 4 < target_wgs * 3) {              // below 75 % of the target: split K
    splits = (target_wgs + tiles - 1) / tiles;
    int max_splits = ktiles / 4;                 // keep >= 4 k-tiles per split
    if (splits > max_splits) splits = max_splits;
    if (splits > 16) splits = 16;
    if (splits < 1) splits = 1;
  }
  int kps = ((ktiles + splits - 1) / splits) * BK;
  p.splits = (k + kps - 1) / kps;
  p.k_per_split = kps;
  return p;
}

struct FusedNorm { const void* w; void* xn; float eps; bool done; void* q = nullptr; float* qscale = nullptr; };   // q: also / instead the e4m3 row + per-token scale

// second pass of a split-K GEMM: sum the fp32 slabs and apply the epilogue (fused with the next RMSNorm for the residual projections)
template <typename T, int EPI>
int reduce_splits(const float* partial, void* c, int m, int n, int ldc, int splits, hipStream_t st, FusedNorm* fn, int pk) {
  const bool v4 = (n % 4) == 0 && (ldc % 4) == 0 && ((uintptr_t)partial & 15) == 0;
  if constexpr (EPI == EPI_RESID) {
    if constexpr (sizeof(T) == 2) {
      if (fn && fn->q && n <= 8192 && v4) {                     // W8A8 consumer: e4m3 row + scale (and xn if asked for)
        if (n <= 4096) splitk_resid_rmsnorm_kernel<T, 4, 4, true><<<m, 1024, 0, st>>>(partial, (T*)c, (const T*)fn->w, (T*)fn->xn, m, n, ldc, splits, fn->eps, pk, (unsigned char*)fn->q, fn->qscale);
        else           splitk_resid_rmsnorm_kernel<T, 8, 4, true><<<m, 1024, 0, st>>>(partial, (T*)c, (const T*)fn->w, (T*)fn->xn, m, n, ldc, splits, fn->eps, pk, (unsigned char*)fn->q, fn->qscale);
        ATS_LAUNCH_CHECK();
        fn->done = true;
        return ATSPEED_OK;
      }
    }
    if (fn && !fn->q && n <= 8192) {
      if (n <= 4096) {
        if (v4) splitk_resid_rmsnorm_kernel<T, 4, 4><<<m, 1024, 0, st>>>(partial, (T*)c, (const T*)fn->w, (T*)fn->xn, m, n, ldc, splits, fn->eps, pk);
        else    splitk_resid_rmsnorm_kernel<T, 4, 1><<<m, 1024, 0, st>>>(partial, (T*)c, (const T*)fn->w, (T*)fn->xn, m, n, ldc, splits, fn->eps, pk);
      } else {
        if (v4) splitk_resid_rmsnorm_kernel<T, 8, 4><<<m, 1024, 0, st>>>(partial, (T*)c, (const T*)fn->w, (T*)fn->xn, m, n, ldc, splits, fn->eps, pk);
        else    splitk_resid_rmsnorm_kernel<T, 8, 1><<<m, 1024, 0, st>>>(partial, (T*)c, (const T*)fn->w, (T*)fn->xn, m, n, ldc, splits, fn->eps, pk);
      }
      ATS_LAUNCH_CHECK();
      fn->done = true;
      return ATSPEED_OK;
    }
  }
  size_t outs = EPI == EPI_SWIGLU ? (size_t)m * (n / 2) : (size_t)m * n;
  if (v4) splitk_reduce_kernel<T, EPI, 4><<<(unsigned)((outs / 4 + 255) / 256), 256, 0, st>>>(partial, c, m, n, ldc, splits, pk);
  else    splitk_reduce_kernel<T, EPI, 1><<<(unsigned)((outs + 255) / 256), 256, 0, st>>>(partial, c, m, n, ldc, splits, pk);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

// One user's tokens through the ring kernel: tiles x splits <= 256 workgroups (one per CU: the ring takes 96-128 KB of LDS).
// 257-512 tokens (the first verification of a long prompt) take two 256-row token tiles per weight tile.
// The LDS-DMA kernels address an operand as a 64-bit scalar base + a 32-bit per-lane byte offset that holds the row: an operand of 4 GB or
// more (rows x row bytes) does not fit and takes the LDS-tiled kernel (64-bit pointers) instead
static bool dma_offsets_fit(long long rows, long long ld_elems, int esz) { return (rows + 1) * ld_elems * esz <= 0xffffffffll; }

static bool panel_applies(int m, int n, int k, int lda);
static int ring_split_count(int m, int n, int k) {
  constexpr int min_m = 33, max_m = 512;
  // measured (tools/yardstick_small.py, cold weights): wins 5-15 % over the LDS-tiled kernel on the wide projections (qkv, gate_up) at
  // 33-256 tokens, loses on N = 4096 where 16 slabs of partials outweigh the weight stream
  constexpr int min_n = 8192;
  constexpr int max_s = 256;
  if (m < min_m || m > max_m || k % 128 != 0 || k < 256 || n < min_n || !dma_offsets_fit(n, k, 2)) return 0;
  const int tiles = ((n + 255) / 256) * ((m + 255) / 256), units = k / 128;
  int s = 256 / tiles;
  if (s > units) s = units;
  if (s > max_s) s = max_s;
  return s >= 1 ? s : 0;
}
template <int EPI>
int launch_ring_split(const bf16_t* a, const bf16_t* w, void* c, int m, int n, int k, int lda, int ldc, int splits, float* partial,
                      hipStream_t st, FusedNorm* fn, int pk) {
  static thread_local AtsPerDeviceFlag attr_flag;
  bool& attr_done = attr_flag.cur();
  if (!attr_done) {
    ATS_HIP(hipFuncSetAttribute((const void*)gemm_ring_kernel<EPI_F32, 8, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    ATS_HIP(hipFuncSetAttribute((const void*)gemm_ring_kernel<EPI_F32, 4, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    attr_done = true;
  }
  const int tiles_n = (n + 255) / 256;
  const float* none = nullptr;
  const int tiles_m = (m + 255) / 256;
  if (m > 128)
    hipLaunchKernelGGL((gemm_ring_kernel<EPI_F32, 8, false, true>), dim3(tiles_n * tiles_m * splits), dim3(512), 128 * 1024, st, (const void*)a, (const void*)w,
                       none, none, (void*)partial, m, n, k, lda, n, tiles_n, tiles_m, 1, splits, (float*)nullptr, (const unsigned char*)nullptr, pk);
  else
    hipLaunchKernelGGL((gemm_ring_kernel<EPI_F32, 4, false, true>), dim3(tiles_n * splits), dim3(512), 96 * 1024, st, (const void*)a, (const void*)w,
                       none, none, (void*)partial, m, n, k, lda, n, tiles_n, 1, 1, splits, (float*)nullptr, (const unsigned char*)nullptr, pk);
  ATS_LAUNCH_CHECK();
  ats_count_path(ATS_PATH_RING_SPLIT);
  if (c == nullptr) return ATSPEED_OK;          // partials only: the caller's next kernel sums the slabs itself (ats_gemm_partials)
  return reduce_splits<bf16_t, EPI>(partial, c, m, n, ldc, splits, st, fn, pk);
}

// ---- K-cut form of the ring kernel for the bf16 / fp16 N <= 4096 projections (o_proj, down) at 257-1100 tokens (round 6; 4-16 users in lock
// step: K-token continuation forwards x users, beamSD.py:579-588).  Their 256-wide tile grid is 16 x 2-9 tiles on 256 CUs; the in-launch
// split-K tail spreads the k-steps but pays a seam per tile, the panel form covers 257-384 tokens only.  This is the plain form that won in fp8
// (gemm_ring_mx_kernel<..., SPLITK>, round 5): grid = tiles x parts ~ one round of 256 workgroups, part z takes its share of the 128-k units
// (>= 4 units = 512 k per part), 128-row token tiles while two parts of them fit a round (else 256-row ones), fp32 slabs [z][M][N] in the
// caller's workspace, finished by the reduce kernels of every other split form (splitk_resid_rmsnorm_kernel fuses residual + next RMSNorm).
// Switch "gemm_kcut" (ATSPEED_GEMM_KCUT): 0 off, 1 where the panel form does not apply, 2 before the panel form too (A/B).
static int kcut_split_count(int m, int n, int k, int lda, bool* rows256 = nullptr) {
  if (m < 257 || m > 1100 || n > 4096) return 0;                       // (cheap guards first: every GEMM of every forward passes through here)
  const int on = ats_switch(ATS_SW_GEMM_KCUT);
  if (!on || n < 1024 || k % 128 != 0 || k < 2048 || (lda % 8) != 0 || (n % 4) != 0 || !dma_offsets_fit(n, k, 2) || !dma_offsets_fit(m, lda, 2)) return 0;
  if (on < 2 && panel_applies(m, n, k, lda)) return 0;
  const int tn = (n + 255) / 256, t256 = tn * ((m + 255) / 256), t128 = tn * ((m + 127) / 128);
  const bool r256 = t128 > 128;
  const int tiles = r256 ? t256 : t128;
  if (rows256) *rows256 = r256;
  const int s_ = std::min(256 / tiles, (k / 128) / 4);
  return s_ >= 2 ? s_ : 0;
}
static int launch_ring_kcut(const bf16_t* a, const bf16_t* w, float* partial, int m, int n, int k, int lda, int splits, bool rows256, hipStream_t st, int pk) {
  static thread_local AtsPerDeviceFlag attr_flag;
  bool& attr_done = attr_flag.cur();
  if (!attr_done) {
    ATS_HIP(hipFuncSetAttribute((const void*)gemm_ring_kernel<EPI_F32, 8, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    ATS_HIP(hipFuncSetAttribute((const void*)gemm_ring_kernel<EPI_F32, 4, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    attr_done = true;
  }
  const int tiles_n = (n + 255) / 256, tiles_m = rows256 ? (m + 255) / 256 : (m + 127) / 128;
  const float* none = nullptr;
  if (rows256)
    hipLaunchKernelGGL((gemm_ring_kernel<EPI_F32, 8, false, true>), dim3(tiles_n * tiles_m * splits), dim3(512), 128 * 1024, st, (const void*)a, (const void*)w,
                       none, none, (void*)partial, m, n, k, lda, n, tiles_n, tiles_m, 4, splits, (float*)nullptr, (const unsigned char*)nullptr, pk);
  else
    hipLaunchKernelGGL((gemm_ring_kernel<EPI_F32, 4, false, true>), dim3(tiles_n * tiles_m * splits), dim3(512), 96 * 1024, st, (const void*)a, (const void*)w,
                       none, none, (void*)partial, m, n, k, lda, n, tiles_n, tiles_m, 4, splits, (float*)nullptr, (const unsigned char*)nullptr, pk);
  ATS_LAUNCH_CHECK();
  ats_count_path(ATS_PATH_RING_SPLIT);
  return ATSPEED_OK;
}

template <typename T, int BM, int BN, int WM, int WN, int EPI>
int launch_cfg(const T* a, const T* w, void* c, int m, int n, int k, int lda, int ldc, const Plan& p, float* partial,
               hipStream_t st, FusedNorm* fn, int pk) {
  using Cfg = TileCfg<T, BM, BN, WM, WN>;
  dim3 grid((n + BN - 1) / BN, (m + BM - 1) / BM, p.splits);
  size_t lds = 2 * Cfg::STAGE_BYTES;
  if (p.splits > 1) {
    auto kern = gemm_kernel<T, BM, BN, WM, WN, EPI, true>;
    hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, st, a, w, c, m, n, k, lda, ldc, p.k_per_split, partial, pk);
    ATS_LAUNCH_CHECK();
    ats_count_path(ATS_PATH_TILED);
    if (c == nullptr) return ATSPEED_OK;        // partials only: the caller's next kernel sums the slabs itself (ats_gemm_partials)
    return reduce_splits<T, EPI>(partial, c, m, n, ldc, p.splits, st, fn, pk);
  } else {
    auto kern = gemm_kernel<T, BM, BN, WM, WN, EPI, false>;
    hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, st, a, w, c, m, n, k, lda, ldc, p.k_per_split, partial, pk);
    ATS_LAUNCH_CHECK();
    ats_count_path(ATS_PATH_TILED);
  }
  return ATSPEED_OK;
}


// dispatch of gemm_wdma_kernel: bf16, 33-256 tokens, K % 64 == 0, and an N that gives 150-256 workgroups of 192 or 128 weight rows (one per
// CU, the deepest ring that fits): gate_up (22016 = 172 x 128) and the lm_head (32859 = 172 x 192).  Measured in the engine's form (packed
// operands, tools/yardstick_engine_like.py, us per launch at 60 / 100 / 121 tokens): gate_up 48.3 / 53.2 / 56.2 -> 44.8 / 47.8 / 50.0, lm_head
// 75.9 / 85.5 / 89.6 -> 64.1 / 70.4 / 73.1 (4.2 TB/s at 60 tokens).  NOT the qkv projection (12288 rows: 96 tiles of 128 leave 160 CUs idle,
// 35-42 us; 192 tiles of 64 give 27.7 / 32.6 against 30.9 / 35.4, but the split-K form hands its slabs to the RoPE kernel, which the
// bf16 store + separate RoPE pass of this form gives back: 22.71 vs 22.65 ms per user) and not N = 4096 (split-K + fused reduce / norm).
static bool wdma_applies(int m, int n, int k, int lda, int epilogue) {
  static const int on = env_int("ATSPEED_GEMM_WDMA", 1);
  constexpr int min_m = 33;
  constexpr int max_m = 256;
  if (!on || m < min_m || m > max_m || k % 64 != 0 || k < 512 || (lda % 8) != 0 || !dma_offsets_fit(n, k, 2) || !dma_offsets_fit(m, lda, 2)) return false;
  const int t192 = (n + 191) / 192, t128 = (n + 127) / 128;
  const bool ok128 = t128 >= 150 && t128 <= 256, ok192 = t192 >= 150 && t192 <= 256;
  if (!(ok128 || (ok192 && m <= 128))) return false;                  // 129-256 rows: 128-row tiles only
  return epilogue == EPI_STORE || epilogue == EPI_F32 || (epilogue == EPI_SWIGLU && n % 32 == 0);
}
template <int BM, int BN, int NST, int EPI, int WM = 2>
int launch_wdma_cfg(const bf16_t* a, const bf16_t* w, void* c, int m, int n, int k, int lda, int ldc, hipStream_t st, int pk) {
  auto kern = gemm_wdma_kernel<BM, BN, NST, EPI, false, WM>;
  constexpr int lds = NST * (BM + BN) * 128;
  static thread_local AtsPerDeviceFlag attr_flag;
  bool& attr_done = attr_flag.cur();
  if (!attr_done) {
    ATS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((n + BN - 1) / BN), dim3(128 * WM), lds, st, (const void*)a, (const void*)w, c, m, n, k, lda, ldc, pk, 1, (const float*)nullptr, (const float*)nullptr, RopeEpi{});
  ATS_LAUNCH_CHECK();
  ats_count_path(ATS_PATH_WDMA);
  return ATSPEED_OK;
}
// split-K form: projections whose N gives too few 128-row tiles (qkv: 96, o_proj and down: 32) take tiles x splits = 150-256 workgroups
static int wdma_split_count(int m, int n, int k, int lda) {
  static const int on_all = env_int("ATSPEED_GEMM_WDMA", 1);
  constexpr int min_m = 33;
  if (!on_all || m < min_m || m > 256 || k % 64 != 0 || (lda % 8) != 0 || n < 2048 || !dma_offsets_fit(n, k, 2) || !dma_offsets_fit(m, lda, 2)) return 0;
  const int t128 = (n + 127) / 128, n_kt = k / 64;
  if (t128 >= 150) return 0;                                           // wide enough for the no-split form (or too wide for one round)
  // k-tiles per part at least: 8 up to 128 tokens (o_proj, K = 4096: 8 parts of 8 tiles, 15.8 / 18.8 -> 13.5 / 16.0 us at 60 / 100 tokens), 16 above
  // (at 225 tokens the 8 x 8 form lost: 24.4 vs 23.7 us)
  const int min_tiles = m <= 128 ? 8 : 16;
  const int s = std::min(256 / t128, n_kt / min_tiles);
  return (s >= 2 && t128 * s >= 150) ? s : 0;
}
template <int BM, int NST, int WM = 2>
int launch_wdma_split_cfg(const bf16_t* a, const bf16_t* w, float* partial, int m, int n, int k, int lda, int splits, hipStream_t st, int pk) {
  auto kern = gemm_wdma_kernel<BM, 128, NST, EPI_F32, true, WM>;
  constexpr int lds = NST * (BM + 128) * 128;
  static thread_local AtsPerDeviceFlag attr_flag;
  bool& attr_done = attr_flag.cur();
  if (!attr_done) {
    ATS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((n + 127) / 128, splits), dim3(128 * WM), lds, st, (const void*)a, (const void*)w, (void*)partial, m, n, k, lda, n, pk, splits, (const float*)nullptr, (const float*)nullptr, RopeEpi{});
  ATS_LAUNCH_CHECK();
  ats_count_path(ATS_PATH_WDMA_SPLIT);
  return ATSPEED_OK;
}
static int launch_wdma_split(const bf16_t* a, const bf16_t* w, float* partial, int m, int n, int k, int lda, int splits, hipStream_t st, int pk) {
  if (m <= 64)  return launch_wdma_split_cfg<64, 6>(a, w, partial, m, n, k, lda, splits, st, pk);
  if (m <= 128) return launch_wdma_split_cfg<128, 4>(a, w, partial, m, n, k, lda, splits, st, pk);
  return launch_wdma_split_cfg<256, 3, 4>(a, w, partial, m, n, k, lda, splits, st, pk);      // 2 x 4 waves (the 2 x 2 form: +12-15 % at 150-256 tokens, round 3)
}
template <int EPI>
int launch_wdma(const bf16_t* a, const bf16_t* w, void* c, int m, int n, int k, int lda, int ldc, hipStream_t st, int pk) {
  const int t192 = (n + 191) / 192, t128 = (n + 127) / 128;
  const bool ok128 = t128 >= 150 && t128 <= 256;
  // tile width: 128 weight rows when that gives 150-256 workgroups, else 192 (the lm_head: 257 tiles of 128 would spill one into a second round)
  if (m <= 64) {
    if (ok128) return launch_wdma_cfg<64, 128, 6, EPI>(a, w, c, m, n, k, lda, ldc, st, pk);           // 24 KB x 6
    return launch_wdma_cfg<64, 192, 4, EPI>(a, w, c, m, n, k, lda, ldc, st, pk);                      // 32 KB x 4
  }
  if (m <= 128) {
    if (ok128) return launch_wdma_cfg<128, 128, 4, EPI>(a, w, c, m, n, k, lda, ldc, st, pk);          // 32 KB x 4 (x 5 = all 160 KB: no faster)
    return launch_wdma_cfg<128, 192, 3, EPI>(a, w, c, m, n, k, lda, ldc, st, pk);                     // 40 KB x 3
  }
  (void)t192;
  return launch_wdma_cfg<256, 128, 3, EPI, 4>(a, w, c, m, n, k, lda, ldc, st, pk);                    // 129-256 tokens: 48 KB x 3 (the X rows are two thirds of a stage), 8 waves
}

// ---- panel form of the ring kernel (gemm_ring_kernel<..., WN = 2, WM = 4>): 257-384 tokens in ONE launch (16 users' K-token continuation
// forwards, beamSD.py:579-588; a long prompt's first verification).  A workgroup owns a 128-row weight panel and all token rows (384): wide
// projections (150-256 panels: gate_up's 172) run one round without split; narrow ones are cut in K so that panels x parts fill the chip
// (qkv 96 x 2, down 32 x 8) and leave fp32 slabs to the consumers of every other split form.  Before: two 256-row token tiles of which 38 %
// were padding at 320 tokens.  Measured (tools/panel_sweep.py, profiles/r05_panel_sweep.txt, us per launch at 320 tokens): qkv 69.2 -> 56.5,
// gate_up 77.6 -> 70.4, down 58.9 -> 50.1; NOT o_proj (8 parts of 16 k-steps are all prologue: 27.8 -> 30.3), not above 384 tokens (the
// 512-row panel, 160 KB of LDS, lost on every projection: gate_up 82 -> 92 at 512) and gate_up only while the padding stays below ~10 %.
// 172 workgroups at ~4.8 TF each is what this form gives: the band stays MFMA-inefficient (0.33 of nominal at 320 tokens), see DESIGN section 9.
static int panel_split_count(int n, int k);
static bool panel_applies(int m, int n, int k, int lda) {
  if (m < 257 || m > 384) return false;                                // (the cheap guard first: every one-user GEMM passes through here)
  const int on = ats_switch(ATS_SW_GEMM_PANEL);                        // tools/panel_sweep.py flips it in-process (atspeed_set_switch)
  if (!on || k % 128 != 0 || k < 2048 || (lda % 8) != 0 || !dma_offsets_fit(n, k, 2) || !dma_offsets_fit(m, lda, 2)) return false;
  const int t128 = (n + 127) / 128;
  if (t128 > 256 || t128 < 16) return false;                           // (K >= 2048: the target's projections; a 68M draft's thin GEMMs stay where they were)
  if (on >= 2) return true;                                            // 2: every shape the kernel can take (tests, sweeps)
  if (panel_split_count(n, k) == 1) return m <= 352;                   // one round of panels: until the 384-row tile's padding costs more than the second token tile did
  return t128 >= 64 || k >= 8192;                                      // split: qkv (96 panels x 2) and down (K = 11008); o_proj's parts are too short
}
static int panel_split_count(int n, int k) {                           // 1: no split
  const int t128 = (n + 127) / 128, units = k / 128;
  if (t128 >= 150) return 1;
  return std::max(1, std::min(256 / t128, units / 4));                 // at least 4 units (512 k) per part
}
template <int EPI, int MT2, bool SPLIT>
int launch_panel_cfg(const bf16_t* x, const bf16_t* w, void* c, int m, int n, int k, int ldx, int ldc, int splits, hipStream_t st, int pk) {
  auto kern = gemm_ring_kernel<EPI, MT2, false, SPLIT, 4, false, 2, 4>;
  constexpr int lds = 4 * (128 + 4 * MT2 * 16) * 64;                   // 128 KB at 384 token rows
  static thread_local AtsPerDeviceFlag attr_flag;
  bool& attr_done = attr_flag.cur();
  if (!attr_done) {
    ATS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  const int tiles_n = (n + 127) / 128;
  const float* none = nullptr;
  hipLaunchKernelGGL(kern, dim3(tiles_n * (SPLIT ? splits : 1)), dim3(512), lds, st, (const void*)x, (const void*)w, none, none, c, m, n, k, ldx, ldc,
                     tiles_n, 1, 1, SPLIT ? splits : 1, (float*)nullptr, (const unsigned char*)nullptr, pk, RopeEpi{}, SkTail{});
  ATS_LAUNCH_CHECK();
  ats_count_path(SPLIT ? ATS_PATH_PANEL_SPLIT : ATS_PATH_PANEL);
  return ATSPEED_OK;
}
template <int EPI>
int launch_panel(const bf16_t* x, const bf16_t* w, void* c, int m, int n, int k, int ldx, int ldc, hipStream_t st, int pk) {
  return launch_panel_cfg<EPI, 6, false>(x, w, c, m, n, k, ldx, ldc, 1, st, pk);
}
static int launch_panel_split(const bf16_t* x, const bf16_t* w, float* partial, int m, int n, int k, int ldx, int splits, hipStream_t st, int pk) {
  return launch_panel_cfg<EPI_F32, 6, true>(x, w, (void*)partial, m, n, k, ldx, n, splits, st, pk);
}

template <typename T, int EPI>
int launch_epi(const T* a, const T* w, void* c, int m, int n, int k, int lda, int ldc, float* partial, size_t ws_bytes,
               hipStream_t st, FusedNorm* fn, int pk) {
  if constexpr (sizeof(T) == 2 && (EPI == EPI_STORE || EPI == EPI_F32 || EPI == EPI_RESID)) {
    bool r256 = false;
    const int ks = kcut_split_count(m, n, k, lda, &r256);
    if (ks >= 2 && (size_t)ks * m * n * sizeof(float) <= ws_bytes && ((uintptr_t)partial & 15) == 0) {
      ATS_TRY(launch_ring_kcut(a, w, partial, m, n, k, lda, ks, r256, st, pk));
      return reduce_splits<bf16_t, EPI>(partial, c, m, n, ldc, ks, st, fn, pk);
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (panel_applies(m, n, k, lda) && (EPI != EPI_SWIGLU || (n % 32 == 0 && (ldc & 3) == 0))) {
      const int ps = panel_split_count(n, k);
      if (ps == 1) return launch_panel<EPI>(a, w, c, m, n, k, lda, ldc, st, pk);
      if ((size_t)ps * m * n * sizeof(float) <= ws_bytes && ((uintptr_t)partial & 15) == 0) {
        ATS_TRY(launch_panel_split(a, w, partial, m, n, k, lda, ps, st, pk));
        return reduce_splits<bf16_t, EPI>(partial, c, m, n, ldc, ps, st, fn, pk);
      }
    }
  }
  if constexpr (sizeof(T) == 2 && (EPI == EPI_STORE || EPI == EPI_F32 || EPI == EPI_SWIGLU)) {
    if (wdma_applies(m, n, k, lda, EPI)) return launch_wdma<EPI>(a, w, c, m, n, k, lda, ldc, st, pk);
  }
  if constexpr (sizeof(T) == 2) {
    const int s = wdma_split_count(m, n, k, lda);
    if (s >= 2 && (size_t)s * m * n * sizeof(float) <= ws_bytes && ((uintptr_t)partial & 15) == 0 && (EPI != EPI_SWIGLU || n % 32 == 0)) {
      ATS_TRY(launch_wdma_split(a, w, partial, m, n, k, lda, s, st, pk));
      return reduce_splits<bf16_t, EPI>(partial, c, m, n, ldc, s, st, fn, pk);
    }
  }
  if constexpr (sizeof(T) == 2) {
    const int rs = ring_split_count(m, n, k);
    if (rs >= 1 && (lda % 8) == 0 && (size_t)rs * m * n * sizeof(float) <= ws_bytes && (EPI != EPI_SWIGLU || n % 32 == 0))
      return launch_ring_split<EPI>(a, w, c, m, n, k, lda, ldc, rs, partial, st, fn, pk);
  }
  Plan p = make_plan<T>(m, n, k);
  if (p.bn == 64 && p.bm != 128) p.bn = 128;
  if (p.splits > 1 && (size_t)p.splits * m * n * sizeof(float) > ws_bytes) {   // not enough workspace: no split
    p.splits = 1;
    p.k_per_split = ((k + GemmTraits<T>::BK - 1) / GemmTraits<T>::BK) * GemmTraits<T>::BK;
  }
  if (p.bn == 64 && p.bm == 128) return launch_cfg<T, 128, 64, 2, 2, EPI>(a, w, c, m, n, k, lda, ldc, p, partial, st, fn, pk);
  switch (p.bm) {
    case 16:  return launch_cfg<T, 16, 128, 1, 4, EPI>(a, w, c, m, n, k, lda, ldc, p, partial, st, fn, pk);
    case 32:  return launch_cfg<T, 32, 128, 1, 4, EPI>(a, w, c, m, n, k, lda, ldc, p, partial, st, fn, pk);
    case 64:  return launch_cfg<T, 64, 128, 1, 4, EPI>(a, w, c, m, n, k, lda, ldc, p, partial, st, fn, pk);
    default:  return launch_cfg<T, 128, 128, 2, 2, EPI>(a, w, c, m, n, k, lda, ldc, p, partial, st, fn, pk);
  }
}

template <typename T>
int launch_typed(const void* a, const void* w, void* c, int m, int n, int k, int lda, int ldc, int epi, void* ws,
                 size_t ws_bytes, hipStream_t st, int pk) {
  const T* A = (const T*)a; const T* Wt = (const T*)w; float* P = (float*)ws;
  switch (epi) {
    case EPI_STORE:  return launch_epi<T, EPI_STORE>(A, Wt, c, m, n, k, lda, ldc, P, ws_bytes, st, nullptr, pk);
    case EPI_F32:    return launch_epi<T, EPI_F32>(A, Wt, c, m, n, k, lda, ldc, P, ws_bytes, st, nullptr, pk);
    case EPI_RESID:  return launch_epi<T, EPI_RESID>(A, Wt, c, m, n, k, lda, ldc, P, ws_bytes, st, nullptr, pk);
    case EPI_SWIGLU: return launch_epi<T, EPI_SWIGLU>(A, Wt, c, m, n, k, lda, ldc, P, ws_bytes, st, nullptr, pk);
  }
  atspeed_set_error("gemm: unknown epilogue %d", epi);
  return ATSPEED_ERR_INVALID;
}

}  // namespace

size_t ats_gemm_workspace_bytes(int m, int n, int k, int dtype) {
  Plan p = dtype == ATS_HALF ? make_plan<bf16_t>(m, n, k) : make_plan<float>(m, n, k);
  size_t b = p.splits > 1 ? (size_t)p.splits * m * n * sizeof(float) : 0;
  if (dtype == ATS_HALF) b = std::max(b, (size_t)ring_split_count(m, n, k) * m * n * sizeof(float));
  if (dtype == ATS_HALF) b = std::max(b, (size_t)wdma_split_count(m, n, k, k) * m * n * sizeof(float));
  if (dtype == ATS_HALF && panel_applies(m, n, k, k) && panel_split_count(n, k) > 1) b = std::max(b, (size_t)panel_split_count(n, k) * m * n * sizeof(float));
  if (dtype == ATS_HALF) b = std::max(b, (size_t)kcut_split_count(m, n, k, k) * m * n * sizeof(float));
  return b;
}

static bool big_kernel_applies(int m, int n, int k, int lda, int ldc, int dtype, int epilogue, bool lm_head = false);
// does this launch take the K-cut form (launch_epi)?  Then the ring kernel's dispatch (big_kernel_applies) does not get it.
static bool kcut_takes(int m, int n, int k, int lda, int dtype, int epilogue, const void* ws, size_t ws_bytes) {
  if (dtype != ATS_HALF || !(epilogue == EPI_STORE || epilogue == EPI_F32 || epilogue == EPI_RESID)) return false;
  const int ks = kcut_split_count(m, n, k, lda);
  return ks >= 2 && ws && ((uintptr_t)ws & 15) == 0 && (size_t)ks * m * n * sizeof(float) <= ws_bytes;
}

// One user's wide bf16 projection as fp32 split-K slabs [splits][m][n] in the workspace, WITHOUT the reduce pass: the consumer sums the
// slabs while it reads them (RoPE + KV scatter after the qkv projection: one launch less per layer).  *splits_out = 0 when this shape does
// not take the split-K ring path (the caller then runs the ordinary ats_gemm).
int ats_gemm_partials(const void* a, const void* w, int m, int n, int k, int lda, int dtype, void* workspace, size_t workspace_bytes,
                      hipStream_t st, int* splits_out, int pk) {
  *splits_out = 0;
  if (dtype != ATS_HALF || m <= 0 || big_kernel_applies(m, n, k, lda, n, dtype, EPI_STORE)) return ATSPEED_OK;
  if (panel_applies(m, n, k, lda)) {
    const int ps = panel_split_count(n, k);
    if (ps >= 2 && (n % 4) == 0 && ((uintptr_t)workspace & 15) == 0 && (size_t)ps * m * n * sizeof(float) <= workspace_bytes) {
      ATS_REQUIRE(a && w && workspace, ATSPEED_ERR_INVALID, "gemm: null operand");
      ATS_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)w & 15) == 0, ATSPEED_ERR_INVALID, "gemm: operands must be 16-byte aligned");
      ATS_TRY(launch_panel_split((const bf16_t*)a, (const bf16_t*)w, (float*)workspace, m, n, k, lda, ps, st, pk));
      *splits_out = ps;
    }
    return ATSPEED_OK;
  }
  if (wdma_applies(m, n, k, lda, EPI_STORE)) return ATSPEED_OK;        // the no-split kernel writes bf16 qkv itself: the caller runs ats_gemm + the plain RoPE pass
  {
    const int s = wdma_split_count(m, n, k, lda);
    if (s >= 2 && (n % 4) == 0 && ((uintptr_t)workspace & 15) == 0 && (size_t)s * m * n * sizeof(float) <= workspace_bytes) {
      ATS_REQUIRE(a && w && workspace, ATSPEED_ERR_INVALID, "gemm: null operand");
      ATS_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)w & 15) == 0, ATSPEED_ERR_INVALID, "gemm: operands must be 16-byte aligned");
      ATS_TRY(launch_wdma_split((const bf16_t*)a, (const bf16_t*)w, (float*)workspace, m, n, k, lda, s, st, pk));
      *splits_out = s;
      return ATSPEED_OK;
    }
  }
  const int rs = ring_split_count(m, n, k);
  if (rs >= 1 && (lda % 8) == 0 && (n % 4) == 0 && ((uintptr_t)workspace & 15) == 0 && (size_t)rs * m * n * sizeof(float) <= workspace_bytes) {
    ATS_REQUIRE(a && w && workspace, ATSPEED_ERR_INVALID, "gemm: null operand");
    ATS_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)w & 15) == 0, ATSPEED_ERR_INVALID, "gemm: operands must be 16-byte aligned");
    ATS_TRY(launch_ring_split<EPI_STORE>((const bf16_t*)a, (const bf16_t*)w, nullptr, m, n, k, lda, n, rs, (float*)workspace, st, nullptr, pk));
    *splits_out = rs;
    return ATSPEED_OK;
  }
  // Round 6: the LDS-tiled split-K kernel's slabs too (up to 32 tokens: the K-beam final step of one user, beamSD.py:505-509; the draft's forwards).
  // Its reduce pass was one more launch per layer; the RoPE kernel sums the same slabs in the same order and rounds as the reduce pass stored.
  if (k % 8 == 0 && lda % 8 == 0 && (n % 4) == 0 && workspace && ((uintptr_t)workspace & 15) == 0 && a && w && (((uintptr_t)a | (uintptr_t)w) & 15) == 0) {
    Plan p = make_plan<bf16_t>(m, n, k);
    if (p.bn == 64 && p.bm != 128) p.bn = 128;
    if (p.splits > 1 && p.bn == 128 && (size_t)p.splits * m * n * sizeof(float) <= workspace_bytes) {
      const bf16_t* A = (const bf16_t*)a; const bf16_t* Wt = (const bf16_t*)w; float* P = (float*)workspace;
      int rc;
      switch (p.bm) {
        case 16:  rc = launch_cfg<bf16_t, 16, 128, 1, 4, EPI_STORE>(A, Wt, nullptr, m, n, k, lda, n, p, P, st, nullptr, pk); break;
        case 32:  rc = launch_cfg<bf16_t, 32, 128, 1, 4, EPI_STORE>(A, Wt, nullptr, m, n, k, lda, n, p, P, st, nullptr, pk); break;
        case 64:  rc = launch_cfg<bf16_t, 64, 128, 1, 4, EPI_STORE>(A, Wt, nullptr, m, n, k, lda, n, p, P, st, nullptr, pk); break;
        default:  rc = launch_cfg<bf16_t, 128, 128, 2, 2, EPI_STORE>(A, Wt, nullptr, m, n, k, lda, n, p, P, st, nullptr, pk); break;
      }
      ATS_TRY(rc);
      *splits_out = p.splits;
    }
  }
  return ATSPEED_OK;
}

int ats_gemm(const void* a, const void* w, void* c, int m, int n, int k, int lda, int ldc, int dtype, int epilogue,
             void* workspace, size_t workspace_bytes, hipStream_t st, int pk, const SkArena* sk) {
  if (m <= 0 || n <= 0) return ATSPEED_OK;
  int epc = dtype == ATSPEED_F32 ? 4 : 8;
  // packed operands (common.h): bf16 only, K and the row strides multiples of one 64-byte k-block, SwiGLU output rows likewise
  ATS_REQUIRE(!pk || (dtype == ATS_HALF && k % 32 == 0 && lda == k && (epilogue != EPI_SWIGLU || ldc % 32 == 0)), ATSPEED_ERR_INVALID,
              "gemm: packed operands need bf16, K %% 32 == 0, lda == K (K=%d lda=%d ldc=%d)", k, lda, ldc);
  ATS_REQUIRE(a && w && c && k > 0, ATSPEED_ERR_INVALID, "gemm: null operand");
  ATS_REQUIRE(k % epc == 0 && lda % epc == 0, ATSPEED_ERR_INVALID, "gemm: K=%d / lda=%d must be multiples of %d", k, lda, epc);
  ATS_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)w & 15) == 0, ATSPEED_ERR_INVALID, "gemm: operands must be 16-byte aligned");
  ATS_REQUIRE(epilogue != EPI_SWIGLU || n % 32 == 0, ATSPEED_ERR_INVALID, "gemm: SwiGLU needs N %% 32 == 0 (N=%d)", n);
  if (dtype == ATSPEED_F32) return launch_typed<float>(a, w, c, m, n, k, lda, ldc, epilogue, workspace, workspace_bytes, st, 0);
  if (dtype == ATS_HALF) {
    if (!kcut_takes(m, n, k, lda, dtype, epilogue, workspace, workspace_bytes) && big_kernel_applies(m, n, k, lda, ldc, dtype, epilogue)) {
      const bf16_t* X = (const bf16_t*)a; const bf16_t* Wt = (const bf16_t*)w;
      switch (epilogue) {
        case EPI_STORE:  return launch_big<EPI_STORE>(X, Wt, c, m, n, k, lda, ldc, st, pk, RopeEpi{}, nullptr, nullptr, sk);
        case EPI_F32:    return launch_big<EPI_F32>(X, Wt, c, m, n, k, lda, ldc, st, pk, RopeEpi{}, nullptr, nullptr, sk);
        case EPI_RESID:  return launch_big<EPI_RESID>(X, Wt, c, m, n, k, lda, ldc, st, pk, RopeEpi{}, nullptr, nullptr, sk);
        case EPI_SWIGLU: return launch_big<EPI_SWIGLU>(X, Wt, c, m, n, k, lda, ldc, st, pk, RopeEpi{}, nullptr, nullptr, sk);
      }
    }
    return launch_typed<bf16_t>(a, w, c, m, n, k, lda, ldc, epilogue, workspace, workspace_bytes, st, pk);
  }
  atspeed_set_error("gemm: unknown dtype %d", dtype);
  return ATSPEED_ERR_INVALID;
}

bool ats_gemm_qkv_rope_applies(int m, int hidden, int head_dim, int dtype) {
  return ats_switch(ATS_SW_FUSE_QKV_ROPE) != 0 && dtype == ATS_HALF && head_dim == 128 && hidden % 256 == 0 && big_kernel_applies(m, 3 * hidden, hidden, hidden, 3 * hidden, dtype, EPI_STORE);
}

int ats_gemm_qkv_rope(const void* x, const void* wqkv, void* qkv, int m, int hidden, const RopeEpi& rope, hipStream_t st, int pk, const SkArena* sk) {
  ATS_REQUIRE(x && wqkv && qkv && rope.rows && rope.cos_tab && rope.sin_tab && rope.hidden == hidden, ATSPEED_ERR_INVALID, "gemm_qkv_rope: null / inconsistent argument");
  ATS_REQUIRE(hidden % 256 == 0 && big_kernel_applies(m, 3 * hidden, hidden, hidden, 3 * hidden, ATS_HALF, EPI_STORE), ATSPEED_ERR_INVALID,
              "gemm_qkv_rope: shape %d x %d is not the batched kernel's", m, hidden);
  ATS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)wqkv & 15) == 0 && ((uintptr_t)qkv & 15) == 0, ATSPEED_ERR_INVALID, "gemm_qkv_rope: operands must be 16-byte aligned");
  return launch_big<EPI_QKV_ROPE>((const bf16_t*)x, (const bf16_t*)wqkv, qkv, m, 3 * hidden, hidden, hidden, 3 * hidden, st, pk, rope, nullptr, nullptr, sk);
}

size_t ats_lmhead_lse_part_bytes(int m, int n) { return (size_t)m * ((n + 255) / 256) * 2 * sizeof(float); }

// logits = a * w^T (fp32) and lse[row] = log sum exp over ALL n columns of the row.  On the batched path (bf16, ring kernel) the
// normaliser comes out of the GEMM epilogue and only the 256-column tiles flagged in tile_store (device bytes, one per tile; NULL = all)
// are written; otherwise the plain GEMM + the streaming lse_rows_kernel.  *fused_out tells which.
int ats_lmhead_lse(const void* a, const void* w, float* logits, int m, int n, int k, int lda, int ldc, int dtype, const unsigned char* tile_store,
                   float* part, size_t part_bytes, float* lse, void* workspace, size_t workspace_bytes, hipStream_t st, int* fused_out, int pk,
                   const SkArena* sk) {
  static const int fuse = env_int("ATSPEED_FUSE_LSE", 1);
  if (fused_out) *fused_out = 0;
  if (m <= 0) return ATSPEED_OK;
  if (fuse && dtype == ATS_HALF && part && part_bytes >= ats_lmhead_lse_part_bytes(m, n) && ((uintptr_t)part & 7) == 0 &&
      big_kernel_applies(m, n, k, lda, ldc, dtype, EPI_F32, true)) {
    ATS_REQUIRE(a && w && logits && lse, ATSPEED_ERR_INVALID, "lmhead_lse: null operand");
    ATS_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)w & 15) == 0, ATSPEED_ERR_INVALID, "lmhead_lse: operands must be 16-byte aligned");
    if (fused_out) *fused_out = 1;
    return launch_big_lse((const bf16_t*)a, (const bf16_t*)w, logits, m, n, k, lda, ldc, part, tile_store, lse, st, pk, sk);
  }
  ATS_TRY(ats_gemm(a, w, logits, m, n, k, lda, ldc, dtype, EPI_F32, workspace, workspace_bytes, st, pk, sk));
  return ats_lse_rows(logits, m, n, ldc, lse, st);
}

// h += a * w^T, then xn = rmsnorm(h) * norm_w  (split-K path fuses the reduce, the residual and the norm)
static bool big_kernel_applies(int m, int n, int k, int lda, int ldc, int dtype, int epilogue, bool lm_head) {
  // the 256-wide ring kernel vs the 128-wide LDS-tiled kernel (with split-K): the ring kernel wins once its tile grid keeps
  // a fair share of the 256 CUs busy (measured, tools/gemm_ab.py with ATSPEED_GEMM_BIG_MIN_FILL)
  // from 257 tokens (two token tiles): measured against the split-K mode at 300-500 tokens, gate_up 115-138 -> 96-108 us, qkv 80 -> 75 us
  constexpr int big_min_m = 257;
  constexpr int min_fill = 60;   // crossover measured at ~50-60 % (o_proj, down, qkv, gate_up at 512-1920 tokens)
  if (dtype != ATS_HALF || m < big_min_m || k % 128 != 0 || (lda % 8) != 0 || !dma_offsets_fit(n, k, 2) || !dma_offsets_fit(m, lda, 2)) return false;
  if (epilogue == EPI_SWIGLU && ((ldc & 3) != 0 || n % 32 != 0)) return false;
  // 257-384 tokens on a projection of 16-256 panels: the panel form (launch_epi).  Not the lm_head with the fused normaliser (a vocabulary of
  // 150-256 panels, e.g. the stock 32000, at 257-352 logit rows would leave the fused-LSE / tile_store ring path for panel<EPI_F32> + a separate LSE pass,
  // a regime the panel sweep never measured; Beauty's 32859 misses the band by one panel)
  if (!lm_head && panel_applies(m, n, k, lda)) return false;
  const int tn = (n + 255) / 256;
  if (big_fill_pct(tn * ((m + 255) / 256)) >= min_fill || big_fill_pct(tn * ((m + 127) / 128)) >= min_fill) return true;
  // a thin grid whose k-steps the split-K tail spreads over the chip: from 48 tiles of 128 rows (four parts per tile on 192 CUs; N = 4096
  // from 257 tokens), where it overtakes the LDS-tiled split-K kernel + reduce pass (tools/sk_sweep.py: o_proj at 320 tokens 36.7 -> 34.1 us,
  // down 73.6 -> 70.3; at 400 tokens 33.7 -> 29.6 and 68.4 -> 59.7)
  constexpr int sk_min_tiles = 48;
  const int t128 = tn * ((m + 127) / 128);
  return t128 >= sk_min_tiles && sk_any_plan(t128, k);
}

int ats_gemm_resid_norm(const void* a, const void* w, void* h, int m, int n, int k, int lda, int ldh, int dtype,
                        const void* norm_w, void* xn, float eps, void* workspace, size_t workspace_bytes, hipStream_t st, int pk, const SkArena* sk) {
  if (m <= 0) return ATSPEED_OK;
  FusedNorm fn{norm_w, xn, eps, false};
  int epc = dtype == ATSPEED_F32 ? 4 : 8;
  if ((kcut_takes(m, n, k, lda, dtype, EPI_RESID, workspace, workspace_bytes) || !big_kernel_applies(m, n, k, lda, ldh, dtype, EPI_RESID)) && k % epc == 0 && lda % epc == 0) {
    int rc;
    if (dtype == ATSPEED_F32)
      rc = launch_epi<float, EPI_RESID>((const float*)a, (const float*)w, h, m, n, k, lda, ldh, (float*)workspace, workspace_bytes, st, &fn, 0);
    else
      rc = launch_epi<bf16_t, EPI_RESID>((const bf16_t*)a, (const bf16_t*)w, h, m, n, k, lda, ldh, (float*)workspace, workspace_bytes, st, &fn, pk);
    if (rc != ATSPEED_OK) return rc;
  } else {
    ATS_TRY(ats_gemm(a, w, h, m, n, k, lda, ldh, dtype, EPI_RESID, workspace, workspace_bytes, st, pk, sk));
  }
  if (!fn.done) return ats_rmsnorm(h, norm_w, xn, m, n, eps, dtype, st, pk);
  return ATSPEED_OK;
}

// ---- W8A8 projections whose 256-wide tile grid is thin (N = 4096 at 257-2000 tokens: o_proj, down): the block-scaled ring kernel cut in K
// (gemm_ring_mx_kernel<EPI_F32, 4, 8, SPLITK>): 128-row token tiles x parts ~ one round of 256 workgroups, scaled fp32 slabs, the usual reduce.
// Before: 32-80 workgroups on 256 CUs (down at 300 tokens 71 us against 50 us in bf16), and from 512 tokens these projections ran bf16.
static int mx_split_count(int m, int n, int k, bool* rows256 = nullptr) {   // 0: this shape takes the plain ring kernel (or is not an fp8 shape at all)
  if (m < 257 || k % 256 != 0 || !dma_offsets_fit(n, k, 1) || !dma_offsets_fit(m, k, 1)) return 0;
  const int tn = (n + 255) / 256, t256 = tn * ((m + 255) / 256), t128 = tn * ((m + 127) / 128);
  if (big_fill_pct(t256) >= 60 || big_fill_pct(t128) >= 60) return 0;
  // measured (tools/gemm_fp8_ab.py with / without a workspace, us per launch incl. the reduce): down 70.9 -> 37.4 at 300 tokens, 72.1 -> 42.7 at 456,
  // 76.5 -> 54.4 at 640, 77.7 -> 66.2 at 912; o_proj (K = 4096: short parts) 30.7 -> 27.6 at 300, 36.2 -> 32.4 at 456 but 35.5 -> 39.1 at 640; qkv (96
  // tiles of 256 rows already) 32.3 -> 50.1 at 300: only really thin grids, and K = 4096 only up to 512 tokens
  if (t256 > 64 || (k < 8192 && m > 512)) return 0;
  const bool r256 = t128 > 128;                                        // 128-row token tiles while two parts of them fit a round, else 256-row ones
  const int tiles = r256 ? t256 : t128;
  if (rows256) *rows256 = r256;
  const int s_ = std::min(256 / tiles, (k / 256) / 2);                // at least 2 units (512 k) per part
  return s_ >= 2 ? s_ : 0;
}
static int launch_mx_split(const unsigned char* x, const float* sx, const unsigned char* w, const float* sw, float* partial, int m, int n, int k,
                           int splits, hipStream_t st, int pk) {
  static thread_local AtsPerDeviceFlag attr_flag;
  bool& attr_done = attr_flag.cur();
  if (!attr_done) {
    ATS_HIP(hipFuncSetAttribute((const void*)gemm_ring_mx_kernel<EPI_F32, 4, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    ATS_HIP(hipFuncSetAttribute((const void*)gemm_ring_mx_kernel<EPI_F32, 8, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    attr_done = true;
  }
  bool r256 = false;
  (void)mx_split_count(m, n, k, &r256);
  const int tiles_n = (n + 255) / 256, tiles_m = r256 ? (m + 255) / 256 : (m + 127) / 128;
  if (r256) hipLaunchKernelGGL((gemm_ring_mx_kernel<EPI_F32, 8, 8, true>), dim3(tiles_n * tiles_m * splits), dim3(512), 128 * 1024, st, (const void*)x, (const void*)w, sx, sw,
                               (void*)partial, m, n, k, n, tiles_n, tiles_m, 4, pk, RopeEpi{}, splits);
  else      hipLaunchKernelGGL((gemm_ring_mx_kernel<EPI_F32, 4, 8, true>), dim3(tiles_n * tiles_m * splits), dim3(512), 96 * 1024, st, (const void*)x, (const void*)w, sx, sw,
                               (void*)partial, m, n, k, n, tiles_n, tiles_m, 4, pk, RopeEpi{}, splits);
  ATS_LAUNCH_CHECK();
  ats_count_path(ATS_PATH_FP8_RING_SPLIT);
  return ATSPEED_OK;
}

// ---- one user's W8A8 projections (1-256 tokens): gemm_wdma_kernel<..., F8 = true>.  Every launch is a pass over the e4m3 weights (half the
// bytes of the 16-bit form).  Wide projections (150-256 tiles of 128 weight rows: gate_up) run without split and apply their epilogue
// directly; the others are cut in K so that tiles x parts fill the chip (qkv 96 x 2, o_proj / down 32 x 8) and leave scaled fp32 slabs to
// the 16-bit form's consumers (reduce + residual + RMSNorm [+ e4m3 quantisation for the next projection], RoPE + KV scatter).
static bool wdma8_applies(int m, int n, int k) {
  static const int on = env_int("ATSPEED_FP8_SMALL", 1);               // 0: one user's forwards stay on the 16-bit kernels (A/B)
  return on && m >= 1 && m <= 256 && k % 128 == 0 && k >= 512 && n >= 16 && dma_offsets_fit(n, k, 1) && dma_offsets_fit(m, k, 1);
}
// 64-row weight tiles for the projections of up to 256 such tiles (N <= 16384: qkv, o_proj, down): the PMC passes of the 128-row form showed the split
// launches' traffic to be their fp32 slabs (down at 228 tokens: 29.9 MB written + 31.8 MB re-read next to 47.6 MB of operands), and 64 tiles x 4 parts
// halve them; qkv's 192 tiles need no split at all (bf16 store + the plain RoPE pass instead of 2 slabs).  One user 16.8 -> 15.75 ms at zero
// acceptance, 5.97 -> 5.48 ms at three accepted steps (A/B ATSPEED_FP8_SMALL_BN64 = 0 / 1 (N <= 4096 only: 16.0 / 5.6) / 2 on one box).
static bool wdma8_bn64(int n) {
  static const int on = env_int("ATSPEED_FP8_SMALL_BN64", 2);
  return on && n <= (on >= 2 ? 16384 : 4096) && n % 64 == 0;
}
static int wdma8_split_count(int n, int k) {                           // 1: no split
  const int t128 = wdma8_bn64(n) ? (n + 63) / 64 : (n + 127) / 128, n_kt = k / 128;
  if (t128 >= 150) return 1;
  return std::max(1, std::min(256 / t128, n_kt / 4));                  // at least 4 tiles (512 k) per part (8: the same within 2 %, 16: +3 % per user)
}
template <int BM, int NST, int EPI, bool SPLIT, int WM = 2, int BN = 128>
int launch_wdma8_cfg(const unsigned char* xq, const float* sx, const unsigned char* wq, const float* sw, void* c, int m, int n, int k, int ldc,
                     int splits, hipStream_t st, int pk, const RopeEpi& rope = RopeEpi{}) {
  auto kern = gemm_wdma_kernel<BM, BN, NST, EPI, SPLIT, WM, true>;
  constexpr int lds = NST * (BM + BN) * 128;
  static thread_local AtsPerDeviceFlag attr_flag;
  bool& attr_done = attr_flag.cur();
  if (!attr_done) {
    ATS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((n + BN - 1) / BN, SPLIT ? splits : 1), dim3(128 * WM), lds, st, (const void*)xq, (const void*)wq, c, m, n, k, k, ldc, pk, splits, sx, sw, rope);
  ATS_LAUNCH_CHECK();
  ats_count_path(SPLIT ? ATS_PATH_FP8_WDMA_SPLIT : ATS_PATH_FP8_WDMA);
  return ATSPEED_OK;
}
template <int EPI, bool SPLIT>
int launch_wdma8(const unsigned char* xq, const float* sx, const unsigned char* wq, const float* sw, void* c, int m, int n, int k, int ldc,
                 int splits, hipStream_t st, int pk, const RopeEpi& rope = RopeEpi{}) {
  {
    if (EPI == EPI_QKV_ROPE || wdma8_bn64(n)) {                        // (the RoPE epilogue exists for 64-row tiles only: the caller asked ats_gemm_fp8_qkv_rope_applies)
      if (m <= 32)  return launch_wdma8_cfg<32, 8, EPI, SPLIT, 2, 64>(xq, sx, wq, sw, c, m, n, k, ldc, splits, st, pk, rope);     // 12 KB x 8
      if (m <= 64)  return launch_wdma8_cfg<64, 8, EPI, SPLIT, 2, 64>(xq, sx, wq, sw, c, m, n, k, ldc, splits, st, pk, rope);     // 16 KB x 8
      if (m <= 128) return launch_wdma8_cfg<128, 6, EPI, SPLIT, 2, 64>(xq, sx, wq, sw, c, m, n, k, ldc, splits, st, pk, rope);    // 24 KB x 6
      return launch_wdma8_cfg<256, 4, EPI, SPLIT, 4, 64>(xq, sx, wq, sw, c, m, n, k, ldc, splits, st, pk, rope);                  // 40 KB x 4, 8 waves
    }
  }
  if constexpr (EPI == EPI_QKV_ROPE) return ATSPEED_ERR_INVALID;
  else {
  if (m <= 32)  return launch_wdma8_cfg<32, 6, EPI, SPLIT>(xq, sx, wq, sw, c, m, n, k, ldc, splits, st, pk);        // 20 KB x 6
  if (m <= 64)  return launch_wdma8_cfg<64, 6, EPI, SPLIT>(xq, sx, wq, sw, c, m, n, k, ldc, splits, st, pk);        // 24 KB x 6
  if (m <= 128) return launch_wdma8_cfg<128, 4, EPI, SPLIT>(xq, sx, wq, sw, c, m, n, k, ldc, splits, st, pk);       // 32 KB x 4
  return launch_wdma8_cfg<256, 3, EPI, SPLIT, 4>(xq, sx, wq, sw, c, m, n, k, ldc, splits, st, pk);                  // 48 KB x 3, 8 waves
  }
}
// the slabs of a split launch: [splits][m][n] fp32 in `ws`
static bool wdma8_ws_ok(int m, int n, int splits, const void* ws, size_t ws_bytes) {
  return ws && ((uintptr_t)ws & 15) == 0 && (size_t)splits * m * n * sizeof(float) <= ws_bytes;
}

bool ats_gemm_fp8_applies(int m, int n, int k, int ldc, int epilogue) {
  if (epilogue == EPI_SWIGLU && ((ldc & 3) != 0 || n % 32 != 0)) return false;
  if (m <= 256) return wdma8_applies(m, n, k);
  if (k % 256 != 0) return false;
  if (m < 512) return true;                                            // 257-511 tokens (a long prompt's first verification): the ring kernel whatever its fill
  if (mx_split_count(m, n, k) >= 2) return true;                       // thin grids: the ring kernel cut in K
  const int tn = (n + 255) / 256;
  return big_fill_pct(tn * ((m + 255) / 256)) >= 60 || big_fill_pct(tn * ((m + 127) / 128)) >= 60;
}
size_t ats_gemm_fp8_workspace_bytes(int m, int n, int k) {
  if (m <= 256) return wdma8_applies(m, n, k) ? (size_t)wdma8_split_count(n, k) * m * n * sizeof(float) : 0;
  return (size_t)mx_split_count(m, n, k) * m * n * sizeof(float);
}

int ats_gemm_fp8(const void* xq, const float* sx, const void* wq, const float* sw, void* c, int m, int n, int k, int ldc,
                 int epilogue, hipStream_t st, int pk, void* ws, size_t ws_bytes) {
  ATS_REQUIRE(xq && sx && wq && sw && c, ATSPEED_ERR_INVALID, "gemm_fp8: null argument");
  const unsigned char* X = (const unsigned char*)xq; const unsigned char* Wq = (const unsigned char*)wq;
  if (m >= 1 && m <= 256 && wdma8_applies(m, n, k)) {
    ATS_REQUIRE(epilogue != EPI_SWIGLU || ((ldc & 3) == 0 && n % 32 == 0), ATSPEED_ERR_INVALID, "gemm_fp8: SwiGLU needs N %% 32 == 0 and ldc %% 4 == 0");
    ATS_REQUIRE((((uintptr_t)xq | (uintptr_t)wq) & 15) == 0, ATSPEED_ERR_INVALID, "gemm_fp8: operands must be 16-byte aligned");
    int s_ = wdma8_split_count(n, k);
    if (s_ > 1 && !wdma8_ws_ok(m, n, s_, ws, ws_bytes)) s_ = 1;         // no room for the slabs: one part per tile
    if (s_ == 1 && epilogue != EPI_RESID) {
      switch (epilogue) {
        case EPI_STORE:  return launch_wdma8<EPI_STORE, false>(X, sx, Wq, sw, c, m, n, k, ldc, 1, st, pk);
        case EPI_F32:    return launch_wdma8<EPI_F32, false>(X, sx, Wq, sw, c, m, n, k, ldc, 1, st, pk);
        case EPI_SWIGLU: return launch_wdma8<EPI_SWIGLU, false>(X, sx, Wq, sw, c, m, n, k, ldc, 1, st, pk);
      }
      atspeed_set_error("gemm_fp8: unknown epilogue %d", epilogue);
      return ATSPEED_ERR_INVALID;
    }
    if (!wdma8_ws_ok(m, n, s_, ws, ws_bytes) && epilogue == EPI_RESID && k % 256 == 0)       // residual epilogue without room for the slabs: the ring kernel's own (as before round 5)
      return launch_big_fp8<EPI_RESID>(X, sx, Wq, sw, c, m, n, k, ldc, st, pk);
    ATS_REQUIRE(wdma8_ws_ok(m, n, s_, ws, ws_bytes), ATSPEED_ERR_CAPACITY, "gemm_fp8: %d x %d x %d parts of fp32 partial sums do not fit the workspace (%zu bytes)", s_, m, n, ws_bytes);
    ATS_TRY((launch_wdma8<EPI_F32, true>(X, sx, Wq, sw, ws, m, n, k, n, s_, st, pk)));
    switch (epilogue) {
      case EPI_STORE:  return reduce_splits<bf16_t, EPI_STORE>((const float*)ws, c, m, n, ldc, s_, st, nullptr, pk);
      case EPI_F32:    return reduce_splits<bf16_t, EPI_F32>((const float*)ws, c, m, n, ldc, s_, st, nullptr, pk);
      case EPI_RESID:  return reduce_splits<bf16_t, EPI_RESID>((const float*)ws, c, m, n, ldc, s_, st, nullptr, pk);
      case EPI_SWIGLU: return reduce_splits<bf16_t, EPI_SWIGLU>((const float*)ws, c, m, n, ldc, s_, st, nullptr, pk);
    }
    atspeed_set_error("gemm_fp8: unknown epilogue %d", epilogue);
    return ATSPEED_ERR_INVALID;
  }
  ATS_REQUIRE(m >= 1 && n >= 1 && k % 256 == 0, ATSPEED_ERR_INVALID, "gemm_fp8: K=%d must be a multiple of 256", k);
  ATS_REQUIRE(dma_offsets_fit(n, k, 1) && dma_offsets_fit(m, k, 1), ATSPEED_ERR_CAPACITY, "gemm_fp8: an operand of %d x %d or %d x %d bytes exceeds the kernel's 32-bit row offsets", n, k, m, k);
  ATS_REQUIRE(epilogue != EPI_SWIGLU || ((ldc & 3) == 0 && n % 32 == 0), ATSPEED_ERR_INVALID, "gemm_fp8: SwiGLU needs N %% 32 == 0 and ldc %% 4 == 0");
  {
    const int ms = mx_split_count(m, n, k);
    if (ms >= 2 && wdma8_ws_ok(m, n, ms, ws, ws_bytes) && (n % 4) == 0) {
      ATS_TRY(launch_mx_split(X, sx, Wq, sw, (float*)ws, m, n, k, ms, st, pk));
      switch (epilogue) {
        case EPI_STORE:  return reduce_splits<bf16_t, EPI_STORE>((const float*)ws, c, m, n, ldc, ms, st, nullptr, pk);
        case EPI_F32:    return reduce_splits<bf16_t, EPI_F32>((const float*)ws, c, m, n, ldc, ms, st, nullptr, pk);
        case EPI_RESID:  return reduce_splits<bf16_t, EPI_RESID>((const float*)ws, c, m, n, ldc, ms, st, nullptr, pk);
        case EPI_SWIGLU: return reduce_splits<bf16_t, EPI_SWIGLU>((const float*)ws, c, m, n, ldc, ms, st, nullptr, pk);
      }
      atspeed_set_error("gemm_fp8: unknown epilogue %d", epilogue);
      return ATSPEED_ERR_INVALID;
    }
  }
  switch (epilogue) {
    case EPI_STORE:  return launch_big_fp8<EPI_STORE>(X, sx, Wq, sw, c, m, n, k, ldc, st, pk);
    case EPI_F32:    return launch_big_fp8<EPI_F32>(X, sx, Wq, sw, c, m, n, k, ldc, st, pk);
    case EPI_RESID:  return launch_big_fp8<EPI_RESID>(X, sx, Wq, sw, c, m, n, k, ldc, st, pk);
    case EPI_SWIGLU: return launch_big_fp8<EPI_SWIGLU>(X, sx, Wq, sw, c, m, n, k, ldc, st, pk);
  }
  atspeed_set_error("gemm_fp8: unknown epilogue %d", epilogue);
  return ATSPEED_ERR_INVALID;
}

// One user's W8A8 qkv projection as scaled fp32 split-K slabs [splits][m][n] in the workspace, without the reduce pass (ats_gemm_partials'
// fp8 form: RoPE + the KV scatter sum the slabs).  *splits_out = 0: this shape takes another path, the caller runs ats_gemm_fp8.
int ats_gemm_fp8_partials(const void* xq, const float* sx, const void* wq, const float* sw, int m, int n, int k, void* ws, size_t ws_bytes,
                          hipStream_t st, int* splits_out, int pk) {
  *splits_out = 0;
  if (m < 1 || m > 256 || !wdma8_applies(m, n, k) || (n % 4) != 0) return ATSPEED_OK;
  const int s_ = wdma8_split_count(n, k);
  if (s_ < 2 || !wdma8_ws_ok(m, n, s_, ws, ws_bytes)) return ATSPEED_OK;
  ATS_REQUIRE(xq && sx && wq && sw && (((uintptr_t)xq | (uintptr_t)wq) & 15) == 0, ATSPEED_ERR_INVALID, "gemm_fp8_partials: null / unaligned operand");
  ATS_TRY((launch_wdma8<EPI_F32, true>((const unsigned char*)xq, sx, (const unsigned char*)wq, sw, ws, m, n, k, n, s_, st, pk)));
  *splits_out = s_;
  return ATSPEED_OK;
}

// h += xq wq^T (W8A8), then the NEXT op's input norm of the updated rows: xn (16-bit operand; may be null when q_out is given) and / or the
// e4m3 rows + per-token scales (q_out, s_out) a following W8A8 projection consumes.  One user's tokens: split-K slabs + ONE fused reduce /
// residual / norm / quantisation pass; batched: the ring kernel's residual epilogue, then the norm kernels.
int ats_gemm_fp8_resid_norm(const void* xq, const float* sx, const void* wq, const float* sw, void* h, int m, int n, int k, int ldh,
                            const void* norm_w, void* xn, void* q_out, float* s_out, float eps, void* ws, size_t ws_bytes, hipStream_t st, int pk) {
  if (m <= 0) return ATSPEED_OK;
  ATS_REQUIRE(xn || (q_out && s_out), ATSPEED_ERR_INVALID, "gemm_fp8_resid_norm: no output for the norm");
  if (m <= 256 && wdma8_applies(m, n, k) && n <= 8192 && (n % 4) == 0 && (ldh % 4) == 0) {
    const int s_ = wdma8_split_count(n, k);
    if (wdma8_ws_ok(m, n, s_, ws, ws_bytes)) {
      ATS_REQUIRE(xq && sx && wq && sw && h && norm_w && (((uintptr_t)xq | (uintptr_t)wq) & 15) == 0, ATSPEED_ERR_INVALID, "gemm_fp8_resid_norm: null / unaligned operand");
      ATS_TRY((launch_wdma8<EPI_F32, true>((const unsigned char*)xq, sx, (const unsigned char*)wq, sw, ws, m, n, k, n, s_, st, pk)));
      FusedNorm fn{norm_w, xn, eps, false, q_out, s_out};
      ATS_TRY((reduce_splits<bf16_t, EPI_RESID>((const float*)ws, h, m, n, ldh, s_, st, &fn, pk)));
      if (fn.done) return ATSPEED_OK;
      return q_out ? ats_rmsnorm_quant_fp8(h, norm_w, xn, q_out, s_out, m, n, eps, st, pk) : ats_rmsnorm(h, norm_w, xn, m, n, eps, ATS_HALF, st, pk);
    }
  }
  {
    const int ms = mx_split_count(m, n, k);                           // thin grid: K-split ring + ONE fused reduce / residual / norm / quantisation pass
    if (ms >= 2 && wdma8_ws_ok(m, n, ms, ws, ws_bytes) && n <= 8192 && (n % 4) == 0 && (ldh % 4) == 0) {
      ATS_REQUIRE(xq && sx && wq && sw && h && norm_w, ATSPEED_ERR_INVALID, "gemm_fp8_resid_norm: null operand");
      ATS_TRY(launch_mx_split((const unsigned char*)xq, sx, (const unsigned char*)wq, sw, (float*)ws, m, n, k, ms, st, pk));
      FusedNorm fn{norm_w, xn, eps, false, q_out, s_out};
      ATS_TRY((reduce_splits<bf16_t, EPI_RESID>((const float*)ws, h, m, n, ldh, ms, st, &fn, pk)));
      if (fn.done) return ATSPEED_OK;
      return q_out ? ats_rmsnorm_quant_fp8(h, norm_w, xn, q_out, s_out, m, n, eps, st, pk) : ats_rmsnorm(h, norm_w, xn, m, n, eps, ATS_HALF, st, pk);
    }
  }
  ATS_TRY(ats_gemm_fp8(xq, sx, wq, sw, h, m, n, k, ldh, EPI_RESID, st, pk, ws, ws_bytes));
  return q_out ? ats_rmsnorm_quant_fp8(h, norm_w, xn, q_out, s_out, m, n, eps, st, pk) : ats_rmsnorm(h, norm_w, xn, m, n, eps, ATS_HALF, st, pk);
}

// the fp8 qkv projection with RoPE + the KV scatter in its epilogue (see ats_gemm_qkv_rope); the caller checks ats_gemm_fp8_applies,
// head_dim == 128 and hidden % 256 == 0 (ats_gemm_fp8_qkv_rope_applies)
bool ats_gemm_fp8_qkv_rope_applies(int m, int hidden, int head_dim) {
  if (!ats_switch(ATS_SW_FUSE_QKV_ROPE) || m < 1 || head_dim != 128 || hidden % 256 != 0) return false;
  // one user's tokens: the weight-streaming kernel's epilogue, where the projection runs as 150-256 unsplit tiles of 64 weight rows (Llama-7B: 192)
  if (m <= 256) return wdma8_applies(m, 3 * hidden, hidden) && wdma8_bn64(3 * hidden) && wdma8_split_count(3 * hidden, hidden) == 1 && (3 * hidden + 63) / 64 <= 256;
  return ats_gemm_fp8_applies(m, 3 * hidden, hidden, 3 * hidden, EPI_STORE);   // the ring kernel's epilogue
}

int ats_gemm_fp8_qkv_rope(const void* xq, const float* sx, const void* wq, const float* sw, void* qkv, int m, int hidden, const RopeEpi& rope,
                          hipStream_t st, int pk) {
  ATS_REQUIRE(xq && sx && wq && sw && qkv && rope.rows && rope.cos_tab && rope.sin_tab && rope.hidden == hidden, ATSPEED_ERR_INVALID,
              "gemm_fp8_qkv_rope: null / inconsistent argument");
  ATS_REQUIRE(m >= 1 && hidden % 256 == 0, ATSPEED_ERR_INVALID, "gemm_fp8_qkv_rope: hidden=%d must be a multiple of 256", hidden);
  if (m <= 256) {
    ATS_REQUIRE(ats_gemm_fp8_qkv_rope_applies(m, hidden, 128) && (((uintptr_t)xq | (uintptr_t)wq) & 15) == 0, ATSPEED_ERR_INVALID,
                "gemm_fp8_qkv_rope: %d x %d is not a shape of the weight-streaming kernel's RoPE epilogue", m, hidden);
    return launch_wdma8<EPI_QKV_ROPE, false>((const unsigned char*)xq, sx, (const unsigned char*)wq, sw, qkv, m, 3 * hidden, hidden, 3 * hidden, 1, st, pk, rope);
  }
  return launch_big_fp8<EPI_QKV_ROPE>((const unsigned char*)xq, sx, (const unsigned char*)wq, sw, qkv, m, 3 * hidden, hidden, 3 * hidden, st, pk, rope);
}



// the same GEMMs on operands in the packed layout (what the bf16 / fp8 engine runs; atspeed_pack_rows makes them): a, w packed;
// the SwiGLU epilogue's output packed too (it is the down projection's operand), every other output row-major

}  // namespace ATS_NS

#ifndef ATS_F16_FLAVOUR          // the C ABI exists once; it picks the flavour by the dtype code (fp8 and packed entry points: bf16)
extern "C" int atspeed_gemm_fp8(const void* xq, const float* sx, const void* wq, const float* sw, void* c, int32_t m, int32_t n,
                                int32_t k, int32_t ldc, int32_t epilogue, void* workspace, size_t workspace_bytes, void* stream) {
  return ats_bf16::ats_gemm_fp8(xq, sx, wq, sw, c, m, n, k, ldc, epilogue, (hipStream_t)stream, 0, workspace, workspace_bytes);
}

extern "C" int atspeed_gemm(const void* a, const void* w, void* c, int32_t m, int32_t n, int32_t k, int32_t lda,
                            int32_t ldc, int32_t dtype, int32_t epilogue, void* workspace, size_t workspace_bytes,
                            void* stream) {
  return ATS_KD(dtype, ats_gemm(a, w, c, m, n, k, lda, ldc, dtype, epilogue, workspace, workspace_bytes, (hipStream_t)stream, 0, nullptr));
}

extern "C" int atspeed_gemm_packed(const void* a, const void* w, void* c, int32_t m, int32_t n, int32_t k, int32_t ldc, int32_t epilogue,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  return ats_bf16::ats_gemm(a, w, c, m, n, k, k, ldc, ATSPEED_BF16, epilogue, workspace, workspace_bytes, (hipStream_t)stream, 1, nullptr);
}

extern "C" int atspeed_gemm_fp8_packed(const void* xq, const float* sx, const void* wq, const float* sw, void* c, int32_t m, int32_t n,
                                       int32_t k, int32_t ldc, int32_t epilogue, void* workspace, size_t workspace_bytes, void* stream) {
  return ats_bf16::ats_gemm_fp8(xq, sx, wq, sw, c, m, n, k, ldc, epilogue, (hipStream_t)stream, 1, workspace, workspace_bytes);
}
#endif
