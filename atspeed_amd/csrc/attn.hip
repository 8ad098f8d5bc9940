// Tree attention over a slot-addressed KV cache.
//
// The reference feeds HF attention a dense additive mask [1,1,T,S] of 0 / finfo.min built by
// concatenating parent rows and identity blocks (beamSD.py:89,204-209,396-400).  Here a query
// row carries a bitset over cache slots (prompt prefix + one ancestor per accepted/draft
// block + itself); the kernel compacts the set bits and touches only visible slots.
//
// Two kernels:
//  * tree_attn_mfma_kernel (bf16, head_dim 64/128): flash-style, one workgroup per (64 query
//    rows, head), K tile and transposed V tile staged in LDS, both products on MFMA 16x16x32.
//    Products are arranged so nothing moves between them: S^T = K.Q^T puts the key on the
//    accumulator rows and the query on the lane (col = lane & 15), so the softmax statistics of a
//    query live in the 4 lanes that share lane & 15, and the exponentiated S^T registers ARE the
//    B operand of O^T = V^T.P^T (k-slots permuted identically on the V^T operand).
//  * tree_attn_kernel (fp32 parity mode / odd head dims): one wave per (token, head), scalar.
#include "internal.h"
#include <type_traits>
#include <cstdlib>

namespace ATS_NS {

namespace {

constexpr int kMaxSlots = 2048;
template <typename T>
__global__ __launch_bounds__(256) void tree_attn_kernel(const T* __restrict__ q, int ldq, const SegTable* __restrict__ tab, size_t layer_off,
                                                        int vis_words, T* __restrict__ out, int ldo, int pk,
                                                        int n_heads, int head_dim, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = blockIdx.x;                        // global row
  int si = 0, hi = tab->n;
  while (hi - si > 1) { int mid = (si + hi) >> 1; if (tab->seg[mid].row0 <= t) si = mid; else hi = mid; }
  const Seg& sg = tab->seg[si];
  const T* kc = reinterpret_cast<const T*>(reinterpret_cast<const char*>(sg.kc) + layer_off);
  const T* vc = reinterpret_cast<const T*>(reinterpret_cast<const char*>(sg.vc) + layer_off);
  const uint64_t* vis_row = sg.vis + (size_t)(t - sg.row0) * vis_words;
  const int n_slots = sg.n_slots;
  const int h = blockIdx.y * 4 + wave;
  // per-wave LDS: q vector (head_dim f32), slot list (n_cap int32), probs (n_cap f32)
  const int n_cap = vis_words * 64;
  float* qs = reinterpret_cast<float*>(smem) + wave * (head_dim + 2 * n_cap);
  int* slot_list = reinterpret_cast<int*>(qs + head_dim);
  float* prob = reinterpret_cast<float*>(slot_list + n_cap);
  if (h >= n_heads) return;                        // whole wave exits together (h is wave-uniform)

  const int hidden = n_heads * head_dim;
  const T* qrow = q + (size_t)t * ldq + h * head_dim;
  for (int d = lane; d < head_dim; d += 64) qs[d] = Elt<T>::load(qrow + d) * scale;

  // compact visible slots (ascending)
  int n_vis = 0;
  const int words = (n_slots + 63) >> 6;
  for (int w = 0; w < words; ++w) {
    uint64_t bits = vis_row[w];
    if (w == words - 1 && (n_slots & 63)) bits &= (~0ull) >> (64 - (n_slots & 63));
    if ((bits >> lane) & 1ull) slot_list[n_vis + __popcll(bits & ((1ull << lane) - 1ull))] = w * 64 + lane;
    n_vis += __popcll(bits);
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // phase 1: scores
  float mx = -INFINITY;
  for (int i = lane; i < n_vis; i += 64) {
    const T* kr = kc + (size_t)slot_list[i] * hidden + h * head_dim;
    float acc = 0.f;
    if constexpr (sizeof(T) == 2) {
      for (int d = 0; d < head_dim; d += 8) {
        uint4 raw = *reinterpret_cast<const uint4*>(kr + d);
        const bf16_t* e = reinterpret_cast<const bf16_t*>(&raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += qs[d + j] * bf2f(e[j]);
      }
    } else {
      for (int d = 0; d < head_dim; d += 4) {
        float4 raw = *reinterpret_cast<const float4*>(kr + d);
        acc += qs[d] * raw.x + qs[d + 1] * raw.y + qs[d + 2] * raw.z + qs[d + 3] * raw.w;
      }
    }
    prob[i] = acc;
    mx = fmaxf(mx, acc);
  }
  mx = wave_max_f32(mx);
  // phase 2: softmax (fp32, like HF eager attention)
  float sum = 0.f;
  for (int i = lane; i < n_vis; i += 64) {
    float p = __expf(prob[i] - mx);
    prob[i] = p;
    sum += p;
  }
  sum = wave_sum_f32(sum);
  const float inv = n_vis > 0 ? 1.f / sum : 0.f;
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // phase 3: out[d] = sum_i p_i V[slot_i][d]
  for (int d = lane; d < head_dim; d += 64) {
    float acc = 0.f;
    for (int i = 0; i < n_vis; ++i) acc += prob[i] * Elt<T>::load(vc + (size_t)slot_list[i] * hidden + h * head_dim + d);
    Elt<T>::store(out + ats_opnd_idx<sizeof(T)>(pk, t, h * head_dim + d, ldo), acc * inv);      // the o_proj operand: packed when pk
  }
}

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned lds_off(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}

// Two pairs of transposed V reads AND their wait in ONE asm statement: the outputs are filled asynchronously, so anything the compiler
// may put between a read and the wait (a copy, or a spill store when registers are tight: the 16-wave form is capped at 128 VGPRs)
// would see stale registers.  Early-clobber outputs: they must not share a register with the address.
template <int OA0, int OB0, int OA1, int OB1>
__device__ __forceinline__ void tr_read_2pairs(u32x2_t& a0, u32x2_t& b0, u32x2_t& a1, u32x2_t& b1, unsigned addr) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %4 offset:%5\n\tds_read_b64_tr_b16 %1, %4 offset:%6\n\t"
      "ds_read_b64_tr_b16 %2, %4 offset:%7\n\tds_read_b64_tr_b16 %3, %4 offset:%8\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(a0), "=&v"(b0), "=&v"(a1), "=&v"(b1)
      : "v"(addr), "i"(OA0), "i"(OB0), "i"(OA1), "i"(OB1)
      : "memory");
}

// K / V tiles HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers).  A DMA instruction fills 64 consecutive 16-byte LDS
// positions, so a swizzle or a row padding of the LDS image is applied on the source side: lane i of instruction j fetches whatever
// belongs at position 64 j + i.
// (m0 is written here without an "m0" clobber on purpose: m0 is a RESERVED register for LLVM's AMDGPU backend -- it never keeps a value
// live in it across instructions, it re-materialises m0 glued to each of its own m0 readers -- and hipcc rejects the clobber with
// -Winline-asm "clobber list contains reserved registers ... may lead to undefined behaviour".)
#define ATS_ATTN_DMA16(voff, sbase, m0v) \
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(m0v) : "memory")

// ---------------------------------------------------------------------------- MFMA kernel (bf16)
// blockIdx.x walks the 64-row query tiles of all segments (tile -> segment through the table's qtile arrays)
// NSTG = 0: K / V tiles staged through registers into one static LDS tile, the next tile's loads in flight during a tile's products (two
// barriers per tile).  NSTG >= 3 (one user's forwards: up to 256 workgroups, one per CU): a RING of NSTG tiles filled by LDS-DMA, NSTG - 1
// tiles in flight -- with 2-5 tiles of K/V per user the whole cache is on its way before the first product, where the register form exposed
// a memory round trip per tile (2.1 us per tile against 0.6 us of work: tools/su_trace.sh) -- hand-counted vmcnt, one barrier per tile.
// Nothing the compiler knows to be a vector-memory load may sit in that loop (its waits would drain the ring: the counter is in order), so
// the visibility words of the tile's rows are copied to LDS first and the query fragments are forced to arrive before the first DMA.
template <int DH, int NW, int NSTG = 0>   // NW waves per workgroup = 16*NW query rows per tile
__global__ __launch_bounds__(64 * NW) void tree_attn_mfma_kernel(const bf16_t* __restrict__ q, int ldq, const SegTable* __restrict__ tab,
                                                             size_t layer_off, int vis_words,
                                                             bf16_t* __restrict__ out, int ldo, int pk,
                                                             int n_heads, float scale) {
  constexpr int KCH = DH / 8;                 // 16-byte chunks per K row
  constexpr int VROW = DH * 2 + 32;           // V tile row stride in bytes: 8 rows x 32 B of a transposed read cover all 64 banks
  constexpr int VCH = VROW / 16;              // 16-byte positions per V row (the last two are padding)
  constexpr int DT = DH / 16;                 // output d-tiles
  constexpr int KS = DH / 32;                 // k-steps of the QK product
  constexpr int KBYTES = 64 * DH * 2, TILE = KBYTES + 64 * VROW;
  constexpr bool RING = NSTG >= 3;
  __shared__ __attribute__((aligned(16))) unsigned char ks_static[RING ? 16 : 64 * DH * 2];
  __shared__ __attribute__((aligned(16))) unsigned char vs_static[RING ? 16 : 64 * VROW];   // V tile row-major; consumed column-wise by ds_read_b64_tr_b16
  extern __shared__ __attribute__((aligned(16))) unsigned char ring_smem[];                   // RING: [NSTG][K tile | V tile][16 NW rows x n_tiles visibility words]
  unsigned char* ks_lds = ks_static;
  unsigned char* vs_lds = vs_static;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, lq = lane & 15;
  // 1-D grid over (head, query tile).  Workgroup L runs on XCD L % 8: give every XCD a contiguous range of work items, heads
  // outermost, so that the query tiles of one (user, head) -- which read the same K/V rows -- run back to back behind one L2
  const int n_qt = tab->n_qtiles;
  int wi;
  {
    const int total = gridDim.x, q8 = total >> 3, r8 = total & 7, x = blockIdx.x & 7;
    wi = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (blockIdx.x >> 3);
  }
  const int h = wi / n_qt, qt = wi - h * n_qt;
  const int hidden = n_heads * DH;
  const Seg& sg = tab->seg[tab->qtile_seg[qt]];
  const bf16_t* kc = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(sg.kc) + layer_off);
  const bf16_t* vc = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(sg.vc) + layer_off);
  const int n_slots = sg.n_slots;
  const int lrow = tab->qtile_idx[qt] * (16 * NW) + wave * 16 + lq;  // row inside the segment
  const bool qok = lrow < sg.n_tok;
  const int qrow = sg.row0 + lrow;                                           // row in the batched buffers
  const uint64_t* vis_row = sg.vis + (size_t)lrow * vis_words;

  s16x8_t qf[KS];
  auto load_q = [&]() {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (qok) qf[ks] = *reinterpret_cast<const s16x8_t*>(q + (size_t)qrow * ldq + h * DH + ks * 32 + g * 8);
      else qf[ks] = s16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
    }
  };
  if constexpr (NSTG < 3) load_q();            // (the ring form starts its K/V tiles first and fetches the queries behind them)
  f32x4_t o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) o[d] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const int n_tiles = (n_slots + 63) >> 6;
  // K/V of tile kt+1 travel to registers while tile kt is being multiplied (one tile of global-load latency hidden)
  constexpr int NLD = RING ? 1 : (64 * KCH + 64 * NW - 1) / (64 * NW);
  uint4 kreg[NLD], vreg[NLD];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int qi = tid + i * (64 * NW);
      const int r = qi / KCH, c = qi % KCH;
      const int key = kt * 64 + r;
      kreg[i] = make_uint4(0, 0, 0, 0); vreg[i] = make_uint4(0, 0, 0, 0);
      if (qi < 64 * KCH && key < n_slots) {
        const size_t off = (size_t)key * hidden + h * DH + c * 8;
        kreg[i] = *reinterpret_cast<const uint4*>(kc + off);
        vreg[i] = *reinterpret_cast<const uint4*>(vc + off);
      }
    }
  };
  // RING: the same LDS image by DMA.  NI instructions per tile (K: KCH, V: VCH), dealt round-robin to the waves; rows past n_slots (last
  // tile) are fetched from the last valid row: finite values, masked out by the visibility word
  constexpr int NI = KCH + VCH;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int lds0 = RING ? __builtin_amdgcn_readfirstlane((int)lds_off(ring_smem)) : 0;
  const unsigned long long kbase = (unsigned long long)kc + (size_t)h * DH * 2, vbase = (unsigned long long)vc + (size_t)h * DH * 2;
  auto dma_tile = [&](int kt, int buf) {
#pragma unroll
    for (int j0 = 0; j0 < NI; j0 += NW) {
      const int j = j0 + wave_u;                       // wave-uniform (SGPR): the DMA's M0 and branch are scalar
      if (j < KCH) {
        const int P = j * 64 + lane, r = P / KCH, cs = P % KCH;
        const int key = min(kt * 64 + r, n_slots - 1);
        const unsigned voff = (unsigned)key * (unsigned)(hidden * 2) + ((cs ^ (r & 7)) * 16);
        ATS_ATTN_DMA16(voff, kbase, lds0 + buf * TILE + j * 1024);
      } else if (j < NI) {
        const int P = (j - KCH) * 64 + lane, r = P / VCH, c = P % VCH;
        const int key = min(kt * 64 + r, n_slots - 1);
        const unsigned voff = (unsigned)key * (unsigned)(hidden * 2) + (c < KCH ? c * 16 : 0);
        ATS_ATTN_DMA16(voff, vbase, lds0 + buf * TILE + KBYTES + (j - KCH) * 1024);
      }
    }
  };
  uint64_t* vis_lds = reinterpret_cast<uint64_t*>(ring_smem + (RING ? NSTG : 0) * TILE) + (size_t)(wave * 16 + lq) * n_tiles;   // this lane's row
  if constexpr (RING) {
    // the first NSTG - 1 tiles leave before anything else is fetched; queries and visibility words travel behind them and everything is
    // awaited once (one memory round trip instead of three in a row), so that no load the compiler knows of is pending inside the ring
#pragma unroll
    for (int t = 0; t < NSTG - 1; ++t) if (t < n_tiles) dma_tile(t, t);
    load_q();
    for (int w = g; w < n_tiles; w += 4) vis_lds[w] = qok ? vis_row[w] : 0ull;     // the 4 lanes of a row share its words
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(qf[ks]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    load_tile(0);
  }
  for (int kt = 0; kt < n_tiles; ++kt) {
    uint64_t word;
    if constexpr (RING) {
      // tile kt has landed: all but the pieces of the NSTG - 2 younger tiles (this wave issues (NI - wave + NW - 1) / NW pieces per tile)
      if (kt + NSTG - 2 < n_tiles) {
        if (wave_u < NI % NW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * ((NI + NW - 1) / NW)) : "memory");
        else                  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * (NI / NW)) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();                                 // everyone's pieces of tile kt; and tile kt - 1 is fully consumed
      if (kt + NSTG - 1 < n_tiles) dma_tile(kt + NSTG - 1, (kt + NSTG - 1) % NSTG);
      ks_lds = ring_smem + (kt % NSTG) * TILE;
      vs_lds = ks_lds + KBYTES;
      word = vis_lds[kt];
    } else {
    __syncthreads();                                   // previous tile fully consumed
    // ---- stage K (swizzled rows) and V (row-major, padded rows)
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int qi = tid + i * (64 * NW);
      const int r = qi / KCH, c = qi % KCH;
      if (qi < 64 * KCH) {
        *reinterpret_cast<uint4*>(ks_lds + (r * KCH + (c ^ (r & 7))) * 16) = kreg[i];
        *reinterpret_cast<uint4*>(vs_lds + r * VROW + c * 16) = vreg[i];
      }
    }
    __syncthreads();
    if (kt + 1 < n_tiles) load_tile(kt + 1);
    word = qok ? vis_row[kt] : 0ull;
    }
    if (kt == n_tiles - 1 && (n_slots & 63)) word &= (~0ull) >> (64 - (n_slots & 63));
    if (__ballot(word != 0ull) == 0ull) continue;      // this wave's 16 rows see nothing here (wave-uniform)

    // ---- S^T = K_tile . Q^T : s[c][r] = score(key = kt*64 + c*16 + g*4 + r, query = lq)
    f32x4_t s[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      s[c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        int r = c * 16 + lq, ch = ks * 4 + g;
        s16x8_t kf = *reinterpret_cast<const s16x8_t*>(ks_lds + (r * KCH + (ch ^ (r & 7))) * 16);
        s[c] = ATS_MFMA_16x16x32(__builtin_bit_cast(bf16x8_t, kf), __builtin_bit_cast(bf16x8_t, qf[ks]), s[c]);
      }
    }
    float mt = -INFINITY;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bool v = (word >> (c * 16 + g * 4 + r)) & 1ull;
        float x = v ? s[c][r] * scale : -INFINITY;
        s[c][r] = x;
        mt = fmaxf(mt, x);
      }
    mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_new);
    float psum = 0.f;
    u32x4_t pf[2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const float p0 = (s[c][r] == -INFINITY) ? 0.f : __expf(s[c][r] - m_new);
        const float p1 = (s[c][r + 1] == -INFINITY) ? 0.f : __expf(s[c][r + 1] - m_new);
        psum += p0 + p1;
        // packed explicitly: element-wise (short)f2bf(p) into the 8-vector was miscompiled in the <64, 16> instantiation (1024 threads,
        // 128-VGPR cap: wrong rows >= 64) once f2bf became the hardware conversion
        pf[c >> 1][(c & 1) * 2 + (r >> 1)] = f2bf_pk(p0, p1);
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      o[d][0] *= alpha; o[d][1] *= alpha; o[d][2] *= alpha; o[d][3] *= alpha;
    }
    // ---- O^T += V^T . P^T  (k-slot j<4 -> key block 2kk, j>=4 -> key block 2kk+1, both keys 4g + (j&3))
    // the A operand (V^T) comes straight out of the row-major V tile: per 16-lane group a transposed read takes the 4 keys
    // 16*blk + 4g .. +3 x 16 columns of d-tile d; lane 4q+p addresses key q, columns 4p..4p+3, lane lq receives column lq
    const unsigned vaddr = lds_off(vs_lds) + (4 * g + (lq >> 2)) * VROW + (lq & 3) * 8;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      // two d-tiles per statement
      auto pv_pair = [&](auto dc) {
        constexpr int d = decltype(dc)::value;
        u32x2_t a0, b0, a1, b1;
        if (kk == 0) tr_read_2pairs<0 * 16 * VROW + d * 32, 1 * 16 * VROW + d * 32, 0 * 16 * VROW + (d + 1) * 32, 1 * 16 * VROW + (d + 1) * 32>(a0, b0, a1, b1, vaddr);
        else         tr_read_2pairs<2 * 16 * VROW + d * 32, 3 * 16 * VROW + d * 32, 2 * 16 * VROW + (d + 1) * 32, 3 * 16 * VROW + (d + 1) * 32>(a0, b0, a1, b1, vaddr);
        u32x4_t v0 = {a0[0], a0[1], b0[0], b0[1]}, v1 = {a1[0], a1[1], b1[0], b1[1]};
        o[d] = ATS_MFMA_16x16x32(__builtin_bit_cast(bf16x8_t, v0), __builtin_bit_cast(bf16x8_t, pf[kk]), o[d]);
        o[d + 1] = ATS_MFMA_16x16x32(__builtin_bit_cast(bf16x8_t, v1), __builtin_bit_cast(bf16x8_t, pf[kk]), o[d + 1]);
      };
      pv_pair(std::integral_constant<int, 0>{}); pv_pair(std::integral_constant<int, 2>{});
      if constexpr (DT == 8) { pv_pair(std::integral_constant<int, 4>{}); pv_pair(std::integral_constant<int, 6>{}); }
    }
  }
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  if (qok) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      ushort4 v4;
      v4.x = f2bf(o[d][0] * inv); v4.y = f2bf(o[d][1] * inv); v4.z = f2bf(o[d][2] * inv); v4.w = f2bf(o[d][3] * inv);
      *reinterpret_cast<ushort4*>(out + ats_opnd_idx<2>(pk, qrow, h * DH + d * 16 + g * 4, ldo)) = v4;
    }
  }
}


// ---------------------------------------------------------------------------- MFMA kernel, 32 query rows per wave (bf16)
// Same algorithm on v_mfma_f32_32x32x16_bf16: a wave owns 32 query rows, so one pass over the K / V tile in LDS feeds twice the
// MFMA work of the 16-row form (which is bound by exactly those LDS reads: every wave re-reads the whole 32 KB tile).
//   S^T block = K[32 keys] . Q^T[32 queries]: lane (lc = lane & 31, hi = lane >> 5) holds query lc and, in accumulator register i,
//   key 8*(i/4) + 4*hi + (i%4) of the block -> a query's statistics live in the lane pair (lc, lc + 32).
//   O^T += V^T . P^T in 16-key steps: the lane's 8 exponentiated registers of a step ARE its B operand (k-slot j <-> key
//   16t + 8*(j/4) + 4*hi + (j%4)); the A operand takes the same keys out of the row-major V tile with two transposed reads.
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// K / V tiles travel HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers) into a double buffer, tile kt+1 while
// tile kt is multiplied; ONE barrier per tile.  A DMA instruction fills 64 consecutive 16-byte LDS positions, so the K swizzle and
// the V row padding are applied on the source side: lane i of instruction j fetches whatever belongs at position 64 j + i.
template <int DH, int NW>   // NW waves per workgroup = 32*NW query rows per tile
__global__ __launch_bounds__(64 * NW, 2) void tree_attn32_kernel(const bf16_t* __restrict__ q, int ldq, const SegTable* __restrict__ tab,
                                                                 size_t layer_off, int vis_words, bf16_t* __restrict__ out, int ldo, int pk,
                                                                 int n_heads, float scale) {
  constexpr int KCH = DH / 8;                 // 16-byte chunks per K row
  constexpr int VROW = DH * 2 + 32;           // V tile row stride in bytes
  constexpr int VCH = VROW / 16;              // 16-byte positions per V row (the last two are padding)
  constexpr int DB = DH / 32;                 // 32-row blocks of O^T
  constexpr int KS = DH / 16;                 // k-steps of the QK product
  constexpr int KBYTES = 64 * DH * 2, TILE = KBYTES + 64 * VROW;
  constexpr int NI = KCH + VCH;               // DMA instructions per tile (K: KCH, V: VCH), dealt round-robin to the waves
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2][K tile | V tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 31, hi = lane >> 5;
  const int n_qt = tab->n_qtiles;
  int wi;
  {
    const int total = gridDim.x, q8 = total >> 3, r8 = total & 7, x = blockIdx.x & 7;
    wi = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (blockIdx.x >> 3);
  }
  const int h = wi / n_qt, qt = wi - h * n_qt;
  const int hidden = n_heads * DH;
  const Seg& sg = tab->seg[tab->qtile_seg[qt]];
  const unsigned long long kbase = (unsigned long long)sg.kc + layer_off + (size_t)h * DH * 2;
  const unsigned long long vbase = (unsigned long long)sg.vc + layer_off + (size_t)h * DH * 2;
  const int n_slots = sg.n_slots;
  const int lrow = tab->qtile_idx[qt] * (32 * NW) + wave * 32 + lc;
  const bool qok = lrow < sg.n_tok;
  const int qrow = sg.row0 + lrow;
  const uint64_t* vis_row = sg.vis + (size_t)lrow * vis_words;
  const int lds0 = __builtin_amdgcn_readfirstlane((int)lds_off(smem));
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  // rows past n_slots (last tile) are fetched from the last valid row: finite values, masked out by the visibility word
  auto dma_tile = [&](int kt, int buf) {
#pragma unroll
    for (int j0 = 0; j0 < NI; j0 += NW) {
      const int j = j0 + wave_u;                       // wave-uniform (SGPR): the DMA's M0 and branch are scalar
      if (j < KCH) {
        const int P = j * 64 + lane, r = P / KCH, cs = P % KCH;
        const int key = min(kt * 64 + r, n_slots - 1);
        const unsigned voff = (unsigned)key * (unsigned)(hidden * 2) + ((cs ^ (r & (KCH - 1))) * 16);
        ATS_ATTN_DMA16(voff, kbase, lds0 + buf * TILE + j * 1024);
      } else if (j < NI) {
        const int P = (j - KCH) * 64 + lane, r = P / VCH, c = P % VCH;
        const int key = min(kt * 64 + r, n_slots - 1);
        const unsigned voff = (unsigned)key * (unsigned)(hidden * 2) + (c < KCH ? c * 16 : 0);
        ATS_ATTN_DMA16(voff, vbase, lds0 + buf * TILE + KBYTES + (j - KCH) * 1024);
      }
    }
  };

  const int n_tiles = (n_slots + 63) >> 6;
  if (n_tiles > 0) dma_tile(0, 0);
  s16x8_t qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (qok) qf[ks] = *reinterpret_cast<const s16x8_t*>(q + (size_t)qrow * ldq + h * DH + ks * 16 + hi * 8);
    else qf[ks] = s16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
  }
  f32x16_t o[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
  const float sl2 = scale * 1.4426950408889634f;        // scores carried in the log2 domain
  float m_run = -INFINITY, l_run = 0.f;

  auto kpos = [](int r, int c) { return (r * KCH + (c ^ (r & (KCH - 1)))) * 16; };
  for (int kt = 0; kt < n_tiles; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile kt have landed
    __syncthreads();                                   // everyone's have, and the other buffer is fully consumed
    if (kt + 1 < n_tiles) dma_tile(kt + 1, (kt + 1) & 1);
    const unsigned char* ks_lds = smem + (kt & 1) * TILE;
    const unsigned char* vs_lds = ks_lds + KBYTES;
    uint64_t word = qok ? vis_row[kt] : 0ull;
    if (kt == n_tiles - 1 && (n_slots & 63)) word &= (~0ull) >> (64 - (n_slots & 63));
    if (__ballot(word != 0ull) == 0ull) continue;      // this wave's 32 rows see nothing here (wave-uniform)

    // V^T fragments: 16-lane group gg = lane >> 4 takes d-columns 16*(gg & 1) .. +15 of the block and the 4 keys 4*hi .. +3 (second
    // read: +8); inside the group lane 4q + p addresses key q, columns 4p .. 4p+3 and receives column (lane & 15)
    const unsigned vaddr = lds_off(vs_lds) + (4 * hi + ((lane & 15) >> 2)) * VROW + (((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;
    // the 64-key tile is consumed as two 32-key blocks, each one online-softmax step (16 score registers live instead of 32)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const unsigned wb = (unsigned)(word >> (32 * b));
      if (__ballot(wb != 0u) == 0ull) continue;        // wave-uniform
      f32x16_t sc;
#pragma unroll
      for (int i = 0; i < 16; ++i) sc[i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        s16x8_t kf = *reinterpret_cast<const s16x8_t*>(ks_lds + kpos(b * 32 + lc, ks * 2 + hi));
        sc = ATS_MFMA_32x32x16(__builtin_bit_cast(bf16x8_t, kf), __builtin_bit_cast(bf16x8_t, qf[ks]), sc);
      }
      // softmax in the log2 domain (one v_exp_f32 per score); the VALU work per score is what bounds this kernel, not the MFMAs:
      // masks from a pre-shifted word with compile-time bit positions, hardware bf16 packing, rescale of O only when a maximum moved
      const unsigned wsh = wb >> (4 * hi);
      float mt = -INFINITY;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float x = (wsh & (1u << (8 * (i >> 2) + (i & 3)))) ? sc[i] * sl2 : -INFINITY;
        sc[i] = x;
        mt = fmaxf(mt, x);
      }
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float m_new = fmaxf(m_run, mt);
      const float m_sub = (m_new == -INFINITY) ? 0.f : m_new;          // nothing visible yet: exp2(-inf - 0) = 0
      float psum = 0.f;
      u32x4_t pf[2];
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const float p0 = __builtin_amdgcn_exp2f(sc[i] - m_sub), p1 = __builtin_amdgcn_exp2f(sc[i + 1] - m_sub);
        psum += p0 + p1;
        unsigned pk;
        asm(ATS_CVT_PK_NAME " %0, %1, %2" : "=v"(pk) : "v"(p0), "v"(p1));
        pf[i >> 3][(i & 7) >> 1] = pk;
      }
      if (__ballot(m_new != m_run) != 0ull) {          // wave-uniform: some query's maximum moved
        const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
          for (int i = 0; i < 16; ++i) o[d][i] *= alpha;
      }
      l_run += psum;
      m_run = m_new;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * b + tt;
        auto pv_pair = [&](auto dc) {
          constexpr int d = decltype(dc)::value;
          u32x2_t a0, b0, a1, b1;
          if (t == 0)      tr_read_2pairs<0 * 16 * VROW + d * 64, (0 * 16 + 8) * VROW + d * 64, 0 * 16 * VROW + (d + 1) * 64, (0 * 16 + 8) * VROW + (d + 1) * 64>(a0, b0, a1, b1, vaddr);
          else if (t == 1) tr_read_2pairs<1 * 16 * VROW + d * 64, (1 * 16 + 8) * VROW + d * 64, 1 * 16 * VROW + (d + 1) * 64, (1 * 16 + 8) * VROW + (d + 1) * 64>(a0, b0, a1, b1, vaddr);
          else if (t == 2) tr_read_2pairs<2 * 16 * VROW + d * 64, (2 * 16 + 8) * VROW + d * 64, 2 * 16 * VROW + (d + 1) * 64, (2 * 16 + 8) * VROW + (d + 1) * 64>(a0, b0, a1, b1, vaddr);
          else             tr_read_2pairs<3 * 16 * VROW + d * 64, (3 * 16 + 8) * VROW + d * 64, 3 * 16 * VROW + (d + 1) * 64, (3 * 16 + 8) * VROW + (d + 1) * 64>(a0, b0, a1, b1, vaddr);
          u32x4_t v0 = {a0[0], a0[1], b0[0], b0[1]}, v1 = {a1[0], a1[1], b1[0], b1[1]};
          o[d] = ATS_MFMA_32x32x16(__builtin_bit_cast(bf16x8_t, v0), __builtin_bit_cast(bf16x8_t, pf[tt]), o[d]);
          o[d + 1] = ATS_MFMA_32x32x16(__builtin_bit_cast(bf16x8_t, v1), __builtin_bit_cast(bf16x8_t, pf[tt]), o[d + 1]);
        };
        pv_pair(std::integral_constant<int, 0>{});
        if constexpr (DB == 4) pv_pair(std::integral_constant<int, 2>{});
      }
    }
  }
  l_run += __shfl_xor(l_run, 32, 64);
  if (qok) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        ushort4 v4;
        v4.x = f2bf(o[d][i4 * 4 + 0] * inv); v4.y = f2bf(o[d][i4 * 4 + 1] * inv);
        v4.z = f2bf(o[d][i4 * 4 + 2] * inv); v4.w = f2bf(o[d][i4 * 4 + 3] * inv);
        *reinterpret_cast<ushort4*>(out + ats_opnd_idx<2>(pk, qrow, h * DH + d * 32 + 8 * i4 + 4 * hi, ldo)) = v4;
      }
  }
}

}  // namespace

int ats_tree_attention_segs(const void* q, int ldq, const SegTable& t, const SegTable* dt, size_t layer_off_bytes, int vis_words,
                            void* out, int ldo, int n_heads, int head_dim, int dtype, hipStream_t st, int rows_per_wave, int pk) {
  if (t.total_tok <= 0) return ATSPEED_OK;
  ATS_REQUIRE(!pk || (dtype == ATS_HALF && ldo % 32 == 0), ATSPEED_ERR_INVALID, "attention: packed output needs bf16 and ldo %% 32 == 0");
  ATS_REQUIRE(head_dim % 8 == 0 && head_dim <= 256, ATSPEED_ERR_INVALID, "attention: head_dim %d unsupported", head_dim);
  ATS_REQUIRE(vis_words * 64 <= kMaxSlots, ATSPEED_ERR_CAPACITY, "attention: visibility bitset too wide (%d words)", vis_words);
  for (int i = 0; i < t.n; ++i)
    ATS_REQUIRE(t.seg[i].n_slots <= vis_words * 64, ATSPEED_ERR_CAPACITY, "attention: %d slots exceed the visibility bitset", t.seg[i].n_slots);
  float scale = 1.0f / sqrtf((float)head_dim);
  if (dtype == ATS_HALF && (head_dim == 64 || head_dim == 128) && (ldq % 8) == 0 && (ldo % 4) == 0) {
    dim3 mgrid(t.n_qtiles * n_heads);
    ATS_REQUIRE(t.qtile_rows == 64 || t.qtile_rows == 128 || t.qtile_rows == 256, ATSPEED_ERR_INVALID, "attention: query tile of %d rows", t.qtile_rows);
    static const int rows32 = getenv("ATSPEED_ATTN32") ? atoi(getenv("ATSPEED_ATTN32")) : 1;
    // small grids (one user: 32-64 workgroups) are latency-bound per workgroup and keep the 16-rows-per-wave kernel (twice the waves per tile)
    constexpr int rows32_min_wgs = 512;       // (round 6: a constant -- no test or tool set ATSPEED_ATTN32_MIN_WGS; measured in round 2)
    if (rows_per_wave == 32 || (rows_per_wave == 0 && rows32 && t.n_qtiles * n_heads >= rows32_min_wgs)) {
#define ATS_ATTN32(DHV, NWV)                                                                                                   \
  {                                                                                                                            \
    constexpr int lds_bytes = 2 * (64 * DHV * 2 + 64 * (DHV * 2 + 32));                                                        \
    static thread_local AtsPerDeviceFlag attr_flag;                                                                            \
    bool& attr_done = attr_flag.cur();                                                                                         \
    if (!attr_done) {                                                                                                          \
      ATS_HIP(hipFuncSetAttribute((const void*)tree_attn32_kernel<DHV, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)); \
      attr_done = true;                                                                                                        \
    }                                                                                                                          \
    tree_attn32_kernel<DHV, NWV><<<mgrid, 64 * NWV, lds_bytes, st>>>((const bf16_t*)q, ldq, dt, layer_off_bytes, vis_words,    \
                                                                     (bf16_t*)out, ldo, pk, n_heads, scale);                       \
  }
      if (head_dim == 128) { if (t.qtile_rows == 256) ATS_ATTN32(128, 8) else if (t.qtile_rows == 128) ATS_ATTN32(128, 4) else ATS_ATTN32(128, 2) }
      else                 { if (t.qtile_rows == 256) ATS_ATTN32(64, 8)  else if (t.qtile_rows == 128) ATS_ATTN32(64, 4)  else ATS_ATTN32(64, 2) }
#undef ATS_ATTN32
      ATS_LAUNCH_CHECK();
      return ATSPEED_OK;
    }
    // one user's forwards (at most one workgroup per CU): the DMA-ring form, the whole K/V of a user in flight before the first product
    static const int ring_on = getenv("ATSPEED_ATTN_RING") ? atoi(getenv("ATSPEED_ATTN_RING")) : 1;
    constexpr int ring_max_wgs = 256;         // (round 6: a constant -- one user's forwards; ATSPEED_ATTN_RING_MAX_WGS was set by nothing)
    {
      const int nw = t.qtile_rows / 16;
      const size_t tile = (size_t)64 * head_dim * 2 + (size_t)64 * (head_dim * 2 + 32);
      const size_t ring_lds = 4 * tile + (size_t)16 * nw * vis_words * sizeof(uint64_t);
      // (not the 256-row tile: 16 waves cap a lane at 128 registers and the compiler's spill traffic would sit in the hand-counted vmcnt window)
      if (ring_on && nw <= 8 && (int)(t.n_qtiles * n_heads) <= ring_max_wgs && ring_lds <= 160 * 1024 - 64) {
#define ATS_ATTN_RING(DHV, NWV)                                                                                                \
  {                                                                                                                            \
    static thread_local AtsPerDeviceFlag attr_flag;                                                                            \
    bool& attr_done = attr_flag.cur();                                                                                         \
    if (!attr_done) {                                                                                                          \
      ATS_HIP(hipFuncSetAttribute((const void*)tree_attn_mfma_kernel<DHV, NWV, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64)); \
      attr_done = true;                                                                                                        \
    }                                                                                                                          \
    tree_attn_mfma_kernel<DHV, NWV, 4><<<mgrid, 64 * NWV, ring_lds, st>>>((const bf16_t*)q, ldq, dt, layer_off_bytes, vis_words, \
                                                                          (bf16_t*)out, ldo, pk, n_heads, scale);                  \
  }
        if (head_dim == 128) { if (nw == 8) ATS_ATTN_RING(128, 8) else ATS_ATTN_RING(128, 4) }
        else                 { if (nw == 8) ATS_ATTN_RING(64, 8)  else ATS_ATTN_RING(64, 4) }
#undef ATS_ATTN_RING
        ATS_LAUNCH_CHECK();
        return ATSPEED_OK;
      }
    }
    if (t.qtile_rows == 256) {
      if (head_dim == 128)
        tree_attn_mfma_kernel<128, 16><<<mgrid, 1024, 0, st>>>((const bf16_t*)q, ldq, dt, layer_off_bytes, vis_words, (bf16_t*)out, ldo, pk, n_heads, scale);
      else
        tree_attn_mfma_kernel<64, 16><<<mgrid, 1024, 0, st>>>((const bf16_t*)q, ldq, dt, layer_off_bytes, vis_words, (bf16_t*)out, ldo, pk, n_heads, scale);
    } else if (t.qtile_rows == 128) {
      if (head_dim == 128)
        tree_attn_mfma_kernel<128, 8><<<mgrid, 512, 0, st>>>((const bf16_t*)q, ldq, dt, layer_off_bytes, vis_words, (bf16_t*)out, ldo, pk, n_heads, scale);
      else
        tree_attn_mfma_kernel<64, 8><<<mgrid, 512, 0, st>>>((const bf16_t*)q, ldq, dt, layer_off_bytes, vis_words, (bf16_t*)out, ldo, pk, n_heads, scale);
    } else {
      if (head_dim == 128)
        tree_attn_mfma_kernel<128, 4><<<mgrid, 256, 0, st>>>((const bf16_t*)q, ldq, dt, layer_off_bytes, vis_words, (bf16_t*)out, ldo, pk, n_heads, scale);
      else
        tree_attn_mfma_kernel<64, 4><<<mgrid, 256, 0, st>>>((const bf16_t*)q, ldq, dt, layer_off_bytes, vis_words, (bf16_t*)out, ldo, pk, n_heads, scale);
    }
    ATS_LAUNCH_CHECK();
    return ATSPEED_OK;
  }
  dim3 grid(t.total_tok, (n_heads + 3) / 4);
  size_t lds = (size_t)4 * (head_dim + 2 * vis_words * 64) * sizeof(float);
  if (dtype == ATSPEED_F32)
    tree_attn_kernel<float><<<grid, 256, lds, st>>>((const float*)q, ldq, dt, layer_off_bytes, vis_words, (float*)out, ldo, 0, n_heads, head_dim, scale);
  else
    tree_attn_kernel<bf16_t><<<grid, 256, lds, st>>>((const bf16_t*)q, ldq, dt, layer_off_bytes, vis_words, (bf16_t*)out, ldo, pk, n_heads, head_dim, scale);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

int ats_tree_attention(const void* q, int ldq, const void* kcache, const void* vcache, const uint64_t* vis,
                       int vis_words, void* out, int ldo, int n_tokens, int n_slots, int n_heads, int head_dim,
                       int dtype, hipStream_t st, int qtile_rows, int rows_per_wave) {
  SegTable t{};
  t.n = 1; t.total_tok = n_tokens; t.total_logit = 0;
  t.seg[0].vis = vis; t.seg[0].kc = const_cast<void*>(kcache); t.seg[0].vc = const_cast<void*>(vcache);
  t.seg[0].row0 = 0; t.seg[0].n_tok = n_tokens; t.seg[0].n_slots = n_slots;
  ATS_REQUIRE(n_slots <= vis_words * 64, ATSPEED_ERR_CAPACITY, "attention: %d slots exceed the visibility bitset (%d words)", n_slots, vis_words);
  ATS_REQUIRE(qtile_rows == 0 || qtile_rows == 64 || qtile_rows == 128 || qtile_rows == 256, ATSPEED_ERR_INVALID, "attention: query tile of %d rows", qtile_rows);
  ATS_REQUIRE(rows_per_wave == 0 || rows_per_wave == 16 || rows_per_wave == 32, ATSPEED_ERR_INVALID, "attention: %d rows per wave", rows_per_wave);
  t.n_qtiles = 0; t.qtile_rows = qtile_rows ? qtile_rows : (n_tokens > 160 ? 256 : (n_tokens > 96 ? 128 : 64));
  ATS_REQUIRE((n_tokens + t.qtile_rows - 1) / t.qtile_rows <= ATS_MAX_QTILES && (n_tokens + t.qtile_rows - 1) / t.qtile_rows <= 255, ATSPEED_ERR_CAPACITY,
              "attention: too many query rows");
  for (int j = 0; j * t.qtile_rows < n_tokens; ++j) { t.qtile_seg[t.n_qtiles] = 0; t.qtile_idx[t.n_qtiles++] = (unsigned char)j; }
  const void* dt = nullptr;
  ATS_TRY(ats_stage(&t, sizeof(t), &dt, st));
  return ats_tree_attention_segs(q, ldq, t, (const SegTable*)dt, 0, vis_words, out, ldo, n_heads, head_dim, dtype, st, rows_per_wave, 0);
}

}  // namespace ATS_NS

#ifndef ATS_F16_FLAVOUR          // the C ABI exists once; it picks the flavour by the dtype code
extern "C" int atspeed_tree_attention(const void* q, int32_t ldq, const void* kcache, const void* vcache,
                                      const uint64_t* vis, int32_t vis_words, void* out, int32_t n_tokens,
                                      int32_t n_slots, int32_t n_heads, int32_t head_dim, int32_t dtype, void* stream) {
  ATS_REQUIRE(q && kcache && vcache && vis && out, ATSPEED_ERR_INVALID, "attention: null argument");
  return ATS_KD(dtype, ats_tree_attention(q, ldq, kcache, vcache, vis, vis_words, out, n_heads * head_dim, n_tokens, n_slots, n_heads,
                            head_dim, dtype, (hipStream_t)stream, 0, 0));
}

extern "C" int atspeed_tree_attention_tiled(const void* q, int32_t ldq, const void* kcache, const void* vcache,
                                            const uint64_t* vis, int32_t vis_words, void* out, int32_t n_tokens,
                                            int32_t n_slots, int32_t n_heads, int32_t head_dim, int32_t dtype, int32_t qtile_rows,
                                            int32_t rows_per_wave, void* stream) {
  ATS_REQUIRE(q && kcache && vcache && vis && out, ATSPEED_ERR_INVALID, "attention: null argument");
  return ATS_KD(dtype, ats_tree_attention(q, ldq, kcache, vcache, vis, vis_words, out, n_heads * head_dim, n_tokens, n_slots, n_heads,
                            head_dim, dtype, (hipStream_t)stream, qtile_rows, rows_per_wave));
}
#endif
