// Streaming kernels of the Llama forward: synthetic fill, embedding gather, RMSNorm,
// RoPE + KV scatter.  All HBM-bound: 16-byte coalesced accesses, wave64 reductions.
#include "internal.h"

namespace ATS_NS {

// ---------------------------------------------------------------------------- fill
__device__ __forceinline__ uint32_t hash_u32(uint32_t idx, uint32_t seed) { return ats_hash_u32(idx, seed); }

template <typename T>
__global__ void fill_hash_normal_kernel(T* dst, size_t n, uint32_t seed, float scale, float add, uint64_t offset) {
#pragma clang fp contract(off)   // the multiply and the add must round separately (numpy does)
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t idx = (uint32_t)(i + offset);
    uint32_t h1 = hash_u32(idx, seed), h2 = hash_u32(idx, seed ^ 0x5BD1E995u);
    int s = (int)((h1 & 0xffffu) + (h1 >> 16) + (h2 & 0xffffu) + (h2 >> 16)) - 131070;
    float v = (float)s * scale;          // exact int->float, ONE fp32 multiply: matches numpy bit for bit
    if (add != 0.0f) v = add + v;
    Elt<T>::store(dst + i, v);
  }
}


// ---------------------------------------------------------------------------- embed
// one 16-byte chunk per thread; a token row is hidden*sizeof(T) contiguous bytes
__global__ void embed_kernel(const uint4* __restrict__ table, const int32_t* __restrict__ ids, uint4* __restrict__ out,
                             int n_tokens, int chunks_per_row, int vocab) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int total = n_tokens * chunks_per_row;
  if (i >= total) return;
  int t = i / chunks_per_row, c = i - t * chunks_per_row;
  int id = ids[t];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  out[i] = table[(size_t)id * chunks_per_row + c];
}

int ats_embed(const void* table, const int32_t* ids, void* out, int n_tokens, int hidden, int vocab, int dtype,
              hipStream_t st) {
  int esz = dtype == ATSPEED_F32 ? 4 : 2;
  int cpr = hidden * esz / 16;
  int total = n_tokens * cpr;
  embed_kernel<<<(total + 255) / 256, 256, 0, st>>>((const uint4*)table, ids, (uint4*)out, n_tokens, cpr, vocab);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

// ---------------------------------------------------------------------------- rmsnorm
// one workgroup per row; fp32 statistics (HF LlamaRMSNorm upcasts); y = w * (x * rsqrt(mean(x^2)+eps))
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const T* __restrict__ x, const T* __restrict__ w, T* __restrict__ y,
                                                      int hidden, float eps, int pk) {
  __shared__ float red[4];
  const T* xr = x + (size_t)blockIdx.x * hidden;
  float ss = 0.f;
  for (int i = threadIdx.x; i < hidden; i += 256) { float v = Elt<T>::load(xr + i); ss += v * v; }
  ss = wave_sum_f32(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  float tot = red[0] + red[1] + red[2] + red[3];
  float rs = rsqrtf(tot / (float)hidden + eps);
  for (int i = threadIdx.x; i < hidden; i += 256) {
    float v = Elt<T>::load(xr + i) * rs;
    if constexpr (sizeof(T) == 2) v = bf2f(f2bf(v));    // HF casts the normalised value to the input dtype first
    Elt<T>::store(y + ats_opnd_idx<sizeof(T)>(pk, blockIdx.x, i, hidden), Elt<T>::load(w + i) * v);   // y: a GEMM operand (packed when pk)
  }
}

// bf16 rows of up to 8192 elements: 16-byte loads, the row stays in registers between the two passes (read once, written once).
// QUANT: also emit the row as OCP e4m3 with its scale (= what quant_rows_fp8_kernel makes of the bf16 output, bit for bit), so the
// W8A8 projection that consumes the norm needs no quantisation pass of its own; y may then be null.
template <int NC, bool QUANT>
__global__ __launch_bounds__(256) void rmsnorm_bf16_vec_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                               bf16_t* __restrict__ y, int hidden, float eps,
                                                               unsigned char* __restrict__ q, float* __restrict__ scale, int pk) {
  __shared__ float red[4], red2[4];
  const int nchunk = hidden >> 3;
  // packed output: rows 2i and 2i+1 share every 128-byte line (64 bytes each).  Workgroup L runs on XCD L % 8, so consecutive workgroups
  // would leave the two halves of a line dirty in two different L2s; inside a group of 16 workgroups the pair goes to the SAME XCD
  // (workgroups x and x + 8 take rows 2x and 2x + 1) and its lines leave one L2 whole
  int row = blockIdx.x;
  if (pk && (row | 15) < (int)gridDim.x && !(pk & 2)) { const int b = row & 15; row = (row & ~15) + (b & 7) * 2 + (b >> 3); }
  const uint4* xr = reinterpret_cast<const uint4*>(x + (size_t)row * hidden);
  const uint4* wr = reinterpret_cast<const uint4*>(w);
  uint4 v[NC];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = threadIdx.x + c * 256;
    v[c] = i < nchunk ? xr[i] : make_uint4(0, 0, 0, 0);
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&v[c]);
#pragma unroll
    for (int j = 0; j < 8; ++j) { float f = bf2f(e[j]); ss += f * f; }
  }
  ss = wave_sum_f32(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float rs = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)hidden + eps);
  float amax = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = threadIdx.x + c * 256;
    if (i < nchunk) {
      const uint4 wv = wr[i];
      const bf16_t* e = reinterpret_cast<const bf16_t*>(&v[c]);
      const bf16_t* we = reinterpret_cast<const bf16_t*>(&wv);
      uint4 o;
      bf16_t* oe = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        oe[j] = f2bf(bf2f(we[j]) * bf2f(f2bf(bf2f(e[j]) * rs)));   // HF casts the normalised value first
        if constexpr (QUANT) amax = fmaxf(amax, fabsf(bf2f(oe[j])));
      }
      if (y) *reinterpret_cast<uint4*>(y + ats_opnd_idx<2>(pk & 1, row, (size_t)i * 8, hidden)) = o;   // outputs are GEMM operands: packed when pk
      v[c] = o;
    }
  }
  if constexpr (QUANT) {
    amax = wave_max_f32(amax);
    if ((threadIdx.x & 63) == 0) red2[threadIdx.x >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red2[0], red2[1]), fmaxf(red2[2], red2[3]));
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (threadIdx.x == 0) scale[row] = sc;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int i = threadIdx.x + c * 256;
      if (i < nchunk) {
        const bf16_t* e = reinterpret_cast<const bf16_t*>(&v[c]);
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = fminf(fmaxf(bf2f(e[j]) * inv, -448.f), 448.f);
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
        *reinterpret_cast<uint2*>(q + ats_opnd_idx<1>(pk & 1, row, (size_t)i * 8, hidden)) = make_uint2((unsigned)lo, (unsigned)hi);
      }
    }
  }
}

// kernel argument `pk` of rmsnorm_bf16_vec_kernel: bit 0 = packed output, bit 1 = keep the workgroup -> row map linear (A/B switch
// ATSPEED_RMSNORM_PAIRS=0)
static int rmsnorm_pk_arg(int pk) {
  static const bool pairs_off = getenv("ATSPEED_RMSNORM_PAIRS") && atoi(getenv("ATSPEED_RMSNORM_PAIRS")) == 0;
  return pk ? (pairs_off ? 3 : 1) : 0;
}

// RMSNorm whose consumer is a W8A8 projection: y (bf16, optional) and the e4m3 row + scale in one pass
int ats_rmsnorm_quant_fp8(const void* x, const void* w, void* y, void* q, float* scale, int rows, int hidden, float eps, hipStream_t st, int pk) {
  if (rows <= 0) return ATSPEED_OK;
  ATS_REQUIRE(hidden % 8 == 0 && hidden <= 8192 && (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)q) & 15) == 0,
              ATSPEED_ERR_INVALID, "rmsnorm_quant: hidden %d unsupported", hidden);
  const bf16_t *xb = (const bf16_t*)x, *wb = (const bf16_t*)w;
  bf16_t* yb = (bf16_t*)y;
  unsigned char* qb = (unsigned char*)q;
  if (hidden <= 2048)      rmsnorm_bf16_vec_kernel<1, true><<<rows, 256, 0, st>>>(xb, wb, yb, hidden, eps, qb, scale, rmsnorm_pk_arg(pk));
  else if (hidden <= 4096) rmsnorm_bf16_vec_kernel<2, true><<<rows, 256, 0, st>>>(xb, wb, yb, hidden, eps, qb, scale, rmsnorm_pk_arg(pk));
  else                     rmsnorm_bf16_vec_kernel<4, true><<<rows, 256, 0, st>>>(xb, wb, yb, hidden, eps, qb, scale, rmsnorm_pk_arg(pk));
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

int ats_rmsnorm(const void* x, const void* w, void* y, int rows, int hidden, float eps, int dtype, hipStream_t st, int pk) {
  if (rows <= 0) return ATSPEED_OK;
  ATS_REQUIRE(!pk || (dtype == ATS_HALF && hidden % 32 == 0), ATSPEED_ERR_INVALID, "rmsnorm: packed output needs bf16 and hidden %% 32 == 0");
  const bool vec_ok = dtype == ATS_HALF && hidden % 8 == 0 && hidden <= 8192 &&
                      (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0;
  if (vec_ok) {
    const bf16_t *xb = (const bf16_t*)x, *wb = (const bf16_t*)w;
    bf16_t* yb = (bf16_t*)y;
    if (hidden <= 2048)      rmsnorm_bf16_vec_kernel<1, false><<<rows, 256, 0, st>>>(xb, wb, yb, hidden, eps, nullptr, nullptr, rmsnorm_pk_arg(pk));
    else if (hidden <= 4096) rmsnorm_bf16_vec_kernel<2, false><<<rows, 256, 0, st>>>(xb, wb, yb, hidden, eps, nullptr, nullptr, rmsnorm_pk_arg(pk));
    else                     rmsnorm_bf16_vec_kernel<4, false><<<rows, 256, 0, st>>>(xb, wb, yb, hidden, eps, nullptr, nullptr, rmsnorm_pk_arg(pk));
    ATS_LAUNCH_CHECK();
    return ATSPEED_OK;
  }
  if (dtype == ATSPEED_F32)
    rmsnorm_kernel<float><<<rows, 256, 0, st>>>((const float*)x, (const float*)w, (float*)y, hidden, eps, 0);
  else
    rmsnorm_kernel<bf16_t><<<rows, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, hidden, eps, pk);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}



// ---------------------------------------------------------------------------- rope + kv scatter
// rotate-half convention of HF Llama: pairs (i, i + dh/2); cos/sin tables [max_pos][dh/2] fp32.
// thread = (token, head, pair i): rotates q in place, writes rotated k and v to the cache slot.
template <typename T>
__global__ void rope_kv_kernel(T* __restrict__ qkv, const int32_t* __restrict__ pos, const int32_t* __restrict__ slots,
                               const float* __restrict__ cos_tab, const float* __restrict__ sin_tab,
                               T* __restrict__ kcache, T* __restrict__ vcache, int n_tokens, int n_heads, int head_dim,
                               int max_pos) {
  int half = head_dim >> 1;
  int hidden = n_heads * head_dim;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int total = n_tokens * n_heads * half;
  if (i >= total) return;
  int p = i % half;
  int h = (i / half) % n_heads;
  int t = i / (half * n_heads);
  int ps = pos[t];
  ps = ps < 0 ? 0 : (ps >= max_pos ? max_pos - 1 : ps);
  float c = cos_tab[(size_t)ps * half + p], s = sin_tab[(size_t)ps * half + p];
  T* row = qkv + (size_t)t * 3 * hidden;
  int d0 = h * head_dim + p, d1 = d0 + half;
  float q0 = Elt<T>::load(row + d0), q1 = Elt<T>::load(row + d1);
  Elt<T>::store(row + d0, q0 * c - q1 * s);
  Elt<T>::store(row + d1, q1 * c + q0 * s);
  float k0 = Elt<T>::load(row + hidden + d0), k1 = Elt<T>::load(row + hidden + d1);
  size_t co = (size_t)slots[t] * hidden;
  Elt<T>::store(kcache + co + d0, k0 * c - k1 * s);
  Elt<T>::store(kcache + co + d1, k1 * c + k0 * s);
  vcache[co + d0] = row[2 * hidden + d0];
  vcache[co + d1] = row[2 * hidden + d1];
}

int ats_rope_kv(void* qkv, const int32_t* pos, const int32_t* slots, const float* cos_tab, const float* sin_tab,
                void* kcache, void* vcache, int n_tokens, int n_heads, int head_dim, int max_pos, int dtype,
                hipStream_t st) {
  int total = n_tokens * n_heads * (head_dim / 2);
  if (total <= 0) return ATSPEED_OK;
  if (dtype == ATSPEED_F32)
    rope_kv_kernel<float><<<(total + 255) / 256, 256, 0, st>>>((float*)qkv, pos, slots, cos_tab, sin_tab, (float*)kcache,
                                                                (float*)vcache, n_tokens, n_heads, head_dim, max_pos);
  else
    rope_kv_kernel<bf16_t><<<(total + 255) / 256, 256, 0, st>>>((bf16_t*)qkv, pos, slots, cos_tab, sin_tab,
                                                                 (bf16_t*)kcache, (bf16_t*)vcache, n_tokens, n_heads,
                                                                 head_dim, max_pos);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

// ---------------------------------------------------------------------------- multi-user (segment) variants
__device__ __forceinline__ int seg_of_row(const SegTable* t, int row) {
  int lo = 0, hi = t->n;                           // last segment with row0 <= row
  while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (t->seg[mid].row0 <= row) lo = mid; else hi = mid; }
  return lo;
}

// Per batched row: the caches of the row's user, its cache slot and its (clamped) rotation index -- resolved once per forward so that
// the qkv projection's epilogue (gemm.hip: EPI_QKV_ROPE) finds them with one 24-byte load instead of a segment search per layer.
__global__ void row_info_kernel(const SegTable* __restrict__ t, RowInfo* __restrict__ out, int max_pos) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= t->total_tok) return;
  const Seg& sg = t->seg[seg_of_row(t, row)];
  const int lt = row - sg.row0;
  int ps = sg.pos[lt];
  ps = ps < 0 ? 0 : (ps >= max_pos ? max_pos - 1 : ps);
  out[row] = RowInfo{sg.kc, sg.vc, ps, sg.slot[lt]};
}

int ats_row_info(const SegTable& t, const SegTable* dt, RowInfo* out, int max_pos, hipStream_t st) {
  if (t.total_tok <= 0) return ATSPEED_OK;
  ATS_REQUIRE(dt && out, ATSPEED_ERR_INVALID, "row_info: null argument");
  row_info_kernel<<<(t.total_tok + 255) / 256, 256, 0, st>>>(dt, out, max_pos);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

__global__ void embed_segs_kernel(const uint4* __restrict__ table, const SegTable* __restrict__ t, uint4* __restrict__ out,
                                  int chunks_per_row, int vocab) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= t->total_tok * chunks_per_row) return;
  int row = i / chunks_per_row, c = i - row * chunks_per_row;
  const Seg& sg = t->seg[seg_of_row(t, row)];
  int id = sg.ids[row - sg.row0];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  out[i] = table[(size_t)id * chunks_per_row + c];
}

int ats_embed_segs(const void* table, const SegTable& t, const SegTable* dt, void* out, int hidden, int vocab, int dtype, hipStream_t st) {
  int cpr = hidden * (dtype == ATSPEED_F32 ? 4 : 2) / 16;
  int total = t.total_tok * cpr;
  if (total <= 0) return ATSPEED_OK;
  embed_segs_kernel<<<(total + 255) / 256, 256, 0, st>>>((const uint4*)table, dt, (uint4*)out, cpr, vocab);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

template <typename T>
__global__ void rope_kv_segs_kernel(T* __restrict__ qkv, const SegTable* __restrict__ t, const float* __restrict__ cos_tab,
                                    const float* __restrict__ sin_tab, size_t layer_off, int n_heads, int head_dim, int max_pos) {
  int half = head_dim >> 1;
  int hidden = n_heads * head_dim;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= t->total_tok * n_heads * half) return;
  int p = i % half;
  int h = (i / half) % n_heads;
  int row = i / (half * n_heads);
  const Seg& sg = t->seg[seg_of_row(t, row)];
  int lt = row - sg.row0;
  int ps = sg.pos[lt];
  ps = ps < 0 ? 0 : (ps >= max_pos ? max_pos - 1 : ps);
  float c = cos_tab[(size_t)ps * half + p], s = sin_tab[(size_t)ps * half + p];
  T* r = qkv + (size_t)row * 3 * hidden;
  int d0 = h * head_dim + p, d1 = d0 + half;
  float q0 = Elt<T>::load(r + d0), q1 = Elt<T>::load(r + d1);
  Elt<T>::store(r + d0, q0 * c - q1 * s);
  Elt<T>::store(r + d1, q1 * c + q0 * s);
  float k0 = Elt<T>::load(r + hidden + d0), k1 = Elt<T>::load(r + hidden + d1);
  T* kc = reinterpret_cast<T*>(reinterpret_cast<char*>(sg.kc) + layer_off) + (size_t)sg.slot[lt] * hidden;
  T* vc = reinterpret_cast<T*>(reinterpret_cast<char*>(sg.vc) + layer_off) + (size_t)sg.slot[lt] * hidden;
  Elt<T>::store(kc + d0, k0 * c - k1 * s);
  Elt<T>::store(kc + d1, k1 * c + k0 * s);
  vc[d0] = r[2 * hidden + d0];
  vc[d1] = r[2 * hidden + d1];
}

// bf16, head_dim % 16 == 0: a thread rotates 8 consecutive (i, i + dh/2) pairs of q and k with 16-byte accesses
__global__ void rope_kv_segs_vec_kernel(bf16_t* __restrict__ qkv, const SegTable* __restrict__ t, const float* __restrict__ cos_tab,
                                        const float* __restrict__ sin_tab, size_t layer_off, int n_heads, int head_dim, int max_pos) {
  const int half = head_dim >> 1, groups = half >> 3;
  const int hidden = n_heads * head_dim;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= t->total_tok * n_heads * groups) return;
  const int gi = i % groups;
  const int h = (i / groups) % n_heads;
  const int row = i / (groups * n_heads);
  const Seg& sg = t->seg[seg_of_row(t, row)];
  const int lt = row - sg.row0;
  int ps = sg.pos[lt];
  ps = ps < 0 ? 0 : (ps >= max_pos ? max_pos - 1 : ps);
  const float* cp = cos_tab + (size_t)ps * half + gi * 8;
  const float* sp = sin_tab + (size_t)ps * half + gi * 8;
  bf16_t* r = qkv + (size_t)row * 3 * hidden;
  const int d0 = h * head_dim + gi * 8, d1 = d0 + half;
  const size_t co = (size_t)sg.slot[lt] * hidden;
  bf16_t* kc = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(sg.kc) + layer_off) + co;
  bf16_t* vc = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(sg.vc) + layer_off) + co;
  uint4 q0v = *reinterpret_cast<const uint4*>(r + d0), q1v = *reinterpret_cast<const uint4*>(r + d1);
  uint4 k0v = *reinterpret_cast<const uint4*>(r + hidden + d0), k1v = *reinterpret_cast<const uint4*>(r + hidden + d1);
  uint4 qo0, qo1, ko0, ko1;
  const uint32_t *q0 = (const uint32_t*)&q0v, *q1 = (const uint32_t*)&q1v, *k0 = (const uint32_t*)&k0v, *k1 = (const uint32_t*)&k1v;
  uint32_t *a0 = (uint32_t*)&qo0, *a1 = (uint32_t*)&qo1, *b0 = (uint32_t*)&ko0, *b1 = (uint32_t*)&ko1;
#pragma unroll
  for (int e = 0; e < 4; ++e) {                                   // two elements per packed register
    const float ca = cp[2 * e], sa = sp[2 * e], cb = cp[2 * e + 1], sb = sp[2 * e + 1];
    a0[e] = f2bf_pk(rope_first(bf_lo(q0[e]), bf_lo(q1[e]), ca, sa), rope_first(bf_hi(q0[e]), bf_hi(q1[e]), cb, sb));
    a1[e] = f2bf_pk(rope_second(bf_lo(q0[e]), bf_lo(q1[e]), ca, sa), rope_second(bf_hi(q0[e]), bf_hi(q1[e]), cb, sb));
    b0[e] = f2bf_pk(rope_first(bf_lo(k0[e]), bf_lo(k1[e]), ca, sa), rope_first(bf_hi(k0[e]), bf_hi(k1[e]), cb, sb));
    b1[e] = f2bf_pk(rope_second(bf_lo(k0[e]), bf_lo(k1[e]), ca, sa), rope_second(bf_hi(k0[e]), bf_hi(k1[e]), cb, sb));
  }
  *reinterpret_cast<uint4*>(r + d0) = qo0; *reinterpret_cast<uint4*>(r + d1) = qo1;
  *reinterpret_cast<uint4*>(kc + d0) = ko0; *reinterpret_cast<uint4*>(kc + d1) = ko1;
  *reinterpret_cast<uint4*>(vc + d0) = *reinterpret_cast<const uint4*>(r + 2 * hidden + d0);
  *reinterpret_cast<uint4*>(vc + d1) = *reinterpret_cast<const uint4*>(r + 2 * hidden + d1);
}

// The same on the qkv projection's split-K slabs (one user's forward): sums the fp32 slabs in slab order, rounds to bf16 exactly as the
// separate reduce pass stored them, rotates, writes q to the qkv buffer and k / v straight to the caches.
__global__ void rope_kv_segs_slab_kernel(const float* __restrict__ slabs, int splits, bf16_t* __restrict__ qkv, const SegTable* __restrict__ t,
                                         const float* __restrict__ cos_tab, const float* __restrict__ sin_tab, size_t layer_off, int n_heads,
                                         int head_dim, int max_pos) {
  // a thread owns 8 consecutive (i, i + dh/2) pairs of ONE of q / k / v (part): 4 float4 loads per slab, four slabs' loads in flight
  const int half = head_dim >> 1, groups = half >> 3;
  const int hidden = n_heads * head_dim;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= t->total_tok * 3 * n_heads * groups) return;
  const int gi = i % groups;
  const int h = (i / groups) % n_heads;
  const int part = (i / (groups * n_heads)) % 3;
  const int row = i / (groups * n_heads * 3);
  const Seg& sg = t->seg[seg_of_row(t, row)];
  const int lt = row - sg.row0;
  const int d0 = h * head_dim + gi * 8, d1 = d0 + half;
  const size_t slab = (size_t)t->total_tok * 3 * hidden;
  const float* p0 = slabs + (size_t)row * 3 * hidden + part * hidden + d0;
  const float* p1 = p0 + half;
  float lo[8], hi[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) lo[e] = hi[e] = 0.f;
  auto add = [](float (&acc)[8], const float4& a, const float4& b) {
    acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w; acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
  };
  int z = 0;
  for (; z + 4 <= splits; z += 4) {
    float4 v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t o = (size_t)(z + u) * slab;
      v[u][0] = *reinterpret_cast<const float4*>(p0 + o); v[u][1] = *reinterpret_cast<const float4*>(p0 + o + 4);
      v[u][2] = *reinterpret_cast<const float4*>(p1 + o); v[u][3] = *reinterpret_cast<const float4*>(p1 + o + 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { add(lo, v[u][0], v[u][1]); add(hi, v[u][2], v[u][3]); }
  }
  for (; z < splits; ++z) {
    const size_t o = (size_t)z * slab;
    const float4 a = *reinterpret_cast<const float4*>(p0 + o), b = *reinterpret_cast<const float4*>(p0 + o + 4);
    const float4 c = *reinterpret_cast<const float4*>(p1 + o), d = *reinterpret_cast<const float4*>(p1 + o + 4);
    add(lo, a, b); add(hi, c, d);
  }
  uint4 o0, o1;
  uint32_t *w0 = (uint32_t*)&o0, *w1 = (uint32_t*)&o1;
  if (part == 2) {                                   // v: the projection's bf16 output as it is
#pragma unroll
    for (int e = 0; e < 8; e += 2) { w0[e >> 1] = f2bf_pk(lo[e], lo[e + 1]); w1[e >> 1] = f2bf_pk(hi[e], hi[e + 1]); }
  } else {
    int ps = sg.pos[lt];
    ps = ps < 0 ? 0 : (ps >= max_pos ? max_pos - 1 : ps);
    const float* cp = cos_tab + (size_t)ps * half + gi * 8;
    const float* sp = sin_tab + (size_t)ps * half + gi * 8;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      // the projection's bf16 outputs first (what the reduce pass stored), then the rotation on those
      const uint32_t x0 = f2bf_pk(lo[e], lo[e + 1]), x1 = f2bf_pk(hi[e], hi[e + 1]);
      const float ca = cp[e], sa = sp[e], cb = cp[e + 1], sb = sp[e + 1];
      w0[e >> 1] = f2bf_pk(rope_first(bf_lo(x0), bf_lo(x1), ca, sa), rope_first(bf_hi(x0), bf_hi(x1), cb, sb));
      w1[e >> 1] = f2bf_pk(rope_second(bf_lo(x0), bf_lo(x1), ca, sa), rope_second(bf_hi(x0), bf_hi(x1), cb, sb));
    }
  }
  bf16_t* dst;
  if (part == 0) dst = qkv + (size_t)row * 3 * hidden;
  else dst = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(part == 1 ? sg.kc : sg.vc) + layer_off) + (size_t)sg.slot[lt] * hidden;
  *reinterpret_cast<uint4*>(dst + d0) = o0;
  *reinterpret_cast<uint4*>(dst + d1) = o1;
}

int ats_rope_kv_segs_slabs(const float* qkv_slabs, int splits, void* qkv, const SegTable& t, const SegTable* dt, const float* cos_tab,
                           const float* sin_tab, size_t layer_off_bytes, int n_heads, int head_dim, int max_pos, hipStream_t st) {
  ATS_REQUIRE(head_dim % 16 == 0 && splits >= 1 && qkv_slabs, ATSPEED_ERR_INVALID, "rope: slab input needs head_dim %% 16 == 0");
  const int totalv = t.total_tok * 3 * n_heads * (head_dim / 16);
  if (totalv <= 0) return ATSPEED_OK;
  rope_kv_segs_slab_kernel<<<(totalv + 255) / 256, 256, 0, st>>>(qkv_slabs, splits, (bf16_t*)qkv, dt, cos_tab, sin_tab, layer_off_bytes, n_heads,
                                                                 head_dim, max_pos);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

int ats_rope_kv_segs(void* qkv, const SegTable& t, const SegTable* dt, const float* cos_tab, const float* sin_tab, size_t layer_off_bytes,
                     int n_heads, int head_dim, int max_pos, int dtype, hipStream_t st) {
  if (dtype == ATS_HALF && head_dim % 16 == 0) {
    int totalv = t.total_tok * n_heads * (head_dim / 16);
    if (totalv <= 0) return ATSPEED_OK;
    rope_kv_segs_vec_kernel<<<(totalv + 255) / 256, 256, 0, st>>>((bf16_t*)qkv, dt, cos_tab, sin_tab, layer_off_bytes, n_heads, head_dim, max_pos);
    ATS_LAUNCH_CHECK();
    return ATSPEED_OK;
  }
  int total = t.total_tok * n_heads * (head_dim / 2);
  if (total <= 0) return ATSPEED_OK;
  if (dtype == ATSPEED_F32)
    rope_kv_segs_kernel<float><<<(total + 255) / 256, 256, 0, st>>>((float*)qkv, dt, cos_tab, sin_tab, layer_off_bytes, n_heads, head_dim, max_pos);
  else
    rope_kv_segs_kernel<bf16_t><<<(total + 255) / 256, 256, 0, st>>>((bf16_t*)qkv, dt, cos_tab, sin_tab, layer_off_bytes, n_heads, head_dim, max_pos);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

// copy the last n_logit rows of every segment to consecutive rows (input of the final norm + lm_head)
__global__ void gather_logit_rows_kernel(const uint4* __restrict__ h, const SegTable* __restrict__ t, uint4* __restrict__ out,
                                         int chunks_per_row) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= t->total_logit * chunks_per_row) return;
  int lr = i / chunks_per_row, c = i - lr * chunks_per_row;
  int lo = 0, hi = t->n;
  while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (t->seg[mid].logit_row0 <= lr) lo = mid; else hi = mid; }
  const Seg& sg = t->seg[lo];
  int src_row = sg.row0 + sg.n_tok - sg.n_logit + (lr - sg.logit_row0);
  out[i] = h[(size_t)src_row * chunks_per_row + c];
}

int ats_gather_logit_rows(const void* h, const SegTable& t, const SegTable* dt, void* out, int hidden, int dtype, hipStream_t st) {
  int cpr = hidden * (dtype == ATSPEED_F32 ? 4 : 2) / 16;
  int total = t.total_logit * cpr;
  if (total <= 0) return ATSPEED_OK;
  gather_logit_rows_kernel<<<(total + 255) / 256, 256, 0, st>>>((const uint4*)h, dt, (uint4*)out, cpr);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

// ---------------------------------------------------------------------------- fp8 (OCP e4m3) row quantisation
// q[r][c] = e4m3(x[r][c] / scale[r]),  scale[r] = max|x[r][:]| / 448  (per-row = per-token / per-output-channel scales).
// One workgroup per row, 16-byte loads, 8-byte stores.  Used once for the weights and per forward for the activations
// feeding the fp8 projections (BASELINE config 5).
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16_t* __restrict__ x, int cols, int ld,
                                                             unsigned char* __restrict__ q, float* __restrict__ scale, int pk) {
  __shared__ float red[4];
  float amax = 0.f;
  for (int c = threadIdx.x * 8; c < cols; c += 256 * 8) {
    uint4 v = *reinterpret_cast<const uint4*>(x + ats_opnd_idx<2>(pk, blockIdx.x, c, ld));      // input and output are GEMM operands: both packed when pk
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(bf2f(e[j])));
  }
  amax = wave_max_f32(amax);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
  const float inv = 1.0f / sc;
  if (threadIdx.x == 0) scale[blockIdx.x] = sc;
  for (int c = threadIdx.x * 8; c < cols; c += 256 * 8) {
    uint4 v = *reinterpret_cast<const uint4*>(x + ats_opnd_idx<2>(pk, blockIdx.x, c, ld));
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = fminf(fmaxf(bf2f(e[j]) * inv, -448.f), 448.f);
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    *reinterpret_cast<uint2*>(q + ats_opnd_idx<1>(pk, blockIdx.x, c, cols)) = make_uint2((unsigned)lo, (unsigned)hi);
  }
}

// The same for PACKED operands, one workgroup per ROW PAIR: the two rows share every 128-byte line of the bf16 input (64 bytes each), so a
// workgroup per row fetches each line twice (through two L2s, the neighbours of a pair landing on different XCDs) -- 2.9 TB/s of algorithmic
// bytes in the fp8 step's profile.  Here a thread owns 16-byte chunk (tid & 7) of line (tid >> 3) + 32 i: chunks 0-3 are the even row's, 4-7 the
// odd row's, so its row is fixed; the pair stays in registers between the maximum and the conversion (read once), and 16 consecutive threads
// fill one whole 128-byte line of the packed e4m3 output.  Bit-identical to quant_rows_fp8_kernel.
template <int NC>
__global__ __launch_bounds__(256) void quant_row_pairs_fp8_kernel(const bf16_t* __restrict__ x, int rows, int cols, unsigned char* __restrict__ q,
                                                                  float* __restrict__ scale) {
  __shared__ float red[2][4];
  const int n_lines = cols >> 5;                                   // 128-byte lines of the pair: 32 bf16 of each row per line
  const uint4* xp = reinterpret_cast<const uint4*>(x) + (size_t)blockIdx.x * n_lines * 8;
  const int sub = threadIdx.x & 7, odd = sub >> 2, l0 = threadIdx.x >> 3;
  uint4 v[NC];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int line = l0 + 32 * i;
    v[i] = line < n_lines ? xp[(size_t)line * 8 + sub] : make_uint4(0, 0, 0, 0);
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&v[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(bf2f(e[j])));
  }
  const float m0 = wave_max_f32(odd ? 0.f : amax), m1 = wave_max_f32(odd ? amax : 0.f);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = m0; red[1][threadIdx.x >> 6] = m1; }
  __syncthreads();
  amax = fmaxf(fmaxf(red[odd][0], red[odd][1]), fmaxf(red[odd][2], red[odd][3]));
  const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
  const float inv = 1.0f / sc;
  const int row = 2 * blockIdx.x + odd;
  if ((threadIdx.x & 3) == 0 && l0 == 0 && row < rows) scale[row] = sc;      // threads 0 and 4
  // packed e4m3: line L' of the pair holds columns 64 L' .. 64 L' + 63 of each row (64 bytes each); input line L = columns 32 L .. 32 L + 31
  unsigned char* qp = q + (size_t)blockIdx.x * (size_t)cols * 2;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int line = l0 + 32 * i;
    if (line < n_lines) {
      const bf16_t* e = reinterpret_cast<const bf16_t*>(&v[i]);
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = fminf(fmaxf(bf2f(e[j]) * inv, -448.f), 448.f);
      int lo = 0, hi = 0;
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
      const int col = line * 32 + (sub & 3) * 8;                   // first of this chunk's 8 columns
      *reinterpret_cast<uint2*>(qp + (size_t)(col >> 6) * 128 + odd * 64 + (col & 63)) = make_uint2((unsigned)lo, (unsigned)hi);
    }
  }
}

int ats_quant_rows_fp8(const void* x, int rows, int cols, int ld, void* q, float* scale, hipStream_t st, int pk) {
  if (rows <= 0) return ATSPEED_OK;
  ATS_REQUIRE(cols % 8 == 0 && ld % 8 == 0, ATSPEED_ERR_INVALID, "quant_fp8: cols=%d / ld=%d must be multiples of 8", cols, ld);
  ATS_REQUIRE(!pk || (cols % 64 == 0 && ld == cols), ATSPEED_ERR_INVALID, "quant_fp8: packed operands need cols %% 64 == 0 and ld == cols");
  static const bool pairs_off = getenv("ATSPEED_QUANT_PAIRS") && atoi(getenv("ATSPEED_QUANT_PAIRS")) == 0;
  if (pk && !pairs_off && cols <= 12 * 1024) {                     // a pair of up to 12288 columns in registers (12 x 16 bytes per thread)
    const int n_pairs = (rows + 1) / 2, nc = ((cols >> 5) + 31) / 32;     // the pad row of an odd count is allocated (operands hold an even number of rows)
    if (nc <= 4)       quant_row_pairs_fp8_kernel<4><<<n_pairs, 256, 0, st>>>((const bf16_t*)x, rows, cols, (unsigned char*)q, scale);
    else if (nc <= 8)  quant_row_pairs_fp8_kernel<8><<<n_pairs, 256, 0, st>>>((const bf16_t*)x, rows, cols, (unsigned char*)q, scale);
    else               quant_row_pairs_fp8_kernel<12><<<n_pairs, 256, 0, st>>>((const bf16_t*)x, rows, cols, (unsigned char*)q, scale);
    ATS_LAUNCH_CHECK();
    return ATSPEED_OK;
  }
  quant_rows_fp8_kernel<<<rows, 256, 0, st>>>((const bf16_t*)x, cols, ld, (unsigned char*)q, scale, pk);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}




// ---------------------------------------------------------------------------- packed operand layout (common.h: ats_pk_byte)
// row-major [rows][row_bytes] <-> packed; 16-byte chunks, one thread each; the pad row of an odd row count is written as zeros
__global__ void pack_rows_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int rows, int row_bytes, int to_packed) {
  const size_t cpr = (size_t)row_bytes >> 4;                     // 16-byte chunks per row
  const size_t rows_even = ((size_t)rows + 1) & ~(size_t)1;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows_even * cpr) return;
  const size_t r = i / cpr, c = i % cpr;
  const size_t pk = ats_pk_byte(r, c << 4, (size_t)row_bytes) >> 4;
  if (to_packed) dst[pk] = r < (size_t)rows ? src[r * cpr + c] : make_uint4(0, 0, 0, 0);
  else if (r < (size_t)rows) dst[r * cpr + c] = src[pk];
}

int ats_pack_rows(const void* src, void* dst, int rows, int row_bytes, int to_packed, hipStream_t st) {
  if (rows <= 0) return ATSPEED_OK;
  ATS_REQUIRE(src && dst && src != dst && row_bytes > 0 && row_bytes % 64 == 0, ATSPEED_ERR_INVALID,
              "pack_rows: row size %d must be a positive multiple of 64 bytes (out of place)", row_bytes);
  ATS_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, ATSPEED_ERR_INVALID, "pack_rows: buffers must be 16-byte aligned");
  const size_t n = (((size_t)rows + 1) & ~(size_t)1) * ((size_t)row_bytes >> 4);
  pack_rows_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>((const uint4*)src, (uint4*)dst, rows, row_bytes, to_packed);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

int ats_fill_hash_normal(void* dst, size_t n, uint32_t seed, float scale, float add, int dtype, uint64_t offset, hipStream_t st) {
  ATS_REQUIRE(dst && (dtype == ATSPEED_F32 || dtype == ATS_HALF), ATSPEED_ERR_INVALID, "fill: bad arguments");
  if (n == 0) return ATSPEED_OK;
  size_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (dtype == ATSPEED_F32)
    fill_hash_normal_kernel<float><<<(unsigned)blocks, 256, 0, st>>>((float*)dst, n, seed, scale, add, offset);
  else
    fill_hash_normal_kernel<bf16_t><<<(unsigned)blocks, 256, 0, st>>>((bf16_t*)dst, n, seed, scale, add, offset);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

}  // namespace ATS_NS

#ifndef ATS_F16_FLAVOUR          // the C ABI exists once; it picks the flavour by the dtype code (fp8 entry points: bf16 input)
extern "C" int atspeed_fill_hash_normal(void* dst, size_t n, uint32_t seed, float scale, float add, int dtype,
                                        uint64_t offset, void* stream) {
  return ATS_KD(dtype, ats_fill_hash_normal(dst, n, seed, scale, add, dtype, offset, (hipStream_t)stream));
}

extern "C" int atspeed_rmsnorm_quant_fp8(const void* x, const void* w, void* y, void* q, float* scale, int32_t rows, int32_t hidden,
                                         float eps, void* stream) {
  ATS_REQUIRE(x && w && q && scale && hidden > 0, ATSPEED_ERR_INVALID, "rmsnorm_quant_fp8: bad arguments");
  return ats_bf16::ats_rmsnorm_quant_fp8(x, w, y, q, scale, rows, hidden, eps, (hipStream_t)stream, 0);
}

extern "C" int atspeed_rmsnorm(const void* x, const void* w, void* y, int32_t rows, int32_t hidden, float eps,
                               int32_t dtype, void* stream) {
  ATS_REQUIRE(x && w && y && hidden > 0, ATSPEED_ERR_INVALID, "rmsnorm: bad arguments");
  return ATS_KD(dtype, ats_rmsnorm(x, w, y, rows, hidden, eps, dtype, (hipStream_t)stream, 0));
}

extern "C" int atspeed_quant_rows_fp8(const void* x, int32_t rows, int32_t cols, void* q, float* scale, void* stream) {
  ATS_REQUIRE(x && q && scale, ATSPEED_ERR_INVALID, "quant_fp8: null argument");
  return ats_bf16::ats_quant_rows_fp8(x, rows, cols, cols, q, scale, (hipStream_t)stream, 0);
}

extern "C" int atspeed_quant_rows_fp8_packed(const void* x, int32_t rows, int32_t cols, void* q, float* scale, void* stream) {
  ATS_REQUIRE(x && q && scale, ATSPEED_ERR_INVALID, "quant_fp8: null argument");
  return ats_bf16::ats_quant_rows_fp8(x, rows, cols, cols, q, scale, (hipStream_t)stream, 1);
}

extern "C" int atspeed_pack_rows(const void* src, void* dst, int32_t rows, int32_t row_bytes, void* stream) {
  return ats_bf16::ats_pack_rows(src, dst, rows, row_bytes, 1, (hipStream_t)stream);       // bytes only: no flavour
}
extern "C" int atspeed_unpack_rows(const void* src, void* dst, int32_t rows, int32_t row_bytes, void* stream) {
  return ats_bf16::ats_pack_rows(src, dst, rows, row_bytes, 0, (hipStream_t)stream);
}
#endif
