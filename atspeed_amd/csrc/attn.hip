// Tree attention over a slot-addressed KV cache.
//
// The reference feeds HF attention a dense additive mask [1,1,T,S] of 0 / finfo.min built by
// concatenating parent rows and identity blocks (beamSD.py:89,204-209,396-400).  Here a query
// row carries a bitset over cache slots (prompt prefix + one ancestor per accepted/draft
// block + itself); the kernel compacts the set bits and touches only visible slots.
//
// v1 kernel (round 1): one wave per (token, head).  Phase 1: lanes own visible slots and
// compute q.k from global K rows; phase 2: wave softmax in fp32; phase 3: lanes own head
// dims and accumulate p*V with coalesced V reads.  Attention is < 2 % of the forward's
// bytes/FLOPs at these sizes (S <= ~250); an MFMA/LDS-tiled version is the next step once
// the GEMMs stop dominating.
#include "internal.h"

namespace {

constexpr int kMaxSlots = 2048;

template <typename T>
__global__ __launch_bounds__(256) void tree_attn_kernel(const T* __restrict__ q, int ldq, const T* __restrict__ kc,
                                                        const T* __restrict__ vc, const uint64_t* __restrict__ vis,
                                                        int vis_words, T* __restrict__ out, int ldo, int n_slots,
                                                        int n_heads, int head_dim, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = blockIdx.x;
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
    uint64_t bits = vis[(size_t)t * vis_words + w];
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
  T* orow = out + (size_t)t * ldo + h * head_dim;
  for (int d = lane; d < head_dim; d += 64) {
    float acc = 0.f;
    for (int i = 0; i < n_vis; ++i) acc += prob[i] * Elt<T>::load(vc + (size_t)slot_list[i] * hidden + h * head_dim + d);
    Elt<T>::store(orow + d, acc * inv);
  }
}

}  // namespace

int ats_tree_attention(const void* q, int ldq, const void* kcache, const void* vcache, const uint64_t* vis,
                       int vis_words, void* out, int ldo, int n_tokens, int n_slots, int n_heads, int head_dim,
                       int dtype, hipStream_t st) {
  if (n_tokens <= 0) return ATSPEED_OK;
  ATS_REQUIRE(head_dim % 8 == 0 && head_dim <= 256, ATSPEED_ERR_INVALID, "attention: head_dim %d unsupported", head_dim);
  ATS_REQUIRE(n_slots <= vis_words * 64 && vis_words * 64 <= kMaxSlots, ATSPEED_ERR_CAPACITY,
              "attention: %d slots exceed the visibility bitset (%d words)", n_slots, vis_words);
  dim3 grid(n_tokens, (n_heads + 3) / 4);
  size_t lds = (size_t)4 * (head_dim + 2 * vis_words * 64) * sizeof(float);
  float scale = 1.0f / sqrtf((float)head_dim);
  if (dtype == ATSPEED_F32)
    tree_attn_kernel<float><<<grid, 256, lds, st>>>((const float*)q, ldq, (const float*)kcache, (const float*)vcache, vis,
                                                    vis_words, (float*)out, ldo, n_slots, n_heads, head_dim, scale);
  else
    tree_attn_kernel<bf16_t><<<grid, 256, lds, st>>>((const bf16_t*)q, ldq, (const bf16_t*)kcache, (const bf16_t*)vcache,
                                                     vis, vis_words, (bf16_t*)out, ldo, n_slots, n_heads, head_dim, scale);
  ATS_LAUNCH_CHECK();
  return ATSPEED_OK;
}

extern "C" int atspeed_tree_attention(const void* q, int32_t ldq, const void* kcache, const void* vcache,
                                      const uint64_t* vis, int32_t vis_words, void* out, int32_t n_tokens,
                                      int32_t n_slots, int32_t n_heads, int32_t head_dim, int32_t dtype, void* stream) {
  ATS_REQUIRE(q && kcache && vcache && vis && out, ATSPEED_ERR_INVALID, "attention: null argument");
  return ats_tree_attention(q, ldq, kcache, vcache, vis, vis_words, out, n_heads * head_dim, n_tokens, n_slots, n_heads,
                            head_dim, dtype, (hipStream_t)stream);
}
