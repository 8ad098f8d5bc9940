// Shared device helpers and host-side error plumbing for libatspeed_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/atspeed_hip.h"

// ---------------------------------------------------------------- the 16-bit flavour of a kernel translation unit
// The engine computes in fp32, bf16 or fp16 (ATSPEED_F16: the type the reference loads both models in, code/inference.py:75-100).  bf16 and
// fp16 kernels are the SAME source: gemm.hip, attn.hip and elementwise.hip are compiled twice (Makefile), once per flavour, each into its
// own namespace (ats_bf16 / ats_f16); what differs is confined to this block -- the element conversions, the MFMA instruction of the type
// (v_mfma_f32_16x16x32_{bf16,f16}, v_mfma_f32_32x32x16_{bf16,f16}) and the packed conversion (v_cvt_pk_{bf16,f16}_f32).  `bf16_t` and the
// helper names (bf2f, f2bf, f2bf_pk, bf_lo, bf_hi) keep their names in both flavours: "the engine's 16-bit type".  engine.hip picks the
// namespace by the model's dtype (ATS_K).
typedef unsigned short bf16_t;   // raw bits of the flavour's 16-bit type
#ifdef ATS_F16_FLAVOUR
#define ATS_NS ats_f16
#define ATS_HALF ATSPEED_F16
#define ATS_MFMA_16x16x32_NAME "v_mfma_f32_16x16x32_f16"
#define ATS_CVT_PK_NAME "v_cvt_pk_f16_f32"
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8_t;
#define ATS_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define ATS_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#else
#define ATS_NS ats_bf16
#define ATS_HALF ATSPEED_BF16
#define ATS_MFMA_16x16x32_NAME "v_mfma_f32_16x16x32_bf16"
#define ATS_CVT_PK_NAME "v_cvt_pk_bf16_f32"
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
#define ATS_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define ATS_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif
// a call of a flavoured internal function, by dtype code: ATS_KD(dt, ats_gemm(...)) -> ats_f16::ats_gemm(...) or ats_bf16::ats_gemm(...)
// (the bf16 build also holds the fp32 parity kernels)
#define ATS_KD(dtype, call) ((dtype) == ATSPEED_F16 ? ats_f16::call : ats_bf16::call)
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
// 16-byte register value as an SSA vector: HIP's uint4 STRUCT in a register ring ended up in scratch (hipcc 7.2)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- host error plumbing
void atspeed_set_error(const char* fmt, ...);

#define ATS_HIP(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      atspeed_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return ATSPEED_ERR_HIP;                                                                      \
    }                                                                                              \
  } while (0)

#define ATS_REQUIRE(cond, code, ...)     \
  do {                                   \
    if (!(cond)) {                       \
      atspeed_set_error(__VA_ARGS__);    \
      return (code);                     \
    }                                    \
  } while (0)

#define ATS_LAUNCH_CHECK() ATS_HIP(hipGetLastError())

#define ATS_TRY(expr)            \
  do {                           \
    int _s = (expr);             \
    if (_s != ATSPEED_OK) return _s; \
  } while (0)

// per-(thread, device) state: one process may drive several GPUs from one thread, so staging buffers, events and the
// "function attribute set" flags are kept per HIP device, selected by the device current at the call
constexpr int ATS_MAX_DEVICES = 64;
// index of the current HIP device in the per-device tables, or -1 when hipGetDevice fails or the id is beyond the tables: callers
// must not alias another device's staging memory / events (they return ATSPEED_ERR_NO_DEVICE)
inline int ats_cur_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= ATS_MAX_DEVICES) return -1;
  return d;
}
struct AtsPerDeviceFlag {
  bool done[ATS_MAX_DEVICES] = {};
  bool never = false;                 // unknown device: "not done yet" every time (the attribute call is simply repeated)
  bool& cur() { const int d = ats_cur_device(); if (d < 0) { never = false; return never; } return done[d]; }
};

// ---------------------------------------------------------------- packed GEMM-operand layout
// What the LDS-DMA of the ring GEMMs wants from HBM (measured, tools/probe/dma_depth.hip: 83 GB/s per CU against 55 GB/s): every
// 1 KB DMA piece (16 rows x 64 bytes of K) made of FULL 128-byte lines.  A row-major operand gives it 16 half lines.  In the packed
// layout two consecutive rows share a line per 64-byte k-block:
//   byte b of row r  ->  ((r >> 1) * (row_bytes / 64) + (b >> 6)) * 128 + (r & 1) * 64 + (b & 63)
// (row_bytes % 64 == 0, buffers hold an even number of rows).  The bf16 / fp8 engine keeps every GEMM operand in it -- projection
// weights, lm_head, and the activations a projection reads (RMSNorm output, attention output, SwiGLU output, their e4m3 forms) --
// written that way by their producers; everything else (residual stream, qkv, logits, KV cache, fp32 parity mode) stays row-major.
__host__ __device__ inline size_t ats_pk_byte(size_t row, size_t byte_in_row, size_t row_bytes) {
  return ((row >> 1) * (row_bytes >> 6) + (byte_in_row >> 6)) * 128 + (row & 1) * 64 + (byte_in_row & 63);
}
// element index of (row, col) in an operand of `ld` elements per row, element size 2^esz_log2 bytes, packed or row-major
template <int ESZ>
__host__ __device__ inline size_t ats_opnd_idx(int pk, size_t row, size_t col, size_t ld) {
  return pk ? ats_pk_byte(row, col * ESZ, ld * ESZ) / ESZ : row * ld + col;
}

// counter-based hash shared by the synthetic-weight fill and the sampling kernels (= atspeed_amd/synth.py:hash_u32)
__host__ __device__ inline uint32_t ats_fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__host__ __device__ inline uint32_t ats_hash_u32(uint32_t idx, uint32_t seed) { return ats_fmix32(idx * 0x9E3779B1u + seed); }
// sub-seed of one random stream: purpose | round << 8 | step << 16 | model tag << 24 (oracle/beamsd_sample_ref.py:HashRng.begin)
enum { ATS_RNG_STEP = 1, ATS_RNG_ACCEPT = 2, ATS_RNG_PERM = 3, ATS_RNG_RESID = 4, ATS_RNG_BONUS = 5 };
__host__ __device__ inline uint32_t ats_rng_sub(uint32_t seed, int purpose, int round, int step, int tag) {
  return ats_hash_u32((uint32_t)(purpose & 0xff) | ((uint32_t)(round & 0xff) << 8) | ((uint32_t)(step & 0xff) << 16) | ((uint32_t)(tag & 0xff) << 24), seed);
}

// ---------------------------------------------------------------- device helpers
#if defined(__HIPCC__)
__device__ __forceinline__ float ats_u01(uint32_t h) { return ((float)(h >> 9) + 0.5f) * 1.1920928955078125e-07f; }   // (k + 1/2) 2^-23, exact
__device__ __forceinline__ float ats_gumbel(uint32_t h) { return -logf(-logf(ats_u01(h))); }
typedef __attribute__((ext_vector_type(2))) float ats_f32x2_t;
#ifdef ATS_F16_FLAVOUR
// fp16 flavour: hardware conversions both ways (v_cvt_f32_f16 / v_cvt_f16_f32, round to nearest even; v_cvt_pk_f16_f32 for pairs on gfx950)
typedef __attribute__((ext_vector_type(2))) _Float16 ats_bf16x2_t;
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }
__device__ __forceinline__ uint32_t f2bf_pk(float lo, float hi) {
  ats_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, ats_bf16x2_t));
}
__device__ __forceinline__ float bf_lo(uint32_t pk) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(pk & 0xffffu)); }
__device__ __forceinline__ float bf_hi(uint32_t pk) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(pk >> 16)); }
#else
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round to nearest even in hardware (v_cvt_pk_bf16_f32 on gfx950: the compiler pairs neighbouring conversions); the six-instruction
// integer form this replaces was a sixth of the ring GEMM's epilogue and most of the attention softmax's VALU work
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
// two values -> one packed register (low half = lo); and the two values back as floats (a bf16 is the high half of its fp32)
typedef __attribute__((ext_vector_type(2))) __bf16 ats_bf16x2_t;
__device__ __forceinline__ uint32_t f2bf_pk(float lo, float hi) {
  ats_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, ats_bf16x2_t));
}
__device__ __forceinline__ float bf_lo(uint32_t pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float bf_hi(uint32_t pk) { return __uint_as_float(pk & 0xffff0000u); }
#endif
// SiLU.  bf16 engine (EXACT = false): the quotient through the hardware reciprocal (v_rcp_f32, 1 ulp of fp32 -- far below the bf16 rounding that
// follows); the correctly rounded fp32 division is a ten-instruction sequence and made the ring GEMM's SwiGLU epilogue VALU-bound (64 outputs per
// thread).  fp32 engine (parity mode): the exact quotient.
template <bool EXACT> __device__ __forceinline__ float ats_silu(float g) {
  const float d = 1.f + __expf(-g);
  if constexpr (EXACT) return g / d;
  else return g * __builtin_amdgcn_rcpf(d);
}
// The rotary pair (x0, x1) = (x[d], x[d + head_dim/2]) of the bf16 engine, with the contraction spelled out: the separate RoPE pass, the
// slab-summing one and the qkv GEMM's fused epilogue must round identically (the compiler is otherwise free to pick which product it fuses).
__device__ __forceinline__ float rope_first(float x0, float x1, float c, float s) { return __fmaf_rn(x0, c, -__fmul_rn(x1, s)); }    // x0 cos - x1 sin
__device__ __forceinline__ float rope_second(float x0, float x1, float c, float s) { return __fmaf_rn(x1, c, __fmul_rn(x0, s)); }    // x1 cos + x0 sin

template <typename T> struct Elt;
template <> struct Elt<float> {
  static constexpr int kPerChunk = 4;   // elements per 16-byte chunk
  __device__ static __forceinline__ float load(const float* p) { return *p; }
  __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct Elt<bf16_t> {
  static constexpr int kPerChunk = 8;
  __device__ static __forceinline__ float load(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void store(bf16_t* p, float v) { *p = f2bf(v); }
};

__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// maximum of a 64-bit key over the wave, returned wave-uniform.  Data-parallel-primitive moves instead of ds_bpermute shuffles (a block top-K
// runs this once per extracted key: 12 LDS round trips per call were most of a one-user beam step's 85 us): row_shr 1/2/4/8 leave a 16-lane
// row's maximum in its lane 15 (max is idempotent, an invalid source lane keeps the own value), row_bcast15 / row_bcast31 carry it to
// lane 63, one readlane pair makes it uniform.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long ats_dpp_max_u64(unsigned long long v) {
  const int lo = (int)(unsigned)(v & 0xffffffffull), hi = (int)(unsigned)(v >> 32);
  const unsigned wlo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
  const unsigned whi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
  const unsigned long long w = ((unsigned long long)whi << 32) | wlo;
  return w > v ? w : v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  v = ats_dpp_max_u64<0x111, 0xf>(v);        // row_shr:1
  v = ats_dpp_max_u64<0x112, 0xf>(v);        // row_shr:2
  v = ats_dpp_max_u64<0x114, 0xf>(v);        // row_shr:4
  v = ats_dpp_max_u64<0x118, 0xf>(v);        // row_shr:8   -> lane 15 of every row: the row's maximum
  v = ats_dpp_max_u64<0x142, 0xa>(v);        // row_bcast15 into rows 1 and 3
  v = ats_dpp_max_u64<0x143, 0xc>(v);        // row_bcast31 into rows 2 and 3 -> lane 63: the wave's maximum
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffull), 63);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}

// monotone float <-> uint32 map: a > b  <=>  ford(a) > ford(b)   (-inf -> 0x007fffff)
__device__ __forceinline__ uint32_t ford(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ford_inv(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}
#endif
