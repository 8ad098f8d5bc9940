// Why the engine's fp16 flavour runs ~5 % behind bf16 (configs.fp16 4742 vs 4999 items/s, gate_up 0.566 vs 0.593 of nominal): the two
// ring kernels have the same ISA (same 246 VGPRs, 256 MFMAs, 96 ds_read_b128 and 32 LDS-DMA per unrolled loop: tools/isa_mix.py), so what is
// left is the matrix pipe itself under the chip's power limit.  Registers-only loop, 8 waves per CU, all CUs, N(0, 1) operands:
//   (a) v_mfma_f32_16x16x32_bf16 on the values rounded to bf16 (8-bit significands);
//   (b) v_mfma_f32_16x16x32_f16 on the values rounded to fp16 (11-bit significands);
//   (c) v_mfma_f32_16x16x32_f16 on the BF16-ROUNDED values held as fp16 (the low 3 significand bits zero): same instruction as (b), data of (a).
// If (b) < (a) ~ (c), the deficit is the wider significands toggling more of the multiplier array (data, not the instruction).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_f16_vs_bf16.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <bool F16>
__global__ __launch_bounds__(512, 1) void k(const u32x4_t* __restrict__ in, float* out, int iters) {
  u32x4_t a[4], b[8];
  for (int i = 0; i < 4; ++i) a[i] = in[(threadIdx.x * 12 + i) & 4095];
  for (int i = 0; i < 8; ++i) b[i] = in[(threadIdx.x * 12 + 4 + i) & 4095];
  f32x4_t acc[4][8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (F16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a[i]), __builtin_bit_cast(f16x8_t, b[j]), acc[i][j], 0, 0, 0);
        else               acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[i]), __builtin_bit_cast(bf16x8_t, b[j]), acc[i][j], 0, 0, 0);
      }
  }
  float s = 0; for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j][0];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
static uint16_t to_bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float from_bf16(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t to_f16(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
int main() {
  const int n = 4096 * 8;
  std::mt19937 rng(7); std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<uint16_t> hb(n), hf(n), hfb(n);
  for (int i = 0; i < n; ++i) { const float v = nd(rng); hb[i] = to_bf16(v); hf[i] = to_f16(v); hfb[i] = to_f16(from_bf16(hb[i])); }
  u32x4_t *db, *df, *dfb; float* dout;
  hipMalloc(&db, n * 2); hipMalloc(&df, n * 2); hipMalloc(&dfb, n * 2); hipMalloc(&dout, 4096 * 512 * 4);
  hipMemcpy(db, hb.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(df, hf.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(dfb, hfb.data(), n * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 6000, blocks = 256 * 8;
  const char* names[] = {"(a) bf16 MFMA, bf16-rounded values", "(b) f16 MFMA, fp16-rounded values ", "(c) f16 MFMA, bf16-rounded values "};
  for (int round = 0; round < 3; ++round)
    for (int v = 0; v < 3; ++v) {
      hipEventRecord(e0);
      if (v == 0) k<false><<<blocks, 512>>>(db, dout, iters);
      if (v == 1) k<true><<<blocks, 512>>>(df, dout, iters);
      if (v == 2) k<true><<<blocks, 512>>>(dfb, dout, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)blocks * 8 * iters * 32 * 16384.0;
      if (round) printf("%s: %7.1f ms  %6.0f TF\n", names[v], ms, flops / ms / 1e9);
    }
  return 0;
}
