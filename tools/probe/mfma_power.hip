// Sustained MFMA rate on random operands (power-limited clock) for the two bf16 shapes: 16x16x32 vs 32x32x16.
// 8 waves per workgroup, 1 workgroup per CU x 256 CUs x several rounds; registers only (no memory traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(512, 1) void k(const u32x4_t* __restrict__ in, float* out, int iters) {
  u32x4_t a[4], b[8];
  for (int i = 0; i < 4; ++i) a[i] = in[(threadIdx.x * 12 + i) & 4095];
  for (int i = 0; i < 8; ++i) b[i] = in[(threadIdx.x * 12 + 4 + i) & 4095];
  if constexpr (SHAPE == 16) {
    f32x4_t acc[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[i]), __builtin_bit_cast(bf16x8_t, b[j]), acc[i][j], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    f32x16_t acc[2][4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[i * 2 + ks]), __builtin_bit_cast(bf16x8_t, b[j * 2 + ks]), acc[i][j], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}
int main() {
  std::vector<uint32_t> h(4096 * 4);
  uint32_t x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (x & 0x7fff7fffu) % 0x3f803f80u + 0x3c003c00u; }   // bf16 pairs of moderate magnitude
  u32x4_t* din; float* dout; hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 4096 * 512 * 4);
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000, blocks = 256 * 8;
  for (int shape : {16, 32, 16, 32}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (shape == 16) k<16><<<blocks, 512>>>(din, dout, iters); else k<32><<<blocks, 512>>>(din, dout, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)blocks * 8 * iters * 32 * 16384.0;     // per wave and iteration: 32 MFMAs of 16384 flop (or 16 of 32768)
      if (rep) printf("mfma %s: %.1f ms  %.0f TF\n", shape == 16 ? "16x16x32" : "32x32x16", ms, flops / ms / 1e9);
    }
  }
  return 0;
}
