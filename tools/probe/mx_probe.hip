// Probe of the block-scaled fp8 MFMA on gfx950 (tuning aid, not part of the product):
//   1. operand / result lane maps of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands and unit scales, checked with exact
//      small-integer data against a host product (assumed map: lane l holds row l&31, bytes b = 0..31 <-> k = 32*(l>>5) + b;
//      C/D as the bf16 32x32 forms: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5));
//   2. register-only issue rate of that MFMA against v_mfma_f32_32x32x16_bf16 and v_mfma_f32_16x16x32_fp8_fp8 on random operands.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/probe/mx_probe.hip -o /tmp/mx_probe && /tmp/mx_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

static float e4m3_to_float(uint8_t v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f;
  if (e == 0) f = ldexpf((float)m, -9);
  else if (e == 15 && m == 7) f = NAN;
  else f = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -f : f;
}

__global__ void one_mfma(const v8i* a, const v8i* b, v16f* c, int scale_a, int scale_b) {
  v16f acc = {0};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, scale_a, 0, scale_b);
  c[threadIdx.x] = acc;
}

template <int KIND>
__global__ __launch_bounds__(512) void rate_kernel(float* out, int iters, unsigned seed) {
  unsigned h = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + seed;
  v8i a, b;
  for (int i = 0; i < 8; ++i) { h = h * 1664525u + 1013904223u; a[i] = (int)(h & 0x3f3f3f3f) | 0x20202020; h = h * 1664525u + 1013904223u; b[i] = (int)(h & 0x3f3f3f3f) | 0x20202020; }
  v16f acc[4] = {{0}, {0}, {0}, {0}};
  v4f acc4[8] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}};
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[u], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    } else if constexpr (KIND == 1) {
      v8bf ab = __builtin_bit_cast(v8bf, __builtin_shufflevector(a, a, 0, 1, 2, 3)), bb = __builtin_bit_cast(v8bf, __builtin_shufflevector(b, b, 0, 1, 2, 3));
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[u], 0, 0, 0);
    } else if constexpr (KIND == 2) {
      long al = ((long)a[1] << 32) | (unsigned)a[0], bl = ((long)b[1] << 32) | (unsigned)b[0];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc4[u] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(al, bl, acc4[u], 0, 0, 0);
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc4[u] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc4[u], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
  }
  float s = 0.f;
  for (int u = 0; u < 4; ++u) for (int i = 0; i < 16; ++i) s += acc[u][i];
  for (int u = 0; u < 8; ++u) for (int i = 0; i < 4; ++i) s += acc4[u][i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  // ---- 1. lane maps
  std::vector<uint8_t> A(32 * 64), B(32 * 64);        // A[row][k], B[col][k] as e4m3 bytes
  srand(5);
  const uint8_t vals[] = {0x00, 0x30, 0x38, 0x3c, 0x40, 0x44, 0xb0, 0xb8, 0xc0, 0x28, 0xa8, 0x48};   // 0, .5, 1, 1.5, 2, 3, -.5, -1, -2, .25, -.25, 4
  for (auto& x : A) x = vals[rand() % 12];
  for (auto& x : B) x = vals[rand() % 12];
  std::vector<float> ref(32 * 32, 0.f);
  for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) { float s = 0; for (int k = 0; k < 64; ++k) s += e4m3_to_float(A[r * 64 + k]) * e4m3_to_float(B[c * 64 + k]); ref[r * 32 + c] = s; }
  std::vector<uint8_t> la(64 * 32), lb(64 * 32);
  for (int l = 0; l < 64; ++l) for (int b = 0; b < 32; ++b) { la[l * 32 + b] = A[(l & 31) * 64 + 32 * (l >> 5) + b]; lb[l * 32 + b] = B[(l & 31) * 64 + 32 * (l >> 5) + b]; }
  void *da, *db, *dc;
  hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dc, 64 * 64);
  hipMemcpy(da, la.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, lb.data(), 2048, hipMemcpyHostToDevice);
  for (int sc = 0; sc < 3; ++sc) {
    const int sa = sc == 0 ? 0x7f7f7f7f : sc == 1 ? 0x80808080 : 0x7f7f7f7f, sb = sc == 2 ? 0x7e7e7e7e : 0x7f7f7f7f;
    one_mfma<<<1, 64>>>((const v8i*)da, (const v8i*)db, (v16f*)dc, sa, sb);
    std::vector<float> out(64 * 16);
    hipMemcpy(out.data(), dc, 64 * 64, hipMemcpyDeviceToHost);
    const float want_scale = sc == 0 ? 1.f : sc == 1 ? 2.f : 0.5f;
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int reg = 0; reg < 16; ++reg) {
      int col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
      if (out[l * 16 + reg] != want_scale * ref[row * 32 + col]) { if (bad < 4) printf("  mismatch lane %d reg %d: got %g want %g\n", l, reg, out[l * 16 + reg], want_scale * ref[row * 32 + col]); ++bad; }
    }
    printf("lane-map check (scale_a %08x scale_b %08x, expect x%g): %s (%d of 1024 differ)\n", sa, sb, want_scale, bad ? "FAIL" : "ok", bad);
  }
  // ---- 2. issue rates (random operands, 8 waves x 1024 workgroups)
  float* dout;
  hipMalloc(&dout, 1024 * 512 * 4);
  const char* names[] = {"v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3)", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_16x16x32_fp8_fp8", "v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3)"};
  const double flop_per_it[] = {4.0 * 2 * 32 * 32 * 64, 4.0 * 2 * 32 * 32 * 16, 8.0 * 2 * 16 * 16 * 32, 8.0 * 2 * 16 * 16 * 128};
  for (int kind = 0; kind < 4; ++kind) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (kind == 0) rate_kernel<0><<<1024, 512>>>(dout, iters, 7u);
      if (kind == 1) rate_kernel<1><<<1024, 512>>>(dout, iters, 7u);
      if (kind == 2) rate_kernel<2><<<1024, 512>>>(dout, iters, 7u);
      if (kind == 3) rate_kernel<3><<<1024, 512>>>(dout, iters, 7u);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s %8.1f TFLOP/s  (%.2f ms)\n", names[kind], flop_per_it[kind] * iters * 1024.0 * 8 / (ms * 1e-3) / 1e12, ms);
  }
  return 0;
}
