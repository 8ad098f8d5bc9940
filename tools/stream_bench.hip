// Micro-benchmark: how fast can 256 CUs stream a weight matrix W[N][K] (bf16) with different
// per-wave access shapes?  (tuning tool; not part of the library)
//   mode 0: fully contiguous: wave w reads 1 KB pieces of its own contiguous span
//   mode 1: MFMA-fragment shape on a ROW-MAJOR matrix: 16 rows x 64 B per instruction (row stride K*2)
//   mode 2: tile-row shape on row-major: 8 rows x 128 B per instruction
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const unsigned short* __restrict__ W, int N, int K, int rows_per_wave, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * 4 + wave;                  // global wave id
  const int n0 = gw * rows_per_wave;                     // this wave owns rows [n0, n0 + rows_per_wave)
  if (n0 >= N) return;
  u32x4 acc = {0, 0, 0, 0};
  const size_t row_bytes = (size_t)K * 2;
  const char* base = (const char*)W;
  if (MODE == 0) {
    const char* p = base + (size_t)n0 * row_bytes + lane * 16;
    const size_t total = (size_t)rows_per_wave * row_bytes;
    for (size_t off = 0; off < total; off += 1024 * DEPTH) {
      u32x4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) v[d] = *(const u32x4*)(p + min(off + (size_t)d * 1024, total - 1024));
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
    }
  } else if (MODE == 1) {
    for (int r0 = 0; r0 < rows_per_wave; r0 += 16) {
      const char* p = base + (size_t)(n0 + r0 + (lane & 15)) * row_bytes + (lane >> 4) * 16;
      for (int kb = 0; kb < K * 2; kb += 64 * DEPTH) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = *(const u32x4*)(p + min(kb + d * 64, K * 2 - 64));
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
      }
    }
  } else {
    for (int r0 = 0; r0 < rows_per_wave; r0 += 8) {
      const char* p = base + (size_t)(n0 + r0 + (lane >> 3)) * row_bytes + (lane & 7) * 16;
      for (int kb = 0; kb < K * 2; kb += 128 * DEPTH) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = *(const u32x4*)(p + min(kb + d * 128, K * 2 - 128));
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
      }
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int MODE, int DEPTH>
float run(const unsigned short* W, int N, int K, int rpw, unsigned* sink, int iters) {
  int waves = (N + rpw - 1) / rpw;
  int blocks = (waves + 3) / 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  stream_kernel<MODE, DEPTH><<<blocks, 256>>>(W, N, K, rpw, sink);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) stream_kernel<MODE, DEPTH><<<blocks, 256>>>(W + (size_t)(i % 4) * N * K, N, K, rpw, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

int main() {
  const int configs[][2] = {{12288, 4096}, {4096, 11008}, {22016, 4096}};
  unsigned* sink; hipMalloc(&sink, 4);
  for (auto& c : configs) {
    int N = c[0], K = c[1];
    size_t bytes = (size_t)N * K * 2;
    unsigned short* W; hipMalloc(&W, bytes * 4);
    hipMemset(W, 1, bytes * 4);
    for (int rpw : {16, 32, 64}) {
      float a = run<0, 8>(W, N, K, rpw, sink, 20), b = run<1, 8>(W, N, K, rpw, sink, 20), cc = run<2, 8>(W, N, K, rpw, sink, 20);
      float b16 = run<1, 16>(W, N, K, rpw, sink, 20);
      printf("N=%5d K=%5d rows/wave=%2d waves=%5d | contiguous %7.1f GB/s | frag16x64B %7.1f GB/s (depth16 %7.1f) | tile8x128B %7.1f GB/s\n",
             N, K, rpw, (N + rpw - 1) / rpw, bytes / a / 1e6, bytes / b / 1e6, bytes / b16 / 1e6, bytes / cc / 1e6);
    }
    hipFree(W);
  }
  return 0;
}
