// Is the ring GEMM's L2 -> LDS stream latency-bound?  (tuning aid, not part of the product)
// Every workgroup (8 waves, one per CU) streams the operand panels of a 256 x 256 GEMM tile exactly as gemm_ring_kernel does -- per 32-deep
// k-step 32 KB by 32 LDS-DMA pieces of 16 rows x 64 B (global_load_lds_dwordx4), same tile -> workgroup order (XCD-contiguous, 4-row bands),
// one barrier per k-step -- but computes nothing and never reads LDS.  D = k-steps kept in flight (the product ring: 3).  If the rate per CU
// grows with D the stream is bound by latency x bytes in flight, not by a bandwidth.
// ROWB = bytes of K per row and DMA piece: 64 is what the product ring moves (16 rows x 64 B per piece: half cache lines); 128 / 256 move the same
// bytes as full 128-byte lines (8 x 128 B, 4 x 256 B per piece).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/probe/dma_depth.hip -o /tmp/dma_depth && /tmp/dma_depth
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p; }
#define DMA16(voff, sbase, m0v) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(m0v) : "memory")

template <int D, int ROWB>
__global__ __launch_bounds__(512, 1) void stream_kernel(const unsigned short* X, const unsigned short* W, int M, int N, int K, int tiles_n, int tiles_m, int GM,
                                                         unsigned long long* cycles) {
  constexpr int STAGE = 512 * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwg = tiles_n * tiles_m;
  int bid = blockIdx.x;
  { const int q = nwg / 8, r = nwg % 8, x = bid % 8; bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8; }
  const int band = bid / (GM * tiles_n), rem = bid % (GM * tiles_n);
  const int band_rows = min(GM, tiles_m - band * GM);
  const int tn = rem / band_rows, tm = band * GM + rem % band_rows;
  const int n0 = tn * 256, m0 = tm * 256;
  const unsigned lbase = lds_addr(smem);
  unsigned woff[2], xoff[2]; int m0w[2], m0x[2];
  // a piece = 1 KB = RPP rows x ROWB bytes; a "k-step" here always moves 32 KB per workgroup (4 pieces per wave), so with ROWB = 128 the
  // 16 pieces of one operand cover 128 of its 256 rows per k-step and the k offset advances every second k-step (same bytes, full lines)
  constexpr int RPP = 1024 / ROWB, LPR = ROWB / 16;
  for (int j = 0; j < 2; ++j) {
    const int row = (wave * 2 + j) * RPP + (lane / LPR);
    woff[j] = (unsigned)min(n0 + row, N - 1) * (unsigned)(K * 2) + (lane % LPR) * 16;
    xoff[j] = (unsigned)min(m0 + row, M - 1) * (unsigned)(K * 2) + (lane % LPR) * 16;
    m0w[j] = __builtin_amdgcn_readfirstlane((int)lbase + (wave * 2 + j) * 1024);
    m0x[j] = __builtin_amdgcn_readfirstlane((int)lbase + 256 * 64 + (wave * 2 + j) * 1024);
  }
  const unsigned long long wb = (unsigned long long)W, xb = (unsigned long long)X;
  const int nks = K / 32;
  constexpr int SUB = ROWB / 64;                       // k-steps per sweep of all 256 rows
  auto issue = [&](int ks, int q) {
    const unsigned long long koff = (unsigned long long)(ks / SUB) * ROWB + (unsigned long long)(ks % SUB) * (256 / SUB) * (unsigned long long)(K * 2);
    DMA16(woff[0], wb + koff, m0w[0] + q * STAGE);
    DMA16(woff[1], wb + koff, m0w[1] + q * STAGE);
    DMA16(xoff[0], xb + koff, m0x[0] + q * STAGE);
    DMA16(xoff[1], xb + koff, m0x[1] + q * STAGE);
  };
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int q = 0; q < D; ++q) issue(q, q);
  int q = 0;
  for (int ks = 0; ks < nks; ++ks) {
    // k-step ks must have landed: all but the D - 1 younger k-steps (4 pieces each)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (D - 1)) : "memory");
    asm volatile("s_barrier" ::: "memory");
    if (ks + D < nks) issue(ks + D, q); else { issue(nks - 1, q); }      // keep the count of outstanding pieces constant to the end
    q = q + 1 == D ? 0 : q + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) cycles[blockIdx.x] = __builtin_readcyclecounter() - t0;
}

template <int D, int ROWB>
void run(const unsigned short* X, const unsigned short* W, int M, int N, int K, unsigned long long* cyc) {
  const int tiles_n = (N + 255) / 256, tiles_m = (M + 255) / 256, nwg = tiles_n * tiles_m;
  hipFuncSetAttribute((const void*)stream_kernel<D, ROWB>, hipFuncAttributeMaxDynamicSharedMemorySize, D * 32768);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    stream_kernel<D, ROWB><<<nwg, 512, D * 32768>>>(X, W, M, N, K, tiles_n, tiles_m, 4, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double bytes = (double)nwg * (K / 32) * 32768.0;
  const double rounds = (double)nwg / 256.0;
  printf("rows of %3d B, D=%d k-steps in flight (%3d KB per CU): %8.1f us per launch, %6.1f GB/s per CU, %5.2f TB/s L2->LDS, %.3f us per k-step (err %s)\n", ROWB, D, D * 32, best * 1e3,
         bytes / 256 / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e12, best * 1e3 / (rounds * (K / 32)), hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 26000, N = argc > 2 ? atoi(argv[2]) : 22016, K = argc > 3 ? atoi(argv[3]) : 4096;
  unsigned short *X, *W; unsigned long long* cyc;
  hipMalloc(&X, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&cyc, 16384 * 8);
  hipMemset(X, 0x3c, (size_t)M * K * 2); hipMemset(W, 0x3c, (size_t)N * K * 2);
  printf("operand panels of a %d x %d x %d bf16 GEMM, 256 x 256 tiles, %d workgroups\n", M, N, K, ((N + 255) / 256) * ((M + 255) / 256));
  run<2, 64>(X, W, M, N, K, cyc);
  run<3, 64>(X, W, M, N, K, cyc);
  run<4, 64>(X, W, M, N, K, cyc);
  run<5, 64>(X, W, M, N, K, cyc);
  run<3, 128>(X, W, M, N, K, cyc);
  run<4, 128>(X, W, M, N, K, cyc);
  run<3, 256>(X, W, M, N, K, cyc);
  run<4, 256>(X, W, M, N, K, cyc);
  return 0;
}
