// How fast can ONE launch stream a projection's weights, and what does a phase boundary cost?  (tuning aid, not part of the product)
//
// One user's forward is a chain of weight streams (gate_up 180 MB, down 90 MB, qkv 100 MB, o_proj 34 MB per layer) with a dependency between
// them.  This probe moves the bytes exactly as gemm_wdma_kernel does -- 64-k tiles of ROWS weight rows x 128 bytes by LDS-DMA into a ring of
// NSTG stages, hand-counted vmcnt + one barrier per tile -- but computes nothing, so it answers:
//   1. rate vs workgroups and bytes in flight per CU (is the product kernel's 4.0 TB/s a latency x in-flight bound?);
//   2. two dependent streams as two launches vs ONE persistent launch with a grid barrier between them, with and without the second stream's
//      first tiles already in flight across the barrier (what a fused MLP kernel could gain).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/probe/wstream_depth.hip -o /tmp/wstream_depth && /tmp/wstream_depth
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <functional>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p; }
#define DMA16(voff, sbase, m0v) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(m0v) : "memory")

// stream rows [n0, n0 + ROWS) x K bf16 of W through the ring; returns a checksum word so the loads cannot be dropped
template <int ROWS, int NSTG>
__device__ __forceinline__ void stream_rows(const unsigned short* W, int N, int K, int n0, unsigned char* smem, int n_pre_issued) {
  constexpr int NPIECE = ROWS / 8, NP = (NPIECE + 3) / 4, STAGE = ROWS * 128;      // fewer than 4 pieces: the first NPIECE waves move one each
  static_assert(ROWS % 8 == 0 && (NPIECE % 4 == 0 || NPIECE < 4) && (NSTG - 2) * NP <= 63, "pieces per wave / vmcnt range");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool active = wave * NP < NPIECE;
  const unsigned lbase = lds_addr(smem);
  unsigned voff[NP]; int m0p[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int piece = wave * NP + j, row = piece * 8 + (lane >> 3), pos = lane & 7;
    voff[j] = (unsigned)min(n0 + row, N - 1) * (unsigned)(K * 2) + pos * 16;
    m0p[j] = __builtin_amdgcn_readfirstlane((int)lbase + piece * 1024);
  }
  const unsigned long long wb = (unsigned long long)W;
  const int n_kt = K / 64;
  auto issue = [&](int kt) {
    const int so = (kt % NSTG) * STAGE;
    if (active) {
#pragma unroll
      for (int j = 0; j < NP; ++j) DMA16(voff[j], wb + (unsigned long long)kt * 128, m0p[j] + so);
    }
  };
#pragma unroll
  for (int t = 0; t < NSTG - 1; ++t) if (t >= n_pre_issued && t < n_kt) issue(t);
  for (int kt = 0; kt < n_kt; ++kt) {
    if (kt + NSTG - 2 < n_kt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * NP) : "memory");
    else                      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (kt + NSTG - 1 < n_kt) issue(kt + NSTG - 1);
  }
}
// only the first NSTG - 1 tiles' issue (the part of the next stream that can fly across a grid barrier)
template <int ROWS, int NSTG>
__device__ __forceinline__ void prefetch_rows(const unsigned short* W, int N, int K, int n0, unsigned char* smem) {
  constexpr int NPIECE = ROWS / 8, NP = (NPIECE + 3) / 4, STAGE = ROWS * 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned lbase = lds_addr(smem);
  const unsigned long long wb = (unsigned long long)W;
  if (wave * NP >= NPIECE) return;
#pragma unroll
  for (int t = 0; t < NSTG - 1; ++t)
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int piece = wave * NP + j, row = piece * 8 + (lane >> 3), pos = lane & 7;
      const unsigned voff = (unsigned)min(n0 + row, N - 1) * (unsigned)(K * 2) + pos * 16;
      DMA16(voff, wb + (unsigned long long)t * 128, __builtin_amdgcn_readfirstlane((int)lbase + piece * 1024) + t * STAGE);
    }
}

template <int ROWS, int NSTG>
__global__ __launch_bounds__(256) void one_stream(const unsigned short* W, int N, int K, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  stream_rows<ROWS, NSTG>(W, N, K, blockIdx.x * ROWS, smem, 0);
  if (threadIdx.x == 0 && smem[blockIdx.x & 1023] == 0x5a && smem[17] == 0xa5) sink[0] = 1;      // keep the LDS image observable
}

// grid barrier: monotone counter, bounded spin (a workgroup that never arrives must not hang the box)
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(counter, 1u);
    int spins = 0; bool good = true;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 18)) { good = false; break; }
    }
    ok = good ? 1 : 0;
  }
  __syncthreads();
  return ok != 0;
}

// persistent: stream A (rows split evenly over the grid in 32-row units), grid barrier, stream B.  PREFETCH: B's first tiles are issued
// before the barrier into a second ring.
template <int RA, int RB, int NSTG_A, int NSTG_B, bool PREFETCH>
__global__ __launch_bounds__(256) void two_streams(const unsigned short* WA, int NA, int KA, const unsigned short* WB, int NB, int KB, unsigned* counter,
                                                   unsigned epoch, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ringB = smem + NSTG_A * RA * 128;
  stream_rows<RA, NSTG_A>(WA, NA, KA, blockIdx.x * RA, smem, 0);
  if (PREFETCH) prefetch_rows<RB, NSTG_B>(WB, NB, KB, blockIdx.x * RB, ringB);
  if (!grid_barrier(counter, epoch * gridDim.x)) { if (threadIdx.x == 0) sink[1] = 0xdead; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
  stream_rows<RB, NSTG_B>(WB, NB, KB, blockIdx.x * RB, PREFETCH ? ringB : smem, PREFETCH ? NSTG_B - 1 : 0);
  if (threadIdx.x == 0 && smem[blockIdx.x & 1023] == 0x5a && smem[17] == 0xa5) sink[0] = 1;
}

static float time_it(int reps, const std::function<void(int)>& f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f(i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) f(i);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms * 1000.f / reps;
}

template <int ROWS, int NSTG>
static void run_one(const char* name, std::vector<unsigned short*>& W, int N, int K, unsigned* sink) {
  const int lds = NSTG * ROWS * 128, grid = (N + ROWS - 1) / ROWS;
  CK(hipFuncSetAttribute((const void*)one_stream<ROWS, NSTG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, one_stream<ROWS, NSTG>, 256, lds));
  float us = time_it(40, [&](int i) { hipLaunchKernelGGL((one_stream<ROWS, NSTG>), dim3(grid), dim3(256), lds, 0, W[i % W.size()], N, K, sink); });
  const double bytes = (double)N * K * 2;
  printf("%-10s N=%5d K=%5d rows/WG=%3d WGs=%3d ring=%2d x %2d KB (%3d KB, %d WG/CU) in flight/WG %3d KB : %6.1f us  %5.2f TB/s\n", name, N, K, ROWS, grid, NSTG,
         ROWS * 128 / 1024, lds / 1024, occ, (NSTG - 1) * ROWS * 128 / 1024, us, bytes / us / 1e6);
}

int main() {
  const int NGU = 22016, KGU = 4096, ND = 4096, KD = 11008;
  const int NBUF = 6;                                   // 6 x (180 + 90) MB > the 256 MB Infinity Cache: every launch streams from HBM
  std::vector<unsigned short*> WA(NBUF), WB(NBUF);
  for (int i = 0; i < NBUF; ++i) {
    CK(hipMalloc(&WA[i], (size_t)NGU * KGU * 2)); CK(hipMemset(WA[i], 0x11 + i, (size_t)NGU * KGU * 2));
    CK(hipMalloc(&WB[i], (size_t)ND * KD * 2));   CK(hipMemset(WB[i], 0x21 + i, (size_t)ND * KD * 2));
  }
  unsigned* sink; CK(hipMalloc(&sink, 64)); CK(hipMemset(sink, 0, 64));
  unsigned* counter; CK(hipMalloc(&counter, 64)); CK(hipMemset(counter, 0, 64));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);

  printf("--- 1. one launch, gate_up's weights (180 MB): rate vs workgroups and ring depth\n");
  run_one<128, 3>("gu", WA, NGU, KGU, sink);
  run_one<128, 4>("gu", WA, NGU, KGU, sink);
  run_one<128, 6>("gu", WA, NGU, KGU, sink);
  run_one<128, 8>("gu", WA, NGU, KGU, sink);
  run_one<128, 10>("gu", WA, NGU, KGU, sink);
  run_one<96, 4>("gu", WA, NGU, KGU, sink);
  run_one<96, 7>("gu", WA, NGU, KGU, sink);
  run_one<96, 10>("gu", WA, NGU, KGU, sink);
  run_one<96, 13>("gu", WA, NGU, KGU, sink);
  run_one<64, 5>("gu", WA, NGU, KGU, sink);
  run_one<64, 9>("gu", WA, NGU, KGU, sink);
  run_one<64, 10>("gu", WA, NGU, KGU, sink);
  run_one<64, 17>("gu", WA, NGU, KGU, sink);
  run_one<32, 5>("gu", WA, NGU, KGU, sink);
  run_one<32, 9>("gu", WA, NGU, KGU, sink);
  run_one<32, 17>("gu", WA, NGU, KGU, sink);
  printf("--- the other projections, one launch each\n");
  run_one<32, 9>("down", WB, ND, KD, sink);             // 128 WGs
  run_one<32, 17>("down", WB, ND, KD, sink);
  run_one<16, 17>("down", WB, ND, KD, sink);            // 256 WGs
  run_one<16, 31>("down", WB, ND, KD, sink);

  printf("--- 2. gate_up then down: two launches vs one persistent launch with a grid barrier (256 WGs: 96 + 16 rows each; 230 / 256 of them stream real rows)\n");
  {
    constexpr int RA = 96, RB = 16, SA = 8, SB = 31;
    const int ldsA = SA * RA * 128, ldsB = SB * RB * 128;
    CK(hipFuncSetAttribute((const void*)one_stream<RA, SA>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsA));
    CK(hipFuncSetAttribute((const void*)one_stream<RB, SB>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsB));
    float us2 = time_it(40, [&](int i) {
      hipLaunchKernelGGL((one_stream<RA, SA>), dim3((NGU + RA - 1) / RA), dim3(256), ldsA, 0, WA[i % NBUF], NGU, KGU, sink);
      hipLaunchKernelGGL((one_stream<RB, SB>), dim3((ND + RB - 1) / RB), dim3(256), ldsB, 0, WB[i % NBUF], ND, KD, sink);
    });
    printf("two launches            : %6.1f us  %5.2f TB/s\n", us2, ((double)NGU * KGU + (double)ND * KD) * 2 / us2 / 1e6);
    unsigned epoch = 0;
    const int G = 256;
    auto fused = [&](auto kern, const char* nm, int lds) {
      CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, lds));
      if (occ * prop.multiProcessorCount < G) { printf("%s: grid does not fit (%d x %d)\n", nm, occ, prop.multiProcessorCount); return; }
      float us = time_it(40, [&](int i) { ++epoch; hipLaunchKernelGGL(kern, dim3(G), dim3(256), lds, 0, WA[i % NBUF], NGU, KGU, WB[i % NBUF], ND, KD, counter, epoch, sink); });
      printf("%-24s: %6.1f us  %5.2f TB/s\n", nm, us, ((double)NGU * KGU + (double)ND * KD) * 2 / us / 1e6);
    };
    // rows past N are clamped to the last row (the same line re-read from L2): 256 x 96 = 24576 >= 22016 for gate_up; 256 x 16 = 4096 for down
    fused(two_streams<RA, RB, SA, SB, false>, "persistent, no prefetch", (SA * RA) * 128);
    fused(two_streams<RA, RB, SA, SB, true>, "persistent, prefetch", (SA * RA + SB * RB) * 128);
    unsigned h[16]; CK(hipMemcpy(h, sink, 64, hipMemcpyDeviceToHost));
    if (h[1] == 0xdead) printf("GRID BARRIER TIMED OUT in some launch\n");
  }
  return 0;
}
