// Measured-peak probes for bench.py ("nominal peaks must be re-measured on the box"): a register-only bf16 MFMA loop and a
// read-only HBM stream, both timed with HIP events on the caller's stream.  Measurement hooks; no reference counterpart.
#include "common.h"
#include "internal.h"
#include <algorithm>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 probe_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float probe_f32x4_t;
typedef unsigned int probe_u32x4_t __attribute__((ext_vector_type(4)));

// 8 waves per workgroup, 4 x 8 accumulator tiles per wave (the ring GEMM's register shape), operands of moderate random magnitude
// (all-zero operands draw less power and clock higher: profiles/README.md), no memory traffic inside the loop.
__global__ __launch_bounds__(512, 1) void probe_mfma_kernel(float* __restrict__ out, int iters, unsigned seed) {
  probe_u32x4_t a[4], b[8];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    probe_u32x4_t v;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const unsigned h = ats_hash_u32((threadIdx.x * 12 + i) * 4 + c, seed);
      v[c] = (h & 0x007f007fu) | 0x3c003c00u | ((h >> 8) & 0x03800380u);          // two bf16 in [2^-7, 2^0)
    }
    if (i < 4) a[i] = v; else b[i - 4] = v;
  }
  probe_f32x4_t acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = probe_f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(probe_bf16x8_t, a[i]), __builtin_bit_cast(probe_bf16x8_t, b[j]),
                                                            acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[i][j][0];
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

// every thread: 8 independent non-temporal 16-byte loads per trip (the verify scan's access shape).  CONTIG = false: grid-stride over the
// whole buffer; true: every workgroup streams its own contiguous span (how the scan walks a logit row: better DRAM page locality)
template <bool CONTIG>
__global__ __launch_bounds__(256) void probe_read_kernel(const probe_u32x4_t* __restrict__ p, size_t n_vec, unsigned* __restrict__ sink) {
  probe_u32x4_t x = {0, 0, 0, 0};
  size_t i, end, stride;
  if constexpr (CONTIG) {
    const size_t span = (n_vec + gridDim.x - 1) / gridDim.x;
    i = (size_t)blockIdx.x * span + threadIdx.x; end = min((size_t)(blockIdx.x + 1) * span, n_vec); stride = 256;
  } else {
    i = (size_t)blockIdx.x * 256 + threadIdx.x; end = n_vec; stride = (size_t)gridDim.x * 256;
  }
  for (; i + 7 * stride < end; i += 8 * stride) {
    probe_u32x4_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) x ^= v[u];
  }
  for (; i < end; i += stride) x ^= __builtin_nontemporal_load(p + i);
  if ((x[0] ^ x[1] ^ x[2] ^ x[3]) == 0x9e3779b1u) sink[0] = 1;     // keeps the loads alive
}

struct EventPair {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ~EventPair() { if (e0) hipEventDestroy(e0); if (e1) hipEventDestroy(e1); }
};

}  // namespace

extern "C" int atspeed_probe_mfma_bf16(int32_t iters, void* scratch_dev, size_t scratch_bytes, void* stream, double* tflops_out) {
  const int blocks = 256 * 4;
  ATS_REQUIRE(iters >= 1 && scratch_dev && tflops_out, ATSPEED_ERR_INVALID, "probe_mfma: bad argument");
  ATS_REQUIRE(scratch_bytes >= (size_t)blocks * 512 * sizeof(float), ATSPEED_ERR_CAPACITY, "probe_mfma: scratch needs %zu bytes",
              (size_t)blocks * 512 * sizeof(float));
  hipStream_t st = (hipStream_t)stream;
  EventPair ev;
  ATS_HIP(hipEventCreate(&ev.e0)); ATS_HIP(hipEventCreate(&ev.e1));
  probe_mfma_kernel<<<blocks, 512, 0, st>>>((float*)scratch_dev, iters / 8 + 1, 1u);       // warm-up (clock ramp)
  ATS_HIP(hipEventRecord(ev.e0, st));
  probe_mfma_kernel<<<blocks, 512, 0, st>>>((float*)scratch_dev, iters, 2u);
  ATS_HIP(hipEventRecord(ev.e1, st));
  ATS_LAUNCH_CHECK();
  ATS_HIP(hipEventSynchronize(ev.e1));
  float ms = 0.f;
  ATS_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
  *tflops_out = (double)blocks * 8 * iters * 32 * 16384.0 / (ms * 1e-3) / 1e12;    // per wave and trip: 32 MFMAs of 2*16*16*32 flop
  return ATSPEED_OK;
}

extern "C" int atspeed_probe_hbm_read(const void* buf_dev, size_t bytes, int32_t reps, void* scratch_dev, void* stream, double* gbs_out) {
  ATS_REQUIRE(buf_dev && scratch_dev && gbs_out && reps >= 1 && bytes >= (1u << 20) && ((uintptr_t)buf_dev & 15) == 0, ATSPEED_ERR_INVALID,
              "probe_hbm: bad argument");
  hipStream_t st = (hipStream_t)stream;
  EventPair ev;
  ATS_HIP(hipEventCreate(&ev.e0)); ATS_HIP(hipEventCreate(&ev.e1));
  const size_t n_vec = bytes / 16;
  double best = 0.0;
  for (int variant = 0; variant < 2; ++variant)
    for (int per_cu : {4, 8, 16, 32, 64}) {                // resident workgroups per CU; the best (shape, grid) is the measured peak
      const int blocks = 256 * per_cu;
      auto launch = [&]() {
        if (variant) probe_read_kernel<true><<<blocks, 256, 0, st>>>((const probe_u32x4_t*)buf_dev, n_vec, (unsigned*)scratch_dev);
        else         probe_read_kernel<false><<<blocks, 256, 0, st>>>((const probe_u32x4_t*)buf_dev, n_vec, (unsigned*)scratch_dev);
      };
      launch();
      ATS_HIP(hipEventRecord(ev.e0, st));
      for (int r = 0; r < reps; ++r) launch();
      ATS_HIP(hipEventRecord(ev.e1, st));
      ATS_LAUNCH_CHECK();
      ATS_HIP(hipEventSynchronize(ev.e1));
      float ms = 0.f;
      ATS_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
      best = std::max(best, (double)n_vec * 16 * reps / (ms * 1e-3) / 1e9);
    }
  *gbs_out = best;
  return ATSPEED_OK;
}
