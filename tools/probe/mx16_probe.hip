// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 operands, unit E8M0 scales) on gfx950 -- tuning aid, not part of the product:
//   operand / result lane maps checked with exact small-integer data against a host product.  Assumed: lane l holds row (A) / column (B)
//   l & 15, bytes b = 0..31 <-> k = 32 * (l >> 4) + b; C/D as the other 16x16 forms: col = l & 15, row = 4 * (l >> 4) + reg.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/probe/mx16_probe.hip -o /tmp/mx16_probe && /tmp/mx16_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

static float e4m3_to_float(uint8_t v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f;
  if (e == 0) f = ldexpf((float)m, -9);
  else if (e == 15 && m == 7) f = NAN;
  else f = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -f : f;
}

__global__ void one_mfma(const v8i* a, const v8i* b, v4f* c, int scale_a, int scale_b) {
  v4f acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, scale_a, 0, scale_b);
  c[threadIdx.x] = acc;
}

int main() {
  std::vector<uint8_t> A(16 * 128), B(16 * 128);        // A[row][k], B[col][k] as e4m3 bytes
  srand(7);
  const uint8_t vals[] = {0x00, 0x30, 0x38, 0x3c, 0x40, 0x44, 0xb0, 0xb8, 0xc0, 0x28, 0xa8, 0x48};   // 0, .5, 1, 1.5, 2, 3, -.5, -1, -2, .25, -.25, 4
  for (auto& x : A) x = vals[rand() % 12];
  for (auto& x : B) x = vals[rand() % 12];
  std::vector<float> ref(16 * 16, 0.f);
  for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) { float s = 0; for (int k = 0; k < 128; ++k) s += e4m3_to_float(A[r * 128 + k]) * e4m3_to_float(B[c * 128 + k]); ref[r * 16 + c] = s; }
  std::vector<uint8_t> la(64 * 32), lb(64 * 32);
  for (int l = 0; l < 64; ++l) for (int b = 0; b < 32; ++b) { la[l * 32 + b] = A[(l & 15) * 128 + 32 * (l >> 4) + b]; lb[l * 32 + b] = B[(l & 15) * 128 + 32 * (l >> 4) + b]; }
  void *da, *db, *dc;
  hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dc, 64 * 16);
  hipMemcpy(da, la.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, lb.data(), 2048, hipMemcpyHostToDevice);
  int fails = 0;
  for (int sc = 0; sc < 3; ++sc) {
    const int sa = sc == 1 ? 0x80808080 : 0x7f7f7f7f, sb = sc == 2 ? 0x7e7e7e7e : 0x7f7f7f7f;
    one_mfma<<<1, 64>>>((const v8i*)da, (const v8i*)db, (v4f*)dc, sa, sb);
    std::vector<float> out(64 * 4);
    hipMemcpy(out.data(), dc, 64 * 16, hipMemcpyDeviceToHost);
    const float want_scale = sc == 0 ? 1.f : sc == 1 ? 2.f : 0.5f;
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int reg = 0; reg < 4; ++reg) {
      int col = l & 15, row = 4 * (l >> 4) + reg;
      if (out[l * 4 + reg] != want_scale * ref[row * 16 + col]) { if (bad < 4) printf("  mismatch lane %d reg %d: got %g want %g\n", l, reg, out[l * 4 + reg], want_scale * ref[row * 16 + col]); ++bad; }
    }
    printf("16x16x128 lane-map check (scale_a %08x scale_b %08x, expect x%g): %s (%d of 256 differ)\n", sa, sb, want_scale, bad ? "FAIL" : "ok", bad);
    fails += bad;
  }
  return fails ? 1 : 0;
}
