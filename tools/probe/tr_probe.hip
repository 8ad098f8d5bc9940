// What does ds_read_b64_tr_b16 deliver?  LDS image [64 rows][row stride RS bytes] of u16 = row*256 + col.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__global__ void probe(uint16_t* out, int RS) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 320];
  for (int i = threadIdx.x; i < 64 * 128; i += 64) { int r = i / 128, c = i % 128; *(uint16_t*)(lds + r * RS + c * 2) = (uint16_t)(r * 256 + c); }
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, lq = lane & 15;
  unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)lds;
  unsigned addr = base + (4 * g + (lq >> 2)) * RS + (lq & 3) * 8;
  u32x2_t v0, v1;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "=v"(v0) : "v"(addr));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:32" : "=v"(v1) : "v"(addr));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  uint16_t* o = out + lane * 8;
  o[0] = v0[0] & 0xffff; o[1] = v0[0] >> 16; o[2] = v0[1] & 0xffff; o[3] = v0[1] >> 16;
  o[4] = v1[0] & 0xffff; o[5] = v1[0] >> 16; o[6] = v1[1] & 0xffff; o[7] = v1[1] >> 16;
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 8 * 2);
  for (int RS : {256, 288}) {
    probe<<<1, 64>>>(d, RS);
    uint16_t h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("RS=%d\n", RS);
    for (int lane : {0, 1, 5, 15, 16, 17, 37, 63}) {
      printf(" lane %2d:", lane);
      for (int j = 0; j < 8; ++j) printf(" (r%d,c%d)", h[lane * 8 + j] >> 8, h[lane * 8 + j] & 255);
      printf("\n");
    }
  }
  return 0;
}
