// Probe (not part of the product): lane / element mapping of ds_read_b64_tr_b16 on gfx950.
// LDS holds bf16 value = its own element index (0 .. 4095 fits bf16 exactly up to 256 -- use uint16 codes instead and print raw).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint16_t *out, int pitch_elems) {
  __shared__ __attribute__((aligned(16))) uint16_t img[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) img[i] = static_cast<uint16_t>(i);
  __syncthreads();
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  // each lane points at 4 contiguous elements: row (4 g' + (i >> 2)), cols 4 (i & 3) .. +3 of a [rows][pitch] image; here g' = g
  const unsigned addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>(img)) + ((4 * g + (i >> 2)) * pitch_elems + 4 * (i & 3)) * 2;
  uint64_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = static_cast<uint16_t>(v >> (16 * j));
}
int main() {
  uint16_t *d; hipMalloc(&d, 64 * 4 * 2);
  for (int pitch : {16, 64}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pitch);
    uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pitch %d elements: lane -> 4 values as (row, col) of the [rows][pitch] image\n", pitch);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d (i=%2d g=%d):", l, l & 15, l >> 4);
      for (int j = 0; j < 4; ++j) printf(" (%d,%d)", h[l * 4 + j] / pitch, h[l * 4 + j] % pitch);
      printf("\n");
      if ((l & 15) == 3 && l > 16) l += 12;   // print lanes 0-19, then 4 of each further group
    }
  }
  return 0;
}
