#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* a, float* o) {
  unsigned w = a[threadIdx.x];
  float c = 10.f;
  c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), __builtin_bit_cast(bf16x2_t, 0x3f803f80u), c, false);
  o[threadIdx.x] = c;
}
int main() {
  unsigned h[4] = {0x40203fc0u /* (1.5, 2.5) */, 0x3f803f80u, 0xc0004000u /* (2,-2) */, 0x00003f80u};
  unsigned* d; float* o; float r[4];
  hipMalloc(&d, 16); hipMalloc(&o, 16); hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, d, o);
  hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
  printf("%g %g %g %g (expect 14 12 10 11)\n", r[0], r[1], r[2], r[3]);
  return 0;
}
