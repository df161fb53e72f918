// Development check of wave_prefix_sum / wave_suffix_sum (csrc/wave_reduce.h) against sequential sums on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../../human-body-reconstruction_amd/csrc/wave_reduce.h"
__global__ void k(const float* in, float* out) {  // out[6][N]
  const int i = blockIdx.x * 64 + threadIdx.x, N = gridDim.x * 64;
  const hbr::WaveScan a = hbr::wave_prefix_sum(in[i], threadIdx.x), b = hbr::wave_suffix_sum(in[i], threadIdx.x);
  out[i] = a.incl; out[N + i] = a.excl; out[2 * N + i] = a.total; out[3 * N + i] = b.incl; out[4 * N + i] = b.excl; out[5 * N + i] = b.total;
}
int main() {
  const int B = 64, N = B * 64;
  std::vector<float> h(N), o(6 * N);
  unsigned s = 777;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (float)((int)(s >> 8) - (1 << 23)) / 8388608.f; }
  float *d, *r;
  (void)hipMalloc(&d, N * 4); (void)hipMalloc(&r, 6 * N * 4);
  (void)hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(B), dim3(64), 0, 0, d, r);
  (void)hipMemcpy(o.data(), r, 6 * N * 4, hipMemcpyDeviceToHost);
  int bad = 0; double worst = 0;
  auto chk = [&](float got, double want) { const double e = fabs(got - want); if (e > worst) worst = e; if (e > 2e-5) ++bad; };
  for (int b = 0; b < B; ++b) {
    double tot = 0; for (int i = 0; i < 64; ++i) tot += h[b * 64 + i];
    double run = 0;
    for (int i = 0; i < 64; ++i) {
      const int j = b * 64 + i;
      chk(o[N + j], run); run += h[j]; chk(o[j], run); chk(o[2 * N + j], tot);
      chk(o[3 * N + j], tot - run + h[j]); chk(o[4 * N + j], tot - run); chk(o[5 * N + j], tot);
    }
  }
  printf("dpp scans: %d mismatches, worst abs error %.3g\n", bad, worst);
  return bad != 0;
}
