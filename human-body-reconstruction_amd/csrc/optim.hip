// a12: dense Adam / AdamW over a flat fp32 buffer (torch.optim.Adam/AdamW single-tensor semantics,
// reference train_hash2.py:141-142,227-228).  Pure HBM streaming: 16 B read of each of p,g,m,v and
// 16 B write of p,m,v per 4 elements, one float4 per lane.
#include "hbr_common.h"

namespace hbr {

struct AdamArgs {
  float lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale;
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamArgs& a) {
  g *= a.gscale;
  p = p * (1.f - a.lr * a.wd);              // decoupled decay (no-op when wd == 0)
  m = m + (1.f - a.b1) * (g - m);           // torch: exp_avg.lerp_(grad, 1-beta1)
  v = a.b2 * v + (1.f - a.b2) * g * g;      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  p = p - (a.lr / a.bc1) * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, AdamArgs a) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 P = ((float4*)p)[i], G = ((const float4*)g)[i], M = ((float4*)m)[i], V = ((float4*)v)[i];
    adam1(P.x, G.x, M.x, V.x, a); adam1(P.y, G.y, M.y, V.y, a);
    adam1(P.z, G.z, M.z, V.z, a); adam1(P.w, G.w, M.w, V.w, a);
    ((float4*)p)[i] = P; ((float4*)m)[i] = M; ((float4*)v)[i] = V;
  }
  // tail (n not a multiple of 4)
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) adam1(p[i], g[i], m[i], v[i], a);
}

}  // namespace hbr

using namespace hbr;

extern "C" int hbr_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int64_t step, float grad_scale, void* stream) {
  if (!p || !g || !m || !v || n < 0 || step < 1) return HBR_EINVAL;
  if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) != 0) return HBR_EINVAL;  // float4 path
  if (n == 0) return HBR_OK;
  AdamArgs a;
  a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.wd = weight_decay; a.gscale = grad_scale;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  int64_t blocks = ((n >> 2) + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, a);
  HBR_RETURN_IF_LAUNCH_FAILED();
  return HBR_OK;
}
