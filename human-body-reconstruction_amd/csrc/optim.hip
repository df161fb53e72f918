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

// the same over up to four (p, g, m, v) segments in ONE launch - the tables and the MLP block of a training step, each
// with its own learning rate / decay: blocks [first[k], first[k+1]) sweep segment k
struct AdamSegs {
  float *p[4], *m[4], *v[4];
  const float* g[4];
  int64_t n[4];
  AdamArgs a[4];
  int first[5];
};
__global__ __launch_bounds__(256) void adam_multi_kernel(AdamSegs s) {
  int k = 0;
#pragma unroll
  for (int j = 1; j < 4; ++j) k += (int)blockIdx.x >= s.first[j] ? 1 : 0;
  const int64_t block = (int64_t)blockIdx.x - s.first[k], nblocks = s.first[k + 1] - s.first[k];
  float *p = s.p[k], *m = s.m[k], *v = s.v[k];
  const float* g = s.g[k];
  const AdamArgs a = s.a[k];
  const int64_t n = s.n[k], n4 = n >> 2, stride = nblocks * 256;
  for (int64_t i = block * 256 + threadIdx.x; i < n4; i += stride) {
    float4 P = ((float4*)p)[i], G = ((const float4*)g)[i], M = ((float4*)m)[i], V = ((float4*)v)[i];
    adam1(P.x, G.x, M.x, V.x, a); adam1(P.y, G.y, M.y, V.y, a);
    adam1(P.z, G.z, M.z, V.z, a); adam1(P.w, G.w, M.w, V.w, a);
    ((float4*)p)[i] = P; ((float4*)m)[i] = M; ((float4*)v)[i] = V;
  }
  for (int64_t i = (n4 << 2) + block * 256 + threadIdx.x; i < n; i += stride) adam1(p[i], g[i], m[i], v[i], a);
}

static AdamArgs make_args(float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale) {
  AdamArgs a;
  a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.wd = weight_decay; a.gscale = grad_scale;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  return a;
}

}  // namespace hbr

using namespace hbr;

extern "C" int hbr_adam_step_multi(int nseg, const HbrAdamSegment* segs_host, void* stream) {
  if (nseg < 1 || nseg > 4 || !segs_host) return HBR_EINVAL;
  AdamSegs s{};
  int blocks = 0;
  for (int k = 0; k < 4; ++k) {
    s.first[k] = blocks;
    if (k >= nseg) continue;
    const HbrAdamSegment& h = segs_host[k];
    if (!h.p || !h.g || !h.m || !h.v || h.n < 0 || h.step < 1) return HBR_EINVAL;
    if ((((uintptr_t)h.p | (uintptr_t)h.g | (uintptr_t)h.m | (uintptr_t)h.v) & 15) != 0) return HBR_EINVAL;  // float4 path
    s.p[k] = h.p; s.g[k] = h.g; s.m[k] = h.m; s.v[k] = h.v; s.n[k] = h.n;
    s.a[k] = make_args(h.lr, h.beta1, h.beta2, h.eps, h.weight_decay, h.step, h.grad_scale);
    int64_t b = ((h.n >> 2) + 255) / 256;
    b = b > 2048 ? 2048 : (b < 1 ? 1 : b);
    blocks += (int)b;
  }
  s.first[4] = blocks;
  for (int k = nseg; k < 4; ++k) s.first[k] = blocks;  // empty segments own no block
  hipLaunchKernelGGL(adam_multi_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, s);
  HBR_RETURN_IF_LAUNCH_FAILED();
  return HBR_OK;
}

extern "C" int hbr_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int64_t step, float grad_scale, void* stream) {
  if (!p || !g || !m || !v || n < 0 || step < 1) return HBR_EINVAL;
  if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) != 0) return HBR_EINVAL;  // float4 path
  if (n == 0) return HBR_OK;
  const AdamArgs a = make_args(lr, beta1, beta2, eps, weight_decay, step, grad_scale);
  int64_t blocks = ((n >> 2) + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, a);
  HBR_RETURN_IF_LAUNCH_FAILED();
  return HBR_OK;
}
