// K0: what vol_render does before the encoder sees a point (reference vol_renderer.py:163-175) -
//   * the stratified depths t[S] shared by all rays of a batch (strat_sampler, helper.py:210-237, non-exp branch),
//   * the occupancy-grid lookup of every sample (Volume_Renderer.get_mask, vol_renderer.py:133-140).
// The points themselves (o + d*t, vol_renderer.py:165) are generated inside the encoder kernels (hbr_common.h).
#include "hash_common.h"

namespace hbr {

// Philox4x32-10 (Salmon et al. 2011): a counter-based generator - the draw for (seed, offset, s) depends on nothing
// else, so every rank that uses the same seed and step gets the same jitter without any state to share.
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
    const uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
    ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
    key.x += W0; key.y += W1;
  }
  return ctr;
}

// t[s] = linspace(tn, tf, S)[s] + (u[s] * (tf - tn)) / S      (helper.py:234-235; one jitter per sample index)
// linspace as torch evaluates it in fp32: step = (tf - tn) / (S - 1); the lower half counts up from tn, the upper half
// down from tf.  u: the caller's uniform draw, or (u == nullptr) 24-bit uniforms in [0, 1) from Philox.
__global__ __launch_bounds__(256) void strat_sample_kernel(float tn, float tf, uint32_t S, const float* __restrict__ u,
                                                           uint64_t seed, uint64_t offset, float* __restrict__ t) {
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  if (s >= S) return;
  float us;
  if (u) {
    us = u[s];
  } else {
    const uint4 r = philox4x32_10(make_uint4(s, (uint32_t)offset, (uint32_t)(offset >> 32), 0u),
                                  make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    us = (float)(r.x >> 8) * 5.9604644775390625e-8f;  // 2^-24
  }
  const float span = __fsub_rn(tf, tn);
  float lin = tn;
  if (S > 1) {
    const float step = __fdiv_rn(span, (float)(S - 1));
    lin = s < S / 2 ? __fadd_rn(tn, __fmul_rn(step, (float)s)) : __fsub_rn(tf, __fmul_rn(step, (float)(S - 1 - s)));
  }
  t[s] = __fadd_rn(lin, __fdiv_rn(__fmul_rn(us, span), (float)S));
}

// keep[n] = grid[cx, cy, cz] with c = trunc(((p - mu) / sigma_val) * G) - the reference's three separately rounded
// fp32 ops and `.long()`.  A negative index counts from the end, as torch's indexing does; anything still outside
// [0, G) (where torch raises IndexError) is "not kept".
__global__ __launch_bounds__(256) void occupancy_mask_kernel(PointSrc ps, uint32_t N, HashGeom g, const uint8_t* __restrict__ grid,
                                                             int G, uint8_t* __restrict__ keep) {
  const uint32_t n = blockIdx.x * 256u + threadIdx.x;
  if (n >= N) return;
  float px, py, pz, nx, ny, nz;
  load_point(ps, n, px, py, pz);
  normalise(g, px, py, pz, nx, ny, nz);
  const float fG = (float)G;
  int c[3] = {(int)__fmul_rn(nx, fG), (int)__fmul_rn(ny, fG), (int)__fmul_rn(nz, fG)};
  bool in = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (c[a] < 0) c[a] += G;
    in = in && c[a] >= 0 && c[a] < G;
  }
  keep[n] = in ? grid[((size_t)c[0] * G + c[1]) * G + c[2]] : (uint8_t)0;
}

}  // namespace hbr

using namespace hbr;

extern "C" int hbr_strat_sample(float tn, float tf, int64_t S, const float* u, uint64_t seed, uint64_t offset, float* t,
                                void* stream) {
  if (!t || S < 0) return HBR_EINVAL;
  if (S > 0x7fffffffLL) return HBR_EUNSUPPORTED;
  if (S == 0) return HBR_OK;
  hipLaunchKernelGGL(strat_sample_kernel, dim3((uint32_t)((S + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tn, tf, (uint32_t)S, u,
                     seed, offset, t);
  HBR_RETURN_IF_LAUNCH_FAILED();
  return HBR_OK;
}

extern "C" int hbr_occupancy_mask(const float* x, const float* rays_o, const float* rays_d, const float* t, int64_t R, int64_t S,
                                  const uint8_t* grid, int G, const float* mu_host, float sigma_val, uint8_t* keep, void* stream) {
  if (!grid || !keep || !mu_host || G < 1 || G > 2048) return HBR_EINVAL;
  PointSrc ps;
  uint32_t N;
  int rc = check_points(x, rays_o, rays_d, t, R, S, ps, N);
  if (rc) return rc;
  if (N == 0) return HBR_OK;
  HashGeom g{};
  g.mu[0] = mu_host[0]; g.mu[1] = mu_host[1]; g.mu[2] = mu_host[2];
  g.sigma = sigma_val;
  hipLaunchKernelGGL(occupancy_mask_kernel, dim3((N + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, ps, N, g, grid, G, keep);
  HBR_RETURN_IF_LAUNCH_FAILED();
  return HBR_OK;
}
