// K0: what vol_render does before the encoder sees a point (reference vol_renderer.py:163-175) -
//   * the stratified depths t[S] shared by all rays of a batch (strat_sampler, helper.py:210-237, non-exp branch),
//   * the occupancy-grid lookup of every sample (Volume_Renderer.get_mask, vol_renderer.py:133-140).
// The points themselves (o + d*t, vol_renderer.py:165) are generated inside the encoder kernels (hbr_common.h).
#include "hash_common.h"
#include "sample_common.h"

namespace hbr {

__global__ __launch_bounds__(256) void strat_sample_kernel(StratArgs a) { strat_sample_one(a, blockIdx.x * 256u + threadIdx.x); }

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
  hipLaunchKernelGGL(strat_sample_kernel, dim3((uint32_t)((S + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     StratArgs{tn, tf, (uint32_t)S, u, seed, offset, t});
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
