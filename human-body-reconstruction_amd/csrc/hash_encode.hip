// K1: multiresolution hash-grid encode, and the global-atomics form of its gradient scatter-add, gfx950.
//
// Forward: one thread = one point, 64 lanes = 64 consecutive samples of a ray, so at the coarse
// levels the lanes of a wave fall into a handful of cells and the texture path coalesces their
// gathers.  Levels are dealt to workgroups by blockIdx % 8: workgroups are dispatched round-robin
// over the 8 XCDs, so each XCD's private 4 MiB L2 only ever sees L/8 levels (1 MiB of fp32
// tables at L=16, T=2^16) instead of thrashing on all 8 MiB.  Placement is a speed assumption
// only; results do not depend on it.
//
// Backward, algo 1 (kept for small N and as a cross-check): one global float atomic per corner-feature.  The LDS
// algorithms that the trainer uses live in hash_scatter.hip.
#include <cstdlib>

#include "hash_common.h"

namespace hbr {

// ------------------------------------------------------------------------------------------------
// K1 forward
// ------------------------------------------------------------------------------------------------
template <bool POW2, int LAYOUT, int DTYPE>
__global__ __launch_bounds__(kFwdThreads) void hash_fwd_kernel(PointSrc ps, uint32_t N, const float* __restrict__ tables,
                                                               HashGeom g, void* __restrict__ y, int64_t y_stride,
                                                               int j_begin, int j_end, int mirror) {
  const int group = blockIdx.x % kXcds;
  const uint32_t tile = blockIdx.x / kXcds;
  const uint32_t n = tile * kFwdThreads + threadIdx.x;
  if (n >= N) return;

  float px, py, pz, nx, ny, nz;
  load_point(ps, n, px, py, pz);
  normalise(g, px, py, pz, nx, ny, nz);
#ifdef HBR_K1_ONLY_GROUP  // timing-only variant: one XCD group's work alone (profiles/r03_k1_level_costs.txt)
  if (group != HBR_K1_ONLY_GROUP) return;
#endif

  for (int j = j_begin; j < j_end; ++j) {
    const int l = group_level(group, j, tile, mirror != 0);
    if (l >= g.L) continue;
#ifdef HBR_K1_ONLY_LEVEL  // timing-only variant: one level's work alone
    if (l != HBR_K1_ONLY_LEVEL) continue;
#endif
    Cell c = locate(nx, ny, nz, g.scale[l]);
    uint32_t rows[8];
    float w[8];
    corner_rows<POW2>(g, c, rows);
    corner_weights(c, w);
    const float2* tab = (const float2*)tables + (size_t)l * g.T;
    float2 fv[8];
#ifdef HBR_K1_ABLATE_BELOW  // timing-only variant: levels below this take no table reads (upper bound of an LDS-staged tile)
    if (l < HBR_K1_ABLATE_BELOW) {
#pragma unroll
      for (int k = 0; k < 8; ++k) fv[k] = make_float2(__uint_as_float(rows[k]), w[k]);
    } else
#endif
#pragma unroll
    for (int k = 0; k < 8; ++k) fv[k] = tab[rows[k]];
    // (fv*w).sum(-2): products rounded, then added (hash_encoding.py:144)
    float a0 = __fmul_rn(fv[0].x, w[0]), a1 = __fmul_rn(fv[0].y, w[0]);
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      a0 = __fadd_rn(a0, __fmul_rn(fv[k].x, w[k]));
      a1 = __fadd_rn(a1, __fmul_rn(fv[k].y, w[k]));
    }
    store_feat<LAYOUT, DTYPE>(y, n, l, N, y_stride, a0, a1);
  }
}

// ------------------------------------------------------------------------------------------------
// K2 backward, algo 1: global float atomics
// ------------------------------------------------------------------------------------------------
template <bool POW2, int LAYOUT, int DTYPE>
__global__ __launch_bounds__(kFwdThreads) void hash_bwd_atomic_kernel(PointSrc ps, uint32_t N, const void* __restrict__ dy,
                                                                      int64_t dy_stride, HashGeom g,
                                                                      float* __restrict__ dtables, int levels_per_group) {
  const int group = blockIdx.x % kXcds;
  const uint32_t tile = blockIdx.x / kXcds;
  const uint32_t n = tile * kFwdThreads + threadIdx.x;
  if (n >= N) return;

  float px, py, pz, nx, ny, nz;
  load_point(ps, n, px, py, pz);
  normalise(g, px, py, pz, nx, ny, nz);

  for (int j = 0; j < levels_per_group; ++j) {
    const int l = group_level(group, j, tile);
    if (l >= g.L) continue;
    float d0, d1;
    load_feat<LAYOUT, DTYPE>(dy, n, l, N, dy_stride, d0, d1);
    Cell c = locate(nx, ny, nz, g.scale[l]);
    uint32_t rows[8];
    float w[8];
    corner_rows<POW2>(g, c, rows);
    corner_weights(c, w);
    float* tab = dtables + (size_t)l * g.T * 2;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      unsafeAtomicAdd(tab + (size_t)rows[k] * 2 + 0, __fmul_rn(w[k], d0));
      unsafeAtomicAdd(tab + (size_t)rows[k] * 2 + 1, __fmul_rn(w[k], d1));
    }
  }
}

// The mirrored level map (hash_common.h) balances the XCDs but gives each of them FOUR level tables; it pays while those fit
// the XCD's 4 MiB L2 (T <= 2^17 at L = 16).  Beyond, the plain map (two tables per XCD) is faster - measured (hash_fwd, ms;
// gpurun_out/k1_split.txt) at T = 2^16 .. 2^20: mirrored 0.250 / 0.256 / 0.369 / 0.465 / 0.567, plain 0.281 / 0.286 / 0.292 /
// 0.397 / 0.552.  (Also measured, and worse everywhere: one launch per level index so that an XCD works on one level at a
// time - 0.33 - 0.65 ms: the points are regenerated per launch and each launch has half the work in flight.)
// HBR_K1_MIRROR_RT=0/1: tuning override.
template <bool POW2, int LAYOUT>
static int launch_fwd_dtype(int dtype, dim3 grid, hipStream_t st, PointSrc ps, uint32_t N, const float* tables,
                            const HashGeom& g, void* y, int64_t stride, int lpg) {
  static const char* e_mirror = getenv("HBR_K1_MIRROR_RT");
  int mirror = g.T * 8 * lpg * 2 <= (4LL << 20) ? 1 : 0;
  if (e_mirror) mirror = atoi(e_mirror) != 0;
  if (dtype == HBR_F32)
    hipLaunchKernelGGL((hash_fwd_kernel<POW2, LAYOUT, HBR_F32>), grid, dim3(kFwdThreads), 0, st, ps, N, tables, g, y, stride, 0, lpg, mirror);
  else
    hipLaunchKernelGGL((hash_fwd_kernel<POW2, LAYOUT, HBR_BF16>), grid, dim3(kFwdThreads), 0, st, ps, N, tables, g, y, stride, 0, lpg, mirror);
  return HBR_OK;
}

int launch_hash_bwd_atomic(hipStream_t st, PointSrc ps, uint32_t N, const void* dy, int layout, int64_t dy_stride, int dy_dtype,
                           const HashGeom& g, float* dtables) {
  const int lpg = (g.L + kXcds - 1) / kXcds;
  const uint32_t tiles = (N + kFwdThreads - 1) / kFwdThreads;
  const dim3 grid(tiles * kXcds), block(kFwdThreads);
#define HBR_A1(P, LY, DT) hipLaunchKernelGGL((hash_bwd_atomic_kernel<P, LY, DT>), grid, block, 0, st, ps, N, dy, dy_stride, g, dtables, lpg)
  if (g.pow2) {
    if (layout == HBR_LAYOUT_PLANAR) { if (dy_dtype == HBR_F32) HBR_A1(true, HBR_LAYOUT_PLANAR, HBR_F32); else HBR_A1(true, HBR_LAYOUT_PLANAR, HBR_BF16); }
    else { if (dy_dtype == HBR_F32) HBR_A1(true, HBR_LAYOUT_ROWS, HBR_F32); else HBR_A1(true, HBR_LAYOUT_ROWS, HBR_BF16); }
  } else {
    if (layout == HBR_LAYOUT_PLANAR) { if (dy_dtype == HBR_F32) HBR_A1(false, HBR_LAYOUT_PLANAR, HBR_F32); else HBR_A1(false, HBR_LAYOUT_PLANAR, HBR_BF16); }
    else { if (dy_dtype == HBR_F32) HBR_A1(false, HBR_LAYOUT_ROWS, HBR_F32); else HBR_A1(false, HBR_LAYOUT_ROWS, HBR_BF16); }
  }
#undef HBR_A1
  return HBR_OK;
}

}  // namespace hbr

using namespace hbr;

extern "C" int hbr_hash_encode_fwd(const float* x, const float* rays_o, const float* rays_d, const float* t, int64_t R,
                                   int64_t S, const float* tables, const float* scales_host, const float* mu_host,
                                   float sigma, int L, int64_t T, int F, void* y, int layout, int64_t y_stride, int y_dtype,
                                   void* stream) {
  if (!tables || !y) return HBR_EINVAL;
  if (F != 2) return HBR_EUNSUPPORTED;
  if (layout != HBR_LAYOUT_ROWS && layout != HBR_LAYOUT_PLANAR) return HBR_EINVAL;
  if (y_dtype != HBR_F32 && y_dtype != HBR_BF16) return HBR_EINVAL;
  if (layout == HBR_LAYOUT_ROWS && y_stride < (int64_t)L * F) return HBR_EINVAL;
  HashGeom g;
  int rc = fill_geom(g, scales_host, mu_host, sigma, L, T);
  if (rc) return rc;
  PointSrc ps;
  uint32_t N;
  rc = check_points(x, rays_o, rays_d, t, R, S, ps, N);
  if (rc) return rc;
  if (N == 0) return HBR_OK;
  hipStream_t st = (hipStream_t)stream;
  const int lpg = (L + kXcds - 1) / kXcds;
  const uint32_t tiles = (N + kFwdThreads - 1) / kFwdThreads;
  dim3 grid(tiles * kXcds);
  if (g.pow2) {
    if (layout == HBR_LAYOUT_PLANAR) launch_fwd_dtype<true, HBR_LAYOUT_PLANAR>(y_dtype, grid, st, ps, N, tables, g, y, y_stride, lpg);
    else launch_fwd_dtype<true, HBR_LAYOUT_ROWS>(y_dtype, grid, st, ps, N, tables, g, y, y_stride, lpg);
  } else {
    if (layout == HBR_LAYOUT_PLANAR) launch_fwd_dtype<false, HBR_LAYOUT_PLANAR>(y_dtype, grid, st, ps, N, tables, g, y, y_stride, lpg);
    else launch_fwd_dtype<false, HBR_LAYOUT_ROWS>(y_dtype, grid, st, ps, N, tables, g, y, y_stride, lpg);
  }
  HBR_RETURN_IF_LAUNCH_FAILED();
  return HBR_OK;
}

