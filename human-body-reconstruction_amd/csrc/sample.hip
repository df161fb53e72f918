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

// ------------------------------------------------------------------------------------------------
// a13 / f4: the second pass's resampling (hierarchical_sampling, reference helper.py:23-51) - one wave per ray
// ------------------------------------------------------------------------------------------------
namespace hbr {

constexpr int kResampleWaves = 4;

// uniform draw k of stream `which` (1: the per-ray u, 2: the shared sample vector) from (seed, offset)
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t offset, uint64_t k, uint32_t which) {
  const uint4 r = philox4x32_10(make_uint4((uint32_t)k, (uint32_t)offset, (uint32_t)(offset >> 32), which + ((uint32_t)(k >> 32) << 4)),
                                make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
  return (float)(r.x >> 8) * 5.9604644775390625e-8f;
}

// number of entries of the ascending a[0..n) that are < x (strict == false: <= x)
template <bool STRICT>
__device__ __forceinline__ int count_below(const float* a, int n, float x) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const bool below = STRICT ? a[mid] < x : a[mid] <= x;
    lo = below ? mid + 1 : lo;
    hi = below ? hi : mid;
  }
  return lo;
}

// Per ray (helper.py:36-47):  w = max(weights, 0);  pdf = (w + 1e-5) / sum(w + 1e-5);  cdf = cumsum(pdf) - both
// sequential fp32 sums by one lane, as torch's CPU cumsum is (the sampler is discontinuous in the last bit of the cdf);
// inds [S] = clamp(searchsorted(cdf, u, right=True), 0, n - 1);  new depth j = samples[inds[j]] where samples [n] =
// samples01 * (tf - tn) + tn is ONE vector shared by all rays (the reference's quirk: the new depths are not drawn
// inside the selected bins);  t_fine = sort(cat(z_vals, new depths)).  The sort: the n new depths are ranked among
// themselves by counting (S^2 / 64 compares per lane), then merged with the ascending z_vals by binary searches;
// z_vals that are not ascending take an all-pairs ranking of the 2S values instead.  (A ray gets S new depths - the
// draws have cdf's shape - whatever n_samples is; n_samples is the length of the shared vector and the clamp.)
__global__ __launch_bounds__(kResampleWaves * 64) void resample_kernel(const float* __restrict__ weights, const float* __restrict__ z,
                                                                       int64_t z_stride, const float* __restrict__ u, const float* __restrict__ samples01,
                                                                       uint64_t seed, uint64_t offset, float tn, float span, int64_t R, int S, int n,
                                                                       float* __restrict__ t_fine) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * kResampleWaves + wv;
  if (r >= R) return;
  const int M = 2 * S;  // one new depth per draw, and there are S draws per ray (u has cdf's shape, helper.py:40)
  float* cdf = lds + (size_t)wv * (4 * (size_t)S + n);  // [S]
  float* zs = cdf + S;                                   // [S] this ray's z_vals
  float* nv = zs + S;                                    // [S] new depths, unsorted
  float* ns = nv + S;                                    // [S] new depths, ascending
  float* smp = ns + S;                                   // [n] the shared sample vector
  const float* wr = weights + r * S;
  const float* zr = z + r * z_stride;
  for (int s = lane; s < S; s += 64) {
    const float w = wr[s];
    cdf[s] = __fadd_rn(w < 0.f ? 0.f : w, 1e-5f);  // helper.py:36,38
    zs[s] = zr[s];
  }
  for (int k = lane; k < n; k += 64) {
    const float s01 = samples01 ? samples01[k] : philox_uniform(seed, offset, (uint64_t)k, 2u);
    smp[k] = __fadd_rn(__fmul_rn(s01, span), tn);  // helper.py:43
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    float tot = 0.f;
    for (int s = 0; s < S; ++s) tot = __fadd_rn(tot, cdf[s]);
    float c = 0.f;
    for (int s = 0; s < S; ++s) {
      c = __fadd_rn(c, __fdiv_rn(cdf[s], tot));  // :38-39
      cdf[s] = c;
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  bool ascending = true;
  for (int s = lane; s + 1 < S; s += 64) ascending = ascending && zs[s] <= zs[s + 1];
  ascending = __all(ascending);
  for (int j = lane; j < S; j += 64) {
    const float uj = u ? u[r * S + j] : philox_uniform(seed, offset, (uint64_t)(r * S + j), 1u);
    int ind = count_below<false>(cdf, S, uj);  // searchsorted(right=True): entries <= u  (:41)
    ind = ind > n - 1 ? n - 1 : ind;           // :44
    nv[j] = smp[ind];                          // :45
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  float* out = t_fine + r * M;
  if (ascending) {
    for (int j = lane; j < S; j += 64) {  // rank among the new depths (ties by index)
      const float x = nv[j];
      int rank = 0;
      for (int k = 0; k < S; ++k) {
        const float y = nv[k];
        rank += (y < x || (y == x && k < j)) ? 1 : 0;
      }
      ns[rank] = x;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < S; i += 64) out[i + count_below<true>(ns, S, zs[i])] = zs[i];    // z first on ties
    for (int q = lane; q < S; q += 64) out[q + count_below<false>(zs, S, ns[q])] = ns[q];
  } else {
    for (int i = lane; i < M; i += 64) {
      const float x = i < S ? zs[i] : nv[i - S];
      int rank = 0;
      for (int k = 0; k < M; ++k) {
        const float y = k < S ? zs[k] : nv[k - S];
        rank += (y < x || (y == x && k < i)) ? 1 : 0;
      }
      out[rank] = x;
    }
  }
}

// Volume_Renderer.update_grid (vol_renderer.py:116-131).  The reference writes tmp_arr[cell] += ceil(alpha) through an
// index_put WITHOUT accumulation into an int8 array: when several points share a cell the LAST one (in point order, on the
// CPU) decides, and its ceil(alpha) is wrapped to int8 - a cell is marked iff that wrapped value is > 0; alpha <= 0
// counts as 0.  Pass 1 finds each touched cell's last point (atomicMax of the point index), pass 2 lets exactly that
// point mark the cell (bool_grid[cell] = True) and counts the marks; if nothing was marked the whole grid becomes True
// (:126-127).  Cells are only ever set, never cleared.  `tmp` (optional) is the reference's tmp_arr, carried between
// calls: positive entries are reset (:131), a count that wrapped negative is not, and is added to the next call's.
// Indices outside [-G, G) - where torch raises - are skipped.
__device__ __forceinline__ int64_t grid_cell(const PointSrc& ps, const HashGeom& g, uint32_t n, int G) {
  float px, py, pz, nx, ny, nz;
  load_point(ps, n, px, py, pz);
  normalise(g, px, py, pz, nx, ny, nz);
  const float fG = (float)G;
  int c[3] = {(int)__fmul_rn(nx, fG), (int)__fmul_rn(ny, fG), (int)__fmul_rn(nz, fG)};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (c[a] < 0) c[a] += G;
    if (c[a] < 0 || c[a] >= G) return -1;
  }
  return ((int64_t)c[0] * G + c[1]) * G + c[2];
}
__global__ __launch_bounds__(256) void grid_last_point_kernel(PointSrc ps, uint32_t N, HashGeom g, int G, int* __restrict__ winner) {
  const uint32_t n = blockIdx.x * 256u + threadIdx.x;
  if (n >= N) return;
  const int64_t cell = grid_cell(ps, g, n, G);
  if (cell >= 0) atomicMax(winner + cell, (int)n);
}
__global__ __launch_bounds__(256) void grid_mark_kernel(PointSrc ps, uint32_t N, HashGeom g, int G, const int* __restrict__ winner,
                                                        const float* __restrict__ alpha, uint8_t* __restrict__ grid,
                                                        int8_t* __restrict__ tmp, unsigned* __restrict__ marked) {
  const uint32_t n = blockIdx.x * 256u + threadIdx.x;
  bool mark = false;
  if (n < N) {
    const int64_t cell = grid_cell(ps, g, n, G);
    if (cell >= 0 && winner[cell] == (int)n) {
      const float a = alpha[n];
      const int v = a <= 0.f ? 0 : (int)ceilf(a);               // alpha[alpha <= 0] = 0; ceil(alpha).int()
      const int old = tmp ? (int)tmp[cell] : 0;
      const int8_t now = (int8_t)(uint8_t)((old + v) & 0xff);  // int32 sum stored back through the int8 array: wraps
      mark = now > 0;
      if (mark) grid[cell] = 1;
      if (tmp) tmp[cell] = mark ? (int8_t)0 : now;              // tmp_arr[tmp_arr > 0] = 0: a wrapped (negative) count stays

    }
  }
  const unsigned long long b = __ballot(mark);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(marked, (unsigned)__popcll(b));
}
__global__ __launch_bounds__(256) void grid_fill_if_none_kernel(const unsigned* __restrict__ marked, uint8_t* __restrict__ grid, size_t cells) {
  if (*marked != 0u) return;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < cells; i += (size_t)gridDim.x * 256) grid[i] = 1;
}

}  // namespace hbr

extern "C" int hbr_hierarchical_resample(const float* weights, const float* z_vals, int64_t z_stride, const float* u,
                                         const float* samples01, uint64_t seed, uint64_t offset, float tn, float tf, int64_t R,
                                         int64_t S, int64_t n_samples, float* t_fine, void* stream) {
  if (!weights || !z_vals || !t_fine || R < 0 || S < 1 || n_samples < 1) return HBR_EINVAL;
  if (z_stride != 0 && z_stride < S) return HBR_EINVAL;
  const int64_t lds = (4 * S + n_samples) * (int64_t)sizeof(float) * kResampleWaves;
  if (lds > 160 * 1024) return HBR_EUNSUPPORTED;  // up to ~2000 samples per ray
  if (R == 0) return HBR_OK;
  const int64_t blocks = (R + kResampleWaves - 1) / kResampleWaves;
  if (blocks > 0x7fffffffLL) return HBR_EUNSUPPORTED;
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)resample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return HBR_ELAUNCH;
  // (tf - tn) is a python-float difference in the reference, turned into an fp32 scalar when it meets the tensor
  hipLaunchKernelGGL(resample_kernel, dim3((uint32_t)blocks), dim3(kResampleWaves * 64), (size_t)lds, (hipStream_t)stream, weights, z_vals,
                     z_stride, u, samples01, seed, offset, tn, (float)((double)tf - (double)tn), R, (int)S, (int)n_samples, t_fine);
  HBR_RETURN_IF_LAUNCH_FAILED();
  return HBR_OK;
}

extern "C" int64_t hbr_occupancy_update_workspace_bytes(int G) { return G < 1 ? 0 : (int64_t)G * G * G * 4 + 256; }

extern "C" int hbr_occupancy_update(const float* x, const float* rays_o, const float* rays_d, const float* t, int64_t R, int64_t S,
                                    const float* alpha, uint8_t* grid, int8_t* tmp_arr, int G, const float* mu_host, float sigma_val,
                                    void* ws, int64_t ws_bytes, void* stream) {
  if (!grid || !alpha || !mu_host || !ws || G < 1 || G > 1024) return HBR_EINVAL;
  if (ws_bytes < hbr_occupancy_update_workspace_bytes(G)) return HBR_EWORKSPACE;
  PointSrc ps;
  uint32_t N;
  int rc = check_points(x, rays_o, rays_d, t, R, S, ps, N);
  if (rc) return rc;
  HashGeom g{};
  g.mu[0] = mu_host[0]; g.mu[1] = mu_host[1]; g.mu[2] = mu_host[2];
  g.sigma = sigma_val;
  hipStream_t st = (hipStream_t)stream;
  const size_t cells = (size_t)G * G * G;
  unsigned* marked = (unsigned*)ws;
  int* winner = (int*)((char*)ws + 256);
  if (hipMemsetAsync(marked, 0, 256, st) != hipSuccess) return HBR_ELAUNCH;
  if (hipMemsetAsync(winner, 0xff, cells * 4, st) != hipSuccess) return HBR_ELAUNCH;  // -1: no point yet
  if (N > 0) {
    hipLaunchKernelGGL(grid_last_point_kernel, dim3((N + 255u) / 256u), dim3(256), 0, st, ps, N, g, G, winner);
    hipLaunchKernelGGL(grid_mark_kernel, dim3((N + 255u) / 256u), dim3(256), 0, st, ps, N, g, G, (const int*)winner, alpha, grid, tmp_arr, marked);
  }
  hipLaunchKernelGGL(grid_fill_if_none_kernel, dim3(1024), dim3(256), 0, st, (const unsigned*)marked, grid, cells);
  HBR_RETURN_IF_LAUNCH_FAILED();
  return HBR_OK;
}
