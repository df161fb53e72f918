// K1 / K2: multiresolution hash-grid encode and its gradient scatter-add, gfx950.
//
// Forward: one thread = one point, 64 lanes = 64 consecutive samples of a ray, so at the coarse
// levels the lanes of a wave fall into a handful of cells and the texture path coalesces their
// gathers.  Levels are dealt to workgroups by blockIdx % 8: workgroups are dispatched round-robin
// over the 8 XCDs, so each XCD's private 4 MiB L2 only ever sees L/8 levels (1 MiB of fp32
// tables at L=16, T=2^16) instead of thrashing on all 8 MiB.  Placement is a speed assumption
// only; results do not depend on it.
//
// Backward: (algo 1) one float atomic per corner-feature; (algo 2) a workgroup owns one feature of a
// 16384-row slice of one level in LDS as fp64 accumulators (128 KiB), sweeps a chunk of the points,
// accumulates the corners that fall into its slice with ds_add_f64 and flushes the slice once with
// contiguous 256-B global float atomics (MI355X_MICROARCH "Global float atomics": contiguous atomics
// run 17x the one-row-per-lane rate).  fp64 because on gfx950 ds_add_f32 costs ~190 cycles per
// wave-instruction while ds_add_f64 costs ~21 (measured: tools/lds_atomic_bench.hip); as a bonus the
// 53-bit sums make the result independent of arrival order to well below one fp32 ulp.
#include "hbr_common.h"

namespace hbr {

constexpr int kFwdThreads = 256;
constexpr int kXcds = 8;

// j-th level of XCD group `group`: groups pair a coarse level (cheap: the wave's gathers coalesce) with a fine one
// (texture-rate bound) - {k, 15-k} for L = 16 - so that the 8 XCDs finish together; each XCD's L2 still only sees
// L/8 levels.
__device__ __forceinline__ int group_level(int group, int j) { return 8 * j + ((j & 1) ? 7 - group : group); }

template <int LAYOUT, int DTYPE>
__device__ __forceinline__ void store_feat(void* y, uint32_t n, int l, uint32_t N, int64_t stride, float f0, float f1) {
  size_t off = (LAYOUT == HBR_LAYOUT_PLANAR) ? ((size_t)l * N + n) * 2 : (size_t)n * stride + (size_t)l * 2;
  if (DTYPE == HBR_F32) {
    float* p = (float*)y + off;
    if (LAYOUT == HBR_LAYOUT_PLANAR) {
      *(float2*)p = make_float2(f0, f1);
    } else {
      p[0] = f0; p[1] = f1;
    }
  } else {
    uint16_t* p = (uint16_t*)y + off;
    if (LAYOUT == HBR_LAYOUT_PLANAR) {
      *(uint32_t*)p = pack_bf16x2(f0, f1);
    } else {
      uint32_t v = pack_bf16x2(f0, f1);
      p[0] = (uint16_t)v; p[1] = (uint16_t)(v >> 16);
    }
  }
}

template <int LAYOUT, int DTYPE>
__device__ __forceinline__ void load_feat(const void* y, uint32_t n, int l, uint32_t N, int64_t stride, float& f0, float& f1) {
  size_t off = (LAYOUT == HBR_LAYOUT_PLANAR) ? ((size_t)l * N + n) * 2 : (size_t)n * stride + (size_t)l * 2;
  if (DTYPE == HBR_F32) {
    const float* p = (const float*)y + off;
    if (LAYOUT == HBR_LAYOUT_PLANAR) {
      float2 v = *(const float2*)p; f0 = v.x; f1 = v.y;
    } else {
      f0 = p[0]; f1 = p[1];
    }
  } else {
    const uint16_t* p = (const uint16_t*)y + off;
    f0 = __uint_as_float((uint32_t)p[0] << 16);
    f1 = __uint_as_float((uint32_t)p[1] << 16);
  }
}

// The same load split in two, for software prefetch: `load_feat_raw` only moves bits (nothing waits for the data),
// `decode_feat` turns them into the two features when they are consumed.
template <int LAYOUT, int DTYPE>
__device__ __forceinline__ uint2 load_feat_raw(const void* y, uint32_t n, int l, uint32_t N, int64_t stride) {
  size_t off = (LAYOUT == HBR_LAYOUT_PLANAR) ? ((size_t)l * N + n) * 2 : (size_t)n * stride + (size_t)l * 2;
  if (DTYPE == HBR_F32) {
    const uint32_t* p = (const uint32_t*)y + off;
    if (LAYOUT == HBR_LAYOUT_PLANAR) return *(const uint2*)p;
    return make_uint2(p[0], p[1]);
  } else {
    const uint16_t* p = (const uint16_t*)y + off;
    if (LAYOUT == HBR_LAYOUT_PLANAR) return make_uint2(*(const uint32_t*)p, 0u);  // planar pairs are 4-byte aligned
    return make_uint2((uint32_t)p[0] | ((uint32_t)p[1] << 16), 0u);
  }
}
template <int DTYPE>
__device__ __forceinline__ void decode_feat(uint2 raw, float& f0, float& f1) {
  if (DTYPE == HBR_F32) {
    f0 = __uint_as_float(raw.x); f1 = __uint_as_float(raw.y);
  } else {
    f0 = bf16_lo(raw.x); f1 = bf16_hi(raw.x);
  }
}

// ------------------------------------------------------------------------------------------------
// K1 forward
// ------------------------------------------------------------------------------------------------
template <bool POW2, int LAYOUT, int DTYPE>
__global__ __launch_bounds__(kFwdThreads) void hash_fwd_kernel(PointSrc ps, uint32_t N, const float* __restrict__ tables,
                                                               HashGeom g, void* __restrict__ y, int64_t y_stride,
                                                               int levels_per_group) {
  const int group = blockIdx.x % kXcds;
  const uint32_t tile = blockIdx.x / kXcds;
  const uint32_t n = tile * kFwdThreads + threadIdx.x;
  if (n >= N) return;

  float px, py, pz, nx, ny, nz;
  load_point(ps, n, px, py, pz);
  normalise(g, px, py, pz, nx, ny, nz);

  for (int j = 0; j < levels_per_group; ++j) {
    const int l = group_level(group, j);
    if (l >= g.L) continue;
    Cell c = locate(nx, ny, nz, g.scale[l]);
    uint32_t rows[8];
    float w[8];
    corner_rows<POW2>(g, c, rows);
    corner_weights(c, w);
    const float2* tab = (const float2*)tables + (size_t)l * g.T;
    float2 fv[8];
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
    const int l = group_level(group, j);
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

// ------------------------------------------------------------------------------------------------
// K2 backward, algo 2: LDS-resident table slice per workgroup
// ------------------------------------------------------------------------------------------------
constexpr int kSliceLog2 = 14;                 // 16384 rows * 1 double * 8 B = 128 KiB of the CU's 160 KiB LDS
constexpr int kSliceRows = 1 << kSliceLog2;
constexpr int kLdsBwdThreads = 1024;

// K2 prologue: the level-independent part of the index computation, (x - mu) / sigma, once per point instead of once
// per (point, level, slice, feature) visit (three IEEE divisions + the point generation are ~40 % of a visit's VALU
// work).  Stored in the order the scatter kernel's threads consume it - entry [stripe*1024 + tid] holds point
// stripe*1024 + perm(tid) - so that its 16-byte reads are fully coalesced.
__device__ __forceinline__ uint32_t stripe_perm(uint32_t tid) { return (tid & 63u) * 16u + (tid >> 6); }

__global__ __launch_bounds__(1024) void normalise_kernel(PointSrc ps, uint32_t N, HashGeom g, float* __restrict__ out) {
  const uint32_t base = blockIdx.x * 1024u;
  const uint32_t n_raw = base + stripe_perm(threadIdx.x);
  const uint32_t n = min(n_raw, N - 1);
  float px, py, pz, nx, ny, nz;
  load_point(ps, n, px, py, pz);
  normalise(g, px, py, pz, nx, ny, nz);
  float* q = out + (size_t)(base + threadIdx.x) * 3;  // 12 B per entry: K2 re-reads this buffer 128x per call
  q[0] = nx; q[1] = ny; q[2] = nz;
}

// One workgroup = (level, 16384-row slice, feature f, chunk of points).  Splitting the two features of a row over
// two workgroups keeps the slice at 16384 rows with fp64 accumulators, so a point is still visited 8 times per
// level (4 slices x 2 features) but each visit issues 8 LDS atomics instead of 16.
template <bool POW2, int LAYOUT, int DTYPE, bool CACHED>
__global__ __launch_bounds__(kLdsBwdThreads) void hash_bwd_lds_kernel(PointSrc ps, uint32_t N, const void* __restrict__ dy,
                                                                      int64_t dy_stride, HashGeom g,
                                                                      float* __restrict__ dtables, int slices_per_level,
                                                                      int chunks, const float* __restrict__ xnorm) {
  extern __shared__ double acc[];  // [kSliceRows]
  // block -> (level, slice, feature, chunk); chunk varies fastest so the blocks of one slice start together
  const uint32_t b = blockIdx.x;
  const uint32_t chunk = b % chunks;
  const uint32_t lsf = b / chunks;
  const int f = lsf & 1;
  const uint32_t slice = (lsf >> 1) % slices_per_level;
  const int l = (lsf >> 1) / slices_per_level;

  for (int i = threadIdx.x; i < kSliceRows; i += kLdsBwdThreads) acc[i] = 0.0;
  __syncthreads();

  const uint32_t row_lo = slice << kSliceLog2;
  // chunks are whole 1024-point stripes so that the stripe permutation is the same in every kernel
  const uint32_t per = ((N + chunks - 1) / chunks + kLdsBwdThreads - 1) / kLdsBwdThreads * kLdsBwdThreads;
  const uint32_t n_begin = chunk * per;
  const uint32_t n_end = min(N, n_begin + per);
  const float scale = g.scale[l];

  // Within each 1024-point stripe the 64 lanes of a wave take points 16 apart (lane i of wave w -> offset 16 i + w):
  // consecutive samples of a ray share their cell at the coarse levels, and 64 lanes adding to the same LDS address
  // serialise (level 0 cost 3.2x a fine level with the natural mapping).  The stripe's 16 waves together still read
  // every dy cache line completely, so the reads stay L1-friendly.
  const uint32_t perm = (threadIdx.x & 63u) * (kLdsBwdThreads / 64) + (threadIdx.x >> 6);
  // One stripe ahead: the next visit's coordinates and dy are requested before this visit's arithmetic, so their
  // latency overlaps it (an iteration was ~2100 cycles per wave, most of it waiting for these two loads at four waves
  // per SIMD).  The prefetch is unconditional - indices are clamped into range instead of branching around the loads,
  // because a load inside a divergent branch is waited for at the join - and a clamped (out-of-range) visit is
  // neutralised by a zero dy.
  struct Visit { float nx, ny, nz; uint2 raw; bool live; };
  auto fetch = [&](uint32_t base) {
    Visit v;
    v.live = base < n_end && base + perm < n_end;
    const uint32_t b = base < n_end ? base : n_begin;            // whole stripes: b + threadIdx.x stays inside the cache
    const uint32_t n = v.live ? base + perm : n_end - 1;         // a valid point index either way
    if (CACHED) {
      const float* q = xnorm + (size_t)(b + threadIdx.x) * 3;
      v.nx = q[0]; v.ny = q[1]; v.nz = q[2];
    } else {
      float px, py, pz;
      load_point(ps, n, px, py, pz);
      normalise(g, px, py, pz, v.nx, v.ny, v.nz);
    }
    v.raw = load_feat_raw<LAYOUT, DTYPE>(dy, n, l, N, dy_stride);
    return v;
  };
  if (n_begin >= n_end) return;  // uniform over the workgroup (an empty chunk): nothing to add, nothing to flush
  Visit nxt = fetch(n_begin);
  for (uint32_t base = n_begin; base < n_end; base += kLdsBwdThreads) {
    const Visit cur = nxt;
    nxt = fetch(base + kLdsBwdThreads);
    const float nx = cur.nx, ny = cur.ny, nz = cur.nz;
    float d0, d1;
    decode_feat<DTYPE>(cur.raw, d0, d1);
    const float dv = cur.live ? (f ? d1 : d0) : 0.f;
    Cell c = locate(nx, ny, nz, scale);
    if constexpr (POW2) {
      // The loop is VALU-bound (93 instructions per visit, SIMDs 89 % busy with the atomics removed), so the visit is
      // written for instruction count.  Hash components are pre-shifted by 3 - (h << 3) distributes over ^ and &, and
      // (c * P) << 3 == c * (P << 3) mod 2^32 - so each corner's masked hash IS its byte offset in the fp64 slice:
      // one bitop, one subtract, one compare per corner.  The corner weight is split as (x*y) * (z*dy): four xy products
      // and two z*dy products per visit, converted to fp64 once (6 conversions instead of 8), and the last product is
      // taken in fp64 inside the predicated part (1 instruction instead of mul, mul, cvt, shift).  The contribution
      // differs from fl(fl(fl(x*y)*z)*dy) by at most an ulp of fp32 - it is then accumulated in fp64 as before.
      // Slices are aligned blocks of kSliceRows rows, so with the slice's first byte offset XOR-ed into the y/z terms
      // the masked hash is < 8*kSliceRows exactly when the row is in the slice, and is then the byte offset inside it.
      const uint32_t mask8 = g.mask << 3, lo8 = row_lo << 3;
      const uint32_t x0 = (uint32_t)c.cx << 3, x1 = x0 + 8u;
      const uint32_t y0 = (uint32_t)c.cy * (kPrimeY << 3), y1 = y0 + (kPrimeY << 3);
      const uint32_t zz0 = (uint32_t)c.cz * (kPrimeZ << 3), zz1 = zz0 + (kPrimeZ << 3);
      const uint32_t a[4] = {y0 ^ zz0 ^ lo8, y1 ^ zz0 ^ lo8, y0 ^ zz1 ^ lo8, y1 ^ zz1 ^ lo8};
      const float gx = __fsub_rn(1.0f, c.fx), gy = __fsub_rn(1.0f, c.fy), gz = __fsub_rn(1.0f, c.fz);
      const double xy[4] = {(double)__fmul_rn(gx, gy), (double)__fmul_rn(c.fx, gy), (double)__fmul_rn(gx, c.fy),
                            (double)__fmul_rn(c.fx, c.fy)};
      const double zd[2] = {(double)__fmul_rn(gz, dv), (double)__fmul_rn(c.fz, dv)};
#pragma unroll
      for (int k = 0; k < 8; ++k) {  // corner k: +1 on x / y / z iff bit 0 / 1 / 2 (hash_encoding.py:34-37)
        const uint32_t off = (((k & 1) ? x1 : x0) ^ a[k >> 1]) & mask8;
        if (off < (uint32_t)(kSliceRows << 3)) {
          const double v = xy[k & 3] * zd[k >> 2];
#ifdef HBR_ABL_NO_DSADD
          asm volatile("" ::"v"(off), "v"(v));
#else
          // (an inline-asm ds_add_f64 on the raw offset saves the compiler's `v_add_u32 ..., 0` per corner, but makes it
          // branch around every predicated block: measured 1.5 % slower)
          atomicAdd((double*)((char*)acc + off), v);
#endif
        }
      }
    } else {
      uint32_t rows[8];
      float w[8];
      corner_rows<POW2>(g, c, rows);
      corner_weights(c, w);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        uint32_t rel = rows[k] - row_lo;  // wraps to a huge value when the row is below the slice
        if (rel < (uint32_t)kSliceRows) atomicAdd(&acc[rel], (double)__fmul_rn(w[k], dv));
      }
    }
  }
  __syncthreads();

  // flush: contiguous wave-instructions of float atomics (stride 2 floats); skip exact zeros (untouched rows)
  const int64_t rows_here = min((int64_t)kSliceRows, g.T - (int64_t)row_lo);
  float* out = dtables + ((size_t)l * g.T + row_lo) * 2 + f;
  for (int64_t i = threadIdx.x; i < rows_here; i += kLdsBwdThreads) {
    const float v = (float)acc[i];
    if (v != 0.f) unsafeAtomicAdd(out + 2 * i, v);
  }
}

template <bool POW2, int LAYOUT>
static int launch_fwd_dtype(int dtype, dim3 grid, hipStream_t st, PointSrc ps, uint32_t N, const float* tables,
                            const HashGeom& g, void* y, int64_t stride, int lpg) {
  if (dtype == HBR_F32)
    hipLaunchKernelGGL((hash_fwd_kernel<POW2, LAYOUT, HBR_F32>), grid, dim3(kFwdThreads), 0, st, ps, N, tables, g, y, stride, lpg);
  else
    hipLaunchKernelGGL((hash_fwd_kernel<POW2, LAYOUT, HBR_BF16>), grid, dim3(kFwdThreads), 0, st, ps, N, tables, g, y, stride, lpg);
  return HBR_OK;
}

template <bool POW2, int LAYOUT, int DTYPE>
static void launch_bwd(int algo, hipStream_t st, PointSrc ps, uint32_t N, const void* dy, int64_t stride, const HashGeom& g,
                       float* dtables, float* xnorm) {
  if (algo == 1) {
    const int lpg = (g.L + kXcds - 1) / kXcds;
    const uint32_t tiles = (N + kFwdThreads - 1) / kFwdThreads;
    hipLaunchKernelGGL((hash_bwd_atomic_kernel<POW2, LAYOUT, DTYPE>), dim3(tiles * kXcds), dim3(kFwdThreads), 0, st, ps, N, dy,
                       stride, g, dtables, lpg);
  } else {
    const int spl = (int)((g.T + kSliceRows - 1) / kSliceRows);
    // Chunks of ~128 Ki points (measured at N = 2M, L = 16: 4 chunks -> 1.20 ms, 8 -> 1.02, 16 -> 0.97, 32 -> 1.02: a
    // workgroup's 128 KiB zero + flush must be amortised, yet the grid has to fill 256 CUs several times over).  The
    // count depends on N, not on L, so a launch over a sub-range of the levels (the staged multi-GPU all-reduce) runs
    // the same per-workgroup shape as the full one.  With the chunk index varying fastest and a multiple of 8 chunks,
    // chunk c of every (level, slice, feature) lands on XCD c % 8, so the cached coordinates and dy of a chunk are
    // re-read from that XCD's L2.  Small problems still get >= 512 workgroups; at most one chunk per 1024-point stripe.
    constexpr int64_t kChunkPoints = 128 * 1024;
    constexpr int kMinBlocks = 512;
    int chunks = (int)((N + kChunkPoints / 2) / kChunkPoints);
    if (chunks >= 8) chunks = (chunks + 4) / 8 * 8;
    const int min_chunks = (kMinBlocks + g.L * spl * 2 - 1) / (g.L * spl * 2);
    if (chunks < min_chunks) chunks = min_chunks;
    int max_chunks = (int)((N + kLdsBwdThreads - 1) / kLdsBwdThreads);
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    const int lds = kSliceRows * (int)sizeof(double);
    const dim3 grid((uint32_t)(g.L * spl * 2 * chunks));
    if (xnorm) {
      auto kern = hash_bwd_lds_kernel<POW2, LAYOUT, DTYPE, true>;
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL(kern, grid, dim3(kLdsBwdThreads), lds, st, ps, N, dy, stride, g, dtables, spl, chunks, (const float*)xnorm);
    } else {
      auto kern = hash_bwd_lds_kernel<POW2, LAYOUT, DTYPE, false>;
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL(kern, grid, dim3(kLdsBwdThreads), lds, st, ps, N, dy, stride, g, dtables, spl, chunks, (const float*)nullptr);
    }
  }
}

static int check_points(const float* x, const float* o, const float* d, const float* t, int64_t R, int64_t S, PointSrc& ps,
                        uint32_t& N) {
  if (R < 0 || S < 1) return HBR_EINVAL;
  if (R * S > 0x7fffffffLL) return HBR_EUNSUPPORTED;
  N = (uint32_t)(R * S);
  if (x) {
    ps = make_point_src(x, nullptr, nullptr, nullptr, (uint32_t)S);
  } else {
    if (!o || !d || !t) return HBR_EINVAL;
    ps = make_point_src(nullptr, o, d, t, (uint32_t)S);
  }
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

// optional workspace of the LDS-slice algorithm: normalised coordinates, 3 floats per point, padded to whole stripes
extern "C" int64_t hbr_hash_bwd_workspace_bytes(int64_t N, int, int64_t, int, int algo) {
  if (algo == 1 || N < 65536) return 0;
  return (N + 1023) / 1024 * 1024 * 3 * (int64_t)sizeof(float);
}

extern "C" int hbr_hash_encode_bwd(const float* x, const float* rays_o, const float* rays_d, const float* t, int64_t R,
                                   int64_t S, const void* dy, int layout, int64_t dy_stride, int dy_dtype,
                                   const float* scales_host, const float* mu_host, float sigma, int L, int64_t T, int F,
                                   float* dtables, int algo, void* ws, int64_t ws_bytes, void* stream) {
  if (!dy || !dtables) return HBR_EINVAL;
  if (F != 2) return HBR_EUNSUPPORTED;
  if (layout != HBR_LAYOUT_ROWS && layout != HBR_LAYOUT_PLANAR) return HBR_EINVAL;
  if (dy_dtype != HBR_F32 && dy_dtype != HBR_BF16) return HBR_EINVAL;
  if (layout == HBR_LAYOUT_ROWS && dy_stride < (int64_t)L * F) return HBR_EINVAL;
  if (algo < 0 || algo > 2) return HBR_EINVAL;
  HashGeom g;
  int rc = fill_geom(g, scales_host, mu_host, sigma, L, T);
  if (rc) return rc;
  PointSrc ps;
  uint32_t N;
  rc = check_points(x, rays_o, rays_d, t, R, S, ps, N);
  if (rc) return rc;
  if (N == 0) return HBR_OK;
  // auto: the LDS-slice kernel wins once there are enough points to amortise its fixed 128 KiB flush per block
  if (algo == 0) algo = (N >= 65536u) ? 2 : 1;
  hipStream_t st = (hipStream_t)stream;
  float* xnorm = nullptr;
  const int64_t stripes = ((int64_t)N + 1023) / 1024;
  if (algo == 2 && ws && ws_bytes >= stripes * 1024 * 3 * (int64_t)sizeof(float) && (((uintptr_t)ws) & 15) == 0) {
    xnorm = (float*)ws;
    hipLaunchKernelGGL(normalise_kernel, dim3((uint32_t)stripes), dim3(1024), 0, st, ps, N, g, xnorm);
  }
#define HBR_BWD(P, LY, DT) launch_bwd<P, LY, DT>(algo, st, ps, N, dy, dy_stride, g, dtables, xnorm)
  if (g.pow2) {
    if (layout == HBR_LAYOUT_PLANAR) { if (dy_dtype == HBR_F32) HBR_BWD(true, HBR_LAYOUT_PLANAR, HBR_F32); else HBR_BWD(true, HBR_LAYOUT_PLANAR, HBR_BF16); }
    else { if (dy_dtype == HBR_F32) HBR_BWD(true, HBR_LAYOUT_ROWS, HBR_F32); else HBR_BWD(true, HBR_LAYOUT_ROWS, HBR_BF16); }
  } else {
    if (layout == HBR_LAYOUT_PLANAR) { if (dy_dtype == HBR_F32) HBR_BWD(false, HBR_LAYOUT_PLANAR, HBR_F32); else HBR_BWD(false, HBR_LAYOUT_PLANAR, HBR_BF16); }
    else { if (dy_dtype == HBR_F32) HBR_BWD(false, HBR_LAYOUT_ROWS, HBR_F32); else HBR_BWD(false, HBR_LAYOUT_ROWS, HBR_BF16); }
  }
#undef HBR_BWD
  HBR_RETURN_IF_LAUNCH_FAILED();
  return HBR_OK;
}
