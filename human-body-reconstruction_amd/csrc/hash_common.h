// Helpers shared by the hash-grid kernels (hash_encode.hip: K1 forward + the global-atomics backward;
// hash_scatter.hip: the LDS backward): feature-buffer addressing, the XCD-group level map, argument checks.
#pragma once
#include "hbr_common.h"

namespace hbr {

constexpr int kFwdThreads = 256;
constexpr int kXcds = 8;

// j-th level of XCD group `group`: groups pair a coarse level (cheap: the wave's gathers coalesce) with a fine one
// (texture-rate bound) - {k, 15-k} for L = 16 - so that the 8 XCDs finish together; each XCD's L2 still only sees
// L/8 levels.
// Round 3: measured one level at a time (profiles/r03_k1_level_costs.txt) the cost keeps rising to the finest level -
// 7 / 40 / 182 us beyond the per-point floor for levels 0 / 8 / 15 - so {0,15} takes 187 us where {7,8} takes 63, and the
// launch waits for group 0.  Odd point tiles therefore take the MIRRORED pair ({7-k, 8+k} on the XCD that has {k, 15-k}
// for the even ones): every XCD then carries (187 + 63) / 2 ... (142 + 122) / 2 = 125-132 us, and its L2 holds four
// tables (2 of its 4 MiB) instead of two.
// `mirror` (run time, round 4): off for tables so large that an XCD's L2 cannot hold the four level tables the mirrored map
// gives it (hbr_hash_encode_fwd decides; the global-atomics backward always mirrors).
__device__ __forceinline__ int group_level(int group, int j, uint32_t tile, bool mirror = true) {
#ifndef HBR_K1_MIRROR
#define HBR_K1_MIRROR 1
#endif
  const bool mirrored = mirror && (HBR_K1_MIRROR == 2 ? tile >= (gridDim.x / kXcds + 1) / 2 : (HBR_K1_MIRROR == 1 && (tile & 1u)));
  const int g = mirrored ? 7 - group : group;
  return 8 * j + ((j & 1) ? 7 - g : g);
}

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

// points of a launch: explicit x[N,3] or rays (o, d, t); N = R*S < 2^31
inline int check_points(const float* x, const float* o, const float* d, const float* t, int64_t R, int64_t S, PointSrc& ps,
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

// algo 1 of hbr_hash_encode_bwd (hash_encode.hip): one global float atomic per corner-feature
int launch_hash_bwd_atomic(hipStream_t st, PointSrc ps, uint32_t N, const void* dy, int layout, int64_t dy_stride, int dy_dtype,
                           const HashGeom& g, float* dtables);

}  // namespace hbr
