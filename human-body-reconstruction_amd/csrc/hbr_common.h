// Shared host/device helpers for libhbr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hbr_hip.h"

#define HBR_WAVE 64

// launch-error check: kernels are enqueued asynchronously; this only catches launch-time failures
#define HBR_RETURN_IF_LAUNCH_FAILED()                \
  do {                                               \
    if (hipGetLastError() != hipSuccess) return HBR_ELAUNCH; \
  } while (0)

namespace hbr {

// hash multipliers (reference hash_encoding.py:24; the int32 wrap of 2654435761 is -1640531535)
constexpr uint32_t kPrimeY = 2654435761u;
constexpr uint32_t kPrimeZ = 805459861u;
constexpr int64_t kPrimeY64 = -1640531535LL;
constexpr int64_t kPrimeZ64 = 805459861LL;

struct HashGeom {
  float scale[HBR_MAX_LEVELS];  // N_l, computed on the host with the reference's torch ops
  float mu[3];
  float sigma;
  int L;
  int pow2;        // 1 -> row = h & mask
  uint32_t mask;   // T-1 when pow2
  int64_t T;
};

// where the points of a launch come from: explicit [N,3] or generated as o + d*t (n = r*S + s)
struct PointSrc {
  const float* x;
  const float* o;
  const float* d;
  const float* t;
  uint32_t S;
  uint32_t magic, shift;  // n / S == (umulhi(n, magic) + n) >> shift for n < 2^31 (host: make_point_src)
};

// round-up magic number for dividing 31-bit n by S (Hacker's Delight 10-9, the "add" form): exact for n < 2^31
inline PointSrc make_point_src(const float* x, const float* o, const float* d, const float* t, uint32_t S) {
  PointSrc ps{x, o, d, t, S, 0u, 0u};
  uint32_t sh = 0;
  while ((1ull << sh) < S) ++sh;
  ps.shift = sh;
  ps.magic = (uint32_t)(((1ull << 32) * ((1ull << sh) - S)) / S + 1);
  return ps;
}

__device__ __forceinline__ void load_point(const PointSrc& ps, uint32_t n, float& px, float& py, float& pz) {
  if (ps.x) {
    const float* p = ps.x + (size_t)n * 3;
    px = p[0]; py = p[1]; pz = p[2];
  } else {
    uint32_t r = (__umulhi(n, ps.magic) + n) >> ps.shift;
    uint32_t s = n - r * ps.S;
    float tt = ps.t[s];
    const float* o = ps.o + (size_t)r * 3;
    const float* d = ps.d + (size_t)r * 3;
    // vol_renderer.py:165: mul then add, separately rounded
    px = __fadd_rn(o[0], __fmul_rn(d[0], tt));
    py = __fadd_rn(o[1], __fmul_rn(d[1], tt));
    pz = __fadd_rn(o[2], __fmul_rn(d[2], tt));
  }
}

// (x - mu) / sigma : level independent (hash_encoding.py:154, first two ops)
__device__ __forceinline__ void normalise(const HashGeom& g, float px, float py, float pz, float& nx, float& ny,
                                          float& nz) {
  nx = __fdiv_rn(__fsub_rn(px, g.mu[0]), g.sigma);
  ny = __fdiv_rn(__fsub_rn(py, g.mu[1]), g.sigma);
  nz = __fdiv_rn(__fsub_rn(pz, g.mu[2]), g.sigma);
}

struct Cell {
  int cx, cy, cz;    // truncated-toward-zero cell (hash_encoding.py:157)
  float fx, fy, fz;  // un_x - x0 (:158)
};

__device__ __forceinline__ Cell locate(float nx, float ny, float nz, float scale) {
  Cell c;
  float ux = __fmul_rn(nx, scale), uy = __fmul_rn(ny, scale), uz = __fmul_rn(nz, scale);
  c.cx = (int)ux; c.cy = (int)uy; c.cz = (int)uz;
  c.fx = __fsub_rn(ux, (float)c.cx);
  c.fy = __fsub_rn(uy, (float)c.cy);
  c.fz = __fsub_rn(uz, (float)c.cz);
  return c;
}

// rows[n] for corner n: +1 on axis d iff bit d of n (hash_encoding.py:34-37,:135), hash :49-53
template <bool POW2>
__device__ __forceinline__ void corner_rows(const HashGeom& g, const Cell& c, uint32_t rows[8]) {
  if (POW2) {
    uint32_t hx0 = (uint32_t)c.cx, hx1 = hx0 + 1u;
    uint32_t hy0 = (uint32_t)c.cy * kPrimeY, hy1 = hy0 + kPrimeY;
    uint32_t hz0 = (uint32_t)c.cz * kPrimeZ, hz1 = hz0 + kPrimeZ;
    uint32_t a00 = hy0 ^ hz0, a10 = hy1 ^ hz0, a01 = hy0 ^ hz1, a11 = hy1 ^ hz1;
    rows[0] = (hx0 ^ a00) & g.mask; rows[1] = (hx1 ^ a00) & g.mask;
    rows[2] = (hx0 ^ a10) & g.mask; rows[3] = (hx1 ^ a10) & g.mask;
    rows[4] = (hx0 ^ a01) & g.mask; rows[5] = (hx1 ^ a01) & g.mask;
    rows[6] = (hx0 ^ a11) & g.mask; rows[7] = (hx1 ^ a11) & g.mask;
  } else {
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      int64_t ix = (int64_t)c.cx + (n & 1), iy = (int64_t)c.cy + ((n >> 1) & 1), iz = (int64_t)c.cz + ((n >> 2) & 1);
      int64_t v = ix ^ (iy * kPrimeY64) ^ (iz * kPrimeZ64);
      int64_t r = v % g.T;  // C remainder has the dividend's sign; the reference floor-mods
      if (r < 0) r += g.T;
      rows[n] = (uint32_t)r;
    }
  }
}

// trilinear weights, multiplied in axis order x,y,z (hash_encoding.py:142-143)
__device__ __forceinline__ void corner_weights(const Cell& c, float w[8]) {
  float gx = __fsub_rn(1.0f, c.fx), gy = __fsub_rn(1.0f, c.fy), gz = __fsub_rn(1.0f, c.fz);
  float xy00 = __fmul_rn(gx, gy), xy10 = __fmul_rn(c.fx, gy), xy01 = __fmul_rn(gx, c.fy), xy11 = __fmul_rn(c.fx, c.fy);
  w[0] = __fmul_rn(xy00, gz); w[1] = __fmul_rn(xy10, gz); w[2] = __fmul_rn(xy01, gz); w[3] = __fmul_rn(xy11, gz);
  w[4] = __fmul_rn(xy00, c.fz); w[5] = __fmul_rn(xy10, c.fz); w[6] = __fmul_rn(xy01, c.fz); w[7] = __fmul_rn(xy11, c.fz);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {  // one v_cvt_pk_bf16_f32 (round to nearest even)
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  f32x2_t v;
  v[0] = a; v[1] = b;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf16_lo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }

inline int fill_geom(HashGeom& g, const float* scales_host, const float* mu_host, float sigma, int L, int64_t T) {
  if (!scales_host || !mu_host || L < 1 || L > HBR_MAX_LEVELS || T < 1 || T > (1LL << 31)) return HBR_EINVAL;
  for (int l = 0; l < HBR_MAX_LEVELS; ++l) g.scale[l] = l < L ? scales_host[l] : 0.f;
  g.mu[0] = mu_host[0]; g.mu[1] = mu_host[1]; g.mu[2] = mu_host[2];
  g.sigma = sigma;
  g.L = L;
  g.T = T;
  g.pow2 = (T & (T - 1)) == 0;
  g.mask = g.pow2 ? (uint32_t)(T - 1) : 0u;
  return HBR_OK;
}

}  // namespace hbr
