// Wave-wide reduction with DPP row operations + four v_readlane: ~11 VALU per value, no LDS traffic.
// (`__shfl_xor` compiles to ds_bpermute_b32: 42 of them per thread made normalise_kernel LDS-bound, 26.8 us instead
// of its 11.8 us of HBM time.)  Requires all 64 lanes active.  The result is wave-uniform.
#pragma once
#include <hip/hip_runtime.h>

namespace hbr {

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// op must be commutative and associative (min, max).  Reduction over each row of 16 lanes, result in all 16.
template <class Op>
__device__ __forceinline__ float row_reduce(float v, Op op) {
  v = op(v, dpp_move<0xB1>(v));   // quad_perm [1,0,3,2]: lane ^ 1
  v = op(v, dpp_move<0x4E>(v));   // quad_perm [2,3,0,1]: lane ^ 2          -> uniform over each quad
  v = op(v, dpp_move<0x141>(v));  // row_half_mirror: i <-> 7 - i           -> uniform over each 8 lanes
  v = op(v, dpp_move<0x140>(v));  // row_mirror: i <-> 15 - i               -> uniform over each row of 16
  return v;
}
template <class Op>
__device__ __forceinline__ float wave_reduce(float v, Op op) {
  v = row_reduce(v, op);
  const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return op(op(a, b), op(c, d));
}

// ---- 64-lane prefix / suffix sums: Kogge-Stone inside each row of 16 with DPP row shifts (a lane without a source
// keeps `old` = 0), then the three / four row totals by v_readlane.  ~14 VALU + 4 readlanes instead of 7 ds_bpermute.
template <int CTRL>
__device__ __forceinline__ float dpp_or_zero(float v) {  // lanes whose source falls outside the row read 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
struct WaveScan {
  float incl, excl, total;
};
__device__ __forceinline__ float lane_value(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// incl[i] = sum_{k<=i} v[k], excl[i] = sum_{k<i} v[k]
__device__ __forceinline__ WaveScan wave_prefix_sum(float v, int lane) {
  v += dpp_or_zero<0x111>(v);  // row_shr:1
  v += dpp_or_zero<0x112>(v);  // row_shr:2
  v += dpp_or_zero<0x114>(v);  // row_shr:4
  v += dpp_or_zero<0x118>(v);  // row_shr:8
  const float t0 = lane_value(v, 15), t1 = lane_value(v, 31), t2 = lane_value(v, 47), t3 = lane_value(v, 63);
  const float t01 = t0 + t1, t012 = t01 + t2;
  const int row = lane >> 4;
  const float off = row == 0 ? 0.f : (row == 1 ? t0 : (row == 2 ? t01 : t012));
  WaveScan r;
  r.incl = v + off;
  r.excl = dpp_or_zero<0x111>(v) + off;
  r.total = t012 + t3;
  return r;
}
// incl[i] = sum_{k>=i} v[k], excl[i] = sum_{k>i} v[k]
__device__ __forceinline__ WaveScan wave_suffix_sum(float v, int lane) {
  v += dpp_or_zero<0x101>(v);  // row_shl:1
  v += dpp_or_zero<0x102>(v);  // row_shl:2
  v += dpp_or_zero<0x104>(v);  // row_shl:4
  v += dpp_or_zero<0x108>(v);  // row_shl:8
  const float t0 = lane_value(v, 0), t1 = lane_value(v, 16), t2 = lane_value(v, 32), t3 = lane_value(v, 48);
  const float t32 = t3 + t2, t321 = t32 + t1;
  const int row = lane >> 4;
  const float off = row == 3 ? 0.f : (row == 2 ? t3 : (row == 1 ? t32 : t321));
  WaveScan r;
  r.incl = v + off;
  r.excl = dpp_or_zero<0x101>(v) + off;
  r.total = t321 + t0;
  return r;
}

}  // namespace hbr
