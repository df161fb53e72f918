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

}  // namespace hbr
