// Device pieces of K0 shared by sample.hip (hbr_strat_sample) and mlp.hip (hbr_render_prologue): the counter-based
// generator and the stratified depths t[S] (strat_sampler, reference helper.py:210-237, non-exp branch).
#pragma once
#include "hbr_common.h"

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

struct StratArgs {
  float tn, tf;
  uint32_t S;
  const float* u;  // the caller's uniform draw [S], or nullptr: 24-bit uniforms in [0, 1) from Philox(seed, offset, s)
  uint64_t seed, offset;
  float* t;        // [S] out
};

// t[s] = linspace(tn, tf, S)[s] + (u[s] * (tf - tn)) / S      (helper.py:234-235; one jitter per sample index)
// linspace as torch evaluates it in fp32: step = (tf - tn) / (S - 1); the lower half counts up from tn, the upper half
// down from tf.
__device__ __forceinline__ void strat_sample_one(const StratArgs& a, uint32_t s) {
  if (s >= a.S) return;
  float us;
  if (a.u) {
    us = a.u[s];
  } else {
    const uint4 r = philox4x32_10(make_uint4(s, (uint32_t)a.offset, (uint32_t)(a.offset >> 32), 0u),
                                  make_uint2((uint32_t)a.seed, (uint32_t)(a.seed >> 32)));
    us = (float)(r.x >> 8) * 5.9604644775390625e-8f;  // 2^-24
  }
  const float span = __fsub_rn(a.tf, a.tn);
  float lin = a.tn;
  if (a.S > 1) {
    const float step = __fdiv_rn(span, (float)(a.S - 1));
    lin = s < a.S / 2 ? __fadd_rn(a.tn, __fmul_rn(step, (float)s)) : __fsub_rn(a.tf, __fmul_rn(step, (float)(a.S - 1 - s)));
  }
  a.t[s] = __fadd_rn(lin, __fdiv_rn(__fmul_rn(us, span), (float)a.S));
}

}  // namespace hbr
