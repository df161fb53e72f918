// One ray's compositing forward, 2*MSE loss term and compositing backward in ONE wave (calc_color, reference
// helper.py:53-107 non-SDF branch; train_hash2.py:221 with Cf = Cr): lane i holds samples 64 c + i, c < NCH, as the
// (r, g, b, sigma) vectors the MLP writes.  Shared by composite.hip's composite_loss_vec_kernel (samples from the [N,4]
// buffer) and mlp.hip's fused render + backward kernel (samples from LDS): the same operations in the same order, so the
// two give the same bits on the same inputs.
#pragma once
#include "wave_reduce.h"

namespace hbr {

template <int NCH>
struct RayComposite {
  float4 v[NCH];               // in: (r, g, b, sigma) of the lane's samples; 0 beyond S
  float dl[NCH];               // in: delta_s = (t[s+1] - t[s]) * dir_norm, 0 for the last sample and beyond S
  float4 d[NCH];               // out: gradient wrt (r, g, b, sigma)
  float c0, c1, c2;            // out: the ray's colour (wave-uniform)
  float se;                    // out: sum over the three channels of (C - gt)^2 (wave-uniform)

  // k = gscale * 4 / (3 R): dL/dC = k (C - gt)
  __device__ __forceinline__ void run(int S, int lane, float gt0, float gt1, float gt2, float k) {
    float Tr[NCH], ex[NCH];
    bool live[NCH];
    float carry = 0.f;
    c0 = c1 = c2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int s = c * 64 + lane;
      float sg = v[c].w;
      live[c] = s < S && !(sg < -10.f);  // helper.py:76
      if (sg < -10.f) sg = -10.f;
      const float p = s < S ? __fmul_rn(sg, dl[c]) : 0.f;
      const WaveScan sc = wave_prefix_sum(p, lane);
      Tr[c] = expf(-(carry + sc.excl));   // helper.py:93-95
      ex[c] = expf(-p);
      const float w = Tr[c] * (1.f - ex[c]);  // :91,:102
      c0 += w * v[c].x; c1 += w * v[c].y; c2 += w * v[c].z;  // (padding lanes: v = 0)
      carry += sc.total;
    }
    auto add = [](float a, float b) { return a + b; };
    c0 = wave_reduce(c0, add); c1 = wave_reduce(c1, add); c2 = wave_reduce(c2, add);
    const float e0 = c0 - gt0, e1 = c1 - gt1, e2 = c2 - gt2;
    se = (e0 * e0 + e1 * e1) + e2 * e2;
    const float g0 = k * e0, g1 = k * e1, g2 = k * e2;
    float suffix = 0.f;
#pragma unroll
    for (int c = NCH - 1; c >= 0; --c) {
      const float w = Tr[c] * (1.f - ex[c]);
      const float g = g0 * v[c].x + g1 * v[c].y + g2 * v[c].z;
      const WaveScan rs = wave_suffix_sum(g * w, lane);
      const float dp = g * Tr[c] * ex[c] - (suffix + rs.excl);
      d[c] = make_float4(w * g0, w * g1, w * g2, live[c] ? dp * dl[c] : 0.f);
      suffix += rs.total;
    }
  }
};

}  // namespace hbr
