// K5: alpha compositing along rays (calc_color, reference helper.py:53-107, non-SDF branch) and the
// 2*MSE loss of train_hash2.py:221.  One wavefront per ray: the S samples are swept in chunks of 64
// lanes, the transmittance prefix is a wave-level scan (no LDS round trip for the scan itself).
#include "hbr_common.h"
#include "wave_reduce.h"
#include "composite_ray.h"

namespace hbr {

constexpr int kRaysPerBlock = 4;
constexpr int kMaxChunks = 64;  // S <= 4096

// 64-lane sums and prefix / suffix sums by DPP row operations (wave_reduce.h): `__shfl_*` compiles to ds_bpermute,
// ~42 of them per ray in the backward kernel, which made it LDS-bound (17 us for 11 us of HBM traffic).
__device__ __forceinline__ float wave_sum(float v) {
  return wave_reduce(v, [](float a, float b) { return a + b; });
}

struct RayIn {
  const float* t;
  int64_t t_stride;  // 0: one t[S] shared by all rays (vol_render's first pass); S: per-ray t[R,S] (hierarchical pass)
  const float* rgb;
  int64_t rgb_stride;
  const float* sigma;
  int64_t sigma_stride;
  const float* dir_norm;
  int64_t R, S;
};

// p = clamp(sigma)*delta for sample s of ray r; `live` says whether sigma's gradient flows (helper.py:76)
__device__ __forceinline__ float sample_p(const RayIn& in, int64_t r, int64_t s, float dn, float& delta, bool& live) {
  delta = 0.f;
  live = false;
  if (s >= in.S) return 0.f;
  const float* tr = in.t + r * in.t_stride;
  if (s < in.S - 1) delta = __fmul_rn(__fsub_rn(tr[s + 1], tr[s]), dn);  // helper.py:67,71; last delta stays 0
  float sg = in.sigma[(r * in.S + s) * in.sigma_stride];
  live = !(sg < -10.f);
  if (!live) sg = -10.f;
  return __fmul_rn(sg, delta);
}

__global__ __launch_bounds__(kRaysPerBlock * 64) void composite_fwd_kernel(RayIn in, float* __restrict__ Cr,
                                                                           float* __restrict__ wts) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
  if (r >= in.R) return;
  const float dn = in.dir_norm ? in.dir_norm[r] : 1.f;
  float carry = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
  for (int64_t base = 0; base < in.S; base += 64) {
    const int64_t s = base + lane;
    float delta;
    bool live;
    const float p = sample_p(in, r, s, dn, delta, live);
    const WaveScan sc = wave_prefix_sum(p, lane);
    const float Tr = expf(-(carry + sc.excl));   // helper.py:93-95: exclusive transmittance, T_0 = 1
    const float alpha = 1.f - expf(-p);          // :91
    const float w = Tr * alpha;                  // :102
    if (s < in.S) {
      const float* c = in.rgb + (r * in.S + s) * in.rgb_stride;
      c0 += w * c[0]; c1 += w * c[1]; c2 += w * c[2];
      if (wts) wts[r * in.S + s] = w;
    }
    carry += sc.total;
  }
  c0 = wave_sum(c0); c1 = wave_sum(c1); c2 = wave_sum(c2);
  if (lane == 0) {
    Cr[r * 3 + 0] = c0; Cr[r * 3 + 1] = c1; Cr[r * 3 + 2] = c2;
  }
}

// dL/dp_s = g_s T_s exp(-p_s) - sum_{k>s} g_k w_k,   g_s = dC . rgb_s ;  d sigma_s = live ? dL/dp_s * delta_s : 0 ;
// d rgb_s = w_s dC.   Pass 1 records the per-chunk cumulative p; pass 2 walks the chunks backwards so the
// suffix sum is accumulated from the far end exactly as autograd's reverse cumsum does.
__global__ __launch_bounds__(kRaysPerBlock * 64) void composite_bwd_kernel(RayIn in, const float* __restrict__ dCr,
                                                                           float* __restrict__ d_rgb,
                                                                           float* __restrict__ d_sigma,
                                                                           const uint8_t* __restrict__ keep) {
  __shared__ float chunk_carry[kRaysPerBlock][kMaxChunks];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * kRaysPerBlock + wv;
  if (r >= in.R) return;
  const float dn = in.dir_norm ? in.dir_norm[r] : 1.f;
  const float g0 = dCr[r * 3 + 0], g1 = dCr[r * 3 + 1], g2 = dCr[r * 3 + 2];
  const int nchunks = (int)((in.S + 63) / 64);
  float carry = 0.f;
  for (int c = 0; c < nchunks; ++c) {
    float delta;
    bool live;
    const float p = sample_p(in, r, (int64_t)c * 64 + lane, dn, delta, live);
    if (lane == 0) chunk_carry[wv][c] = carry;
    carry += wave_sum(p);
  }
  // same-wave LDS hand-off: make the stores visible before the reads below
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  float suffix = 0.f;
  for (int c = nchunks - 1; c >= 0; --c) {
    const int64_t s = (int64_t)c * 64 + lane;
    float delta;
    bool live;
    const float p = sample_p(in, r, s, dn, delta, live);
    const float Tr = expf(-(chunk_carry[wv][c] + wave_prefix_sum(p, lane).excl));
    const float e = expf(-p);
    const float w = Tr * (1.f - e);
    float g = 0.f;
    const float* col = nullptr;
    if (s < in.S) {
      col = in.rgb + (r * in.S + s) * in.rgb_stride;
      g = g0 * col[0] + g1 * col[1] + g2 * col[2];
    }
    const float gw = g * w;
    const WaveScan rs = wave_suffix_sum(gw, lane);
    const float dp = g * Tr * e - (suffix + rs.excl);
    if (s < in.S) {
      // nothing flows back through a sample the occupancy grid masked out (its sigma/rgb are constants, not MLP outputs)
      const bool kept = !keep || keep[r * in.S + s];
      d_sigma[(r * in.S + s) * in.sigma_stride] = (live && kept) ? dp * delta : 0.f;
      float* dc = d_rgb + (r * in.S + s) * in.rgb_stride;
      dc[0] = kept ? w * g0 : 0.f; dc[1] = kept ? w * g1 : 0.f; dc[2] = kept ? w * g2 : 0.f;
    }
    suffix += rs.total;
  }
}

// loss = 2*mean((Cr-gt)^2) over R*3 elements; dCr = gscale * 4*(Cr-gt)/(3R).
// With a scratch block (`partials` [kLossMaxBlocks] floats + one zeroed ticket word behind them) the block sums are
// parked there and the last block to finish adds them up in block order - a fixed summation order, so the loss is
// bitwise reproducible - and resets the ticket for the next call.  Without it: one float atomic per block.
constexpr int kLossMaxBlocks = 1024;
__global__ __launch_bounds__(256) void mse2_kernel(const float* __restrict__ Cr, const float* __restrict__ gt, int64_t n,
                                                   float inv_n, float gscale, float* __restrict__ loss,
                                                   float* __restrict__ dCr, float* __restrict__ partials) {
  __shared__ float part[4];
  __shared__ bool last;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = Cr[i] - gt[i];
    acc += d * d;
    if (dCr) dCr[i] = gscale * 4.f * inv_n * d;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (!loss) return;
  const float mine = (part[0] + part[1]) + (part[2] + part[3]);
  if (!partials) {
    if (threadIdx.x == 0) unsafeAtomicAdd(loss, 2.f * inv_n * mine);
    return;
  }
  unsigned* ticket = (unsigned*)(partials + kLossMaxBlocks);
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = mine;
    __threadfence();
    last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last || threadIdx.x >= 64) return;
  __threadfence();
  float s = 0.f;
  for (unsigned b = threadIdx.x; b < gridDim.x; b += 64) s += __builtin_nontemporal_load(partials + b);  // lane-strided, fixed order
  s = wave_sum(s);
  if (threadIdx.x == 0) {
    *loss += 2.f * inv_n * s;
    *ticket = 0u;
  }
}

// composite_fwd + the 2*MSE loss + composite_bwd of ONE ray in one wave: the colour of a ray depends on nothing but its
// own samples, and the gradient the loss sends back into it is dCr = gscale * 4 (Cr - gt) / (3R) (mse2_kernel's formula),
// so the three launches of a training step (8.8 + 7.9 + 15 us) become one pass over the MLP output that is still in L1
// when the backward sweep re-reads it.  Every per-ray value is computed exactly as the separate kernels compute it (same
// operations in the same order: bit-identical Cr and gradients).
// Loss: a block's 16 waves take rays block*16 + wave, + 16*gridDim, ... and add up their (Cr - gt)^2 in that order; the
// block sums are parked in `partials` [gridDim] and added in a fixed order by closs_finish_kernel - bitwise reproducible
// for a given R; `loss` is WRITTEN, not accumulated into.  At most 512 blocks of 16 waves = 32 waves per CU: with 16 the
// per-ray dependency chain (load, scan, exp, load, scan) left the kernel at 73 us.
constexpr int kLossWaves = 16;       // rays in flight per block of composite_loss_kernel (one per wave)
constexpr int kClossMaxBlocks = 512;  // x 16 waves = 32 waves on each of the 256 CUs
// a block's share of sum (Cr - gt)^2: its waves' sums added in wave order, parked in partials[block]
__device__ __forceinline__ void closs_block_sum(float se, float* ray_se, float* __restrict__ partials) {
  if ((threadIdx.x & 63) == 0) ray_se[threadIdx.x >> 6] = se;
  __syncthreads();
  if (threadIdx.x == 0) {
    float bs = 0.f;
    for (int w = 0; w < kLossWaves; ++w) bs += ray_se[w];
    partials[blockIdx.x] = bs;
  }
}
// loss = 2 * inv_n * sum of the block sums, added in a fixed (lane-strided) order: bitwise reproducible.  A launch of
// its own (one wave) behind the compositing kernel: the kernel boundary makes the partial sums visible, where a
// last-block-finishes scheme needs an agent-scope release in every block - with 33 MB of gradient rows freshly dirty in
// the L2s that fence, not the arithmetic, set the kernel's time (34 us for ~15 us of work).
__global__ __launch_bounds__(64) void closs_finish_kernel(const float* __restrict__ partials, int n, float scale, float* __restrict__ loss) {
  float s = 0.f;
  for (int b = threadIdx.x; b < n; b += 64) s += partials[b];
  s = wave_sum(s);
  if (threadIdx.x == 0) *loss = scale * s;
}
__global__ __launch_bounds__(kLossWaves * 64) void composite_loss_kernel(RayIn in, const float* __restrict__ gt, float inv_n,
                                                                            float gscale, float* __restrict__ loss, float* __restrict__ Cr,
                                                                            float* __restrict__ d_rgb, float* __restrict__ d_sigma,
                                                                            const uint8_t* __restrict__ keep, float* __restrict__ partials) {
  __shared__ float chunk_carry[kLossWaves][kMaxChunks];
  __shared__ float ray_se[kLossWaves];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int nchunks = (int)((in.S + 63) / 64);
  const float k = gscale * 4.f * inv_n;
  float se = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * kLossWaves + wv; r < in.R; r += (int64_t)gridDim.x * kLossWaves) {
    const float dn = in.dir_norm ? in.dir_norm[r] : 1.f;
    // ---- forward (composite_fwd_kernel); the cumulative p at each chunk's start is recorded as composite_bwd_kernel's
    // first pass computes it (wave_sum of the chunk, not the scan's total: the two differ in the last bit)
    float carry = 0.f, bcarry = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int c = 0; c < nchunks; ++c) {
      const int64_t s = (int64_t)c * 64 + lane;
      float delta;
      bool live;
      const float p = sample_p(in, r, s, dn, delta, live);
      const WaveScan sc = wave_prefix_sum(p, lane);
      const float Tr = expf(-(carry + sc.excl));
      const float alpha = 1.f - expf(-p);
      const float w = Tr * alpha;
      if (s < in.S) {
        const float* col = in.rgb + (r * in.S + s) * in.rgb_stride;
        c0 += w * col[0]; c1 += w * col[1]; c2 += w * col[2];
      }
      if (lane == 0) chunk_carry[wv][c] = bcarry;
      carry += sc.total;
      bcarry += wave_sum(p);
    }
    c0 = wave_sum(c0); c1 = wave_sum(c1); c2 = wave_sum(c2);
    // ---- loss and its gradient (mse2_kernel)
    const float e0 = c0 - gt[r * 3 + 0], e1 = c1 - gt[r * 3 + 1], e2 = c2 - gt[r * 3 + 2];
    se += (e0 * e0 + e1 * e1) + e2 * e2;
    const float g0 = k * e0, g1 = k * e1, g2 = k * e2;
    if (lane == 0 && Cr) {
      Cr[r * 3 + 0] = c0; Cr[r * 3 + 1] = c1; Cr[r * 3 + 2] = c2;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's chunk_carry stores before its reads below
    // ---- backward (composite_bwd_kernel's second pass)
    float suffix = 0.f;
    for (int c = nchunks - 1; c >= 0; --c) {
      const int64_t s = (int64_t)c * 64 + lane;
      float delta;
      bool live;
      const float p = sample_p(in, r, s, dn, delta, live);
      const float Tr = expf(-(chunk_carry[wv][c] + wave_prefix_sum(p, lane).excl));
      const float e = expf(-p);
      const float w = Tr * (1.f - e);
      float g = 0.f;
      if (s < in.S) {
        const float* col = in.rgb + (r * in.S + s) * in.rgb_stride;
        g = g0 * col[0] + g1 * col[1] + g2 * col[2];
      }
      const float gw = g * w;
      const WaveScan rs = wave_suffix_sum(gw, lane);
      const float dp = g * Tr * e - (suffix + rs.excl);
      if (s < in.S) {
        const bool kept = !keep || keep[r * in.S + s];
        d_sigma[(r * in.S + s) * in.sigma_stride] = (live && kept) ? dp * delta : 0.f;
        float* dc = d_rgb + (r * in.S + s) * in.rgb_stride;
        dc[0] = kept ? w * g0 : 0.f; dc[1] = kept ? w * g1 : 0.f; dc[2] = kept ? w * g2 : 0.f;
      }
      suffix += rs.total;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // the reads of chunk_carry before the next ray's stores
  }
  closs_block_sum(se, ray_se, partials);
}

// The same for the layout the training step actually has - rgb and sigma interleaved as the MLP's [N,4] rows, at most
// 64 * NCH samples per ray: a lane loads its sample as ONE 16-byte vector, keeps it (and p, T, alpha) in registers for
// the backward sweep and stores the gradient as one 16-byte vector; nothing is read twice, no scan or exp is redone.
// The backward uses the forward's transmittance as it stands (the separate composite_bwd_kernel recomputes it from a
// differently associated chunk sum: the two agree to the last bit or two).
template <int NCH>
__global__ __launch_bounds__(kLossWaves * 64) void composite_loss_vec_kernel(RayIn in, const float* __restrict__ gt, float inv_n,
                                                                               float gscale, float* __restrict__ loss, float* __restrict__ Cr,
                                                                               float4* __restrict__ d_out, const uint8_t* __restrict__ keep,
                                                                               float* __restrict__ partials) {
  __shared__ float ray_se[kLossWaves];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const float k = gscale * 4.f * inv_n;
  const int S = (int)in.S;
  float se = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * kLossWaves + wv; r < in.R; r += (int64_t)gridDim.x * kLossWaves) {
    const float dn = in.dir_norm ? in.dir_norm[r] : 1.f;
    const float* tr = in.t + r * in.t_stride;
    const float4* row = (const float4*)in.rgb + r * S;
    RayComposite<NCH> rc;  // (composite_ray.h: shared with the fused render + backward kernel of mlp.hip)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {  // all loads first
      const int s = c * 64 + lane;
      rc.v[c] = s < S ? row[s] : make_float4(0.f, 0.f, 0.f, 0.f);
      rc.dl[c] = s < S - 1 ? __fmul_rn(__fsub_rn(tr[s + 1], tr[s]), dn) : 0.f;  // helper.py:67,71; last delta stays 0
    }
    rc.run(S, lane, gt[r * 3 + 0], gt[r * 3 + 1], gt[r * 3 + 2], k);
    se += rc.se;
    if (lane == 0 && Cr) {
      Cr[r * 3 + 0] = rc.c0; Cr[r * 3 + 1] = rc.c1; Cr[r * 3 + 2] = rc.c2;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int s = c * 64 + lane;
      if (s < S) {
        const bool kept = !keep || keep[r * S + s];
        d_out[r * S + s] = kept ? rc.d[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  closs_block_sum(se, ray_se, partials);
}

static int check_ray_in(const RayIn& in) {
  if (!in.t || !in.rgb || !in.sigma || in.R < 0 || in.S < 1 || in.rgb_stride < 3 || in.sigma_stride < 1) return HBR_EINVAL;
  if (in.t_stride != 0 && in.t_stride < in.S) return HBR_EINVAL;
  if (in.S > (int64_t)kMaxChunks * 64) return HBR_EUNSUPPORTED;
  return HBR_OK;
}

}  // namespace hbr

using namespace hbr;

extern "C" int hbr_composite_fwd(const float* t, int64_t t_stride, const float* rgb, int64_t rgb_stride, const float* sigma,
                                 int64_t sigma_stride, const float* dir_norm, int64_t R, int64_t S, float* Cr, float* wts,
                                 void* stream) {
  RayIn in{t, t_stride, rgb, rgb_stride, sigma, sigma_stride, dir_norm, R, S};
  int rc = check_ray_in(in);
  if (rc) return rc;
  if (!Cr) return HBR_EINVAL;
  if (R == 0) return HBR_OK;
  const int64_t blocks = (R + kRaysPerBlock - 1) / kRaysPerBlock;
  if (blocks > 0x7fffffffLL) return HBR_EUNSUPPORTED;
  hipLaunchKernelGGL(composite_fwd_kernel, dim3((uint32_t)blocks), dim3(kRaysPerBlock * 64), 0, (hipStream_t)stream, in, Cr, wts);
  HBR_RETURN_IF_LAUNCH_FAILED();
  return HBR_OK;
}

extern "C" int hbr_composite_bwd(const float* t, int64_t t_stride, const float* rgb, int64_t rgb_stride, const float* sigma,
                                 int64_t sigma_stride, const float* dir_norm, int64_t R, int64_t S, const float* dCr,
                                 float* d_rgb, float* d_sigma, const uint8_t* keep, void* stream) {
  RayIn in{t, t_stride, rgb, rgb_stride, sigma, sigma_stride, dir_norm, R, S};
  int rc = check_ray_in(in);
  if (rc) return rc;
  if (!dCr || !d_rgb || !d_sigma) return HBR_EINVAL;
  if (R == 0) return HBR_OK;
  const int64_t blocks = (R + kRaysPerBlock - 1) / kRaysPerBlock;
  if (blocks > 0x7fffffffLL) return HBR_EUNSUPPORTED;
  hipLaunchKernelGGL(composite_bwd_kernel, dim3((uint32_t)blocks), dim3(kRaysPerBlock * 64), 0, (hipStream_t)stream, in, dCr, d_rgb,
                     d_sigma, keep);
  HBR_RETURN_IF_LAUNCH_FAILED();
  return HBR_OK;
}

extern "C" int64_t hbr_mse2_workspace_bytes(void) { return (kLossMaxBlocks + 1) * (int64_t)sizeof(float); }

extern "C" int hbr_mse2_loss_fwd_bwd(const float* Cr, const float* gt, int64_t R, float gscale, float* loss_out, float* dCr,
                                     void* ws, void* stream) {
  if (!Cr || !gt || R < 0) return HBR_EINVAL;
  if (R == 0) return HBR_OK;
  const int64_t n = R * 3;
  int64_t blocks = (n + 255) / 256;
  if (blocks > kLossMaxBlocks) blocks = kLossMaxBlocks;
  hipLaunchKernelGGL(mse2_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, Cr, gt, n, 1.0f / (float)n, gscale,
                     loss_out, dCr, (float*)ws);
  HBR_RETURN_IF_LAUNCH_FAILED();
  return HBR_OK;
}

static int64_t closs_blocks(int64_t R) {
  const int64_t b = (R + kLossWaves - 1) / kLossWaves;
  return b < 1 ? 1 : (b > kClossMaxBlocks ? kClossMaxBlocks : b);
}
extern "C" int64_t hbr_composite_loss_workspace_bytes(int64_t R) { return closs_blocks(R) * (int64_t)sizeof(float); }

extern "C" int hbr_composite_loss_fwd_bwd(const float* t, int64_t t_stride, const float* rgb, int64_t rgb_stride, const float* sigma,
                                          int64_t sigma_stride, const float* dir_norm, int64_t R, int64_t S, const float* gt,
                                          float gscale, float* loss_out, float* Cr, float* d_rgb, float* d_sigma,
                                          const uint8_t* keep, void* ws, void* stream) {
  RayIn in{t, t_stride, rgb, rgb_stride, sigma, sigma_stride, dir_norm, R, S};
  int rc = check_ray_in(in);
  if (rc) return rc;
  if (!gt || !loss_out || !d_rgb || !d_sigma || !ws || R < 1) return HBR_EINVAL;
  const int64_t blocks = closs_blocks(R);
  // the training step's layout: [N,4] rows (r,g,b,sigma), gradients likewise -> the vector kernel
  const bool vec = rgb_stride == 4 && sigma_stride == 4 && sigma == rgb + 3 && d_sigma == d_rgb + 3 && S <= 256 &&
                   ((((uintptr_t)rgb) | ((uintptr_t)d_rgb)) & 15) == 0;
  if (vec) {
    const float inv_n = 1.0f / (float)(R * 3);
    hipStream_t st = (hipStream_t)stream;
    dim3 g((uint32_t)blocks), b(kLossWaves * 64);
    if (S <= 64) hipLaunchKernelGGL(composite_loss_vec_kernel<1>, g, b, 0, st, in, gt, inv_n, gscale, loss_out, Cr, (float4*)d_rgb, keep, (float*)ws);
    else if (S <= 128) hipLaunchKernelGGL(composite_loss_vec_kernel<2>, g, b, 0, st, in, gt, inv_n, gscale, loss_out, Cr, (float4*)d_rgb, keep, (float*)ws);
    else hipLaunchKernelGGL(composite_loss_vec_kernel<4>, g, b, 0, st, in, gt, inv_n, gscale, loss_out, Cr, (float4*)d_rgb, keep, (float*)ws);
    hipLaunchKernelGGL(closs_finish_kernel, dim3(1), dim3(64), 0, st, (const float*)ws, (int)blocks, 2.f * inv_n, loss_out);
    HBR_RETURN_IF_LAUNCH_FAILED();
    return HBR_OK;
  }
  hipLaunchKernelGGL(composite_loss_kernel, dim3((uint32_t)blocks), dim3(kLossWaves * 64), 0, (hipStream_t)stream, in, gt,
                     1.0f / (float)(R * 3), gscale, loss_out, Cr, d_rgb, d_sigma, keep, (float*)ws);
  hipLaunchKernelGGL(closs_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)ws, (int)blocks, 2.f * (1.0f / (float)(R * 3)), loss_out);
  HBR_RETURN_IF_LAUNCH_FAILED();
  return HBR_OK;
}
