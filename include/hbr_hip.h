/* libhbr_hip.so -- C ABI of the MI355X (gfx950) hash-NeRF render/train hot path.
 *
 * The reference (RishabhSri14/Human-Body-Reconstruction) is 100 % Python and exposes no FFI;
 * its boundary is a set of Python call signatures.  Each entry point below replaces the chain of
 * ATen ops behind one of those calls (reference file:line cited per function).  INTEGRATION.md
 * shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - return 0 on success, a negative HBR_E* code otherwise; never throw, never allocate,
 *     never synchronise: the caller passes outputs and workspace, work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream);
 *   - pointers are DEVICE pointers to contiguous row-major buffers unless the parameter name
 *     ends in `_host`;
 *   - sizes are int64_t; N = R*S points are ordered ray-major (n = r*S + s);
 *   - fp32 arithmetic follows the reference's op order (no FMA contraction on the index path).
 */
#ifndef HBR_HIP_H
#define HBR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: only the entry points declared here are exported. */
#define HBR_API __attribute__((visibility("default")))

#define HBR_VERSION 300 /* 0.3.0: checked algo-3 contract, hidden visibility */

enum {
  HBR_OK = 0,
  HBR_EINVAL = -1,       /* null pointer / negative size / inconsistent shape */
  HBR_EUNSUPPORTED = -2, /* configuration outside what the kernels are built for */
  HBR_ELAUNCH = -3,      /* hipLaunchKernel / hipMemsetAsync reported an error */
  HBR_EWORKSPACE = -4    /* workspace too small (see hbr_workspace_bytes) */
};

enum { HBR_MAX_LEVELS = 32 };

/* feature-buffer layouts produced by the encoder and consumed by the MLP kernels */
enum {
  HBR_LAYOUT_ROWS = 0,  /* y[n*stride + l*F + f]            (what HashEncoder.forward returns) */
  HBR_LAYOUT_PLANAR = 1 /* y[(l*N + n)*F + f]  level-major  (coalesced; internal pipeline)     */
};

enum { HBR_F32 = 0, HBR_BF16 = 1 };
/* OR-ed into hbr_mlp_fwd's / hbr_mlp_bwd's `precision`: the workspace still holds the weight-fragment image that the
 * previous hbr_render_prologue / hbr_mlp_fwd / hbr_mlp_bwd call built from the same `params` at the same precision
 * (prologue, forward, then backward of one training step), so this call need not pack it again. */
enum { HBR_IMAGE_READY = 0x100 };
/* OR-ed into hbr_hash_encode_bwd's `algo` and hbr_mlp_bwd's `precision`: the gradient outputs (dtables / dparams) are
 * WRITTEN instead of accumulated into, so the caller need not zero them first (a training step's 8 MiB memset).
 * hbr_hash_encode_bwd honours it only where every row has exactly one writer - the LDS kernels with the full
 * workspace (algo 0 / 2 / 3 resolving to that) - and returns HBR_EUNSUPPORTED otherwise, before launching anything:
 * the caller then zeroes the buffer and calls again without the flag. */
enum { HBR_OVERWRITE = 0x200 };

HBR_API int hbr_version(void);
HBR_API const char* hbr_strerror(int code);
/* 1 if a HIP device with gcnArchName gfx950 is visible, 0 otherwise */
HBR_API int hbr_device_ok(void);

/* ---- K0: sampling along rays ------------------------------------------------------------------
 * hbr_strat_sample replaces strat_sampler, helper.py:210-237 (non-exp branch):
 *   t[s] = linspace(tn, tf, S)[s] + (u[s] * (tf - tn)) / S  - one jitter per sample index, shared by all rays.
 *   u   DEVICE [S] uniform draws in [0,1), or NULL: drawn on the device by Philox4x32-10 from (seed, offset, s),
 *       24-bit uniforms - reproducible across ranks and runs without shared generator state
 *   t   DEVICE [S] out
 * hbr_occupancy_mask replaces Volume_Renderer.get_mask, vol_renderer.py:133-140:
 *   keep[n] = grid[cx,cy,cz], c = trunc(((p - mu)/sigma_val) * G); points as in K1 (x, or rays_o/rays_d/t)
 *   grid DEVICE [G,G,G] bytes (a torch.bool tensor), read live; keep DEVICE [N] bytes out.
 *   A negative index counts from the end (torch indexing); outside [-G, G) - where torch raises - is "not kept".
 */
HBR_API int hbr_strat_sample(float tn, float tf, int64_t S, const float* u, uint64_t seed, uint64_t offset, float* t,
                     void* stream);
HBR_API int hbr_occupancy_mask(const float* x, const float* rays_o, const float* rays_d, const float* t, int64_t R,
                       int64_t S, const uint8_t* grid, int G, const float* mu_host, float sigma_val,
                       uint8_t* keep, void* stream);

/* ---- a13 / f4: resampling for the hierarchical second pass --------------------------------------------------
 * Replaces hierarchical_sampling, helper.py:23-51 (one wave per ray; torch cumsum / searchsorted / cat / sort otherwise):
 *   weights [R,S] the first pass's compositing weights (negative values count as 0, :36)
 *   z_vals  the first pass's depths: [S] shared (z_stride = 0) or [R, z_stride >= S]
 *   u       [R,S] uniform draws for the inverse-CDF lookup, or NULL: Philox4x32-10 from (seed, offset), stream 1
 *   samples01 [n_samples] uniform draws of the ONE depth vector all rays index (:43-45), or NULL: Philox stream 2
 *   t_fine  [R, 2S] out: sort(cat(z_vals, samples[clamp(searchsorted(cdf, u, right=True), 0, n_samples-1)])) - a ray
 *           gets S new depths (u has the cdf's shape) whatever n_samples is
 * pdf and cdf are summed sequentially in fp32 (as torch's CPU cumsum does): the lookup is discontinuous in the cdf's
 * last bit. */
HBR_API int hbr_hierarchical_resample(const float* weights, const float* z_vals, int64_t z_stride, const float* u,
                                      const float* samples01, uint64_t seed, uint64_t offset, float tn, float tf,
                                      int64_t R, int64_t S, int64_t n_samples, float* t_fine, void* stream);

/* ---- f4: occupancy-grid update ----------------------------------------------------------------------------------
 * Replaces Volume_Renderer.update_grid, vol_renderer.py:116-131, on the [G,G,G] byte grid hbr_occupancy_mask reads:
 * a cell is set to True iff the LAST point (in point order) that falls into it has int8(ceil(max(alpha, 0))) > 0 - the
 * reference's non-accumulating index_put through an int8 scratch array, wrap-around included; if no cell gets set the
 * whole grid becomes True (:126-127).  Cells are never cleared.  Points as in K1 (x, or rays_o / rays_d / t);
 * alpha [N] fp32 (the reference passes the model's density column).  tmp_arr: the reference's [G,G,G] int8 scratch,
 * state carried between calls (a count that wrapped negative survives the reset at :131), or NULL (treated as zeros).
 * ws: hbr_occupancy_update_workspace_bytes(G). */
HBR_API int64_t hbr_occupancy_update_workspace_bytes(int G);
HBR_API int hbr_occupancy_update(const float* x, const float* rays_o, const float* rays_d, const float* t, int64_t R,
                                 int64_t S, const float* alpha, uint8_t* grid, int8_t* tmp_arr, int G,
                                 const float* mu_host, float sigma_val, void* ws, int64_t ws_bytes, void* stream);

/* ---- K0 + a7 + weight packing in ONE launch ---------------------------------------------------------
 * What a render call does before the encoder runs (vol_renderer.py:163,177-181 + the MLP's weight staging):
 *   t      [S] out as hbr_strat_sample(tn, tf, S, u, seed, offset) writes it; NULL: skipped
 *   pe     [R,24] out = hbr_dir_encode(rays_d [R,3], d_model 3, num_freq 4); NULL (with rays_d NULL): skipped
 *   params the MLP's flat parameter block (see hbr_mlp_fwd): its MFMA-fragment image is built in `ws`
 *          (hbr_mlp_workspace_bytes(precision) bytes) exactly as hbr_mlp_fwd builds it; a following hbr_mlp_fwd /
 *          hbr_mlp_bwd on this stream with HBR_IMAGE_READY then skips its own packing launch.  NULL: skipped.
 */
HBR_API int hbr_render_prologue(float tn, float tf, int64_t S, const float* u, uint64_t seed, uint64_t offset, float* t,
                                const float* rays_d, int64_t R, float* pe, const float* params, int precision, void* ws,
                                int64_t ws_bytes, void* stream);

/* ---- K1: multiresolution hash-grid encode --------------------------------------------------
 * Replaces HashEncoder.forward, hash_encoding.py:146-170 (per level: scale :153-154, trunc :157,
 * frac :158, 8 corners :135, spatial hash :49-53, embedding gather :163, trilinear :142-144).
 *   x        [N,3] fp32 points (may be NULL when rays are given, see below)
 *   rays_o/rays_d [R,3], t [S]: if x == NULL the points are generated on chip as
 *            o + d*t (vol_renderer.py:165), N = R*S
 *   tables   [L,T,F] fp32, stacked Embedding_list[l].weight
 *   scales_host [L] fp32 level scales N_l computed by the host exactly as hash_encoding.py:13,153
 *   mu_host  [3], sigma: un_x = ((x-mu)/sigma)*N_l
 *   T        rows per level; any T>=1 (power of two takes the uint32 fast path)
 *   F        features per row: 2
 *   y        output, layout per `layout`; row stride `y_stride` elements for HBR_LAYOUT_ROWS
 *            (>= L*F; extra columns are left untouched -- the reference's E aux columns)
 */
HBR_API int hbr_hash_encode_fwd(const float* x, const float* rays_o, const float* rays_d, const float* t,
                        int64_t R, int64_t S, const float* tables, const float* scales_host,
                        const float* mu_host, float sigma, int L, int64_t T, int F, void* y,
                        int layout, int64_t y_stride, int y_dtype, void* stream);

/* ---- K2: gradient scatter-add into the tables ------------------------------------------------
 * Replaces autograd of the above (16x aten::embedding_dense_backward + mul/sum backward).
 *   dy       same layout/dtype conventions as y
 *   dy_absmax optional DEVICE [L] fp32: max |dy| per level (e.g. produced by the kernel that wrote dy);
 *            NULL => computed here by one extra pass over dy.  Must be >= the true maximum.
 *   dtables  [L,T,F] fp32, ACCUMULATED INTO (caller zeroes it when a fresh gradient is wanted)
 *   algo     1 = one global float atomic per corner-feature (no workspace; any T; order-dependent rounding)
 *            2 = LDS-partitioned accumulation in 64-bit fixed point (scale 2^k per level from max |dy|, so every
 *                contribution is rounded once and the integer sums are order-independent): needs `ws`, T <= 2^28.
 *                With the full workspace the chunk partials are reduced in a fixed order and the result of a launch
 *                is bitwise reproducible; with only the first hbr_hash_bwd_workspace_bytes_min() bytes they are
 *                added with float atomics (same values up to the order of <= chunks fp32 additions per entry).
 *            3 = 2, re-using the normalised coordinates the PREVIOUS algo-2 call left in `ws` (a launch over another
 *                range of levels - the multi-GPU trainer scatters the levels in two calls so that each half's
 *                all-reduce overlaps the other half's kernel).  The contract is CHECKED: the library records, per `ws`
 *                pointer, which points the last algo-2 call normalised into it (x / rays_o / rays_d / t pointers, R, S,
 *                mu, sigma) and on which stream; algo 3 returns HBR_EINVAL unless this call names exactly those - e.g.
 *                when another caller ran an algo-2 backward on the same workspace in between.  (What it cannot see: a
 *                kernel of another stream overwriting `ws`; a workspace belongs to one stream.)
 *            0 = auto: 2 when N >= 4096 (where the two cost the same; 65536 until round 4), T <= 2^28 and the workspace suffices, else 1
 *   ws       16-byte-aligned scratch of hbr_hash_bwd_workspace_bytes() bytes (normalised coordinates, per-level
 *            maxima, per-chunk partial tables); contents are dead after the call
 *   errors   algo 2 without enough workspace -> HBR_EWORKSPACE; algo 2 with T > 2^28 -> HBR_EUNSUPPORTED
 */
HBR_API int hbr_hash_encode_bwd(const float* x, const float* rays_o, const float* rays_d, const float* t,
                        int64_t R, int64_t S, const void* dy, int layout, int64_t dy_stride,
                        int dy_dtype, const float* dy_absmax, const float* scales_host,
                        const float* mu_host, float sigma, int L, int64_t T, int F, float* dtables,
                        int algo, void* ws, int64_t ws_bytes, void* stream);
/* full workspace of algo 2 for this shape (0 when `algo`/N/T select the global-atomics kernel) */
HBR_API int64_t hbr_hash_bwd_workspace_bytes(int64_t N, int L, int64_t T, int F, int algo);
/* the part of it algo 2 cannot run without (coordinates + maxima, 12 B per point) */
HBR_API int64_t hbr_hash_bwd_workspace_bytes_min(int64_t N, int L, int64_t T, int F, int algo);

/* ---- K5: alpha compositing along rays ---------------------------------------------------------
 * Replaces calc_color, helper.py:53-107 (non-SDF branch).
 *   t [S] sample depths shared by all rays (t_stride = 0), or per-ray t [R, t_stride >= S] (the hierarchical
 *   pass, vol_renderer.py:242); rgb [R,S,3]; sigma [R,S]; dir_norm [R] (NULL => 1)
 *   rgbs: alternatively rgb and sigma interleaved as the MLP's [R*S,4] (r,g,b,sigma) output:
 *         pass rgb = out, sigma = out+3 and elem strides rgb_stride = sigma_stride = 4
 *   Cr [R,3]; wts [R,S] (may be NULL)
 */
HBR_API int hbr_composite_fwd(const float* t, int64_t t_stride, const float* rgb, int64_t rgb_stride, const float* sigma,
                      int64_t sigma_stride, const float* dir_norm, int64_t R, int64_t S, float* Cr,
                      float* wts, void* stream);
/* d_rgb / d_sigma use the same strides as their forward counterparts.
 * keep: optional DEVICE [R*S] bytes (hbr_occupancy_mask): gradients of samples with keep == 0 are written as 0
 * (vol_renderer.py:209-221: their sigma/rgb are zeros, not model outputs) */
HBR_API int hbr_composite_bwd(const float* t, int64_t t_stride, const float* rgb, int64_t rgb_stride, const float* sigma,
                      int64_t sigma_stride, const float* dir_norm, int64_t R, int64_t S,
                      const float* dCr, float* d_rgb, float* d_sigma, const uint8_t* keep, void* stream);

/* ---- K5 forward + a11 + K5 backward in ONE launch (the fused training step) ----------------------------------
 * hbr_composite_fwd, hbr_mse2_loss_fwd_bwd and hbr_composite_bwd for the case Cf is Cr (hierarchical off): a ray's
 * colour, its share of loss = 2*mean((Cr-gt)^2) and the gradient dCr = gscale*4*(Cr-gt)/(3R) flowing back into its own
 * samples are computed by one wave in one pass.  Arguments as in those three calls, except:
 *   loss_out  one fp32, WRITTEN (block sums added in a fixed order by a one-wave second launch: bitwise reproducible)
 *   Cr        [R,3] out, or NULL when the colours themselves are not needed
 *   ws        hbr_composite_loss_workspace_bytes(R) bytes of scratch (the block sums), dead after the call
 */
HBR_API int64_t hbr_composite_loss_workspace_bytes(int64_t R);
HBR_API int hbr_composite_loss_fwd_bwd(const float* t, int64_t t_stride, const float* rgb, int64_t rgb_stride,
                                       const float* sigma, int64_t sigma_stride, const float* dir_norm, int64_t R,
                                       int64_t S, const float* gt, float gscale, float* loss_out, float* Cr,
                                       float* d_rgb, float* d_sigma, const uint8_t* keep, void* ws, void* stream);

/* ---- a7: view-direction encoding ---------------------------------------------------------------
 * Replaces PositionalEncoder.forward, encoder.py:25-32: for each row and coordinate c,
 * out[row, c*2nf + k] = sin(2*x_c*k), out[row, c*2nf + nf + k] = cos(2*x_c*k), k = 0..nf-1.
 *   x [rows, d_model] fp32 -> out [rows, d_model*2*num_freq] fp32 */
HBR_API int hbr_dir_encode(const float* x, int64_t rows, int d_model, int num_freq, float* out, void* stream);

/* ---- K3/K4: fused density/colour MLP on the matrix cores --------------------------------------
 * Replaces MLP_3D.forward (test_hash.py:52-72) for the instance built at train_hash2.py:127
 * (32 -> 64 -> 64 -> 16 ; 15+24 -> 64 -> 64 -> 3; LeakyReLU(0.01) density, ELU rgb).
 *   feat     [N,32] features (layout/dtype as produced by K1; HBR_LAYOUT_ROWS needs stride % 4 == 0)
 *   viewdirs_enc [G,24] fp32 encoded view directions (hbr_dir_encode with d_model 3, num_freq 4);
 *            point n uses row n / group  (group = S for one row per ray, 1 for one row per point)
 *   params   fp32 flat parameter block, nn.Linear layout, in this order:
 *            sig0.W[64,32] sig0.b[64] sig2.W[64,64] sig2.b[64] sig4.W[16,64] sig4.b[16]
 *            col0.W[64,39] col0.b[64] col2.W[64,64] col2.b[64] col4.W[3,64] col4.b[3]   (14227 floats)
 *   precision HBR_F32 (exact-fp32 MFMA, v_mfma_f32_32x32x2_f32) or HBR_BF16 (bf16 operands, fp32 accumulate)
 *   out      [N,4] fp32 (r,g,b,sigma)
 *   keep     (hbr_mlp_fwd) optional DEVICE [N] bytes (hbr_occupancy_mask): out rows with keep == 0 are written as zeros
 *   ws       scratch of hbr_mlp_workspace_bytes(precision) bytes (about 28 MB), 16-byte aligned: the weights
 *            re-packed in MFMA-fragment order (rebuilt on every call, nothing is cached), followed by the
 *            backward's per-workgroup weight-gradient slabs and per-wave feature-gradient maxima (written and
 *            reduced within one hbr_mlp_bwd call)
 */
enum { HBR_MLP_PARAM_FLOATS = 14227 };
HBR_API int64_t hbr_mlp_workspace_bytes(int precision);
HBR_API int hbr_mlp_fwd(const void* feat, int layout, int64_t feat_stride, int feat_dtype,
                const float* viewdirs_enc, int64_t N, int64_t group, const float* params,
                int precision, float* out, const uint8_t* keep, void* ws, int64_t ws_bytes, void* stream);
/* backward: recomputes the forward activations from `feat`, then
 *   dout     [N,4] fp32 gradient of out
 *   dfeat    gradient wrt feat (same layout/dtype and the same row stride `feat_stride` as feat); may be NULL
 *   dfeat_absmax optional DEVICE [16] fp32: max |dfeat| per level (of the values as stored), for hbr_hash_encode_bwd's
 *            dy_absmax - saves that call its own pass over dfeat; NULL or dfeat == NULL => not produced
 *   dparams  [14227] fp32, ACCUMULATED INTO (fixed summation order: bitwise reproducible)
 */
HBR_API int hbr_mlp_bwd(const void* feat, int layout, int64_t feat_stride, int feat_dtype,
                const float* viewdirs_enc, int64_t N, int64_t group, const float* params,
                int precision, const float* dout, void* dfeat, float* dfeat_absmax, float* dparams,
                void* ws, int64_t ws_bytes, void* stream);

/* ---- the MLP + compositing + loss of a training step, forward AND backward, in one call (round 4) -----------------
 * hbr_mlp_fwd + hbr_composite_loss_fwd_bwd + hbr_mlp_bwd fused: replaces MLP_3D.forward (test_hash.py:52-72), calc_color
 * (helper.py:53-107), loss = MSE(Cr, gt) + MSE(Cf, gt) with Cf = Cr (train_hash2.py:221) and their autograd.  The backward
 * kernel recomputes the forward of its 32-point tile anyway; with whole rays inside a workgroup round (S in {32, 64, 128},
 * shared depths t[S]) it composites them in LDS and forms d out itself - the forward launch, the compositing launches and
 * the [N,4] out / d out buffers disappear.  Same arithmetic as the three calls (compositing: the same operations in the
 * same order; the MLP forward is the backward kernel's recompute, i.e. the bias enters by an extra MFMA k-step: colours
 * agree to ~1e-6 relative).
 *   feat, layout, feat_stride, feat_dtype, viewdirs_enc [R,24], params, dfeat, dfeat_absmax, dparams, ws: as hbr_mlp_bwd
 *            (N = R * S points, ray-major; group = S); precision: HBR_BF16 (| HBR_IMAGE_READY | HBR_OVERWRITE)
 *   t [S], dir_norm [R] or NULL, gt [R,3], gscale: as hbr_composite_loss_fwd_bwd
 *   loss_out one fp32, WRITTEN (fixed summation order: bitwise reproducible);  Cr [R,3] out or NULL
 *   returns HBR_EUNSUPPORTED - before launching anything - for fp32 precision, the rows layout or another S: the caller
 *   then issues the three separate calls.
 */
HBR_API int hbr_mlp_render_bwd(const void* feat, int layout, int64_t feat_stride, int feat_dtype, const float* viewdirs_enc,
                       int64_t R, int64_t S, const float* params, int precision, const float* t, const float* dir_norm,
                       const float* gt, float gscale, float* loss_out, float* Cr, void* dfeat, float* dfeat_absmax,
                       float* dparams, void* ws, int64_t ws_bytes, void* stream);

/* ---- vol_render, inference half, in one call ----------------------------------------------------
 * Replaces vol_renderer.py:141-223 under no_grad (the image-write loop, train_hash2.py:277-292; hierarchical off):
 * direction encoding -> K1 (points o + d*t generated on chip, planar features in `feat_dtype`) -> K3 -> K5, enqueued
 * back to back on `stream`.  Shared depths t [S]; num_freq = 4, L = 16, F = 2 (the MLP's 32 + 24 inputs).
 *   keep   optional [R*S] bytes from hbr_occupancy_mask;  dir_norm [R] or NULL (=1)
 *   Cr [R,3] out;  wts [R,S] out or NULL;  out [R*S,4] (r,g,b,sigma) out, or NULL to keep it inside the workspace
 *   ws     256-byte-aligned scratch of hbr_render_fwd_workspace_bytes(R, S, L, precision, feat_dtype, out == NULL)
 */
HBR_API int64_t hbr_render_fwd_workspace_bytes(int64_t R, int64_t S, int L, int precision, int feat_dtype, int own_out);
HBR_API int hbr_render_fwd(const float* rays_o, const float* rays_d, const float* t, const float* dir_norm, int64_t R,
                   int64_t S, const float* tables, const float* scales_host, const float* mu_host, float sigma,
                   int L, int64_t T, int F, const float* params, int precision, int feat_dtype,
                   const uint8_t* keep, float* Cr, float* wts, float* out, void* ws, int64_t ws_bytes,
                   void* stream);

/* ---- a11: loss + its gradient -----------------------------------------------------------------
 * train_hash2.py:177,221 with hierarchical off: loss = 2*mean((Cr-gt)^2); dCr = 4*(Cr-gt)/(3R) * gscale.
 * loss_out: one fp32, accumulated into (caller zeroes).
 * ws: optional scratch of hbr_mse2_workspace_bytes() bytes whose LAST 4 bytes are zero on first use (the kernel
 *     leaves them zero): block sums are then added in a fixed order and the loss is bitwise reproducible.
 *     NULL => one float atomic per block.  */
HBR_API int64_t hbr_mse2_workspace_bytes(void);
HBR_API int hbr_mse2_loss_fwd_bwd(const float* Cr, const float* gt, int64_t R, float gscale, float* loss_out,
                          float* dCr, void* ws, void* stream);

/* ---- a12: dense Adam / AdamW over a flat fp32 buffer -----------------------------------------
 * torch.optim.Adam / AdamW single-tensor semantics (train_hash2.py:141-142): decoupled weight decay
 * p *= 1-lr*wd (AdamW; wd = 0 for Adam), m,v EMA, bias correction with `step` (1-based),
 * p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps).  grad_scale multiplies g first (1/world_size after
 * the all-reduce). */
HBR_API int hbr_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                  void* stream);

/* The same for up to four parameter segments in ONE launch (a training step's Adam on the tables + AdamW on the MLP,
 * train_hash2.py:227-228): segs_host is a HOST array of nseg descriptors, field meanings as hbr_adam_step's arguments. */
typedef struct HbrAdamSegment {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
  float lr, beta1, beta2, eps, weight_decay;
  int64_t step;
  float grad_scale;
} HbrAdamSegment;
HBR_API int hbr_adam_step_multi(int nseg, const HbrAdamSegment* segs_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HBR_HIP_H */
