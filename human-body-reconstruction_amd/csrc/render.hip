// hbr_render_fwd: the inference half of vol_render in ONE call (reference vol_renderer.py:141-223 under no_grad, as
// the image-write loop uses it, train_hash2.py:277-292): direction encoding -> K1 (points generated on chip, planar
// features) -> K3 -> K5, enqueued back to back on the caller's stream with the caller's workspace.  Nothing comes
// back to the host between the four launches; the [N,32] features never take the reference's row layout.
#include "hbr_common.h"

namespace {
int64_t up256(int64_t v) { return (v + 255) / 256 * 256; }
struct Layout {
  int64_t pe, feat, out, mlp, total;
};
Layout layout(int64_t R, int64_t S, int L, int precision, int feat_dtype, bool own_out) {
  Layout w;
  const int64_t N = R * S;
  w.pe = 0;
  w.feat = up256(R * 24 * 4);
  w.out = w.feat + up256(N * L * 2 * (feat_dtype == HBR_F32 ? 4 : 2));
  w.mlp = w.out + (own_out ? up256(N * 16) : 0);
  w.total = w.mlp + hbr_mlp_workspace_bytes(precision);
  return w;
}
}  // namespace

extern "C" int64_t hbr_render_fwd_workspace_bytes(int64_t R, int64_t S, int L, int precision, int feat_dtype, int own_out) {
  if (R < 0 || S < 1 || L < 1) return 0;
  return layout(R, S, L, precision, feat_dtype, own_out != 0).total;
}

extern "C" int hbr_render_fwd(const float* rays_o, const float* rays_d, const float* t, const float* dir_norm, int64_t R, int64_t S,
                              const float* tables, const float* scales_host, const float* mu_host, float sigma, int L, int64_t T,
                              int F, const float* params, int precision, int feat_dtype, const uint8_t* keep, float* Cr,
                              float* wts, float* out, void* ws, int64_t ws_bytes, void* stream) {
  if (!rays_o || !rays_d || !t || !tables || !params || !Cr || !ws || R < 0 || S < 1) return HBR_EINVAL;
  if (L != 16 || F != 2) return HBR_EUNSUPPORTED;  // the MLP kernels take 32 input features
  if (((uintptr_t)ws & 255) != 0) return HBR_EINVAL;
  const Layout w = layout(R, S, L, precision, feat_dtype, out == nullptr);
  if (ws_bytes < w.total) return HBR_EWORKSPACE;
  if (R == 0) return HBR_OK;
  char* b = (char*)ws;
  float* pe = (float*)(b + w.pe);
  void* feat = b + w.feat;
  float* o4 = out ? out : (float*)(b + w.out);
  int rc = hbr_dir_encode(rays_d, R, 3, 4, pe, stream);
  if (rc) return rc;
  rc = hbr_hash_encode_fwd(nullptr, rays_o, rays_d, t, R, S, tables, scales_host, mu_host, sigma, L, T, F, feat, HBR_LAYOUT_PLANAR, 0,
                           feat_dtype, stream);
  if (rc) return rc;
  rc = hbr_mlp_fwd(feat, HBR_LAYOUT_PLANAR, 0, feat_dtype, pe, R * S, S, params, precision, o4, keep, b + w.mlp, ws_bytes - w.mlp, stream);
  if (rc) return rc;
  return hbr_composite_fwd(t, 0, o4, 4, o4 + 3, 4, dir_norm, R, S, Cr, wts, stream);
}
