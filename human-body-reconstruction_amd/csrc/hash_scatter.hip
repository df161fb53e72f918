// K2: gradient scatter-add into the hash tables (autograd of reference hash_encoding.py:146-170, i.e. 16 x
// aten::embedding_dense_backward + the mul/sum backward), gfx950.  Entry point: hbr_hash_encode_bwd.
//
// algo 1 (hash_encode.hip): one global float atomic per corner-feature.  ~55 ms at the README batch - memory-side
//   atomics on scattered rows run at ~10 G/s - kept for fewer than 4096 points and as a cross-check.
// algo 2 (this file): accumulate in LDS, in 64-bit FIXED POINT.
//   Measured on MI355X (tools/lds_atomic_bench2.hip), cycles per wave-instruction at all / a quarter of the lanes:
//     ds_add_f32 193 / 48 (3 cycles per active lane whatever the FP mode; ds_pk_add_bf16/f16 and the rtn form the same),
//     ds_add_f64 20.7 / 10.3,  ds_add_u64 10.5 / 7.0,  ds_add_u32 5.8 / 4.5.
//   Every contribution w*dy is scaled by a per-level power of two 2^k (from the level's max |dy|) and rounded to an
//   integer exactly once; integer addition is associative, so the result of a launch does not depend on the order in
//   which lanes, waves, workgroups or XCDs happen to run - bitwise reproducible, which float atomics cannot give.
//   |dy * 2^k| < 2^fixbits with fixbits = 62 - max(18, ceil(log2 N)): a point's corner weights sum to 1, so no sum of
//   any subset of the launch's contributions reaches 2^62.  The quantum is 2^-fixbits of the level's largest
//   gradient (2^-41 at N = 2 M): far below an fp32 ulp of anything that is not itself 2^-17 of that maximum.
//
//   Two kernels share the levels, decided ON THE DEVICE from the cell bounding box of the launch's points:
//   * dense (coarse levels: the box of touched vertices fits in LDS, <= 10 000 vertices with both features or
//     20 000 with one): ONE visit per point-level.  A lane walks 16 consecutive samples, keeps the eight corner sums
//     of the cell it is in in registers and flushes them into a dense vertex table in LDS when the cell changes
//     (coarse cells hold up to 25 consecutive samples of a ray).  The vertex table is hashed into rows only at the end.
//   * hashed slices (all other levels): a workgroup owns one feature of a 16384-row slice (128 KiB of 8-byte
//     accumulators) and sweeps a chunk of the points; a point is visited by the 4 x 2 slice/feature owners of its
//     level, each keeping the corners that fall into its slice.  The x term of the hash is the cell coordinate
//     itself, so the two x-neighbours of a (y, z) pair are always in the slice together: four tests, not eight.
//   Chunk partials leave the workgroups as plain stores and are summed in a fixed order (slab_reduce_kernel).
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "hash_common.h"
#include "wave_reduce.h"

namespace hbr {

constexpr int kSliceLog2 = 14;                 // 16384 rows * 8 B = 128 KiB of the CU's 160 KiB LDS
constexpr int kSliceRows = 1 << kSliceLog2;
constexpr uint32_t kSliceBytes = (uint32_t)kSliceRows << 3;
constexpr int kLdsBwdThreads = 1024;
constexpr int kSeg = kLdsBwdThreads / 64;      // 16: samples a lane of the dense kernel walks = waves of a stripe
constexpr int64_t kMaxLdsT = 1LL << 28;        // row offsets are shifted left by 3 in 32 bits
constexpr int kChunkPointsLog2Max = 19;        // a hashed-slice workgroup sweeps fewer than 2^19 points
constexpr double kRoundMagic = 6755399441055744.0;  // 1.5 * 2^52: x + magic has round-to-nearest(x) in its low mantissa bits
constexpr uint32_t kRoundMagicHi = 0x43380000u;      // its high word (the low word is 0)
constexpr int kAbsBlocks = 512;                // partial maxima per level (absmax kernels' grid.x)
// `algo 0` (auto) takes the LDS kernels from this many points.  Until round 4 it was 65 536 ("enough points to amortise the
// 128 KiB flush per workgroup") - never measured: global float atomics cost 24 ns per point whatever N, the LDS kernels
// ~0.06 ms up to 16 Ki points and 0.08 / 0.12 ms at 32 / 64 Ki, so at 8 / 32 Ki points auto ran 3.5x / 9x slower than it had to
// (tools/k2_small_n.py: 0.204 / 0.764 ms against 0.058 / 0.083; the two meet at 2048 points).
constexpr uint32_t kLdsAutoMinPoints = 4096;
constexpr int kDenseLevels = 6;                // dense levels are a prefix of the levels; at most this many
constexpr int kDenseCap = 10000;               // vertices of a dense table with both features (160 000 B of LDS)
constexpr int64_t kDenseMaxT = 1LL << 22;      // the per-level int64 row table of the dense path is T*16 B (cleared and re-read per call)
// Tables of 16 or more slices per level (train_hash2.py:36 --hash_size 18 and up) take the MASKED form of the
// hashed branch: a small kernel first records, per (level, slice, 64-point step), which points have a corner in the
// slice; a slice owner then loads its chunk's coordinates as before but VISITS only the flagged points, compacted
// through a per-wave LDS ring - 4 (1 - (1 - 1/spl)^4) / spl of the points instead of all of them (12 % at spl = 32).
// The unmasked form is linear in T: 0.52 / 0.94 / 1.65 / 4.24 / 7.44 ms at T = 2^16 .. 2^20 (profiles/r04_k2_vs_T.txt);
// masked: 1.20 / 1.87 / 3.08 ms at 2^18 .. 2^20.  What is left is bandwidth, not arithmetic: every slice owner still
// STREAMS its chunk's coordinates and dy (16 B per point) to pick its 12 % - 64 owners per level at 2^19 = 33 GB of
// L2 -> L1 traffic per call; a 128-byte line holds ten points, so gathering only the flagged ones would fetch 3/4 of
// the lines anyway, and routing (x, y, z, dy) payloads to the owners through memory is 6 GB of HBM traffic.
constexpr int kMaskMinSlices = 16, kMaskMaxSlices = 256;  // (measured at 8 slices, T = 2^17: 0.973 ms masked, 0.935 unmasked)
constexpr int kRing = 128;                     // entries (16 B) of a wave's ring: a step adds <= 64, a pop takes 64
constexpr uint32_t kMaskedLdsBytes = kSliceBytes + (kLdsBwdThreads / 64) * kRing * 16;  // 160 KiB: the CU's whole LDS
static_assert(kMaskedLdsBytes <= 160 * 1024, "LDS of a gfx950 CU");
constexpr int kDenseStripesPerWg = 64;         // 64 Ki points per dense workgroup (measured at N = 2M: 16 -> 0.663 ms, 32 -> 0.633, 64 -> 0.617, 128 -> 0.629)

// ------------------------------------------------------------------------------------------------
// per-launch facts the kernels agree on, kept in the workspace
// ------------------------------------------------------------------------------------------------
struct Meta {
  uint32_t absmax[HBR_MAX_LEVELS];  // max |dy| per level, fp32 bit pattern
  float lo[3], hi[3];               // bounding box of the normalised coordinates (x - mu) / sigma
  uint32_t finite;                  // 1 if every normalised coordinate is finite
  uint32_t pad[64 - HBR_MAX_LEVELS - 7];
};
static_assert(sizeof(Meta) == 256, "Meta block");

// fixed-point scale of a level from its max |dy| (bit pattern `ab`): |dy| < 2^(e+1) => k = fixbits - 1 - e.
// 2^k must be an fp32 normal, so k is clamped to [-126, 127] (a level whose largest gradient is below 2^-84 keeps
// fewer fractional bits - 20 orders of magnitude under Adam's epsilon).
struct FixScale {
  float mul;      // 2^k
  double inv;     // 2^-k
  int state;      // 0: all-zero gradient, 1: finite, 2: non-finite (NaN/inf somewhere in dy)
};
__device__ __forceinline__ FixScale fix_scale(uint32_t ab, int fixbits) {
  FixScale s;
  const int be = (int)(ab >> 23);                  // biased exponent
  s.state = ab == 0 ? 0 : (be == 255 ? 2 : 1);
  int k = fixbits - 1 - (be - 127);
  k = k > 127 ? 127 : (k < -126 ? -126 : k);
  s.mul = __uint_as_float((uint32_t)(k + 127) << 23);
  s.inv = __longlong_as_double((long long)(1023 - k) << 52);
  return s;
}

// round(v) of a double as a 64-bit integer: v + magic lands in [2^52, 2^53), where the mantissa counts in ones
// (two's complement around 1.5 * 2^52), and the magic's low word is zero - one fp64 op + one integer add.
__device__ __forceinline__ unsigned long long fix_bits(double t) {
  return (unsigned long long)__double_as_longlong(t) - ((unsigned long long)kRoundMagicHi << 32);
}
__device__ __forceinline__ unsigned long long fix_product(double xy, double zd) { return fix_bits(__fma_rn(xy, zd, kRoundMagic)); }
__device__ __forceinline__ unsigned long long fix_value(float v) { return fix_bits((double)v + kRoundMagic); }

// How level l is handled: by the dense kernel (box of touched vertices [x0, x0+dx) x ... fits in LDS; `split`: one
// feature per workgroup) or by the hashed-slice kernel.  Cells are trunc(n * N_l), monotonic in n, so the box of the
// normalised coordinates bounds the cells; a corner adds one per axis.
struct LevelPlan {
  FixScale fs;
  bool dense, split;
  int x0, y0, z0, dx, dy, dz, V;
};
__device__ __forceinline__ LevelPlan level_plan(const Meta* __restrict__ m, const HashGeom& g, int l, int fixbits, int dense_levels) {
  LevelPlan p;
  p.fs = fix_scale(m->absmax[l], fixbits);
  p.dense = p.split = false;
  p.x0 = p.y0 = p.z0 = p.dx = p.dy = p.dz = p.V = 0;
  if (l >= dense_levels || g.T > kDenseMaxT || p.fs.state != 1 || !m->finite) return p;
  const float s = g.scale[l];
  const float ulo[3] = {__fmul_rn(m->lo[0], s), __fmul_rn(m->lo[1], s), __fmul_rn(m->lo[2], s)};
  const float uhi[3] = {__fmul_rn(m->hi[0], s), __fmul_rn(m->hi[1], s), __fmul_rn(m->hi[2], s)};
  long long dims[3];
  int c0[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (!(fabsf(ulo[a]) < 1.0e9f) || !(fabsf(uhi[a]) < 1.0e9f)) return p;
    c0[a] = (int)ulo[a];
    dims[a] = (long long)(int)uhi[a] - c0[a] + 2;
    if (dims[a] < 2 || dims[a] > 2 * kDenseCap) return p;
  }
  const long long V = dims[0] * dims[1] * dims[2];
  if (V > 2 * kDenseCap) return p;
  p.dense = true;
  p.split = V > kDenseCap;
  p.x0 = c0[0]; p.y0 = c0[1]; p.z0 = c0[2];
  p.dx = (int)dims[0]; p.dy = (int)dims[1]; p.dz = (int)dims[2];
  p.V = (int)V;
  return p;
}

// ------------------------------------------------------------------------------------------------
// prologue
// ------------------------------------------------------------------------------------------------
// (x - mu) / sigma once per point instead of once per visit (three IEEE divisions + the point generation), stored in
// the order the scatter kernels consume it: entry [stripe*1024 + w*64 + i] holds point stripe*1024 + 16*i + w, i.e.
// lane i of wave w of the hashed kernel / sample w of lane i's segment in the dense kernel - fully coalesced 12-byte
// reads in both.  Consecutive samples of a ray share their cell at the coarse levels, and 64 lanes adding to one LDS
// address serialise; with lanes 16 samples apart they do not (level 0 cost 3.2x a fine level with the natural map).
// The same launch takes the bounding box of its 1024 coordinates: part [block][7] = lo xyz, hi xyz, finite (round 2
// first took the boxes of the rays' end points in a launch of its own; per point they are exact, cost nothing next to
// the 12 bytes the thread stores, and that launch is gone).
// It also clears the dense levels' int64 row tables (`zero`, `zero_vec` 16-byte vectors; 6 MiB at L = 16, T = 2^16),
// a grid-stride loop of plain stores next to the 24 MB it writes anyway - instead of a 5 us memset launch.
__global__ __launch_bounds__(1024) void normalise_kernel(PointSrc ps, uint32_t N, HashGeom g, float* __restrict__ out,
                                                         float* __restrict__ part, uint4* __restrict__ zero, size_t zero_vec) {
  __shared__ float red[16][7];
  for (size_t i = (size_t)blockIdx.x * 1024u + threadIdx.x; i < zero_vec; i += (size_t)gridDim.x * 1024u) zero[i] = make_uint4(0u, 0u, 0u, 0u);
  const uint32_t base = blockIdx.x * 1024u;
  const uint32_t n_raw = base + (threadIdx.x & 63u) * 16u + (threadIdx.x >> 6);
  const uint32_t n = min(n_raw, N - 1);
  float px, py, pz, nx, ny, nz;
  load_point(ps, n, px, py, pz);
  normalise(g, px, py, pz, nx, ny, nz);
  float* q = out + (size_t)(base + threadIdx.x) * 3;
  q[0] = nx; q[1] = ny; q[2] = nz;
  auto lo = [](float a, float b) { return fminf(a, b); };
  auto hi = [](float a, float b) { return fmaxf(a, b); };
  if (ps.x == nullptr) {
    // Ray-generated points: a coordinate fl(o + fl(d t)) is monotonic in t, so a ray's extreme coordinates are those of
    // its samples at min t and max t - whatever the order of t[] - and the stripe's box is the box of 2 x (rays in the
    // stripe) end points.  ONE wave computes them (and the finiteness of t); the other fifteen skip the seven 64-lane
    // reductions that were half of this kernel's instructions (17.5 -> ~12 us at the README batch).  The end points are
    // whole-ray extremes: a ray only partly in this stripe widens this stripe's box, not the launch's (every ray is in it whole).
    if (threadIdx.x >= 64) return;
    const uint32_t lane = threadIdx.x;
    float tlo = __uint_as_float(0x7f800000u), thi = -tlo, fin = 1.f;
    for (uint32_t s2 = lane; s2 < ps.S; s2 += 64) {
      const float tt = ps.t[s2];
      tlo = fminf(tlo, tt); thi = fmaxf(thi, tt);
      fin = isfinite(tt) ? fin : 0.f;
    }
    tlo = wave_reduce(tlo, lo); thi = wave_reduce(thi, hi); fin = wave_reduce(fin, lo);
    const uint32_t last = min(base + 1023u, N - 1);
    const uint32_t r0 = (__umulhi(base, ps.magic) + base) >> ps.shift, r1 = (__umulhi(last, ps.magic) + last) >> ps.shift;
    const float inf = __uint_as_float(0x7f800000u);
    float v[7] = {inf, inf, inf, -inf, -inf, -inf, fin};
    for (uint32_t k = lane; k < 2u * (r1 - r0 + 1u); k += 64) {
      const uint32_t r = r0 + (k >> 1);
      const float tt = (k & 1u) ? thi : tlo;
      const float* o = ps.o + (size_t)r * 3;
      const float* d = ps.d + (size_t)r * 3;
      float ex, ey, ez;  // load_point's arithmetic (vol_renderer.py:165), then normalise's
      normalise(g, __fadd_rn(o[0], __fmul_rn(d[0], tt)), __fadd_rn(o[1], __fmul_rn(d[1], tt)), __fadd_rn(o[2], __fmul_rn(d[2], tt)), ex, ey, ez);
      v[0] = fminf(v[0], ex); v[1] = fminf(v[1], ey); v[2] = fminf(v[2], ez);
      v[3] = fmaxf(v[3], ex); v[4] = fmaxf(v[4], ey); v[5] = fmaxf(v[5], ez);
      v[6] = (isfinite(ex) && isfinite(ey) && isfinite(ez)) ? v[6] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = wave_reduce(v[k], lo);
#pragma unroll
    for (int k = 3; k < 6; ++k) v[k] = wave_reduce(v[k], hi);
    v[6] = wave_reduce(v[6], lo);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 7; ++k) part[(size_t)blockIdx.x * 7 + k] = v[k];
    }
    return;
  }
  // explicit points: every wave reduces its 64, then one more row reduction over the 16 waves
  // fminf / fmaxf skip a NaN: `finite` records that the box does not cover such a point
  float v[7] = {nx, ny, nz, nx, ny, nz, (isfinite(nx) && isfinite(ny) && isfinite(nz)) ? 1.f : 0.f};
#pragma unroll
  for (int k = 0; k < 3; ++k) v[k] = wave_reduce(v[k], lo);  // DPP: 42 __shfl_xor (ds_bpermute) made the kernel LDS-bound
#pragma unroll
  for (int k = 3; k < 6; ++k) v[k] = wave_reduce(v[k], hi);
  v[6] = wave_reduce(v[6], lo);
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 7; ++k) red[threadIdx.x >> 6][k] = v[k];
  }
  __syncthreads();
  // the 16 waves' values of entry k meet in one row of 16 lanes (threads 16 k .. 16 k + 15): one more row reduction
  if (threadIdx.x < 128) {  // whole waves, so every row is fully active (row 7 idles on a copy of entry 6)
    const int k = min((int)(threadIdx.x >> 4), 6), w = threadIdx.x & 15;
    const float x = red[w][k];
    const float r = (k >= 3 && k < 6) ? row_reduce(x, hi) : row_reduce(x, lo);
    if (w == 0 && threadIdx.x < 112) part[(size_t)blockIdx.x * 7 + k] = r;
  }
}

// max |dy| per level as fp32 bit patterns (non-negative floats order like unsigned integers; a NaN lands above inf,
// so a non-finite gradient is seen as such).  No atomics: 16 counters share a cache line and memory-side atomics
// on one line serialise (32 768 of them took 0.37 ms); every block stores its maximum to part[l][block].
__device__ __forceinline__ uint32_t absbits_max(uint32_t m, uint32_t w, bool bf16) {
  if (bf16) return max(m, max((w << 16) & 0x7fffffffu, w & 0x7fff0000u));
  return max(m, w & 0x7fffffffu);
}
__device__ __forceinline__ void absmax_store(uint32_t m, uint32_t* __restrict__ part) {
  __shared__ uint32_t red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[(size_t)blockIdx.y * kAbsBlocks + blockIdx.x] = max(max(red[0], red[1]), max(red[2], red[3]));
}
// planar [L][N][2]: a level is one contiguous run of 2N elements, read as 16-byte vectors, four in flight per thread
template <int DTYPE>
__global__ __launch_bounds__(256) void absmax_planar_kernel(const void* __restrict__ dy, uint32_t N, uint32_t* __restrict__ part) {
  const int l = blockIdx.y;
  constexpr int kElem = DTYPE == HBR_F32 ? 4 : 2;
  constexpr bool kBf = DTYPE == HBR_BF16;
  const size_t level_bytes = (size_t)N * 2 * kElem;  // a multiple of 4
  const char* base = (const char*)dy + (size_t)l * level_bytes;
  const size_t head = (16 - ((uintptr_t)base & 15)) & 15;  // peel to 16-byte alignment
  const size_t nvec = level_bytes > head ? (level_bytes - head) / 16 : 0;
  const uint4* v = (const uint4*)(base + head);
  uint32_t m = 0;
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < nvec; i += 4 * stride) {
    const uint4 a = v[i], b = v[i + stride], c = v[i + 2 * stride], d = v[i + 3 * stride];
    const uint32_t w[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
#pragma unroll
    for (int k = 0; k < 16; ++k) m = absbits_max(m, w[k], kBf);
  }
  for (; i < nvec; i += stride) {
    const uint4 a = v[i];
    m = absbits_max(absbits_max(absbits_max(absbits_max(m, a.x, kBf), a.y, kBf), a.z, kBf), a.w, kBf);
  }
  if (blockIdx.x == 0 && threadIdx.x < 8) {  // the < 16-byte head and tail, as 4-byte words
    const size_t tail0 = head + nvec * 16;
    const size_t off = threadIdx.x < 4 ? (size_t)threadIdx.x * 4 : tail0 + (size_t)(threadIdx.x - 4) * 4;
    const bool in = threadIdx.x < 4 ? off < head && off < level_bytes : off < level_bytes;
    if (in) m = absbits_max(m, *(const uint32_t*)(base + off), kBf);
  }
  absmax_store(m, part);
}
template <int LAYOUT, int DTYPE>
__global__ __launch_bounds__(256) void absmax_kernel(const void* __restrict__ dy, uint32_t N, int64_t dy_stride,
                                                     uint32_t* __restrict__ part) {
  const int l = blockIdx.y;
  uint32_t m = 0;
  for (uint32_t n = blockIdx.x * 256u + threadIdx.x; n < N; n += gridDim.x * 256u) {
    const uint2 raw = load_feat_raw<LAYOUT, DTYPE>(dy, n, l, N, dy_stride);
    float d0, d1;
    decode_feat<DTYPE>(raw, d0, d1);
    m = max(m, max(__float_as_uint(d0) & 0x7fffffffu, __float_as_uint(d1) & 0x7fffffffu));
  }
  absmax_store(m, part);
}

// one workgroup: partial maxima / boxes -> Meta.  `given` (optional): per-level maxima handed in by the caller.
// Wave w reduces the maxima of levels w, w + 16 (no workgroup barrier); then all threads reduce the boxes.
__global__ __launch_bounds__(1024) void meta_reduce_kernel(const uint32_t* __restrict__ abs_part, int abs_blocks,
                                                           const float* __restrict__ given, const float* __restrict__ bounds_part,
                                                           uint32_t nboxes, int L, Meta* __restrict__ meta) {
  __shared__ float red[16][7];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int l = wv; l < HBR_MAX_LEVELS; l += 16) {
    uint32_t m = 0;
    if (l < L) {
      if (given) m = __float_as_uint(given[l]) & 0x7fffffffu;
      else for (int b = lane; b < abs_blocks; b += 64) m = max(m, abs_part[(size_t)l * kAbsBlocks + b]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    if (lane == 0) meta->absmax[l] = m;
  }
  const float inf = __uint_as_float(0x7f800000u);
  float v[7] = {inf, inf, inf, -inf, -inf, -inf, 1.f};
  for (uint32_t s = threadIdx.x; s < nboxes; s += 1024) {  // the stripes' boxes (normalise_kernel)
    const float* p = bounds_part + (size_t)s * 7;
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = fminf(v[k], p[k]);
#pragma unroll
    for (int k = 3; k < 6; ++k) v[k] = fmaxf(v[k], p[k]);
    v[6] = fminf(v[6], p[6]);
  }
  auto lo = [](float a, float b) { return fminf(a, b); };
  auto hi = [](float a, float b) { return fmaxf(a, b); };
#pragma unroll
  for (int k = 0; k < 3; ++k) v[k] = wave_reduce(v[k], lo);
#pragma unroll
  for (int k = 3; k < 6; ++k) v[k] = wave_reduce(v[k], hi);
  v[6] = wave_reduce(v[6], lo);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 7; ++k) red[wv][k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < 7) {
    float r = red[0][threadIdx.x];
    for (int w = 1; w < 16; ++w) r = (threadIdx.x >= 3 && threadIdx.x < 6) ? fmaxf(r, red[w][threadIdx.x]) : fminf(r, red[w][threadIdx.x]);
    if (threadIdx.x < 3) meta->lo[threadIdx.x] = r;
    else if (threadIdx.x < 6) meta->hi[threadIdx.x - 3] = r;
    else meta->finite = r > 0.5f ? 1u : 0u;
  }
}

// ------------------------------------------------------------------------------------------------
// a lane's 16 consecutive points ("segment") of one level's dy
// ------------------------------------------------------------------------------------------------
// Both scatter kernels give lane i of a wave the points 1024 s + 16 i + m, m = 0..15, of a stripe s.  In the planar
// layout their dy values are 64 (bf16) or 128 (fp32) contiguous bytes: loaded once per stripe as 16-byte vectors -
// a wave reads 4 or 8 KiB contiguously - instead of one 4-byte load per visit at a 64-byte lane stride (32 cache
// lines per wave-instruction, which made the dense kernel 2.7x slower on fp32 dy than on bf16).
template <int DTYPE>
struct SegDy {
  static constexpr int kVecs = DTYPE == HBR_F32 ? 8 : 4;
  uint4 v[kVecs];
  __device__ __forceinline__ void load(const void* dy, int l, uint32_t N, uint32_t n0) {
    const uint4* p = (const uint4*)((const char*)dy + ((size_t)l * N + n0) * (DTYPE == HBR_F32 ? 8 : 4));
#pragma unroll
    for (int k = 0; k < kVecs; ++k) v[k] = p[k];
  }
  // (d0, d1) of the segment's m-th point; m must be a compile-time constant after unrolling
  __device__ __forceinline__ void get(int m, float& d0, float& d1) const {
    if (DTYPE == HBR_F32) {
      const uint4 q = v[m >> 1];
      d0 = __uint_as_float((m & 1) ? q.z : q.x);
      d1 = __uint_as_float((m & 1) ? q.w : q.y);
    } else {
      const uint4 q = v[m >> 2];
      const uint32_t w = (m & 3) == 0 ? q.x : ((m & 3) == 1 ? q.y : ((m & 3) == 2 ? q.z : q.w));
      d0 = bf16_lo(w);
      d1 = bf16_hi(w);
    }
  }
  // feature f of the segment's m-th point alone.  bf16: ONE v_perm_b32 moves the chosen half of the packed pair into the
  // upper half of a zero word (`sel` = pick_selector(f), wave-uniform) - the hashed visit is VALU-issue bound, and
  // shift + mask + select of the two decoded values were 3 of its ~70 instructions.
  __device__ __forceinline__ static uint32_t pick_selector(int f) { return f ? 0x07060c0cu : 0x05040c0cu; }
  __device__ __forceinline__ float pick(int m, int f, uint32_t sel) const {
    if (DTYPE == HBR_F32) {
      float d0, d1;
      get(m, d0, d1);
      return f ? d1 : d0;
    } else {
      const uint4 q = v[m >> 2];
      const uint32_t w = (m & 3) == 0 ? q.x : ((m & 3) == 1 ? q.y : ((m & 3) == 2 ? q.z : q.w));
      return __uint_as_float(__builtin_amdgcn_perm(w, 0u, sel));
    }
  }
  // vector path usable for this level: planar, and the level's first byte 16-byte aligned
  __device__ __forceinline__ static bool aligned(const void* dy, int l, uint32_t N) {
    return ((((uintptr_t)dy) + (size_t)l * N * (DTYPE == HBR_F32 ? 8 : 4)) & 15) == 0;
  }
};

// ------------------------------------------------------------------------------------------------
// slice-membership masks (tables of kMaskMinSlices or more slices per level)
// ------------------------------------------------------------------------------------------------
// masks[l][slice][stripe][w] bit i = "entry (stripe, w, i) of the coordinate block - the point the scatter kernel's lane i
// visits at step w of that stripe - has at least one of its eight corners in rows [slice * 16384, (slice + 1) * 16384) of
// level l".  One workgroup per stripe, wave w = step w; a lane ORs its bit into its wave's [spl] words in LDS (rows come
// from corner_rows, the same cell arithmetic the visit uses, any T), then the words leave as plain 8-byte stores.  Points
// beyond N (the last stripe's padding) are never flagged; dense levels and all-zero / non-finite levels are skipped (the
// scatter kernel does not sweep them either).
template <bool POW2>
__global__ __launch_bounds__(kLdsBwdThreads) void slice_mask_kernel(uint32_t N, HashGeom g, int spl, int fixbits, int dense_levels,
                                                                     const float* __restrict__ xnorm, const Meta* __restrict__ meta,
                                                                     unsigned long long* __restrict__ masks) {
  __shared__ unsigned long long wm[kLdsBwdThreads / 64][kMaskMaxSlices];
  const uint32_t s = blockIdx.x, stripes = gridDim.x;
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const float* q = xnorm + ((size_t)s * 1024u + threadIdx.x) * 3;
  const float nx = q[0], ny = q[1], nz = q[2];
  const bool live = s * 1024u + lane * kSeg + wv < N;
  const unsigned long long bit = 1ull << lane;
  for (int l = 0; l < g.L; ++l) {
    const LevelPlan plan = level_plan(meta, g, l, fixbits, dense_levels);
    if (plan.dense || plan.fs.state != 1) continue;  // uniform over the workgroup
    for (int i = lane; i < spl; i += 64) wm[wv][i] = 0ull;
    __syncthreads();
    if (live) {
      const Cell c = locate(nx, ny, nz, g.scale[l]);
      uint32_t rows[8];
      corner_rows<POW2>(g, c, rows);
      uint32_t prev = 0xffffffffu;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t sl = rows[k] >> kSliceLog2;
        if (sl != prev) atomicOr(&wm[wv][sl], bit);  // x-neighbours almost always share their slice
        prev = sl;
      }
    }
    __syncthreads();
    for (int i = lane; i < spl; i += 64) masks[(((size_t)l * spl + i) * stripes + s) * kSeg + wv] = wm[wv][i];
  }
}

// ------------------------------------------------------------------------------------------------
// hashed-slice kernel
// ------------------------------------------------------------------------------------------------
// One workgroup = (level, 16384-row slice, feature f, chunk of points).  Splitting the two features of a row over
// two workgroups keeps the slice at 16384 rows of 8-byte accumulators, so a point is visited 8 times per level
// (4 slices x 2 features) and each visit issues at most 8 LDS atomics.
// Flush: `slabs` != nullptr -> the workgroup converts its slice to fp32 and stores it (plain, contiguous) into
// slab [chunk][l][f][row]; slab_reduce_kernel then sums the chunks in a fixed order.  nullptr -> contiguous global
// float atomics straight into dtables (no extra memory, but the fp32 sum of the chunk partials is order-dependent).
template <bool POW2, int LAYOUT, int DTYPE, bool MASKED>
__device__ __forceinline__ void hashed_slice_body(const uint32_t b, unsigned long long* __restrict__ acc /* LDS [kSliceRows] (+ the rings) */,
                                                  uint32_t N, const void* __restrict__ dy, int64_t dy_stride, const HashGeom& g,
                                                  float* __restrict__ dtables, int slices_per_level, int chunks, int fixbits,
                                                  int dense_levels, const float* __restrict__ xnorm,
                                                  const Meta* __restrict__ meta, float* __restrict__ slabs,
                                                  const unsigned long long* __restrict__ masks) {
  // block -> (level, slice, feature, chunk); chunk varies fastest so the blocks of one slice start together and, with
  // a multiple of 8 chunks, chunk c of every (level, slice, feature) lands on XCD c % 8: its coordinates and dy are
  // re-read from that XCD's L2
  const uint32_t chunk = b % chunks;
  const uint32_t lsf = b / chunks;
  const int f = lsf & 1;
  const uint32_t slice = (lsf >> 1) % slices_per_level;
  const int l = (lsf >> 1) / slices_per_level;
  const LevelPlan plan = level_plan(meta, g, l, fixbits, dense_levels);
  if (plan.dense) return;  // the dense kernel owns this level
  const FixScale fs = plan.fs;
  const uint32_t row_lo = slice << kSliceLog2;
  const int64_t rows_here = min((int64_t)kSliceRows, g.T - (int64_t)row_lo);
  // a chunk is a run of whole 1024-point stripes
  const uint32_t stripes = (N + 1023u) / 1024u;
  const uint32_t per = (stripes + chunks - 1) / chunks;
  const uint32_t s_begin = chunk * per, s_end = min(stripes, s_begin + per);
  float* slab = slabs ? slabs + (((size_t)chunk * g.L + l) * 2 + f) * (size_t)g.T + row_lo : nullptr;
  if (s_begin >= s_end || fs.state != 1) {
    // uniform over the workgroup: an empty chunk or an all-zero gradient adds nothing; a non-finite gradient
    // poisons the level (as float accumulation would) instead of being scaled into garbage
    const float fill = fs.state == 2 && s_begin < s_end ? __uint_as_float(0x7fc00000u) : 0.f;
    if (slab) {
      for (int64_t i = threadIdx.x; i < rows_here; i += kLdsBwdThreads) slab[i] = fill;
    } else if (fill != 0.f) {
      float* out = dtables + ((size_t)l * g.T + row_lo) * 2 + f;
      for (int64_t i = threadIdx.x; i < rows_here; i += kLdsBwdThreads) out[2 * i] = fill;
    }
    return;
  }

  for (int i = threadIdx.x; i < kSliceRows; i += kLdsBwdThreads) acc[i] = 0ull;
  __syncthreads();

  const float scale = g.scale[l];
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;

  // one visit: this slice's share of the eight corner contributions of a point; dv = dy of feature f (0 if not live).
  // The loop is VALU-issue bound (71 instructions per visit at the start of round 2).  When the launch's bounding box
  // says no coordinate is negative (NONNEG), u - trunc(u) is one v_fract_f32: for u >= 0 that difference is exactly
  // representable and equals u - floor(u).  (Packed fp32 - v_pk_mul/add_f32 for the coordinate scaling and the
  // weight products - was measured SLOWER, 610 vs 589 us: a packed instruction issues at half rate here.)
  auto visit = [&](auto nonneg_tag, float nx, float ny, float nz, float dv) {
    constexpr bool NONNEG = decltype(nonneg_tag)::value;
    constexpr bool BOXED = NONNEG;  // the launch's bounding box also proves cx < 16383 (see `nonneg` below)
    const float dvs = __fmul_rn(dv, fs.mul);  // exact: a power of two
    Cell c;
    if constexpr (NONNEG) {
      const float ux = __fmul_rn(nx, scale), uy = __fmul_rn(ny, scale), uz = __fmul_rn(nz, scale);
      c.cx = (int)ux; c.cy = (int)uy; c.cz = (int)uz;
      c.fx = __builtin_amdgcn_fractf(ux); c.fy = __builtin_amdgcn_fractf(uy); c.fz = __builtin_amdgcn_fractf(uz);
    } else {
      c = locate(nx, ny, nz, scale);
    }
    // trilinear weight of a corner split as (x*y) * (z*dy): four xy and two z*dy products per visit, each rounded to
    // fp32 as the reference's are; the last product is taken exactly (fma in fp64) and rounded to an integer.  The
    // contribution differs from fl(fl(fl(x*y)*z)*dy) by at most an fp32 ulp.
    const float gx = __fsub_rn(1.0f, c.fx), gy = __fsub_rn(1.0f, c.fy), gz = __fsub_rn(1.0f, c.fz);
    const double xy[4] = {(double)__fmul_rn(gx, gy), (double)__fmul_rn(c.fx, gy), (double)__fmul_rn(gx, c.fy),
                          (double)__fmul_rn(c.fx, c.fy)};
    const double zd[2] = {(double)__fmul_rn(gz, dvs), (double)__fmul_rn(c.fz, dvs)};
    if constexpr (POW2) {
      // Hash components pre-shifted by 3 - (h << 3) distributes over ^ and &, and (c * P) << 3 == c * (P << 3) mod 2^32 -
      // so a corner's masked hash IS its byte offset in the table; with the slice's first byte offset XOR-ed into the
      // y/z terms it is < kSliceBytes exactly when the row is in the slice, and is then the byte offset inside it.
      const uint32_t mask8 = g.mask << 3, lo8 = row_lo << 3;
      const uint32_t xs = (uint32_t)c.cx << 3;
      // (24-bit multiplies - c * P = c * P[23:0] + ((c * P[31:24]) << 24) - instead of v_mul_lo_u32: measured, no difference)
      const uint32_t y0 = (uint32_t)c.cy * (kPrimeY << 3), zz0 = (uint32_t)c.cz * (kPrimeZ << 3);
      const uint32_t y1 = y0 + (kPrimeY << 3), zz1 = zz0 + (kPrimeZ << 3);
      const uint32_t a[4] = {(y0 ^ zz0 ^ lo8) & mask8, (y1 ^ zz0 ^ lo8) & mask8, (y0 ^ zz1 ^ lo8) & mask8,
                             (y1 ^ zz1 ^ lo8) & mask8};
      // The x term of the hash is the cell coordinate itself: while 0 <= cx and cx + 1 < kSliceRows it cannot reach the
      // bits that select the slice (nor exceed the mask of a table with at least that many rows), so the two
      // x-neighbours of a (y, z) pair are in the slice together - four tests per visit instead of eight, and both
      // atomics of a pair run under one exec mask.  Negative cells (points outside the box), coordinates that need
      // N_l * extent > 16384 and tables smaller than a slice take the per-corner form below.
      if ((BOXED || __all((uint32_t)c.cx < (uint32_t)(kSliceRows - 1))) && g.T >= kSliceRows) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {  // pair p: y bit = p & 1, z bit = p >> 1 (corner = x + 2 y + 4 z, hash_encoding.py:34-37)
          if (a[p] < kSliceBytes) {
            atomicAdd((unsigned long long*)((char*)acc + (xs ^ a[p])), fix_product(xy[2 * (p & 1)], zd[p >> 1]));
            atomicAdd((unsigned long long*)((char*)acc + ((xs + 8u) ^ a[p])), fix_product(xy[2 * (p & 1) + 1], zd[p >> 1]));
          }
        }
      } else {
        const uint32_t x0 = xs & mask8, x1 = (xs + 8u) & mask8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const uint32_t off = ((k & 1) ? x1 : x0) ^ a[k >> 1];
          if (off < kSliceBytes) atomicAdd((unsigned long long*)((char*)acc + off), fix_product(xy[k & 3], zd[k >> 2]));
        }
      }
    } else {
      uint32_t rows[8];
      corner_rows<POW2>(g, c, rows);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        uint32_t rel = rows[k] - row_lo;  // wraps to a huge value when the row is below the slice
        if (rel < (uint32_t)kSliceRows) atomicAdd(&acc[rel], fix_product(xy[k & 3], zd[k >> 2]));
      }
    }
  };

  // Each wave takes whole stripes; within one, lane i visits the points 16 i + m in 16 steps: lanes of a step are 16
  // samples apart (consecutive samples of a ray share their cell at the coarse levels, and 64 lanes adding to one LDS
  // address serialise), the coordinates of a step are one contiguous 768-byte read, and the lane's 16 dy values one
  // contiguous 64 / 128 bytes fetched up front.
  const bool vec_ok = LAYOUT == HBR_LAYOUT_PLANAR && SegDy<DTYPE>::aligned(dy, l, N);
  // From the launch's bounding box (uniform over the workgroup): no coordinate negative and every cell's x below the
  // slice-select bits -> the visit needs no per-point range check (NONNEG / BOXED above; 0.594 -> 0.587 ms).
  const bool nonneg = meta->finite && meta->lo[0] >= 0.f && meta->lo[1] >= 0.f && meta->lo[2] >= 0.f && scale >= 0.f &&
                      __fmul_rn(meta->hi[0], scale) < (float)(kSliceRows - 2);
  const uint32_t psel = SegDy<DTYPE>::pick_selector(f);
  auto sweep = [&](auto nonneg_tag) {
    for (uint32_t s = s_begin + wv; s < s_end; s += kLdsBwdThreads / 64) {
      const float* q = xnorm + ((size_t)s * 1024u + lane) * 3;
      const uint32_t n0 = s * 1024u + lane * kSeg;
      if (vec_ok && s * 1024u + 1024u <= N) {
        SegDy<DTYPE> seg;
        seg.load(dy, l, N, n0);
        float nx = q[0], ny = q[1], nz = q[2];
#pragma unroll
        for (int m = 0; m < kSeg; ++m) {
          const float* qn = q + (m + 1 < kSeg ? (m + 1) : m) * 64 * 3;  // next step's coordinates, requested before this step's arithmetic
          const float ax = qn[0], ay = qn[1], az = qn[2];
          visit(nonneg_tag, nx, ny, nz, seg.pick(m, f, psel));
          nx = ax; ny = ay; nz = az;
        }
      } else {  // last (partial) stripe, rows layout, or an unaligned level: one clamped load per visit
        for (int m = 0; m < kSeg; ++m) {
          const uint32_t n = n0 + m;
          const uint2 raw = load_feat_raw<LAYOUT, DTYPE>(dy, min(n, N - 1), l, N, dy_stride);
          float d0, d1;
          decode_feat<DTYPE>(raw, d0, d1);
          visit(nonneg_tag, q[m * 64 * 3], q[m * 64 * 3 + 1], q[m * 64 * 3 + 2], n < N ? (f ? d1 : d0) : 0.f);
        }
      }
    }
  };
  // MASKED: the same stripes, but only the points slice_mask_kernel flagged for this slice are visited.  A wave loads the
  // coordinates of eight steps at a time (coalesced, every lane), pushes the flagged lanes' (x, y, z, dy) into its ring in
  // LDS at tail + (flagged lanes below me) - the step's 64-bit mask is wave-uniform, so the test is two ANDs against the
  // lane's own bit and the rank two v_mbcnt - and visits 64 ring entries whenever that many are waiting.  LDS operations
  // of one wave execute in order, so the ring needs no barrier.  Order of visits differs from the unmasked sweep; the
  // integer sums do not depend on it.
  auto sweep_masked = [&](auto nonneg_tag) {
    float4* ring = (float4*)((char*)acc + kSliceBytes) + wv * kRing;
    uint32_t head = 0, tail = 0;  // wave-uniform
    const uint32_t stripes = (N + 1023u) / 1024u;
    const unsigned long long* mrow = masks + ((size_t)l * slices_per_level + slice) * (size_t)stripes * kSeg;
    const uint32_t bit_lo = lane < 32u ? 1u << lane : 0u, bit_hi = lane >= 32u ? 1u << (lane - 32u) : 0u;
    auto pop = [&](uint32_t count) {  // count <= 64 entries from the head; lanes beyond it visit the origin with dy = 0 (adds 0)
      float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane < count) e = ring[(head + lane) & (kRing - 1)];
      head += count;
      visit(nonneg_tag, e.x, e.y, e.z, e.w);
    };
    // (the wave index as a scalar: the stripe index, and with it the address of the stripe's 16 mask words, is then
    // wave-uniform by construction - the words arrive by scalar loads, no v_readlane per step)
    const uint32_t wv_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)wv);
    for (uint32_t s = s_begin + wv_s; s < s_end; s += kLdsBwdThreads / 64) {
      const unsigned long long* mstripe = mrow + (size_t)s * kSeg;  // step m's mask: mstripe[m]
      const float* q = xnorm + ((size_t)s * 1024u + lane) * 3;
      const uint32_t n0 = s * 1024u + lane * kSeg;
      // one stripe with the dy source fixed at compile time (VEC: the lane's 16 values from one vector load; else a
      // load per flagged point): the choice is per stripe, and inside the 16-step loop it cost six instructions a step
      auto stripe = [&](auto vec_tag) {
        constexpr bool VEC = decltype(vec_tag)::value;
        SegDy<DTYPE> seg;
        if constexpr (VEC) seg.load(dy, l, N, n0);
#pragma unroll
        for (int g8 = 0; g8 < kSeg; g8 += 8) {
          float cx[8], cy[8], cz[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {  // eight steps' coordinates in flight
            const float* qq = q + (g8 + k) * 64 * 3;
            cx[k] = qq[0]; cy[k] = qq[1]; cz[k] = qq[2];
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int m = g8 + k;
            const unsigned long long mask = mstripe[m];
            const uint32_t lo = (uint32_t)mask, hi = (uint32_t)(mask >> 32);
            if ((lo | hi) == 0u) continue;  // uniform
            if (((lo & bit_lo) | (hi & bit_hi)) != 0u) {
              float dv;
              if constexpr (VEC) {
                dv = seg.pick(m, f, psel);
              } else {
                const uint2 raw = load_feat_raw<LAYOUT, DTYPE>(dy, n0 + m, l, N, dy_stride);  // a flagged point is < N
                float d0, d1;
                decode_feat<DTYPE>(raw, d0, d1);
                dv = f ? d1 : d0;
              }
              const uint32_t rank = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
              ring[(tail + rank) & (kRing - 1)] = make_float4(cx[k], cy[k], cz[k], dv);
            }
            tail += (uint32_t)__builtin_popcount(lo) + (uint32_t)__builtin_popcount(hi);
            if (tail - head >= 64u) pop(64u);
          }
        }
      };
      if (vec_ok && s * 1024u + 1024u <= N) stripe(std::true_type{});  // uniform
      else stripe(std::false_type{});
    }
    if (tail != head) pop(tail - head);
  };
  if constexpr (MASKED) {
    if (nonneg) sweep_masked(std::true_type{});
    else sweep_masked(std::false_type{});
  } else {
    if (nonneg) sweep(std::true_type{});
    else sweep(std::false_type{});
  }
  __syncthreads();

  if (slab) {
    for (int64_t i = threadIdx.x; i < rows_here; i += kLdsBwdThreads) slab[i] = (float)((double)(long long)acc[i] * fs.inv);
  } else {
    // contiguous wave-instructions of float atomics (stride 2 floats); skip exact zeros (untouched rows)
    float* out = dtables + ((size_t)l * g.T + row_lo) * 2 + f;
    for (int64_t i = threadIdx.x; i < rows_here; i += kLdsBwdThreads) {
      const float v = (float)((double)(long long)acc[i] * fs.inv);
      if (v != 0.f) unsafeAtomicAdd(out + 2 * i, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// dense kernel (coarse levels)
// ------------------------------------------------------------------------------------------------
// One workgroup = (level, chunk of stripes, feature half).  Lane i of a wave walks the 16 consecutive points
// 1024 s + 16 i + m (m = 0..15) of stripe s - for rays: 16 consecutive samples - and accumulates, in fp32 registers
// and in sample order, the eight corner sums (x both features unless `split`) of the cell it is in; when the cell
// changes, and at the end of the segment, the sums are rounded to fixed point once and added to the dense vertex
// table in LDS.  One visit per point-level instead of eight, and at level 0 one flush per ~10 samples instead of 16
// atomics per sample.  The table leaves the workgroup as integers (dense slab [level][chunk][V*2]).
template <int LAYOUT, int DTYPE>
__device__ __forceinline__ void dense_body(const uint32_t b, unsigned long long* __restrict__ acc /* LDS [V][2] (both features) or [V] (split) */,
                                           uint32_t N, const void* __restrict__ dy, int64_t dy_stride, const HashGeom& g, int dchunks,
                                           int fixbits, int dense_levels, const float* __restrict__ xnorm,
                                           const Meta* __restrict__ meta, unsigned long long* __restrict__ dslab) {
  // heaviest first: the highest dense level has the largest table (and, split, two workgroups per chunk)
  const int f = b & 1;
  const uint32_t chunk = (b >> 1) % dchunks;
  const int l = dense_levels - 1 - (int)((b >> 1) / dchunks);
  const LevelPlan plan = level_plan(meta, g, l, fixbits, dense_levels);
  if (!plan.dense || (!plan.split && f == 1)) return;
  const bool both = !plan.split;
  const int entries = both ? 2 * plan.V : plan.V;
  for (int i = threadIdx.x; i < entries; i += kLdsBwdThreads) acc[i] = 0ull;
  __syncthreads();

  const uint32_t stripes = (N + 1023u) / 1024u;
  const uint32_t per = (stripes + dchunks - 1) / dchunks;
  const uint32_t s_begin = chunk * per, s_end = min(stripes, s_begin + per);
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const float scale = g.scale[l], mul = plan.fs.mul;
  const int sx = both ? 2 : 1;                 // LDS entries per vertex
  const int sy = plan.dx * sx, sz = plan.dx * plan.dy * sx;

  // BOTH: two features per vertex (entries v*2, v*2+1); else this workgroup's feature f only (entry v).
  // VEC: the lane's 16 dy values come from one contiguous vector load (full, aligned planar stripes).
  const bool vec_ok = LAYOUT == HBR_LAYOUT_PLANAR && SegDy<DTYPE>::aligned(dy, l, N);
  auto walk = [&](auto both_tag) {
    constexpr bool BOTH = decltype(both_tag)::value;
    for (uint32_t s = s_begin + wv; s < s_end; s += kLdsBwdThreads / 64) {
      float s0[8], s1[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) s0[k] = s1[k] = 0.f;
      int cur = 0;
      const float* q = xnorm + ((size_t)s * 1024u + lane) * 3;
      const uint32_t n0 = s * 1024u + lane * kSeg;
      // one step: point's coordinates + its (scaled) dy -> leave the old cell if it changed, then accumulate
      auto step = [&](int m, float nx, float ny, float nz, float d0, float d1) {
        if (!BOTH) d0 = f ? d1 : d0;
        d0 = __fmul_rn(d0, mul);  // exact: a power of two
        d1 = __fmul_rn(d1, mul);
        const Cell c = locate(nx, ny, nz, scale);
        const int e0 = (c.cx - plan.x0) * sx + (c.cy - plan.y0) * sy + (c.cz - plan.z0) * sz;
        if (m > 0 && __any(e0 != cur)) {  // some lane leaves its cell: those lanes add their sums to the table and start over
          if (e0 != cur) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const int e = cur + (k & 1) * sx + ((k >> 1) & 1) * sy + (k >> 2) * sz;
              atomicAdd(&acc[e], fix_value(s0[k]));
              if (BOTH) atomicAdd(&acc[e + 1], fix_value(s1[k]));
              s0[k] = s1[k] = 0.f;
            }
          }
        }
        cur = e0;
        float w[8];
        corner_weights(c, w);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          s0[k] = __fmaf_rn(w[k], d0, s0[k]);
          if (BOTH) s1[k] = __fmaf_rn(w[k], d1, s1[k]);
        }
      };
      if (vec_ok && s * 1024u + 1024u <= N) {
        SegDy<DTYPE> seg;
        seg.load(dy, l, N, n0);
#pragma unroll
        for (int m = 0; m < kSeg; ++m) {
          float d0, d1;
          seg.get(m, d0, d1);
          step(m, q[m * 64 * 3], q[m * 64 * 3 + 1], q[m * 64 * 3 + 2], d0, d1);
        }
      } else {
        for (int m = 0; m < kSeg; ++m) {
          const uint32_t n = n0 + m;
          const uint2 raw = load_feat_raw<LAYOUT, DTYPE>(dy, min(n, N - 1), l, N, dy_stride);
          float d0, d1;
          decode_feat<DTYPE>(raw, d0, d1);
          const bool live = n < N;
          step(m, q[m * 64 * 3], q[m * 64 * 3 + 1], q[m * 64 * 3 + 2], live ? d0 : 0.f, live ? d1 : 0.f);
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int e = cur + (k & 1) * sx + ((k >> 1) & 1) * sy + (k >> 2) * sz;
        atomicAdd(&acc[e], fix_value(s0[k]));
        if (BOTH) atomicAdd(&acc[e + 1], fix_value(s1[k]));
      }
    }
  };
  if (both) walk(std::true_type{});
  else walk(std::false_type{});
  __syncthreads();
  // dense slab [l][chunk][v][f]
  unsigned long long* out = dslab + ((size_t)l * dchunks + chunk) * (size_t)(4 * kDenseCap);
  if (both) {
    for (int i = threadIdx.x; i < entries; i += kLdsBwdThreads) out[i] = acc[i];
  } else {
    for (int i = threadIdx.x; i < entries; i += kLdsBwdThreads) out[2 * i + f] = acc[i];
  }
}

// ONE launch for both kinds of workgroup - the dense ones first - so that the dispatcher packs them together: the
// dense levels alone are a few hundred workgroups of uneven length (2.4 rounds of the 256 CUs; measured 104 us for
// 49 us worth of instructions when launched on their own).
// (waves_per_eu 4: the LDS allows one 16-wave workgroup per CU anyway, so take the 128 VGPRs that leaves - at the
// default heuristic the dense walk's sixteen running sums spilled to scratch.)
template <bool POW2, int LAYOUT, int DTYPE, bool MASKED>
__global__ __launch_bounds__(kLdsBwdThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void hash_scatter_kernel(
    uint32_t N, const void* __restrict__ dy, int64_t dy_stride, HashGeom g, float* __restrict__ dtables, int slices_per_level,
    int chunks, int dchunks, uint32_t dense_blocks, int fixbits, int dense_levels, const float* __restrict__ xnorm,
    const Meta* __restrict__ meta, float* __restrict__ slabs, unsigned long long* __restrict__ dslab,
    const unsigned long long* __restrict__ masks) {
  extern __shared__ unsigned long long acc[];
  if (blockIdx.x < dense_blocks)
    dense_body<LAYOUT, DTYPE>(blockIdx.x, acc, N, dy, dy_stride, g, dchunks, fixbits, dense_levels, xnorm, meta, dslab);
  else
    hashed_slice_body<POW2, LAYOUT, DTYPE, MASKED>(blockIdx.x - dense_blocks, acc, N, dy, dy_stride, g, dtables, slices_per_level,
                                                   chunks, fixbits, dense_levels, xnorm, meta, slabs, masks);
}

// dense vertex tables -> rows: one thread per (vertex, feature) of a dense level sums the chunks' integers and adds
// the total to the level's int64 row table g64[l][row][f] with an integer atomic (several vertices can hash to one
// row; integer addition keeps the result independent of their order).  g64 must be zero on entry.
template <bool POW2>
__global__ __launch_bounds__(256) void dense_scatter_kernel(HashGeom g, int dchunks, int fixbits, int dense_levels, const Meta* __restrict__ meta,
                                                            const unsigned long long* __restrict__ dslab,
                                                            unsigned long long* __restrict__ g64) {
  __shared__ unsigned long long part[8][32];
  const int l = blockIdx.y;
  const LevelPlan plan = level_plan(meta, g, l, fixbits, dense_levels);
  if (!plan.dense || (int)blockIdx.x * 32 >= 2 * plan.V) return;  // uniform over the block
  // 32 consecutive entries x 8 parts: part p sums chunks p, p + 8, ... (integers: any order gives the same sum)
  const int el = threadIdx.x & 31, pt = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;
  unsigned long long sum = 0;
  if (e < 2 * plan.V) {
    const unsigned long long* p = dslab + (size_t)l * dchunks * (size_t)(4 * kDenseCap) + e;
    unsigned long long a0 = 0, a1 = 0;
    int c = pt;
    for (; c + 8 < dchunks; c += 16) { a0 += p[(size_t)c * (4 * kDenseCap)]; a1 += p[(size_t)(c + 8) * (4 * kDenseCap)]; }
    if (c < dchunks) a0 += p[(size_t)c * (4 * kDenseCap)];
    sum = a0 + a1;
  }
  part[pt][el] = sum;
  __syncthreads();
  if (pt != 0 || e >= 2 * plan.V) return;
  sum = part[0][el] + part[1][el] + part[2][el] + part[3][el] + part[4][el] + part[5][el] + part[6][el] + part[7][el];
  if (sum == 0) return;
  const int v = e >> 1, fe = e & 1;
  Cell c;
  c.cx = plan.x0 + v % plan.dx;
  c.cy = plan.y0 + (v / plan.dx) % plan.dy;
  c.cz = plan.z0 + v / (plan.dx * plan.dy);
  c.fx = c.fy = c.fz = 0.f;
  uint32_t rows[8];
  corner_rows<POW2>(g, c, rows);  // rows[0]: the vertex itself
  atomicAdd(&g64[((size_t)l * g.T + rows[0]) * 2 + fe], sum);
}

// dtables[l][row][f] += the level's total: hashed levels - sum over chunks (in chunk order) of slab[chunk][l][f][row];
// dense levels - the int64 row table.  One thread per row of a level: every table entry has exactly one writer and
// one summation order.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int chunks, HashGeom g, int fixbits,
                                                          int dense_levels, const Meta* __restrict__ meta, const unsigned long long* __restrict__ g64,
                                                          float* __restrict__ dtables, bool overwrite) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (l, row)
  const int L = g.L;
  const int64_t T = g.T;
  if (i >= (int64_t)L * T) return;
  const int64_t l = i / T, row = i - l * T;
  const LevelPlan plan = level_plan(meta, g, (int)l, fixbits, dense_levels);
  float s0 = 0.f, s1 = 0.f;
  if (plan.dense) {
    const unsigned long long* p = g64 + (size_t)i * 2;
    s0 = (float)((double)(long long)p[0] * plan.fs.inv);
    s1 = (float)((double)(long long)p[1] * plan.fs.inv);
  } else {
    const size_t chunk_stride = (size_t)L * 2 * T;
    const float* p0 = slabs + (size_t)l * 2 * T + row;
    int c = 0;
    for (; c + 4 <= chunks; c += 4) {  // four chunks in flight; added in chunk order
      const float a0 = p0[(size_t)c * chunk_stride], b0 = p0[(size_t)c * chunk_stride + T];
      const float a1 = p0[(size_t)(c + 1) * chunk_stride], b1 = p0[(size_t)(c + 1) * chunk_stride + T];
      const float a2 = p0[(size_t)(c + 2) * chunk_stride], b2 = p0[(size_t)(c + 2) * chunk_stride + T];
      const float a3 = p0[(size_t)(c + 3) * chunk_stride], b3 = p0[(size_t)(c + 3) * chunk_stride + T];
      s0 = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(s0, a0), a1), a2), a3);
      s1 = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(s1, b0), b1), b2), b3);
    }
    for (; c < chunks; ++c) {
      s0 = __fadd_rn(s0, p0[(size_t)c * chunk_stride]);
      s1 = __fadd_rn(s1, p0[(size_t)c * chunk_stride + T]);
    }
  }
  float2* d = (float2*)dtables + i;  // this thread is the row's only writer
  float2 v = overwrite ? make_float2(0.f, 0.f) : *d;
  v.x = __fadd_rn(v.x, s0); v.y = __fadd_rn(v.y, s1);
  *d = v;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// Chunks of ~256 Ki points per workgroup of the hashed-slice kernel: a workgroup's 128 KiB zero + flush and its slab
// must be amortised, yet the grid has to fill 256 CUs a few times over.  Measured at N = 2M, L = 16 with the slab flush
// (total K2 time): 4 chunks 0.717 ms, 6 -> 0.731, 8 -> 0.598, 10 -> 0.631, 12 -> 0.657, 16 -> 0.683 (with the atomic
// flush of round 1 the optimum was 16).  The count depends on N and T only, NOT on L, so a launch over a sub-range of
// the levels (the staged multi-GPU all-reduce) chunks - and therefore rounds - exactly like the full one.  Small
// problems still get >= 512 workgroups at 16 levels; at most one chunk per 1024-point stripe; never 2^19 points or
// more per workgroup.
static int lds_chunks(int64_t N, int spl) {
  constexpr int64_t kChunkPoints = 256 * 1024;  // 8 chunks at N = 2 M: 0.578 ms; 6: 0.712, 12: 0.618, 16: 0.622 (tools/k2_time.py)
  constexpr int kMinBlocks = 512, kRefLevels = 16;
  int chunks = (int)((N + kChunkPoints / 2) / kChunkPoints);
  if (chunks > 8) chunks = (chunks + 4) / 8 * 8;  // a multiple of 8: chunk c of every (level, slice, feature) on XCD c % 8
  const int min_chunks = (kMinBlocks + kRefLevels * spl * 2 - 1) / (kRefLevels * spl * 2);
  if (chunks < min_chunks) chunks = min_chunks;
  // (Round 4, measured and NOT kept: 5 chunks below 384 Ki points / 8 from there, to turn the 1.5 rounds of 384 workgroups
  // into full ones - K2 at 2000 / 4000 / 8000 rays 0.126 / 0.186 / 0.313 ms against 0.112 / 0.183 / 0.321: a wave takes whole
  // stripes, so 50 stripes over 16 waves cost the same four stripe-times as 62, and the dense workgroups are as long as
  // the hashed ones.)
  const int max_chunks = (int)((N + kLdsBwdThreads - 1) / kLdsBwdThreads);
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  while ((N + chunks - 1) / chunks + kLdsBwdThreads > (1LL << kChunkPointsLog2Max)) ++chunks;
  return chunks;
}
static int lds_slices(int64_t T) { return (int)((T + kSliceRows - 1) / kSliceRows); }
// the masked form of the hashed branch (slice_mask_kernel + ring visits) pays from sixteen slices per level
static bool lds_masked(int spl) { return spl >= kMaskMinSlices && spl <= kMaskMaxSlices; }
// Chunks of the masked form: every (level, slice, feature) owner scans the masks of ALL its chunk's stripes whatever it
// visits, and each chunk costs a 64 KiB slab per owner - few chunks.  Measured (HBR_K2_MASK_CHUNKS, gpurun_out/k2_mask_chunks.txt;
// K2 span in ms at 2^18 / 2^19 / 2^20): 1 chunk 1.51 / 1.89 / 3.27, 2 chunks 1.21 / 1.91 / 3.14, 3 chunks 1.32 / 2.25 / 5.05 (an odd
// count puts the chunks of one slice on different XCDs from call to call), 4 chunks 1.24 / 1.86 / 3.24, 8 chunks 1.23 / 1.95 / 3.38:
// two.  Depends on N and T only, like lds_chunks.  (No limit on the points per workgroup is needed: the fixed-point
// scale is derived from N, so no sum over any subset of a launch's contributions can overflow.)
static int lds_chunks_masked(int64_t N, int spl) {
  int chunks = 2;
  (void)spl;
  static const char* force = getenv("HBR_K2_MASK_CHUNKS");  // tuning runs only (tools/k2_vs_T.sh)
  if (force && atoi(force) > 0) chunks = atoi(force);
  const int max_chunks = (int)((N + kLdsBwdThreads - 1) / kLdsBwdThreads);
  if (chunks > max_chunks) chunks = max_chunks;
  return chunks < 1 ? 1 : chunks;
}
static int hashed_chunks(int64_t N, int spl) { return lds_masked(spl) ? lds_chunks_masked(N, spl) : lds_chunks(N, spl); }
static int dense_chunks(int64_t N) {
  const int64_t stripes = (N + 1023) / 1024;
  int64_t c = (stripes + kDenseStripesPerWg - 1) / kDenseStripesPerWg;
  return (int)(c < 1 ? 1 : c);
}
static int fix_bits_for(int64_t N) {
  int lg = 0;
  while ((1LL << lg) < N) ++lg;
  return 62 - (lg > kChunkPointsLog2Max ? lg : kChunkPointsLog2Max);
}

// workspace: [normalised coordinates, 3 floats per point, whole stripes][Meta][partial maxima][partial boxes]
//            | min ends here |  [int64 row tables of the dense levels][dense slabs][chunk slabs]
struct Workspace {
  int64_t xnorm, meta, abs_part, bounds_part, min_total, g64, dslab, slabs, masks, total;
  int dense_levels;  // levels the dense path can take at all (0: disabled for this T)
};
static Workspace workspace(int64_t N, int L, int64_t T) {
  auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
  Workspace w;
  const int64_t stripes = (N + 1023) / 1024;
  w.xnorm = 0;
  w.meta = up(stripes * 1024 * 3 * 4);
  w.abs_part = w.meta + (int64_t)sizeof(Meta);
  w.bounds_part = up(w.abs_part + (int64_t)HBR_MAX_LEVELS * kAbsBlocks * 4);
  w.min_total = up(w.bounds_part + stripes * 7 * 4);
  w.dense_levels = T <= kDenseMaxT ? (L < kDenseLevels ? L : kDenseLevels) : 0;
  w.g64 = w.min_total;
  w.dslab = up(w.g64 + (int64_t)w.dense_levels * T * 2 * 8);
  w.slabs = up(w.dslab + (int64_t)w.dense_levels * dense_chunks(N) * 4 * kDenseCap * 8);
  const int spl = lds_slices(T);
  w.masks = up(w.slabs + (int64_t)hashed_chunks(N, spl) * L * 2 * T * 4);
  w.total = w.masks + (lds_masked(spl) ? (int64_t)L * spl * stripes * kSeg * 8 : 0);  // slice-membership masks [L][spl][stripes][16]
  return w;
}

template <int LAYOUT, int DTYPE>
static void launch_absmax(hipStream_t st, const void* dy, uint32_t N, int64_t stride, int L, uint32_t* part, int& blocks) {
  if (LAYOUT == HBR_LAYOUT_PLANAR) {
    const size_t nvec = (size_t)N * 2 * (DTYPE == HBR_F32 ? 4 : 2) / 16;
    uint32_t bx = (uint32_t)((nvec + 256 * 4 - 1) / (256 * 4));
    bx = bx < 1 ? 1 : (bx > (uint32_t)kAbsBlocks ? (uint32_t)kAbsBlocks : bx);
    blocks = (int)bx;
    hipLaunchKernelGGL((absmax_planar_kernel<DTYPE>), dim3(bx, (uint32_t)L), dim3(256), 0, st, dy, N, part);
    return;
  }
  uint32_t bx = (N + 256 * 8 - 1) / (256 * 8);
  bx = bx < 1 ? 1 : (bx > (uint32_t)kAbsBlocks ? (uint32_t)kAbsBlocks : bx);
  blocks = (int)bx;
  hipLaunchKernelGGL((absmax_kernel<LAYOUT, DTYPE>), dim3(bx, (uint32_t)L), dim3(256), 0, st, dy, N, stride, part);
}

template <bool POW2, int LAYOUT, int DTYPE>
static int launch_lds(hipStream_t st, uint32_t N, const void* dy, int64_t stride, const HashGeom& g, float* dtables, char* ws,
                      const Workspace& w, bool full, bool overwrite, bool g64_cleared) {
  const int spl = lds_slices(g.T);
  const bool masked = full && lds_masked(spl);  // (the masks live behind the slabs: full workspace only)
  const int chunks = masked ? lds_chunks_masked(N, spl) : lds_chunks(N, spl);
  const int fixbits = fix_bits_for(N);
  const float* xnorm = (const float*)(ws + w.xnorm);
  const Meta* meta = (const Meta*)(ws + w.meta);
  float* slabs = full ? (float*)(ws + w.slabs) : nullptr;
  const int dl = full ? w.dense_levels : 0;  // levels the dense kernel may take (it decides per level on the device)
  unsigned long long* g64 = (unsigned long long*)(ws + w.g64);
  unsigned long long* dslab = (unsigned long long*)(ws + w.dslab);
  const int dchunks = dense_chunks(N);
  const uint32_t dense_blocks = (uint32_t)(dl * dchunks * 2);
  if (dl > 0 && !g64_cleared && hipMemsetAsync(g64, 0, (size_t)dl * g.T * 16, st) != hipSuccess) return HBR_ELAUNCH;
  static_assert(2 * kDenseCap * 8 >= kSliceRows * 8 && kMaskedLdsBytes >= 2 * kDenseCap * 8, "LDS users: slice < dense table < slice + rings");
  const uint32_t grid = dense_blocks + (uint32_t)(g.L * spl * 2 * chunks);
  if (masked) {
    unsigned long long* masks = (unsigned long long*)(ws + w.masks);
    const uint32_t stripes = (N + 1023u) / 1024u;
    hipLaunchKernelGGL((slice_mask_kernel<POW2>), dim3(stripes), dim3(kLdsBwdThreads), 0, st, N, g, spl, fixbits, dl, xnorm, meta, masks);
    auto kern = hash_scatter_kernel<POW2, LAYOUT, DTYPE, true>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaskedLdsBytes) != hipSuccess) return HBR_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kLdsBwdThreads), kMaskedLdsBytes, st, N, dy, stride, g, dtables, spl, chunks, dchunks,
                       dense_blocks, fixbits, dl, xnorm, meta, slabs, dslab, (const unsigned long long*)masks);
  } else {
    auto kern = hash_scatter_kernel<POW2, LAYOUT, DTYPE, false>;
    const int lds = dl > 0 ? 2 * kDenseCap * 8 : kSliceRows * 8;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kDenseCap * 8) != hipSuccess) return HBR_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kLdsBwdThreads), lds, st, N, dy, stride, g, dtables, spl, chunks, dchunks, dense_blocks,
                       fixbits, dl, xnorm, meta, slabs, dslab, (const unsigned long long*)nullptr);
  }
  if (dl > 0)
    hipLaunchKernelGGL((dense_scatter_kernel<POW2>), dim3((4 * kDenseCap + 31) / 32, (uint32_t)dl), dim3(256), 0, st, g, dchunks,
                       fixbits, dl, meta, (const unsigned long long*)dslab, g64);
  if (full) {
    const int64_t rows = (int64_t)g.L * g.T;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((uint32_t)((rows + 255) / 256)), dim3(256), 0, st, (const float*)slabs, chunks, g,
                       fixbits, dl, meta, (const unsigned long long*)(ws + w.g64), dtables, overwrite);
  }
  return HBR_OK;
}

static bool lds_shape_ok(int64_t N, int L, int64_t T) { return N >= 1 && L >= 1 && T >= 1 && T <= kMaxLdsT; }

// algo 3's contract, made checkable: what the last algo-2 call that FILLED a workspace's coordinate block normalised
// (point source, shape, mu / sigma) and on which stream.  The record lives on the host, keyed by the workspace pointer
// (a handful of workspaces at most: one per stream in a training process); algo 3 is refused unless it names the same
// points on the same stream - so a second caller that ran hbr_hash_encode_bwd on the same workspace between the two
// halves is detected instead of silently scattering the first caller's gradients to the second caller's cells.
struct CoordToken {
  const void* ws;
  const void *x, *o, *d, *t;
  int64_t R, S;
  float mu[3], sigma;
  void* stream;
};
static std::mutex g_token_mutex;
static CoordToken g_tokens[16];
static int g_token_next = 0;
static bool token_equal(const CoordToken& a, const CoordToken& b) {
  return a.ws == b.ws && a.x == b.x && a.o == b.o && a.d == b.d && a.t == b.t && a.R == b.R && a.S == b.S &&
         a.mu[0] == b.mu[0] && a.mu[1] == b.mu[1] && a.mu[2] == b.mu[2] && a.sigma == b.sigma && a.stream == b.stream;
}
static void token_record(const CoordToken& tk) {
  std::lock_guard<std::mutex> lock(g_token_mutex);
  for (auto& e : g_tokens)
    if (e.ws == tk.ws) { e = tk; return; }
  g_tokens[g_token_next] = tk;
  g_token_next = (g_token_next + 1) % 16;
}
static bool token_matches(const CoordToken& tk) {
  std::lock_guard<std::mutex> lock(g_token_mutex);
  for (auto& e : g_tokens)
    if (e.ws == tk.ws) return token_equal(e, tk);
  return false;
}

}  // namespace hbr

using namespace hbr;

extern "C" int64_t hbr_hash_bwd_workspace_bytes(int64_t N, int L, int64_t T, int, int algo) {
  if (algo == 1 || (algo == 0 && N < (int64_t)kLdsAutoMinPoints) || !lds_shape_ok(N, L, T)) return 0;
  return workspace(N, L, T).total;  // (the coordinate part depends on N only, so it is shared by calls over level sub-ranges)
}
extern "C" int64_t hbr_hash_bwd_workspace_bytes_min(int64_t N, int L, int64_t T, int, int algo) {
  if (algo == 1 || (algo == 0 && N < (int64_t)kLdsAutoMinPoints) || !lds_shape_ok(N, L, T)) return 0;
  return workspace(N, L, T).min_total;
}

extern "C" int hbr_hash_encode_bwd(const float* x, const float* rays_o, const float* rays_d, const float* t, int64_t R,
                                   int64_t S, const void* dy, int layout, int64_t dy_stride, int dy_dtype,
                                   const float* dy_absmax, const float* scales_host, const float* mu_host, float sigma, int L,
                                   int64_t T, int F, float* dtables, int algo, void* ws, int64_t ws_bytes, void* stream) {
  if (!dy || !dtables) return HBR_EINVAL;
  if (F != 2) return HBR_EUNSUPPORTED;
  if (layout != HBR_LAYOUT_ROWS && layout != HBR_LAYOUT_PLANAR) return HBR_EINVAL;
  if (dy_dtype != HBR_F32 && dy_dtype != HBR_BF16) return HBR_EINVAL;
  if (layout == HBR_LAYOUT_ROWS && dy_stride < (int64_t)L * F) return HBR_EINVAL;
  const bool overwrite = (algo & HBR_OVERWRITE) != 0;  // write dtables instead of accumulating (hbr_hip.h)
  algo &= ~HBR_OVERWRITE;
  if (algo < 0 || algo > 3) return HBR_EINVAL;
  const bool reuse_coords = algo == 3;  // algo 3 = algo 2, the coordinates (and their boxes) of the previous call on `ws` still valid
  if (reuse_coords) algo = 2;
  HashGeom g;
  int rc = fill_geom(g, scales_host, mu_host, sigma, L, T);
  if (rc) return rc;
  PointSrc ps;
  uint32_t N;
  rc = check_points(x, rays_o, rays_d, t, R, S, ps, N);
  if (rc) return rc;
  if (N == 0) return overwrite ? HBR_EUNSUPPORTED : HBR_OK;  // nothing would write the rows
  hipStream_t st = (hipStream_t)stream;
  // The LDS kernels shift row offsets left by 3 in 32 bits and need their workspace.  Auto picks them once there are
  // enough points to amortise the fixed 128 KiB flush per workgroup and the workspace is there; asked for explicitly
  // they refuse rather than silently running something else.
  const bool shape_ok = lds_shape_ok(N, L, T);
  const Workspace w = shape_ok ? workspace(N, L, T) : Workspace{};
  const bool ws_ok = ws && (((uintptr_t)ws) & 15) == 0 && shape_ok && ws_bytes >= w.min_total;
  if (algo == 2) {
    if (!shape_ok) return HBR_EUNSUPPORTED;
    if (!ws_ok) return HBR_EWORKSPACE;
  }
  if (algo == 0) algo = (N >= kLdsAutoMinPoints && ws_ok) ? 2 : 1;
  if (overwrite && (algo == 1 || ws_bytes < w.total)) return HBR_EUNSUPPORTED;  // float atomics add to what is there
  if (algo == 1) {
    rc = launch_hash_bwd_atomic(st, ps, N, dy, layout, dy_stride, dy_dtype, g, dtables);
    if (rc) return rc;
    HBR_RETURN_IF_LAUNCH_FAILED();
    return HBR_OK;
  }
  char* wsb = (char*)ws;
  const CoordToken token{ws, x, rays_o, rays_d, t, R, S, {g.mu[0], g.mu[1], g.mu[2]}, g.sigma, stream};
  if (reuse_coords) {
    if (!token_matches(token)) return HBR_EINVAL;  // not the points the last algo-2 call left in this workspace
  } else {
    token_record(token);
  }
  const bool full = ws_bytes >= w.total;  // else: hashed slices for every level, float-atomic flush
  const uint32_t stripes = (N + 1023u) / 1024u;
  const bool clears = !reuse_coords && full && w.dense_levels > 0;  // launch_lds' dl: dense levels only with the full workspace
  if (!reuse_coords)  // coordinates in scatter order + one bounding box per stripe (+ the dense levels' row tables cleared)
    hipLaunchKernelGGL(normalise_kernel, dim3(stripes), dim3(1024), 0, st, ps, N, g, (float*)(wsb + w.xnorm), (float*)(wsb + w.bounds_part),
                       (uint4*)(wsb + w.g64), clears ? (size_t)w.dense_levels * (size_t)g.T : (size_t)0);
  uint32_t* abs_part = (uint32_t*)(wsb + w.abs_part);
  int abs_blocks = 0;
  if (!dy_absmax) {
    if (layout == HBR_LAYOUT_PLANAR) {
      if (dy_dtype == HBR_F32) launch_absmax<HBR_LAYOUT_PLANAR, HBR_F32>(st, dy, N, dy_stride, L, abs_part, abs_blocks);
      else launch_absmax<HBR_LAYOUT_PLANAR, HBR_BF16>(st, dy, N, dy_stride, L, abs_part, abs_blocks);
    } else {
      if (dy_dtype == HBR_F32) launch_absmax<HBR_LAYOUT_ROWS, HBR_F32>(st, dy, N, dy_stride, L, abs_part, abs_blocks);
      else launch_absmax<HBR_LAYOUT_ROWS, HBR_BF16>(st, dy, N, dy_stride, L, abs_part, abs_blocks);
    }
  }
  hipLaunchKernelGGL(meta_reduce_kernel, dim3(1), dim3(1024), 0, st, (const uint32_t*)abs_part, abs_blocks, dy_absmax,
                     (const float*)(wsb + w.bounds_part), stripes, L, (Meta*)(wsb + w.meta));
#define HBR_BWD(P, LY, DT) rc = launch_lds<P, LY, DT>(st, N, dy, dy_stride, g, dtables, wsb, w, full, overwrite, clears)
  if (g.pow2) {
    if (layout == HBR_LAYOUT_PLANAR) { if (dy_dtype == HBR_F32) HBR_BWD(true, HBR_LAYOUT_PLANAR, HBR_F32); else HBR_BWD(true, HBR_LAYOUT_PLANAR, HBR_BF16); }
    else { if (dy_dtype == HBR_F32) HBR_BWD(true, HBR_LAYOUT_ROWS, HBR_F32); else HBR_BWD(true, HBR_LAYOUT_ROWS, HBR_BF16); }
  } else {
    if (layout == HBR_LAYOUT_PLANAR) { if (dy_dtype == HBR_F32) HBR_BWD(false, HBR_LAYOUT_PLANAR, HBR_F32); else HBR_BWD(false, HBR_LAYOUT_PLANAR, HBR_BF16); }
    else { if (dy_dtype == HBR_F32) HBR_BWD(false, HBR_LAYOUT_ROWS, HBR_F32); else HBR_BWD(false, HBR_LAYOUT_ROWS, HBR_BF16); }
  }
#undef HBR_BWD
  if (rc) return rc;
  HBR_RETURN_IF_LAUNCH_FAILED();
  return HBR_OK;
}
