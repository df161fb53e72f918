// K3 / K4: fused density + colour MLP of the hash-NeRF (reference test_hash.py:20-72, the instance
// built at train_hash2.py:127: 32->64->64->16, [15 geo + 24 dir-PE]->64->64->3), forward and backward,
// on the gfx950 matrix cores.
//
// Layout idea ("orientation 1"): every GEMM is computed transposed, Z^T[out][point] = W[out][in] * X^T[in][point],
// with v_mfma_f32_32x32x16_bf16 (or the exact-f32 v_mfma_f32_32x32x2_f32).  The 32x32 result keeps the POINT on
// the lane and 16 output features in the lane's registers, which is exactly the B-operand shape of the next
// layer's MFMA (cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's operand"), so the whole
// 6-layer chain - and the data-gradient chain back - runs in registers with no LDS traffic for activations.
// One wave owns 32 points at a time.
//
// Weight gradients sum over points, i.e. need the point on the K axis (dW^T = X^T_tile * dZ_tile).  The four waves of
// a workgroup each run the chain for their own 32-point tile and SHARE the 18 dW tiles: a wave parks a layer's X and dZ
// fragments in an LDS exchange slot, and the tile's owner accumulates over all four slots.  bf16 writes the slot as a
// [point][feature] image and reads it back with the transposing LDS read (ds_read_b64_tr_b16); exact-fp32 transposes
// with one more MFMA against an identity operand first (every product is x*1 or x*0).  The owners' accumulators stay in
// AGPRs for the whole sweep and leave the kernel as per-workgroup slabs that a small reduce kernel sums in a fixed
// order - no LDS or global float atomics in the loop (ds_add_f32 costs ~190 cycles per wave-instruction on gfx950).
//
// Weights are re-packed into MFMA-fragment order by a tiny kernel on every call (parameters are updated in
// place by the optimiser between calls; nothing is cached across calls - except that a step's backward may declare
// the image its forward packed still valid, HBR_IMAGE_READY), then held in LDS.
// One wave per SIMD in the backward kernel: everything below that looks like scheduling by hand (requests ahead of a
// fence, MFMAs pinned between epilogue words, vectors built whole) is there because of it - DESIGN.md section 3.
#include "hbr_common.h"
#include "sample_common.h"
#include "composite_ray.h"
#include <cstdlib>
#include <type_traits>

namespace hbr {
namespace mlp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// flat parameter block (hbr_hip.h): offsets of the six nn.Linear layers
constexpr int OFF_S0W = 0, OFF_S0B = 2048, OFF_S2W = 2112, OFF_S2B = 6208, OFF_S4W = 6272, OFF_S4B = 7296;
constexpr int OFF_C0W = 7312, OFF_C0B = 9808, OFF_C2W = 9872, OFF_C2B = 13968, OFF_C4W = 14032, OFF_C4B = 14224;
static_assert(OFF_C4B + 3 == HBR_MLP_PARAM_FLOATS, "parameter block size");

// layer ids
enum { L1 = 0, L2 = 1, L3 = 2, C1 = 3, C2 = 4, C3 = 5, NLAYER = 6 };

// Logical, zero-padded weight of layer `l`: row = output feature, col = input slot.
//  L3: 16 real rows (row 0 density, rows 1..15 geo features).
//  C1: input slots are the L3 tile's rows: slot 0 = density (weight 0), slots 1..15 = geo (reference cat order,
//      test_hash.py:66), slots 16..39 = the 24 dir-PE values, so col0.weight column = slot-1.
//  C3: 3 real rows (r,g,b).
__device__ __host__ inline int wlog_offset(int l, int row, int col) {
  switch (l) {
    case L1: return (row < 64 && col < 32) ? OFF_S0W + row * 32 + col : -1;
    case L2: return (row < 64 && col < 64) ? OFF_S2W + row * 64 + col : -1;
    case L3: return (row < 16 && col < 64) ? OFF_S4W + row * 64 + col : -1;
    case C1: return (row < 64 && col >= 1 && col < 40) ? OFF_C0W + row * 39 + (col - 1) : -1;
    case C2: return (row < 64 && col < 64) ? OFF_C2W + row * 64 + col : -1;
    case C3: return (row < 3 && col < 64) ? OFF_C4W + row * 64 + col : -1;
  }
  return -1;
}
__device__ __host__ inline int blog_offset(int l, int row) {
  switch (l) {
    case L1: return row < 64 ? OFF_S0B + row : -1;
    case L2: return row < 64 ? OFF_S2B + row : -1;
    case L3: return row < 16 ? OFF_S4B + row : -1;
    case C1: return row < 64 ? OFF_C0B + row : -1;
    case C2: return row < 64 ? OFF_C2B + row : -1;
    case C3: return row < 3 ? OFF_C4B + row : -1;
  }
  return -1;
}

// row of a 32x32 accumulator tile held in register q of a lane in half h (cdna_hip_programming.md section 3)
__device__ __host__ constexpr int acc_row(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }

// ------------------------------------------------------------------------------------------------
// precision policies
// ------------------------------------------------------------------------------------------------
struct PBf16 {
  using frag = bf16x8;
  static constexpr int S32 = 2, S16 = 1, S8 = 1;  // k-steps covering 32 / 16 / 8 rows of a tile
  static constexpr int ELEMS = 8;
  // logical row (within its 32-row tile) that element j of the step-s fragment of a lane in half h stands for
  __device__ __host__ static constexpr int rho(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }
  __device__ static __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  // accumulate into a tile that lives in the accumulator (AGPR) half of the register file for the whole kernel.
  // Inline asm because the build forces the VGPR form on the builtin MFMAs (the chain's results are read by VALU code
  // right away); these dW tiles are only ever touched by the next MFMA, so keeping them in AGPRs costs no moves.
  // PAD: `s_nop 1` = the wait states between a VALU write of a/b (a v_mov the compiler may have placed) and the MFMA
  // reading them, which hipcc does not insert inside an asm statement (cdna_hip_programming.md 5.7 item 2); operands
  // that came straight from LDS reads need none, and an s_nop costs a 4-cycle issue slot at one wave per SIMD.
  // ORDERED: volatile, i.e. pinned in program order relative to the other volatile statements (xch_take slots VALU
  // work behind each MFMA).
  template <bool ORDERED, bool PAD>
  __device__ static __forceinline__ void mfma_acc_lds(frag a, frag b, f32x16& c) {
    if constexpr (PAD) {
      if constexpr (ORDERED) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
      else asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    } else {
      if constexpr (ORDERED) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
      else asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    }
  }
  // Vectors are always BUILT WHOLE from scalar words (initializer lists), never updated element by element: hipcc 7.2
  // miscompiles element-wise writes into (bit-cast) vectors depending on the surrounding code - seen in round 1 as
  // "all four words took the first word's value" and in round 2 as one instantiation (rows layout, fp32 features) of
  // the forward kernel returning 15 % wrong outputs after an unrelated change elsewhere in the file.
  __device__ static __forceinline__ frag from_words(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    const u32x4 w = {w0, w1, w2, w3};
    return __builtin_bit_cast(frag, w);
  }
  __device__ static __forceinline__ frag from_acc(const f32x16& acc, int s) {
    return from_words(pack_bf16x2(acc[8 * s], acc[8 * s + 1]), pack_bf16x2(acc[8 * s + 2], acc[8 * s + 3]),
                      pack_bf16x2(acc[8 * s + 4], acc[8 * s + 5]), pack_bf16x2(acc[8 * s + 6], acc[8 * s + 7]));
  }
  __device__ static __forceinline__ frag zero() { return from_words(0u, 0u, 0u, 0u); }
  // the three k-steps of the colour net's first layer: L3 tile rows 0..15, PE 0..15, PE 16..23
  __device__ static __forceinline__ void cin(const f32x16& acc3, float4 peA, float4 peB, float4 peC, frag (&out)[3]) {
    out[0] = from_acc(acc3, 0);
    out[1] = from_words(pack_bf16x2(peA.x, peA.y), pack_bf16x2(peA.z, peA.w), pack_bf16x2(peB.x, peB.y), pack_bf16x2(peB.z, peB.w));
    out[2] = from_words(pack_bf16x2(peC.x, peC.y), pack_bf16x2(peC.z, peC.w), 0u, 0u);
  }
  // features of one point for step s: element j <- feature rho(s,h,j); v[4] are the (f0,f1) pairs of levels
  // 8s+2h, 8s+2h+1, 8s+4+2h, 8s+4+2h+1
  __device__ static __forceinline__ frag feat_frag(const float2 (&v)[4]) {
    return from_words(pack_bf16x2(v[0].x, v[0].y), pack_bf16x2(v[1].x, v[1].y), pack_bf16x2(v[2].x, v[2].y), pack_bf16x2(v[3].x, v[3].y));
  }
};

struct PF32 {
  using frag = float;
  static constexpr int S32 = 16, S16 = 8, S8 = 4;
  static constexpr int ELEMS = 1;
  __device__ __host__ static constexpr int rho(int s, int h, int) { return acc_row(s, h); }
  __device__ static __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
  __device__ static __forceinline__ void mfma_acc(frag a, frag b, f32x16& c) {
    asm("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  }
  __device__ static __forceinline__ frag from_acc(const f32x16& acc, int s) { return acc[s]; }
  __device__ static __forceinline__ frag zero() { return 0.f; }
  __device__ static __forceinline__ frag ident(int s, int lane) { return ((lane & 31) == acc_row(s, lane >> 5)) ? 1.f : 0.f; }
  __device__ static __forceinline__ void cin(const f32x16& acc3, float4 peA, float4 peB, float4 peC, frag (&out)[20]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) out[q] = acc3[q];
    out[8] = peA.x; out[9] = peA.y; out[10] = peA.z; out[11] = peA.w;
    out[12] = peB.x; out[13] = peB.y; out[14] = peB.z; out[15] = peB.w;
    out[16] = peC.x; out[17] = peC.y; out[18] = peC.z; out[19] = peC.w;
  }
};

// ------------------------------------------------------------------------------------------------
// fragment-image tables (shared by the pack kernel and the compute kernels)
// ------------------------------------------------------------------------------------------------
// Reading a layer's bias into the accumulators costs 4 KiB of LDS traffic per 32x32 output tile (16 floats per lane) -
// more than the tile's weight fragments in the narrow layers, 40 of the forward's 74 KiB per point tile, and the LDS
// pipe (shared by the CU's four waves) is what bounds these kernels.  In bf16 mode the bias is instead one more
// k-step: a weight fragment whose k = 0, 1, 2 columns hold the bias split into three bf16 parts (hi + mid + lo
// reproduces the fp32 value to 2^-24 relative), multiplied by an activation fragment that is 1 on those k and 0
// elsewhere: 1 KiB and one MFMA per tile, accumulators start from the inline constant 0.
template <class P>
struct Tab {
  // forward: output-feature tiles and k-steps (input slots) per layer
  __device__ __host__ static constexpr int f_out_tiles(int l) { return (l == L3 || l == C3) ? 1 : 2; }
  // bf16: the bias rides on one more k-step per output tile (BIAS_STEP) instead of being read into the accumulators
  static constexpr bool BIAS_STEP = (P::ELEMS == 8);
  __device__ __host__ static constexpr int f_wsteps(int l) { return l == L1 ? P::S32 : (l == C1 ? P::S32 + P::S8 : 2 * P::S32); }
  __device__ __host__ static constexpr int f_ksteps(int l) { return f_wsteps(l) + (BIAS_STEP ? 1 : 0); }
  __device__ __host__ static constexpr int f_base(int l) {
    int b = 0;
    for (int i = 0; i < l; ++i) b += f_out_tiles(i) * f_ksteps(i);
    return b;
  }
  static constexpr int F_FRAGS = f_base(NLAYER);
  // backward (data gradient): output = input-feature tiles, k-steps over the layer's output features
  __device__ __host__ static constexpr int b_out_tiles(int l) { return (l == L1 || l == C1) ? 1 : 2; }
  __device__ __host__ static constexpr int b_ksteps(int l) { return l == L3 ? P::S16 : (l == C3 ? P::S8 : 2 * P::S32); }
  __device__ __host__ static constexpr int b_base(int l) {
    int b = F_FRAGS;
    for (int i = 0; i < l; ++i) b += b_out_tiles(i) * b_ksteps(i);
    return b;
  }
  static constexpr int ALL_FRAGS = b_base(NLAYER);
  static constexpr int FRAG_BYTES = (int)sizeof(typename P::frag) * 64;
  static constexpr int BIAS_OFF_F = F_FRAGS * FRAG_BYTES;      // when only the forward image is staged
  static constexpr int BIAS_OFF_ALL = ALL_FRAGS * FRAG_BYTES;  // in the full image
  static constexpr int BIAS_BYTES = NLAYER * 64 * 4;
  static constexpr int IMG_BYTES = BIAS_OFF_ALL + BIAS_BYTES;
};

// global image: [ALL_FRAGS][64 lanes] fragments, then [6][64] padded biases
template <class P>
__device__ __forceinline__ void pack_body(const float* __restrict__ params, char* __restrict__ img, int block, int nblocks) {
  using T = Tab<P>;
  const int total = T::ALL_FRAGS * 64;
  for (int e = block * 256 + threadIdx.x; e < total + NLAYER * 64; e += nblocks * 256) {
    if (e >= total) {  // biases
      const int b = e - total, l = b / 64, row = b % 64;
      const int off = blog_offset(l, row);
      ((float*)(img + T::BIAS_OFF_ALL))[b] = off >= 0 ? params[off] : 0.f;
      continue;
    }
    const int fid = e / 64, lane = e % 64, r = lane & 31, h = lane >> 5;
    const bool fwd = fid < T::F_FRAGS;
    int l = 0, base = fwd ? 0 : T::F_FRAGS;
    for (;; ++l) {
      const int cnt = fwd ? T::f_out_tiles(l) * T::f_ksteps(l) : T::b_out_tiles(l) * T::b_ksteps(l);
      if (fid < base + cnt) break;
      base += cnt;
    }
    const int nk = fwd ? T::f_ksteps(l) : T::b_ksteps(l);
    const int m = (fid - base) / nk, ks = (fid - base) % nk;
    const int n = ks / P::S32, s = ks % P::S32;
    typename P::frag f;
    float vals[8];
    if (fwd && T::BIAS_STEP && ks == nk - 1) {  // the bias step: k = 0, 1, 2 <- the three bf16 parts of bias[32m + r]
      const int off = blog_offset(l, 32 * m + r);
      float part[3] = {0.f, 0.f, 0.f}, rest = off >= 0 ? params[off] : 0.f;
      for (int i = 0; i < 3; ++i) {
        part[i] = (float)(__bf16)rest;
        rest -= part[i];  // exact: rest and part[i] share their leading bits
      }
#pragma unroll
      for (int j = 0; j < P::ELEMS; ++j) {
        const int kk = P::rho(0, h, j);
        vals[j] = kk < 3 ? part[kk] : 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < P::ELEMS; ++j) {
        const int kk = 32 * n + P::rho(s, h, j);
        const int off = fwd ? wlog_offset(l, 32 * m + r, kk) : wlog_offset(l, kk, 32 * m + r);
        vals[j] = off >= 0 ? params[off] : 0.f;
      }
    }
    if constexpr (P::ELEMS == 8) {
      f = P::from_words(pack_bf16x2(vals[0], vals[1]), pack_bf16x2(vals[2], vals[3]), pack_bf16x2(vals[4], vals[5]), pack_bf16x2(vals[6], vals[7]));
    } else {
      f = vals[0];
    }
    ((typename P::frag*)img)[e] = f;
  }
}
template <class P>
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ params, char* __restrict__ img) {
  pack_body<P>(params, img, (int)blockIdx.x, (int)gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// device building blocks
// ------------------------------------------------------------------------------------------------
// `lofs` = lane * sizeof(frag), laundered through an empty asm once per point tile (opaque_lane_offset) so the
// compiler cannot prove the weight loads loop-invariant: hoisting all ~62 fragments (250 registers) out of the tile
// loop is what it does otherwise, and the kernel then spills.
template <class P>
__device__ __forceinline__ typename P::frag ldw(const char* img, int fid, int lofs) {
  return *(const typename P::frag*)(img + fid * 64 * (int)sizeof(typename P::frag) + lofs);
}
template <class P>
__device__ __forceinline__ int opaque_lane_offset(int lane) {
  int v = lane * (int)sizeof(typename P::frag);
  asm volatile("" : "+v"(v));
  return v;
}

// acc[m] = bias ; acc[m] += sum_ks W_frag(m,ks) * x[ks]      (orientation 1)
// BIAS: 0 none (backward), 1 read into the accumulators, 2 the extra k-step of the bf16 image (a bf16 image always
// holds that fragment behind each output tile's weight fragments, whichever way the bias is applied)
template <class P, int NOUT, int NK, int BIAS>
__device__ __forceinline__ void dense(const char* img, int fbase, const float* bias, int lane, int lofs,
                                      const typename P::frag (&x)[NK], f32x16 (&acc)[NOUT]) {
  const int h = lane >> 5;
  const int bofs = lofs - lane * (int)sizeof(typename P::frag);  // opaque zero: keeps the bias loads inside the loop too
  constexpr bool BSTEP = BIAS == 2 && Tab<P>::BIAS_STEP;
  constexpr int MSTRIDE = NK + ((BIAS && Tab<P>::BIAS_STEP) ? 1 : 0);  // fragments per output tile in the image
#pragma unroll
  for (int m = 0; m < NOUT; ++m) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (BIAS && !BSTEP) b = *(const float4*)((const char*)bias + bofs + 4 * (32 * m + 8 * g + 4 * h));
      acc[m][4 * g + 0] = b.x; acc[m][4 * g + 1] = b.y; acc[m][4 * g + 2] = b.z; acc[m][4 * g + 3] = b.w;
    }
  }
  if constexpr (BSTEP) {
    // activation fragment of the bias step: k = rho(0, h, j) in {0, 1, 2} <-> elements 0..2 of the h = 0 lanes
    const typename P::frag ones = P::from_words(h ? 0u : 0x3f803f80u, h ? 0u : 0x00003f80u, 0u, 0u);
#pragma unroll
    for (int m = 0; m < NOUT; ++m) {  // all of a tile's fragments are requested before its first MFMA
      typename P::frag wf[NK + 1];
#pragma unroll
      for (int g = 0; g <= NK; ++g) wf[g] = ldw<P>(img, fbase + m * (NK + 1) + g, lofs);
#pragma unroll
      for (int g = 0; g < NK; ++g) acc[m] = P::mfma(wf[g], x[g], acc[m]);
      acc[m] = P::mfma(wf[NK], ones, acc[m]);
    }
    return;
  }
  // The weight fragments of a group of k-steps are all requested before the first MFMA of the group: left to itself
  // hipcc emits ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma per k-step, exposing the full LDS latency ~100 times per tile.
  constexpr int G = (P::ELEMS == 8) ? 4 : 8;  // k-steps per group (4 x 4 VGPRs for bf16, 8 x 1 for f32)
#pragma unroll
  for (int m = 0; m < NOUT; ++m) {
#pragma unroll
    for (int k0 = 0; k0 < NK; k0 += G) {
      typename P::frag wf[G];
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (k0 + g < NK) wf[g] = ldw<P>(img, fbase + m * MSTRIDE + k0 + g, lofs);
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (k0 + g < NK) acc[m] = P::mfma(wf[g], x[k0 + g], acc[m]);
    }
  }
}

// Packed 16-bit helpers on bf16 pairs (one VGPR = two activations).  For `fmaxf(x, 0)` hipcc emits a canonicalising
// `v_max_f32 x, x, x` in front of the max - 2 instructions per activation, 256 per tile - and it scalarises 8-wide `short`
// vectors; on 2-wide vectors `__builtin_elementwise_max` / `min` select v_pk_max_i16 / v_pk_max_u16 / v_pk_min_u16 directly.
//   ReLU of a bf16 = signed 16-bit max with 0 (negative floats are negative integers; -0 -> +0).
//   "was the ReLU output positive" = its bits are non-zero: min(bits, 1) is 0 / 1, and a 16-bit multiply by it keeps
//   or clears a gradient's bits.
// Round 4: these were inline asm until the round-3 two-tile experiment's "first tile of every wave is wrong" was traced
// (tools/dev/mfma_operand_hazard_scan.py): gfx950 needs 2 wait states between a VALU write of a VGPR and an MFMA reading
// it as an operand; hipcc pads that pair only when it KNOWS the writer is a VALU instruction.  An `asm` statement is
// opaque to it - and, not being volatile, free to be scheduled one slot ahead of the consuming MFMA, which is what the
// experiment's prologue forward got (nine `v_pk_max_i16 vN ; s_waitcnt ; v_mfma ... vN` triples: stale operands whenever
// the s_waitcnt did not happen to stall).  The shipped kernels carried 23 such triples too (K3: 3 per planar
// instantiation; K4: 2), masked only by their LDS waits.  Builtins put the hazard back into the compiler's hands;
// tests/test_lib_abi.py scans the built library's disassembly for any VALU -> MFMA-operand pair closer than 2 slots.
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_relu_bf16(uint32_t w) {
  const s16x2 z = {0, 0};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), z));
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
// The mask epilogue stays inline asm: it is slotted word by word behind the owners' asm MFMAs (volatile = pinned in
// program order relative to them; one statement for the dependent pair - between two statements hipcc pads an `s_nop 0`,
// 4 issue cycles at one wave per SIMD).  Its outputs are therefore asm-VALU writes: whoever assembles them into MFMA
// operands calls asm_valu_pad() on the finished fragments first.
__device__ __forceinline__ uint32_t pk_keep_where_nonzero_ordered(uint32_t grad, uint32_t act, uint32_t ones /* 0x00010001 */) {
  uint32_t r;
  asm volatile("v_pk_min_u16 %0, %2, %3\n\tv_pk_mul_lo_u16 %0, %1, %0" : "=&v"(r) : "v"(grad), "v"(act), "v"(ones));
  return r;
}
// Two wait states behind inline-asm VALU writes of the fragments `f[0..N)`, in front of every consumer: the statement
// takes the fragments as read-write operands, so it is ordered after their producers and before their readers whatever
// hipcc's scheduler does with the surrounding code.  (8 issue cycles; N <= 4 fragments = 16 VGPR operands per statement.)
template <int N>
__device__ __forceinline__ void asm_valu_pad(bf16x8 (&f)[N]) {
  static_assert(N == 2 || N == 4, "fragments of one or two 32-feature tiles");
  if constexpr (N == 2) asm volatile("s_nop 1" : "+v"(f[0]), "+v"(f[1]));
  else asm volatile("s_nop 1" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
}

// ReLU + re-pack.  No mask is recorded: the backward pass reads the sign back from the packed activations themselves
// (a ReLU output is > 0 exactly where its bits are non-zero).  bf16: round first, then one packed integer max per PAIR
// (rounding is monotonic and keeps the sign, so relu(round(x)) == round(relu(x))); f32: an integer max on the bits.
template <class P, int NT>
__device__ __forceinline__ void relu_frags(f32x16 (&acc)[NT], typename P::frag (&out)[NT * P::S32]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if constexpr (P::ELEMS == 8) {
#pragma unroll
      for (int s = 0; s < P::S32; ++s) {
        const f32x16& a = acc[t];
        out[t * P::S32 + s] = PBf16::from_words(pk_relu_bf16(pack_bf16x2(a[8 * s], a[8 * s + 1])), pk_relu_bf16(pack_bf16x2(a[8 * s + 2], a[8 * s + 3])),
                                                pk_relu_bf16(pack_bf16x2(a[8 * s + 4], a[8 * s + 5])), pk_relu_bf16(pack_bf16x2(a[8 * s + 6], a[8 * s + 7])));
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[t][q] = __int_as_float(max(__float_as_int(acc[t][q]), 0));
#pragma unroll
      for (int s = 0; s < P::S32; ++s) out[t * P::S32 + s] = P::from_acc(acc[t], s);
    }
  }
}

// f32: dZ = dX where the forward activation h was positive, else 0 (bf16 masks packed pairs inside mask_take)
template <class P, int NT>
__device__ __forceinline__ void mask_frags(f32x16 (&acc)[NT], const typename P::frag (&h)[NT * P::S32],
                                           typename P::frag (&out)[NT * P::S32]) {
  static_assert(P::ELEMS == 1, "f32 fragments");
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int s = 0; s < P::S32; ++s) acc[t][s] = (h[t * P::S32 + s] > 0.f) ? acc[t][s] : 0.f;
#pragma unroll
    for (int s = 0; s < P::S32; ++s) out[t * P::S32 + s] = P::from_acc(acc[t], s);
  }
}

// Development-only phase timer (tools/k4_phases.py builds a variant with -DHBR_K4_PROF=1; never in the shipped
// library): wave 0 of workgroup 0 adds the shader-clock cycles between successive marks to k4_prof[phase].
#ifndef HBR_K4_PROF
#define HBR_K4_PROF 0
#endif
#if HBR_K4_PROF
__device__ unsigned long long k4_prof[64];
struct PhaseClock {
  long long t0 = 0;
  bool on = false;
  long long c0 = 0, r0 = 0;
  __device__ __forceinline__ void start() {
    on = blockIdx.x == 0 && threadIdx.x == 0;
    c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    t0 = clock64();
  }
  // shader-clock and 100 MHz reference ticks of the whole sweep -> the clock the chip held (k4_prof[62] / [63] x 100 MHz)
  __device__ __forceinline__ void finish() {
    const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (on) { k4_prof[62] += (unsigned long long)(c1 - c0); k4_prof[63] += (unsigned long long)(r1 - r0); }
  }
  __device__ __forceinline__ void mark(int i) {
    if (HBR_K4_PROF == 2) return;  // whole-sweep stamps only: the unperturbed cycle count and clock
    const long long t1 = clock64();
    if (on) k4_prof[i] += (unsigned long long)(t1 - t0);
    t0 = clock64();
  }
  __device__ __forceinline__ void mark_fine(int i) {
    if (HBR_K4_PROF == 3) { const int keep = HBR_K4_PROF; (void)keep; mark(i); }
  }
  // the same after the values in `v` exist (an MFMA result, a re-packed fragment): pins the producers before the stamp
  template <class V>
  __device__ __forceinline__ void mark_after(int i, V& v) {
    if (HBR_K4_PROF == 2) return;
    asm volatile("" : "+v"(v));
    mark(i);
  }
};
#else
struct PhaseClock {
  __device__ __forceinline__ void start() {}
  __device__ __forceinline__ void finish() {}
  __device__ __forceinline__ void mark(int) {}
  __device__ __forceinline__ void mark_fine(int) {}
  template <class V>
  __device__ __forceinline__ void mark_after(int, V&) {}
};
#endif
struct FeatSrc {
  const void* p;
  int64_t stride;  // rows layout
  uint32_t N;
  uint32_t addr32;  // every byte offset the tile loads / stores form (features, d out, directions) fits 32 bits
};
// N and N / group up to which FeatSrc::addr32 holds: 3 N x 8 B (planar fp32: lane half h reads level 2h + k), 16 B x N
// (d out), 96 B x rays
constexpr int64_t kAddr32MaxN = 1LL << 27, kAddr32MaxRays = 1LL << 25;

struct PeSrc {
  const float* pe;  // [G,24] encoded view directions
  uint32_t group;   // points per row of pe (S for per-ray directions, 1 for per-point)
};

// raw per-lane inputs of one 32-point tile; loaded one tile ahead of use in the backward kernel (software prefetch:
// with one or two waves per SIMD nothing else hides the HBM latency of these loads)
struct TileIn {
  float2 v[8];              // (f0,f1) of the lane's 8 levels; bf16 storage: .x holds the RAW packed pair, unpacked at use
                            // (an unpack at load time would wait for the prefetched tile right away)
  float4 peA, peB, peC;     // the lane's 12 direction-encoding values
  float4 dO;                // d out (backward only)
};

// Where a lane's next tile is: its point n = ray * group + rem.  A sweep advances every lane by the same number of
// points per round, so the ray index is carried along with two adds and a compare instead of a 32-bit division per
// tile (~35 VALU), and in the planar layout the loads take a 32-bit offset from a wave-uniform (SGPR) level base
// instead of 64-bit per-lane address arithmetic.
struct TileCursor {
  uint32_t n, ray, rem;
  uint32_t dn, dray, drem, group;
  __device__ __forceinline__ void start(uint32_t n0, uint32_t points_per_round, uint32_t g) {
    group = g; n = n0; ray = n0 / g; rem = n0 - ray * g;
    dn = points_per_round; dray = points_per_round / g; drem = points_per_round - dray * g;
  }
  __device__ __forceinline__ void advance() {
    n += dn; ray += dray; rem += drem;
    const bool carry = rem >= group;
    rem -= carry ? group : 0u;
    ray += carry ? 1u : 0u;
  }
};

// byte offset of (level 2h, point n) from the start of a planar buffer with ES-byte (f0,f1) pairs; the lane's level
// 4(k>>1) + 2h + (k&1) is (4(k>>1) + (k&1)) * N * ES further - a wave-uniform amount
template <int ES>
__device__ __forceinline__ uint32_t planar_off(uint32_t N, uint32_t n, int h) { return (2u * (uint32_t)h * N + n) * ES; }
template <int ES>
__device__ __forceinline__ size_t planar_level_base(uint32_t N, int k) { return (size_t)(4 * (k >> 1) + (k & 1)) * N * ES; }

template <int LAYOUT, int DT, bool WITH_DOUT>
__device__ __forceinline__ void load_tile_in(const FeatSrc& fs, const PeSrc& ps, const float* dout, const TileCursor& c, bool valid,
                                             int h, TileIn& ti) {
  const uint32_t n = c.n;
  if (LAYOUT == HBR_LAYOUT_PLANAR && fs.addr32) {
    constexpr int ES = DT == HBR_F32 ? 8 : 4;
    const uint32_t off = planar_off<ES>(fs.N, n, h);
#pragma unroll
    for (int k = 0; k < 8; ++k) ti.v[k] = make_float2(0.f, 0.f);
    ti.peA = make_float4(0, 0, 0, 0); ti.peB = ti.peA; ti.peC = ti.peA; ti.dO = ti.peA;
    if (valid) {  // one branch around all of the tile's loads
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const char* lb = (const char*)fs.p + planar_level_base<ES>(fs.N, k);
        if (DT == HBR_F32) ti.v[k] = *(const float2*)(lb + off);
        else ti.v[k].x = __uint_as_float(*(const uint32_t*)(lb + off));
      }
      const char* pr = (const char*)ps.pe + (c.ray * 96u + 16u * (uint32_t)h);
      ti.peA = *(const float4*)pr;
      ti.peB = *(const float4*)(pr + 32);
      ti.peC = *(const float4*)(pr + 64);
      if (WITH_DOUT && dout && h == 0) ti.dO = *(const float4*)((const char*)dout + n * 16u);
    }
    return;
  }
  // feature pair (2l, 2l+1) of level l; a lane in half h owns levels {2h,2h+1, 4+2h,4+2h+1, 8+.., 12+..}
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int lvl = 4 * (k >> 1) + 2 * h + (k & 1);
    ti.v[k] = make_float2(0.f, 0.f);
    if (valid) {
      if (LAYOUT == HBR_LAYOUT_PLANAR) {
        if (DT == HBR_F32) ti.v[k] = ((const float2*)fs.p)[(size_t)lvl * fs.N + n];
        else ti.v[k].x = __uint_as_float(((const uint32_t*)fs.p)[(size_t)lvl * fs.N + n]);
      } else {
        if (DT == HBR_F32) ti.v[k] = *(const float2*)((const float*)fs.p + (size_t)n * fs.stride + 2 * lvl);
        else ti.v[k].x = __uint_as_float(*(const uint32_t*)((const uint16_t*)fs.p + (size_t)n * fs.stride + 2 * lvl));
      }
    }
  }
  ti.peA = make_float4(0, 0, 0, 0); ti.peB = ti.peA; ti.peC = ti.peA; ti.dO = ti.peA;
  if (valid) {
    const float* pr = ps.pe + (size_t)c.ray * 24;
    ti.peA = *(const float4*)(pr + 4 * h);
    ti.peB = *(const float4*)(pr + 8 + 4 * h);
    ti.peC = *(const float4*)(pr + 16 + 4 * h);
    if (WITH_DOUT && dout && h == 0) ti.dO = ((const float4*)dout)[n];
  }
}

// The fast path's loads of one tile in SIX parts (two vector-memory instructions each), for the backward kernel: its four
// waves run in lockstep, so twelve loads issued back to back by each of them queue at the CU's one address unit (16
// cycles per 64-lane instruction: ~660 cycles of issue stall per tile, profiles/r02_h_k4_phases.txt); two at a time
// between the exchange steps they are taken in the shadow of the MFMA / LDS work.  PART 0..3: level pairs; 4: peA, peB;
// 5: peC, d out.  Same addresses and values as load_tile_in's fast path.
template <int DT, int PART>
__device__ __forceinline__ void load_tile_part(const FeatSrc& fs, const PeSrc& ps, const float* dout, const TileCursor& c, bool valid,
                                               int h, TileIn& ti) {
  constexpr int ES = DT == HBR_F32 ? 8 : 4;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (PART < 4) {
    ti.v[2 * PART] = make_float2(0.f, 0.f);
    ti.v[2 * PART + 1] = make_float2(0.f, 0.f);
    if (valid) {
      const uint32_t off = planar_off<ES>(fs.N, c.n, h);
#pragma unroll
      for (int k = 2 * PART; k < 2 * PART + 2; ++k) {
        const char* lb = (const char*)fs.p + planar_level_base<ES>(fs.N, k);
        if (DT == HBR_F32) ti.v[k] = *(const float2*)(lb + off);
        else ti.v[k].x = __uint_as_float(*(const uint32_t*)(lb + off));
      }
    }
  } else if constexpr (PART == 4) {
    ti.peA = z4; ti.peB = z4;
    if (valid) {
      const char* pr = (const char*)ps.pe + (c.ray * 96u + 16u * (uint32_t)h);
      ti.peA = *(const float4*)pr;
      ti.peB = *(const float4*)(pr + 32);
    }
  } else {
    ti.peC = z4; ti.dO = z4;
    if (valid) {
      const char* pr = (const char*)ps.pe + (c.ray * 96u + 16u * (uint32_t)h);
      ti.peC = *(const float4*)(pr + 64);
      if (dout && h == 0) ti.dO = *(const float4*)((const char*)dout + c.n * 16u);  // (dout == nullptr: the render variant forms d out itself)
    }
  }
}

// one point's 32 features -> the S32 fragments of the single input tile
template <class P, int DT>
__device__ __forceinline__ void feat_frags(const TileIn& ti, typename P::frag (&x)[P::S32]) {
  if constexpr (P::ELEMS == 8) {
    // step s, element j <-> feature 16s + 8(j>>2) + 4h + (j&3)  == levels 8s+2h, 8s+2h+1, 8s+4+2h, 8s+4+2h+1
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if constexpr (DT == HBR_BF16) {
        // stored pairs ARE the fragment's words: (f0 | f1 << 16) of the four levels, no conversion
        const u32x4 w = {__float_as_uint(ti.v[4 * s + 0].x), __float_as_uint(ti.v[4 * s + 1].x),
                         __float_as_uint(ti.v[4 * s + 2].x), __float_as_uint(ti.v[4 * s + 3].x)};
        x[s] = __builtin_bit_cast(bf16x8, w);
      } else {
        const float2 w[4] = {ti.v[4 * s + 0], ti.v[4 * s + 1], ti.v[4 * s + 2], ti.v[4 * s + 3]};
        x[s] = PBf16::feat_frag(w);
      }
    }
  } else {
    // step q <-> feature (q&3) + 8(q>>2) + 4h : level 4(q>>2) + 2h + ((q&3)>>1), component q&1
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float2 p = ti.v[2 * (q >> 2) + ((q & 3) >> 1)];
      if constexpr (DT == HBR_BF16) p = make_float2(bf16_lo(__float_as_uint(p.x)), bf16_hi(__float_as_uint(p.x)));
      x[q] = (q & 1) ? p.y : p.x;
    }
  }
}

// everything the backward pass needs from the recomputed forward of one 32-point tile
template <class P>
struct Saved {
  typename P::frag x0[P::S32];
  typename P::frag h1[2 * P::S32];
  typename P::frag h2[2 * P::S32];
  typename P::frag cin[P::S32 + P::S8];
  typename P::frag c1[2 * P::S32];
  typename P::frag c2[2 * P::S32];
  float s0;            // raw density (row 0 of the L3 tile; meaningful on h == 0 lanes)
  float raw[3];        // raw rgb (rows 0..2 of the C3 tile; h == 0 lanes)
};


// One layer of the forward chain on the fragments saved so far.  BSTEP: bf16 bias by the extra k-step (the backward
// kernel: 40 fewer LDS reads and 32 fewer live registers per tile) or read into the accumulators (the forward kernel,
// which has the occupancy to hide those reads and is 10 % faster without the ten extra MFMAs).
template <class P, bool BSTEP, int LAYER>
__device__ __forceinline__ void forward_layer(const char* img, const float* bias, int lane, int lofs, float4 peA, float4 peB,
                                              float4 peC, Saved<P>& sv, PhaseClock& pc) {
  using T = Tab<P>;
  constexpr int B = BSTEP ? 2 : 1;
  if constexpr (LAYER == L1) {
    f32x16 a[2];
    dense<P, 2, P::S32, B>(img, T::f_base(L1), bias + 64 * L1, lane, lofs, sv.x0, a);
    pc.mark_after(21, a[1]);
    relu_frags<P, 2>(a, sv.h1);
    pc.mark_after(22, sv.h1[2 * P::S32 - 1]);
  } else if constexpr (LAYER == L2) {
    f32x16 a[2];
    dense<P, 2, 2 * P::S32, B>(img, T::f_base(L2), bias + 64 * L2, lane, lofs, sv.h1, a);
    pc.mark_after(23, a[1]);
    relu_frags<P, 2>(a, sv.h2);
    pc.mark_after(24, sv.h2[2 * P::S32 - 1]);
  } else if constexpr (LAYER == L3) {
    f32x16 a[1];
    dense<P, 1, 2 * P::S32, B>(img, T::f_base(L3), bias + 64 * L3, lane, lofs, sv.h2, a);
    pc.mark_after(25, a[0]);
    sv.s0 = a[0][0];
    P::cin(a[0], peA, peB, peC, sv.cin);
    pc.mark_after(26, sv.cin[P::S32 + P::S8 - 1]);
  } else if constexpr (LAYER == C1) {
    f32x16 a[2];
    dense<P, 2, P::S32 + P::S8, B>(img, T::f_base(C1), bias + 64 * C1, lane, lofs, sv.cin, a);
    pc.mark_after(27, a[1]);
    relu_frags<P, 2>(a, sv.c1);
    pc.mark_after(28, sv.c1[2 * P::S32 - 1]);
  } else if constexpr (LAYER == C2) {
    f32x16 a[2];
    dense<P, 2, 2 * P::S32, B>(img, T::f_base(C2), bias + 64 * C2, lane, lofs, sv.c1, a);
    pc.mark_after(29, a[1]);
    relu_frags<P, 2>(a, sv.c2);
    pc.mark_after(30, sv.c2[2 * P::S32 - 1]);
  } else {
    f32x16 a[1];
    dense<P, 1, 2 * P::S32, B>(img, T::f_base(C3), bias + 64 * C3, lane, lofs, sv.c2, a);
    pc.mark_after(31, a[0]);
    sv.raw[0] = a[0][0]; sv.raw[1] = a[0][1]; sv.raw[2] = a[0][2];
  }
}

// DT: storage type of the feature buffer the tile was loaded from
template <class P, int DT, bool BSTEP>
__device__ __forceinline__ void forward_tile(const char* img, const float* bias, const TileIn& ti, int lane, Saved<P>& sv,
                                             PhaseClock& pc) {
  const int lofs = opaque_lane_offset<P>(lane);
  feat_frags<P, DT>(ti, sv.x0);
  pc.mark_after(20, sv.x0[P::S32 - 1]);  // the tile's inputs have arrived
  forward_layer<P, BSTEP, L1>(img, bias, lane, lofs, ti.peA, ti.peB, ti.peC, sv, pc);
  forward_layer<P, BSTEP, L2>(img, bias, lane, lofs, ti.peA, ti.peB, ti.peC, sv, pc);
  forward_layer<P, BSTEP, L3>(img, bias, lane, lofs, ti.peA, ti.peB, ti.peC, sv, pc);
  forward_layer<P, BSTEP, C1>(img, bias, lane, lofs, ti.peA, ti.peB, ti.peC, sv, pc);
  forward_layer<P, BSTEP, C2>(img, bias, lane, lofs, ti.peA, ti.peB, ti.peC, sv, pc);
  forward_layer<P, BSTEP, C3>(img, bias, lane, lofs, ti.peA, ti.peB, ti.peC, sv, pc);
}

// The backward kernel's forward recompute (bf16, one wave per SIMD): every layer's weight fragments - its bias step
// included - are requested one layer AHEAD, in front of the previous layer's MFMAs, and a scheduling fence that LDS
// operations may not cross keeps hipcc from sinking the reads back next to their MFMAs (it otherwise emits
// ds_read -> s_waitcnt -> v_mfma per k-step, exposing the LDS latency several times per layer): 12.65 k -> 12.2 k cycles
// per tile, of which the chip takes a part back as clock (2.13 -> 2.08 GHz): 391.7 -> 382.6 us.
constexpr int kFwdFrags = 10;  // 2 output tiles x (4 k-steps + the bias step)
template <int LAYER>
struct FwdShape {
  static constexpr int NOUT = (LAYER == L3 || LAYER == C3) ? 1 : 2;
  static constexpr int NK = Tab<PBf16>::f_wsteps(LAYER);
  static constexpr int FRAGS = NOUT * (NK + 1);
};
template <int LAYER>
__device__ __forceinline__ void fwd_request(const char* img, int lofs, bf16x8 (&w)[kFwdFrags]) {
  static_assert(FwdShape<LAYER>::FRAGS <= kFwdFrags, "fragments of one forward layer");
#pragma unroll
  for (int i = 0; i < FwdShape<LAYER>::FRAGS; ++i) w[i] = ldw<PBf16>(img, Tab<PBf16>::f_base(LAYER) + i, lofs);
  __builtin_amdgcn_sched_barrier(0);  // nothing is scheduled across: the requests stay in front of the previous layer
}
template <int LAYER>
__device__ __forceinline__ void fwd_dense(const bf16x8 (&w)[kFwdFrags], const bf16x8 (&x)[FwdShape<LAYER>::NK], bf16x8 ones,
                                          f32x16 (&acc)[FwdShape<LAYER>::NOUT]) {
  constexpr int NK = FwdShape<LAYER>::NK;
#pragma unroll
  for (int m = 0; m < FwdShape<LAYER>::NOUT; ++m) {
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[m][q] = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) acc[m] = PBf16::mfma(w[m * (NK + 1) + k], x[k], acc[m]);
    acc[m] = PBf16::mfma(w[m * (NK + 1) + NK], ones, acc[m]);
  }
}
template <int DT>
__device__ __forceinline__ void forward_tile_prefetched(const char* img, const TileIn& ti, int lane, Saved<PBf16>& sv, PhaseClock& pc) {
  using P = PBf16;
  const int lofs = opaque_lane_offset<P>(lane);
  const int h = lane >> 5;
  const bf16x8 ones = P::from_words(h ? 0u : 0x3f803f80u, h ? 0u : 0x00003f80u, 0u, 0u);  // dense(): the bias step's activations
  bf16x8 wa[kFwdFrags], wb[kFwdFrags];
  fwd_request<L1>(img, lofs, wa);
  feat_frags<P, DT>(ti, sv.x0);
  pc.mark_after(20, sv.x0[P::S32 - 1]);
  fwd_request<L2>(img, lofs, wb);
  {
    f32x16 a[2];
    fwd_dense<L1>(wa, sv.x0, ones, a);
    pc.mark_after(21, a[1]);
    relu_frags<P, 2>(a, sv.h1);
  }
  fwd_request<L3>(img, lofs, wa);
  {
    f32x16 a[2];
    fwd_dense<L2>(wb, sv.h1, ones, a);
    pc.mark_after(23, a[1]);
    relu_frags<P, 2>(a, sv.h2);
  }
  fwd_request<C1>(img, lofs, wb);
  {
    f32x16 a[1];
    fwd_dense<L3>(wa, sv.h2, ones, a);
    pc.mark_after(25, a[0]);
    sv.s0 = a[0][0];
    P::cin(a[0], ti.peA, ti.peB, ti.peC, sv.cin);
  }
  fwd_request<C2>(img, lofs, wa);
  {
    f32x16 a[2];
    fwd_dense<C1>(wb, sv.cin, ones, a);
    pc.mark_after(27, a[1]);
    relu_frags<P, 2>(a, sv.c1);
  }
  fwd_request<C3>(img, lofs, wb);
  {
    f32x16 a[2];
    fwd_dense<C2>(wa, sv.c1, ones, a);
    pc.mark_after(29, a[1]);
    relu_frags<P, 2>(a, sv.c2);
  }
  {
    f32x16 a[1];
    fwd_dense<C3>(wb, sv.c2, ones, a);
    pc.mark_after(31, a[0]);
    sv.raw[0] = a[0][0]; sv.raw[1] = a[0][1]; sv.raw[2] = a[0][2];
  }
}

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }          // test_hash.py:38,67
__device__ __forceinline__ float lrelu(float x) { return x > 0.f ? x : 0.01f * x; }          // test_hash.py:39,62

// cooperative copy global image -> LDS (16-byte vectors)
__device__ __forceinline__ void stage_image(char* dst, const char* src, int bytes) {
  for (int i = threadIdx.x * 16; i < bytes; i += blockDim.x * 16) *(float4*)(dst + i) = *(const float4*)(src + i);
}

// ------------------------------------------------------------------------------------------------
// forward kernel
// ------------------------------------------------------------------------------------------------
// Waves per workgroup = waves sharing one staged weight image.  With 4 (round 1) the 46 KiB image let three workgroups
// = 3 waves per SIMD onto a CU; 16 waves around one image give the 4 per SIMD the 117-126 VGPRs allow:
// 0.087-0.089 ms (4), 0.078-0.080 (8), 0.075-0.078 (16) for pack + kernel at 2 M points.
constexpr int kFwdWaves = 16;

template <class P, int LAYOUT, int DT>
__global__ __launch_bounds__(kFwdWaves * 64) void mlp_fwd_kernel(const char* __restrict__ gimg, FeatSrc fs, PeSrc ps,
                                                                 float* __restrict__ out, const uint8_t* __restrict__ keep) {
  using T = Tab<P>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // forward fragments, then the bias block right behind them
  stage_image(smem, gimg, T::BIAS_OFF_F);
  stage_image(smem + T::BIAS_OFF_F, gimg + T::BIAS_OFF_ALL, T::BIAS_BYTES);
  __syncthreads();
  const float* bias = (const float*)(smem + T::BIAS_OFF_F);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t ntiles = (fs.N + 31) / 32;
  TileCursor cur;
  cur.start((blockIdx.x * kFwdWaves + wv) * 32 + (lane & 31), gridDim.x * kFwdWaves * 32, ps.group);
  for (uint32_t tile = blockIdx.x * kFwdWaves + wv; tile < ntiles; tile += gridDim.x * kFwdWaves, cur.advance()) {
    const uint32_t n = cur.n;
    const bool valid = n < fs.N;
    Saved<P> sv;
    TileIn ti;
    load_tile_in<LAYOUT, DT, false>(fs, ps, nullptr, cur, valid, lane >> 5, ti);
    PhaseClock pc;
    // (the backward kernel's prefetched forward was measured here too: 0.090-0.096 vs 0.088-0.089 ms - with three
    // waves per SIMD the latency is already hidden and the ten bias MFMAs cost more than the bias reads)
    forward_tile<P, DT, false>(smem, bias, ti, lane, sv, pc);
    if (valid && lane < 32) {
      // a sample whose occupancy cell is False keeps the zeros the reference initialises sigma/rgb with (vol_renderer.py:213-217)
      const bool kept = !keep || keep[n];
      ((float4*)out)[n] = kept ? make_float4(elu1(sv.raw[0]), elu1(sv.raw[1]), elu1(sv.raw[2]), lrelu(sv.s0)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward kernel
// ------------------------------------------------------------------------------------------------
// orientation-2 fragments of a tensor given its orientation-1 fragments: xt[tile][s'] (k = points).
// `colsum[t]` (optional) receives the lane's sum over its 16 point registers of tile t, taken from the fp32 tile
// before it is re-packed - the bias gradient, without unpacking bf16 again.
template <class P, int NT, int NK, bool SUM>
__device__ __forceinline__ void transpose_frags(const typename P::frag (&x)[NK], int lane, typename P::frag (&xt)[NT][P::S32],
                                                float* colsum = nullptr) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
    for (int s = 0; s < P::S32; ++s) {
      if (t * P::S32 + s < NK) acc = P::mfma(x[t * P::S32 + s], P::ident(s, lane), acc);
    }
    if (SUM) {
      float a = (acc[0] + acc[1]) + (acc[2] + acc[3]), b = (acc[4] + acc[5]) + (acc[6] + acc[7]);
      float c = (acc[8] + acc[9]) + (acc[10] + acc[11]), d = (acc[12] + acc[13]) + (acc[14] + acc[15]);
      colsum[t] += (a + b) + (c + d);
    }
#pragma unroll
    for (int s = 0; s < P::S32; ++s) xt[t][s] = P::from_acc(acc, s);
  }
}

// bias-gradient partials of a wave: L1:0,1 L2:2,3 L3:4 C1:5,6 C2:7,8 C3:9
__device__ __host__ constexpr int db_base(int l) {
  constexpr int b[NLAYER] = {0, 2, 4, 5, 7, 9};
  return b[l];
}

struct DFeatDst {
  void* p;
  int64_t stride;
  uint32_t* abs_part;  // optional [16 levels][kAbsWaves]: per-wave max |d feat| of each level (fp32 bit patterns), else nullptr
};
constexpr int kAbsWaves = 1024;  // kMaxBwdBlocks workgroups x 4 waves

// ------------------------------------------------------------------------------------------------
// weight-gradient flush: per-workgroup slabs + one reduce launch
// ------------------------------------------------------------------------------------------------
// Adding every wave's tiles to dparams with global float atomics would put 6.3 M atomics onto the same 14 227 addresses
// (1024 waves x 6 tiles x 1024 lanes-registers): measured 0.20 ms of a 0.61 ms kernel, independent of N.  Instead every wave stores its accumulator
// registers as they stand (coalesced 256-B rows) into its workgroup's slab of the caller's workspace, and
// mlp_dw_reduce_kernel sums the slabs in a fixed order (which also makes the MLP gradient run-to-run reproducible)
// and adds the result into dparams.
constexpr int kSlabRegs = NLAYER * 16 + 10;        // per wave: 6 tiles x 16 registers, then the 10 bias partials
constexpr int kSlabWave = kSlabRegs * 64;          // floats per wave
constexpr int kSlabWg = 4 * kSlabWave;             // floats per workgroup (four tile-owning waves)
static_assert(kSlabWg % 32 == 0, "reduce kernel takes 32 entries per block");
constexpr int kMaxBwdBlocks = 256;                 // one workgroup per CU
__device__ __host__ constexpr int64_t slab_offset_bytes(int64_t img_bytes) { return (img_bytes + 255) / 256 * 256; }

// Column order of the bf16 exchange image (xch_put): a lane parks its 8 fragment elements as ONE 16-byte chunk, so
// position c = 8h + j of a 16-feature group holds feature 8(j>>2) + 4h + (j&3) - bits 2 and 3 of the index swapped.
// The owners' tiles come out with rows and columns in image order; this (an involution) maps a tile index back.
__device__ __host__ constexpr int xch_feature(int pos, bool swapped) {
  return swapped ? ((pos & ~12) | ((pos & 4) << 1) | ((pos & 8) >> 1)) : pos;
}

// flat-parameter offset of slab entry e = (wave, register, lane), or -1 for a padding row/column of its tile
__device__ __forceinline__ int slab_entry_offset(int e, bool swapped) {
  const int lane = e & 63, r = (e >> 6) % kSlabRegs, wv = e / kSlabWave, h = lane >> 5;
  const int out = xch_feature(lane & 31, swapped);
  if (r < NLAYER * 16) {
    const int l = r >> 4, q = r & 15;
    const int nout = (l == L3 || l == C3) ? 1 : 2;
    const int tiles = ((l == L1) ? 1 : 2) * nout;
    const int tau = (tiles == 4) ? wv : (wv >> 1);
    return wlog_offset(l, 32 * (tau % nout) + out, 32 * (tau / nout) + xch_feature(acc_row(q, h), swapped));
  }
  const int i = r - NLAYER * 16;  // bias partial i: layer l, out tile m (db_base)
  int l = NLAYER - 1;
  while (db_base(l) > i) --l;
  return blog_offset(l, 32 * (i - db_base(l)) + out);
}

// Stage 1: 256 threads = 32 consecutive slab entries x 8 parts; part p sums slabs p, p+8, ... with four loads in
// flight, the eight partials are combined through LDS in a fixed order (a single thread per entry walking all 256
// slabs was latency-bound: 62 us) -> tot[entry].  Sixteen extra blocks reduce the per-wave feature-gradient maxima.
__global__ __launch_bounds__(256) void mlp_dw_reduce_kernel(const float* __restrict__ slabs, int nblocks, float* __restrict__ tot,
                                                            const uint32_t* __restrict__ abs_part, float* __restrict__ absmax_out,
                                                            bool swapped) {
  __shared__ float part[8][32];
  if (blockIdx.x >= kSlabWg / 32) {  // 16 extra blocks, one per level: max |d feat| over the waves' partials -> absmax_out[level]
    __shared__ uint32_t wmax[4];
    const int level = blockIdx.x - kSlabWg / 32;
    uint32_t m = 0;
    for (int w = threadIdx.x; w < nblocks * 4; w += 256) m = max(m, abs_part[(size_t)level * kAbsWaves + w]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) absmax_out[level] = __uint_as_float(max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
    return;
  }
  const int el = threadIdx.x & 31, p = threadIdx.x >> 5;
  const float* src = slabs + blockIdx.x * 32 + el;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (slab_entry_offset(blockIdx.x * 32 + el, swapped) >= 0) {  // padding entries (over half of the slab) are never read
    int b = p;
    for (; b + 24 < nblocks; b += 32) {
      a0 += src[(size_t)b * kSlabWg];
      a1 += src[(size_t)(b + 8) * kSlabWg];
      a2 += src[(size_t)(b + 16) * kSlabWg];
      a3 += src[(size_t)(b + 24) * kSlabWg];
    }
    for (; b < nblocks; b += 8) a0 += src[(size_t)b * kSlabWg];
  }
  part[p][el] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (p != 0) return;
  tot[blockIdx.x * 32 + el] = ((part[0][el] + part[1][el]) + (part[2][el] + part[3][el])) + ((part[4][el] + part[5][el]) + (part[6][el] + part[7][el]));
}

// Stage 2: slab entries -> parameters.  Several entries can stand for the same parameter - the two waves that share
// a tile of a two-tile layer, and the 4 waves x 2 lane halves that hold partial sums of one bias row - so ONE of them
// (the "leader": the even wave / wave 0's lower half) adds up its siblings in a fixed order and the others return:
// every parameter has exactly one writer and one summation order, i.e. the MLP gradient is bitwise reproducible.
// (hbr_mlp_render_bwd: one more block adds up the waves' squared-error partials, lane-strided then by DPP - a fixed order -
// and WRITES loss = se_scale * sum.)
__global__ __launch_bounds__(256) void mlp_dw_finalize_kernel(const float* __restrict__ tot, float* __restrict__ dparams, bool swapped, bool overwrite,
                                                              const float* __restrict__ se_part, int n_se, float se_scale, float* __restrict__ loss_out) {
  if (blockIdx.x == (kSlabWg + 255) / 256) {  // the extra block of a render launch
    if (threadIdx.x >= 64) return;
    float sum = 0.f;
    for (int i = threadIdx.x; i < n_se; i += 64) sum += se_part[i];
    sum = wave_reduce(sum, [](float a, float b) { return a + b; });
    if (threadIdx.x == 0) *loss_out = se_scale * sum;
    return;
  }
  const int e = blockIdx.x * 256 + threadIdx.x;  // (wave, register, lane) of the slab layout
  if (e >= kSlabWg) return;
  const int lane = e & 63, r = (e >> 6) % kSlabRegs, wv = e / kSlabWave, h = lane >> 5;
  const int off = slab_entry_offset(e, swapped);
  if (off < 0) return;
  float v = tot[e];
  if (r < NLAYER * 16) {
    const int l = r >> 4;
    const int tiles = ((l == L1) ? 1 : 2) * ((l == L3 || l == C3) ? 1 : 2);
    if (tiles != 4) {
      if (wv & 1) return;
      v += tot[e + kSlabWave];
    }
  } else {
    if (wv != 0 || h != 0) return;
    v += tot[e + 32];
#pragma unroll
    for (int k = 1; k < 4; ++k) v += tot[e + k * kSlabWave] + tot[e + k * kSlabWave + 32];
  }
  dparams[off] = overwrite ? v : dparams[off] + v;  // the only writer of this address
}

// ------------------------------------------------------------------------------------------------
// backward kernel, single pass: the four waves of a workgroup SHARE the weight-gradient accumulators
// ------------------------------------------------------------------------------------------------
// One wave cannot hold all 18 dW tiles (288 registers) next to its working set.  Here every wave runs the whole chain
// for its own 32-point tile, but owns only ONE dW tile per
// layer (<= 6 tiles = 96 accumulator registers): it parks a layer's X and dZ fragments in an LDS exchange slot
// (xch_put), meets the workgroup at a barrier, runs the next layer's dense while the owners' transposing reads are in
// flight, and then accumulates ITS tile of that layer over the fragments of all four waves (Owner; K = 128 points
// per round) with the dense's ReLU-mask epilogue in the shadow of those MFMAs.
// bf16: the slot is a [point][feature] image written straight from the orientation-1 fragments and read back with
// the transposing LDS read; f32: the fragments are transposed with identity MFMAs first (no 32-bit transposing read).  Layers with four dW tiles give one tile to every wave;
// layers with two give each tile to a pair of waves that split the four sources.  The exchange buffer is double-buffered, so one
// barrier per layer suffices: a wave can only overwrite buffer b two layers later, after the next barrier, which
// every wave reaches only after finishing its reads of b.  No LDS atomics, no second recompute of the forward.
// bf16 exchanges through a [point][feature] LDS image read back with ds_read_b64_tr_b16; f32 (no 32-bit transposing
// read) through fragments transposed by identity MFMAs
template <class P>
constexpr bool kXchImage = (P::ELEMS == 8);
template <class P>
constexpr bool kXchSwapped = kXchImage<P>;  // the image's columns are in xch_feature order (16-byte writes)
template <class P>
struct Xch {
  static constexpr int FRAG_B = (int)sizeof(typename P::frag) * 64;
  static constexpr int SLOT_FRAGS = 4 * P::S32;  // XT tiles 0,1 then dZT tiles 0,1, S32 k-steps (of points) each
  static constexpr int SLOT_B = SLOT_FRAGS * FRAG_B;
  static constexpr int BUF_B = 4 * SLOT_B;       // four waves
  static constexpr int BYTES = 2 * BUF_B;
};

// bf16 exchange image: a wave's slot is [32 points][128 features x 2 B] (X in chunks 0..7, dZ in chunks 8..15 of a
// 256-B row), written straight from the orientation-1 fragments and read back by the dW owners with the transposing
// LDS read (ds_read_b64_tr_b16), which yields the k = point fragments the wgrad MFMA needs - no identity-MFMA
// transposes.  16-B chunks are XOR-swizzled by the row (cdna_hip_programming.md T10, image (b)): the transposed reads
// and the 16-byte writes (columns in xch_feature order) are conflict-free.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int xch_off(int row, int chunk) {
  return 256 * row + 16 * (chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}
__device__ __forceinline__ bf16x8 lds_tr_frag(const char* lo, const char* hi) {
  const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)lo);
  const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)hi);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Which dW^T tile [in tile n][out tile m] of a layer a wave owns, and over which source waves.  Four-tile layers:
// wave w owns tile w over all four sources.  Two-tile layers: waves {0,1} own tile 0, waves {2,3} tile 1, each over
// half of the sources - so every wave runs the same instruction stream (no branch around the accumulator, which
// lets it stay in the AGPRs) and the partial tiles simply add up in the final flush.
template <int NIN, int NOUT>
struct Own {
  static constexpr int tiles = NIN * NOUT;
  static constexpr int NSRC = (tiles == 4) ? 4 : 2;
  int n, m, src0;
  __device__ __forceinline__ explicit Own(int wv) {
    const int tau = (tiles == 4) ? wv : (wv >> 1);
    src0 = (tiles == 4) ? 0 : 2 * (wv & 1);
    n = tau / NOUT;
    m = tau % NOUT;
  }
};

// ---- first half of a layer's exchange: park this wave's X and dZ fragments in its slot of buffer `buf`
template <class P, int NIN, int NOUT, int NKX, int NKZ>
__device__ __forceinline__ void xch_put(char* xch, int buf, int lane, int wv, const typename P::frag (&x)[NKX],
                                        const typename P::frag (&dz)[NKZ], float* colsum) {
  using X = Xch<P>;
  if constexpr (kXchImage<P>) {
    static_assert(X::SLOT_B == 32 * 256, "slot = 32 point rows of 256 B");
    const int pt = lane & 31, h = lane >> 5;
    const int sw16 = 16 * (((pt & 3) << 2) | ((pt >> 2) & 3));
    // fragment (tile ks/2, step ks%2) = the 16 features 32n+16s..: this lane's 8 elements (features 4h.., 8+4h..) go
    // to chunk h of the pair as ONE 16-byte write (columns in xch_feature order); the 8 lanes a write cycle serves hit
    // 8 different chunks.  (Two 8-byte writes with the columns in feature order cost the same: a store is priced by
    // its source dwords.)
    char* mine = xch + buf * X::BUF_B + wv * X::SLOT_B + 256 * pt;
#pragma unroll
    for (int ks = 0; ks < NKX; ++ks)
      *(u32x4*)(mine + ((16 * (4 * (ks / 2) + 2 * (ks % 2) + h)) ^ sw16)) = __builtin_bit_cast(u32x4, x[ks]);
#pragma unroll
    for (int ks = 0; ks < NKZ; ++ks)
      *(u32x4*)(mine + ((16 * (8 + 4 * (ks / 2) + 2 * (ks % 2) + h)) ^ sw16)) = __builtin_bit_cast(u32x4, dz[ks]);
  } else {
    typename P::frag xt[NIN][P::S32], zt[NOUT][P::S32];
    transpose_frags<P, NIN, NKX, false>(x, lane, xt);
    transpose_frags<P, NOUT, NKZ, true>(dz, lane, zt, colsum);
    char* mine = xch + buf * X::BUF_B + wv * X::SLOT_B + lane * (int)sizeof(typename P::frag);
#pragma unroll
    for (int n = 0; n < NIN; ++n)
#pragma unroll
      for (int s = 0; s < P::S32; ++s) *(typename P::frag*)(mine + (n * P::S32 + s) * X::FRAG_B) = xt[n][s];
#pragma unroll
    for (int m = 0; m < NOUT; ++m)
#pragma unroll
      for (int s = 0; s < P::S32; ++s) *(typename P::frag*)(mine + ((2 + m) * P::S32 + s) * X::FRAG_B) = zt[m][s];
  }
}

// bf16 bias gradients: ONE accumulator tile for all six layers.  Layer l's sums (over the points, of the out tile this
// wave owns) live in row acc_row(l, 0) of it, i.e. in register l of the lanes 0..31 (column = out feature).
struct BiasAcc {
  f32x16 tile;
  uint32_t onehot[NLAYER];  // per layer: a bf16 pair of ones in the lanes whose MFMA row is that layer's, else 0
  __device__ __forceinline__ void init(int lane) {
#pragma unroll
    for (int q = 0; q < 16; ++q) tile[q] = 0.f;
#pragma unroll
    for (int l = 0; l < NLAYER; ++l) {
      onehot[l] = ((lane & 31) == acc_row(l, 0)) ? 0x3f803f80u : 0u;
      asm volatile("" : "+v"(onehot[l]));  // computed once, long before the asm MFMAs that read it
    }
  }
};

// ---- second half of a layer's exchange
// The owner side, bf16.  begin(): meet the workgroup, then request the first two source waves'
// fragments.  run(): accumulate MY dW^T tile over all source waves (and, by one more MFMA per fragment against a
// one-hot operand, the bias gradients), with the caller's epilogue words slotted behind the MFMAs.  Whatever the
// caller issues between the two (the next dense's MFMAs) runs while the transposing reads are in flight.
template <int NIN, int NOUT>
struct Owner {
  using X = Xch<PBf16>;
  using O = Own<NIN, NOUT>;
  static constexpr int NSRC = O::NSRC;
  static constexpr bool kHalves = NSRC == 4;  // four sources: two at a time (32 instead of 64 fragment registers)
  const char *a0, *a1, *b0, *b1;
  bf16x8 fa[NSRC][2], fb[NSRC][2];
  int n, m;

  __device__ __forceinline__ void request(int w) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {  // points 16s + 8h + j of source wave src0 + w
      const int o = w * X::SLOT_B + s * 16 * 256;
      fa[w][s] = lds_tr_frag(a0 + o, a1 + o);
      fb[w][s] = lds_tr_frag(b0 + o, b1 + o);
    }
  }
  __device__ __forceinline__ void begin(const char* xch, int buf, int lane, int wv, PhaseClock& pc, int ph) {
    pc.mark(ph);      // put (+ dense when it is issued first)
    __syncthreads();
    pc.mark(ph + 1);  // waiting for the slowest wave
    const O own(wv);
    n = own.n; m = own.m;
    // lane (group g of its half, q, p) supplies row q, columns 4p..4p+3 of its group's 4-point x 16-feature block
    const int h = lane >> 5, g = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
    const char* src = xch + buf * X::BUF_B + own.src0 * X::SLOT_B + 8 * (p & 1);
    a0 = src + xch_off(8 * h + q, 4 * n + 2 * g + (p >> 1));
    a1 = src + xch_off(8 * h + 4 + q, 4 * n + 2 * g + (p >> 1));
    b0 = src + xch_off(8 * h + q, 8 + 4 * m + 2 * g + (p >> 1));
    b1 = src + xch_off(8 * h + 4 + q, 8 + 4 * m + 2 * g + (p >> 1));
    // the reads are issued two sources ahead of their MFMAs; each source's fragments are then claimed in turn, so the
    // waits are lgkmcnt(remaining) rather than full drains (hipcc otherwise sinks every read next to its MFMA)
#pragma unroll
    for (int w = 0; w < (kHalves ? 2 : NSRC); ++w) request(w);
  }
  // which sources' dZ this wave sums into the bias gradient: every (out tile, source) pair is summed by exactly one
  // of the waves that read it - the owners of (n, m) for n = 0, 1 split the source list by n
  __device__ __forceinline__ bool sums_bias(int w) const {
    if constexpr (O::tiles == 4) return (w < 2) == (n == 0);
    else if constexpr (NIN == 2) return w == n;  // NOUT == 1: the n = 0 and n = 1 owners read the same two sources
    else return true;                            // NIN == 1: one owner per (m, source pair)
  }
  template <int WORDS, class Epi>
  __device__ __forceinline__ void run(int& buf, f32x16& acc, BiasAcc& ba, int layer, PhaseClock& pc, int ph, Epi epi) {
    const u32x4 ohw = {ba.onehot[layer], ba.onehot[layer], ba.onehot[layer], ba.onehot[layer]};
    const bf16x8 onehot = __builtin_bit_cast(bf16x8, ohw);
#pragma unroll
    for (int w = 0; w < NSRC; ++w) {
      asm volatile("" : "+v"(fa[w][0]), "+v"(fa[w][1]), "+v"(fb[w][0]), "+v"(fb[w][1]));
      if (w == 0) pc.mark_fine(32 + 5 * layer + 4);  // (+ dense and) the first source's fragments have arrived
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        // the claim above may have moved fragments with v_mov: pad the first MFMA behind it (VALU write -> MFMA read)
        if (s == 0) PBf16::mfma_acc_lds<(WORDS > 0), true>(fa[w][s], fb[w][s], acc);
        else PBf16::mfma_acc_lds<(WORDS > 0), false>(fa[w][s], fb[w][s], acc);
        constexpr int NM = 2 * NSRC;
        const int i = 2 * w + s;
#pragma unroll
        for (int k = (i * WORDS) / NM; k < ((i + 1) * WORDS) / NM; ++k) epi(k);
      }
      // bias gradient of out tile m = sum over points of dZ: one more MFMA per fragment, against an operand that is 1 in
      // row `layer`'s slot of the shared bias tile and 0 elsewhere (8 issue cycles; four v_dot2c per fragment cost ~40)
      if (sums_bias(w)) {
        PBf16::mfma_acc_lds<(WORDS > 0), true>(onehot, fb[w][0], ba.tile);  // `onehot` is assembled by v_mov
        PBf16::mfma_acc_lds<(WORDS > 0), false>(onehot, fb[w][1], ba.tile);
      }
      if constexpr (kHalves) {
        if (w + 2 < NSRC) request(w + 2);
      }
      pc.mark_fine(32 + 5 * layer + w);  // source w: two owner MFMAs + its epilogue words (+ two bias MFMAs)
    }
    buf ^= 1;
    pc.mark(ph + 2);  // (dense +) owner MFMAs with the epilogue in their shadow
  }
};

struct NoEpi {
  __device__ __forceinline__ void operator()(int) const {}
};
// `epi(k)`, k = 0..WORDS-1 (bf16 only): the caller's VALU epilogue of the dense it has just issued, cut into words
template <class P, int NIN, int NOUT, int WORDS = 0, class Epi = NoEpi>
__device__ __forceinline__ void xch_take(char* xch, int& buf, int lane, int wv, f32x16& acc, BiasAcc& ba, int layer, PhaseClock& pc,
                                         int ph, Epi epi = Epi()) {
  static_assert(kXchImage<P> || WORDS == 0, "only the bf16 path takes an epilogue");
  if constexpr (kXchImage<P>) {
    Owner<NIN, NOUT> ow;
    ow.begin(xch, buf, lane, wv, pc, ph);
    ow.template run<WORDS>(buf, acc, ba, layer, pc, ph, epi);
  } else {
    using X = Xch<P>;
    using O = Own<NIN, NOUT>;
    pc.mark(ph);
    __syncthreads();
    pc.mark(ph + 1);
    const O own(wv);
    // f32: fragments already transposed by xch_put, one source at a time (16 + 16 registers in flight)
    const char* base = xch + buf * X::BUF_B + own.src0 * X::SLOT_B + lane * (int)sizeof(typename P::frag);
    (void)ba; (void)layer;
    typename P::frag fa[P::S32], fb[P::S32];
#pragma unroll
    for (int w = 0; w < O::NSRC; ++w) {
#pragma unroll
      for (int s = 0; s < P::S32; ++s) {
        fa[s] = *(const typename P::frag*)(base + w * X::SLOT_B + (own.n * P::S32 + s) * X::FRAG_B);
        fb[s] = *(const typename P::frag*)(base + w * X::SLOT_B + ((2 + own.m) * P::S32 + s) * X::FRAG_B);
      }
#pragma unroll
      for (int s = 0; s < P::S32; ++s) P::mfma_acc(fa[s], fb[s], acc);
    }
    buf ^= 1;
    pc.mark(ph + 2);
  }
}

// One backward layer step after the exchange writes (xch_put): the NEXT layer's dense `a = W^T dz`, its ReLU mask
// (-> `out`, the next dZ) and the owner half of THIS layer's exchange.
// bf16 order: barrier, request the owners' transposing reads, THEN the dense MFMAs - on weight fragments the caller
// requested before the exchange writes (`wq`; LDS operations complete in order, so they land first) - and then the
// owner MFMAs with the mask epilogue word by word in their shadow.  The dense covers the latency of the transposing
// reads and no longer waits for weight reads queued behind the 8-16 exchange writes (382.6 vs 391.4 us with the dense
// in front of the barrier).   f32: dense, mask, take.
constexpr int kMaxDenseFrags = 8;
template <class P, int NOUT, int NK>
__device__ __forceinline__ void dense_request(const char* img, int fbase, int lofs, typename P::frag (&wq)[kMaxDenseFrags]) {
  if constexpr (kXchImage<P>) {
    static_assert(NOUT * NK <= kMaxDenseFrags, "weight fragments of one backward dense");
#pragma unroll
    for (int i = 0; i < NOUT * NK; ++i) wq[i] = ldw<P>(img, fbase + i, lofs);
  }
}
template <class P, int NT, int NK, int NIN, int NOUT>
__device__ __forceinline__ void dense_mask_take(const char* img, int fbase, int lofs, const typename P::frag (&wq)[kMaxDenseFrags],
                                                const typename P::frag (&dz)[NK], const typename P::frag (&h)[NT * P::S32],
                                                typename P::frag (&out)[NT * P::S32], char* xch, int& buf, int lane, int wv,
                                                f32x16& acc, BiasAcc& ba, int layer, PhaseClock& pc, int ph) {
  f32x16 a[NT];
  if constexpr (kXchImage<P>) {
    constexpr int WORDS = NT * P::S32 * 4;
    uint32_t ow[WORDS];
    uint32_t ones = 0x00010001u;
    asm volatile("" : "+v"(ones));
    Owner<NIN, NOUT> own;
    own.begin(xch, buf, lane, wv, pc, ph);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int q = 0; q < 16; ++q) a[t][q] = 0.f;
#pragma unroll
      for (int k = 0; k < NK; ++k) a[t] = P::mfma(wq[t * NK + k], dz[k], a[t]);
    }
    own.template run<WORDS>(buf, acc, ba, layer, pc, ph, [&](int k) {
      const int f = k >> 2, t = f / P::S32, s = f % P::S32, j = k & 3;
      const uint32_t g = pack_bf16x2(a[t][8 * s + 2 * j], a[t][8 * s + 2 * j + 1]);
      ow[k] = pk_keep_where_nonzero_ordered(g, __builtin_bit_cast(u32x4, h[f])[j], ones);
    });
#pragma unroll
    for (int f = 0; f < NT * P::S32; ++f) {
      out[f] = PBf16::from_words(ow[4 * f], ow[4 * f + 1], ow[4 * f + 2], ow[4 * f + 3]);
    }
    asm_valu_pad(out);  // the words came out of asm statements: the next dense's MFMAs read them as operands
  } else {
    dense<P, NT, NK, 0>(img, fbase, nullptr, lane, lofs, dz, a);
    mask_frags<P, NT>(a, h, out);
    xch_take<P, NIN, NOUT>(xch, buf, lane, wv, acc, ba, layer, pc, ph);
  }
}

// RENDER (round 4; hbr_mlp_render_bwd): the training step's K3 + K5 + K4 in ONE kernel.  The backward recomputes the forward of
// its tile anyway, and with S in {32, 64, 128} samples per ray the four waves of a workgroup hold 4 x 32 consecutive
// points = whole rays - so the raw (rgb, sigma) the recompute ends in are activated (test_hash.py:62,67), parked in
// LDS (2 KiB), and after ONE barrier every wave composites ITS ray from them (composite_ray.h: the operations of
// composite_loss_vec_kernel in its order, the same bits on the same inputs), forms dL/dC = k (C - gt) and the gradient
// of its own 32 samples, which is the `d out` the rest of the kernel reads from memory otherwise.  The forward launch
// (0.065 ms), the compositing launches (0.020 ms) and 2 x 33 MB of [N,4] traffic are gone; the wave that holds a ray's
// first samples adds its squared error to a per-wave partial (fixed order; summed by mlp_dw_finalize_kernel).
struct RenderArgs {
  const float* t;         // [S] shared depths
  const float* dir_norm;  // [R] or nullptr (= 1)
  const float* gt;        // [R,3]
  float k;                // gscale * 4 / (3 R)
  float* se_part;         // [gridDim.x * 4] per-wave sums of (C - gt)^2, written
  float* Cr;              // optional [R,3] out
  uint32_t S;
};
constexpr int kRenderLdsBytes = 4 * 32 * 16;  // (r, g, b, sigma) of the workgroup's 128 points

template <class P, int LAYOUT, int DT, bool WLDS, bool RENDER = false>
__global__ __launch_bounds__(256) void mlp_bwd_fused_kernel(const char* __restrict__ gimg, FeatSrc fs, PeSrc ps,
                                                            const float* __restrict__ dout, DFeatDst dfd,
                                                            float* __restrict__ slabs, RenderArgs ra) {
  using T = Tab<P>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xch = smem;
  char* limg = smem + Xch<P>::BYTES;
  float4* rays = (float4*)(smem + Xch<P>::BYTES + (WLDS ? (T::IMG_BYTES + 15) / 16 * 16 : 0));  // RENDER only
  if (WLDS) stage_image(limg, gimg, T::IMG_BYTES);
  // the exchange image is read in whole 32-feature tiles even where a layer writes fewer: start from zeros, not from
  // whatever bit patterns the LDS held
  for (int i = threadIdx.x * 16; i < Xch<P>::BYTES; i += 256 * 16) *(uint4*)(xch + i) = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  const char* img = WLDS ? (const char*)limg : gimg;
  const float* bias = (const float*)(img + T::BIAS_OFF_ALL);
  // the wave index as a SCALAR: which dW tile a wave owns, which sources it sums and where it reads them from are then
  // wave-uniform by construction - scalar branches (no EXEC masking around the owner MFMAs) and scalar address terms
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = lane >> 5;
  const uint32_t ntiles = (fs.N + 31) / 32;
  const uint32_t stride = gridDim.x * 4;
  const uint32_t rounds = (ntiles + stride - 1) / stride;  // every wave runs every round: the barriers are workgroup-wide

  f32x16 acc[NLAYER];  // my dW^T tile of each layer
#pragma unroll
  for (int l = 0; l < NLAYER; ++l)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[l][q] = 0.f;
  // running max |d feat| (as stored, fp32 bit patterns) of the lane's 8 levels 4g + 2h + {0,1}: the scatter kernel's
  // fixed-point scale comes from it, so it does not have to re-read the 131 MB it is about to consume
  uint32_t amax[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) amax[i] = 0u;
  float bsum[10];  // f32: the lane's bias-gradient partials (xch_put); bf16: filled from `ba` at the end
#pragma unroll
  for (int i = 0; i < 10; ++i) bsum[i] = 0.f;
  BiasAcc ba;
  ba.init(lane);
  int buf = 0;

  // RENDER: a lane's samples of ITS ray are always 64 c + lane: the depth differences are loop constants
  float tdl[2] = {0.f, 0.f};
  float se_acc = 0.f;
  const int wpr = RENDER ? (int)(ra.S >> 5) : 1;   // waves per ray: 1, 2 or 4
  const int wr = wv & (wpr - 1);                    // this wave's position in its ray
  if constexpr (RENDER) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const uint32_t sidx = 64u * c + (uint32_t)lane;
      if (sidx + 1u < ra.S) tdl[c] = __fsub_rn(ra.t[sidx + 1u], ra.t[sidx]);
    }
  }
  TileIn nxt;
  TileCursor ahead;  // the tile being prefetched
  ahead.start((blockIdx.x * 4 + wv) * 32 + (lane & 31), stride * 32, ps.group);
  load_tile_in<LAYOUT, DT, true>(fs, ps, dout, ahead, ahead.n < fs.N, h, nxt);
  PhaseClock pc;
  pc.start();
#ifndef HBR_K4_SPREAD
#define HBR_K4_SPREAD 1  // 0: all twelve loads of the next tile at the top of the round (A/B builds)
#endif
  for (uint32_t r = 0; r < rounds; ++r) {
    const uint32_t tile = blockIdx.x * 4 + wv + r * stride;
    const uint32_t n = tile * 32 + (lane & 31);
    const bool valid = tile < ntiles && n < fs.N;
    const TileIn cur = nxt;
    ahead.advance();  // n < N implies its tile exists; N < 2^31 and at most one round past the end: no wrap-around
    // Round 3: on the fast path the next tile's loads go out two at a time between the exchange steps (load_tile_part):
    // 0.3957 -> 0.3876 ms stand-alone, same bits.  (Deferring the eight d feat stores the same way was measured too:
    // 0.3937 ms - the carried words cost more than the stores' issue stalls; not kept.)
    const bool spread = HBR_K4_SPREAD && LAYOUT == HBR_LAYOUT_PLANAR && fs.addr32;  // uniform
    const bool nvalid = ahead.n < fs.N;
    if (!spread) load_tile_in<LAYOUT, DT, true>(fs, ps, dout, ahead, nvalid, h, nxt);
    else load_tile_part<DT, 0>(fs, ps, dout, ahead, nvalid, h, nxt);
#define HBR_LOAD_PART(PART) if (spread) load_tile_part<DT, PART>(fs, ps, dout, ahead, nvalid, h, nxt)
    pc.mark(0);  // next tile's loads issued
    Saved<P> sv;
    if constexpr (P::ELEMS == 8) forward_tile_prefetched<DT>(img, cur, lane, sv, pc);
    else forward_tile<P, DT, true>(img, bias, cur, lane, sv, pc);
    const int lofs = opaque_lane_offset<P>(lane);
    float4 dO = cur.dO;  // zero on invalid lanes => every dZ of such a point is zero
    if constexpr (RENDER) {
      // my 32 points' activated outputs -> LDS (what hbr_mlp_fwd writes to out[N,4]: ELU on rgb, leaky ReLU on sigma)
      if (h == 0) rays[wv * 32 + lane] = valid ? make_float4(elu1(sv.raw[0]), elu1(sv.raw[1]), elu1(sv.raw[2]), lrelu(sv.s0)) : make_float4(0.f, 0.f, 0.f, 0.f);
      __syncthreads();  // (the next round's writes come after this round's six exchange barriers: one buffer suffices)
      const uint32_t ray = (tile - (uint32_t)wr) / (uint32_t)wpr;  // tiles of a ray are consecutive and aligned: N = R * S
      const bool rvalid = tile < ntiles;                            // uniform: a ray is in the launch whole or not at all
      dO = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rvalid) {
        const float dn = ra.dir_norm ? ra.dir_norm[ray] : 1.f;
        const float4* row = rays + (wv - wr) * 32;  // the ray's S samples
        auto composite = [&](auto nch_tag) {  // one 64-lane chunk for S <= 64 (train_hash2.py:29's default), two for S = 128
          constexpr int NCH = decltype(nch_tag)::value;
          RayComposite<NCH> rc;
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const uint32_t sidx = 64u * c + (uint32_t)lane;
            rc.v[c] = sidx < ra.S ? row[sidx] : make_float4(0.f, 0.f, 0.f, 0.f);
            rc.dl[c] = __fmul_rn(tdl[c], dn);  // helper.py:67,71 (0 for the last sample and beyond)
          }
          rc.run((int)ra.S, lane, ra.gt[ray * 3 + 0], ra.gt[ray * 3 + 1], ra.gt[ray * 3 + 2], ra.k);
          if (wr == 0) {
            se_acc += rc.se;
            if (ra.Cr && lane == 0) { ra.Cr[ray * 3 + 0] = rc.c0; ra.Cr[ray * 3 + 1] = rc.c1; ra.Cr[ray * 3 + 2] = rc.c2; }
          }
          // my samples are 32 wr + (lane & 31): chunk wr >> 1, lanes 32 (wr & 1) + ...
          float4 mine = rc.d[0];
          if constexpr (NCH == 2) mine = (wr & 2) ? rc.d[1] : rc.d[0];
          if (wr & 1) {
            const int src = (lane & 31) + 32;
            mine = make_float4(__shfl(mine.x, src), __shfl(mine.y, src), __shfl(mine.z, src), __shfl(mine.w, src));
          }
          if (h == 0 && valid) dO = mine;
        };
        if (ra.S > 64u) composite(std::integral_constant<int, 2>{});
        else composite(std::integral_constant<int, 1>{});
      }
    }

    // ---- C3
    typename P::frag dz3[P::S8];
    {
      f32x16 a;
#pragma unroll
      for (int q = 0; q < 16; ++q) a[q] = 0.f;
      // (RENDER, measured and not kept: the ELU derivative as ELU(x) + 1 from the activations the render block just computed -
      // three values kept live across the compositing cost more than the three expf they save: 0.4424 vs 0.4373 ms stand-alone)
      a[0] = dO.x * (sv.raw[0] > 0.f ? 1.f : expf(sv.raw[0]));
      a[1] = dO.y * (sv.raw[1] > 0.f ? 1.f : expf(sv.raw[1]));
      a[2] = dO.z * (sv.raw[2] > 0.f ? 1.f : expf(sv.raw[2]));
#pragma unroll
      for (int s = 0; s < P::S8; ++s) dz3[s] = P::from_acc(a, s);
    }
    pc.mark(1);  // dZ of the output layer (expf)
    typename P::frag wq[kMaxDenseFrags];  // the next dense's weight fragments, requested ahead of the exchange writes
    dense_request<P, 2, P::S8>(img, T::b_base(C3), lofs, wq);
    xch_put<P, 2, 1, 2 * P::S32, P::S8>(xch, buf, lane, wv, sv.c2, dz3, bsum + db_base(C3));
    // ---- C2
    typename P::frag dzc2[2 * P::S32];
    dense_mask_take<P, 2, P::S8, 2, 1>(img, T::b_base(C3), lofs, wq, dz3, sv.c2, dzc2, xch, buf, lane, wv, acc[C3], ba, C3, pc, 2);
    HBR_LOAD_PART(1);
    dense_request<P, 2, 2 * P::S32>(img, T::b_base(C2), lofs, wq);
    xch_put<P, 2, 2, 2 * P::S32, 2 * P::S32>(xch, buf, lane, wv, sv.c1, dzc2, bsum + db_base(C2));
    // ---- C1
    typename P::frag dzc1[2 * P::S32];
    dense_mask_take<P, 2, 2 * P::S32, 2, 2>(img, T::b_base(C2), lofs, wq, dzc2, sv.c1, dzc1, xch, buf, lane, wv, acc[C2], ba, C2, pc, 5);
    HBR_LOAD_PART(2);
    xch_put<P, 2, 2, P::S32 + P::S8, 2 * P::S32>(xch, buf, lane, wv, sv.cin, dzc1, bsum + db_base(C1));  // take: after the next layer's dense
    // ---- L3: ds rows 1..15 = d cin slots 1..15 ; row 0 = d sigma * lrelu'(s0)
    typename P::frag dz_s[P::S16];
    {
      f32x16 a[1];
      dense<P, 1, 2 * P::S32, 0>(img, T::b_base(C1), nullptr, lane, lofs, dzc1, a);
      if (h == 0) a[0][0] = dO.w * (sv.s0 > 0.f ? 1.f : 0.01f);
#pragma unroll
      for (int s = 0; s < P::S16; ++s) dz_s[s] = P::from_acc(a[0], s);
    }
    xch_take<P, 2, 2>(xch, buf, lane, wv, acc[C1], ba, C1, pc, 8);
    HBR_LOAD_PART(3);
    dense_request<P, 2, P::S16>(img, T::b_base(L3), lofs, wq);
    xch_put<P, 2, 1, 2 * P::S32, P::S16>(xch, buf, lane, wv, sv.h2, dz_s, bsum + db_base(L3));
    // ---- L2
    typename P::frag dz2[2 * P::S32];
    dense_mask_take<P, 2, P::S16, 2, 1>(img, T::b_base(L3), lofs, wq, dz_s, sv.h2, dz2, xch, buf, lane, wv, acc[L3], ba, L3, pc, 11);
    HBR_LOAD_PART(4);
    dense_request<P, 2, 2 * P::S32>(img, T::b_base(L2), lofs, wq);
    xch_put<P, 2, 2, 2 * P::S32, 2 * P::S32>(xch, buf, lane, wv, sv.h1, dz2, bsum + db_base(L2));
    // ---- L1
    typename P::frag dz1[2 * P::S32];
    dense_mask_take<P, 2, 2 * P::S32, 2, 2>(img, T::b_base(L2), lofs, wq, dz2, sv.h1, dz1, xch, buf, lane, wv, acc[L2], ba, L2, pc, 14);
    HBR_LOAD_PART(5);
    xch_put<P, 1, 2, P::S32, 2 * P::S32>(xch, buf, lane, wv, sv.x0, dz1, bsum + db_base(L1));  // take: after the next layer's dense
    // ---- d feat
    if (dfd.p) {
      f32x16 a[1];
      dense<P, 1, 2 * P::S32, 0>(img, T::b_base(L1), nullptr, lane, lofs, dz1, a);
      if (valid) {
        constexpr int ES = DT == HBR_F32 ? 8 : 4;
        const uint32_t off = planar_off<ES>(fs.N, n, h);  // the offset this tile's features were loaded from
        // words to store per level pair: fp32 (v0,v1),(v2,v3); bf16 the packed pairs.  The maxima are of the values AS
        // STORED; bf16 keeps them per 16-bit half (both halves of a word belong to one level).
        auto emit = [&](auto fast_tag) {
          constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float v0 = a[0][4 * g], v1 = a[0][4 * g + 1], v2 = a[0][4 * g + 2], v3 = a[0][4 * g + 3];
            const int lvl = 4 * g + 2 * h;
            char* lb0 = (char*)dfd.p + planar_level_base<ES>(fs.N, 2 * g);  // wave-uniform bases of levels 4g + 2h, + 1
            char* lb1 = (char*)dfd.p + planar_level_base<ES>(fs.N, 2 * g + 1);
            if (DT == HBR_F32) {
              amax[2 * g] = max(amax[2 * g], max(__float_as_uint(v0) & 0x7fffffffu, __float_as_uint(v1) & 0x7fffffffu));
              amax[2 * g + 1] = max(amax[2 * g + 1], max(__float_as_uint(v2) & 0x7fffffffu, __float_as_uint(v3) & 0x7fffffffu));
              if (FAST) {
                *(float2*)(lb0 + off) = make_float2(v0, v1);
                *(float2*)(lb1 + off) = make_float2(v2, v3);
              } else if (LAYOUT == HBR_LAYOUT_PLANAR) {
                ((float2*)dfd.p)[(size_t)lvl * fs.N + n] = make_float2(v0, v1);
                ((float2*)dfd.p)[(size_t)(lvl + 1) * fs.N + n] = make_float2(v2, v3);
              } else {
                *(float4*)((float*)dfd.p + (size_t)n * dfd.stride + 2 * lvl) = make_float4(v0, v1, v2, v3);
              }
            } else {
              const uint32_t p01 = pack_bf16x2(v0, v1), p23 = pack_bf16x2(v2, v3);
              amax[2 * g] = pk_max_u16(amax[2 * g], p01 & 0x7fff7fffu);
              amax[2 * g + 1] = pk_max_u16(amax[2 * g + 1], p23 & 0x7fff7fffu);
              if (FAST) {
                *(uint32_t*)(lb0 + off) = p01;
                *(uint32_t*)(lb1 + off) = p23;
              } else if (LAYOUT == HBR_LAYOUT_PLANAR) {
                ((uint32_t*)dfd.p)[(size_t)lvl * fs.N + n] = p01;
                ((uint32_t*)dfd.p)[(size_t)(lvl + 1) * fs.N + n] = p23;
              } else {
                *(uint2*)((uint16_t*)dfd.p + (size_t)n * dfd.stride + 2 * lvl) = make_uint2(p01, p23);
              }
            }
          }
        };
        if (LAYOUT == HBR_LAYOUT_PLANAR && fs.addr32) emit(std::true_type{});
        else emit(std::false_type{});
      }
    }
    xch_take<P, 1, 2>(xch, buf, lane, wv, acc[L1], ba, L1, pc, 17);
#undef HBR_LOAD_PART
  }

  pc.finish();
  // ---- flush: every wave parks its registers in the workgroup's slab (plain coalesced stores; mlp_dw_reduce_kernel
  // turns the slabs into parameter gradients)
  asm volatile("s_nop 15" ::: "memory");  // last asm MFMA's D -> first VALU reader
  if constexpr (kXchImage<P>) {  // layer l, out tile own.m: register l of the lower lane half
    auto park = [&](int l, int m, int nout) {
      const float v = h ? 0.f : ba.tile[l];
      bsum[db_base(l)] = (m == 0) ? v : 0.f;
      if (nout == 2) bsum[db_base(l) + 1] = (m == 1) ? v : 0.f;
    };
    park(L1, Own<1, 2>(wv).m, 2); park(L2, Own<2, 2>(wv).m, 2); park(L3, Own<2, 1>(wv).m, 1);
    park(C1, Own<2, 2>(wv).m, 2); park(C2, Own<2, 2>(wv).m, 2); park(C3, Own<2, 1>(wv).m, 1);
  }
  if constexpr (RENDER) {
    if (lane == 0) ra.se_part[blockIdx.x * 4 + wv] = se_acc;
  }
  float* mine = slabs + ((size_t)blockIdx.x * 4 + wv) * kSlabWave + lane;
#pragma unroll
  for (int l = 0; l < NLAYER; ++l)
#pragma unroll
    for (int q = 0; q < 16; ++q) mine[(l * 16 + q) * 64] = acc[l][q];
#pragma unroll
  for (int i = 0; i < 10; ++i) mine[(NLAYER * 16 + i) * 64] = bsum[i];
  if (dfd.abs_part) {  // the wave's maxima: reduce over the 32 lanes of each half, lanes 0 and 32 store their 8 levels
    if (DT == HBR_BF16) {  // two bf16 maxima per register -> the fp32 bit pattern of the larger
#pragma unroll
      for (int i = 0; i < 8; ++i) amax[i] = max(amax[i] << 16, amax[i] & 0xffff0000u);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) amax[i] = max(amax[i], (uint32_t)__shfl_xor((int)amax[i], o));
    }
    if ((lane & 31) == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) dfd.abs_part[(size_t)(4 * (i >> 1) + 2 * h + (i & 1)) * kAbsWaves + blockIdx.x * 4 + wv] = amax[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// view-direction encoding (a7): out[row, c*2nf + k] = sin(2*x_c*k), out[row, c*2nf + nf + k] = cos(2*x_c*k)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dir_encode_body(const float* __restrict__ x, int64_t rows, int d, int nf, float* __restrict__ out,
                                                int64_t block, int64_t nblocks) {
  const int64_t total = rows * d * nf;
  for (int64_t e = block * 256 + threadIdx.x; e < total; e += nblocks * 256) {
    const int k = (int)(e % nf);
    const int64_t rc = e / nf;
    const int c = (int)(rc % d);
    const int64_t row = rc / d;
    const float ang = __fmul_rn(__fmul_rn(2.f, x[row * d + c]), (float)k);  // encoder.py:27: (2*x)*k
    float* o = out + row * (int64_t)(d * 2 * nf) + c * 2 * nf;
    o[k] = sinf(ang);
    o[nf + k] = cosf(ang);
  }
}
__global__ __launch_bounds__(256) void dir_encode_kernel(const float* __restrict__ x, int64_t rows, int d, int nf,
                                                         float* __restrict__ out) {
  dir_encode_body(x, rows, d, nf, out, (int64_t)blockIdx.x, (int64_t)gridDim.x);
}

// What a render call does before the encoder runs, in ONE launch (three ~4.5 us launches otherwise): the weight-fragment
// image (blocks [0, pack_blocks)), the per-ray direction encoding (the next dir_blocks) and the depths t[S] (the rest).
template <class P>
__global__ __launch_bounds__(256) void prologue_kernel(StratArgs sa, const float* __restrict__ dirs, int64_t rows, float* __restrict__ pe,
                                                       const float* __restrict__ params, char* __restrict__ img, int pack_blocks,
                                                       int dir_blocks) {
  int b = (int)blockIdx.x;
  if (b < pack_blocks) {
    pack_body<P>(params, img, b, pack_blocks);
    return;
  }
  b -= pack_blocks;
  if (b < dir_blocks) {
    dir_encode_body(dirs, rows, 3, 4, pe, b, dir_blocks);
    return;
  }
  strat_sample_one(sa, (uint32_t)(b - dir_blocks) * 256u + threadIdx.x);
}

template <class P>
static int pack(const float* params, char* ws, hipStream_t st) {
  hipLaunchKernelGGL((pack_kernel<P>), dim3(64), dim3(256), 0, st, params, ws);
  return HBR_OK;
}

// raise the kernel's dynamic-LDS limit (above 64 KiB needs the attribute) and launch; a refused attribute is an error,
// not something to find out from a failed launch later
template <class K, class... Args>
static int launch_with_lds(K k, dim3 grid, dim3 block, int lds, hipStream_t st, Args... args) {
  if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return HBR_ELAUNCH;
  hipLaunchKernelGGL(k, grid, block, lds, st, args...);
  return HBR_OK;
}

template <class P, int LAYOUT>
static int launch_fwd(int dt, uint32_t blocks, hipStream_t st, const char* img, FeatSrc fs, PeSrc ps, float* out, const uint8_t* keep) {
  using T = Tab<P>;
  const int lds = T::BIAS_OFF_F + T::BIAS_BYTES;
  if (dt == HBR_F32) return launch_with_lds(mlp_fwd_kernel<P, LAYOUT, HBR_F32>, dim3(blocks), dim3(kFwdWaves * 64), lds, st, img, fs, ps, out, keep);
  return launch_with_lds(mlp_fwd_kernel<P, LAYOUT, HBR_BF16>, dim3(blocks), dim3(kFwdWaves * 64), lds, st, img, fs, ps, out, keep);
}

// render != nullptr (bf16, planar): the RENDER instantiation - d out is formed in the kernel, *loss_out is written
template <class P, int LAYOUT, int DT, bool WLDS>
static int launch_bwd_fused(uint32_t ntiles, hipStream_t st, const char* img, FeatSrc fs, PeSrc ps, const float* dout,
                            DFeatDst dfd, float* dparams, float* absmax_out, bool overwrite, const RenderArgs* render = nullptr,
                            float se_scale = 0.f, float* loss_out = nullptr) {
  using T = Tab<P>;
  const int lds = Xch<P>::BYTES + (WLDS ? (T::IMG_BYTES + 15) / 16 * 16 : 0) + (render ? kRenderLdsBytes : 0);
  uint32_t blocks = (ntiles + 3) / 4;
  if (blocks > kMaxBwdBlocks) blocks = kMaxBwdBlocks;  // one workgroup per CU; each sweeps its share of the tiles in rounds of four
  float* slabs = (float*)(const_cast<char*>(img) + slab_offset_bytes(T::IMG_BYTES));  // behind the fragment image in `ws`
  const bool want_abs = absmax_out && dfd.p;
  dfd.abs_part = want_abs ? (uint32_t*)(slabs + (size_t)kMaxBwdBlocks * kSlabWg) : nullptr;  // behind the slabs
  float* tot = slabs + (size_t)kMaxBwdBlocks * kSlabWg + 16 * (size_t)kAbsWaves;  // behind the maxima
  float* se_part = tot + kSlabWg;                                                   // behind the totals: one float per wave
  int rc;
  if constexpr (std::is_same<P, PBf16>::value && LAYOUT == HBR_LAYOUT_PLANAR && WLDS) {
    if (render) {
      RenderArgs ra = *render;
      ra.se_part = se_part;
      rc = launch_with_lds(mlp_bwd_fused_kernel<P, LAYOUT, DT, WLDS, true>, dim3(blocks), dim3(256), lds, st, img, fs, ps, dout, dfd, slabs, ra);
    } else {
      rc = launch_with_lds(mlp_bwd_fused_kernel<P, LAYOUT, DT, WLDS, false>, dim3(blocks), dim3(256), lds, st, img, fs, ps, dout, dfd, slabs, RenderArgs{});
    }
  } else {
    if (render) return HBR_EUNSUPPORTED;
    rc = launch_with_lds(mlp_bwd_fused_kernel<P, LAYOUT, DT, WLDS, false>, dim3(blocks), dim3(256), lds, st, img, fs, ps, dout, dfd, slabs, RenderArgs{});
  }
  if (rc) return rc;
  hipLaunchKernelGGL(mlp_dw_reduce_kernel, dim3(kSlabWg / 32 + (want_abs ? 16 : 0)), dim3(256), 0, st, (const float*)slabs, (int)blocks,
                     tot, (const uint32_t*)dfd.abs_part, absmax_out, kXchSwapped<P>);
  hipLaunchKernelGGL(mlp_dw_finalize_kernel, dim3((kSlabWg + 255) / 256 + (render ? 1 : 0)), dim3(256), 0, st, (const float*)tot, dparams,
                     kXchSwapped<P>, overwrite, (const float*)se_part, (int)blocks * 4, se_scale, loss_out);
  return HBR_OK;
}

// single pass, dW tiles shared by the four waves of a workgroup through an LDS fragment exchange
template <int LAYOUT, int DT>
static int launch_bwd(int precision, uint32_t ntiles, hipStream_t st, const char* img, FeatSrc fs, PeSrc ps, const float* dout,
                      DFeatDst dfd, float* dparams, float* absmax_out, bool overwrite) {
  if (precision == HBR_BF16) return launch_bwd_fused<PBf16, LAYOUT, DT, true>(ntiles, st, img, fs, ps, dout, dfd, dparams, absmax_out, overwrite);
  return launch_bwd_fused<PF32, LAYOUT, DT, false>(ntiles, st, img, fs, ps, dout, dfd, dparams, absmax_out, overwrite);
}

// HBR_MLP_ADDR64 (any value) in the environment sends every call down the 64-bit addressing path that sizes beyond
// kAddr32MaxN take - the tests use it to cover that path at small sizes
static uint32_t addr32_ok(int64_t N, int64_t group) {
  static const bool force64 = getenv("HBR_MLP_ADDR64") != nullptr;
  return (!force64 && N <= kAddr32MaxN && N / group <= kAddr32MaxRays) ? 1u : 0u;
}

static int check_common(const void* feat, int layout, int64_t stride, int dt, const float* pe, int64_t N, int64_t group,
                        const float* params, int precision, void* ws, int64_t ws_bytes) {
  if (!feat || !pe || !params || !ws || N < 0 || group < 1) return HBR_EINVAL;
  if (layout != HBR_LAYOUT_ROWS && layout != HBR_LAYOUT_PLANAR) return HBR_EINVAL;
  if (dt != HBR_F32 && dt != HBR_BF16) return HBR_EINVAL;
  if (precision != HBR_F32 && precision != HBR_BF16) return HBR_EINVAL;
  if (layout == HBR_LAYOUT_ROWS && (stride < 32 || (stride & 3))) return HBR_EINVAL;  // 16-byte row alignment
  if (N > 0x7fffffffLL) return HBR_EUNSUPPORTED;
  if (ws_bytes < hbr_mlp_workspace_bytes(precision)) return HBR_EWORKSPACE;
  if (((uintptr_t)ws | (uintptr_t)pe) & 15) return HBR_EINVAL;
  return HBR_OK;
}

}  // namespace mlp
}  // namespace hbr

using namespace hbr;
using namespace hbr::mlp;

#if HBR_K4_PROF
extern "C" int hbr_debug_k4_prof(unsigned long long* out64, int reset) {  // development builds only
  if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(k4_prof), sizeof(unsigned long long) * 64) != hipSuccess) return HBR_EINVAL;
  if (reset) {
    unsigned long long z[64] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(k4_prof), z, sizeof(z)) != hipSuccess) return HBR_EINVAL;
  }
  return HBR_OK;
}
#endif

extern "C" int64_t hbr_mlp_workspace_bytes(int precision) {
  // the MFMA-fragment image of the weights, then (backward only) one weight-gradient slab per workgroup
  const int64_t img = precision == HBR_BF16 ? Tab<PBf16>::IMG_BYTES : Tab<PF32>::IMG_BYTES;
  // ... the per-wave feature-gradient maxima, one slab of totals, and (hbr_mlp_render_bwd) one squared-error partial per wave
  return slab_offset_bytes(img) + ((int64_t)kMaxBwdBlocks * kSlabWg + 16 * (int64_t)kAbsWaves + kSlabWg + (int64_t)kMaxBwdBlocks * 4) * (int64_t)sizeof(float);
}

extern "C" int hbr_dir_encode(const float* x, int64_t rows, int d_model, int num_freq, float* out, void* stream) {
  if (!x || !out || rows < 0 || d_model < 1 || num_freq < 1) return HBR_EINVAL;
  if (rows == 0) return HBR_OK;
  int64_t total = rows * d_model * num_freq;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dir_encode_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, x, rows, d_model, num_freq, out);
  HBR_RETURN_IF_LAUNCH_FAILED();
  return HBR_OK;
}

extern "C" int hbr_render_prologue(float tn, float tf, int64_t S, const float* u, uint64_t seed, uint64_t offset, float* t,
                                   const float* rays_d, int64_t R, float* pe, const float* params, int precision, void* ws,
                                   int64_t ws_bytes, void* stream) {
  if (S < 0 || R < 0 || S > 0x7fffffffLL) return HBR_EINVAL;
  if ((pe != nullptr) != (rays_d != nullptr)) return HBR_EINVAL;
  if (params) {
    if (precision != HBR_F32 && precision != HBR_BF16) return HBR_EINVAL;
    if (!ws || ((uintptr_t)ws & 15)) return HBR_EINVAL;
    if (ws_bytes < hbr_mlp_workspace_bytes(precision)) return HBR_EWORKSPACE;
  }
  const int pack_blocks = params ? 64 : 0;
  int64_t dir_blocks = (pe && R > 0) ? (R * 12 + 255) / 256 : 0;
  if (dir_blocks > 4096) dir_blocks = 4096;
  const int64_t strat_blocks = (t && S > 0) ? (S + 255) / 256 : 0;
  const int64_t blocks = pack_blocks + dir_blocks + strat_blocks;
  if (blocks == 0) return HBR_OK;
  if (blocks > 0x7fffffffLL) return HBR_EUNSUPPORTED;
  const StratArgs sa{tn, tf, (uint32_t)(t ? S : 0), u, seed, offset, t};
  hipStream_t st = (hipStream_t)stream;
  if (precision == HBR_BF16)
    hipLaunchKernelGGL((prologue_kernel<PBf16>), dim3((uint32_t)blocks), dim3(256), 0, st, sa, rays_d, R, pe, params, (char*)ws, pack_blocks, (int)dir_blocks);
  else
    hipLaunchKernelGGL((prologue_kernel<PF32>), dim3((uint32_t)blocks), dim3(256), 0, st, sa, rays_d, R, pe, params, (char*)ws, pack_blocks, (int)dir_blocks);
  HBR_RETURN_IF_LAUNCH_FAILED();
  return HBR_OK;
}

extern "C" int hbr_mlp_fwd(const void* feat, int layout, int64_t feat_stride, int feat_dtype, const float* viewdirs_enc,
                           int64_t N, int64_t group, const float* params, int precision, float* out, const uint8_t* keep, void* ws,
                           int64_t ws_bytes, void* stream) {
  const bool image_ready = (precision & HBR_IMAGE_READY) != 0;  // `ws` holds the image of THESE params (hbr_render_prologue)
  precision &= ~HBR_IMAGE_READY;
  int rc = check_common(feat, layout, feat_stride, feat_dtype, viewdirs_enc, N, group, params, precision, ws, ws_bytes);
  if (rc) return rc;
  if (!out || ((uintptr_t)out & 15)) return HBR_EINVAL;
  if (N == 0) return HBR_OK;
  hipStream_t st = (hipStream_t)stream;
  FeatSrc fs{feat, feat_stride, (uint32_t)N, addr32_ok(N, group)};
  PeSrc ps{viewdirs_enc, (uint32_t)group};
  const uint32_t ntiles = (uint32_t)((N + 31) / 32);
  uint32_t blocks = (ntiles + kFwdWaves - 1) / kFwdWaves;
  if (blocks > 4096 / kFwdWaves) blocks = 4096 / kFwdWaves;
  char* img = (char*)ws;
  if (precision == HBR_BF16) {
    if (!image_ready) pack<PBf16>(params, img, st);
    if (layout == HBR_LAYOUT_PLANAR) rc = launch_fwd<PBf16, HBR_LAYOUT_PLANAR>(feat_dtype, blocks, st, img, fs, ps, out, keep);
    else rc = launch_fwd<PBf16, HBR_LAYOUT_ROWS>(feat_dtype, blocks, st, img, fs, ps, out, keep);
  } else {
    if (!image_ready) pack<PF32>(params, img, st);
    if (layout == HBR_LAYOUT_PLANAR) rc = launch_fwd<PF32, HBR_LAYOUT_PLANAR>(feat_dtype, blocks, st, img, fs, ps, out, keep);
    else rc = launch_fwd<PF32, HBR_LAYOUT_ROWS>(feat_dtype, blocks, st, img, fs, ps, out, keep);
  }
  if (rc) return rc;
  HBR_RETURN_IF_LAUNCH_FAILED();
  return HBR_OK;
}

extern "C" int hbr_mlp_bwd(const void* feat, int layout, int64_t feat_stride, int feat_dtype, const float* viewdirs_enc,
                           int64_t N, int64_t group, const float* params, int precision, const float* dout, void* dfeat,
                           float* dfeat_absmax, float* dparams, void* ws, int64_t ws_bytes, void* stream) {
  // HBR_IMAGE_READY: `ws` still holds the fragment image hbr_mlp_fwd / hbr_mlp_bwd built from THESE params at this
  // precision (a training step's forward, then its backward): skip the 5 us pack launch
  const bool image_ready = (precision & HBR_IMAGE_READY) != 0;
  const bool overwrite = (precision & HBR_OVERWRITE) != 0;  // dparams written instead of accumulated into
  precision &= ~(HBR_IMAGE_READY | HBR_OVERWRITE);
  int rc = check_common(feat, layout, feat_stride, feat_dtype, viewdirs_enc, N, group, params, precision, ws, ws_bytes);
  if (rc) return rc;
  if (!dout || !dparams || ((uintptr_t)dout & 15)) return HBR_EINVAL;
  if (N == 0) return overwrite ? HBR_EUNSUPPORTED : HBR_OK;  // nothing would write dparams
  hipStream_t st = (hipStream_t)stream;
  FeatSrc fs{feat, feat_stride, (uint32_t)N, addr32_ok(N, group)};
  PeSrc ps{viewdirs_enc, (uint32_t)group};
  DFeatDst dfd{dfeat, feat_stride, nullptr};
  const uint32_t ntiles = (uint32_t)((N + 31) / 32);
  char* img = (char*)ws;
  if (!image_ready) {
    if (precision == HBR_BF16) pack<PBf16>(params, img, st);
    else pack<PF32>(params, img, st);
  }
  if (layout == HBR_LAYOUT_PLANAR) {
    if (feat_dtype == HBR_F32) rc = launch_bwd<HBR_LAYOUT_PLANAR, HBR_F32>(precision, ntiles, st, img, fs, ps, dout, dfd, dparams, dfeat_absmax, overwrite);
    else rc = launch_bwd<HBR_LAYOUT_PLANAR, HBR_BF16>(precision, ntiles, st, img, fs, ps, dout, dfd, dparams, dfeat_absmax, overwrite);
  } else {
    if (feat_dtype == HBR_F32) rc = launch_bwd<HBR_LAYOUT_ROWS, HBR_F32>(precision, ntiles, st, img, fs, ps, dout, dfd, dparams, dfeat_absmax, overwrite);
    else rc = launch_bwd<HBR_LAYOUT_ROWS, HBR_BF16>(precision, ntiles, st, img, fs, ps, dout, dfd, dparams, dfeat_absmax, overwrite);
  }
  if (rc) return rc;
  HBR_RETURN_IF_LAUNCH_FAILED();
  return HBR_OK;
}

// K3 + K5 + K4 of a training step in one call (mlp_bwd_fused_kernel<..., RENDER>): see include/hbr_hip.h
extern "C" int hbr_mlp_render_bwd(const void* feat, int layout, int64_t feat_stride, int feat_dtype, const float* viewdirs_enc,
                                  int64_t R, int64_t S, const float* params, int precision, const float* t, const float* dir_norm,
                                  const float* gt, float gscale, float* loss_out, float* Cr, void* dfeat, float* dfeat_absmax,
                                  float* dparams, void* ws, int64_t ws_bytes, void* stream) {
  const bool image_ready = (precision & HBR_IMAGE_READY) != 0;
  const bool overwrite = (precision & HBR_OVERWRITE) != 0;
  precision &= ~(HBR_IMAGE_READY | HBR_OVERWRITE);
  if (R < 0 || S < 1) return HBR_EINVAL;
  const int64_t N = R * S;
  int rc = check_common(feat, layout, feat_stride, feat_dtype, viewdirs_enc, N, S, params, precision, ws, ws_bytes);
  if (rc) return rc;
  if (!t || !gt || !loss_out || !dparams) return HBR_EINVAL;
  // whole rays per workgroup round (4 waves x 32 points), bf16 MFMA, planar features: anything else takes the three separate calls
  if (precision != HBR_BF16 || layout != HBR_LAYOUT_PLANAR || (S != 32 && S != 64 && S != 128) || R < 1) return HBR_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  FeatSrc fs{feat, feat_stride, (uint32_t)N, addr32_ok(N, S)};
  PeSrc ps{viewdirs_enc, (uint32_t)S};
  DFeatDst dfd{dfeat, feat_stride, nullptr};
  const uint32_t ntiles = (uint32_t)(N / 32);
  char* img = (char*)ws;
  if (!image_ready) pack<PBf16>(params, img, st);
  RenderArgs ra{t, dir_norm, gt, gscale * 4.f * (1.0f / (float)(R * 3)), nullptr, Cr, (uint32_t)S};
  const float se_scale = 2.f * (1.0f / (float)(R * 3));
  if (feat_dtype == HBR_F32)
    rc = launch_bwd_fused<PBf16, HBR_LAYOUT_PLANAR, HBR_F32, true>(ntiles, st, img, fs, ps, nullptr, dfd, dparams, dfeat_absmax, overwrite, &ra, se_scale, loss_out);
  else
    rc = launch_bwd_fused<PBf16, HBR_LAYOUT_PLANAR, HBR_BF16, true>(ntiles, st, img, fs, ps, nullptr, dfd, dparams, dfeat_absmax, overwrite, &ra, se_scale, loss_out);
  if (rc) return rc;
  HBR_RETURN_IF_LAUNCH_FAILED();
  return HBR_OK;
}
