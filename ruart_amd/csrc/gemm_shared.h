// Helpers shared by the encoder GEMM kernels (gemm.hip, gemm_corr.hip): LDS tile addressing, the GELU epilogue, raw barrier.
#pragma once
#include "common.h"

// byte offset of 16-byte chunk `c` (0..7) of row `r` inside a [rows][64] bf16 tile
__device__ __forceinline__ int lds_off(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }

// GELU for the 16-bit epilogues:  gelu(x) = x * Phi(x) = x / (1 + exp(-x q(x^2))),  x q(x^2) = logit(Phi(x)), q = degree-4
// polynomial in x^2 fitted (weighted minimax, tools/fit_gelu.py) on |x| <= 7.  q stays positive and grows beyond the fitted
// range, so the logistic saturates to exactly 0 / 1 for large |x| (and through inf) without a clamp.  Max abs error 3.4e-6 in
// fp32 evaluation - at or below the half-ulp of an f16 output wherever |gelu| > 0.01, 70x below it at |gelu| ~ 0.5.
// The epilogue is VALU-issue bound (PMC + instruction count: ~12 lane-passes per element here vs ~19 for the A&S erf form; the
// packed v_pk_*_f32 forms take two passes, so they save instructions, not cycles).  Coefficients carry the -log2(e) of the exp2.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_pk(f32x2_t x) {
  const f32x2_t x2 = x * x;
  f32x2_t p = x2 * -3.228988589e-06f + 8.823813550e-05f;
  p = p * x2 + 3.602745419e-04f;
  p = p * x2 + -1.052266881e-01f;
  p = p * x2 + -2.302045345e+00f;
  p = p * x;
  f32x2_t d;
  d.x = 1.0f + __builtin_amdgcn_exp2f(p.x);
  d.y = 1.0f + __builtin_amdgcn_exp2f(p.y);
  f32x2_t r;
  r.x = __builtin_amdgcn_rcpf(d.x);
  r.y = __builtin_amdgcn_rcpf(d.y);
  return x * r;
}
__device__ __forceinline__ f32x4_t gelu4(f32x4_t v) {
  const f32x2_t lo = gelu_pk((f32x2_t){v[0], v[1]}), hi = gelu_pk((f32x2_t){v[2], v[3]});
  return (f32x4_t){lo.x, lo.y, hi.x, hi.y};
}

// Forward value and derivative of the same GELU in one go (the backward epilogue of the trainable encoder's output-dense dX product):
// Phi(x) = 1 / (1 + exp2(p(x))) as above, g = x Phi, dg/dx = Phi + x phi(x) with phi(x) = exp(-x^2 / 2) / sqrt(2 pi).
__device__ __forceinline__ void gelu_fwd_bwd_pk(f32x2_t x, f32x2_t& g, f32x2_t& d) {
  const f32x2_t x2 = x * x;
  f32x2_t p = x2 * -3.228988589e-06f + 8.823813550e-05f;
  p = p * x2 + 3.602745419e-04f;
  p = p * x2 + -1.052266881e-01f;
  p = p * x2 + -2.302045345e+00f;
  p = p * x;
  f32x2_t den, phi;
  den.x = 1.0f + __builtin_amdgcn_exp2f(p.x);
  den.y = 1.0f + __builtin_amdgcn_exp2f(p.y);
  phi.x = __builtin_amdgcn_exp2f(x2.x * -0.72134752044f);
  phi.y = __builtin_amdgcn_exp2f(x2.y * -0.72134752044f);
  f32x2_t cdf;
  cdf.x = __builtin_amdgcn_rcpf(den.x);
  cdf.y = __builtin_amdgcn_rcpf(den.y);
  g = x * cdf;
  d = cdf + x * phi * 0.3989422804f;
}

// GROUP_M of the tile walk (workgroup ids walk groups of GROUP_M row panels, column-major inside a group) as a rule in the problem's TILE
// counts and K - not in the benchmark's row count.  What the sweeps show (tools/gemm_corr_order_sweep.py; round 3 at 42 880 rows x bert-base,
// profiles/r03_gemm_order_sweep_time.log; round 4 at 121 344 rows x bert-large, profiles/r04_gemm_order_sweep.log): a wide output (>= 8
// column tiles) wants tall groups - 16 while an operand panel is short (K < 1024: 393 KB per 256 rows in the fp16c form), 8 for the longer
// panels of bert-large (QKV 1 251 us at 8, 1 291 at 16; FF1 1 831 / 1 906); a narrow output with a long reduction (K >= 2048) a shorter one
// (6); everything else 8; short matrices (< 96 row tiles: the (64, 512) halves of the north-star shape) 4 in the plain 16-bit kernel.
// `precise` = the fp16c kernel (twice the K-tiles per tile).  Differences between neighbouring settings are 1-5 %, the walk is not what
// bounds these kernels.
static inline int ruart_tile_group_m(int row_tiles, int col_tiles, int K, bool precise) {
  int g = 8;
  if (precise) {
    if (col_tiles >= 8) g = K >= 1024 ? 8 : 16;
    else if (K >= 2048) g = 6;
  }
  if (row_tiles < 96 && !precise) g = 4;
  return g;
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA of 16 bytes per lane in the instruction's saddr form: global address = wave-uniform 64-bit base (an SGPR pair) + this lane's
// 32-bit BYTE offset; LDS address = wave-uniform byte address (through M0) + lane * 16.  The builtin takes one 64-bit pointer per lane
// and hipcc then keeps every staged address as a VGPR pair plus two v_lshl_add_u64 per load in the K loop (12+ VGPRs and ~16 64-bit
// VALU adds per K-tile in gemm_16_nt_256p8); this form needs one VGPR per operand piece and scalar pointer arithmetic.
// Not tracked by the compiler's own s_waitcnt insertion: every kernel that uses it counts vmcnt by hand (they already do).  M0 is on
// the clobber list, so a compiler-initialised M0 (builtin LDS-DMA, s_movrel indexing, LDS-direct loads, sendmsg) added to a kernel
// that includes this header is re-established after the statement instead of silently reading ours.
// Precondition: `base_uniform` and `lds_dst_uniform` are wave-uniform (the "s" constraint would otherwise insert a readfirstlane
// and quietly use lane 0's value for every lane).
// RUART_DMA_BUILTIN=1 (diagnostic builds) goes back to the builtin.
#ifndef RUART_DMA_BUILTIN
#define RUART_DMA_BUILTIN 0
#endif
__device__ __forceinline__ void dma16(const void* base_uniform, unsigned lane_byte_off, char* lds_dst_uniform) {
#if RUART_DMA_BUILTIN
  __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(base_uniform) + lane_byte_off), (lptr_t)lds_dst_uniform, 16, 0, 0);
#else
  const unsigned lds_addr = (unsigned)(size_t)(lptr_t)lds_dst_uniform;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(lane_byte_off), "s"(base_uniform) : "memory", "m0");
#endif
}

#define RUART_BAR()                          \
  do {                                       \
    asm volatile("" ::: "memory");           \
    __builtin_amdgcn_s_barrier();            \
    asm volatile("" ::: "memory");           \
  } while (0)

// ---- LayerNorm folded into the projections around it (ruart_bert_forward_folded, bert_forward.hip) --------------------------------
// The encoder's LayerNorm pass (Models/Bert/modeling.py:164-168) reads a fp32 row and writes it back three times over (fp32 + f16 +
// two e4m3 bytes): 12 bytes per element of pure traffic between two GEMMs.  In the folded form nobody materialises a normalised row:
//   * the PRODUCER of the pre-LayerNorm row y (attention-output / output dense, EPI 3) writes y itself - fp32 and in the split operand
//     form - plus, per row and per 256-column tile, the partial (sum, sum of squares) of the row: part[row][4][2] (slot = tile);
//   * the CONSUMER projection (QKV / intermediate dense, FOLD) runs on y with weights W' = W diag(gamma) 2^-s prepared once, and its
//     epilogue finishes the normalisation per output element:
//         LN(y) W^T + b  =  rstd_r 2^s (y W'^T - mu_r c) + d,     c_j = sum_i W'_ji,   d_j = b_j + sum_i beta_i W_ji
//     (mu_r, rstd_r from the row's partials; c and d are vectors prepared with the weights);
//   * whoever needs the normalised row as a RESIDUAL (the next EPI 3 product) or as a layer output (the pooling kernel) applies
//     (y - mu) rstd gamma + beta to the fp32 y it reads anyway.
// Row statistics are one-pass sums in fp32 over 768-1024 elements (var = E[y^2] - mu^2, clamped at 0): |mu| << sigma on these rows.
struct CorrFold {
  const float* in_part;    // FOLD: partials of the A rows, [M][4][2], the first in_np slots used
  const float* colc;       // FOLD: c [N]
  int in_np;
  float wscale;            // FOLD: 2^s
  const float* rs_part;    // EPI 3: partials of the residual rows [M][4][2] (rs_np used); NULL = the residual rows are taken as they are
  const float* rs_g;       // EPI 3: gamma, beta [N] of the residual's LayerNorm
  const float* rs_b;
  int rs_np;
  float* out_part;         // EPI 3: [M][4][2] partials of the rows written (slot = column tile, N / 256 <= 4 of them)
  void* C16;               // EPI 3: the written rows in the split operand form (C16 f16 [M][ldc], C8 e4m3 [M][2 ldc])
  float inv_h;             // 1 / (row length the partials cover)
  float eps;
};
constexpr int kFoldSlots = 4;                // partial slots per row (32 bytes: rows of partials can be gathered 16 bytes at a time); N <= 1024
constexpr int kFoldStatsOff = 72 * 1024;    // LDS: (mu, rstd) of the tile's 256 rows, beyond the epilogue's staging image (68 KB)
constexpr int kFoldPartOff = 76 * 1024;     // LDS: [4 column groups][256 rows] (sum, sumsq) of an EPI 3 tile

// (mu, rstd) of one row from its partials
__device__ __forceinline__ float2 fold_row_stat_of(const float* __restrict__ part, int np, int row, float inv_h, float eps) {
  const float* p = part + (size_t)row * (2 * kFoldSlots);
  float s = 0.f, q = 0.f;
  for (int k = 0; k < np; ++k) {
    s += p[2 * k];
    q += p[2 * k + 1];
  }
  const float mu = s * inv_h;
  const float var = fmaxf(q * inv_h - mu * mu, 0.f);
  return make_float2(mu, 1.0f / sqrtf(var + eps));
}
// (mu, rstd) of the tile's rows m0 .. m0 + 255 from their partials -> LDS; every thread of the workgroup calls it
__device__ __forceinline__ void fold_row_stats(char* smem, const float* __restrict__ part, int np, int m0, float inv_h, float eps) {
  if (threadIdx.x < 256) reinterpret_cast<float2*>(smem + kFoldStatsOff)[threadIdx.x] = fold_row_stat_of(part, np, m0 + threadIdx.x, inv_h, eps);
  __syncthreads();
}
// sum over the 16 lanes of a DPP row (the lanes that share a tile row in the epilogue); every lane gets the total
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));    // quad_perm 1,0,3,2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));    // quad_perm 2,3,0,1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));   // row_mirror
  return v;
}
