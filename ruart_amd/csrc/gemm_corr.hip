// Dense projections of the encoder in the "f16 + fp8 correction" precision mode (RUART_DT_F16C, common.h):
//
//   C[M,N] = epilogue( A . W^T + bias ) [+ residual]      with      A . W^T  ~=  A16 . W16^T  +  2^-18 * A8 . W8^T
//
// Replaces the same nn.Linear sites as gemm.hip (Models/Bert/modeling.py:225-227, 261, 287-288, 300) when the answer scores have
// to stay within 1e-3 of the fp32 reference on every output: a plain 16-bit product carries ~2^-12 relative operand rounding, the
// SDNet trunk amplifies that to ~1e-2 on the worst of 6 464 probabilities.  Here the rounding residuals of
// both operands are multiplied too - not with two more 16-bit products (the split-bf16 "x3" form, 3x the MFMA time) but on the
// block-scaled fp8 matrix instruction of CDNA4, which runs at TWICE the f16 rate:
//     A16 = f16(A)                        W16 = f16(W)                                      k = 0 .. K-1      v_mfma_f32_16x16x32_f16
//     A8  = [ e4m3((A-A16) 2^11) | e4m3(A) ]         W8 = [ e4m3(W16 2^7) | e4m3((W-W16) 2^18) ]     k' = 0 .. 2K-1
//                                                                                 v_mfma_scale_f32_16x16x128_f8f6f4, scale 2^-18
// A correction term needs ~5 significant bits (it is 2^-11 of the product), which e4m3 has; the dropped lo.lo term is 2^-22.
// One byte of A8 / W8 per K element and per half, so a row of A8 is exactly as long as a row of A16 (2K bytes): the fp8 phase is
// the SAME loop over 128-byte-per-row K-tiles - same LDS image, swizzle, staging and fragment reads as the f16 phase - with the
// base pointers switched and one 16x16x128 MFMA (32 cycles) where the f16 phase issues two 16x16x32 (16 cycles each).  The
// accumulators are shared: the scaled MFMA adds 2^-18 * (a8 . w8) straight into the fp32 sums.
//
// Schedule: gemm_16_nt_256p8's (256x256 tile, 8 waves, four phases per K-tile, LDS-DMA prefetch in flight across raw barriers,
// counted vmcnt, staggered wave groups) over 2 K/64 K-tiles.  Epilogues: fp32 out (QKV), fp32 out + fp32 residual (attention
// output / FFN output dense: the residual stream stays fp32), GELU + split out (FFN intermediate: f16 + the two fp8 halves).
#include "common.h"
#include <type_traits>
#include "ruart_hip.h"
#include "gemm_shared.h"

typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x8_t __attribute__((ext_vector_type(8)));

// operand format code of the scaled MFMA: 0 = e4m3 (production).  Diagnostic builds (tools/build_variant.sh fp6 -DRUART_C8_FMT=2)
// make the instruction read the same registers as e2m3 fp6 - wrong numbers, but the 4x-rate timing of an fp6 correction phase.
#ifndef RUART_C8_FMT
#define RUART_C8_FMT 0
#endif

// Counted waits of the K loop: 0 = one wait per K-tile for the whole next tile (three half-tiles in flight; the product since round 1),
// 1 = three waits per K-tile, each one phase ahead of the first read of what it covers (five half-tiles in flight, every DMA piece has
// >= 5 phases to land instead of >= 3).  Round 6 measured 1 against 0: race screen clean, time equal (layer 1 246 vs 1 258 us, step 21.98
// vs 21.98 ms, profiles/r06_waits_ab.log) - the loop does not wait for its prefetch; kept as a diagnostic build.
#ifndef RUART_P8_WAITS
#define RUART_P8_WAITS 0
#endif
#define CBM 256
#define CBN 256
#define CBKB 128   // bytes of one row of one K-tile (64 f16 or 128 fp8)

// GELU of this mode: x Phi(x) with Phi(x) = 0.5 erfc(-x / sqrt 2) taken as  r = exp2(P7(|x|)),  Phi = x < 0 ? r : 1 - r,  where P7 is a
// degree-7 fit of log2(0.5 erfc(a / sqrt 2)) on a in [0, 7.07] (weighted by erfc, so the ABSOLUTE error of Phi is what is minimised:
// 7e-8; beyond 7.07 Phi is 0 or 1 to 1e-12 and |x| is clamped).  |GELU error| <= 5.0e-7 over [-12, 12], rms 7.6e-8 on [-4, 4] -
// the level of the Abramowitz-Stegun 7.1.26 erf this replaces (4.6e-7 / 1.2e-7) with ONE transcendental per element instead of two
// (v_rcp + v_exp run at a quarter of the fma rate).  Measured on the FFN intermediate dense (43 008 x 3072 x 768, diagnostic builds
// -DRUART_ABL_NOGELU / -DRUART_ABL_NOFP8): 403 us per launch, 375 without the GELU, 361 without the fp8 companions, 320 without
// both - the A-S form was 405, so the epilogue is not bound by these instructions alone.  The logistic-polynomial form of the plain
// 16-bit epilogue (3.4e-6) would be the largest error of the whole layer here.
__device__ __forceinline__ float gelu_erfc7(float x) {
  const float a = fminf(fabsf(x), 7.0710678f);
  float p = fmaf(8.470145706e-06f, a, -5.390352871e-05f);
  p = fmaf(p, a, -4.212367312e-04f);
  p = fmaf(p, a, 7.389273853e-03f);
  p = fmaf(p, a, -5.269111326e-02f);
  p = fmaf(p, a, -4.591531477e-01f);
  p = fmaf(p, a, -1.151110943e+00f);
  p = fmaf(p, a, -9.999998964e-01f);
  const float r = __builtin_amdgcn_exp2f(p);
  // x Phi(x) = max(x, 0) - |x| r  (x >= 0: x - x r; x < 0: x r): two instructions instead of compare + select + subtract + multiply
  return fmaf(-fabsf(x), r, fmaxf(x, 0.f));
}
__device__ __forceinline__ f32x4_t gelu4_as(f32x4_t v) {
  f32x4_t r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = gelu_erfc7(v[i]);
  return r;
}

// Epilogue of one 256 x 256 tile through LDS, eight rows per pass (as gemm_16_nt_256p8): bias, then per EPI the residual / GELU + split
// stores.  Shared by the GEMM kernel and by the fix-up kernel of its split tail tiles.  `smem`: >= 8 x 32 x 272 bytes, no longer read as
// operand tiles by any wave.
// Cache policy of the epilogue's streams (experiments, round 5): bit 0 the fp32 QKV rows (394 MB per launch, read once by the attention
// kernel), bit 1 the fp32 pre-LayerNorm rows of the two N = 768 products, bit 3 their residual loads - non-temporal when set.
#ifndef RUART_NT_EPI
#define RUART_NT_EPI 0
#endif

// (CorrFold, fold_row_stats, row16_sum: gemm_shared.h - the plain 16-bit kernel folds its LayerNorms the same way since round 6)
// EPI: 0 fp32 out; 1 fp32 out + fp32 residual; 2 GELU, split out; 3 fp32 out + (LayerNorm of the) residual, split out, row partials.
// FOLD (EPI 0 / 2): the A rows are pre-LayerNorm rows, see CorrFold.
// WN: wave columns of the tile (4: the 256 x 256 tile of eight waves; 2: the 256 x 128 tile of four, gemm_16c_nt_256x128d - EPI 0 / 2 only).
template <int EPI, bool FOLD = false, int WN = 4>
__device__ __forceinline__ void corr_epilogue(f32x4_t (&acc)[4][8], char* smem, int m0, int n0, const float* __restrict__ bias,
                                              const float* __restrict__ R, int ldr, void* __restrict__ C, int ldc,
                                              unsigned char* __restrict__ C8, int N, int hh0, int hh1, const CorrFold& f,
                                              const float2* pre_stats = nullptr) {
  static_assert(WN == 4 || (WN == 2 && EPI != 3 && EPI != 1), "the 256 x 128 tile carries the QKV / intermediate epilogues only");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = WN == 4 ? wave >> 2 : wave >> 1, wn = wave & (WN - 1), fr = lane & 15, fq = lane >> 4;
  constexpr int ERS = 272;
  char* my = smem + wave * (32 * ERS);
  const int rrow = lane >> 4, rcol = (lane & 15) * 4;
  f32x4_t bv = {0.f, 0.f, 0.f, 0.f};
  const int ncol = n0 + wn * 64 + rcol;
  if (bias) bv = *reinterpret_cast<const f32x4_t*>(bias + ncol);
  f32x4_t cv = {0.f, 0.f, 0.f, 0.f}, gv = {1.f, 1.f, 1.f, 1.f}, ev = {0.f, 0.f, 0.f, 0.f};
  bool res_ln = false;
  float wsc = 1.f;
  // (mu, rstd) of the tile's rows -> LDS: from the partials, or - `pre_stats` - the pair thread i < 256 already holds for row m0 + i (the
  // 256 x 128 kernel finishes them ahead of its K loop, where they cost two registers instead of a spill of the accumulators; in the 256 x 256
  // kernel the same move measured equal - QKV 282 against 276 us, FF1 408 / 405 - and is not made)
  auto put_stats = [&](const float* part, int np) {
    if (pre_stats) {
      if (threadIdx.x < 256) reinterpret_cast<float2*>(smem + kFoldStatsOff)[threadIdx.x] = *pre_stats;
      __syncthreads();
    } else {
      fold_row_stats(smem, part, np, m0, f.inv_h, f.eps);
    }
  };
  if constexpr (FOLD) {
    put_stats(f.in_part, f.in_np);
    cv = *reinterpret_cast<const f32x4_t*>(f.colc + ncol);
    wsc = f.wscale;
  }
  if constexpr (EPI == 3) {
    res_ln = f.rs_part != nullptr;
    if (res_ln) {
      put_stats(f.rs_part, f.rs_np);
      gv = *reinterpret_cast<const f32x4_t*>(f.rs_g + ncol);
      ev = *reinterpret_cast<const f32x4_t*>(f.rs_b + ncol);
    }
  }
  const float2* rstat = reinterpret_cast<const float2*>(smem + kFoldStatsOff) + wm * 128 + rrow;
  float2* rpart = reinterpret_cast<float2*>(smem + kFoldPartOff) + wn * 256 + wm * 128 + rrow;
#pragma unroll
  for (int hh = 0; hh < 4; ++hh) {
    if (hh < hh0 || hh >= hh1) continue;           // (the fix-up kernel runs one 32-row pass per workgroup)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        *reinterpret_cast<f32x4_t*>(my + (j * 16 + fr) * ERS + (i * 16 + fq * 4) * 4) = acc[i][hh * 2 + j];
    f32x4_t v[8], res[8];
    const int mrow = m0 + wm * 128 + hh * 32 + rrow;
    // Row addresses as uniform base + 32-bit byte offset (the launcher checks that every operand spans < 4 GB): the stores take the
    // saddr form and a row costs one v_add_u32 instead of a 64-bit multiply-add chain (~2 VALU instructions per element of a VALU-bound epilogue)
    const unsigned e_c = (unsigned)(mrow * ldc + ncol), e_r = (unsigned)(mrow * ldr + ncol);            // element offsets of row mrow
    const unsigned st_c = 4u * (unsigned)ldc, st_r = 4u * (unsigned)ldr;                                   // ... and of four rows on
    auto at = [](const void* base, unsigned byte_off) { return reinterpret_cast<const char*>(base) + byte_off; };
    auto atw = [](void* base, unsigned byte_off) { return reinterpret_cast<char*>(base) + byte_off; };
    if (EPI == 1 || EPI == 3) {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr)
        res[rr] = (RUART_NT_EPI & 8) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(at(R, (e_r + rr * st_r) * 4u)))
                                     : *reinterpret_cast<const f32x4_t*>(at(R, (e_r + rr * st_r) * 4u));
    }
    if constexpr (FOLD) {
      // v = rstd 2^s (acc - mu c) + d   (d arrives as `bias`)
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const float2 st = rstat[hh * 32 + rr * 4];
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(my + (rr * 4 + rrow) * ERS + rcol * 4);
        const float sc = st.y * wsc;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[rr][r] = fmaf(sc, fmaf(-st.x, cv[r], a[r]), bv[r]);
      }
    } else {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) v[rr] = *reinterpret_cast<const f32x4_t*>(my + (rr * 4 + rrow) * ERS + rcol * 4) + bv;
    }
    if constexpr (EPI == 3) {
      // y = acc + bias + residual (the residual rows normalised on the way in when they are pre-LayerNorm rows); out: y fp32, y split,
      // the row's (sum, sumsq) over this wave's 64 columns -> LDS
      f16_t* C16 = reinterpret_cast<f16_t*>(f.C16);
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const unsigned ec = e_c + rr * st_c;                 // element offset of this row's four columns in C / C16
        if (res_ln) {
          const float2 st = rstat[hh * 32 + rr * 4];
#pragma unroll
          for (int r = 0; r < 4; ++r) res[rr][r] = fmaf(gv[r], (res[rr][r] - st.x) * st.y, ev[r]);
        }
        v[rr] += res[rr];
        *reinterpret_cast<f32x4_t*>(atw(C, ec * 4u)) = v[rr];
#ifdef RUART_ABL_SPLIT8
        store_split8_diag(C16 + (size_t)ec, C8 + 2 * (size_t)ec, v[rr]);
#elif !defined(RUART_ABL_FOLD_NOSPLIT)     // (timing diagnostic: kind 3 without its split copy)
        // (a C8 row is 2 ldc bytes: lo8 halves at [0, N), hi8 halves at [N, 2N))
        store_split4(reinterpret_cast<f16_t*>(atw(C16, ec * 2u)), reinterpret_cast<unsigned char*>(atw(C8, ec * 2u - (unsigned)ncol)), N, v[rr]);
#endif
#ifndef RUART_ABL_FOLD_NOSTATS        // (timing diagnostic: the epilogue without the rows' partial sums - wrong results downstream)
        const float s1 = row16_sum((v[rr][0] + v[rr][1]) + (v[rr][2] + v[rr][3]));
        const float s2 = row16_sum(fmaf(v[rr][0], v[rr][0], v[rr][1] * v[rr][1]) + fmaf(v[rr][2], v[rr][2], v[rr][3] * v[rr][3]));
        if ((lane & 15) == 0) rpart[hh * 32 + rr * 4] = make_float2(s1, s2);
#endif
      }
      continue;
    }
#ifndef RUART_ABL_NOGELU            // (diagnostic builds: the epilogue without its GELU / without its fp8 stores)
    if (EPI == 2) {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) v[rr] = gelu4_as(v[rr]);
    }
#endif
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const unsigned ec = e_c + rr * st_c;
      if (EPI == 1) v[rr] += res[rr];
      if (EPI == 2)
#ifdef RUART_ABL_NOFP8
        *reinterpret_cast<f16x4_t*>(atw(C, ec * 2u)) = (f16x4_t){(f16_t)v[rr][0], (f16_t)v[rr][1], (f16_t)v[rr][2], (f16_t)v[rr][3]};
#elif defined(RUART_ABL_SPLIT8)
        store_split8_diag(reinterpret_cast<f16_t*>(C) + (size_t)ec, C8 + 2 * (size_t)ec, v[rr]);
#else
        store_split4(reinterpret_cast<f16_t*>(atw(C, ec * 2u)), reinterpret_cast<unsigned char*>(atw(C8, ec * 2u - (unsigned)ncol)), N, v[rr]);
#endif
      else if ((EPI == 0 && (RUART_NT_EPI & 1)) || (EPI == 1 && (RUART_NT_EPI & 2)))
        __builtin_nontemporal_store(v[rr], reinterpret_cast<f32x4_t*>(atw(C, ec * 4u)));
      else
        *reinterpret_cast<f32x4_t*>(atw(C, ec * 4u)) = v[rr];
    }
  }
  if constexpr (EPI == 3) {
    // the four column groups of a row in a fixed order -> this tile's partial of the row
    __syncthreads();
    if (threadIdx.x < 256) {
      const float2* pp = reinterpret_cast<const float2*>(smem + kFoldPartOff) + threadIdx.x;
      const float2 a = pp[0], b = pp[256], c = pp[512], d = pp[768];
      reinterpret_cast<float2*>(f.out_part)[(size_t)(m0 + threadIdx.x) * kFoldSlots + n0 / 256] = make_float2((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y));
    }
  }
}

// (tm, tn) of logical tile `id` under the GROUP_M walk (groups of `order` row panels, column-major inside a group)
__device__ __forceinline__ void corr_tile_of(int id, int ntm, int ntn, int order, int& tm, int& tn) {
  if (order == 0) {
    tm = id / ntn;
    tn = id % ntn;
  } else {
    const int per_group = order * ntn;
    const int g = id / per_group, first = g * order;
    const int gsz = min(ntm - first, order);
    const int r = id - g * per_group;
    tm = first + r % gsz;
    tn = r / gsz;
  }
}

// Diagnostic builds (tools/build_variant.sh v224 -DRUART_GEMM_VGPR_HALF=112): cap the kernel at 2 x N architectural VGPRs (hipcc doubles
// an amdgpu_num_vgpr request on the unified register file of gfx90a+), so that 2 waves per SIMD leave registers for a co-resident small
// wave of another kernel.  At 224 the compiler spills inside the K loop: not a product setting.
#ifdef RUART_GEMM_VGPR_HALF
#define RUART_VGPR_ATTR __attribute__((amdgpu_num_vgpr(RUART_GEMM_VGPR_HALF)))
#else
#define RUART_VGPR_ATTR
#endif
// EPI: 0 fp32 out; 1 fp32 out + fp32 residual; 2 GELU, split out (C = f16 rows, C8 = fp8 rows of 2N bytes)
template <int EPI, bool FOLD = false>
__global__ RUART_VGPR_ATTR __launch_bounds__(512, 2) void gemm_16c_nt_256p8(const char* __restrict__ A16, const char* __restrict__ A8, int pitch_a,
                                                            const char* __restrict__ W16, const char* __restrict__ W8, int pitch_w,
                                                            const float* __restrict__ bias, const float* __restrict__ R, int ldr,
                                                            void* __restrict__ C, int ldc, unsigned char* __restrict__ C8, int M, int N,
                                                            int K, int order, int n8, int o8, int n_full, int S,
                                                            float* __restrict__ slabs, const CorrFold f
#ifdef RUART_P8_STAMPS
                                                            , unsigned long long* __restrict__ stamps
#endif
) {
  // diagnostic builds (tools/build_variant.sh stamps -DRUART_P8_STAMPS; tools/r06_corr_stamps.py): s_memrealtime (100 MHz) per workgroup at
  // start / pipeline filled / f16 run done / fp8 run done / epilogue's stores drained, and where the workgroup ran (XCC_ID, HW_ID)
#ifdef RUART_P8_STAMPS
  // (per WAVE: [workgroup][wave][8])
#define C8_STAMP(i) do { if (stamps && (threadIdx.x & 63) == 0) stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  if (stamps && (threadIdx.x & 63) == 0) {
    stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + 6] = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (31 << 11));
    stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + 7] = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
  }
#else
#define C8_STAMP(i)
#endif
  C8_STAMP(0);
  constexpr int kHalf = 128 * CBKB;              // 16 KB half-tile
  constexpr int kOper = 2 * kHalf;               // 32 KB per operand K-tile
  constexpr int kBuf = 2 * kOper;                // 64 KB per K-tile
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // 2 * kBuf = 128 KB, the ONLY LDS object
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int ntn = N / CBN, ntm = M / CBM;
  // Workgroups [0, n_full) own whole tiles (XCD-contiguous walk over them); the rest of the grid are the K SLICES of the last
  // tiles - the tail split of the launcher (corr_tail_plan): S workgroups per tile, dispatched last, each over NT / S K-tiles of
  // one phase, parking its partial sums in `slabs` for gemm_16c_fixup.
  const int bid = blockIdx.x;
  int id, slice = -1;
  if (bid < n_full) {
    id = xcd_remap(bid, n_full);
  } else {
    const int p = bid - n_full;
    id = n_full + p / S;
    slice = p - (p / S) * S;
  }
  // (integer division runs on the vector ALU: pin the wave-uniform results to SGPRs, dma16's operands must be scalar)
  id = __builtin_amdgcn_readfirstlane(id);
  slice = __builtin_amdgcn_readfirstlane(slice);
  int tm, tn;
  corr_tile_of(id, ntm, ntn, order, tm, tn);
  tm = __builtin_amdgcn_readfirstlane(tm);
  tn = __builtin_amdgcn_readfirstlane(tn);
  const int m0 = tm * CBM, n0 = tn * CBN;
  const int nt = K / 64;                         // K-tiles of the f16 phase; the full fp8 phase has as many (2K bytes per row)
  // n8 fp8 K-tiles starting at fp8 tile o8: (nt, 0) = both correction products (production); (nt / 2, 0) = a_lo . w_hi only,
  // (nt / 2, nt / 2) = a_hi . w_lo only, (0, 0) = none - the ablation forms of ruart_gemm_16c_nt_sel
  const int NT = nt + n8;

  // staging: wave w fills local rows 16w .. 16w+15 of a half-tile (two 1 KB pieces of 8 rows x 128 B, lane-linear); the XOR
  // swizzle sits on the SOURCE chunk.  Rows of A8 / W8 have the pitch of A16 / W16, so one per-lane offset serves both phases.
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const unsigned a_lane = (unsigned)(srow * pitch_a + schunk * 16), w_lane = (unsigned)(srow * pitch_w + schunk * 16);
  const size_t a_row0 = (size_t)(m0 + (wave >> 2) * 128 + (wave & 3) * 16) * pitch_a;
  const size_t w_row0 = (size_t)(n0 + (wave >> 1) * 64 + (wave & 1) * 16) * pitch_w;
  const size_t a8r = (size_t)8 * pitch_a, w8r = (size_t)8 * pitch_w, a_h = (size_t)64 * pitch_a, w_h = (size_t)32 * pitch_w;
  char* const st_base = smem + wave * 2048;
  auto stage_a = [&](int d, int h, int kt) {
    char* dst = st_base + d * kBuf + h * kHalf;
    const char* src = (kt < nt ? A16 + (size_t)kt * CBKB : A8 + (size_t)(kt - nt + o8) * CBKB) + a_row0 + h * a_h;
    dma16(src, a_lane, dst);
    dma16(src + a8r, a_lane, dst + 1024);
  };
  auto stage_w = [&](int d, int h, int kt) {
    char* dst = st_base + d * kBuf + kOper + h * kHalf;
    const char* src = (kt < nt ? W16 + (size_t)kt * CBKB : W8 + (size_t)(kt - nt + o8) * CBKB) + w_row0 + h * w_h;
    dma16(src, w_lane, dst);
    dma16(src + w8r, w_lane, dst + 1024);
  };

  f32x4_t acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  // A fragment = the 16-byte chunks fq and 4 + fq of a tile row: for the f16 phase the k-steps 0 and 1 of v_mfma_16x16x32, for
  // the fp8 phase the 32 bytes of one v_mfma_scale_16x16x128 operand (both operands use the same chunk pair, so the k pairing
  // is consistent; the order of k inside a dot product is free).
  i32x8_t af[4], wf0[2], wf1[2];
  auto frag = [&](const char* base, int row) {
    const i32x4_t lo = *reinterpret_cast<const i32x4_t*>(base + lds_off(row, fq));
    const i32x4_t hi = *reinterpret_cast<const i32x4_t*>(base + lds_off(row, 4 + fq));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto read_a = [&](int d, int h) {
    const char* sa = smem + d * kBuf + h * kHalf;
#pragma unroll
    for (int j = 0; j < 4; ++j) af[j] = frag(sa, wm * 64 + j * 16 + fr);
  };
  auto read_w = [&](int d, int h, i32x8_t (&wf)[2]) {
    const char* sw = smem + d * kBuf + kOper + h * kHalf;
#pragma unroll
    for (int i = 0; i < 2; ++i) wf[i] = frag(sw, wn * 32 + i * 16 + fr);
  };
  const int scale_w = 0x01010101 * (127 - RUART_C8_SHIFT), scale_a = 0x7f7f7f7f;     // E8M0: 2^-RUART_C8_SHIFT (2^-18) and 2^0
  auto quad = [&](auto f8tag, int hc, int hr, i32x8_t (&wf)[2]) {
    constexpr bool F8 = decltype(f8tag)::value;
    __builtin_amdgcn_s_setprio(1);
    if constexpr (F8) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[hc * 2 + i][hr * 4 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[i], af[j], acc[hc * 2 + i][hr * 4 + j], RUART_C8_FMT,
                                                                                         RUART_C8_FMT, 0, scale_w, 0, scale_a);
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const i32x4_t w4 = ks ? __builtin_shufflevector(wf[i], wf[i], 4, 5, 6, 7) : __builtin_shufflevector(wf[i], wf[i], 0, 1, 2, 3);
            const i32x4_t a4 = ks ? __builtin_shufflevector(af[j], af[j], 4, 5, 6, 7) : __builtin_shufflevector(af[j], af[j], 0, 1, 2, 3);
            acc[hc * 2 + i][hr * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, w4), __builtin_bit_cast(f16x8_t, a4),
                                                                                  acc[hc * 2 + i][hr * 4 + j], 0, 0, 0);
          }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // one K-tile = four phases (see gemm_16_nt_256p8 for the slot / vmcnt bookkeeping, which is unchanged)
  auto tile = [&](auto f8tag, auto dtag, auto n1tag, auto n2tag, int t) {
    constexpr int D = decltype(dtag)::value;
    constexpr bool N1 = decltype(n1tag)::value, N2 = decltype(n2tag)::value;
    read_w(D, 0, wf0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(D, 0);
    if (N1) stage_a(D ^ 1, 1, t + 1);
#if RUART_P8_WAITS
    // (this tile's W-h1, read one phase on, has landed; five younger half-tiles may be in flight - P8_VMCNT below)
    if (N1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
#endif
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");     // the 4 W-h0 reads (issued first) are back: its slot may be restaged
    RUART_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    quad(f8tag, 0, 0, wf0);
    RUART_BAR();
    read_w(D, 1, wf1);
    if (N2) stage_w(D, 0, t + 2);
#if RUART_P8_WAITS
    // (this tile's A-h1 has landed)
    if (N2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else if (N1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    RUART_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    quad(f8tag, 1, 0, wf1);
    RUART_BAR();
    read_a(D, 1);
    if (N2) stage_a(D, 0, t + 2);
    RUART_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    quad(f8tag, 1, 1, wf1);
    RUART_BAR();
#if RUART_P8_WAITS
    // (K-tile t+1's W-h0 and A-h0 have landed)
    if (N2) {
      stage_w(D, 1, t + 2);
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    } else if (N1) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
#else
    if (N2) {
      stage_w(D, 1, t + 2);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // K-tile t+1 complete; the 3 youngest half-tiles stay in flight
    } else if (N1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#endif
    RUART_BAR();
    quad(f8tag, 0, 1, wf0);
    RUART_BAR();
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using Tt = std::true_type;
  using Ff = std::false_type;
  // MFMA of the two runs.  Diagnostic builds (wrong numbers, right timing): RUART_ABL_MFMA 1 = the fp8 instruction in BOTH runs, 2 = the f16
  // instruction in both - which of the run's properties makes an f16 K-tile 13 % longer than an fp8 K-tile
#ifndef RUART_ABL_MFMA
#define RUART_ABL_MFMA 0
#endif
  using FA = std::conditional_t<RUART_ABL_MFMA == 1, Tt, Ff>;
  using FB = std::conditional_t<RUART_ABL_MFMA == 2, Ff, Tt>;

  // K-tiles [kb, ke) of this workgroup: everything, or slice `slice` of S (S even: a slice never straddles the f16 / fp8 boundary)
  int kb = 0, ke = NT;
  if (slice >= 0) {
    const int L = __builtin_amdgcn_readfirstlane(NT / S);
    kb = slice * L;
    ke = kb + L;
  }
  stage_w(0, 0, kb);
  stage_a(0, 0, kb);
  stage_w(0, 1, kb);
  stage_a(0, 1, kb);
  stage_w(1, 0, kb + 1);
  stage_a(1, 0, kb + 1);
  stage_w(1, 1, kb + 1);
#if RUART_P8_WAITS
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");          // K-tile kb's W-h0 and A-h0 landed (this wave's share)
#else
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");           // K-tile kb landed (this wave's share)
#endif
  RUART_BAR();
  C8_STAMP(1);
#ifdef RUART_P8_STAMPS
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime();      // shader clock over the K loop (slot 5): cycles per K-tile, in-loop clock
#endif
  if (wave >= 4) RUART_BAR();                                 // stagger: waves 4-7 run one barrier behind
  // The K loop as two optional runs - K-tiles [a0, a1) of the f16 phase, then [b0, b1) of the fp8 phase (each empty or an even count
  // >= 2): the whole product is (0, nt, nt, NT), a slice lives in one of the two, the no-correction ablation is (0, nt) alone.  The
  // last two tiles of the LAST run stop prefetching.
  int a0 = 0, a1 = nt, b0 = nt, b1 = NT;
  if (slice >= 0) {
    if (kb < nt) { a0 = kb; a1 = ke; b0 = b1 = 0; }
    else { a0 = a1 = 0; b0 = kb; b1 = ke; }
  }
  const bool a_last = b0 >= b1;
  if (a0 < a1) {
    int t = a0;
    const int body_end = a_last ? a1 - 2 : a1;
    for (; t < body_end; t += 2) {
      tile(FA{}, I0{}, Tt{}, Tt{}, t);
      tile(FA{}, I1{}, Tt{}, Tt{}, t + 1);
    }
    if (a_last) {
      tile(FA{}, I0{}, Tt{}, Ff{}, t);
      tile(FA{}, I1{}, Ff{}, Ff{}, t + 1);
    }
  }
  C8_STAMP(2);
  if (b0 < b1) {
    int t = b0;
    for (; t + 2 < b1; t += 2) {
      tile(FB{}, I0{}, Tt{}, Tt{}, t);
      tile(FB{}, I1{}, Tt{}, Tt{}, t + 1);
    }
    tile(FB{}, I0{}, Tt{}, Ff{}, t);
    tile(FB{}, I1{}, Ff{}, Ff{}, t + 1);
  }
  if (wave < 4) RUART_BAR();                                  // waves 0-3 pair the lagging group's last barrier
  RUART_BAR();                                                // every wave is done reading operand tiles
  C8_STAMP(3);
#ifdef RUART_P8_STAMPS
  if (stamps && (threadIdx.x & 63) == 0) stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + 5] = __builtin_amdgcn_s_memtime() - clk0;
#endif

  if (slice >= 0) {
    // partial sums of this slice, thread-major ([i][j][tid] x 4 floats: 16-byte coalesced stores, read back the same way)
    float* slab = slabs + ((size_t)(id - n_full) * S + slice) * (CBM * CBN);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4_t*>(slab + ((i * 8 + j) * 512 + tid) * 4) = acc[i][j];
    return;
  }
  corr_epilogue<EPI, FOLD>(acc, smem, m0, n0, bias, R, ldr, C, ldc, C8, N, 0, 4, f);
#ifdef RUART_P8_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  C8_STAMP(4);
}

// ---- 256 x 128 tiles, TWO resident workgroups per CU ("dual" form; round 6) -----------------------------------------------------------
// The stamps of gemm_16c_nt_256p8 say its K loop runs at 85 % of the matrix pipe's time and that what is left of a
// tile is the part in which the pipe does nothing at all: 2 us of pipeline fill and 5-17 us of epilogue in a 41-55 us tile - the CU has ONE
// workgroup, and that workgroup is either multiplying or storing.  This form halves the tile along N and the workgroup to four waves (one
// per SIMD, the same 128 x 64 sub-tile, fragment reads and 32 MFMAs per K-tile as a wave of the 256 x 256 kernel), so that a CU holds two
// workgroups (2 x 80 KB of LDS, 2 x 256 registers per SIMD lane): while one stores its tile or fills its pipeline the other multiplies, and
// while both multiply they share each SIMD's matrix pipe the way the two wave groups of the big kernel do - without a barrier between them.
// Costs: an A K-tile (32 KB) now feeds 128 output columns instead of 256: +50 % operand bytes from L2 per flop.
//   * LDS (80 KB): A half-tiles (128 rows x 128 B = 16 KB; half h = rows wm*128 + h*64 .. +64 of both wave rows) in a RING OF THREE - half
//     s = 2t + h of the stream lives in slot s % 3 -, W half-tiles (64 rows = 8 KB; half h = rows wn*64 + h*32 .. +32) double-buffered.
//   * Phase p of K-tile t:  fragment reads | ONE half-tile prefetch | counted wait | lgkmcnt(0) | s_barrier | 8 (fp8) / 16 (f16) MFMAs.
//     ONE barrier per phase (no wave group runs a barrier behind here): a slot is restaged in a later phase than its reads, and every wave
//     completed those reads before the barrier the staging wave has passed; a piece is read in a later phase than the wait that covers
//     it, behind the barrier every wave reaches after its own wait.
//       p0: read W-h0(t), A-h0(t)   stage A-h0(t+1) -> the slot of A-h1(t-1)                         quadrant (0, 0)
//       p1: read W-h1(t)            stage W-h0(t+2) -> its own slot        wait: A-h1(t) landed       quadrant (1, 0)
//       p2: read A-h1(t)            stage A-h1(t+1) -> the slot of A-h0(t)                            quadrant (1, 1)
//       p3:                         stage W-h1(t+2) -> its own slot        wait: A-h0(t+1) landed     quadrant (0, 1)
//     Per wave an A half is 4 LDS-DMA instructions, a W half 2; both waits leave the 8 youngest in flight (vmcnt(8)) and give every piece
//     >= 3 phases to land, as the big kernel's single wait does.
// EPI 0 / 2 (QKV, intermediate dense; FOLD or not), both correction products, no tail split.  M % 256 == 0, N % 128 == 0, K % 128 == 0.
#define DBN 128
// diagnostic builds: RUART_D_PRIO 0 = s_setprio 1 around every quadrant's MFMAs (as the 256 x 256 kernel), 1 = no priority changes, 2 = static
// priority by seat; RUART_D_BAR2 1 = a second barrier behind every quadrant (the 256 x 256 kernel's phase shape)
#ifndef RUART_D_PRIO
#define RUART_D_PRIO 0
#endif
#ifndef RUART_D_BAR2
#define RUART_D_BAR2 0
#endif
#if RUART_D_BAR2
#define D_BAR2() RUART_BAR()
#else
#define D_BAR2()
#endif
template <int EPI, bool FOLD = false>
__global__ __launch_bounds__(256, 2) void gemm_16c_nt_256x128d(const char* __restrict__ A16, const char* __restrict__ A8, int pitch_a,
                                                               const char* __restrict__ W16, const char* __restrict__ W8, int pitch_w,
                                                               const float* __restrict__ bias, void* __restrict__ C, int ldc,
                                                               unsigned char* __restrict__ C8, int M, int N, int K, int order, const CorrFold f
#ifdef RUART_P8_STAMPS
                                                               , unsigned long long* __restrict__ stamps
#endif
) {
#ifdef RUART_P8_STAMPS
#define D8_STAMP(i) do { if (stamps && (threadIdx.x & 63) == 0) stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  if (stamps && (threadIdx.x & 63) == 0) {
    stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + 6] = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (31 << 11));
    stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + 7] = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
  }
#else
#define D8_STAMP(i)
#endif
  D8_STAMP(0);
#if RUART_D_PRIO == 2
  // static priority by SEAT (the wave slot on its SIMD: the two workgroups of a CU sit on different slots): one workgroup of the CU always
  // takes the matrix pipe first, the other fills its gaps
  if (__builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (3 << 11)) & 1) __builtin_amdgcn_s_setprio(1);
#endif
  constexpr int kHalfA = 128 * CBKB;             // 16 KB
  constexpr int kHalfW = 64 * CBKB;              // 8 KB
  constexpr int kWOff = 3 * kHalfA;              // the W slots [D][h] behind the A ring
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // 3 * 16 + 4 * 8 = 80 KB, the ONLY LDS object
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = N / DBN, ntm = M / CBM;
  int id = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x));
  int tm, tn;
  corr_tile_of(id, ntm, ntn, order, tm, tn);
  tm = __builtin_amdgcn_readfirstlane(tm);
  tn = __builtin_amdgcn_readfirstlane(tn);
  const int m0 = tm * CBM, n0 = tn * DBN;
  const int nt = K / 64, NT = 2 * nt;
  // FOLD: thread i finishes the statistics of row m0 + i here, ahead of the loop (two registers), so that the epilogue does not wait for them
  float2 row_stat = make_float2(0.f, 1.f);
  if constexpr (FOLD) row_stat = fold_row_stat_of(f.in_part, f.in_np, m0 + tid, f.inv_h, f.eps);

  // staging: wave w fills local rows 32w .. 32w+31 of an A half (four 1 KB pieces of 8 rows x 128 B, lane-linear) and 16w .. 16w+15 of a
  // W half (two pieces); the XOR swizzle sits on the SOURCE chunk, one per-lane offset serves both phases (gemm_16c_nt_256p8)
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const unsigned a_lane = (unsigned)(srow * pitch_a + schunk * 16), w_lane = (unsigned)(srow * pitch_w + schunk * 16);
  const size_t a_row0 = (size_t)(m0 + (wave >> 1) * 128 + (wave & 1) * 32) * pitch_a;
  const size_t w_row0 = (size_t)(n0 + (wave >> 1) * 64 + (wave & 1) * 16) * pitch_w;
  const size_t a8r = (size_t)8 * pitch_a, w8r = (size_t)8 * pitch_w, a_h = (size_t)64 * pitch_a, w_h = (size_t)32 * pitch_w;
  auto stage_a = [&](int slot, int h, int kt) {
    char* dst = smem + slot * kHalfA + wave * 4096;
    const char* src = (kt < nt ? A16 + (size_t)kt * CBKB : A8 + (size_t)(kt - nt) * CBKB) + a_row0 + h * a_h;
#pragma unroll
    for (int p = 0; p < 4; ++p) dma16(src + p * a8r, a_lane, dst + p * 1024);
  };
  auto stage_w = [&](int d, int h, int kt) {
    char* dst = smem + kWOff + (d * 2 + h) * kHalfW + wave * 2048;
    const char* src = (kt < nt ? W16 + (size_t)kt * CBKB : W8 + (size_t)(kt - nt) * CBKB) + w_row0 + h * w_h;
    dma16(src, w_lane, dst);
    dma16(src + w8r, w_lane, dst + 1024);
  };

  f32x4_t acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  i32x8_t af[4], wf0[2], wf1[2];
  auto frag = [&](const char* base, int row) {
    const i32x4_t lo = *reinterpret_cast<const i32x4_t*>(base + lds_off(row, fq));
    const i32x4_t hi = *reinterpret_cast<const i32x4_t*>(base + lds_off(row, 4 + fq));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto read_a = [&](int slot) {
    const char* sa = smem + slot * kHalfA;
#pragma unroll
    for (int j = 0; j < 4; ++j) af[j] = frag(sa, wm * 64 + j * 16 + fr);
  };
  auto read_w = [&](int d, int h, i32x8_t (&wf)[2]) {
    const char* sw = smem + kWOff + (d * 2 + h) * kHalfW;
#pragma unroll
    for (int i = 0; i < 2; ++i) wf[i] = frag(sw, wn * 32 + i * 16 + fr);
  };
  const int scale_w = 0x01010101 * (127 - RUART_C8_SHIFT), scale_a = 0x7f7f7f7f;
  auto quad = [&](auto f8tag, int hc, int hr, i32x8_t (&wf)[2]) {
    constexpr bool F8 = decltype(f8tag)::value;
#if RUART_D_PRIO == 0
    __builtin_amdgcn_s_setprio(1);
#endif
    if constexpr (F8) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[hc * 2 + i][hr * 4 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[i], af[j], acc[hc * 2 + i][hr * 4 + j], RUART_C8_FMT,
                                                                                         RUART_C8_FMT, 0, scale_w, 0, scale_a);
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const i32x4_t w4 = ks ? __builtin_shufflevector(wf[i], wf[i], 4, 5, 6, 7) : __builtin_shufflevector(wf[i], wf[i], 0, 1, 2, 3);
            const i32x4_t a4 = ks ? __builtin_shufflevector(af[j], af[j], 4, 5, 6, 7) : __builtin_shufflevector(af[j], af[j], 0, 1, 2, 3);
            acc[hc * 2 + i][hr * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, w4), __builtin_bit_cast(f16x8_t, a4),
                                                                                  acc[hc * 2 + i][hr * 4 + j], 0, 0, 0);
          }
    }
#if RUART_D_PRIO == 0
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  int rd = 0;                                     // ring slot of the A half read next (half s -> slot s % 3); the half staged next goes to rd - 1
  auto ring_next = [](int s) { return s == 2 ? 0 : s + 1; };
  auto ring_prev = [](int s) { return s == 0 ? 2 : s - 1; };
  // D: W buffer of this tile; N1 / N2: K-tile t+1 / t+2 exists
  auto tile = [&](auto f8tag, auto dtag, auto n1tag, auto n2tag, int t) {
    constexpr int D = decltype(dtag)::value;
    constexpr bool N1 = decltype(n1tag)::value, N2 = decltype(n2tag)::value;
    // p0
    read_w(D, 0, wf0);
    read_a(rd);
    if (N1) stage_a(ring_prev(rd), 0, t + 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    RUART_BAR();
    quad(f8tag, 0, 0, wf0);
    D_BAR2();
    // p1
    read_w(D, 1, wf1);
    if (N2) stage_w(D, 0, t + 2);
    if (N2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // A-h1(t) has landed; W-h1(t+1), A-h0(t+1), W-h0(t+2) may be in flight
    else if (N1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    RUART_BAR();
    quad(f8tag, 1, 0, wf1);
    D_BAR2();
    // p2
    rd = ring_next(rd);
    read_a(rd);
    if (N1) stage_a(ring_prev(rd), 1, t + 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    RUART_BAR();
    quad(f8tag, 1, 1, wf1);
    D_BAR2();
    // p3
    if (N2) {
      stage_w(D, 1, t + 2);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                // K-tile t+1's W-h0, W-h1, A-h0 have landed; W-h0(t+2), A-h1(t+1), W-h1(t+2) may be in flight
    } else if (N1) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    RUART_BAR();
    quad(f8tag, 0, 1, wf0);
    D_BAR2();
    rd = ring_next(rd);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using Tt = std::true_type;
  using Ff = std::false_type;

  stage_w(0, 0, 0);
  stage_a(0, 0, 0);
  stage_w(0, 1, 0);
  stage_w(1, 0, 1);
  stage_a(1, 1, 0);
  stage_w(1, 1, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             // K-tile 0's W-h0, A-h0, W-h1 landed (this wave's share)
  RUART_BAR();
  D8_STAMP(1);
#ifdef RUART_P8_STAMPS
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime();
#endif
  for (int t = 0; t < nt; t += 2) {                            // the f16 run (the fp8 run follows: every tile prefetches)
    tile(Ff{}, I0{}, Tt{}, Tt{}, t);
    tile(Ff{}, I1{}, Tt{}, Tt{}, t + 1);
  }
  D8_STAMP(2);
  {
    int t = nt;
    for (; t + 2 < NT; t += 2) {
      tile(Tt{}, I0{}, Tt{}, Tt{}, t);
      tile(Tt{}, I1{}, Tt{}, Tt{}, t + 1);
    }
    tile(Tt{}, I0{}, Tt{}, Ff{}, t);
    tile(Tt{}, I1{}, Ff{}, Ff{}, t + 1);
  }
  RUART_BAR();                                                 // every wave is done reading operand tiles
  D8_STAMP(3);
#ifdef RUART_P8_STAMPS
  if (stamps && (threadIdx.x & 63) == 0) stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + 5] = __builtin_amdgcn_s_memtime() - clk0;
#endif
  corr_epilogue<EPI, FOLD, 2>(acc, smem, m0, n0, bias, nullptr, 0, C, ldc, C8, N, 0, 4, f, FOLD ? &row_stat : nullptr);
#ifdef RUART_P8_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  D8_STAMP(4);
}

// Second launch of a tail-split product: tile n_full + blockIdx.x = the sum of its S slices IN SLICE ORDER (deterministic), then the
// tile's epilogue exactly as the GEMM kernel runs it.
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_16c_fixup(const float* __restrict__ slabs, int S, int n_full, const float* __restrict__ bias,
                                                         const float* __restrict__ R, int ldr, void* __restrict__ C, int ldc,
                                                         unsigned char* __restrict__ C8, int M, int N, int order) {
#define TILE_OF(id_, tm_, tn_) corr_tile_of(id_, M / CBM, N / CBN, order, tm_, tn_)
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x;
  const int q = blockIdx.x >> 2, hh = blockIdx.x & 3;       // one 32-rows-per-wave pass of the epilogue per workgroup
  int tm, tn;
  TILE_OF(n_full + q, tm, tn);
  const float* slab = slabs + (size_t)q * S * (CBM * CBN);
  f32x4_t acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int hc = 0; hc < 4; ++hc) {
    if (hc != hh) continue;                                  // (wave-uniform; keeps the accumulator indices static)
    // slices in slice order, two slabs (16 loads per thread) in flight; the second of a pair is clamped and masked at an odd tail
    for (int sl = 0; sl < S; sl += 2) {
      const bool two = sl + 1 < S;
      const float* p0 = slab + (size_t)sl * (CBM * CBN);
      const float* p1 = slab + (size_t)(two ? sl + 1 : sl) * (CBM * CBN);
      f32x4_t a[8], b[8];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          a[i * 2 + j] = *reinterpret_cast<const f32x4_t*>(p0 + ((i * 8 + hc * 2 + j) * 512 + tid) * 4);
          b[i * 2 + j] = *reinterpret_cast<const f32x4_t*>(p1 + ((i * 8 + hc * 2 + j) * 512 + tid) * 4);
        }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][hc * 2 + j] += a[i * 2 + j];
          if (two) acc[i][hc * 2 + j] += b[i * 2 + j];
        }
    }
  }
  corr_epilogue<EPI>(acc, smem, tm * CBM, tn * CBN, bias, R, ldr, C, ldc, C8, N, hh, hh + 1, CorrFold{});
#undef TILE_OF
}

// the epilogue's 32-bit byte offsets: the output (fp32 at most: 4 bytes per element; its split companions are smaller) and the residual
// must span less than 4 GB each
static inline bool corr_spans_ok(int M, int ldc, int ldr) {
  return (size_t)M * (size_t)ldc * 4 < ((size_t)1 << 32) && (size_t)M * (size_t)ldr * 4 < ((size_t)1 << 32);
}
extern int g_tile_order, g_tile_order_auto;
#ifdef RUART_P8_STAMPS
extern unsigned long long* g_p8_stamps;
#endif
// GROUP_M of the tile walk: ruart_tile_group_m (gemm_shared.h) unless ruart_gemm_set_tile_order pinned a value.  L2-miss traffic moves the
// OTHER way (smallest at GROUP_M 2-3, profiles/r03_gemm_order_sweep_fetch.log): it is not what bounds this kernel.
static inline int corr_tile_order(int M, int N, int K) {
  if (!g_tile_order_auto) return g_tile_order;
  return ruart_tile_group_m(M / CBM, N / CBN, K, true);
}
void* ruart_prof_begin_(hipStream_t s, int M, int N, int K);
void ruart_prof_end_(void* rec, hipStream_t s);

// ---- tail split ------------------------------------------------------------------------------------------------------------------
// One 256 x 256 tile per CU means a product runs in whole ROUNDS of `cus` tiles, and the encoder's shapes do not fill their last one:
// at the bench's 167 row tiles the N = 768 products are 501 tiles - on the 240 CUs of the run-ahead stream two full rounds and a
// third of 21 tiles (2.09 -> 3 rounds: the mask cost the attention-output / output dense 44 % of their time), on all 256 CUs 1.96; the
// (64, 512) north-star halves are 1.5 rounds.  With a workspace, the tiles of that last partial round are cut along K into S slices
// each, dispatched behind the full tiles: r * S short workgroups instead of r full-length ones on an otherwise idle chip, then one
// small launch (gemm_16c_fixup) adds a tile's slices in slice order and runs its epilogue.  Deterministic; the plan depends only on
// (M, N, K, cus) - never on the stream the call happens to run on, so a pass on the CU-masked stream and an inline pass agree bit for bit.
struct TailPlan { int n_full, r, S; };
static TailPlan corr_tail_plan(int tiles, int NT, int cus) {
  TailPlan p{tiles, 0, 0};
  if (cus <= 0 || tiles <= cus) return p;                 // (a launch that is one partial round is left alone)
  const int r = tiles % cus;
  // the second launch (slab traffic, ~10 us) pays when the last round is nearly empty, or - up to 60 % full - when a tile is long (K >= 2048)
  if (r == 0 || (4 * r > cus && !(5 * r <= 3 * cus && NT >= 64))) return p;
  int S = cus / r;
  if (S > 8) S = 8;                                        // slabs: S x 256 KB per tile, written and read once
  S &= ~1;                                                 // even: a slice stays inside the f16 or the fp8 phase
  while (S >= 2 && (NT % S != 0 || (NT / S) % 2 != 0 || NT / S < 2)) S -= 2;
  if (S < 2) return p;
  p.n_full = tiles - r;
  p.r = r;
  p.S = S;
  return p;
}
extern "C" size_t ruart_gemm_16c_tail_ws_bytes(int M, int N, int K, int cus) {
  if (M <= 0 || N <= 0 || K <= 0 || M % CBM || N % CBN || K % 128) return 0;
  const TailPlan p = corr_tail_plan((M / CBM) * (N / CBN), 2 * (K / 64), cus);
  return (size_t)p.r * p.S * CBM * CBN * sizeof(float);
}

template <int EPI, bool FOLD = false>
static void launch_corr(const void* A16, const void* A8, int lda, const void* W16, const void* W8, int ldw, const float* bias,
                        const float* residual, int ldr, void* C, int ldc, void* C8, int M, int N, int K, int corr, void* tail_ws,
                        size_t tail_ws_bytes, int cus, hipStream_t s, const CorrFold& fold = CorrFold{}) {
  constexpr int lds = 2 * 2 * CBM * CBKB;                // 128 KB
  // (the epilogue addresses C, C16 / C8 and the residual with 32-bit byte offsets; every caller checks spans_ok first)
  const int nt = K / 64;
  const int n8 = corr == 3 ? nt : (corr ? nt / 2 : 0), o8 = corr == 2 ? nt / 2 : 0;
  const int tiles = (M / CBM) * (N / CBN), order = corr_tile_order(M, N, K);
  TailPlan p{tiles, 0, 0};
  if (corr == 3 && tail_ws) {
    p = corr_tail_plan(tiles, 2 * nt, cus);
    if ((size_t)p.r * p.S * CBM * CBN * sizeof(float) > tail_ws_bytes) p = TailPlan{tiles, 0, 0};
  }
  auto kern = gemm_16c_nt_256p8<EPI, FOLD>;
  static bool done = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds), true);
  (void)done;
  hipLaunchKernelGGL(kern, dim3(p.n_full + p.r * p.S), dim3(512), lds, s, (const char*)A16, (const char*)A8, 2 * lda, (const char*)W16,
                     (const char*)W8, 2 * ldw, bias, residual, ldr, C, ldc, (unsigned char*)C8, M, N, K, order, n8, o8, p.n_full, p.S,
                     (float*)tail_ws, fold
#ifdef RUART_P8_STAMPS
                     , g_p8_stamps
#endif
                     );
  if (p.r > 0) {
    constexpr int flds = 8 * 32 * 272;                    // the epilogue's staging image
    auto fix = gemm_16c_fixup<EPI>;
    static bool fdone = (hipFuncSetAttribute((const void*)fix, hipFuncAttributeMaxDynamicSharedMemorySize, flds), true);
    (void)fdone;
    hipLaunchKernelGGL(fix, dim3(4 * p.r), dim3(512), flds, s, (const float*)tail_ws, p.S, p.n_full, bias, residual, ldr, C, ldc,
                       (unsigned char*)C8, M, N, order);
  }
}

// The 256 x 128 two-workgroups-per-CU form (gemm_16c_nt_256x128d) for the products whose epilogue it carries: EPI 0 / 2, both correction
// products, single launch.  ruart_gemm_16c_set_dual(1) routes them here.
int g_corr_dual = 0;
extern "C" int ruart_gemm_16c_set_dual(int on) {
  RUART_ENTRY();
  const int was = g_corr_dual;
  g_corr_dual = on != 0;
  return was;
}
template <int EPI, bool FOLD = false>
static void launch_corr_dual(const void* A16, const void* A8, int lda, const void* W16, const void* W8, int ldw, const float* bias, void* C,
                             int ldc, void* C8, int M, int N, int K, hipStream_t s, const CorrFold& fold = CorrFold{}) {
  constexpr int lds = 3 * 128 * CBKB + 4 * 64 * CBKB;   // 80 KB
  const int order = g_tile_order_auto ? ruart_tile_group_m(M / CBM, N / DBN, K, true) : g_tile_order;
  auto kern = gemm_16c_nt_256x128d<EPI, FOLD>;
  static bool done = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds), true);
  (void)done;
  hipLaunchKernelGGL(kern, dim3((M / CBM) * (N / DBN)), dim3(256), lds, s, (const char*)A16, (const char*)A8, 2 * lda, (const char*)W16,
                     (const char*)W8, 2 * ldw, bias, C, ldc, (unsigned char*)C8, M, N, K, order, fold
#ifdef RUART_P8_STAMPS
                     , g_p8_stamps
#endif
                     );
}

// ruart_gemm_16c_nt_sel with the tail split (above): tail_ws = ruart_gemm_16c_tail_ws_bytes(M, N, K, cus) bytes of scratch the call
// may use until it has finished on `stream`, cus = the CU count the plan is made for (the stream's CU mask, or the device's CUs);
// tail_ws NULL / too small or cus <= 0: the single-launch form.
extern "C" int ruart_gemm_16c_nt_ws(const void* A16, const void* A8, int lda, const void* W16, const void* W8, int ldw, const float* bias,
                                    const float* residual, int ldr, void* C, int ldc, void* C8, int M, int N, int K, int act, int corr,
                                    void* tail_ws, size_t tail_ws_bytes, int cus, void* stream) {
  RUART_ENTRY();
  if (corr < 0 || corr > 3 || ((corr == 1 || corr == 2) && K % 256)) return (int)hipErrorInvalidValue;
  if (M % CBM || N % CBN || K % 128 || (lda & 7) || (ldw & 7) || (ldc & 3) || lda < K || ldw < K) return (int)hipErrorInvalidValue;
  if (!A16 || !A8 || !W16 || !W8 || !C) return (int)hipErrorInvalidValue;
  if (!corr_spans_ok(M, ldc, residual ? ldr : 0)) return (int)hipErrorInvalidValue;
  // every argument check sits in front of ruart_prof_begin_: an error return never leaves a profiling event open
  if (act == RUART_ACT_GELU) {
    if (residual || !C8 || ldc < N) return (int)hipErrorInvalidValue;
  } else if (act == RUART_ACT_NONE) {
    if (C8) return (int)hipErrorInvalidValue;
  } else {
    return (int)hipErrorInvalidValue;
  }
  hipStream_t s = (hipStream_t)stream;
  void* rec = ruart_prof_begin_(s, M, N, K);
  const bool dual = g_corr_dual && corr == 3 && !residual && !(tail_ws && cus > 0);
  if (dual && act == RUART_ACT_GELU)
    launch_corr_dual<2>(A16, A8, lda, W16, W8, ldw, bias, C, ldc, C8, M, N, K, s);
  else if (dual)
    launch_corr_dual<0>(A16, A8, lda, W16, W8, ldw, bias, C, ldc, nullptr, M, N, K, s);
  else if (act == RUART_ACT_GELU)
    launch_corr<2>(A16, A8, lda, W16, W8, ldw, bias, nullptr, 0, C, ldc, C8, M, N, K, corr, tail_ws, tail_ws_bytes, cus, s);
  else if (residual)
    launch_corr<1>(A16, A8, lda, W16, W8, ldw, bias, residual, ldr, C, ldc, nullptr, M, N, K, corr, tail_ws, tail_ws_bytes, cus, s);
  else
    launch_corr<0>(A16, A8, lda, W16, W8, ldw, bias, nullptr, 0, C, ldc, nullptr, M, N, K, corr, tail_ws, tail_ws_bytes, cus, s);
  ruart_prof_end_(rec, s);
  RUART_CHECK_LAUNCH();
  return 0;
}

// The projections of the LayerNorm-folded encoder pass (CorrFold above; ruart_bert_forward_folded).  Single launch, both correction
// products.  kind 0: C fp32 = rstd 2^s (A W'^T - mu c) + d (QKV; `bias` = d);  kind 2: the same, then GELU, split out (C f16, C8);
// kind 3: y = A W^T + bias + residual - the residual rows normalised with (res_part, res_gamma, res_beta) when res_part != NULL -,
// out: C fp32, C16 / C8 split, out_part[M][4][2] (N <= 1024).  Partials: four (sum, sumsq) slots per row.  `in_part` NULL with kind 0 / 2 = the plain product (rows already normalised).
extern "C" int ruart_gemm_16c_nt_fold(const void* A16, const void* A8, int lda, const void* W16, const void* W8, int ldw, const float* bias,
                                      int kind, const float* in_part, int in_np, const float* colc, float wscale, const float* residual,
                                      int ldr, const float* res_part, int res_np, const float* res_gamma, const float* res_beta, void* C,
                                      int ldc, void* C16, void* C8, float* out_part, int M, int N, int K, int stat_len, float eps,
                                      void* stream) {
  RUART_ENTRY();
  if (M <= 0 || M % CBM || N % CBN || K % 128 || (lda & 7) || (ldw & 7) || (ldc & 3) || lda < K || ldw < K) return (int)hipErrorInvalidValue;
  if (!A16 || !A8 || !W16 || !W8 || !C || stat_len <= 0) return (int)hipErrorInvalidValue;
  if (!corr_spans_ok(M, ldc, residual ? ldr : 0)) return (int)hipErrorInvalidValue;
  if (kind == 3) {
    if (!residual || !C16 || !C8 || !out_part || ldc < N || N > 256 * kFoldSlots || res_np > kFoldSlots || (res_part && (!res_gamma || !res_beta || res_np <= 0))) return (int)hipErrorInvalidValue;
  } else if (kind == 0 || kind == 2) {
    if (in_part && (!colc || in_np <= 0 || in_np > kFoldSlots)) return (int)hipErrorInvalidValue;
    if (kind == 2 ? (!C8 || ldc < N) : (C8 != nullptr)) return (int)hipErrorInvalidValue;
  } else {
    return (int)hipErrorInvalidValue;
  }
  hipStream_t s = (hipStream_t)stream;
  CorrFold f{};
  f.in_part = in_part; f.colc = colc; f.in_np = in_np; f.wscale = wscale;
  f.rs_part = res_part; f.rs_g = res_gamma; f.rs_b = res_beta; f.rs_np = res_np;
  f.out_part = out_part; f.C16 = C16; f.inv_h = 1.0f / (float)stat_len; f.eps = eps;
  void* rec = ruart_prof_begin_(s, M, N, K);
  if (kind == 3)
    launch_corr<3>(A16, A8, lda, W16, W8, ldw, bias, residual, ldr, C, ldc, C8, M, N, K, 3, nullptr, 0, 0, s, f);
  else if (g_corr_dual && kind == 2 && in_part)
    launch_corr_dual<2, true>(A16, A8, lda, W16, W8, ldw, bias, C, ldc, C8, M, N, K, s, f);
  else if (g_corr_dual && kind == 2)
    launch_corr_dual<2>(A16, A8, lda, W16, W8, ldw, bias, C, ldc, C8, M, N, K, s);
  else if (g_corr_dual && in_part)
    launch_corr_dual<0, true>(A16, A8, lda, W16, W8, ldw, bias, C, ldc, nullptr, M, N, K, s, f);
  else if (g_corr_dual)
    launch_corr_dual<0>(A16, A8, lda, W16, W8, ldw, bias, C, ldc, nullptr, M, N, K, s);
  else if (kind == 2 && in_part)
    launch_corr<2, true>(A16, A8, lda, W16, W8, ldw, bias, nullptr, 0, C, ldc, C8, M, N, K, 3, nullptr, 0, 0, s, f);
  else if (kind == 2)
    launch_corr<2>(A16, A8, lda, W16, W8, ldw, bias, nullptr, 0, C, ldc, C8, M, N, K, 3, nullptr, 0, 0, s);
  else if (in_part)
    launch_corr<0, true>(A16, A8, lda, W16, W8, ldw, bias, nullptr, 0, C, ldc, nullptr, M, N, K, 3, nullptr, 0, 0, s, f);
  else
    launch_corr<0>(A16, A8, lda, W16, W8, ldw, bias, nullptr, 0, C, ldc, nullptr, M, N, K, 3, nullptr, 0, 0, s);
  ruart_prof_end_(rec, s);
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" int ruart_gemm_16c_nt_sel(const void* A16, const void* A8, int lda, const void* W16, const void* W8, int ldw, const float* bias,
                                     const float* residual, int ldr, void* C, int ldc, void* C8, int M, int N, int K, int act, int corr,
                                     void* stream) {
  return ruart_gemm_16c_nt_ws(A16, A8, lda, W16, W8, ldw, bias, residual, ldr, C, ldc, C8, M, N, K, act, corr, nullptr, 0, 0, stream);
}

extern "C" int ruart_gemm_16c_nt(const void* A16, const void* A8, int lda, const void* W16, const void* W8, int ldw, const float* bias,
                                 const float* residual, int ldr, void* C, int ldc, void* C8, int M, int N, int K, int act,
                                 void* stream) {
  return ruart_gemm_16c_nt_sel(A16, A8, lda, W16, W8, ldw, bias, residual, ldr, C, ldc, C8, M, N, K, act, 3, stream);
}

// the power-of-two exponents of the e4m3 companions this library was built with (common.h): {SA_LO, SA_HI, SW_HI, SW_LO} - the host side
// prepares weights (and tests prepare operands) with exactly these
extern "C" int ruart_f16c_shifts(int* out4) {
  if (!out4) return (int)hipErrorInvalidValue;
  out4[0] = RUART_C8_SA_LO;
  out4[1] = RUART_C8_SA_HI;
  out4[2] = RUART_C8_SW_HI;
  out4[3] = RUART_C8_SW_LO;
  return 0;
}
