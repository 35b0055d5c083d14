// Shared device helpers for the gfx950 (CDNA4) kernels.  Wave width is 64 everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define RUART_DT_F32 0
#define RUART_DT_BF16 1
#define RUART_DT_F16 2

// hipGetLastError() is per-thread and sticky across libraries: RCCL / torch probe pointers and events in ways that leave
// hipErrorInvalidValue behind.  Every entry point clears the slot first, so RUART_CHECK_LAUNCH reports only its own launches.
#define RUART_ENTRY() (void)hipGetLastError()

#define RUART_CHECK_LAUNCH() \
  do {                       \
    hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) return (int)e_; \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x == 64 * NW; `red` is NW floats of LDS.  All threads get the result.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  if (NW == 1) return v;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += red[i];
  return t;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  if (NW == 1) return v;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
  return t;
}

// ---- typed 4-element row access (float or bf16 storage, fp32 math) ----
__device__ __forceinline__ f32x4_t load4(const float* p) { return *reinterpret_cast<const f32x4_t*>(p); }
// Once-read streams (the encoder's fp32 Q / K / V rows in the attention kernel, the twelve layer matrices in sub-word pooling): the
// non-temporal cache policy.  With the default policy every line of such a stream is allocated in L2 and the Infinity Cache and pushes
// out lines other kernels still want; measured on attn_flash_split_kernel: 155 -> 124-131 us per call (profiles/r05_attn_split_nt.log).
// RUART_NT_STREAM=0 (a -D flag) restores the default policy everywhere (A/B builds).
#ifndef RUART_NT_STREAM
#define RUART_NT_STREAM 1
#endif
__device__ __forceinline__ f32x4_t load4_stream(const float* p) {
#if RUART_NT_STREAM
  return __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
#else
  return load4(p);
#endif
}
__device__ __forceinline__ f32x4_t load4(const bf16_t* p) {
  bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(p);
  f32x4_t r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  return r;
}
__device__ __forceinline__ f32x4_t load4(const f16_t* p) {
  f16x4_t v = *reinterpret_cast<const f16x4_t*>(p);
  f32x4_t r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  return r;
}
__device__ __forceinline__ f32x4_t load4_stream(const bf16_t* p) {
#if RUART_NT_STREAM
  bf16x4_t v = __builtin_nontemporal_load(reinterpret_cast<const bf16x4_t*>(p));
  f32x4_t r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  return r;
#else
  return load4(p);
#endif
}
__device__ __forceinline__ f32x4_t load4_stream(const f16_t* p) {
#if RUART_NT_STREAM
  f16x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f16x4_t*>(p));
  f32x4_t r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  return r;
#else
  return load4(p);
#endif
}
__device__ __forceinline__ void store4(f16_t* p, f32x4_t v) {
  f16x4_t r = {(f16_t)v[0], (f16_t)v[1], (f16_t)v[2], (f16_t)v[3]};
  *reinterpret_cast<f16x4_t*>(p) = r;
}
__device__ __forceinline__ void store4(float* p, f32x4_t v) { *reinterpret_cast<f32x4_t*>(p) = v; }
__device__ __forceinline__ void store4(bf16_t* p, f32x4_t v) {
  bf16x4_t r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
  *reinterpret_cast<bf16x4_t*>(p) = r;
}

// 16-bit MFMA forms: identical rate and fragment layout for bf16 and f16 (MI355X_MICROARCH.md, Matrix cores)
template <typename T> struct Vec8;
template <> struct Vec8<bf16_t> { typedef bf16x8_t type; };
template <> struct Vec8<f16_t> { typedef f16x8_t type; };
__device__ __forceinline__ f32x4_t mfma_16x16x32(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t mfma_16x16x32(f16x8_t a, f16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// XCD-aware remap of a linear workgroup id (cdna_hip_programming.md §5 T1, bijective form):
// ids that share `id % 8` run on one XCD; give each XCD a contiguous chunk of the logical grid
// so that neighbouring tiles (which share an operand panel) hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// ---- "f16 + fp8 correction" storage (RUART_DT_F16C): a value v travels as  hi = f16(v)  plus two e4m3 bytes
//   lo8 = fp8((v - hi) * 2^SA_LO)   and   hi8 = fp8(v * 2^SA_HI)
// so that a GEMM can take  v.w  as  hi.w_hi (f16 MFMA)  +  2^-SHIFT (lo8.w_hi8 + hi8.w_lo8) (block-scaled fp8 MFMA, twice the
// f16 rate), with w_hi8 = fp8(w_hi * 2^SW_HI), w_lo8 = fp8((w - w_hi) * 2^SW_LO) prepared once per weight matrix.
// Exponents: SA_LO + SW_HI == SA_HI + SW_LO == RUART_C8_SHIFT.  Ranges: |v| < 448 and |w| < 3.5 stay below e4m3's 448 (larger
// values saturate - in a correction term only); the f16 rounding residual of |v| >= 2^-6 stays a normal e4m3 number (smaller
// activations carry a subnormal, 1-2 bit residual: their products are negligible next to the row's typical |v| ~ 1).
// Round 3: (SA_LO, SA_HI) moved from (13, 2) - range 112 - to (11, 0): encoder weights with the heavy tails of a pretrained BERT
// (a few LayerNorm gains x 10-30: layer outputs up to |v| ~ 450, tests/golden/sdnet_e2e_outliers.npz) saturated both activation
// companions and the probabilities were off by 1.85e-3; with the wider range that fixture holds 1.6e-4 and the N(0, s) goldens are
// unchanged to within their noise (bench B = 64: 3.9e-5, ragged: 3.3e-4, stress: 1.4e-5).
#ifndef RUART_C8_SA_LO
#define RUART_C8_SA_LO 11
#endif
#ifndef RUART_C8_SA_HI
#define RUART_C8_SA_HI 0
#endif
#define RUART_C8_SW_HI 7
#define RUART_C8_SW_LO (RUART_C8_SA_LO + RUART_C8_SW_HI - RUART_C8_SA_HI)
#define RUART_C8_SHIFT (RUART_C8_SA_LO + RUART_C8_SW_HI)
static_assert(RUART_C8_SA_LO + RUART_C8_SW_HI == RUART_C8_SA_HI + RUART_C8_SW_LO, "both correction products share one scale");

__device__ __forceinline__ unsigned pack_fp8x4(f32x4_t v, float scale) {
  f32x4_t s;
#pragma unroll
  for (int r = 0; r < 4; ++r) s[r] = __builtin_amdgcn_fmed3f(v[r] * scale, -448.0f, 448.0f);
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(s[0], s[1], w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(s[2], s[3], w, true);
  return (unsigned)w;
}
// The same bytes with the scale folded into the conversion (round 6): e4m3(clamp(x 2^s, -448, 448)) == v_cvt_scalef32_pk_fp8_f32 of
// clamp(x, -448 / 2^s, 448 / 2^s) with the scale operand 2^-s - bit for bit on every non-NaN f32 (tools/r06_cvt_probe.hip swept 10^8
// patterns on the device; the scaled conversion by itself does NOT saturate, an overflow gives the NaN code, so the clamp stays).
// One v_mul_f32 per element less than pack_fp8x4.  `s` = log2 of the scale, a compile-time constant >= 1.
template <int S>
__device__ __forceinline__ unsigned pack_fp8x4_shift(f32x4_t v) {
  typedef short s16x2_t __attribute__((ext_vector_type(2)));
  constexpr float lim = 448.0f / (float)(1 << S), inv = 1.0f / (float)(1 << S);
  f32x4_t c;
#pragma unroll
  for (int r = 0; r < 4; ++r) c[r] = __builtin_amdgcn_fmed3f(v[r], -lim, lim);
  s16x2_t w = {0, 0};
  w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, c[0], c[1], inv, false);
  w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, c[2], c[3], inv, true);
  return (unsigned)__builtin_bit_cast(int, w);
}
// x - (float)h for the low / high half of a packed f16 pair: one v_fma_mix_f32 (x * 1.0 - h with the f16 operand read straight from the
// packed register; exact in fp32, so the same bits as the C++ form, which hipcc compiles to v_cvt_f32_f16 + v_sub_f32)
__device__ __forceinline__ float sub_f16_lo(float x, unsigned h2) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(h2));
  return r;
}
__device__ __forceinline__ float sub_f16_hi(float x, unsigned h2) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(h2));
  return r;
}
// RUART_SPLIT_FAST=0 (a -D flag): the round-2 instruction sequence of the split stores (A/B builds); the bytes are the same
#ifndef RUART_SPLIT_FAST
#define RUART_SPLIT_FAST 1
#endif
// the three parts of 4 consecutive values in the split form: f16 x 4 (two packed words), the lo8 word, the hi8 word
__device__ __forceinline__ void split4_words(f32x4_t v, unsigned& h01, unsigned& h23, unsigned& lo8, unsigned& hi8) {
  typedef f16_t f16x2v __attribute__((ext_vector_type(2)));
  const f16x2v a = {(f16_t)v[0], (f16_t)v[1]}, b = {(f16_t)v[2], (f16_t)v[3]};
  h01 = __builtin_bit_cast(unsigned, a);
  h23 = __builtin_bit_cast(unsigned, b);
#if RUART_SPLIT_FAST
  const f32x4_t lo = {sub_f16_lo(v[0], h01), sub_f16_hi(v[1], h01), sub_f16_lo(v[2], h23), sub_f16_hi(v[3], h23)};
  lo8 = pack_fp8x4_shift<RUART_C8_SA_LO>(lo);
#else
  const f32x4_t lo = {v[0] - (float)a[0], v[1] - (float)a[1], v[2] - (float)b[0], v[3] - (float)b[1]};
  lo8 = pack_fp8x4(lo, (float)(1 << RUART_C8_SA_LO));
#endif
#if RUART_C8_SA_HI == 0
  hi8 = pack_fp8x4(v, 1.0f);
#else
  hi8 = pack_fp8x4(v, (float)(1 << RUART_C8_SA_HI));
#endif
}
// store 4 consecutive values in the split form: p16 -> f16 row, p8 -> the row's lo8 bytes, p8 + hi_off -> its hi8 bytes
__device__ __forceinline__ void store_split4(f16_t* p16, unsigned char* p8, int hi_off, f32x4_t v) {
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  unsigned h01, h23, lo8, hi8;
  split4_words(v, h01, h23, lo8, hi8);
  const u32x2_t h = {h01, h23};
  *reinterpret_cast<u32x2_t*>(p16) = h;
  *reinterpret_cast<unsigned*>(p8) = lo8;
  *reinterpret_cast<unsigned*>(p8 + hi_off) = hi8;
}
// Timing diagnostic (-DRUART_ABL_SPLIT8 in gemm_corr.hip; WRONG operand layout): both companions in one 8-byte store at p8x2 = the
// row's byte 2 * column - what a [lo4 | hi4]-interleaved companion layout would allow.
__device__ __forceinline__ void store_split8_diag(f16_t* p16, unsigned char* p8x2, f32x4_t v) {
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const f16x4_t h = {(f16_t)v[0], (f16_t)v[1], (f16_t)v[2], (f16_t)v[3]};
  *reinterpret_cast<f16x4_t*>(p16) = h;
  const f32x4_t lo = {v[0] - (float)h[0], v[1] - (float)h[1], v[2] - (float)h[2], v[3] - (float)h[3]};
  const u32x2_t w = {pack_fp8x4(lo, (float)(1 << RUART_C8_SA_LO)), pack_fp8x4(v, (float)(1 << RUART_C8_SA_HI))};
  *reinterpret_cast<u32x2_t*>(p8x2) = w;
}

// ---- dropout without stored masks: a counter-based hash of (stream seed, element index) decides every element, so the backward
// pass regenerates the forward's mask (training kernels of the encoder, bert_train_*.hip)
__device__ __forceinline__ unsigned hash32(unsigned x) {       // "lowbias32": a well-mixed 32-bit integer hash
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// multiplier of element `idx` under dropout(p) with stream `seed`: 0 or keep_inv = 1/(1-p).
// Two keyed rounds: the streams of different sites / layers / heads differ in `seed` only, and any key that is ADDED to (or XORed
// into) the index once makes one site's mask a translated copy of another's (mask_B[i] == mask_A[i + d], d fixed) - exactly
// correlated wherever both indices are in range.  Here the index is hashed under one key and the result is keyed again, so no
// index map carries one stream onto another.
__device__ __forceinline__ float drop_scale(unsigned seed, unsigned idx, float p, float keep_inv) {
  const unsigned k2 = hash32(seed ^ 0xA511E9B3u);
  const float u = (float)(hash32(hash32(idx + seed) ^ k2) >> 8) * (1.0f / 16777216.0f);
  return u >= p ? keep_inv : 0.f;
}
