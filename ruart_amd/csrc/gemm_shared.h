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

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define RUART_BAR()                          \
  do {                                       \
    asm volatile("" ::: "memory");           \
    __builtin_amdgcn_s_barrier();            \
    asm volatile("" ::: "memory");           \
  } while (0)


