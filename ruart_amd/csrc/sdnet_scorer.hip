// Fused answer scorer of the SDNet trunk: Models/Layers.py:352-432 (GetFinalScores, useES + no_answer branches of the shipped conf)
// with its two BilinearSeqAttn (:435-468) and get_single_score (:421-432) - SURVEY's K14.
//
//   score_i  = x_i . u(i),  u(i) = u2 for the first ES slots (attn2, the ES candidates) and u1 for the OCR slots (attn); -inf where the
//              slot is masked (mask_flag)
//   a        = softmax_i(mask(x_i . uh)),  pooled = sum_i a_i x_i,  s_na = w . pooled + b          (the no-answer score)
//   probs    = softmax([score_0 .. score_{L-1}, s_na])
// u1 = W_attn h0 + b, u2 = W_attn2 h0 + b, uh = W_na h0 + b are three small products the caller takes on ruart_gemm_x3 (with the
// variational-dropout masks of x folded into u1 / u2: (x o m) . u = x . (m o u)).  Unfused this was ~12 launches forward and ~25
// backward around a (B, L, D) broadcast product; here one workgroup per sample does each direction in one launch, x (200 KB per
// sample, L2-resident) read three times, every sum in a fixed order.
#include "common.h"
#include "ruart_hip.h"

#define SC_LMAX 1024
extern int* ruart_nan_flag_ptr;

namespace {

// wave-level dot products of row x_i (D floats, D % 4 == 0) with up to two vectors; lanes stride float4
__device__ __forceinline__ void row_dots(const float* __restrict__ xr, const float* __restrict__ ua, const float* __restrict__ ub, int D,
                                         float& da, float& db) {
  const int lane = threadIdx.x & 63;
  float sa = 0.f, sb = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4_t v = load4(xr + c), a = load4(ua + c), b = load4(ub + c);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sa = fmaf(v[r], a[r], sa);
      sb = fmaf(v[r], b[r], sb);
    }
  }
  da = wave_sum(sa);
  db = wave_sum(sb);
}

__global__ __launch_bounds__(256) void scorer_fwd_kernel(const float* __restrict__ x, const float* __restrict__ u1, const float* __restrict__ u2,
                                                         const float* __restrict__ uh, const float* __restrict__ w, const float* __restrict__ bna,
                                                         const unsigned char* __restrict__ mask, float* __restrict__ probs,
                                                         float* __restrict__ a_out, int L, int D, int ES, int mask_flag,
                                                         int* __restrict__ nan_flag) {
  __shared__ float sc[SC_LMAX + 1], z[SC_LMAX];
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* xb = x + (size_t)b * L * D;
  const float *p1 = u1 + (size_t)b * D, *p2 = u2 + (size_t)b * D, *ph = uh + (size_t)b * D;
  const unsigned char* mb = mask + (size_t)b * L;
  for (int i = wave; i < L; i += 4) {
    float s, zz;
    row_dots(xb + (size_t)i * D, i < ES ? p2 : p1, ph, D, s, zz);
    if (lane == 0) {
      const bool on = mb[i] != 0;
      sc[i] = (mask_flag && !on) ? -INFINITY : s;
      z[i] = on ? zz : -INFINITY;
    }
  }
  __syncthreads();
  // a = softmax(z)
  float mx = -INFINITY;
  for (int i = tid; i < L; i += 256) mx = fmaxf(mx, z[i]);
  mx = block_max<4>(mx, red);
  float sum = 0.f;
  for (int i = tid; i < L; i += 256) sum += __expf(z[i] - mx);
  sum = block_sum<4>(sum, red);
  const float inv = 1.0f / sum;
  __syncthreads();
  for (int i = tid; i < L; i += 256) {
    const float a = __expf(z[i] - mx) * inv;
    z[i] = a;
    a_out[(size_t)b * L + i] = a;
  }
  __syncthreads();
  // pooled[d] = sum_i a_i x_i[d] (rows in order), s_na = w . pooled + b
  float part = 0.f;
  for (int d = tid; d < D; d += 256) {
    float p = 0.f;
    for (int i = 0; i < L; ++i) p = fmaf(z[i], xb[(size_t)i * D + d], p);
    part = fmaf(w[d], p, part);
  }
  part = block_sum<4>(part, red);
  if (tid == 0) sc[L] = part + bna[0];
  __syncthreads();
  // probs = softmax(sc[0..L])
  float m2 = -INFINITY;
  for (int i = tid; i <= L; i += 256) m2 = fmaxf(m2, sc[i]);
  m2 = block_max<4>(m2, red);
  float s2 = 0.f;
  for (int i = tid; i <= L; i += 256) s2 += __expf(sc[i] - m2);
  s2 = block_sum<4>(s2, red);
  const float inv2 = 1.0f / s2;
  bool bad = false;
  for (int i = tid; i <= L; i += 256) {
    const float p = __expf(sc[i] - m2) * inv2;
    probs[(size_t)b * (L + 1) + i] = p;
    bad |= !(p == p);
  }
  if (bad && nan_flag) atomicOr(nan_flag, 1);
}

__global__ __launch_bounds__(256) void scorer_bwd_kernel(const float* __restrict__ x, const float* __restrict__ u1, const float* __restrict__ u2,
                                                         const float* __restrict__ uh, const float* __restrict__ w,
                                                         const float* __restrict__ probs, const float* __restrict__ a_in,
                                                         const float* __restrict__ gprobs, float* __restrict__ gx, float* __restrict__ gu1,
                                                         float* __restrict__ gu2, float* __restrict__ guh, float* __restrict__ gw_part,
                                                         float* __restrict__ gb_part, int L, int D, int ES) {
  __shared__ float ds[SC_LMAX + 1], av[SC_LMAX], dz[SC_LMAX];
  __shared__ float red[4];
  extern __shared__ float dpool[];                       // D floats: d(loss)/d(pooled)
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* xb = x + (size_t)b * L * D;
  const float *p1 = u1 + (size_t)b * D, *p2 = u2 + (size_t)b * D, *ph = uh + (size_t)b * D;
  const float* pb = probs + (size_t)b * (L + 1);
  const float* gb = gprobs + (size_t)b * (L + 1);
  // softmax backward: ds_j = p_j (g_j - sum_k g_k p_k)   (masked slots have p = 0)
  float dot = 0.f;
  for (int i = tid; i <= L; i += 256) dot = fmaf(gb[i], pb[i], dot);
  dot = block_sum<4>(dot, red);
  __syncthreads();
  for (int i = tid; i <= L; i += 256) ds[i] = pb[i] * (gb[i] - dot);
  for (int i = tid; i < L; i += 256) av[i] = a_in[(size_t)b * L + i];
  __syncthreads();
  const float ds_na = ds[L];
  // the no-answer branch: pooled again (rows in order), its gradients to w / b, d(pooled) = ds_na * w
  for (int d = tid; d < D; d += 256) {
    float p = 0.f;
    for (int i = 0; i < L; ++i) p = fmaf(av[i], xb[(size_t)i * D + d], p);
    gw_part[(size_t)b * D + d] = ds_na * p;
    dpool[d] = ds_na * w[d];
  }
  if (tid == 0) gb_part[b] = ds_na;
  __syncthreads();
  // da_i = dpool . x_i ;  dz = a o (da - sum_j a_j da_j)
  for (int i = wave; i < L; i += 4) {
    float da, unused;
    row_dots(xb + (size_t)i * D, dpool, dpool, D, da, unused);
    if (lane == 0) dz[i] = da;
  }
  __syncthreads();
  float adot = 0.f;
  for (int i = tid; i < L; i += 256) adot = fmaf(av[i], dz[i], adot);
  adot = block_sum<4>(adot, red);
  __syncthreads();
  for (int i = tid; i < L; i += 256) dz[i] = av[i] * (dz[i] - adot);
  __syncthreads();
  // gx_i = ds_i u(i) + a_i dpool + dz_i uh
  for (int i = wave; i < L; i += 4) {
    const float* ui = i < ES ? p2 : p1;
    const float s = ds[i], a = av[i], zz = dz[i];
    float* g = gx + ((size_t)b * L + i) * D;
    for (int c = lane * 4; c < D; c += 256) {
      const f32x4_t uv = load4(ui + c), hv = load4(ph + c), dv = *reinterpret_cast<const f32x4_t*>(dpool + c);
      store4(g + c, uv * s + dv * a + hv * zz);
    }
  }
  // gu1 = sum_{i >= ES} ds_i x_i, gu2 = sum_{i < ES} ds_i x_i, guh = sum_i dz_i x_i    (rows in order)
  for (int d = tid; d < D; d += 256) {
    float a1 = 0.f, a2 = 0.f, ah = 0.f;
    for (int i = 0; i < L; ++i) {
      const float v = xb[(size_t)i * D + d];
      if (i < ES) a2 = fmaf(ds[i], v, a2);
      else a1 = fmaf(ds[i], v, a1);
      ah = fmaf(dz[i], v, ah);
    }
    gu1[(size_t)b * D + d] = a1;
    gu2[(size_t)b * D + d] = a2;
    guh[(size_t)b * D + d] = ah;
  }
}

}  // namespace

extern "C" int ruart_scorer_fwd(const float* x, const float* u1, const float* u2, const float* uh, const float* w, const float* bna,
                                const unsigned char* mask, float* probs, float* a_out, int B, int L, int D, int ES, int mask_flag,
                                void* stream) {
  RUART_ENTRY();
  if (B <= 0 || L <= 0 || L > SC_LMAX || D <= 0 || (D & 3) || ES < 0 || ES > L || !x || !u1 || !u2 || !uh || !w || !bna || !mask || !probs || !a_out)
    return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(scorer_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, u1, u2, uh, w, bna, mask, probs, a_out, L, D, ES,
                     mask_flag, ruart_nan_flag_ptr);
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" int ruart_scorer_bwd(const float* x, const float* u1, const float* u2, const float* uh, const float* w, const float* probs,
                                const float* a_in, const float* gprobs, float* gx, float* gu1, float* gu2, float* guh, float* gw_part,
                                float* gb_part, int B, int L, int D, int ES, void* stream) {
  RUART_ENTRY();
  if (B <= 0 || L <= 0 || L > SC_LMAX || D <= 0 || (D & 3) || D > 8192 || ES < 0 || ES > L) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(scorer_bwd_kernel, dim3(B), dim3(256), (size_t)D * sizeof(float), (hipStream_t)stream, x, u1, u2, uh, w, probs, a_in,
                     gprobs, gx, gu1, gu2, guh, gw_part, gb_part, L, D, ES);
  RUART_CHECK_LAUNCH();
  return 0;
}
