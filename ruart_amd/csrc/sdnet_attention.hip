// Fused masked cross-attention of the SDNet trunk (fp32, exact-fp32 MFMA).
//
// Reference: Models/Layers.py:244 (scores = x1_rep.bmm(x2_rep^T)), :275-276 (masked_fill -inf on keys),
// :284-288 (softmax over keys, alpha.bmm(x3)).  10 call sites per forward:
// pre-align x2, deep_attn 3x2, self-attention x2, od_ocr_attn, position_attn, ques_self_attn.
// The ReLU projections a = ReLU(x1 W^T) * diag and k = ReLU(x2 W^T) (Layers.py:226-231) are plain GEMMs done
// by the caller; this file is score / softmax / context and their gradients.
//
// Forward: one workgroup (4 waves) per (batch, 16 query rows).  The 16 x h query tile and the L2 x h key panel are
// staged through LDS in 64-wide chunks of h and multiplied with v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain);
// the 16 x L2 score tile never leaves LDS: masked softmax with 16 lanes per row (shuffle reductions), then the
// probabilities are the A operand of the context product against the value panel staged in LDS.
// LDS row strides are chosen so every MFMA operand read is bank-conflict free:
//   A-type reads  [(lane&15) * ld + 4*kk + (lane>>4)]  want ld % 32 == 2,
//   B-type reads  [(4*kk + (lane>>4)) * ld + (lane&15)] want ld % 32 == 16.
// Backward: kernel A per (batch, 16 query rows): dP = gO . v^T, dS = P * (dP - rowsum(dP * P)), grad_a = dS . k;
// kernel B per (batch, 16 key rows): grad_k = dS^T . a, grad_v = P^T . gO.  No atomics, deterministic.
#include <cstdlib>
#include "common.h"
#include "ruart_hip.h"

#define CH 64          // chunk of the contraction / output dimension staged per pass
#define LDA_ 66        // A-type stride for a 64-wide chunk   (66 % 32 == 2)
#define LDB_ 80        // B-type stride for a 64-wide chunk   (80 % 32 == 16)
#define NTW 6          // key tiles of 16 per wave: 4 waves x 6 x 16 = up to 384 keys
#define ATTN_MAX_L2 384

int* ruart_nan_flag_ptr = nullptr;   // host copy of the device flag address, passed to kernels as an argument

__device__ __forceinline__ f32x4_t mma16(const float* As, int lda, const float* Bs, int sbk, int sbj, int Kc, f32x4_t acc) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  for (int kk = 0; kk < Kc; kk += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(As[r * lda + kk + q], Bs[(kk + q) * sbk + r * sbj], acc, 0, 0, 0);
  return acc;
}

// Activation folded into the staging of the projected operands (Layers.py:228-231): a = ReLU(p1) * diag, k = ReLU(p2).
// diag_len: 0 = none, 1 = one scalar (do_similarity), otherwise a per-column vector of that length.
struct Act {
  int relu;
  const float* diag;
  int diag_len;
  __device__ __forceinline__ float operator()(float v, int col) const {
    if (relu) v = fmaxf(v, 0.f);
    if (diag_len == 1) v *= diag[0];
    else if (diag_len > 1) v *= diag[col];
    return v;
  }
};
__device__ __forceinline__ Act no_act() { return Act{0, nullptr, 0}; }

// stage rows [r0, r0+nr) x cols [c0, c0+CH) of a row-major (rows x cols, ld) matrix into dst[nr_pad][ldd], zero-filled.
// Sixteen independent loads in flight per thread before the first LDS store.  Every load is UNCONDITIONAL - out-of-range elements
// read a clamped (valid) address and are zeroed by a select when they are stored.  Rounds 1-3 wrote `in_range ? src[..] : 0`: hipcc
// branches around such a load and waits vmcnt(0) behind it (cdna_hip_programming.md, projection GEMM item 4c), so the "sixteen in
// flight" were sixteen dependent L2 round trips and the three attention kernels ran at 0.85 TB/s.
template <int U>
__device__ __forceinline__ void stage_u(float* dst, int ldd, int nr_pad, const float* src, int ld, int r0, int rows, int c0, int cols, Act act) {
  const int total = nr_pad * CH;
  for (int e0 = threadIdx.x; e0 < total; e0 += U * blockDim.x) {
    float v[U], dv[U];
    unsigned ok = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * blockDim.x;
      const int r = e / CH, c = e % CH;
      const int gr = r0 + r, gc = c0 + c;
      if (e < total && gr < rows && gc < cols) ok |= 1u << u;
      v[u] = src[(size_t)min(gr, rows - 1) * ld + min(gc, cols - 1)];
    }
    if (act.diag_len > 1) {                      // (wave-uniform: one branch around the group, not one per load)
#pragma unroll
      for (int u = 0; u < U; ++u) dv[u] = act.diag[min(c0 + (e0 + u * (int)blockDim.x) % CH, cols - 1)];
    } else {
      const float d1 = act.diag_len == 1 ? act.diag[0] : 1.f;
#pragma unroll
      for (int u = 0; u < U; ++u) dv[u] = d1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * blockDim.x;
      float x = v[u];
      if (act.relu) x = fmaxf(x, 0.f);
      x *= dv[u];
      if (e < total) dst[(e / CH) * ldd + (e % CH)] = ((ok >> u) & 1u) ? x : 0.f;
    }
  }
}
// the loads in flight per thread follow the tile: a 16-row tile is 4 elements per thread (the trainable encoder's short windows call
// these kernels with 16-row key panels too - sixteen clamped loads each would be 4x the traffic), a 64-row panel 16
__device__ __forceinline__ void stage(float* dst, int ldd, int nr_pad, const float* src, int ld, int r0, int rows, int c0, int cols,
                                      Act act = Act{0, nullptr, 0}) {
  if (nr_pad <= 16) stage_u<4>(dst, ldd, nr_pad, src, ld, r0, rows, c0, cols, act);
  else if (nr_pad <= 32) stage_u<8>(dst, ldd, nr_pad, src, ld, r0, rows, c0, cols, act);
  else stage_u<16>(dst, ldd, nr_pad, src, ld, r0, rows, c0, cols, act);
}
// same but transposed on the fly: dst[c][r] = src[r0 + r][c0 + c]   (dst is [CH][ldd], r < 16)
__device__ __forceinline__ void stage_t16(float* dst, int ldd, const float* src, int ld, int r0, int rows, int c0, int cols,
                                          const float* mul = nullptr) {
  float v[4], m[4];                            // 16 * CH = 1024 elements, 256 threads: all four loads in flight together
  unsigned ok = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = threadIdx.x + u * 256;
    const int c = e & 15, r = e >> 4;          // c: 16 source columns (fast), r: 64 source rows
    const int gr = r0 + r, gc = c0 + c;
    if (gr < rows && gc < cols) ok |= 1u << u;
    const size_t at = (size_t)min(gr, rows - 1) * ld + min(gc, cols - 1);
    v[u] = src[at];
    m[u] = mul ? mul[at] : 1.f;                // (wave-uniform condition; the address is valid either way)
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = threadIdx.x + u * 256;
    dst[(e & 15) * ldd + (e >> 4)] = ((ok >> u) & 1u) ? v[u] * m[u] : 0.f;
  }
}

__device__ __forceinline__ int lds_probs_stride(int L2p) { return ((L2p + 31) / 32) * 32 + 2; }

// ------------------------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) float dsm[];

__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ a, const float* __restrict__ k,
                                                       const float* __restrict__ v, const unsigned char* __restrict__ mask,
                                                       float* __restrict__ out, float* __restrict__ probs, int L1, int L2, int h,
                                                       int D3, int* __restrict__ nan_flag, int relu, const float* __restrict__ diag,
                                                       int diag_len, const float* __restrict__ pscale) {
  const int b = blockIdx.y, i0 = blockIdx.x * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L2p = (L2 + 15) & ~15, ldS = lds_probs_stride(L2p);
  float* a_s = dsm;                       // [16][LDA_]
  float* S_s = a_s + 16 * LDA_;           // [16][ldS]
  float* kv_s = S_s + 16 * ldS;           // [L2p][LDB_]  (keys use stride LDA_, values LDB_)
  const float* ab = a + (size_t)b * L1 * h;
  const float* kb = k + (size_t)b * L2 * h;
  const float* vb = v + (size_t)b * L2 * D3;
  const int ntile = L2p / 16;

  // ---- scores: S[16][L2p] = a_tile . k^T, accumulated over chunks of h
  f32x4_t acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < h; c0 += CH) {
    __syncthreads();
    stage(a_s, LDA_, 16, ab, h, i0, L1, c0, h, Act{relu, diag, diag_len});
    stage(kv_s, LDA_, L2p, kb, h, 0, L2, c0, h, Act{relu, nullptr, 0});
    __syncthreads();
    const int kc = min(CH, (h - c0 + 3) & ~3);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int jt = wave + 4 * t;
      if (jt < ntile) acc[t] = mma16(a_s, LDA_, kv_s + jt * 16 * LDA_, 1, LDA_, kc, acc[t]);
    }
  }
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int jt = wave + 4 * t;
    if (jt < ntile) {
#pragma unroll
      for (int r = 0; r < 4; ++r) S_s[((lane >> 4) * 4 + r) * ldS + jt * 16 + (lane & 15)] = acc[t][r];
    }
  }
  __syncthreads();
  // ---- masked softmax: 16 lanes per row
  {
    const int row = threadIdx.x >> 4, c = threadIdx.x & 15;
    const unsigned char* mb = mask + (size_t)b * L2;
    float* Sr = S_s + row * ldS;
    float mx = -INFINITY;
    for (int j = c; j < L2; j += 16) {
      const float s = mb[j] ? Sr[j] : -INFINITY;
      Sr[j] = s;
      mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int j = c; j < L2; j += 16) {
      const float e = expf(Sr[j] - mx);     // all keys masked => -inf - -inf = NaN, as the reference (Layers.py:290 asserts)
      Sr[j] = e;
      sum += e;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = 1.0f / sum;
    const bool live = (i0 + row) < L1;
    bool bad = false;
    for (int j = c; j < L2p; j += 16) {
      const float p = (j < L2) ? Sr[j] * inv : 0.f;
      // probability dropout (BERT, modeling.py:244-246): the context uses p * pscale, the saved probabilities stay pre-dropout
      const float ps = (pscale && live && j < L2) ? pscale[((size_t)b * L1 + i0 + row) * L2 + j] : 1.f;
      Sr[j] = live ? p * ps : 0.f;
      bad |= live && !(p == p);
      if (probs && live && j < L2) probs[((size_t)b * L1 + i0 + row) * L2 + j] = p;
    }
    if (bad && nan_flag) atomicOr(nan_flag, 1);
  }
  // ---- context: out[16][D3] = P . v, 64 output columns per pass (one 16-col tile per wave)
  for (int d0 = 0; d0 < D3; d0 += CH) {
    __syncthreads();
    stage(kv_s, LDB_, L2p, vb, D3, 0, L2, d0, D3);
    __syncthreads();
    f32x4_t o = {0.f, 0.f, 0.f, 0.f};
    o = mma16(S_s, ldS, kv_s + wave * 16, LDB_, 1, L2p, o);
    const int d = d0 + wave * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + (lane >> 4) * 4 + r;
      if (i < L1 && d < D3) out[((size_t)b * L1 + i) * D3 + d] = o[r];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward A: per (batch, 16 query rows)
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                         const float* __restrict__ probs, const float* __restrict__ gout,
                                                         float* __restrict__ grad_a, float* __restrict__ dS, int L1, int L2, int h,
                                                         int D3, const float* __restrict__ pa, int relu,
                                                         const float* __restrict__ diag, int diag_len,
                                                         float* __restrict__ grad_diag, const float* __restrict__ pscale) {
  const int b = blockIdx.y, i0 = blockIdx.x * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L2p = (L2 + 15) & ~15, ldS = lds_probs_stride(L2p);
  float* g_s = dsm;                       // [16][LDA_]   grad_out chunk
  float* S_s = g_s + 16 * LDA_;           // [16][ldS]    dP then dS
  float* kv_s = S_s + 16 * ldS;           // [L2p][LDB_]
  const float* kb = k + (size_t)b * L2 * h;
  const float* vb = v + (size_t)b * L2 * D3;
  const float* gb = gout + (size_t)b * L1 * D3;
  const int ntile = L2p / 16;

  f32x4_t acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < D3; c0 += CH) {     // dP = gO . v^T
    __syncthreads();
    stage(g_s, LDA_, 16, gb, D3, i0, L1, c0, D3);
    stage(kv_s, LDA_, L2p, vb, D3, 0, L2, c0, D3);
    __syncthreads();
    const int kc = min(CH, (D3 - c0 + 3) & ~3);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int jt = wave + 4 * t;
      if (jt < ntile) acc[t] = mma16(g_s, LDA_, kv_s + jt * 16 * LDA_, 1, LDA_, kc, acc[t]);
    }
  }
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int jt = wave + 4 * t;
    if (jt < ntile) {
#pragma unroll
      for (int r = 0; r < 4; ++r) S_s[((lane >> 4) * 4 + r) * ldS + jt * 16 + (lane & 15)] = acc[t][r];
    }
  }
  __syncthreads();
  {
    const int row = threadIdx.x >> 4, c = threadIdx.x & 15;
    const bool live = (i0 + row) < L1;
    const float* Pr = probs + ((size_t)b * L1 + (live ? i0 + row : 0)) * L2;
    float* Sr = S_s + row * ldS;
    float dot = 0.f;
    if (pscale && live) {                  // d/dP of (P * pscale) . v
      const float* Mr = pscale + ((size_t)b * L1 + i0 + row) * L2;
      for (int j = c; j < L2; j += 16) Sr[j] *= Mr[j];
    }
    for (int j = c; j < L2; j += 16) dot += Sr[j] * Pr[j];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    for (int j = c; j < L2p; j += 16) {
      float d = 0.f;
      if (live && j < L2) {
        const float p = Pr[j];
        d = p * (Sr[j] - dot);
        dS[((size_t)b * L1 + i0 + row) * L2 + j] = d;
      }
      Sr[j] = d;
    }
  }
  for (int d0 = 0; d0 < h; d0 += CH) {      // grad_a = dS . k
    __syncthreads();
    stage(kv_s, LDB_, L2p, kb, h, 0, L2, d0, h, Act{relu, nullptr, 0});
    __syncthreads();
    f32x4_t o = {0.f, 0.f, 0.f, 0.f};
    o = mma16(S_s, ldS, kv_s + wave * 16, LDB_, 1, L2p, o);
    const int d = d0 + wave * 16 + (lane & 15);
    // o = d(loss)/d(a) with a = ReLU(pa) * diag: chain to pa and to diag
    float gd = 0.f;
    const float dsc = (d < h) ? (diag_len == 1 ? diag[0] : (diag_len > 1 ? diag[d] : 1.f)) : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + (lane >> 4) * 4 + r;
      if (i < L1 && d < h) {
        const size_t at = ((size_t)b * L1 + i) * h + d;
        float g = o[r];
        if (pa) {
          const float pv = pa[at];
          gd += g * (relu ? fmaxf(pv, 0.f) : pv);
          g = (relu && pv <= 0.f) ? 0.f : g * dsc;
        }
        grad_a[at] = g;
      }
    }
    if (grad_diag) {      // vector diag only: sum over this tile's 16 rows = the 4 lanes sharing a column; one partial row
      gd += __shfl_xor(gd, 16, 64);     // per workgroup, summed by the caller in a fixed order (no atomics: deterministic)
      gd += __shfl_xor(gd, 32, 64);
      if ((lane >> 4) == 0 && d < h) grad_diag[((size_t)b * gridDim.x + blockIdx.x) * h + d] = gd;
    }
  }
}

// backward B: per (batch, 16 key rows): C[16 j][N] = X^T . Y with X (L1 x L2) in {dS, P}, Y (L1 x N) in {a, gO}
__device__ __forceinline__ void tn_product(const float* X, int L1, int L2, int j0, const float* Y, int N, float* C, float* xt_s,
                                           float* y_s, Act yact, const float* out_gate, const float* xmul = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int n0 = 0; n0 < N; n0 += CH) {
    f32x4_t o = {0.f, 0.f, 0.f, 0.f};
    for (int r0 = 0; r0 < L1; r0 += CH) {
      __syncthreads();
      stage_t16(xt_s, LDA_, X, L2, r0, L1, j0, L2, xmul);     // xt_s[j][i]
      stage(y_s, LDB_, CH, Y, N, r0, L1, n0, N, yact);   // y_s[i][n]
      __syncthreads();
      const int kc = min(CH, (L1 - r0 + 3) & ~3);
      o = mma16(xt_s, LDA_, y_s + wave * 16, LDB_, 1, kc, o);
    }
    const int n = n0 + wave * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = j0 + (lane >> 4) * 4 + r;
      if (j < L2 && n < N) {
        const size_t at = (size_t)j * N + n;
        C[at] = (out_gate && out_gate[at] <= 0.f) ? 0.f : o[r];       // d/d(pk) of ReLU(pk)
      }
    }
  }
}

__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const float* __restrict__ a, const float* __restrict__ probs,
                                                          const float* __restrict__ dS, const float* __restrict__ gout,
                                                          float* __restrict__ grad_k, float* __restrict__ grad_v, int L1, int L2, int h,
                                                          int D3, const float* __restrict__ pk, int relu,
                                                          const float* __restrict__ diag, int diag_len,
                                                          const float* __restrict__ pscale) {
  const int b = blockIdx.y, j0 = blockIdx.x * 16;
  float* xt_s = dsm;                   // [16][LDA_]
  float* y_s = xt_s + 16 * LDA_;       // [CH][LDB_]
  const size_t pb = (size_t)b * L1 * L2;
  tn_product(dS + pb, L1, L2, j0, a + (size_t)b * L1 * h, h, grad_k + (size_t)b * L2 * h, xt_s, y_s, Act{relu, diag, diag_len},
             (relu && pk) ? pk + (size_t)b * L2 * h : nullptr);
  tn_product(probs + pb, L1, L2, j0, gout + (size_t)b * L1 * D3, D3, grad_v + (size_t)b * L2 * D3, xt_s, y_s, no_act(), nullptr,
             pscale ? pscale + pb : nullptr);
}

// ------------------------------------------------------------------------------------------------
// Round 5: the three kernels above with their operand chunks REGISTER-PREFETCHED.  The forms above stage a 64-wide chunk (global loads
// -> wait -> LDS), synchronise, multiply, synchronise - a workgroup's ~8 chunks are ~8 exposed L2 / HBM round trips and the kernels ran
// at 13-15 % of the HBM rate for 41-45 us (profiles/r04_pmc_hbm_per_kernel.csv).  Here a chunk's loads are issued right after the
// previous chunk's registers have gone to LDS, i.e. they fly under that chunk's MFMAs (and the first chunk of a kernel's NEXT product
// under the softmax / dS arithmetic); every load is unconditional (clamped addresses, validity bits applied at the LDS store), and the
// arithmetic - operand values, MFMA order, reductions - is the old kernels' exactly: bit-identical outputs (tested).
// A panel of nr_pad x 64 elements needs nr_pad / 4 registers per thread: instantiated for key counts up to 48 / 64 / 112 / 128; longer
// key panels take the forms above.
template <int U, bool DIAG = false>
struct Panel {
  float v[U];
  float dv[DIAG ? U : 1];
  unsigned ok;
  // rows [r0, r0 + nr_pad) x cols [c0, c0 + CH) of a row-major (rows x cols, ld) matrix; nr_pad * CH <= U * 256
  __device__ __forceinline__ void load(const float* __restrict__ src, int ld, int nr_pad, int r0, int rows, int c0, int cols,
                                       const float* __restrict__ diag = nullptr, int diag_len = 0) {
    ok = 0;
    const int total = nr_pad * CH;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = (int)threadIdx.x + u * 256;
      const int r = e / CH, c = e % CH;
      const int gr = r0 + r, gc = c0 + c;
      if (e < total && gr < rows && gc < cols) ok |= 1u << u;
      v[u] = src[(size_t)min(gr, rows - 1) * ld + min(gc, cols - 1)];
      if (DIAG) dv[u] = diag_len > 1 ? diag[min(gc, cols - 1)] : (diag_len == 1 ? diag[0] : 1.f);
    }
  }
  __device__ __forceinline__ void store(float* dst, int ldd, int nr_pad, int relu) const {
    const int total = nr_pad * CH;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = (int)threadIdx.x + u * 256;
      float x = v[u];
      if (relu) x = fmaxf(x, 0.f);
      x *= DIAG ? dv[u] : 1.f;
      if (e < total) dst[(e / CH) * ldd + (e % CH)] = ((ok >> u) & 1u) ? x : 0.f;
    }
  }
};

template <int UK>
__global__ __launch_bounds__(256) void attn_fwd_pf_kernel(const float* __restrict__ a, const float* __restrict__ k,
                                                          const float* __restrict__ v, const unsigned char* __restrict__ mask,
                                                          float* __restrict__ out, float* __restrict__ probs, int L1, int L2, int h,
                                                          int D3, int* __restrict__ nan_flag, int relu, const float* __restrict__ diag,
                                                          int diag_len, const float* __restrict__ pscale) {
  const int b = blockIdx.y, i0 = blockIdx.x * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L2p = (L2 + 15) & ~15, ldS = lds_probs_stride(L2p);
  float* a_s = dsm;                       // [16][LDA_]
  float* S_s = a_s + 16 * LDA_;           // [16][ldS]
  float* kv_s = S_s + 16 * ldS;           // [L2p][LDB_]  (keys use stride LDA_, values LDB_)
  const float* ab = a + (size_t)b * L1 * h;
  const float* kb = k + (size_t)b * L2 * h;
  const float* vb = v + (size_t)b * L2 * D3;
  const int ntile = L2p / 16;

  Panel<4, true> pa;
  Panel<UK> pk;
  pa.load(ab, h, 16, i0, L1, 0, h, diag, diag_len);
  pk.load(kb, h, L2p, 0, L2, 0, h);
  f32x4_t acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < h; c0 += CH) {
    __syncthreads();
    pa.store(a_s, LDA_, 16, relu);
    pk.store(kv_s, LDA_, L2p, relu);
    {                                       // the next chunk - or the first chunk of the value panel - in flight under the MFMAs
      const bool more = c0 + CH < h;
      pa.load(ab, h, 16, i0, L1, more ? c0 + CH : 0, h, diag, diag_len);
      pk.load(more ? kb : vb, more ? h : D3, L2p, 0, L2, more ? c0 + CH : 0, more ? h : D3);
    }
    __syncthreads();
    const int kc = min(CH, (h - c0 + 3) & ~3);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int jt = wave + 4 * t;
      if (jt < ntile) acc[t] = mma16(a_s, LDA_, kv_s + jt * 16 * LDA_, 1, LDA_, kc, acc[t]);
    }
  }
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int jt = wave + 4 * t;
    if (jt < ntile) {
#pragma unroll
      for (int r = 0; r < 4; ++r) S_s[((lane >> 4) * 4 + r) * ldS + jt * 16 + (lane & 15)] = acc[t][r];
    }
  }
  __syncthreads();
  {
    const int row = threadIdx.x >> 4, c = threadIdx.x & 15;
    const unsigned char* mb = mask + (size_t)b * L2;
    float* Sr = S_s + row * ldS;
    float mx = -INFINITY;
    for (int j = c; j < L2; j += 16) {
      const float s = mb[j] ? Sr[j] : -INFINITY;
      Sr[j] = s;
      mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int j = c; j < L2; j += 16) {
      const float e = expf(Sr[j] - mx);
      Sr[j] = e;
      sum += e;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = 1.0f / sum;
    const bool live = (i0 + row) < L1;
    bool bad = false;
    for (int j = c; j < L2p; j += 16) {
      const float p = (j < L2) ? Sr[j] * inv : 0.f;
      const float ps = (pscale && live && j < L2) ? pscale[((size_t)b * L1 + i0 + row) * L2 + j] : 1.f;
      Sr[j] = live ? p * ps : 0.f;
      bad |= live && !(p == p);
      if (probs && live && j < L2) probs[((size_t)b * L1 + i0 + row) * L2 + j] = p;
    }
    if (bad && nan_flag) atomicOr(nan_flag, 1);
  }
  for (int d0 = 0; d0 < D3; d0 += CH) {
    __syncthreads();
    pk.store(kv_s, LDB_, L2p, 0);
    pk.load(vb, D3, L2p, 0, L2, d0 + CH < D3 ? d0 + CH : 0, D3);
    __syncthreads();
    f32x4_t o = {0.f, 0.f, 0.f, 0.f};
    o = mma16(S_s, ldS, kv_s + wave * 16, LDB_, 1, L2p, o);
    const int d = d0 + wave * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + (lane >> 4) * 4 + r;
      if (i < L1 && d < D3) out[((size_t)b * L1 + i) * D3 + d] = o[r];
    }
  }
}

template <int UK>
__global__ __launch_bounds__(256) void attn_bwd_q_pf_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                            const float* __restrict__ probs, const float* __restrict__ gout,
                                                            float* __restrict__ grad_a, float* __restrict__ dS, int L1, int L2, int h,
                                                            int D3, const float* __restrict__ pa_, int relu,
                                                            const float* __restrict__ diag, int diag_len,
                                                            float* __restrict__ grad_diag, const float* __restrict__ pscale) {
  const int b = blockIdx.y, i0 = blockIdx.x * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L2p = (L2 + 15) & ~15, ldS = lds_probs_stride(L2p);
  float* g_s = dsm;                       // [16][LDA_]   grad_out chunk
  float* S_s = g_s + 16 * LDA_;           // [16][ldS]    dP then dS
  float* kv_s = S_s + 16 * ldS;           // [L2p][LDB_]
  const float* kb = k + (size_t)b * L2 * h;
  const float* vb = v + (size_t)b * L2 * D3;
  const float* gb = gout + (size_t)b * L1 * D3;
  const int ntile = L2p / 16;

  Panel<4> pg;
  Panel<UK> pk;
  pg.load(gb, D3, 16, i0, L1, 0, D3);
  pk.load(vb, D3, L2p, 0, L2, 0, D3);
  f32x4_t acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < D3; c0 += CH) {     // dP = gO . v^T
    __syncthreads();
    pg.store(g_s, LDA_, 16, 0);
    pk.store(kv_s, LDA_, L2p, 0);
    {
      const bool more = c0 + CH < D3;       // next chunk, or the first key chunk of grad_a = dS . k
      pg.load(gb, D3, 16, i0, L1, more ? c0 + CH : 0, D3);
      pk.load(more ? vb : kb, more ? D3 : h, L2p, 0, L2, more ? c0 + CH : 0, more ? D3 : h);
    }
    __syncthreads();
    const int kc = min(CH, (D3 - c0 + 3) & ~3);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int jt = wave + 4 * t;
      if (jt < ntile) acc[t] = mma16(g_s, LDA_, kv_s + jt * 16 * LDA_, 1, LDA_, kc, acc[t]);
    }
  }
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int jt = wave + 4 * t;
    if (jt < ntile) {
#pragma unroll
      for (int r = 0; r < 4; ++r) S_s[((lane >> 4) * 4 + r) * ldS + jt * 16 + (lane & 15)] = acc[t][r];
    }
  }
  __syncthreads();
  {
    const int row = threadIdx.x >> 4, c = threadIdx.x & 15;
    const bool live = (i0 + row) < L1;
    const float* Pr = probs + ((size_t)b * L1 + (live ? i0 + row : 0)) * L2;
    float* Sr = S_s + row * ldS;
    float dot = 0.f;
    if (pscale && live) {
      const float* Mr = pscale + ((size_t)b * L1 + i0 + row) * L2;
      for (int j = c; j < L2; j += 16) Sr[j] *= Mr[j];
    }
    for (int j = c; j < L2; j += 16) dot += Sr[j] * Pr[j];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    for (int j = c; j < L2p; j += 16) {
      float d = 0.f;
      if (live && j < L2) {
        const float p = Pr[j];
        d = p * (Sr[j] - dot);
        dS[((size_t)b * L1 + i0 + row) * L2 + j] = d;
      }
      Sr[j] = d;
    }
  }
  for (int d0 = 0; d0 < h; d0 += CH) {      // grad_a = dS . k
    __syncthreads();
    pk.store(kv_s, LDB_, L2p, relu);
    pk.load(kb, h, L2p, 0, L2, d0 + CH < h ? d0 + CH : 0, h);
    __syncthreads();
    f32x4_t o = {0.f, 0.f, 0.f, 0.f};
    o = mma16(S_s, ldS, kv_s + wave * 16, LDB_, 1, L2p, o);
    const int d = d0 + wave * 16 + (lane & 15);
    float gd = 0.f;
    const float dsc = (d < h) ? (diag_len == 1 ? diag[0] : (diag_len > 1 ? diag[d] : 1.f)) : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + (lane >> 4) * 4 + r;
      if (i < L1 && d < h) {
        const size_t at = ((size_t)b * L1 + i) * h + d;
        float g = o[r];
        if (pa_) {
          const float pv = pa_[at];
          gd += g * (relu ? fmaxf(pv, 0.f) : pv);
          g = (relu && pv <= 0.f) ? 0.f : g * dsc;
        }
        grad_a[at] = g;
      }
    }
    if (grad_diag) {
      gd += __shfl_xor(gd, 16, 64);
      gd += __shfl_xor(gd, 32, 64);
      if ((lane >> 4) == 0 && d < h) grad_diag[((size_t)b * gridDim.x + blockIdx.x) * h + d] = gd;
    }
  }
}

// backward B with prefetch: the (n0, r0) chunk pairs of a product in one flattened walk, the next pair's operands in flight under this
// pair's MFMAs.  X^T tile: 16 source columns x 64 rows (4 elements per thread, optionally times `xmul`), Y chunk: 64 x 64 (16 per thread).
struct PanelT16 {
  float v[4], m[4];
  unsigned ok;
  __device__ __forceinline__ void load(const float* __restrict__ src, int ld, int r0, int rows, int c0, int cols, const float* __restrict__ mul) {
    ok = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = (int)threadIdx.x + u * 256;
      const int c = e & 15, r = e >> 4;
      const int gr = r0 + r, gc = c0 + c;
      if (gr < rows && gc < cols) ok |= 1u << u;
      const size_t at = (size_t)min(gr, rows - 1) * ld + min(gc, cols - 1);
      v[u] = src[at];
      m[u] = mul ? mul[at] : 1.f;
    }
  }
  __device__ __forceinline__ void store(float* dst, int ldd) const {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = (int)threadIdx.x + u * 256;
      dst[(e & 15) * ldd + (e >> 4)] = ((ok >> u) & 1u) ? v[u] * m[u] : 0.f;
    }
  }
};

__device__ __forceinline__ void tn_product_pf(const float* X, int L1, int L2, int j0, const float* Y, int N, float* C, float* xt_s,
                                              float* y_s, Act yact, const float* out_gate, const float* xmul) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nr = (L1 + CH - 1) / CH, nn = (N + CH - 1) / CH, steps = nr * nn;
  PanelT16 px;
  Panel<16, true> py;
  px.load(X, L2, 0, L1, j0, L2, xmul);
  py.load(Y, N, CH, 0, L1, 0, N, yact.diag, yact.diag_len);
  f32x4_t o = {0.f, 0.f, 0.f, 0.f};
  int ri = 0, ni = 0;
  for (int s = 0; s < steps; ++s) {
    const int r0 = ri * CH, n0 = ni * CH;
    __syncthreads();
    px.store(xt_s, LDA_);
    py.store(y_s, LDB_, CH, yact.relu);
    int rn = ri + 1, nx = ni;
    if (rn == nr) { rn = 0; nx = ni + 1; }
    if (nx == nn) { rn = 0; nx = 0; }              // (past the end: a clamped re-read of the first pair, never stored)
    px.load(X, L2, rn * CH, L1, j0, L2, xmul);
    py.load(Y, N, CH, rn * CH, L1, nx * CH, N, yact.diag, yact.diag_len);
    __syncthreads();
    const int kc = min(CH, (L1 - r0 + 3) & ~3);
    o = mma16(xt_s, LDA_, y_s + wave * 16, LDB_, 1, kc, o);
    if (ri == nr - 1) {                              // the chunk column is complete
      const int n = n0 + wave * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = j0 + (lane >> 4) * 4 + r;
        if (j < L2 && n < N) {
          const size_t at = (size_t)j * N + n;
          C[at] = (out_gate && out_gate[at] <= 0.f) ? 0.f : o[r];
        }
      }
      o = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    ri = rn;
    ni = nx;
  }
}

__global__ __launch_bounds__(256) void attn_bwd_kv_pf_kernel(const float* __restrict__ a, const float* __restrict__ probs,
                                                             const float* __restrict__ dS, const float* __restrict__ gout,
                                                             float* __restrict__ grad_k, float* __restrict__ grad_v, int L1, int L2, int h,
                                                             int D3, const float* __restrict__ pk, int relu,
                                                             const float* __restrict__ diag, int diag_len,
                                                             const float* __restrict__ pscale) {
  const int b = blockIdx.y, j0 = blockIdx.x * 16;
  float* xt_s = dsm;                   // [16][LDA_]
  float* y_s = xt_s + 16 * LDA_;       // [CH][LDB_]
  const size_t pb = (size_t)b * L1 * L2;
  tn_product_pf(dS + pb, L1, L2, j0, a + (size_t)b * L1 * h, h, grad_k + (size_t)b * L2 * h, xt_s, y_s, Act{relu, diag, diag_len},
                (relu && pk) ? pk + (size_t)b * L2 * h : nullptr, nullptr);
  tn_product_pf(probs + pb, L1, L2, j0, gout + (size_t)b * L1 * D3, D3, grad_v + (size_t)b * L2 * D3, xt_s, y_s, no_act(), nullptr,
                pscale ? pscale + pb : nullptr);
}

// ------------------------------------------------------------------------------------------------
// Whole-tensor layer norm (Layers.py:167-168): mean / biased variance over ALL n elements, no affine.
// Two-pass (mean, then centred sum of squares) with fixed-order partials => deterministic.
// ------------------------------------------------------------------------------------------------
#ifndef WLN_BLOCKS
#define WLN_BLOCKS 256
#endif
__device__ __forceinline__ float grid_partial_total(const float* part, float* red) {
  // every block re-reduces the WLN_BLOCKS partials in the same order
  float s = (threadIdx.x < WLN_BLOCKS) ? part[threadIdx.x] : 0.f;
  return block_sum<4>(s, red);
}

// The streaming loops below keep FOUR independent 16-byte loads per thread in flight (clamped indices, a select afterwards: no load sits
// behind a branch); n % 4 != 0 or an unaligned base takes the scalar tail form for everything.
#define WLN_FOR4(n4, ...)                                                                          \
  for (long long i_ = blockIdx.x * 256LL + threadIdx.x; i_ < (n4); i_ += 4LL * 256 * WLN_BLOCKS) { \
    long long ix_[4];                                                                              \
    bool ok_[4];                                                                                   \
    _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) {                                             \
      const long long j_ = i_ + (long long)u_ * 256 * WLN_BLOCKS;                                  \
      ok_[u_] = j_ < (n4);                                                                         \
      ix_[u_] = ok_[u_] ? j_ : (n4) - 1;                                                           \
    }                                                                                              \
    __VA_ARGS__                                                                                    \
  }
__device__ __forceinline__ bool wln_vec_ok(const void* a, const void* b, const void* c, long long n) {
  return ((n & 3) == 0) && (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0) && n >= 4;
}

__global__ __launch_bounds__(256) void wln_sum_kernel(const float* __restrict__ x, const float* __restrict__ x2, long long n,
                                                      float* __restrict__ part) {
  __shared__ float red[4];
  float s = 0.f, t = 0.f;
  if (wln_vec_ok(x, x2, nullptr, n)) {
    const f32x4_t* x4 = reinterpret_cast<const f32x4_t*>(x);
    const f32x4_t* y4 = reinterpret_cast<const f32x4_t*>(x2);
    const long long n4 = n >> 2;
    if (x2) {
      WLN_FOR4(n4, f32x4_t a[4], b[4];
               _Pragma("unroll") for (int u = 0; u < 4; ++u) { a[u] = x4[ix_[u]]; b[u] = y4[ix_[u]]; }
               _Pragma("unroll") for (int u = 0; u < 4; ++u) if (ok_[u]) {
                 s += (a[u][0] + a[u][1]) + (a[u][2] + a[u][3]);
                 t += (a[u][0] * b[u][0] + a[u][1] * b[u][1]) + (a[u][2] * b[u][2] + a[u][3] * b[u][3]);
               })
    } else {
      WLN_FOR4(n4, f32x4_t a[4];
               _Pragma("unroll") for (int u = 0; u < 4; ++u) a[u] = x4[ix_[u]];
               _Pragma("unroll") for (int u = 0; u < 4; ++u) if (ok_[u]) s += (a[u][0] + a[u][1]) + (a[u][2] + a[u][3]);)
    }
  } else {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += 256LL * WLN_BLOCKS) {
      const float a = x[i];
      s += a;
      if (x2) t += a * x2[i];
    }
  }
  s = block_sum<4>(s, red);
  if (x2) t = block_sum<4>(t, red);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = s;
    if (x2) part[WLN_BLOCKS + blockIdx.x] = t;
  }
}
__global__ __launch_bounds__(256) void wln_var_kernel(const float* __restrict__ x, long long n, const float* __restrict__ part,
                                                      float* __restrict__ part2) {
  __shared__ float red[4];
  const float mean = grid_partial_total(part, red) / (float)n;
  float q = 0.f;
  if (wln_vec_ok(x, nullptr, nullptr, n)) {
    const f32x4_t* x4 = reinterpret_cast<const f32x4_t*>(x);
    const long long n4 = n >> 2;
    WLN_FOR4(n4, f32x4_t a[4];
             _Pragma("unroll") for (int u = 0; u < 4; ++u) a[u] = x4[ix_[u]];
             _Pragma("unroll") for (int u = 0; u < 4; ++u) if (ok_[u]) {
               const f32x4_t d = a[u] - mean;
               q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
             })
  } else {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += 256LL * WLN_BLOCKS) {
      const float d = x[i] - mean;
      q += d * d;
    }
  }
  q = block_sum<4>(q, red);
  if (threadIdx.x == 0) part2[blockIdx.x] = q;
}
__global__ __launch_bounds__(256) void wln_apply_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, float eps,
                                                        const float* __restrict__ part, const float* __restrict__ part2,
                                                        float* __restrict__ stats, int* __restrict__ nan_flag) {
  __shared__ float red[4];
  const float mean = grid_partial_total(part, red) / (float)n;
  const float var = grid_partial_total(part2, red) / (float)n;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    stats[0] = mean;
    stats[1] = rstd;
  }
  bool bad = false;
  if (wln_vec_ok(x, y, nullptr, n)) {
    const f32x4_t* x4 = reinterpret_cast<const f32x4_t*>(x);
    f32x4_t* y4 = reinterpret_cast<f32x4_t*>(y);
    const long long n4 = n >> 2;
    WLN_FOR4(n4, f32x4_t a[4];
             _Pragma("unroll") for (int u = 0; u < 4; ++u) a[u] = x4[ix_[u]];
             _Pragma("unroll") for (int u = 0; u < 4; ++u) if (ok_[u]) {
               const f32x4_t o = (a[u] - mean) * rstd;
               y4[ix_[u]] = o;
               bad |= !(o[0] == o[0]) || !(o[1] == o[1]) || !(o[2] == o[2]) || !(o[3] == o[3]);
             })
  } else {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += 256LL * WLN_BLOCKS) {
      const float o = (x[i] - mean) * rstd;
      y[i] = o;
      bad |= !(o == o);
    }
  }
  if (bad && nan_flag) atomicOr(nan_flag, 1);
}
__global__ __launch_bounds__(256) void wln_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gy,
                                                      const float* __restrict__ stats, float* __restrict__ gx, long long n,
                                                      const float* __restrict__ part) {
  __shared__ float red[4];
  const float mg = grid_partial_total(part, red) / (float)n;                 // mean(gy)
  const float mgy = grid_partial_total(part + WLN_BLOCKS, red) / (float)n;   // mean(gy * y)
  const float rstd = stats[1];
  if (wln_vec_ok(y, gy, gx, n)) {
    const f32x4_t* y4 = reinterpret_cast<const f32x4_t*>(y);
    const f32x4_t* g4 = reinterpret_cast<const f32x4_t*>(gy);
    f32x4_t* o4 = reinterpret_cast<f32x4_t*>(gx);
    const long long n4 = n >> 2;
    WLN_FOR4(n4, f32x4_t a[4], b[4];
             _Pragma("unroll") for (int u = 0; u < 4; ++u) { a[u] = y4[ix_[u]]; b[u] = g4[ix_[u]]; }
             _Pragma("unroll") for (int u = 0; u < 4; ++u) if (ok_[u]) o4[ix_[u]] = rstd * (b[u] - mg - a[u] * mgy);)
  } else {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += 256LL * WLN_BLOCKS) gx[i] = rstd * (gy[i] - mg - y[i] * mgy);
  }
}

// ------------------------------------------------------------------------------------------------
static bool g_attn_prefetch = !(getenv("RUART_ATTN_PREFETCH") && atoi(getenv("RUART_ATTN_PREFETCH")) == 0);
template <typename K> static void attn_big_lds(K kern) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
static void attn_allow_big_lds() {
  static bool done = (attn_big_lds(attn_fwd_kernel), attn_big_lds(attn_bwd_q_kernel), attn_big_lds(attn_fwd_pf_kernel<12>),
                      attn_big_lds(attn_fwd_pf_kernel<16>), attn_big_lds(attn_fwd_pf_kernel<28>), attn_big_lds(attn_fwd_pf_kernel<32>),
                      attn_big_lds(attn_bwd_q_pf_kernel<12>), attn_big_lds(attn_bwd_q_pf_kernel<16>), attn_big_lds(attn_bwd_q_pf_kernel<28>),
                      attn_big_lds(attn_bwd_q_pf_kernel<32>), true);
  (void)done;
}

static size_t attn_lds_bytes(int L2) {
  const int L2p = (L2 + 15) & ~15;
  const int ldS = ((L2p + 31) / 32) * 32 + 2;
  return sizeof(float) * (size_t)(16 * LDA_ + 16 * ldS + L2p * LDB_);
}

// ------------------------------------------------------------------------------------------------
// Single-query form (L1 == 1, no activation): LinearSelfAttn.merge and the no-answer score (Layers.py:310-322, 421-432) attend with
// ONE query row per batch element.  The tiled kernels above would run a 16-row MFMA tile for it and walk the key / value
// panels through LDS with two barriers per 64 columns (~100 us for 64 x 100 x 500); here one workgroup per batch element does
// the dot products with wave reductions and the context with coalesced row reads (~10 us).  Same arithmetic order per output
// on every run (deterministic).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn1_fwd_kernel(const float* __restrict__ a, const float* __restrict__ k,
                                                        const float* __restrict__ v, const unsigned char* __restrict__ mask,
                                                        float* __restrict__ out, float* __restrict__ probs, int L2, int h, int D3,
                                                        int* __restrict__ nan_flag) {
  __shared__ float p_s[ATTN_MAX_L2];
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* ab = a + (size_t)b * h;
  const float* kb = k + (size_t)b * L2 * h;
  const float* vb = v + (size_t)b * L2 * D3;
  for (int j0 = wave * 4; j0 < L2; j0 += 16) {         // a wave takes 4 keys at a time: 4 independent coalesced dot products
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d = lane; d < h; d += 64) {
      const float av = ab[d];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (j0 + u < L2) s[u] = fmaf(av, kb[(size_t)(j0 + u) * h + d], s[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float t = wave_sum(s[u]);
      if (lane == 0 && j0 + u < L2) p_s[j0 + u] = mask[(size_t)b * L2 + j0 + u] ? t : -INFINITY;
    }
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int j = tid; j < L2; j += 256) mx = fmaxf(mx, p_s[j]);
  mx = block_max<4>(mx, red);
  float sum = 0.f;
  for (int j = tid; j < L2; j += 256) {
    const float e = expf(p_s[j] - mx);                  // all keys masked => NaN, as the reference
    p_s[j] = e;
    sum += e;
  }
  sum = block_sum<4>(sum, red);
  __syncthreads();
  const float inv = 1.0f / sum;
  bool bad = false;
  for (int j = tid; j < L2; j += 256) {
    const float p = p_s[j] * inv;
    p_s[j] = p;
    bad |= !(p == p);
    if (probs) probs[(size_t)b * L2 + j] = p;
  }
  if (bad && nan_flag) atomicOr(nan_flag, 1);
  __syncthreads();
  for (int d = tid; d < D3; d += 256) {                 // context: rows of v read coalesced, keys in order, 8 loads in flight
    float o = 0.f;
    int j = 0;
    for (; j + 8 <= L2; j += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = vb[(size_t)(j + u) * D3 + d];
#pragma unroll
      for (int u = 0; u < 8; ++u) o = fmaf(p_s[j + u], t[u], o);
    }
    for (; j < L2; ++j) o = fmaf(p_s[j], vb[(size_t)j * D3 + d], o);
    out[(size_t)b * D3 + d] = o;
  }
}

// backward of the single-query form: grad_v[j] = p_j gO, dp_j = gO . v_j, ds = p (dp - sum p dp), grad_a = sum_j ds_j k_j, grad_k[j] = ds_j a
__global__ __launch_bounds__(256) void attn1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ k,
                                                        const float* __restrict__ v, const float* __restrict__ probs,
                                                        const float* __restrict__ gout, float* __restrict__ grad_a,
                                                        float* __restrict__ grad_k, float* __restrict__ grad_v, int L2, int h, int D3) {
  __shared__ float ds_s[ATTN_MAX_L2];
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* ab = a + (size_t)b * h;
  const float* kb = k + (size_t)b * L2 * h;
  const float* vb = v + (size_t)b * L2 * D3;
  const float* gb = gout + (size_t)b * D3;
  const float* pb = probs + (size_t)b * L2;
  for (int j0 = wave * 4; j0 < L2; j0 += 16) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d = lane; d < D3; d += 64) {
      const float gv = gb[d];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (j0 + u < L2) s[u] = fmaf(gv, vb[(size_t)(j0 + u) * D3 + d], s[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float t = wave_sum(s[u]);
      if (lane == 0 && j0 + u < L2) ds_s[j0 + u] = t;   // dp_j
    }
  }
  __syncthreads();
  float dot = 0.f;
  for (int j = tid; j < L2; j += 256) dot += ds_s[j] * pb[j];
  dot = block_sum<4>(dot, red);
  __syncthreads();
  for (int j = tid; j < L2; j += 256) ds_s[j] = pb[j] * (ds_s[j] - dot);
  __syncthreads();
  for (int d = tid; d < h; d += 256) {
    float g = 0.f;
    int j = 0;
    for (; j + 8 <= L2; j += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = kb[(size_t)(j + u) * h + d];
#pragma unroll
      for (int u = 0; u < 8; ++u) g = fmaf(ds_s[j + u], t[u], g);
    }
    for (; j < L2; ++j) g = fmaf(ds_s[j], kb[(size_t)j * h + d], g);
    grad_a[(size_t)b * h + d] = g;
  }
  for (int e = tid; e < L2 * h; e += 256) grad_k[(size_t)b * L2 * h + e] = ds_s[e / h] * ab[e % h];
  for (int e = tid; e < L2 * D3; e += 256) grad_v[(size_t)b * L2 * D3 + e] = pb[e / D3] * gb[e % D3];
}

extern "C" int ruart_attn_fwd(const float* a, const float* k, const float* v, const unsigned char* mask, const float* diag,
                              int diag_len, int relu, float* out, float* probs, int B, int L1, int L2, int h, int D3, void* stream) {
  RUART_ENTRY();
  return ruart_attn_fwd_pscale(a, k, v, mask, diag, diag_len, relu, nullptr, out, probs, B, L1, L2, h, D3, stream);
}

extern "C" int ruart_attn_fwd_pscale(const float* a, const float* k, const float* v, const unsigned char* mask, const float* diag,
                                     int diag_len, int relu, const float* prob_scale, float* out, float* probs, int B, int L1, int L2,
                                     int h, int D3, void* stream) {
  RUART_ENTRY();
  if (B <= 0 || L1 <= 0 || L2 <= 0 || L2 > ATTN_MAX_L2 || h <= 0 || D3 <= 0) return (int)hipErrorInvalidValue;
  if (diag_len != 0 && diag_len != 1 && diag_len != h) return (int)hipErrorInvalidValue;
  if (L1 == 1 && !relu && diag_len == 0 && !prob_scale) {              // single-query fast path
    hipLaunchKernelGGL(attn1_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, a, k, v, mask, out, probs, L2, h, D3,
                       ruart_nan_flag_ptr);
    RUART_CHECK_LAUNCH();
    return 0;
  }
  const dim3 grid(ceil_div(L1, 16), B), block(256);
  attn_allow_big_lds();
  const int L2p = (L2 + 15) & ~15;
#define RUART_AF(U) hipLaunchKernelGGL(attn_fwd_pf_kernel<U>, grid, block, attn_lds_bytes(L2), (hipStream_t)stream, a, k, v, mask, out, probs, \
                                       L1, L2, h, D3, ruart_nan_flag_ptr, relu, diag_len ? diag : nullptr, diag_len, prob_scale)
  if (!g_attn_prefetch || L2p > 128)
    hipLaunchKernelGGL(attn_fwd_kernel, grid, block, attn_lds_bytes(L2), (hipStream_t)stream, a, k, v, mask, out, probs, L1, L2, h, D3,
                       ruart_nan_flag_ptr, relu, diag_len ? diag : nullptr, diag_len, prob_scale);
  else if (L2p <= 48) RUART_AF(12);
  else if (L2p <= 64) RUART_AF(16);
  else if (L2p <= 112) RUART_AF(28);
  else RUART_AF(32);
#undef RUART_AF
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" int ruart_attn_set_prefetch(int on) {
  RUART_ENTRY();
  g_attn_prefetch = on != 0;
  return 0;
}

extern "C" int ruart_attn_bwd(const float* a, const float* k, const float* v, const float* probs, const float* grad_out,
                              const float* diag, int diag_len, int relu, float* grad_a, float* grad_k, float* grad_v,
                              float* grad_diag, float* ds_ws, int B, int L1, int L2, int h, int D3, void* stream) {
  RUART_ENTRY();
  return ruart_attn_bwd_pscale(a, k, v, probs, grad_out, diag, diag_len, relu, nullptr, grad_a, grad_k, grad_v, grad_diag, ds_ws, B, L1,
                               L2, h, D3, stream);
}

extern "C" int ruart_attn_bwd_pscale(const float* a, const float* k, const float* v, const float* probs, const float* grad_out,
                                     const float* diag, int diag_len, int relu, const float* prob_scale, float* grad_a, float* grad_k,
                                     float* grad_v, float* grad_diag, float* ds_ws, int B, int L1, int L2, int h, int D3,
                                     void* stream) {
  RUART_ENTRY();
  if (B <= 0 || L1 <= 0 || L2 <= 0 || L2 > ATTN_MAX_L2 || h <= 0 || D3 <= 0) return (int)hipErrorInvalidValue;
  if (diag_len != 0 && diag_len != 1 && diag_len != h) return (int)hipErrorInvalidValue;
  if (L1 == 1 && !relu && diag_len == 0 && !prob_scale) {              // single-query fast path (ds_ws unused)
    hipLaunchKernelGGL(attn1_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, a, k, v, probs, grad_out, grad_a, grad_k, grad_v,
                       L2, h, D3);
    RUART_CHECK_LAUNCH();
    return 0;
  }
  const bool act = relu || diag_len;
  const float* dg = diag_len ? diag : nullptr;
  attn_allow_big_lds();
  const int L2p = (L2 + 15) & ~15;
#define RUART_AQ(U) hipLaunchKernelGGL(attn_bwd_q_pf_kernel<U>, dim3(ceil_div(L1, 16), B), dim3(256), attn_lds_bytes(L2), (hipStream_t)stream, \
                                       k, v, probs, grad_out, grad_a, ds_ws, L1, L2, h, D3, act ? a : nullptr, relu, dg, diag_len,           \
                                       (diag_len > 1) ? grad_diag : nullptr, prob_scale)
  if (!g_attn_prefetch || L2p > 128)
    hipLaunchKernelGGL(attn_bwd_q_kernel, dim3(ceil_div(L1, 16), B), dim3(256), attn_lds_bytes(L2), (hipStream_t)stream, k, v, probs,
                       grad_out, grad_a, ds_ws, L1, L2, h, D3, act ? a : nullptr, relu, dg, diag_len,
                       (diag_len > 1) ? grad_diag : nullptr, prob_scale);
  else if (L2p <= 48) RUART_AQ(12);
  else if (L2p <= 64) RUART_AQ(16);
  else if (L2p <= 112) RUART_AQ(28);
  else RUART_AQ(32);
#undef RUART_AQ
  RUART_CHECK_LAUNCH();
  const size_t lds = sizeof(float) * (16 * LDA_ + CH * LDB_);
  if (g_attn_prefetch)
    hipLaunchKernelGGL(attn_bwd_kv_pf_kernel, dim3(ceil_div(L2, 16), B), dim3(256), lds, (hipStream_t)stream, a, probs, ds_ws, grad_out,
                       grad_k, grad_v, L1, L2, h, D3, k, relu, dg, diag_len, prob_scale);
  else
    hipLaunchKernelGGL(attn_bwd_kv_kernel, dim3(ceil_div(L2, 16), B), dim3(256), lds, (hipStream_t)stream, a, probs, ds_ws, grad_out,
                       grad_k, grad_v, L1, L2, h, D3, k, relu, dg, diag_len, prob_scale);
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" int ruart_whole_ln_fwd(const float* x, float* y, float* stats, float* ws, long long n, float eps, void* stream) {
  RUART_ENTRY();
  if (n <= 0) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(wln_sum_kernel, dim3(WLN_BLOCKS), dim3(256), 0, s, x, (const float*)nullptr, n, ws);
  hipLaunchKernelGGL(wln_var_kernel, dim3(WLN_BLOCKS), dim3(256), 0, s, x, n, ws, ws + 2 * WLN_BLOCKS);
  hipLaunchKernelGGL(wln_apply_kernel, dim3(WLN_BLOCKS), dim3(256), 0, s, x, y, n, eps, ws, ws + 2 * WLN_BLOCKS, stats, ruart_nan_flag_ptr);
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" int ruart_whole_ln_bwd(const float* y, const float* grad_y, const float* stats, float* grad_x, float* ws, long long n,
                                  void* stream) {
  RUART_ENTRY();
  if (n <= 0) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(wln_sum_kernel, dim3(WLN_BLOCKS), dim3(256), 0, s, grad_y, y, n, ws);
  hipLaunchKernelGGL(wln_bwd_kernel, dim3(WLN_BLOCKS), dim3(256), 0, s, y, grad_y, stats, grad_x, n, ws);
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" int ruart_stream_create_cu_masked(int n_cus, void** stream_out) {
  RUART_ENTRY();
  if (!stream_out) return -1;
  int dev = 0, total = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -2;
  if (hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -2;
  hipStream_t s = nullptr;
  hipError_t e;
  if (n_cus < 0 && -n_cus < total) {
    // the LAST |n_cus| bits (experiments: a second set of streams kept off most of the CUs a prefix-masked stream uses)
    uint32_t mask[16] = {0};
    for (int i = total + n_cus; i < total && i < 512; ++i) mask[i >> 5] |= 1u << (i & 31);
    e = hipExtStreamCreateWithCUMask(&s, (uint32_t)((total + 31) / 32), mask);
  } else if (n_cus <= 0 || n_cus >= total) {
    e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  } else {
    // the FIRST n_cus bits.  Measured on MI355X (fp16c bench, encoder stream masked): n_cus = 240 / 224 of 256 -> 25.8 ms per step
    // against 26.7 unmasked, while leaving out the same number of CUs evenly spaced over the bit range made it 28-35 ms: the
    // runtime's bit order interleaves the XCDs, so a prefix that is a multiple of 8 shrinks every XCD alike, whereas scattered
    // holes unbalance them (workgroups are dealt round-robin over the XCDs: the smallest XCD sets the pace).
    uint32_t mask[16] = {0};
    for (int i = 0; i < n_cus && i < 512; ++i) mask[i >> 5] |= 1u << (i & 31);
    e = hipExtStreamCreateWithCUMask(&s, (uint32_t)((total + 31) / 32), mask);
  }
  if (e != hipSuccess) return -3;
  *stream_out = (void*)s;
  return 0;
}

extern "C" int ruart_stream_create_priority(int priority, void** stream_out) {
  RUART_ENTRY();
  if (!stream_out) return -1;
  int least = 0, greatest = 0;                       // HIP: numerically lower = higher priority; the range is [greatest, least]
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return -2;
  if (priority > least) priority = least;
  if (priority < greatest) priority = greatest;
  hipStream_t s = nullptr;
  if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority) != hipSuccess) return -3;
  *stream_out = (void*)s;
  return priority - greatest;                        // >= 0: the level granted, counted from the highest
}

extern "C" int ruart_stream_destroy(void* stream) {
  RUART_ENTRY();
  return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? 0 : -1;
}

extern "C" int ruart_set_nan_flag(int* flag) {
  RUART_ENTRY();
  ruart_nan_flag_ptr = flag;
  return 0;
}
