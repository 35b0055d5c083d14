// Self-attention of the TRAINABLE encoder's 16-bit path (Models/Bert/modeling.py:224-250 and its backward), on the packed token
// stream: windows of whole short sequences (<= 64 word pieces per block, block-diagonal mask by the keys' first-token index), one
// workgroup of four waves per (window, head), every product on v_mfma_f32_16x16x32, "query on lane & 15" as in bert_kernels.hip.
//
//   forward   f16 operands; S^T = K . Q^T, softmax over the keys of the query's own sequence, attention-probability dropout as a
//             hash-generated multiplier on the probabilities that enter P . V (the normaliser sums the undropped ones), O^T = V^T . P^T
//   backward  f16 operands too, the incoming context gradient scaled per window by a power of two into f16's range; recomputes S and P
//             from Q, K; then
//                 dPd^T = V . dO^T       delta_q = sum_k Pd dPd       dS^T = P o (D o dPd - delta)
//                 dQ^T  = K^T . dS^T     (dS^T's accumulator tiles are the B operand directly)
//                 dV^T  = dO^T . Pd      dK^T = Q^T . dS        (sums over queries: Pd^T and dS^T pass through LDS once)
//             and writes dQ | dK | dV rows in bf16.  No atomics: every (token, head) row is written by exactly one lane.
// Sequences longer than 64 pieces: the *_long kernels at the end of this file (chunks of <= 64 tokens against the whole sequence).
#include "common.h"
#include "ruart_hip.h"

typedef __attribute__((__vector_size__(4 * sizeof(short)))) short tr16x4_t;
typedef __attribute__((address_space(3))) tr16x4_t* tr_ptr_t;
typedef __attribute__((__vector_size__(4 * sizeof(int)))) int i32x4_t;

#define ARS 144          // LDS row stride in bytes: 64 x 2 + 16 (conflict-light for the row reads and the transposed reads)

__device__ __forceinline__ unsigned attn_drop_idx(int q_tok, int key_tok, int seq_lo) { return (unsigned)q_tok * 4096u + (unsigned)(key_tok - seq_lo); }

// softmax over the 64 staged keys of one query column: s[it][r] = score(key it*16 + g*4 + r, query fr) with masked entries at
// -1e30 -> probabilities in place; returns nothing else (single tile: no running state)
__device__ __forceinline__ void softmax_cols(f32x4_t (&s)[4]) {
  constexpr float kLog2e = 1.4426950408889634f;
  float mx = -1e30f;
#pragma unroll
  for (int it = 0; it < 4; ++it)
#pragma unroll
    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[it][r]);
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = s[it][r] > -1e29f ? __builtin_amdgcn_exp2f((s[it][r] - mx) * kLog2e) : 0.f;
      s[it][r] = p;
      sum += p;
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it) s[it] *= inv;
}

template <typename T16>
__device__ __forceinline__ void pack_cols(const f32x4_t (&s)[4], typename Vec8<T16>::type (&pf)[2]) {
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pf[s2][j] = (T16)s[2 * s2][j];
      pf[s2][4 + j] = (T16)s[2 * s2 + 1][j];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_train_fwd_kernel(const f16_t* __restrict__ qkv, int ld, f16_t* __restrict__ ctx, int ldc, int H,
                                                                const int* __restrict__ bq0, const int* __restrict__ bq1,
                                                                const int* __restrict__ tok_lo, float p_drop, unsigned seed) {
  __shared__ __attribute__((aligned(16))) char Ks[64 * ARS];
  __shared__ __attribute__((aligned(16))) char Vs[64 * ARS];
  __shared__ __attribute__((aligned(16))) int Ls[64];
  typedef f16x8_t frag_t;
  const int b = blockIdx.x, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int q0 = bq0[b], n = bq1[b] - q0;
  const int qi = wave * 16 + fr, tq = q0 + qi;
  const bool qvalid = qi < n;
  const int lo = tok_lo[qvalid ? tq : q0];
  frag_t qf[2];
  {
    const f16_t* qp = qkv + (size_t)(qvalid ? tq : q0) * ld + h * 64 + g * 8;
    qf[0] = *reinterpret_cast<const frag_t*>(qp);
    qf[1] = *reinterpret_cast<const frag_t*>(qp + 32);
  }
  const int srow = tid >> 2, sc0 = (tid & 3) * 2;
  {
    uint4 kv[2], vv[2];
    if (srow < n) {
      const f16_t* kp = qkv + (size_t)(q0 + srow) * ld + H + h * 64 + sc0 * 8;
      kv[0] = *reinterpret_cast<const uint4*>(kp);
      kv[1] = *reinterpret_cast<const uint4*>(kp + 8);
      vv[0] = *reinterpret_cast<const uint4*>(kp + H);
      vv[1] = *reinterpret_cast<const uint4*>(kp + H + 8);
    } else {
      kv[0] = kv[1] = vv[0] = vv[1] = make_uint4(0, 0, 0, 0);
    }
    *reinterpret_cast<uint4*>(Ks + srow * ARS + sc0 * 16) = kv[0];
    *reinterpret_cast<uint4*>(Ks + srow * ARS + sc0 * 16 + 16) = kv[1];
    *reinterpret_cast<uint4*>(Vs + srow * ARS + sc0 * 16) = vv[0];
    *reinterpret_cast<uint4*>(Vs + srow * ARS + sc0 * 16 + 16) = vv[1];
    if (tid < 64) Ls[tid] = tid < n ? tok_lo[q0 + tid] : -1;
  }
  __syncthreads();
  f32x4_t s[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    s[it] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const frag_t kf = *reinterpret_cast<const frag_t*>(Ks + (it * 16 + fr) * ARS + (ks * 32 + g * 8) * 2);
      s[it] = mfma_16x16x32(kf, qf[ks], s[it]);
    }
    const i32x4_t lk = *reinterpret_cast<const i32x4_t*>(&Ls[it * 16 + g * 4]);
#pragma unroll
    for (int r = 0; r < 4; ++r) s[it][r] = lk[r] == lo ? s[it][r] : -1e30f;
  }
  softmax_cols(s);
  if (p_drop > 0.f) {
    const float keep_inv = 1.0f / (1.0f - p_drop);
    const unsigned sd = seed + (unsigned)h * 0x9E3779B1u;
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[it][r] *= drop_scale(sd, attn_drop_idx(tq, q0 + it * 16 + g * 4 + r, lo), p_drop, keep_inv);
  }
  frag_t pf[2];
  pack_cols<f16_t>(s, pf);
  f32x4_t o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const char* base = Vs + (32 * s2 + 4 * g + (fr >> 2)) * ARS + (dt * 16 + (fr & 3) * 4) * 2;
      union { struct { tr16x4_t a, b; } s; frag_t f; } u;
      u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)base);
      u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(base + 16 * ARS));
      o[dt] = mfma_16x16x32(u.f, pf[s2], o[dt]);
    }
  }
  if (qvalid) {
    f16_t* op = ctx + (size_t)tq * ldc + h * 64 + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store4(op + dt * 16, o[dt]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Backward.  All operands are f16 (11 significant bits; bf16's 8 put 3 % on the smallest weight-gradient norms): the context
// gradient arrives in bf16 at magnitudes of 1e-6 .. 1e-3, far below f16's normal range, so every window scales it by a power of two
// that brings its largest entry to [0.5, 1) before the conversion and divides the three results by it again on the way out.
__global__ __launch_bounds__(256, 2) void attn_train_bwd_kernel(const f16_t* __restrict__ qkv, int ld, const bf16_t* __restrict__ dctx, int ldc,
                                                                bf16_t* __restrict__ dqkv, int ldd, int H, const int* __restrict__ bq0,
                                                                const int* __restrict__ bq1, const int* __restrict__ tok_lo, float p_drop,
                                                                unsigned seed, float* __restrict__ bias_part) {
  __shared__ __attribute__((aligned(16))) char Qs[64 * ARS];
  __shared__ __attribute__((aligned(16))) char Ks[64 * ARS];
  __shared__ __attribute__((aligned(16))) char Vs[64 * ARS];
  __shared__ __attribute__((aligned(16))) char Ds[64 * ARS];      // dO * scale
  __shared__ __attribute__((aligned(16))) float bsum[2][4][64];   // per wave: column sums of its 16 tokens' dQ and dV rows
  // Pd^T and dS^T ([key][query] images for the two products that sum over queries) take over the V and K images once every wave is
  // done with them: 37 KB of LDS per workgroup instead of 55 KB - four workgroups per CU instead of two
  char* const PT = Vs;
  char* const ST = Ks;
  __shared__ __attribute__((aligned(16))) int Ls[64];
  __shared__ float red[4];
  typedef f16x8_t frag_t;
  const int b = blockIdx.x, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int q0 = bq0[b], n = bq1[b] - q0;
  const int qi = wave * 16 + fr, tq = q0 + qi;
  const bool qvalid = qi < n;
  const int lo = tok_lo[qvalid ? tq : q0];
  // stage Q, K, V (f16) and the scaled dO rows of the window; rows past its end are zero
  const int srow = tid >> 2, sc0 = (tid & 3) * 2;
  float dsc;                                                   // the window's power-of-two scale of dO
  {
    uint4 q4[2], k4[2], v4[2];
    f32x4_t d4[4];
    float mx = 0.f;
    if (srow < n) {
      const f16_t* qp = qkv + (size_t)(q0 + srow) * ld + h * 64 + sc0 * 8;
      const bf16_t* dp = dctx + (size_t)(q0 + srow) * ldc + h * 64 + sc0 * 8;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        q4[c] = *reinterpret_cast<const uint4*>(qp + c * 8);
        k4[c] = *reinterpret_cast<const uint4*>(qp + H + c * 8);
        v4[c] = *reinterpret_cast<const uint4*>(qp + 2 * H + c * 8);
        d4[2 * c] = load4(dp + c * 8);
        d4[2 * c + 1] = load4(dp + c * 8 + 4);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, fabsf(d4[i][r]));
    } else {
      q4[0] = q4[1] = k4[0] = k4[1] = v4[0] = v4[1] = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) d4[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      *reinterpret_cast<uint4*>(Qs + srow * ARS + (sc0 + c) * 16) = q4[c];
      *reinterpret_cast<uint4*>(Ks + srow * ARS + (sc0 + c) * 16) = k4[c];
      *reinterpret_cast<uint4*>(Vs + srow * ARS + (sc0 + c) * 16) = v4[c];
    }
    if (tid < 64) Ls[tid] = tid < n ? tok_lo[q0 + tid] : -1;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    int e = 0;
    if (mx > 0.f) frexpf(mx, &e);                              // mx = m * 2^e, m in [0.5, 1)
    dsc = ldexpf(1.0f, -e);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      f16x8_t v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = (f16_t)(d4[2 * c][r] * dsc);
        v[4 + r] = (f16_t)(d4[2 * c + 1][r] * dsc);
      }
      *reinterpret_cast<f16x8_t*>(Ds + srow * ARS + (sc0 + c) * 16) = v;
    }
  }
  __syncthreads();
  const float inv_dsc = 1.0f / dsc;
  // this wave's query fragments (rows qi of Q and dO): k-step ks covers d = ks*32 + g*8 .. +7
  frag_t qf[2], df[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    qf[ks] = *reinterpret_cast<const frag_t*>(Qs + qi * ARS + (ks * 32 + g * 8) * 2);
    df[ks] = *reinterpret_cast<const frag_t*>(Ds + qi * ARS + (ks * 32 + g * 8) * 2);
  }
  f32x4_t s[4], dp[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    s[it] = dp[it] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const frag_t kf = *reinterpret_cast<const frag_t*>(Ks + (it * 16 + fr) * ARS + (ks * 32 + g * 8) * 2);
      const frag_t vf = *reinterpret_cast<const frag_t*>(Vs + (it * 16 + fr) * ARS + (ks * 32 + g * 8) * 2);
      s[it] = mfma_16x16x32(kf, qf[ks], s[it]);
      dp[it] = mfma_16x16x32(vf, df[ks], dp[it]);           // dPd^T[key][q] = sum_d V[key][d] dO[q][d]
    }
    const i32x4_t lk = *reinterpret_cast<const i32x4_t*>(&Ls[it * 16 + g * 4]);
#pragma unroll
    for (int r = 0; r < 4; ++r) s[it][r] = (lk[r] == lo && qvalid) ? s[it][r] : -1e30f;
  }
  softmax_cols(s);                                            // s = P (zero for masked keys and for lanes without a query)
  // D o dPd, delta, dS; Pd for the value gradient
  const float keep_inv = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
  const unsigned sd = seed + (unsigned)h * 0x9E3779B1u;
  f32x4_t pd[4];
  float delta = 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float D = p_drop > 0.f ? drop_scale(sd, attn_drop_idx(tq, q0 + it * 16 + g * 4 + r, lo), p_drop, keep_inv) : 1.0f;
      pd[it][r] = s[it][r] * D;
      dp[it][r] *= D;                                        // dP = D o dPd
      delta += s[it][r] * dp[it][r];
    }
  delta += __shfl_xor(delta, 16, 64);
  delta += __shfl_xor(delta, 32, 64);
#pragma unroll
  for (int it = 0; it < 4; ++it)
#pragma unroll
    for (int r = 0; r < 4; ++r) s[it][r] *= dp[it][r] - delta;     // s = dS^T (in units of the scaled dO)
  // dQ^T = K^T . dS^T with the dS^T accumulator tiles as the B operand (permuted key order, matched by the transposed reads)
  frag_t sf[2];
  pack_cols<f16_t>(s, sf);
  f32x4_t dq[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    dq[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const char* base = Ks + (32 * s2 + 4 * g + (fr >> 2)) * ARS + (dt * 16 + (fr & 3) * 4) * 2;
      union { struct { tr16x4_t a, b; } s; frag_t f; } u;
      u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)base);
      u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(base + 16 * ARS));
      dq[dt] = mfma_16x16x32(u.f, sf[s2], dq[dt]);
    }
  }
  if (qvalid) {
    bf16_t* op = dqkv + (size_t)tq * ldd + h * 64 + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store4(op + dt * 16, dq[dt] * inv_dsc);
  }
  // bias gradients: the window's column sums of dQ (here) and dV (below), before rounding; tokens past the window's end carry zeros.
  // Lane (fr, g) holds columns dt*16 + g*4 .. +3 of token fr: sum over the 16 tokens of the wave by shuffles, over the waves in LDS.
  auto colsum16 = [&](const f32x4_t (&v)[4], int which) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4_t t = v[dt];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] += __shfl_xor(t[r], m, 64);
      if (fr == 0) *reinterpret_cast<f32x4_t*>(&bsum[which][wave][dt * 16 + g * 4]) = t;
    }
  };
  if (bias_part) colsum16(dq, 0);
  __syncthreads();                                            // every wave has read K (dQ) and V (dPd) for the last time
  // Pd^T and dS^T -> LDS [key][query] for the products that sum over queries
#pragma unroll
  for (int it = 0; it < 4; ++it)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = it * 16 + g * 4 + r;
      *reinterpret_cast<f16_t*>(PT + key * ARS + qi * 2) = (f16_t)pd[it][r];
      *reinterpret_cast<f16_t*>(ST + key * ARS + qi * 2) = (f16_t)s[it][r];
    }
  __syncthreads();
  // dV^T[d][key] = sum_q dO[q][d] Pd[q][key],  dK^T[d][key] = sum_q Q[q][d] dS[q][key]  for this wave's 16 keys (key = qi)
  f32x4_t dv[4], dk[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    dv[dt] = dk[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      // B operands: 8 queries of key qi in the order the transposed A fragment uses: 32*s2 + 4g + {0..3} and + 16
      union { struct { unsigned long long a, b; } s; frag_t f; } pb, sb;
      pb.s.a = *reinterpret_cast<const unsigned long long*>(PT + qi * ARS + (32 * s2 + 4 * g) * 2);
      pb.s.b = *reinterpret_cast<const unsigned long long*>(PT + qi * ARS + (32 * s2 + 16 + 4 * g) * 2);
      sb.s.a = *reinterpret_cast<const unsigned long long*>(ST + qi * ARS + (32 * s2 + 4 * g) * 2);
      sb.s.b = *reinterpret_cast<const unsigned long long*>(ST + qi * ARS + (32 * s2 + 16 + 4 * g) * 2);
      const int off = (32 * s2 + 4 * g + (fr >> 2)) * ARS + (dt * 16 + (fr & 3) * 4) * 2;
      union { struct { tr16x4_t a, b; } s; frag_t f; } ud, uq;
      ud.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(Ds + off));
      ud.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(Ds + off + 16 * ARS));
      uq.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(Qs + off));
      uq.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(Qs + off + 16 * ARS));
      dv[dt] = mfma_16x16x32(ud.f, pb.f, dv[dt]);
      dk[dt] = mfma_16x16x32(uq.f, sb.f, dk[dt]);
    }
  }
  if (qvalid) {                                               // (token qi as a KEY: the window's queries and keys are the same tokens)
    bf16_t* kp = dqkv + (size_t)tq * ldd + H + h * 64 + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      store4(kp + dt * 16, dk[dt] * inv_dsc);
      store4(kp + H + dt * 16, dv[dt] * inv_dsc);
    }
  }
  if (bias_part) {
    colsum16(dv, 1);
    __syncthreads();
    if (tid < 128) {                                          // 64 columns of dQ, 64 of dV: the four waves' sums in wave order
      const int which = tid >> 6, c = tid & 63;
      const float t = ((bsum[which][0][c] + bsum[which][1][c]) + (bsum[which][2][c] + bsum[which][3][c])) * inv_dsc;
      bias_part[(size_t)b * 2 * H + which * H + h * 64 + c] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Sequences longer than one window (65 .. 512 word pieces: a question or a caption; Models/Bert/Bert.py:96-99 windows longer inputs).
// The host cuts such a sequence into CHUNKS of <= 64 consecutive tokens; every kernel below takes one workgroup per (chunk, head):
//   forward   the chunk's queries against the key tiles of the whole sequence, online softmax (running maximum / sum in fp32, the
//             dropped probabilities enter P . V, the undropped ones the normaliser), and the row's log2-sum-exp for the backward
//   dQ        the chunk's queries again, two passes over the key tiles: delta_q = sum_k P (D o dPd) with P = exp2(s - lse), then
//             dS^T = P o (D o dPd - delta), dQ^T += K^T . dS^T; leaves delta and the chunk's dO scale for the third
//   dK / dV   the chunk's tokens as KEYS against the query tiles of the whole sequence: P^T and dS^T per tile through LDS as in the
//             window kernel, dV^T += dO^T . Pd, dK^T += Q^T . dS - every (token, head) row is still written by exactly one lane of
//             one workgroup, so there are no atomics and the bits repeat.
// Same operand types, same dropout hash (query token, key offset inside the sequence), same power-of-two scaling of dO as the window
// kernels.  chunk arrays: cq0 / cq1 = the chunk's tokens, ck0 / ck1 = its sequence, cfirst = index of the sequence's first chunk
// (the chunks of a sequence are consecutive and start at its first token in steps of 64).
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void stage_rows16(char* img, const f16_t* src, int ld, int row0, int n_rows, int srow, int sc0) {
  uint4 v[2];
  if (srow < n_rows) {
    const f16_t* p = src + (size_t)(row0 + srow) * ld + sc0 * 8;
    v[0] = *reinterpret_cast<const uint4*>(p);
    v[1] = *reinterpret_cast<const uint4*>(p + 8);
  } else {
    v[0] = v[1] = make_uint4(0, 0, 0, 0);
  }
  *reinterpret_cast<uint4*>(img + srow * ARS + sc0 * 16) = v[0];
  *reinterpret_cast<uint4*>(img + srow * ARS + sc0 * 16 + 16) = v[1];
}

__global__ __launch_bounds__(256, 2) void attn_train_fwd_long_kernel(const f16_t* __restrict__ qkv, int ld, f16_t* __restrict__ ctx, int ldc, int H,
                                                                     int nh, const int* __restrict__ cq0, const int* __restrict__ cq1,
                                                                     const int* __restrict__ ck0, const int* __restrict__ ck1, float p_drop,
                                                                     unsigned seed, float* __restrict__ lse2) {
  __shared__ __attribute__((aligned(16))) char Ks[64 * ARS];
  __shared__ __attribute__((aligned(16))) char Vs[64 * ARS];
  typedef f16x8_t frag_t;
  constexpr float kLog2e = 1.4426950408889634f;
  const int b = blockIdx.x, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int q0 = cq0[b], n = cq1[b] - q0, k0 = ck0[b], k1 = ck1[b];
  const int qi = wave * 16 + fr, tq = q0 + qi;
  const bool qvalid = qi < n;
  frag_t qf[2];
  {
    const f16_t* qp = qkv + (size_t)(qvalid ? tq : q0) * ld + h * 64 + g * 8;
    qf[0] = *reinterpret_cast<const frag_t*>(qp);
    qf[1] = *reinterpret_cast<const frag_t*>(qp + 32);
  }
  const int srow = tid >> 2, sc0 = (tid & 3) * 2;
  const float keep_inv = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
  const unsigned sd = seed + (unsigned)h * 0x9E3779B1u;
  float m = -1e30f, l = 0.f;
  f32x4_t o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int kt = k0; kt < k1; kt += 64) {
    const int tn = min(64, k1 - kt);
    __syncthreads();                                            // the previous tile's images are dead
    stage_rows16(Ks, qkv + H + h * 64, ld, kt, tn, srow, sc0);
    stage_rows16(Vs, qkv + 2 * H + h * 64, ld, kt, tn, srow, sc0);
    __syncthreads();
    f32x4_t s[4];
    float mx = -1e30f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      s[it] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const frag_t kf = *reinterpret_cast<const frag_t*>(Ks + (it * 16 + fr) * ARS + (ks * 32 + g * 8) * 2);
        s[it] = mfma_16x16x32(kf, qf[ks], s[it]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[it][r] = (it * 16 + g * 4 + r) < tn ? s[it][r] : -1e30f;
        mx = fmaxf(mx, s[it][r]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f((m - mn) * kLog2e);
    float rs = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = s[it][r] > -1e29f ? __builtin_amdgcn_exp2f((s[it][r] - mn) * kLog2e) : 0.f;
        rs += p;
        s[it][r] = p_drop > 0.f ? p * drop_scale(sd, attn_drop_idx(tq, kt + it * 16 + g * 4 + r, k0), p_drop, keep_inv) : p;
      }
    rs += __shfl_xor(rs, 16, 64);
    rs += __shfl_xor(rs, 32, 64);
    l = l * alpha + rs;
    m = mn;
    frag_t pf[2];
    pack_cols<f16_t>(s, pf);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      o[dt] *= alpha;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const char* base = Vs + (32 * s2 + 4 * g + (fr >> 2)) * ARS + (dt * 16 + (fr & 3) * 4) * 2;
        union { struct { tr16x4_t a, b; } s; frag_t f; } u;
        u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)base);
        u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(base + 16 * ARS));
        o[dt] = mfma_16x16x32(u.f, pf[s2], o[dt]);
      }
    }
  }
  if (qvalid) {
    const float inv = 1.0f / l;
    f16_t* op = ctx + (size_t)tq * ldc + h * 64 + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store4(op + dt * 16, o[dt] * inv);
    if (g == 0) lse2[(size_t)tq * nh + h] = m * kLog2e + __builtin_amdgcn_logf(l);     // log2 of the row's sum of exp(score)
  }
}

// stage the chunk's dO rows scaled by the power of two that brings their largest entry to [0.5, 1) (returns it; `red`: 4 floats of
// LDS), or by `fixed_scale` when that is non-zero (the dK / dV kernel re-uses the scale the dQ kernel chose for the same rows)
__device__ __forceinline__ float stage_scaled_dO(char* Ds, const bf16_t* __restrict__ dctx, int ldc, int row0, int n_rows, int col0, int srow,
                                                 int sc0, int lane, int wave, float* red, float fixed_scale) {
  f32x4_t d4[4];
  float mx = 0.f;
  if (srow < n_rows) {
    const bf16_t* dp = dctx + (size_t)(row0 + srow) * ldc + col0 + sc0 * 8;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      d4[2 * c] = load4(dp + c * 8);
      d4[2 * c + 1] = load4(dp + c * 8 + 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, fabsf(d4[i][r]));
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) d4[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  float dsc = fixed_scale;
  if (fixed_scale == 0.f) {
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    int e = 0;
    if (mx > 0.f) frexpf(mx, &e);                              // mx = m * 2^e, m in [0.5, 1)
    dsc = ldexpf(1.0f, -e);
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    f16x8_t v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = (f16_t)(d4[2 * c][r] * dsc);
      v[4 + r] = (f16_t)(d4[2 * c + 1][r] * dsc);
    }
    *reinterpret_cast<f16x8_t*>(Ds + srow * ARS + (sc0 + c) * 16) = v;
  }
  return dsc;
}

__global__ __launch_bounds__(256, 2) void attn_train_bwd_long_dq_kernel(const f16_t* __restrict__ qkv, int ld, const bf16_t* __restrict__ dctx, int ldc,
                                                                        bf16_t* __restrict__ dqkv, int ldd, int H, int nh, const int* __restrict__ cq0, const int* __restrict__ cq1,
                                                                        const int* __restrict__ ck0, const int* __restrict__ ck1, float p_drop,
                                                                        unsigned seed, const float* __restrict__ lse2, float* __restrict__ delta_out,
                                                                        float* __restrict__ dsc_out, float* __restrict__ bias_part) {
  __shared__ __attribute__((aligned(16))) char Qs[64 * ARS];
  __shared__ __attribute__((aligned(16))) char Ds[64 * ARS];
  __shared__ __attribute__((aligned(16))) char Ks[64 * ARS];
  __shared__ __attribute__((aligned(16))) char Vs[64 * ARS];
  __shared__ __attribute__((aligned(16))) float bsum[4][64];
  __shared__ float red[4];
  typedef f16x8_t frag_t;
  constexpr float kLog2e = 1.4426950408889634f;
  const int b = blockIdx.x, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int q0 = cq0[b], n = cq1[b] - q0, k0 = ck0[b], k1 = ck1[b];
  const int qi = wave * 16 + fr, tq = q0 + qi;
  const bool qvalid = qi < n;
  const int srow = tid >> 2, sc0 = (tid & 3) * 2;
  stage_rows16(Qs, qkv + h * 64, ld, q0, n, srow, sc0);
  const float dsc = stage_scaled_dO(Ds, dctx, ldc, q0, n, h * 64, srow, sc0, lane, wave, red, 0.f);
  __syncthreads();
  const float inv_dsc = 1.0f / dsc;
  frag_t qf[2], df[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    qf[ks] = *reinterpret_cast<const frag_t*>(Qs + qi * ARS + (ks * 32 + g * 8) * 2);
    df[ks] = *reinterpret_cast<const frag_t*>(Ds + qi * ARS + (ks * 32 + g * 8) * 2);
  }
  const float lse = qvalid ? lse2[(size_t)tq * nh + h] : 0.f;
  const float keep_inv = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
  const unsigned sd = seed + (unsigned)h * 0x9E3779B1u;
  // Pass 1 over the key tiles: delta_q = sum_k P (D o dPd), from the SAME rounded operands the second pass forms dS with, so that
  // sum_k dS_qk = 0 holds to fp32 rounding.  (<dO_q, O_q> with the forward's saved f16 context rows is the same number in exact
  // arithmetic, but its rounding differs from the recomputed products' by ~2^-11 dPd: a common-mode error P_qk eps in every dS_qk of
  // the row, which dQ = dS . K and dK = dS^T . Q multiply by the MEAN key / query - the query- and key-weight gradient norms of the
  // reference's long-question pass came out 1.6x too large that way.)
  float delta = 0.f;
  for (int kt = k0; kt < k1; kt += 64) {
    const int tn = min(64, k1 - kt);
    if (kt != k0) __syncthreads();
    stage_rows16(Ks, qkv + H + h * 64, ld, kt, tn, srow, sc0);
    stage_rows16(Vs, qkv + 2 * H + h * 64, ld, kt, tn, srow, sc0);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      f32x4_t sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const frag_t kf = *reinterpret_cast<const frag_t*>(Ks + (it * 16 + fr) * ARS + (ks * 32 + g * 8) * 2);
        const frag_t vf = *reinterpret_cast<const frag_t*>(Vs + (it * 16 + fr) * ARS + (ks * 32 + g * 8) * 2);
        sc = mfma_16x16x32(kf, qf[ks], sc);
        dp = mfma_16x16x32(vf, df[ks], dp);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = it * 16 + g * 4 + r;
        const float P = (key < tn && qvalid) ? __builtin_amdgcn_exp2f(sc[r] * kLog2e - lse) : 0.f;
        const float D = p_drop > 0.f ? drop_scale(sd, attn_drop_idx(tq, kt + key, k0), p_drop, keep_inv) : 1.0f;
        delta += P * (D * dp[r]);
      }
    }
  }
  delta += __shfl_xor(delta, 16, 64);
  delta += __shfl_xor(delta, 32, 64);
  f32x4_t dq[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int kt = k0; kt < k1; kt += 64) {
    const int tn = min(64, k1 - kt);
    __syncthreads();                                            // the previous tile's images are dead
    stage_rows16(Ks, qkv + H + h * 64, ld, kt, tn, srow, sc0);
    stage_rows16(Vs, qkv + 2 * H + h * 64, ld, kt, tn, srow, sc0);
    __syncthreads();
    f32x4_t s[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      f32x4_t sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const frag_t kf = *reinterpret_cast<const frag_t*>(Ks + (it * 16 + fr) * ARS + (ks * 32 + g * 8) * 2);
        const frag_t vf = *reinterpret_cast<const frag_t*>(Vs + (it * 16 + fr) * ARS + (ks * 32 + g * 8) * 2);
        sc = mfma_16x16x32(kf, qf[ks], sc);
        dp = mfma_16x16x32(vf, df[ks], dp);                    // dPd^T[key][q] = sum_d V[key][d] dO[q][d]
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = it * 16 + g * 4 + r;
        const float P = (key < tn && qvalid) ? __builtin_amdgcn_exp2f(sc[r] * kLog2e - lse) : 0.f;
        const float D = p_drop > 0.f ? drop_scale(sd, attn_drop_idx(tq, kt + key, k0), p_drop, keep_inv) : 1.0f;
        s[it][r] = P * (D * dp[r] - delta);                    // dS^T (in units of the scaled dO)
      }
    }
    frag_t sf[2];
    pack_cols<f16_t>(s, sf);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const char* base = Ks + (32 * s2 + 4 * g + (fr >> 2)) * ARS + (dt * 16 + (fr & 3) * 4) * 2;
        union { struct { tr16x4_t a, b; } s; frag_t f; } u;
        u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)base);
        u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(base + 16 * ARS));
        dq[dt] = mfma_16x16x32(u.f, sf[s2], dq[dt]);
      }
  }
  if (qvalid) {
    bf16_t* op = dqkv + (size_t)tq * ldd + h * 64 + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store4(op + dt * 16, dq[dt] * inv_dsc);
    if (g == 0) delta_out[(size_t)tq * nh + h] = delta;         // (scaled units: the dK / dV kernel scales dO by the same dsc)
  }
  if (tid == 0) dsc_out[(size_t)b * nh + h] = dsc;
  if (bias_part) {                                              // the chunk's column sums of dQ (query bias gradient), before rounding
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4_t t = dq[dt];
#pragma unroll
      for (int mm = 1; mm < 16; mm <<= 1)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] += __shfl_xor(t[r], mm, 64);
      if (fr == 0) *reinterpret_cast<f32x4_t*>(&bsum[wave][dt * 16 + g * 4]) = t;
    }
    __syncthreads();
    if (tid < 64) bias_part[(size_t)b * 2 * H + h * 64 + tid] = ((bsum[0][tid] + bsum[1][tid]) + (bsum[2][tid] + bsum[3][tid])) * inv_dsc;
  }
}

__global__ __launch_bounds__(256, 2) void attn_train_bwd_long_dkv_kernel(const f16_t* __restrict__ qkv, int ld, const bf16_t* __restrict__ dctx, int ldc,
                                                                         bf16_t* __restrict__ dqkv, int ldd, int H, int nh,
                                                                         const int* __restrict__ cq0, const int* __restrict__ cq1,
                                                                         const int* __restrict__ ck0, const int* __restrict__ ck1,
                                                                         const int* __restrict__ cfirst, float p_drop, unsigned seed,
                                                                         const float* __restrict__ lse2, const float* __restrict__ delta_in,
                                                                         const float* __restrict__ dsc_in, float* __restrict__ bias_part) {
  __shared__ __attribute__((aligned(16))) char Ks[64 * ARS];      // the chunk's tokens as keys: staged once
  __shared__ __attribute__((aligned(16))) char Vs[64 * ARS];
  __shared__ __attribute__((aligned(16))) char Qs[64 * ARS];      // one query tile of the sequence at a time
  __shared__ __attribute__((aligned(16))) char Ds[64 * ARS];
  __shared__ __attribute__((aligned(16))) char PT[64 * ARS];      // Pd^T and dS^T of the tile, [key][query]
  __shared__ __attribute__((aligned(16))) char ST[64 * ARS];
  __shared__ __attribute__((aligned(16))) float bsum[4][64];
  typedef f16x8_t frag_t;
  constexpr float kLog2e = 1.4426950408889634f;
  const int b = blockIdx.x, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int kq0 = cq0[b], nk = cq1[b] - kq0, s0 = ck0[b], s1 = ck1[b], first = cfirst[b];
  const int qi = wave * 16 + fr;                                  // query index inside a tile; ALSO this lane's key for dK / dV
  const int srow = tid >> 2, sc0 = (tid & 3) * 2;
  stage_rows16(Ks, qkv + H + h * 64, ld, kq0, nk, srow, sc0);
  stage_rows16(Vs, qkv + 2 * H + h * 64, ld, kq0, nk, srow, sc0);
  const float keep_inv = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
  const unsigned sd = seed + (unsigned)h * 0x9E3779B1u;
  f32x4_t dv[4], dk[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dv[dt] = dk[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  int t = 0;
  for (int qt = s0; qt < s1; qt += 64, ++t) {
    const int nq = min(64, s1 - qt);
    const float dsc = dsc_in[(size_t)(first + t) * nh + h];
    const float inv_dsc = 1.0f / dsc;
    __syncthreads();                                              // the previous tile's Qs / Ds / PT / ST reads are done
    stage_rows16(Qs, qkv + h * 64, ld, qt, nq, srow, sc0);
    stage_scaled_dO(Ds, dctx, ldc, qt, nq, h * 64, srow, sc0, lane, wave, nullptr, dsc);
    __syncthreads();
    const int tq = qt + qi;
    const bool qvalid = qi < nq;
    frag_t qf[2], df[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[ks] = *reinterpret_cast<const frag_t*>(Qs + qi * ARS + (ks * 32 + g * 8) * 2);
      df[ks] = *reinterpret_cast<const frag_t*>(Ds + qi * ARS + (ks * 32 + g * 8) * 2);
    }
    const float lse = qvalid ? lse2[(size_t)tq * nh + h] : 0.f;
    const float delta = qvalid ? delta_in[(size_t)tq * nh + h] : 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      f32x4_t sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const frag_t kf = *reinterpret_cast<const frag_t*>(Ks + (it * 16 + fr) * ARS + (ks * 32 + g * 8) * 2);
        const frag_t vf = *reinterpret_cast<const frag_t*>(Vs + (it * 16 + fr) * ARS + (ks * 32 + g * 8) * 2);
        sc = mfma_16x16x32(kf, qf[ks], sc);
        dp = mfma_16x16x32(vf, df[ks], dp);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = it * 16 + g * 4 + r;
        const float P = (key < nk && qvalid) ? __builtin_amdgcn_exp2f(sc[r] * kLog2e - lse) : 0.f;
        const float D = p_drop > 0.f ? drop_scale(sd, attn_drop_idx(tq, kq0 + key, s0), p_drop, keep_inv) : 1.0f;
        *reinterpret_cast<f16_t*>(PT + key * ARS + qi * 2) = (f16_t)(P * D);
        *reinterpret_cast<f16_t*>(ST + key * ARS + qi * 2) = (f16_t)(P * (D * dp[r] - delta));
      }
    }
    __syncthreads();
    // dV^T[d][key] += sum_q dO[q][d] Pd[q][key],  dK^T[d][key] += sum_q Q[q][d] dS[q][key]  for this wave's 16 keys (key = qi)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4_t tv = {0.f, 0.f, 0.f, 0.f}, tk = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        union { struct { unsigned long long a, b; } s; frag_t f; } pb, sb;
        pb.s.a = *reinterpret_cast<const unsigned long long*>(PT + qi * ARS + (32 * s2 + 4 * g) * 2);
        pb.s.b = *reinterpret_cast<const unsigned long long*>(PT + qi * ARS + (32 * s2 + 16 + 4 * g) * 2);
        sb.s.a = *reinterpret_cast<const unsigned long long*>(ST + qi * ARS + (32 * s2 + 4 * g) * 2);
        sb.s.b = *reinterpret_cast<const unsigned long long*>(ST + qi * ARS + (32 * s2 + 16 + 4 * g) * 2);
        const int off = (32 * s2 + 4 * g + (fr >> 2)) * ARS + (dt * 16 + (fr & 3) * 4) * 2;
        union { struct { tr16x4_t a, b; } s; frag_t f; } ud, uq;
        ud.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(Ds + off));
        ud.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(Ds + off + 16 * ARS));
        uq.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(Qs + off));
        uq.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(Qs + off + 16 * ARS));
        tv = mfma_16x16x32(ud.f, pb.f, tv);
        tk = mfma_16x16x32(uq.f, sb.f, tk);
      }
      dv[dt] += tv * inv_dsc;                                     // every query tile carries its own dO scale
      dk[dt] += tk * inv_dsc;
    }
  }
  if (qi < nk) {
    bf16_t* kp = dqkv + (size_t)(kq0 + qi) * ldd + H + h * 64 + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      store4(kp + dt * 16, dk[dt]);
      store4(kp + H + dt * 16, dv[dt]);
    }
  }
  if (bias_part) {                                                // the chunk's column sums of dV (value bias gradient); keys past its end hold zeros
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4_t tt = dv[dt];
#pragma unroll
      for (int mm = 1; mm < 16; mm <<= 1)
#pragma unroll
        for (int r = 0; r < 4; ++r) tt[r] += __shfl_xor(tt[r], mm, 64);
      if (fr == 0) *reinterpret_cast<f32x4_t*>(&bsum[wave][dt * 16 + g * 4]) = tt;
    }
    __syncthreads();
    if (tid < 64) bias_part[(size_t)b * 2 * H + H + h * 64 + tid] = (bsum[0][tid] + bsum[1][tid]) + (bsum[2][tid] + bsum[3][tid]);
  }
}

extern "C" int ruart_attn_train_fwd(const void* qkv16, int ld, void* ctx16, int ldc, int H, int n_heads, int n_blocks, const int* blk_q0,
                                    const int* blk_q1, const int* tok_lo, float p_drop, unsigned seed, void* stream) {
  RUART_ENTRY();
  if (n_heads * 64 != H || n_blocks <= 0 || (ld & 7) || (ldc & 3) || p_drop < 0.f || p_drop >= 1.f) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(attn_train_fwd_kernel, dim3(n_blocks, n_heads), dim3(256), 0, (hipStream_t)stream, (const f16_t*)qkv16, ld, (f16_t*)ctx16, ldc,
                     H, blk_q0, blk_q1, tok_lo, p_drop, seed);
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" int ruart_attn_train_bwd(const void* qkv16, int ld, const void* dctx_bf16, int ldc, void* dqkv_bf16, int ldd, int H, int n_heads,
                                    int n_blocks, const int* blk_q0, const int* blk_q1, const int* tok_lo, float p_drop, unsigned seed,
                                    float* bias_part, void* stream) {
  RUART_ENTRY();
  if (n_heads * 64 != H || n_blocks <= 0 || (ld & 7) || (ldc & 7) || (ldd & 3) || p_drop < 0.f || p_drop >= 1.f) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(attn_train_bwd_kernel, dim3(n_blocks, n_heads), dim3(256), 0, (hipStream_t)stream, (const f16_t*)qkv16, ld,
                     (const bf16_t*)dctx_bf16, ldc, (bf16_t*)dqkv_bf16, ldd, H, blk_q0, blk_q1, tok_lo, p_drop, seed, bias_part);
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" int ruart_attn_train_fwd_long(const void* qkv16, int ld, void* ctx16, int ldc, int H, int n_heads, int n_chunks, const int* chunk_q0,
                                         const int* chunk_q1, const int* chunk_k0, const int* chunk_k1, float p_drop, unsigned seed, float* lse2,
                                         void* stream) {
  RUART_ENTRY();
  if (n_heads * 64 != H || n_chunks <= 0 || (ld & 7) || (ldc & 3) || p_drop < 0.f || p_drop >= 1.f || !lse2) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(attn_train_fwd_long_kernel, dim3(n_chunks, n_heads), dim3(256), 0, (hipStream_t)stream, (const f16_t*)qkv16, ld, (f16_t*)ctx16,
                     ldc, H, n_heads, chunk_q0, chunk_q1, chunk_k0, chunk_k1, p_drop, seed, lse2);
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" int ruart_attn_train_bwd_long(const void* qkv16, int ld, const void* dctx_bf16, int ldc, void* dqkv_bf16, int ldd, int H, int n_heads, int n_chunks, const int* chunk_q0, const int* chunk_q1, const int* chunk_k0,
                                         const int* chunk_k1, const int* chunk_first, float p_drop, unsigned seed, const float* lse2, float* delta_ws,
                                         float* scale_ws, float* bias_part, void* stream) {
  RUART_ENTRY();
  if (n_heads * 64 != H || n_chunks <= 0 || (ld & 7) || (ldc & 7) || (ldd & 3) || p_drop < 0.f || p_drop >= 1.f || !lse2 || !delta_ws || !scale_ws)
    return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(attn_train_bwd_long_dq_kernel, dim3(n_chunks, n_heads), dim3(256), 0, (hipStream_t)stream, (const f16_t*)qkv16, ld,
                     (const bf16_t*)dctx_bf16, ldc, (bf16_t*)dqkv_bf16, ldd, H, n_heads, chunk_q0, chunk_q1, chunk_k0, chunk_k1, p_drop, seed, lse2, delta_ws,
                     scale_ws, bias_part);
  RUART_CHECK_LAUNCH();
  hipLaunchKernelGGL(attn_train_bwd_long_dkv_kernel, dim3(n_chunks, n_heads), dim3(256), 0, (hipStream_t)stream, (const f16_t*)qkv16, ld,
                     (const bf16_t*)dctx_bf16, ldc, (bf16_t*)dqkv_bf16, ldd, H, n_heads, chunk_q0, chunk_q1, chunk_k0, chunk_k1, chunk_first, p_drop, seed,
                     lse2, delta_ws, scale_ws, bias_part);
  RUART_CHECK_LAUNCH();
  return 0;
}
