// Persistent LSTM recurrence for the SDNet trunk (fp32).
//
// Reference: Models/Layers.py:166 `self.rnns[i](rnn_input)[0]` -> torch nn.LSTM(batch_first, 1 layer, (bi)directional),
// gate order i, f, g, o, zero initial state, padding NOT masked.  Seven instances, hidden 125
// (Models/SDNet.py:147, 150, 172, 188; Models/Layers.py:489), T up to 100 sequential steps: latency-bound.
//
// The input projection x W_ih^T + b_ih + b_hh for all time steps and both directions is ONE plain GEMM done by the
// caller (xproj).  This file is the sequential part: one workgroup per (batch row, direction) keeps its direction's
// W_hh (4h x h, <= 512 x 128 fp32 = 256 KB) entirely in VGPRs - 128 weights per thread - for the whole sequence, the
// running h in LDS (broadcast reads), c in a register.  Two barriers per step, no global traffic for weights.
// Backward (BPTT) is the mirror image with W_hh^T columns in VGPRs; it emits grad_xproj, from which the caller gets
// grad_W_ih, grad_b, grad_x and grad_W_hh with three plain GEMMs.
#include "common.h"
#include "ruart_hip.h"

#define HP 128      // padded hidden size (h <= 128)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(512) void lstm_fwd_kernel(const float* __restrict__ xproj, const float* __restrict__ w_hh,
                                                       float* __restrict__ y, float* __restrict__ gates, float* __restrict__ cells,
                                                       float* __restrict__ hprev, int T, int h, int ndir, int* __restrict__ nan_flag) {
  __shared__ __attribute__((aligned(16))) float h_s[HP];
  __shared__ float pre_s[4 * HP];
  const int b = blockIdx.x, d = blockIdx.y, g = threadIdx.x;
  const int G = 4 * h;
  const bool row = g < G;
  float w[HP];
  {
    const float* wr = w_hh + ((size_t)d * G + (row ? g : 0)) * h;
#pragma unroll
    for (int j = 0; j < HP; ++j) w[j] = (row && j < h) ? wr[j] : 0.f;
  }
  if (g < HP) h_s[g] = 0.f;
  float c = 0.f;
  const size_t ldx = (size_t)ndir * G, ldy = (size_t)ndir * h;
  const float* xp = xproj + (size_t)b * T * ldx + (size_t)d * G + (row ? g : 0);
  const int t0 = d ? T - 1 : 0, dt = d ? -1 : 1;
  float xnext = row ? xp[(size_t)t0 * ldx] : 0.f;
  __syncthreads();
  bool bad = false;
  for (int s = 0, t = t0; s < T; ++s, t += dt) {
    float pre = xnext;
    if (s + 1 < T && row) xnext = xp[(size_t)(t + dt) * ldx];
    {
      // four independent partial sums: the 128-long dot product is the step's longest dependent chain (one wave per SIMD pair
      // cannot hide a 4-cycle FMA latency 128 times), so it is cut to 32 deep; fixed combination order keeps it deterministic
      float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
      // software-pipelined by hand: the next four 16-byte LDS reads are issued before the 16 FMAs of the current four
      f32x4_t ha[4], hb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) ha[u] = *reinterpret_cast<const f32x4_t*>(h_s + 4 * u);
#pragma unroll
      for (int jb = 0; jb < HP; jb += 32) {
#pragma unroll
        for (int u = 0; u < 4; ++u) hb[u] = *reinterpret_cast<const f32x4_t*>(h_s + jb + 16 + 4 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = jb + 4 * u;
          p0 = fmaf(w[j], ha[u][0], p0); p1 = fmaf(w[j + 1], ha[u][1], p1);
          p2 = fmaf(w[j + 2], ha[u][2], p2); p3 = fmaf(w[j + 3], ha[u][3], p3);
        }
        if (jb + 32 < HP) {
#pragma unroll
          for (int u = 0; u < 4; ++u) ha[u] = *reinterpret_cast<const f32x4_t*>(h_s + jb + 32 + 4 * u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = jb + 16 + 4 * u;
          p0 = fmaf(w[j], hb[u][0], p0); p1 = fmaf(w[j + 1], hb[u][1], p1);
          p2 = fmaf(w[j + 2], hb[u][2], p2); p3 = fmaf(w[j + 3], hb[u][3], p3);
        }
      }
      pre += (p0 + p1) + (p2 + p3);
    }
    // every gate row applies its own activation (one v_exp per thread, no divergence: tanh(x) = 2 sigmoid(2x) - 1), so the
    // serial combine below is left with a single tanh per hidden unit
    const size_t o = ((size_t)b * T + t);
    if (row) {
      const bool is_g = (g >= 2 * h) && (g < 3 * h);
      const float sg = 1.0f / (1.0f + __expf(is_g ? -2.0f * pre : -pre));
      const float a = is_g ? 2.0f * sg - 1.0f : sg;
      pre_s[g] = a;
      if (gates) gates[o * ldx + (size_t)d * G + g] = a;
    }
    __syncthreads();
    if (g < h) {
      const float ig = pre_s[g], fg = pre_s[h + g], gg = pre_s[2 * h + g], og = pre_s[3 * h + g];
      c = fg * c + ig * gg;
      const float tc = 2.0f / (1.0f + __expf(-2.0f * c)) - 1.0f;
      const float hh = og * tc;
      if (hprev) hprev[o * ldy + (size_t)d * h + g] = h_s[g];      // h of the previous step of THIS direction (0 at its start)
      h_s[g] = hh;
      y[o * ldy + (size_t)d * h + g] = hh;
      bad |= !(hh == hh);
      if (cells) cells[o * ldy + (size_t)d * h + g] = c;
    }
    __syncthreads();
  }
  if (bad && nan_flag) atomicOr(nan_flag, 1);
}

__global__ __launch_bounds__(512) void lstm_bwd_kernel(const float* __restrict__ grad_y, const float* __restrict__ w_hh,
                                                       const float* __restrict__ gates, const float* __restrict__ cells,
                                                       float* __restrict__ grad_xproj, int T, int h, int ndir) {
  __shared__ __attribute__((aligned(16))) float da_s[4 * HP];
  __shared__ float part_s[4 * HP];
  const int b = blockIdx.x, d = blockIdx.y;
  const int p = threadIdx.x >> 7, j = threadIdx.x & (HP - 1);
  const int G = 4 * h;
  const bool live = j < h;
  float wt[HP];      // wt[r] = W_hh[d][p*h + r][j]
  {
    const float* wc = w_hh + (size_t)d * G * h + (size_t)p * h * h + (live ? j : 0);
#pragma unroll
    for (int r = 0; r < HP; ++r) wt[r] = (live && r < h) ? wc[(size_t)r * h] : 0.f;
  }
  da_s[threadIdx.x] = 0.f;      // pad lanes stay zero for the whole run
  const size_t ldx = (size_t)ndir * G, ldy = (size_t)ndir * h;
  // walk the forward order backwards: dir 0 ran t = 0..T-1, dir 1 ran t = T-1..0
  const int t0 = d ? 0 : T - 1, dt = d ? 1 : -1;
  float dh_rec = 0.f, dc_next = 0.f;
  __syncthreads();
  for (int s = 0, t = t0; s < T; ++s, t += dt) {
    if (p == 0 && live) {
      const size_t o = (size_t)b * T + t;
      const float* gp = gates + o * ldx + (size_t)d * G + j;
      const float ig = gp[0], fg = gp[h], gg = gp[2 * h], og = gp[3 * h];
      const float c = cells[o * ldy + (size_t)d * h + j];
      const float c_prev = (s + 1 < T) ? cells[((size_t)b * T + t + dt) * ldy + (size_t)d * h + j] : 0.f;
      const float dh = grad_y[o * ldy + (size_t)d * h + j] + dh_rec;
      const float tc = tanhf(c);
      const float dc = dc_next + dh * og * (1.f - tc * tc);
      const float da_i = dc * gg * ig * (1.f - ig);
      const float da_f = dc * c_prev * fg * (1.f - fg);
      const float da_g = dc * ig * (1.f - gg * gg);
      const float da_o = dh * tc * og * (1.f - og);
      dc_next = dc * fg;
      da_s[j] = da_i; da_s[HP + j] = da_f; da_s[2 * HP + j] = da_g; da_s[3 * HP + j] = da_o;
      float* gx = grad_xproj + o * ldx + (size_t)d * G + j;
      gx[0] = da_i; gx[h] = da_f; gx[2 * h] = da_g; gx[3 * h] = da_o;
    }
    __syncthreads();
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;      // four partial sums, as in the forward kernel
#pragma unroll
    for (int r = 0; r < HP; r += 4) {
      const f32x4_t dv = *reinterpret_cast<const f32x4_t*>(da_s + p * HP + r);
      a0 = fmaf(wt[r], dv[0], a0); a1 = fmaf(wt[r + 1], dv[1], a1);
      a2 = fmaf(wt[r + 2], dv[2], a2); a3 = fmaf(wt[r + 3], dv[3], a3);
    }
    part_s[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (p == 0 && live) dh_rec = (part_s[j] + part_s[HP + j]) + (part_s[2 * HP + j] + part_s[3 * HP + j]);
  }
}

// ---------------------------------------------------------------------------------------------------------
// The same recurrence with 16 batch rows per workgroup on the matrix cores (round 3; default, ruart_lstm_set_variant).
//
// The per-row kernels above pin one CU per (batch row, direction) - 128 workgroups x ~110 us for a (64, 100) layer - and every CU
// they sit on is lost to an encoder GEMM workgroup, which needs a whole CU.  Here a workgroup advances 16
// batch rows at once: the step's recurrent product P (16 x 4h) = h_{t-1} (16 x h) . W_hh^T is an MFMA product with the SDNet
// trunk's split-bf16 arithmetic (hi.hi + hi.lo + lo.hi, fp32 accumulation: ~2^-16 per product, sdnet_gemm.hip), W_hh's hi / lo
// fragments live in VGPRs for the whole sequence (128 per lane, as before), the running h in LDS as the bf16 hi / lo operand
// image (double-buffered: ONE barrier per step).  B x ndir / 16 = 8 workgroups instead of 128.
//
// Ownership.  Wave w, MFMA row m (0..15) stands for hidden unit  unit(w, m) = min(16 w + (m & ~3), h - 4) + (m & 3)  and owns all
// FOUR gate rows of it (the W rows are gathered that way once), so after the MFMAs lane (fr, fq) holds the i, f, g, o
// pre-activations of the four CONSECUTIVE units u0 .. u0 + 3, u0 = min(16 w + 4 fq, h - 4), of batch row fr: the cell update is
// lane-local, c and the previous h stay in registers, nothing is exchanged but the new h.  The clamp makes every lane's four units
// real ones: where 16 w + 4 fq + 3 runs past h (h = 125: the lanes of "units 124..127") the lane recomputes the last four real
// units instead - the same bits as their first owner (same operands, same instruction) - so EVERY lane issues the same full
// 16-byte loads and stores and the loop has no divergent path at all (the first version special-cased the ragged end: ~900
// instructions of exec-mask bookkeeping per step).  Duplicated positions must not be counted twice in a contraction: the W
// entries of a duplicate position are zero.  Batch rows past B are clamped to B - 1 the same way (duplicate, identical stores).
// The backward kernel mirrors it: the lane computes da_i..da_o of its units from the saved gates / cells, writes them to the
// global grad_xproj and to an LDS operand image da (16 x 512 positions, bf16 hi / lo), and after the barrier every wave takes
// dh_{t-1} of ITS 16 positions = da . W_hh[:, units] over all positions from W_hh^T fragments in VGPRs - the result lands in
// the lanes that own those units.
// Memory waits: vmcnt counts loads and stores in one in-order queue, so each step first waits for the NEXT step's operands
// (issued at the top of the step, a whole step ago) and only then issues its own stores - otherwise the wait for the loads would
// also drain the stores (a write latency per step: measured, 1.1 us).
// ---------------------------------------------------------------------------------------------------------
#ifndef RUART_LSTM_ABL
#define RUART_LSTM_ABL 0       // diagnostic builds: 1 no x loads in the loop, 2 no global stores, 4 no gate transcendental, 8 no MFMAs
#endif
#define LRB 16                 // batch rows per workgroup
#define LHS 272                // forward h image: 128 bf16 per row + 16 bytes pad
#define LDS_DA 1040            // backward da image: 512 bf16 per row + 16 bytes pad
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));   // 4 consecutive floats at any 4-byte aligned address

__device__ __forceinline__ float sigm_(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }
__device__ __forceinline__ float tanh_(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -2.8853900817779268f)) - 1.0f; }

__device__ __forceinline__ void split_bf16_(float x, bf16_t& hi, bf16_t& lo) {
  hi = (bf16_t)x;
  lo = (bf16_t)(x - (float)hi);
}
// hidden unit of ownership position (w, m), and whether that position is the unit's first (counted) one
__device__ __forceinline__ int lstm_unit(int w, int m, int h) { return min(16 * w + (m & ~3), h - 4) + (m & 3); }
// Is ownership position q = 16 w + m the one whose weights COUNT in a contraction over positions?  Natural positions (group base
// q & ~3 <= h - 4) always; of the clamped groups only the first one, and there only the units no natural position covers.
__device__ __forceinline__ bool lstm_counted(int q, int h) {
  const int base = q & ~3, b0 = (h - 4) & ~3;
  if (base <= h - 4) return true;
  return base == b0 + 4 && (h - 4 + (q & 3)) > b0 + 3;
}

__global__ __launch_bounds__(512) void lstm_fwd_mfma_kernel(const float* __restrict__ xproj, const float* __restrict__ w_hh,
                                                            float* __restrict__ y, float* __restrict__ gates, float* __restrict__ cells,
                                                            float* __restrict__ hprev, int B, int T, int h, int ndir,
                                                            int* __restrict__ nan_flag) {
  __shared__ __attribute__((aligned(16))) char himg[2][2][LRB * LHS];      // [buffer][hi | lo], k = ownership position 16 w + m
  const int d = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int G = 4 * h;
  // W_hh fragments: first MFMA operand of tile (gate type i, k-step ks) = gate row i*h + unit(w, fr), contraction positions
  // k = 32 ks + 8 fq .. + 7 -> hidden unit of that position; duplicate positions carry zero weights
  bf16x8_t wh[4][4], wl[4][4];
  {
    const int unit = lstm_unit(w, fr, h);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = ks * 32 + fq * 8 + e, kw = k >> 4, km = k & 15;
        const int ku = lstm_unit(kw, km, h);
        const bool counted = lstm_counted(k, h);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = counted ? w_hh[((size_t)d * G + (size_t)i * h + unit) * h + ku] : 0.f;
          bf16_t a, b;
          split_bf16_(v, a, b);
          wh[i][ks][e] = a;
          wl[i][ks][e] = b;
        }
      }
  }
  for (int i = tid; i < 2 * 2 * LRB * LHS / 4; i += 512) reinterpret_cast<int*>(&himg[0][0][0])[i] = 0;     // h_0 = 0; pads stay 0
  const int b = min((int)blockIdx.x * LRB + fr, B - 1);
  const int u0 = min(16 * w + 4 * fq, h - 4);          // the lane's units u0 .. u0 + 3 of batch row b
  float c[4] = {0.f, 0.f, 0.f, 0.f}, hp[4] = {0.f, 0.f, 0.f, 0.f};
  const size_t ldx = (size_t)ndir * G, ldy = (size_t)ndir * h;
  const int t0 = d ? T - 1 : 0, dt = d ? -1 : 1;
  const long long sx = (long long)dt * (long long)ldx, sy = (long long)dt * (long long)ldy;      // pointer steps per time step
  const float* xp = xproj + ((size_t)b * T + t0) * ldx + (size_t)d * G + u0;
  const size_t oy = ((size_t)b * T + t0) * ldy + (size_t)d * h + u0, og_ = ((size_t)b * T + t0) * ldx + (size_t)d * G + u0;
  float* yp = y + oy;
  float* cp_ = cells ? cells + oy : nullptr;
  float* hpp = hprev ? hprev + oy : nullptr;
  float* gp = gates ? gates + og_ : nullptr;
  auto load_x = [&](const float* p, f32x4_t (&x)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4_u v = *reinterpret_cast<const f32x4_u*>(p + (size_t)i * h);
      x[i] = (f32x4_t){v[0], v[1], v[2], v[3]};
    }
  };
  f32x4_t xcur[4], xnext[4];
  load_x(xp, xcur);
#pragma unroll
  for (int i = 0; i < 4; ++i) xnext[i] = xcur[i];
  __syncthreads();
  bool bad = false;
  int cur = 0;
  for (int s = 0; s < T; ++s, cur ^= 1) {
    xp += sx;
    if (s + 1 < T && !(RUART_LSTM_ABL & 1)) load_x(xp, xnext);
    bf16x8_t hh[4], hl[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      hh[ks] = *reinterpret_cast<const bf16x8_t*>(&himg[cur][0][fr * LHS + (ks * 32 + fq * 8) * 2]);
      hl[ks] = *reinterpret_cast<const bf16x8_t*>(&himg[cur][1][fr * LHS + (ks * 32 + fq * 8) * 2]);
    }
    f32x4_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (!(RUART_LSTM_ABL & 8)) {
      // (consecutive MFMAs never touch the same accumulator: a dependent pair stalls for the matrix pipe's latency)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = mfma_16x16x32(wl[i][ks], hh[ks], acc[i]);       // small terms first
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = mfma_16x16x32(wh[i][ks], hl[ks], acc[i]);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = mfma_16x16x32(wh[i][ks], hh[ks], acc[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][0] = (float)hh[i][0] + (float)hl[i][1] + (float)wh[i][0][0];
    }
    // lane-local cell update of units u0 + r, batch row b
    f32x4_t ga[4], hv, cv, hpv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pi = acc[0][r] + xcur[0][r], pf = acc[1][r] + xcur[1][r], pg = acc[2][r] + xcur[2][r], po = acc[3][r] + xcur[3][r];
#if RUART_LSTM_ABL & 4
      const float ig = pi * 0.1f, fg = pf * 0.1f, gg = pg * 0.1f, og = po * 0.1f;
      c[r] = fg * c[r] + ig * gg;
      const float tc = c[r];
#else
      // sigmoid(x) = rcp(1 + exp2(-x log2 e)), tanh(x) = 2 sigmoid(2x) - 1 on the raw v_exp_f32 / v_rcp_f32 (1 ulp each): the IEEE
      // divide of 1.0f / (...) is a ten-instruction sequence
      const float ig = sigm_(pi), fg = sigm_(pf), gg = tanh_(pg), og = sigm_(po);
      c[r] = fg * c[r] + ig * gg;
      const float tc = tanh_(c[r]);
#endif
      const float hn = og * tc;
      ga[0][r] = ig; ga[1][r] = fg; ga[2][r] = gg; ga[3][r] = og;
      hv[r] = hn;
      cv[r] = c[r];
      hpv[r] = hp[r];
      hp[r] = hn;
      bad |= !(hn == hn);
    }
    {
      // the new h as next step's second operand, at the lane's ownership positions 16 w + 4 fq .. + 3 of row fr
      bf16x4_t vh, vl;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bf16_t a, bb;
        split_bf16_(hv[r], a, bb);
        vh[r] = a;
        vl[r] = bb;
      }
      *reinterpret_cast<bf16x4_t*>(&himg[cur ^ 1][0][fr * LHS + (16 * w + 4 * fq) * 2]) = vh;
      *reinterpret_cast<bf16x4_t*>(&himg[cur ^ 1][1][fr * LHS + (16 * w + 4 * fq) * 2]) = vl;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // next step's x landed; only the PREVIOUS step's stores were behind it
#pragma unroll
    for (int i = 0; i < 4; ++i) xcur[i] = xnext[i];
    __builtin_amdgcn_sched_barrier(0);
    if (!((RUART_LSTM_ABL & 2) && s + 1 < T)) {
      *reinterpret_cast<f32x4_u*>(yp) = (f32x4_u){hv[0], hv[1], hv[2], hv[3]};
      if (cp_) *reinterpret_cast<f32x4_u*>(cp_) = (f32x4_u){cv[0], cv[1], cv[2], cv[3]};
      if (hpp) *reinterpret_cast<f32x4_u*>(hpp) = (f32x4_u){hpv[0], hpv[1], hpv[2], hpv[3]};
      if (gp) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4_u*>(gp + (size_t)i * h) = (f32x4_u){ga[i][0], ga[i][1], ga[i][2], ga[i][3]};
      }
    }
    yp += sy;
    if (cp_) cp_ += sy;
    if (hpp) hpp += sy;
    if (gp) gp += sx;
    __syncthreads();                                   // the new h image is complete; the old one is dead for every wave
  }
  if (bad && nan_flag) atomicOr(nan_flag, 1);
}

__global__ __launch_bounds__(512) void lstm_bwd_mfma_kernel(const float* __restrict__ grad_y, const float* __restrict__ w_hh,
                                                            const float* __restrict__ gates, const float* __restrict__ cells,
                                                            float* __restrict__ grad_xproj, int B, int T, int h, int ndir) {
  // dh_{t-1} (16 rows x 128 positions) = da_t (16 x 512 positions) . W^T.  Each wave contracts over ITS OWN 64 positions - whose da it
  // has just computed, in registers: the second MFMA operand of k-step j is the lane's own da of gate types 2j and 2j + 1 for its four
  // units, no LDS image of da at all - for ALL 128 output positions (8 tiles), and the eight waves' partial sums meet in LDS: 8 KB
  // written and 8 KB read per wave and step (the first form had every wave read the whole 32 KB da image: 256 KB of LDS reads per step).
  __shared__ __attribute__((aligned(16))) float part[2][8][8][64 * 4];      // [buffer][producer wave][output tile][lane x 4], 128 KB
  const int d = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int G = 4 * h;
  // W_hh^T fragments: first operand of (output tile t8, k-step j) = output position (t8, fr) -> unit(t8, fr); contraction element e of
  // lane group fq = gate type 2j + (e >> 2) of this wave's position 4 fq + (e & 3); duplicate contraction positions carry zero weights
  bf16x8_t wh[8][2], wl[8][2];
  {
#pragma unroll
    for (int t8 = 0; t8 < 8; ++t8) {
      const int unit = lstm_unit(t8, fr, h);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int type = 2 * j + (e >> 2), m = 4 * fq + (e & 3);
          const int gu = lstm_unit(w, m, h);
          const float v = lstm_counted(16 * w + m, h) ? w_hh[((size_t)d * G + (size_t)type * h + gu) * h + unit] : 0.f;
          bf16_t a, b;
          split_bf16_(v, a, b);
          wh[t8][j][e] = a;
          wl[t8][j][e] = b;
        }
    }
  }
  const int b = min((int)blockIdx.x * LRB + fr, B - 1);
  const int u0 = min(16 * w + 4 * fq, h - 4);
  const size_t ldx = (size_t)ndir * G, ldy = (size_t)ndir * h;
  const int t0 = d ? 0 : T - 1, dt = d ? 1 : -1;       // the forward order walked backwards
  const long long sx = (long long)dt * (long long)ldx, sy = (long long)dt * (long long)ldy;
  float dh_rec[4] = {0.f, 0.f, 0.f, 0.f}, dc_next[4] = {0.f, 0.f, 0.f, 0.f};
  struct StepIn { f32x4_t ig, fg, gg, og, cc, cp, gy; };
  const float* gp = gates + ((size_t)b * T + t0) * ldx + (size_t)d * G + u0;
  const float* cp_ = cells + ((size_t)b * T + t0) * ldy + (size_t)d * h + u0;
  const float* gyp = grad_y + ((size_t)b * T + t0) * ldy + (size_t)d * h + u0;
  float* gx = grad_xproj + ((size_t)b * T + t0) * ldx + (size_t)d * G + u0;
  auto ld = [&](const float* p) -> f32x4_t {
    const f32x4_u v = *reinterpret_cast<const f32x4_u*>(p);
    return (f32x4_t){v[0], v[1], v[2], v[3]};
  };
  // one unconditional 16-byte load per operand, issued ONE STEP AHEAD of their use; `prev`: the forward's previous step exists
  auto load_step = [&](bool prev, StepIn& in) {
    in.ig = ld(gp);
    in.fg = ld(gp + h);
    in.gg = ld(gp + 2 * (size_t)h);
    in.og = ld(gp + 3 * (size_t)h);
    in.cc = ld(cp_);
    in.cp = ld(prev ? cp_ + sy : cp_);                 // (multiplied by 0 below when there is none)
    in.gy = ld(gyp);
  };
  StepIn in, nxt;
  load_step(T > 1, in);
  nxt = in;
  for (int s = 0; s < T; ++s) {
    gp += sx;
    cp_ += sy;
    gyp += sy;
    if (s + 1 < T) load_step(s + 2 < T, nxt);
    const float has_prev = (s + 1 < T) ? 1.f : 0.f;
    f32x4_t da[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float dh = in.gy[r] + dh_rec[r];
      const float tc = tanh_(in.cc[r]);
      const float dc = dc_next[r] + dh * in.og[r] * (1.f - tc * tc);
      da[0][r] = dc * in.gg[r] * in.ig[r] * (1.f - in.ig[r]);
      da[1][r] = dc * (in.cp[r] * has_prev) * in.fg[r] * (1.f - in.fg[r]);
      da[2][r] = dc * in.ig[r] * (1.f - in.gg[r] * in.gg[r]);
      da[3][r] = dh * tc * in.og[r] * (1.f - in.og[r]);
      dc_next[r] = dc * in.fg[r];
    }
    // da as the second MFMA operand of k-step j: [type 2j: units 0..3 | type 2j + 1: units 0..3] of this lane, hi and lo
    bf16x8_t dh_[2], dl_[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        bf16_t a, bb;
        split_bf16_(da[2 * j + (e >> 2)][e & 3], a, bb);
        dh_[j][e] = a;
        dl_[j][e] = bb;
      }
    f32x4_t acc[8];
#pragma unroll
    for (int t8 = 0; t8 < 8; ++t8) acc[t8] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {                     // (consecutive MFMAs never touch the same accumulator)
#pragma unroll
      for (int t8 = 0; t8 < 8; ++t8) acc[t8] = mfma_16x16x32(wl[t8][j], dh_[j], acc[t8]);       // small terms first
#pragma unroll
      for (int t8 = 0; t8 < 8; ++t8) acc[t8] = mfma_16x16x32(wh[t8][j], dl_[j], acc[t8]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t8 = 0; t8 < 8; ++t8) acc[t8] = mfma_16x16x32(wh[t8][j], dh_[j], acc[t8]);
    float* mine = &part[s & 1][w][0][lane * 4];
#pragma unroll
    for (int t8 = 0; t8 < 8; ++t8) *reinterpret_cast<f32x4_t*>(mine + t8 * 256) = acc[t8];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // next step's operands in registers before this step's stores are issued
    in = nxt;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4_u*>(gx + (size_t)i * h) = (f32x4_u){da[i][0], da[i][1], da[i][2], da[i][3]};
    gx += sx;
    __syncthreads();                                   // (double-buffered: the next step writes the other half)
    // dh of this wave's output tile: the eight waves' partial sums, in wave order
    f32x4_t sum = *reinterpret_cast<const f32x4_t*>(&part[s & 1][0][w][lane * 4]);
#pragma unroll
    for (int pw = 1; pw < 8; ++pw) sum += *reinterpret_cast<const f32x4_t*>(&part[s & 1][pw][w][lane * 4]);
#pragma unroll
    for (int r = 0; r < 4; ++r) dh_rec[r] = sum[r];
  }
}

// ---------------------------------------------------------------------------------------------------------
// Fused LSTM cell for the wide, short `multi2one` LSTM (hidden 300: W_hh does not fit one CU's registers, and the
// sequences are only 1-3 real words long).  The recurrent product h W_hh^T is a plain GEMM per step; this kernel does the
// whole pointwise part of a step in one pass over a ragged, length-sorted batch: rows < n_active are advanced, rows
// >= n_active (items already finished) are carried through unchanged - replacing ~12 elementwise launches forward and ~25
// backward per step.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(const float* __restrict__ pre, const float* __restrict__ h_prev,
                                                            const float* __restrict__ c_prev, float* __restrict__ h_out,
                                                            float* __restrict__ c_out, float* __restrict__ acts, int n_active,
                                                            int n_rows, int h) {
  const long long total = (long long)n_rows * h;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += 256LL * gridDim.x) {
    const int r = (int)(e / h), j = (int)(e - (long long)r * h);
    if (r < n_active) {
      const float* p = pre + (size_t)r * 4 * h + j;
      const float ig = 1.0f / (1.0f + __expf(-p[0])), fg = 1.0f / (1.0f + __expf(-p[h]));
      const float gg = 2.0f / (1.0f + __expf(-2.0f * p[2 * h])) - 1.0f, og = 1.0f / (1.0f + __expf(-p[3 * h]));
      const float c = fg * c_prev[e] + ig * gg;
      const float tc = 2.0f / (1.0f + __expf(-2.0f * c)) - 1.0f;
      c_out[e] = c;
      h_out[e] = og * tc;
      float* a = acts + (size_t)r * 4 * h + j;
      a[0] = ig; a[h] = fg; a[2 * h] = gg; a[3 * h] = og;
    } else {
      c_out[e] = c_prev[e];
      h_out[e] = h_prev[e];
    }
  }
}

__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float* __restrict__ gh, const float* __restrict__ gc,
                                                            const float* __restrict__ acts, const float* __restrict__ c_prev,
                                                            const float* __restrict__ c_out, float* __restrict__ g_pre,
                                                            float* __restrict__ g_hprev, float* __restrict__ g_cprev, int n_active,
                                                            int n_rows, int h) {
  const long long total = (long long)n_rows * h;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += 256LL * gridDim.x) {
    const int r = (int)(e / h), j = (int)(e - (long long)r * h);
    const float dh = gh ? gh[e] : 0.f, dcn = gc ? gc[e] : 0.f;
    if (r < n_active) {
      const float* a = acts + (size_t)r * 4 * h + j;
      const float ig = a[0], fg = a[h], gg = a[2 * h], og = a[3 * h];
      const float c = c_out[e];
      const float tc = 2.0f / (1.0f + __expf(-2.0f * c)) - 1.0f;
      const float dc = dcn + dh * og * (1.f - tc * tc);
      float* g = g_pre + (size_t)r * 4 * h + j;
      g[0] = dc * gg * ig * (1.f - ig);
      g[h] = dc * c_prev[e] * fg * (1.f - fg);
      g[2 * h] = dc * ig * (1.f - gg * gg);
      g[3 * h] = dh * tc * og * (1.f - og);
      g_cprev[e] = dc * fg;
      g_hprev[e] = 0.f;              // the recurrent path of h_prev goes through the GEMM (autograd adds it)
    } else {
      g_cprev[e] = dcn;
      g_hprev[e] = dh;
    }
  }
}

extern int* ruart_nan_flag_ptr;

// The operands one bidirectional nn.LSTM layer's kernels take, assembled from its eight parameter tensors in ONE launch (torch: two
// cats, two adds and a stack - five launches per layer and step, 55 of a step's ~900):
//   w (2 G, K) = [w_ih ; w_ih_r],  b (2 G) = [b_ih + b_hh ; b_ih_r + b_hh_r],  whh (2, G, h) = [w_hh , w_hh_r]      (G = 4 h)
__global__ void lstm_pack_params_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_ih_r, const float* __restrict__ b_ih,
                                        const float* __restrict__ b_hh, const float* __restrict__ b_ih_r, const float* __restrict__ b_hh_r,
                                        const float* __restrict__ w_hh, const float* __restrict__ w_hh_r, float* __restrict__ w,
                                        float* __restrict__ b, float* __restrict__ whh, long long n_w, long long n_hh, int G) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n_w + 2 * n_hh + 2 * G; i += stride) {
    if (i < n_w) {
      w[i] = w_ih[i];
    } else if (i < 2 * n_w) {
      w[i] = w_ih_r[i - n_w];
    } else if (i < 2 * n_w + n_hh) {
      whh[i - 2 * n_w] = w_hh[i - 2 * n_w];
    } else if (i < 2 * n_w + 2 * n_hh) {
      whh[i - 2 * n_w] = w_hh_r[i - 2 * n_w - n_hh];
    } else {
      const int j = (int)(i - 2 * n_w - 2 * n_hh);
      b[j] = j < G ? b_ih[j] + b_hh[j] : b_ih_r[j - G] + b_hh_r[j - G];
    }
  }
}

extern "C" int ruart_lstm_pack_params(const float* w_ih, const float* w_ih_r, const float* b_ih, const float* b_hh, const float* b_ih_r,
                                      const float* b_hh_r, const float* w_hh, const float* w_hh_r, float* w, float* b, float* whh, int G,
                                      int K, int h, void* stream) {
  RUART_ENTRY();
  if (G <= 0 || K <= 0 || h <= 0 || !w_ih || !w_ih_r || !b_ih || !b_hh || !b_ih_r || !b_hh_r || !w_hh || !w_hh_r || !w || !b || !whh)
    return (int)hipErrorInvalidValue;
  const long long n_w = (long long)G * K, n_hh = (long long)G * h;
  const long long total = 2 * n_w + 2 * n_hh + 2 * G;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(lstm_pack_params_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_ih, w_ih_r, b_ih, b_hh, b_ih_r, b_hh_r, w_hh,
                     w_hh_r, w, b, whh, n_w, n_hh, G);
  RUART_CHECK_LAUNCH();
  return 0;
}

static int g_lstm_variant = 1;       // 1: 16 batch rows per workgroup on the matrix cores (default); 0: one workgroup per row, fp32 VALU
extern "C" int ruart_lstm_set_variant(int v) {
  if (v != 0 && v != 1) return (int)hipErrorInvalidValue;
  g_lstm_variant = v;
  return 0;
}

extern "C" int ruart_lstm_fwd(const float* xproj, const float* w_hh, float* y, float* gates, float* cells, float* hprev, int B, int T,
                              int h, int ndir, void* stream) {
  RUART_ENTRY();
  if (B <= 0 || T <= 0 || h <= 0 || h > HP || ndir < 1 || ndir > 2) return (int)hipErrorInvalidValue;
  if (g_lstm_variant == 1 && h >= 4) {
    hipLaunchKernelGGL(lstm_fwd_mfma_kernel, dim3((B + LRB - 1) / LRB, ndir), dim3(512), 0, (hipStream_t)stream, xproj, w_hh, y, gates, cells,
                       hprev, B, T, h, ndir, ruart_nan_flag_ptr);
    RUART_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(lstm_fwd_kernel, dim3(B, ndir), dim3(512), 0, (hipStream_t)stream, xproj, w_hh, y, gates, cells, hprev, T, h, ndir,
                     ruart_nan_flag_ptr);
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" int ruart_lstm_bwd(const float* grad_y, const float* w_hh, const float* gates, const float* cells, float* grad_xproj,
                              int B, int T, int h, int ndir, void* stream) {
  RUART_ENTRY();
  if (B <= 0 || T <= 0 || h <= 0 || h > HP || ndir < 1 || ndir > 2) return (int)hipErrorInvalidValue;
  if (g_lstm_variant == 1 && h >= 4) {
    hipLaunchKernelGGL(lstm_bwd_mfma_kernel, dim3((B + LRB - 1) / LRB, ndir), dim3(512), 0, (hipStream_t)stream, grad_y, w_hh, gates, cells,
                       grad_xproj, B, T, h, ndir);
    RUART_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(lstm_bwd_kernel, dim3(B, ndir), dim3(512), 0, (hipStream_t)stream, grad_y, w_hh, gates, cells, grad_xproj, T, h,
                     ndir);
  RUART_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// out[w][c] = x[w][c] * m[idx[w]][c]: the variational dropout of a PACKED (words, D) matrix whose mask is drawn per (item, feature)
// (layers.row_dropout; Models/Layers.py:23-30 on the reference's padded (items, Lw, D) layout).  One pass instead of torch's gather of the
// mask rows into a (words, D) matrix followed by a multiply (3x the traffic); the backward is the same kernel on the gradient.
// One wave per row, four 16-byte loads in flight per lane; D % 4 == 0, rows 16-byte aligned.
__global__ __launch_bounds__(256) void rows_scale_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ m, int ldm,
                                                         const long long* __restrict__ idx, float* __restrict__ out, int ldo, int rows, int D) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= rows) return;
  const float* xr = x + (size_t)w * ldx;
  const float* mr = m + (size_t)idx[w] * ldm;
  float* orow = out + (size_t)w * ldo;
  for (int c0 = lane * 4; c0 < D; c0 += 4 * 256) {
    f32x4_t a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = min(c0 + u * 256, D - 4);                 // clamped: every load is unconditional
      a[u] = load4(xr + c);
      b[u] = load4(mr + c);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (c0 + u * 256 < D) store4(orow + c0 + u * 256, a[u] * b[u]);
  }
}

extern "C" int ruart_rows_scale(const float* x, int ldx, const float* mask, int ldm, const long long* row_of, float* out, int ldo, int rows,
                                int D, void* stream) {
  RUART_ENTRY();
  if (rows <= 0 || D < 4 || (D & 3) || (ldx & 3) || (ldm & 3) || (ldo & 3) || !x || !mask || !row_of || !out) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(rows_scale_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, mask, ldm, row_of, out, ldo, rows, D);
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" int ruart_lstm_cell_fwd(const float* pre, const float* h_prev, const float* c_prev, float* h_out, float* c_out,
                                   float* acts, int n_active, int n_rows, int h, void* stream) {
  RUART_ENTRY();
  if (n_rows <= 0 || h <= 0 || n_active < 0 || n_active > n_rows) return (int)hipErrorInvalidValue;
  const long long total = (long long)n_rows * h;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pre, h_prev, c_prev, h_out, c_out, acts,
                     n_active, n_rows, h);
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" int ruart_lstm_cell_bwd(const float* grad_h, const float* grad_c, const float* acts, const float* c_prev,
                                   const float* c_out, float* grad_pre, float* grad_h_prev, float* grad_c_prev, int n_active,
                                   int n_rows, int h, void* stream) {
  RUART_ENTRY();
  if (n_rows <= 0 || h <= 0 || n_active < 0 || n_active > n_rows) return (int)hipErrorInvalidValue;
  const long long total = (long long)n_rows * h;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grad_h, grad_c, acts, c_prev, c_out,
                     grad_pre, grad_h_prev, grad_c_prev, n_active, n_rows, h);
  RUART_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Embedding weight gradient from a HOST-prepared sort (Models/SDNet.py:439-493 looks words / POS / entity ids up in
// nn.Embedding tables; torch's backward sorts the ids on the device every step - ~100 launches and ~1 ms for the step's nine
// lookups).  The ids of a batch are known when it is collated, so ruart_amd.batch.BatchIndex sorts them there (numpy, in the
// loader worker): order[] lists the lookup positions grouped by table row, seg_start[s] .. seg_start[s+1] is the slice of
// order[] that hit row seg_row[s].  One workgroup per distinct row adds its gradient rows in that fixed order (no atomics:
// deterministic); rows that were never looked up keep the zero the caller filled gw with.
// A row with very many occurrences (a frequent word; [CLS], [SEP] and the first positions of the trainable encoder's tables: thousands
// per batch) would be one long chain of dependent loads in one workgroup, so the host cuts such rows into sub-segments of <= 64
// occurrences (batch._sort_ids): ruart_embedding_bwd_split sums the sub-segments into a workspace with this kernel (seg_row == NULL:
// segment s -> row s) and then the sub-segment sums of each row into gw with it again (order == NULL: occurrence i is workspace row i).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embedding_bwd_sorted_kernel(const float* __restrict__ gy, const int* __restrict__ order,
                                                                   const int* __restrict__ seg_start, const int* __restrict__ seg_row,
                                                                   float* __restrict__ gw, int D, int dpad) {
  __shared__ float red[256];
  const int s = blockIdx.x, tid = threadIdx.x;
  const int beg = seg_start[s], end = seg_start[s + 1];
  float* dst = gw + (size_t)(seg_row ? seg_row[s] : s) * D;
  const int G = 256 / dpad;                     // occurrence groups working side by side on narrow tables (dpad = 8..256)
  const int g = tid / dpad, dl = tid % dpad;
  if (G == 1) {
    // wide tables (D > 128; blockIdx.y = chunk of 256 columns): a frequent row - the position table of the trainable encoder has 64
    // rows with ~700 occurrences each - is a chain of dependent loads, so eight occurrences are in flight per lane and the eight partial
    // sums are combined in a fixed order
    const int d = blockIdx.y * 256 + tid;
    if (d >= D) return;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int i = beg;
    for (; i + 7 < end; i += 8) {
      int o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = order ? order[i + k] : i + k;
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] += gy[(size_t)o[k] * D + d];
    }
    for (int k = 0; i < end; ++i, ++k) a[k] += gy[(size_t)(order ? order[i] : i) * D + d];
    dst[d] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    return;
  }
  for (int d0 = 0; d0 < D; d0 += dpad) {
    const int d = d0 + dl;
    float acc = 0.f;
    if (d < D)
      for (int i = beg + g; i < end; i += G) acc += gy[(size_t)(order ? order[i] : i) * D + d];
    __syncthreads();
    red[tid] = acc;
    __syncthreads();
    if (g == 0 && d < D) {
      float t = 0.f;
      for (int q = 0; q < G; ++q) t += red[q * dpad + dl];          // fixed order
      dst[d] = t;
    }
  }
}

static void launch_embedding_bwd(const float* gy, const int* order, const int* seg_start, const int* seg_row, int n_seg, int D, float* out,
                                 hipStream_t stream) {
  int dpad = 8;
  while (dpad < D && dpad < 256) dpad <<= 1;
  hipLaunchKernelGGL(embedding_bwd_sorted_kernel, dim3(n_seg, dpad == 256 ? (D + 255) / 256 : 1), dim3(256), 0, stream, gy, order, seg_start,
                     seg_row, out, D, dpad);
}

extern "C" int ruart_embedding_bwd_sorted(const float* grad_out, const int* order, const int* seg_start, const int* seg_row, int n_seg,
                                          int D, float* grad_weight, void* stream) {
  RUART_ENTRY();
  if (n_seg < 0 || D <= 0 || D > 4096 || (n_seg && (!order || !seg_start || !seg_row))) return (int)hipErrorInvalidValue;
  if (n_seg == 0) return 0;
  launch_embedding_bwd(grad_out, order, seg_start, seg_row, n_seg, D, grad_weight, (hipStream_t)stream);
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" int ruart_embedding_bwd_split(const float* grad_out, const int* order, const int* sub_start, int n_sub, const int* row_first,
                                         const int* row_id, int n_rows, int D, float* ws, float* grad_weight, void* stream) {
  RUART_ENTRY();
  if (n_sub < 0 || n_rows < 0 || n_rows > n_sub || D <= 0 || D > 4096 || (n_sub && (!order || !sub_start || !row_first || !row_id || !ws)))
    return (int)hipErrorInvalidValue;
  if (n_sub == 0) return 0;
  launch_embedding_bwd(grad_out, order, sub_start, nullptr, n_sub, D, ws, (hipStream_t)stream);          // ws[s] = sum of sub-segment s
  launch_embedding_bwd(ws, nullptr, row_first, row_id, n_rows, D, grad_weight, (hipStream_t)stream);      // gw[row] = its sub-segments, in order
  RUART_CHECK_LAUNCH();
  return 0;
}
