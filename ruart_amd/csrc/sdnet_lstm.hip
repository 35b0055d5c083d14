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

extern "C" int ruart_lstm_fwd(const float* xproj, const float* w_hh, float* y, float* gates, float* cells, float* hprev, int B, int T,
                              int h, int ndir, void* stream) {
  RUART_ENTRY();
  if (B <= 0 || T <= 0 || h <= 0 || h > HP || ndir < 1 || ndir > 2) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(lstm_fwd_kernel, dim3(B, ndir), dim3(512), 0, (hipStream_t)stream, xproj, w_hh, y, gates, cells, hprev, T, h, ndir,
                     ruart_nan_flag_ptr);
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" int ruart_lstm_bwd(const float* grad_y, const float* w_hh, const float* gates, const float* cells, float* grad_xproj,
                              int B, int T, int h, int ndir, void* stream) {
  RUART_ENTRY();
  if (B <= 0 || T <= 0 || h <= 0 || h > HP || ndir < 1 || ndir > 2) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(lstm_bwd_kernel, dim3(B, ndir), dim3(512), 0, (hipStream_t)stream, grad_y, w_hh, gates, cells, grad_xproj, T, h,
                     ndir);
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
