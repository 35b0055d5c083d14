// Row-wise (HBM-bound) kernels of the TRAINABLE encoder's 16-bit path (conf without LOCK_BERT, opt['bert_train_gemm'] = '16';
// Models/SDNet.py:88-94 puts the encoder's parameters into the optimizer, Models/Bert/modeling.py:171-303 is what is differentiated).
// Activations are stored in f16, gradients that feed a GEMM in bf16 (they need the exponent range), the residual-stream gradient in
// fp32; all arithmetic is fp32.  Dropout masks are never stored: a counter-based hash of (seed, row, column) regenerates them in
// the backward pass.
//
//   ruart_ln_train_fwd / _bwd      BertSelfOutput / BertOutput: LayerNorm(dropout(dense) + input)       modeling.py:260-264, 299-303
//                                  (and the embeddings' LayerNorm-then-dropout, :196-199, with `post` = 1)
//                                  (the GELU of modeling.py:52-57 and its backward ride in GEMM epilogues: ruart_gemm_16_nt_gelu2 / _gelu_bwd)
//   ruart_colsum_bf16 / _f32_rows  bias gradients: column sums of a gradient matrix / of per-strip partial sums
//   ruart_weight_prep              fp32 master weight -> f16 operand of the forward + transposed bf16 operand of dX = dY . W, one pass
//   ruart_f16_to_bf16              saved activations as the bf16 operand of a weight-gradient product
//   ruart_transpose16              (rows, cols) -> (cols, rows) of a 16-bit matrix (the first form of the weight-gradient products fed
//                                  the NT kernel transposed copies; the TN kernel of gemm_tn.hip replaced it - kept for the tests that
//                                  hold the two forms against each other)
//   ruart_splitk_reduce            sums the fp32 partial slabs of a split-K GEMM (fixed order: deterministic)
//   ruart_mix_rows / _bwd          mixed[r] = sum_l w[l] * layer_l[r] (Models/SDNet.py:573-581 on the token stream) and d w[l]
#include "common.h"
#include "ruart_hip.h"

#define TMAXG 4          // H <= 1024, one wave per row, lane owns columns (i*64 + lane)*4 .. +3

// y = LN(drop(x) + res) (post == 0)  or  y = drop(LN(x)) (post == 1, res ignored).  x fp32 (the dense output, bias included),
// res f16 or NULL.  Saved for the backward: pre16 = f16(the LayerNorm's input), stats[row] = (mean, rstd).
template <int NG>
__global__ __launch_bounds__(256) void ln_train_fwd_kernel(const float* __restrict__ x, int ldx, const f16_t* __restrict__ res, int ldr,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                           float p, unsigned seed, int post, f16_t* __restrict__ y, f16_t* __restrict__ pre16,
                                                           float* __restrict__ stats, int ld16, int rows, int H) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float keep_inv = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  f32x4_t v[NG];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const int c = (i * 64 + lane) * 4;
    v[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (c < H) {
      v[i] = load4(x + (size_t)row * ldx + c);
      if (!post) {
        if (p > 0.f) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[i][r] *= drop_scale(seed, (unsigned)(row * H + c + r), p, keep_inv);
        }
        if (res) v[i] += load4(res + (size_t)row * ldr + c);
      }
      store4(pre16 + (size_t)row * ld16 + c, v[i]);
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
  }
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = v[i][r] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
  if (lane == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
  }
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      const f32x4_t g = load4(gamma + c), b = load4(beta + c);
      f32x4_t o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[r] = g[r] * ((v[i][r] - mean) * rstd) + b[r];
        if (post && p > 0.f) o[r] *= drop_scale(seed, (unsigned)(row * H + c + r), p, keep_inv);
      }
      store4(y + (size_t)row * ld16 + c, o);
    }
  }
}

// Backward of the op above.  dy fp32 (+ add_scale * add when add != NULL: the layer-mix gradient joins here).
//   post == 0: d_res (fp32) = gradient w.r.t. the LayerNorm's input (what flows on through the residual connection),
//              d_gemm (bf16) = the same times the dropout multiplier (gradient w.r.t. the dense output);
//   post == 1: dy is first multiplied by the dropout multiplier; only d_res is written (gradient w.r.t. the embedding sum).
// Partial column sums of d(gamma), d(beta) and (post == 0) of the unrounded d_gemm - the bias gradient of the dense layer in front - go
// to part[(block, 0/1/2, H)]; ruart_ln_train_bwd reduces them in block order.
template <int NG>
__global__ __launch_bounds__(256) void ln_train_bwd_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ add,
                                                           const float* __restrict__ add_scale, const f16_t* __restrict__ pre16, int ld16,
                                                           const float* __restrict__ stats, const float* __restrict__ gamma, float p,
                                                           unsigned seed, int post, float* __restrict__ d_res, int ldd,
                                                           bf16_t* __restrict__ d_gemm, int ldg, float* __restrict__ part, int rows, int H) {
  __shared__ float red[3][3 * 256 * NG];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float keep_inv = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  const float a = add ? add_scale[0] : 0.f;
  f32x4_t dgam[NG], dbet[NG], dbia[NG], gam[NG];
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const int c = (i * 64 + lane) * 4;
    dgam[i] = dbet[i] = dbia[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    gam[i] = c < H ? load4(gamma + c) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  for (int row = blockIdx.x * 4 + wv; row < rows; row += gridDim.x * 4) {
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    f32x4_t g[NG], xh[NG];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int c = (i * 64 + lane) * 4;
      g[i] = xh[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      if (c < H) {
        f32x4_t d = load4(dy + (size_t)row * ldy + c);
        if (add) d += load4(add + (size_t)row * ldy + c) * a;
        if (post && p > 0.f) {
#pragma unroll
          for (int r = 0; r < 4; ++r) d[r] *= drop_scale(seed, (unsigned)(row * H + c + r), p, keep_inv);
        }
        const f32x4_t pr = load4(pre16 + (size_t)row * ld16 + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          xh[i][r] = (pr[r] - mean) * rstd;
          dgam[i][r] += d[r] * xh[i][r];
          dbet[i][r] += d[r];
          g[i][r] = d[r] * gam[i][r];
          s1 += g[i][r];
          s2 += g[i][r] * xh[i][r];
        }
      }
    }
    const float m1 = wave_sum(s1) / (float)H, m2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < H) {
        f32x4_t dx;
#pragma unroll
        for (int r = 0; r < 4; ++r) dx[r] = rstd * (g[i][r] - m1 - xh[i][r] * m2);
        store4(d_res + (size_t)row * ldd + c, dx);
        if (!post) {
          if (p > 0.f) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dx[r] *= drop_scale(seed, (unsigned)(row * H + c + r), p, keep_inv);
          }
          dbia[i] += dx;
          store4(d_gemm + (size_t)row * ldg + c, dx);
        }
      }
    }
  }
  // the four waves' partial sums -> one row pair per block
  if (wv > 0) {
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      *reinterpret_cast<f32x4_t*>(&red[wv - 1][(i * 64 + lane) * 4]) = dgam[i];
      *reinterpret_cast<f32x4_t*>(&red[wv - 1][256 * NG + (i * 64 + lane) * 4]) = dbet[i];
      *reinterpret_cast<f32x4_t*>(&red[wv - 1][2 * 256 * NG + (i * 64 + lane) * 4]) = dbia[i];
    }
  }
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < H) {
        f32x4_t sg = dgam[i], sb = dbet[i], sx = dbia[i];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          sg += *reinterpret_cast<const f32x4_t*>(&red[k][c]);
          sb += *reinterpret_cast<const f32x4_t*>(&red[k][256 * NG + c]);
          sx += *reinterpret_cast<const f32x4_t*>(&red[k][2 * 256 * NG + c]);
        }
        store4(part + ((size_t)blockIdx.x * 3) * H + c, sg);
        store4(part + ((size_t)blockIdx.x * 3 + 1) * H + c, sb);
        store4(part + ((size_t)blockIdx.x * 3 + 2) * H + c, sx);
      }
    }
  }
}

// out[j] (+)= sum over `n` rows of part[row * stride + j]: 32 columns per block, eight row groups (rows g, g + 8, ...) summed in a fixed
// order - the same bits every run.  blockIdx.y selects one of up to three (part offset, out) pairs, so the LayerNorm backward reduces its
// gamma / beta / bias partials in one launch.
struct ColReduceOut {
  float* out[3];
  size_t off[3];
};
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ part, int n, size_t stride, ColReduceOut o, int cols,
                                                        int accumulate) {
  __shared__ float red[8][32];
  const int cx = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + cx;
  float* __restrict__ out = o.out[blockIdx.y];
  part += o.off[blockIdx.y];
  float s0 = 0.f, s1 = 0.f;
  if (j < cols) {
    int r = g;
    for (; r + 8 < n; r += 16) {
      s0 += part[(size_t)r * stride + j];
      s1 += part[(size_t)(r + 8) * stride + j];
    }
    if (r < n) s0 += part[(size_t)r * stride + j];
  }
  red[g][cx] = s0 + s1;
  __syncthreads();
  if (g == 0 && j < cols) {
    float s = red[0][cx];
#pragma unroll
    for (int k = 1; k < 8; ++k) s += red[k][cx];
    out[j] = accumulate ? out[j] + s : s;
  }
}
static void colreduce(const float* part, int n, size_t stride, float* out, int cols, int accumulate, hipStream_t s) {
  ColReduceOut o = {{out, nullptr, nullptr}, {0, 0, 0}};
  hipLaunchKernelGGL(colreduce_kernel, dim3(ceil_div(cols, 32)), dim3(256), 0, s, part, n, stride, o, cols, accumulate);
}

// column sums of a bf16 matrix in two deterministic stages: block (chunk of 256 rows, 256 columns) -> part[chunk][cols]
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ x, int ld, int rows, int cols, float* __restrict__ part) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const int r0 = blockIdx.y * 256;
  f32x4_t s = {0.f, 0.f, 0.f, 0.f};
  if (c < cols) {
    const int r1 = min(rows, r0 + 256);
    for (int r = r0 + wv; r < r1; r += 4) s += load4(x + (size_t)r * ld + c);
  }
  *reinterpret_cast<f32x4_t*>(&red[wv][lane * 4]) = s;
  __syncthreads();
  if (wv == 0 && c < cols) {
    f32x4_t t = s;
#pragma unroll
    for (int k = 1; k < 4; ++k) t += *reinterpret_cast<const f32x4_t*>(&red[k][lane * 4]);
    store4(part + (size_t)blockIdx.y * cols + c, t);
  }
}

// the same for an fp32 matrix of any width (the trunk's bias gradients: widths 250, 300, 1000 ... - one column per thread, 128 rows per
// block, four rows in flight)
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ x, int ld, int rows, int cols, float* __restrict__ part) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const int r0 = blockIdx.y * 128, r1 = min(rows, r0 + 128);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = r0;
  for (; r + 3 < r1; r += 4) {
    s0 += x[(size_t)r * ld + c];
    s1 += x[(size_t)(r + 1) * ld + c];
    s2 += x[(size_t)(r + 2) * ld + c];
    s3 += x[(size_t)(r + 3) * ld + c];
  }
  for (; r < r1; ++r) s0 += x[(size_t)r * ld + c];
  part[(size_t)blockIdx.y * cols + c] = (s0 + s1) + (s2 + s3);
}

// 64 x 64 tile transpose of a 16-bit matrix through LDS (row stride padded by 2 elements: conflict-light both ways)
template <bool F16_TO_BF16>
__global__ __launch_bounds__(256) void transpose16_kernel(const unsigned short* __restrict__ in, int ldi, unsigned short* __restrict__ out,
                                                          int ldo, int rows, int cols) {
  __shared__ unsigned short tile[64][66];
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    unsigned short v = (r0 + r < rows && c0 + tx < cols) ? in[(size_t)(r0 + r) * ldi + c0 + tx] : (unsigned short)0;
    if (F16_TO_BF16) v = __builtin_bit_cast(unsigned short, (bf16_t)(float)__builtin_bit_cast(f16_t, v));
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int c = ty; c < 64; c += 4)
    if (c0 + c < cols && r0 + tx < rows) out[(size_t)(c0 + c) * ldo + r0 + tx] = tile[tx][c];
}

// One pass over fp32 master weights (rows x cols each): out16 = f16(scale * W) in place of the forward's operand, outT = bf16(scale * W)^T
// (cols x rows) for dX = dY . W as an NT product.  64 x 64 tiles through LDS; up to eight weights per launch (a BERT layer has six: the
// per-weight launches were 72 x 12 us of latency per step).
struct WPrepBatch {
  ruart_wprep_item it[8];
  int tile0[9];      // first tile of item i; tile0[n] = total
  int n;
};
__global__ __launch_bounds__(256) void weight_prep_kernel(WPrepBatch b) {
  __shared__ float tile[64][65];
  int i = 0;
  while (i + 1 < b.n && (int)blockIdx.x >= b.tile0[i + 1]) ++i;
  const ruart_wprep_item& w = b.it[i];
  const int t = blockIdx.x - b.tile0[i], tcols = (w.cols + 63) / 64;
  const int c0 = (t % tcols) * 64, r0 = (t / tcols) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  f16_t* out16 = (f16_t*)w.out16;
  bf16_t* outT = (bf16_t*)w.outT_bf16;
  for (int r = ty; r < 64; r += 4) {
    float v = 0.f;
    if (r0 + r < w.rows && c0 + tx < w.cols) {
      v = w.w[(size_t)(r0 + r) * w.ldw + c0 + tx] * w.scale;
      if (out16) out16[(size_t)(r0 + r) * w.ld16 + c0 + tx] = (f16_t)v;
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  if (outT)
    for (int c = ty; c < 64; c += 4)
      if (c0 + c < w.cols && r0 + tx < w.rows) outT[(size_t)(c0 + c) * w.ldT + r0 + tx] = (bf16_t)tile[tx][c];
}

// C[m][n] (+)= sum_z part[z][m][n]  (fp32, z in order)
__global__ void splitk_reduce_kernel(const float* __restrict__ part, size_t slab, int nz, float* __restrict__ C, size_t n4, float scale,
                                     int accumulate) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4_t s = load4(part + i * 4);
    for (int z = 1; z < nz; ++z) s += load4(part + (size_t)z * slab + i * 4);
    s *= scale;
    if (accumulate) s += load4(C + i * 4);
    store4(C + i * 4, s);
  }
}

// mixed[r][:] = sum_l w[l] * layers[l][r][:]   (f16 layers, fp32 out), one wave per row
__global__ __launch_bounds__(256) void mix_rows_kernel(const f16_t* __restrict__ layers, size_t layer_stride, int ld, int NL,
                                                       const float* __restrict__ w, float* __restrict__ out, int ldo, int rows, int H) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4_t acc[TMAXG];
#pragma unroll
  for (int i = 0; i < TMAXG; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int l = 0; l < NL; ++l) {
    const float wl = w[l];
    const f16_t* base = layers + (size_t)l * layer_stride + (size_t)row * ld;
#pragma unroll
    for (int i = 0; i < TMAXG; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < H) acc[i] += load4(base + c) * wl;
    }
  }
#pragma unroll
  for (int i = 0; i < TMAXG; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) store4(out + (size_t)row * ldo + c, acc[i]);
  }
}
// part[block][l] = sum over the block's rows of <g[r], layers[l][r]>  (d w[l]); reduced by colreduce_kernel
#define MIX_MAX_LAYERS 32
__global__ __launch_bounds__(256) void mix_rows_bwd_kernel(const f16_t* __restrict__ layers, size_t layer_stride, int ld, int NL,
                                                           const float* __restrict__ g, int ldg, float* __restrict__ part, int rows, int H) {
  __shared__ float red[4][MIX_MAX_LAYERS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float mine[MIX_MAX_LAYERS];
#pragma unroll
  for (int l = 0; l < MIX_MAX_LAYERS; ++l) mine[l] = 0.f;
  for (int row = blockIdx.x * 4 + wv; row < rows; row += gridDim.x * 4) {
    f32x4_t gv[TMAXG];
#pragma unroll
    for (int i = 0; i < TMAXG; ++i) {
      const int c = (i * 64 + lane) * 4;
      gv[i] = c < H ? load4(g + (size_t)row * ldg + c) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int l = 0; l < MIX_MAX_LAYERS; ++l) {
      if (l < NL) {
        const f16_t* base = layers + (size_t)l * layer_stride + (size_t)row * ld;
#pragma unroll
        for (int i = 0; i < TMAXG; ++i) {
          const int c = (i * 64 + lane) * 4;
          if (c < H) {
            const f32x4_t v = load4(base + c);
            mine[l] += v[0] * gv[i][0] + v[1] * gv[i][1] + v[2] * gv[i][2] + v[3] * gv[i][3];
          }
        }
      }
    }
  }
#pragma unroll
  for (int l = 0; l < MIX_MAX_LAYERS; ++l) {
    if (l < NL) {
      const float s = wave_sum(mine[l]);
      if (lane == 0) red[wv][l] = s;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < NL) part[(size_t)blockIdx.x * NL + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// ------------------------------------------------------------------------------------------------------------------------------
// workgroups of the LayerNorm backward (a wave walks rows grid-stride, one row in flight): 768 = 12 waves per CU.  Round 6, trained-encoder
// step, interleaved: 512 44.3-44.9 ms, 768 44.0-44.5, 1024 44.1-44.2; the next row's operands prefetched (158 VGPRs): no gain.
#ifndef LN_BWD_BLOCKS
#define LN_BWD_BLOCKS 768
#endif

extern "C" int ruart_ln_train_fwd(const float* x, int ldx, const void* res16, int ldr, const float* gamma, const float* beta, float eps,
                                  float p, unsigned seed, int post, void* y16, void* pre16, float* stats, int ld16, int rows, int H,
                                  void* stream) {
  RUART_ENTRY();
  if (H % 4 || H > 256 * TMAXG || rows <= 0 || (ldx & 3) || (ld16 & 3) || !y16 || !pre16 || !stats || p < 0.f || p >= 1.f)
    return (int)hipErrorInvalidValue;
  // NG = column groups of 256 a row needs: the per-lane state (and the backward's LDS) is sized for the width at hand - H = 768 runs
  // the three-group instance (backward: 102 VGPRs and 27 KB instead of 134 and 36 KB of the H = 1024 one)
#define LN_FWD(NG)                                                                                                                        \
  hipLaunchKernelGGL(ln_train_fwd_kernel<NG>, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, (const f16_t*)res16, ldr, \
                     gamma, beta, eps, p, seed, post, (f16_t*)y16, (f16_t*)pre16, stats, ld16, rows, H)
  switch (ceil_div(H, 256)) {
    case 1: LN_FWD(1); break;
    case 2: LN_FWD(2); break;
    case 3: LN_FWD(3); break;
    default: LN_FWD(4); break;
  }
#undef LN_FWD
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t ruart_ln_train_bwd_ws_floats(int H) { return (size_t)LN_BWD_BLOCKS * 3 * H; }

extern "C" int ruart_ln_train_bwd(const float* dy, int ldy, const float* add, const float* add_scale, const void* pre16, int ld16,
                                  const float* stats, const float* gamma, float p, unsigned seed, int post, float* d_res, int ldd,
                                  void* d_gemm_bf16, int ldg, float* d_gamma, float* d_beta, float* d_bias, int accumulate, float* ws, int rows,
                                  int H, void* stream) {
  RUART_ENTRY();
  if (H % 4 || H > 256 * TMAXG || rows <= 0 || !d_res || (!post && !d_gemm_bf16) || !ws || !d_gamma || !d_beta) return (int)hipErrorInvalidValue;
  const int blocks = min(LN_BWD_BLOCKS, ceil_div(rows, 4));
#define LN_BWD(NG)                                                                                                                       \
  hipLaunchKernelGGL(ln_train_bwd_kernel<NG>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, ldy, add, add_scale, (const f16_t*)pre16, \
                     ld16, stats, gamma, p, seed, post, d_res, ldd, (bf16_t*)d_gemm_bf16, ldg, ws, rows, H)
  switch (ceil_div(H, 256)) {
    case 1: LN_BWD(1); break;
    case 2: LN_BWD(2); break;
    case 3: LN_BWD(3); break;
    default: LN_BWD(4); break;
  }
#undef LN_BWD
  ColReduceOut o = {{d_gamma, d_beta, d_bias}, {0, (size_t)H, (size_t)2 * H}};
  hipLaunchKernelGGL(colreduce_kernel, dim3(ceil_div(H, 32), (d_bias && !post) ? 3 : 2), dim3(256), 0, (hipStream_t)stream, ws, blocks,
                     (size_t)3 * H, o, H, accumulate);
  RUART_CHECK_LAUNCH();
  return 0;
}

// Backward through the GELU of the intermediate dense as a pass of its own (round 5: the fused form, ruart_gemm_16_nt_gelu_bwd, spends
// 300 us of a 476-us launch in its epilogue - 8 waves per CU doing two transcendentals per element and 384 KB of traffic per tile after the
// matrix work, nothing overlapped; the plain product + this pass are 160 + ~190 us).  In place: d (M x N bf16) holds acc = dY . W2 and
// becomes acc * gelu'(H); G = gelu(H) (bf16: the X operand of the output dense's weight gradient); colpart[(M / 128) x N] = column sums
// of the unrounded d per 128-row strip (the intermediate bias gradient after ruart_colsum_f32_rows).  One workgroup = one strip x 256
// columns, a wave takes every fourth row, eight rows of loads in flight per lane.
#include "gemm_shared.h"
__global__ __launch_bounds__(256) void gelu_bwd_rows_kernel(bf16_t* __restrict__ d, const f16_t* __restrict__ Hh, int ld, bf16_t* __restrict__ G,
                                                            float* __restrict__ colpart, int M, int N) {
  __shared__ __attribute__((aligned(16))) float red[3][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int col = blockIdx.x * 256 + lane * 4;
  const int r0 = blockIdx.y * 128;
  f32x4_t cs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int it = 0; it < 4; ++it) {
    f32x4_t a[8], h[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const size_t o = (size_t)(r0 + (it * 8 + k) * 4 + wv) * ld + col;
      a[k] = load4_stream(d + o);
      h[k] = load4_stream(Hh + o);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const size_t o = (size_t)(r0 + (it * 8 + k) * 4 + wv) * ld + col;
      f32x2_t g0, d0, g1, d1;
      gelu_fwd_bwd_pk((f32x2_t){h[k][0], h[k][1]}, g0, d0);
      gelu_fwd_bwd_pk((f32x2_t){h[k][2], h[k][3]}, g1, d1);
      const f32x4_t v = a[k] * (f32x4_t){d0.x, d0.y, d1.x, d1.y};
      cs += v;
      store4(G + o, (f32x4_t){g0.x, g0.y, g1.x, g1.y});
      store4(d + o, v);
    }
  }
  if (wv > 0) *reinterpret_cast<f32x4_t*>(&red[wv - 1][lane * 4]) = cs;
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) cs += *reinterpret_cast<const f32x4_t*>(&red[k][lane * 4]);
    store4(colpart + (size_t)blockIdx.y * N + col, cs);
  }
}
extern "C" int ruart_gelu_bwd_rows(void* d_bf16, const void* h16, int ld, void* g_bf16, float* colpart, int M, int N, void* stream) {
  RUART_ENTRY();
  if (M <= 0 || M % 128 || N <= 0 || N % 256 || (ld & 3) || ld < N || !d_bf16 || !h16 || !g_bf16 || !colpart) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(gelu_bwd_rows_kernel, dim3(N / 256, M / 128), dim3(256), 0, (hipStream_t)stream, (bf16_t*)d_bf16, (const f16_t*)h16, ld,
                     (bf16_t*)g_bf16, colpart, M, N);
  RUART_CHECK_LAUNCH();
  return 0;
}

__global__ void f16_to_bf16_kernel(const f16_t* __restrict__ in, bf16_t* __restrict__ out, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) store4(out + i * 4, load4(in + i * 4));
}
extern "C" int ruart_f16_to_bf16(const void* in16, void* out_bf16, long long n, void* stream) {
  RUART_ENTRY();
  if (n <= 0 || n % 4 || !in16 || !out_bf16) return (int)hipErrorInvalidValue;
  const size_t n4 = (size_t)n / 4;
  hipLaunchKernelGGL(f16_to_bf16_kernel, dim3((unsigned)min((size_t)4096, (n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f16_t*)in16,
                     (bf16_t*)out_bf16, n4);
  RUART_CHECK_LAUNCH();
  return 0;
}

/* out[j] (+)= sum_r x[r][j]; ws: ceil(rows / 256) * cols floats */
extern "C" int ruart_colsum_bf16(const void* x_bf16, int ld, int rows, int cols, float* out, int accumulate, float* ws, void* stream) {
  RUART_ENTRY();
  if (rows <= 0 || cols <= 0 || cols % 4 || (ld & 3) || !ws) return (int)hipErrorInvalidValue;
  const int chunks = ceil_div(rows, 256);
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3(ceil_div(cols, 256), chunks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_bf16, ld, rows,
                     cols, ws);
  colreduce(ws, chunks, (size_t)cols, out, cols, accumulate, (hipStream_t)stream);
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t ruart_colsum_f32_ws_floats(int rows, int cols) { return (size_t)ceil_div(rows, 128) * cols; }

/* out[j] (+)= sum_r x[r][j] of an fp32 matrix (any cols / ld); ws: ruart_colsum_f32_ws_floats(rows, cols) floats; fixed order */
extern "C" int ruart_colsum_f32(const float* x, int ld, int rows, int cols, float* out, int accumulate, float* ws, void* stream) {
  RUART_ENTRY();
  if (!x || !out || !ws || rows <= 0 || cols <= 0 || ld < cols) return (int)hipErrorInvalidValue;
  const int chunks = ceil_div(rows, 128);
  hipLaunchKernelGGL(colsum_f32_kernel, dim3(ceil_div(cols, 256), chunks), dim3(256), 0, (hipStream_t)stream, x, ld, rows, cols, ws);
  colreduce(ws, chunks, (size_t)cols, out, cols, accumulate, (hipStream_t)stream);
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" int ruart_transpose16(const void* in, int ldi, void* out, int ldo, int rows, int cols, int f16_to_bf16, void* stream) {
  RUART_ENTRY();
  if (rows <= 0 || cols <= 0) return (int)hipErrorInvalidValue;
  const dim3 grid(ceil_div(cols, 64), ceil_div(rows, 64));
  if (f16_to_bf16)
    hipLaunchKernelGGL(transpose16_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)in, ldi, (unsigned short*)out, ldo,
                       rows, cols);
  else
    hipLaunchKernelGGL(transpose16_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)in, ldi, (unsigned short*)out, ldo,
                       rows, cols);
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" int ruart_colsum_f32_rows(const float* part, int rows, int ld, int cols, float* out, int accumulate, void* stream) {
  RUART_ENTRY();
  if (!part || !out || rows <= 0 || cols <= 0 || ld < cols) return (int)hipErrorInvalidValue;
  colreduce(part, rows, (size_t)ld, out, cols, accumulate, (hipStream_t)stream);
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" int ruart_weight_prep_batch(const ruart_wprep_item* items, int n, void* stream) {
  RUART_ENTRY();
  if (!items || n <= 0 || n > 8) return (int)hipErrorInvalidValue;
  WPrepBatch b;
  b.n = n;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    const ruart_wprep_item& w = items[i];
    if (!w.w || w.rows <= 0 || w.cols <= 0 || (!w.out16 && !w.outT_bf16)) return (int)hipErrorInvalidValue;
    b.it[i] = w;
    b.tile0[i] = tiles;
    tiles += ceil_div(w.cols, 64) * ceil_div(w.rows, 64);
  }
  b.tile0[n] = tiles;
  hipLaunchKernelGGL(weight_prep_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, b);
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" int ruart_weight_prep(const float* w, int ldw, float scale, void* out16, int ld16, void* outT_bf16, int ldT, int rows, int cols,
                                 void* stream) {
  const ruart_wprep_item one = {w, out16, outT_bf16, ldw, ld16, ldT, rows, cols, scale};
  return ruart_weight_prep_batch(&one, 1, stream);
}

extern "C" int ruart_splitk_reduce(const float* part, long long slab_floats, int nz, float* C, long long n, float scale, int accumulate,
                                   void* stream) {
  RUART_ENTRY();
  if (nz <= 0 || n <= 0 || n % 4) return (int)hipErrorInvalidValue;
  const size_t n4 = (size_t)n / 4;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)min((size_t)2048, (n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part,
                     (size_t)slab_floats, nz, C, n4, scale, accumulate);
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" int ruart_mix_rows(const void* layers16, long long layer_stride, int ld, int n_layers, const float* w, float* out, int ldo, int rows,
                              int H, void* stream) {
  RUART_ENTRY();
  if (H % 4 || H > 256 * TMAXG || rows <= 0 || n_layers <= 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mix_rows_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const f16_t*)layers16, (size_t)layer_stride, ld,
                     n_layers, w, out, ldo, rows, H);
  RUART_CHECK_LAUNCH();
  return 0;
}
/* d_w[l] = sum_r <g[r], layers[l][r]>; ws: 512 * n_layers floats */
extern "C" int ruart_mix_rows_bwd(const void* layers16, long long layer_stride, int ld, int n_layers, const float* g, int ldg, float* d_w,
                                  float* ws, int rows, int H, void* stream) {
  RUART_ENTRY();
  if (H % 4 || H > 256 * TMAXG || rows <= 0 || n_layers <= 0 || n_layers > MIX_MAX_LAYERS || !ws) return (int)hipErrorInvalidValue;
  const int blocks = min(512, ceil_div(rows, 4));
  hipLaunchKernelGGL(mix_rows_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16_t*)layers16, (size_t)layer_stride, ld, n_layers,
                     g, ldg, ws, rows, H);
  colreduce(ws, blocks, (size_t)n_layers, d_w, n_layers, 0, (hipStream_t)stream);
  RUART_CHECK_LAUNCH();
  return 0;
}
