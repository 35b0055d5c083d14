// fp32 GEMM of the SDNet trunk on the 16-bit matrix cores: every fp32 operand element is split into two bf16 values
// (hi = bf16(x), lo = bf16(x - hi): 16 significand bits together, fp32's exponent range, so no scaling is needed for tiny
// gradients) and the product is accumulated in fp32 as  hi.hi + hi.lo + lo.hi  - three v_mfma_f32_16x16x32_bf16 per tile
// step instead of eight v_mfma_f32_16x16x4_f32 at 1/16 of the rate, i.e. ~5x the fp32-MFMA throughput at a relative error
// of ~2^-16 per product (the dropped lo.lo term and the rounding of lo).
//
// Replaces the rocBLAS calls behind torch.mm / addmm in the trunk (Models/Layers.py:155, 166, 226-227 and their backward):
//   C[M,N] = A(M,K) . B(K,N) (+ bias[N]),  fp32 in, fp32 out,
// with A element (m,k) at A[m*sam + k*sak] and B element (k,n) at B[k*sbk + n*sbn] - the forward x.W^T (both K-contiguous),
// dX = dY.W (B is N-contiguous) and dW = dY^T.X (A is M-contiguous, B is N-contiguous) all go through one entry point.
//
// Tile 128x128x32, 4 waves of 64x64 (4x4 MFMA tiles, 64 accumulator VGPRs), register-staged loads (the split happens on
// the way to LDS), two LDS stages, two workgroups per CU.  A K-contiguous operand is stored [row][32 k] with two rows per
// 128-byte line and the XOR swizzle of gemm.hip (conflict-free ds_read_b128); an M/N-contiguous operand is stored as it
// comes, [k][128 m], and its fragments are fetched with ds_read_b64_tr_b16 (the hardware transpose), so neither layout
// needs a transposing store.  Small outputs with a long reduction (the weight gradients: e.g. 500x125 over K = 6400) are
// split along K over up to 64 workgroups; every slice parks its partial tile in a workspace and a second, fully parallel
// launch adds the slices in slice order (+ bias) - a deterministic sum, no float atomics, no inter-workgroup hand-off.
#include "common.h"
#include "ruart_hip.h"

#define XBM 128
#define XBN 128
#define XBK 32
#define X_ARR 8704                 // bytes per LDS operand image: max(128 rows * 64 B, 32 k-rows * 272 B)
#define X_TRS 272                  // row stride of the [k][m] image (256 B + 16 B pad)

typedef __attribute__((__vector_size__(4 * sizeof(short)))) short xtr16x4_t;
typedef __attribute__((address_space(3))) xtr16x4_t* xtr_ptr_t;

namespace {

__device__ __forceinline__ void split_bf16(float x, bf16_t& hi, bf16_t& lo) {
  hi = (bf16_t)x;
  lo = (bf16_t)(x - (float)hi);
}

// byte offset of element (row r, k) in the swizzled [row][32 k] image: two rows share a 128-byte line
__device__ __forceinline__ int kc_off(int r, int chunk /*16-byte chunk of the row, 0..3*/) {
  const int line = r >> 1;
  return line * 128 + ((((r & 1) * 4 + chunk) ^ (line & 7)) << 4);
}

// MODE 0: the operand's K index is contiguous in memory (element (row, k) at P[row*srow + k]);
// MODE 1: its row index is contiguous (element (row, k) at P[k*sk + row]).
template <int MODE, bool VEC>
struct Stager {
  float v[16];

  // rows = M or N (limit of the row index), r0 = first row of the tile, k0 = first k of this step
  __device__ __forceinline__ void load(const float* __restrict__ P, long srow, long sk, int r0, int rows, int k0, int K, int tid) {
    if (MODE == 0) {
      const int kq = (tid & 7) * 4;                   // 4 consecutive k
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int r = r0 + p * 32 + (tid >> 3), k = k0 + kq;
        const float* src = P + (long)r * srow + k;
        if (r < rows && VEC && k + 3 < K) {
          const f32x4_t t = *reinterpret_cast<const f32x4_t*>(src);
          v[p * 4 + 0] = t[0]; v[p * 4 + 1] = t[1]; v[p * 4 + 2] = t[2]; v[p * 4 + 3] = t[3];
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[p * 4 + i] = (r < rows && k + i < K) ? src[i] : 0.f;
        }
      }
    } else {
      const int m4 = (tid & 31) * 4;                  // 4 consecutive rows
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int k = k0 + 2 * (tid >> 5) + (p & 1) + 16 * (p >> 1), r = r0 + m4;
        const float* src = P + (long)k * sk + r;
        if (k < K && VEC && r + 3 < rows) {
          const f32x4_t t = *reinterpret_cast<const f32x4_t*>(src);
          v[p * 4 + 0] = t[0]; v[p * 4 + 1] = t[1]; v[p * 4 + 2] = t[2]; v[p * 4 + 3] = t[3];
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[p * 4 + i] = (k < K && r + i < rows) ? src[i] : 0.f;
        }
      }
    }
  }

  __device__ __forceinline__ void store(char* hi_img, char* lo_img, int tid) const {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      bf16x4_t h, l;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bf16_t a, b;
        split_bf16(v[p * 4 + i], a, b);
        h[i] = a;
        l[i] = b;
      }
      int off;
      if (MODE == 0) {
        const int r = p * 32 + (tid >> 3), kq = tid & 7;                 // 4 k values = 8 bytes inside chunk kq >> 1
        off = kc_off(r, kq >> 1) + (kq & 1) * 8;
      } else {
        const int k = 2 * (tid >> 5) + (p & 1) + 16 * (p >> 1);          // 4 rows = 8 bytes of k-row k
        off = k * X_TRS + (tid & 31) * 8;
      }
      *reinterpret_cast<bf16x4_t*>(hi_img + off) = h;
      *reinterpret_cast<bf16x4_t*>(lo_img + off) = l;
    }
  }
};

// fragment of MFMA tile `t16` (16 rows starting at row base) for lane (fr, fq): 8 consecutive k = 8 fq .. 8 fq + 7
template <int MODE>
__device__ __forceinline__ bf16x8_t frag(const char* img, int row_base, int fr, int fq) {
  if (MODE == 0) {
    return *reinterpret_cast<const bf16x8_t*>(img + kc_off(row_base + fr, fq));
  } else {
    // transposed fetch: lane 4q+p of a 16-lane group supplies k-row q, rows 4p..4p+3, and receives row (lane & 15), k-rows 0..3
    const char* base = img + (8 * fq + (fr >> 2)) * X_TRS + (row_base + (fr & 3) * 4) * 2;
    union { struct { xtr16x4_t a, b; } s; bf16x8_t f; } u;
    u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((xtr_ptr_t)base);
    u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((xtr_ptr_t)(base + 4 * X_TRS));
    return u.f;
  }
}

template <int AMODE, int BMODE, bool VECA, bool VECB>
__global__ __launch_bounds__(256, 2) void gemm_x3_kernel(const float* __restrict__ A, long sam, long sak, const float* __restrict__ B,
                                                         long sbk, long sbn, const float* __restrict__ bias, float* __restrict__ C,
                                                         int ldc, int M, int N, int K, int splitk, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];          // 2 stages x (A_hi, A_lo, B_hi, B_lo)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int ntn = (N + XBN - 1) / XBN, ntm = (M + XBM - 1) / XBM;
  const int ntiles = ntm * ntn;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = id / splitk, slice = id - tile * splitk;             // the slices of a tile are neighbours: same XCD / L2
  const int m0 = (tile / ntn) * XBM, n0 = (tile % ntn) * XBN;
  const int ksteps = (K + XBK - 1) / XBK;
  const int per = (ksteps + splitk - 1) / splitk;
  const int kbeg = slice * per, kend = min(ksteps, kbeg + per);

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  Stager<AMODE, VECA> sa;
  Stager<BMODE, VECB> sb;
  // operand A: rows = m (stride sam) in MODE 0 / k-rows of stride sak in MODE 1; operand B: "rows" = n
  const long a_srow = sam, a_sk = sak, b_srow = sbn, b_sk = sbk;
  if (kbeg < kend) {
    sa.load(A, a_srow, a_sk, m0, M, kbeg * XBK, K, tid);
    sb.load(B, b_srow, b_sk, n0, N, kbeg * XBK, K, tid);
  }
  for (int t = kbeg; t < kend; ++t) {
    char* st = smem + ((t - kbeg) & 1) * (4 * X_ARR);
    sa.store(st, st + X_ARR, tid);
    sb.store(st + 2 * X_ARR, st + 3 * X_ARR, tid);
    __syncthreads();                                  // one barrier per step: the other stage was last read two steps ago
    if (t + 1 < kend) {
      sa.load(A, a_srow, a_sk, m0, M, (t + 1) * XBK, K, tid);
      sb.load(B, b_srow, b_sk, n0, N, (t + 1) * XBK, K, tid);
    }
    bf16x8_t ah[4], al[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ah[j] = frag<AMODE>(st, wm * 64 + j * 16, fr, fq);
      al[j] = frag<AMODE>(st + X_ARR, wm * 64 + j * 16, fr, fq);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16x8_t bh = frag<BMODE>(st + 2 * X_ARR, wn * 64 + i * 16, fr, fq);
      const bf16x8_t bl = frag<BMODE>(st + 3 * X_ARR, wn * 64 + i * 16, fr, fq);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = mfma_16x16x32(bl, ah[j], acc[i][j]);      // small terms first
        acc[i][j] = mfma_16x16x32(bh, al[j], acc[i][j]);
        acc[i][j] = mfma_16x16x32(bh, ah[j], acc[i][j]);
      }
    }
  }

  if (splitk > 1) {
    // park the partial tile, thread-major ([i][j][tid] x 4 floats): x3_reduce_kernel re-reads it with the same mapping
    float* slab = ws + ((size_t)tile * splitk + slice) * (XBM * XBN);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4_t*>(slab + ((i * 4 + j) * 256 + tid) * 4) = acc[i][j];
    return;
  }
  // lane owns rows m = .. + j*16 + fr and four consecutive columns n = .. + i*16 + fq*4 + r
  const bool vec_out = ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + wn * 64 + i * 16 + fq * 4;
    f32x4_t bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = (n + r < N) ? bias[n + r] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wm * 64 + j * 16 + fr;
      if (m >= M || n >= N) continue;
      const f32x4_t v = acc[i][j] + bv;
      float* dst = C + (size_t)m * ldc + n;
      if (vec_out && n + 3 < N) {
        *reinterpret_cast<f32x4_t*>(dst) = v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < N) dst[r] = v[r];
      }
    }
  }
  (void)ntiles;
}

// second launch of a split-K product: one thread per 4 output floats adds the slices in slice order and writes C (+ bias)
__global__ __launch_bounds__(256) void x3_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ bias, float* __restrict__ C,
                                                        int ldc, int M, int N, int splitk) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
  const int ntn = (N + XBN - 1) / XBN;
  const int tile = blockIdx.x >> 4, ij = blockIdx.x & 15, i = ij >> 2, j = ij & 3;
  const int m = (tile / ntn) * XBM + wm * 64 + j * 16 + fr, n = (tile % ntn) * XBN + wn * 64 + i * 16 + fq * 4;
  if (m >= M || n >= N) return;
  const float* p = ws + (size_t)tile * splitk * (XBM * XBN) + (ij * 256 + tid) * 4;
  f32x4_t s = {0.f, 0.f, 0.f, 0.f};
  for (int sl = 0; sl < splitk; ++sl) s += *reinterpret_cast<const f32x4_t*>(p + (size_t)sl * (XBM * XBN));
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (n + r < N) C[(size_t)m * ldc + n + r] = s[r] + (bias ? bias[n + r] : 0.f);
}

int pick_splitk(int M, int N, int K) {
  const int tiles = ((M + XBM - 1) / XBM) * ((N + XBN - 1) / XBN);
  const int ksteps = (K + XBK - 1) / XBK;
  if (tiles >= 160 || ksteps < 16) return 1;
  int s = (448 + tiles - 1) / tiles;                   // aim at ~2 workgroups per CU
  if (s > ksteps / 4) s = ksteps / 4;                  // at least 4 steps per slice
  if (s > 64) s = 64;
  return s < 1 ? 1 : s;
}

template <int AM, int BM_, bool VA, bool VB>
void launch_x3(const float* A, long sam, long sak, const float* B, long sbk, long sbn, const float* bias, float* C, int ldc, int M,
               int N, int K, int splitk, float* ws, hipStream_t s) {
  auto kern = gemm_x3_kernel<AM, BM_, VA, VB>;
  constexpr int lds = 2 * 4 * X_ARR;
  static bool done = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds), true);
  (void)done;
  const int tiles = ((M + XBM - 1) / XBM) * ((N + XBN - 1) / XBN);
  hipLaunchKernelGGL(kern, dim3(tiles * splitk), dim3(256), lds, s, A, sam, sak, B, sbk, sbn, bias, C, ldc, M, N, K, splitk, ws);
  if (splitk > 1) hipLaunchKernelGGL(x3_reduce_kernel, dim3(tiles * 16), dim3(256), 0, s, ws, bias, C, ldc, M, N, splitk);
}

}  // namespace

extern "C" int ruart_gemm_x3_plan(int M, int N, int K, int* splitk, size_t* ws_bytes) {
  if (M <= 0 || N <= 0 || K <= 0) return (int)hipErrorInvalidValue;
  const int s = pick_splitk(M, N, K);
  const int tiles = ((M + XBM - 1) / XBM) * ((N + XBN - 1) / XBN);
  if (splitk) *splitk = s;
  if (ws_bytes) *ws_bytes = s > 1 ? (size_t)tiles * s * XBM * XBN * sizeof(float) : 0;
  return 0;
}

extern "C" int ruart_gemm_x3(const float* A, long long sam, long long sak, const float* B, long long sbk, long long sbn,
                             const float* bias, float* C, int ldc, int M, int N, int K, float* ws, size_t ws_bytes, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || ldc < N) return (int)hipErrorInvalidValue;
  const int amode = (sak == 1) ? 0 : (sam == 1 ? 1 : -1);
  const int bmode = (sbk == 1) ? 0 : (sbn == 1 ? 1 : -1);
  if (amode < 0 || bmode < 0) return (int)hipErrorInvalidValue;        // one unit stride per operand
  int splitk = pick_splitk(M, N, K);
  const int tiles = ((M + XBM - 1) / XBM) * ((N + XBN - 1) / XBN);
  if (splitk > 1 && (!ws || ws_bytes < (size_t)tiles * splitk * XBM * XBN * sizeof(float))) return (int)hipErrorInvalidValue;
  // 16-byte vector loads need an aligned base and a non-unit stride that is a multiple of 4 floats
  const bool va = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (((amode == 0 ? sam : sak) & 3) == 0);
  const bool vb = ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && (((bmode == 0 ? sbn : sbk) & 3) == 0);
  hipStream_t s = (hipStream_t)stream;
#define X3(AM, BM_)                                                                                                               \
  do {                                                                                                                            \
    if (va && vb) launch_x3<AM, BM_, true, true>(A, sam, sak, B, sbk, sbn, bias, C, ldc, M, N, K, splitk, ws, s);       \
    else if (va) launch_x3<AM, BM_, true, false>(A, sam, sak, B, sbk, sbn, bias, C, ldc, M, N, K, splitk, ws, s);       \
    else if (vb) launch_x3<AM, BM_, false, true>(A, sam, sak, B, sbk, sbn, bias, C, ldc, M, N, K, splitk, ws, s);       \
    else launch_x3<AM, BM_, false, false>(A, sam, sak, B, sbk, sbn, bias, C, ldc, M, N, K, splitk, ws, s);              \
  } while (0)
  if (amode == 0 && bmode == 0) X3(0, 0);
  else if (amode == 0 && bmode == 1) X3(0, 1);
  else if (amode == 1 && bmode == 0) X3(1, 0);
  else X3(1, 1);
#undef X3
  RUART_CHECK_LAUNCH();
  return 0;
}
