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
// Tile 128x128x32 (4 waves of 64x64, two workgroups per CU) or 256x256x32 (8 waves of 128x64, one per CU) chosen per
// problem, register-staged loads (the split happens on the way to LDS), two LDS stages.  A K-contiguous operand is stored [row][32 k] with two rows per
// 128-byte line and the XOR swizzle of gemm.hip (conflict-free ds_read_b128); an M/N-contiguous operand is stored as it
// comes, [k][128 m], and its fragments are fetched with ds_read_b64_tr_b16 (the hardware transpose), so neither layout
// needs a transposing store.  Small outputs with a long reduction (the weight gradients: e.g. 500x125 over K = 6400) are
// split along K over up to 64 workgroups; every slice parks its partial tile in a workspace and a second, fully parallel
// launch adds the slices in slice order (+ bias) - a deterministic sum, no float atomics, no inter-workgroup hand-off.
// (Measured alternative: the last workgroup of a tile to finish adds the slices in the same launch, agent-scope release /
// acquire around a ticket counter.  Correct, but every workgroup pays the L2 write-back of the release: 39 -> 76 us on the
// 6400 x 250 x 1800 projection, 20 -> 35 us on 2560 x 250 x 800.  The second launch costs 5-9 us.)
#include <cstdlib>
#include "common.h"
#include "ruart_hip.h"

#define XBK 32
// Square tiles of TM = 128 (4 waves, two workgroups per CU) or 256 (8 waves, one workgroup per CU: twice the flops per
// byte pulled from L2 - the 128 tile is bound by that delivery, ~35 GB/s per CU, at 220 TFLOP/s-equivalent).
template <int TM> struct XT {
  static constexpr int THREADS = TM * 2;                 // 64 x (2 waves along M) x (TM/64 waves along N)
  static constexpr int WN = TM / 64;                     // waves along N, 64 columns each
  static constexpr int JT = TM / 32;                     // 16-row MFMA tiles per wave along M (wave = TM/2 rows)
  static constexpr int TRS = TM * 2 + 16;                // row stride of the [k][rows] image (+16 B pad)
  static constexpr int ARR = 32 * TRS;                   // bytes per LDS operand image (>= TM rows * 64 B)
};

typedef __attribute__((__vector_size__(4 * sizeof(short)))) short xtr16x4_t;
typedef __attribute__((address_space(3))) xtr16x4_t* xtr_ptr_t;

namespace {

__device__ __forceinline__ void split_bf16(float x, bf16_t& hi, bf16_t& lo) {
  hi = (bf16_t)x;
  lo = (bf16_t)(x - (float)hi);
}

// byte offset of 16-byte chunk `chunk` (8 k values) of row r in the swizzled [row][32 k] image: two rows share a 128-byte
// line, chunk XOR line as in gemm.hip - conflict-free for ds_read_b128 fragment reads.
__device__ __forceinline__ int kc_off(int r, int chunk /*0..3*/) {
  const int line = r >> 1;
  return line * 128 + ((((r & 1) * 4 + chunk) ^ (line & 7)) << 4);
}

// MODE 0: the operand's K index is contiguous in memory (element (row, k) at P[row*srow + k]);
// MODE 1: its row index is contiguous (element (row, k) at P[k*sk + row]).
// VW: floats per load instruction along the contiguous index - 4 (16-byte aligned base, stride % 4 == 0, extent % 4 == 0), 2 (8-byte
// analogue: the trunk's 250-wide activations) or 1.  EVERY load is unconditional: out-of-range rows / k are CLAMPED to the last valid
// vector and zeroed by a select afterwards.  (Rounds 1-3 branched around each load - vector load inside the matrix, per-element scalar
// loads at its edges and for every operand whose rows are not 16-byte aligned; hipcc waits vmcnt(0) behind every load it has to branch
// around (cdna_hip_programming.md, projection GEMM item 4c), so an operand with 250-float rows was fetched as 16 DEPENDENT L2 round
// trips per thread and K step.)
// MASKED (compile time): a variational-dropout mask is fused into the operand.  A variational mask holds only 0 and 1/(1-p), so it
// travels as ONE BYTE per element (0 = dropped) plus the scalar 1/(1-p): MODE 0: element (row, k) is kept iff S[(row / rpm) * K + k],
// MODE 1: iff S[(k / rpm) * rows + row]  (rpm = time steps that share one mask row); kept elements are multiplied by `scale`, exactly
// what x * mask computes.  The bytes of a load's VW elements are fetched as one 32- / 16- / 8-bit load from the same (clamped)
// coordinates into their own registers (4 per operand at VW = 4) and applied in store() - a select right behind the loads would
// make the wave wait for them in front of the MFMAs they are meant to overlap.  (Rounds 2-3 read an fp32 mask as per-element scalar
// loads behind a run-time `if (S)`: 1.5-2x slower than a separate multiply pass, so the product materialised x * mask; an fp32 mask
// in vector registers - 16 more per register set - spilled in the 256 tile and in the two-set 128 tile.)
template <int TM, int MODE, int VW, bool MASKED = false>
struct Stager {
  float v[16];
  unsigned mb[MASKED ? 16 / VW : 1];   // VW keep-bytes per word
  unsigned ok;      // bit e: element e is inside the matrix.  Applied in store(): a select right behind the load would make the wave
                    // wait for it there, in front of the MFMAs the load is meant to overlap

  __device__ __forceinline__ static void ldv(const float* p, float* out) {
    if (VW == 4) {
      const f32x4_t t = *reinterpret_cast<const f32x4_t*>(p);
      out[0] = t[0]; out[1] = t[1]; out[2] = t[2]; out[3] = t[3];
    } else if (VW == 2) {
      typedef float f32x2v __attribute__((ext_vector_type(2)));
      const f32x2v t = *reinterpret_cast<const f32x2v*>(p);
      out[0] = t[0]; out[1] = t[1];
    } else {
      out[0] = *p;
    }
  }
  __device__ __forceinline__ static unsigned ldm(const unsigned char* p) {
    if (VW == 4) return *reinterpret_cast<const unsigned*>(p);
    if (VW == 2) return *reinterpret_cast<const unsigned short*>(p);
    return *p;
  }

  // x / rpm for 0 <= x < 2^20, rpm >= 1: one multiply by the reciprocal (exact: the quotient's distance to the next integer is
  // >= 0.5 / rpm, the product's rounding error < x 2^-22)
  __device__ __forceinline__ static int div_rpm(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

  // once per tile (MASKED, MODE 0): offsets of the mask rows of this thread's four operand rows - they do not change along K
  // (kept by the caller: the two register sets of the prefetch share them; a mask has < 2^31 elements)
  __device__ __forceinline__ static void mask_rows(int (&mo)[4], int r0, int rows, int K, int tid, int rpm) {
    const float inv = 1.0f / (float)rpm;
#pragma unroll
    for (int p = 0; p < 4; ++p) mo[p] = div_rpm(min(r0 + p * (TM / 4) + (tid >> 3), rows - 1), inv) * K;
  }

  // rows = M or N (limit of the row index), r0 = first row of the tile, k0 = first k of this step.
  // Addresses are the wave-uniform base pointer + ONE unsigned 32-bit element offset per load (the saddr form of global_load): an
  // operand spans < 2^30 elements (checked on the host).  Four 64-bit row pointers per operand and register set - what the pointer
  // form `P + row * pitch` costs - were the registers the 256 tile spilled inside its K loop once the mask words joined them.
  // ROWS past the matrix are clamped and NOT zeroed: they only reach output rows / columns that nothing stores (the epilogue and the
  // split-K sum test m < M, n < N), and they are copies of real rows, so they are finite.  Only k >= K has to contribute zeros, and
  // that happens in the last K step alone: KFULL (the step lies inside K: no clamp of k, no validity bits, no select in store()) is
  // what every other step runs - the per-element bounds select was 110 of the 233 VALU instructions of a K step (round 4 counters:
  // these kernels are bound by their issue slots, profiles/HISTORY.md round 5 (9)).
  template <bool KFULL>
  __device__ __forceinline__ void load(const float* __restrict__ P, long srow, long sk, int r0, int rows, int k0, int K, int tid,
                                       const unsigned char* __restrict__ S, int rpm, const int (&mo)[4]) {
    ok = KFULL ? 0xffffu : 0u;
    if (MODE == 0) {
      const int kq = (tid & 7) * 4;                   // 4 consecutive k
      const unsigned pitch = (unsigned)srow;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int r = r0 + p * (TM / 4) + (tid >> 3), k = k0 + kq;
        const unsigned ro = (unsigned)min(r, rows - 1) * pitch;
#pragma unroll
        for (int j = 0; j < 4; j += VW) {
          const int kk = k + j, kc = KFULL ? kk : min(kk, K - VW);
          ldv(P + (ro + (unsigned)kc), &v[p * 4 + j]);
          if (!KFULL && kk < K) ok |= ((1u << VW) - 1u) << (p * 4 + j);           // extent % VW == 0: inside or outside as a whole
          if (MASKED) mb[(p * 4 + j) / VW] = ldm(S + (unsigned)(mo[p] + kc));
        }
      }
    } else {
      const int m4 = (tid % (TM / 4)) * 4;            // 4 consecutive rows
      const float inv = 1.0f / (float)rpm;
      const unsigned pitch = (unsigned)sk;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int k = k0 + 2 * (tid / (TM / 4)) + (p & 1) + 16 * (p >> 1), r = r0 + m4;
        const int kc = KFULL ? k : min(k, K - 1);
        const unsigned ko = (unsigned)kc * pitch;
        const unsigned so = MASKED ? (unsigned)(div_rpm(kc, inv) * rows) : 0u;
        if (!KFULL && k < K) ok |= 0xfu << (p * 4);
#pragma unroll
        for (int j = 0; j < 4; j += VW) {
          const int rrc = min(r + j, rows - VW);
          ldv(P + (ko + (unsigned)rrc), &v[p * 4 + j]);
          if (MASKED) mb[(p * 4 + j) / VW] = ldm(S + (so + (unsigned)rrc));
        }
      }
    }
  }

  template <bool KFULL>
  __device__ __forceinline__ void store(char* hi_img, char* lo_img, int tid, bool with_lo = true, float scale = 1.f) const {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      bf16x4_t h, l;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bf16_t a, b;
        const int e = p * 4 + i;
        float x = v[e];
        if (MASKED) x = ((mb[e / VW] >> (8 * (e % VW))) & 0xffu) ? x * scale : 0.f;
        if (!KFULL) x = ((ok >> e) & 1u) ? x : 0.f;
        split_bf16(x, a, b);
        h[i] = a;
        l[i] = b;
      }
      int off;
      if (MODE == 0) {
        const int r = p * (TM / 4) + (tid >> 3), kq = tid & 7;           // 4 k values = 8 bytes inside chunk kq >> 1
        off = kc_off(r, kq >> 1) + (kq & 1) * 8;
      } else {
        const int k = 2 * (tid / (TM / 4)) + (p & 1) + 16 * (p >> 1);    // 4 rows = 8 bytes of k-row k
        off = k * XT<TM>::TRS + (tid % (TM / 4)) * 8;
      }
      *reinterpret_cast<bf16x4_t*>(hi_img + off) = h;
      if (with_lo) *reinterpret_cast<bf16x4_t*>(lo_img + off) = l;
    }
  }
};

// Fragment of one 16-row MFMA tile for lane (fr = lane & 15, fq = lane >> 4): 8 consecutive k = 8 fq .. 8 fq + 7.
// (Tried: the permuted order k = {4g.., 16+4g..} with a 32-mod-256 row stride makes the transposed reads conflict-free, but
// turns the row read into two ds_read_b64 - measured 7 % slower over the trunk's shapes, so the natural order stays.)
template <int TM, int MODE>
__device__ __forceinline__ bf16x8_t frag(const char* img, int row_base, int fr, int fq) {
  if (MODE == 0) {
    return *reinterpret_cast<const bf16x8_t*>(img + kc_off(row_base + fr, fq));
  } else {
    // transposed fetch: lane 4q+p of a 16-lane group supplies k-row q, rows 4p..4p+3, and receives row (lane & 15), k-rows 0..3
    const char* base = img + (8 * fq + (fr >> 2)) * XT<TM>::TRS + (row_base + (fr & 3) * 4) * 2;
    union { struct { xtr16x4_t a, b; } s; bf16x8_t f; } u;
    u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((xtr_ptr_t)base);
    u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((xtr_ptr_t)(base + 4 * XT<TM>::TRS));
    return u.f;
  }
}

// One workgroup's share of a product: output tile `id / splitk`, K slice `id % splitk` (the body of gemm_x3_kernel and of the grouped
// weight-gradient kernel below)
template <int TM, int AMODE, int BMODE, int VWA, int VWB, int NP, int MSK = 0>    // MSK: 0 no fused mask, 1 a_scale, 2 b_scale
__device__ __forceinline__ void x3_tile(const float* __restrict__ A, long sam, long sak, const float* __restrict__ B, long sbk, long sbn,
                                        const float* __restrict__ bias, float* __restrict__ C, int ldc, int M, int N, int K, int splitk,
                                        float* __restrict__ ws, const unsigned char* __restrict__ a_scale, const unsigned char* __restrict__ b_scale,
                                        float keep_scale, const float* __restrict__ c_scale, int rpm, const float* __restrict__ R, int ldr,
                                        int act, int id) {
  typedef XT<TM> T;
  constexpr int ARR = T::ARR, JT = T::JT;
  extern __shared__ __attribute__((aligned(1024))) char smem[];          // 2 stages x (A_hi, A_lo, B_hi, B_lo)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / T::WN, wn = wave % T::WN;                        // 2 x WN waves; a wave owns (TM/2) rows x 64 columns
  const int fr = lane & 15, fq = lane >> 4;
  const int ntn = (N + TM - 1) / TM;
  const int tile = id / splitk, slice = id - tile * splitk;             // the slices of a tile are neighbours: same XCD / L2
  const int m0 = (tile / ntn) * TM, n0 = (tile % ntn) * TM;
  const int ksteps = (K + XBK - 1) / XBK;
  const int per = (ksteps + splitk - 1) / splitk;
  const int kbeg = slice * per, kend = min(ksteps, kbeg + per);

  f32x4_t acc[4][JT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < JT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // operand A: rows = m (stride sam) in MODE 0 / k-rows of stride sak in MODE 1; operand B: "rows" = n
  const long a_srow = sam, a_sk = sak, b_srow = sbn, b_sk = sbk;
  // one K step on LDS stage `st`: fragments of four m-tiles at a time (keeps the fragment registers at 40)
  auto compute = [&](const char* st) {
#pragma unroll
    for (int jh = 0; jh < JT / 4; ++jh) {
      bf16x8_t ah[4], al[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ah[j] = frag<TM, AMODE>(st, wm * (TM / 2) + (jh * 4 + j) * 16, fr, fq);
        if (NP == 3) al[j] = frag<TM, AMODE>(st + ARR, wm * (TM / 2) + (jh * 4 + j) * 16, fr, fq);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8_t bh = frag<TM, BMODE>(st + 2 * ARR, wn * 64 + i * 16, fr, fq);
        bf16x8_t bl = bh;
        if (NP == 3) bl = frag<TM, BMODE>(st + 3 * ARR, wn * 64 + i * 16, fr, fq);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4_t& c = acc[i][jh * 4 + j];
          if (NP == 3) {
            c = mfma_16x16x32(bl, ah[j], c);          // small terms first
            c = mfma_16x16x32(bh, al[j], c);
          }
          c = mfma_16x16x32(bh, ah[j], c);
        }
      }
    }
  };
  int moa[4] = {0, 0, 0, 0}, mob[4] = {0, 0, 0, 0};
  if (MSK == 1 && AMODE == 0) Stager<TM, AMODE, VWA, true>::mask_rows(moa, m0, M, K, tid, rpm);
  if (MSK == 2 && BMODE == 0) Stager<TM, BMODE, VWB, true>::mask_rows(mob, n0, N, K, tid, rpm);
#ifndef RUART_X3_PF
#define RUART_X3_PF 2
#endif
#if RUART_X3_PF == 2
  // Register-staged prefetch TWO K steps deep on the 128 tile (round 4): the trunk's products are short (K = 250 .. 1 800: 8-57 steps of
  // 32) and a step's MFMA work (~0.4 us at two workgroups per CU) is shorter than an L2 round trip, so with one step of lookahead every
  // store waited for its own loads (PMC: waves parked in s_waitcnt / s_barrier 45 % of their cycles).  Two register sets alternate with
  // the two LDS stages; a set's loads are issued right after the barrier that follows its store and are consumed two steps later.
  // (The 256 tile - 8 waves x 128 accumulators - has no registers for a second set.)
  if constexpr (TM == 128) {
    Stager<TM, AMODE, VWA, MSK == 1> sa0, sa1;
    Stager<TM, BMODE, VWB, MSK == 2> sb0, sb1;
    sa0.template load<false>(A, a_srow, a_sk, m0, M, kbeg * XBK, K, tid, a_scale, rpm, moa);
    sb0.template load<false>(B, b_srow, b_sk, n0, N, kbeg * XBK, K, tid, b_scale, rpm, mob);
    sa1.template load<false>(A, a_srow, a_sk, m0, M, (kbeg + 1) * XBK, K, tid, a_scale, rpm, moa);
    sb1.template load<false>(B, b_srow, b_sk, n0, N, (kbeg + 1) * XBK, K, tid, b_scale, rpm, mob);
    // No branch inside a pair loop: the loads past the slice's end are issued all the same (their addresses are clamped into the
    // matrix, their elements are never stored), so hipcc can count vmcnt exactly - a load behind a run-time condition makes it
    // assume the fewest outstanding loads at the join and wait for the YOUNGER set as well.
    char* const st0 = smem;
    char* const st1 = smem + 4 * ARR;
    int t = kbeg;
    // sched_barrier(0) pins the order  store | barrier | issue the loads of two steps ahead | MFMAs : left alone, hipcc sinks a
    // set's loads below the MFMAs they are meant to run beside and hoists the NEXT store (with its vmcnt wait) above them.
    // The pair loop exists twice: while all four K steps an iteration touches (stores t, t + 1; loads t + 2, t + 3) lie inside K it runs
    // the KFULL forms - no k clamp, no validity bits, no select per element -, the last iterations the general ones.
#define X3_PAIR(KF)                                                                                       \
    {                                                                                                     \
      sa0.template store<KF>(st0, st0 + ARR, tid, NP == 3, keep_scale);                                   \
      sb0.template store<KF>(st0 + 2 * ARR, st0 + 3 * ARR, tid, NP == 3, keep_scale);                     \
      __syncthreads();                /* one barrier per step: the other stage was last read two steps ago */ \
      __builtin_amdgcn_sched_barrier(0);                                                                  \
      sa0.template load<KF>(A, a_srow, a_sk, m0, M, (t + 2) * XBK, K, tid, a_scale, rpm, moa);            \
      sb0.template load<KF>(B, b_srow, b_sk, n0, N, (t + 2) * XBK, K, tid, b_scale, rpm, mob);            \
      __builtin_amdgcn_sched_barrier(0);                                                                  \
      compute(st0);                                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                  \
      sa1.template store<KF>(st1, st1 + ARR, tid, NP == 3, keep_scale);                                   \
      sb1.template store<KF>(st1 + 2 * ARR, st1 + 3 * ARR, tid, NP == 3, keep_scale);                     \
      __syncthreads();                                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                  \
      sa1.template load<KF>(A, a_srow, a_sk, m0, M, (t + 3) * XBK, K, tid, a_scale, rpm, moa);            \
      sb1.template load<KF>(B, b_srow, b_sk, n0, N, (t + 3) * XBK, K, tid, b_scale, rpm, mob);            \
      __builtin_amdgcn_sched_barrier(0);                                                                  \
      compute(st1);                                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                  \
    }
    const int kfull = K / XBK;                        // steps [0, kfull) lie inside K
    for (; t + 1 < kend && t + 3 < kfull; t += 2) X3_PAIR(true)
    for (; t + 1 < kend; t += 2) X3_PAIR(false)
#undef X3_PAIR
    if (t < kend) {                                   // odd step count: the last step's operands are in set 0
      sa0.template store<false>(st0, st0 + ARR, tid, NP == 3, keep_scale);
      sb0.template store<false>(st0 + 2 * ARR, st0 + 3 * ARR, tid, NP == 3, keep_scale);
      __syncthreads();
      compute(st0);
    }
  } else
#endif
  {
    Stager<TM, AMODE, VWA, MSK == 1> sa;
    Stager<TM, BMODE, VWB, MSK == 2> sb;
    if (kbeg < kend) {
      sa.template load<false>(A, a_srow, a_sk, m0, M, kbeg * XBK, K, tid, a_scale, rpm, moa);
      sb.template load<false>(B, b_srow, b_sk, n0, N, kbeg * XBK, K, tid, b_scale, rpm, mob);
    }
    const int kfull = K / XBK;
    int t = kbeg;
    for (; t + 1 < kend && t + 1 < kfull; ++t) {        // this step and the next lie inside K: the KFULL forms (see Stager)
      char* st = smem + ((t - kbeg) & 1) * (4 * ARR);
      sa.template store<true>(st, st + ARR, tid, NP == 3, keep_scale);
      sb.template store<true>(st + 2 * ARR, st + 3 * ARR, tid, NP == 3, keep_scale);
      __syncthreads();                                  // one barrier per step: the other stage was last read two steps ago
      sa.template load<true>(A, a_srow, a_sk, m0, M, (t + 1) * XBK, K, tid, a_scale, rpm, moa);
      sb.template load<true>(B, b_srow, b_sk, n0, N, (t + 1) * XBK, K, tid, b_scale, rpm, mob);
      compute(st);
    }
    for (; t < kend; ++t) {
      char* st = smem + ((t - kbeg) & 1) * (4 * ARR);
      sa.template store<false>(st, st + ARR, tid, NP == 3, keep_scale);
      sb.template store<false>(st + 2 * ARR, st + 3 * ARR, tid, NP == 3, keep_scale);
      __syncthreads();
      if (t + 1 < kend) {
        sa.template load<false>(A, a_srow, a_sk, m0, M, (t + 1) * XBK, K, tid, a_scale, rpm, moa);
        sb.template load<false>(B, b_srow, b_sk, n0, N, (t + 1) * XBK, K, tid, b_scale, rpm, mob);
      }
      compute(st);
    }
  }

  if (splitk > 1) {
    // park the partial tile, thread-major ([i][j][tid] x 4 floats): x3_reduce_kernel re-reads it with the same mapping
    float* slab = ws + ((size_t)tile * splitk + slice) * (TM * TM);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < JT; ++j) *reinterpret_cast<f32x4_t*>(slab + ((i * JT + j) * T::THREADS + tid) * 4) = acc[i][j];
    return;
  }
  // lane owns rows m = .. + j*16 + fr and four consecutive columns n = .. + i*16 + fq*4 + r
  const uintptr_t bits = reinterpret_cast<uintptr_t>(C) | (R ? reinterpret_cast<uintptr_t>(R) : 0) | (bias ? reinterpret_cast<uintptr_t>(bias) : 0) |
                         (c_scale ? reinterpret_cast<uintptr_t>(c_scale) : 0);
  // whole-vector epilogue (bias, mask, residual, store as 16-byte accesses, none behind a per-element condition) when every pitch
  // and base allows it; the element-wise form handles the matrix's last columns and odd pitches
  const bool vec_all = (bits & 15) == 0 && (ldc & 3) == 0 && (!R || (ldr & 3) == 0) && (!c_scale || (N & 3) == 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + wn * 64 + i * 16 + fq * 4;
    if (vec_all && n + 3 < N) {
      f32x4_t bv = {0.f, 0.f, 0.f, 0.f};
      if (bias) bv = *reinterpret_cast<const f32x4_t*>(bias + n);
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const int m = m0 + wm * (TM / 2) + j * 16 + fr;
        if (m >= M) continue;
        f32x4_t v = acc[i][j] + bv;
        if (act == RUART_ACT_GELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] * 0.5f * (1.0f + erff(v[r] * 0.70710678118654752440f));
        }
        if (c_scale) v *= *reinterpret_cast<const f32x4_t*>(c_scale + (size_t)(m / rpm) * N + n);
        if (R) v += *reinterpret_cast<const f32x4_t*>(R + (size_t)m * ldr + n);
        *reinterpret_cast<f32x4_t*>(C + (size_t)m * ldc + n) = v;
      }
      continue;
    }
    f32x4_t bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = (n + r < N) ? bias[n + r] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      const int m = m0 + wm * (TM / 2) + j * 16 + fr;
      if (m >= M || n >= N) continue;
      f32x4_t v = acc[i][j] + bv;
      if (act == RUART_ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] * 0.5f * (1.0f + erff(v[r] * 0.70710678118654752440f));
      }
      if (c_scale) {
        const float* sp = c_scale + (size_t)(m / rpm) * N + n;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < N) v[r] *= sp[r];
      }
      if (R) {
        const float* rp = R + (size_t)m * ldr + n;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < N) v[r] += rp[r];
      }
      float* dst = C + (size_t)m * ldc + n;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < N) dst[r] = v[r];
    }
  }
}

template <int TM, int AMODE, int BMODE, int VWA, int VWB, int NP = 3, int MSK = 0>     // NP = 1: the hi.hi product only (plain bf16)
__global__ __launch_bounds__(TM * 2, 2) void gemm_x3_kernel(const float* __restrict__ A, long sam, long sak, const float* __restrict__ B,
                                                            long sbk, long sbn, const float* __restrict__ bias, float* __restrict__ C,
                                                            int ldc, int M, int N, int K, int splitk, float* __restrict__ ws,
                                                            const unsigned char* __restrict__ a_scale,
                                                            const unsigned char* __restrict__ b_scale, float keep_scale,
                                                            const float* __restrict__ c_scale, int rpm, const float* __restrict__ R,
                                                            int ldr, int act) {
  x3_tile<TM, AMODE, BMODE, VWA, VWB, NP, MSK>(A, sam, sak, B, sbk, sbn, bias, C, ldc, M, N, K, splitk, ws, a_scale, b_scale, keep_scale, c_scale,
                                               rpm, R, ldr, act, xcd_remap(blockIdx.x, gridDim.x));
}

// ---- grouped weight gradients --------------------------------------------------------------------------------------------------
// dW_p (M_p x N_p) = dY_p^T . X_p over the rows of a projection's input: ~45 such products per training step, each a small output
// with a long reduction - alone they fill 10-30 % of the chip for 20-50 us and each drags a split-K reduction launch behind it.  Here
// ALL of a step's weight gradients run as one launch (and one reduction launch): the workgroups of problem p are the id range
// [blk0, blk0 + tiles * splitk) of a 1-D grid, each runs x3_tile exactly as the single-problem kernel would (same tiles, same K
// slices, same summation order: bitwise the same dW).  `accum`: dW += (a weight used twice - deep attention, the shared RNNs - gets
// its second contribution in a second wave of the same kind).
#define X3G_MAX 44
struct X3GProb {
  const float* A;      // dY (rows, M): element (m, k) at A[k * lda + m]
  const float* B;      // X  (rows, N): element (k, n) at B[k * ldb + n]
  float* C;            // dW (M, N), row stride ldc
  float* ws;           // this problem's split-K slabs
  int lda, ldb, ldc, M, N, K, splitk, blk0, red0, accum;
};
struct X3GArgs {
  X3GProb p[X3G_MAX];
  int n;
};
template <bool VECA, bool VECB>
__global__ __launch_bounds__(256, 2) void gemm_x3_grouped_tn_kernel(const X3GArgs g) {
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  int q = 0;
  while (q + 1 < g.n && id >= g.p[q + 1].blk0) ++q;          // wave-uniform scan over <= 44 problems
  const X3GProb& P = g.p[q];
  const float* R = (P.accum && P.splitk == 1) ? P.C : nullptr;
  x3_tile<128, 1, 1, VECA ? 4 : 1, VECB ? 4 : 1, 3>(P.A, 1, P.lda, P.B, P.ldb, 1, nullptr, P.C, P.ldc, P.M, P.N, P.K, P.splitk, P.ws, nullptr, nullptr,
                                    1.f, nullptr, 1, R, P.ldc, RUART_ACT_NONE, id - P.blk0);
}

// second launch of a split-K product: one thread per 4 output floats adds the slices in slice order and writes C (+ bias)
template <int TM>
__device__ __forceinline__ void x3_reduce_body(const float* __restrict__ ws, const float* __restrict__ bias, float* __restrict__ C, int ldc,
                                               int M, int N, int splitk, const float* __restrict__ c_scale, int rpm,
                                               const float* __restrict__ R, int ldr, int act, int blk) {
  typedef XT<TM> T;
  constexpr int JT = T::JT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / T::WN, wn = wave % T::WN, fr = lane & 15, fq = lane >> 4;
  const int ntn = (N + TM - 1) / TM;
  const int tile = blk / (4 * JT), ij = blk % (4 * JT), i = ij / JT, j = ij % JT;
  const int m = (tile / ntn) * TM + wm * (TM / 2) + j * 16 + fr, n = (tile % ntn) * TM + wn * 64 + i * 16 + fq * 4;
  if (m >= M || n >= N) return;
  const float* p = ws + (size_t)tile * splitk * (TM * TM) + (ij * T::THREADS + tid) * 4;
  // slices in slice order (the sum's order is part of the contract: deterministic, equal to the grouped form), eight loads in flight
  f32x4_t s = {0.f, 0.f, 0.f, 0.f};
  int sl = 0;
  for (; sl + 8 <= splitk; sl += 8) {
    f32x4_t a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const f32x4_t*>(p + (size_t)(sl + u) * (TM * TM));
#pragma unroll
    for (int u = 0; u < 8; ++u) s += a[u];
  }
  {
    f32x4_t a[8];                                      // tail: clamped loads, masked adds - no load behind a branch
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const f32x4_t*>(p + (size_t)min(sl + u, splitk - 1) * (TM * TM));
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (sl + u < splitk) s += a[u];
  }
  // epilogue: whole-vector form when the four columns are inside the matrix and every row pitch / base allows 16-byte access
  const uintptr_t bits = reinterpret_cast<uintptr_t>(C) | (R ? reinterpret_cast<uintptr_t>(R) : 0) | (bias ? reinterpret_cast<uintptr_t>(bias) : 0) |
                         (c_scale ? reinterpret_cast<uintptr_t>(c_scale) : 0);
  const bool vec = n + 3 < N && (bits & 15) == 0 && (ldc & 3) == 0 && (!R || (ldr & 3) == 0) && (!c_scale || (N & 3) == 0);
  if (vec) {
    f32x4_t v = s;
    if (bias) v += *reinterpret_cast<const f32x4_t*>(bias + n);
    if (act == RUART_ACT_GELU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = v[r] * 0.5f * (1.0f + erff(v[r] * 0.70710678118654752440f));
    }
    if (c_scale) v *= *reinterpret_cast<const f32x4_t*>(c_scale + (size_t)(m / rpm) * N + n);
    if (R) v += *reinterpret_cast<const f32x4_t*>(R + (size_t)m * ldr + n);
    *reinterpret_cast<f32x4_t*>(C + (size_t)m * ldc + n) = v;
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (n + r < N) {
      float v = s[r] + (bias ? bias[n + r] : 0.f);
      if (act == RUART_ACT_GELU) v = v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
      if (c_scale) v *= c_scale[(size_t)(m / rpm) * N + n + r];
      if (R) v += R[(size_t)m * ldr + n + r];
      C[(size_t)m * ldc + n + r] = v;
    }
}

template <int TM>
__global__ __launch_bounds__(TM * 2) void x3_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                                                           float* __restrict__ C, int ldc, int M, int N, int splitk,
                                                           const float* __restrict__ c_scale, int rpm, const float* __restrict__ R,
                                                           int ldr, int act) {
  x3_reduce_body<TM>(ws, bias, C, ldc, M, N, splitk, c_scale, rpm, R, ldr, act, blockIdx.x);
}

// the slice sums of every split problem of a group, one launch (problem p: blocks [red0, red0 + tiles * 16))
__global__ __launch_bounds__(256) void x3_reduce_grouped_kernel(const X3GArgs g) {
  const int id = blockIdx.x;
  int q = 0;
  while (q + 1 < g.n && id >= g.p[q + 1].red0) ++q;
  const X3GProb& P = g.p[q];
  if (P.splitk <= 1) return;                           // (problems without a split own no reduction blocks; defensive)
  x3_reduce_body<128>(P.ws, nullptr, P.C, P.ldc, P.M, P.N, P.splitk, nullptr, 1, P.accum ? P.C : nullptr, P.ldc, RUART_ACT_NONE, id - P.red0);
}

struct Plan { int tm, tiles, splitk; };

// Tile size and K split.  The 256 tile wins when it still fills the chip (>= ~200 workgroups after the split); products with a
// small output and a long reduction (weight gradients) are split along K, at least 4 steps of 32 per slice.
Plan make_plan(int M, int N, int K, int amode, int bmode) {
  const int ksteps = (K + XBK - 1) / XBK;
  // workgroups a split-K plan aims at.  128 since round 5 (448 before: "two small workgroups per CU"): beside the encoder pass what a trunk
  // product costs the step is the CUs it occupies, not its own latency - 23.58-23.64 ms per step at 128 / 224 against 23.85 at 448, 24.2 at 64,
  // 28.1 without any split (the trunk's chain becomes the longer one); profiles/r05_knob_sweep.log.  RUART_X3_FILL overrides (experiments).
  static const int fill = [] { const char* e = getenv("RUART_X3_FILL"); return e ? atoi(e) : 128; }();
  static const int nosplit = [] { const char* e = getenv("RUART_X3_NOSPLIT_TILES"); return e ? atoi(e) : 160; }();
  auto split_for = [&](int tiles) {
    if (tiles >= nosplit || ksteps < 16) return 1;
    int s = (fill + tiles - 1) / tiles;                // aim at ~2 workgroups of the small tile per CU
    if (s > ksteps / 4) s = ksteps / 4;
    if (s > 64) s = 64;
    return s < 1 ? 1 : s;
  };
  const int t256 = ((M + 255) / 256) * ((N + 255) / 256), t128 = ((M + 127) / 128) * ((N + 127) / 128);
  Plan p;
  static const int fill256 = [] { const char* e = getenv("RUART_X3_T256_FILL"); return e ? atoi(e) : 256; }();      // (experiments)
  static const int min256 = [] { const char* e = getenv("RUART_X3_T256_MIN"); return e ? atoi(e) : 192; }();
  int s256 = 1;
  if (t256 < 200 && ksteps >= 16) {
    s256 = (fill256 + t256 - 1) / t256;
    if (s256 > ksteps / 8) s256 = ksteps / 8;
    if (s256 < 1) s256 = 1;
  }
  // padding waste of the big tile must stay moderate, and it needs enough workgroups
  const double waste = (double)t256 * 65536.0 / ((double)M * N);
  // (measured: with a transposed-read B operand the big tile re-reads B's fragments twice per step and loses: 231 vs 203 us)
  if (bmode == 0 && (long long)M * N >= 4096LL * 1024 && t256 * s256 >= min256 && waste < 1.25) {
    (void)amode;
    p.tm = 256; p.tiles = t256; p.splitk = s256;
  } else {
    p.tm = 128; p.tiles = t128; p.splitk = split_for(t128);
  }
  return p;
}

template <int TM, int AM, int BM_, int VA, int VB, int NP = 3, int MSK = 0>
void launch_x3(const float* A, long sam, long sak, const float* B, long sbk, long sbn, const float* bias, float* C, int ldc, int M,
               int N, int K, const Plan& p, float* ws, const unsigned char* a_scale, const unsigned char* b_scale, float keep_scale,
               const float* c_scale, int rpm,
               const float* R, int ldr, int act, hipStream_t s) {
  auto kern = gemm_x3_kernel<TM, AM, BM_, VA, VB, NP, MSK>;
  constexpr int lds = 2 * 4 * XT<TM>::ARR;
  static bool done = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds), true);
  (void)done;
  hipLaunchKernelGGL(kern, dim3(p.tiles * p.splitk), dim3(XT<TM>::THREADS), lds, s, A, sam, sak, B, sbk, sbn, bias, C, ldc, M, N, K,
                     p.splitk, ws, a_scale, b_scale, keep_scale, c_scale, rpm, R, ldr, act);
  if (p.splitk > 1)
    hipLaunchKernelGGL(x3_reduce_kernel<TM>, dim3(p.tiles * 4 * XT<TM>::JT), dim3(XT<TM>::THREADS), 0, s, ws, bias, C, ldc, M, N, p.splitk,
                       c_scale, rpm, R, ldr, act);
}

}  // namespace

extern "C" int ruart_gemm_x3_plan(int M, int N, int K, int a_k_contiguous, int b_k_contiguous, int* splitk, size_t* ws_bytes) {
  RUART_ENTRY();
  if (M <= 0 || N <= 0 || K <= 0) return (int)hipErrorInvalidValue;
  const Plan p = make_plan(M, N, K, a_k_contiguous ? 0 : 1, b_k_contiguous ? 0 : 1);
  if (splitk) *splitk = p.splitk;
  if (ws_bytes) *ws_bytes = p.splitk > 1 ? (size_t)p.tiles * p.splitk * p.tm * p.tm * sizeof(float) : 0;
  return 0;
}

namespace {
template <int NP>
int gemm_xn(const float* A, long long sam, long long sak, const float* B, long long sbk, long long sbn,
            const float* bias, const float* residual, int ldr, int act, float* C, int ldc, int M, int N, int K,
            float* ws, size_t ws_bytes, const unsigned char* a_scale, const unsigned char* b_scale, float keep_scale,
            const float* c_scale, int rows_per_scale_row, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || ldc < N) return (int)hipErrorInvalidValue;
  if ((residual && ldr < N) || (act != RUART_ACT_NONE && act != RUART_ACT_GELU)) return (int)hipErrorInvalidValue;
  const float* R = residual;
  const int rpm = rows_per_scale_row > 0 ? rows_per_scale_row : 1;
  if ((a_scale && sak != 1) || (b_scale && sbn != 1)) return (int)hipErrorInvalidValue;    // see the header
  const int amode = (sak == 1) ? 0 : (sam == 1 ? 1 : -1);
  const int bmode = (sbk == 1) ? 0 : (sbn == 1 ? 1 : -1);
  if (amode < 0 || bmode < 0) return (int)hipErrorInvalidValue;        // one unit stride per operand
  {   // the kernels address an operand as base + a 32-bit element offset
    const long long ea = amode == 0 ? (long long)(M - 1) * sam + K : (long long)(K - 1) * sak + M;
    const long long eb = bmode == 0 ? (long long)(N - 1) * sbn + K : (long long)(K - 1) * sbk + N;
    if (ea >= (1LL << 30) || eb >= (1LL << 30) || sam < 0 || sak < 0 || sbk < 0 || sbn < 0) return (int)hipErrorInvalidValue;
  }
  Plan p = make_plan(M, N, K, amode, bmode);
  if (p.splitk > 1 && (!ws || ws_bytes < (size_t)p.tiles * p.splitk * p.tm * p.tm * sizeof(float))) p.splitk = 1;   // no room: unsplit
  // floats per load along each operand's contiguous index (Stager): base alignment, row pitch and extent must all allow it; a fused
  // mask is read with the operand's width (its pitch is the operand's extent)
  auto width = [](const float* base, long pitch, int extent, const unsigned char* mask) {
    const uintptr_t bits = reinterpret_cast<uintptr_t>(base), mbits = mask ? reinterpret_cast<uintptr_t>(mask) : 0;   // (mask: 1 B / element)
    if ((bits & 15) == 0 && (mbits & 3) == 0 && (pitch & 3) == 0 && (extent & 3) == 0) return 4;
    if ((bits & 7) == 0 && (mbits & 1) == 0 && (pitch & 1) == 0 && (extent & 1) == 0) return 2;
    return 1;
  };
  const int vwa = width(A, amode == 0 ? sam : sak, amode == 0 ? K : M, a_scale);
  const int vwb = width(B, bmode == 0 ? sbn : sbk, bmode == 0 ? K : N, b_scale);
  if (p.tm == 256 && (vwa == 1 || vwb == 1)) {          // the big tile is built for the 16- and 8-byte forms only
    p.tm = 128;
    p.tiles = ((M + 127) / 128) * ((N + 127) / 128);
    p.splitk = 1;
  }
  hipStream_t s = (hipStream_t)stream;
#define X3L(TM, AM, BM_, VA, VB) \
  launch_x3<TM, AM, BM_, VA, VB, NP>(A, sam, sak, B, sbk, sbn, bias, C, ldc, M, N, K, p, ws, a_scale, b_scale, keep_scale, c_scale, rpm, R, ldr, act, s)
#define X3W(TM, AM, BM_)                                         \
  do {                                                           \
    if (vwa == 4 && vwb == 4) X3L(TM, AM, BM_, 4, 4);            \
    else if (vwa == 4 && vwb == 2) X3L(TM, AM, BM_, 4, 2);       \
    else if (vwa == 2 && vwb == 4) X3L(TM, AM, BM_, 2, 4);       \
    else if (vwa == 2 && vwb == 2) X3L(TM, AM, BM_, 2, 2);       \
    else if (TM == 256) return (int)hipErrorInvalidValue;        \
    else X3S(AM, BM_);                                           \
  } while (0)
#define X3S(AM, BM_)                                             \
  do {                                                           \
    if (vwa == 4) X3L(128, AM, BM_, 4, 1);                       \
    else if (vwa == 2) X3L(128, AM, BM_, 2, 1);                  \
    else if (vwb == 4) X3L(128, AM, BM_, 1, 4);                  \
    else if (vwb == 2) X3L(128, AM, BM_, 1, 2);                  \
    else X3L(128, AM, BM_, 1, 1);                                \
  } while (0)
  if (a_scale || b_scale) {
    // One operand mask fused into the loads (ops._Linear: the forward's (x * mask) W^T with a_scale, dW = dY^T (x * mask) with b_scale).
    // Both operands are read with the narrower of their two vector widths (one instantiation per width, not per pair).
    if (a_scale && b_scale) return (int)hipErrorInvalidValue;
    const int vw = vwa < vwb ? vwa : vwb;
#define X3M(TM, AM, BM_, VW_, MSK_) \
  launch_x3<TM, AM, BM_, VW_, VW_, NP, MSK_>(A, sam, sak, B, sbk, sbn, bias, C, ldc, M, N, K, p, ws, a_scale, b_scale, keep_scale, c_scale, rpm, R, ldr, act, s)
    if (a_scale && amode == 0 && bmode == 0) {
      if (p.tm == 256 && vw >= 2) {
        if (vw == 4) X3M(256, 0, 0, 4, 1); else X3M(256, 0, 0, 2, 1);
      } else {
        if (p.tm == 256) { p.tm = 128; p.tiles = ((M + 127) / 128) * ((N + 127) / 128); p.splitk = 1; }
        if (vw == 4) X3M(128, 0, 0, 4, 1); else if (vw == 2) X3M(128, 0, 0, 2, 1); else X3M(128, 0, 0, 1, 1);
      }
    } else {
      if (p.tm == 256) { p.tm = 128; p.tiles = ((M + 127) / 128) * ((N + 127) / 128); p.splitk = 1; }
      if (b_scale && amode == 1 && bmode == 1) {
        if (vw == 4) X3M(128, 1, 1, 4, 2); else if (vw == 2) X3M(128, 1, 1, 2, 2); else X3M(128, 1, 1, 1, 2);
      } else if (a_scale && amode == 0 && bmode == 1) X3M(128, 0, 1, 1, 1);
      else if (b_scale && amode == 0 && bmode == 1) X3M(128, 0, 1, 1, 2);
      else return (int)hipErrorInvalidValue;           // (a_scale needs sak == 1, b_scale sbn == 1: checked above)
    }
#undef X3M
  } else if (p.tm == 256) {                            // (make_plan picks the big tile for a K-contiguous B only)
    if (bmode != 0) return (int)hipErrorInvalidValue;
    if (amode == 0) X3W(256, 0, 0); else X3W(256, 1, 0);
  } else if (amode == 0 && bmode == 0) X3W(128, 0, 0);
  else if (amode == 0 && bmode == 1) X3W(128, 0, 1);
  else if (amode == 1 && bmode == 0) X3W(128, 1, 0);
  else X3W(128, 1, 1);
#undef X3S
#undef X3W
#undef X3L
  RUART_CHECK_LAUNCH();
  return 0;
}
}  // namespace

extern "C" int ruart_gemm_x3(const float* A, long long sam, long long sak, const float* B, long long sbk, long long sbn,
                             const float* bias, const float* residual, int ldr, int act, float* C, int ldc, int M, int N, int K,
                             float* ws, size_t ws_bytes, const unsigned char* a_keep, const unsigned char* b_keep, float keep_scale,
                             const float* c_scale, int rows_per_scale_row, void* stream) {
  RUART_ENTRY();
  return gemm_xn<3>(A, sam, sak, B, sbk, sbn, bias, residual, ldr, act, C, ldc, M, N, K, ws, ws_bytes, a_keep, b_keep, keep_scale, c_scale,
                    rows_per_scale_row, stream);
}

extern "C" int ruart_gemm_x1(const float* A, long long sam, long long sak, const float* B, long long sbk, long long sbn,
                             const float* bias, const float* residual, int ldr, int act, float* C, int ldc, int M, int N, int K,
                             float* ws, size_t ws_bytes, const unsigned char* a_keep, const unsigned char* b_keep, float keep_scale,
                             const float* c_scale, int rows_per_scale_row, void* stream) {
  RUART_ENTRY();
  return gemm_xn<1>(A, sam, sak, B, sbk, sbn, bias, residual, ldr, act, C, ldc, M, N, K, ws, ws_bytes, a_keep, b_keep, keep_scale, c_scale,
                    rows_per_scale_row, stream);
}

extern "C" int ruart_gemm_bf16_tn(const float* A, long long sak_rows, const float* B, long long sbk_rows, float* C, int ldc, int M, int N,
                                  int K, float* ws, size_t ws_bytes, void* stream) {
  RUART_ENTRY();
  // A stored (K, M): element (m, k) at A[k * sak_rows + m]; B stored (K, N)
  return gemm_xn<1>(A, 1, sak_rows, B, sbk_rows, 1, nullptr, nullptr, 0, RUART_ACT_NONE, C, ldc, M, N, K, ws, ws_bytes, nullptr, nullptr,
                    1.f, nullptr, 1, stream);
}

// ---- grouped weight gradients: host side ---------------------------------------------------------------------------------------
extern "C" size_t ruart_gemm_x3_tn_grouped_ws(const ruart_x3_tn_problem* probs, int n) {
  size_t total = 0;
  for (int i = 0; i < n; ++i) {
    const Plan p = make_plan(probs[i].M, probs[i].N, probs[i].K, 1, 1);
    if (p.splitk > 1) total += (size_t)p.tiles * p.splitk * 128 * 128 * sizeof(float);
  }
  return total;
}

extern "C" int ruart_gemm_x3_tn_grouped(const ruart_x3_tn_problem* probs, int n, float* ws, size_t ws_bytes, void* stream) {
  RUART_ENTRY();
  if (n <= 0 || !probs) return (int)hipErrorInvalidValue;
  if (ruart_gemm_x3_tn_grouped_ws(probs, n) > ws_bytes) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  // four classes by 16-byte loadability of the two operands; inside a class, launches of at most X3G_MAX problems
  size_t ws_off = 0;
  for (int cls = 0; cls < 4; ++cls) {
    const bool va = cls & 1, vb = cls & 2;
    X3GArgs g;
    g.n = 0;
    int blk = 0, red = 0;
    auto flush = [&]() -> int {
      if (g.n == 0) return 0;
      constexpr int lds = 2 * 4 * XT<128>::ARR;
#define X3G_LAUNCH(VA, VB)                                                                                                  \
  do {                                                                                                                       \
    auto kern = gemm_x3_grouped_tn_kernel<VA, VB>;                                                                           \
    static bool done = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds), true);    \
    (void)done;                                                                                                              \
    hipLaunchKernelGGL(kern, dim3(blk), dim3(256), lds, s, g);                                                               \
  } while (0)
      if (va && vb) X3G_LAUNCH(true, true);
      else if (va) X3G_LAUNCH(true, false);
      else if (vb) X3G_LAUNCH(false, true);
      else X3G_LAUNCH(false, false);
#undef X3G_LAUNCH
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return (int)e;
      if (red > 0) {
        hipLaunchKernelGGL(x3_reduce_grouped_kernel, dim3(red), dim3(256), 0, s, g);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
      }
      g.n = 0;
      blk = red = 0;
      return 0;
    };
    for (int i = 0; i < n; ++i) {
      const ruart_x3_tn_problem& q = probs[i];
      if (!q.A || !q.B || !q.C || q.M <= 0 || q.N <= 0 || q.K <= 0 || q.lda < q.M || q.ldb < q.N || q.ldc < q.N ||
          (long long)q.K * q.lda >= (1LL << 30) || (long long)q.K * q.ldb >= (1LL << 30))
        return (int)hipErrorInvalidValue;
      const bool qa = ((reinterpret_cast<uintptr_t>(q.A) & 15) == 0) && ((q.lda & 3) == 0) && ((q.M & 3) == 0);
      const bool qb = ((reinterpret_cast<uintptr_t>(q.B) & 15) == 0) && ((q.ldb & 3) == 0) && ((q.N & 3) == 0);
      if (qa != va || qb != vb) continue;
      const Plan pl = make_plan(q.M, q.N, q.K, 1, 1);          // (a transposed-read B operand always plans the 128 tile)
      X3GProb& P = g.p[g.n];
      P.A = q.A; P.B = q.B; P.C = q.C;
      P.lda = q.lda; P.ldb = q.ldb; P.ldc = q.ldc;
      P.M = q.M; P.N = q.N; P.K = q.K;
      P.splitk = pl.splitk;
      P.accum = q.accumulate ? 1 : 0;
      P.blk0 = blk;
      P.red0 = red;
      P.ws = nullptr;
      if (pl.splitk > 1) {
        P.ws = ws + ws_off / sizeof(float);
        ws_off += (size_t)pl.tiles * pl.splitk * 128 * 128 * sizeof(float);
        red += pl.tiles * 4 * XT<128>::JT;
      }
      blk += pl.tiles * pl.splitk;
      if (++g.n == X3G_MAX) {
        // unsplit problems own no reduction blocks: give the scan a sentinel by leaving red0 equal to the next problem's
        const int rc = flush();
        if (rc) return rc;
      }
    }
    const int rc = flush();
    if (rc) return rc;
  }
  return 0;
}
