// Weight-gradient products of the 16-bit trainable encoder:  C[M,N] = P[T,M]^T . Q[T,N]  ("TN": both operands are stored with the
// REDUCTION index - the token row - as their slow axis, exactly as the backward pass holds dY (T x N_out) and X (T x K_in)).
//
// Replaces autograd's  grad_weight = grad_output.t().mm(input)  of every nn.Linear of the reference's trainable BERT
// (Models/Bert/modeling.py:225-227, :261, :287, :300 under Models/SDNet.py:60-63 without LOCK_BERT).  Round 2's first form transposed
// both operands (ruart_transpose16) to feed the NT kernel: 2.2 GB of extra traffic per layer, 9.3 ms of a 74 ms step.
//
// Same 256 x 256 x 64 tile, 8 waves (2 x 4), eight-phase schedule, counted vmcnt and half-wave stagger as gemm_16_nt_256p8 (gemm.hip -
// the schedule is described there); what differs is the LDS image and the fragment reads:
//  * a half-tile is staged as it lies in memory: [64 t-rows][128 columns] (256 B rows), by global_load_lds, wave w filling rows
//    8w .. 8w+7.  Half h of P holds columns wm*128 + h*64 .. +64 for both wave rows wm; half h of Q is the contiguous column block
//    h*128 .. +128, of which wave column wn owns 32 (so a wave's 64 output columns are two runs of 32, 128 apart: whole 256-byte
//    runs per staged row instead of four 64-byte pieces) - every slot dies at the same phase as in the NT kernel.
//  * MFMA fragments want 8 consecutive REDUCTION indices per lane, which here are 8 different LDS rows: ds_read_tr16_b64 reads a
//    4-row x 16-column block per 16 lanes and hands lane i column i.  Two of them (rows 4g.. and 16+4g..) make one K=32 operand; both
//    operands use the same row order, so the products pair up whatever that order is.
//  * one transposed read touches 16 rows x 32 B.  Rows are 256 B apart (all on the same banks), so the 32-byte column index is
//    XOR-swizzled with (row & 7) - applied on the way in by permuting which 16-byte chunk each lane fetches (the DMA writes LDS
//    linearly) - which spreads each half-wave's 8 rows over all 64 banks.
// Output: split over the token rows (blockIdx.y) into fp32 slabs, summed in slice order by ruart_splitk_reduce (deterministic).
#include "common.h"
#include <type_traits>
#include "ruart_hip.h"
#include "gemm_shared.h"

#define BK 64
#define BM4 256
#define BN4 256

typedef short tr16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) tr16x4_t* tr_ptr_t;

template <typename T16>
__global__ __launch_bounds__(512, 2) void gemm_16_tn_256p8(const T16* __restrict__ P, int ldp, const T16* __restrict__ Q, int ldq,
                                                           float* __restrict__ C, int ldc, int M, int N, int T, int order, int tchunk) {
  {
    const int z = blockIdx.y;
    P += (size_t)z * tchunk * ldp;
    Q += (size_t)z * tchunk * ldq;
    T = min(tchunk, T - z * tchunk);
    C += (size_t)z * M * ldc;
  }
  constexpr int kHalf = 64 * 128 * 2;            // 16 KB half-tile: 64 t-rows x 128 columns
  constexpr int kOper = 2 * kHalf;
  constexpr int kBuf = 2 * kOper;
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // 2 * kBuf = 128 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int ntn = N / BN4, ntm = M / BM4;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  int tm, tn;
  if (order == 0) {
    tm = id / ntn;
    tn = id % ntn;
  } else {
    const int per_group = order * ntn;
    const int g = id / per_group, first = g * order;
    const int gsz = min(ntm - first, order);
    const int r = id - g * per_group;
    tm = first + r % gsz;
    tn = r / gsz;
  }
  const int m0 = tm * BM4, n0 = tn * BN4;

  // staging: piece i (0, 1) of a wave is t-rows 8w + 4i .. +3; lane -> (row lane >> 4, LDS chunk lane & 15), fetching the source
  // chunk whose 32-byte column index is the LDS one XOR (row & 7)
  unsigned p_lane[2], q_lane[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 4 * i + (lane >> 4);
    const int c = (lane & 15) ^ (r << 1);
    p_lane[i] = (unsigned)(r * ldp + (c >> 3) * 128 + (c & 7) * 8) * 2;      // bytes
    q_lane[i] = (unsigned)(r * ldq + c * 8) * 2;
  }
  const T16* p_src = P + (size_t)(8 * wave) * ldp + m0;
  const T16* q_src = Q + (size_t)(8 * wave) * ldq + n0;
  const size_t p_t = (size_t)BK * ldp, q_t = (size_t)BK * ldq;
  char* const st_base = smem + wave * 2048;
  auto stage_a = [&](int d, int h, int kt) {
    char* dst = st_base + d * kBuf + h * kHalf;
    const T16* src = p_src + kt * p_t + h * 64;
    dma16(src, p_lane[0], dst);
    dma16(src, p_lane[1], dst + 1024);
  };
  auto stage_w = [&](int d, int h, int kt) {
    char* dst = st_base + d * kBuf + kOper + h * kHalf;
    const T16* src = q_src + kt * q_t + h * 128;
    dma16(src, q_lane[0], dst);
    dma16(src, q_lane[1], dst + 1024);
  };

  f32x4_t acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#ifndef RUART_TN_ABLATE
#define RUART_TN_ABLATE 0
#endif
  // diagnostic builds only (tools/build_variant.sh NAME -DRUART_TN_ABLATE=n): 1 no prefetch in the loop, 2 no fragment reads after the first
  // K-tile, 4 no stagger.  0 in production.  (Issuing a phase's prefetch BEFORE its fragment reads was measured 30 % slower: the reads'
  // issue is the critical path of the segment.)
  constexpr int ab = RUART_TN_ABLATE;
  const int fr = lane & 15, fq = lane >> 4;
  typedef typename Vec8<T16>::type frag_t;
  frag_t af[4][2], wfa[2][2], wfb[2][2];
  const int swz = 4 * (fq & 1) + (fr >> 2);                        // (t-row & 7) of this lane's reads
  const int lane_off = (4 * fq + (fr >> 2)) * 256 + (fr & 3) * 8;
  auto read_frag = [&](const char* base, int sub, int ks) -> frag_t {
    const char* p = base + lane_off + ks * (32 * 256) + ((sub ^ swz) << 5);
    union { struct { tr16x4_t a, b; } s; frag_t f; } u;
    u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)p);
    u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(p + 16 * 256));
    return u.f;
  };
  auto read_a = [&](int d, int h, int j0, int j1) {
    const char* sa = smem + d * kBuf + h * kHalf;
#pragma unroll
    for (int j = j0; j < j1; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[j][ks] = read_frag(sa, wm * 4 + j, ks);
  };
  auto read_w = [&](int d, int h, frag_t (&wf)[2][2]) {
    const char* sw = smem + d * kBuf + kOper + h * kHalf;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) wf[i][ks] = read_frag(sw, wn * 2 + i, ks);
  };
  auto quad = [&](int hc, int hr, frag_t (&wf)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[hc * 2 + i][hr * 4 + j] = mfma_16x16x32(wf[i][ks], af[j][ks], acc[hc * 2 + i][hr * 4 + j]);
    __builtin_amdgcn_s_setprio(0);
  };
  // One K-tile = four phases (gemm.hip describes the schedule).  The fragment reads are spread 16 / 8 / 16 / 8 over the phases instead
  // of the NT kernel's 24 / 8 / 16 / 0 (in transposed-read instructions): phase 3, which has nothing else to fetch, reads the NEXT
  // K-tile's Q-h0 fragments into the register set that held this tile's Q-h1 (dead after phase 2; the two sets swap roles with the
  // LDS buffer), so the longest read segment of the loop - the one every phase's MFMAs wait behind - is a third shorter.  For that read, K-tile t+1's Q-h0 must have landed for
  // BOTH wave groups one barrier earlier than the rest of the tile: the counted wait at the end of phase 2 (5 younger half-tiles may
  // still be in flight) guarantees it before the barrier the other group pairs with.
  auto tile = [&](auto dtag, auto n1tag, auto n2tag, int t) {
    constexpr int D = decltype(dtag)::value;
    constexpr bool N1 = decltype(n1tag)::value, N2 = decltype(n2tag)::value;
    constexpr bool S1 = N1 && !(ab & 1), S2 = N2 && !(ab & 1);
    const bool rd = !(ab & 2) || t == 0;
    auto run = [&](frag_t (&wc)[2][2], frag_t (&wn)[2][2]) {
      // phase 0: quadrant (rows h0, cols h0) - Q-h0 fragments already in wc; prefetch (t+1, P-h1)
      if (rd) read_a(D, 0, 0, 4);
      if (S1) stage_a(D ^ 1, 1, t + 1);
      RUART_BAR();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      quad(0, 0, wc);
      RUART_BAR();
      // phase 1: (rows h0, cols h1); prefetch (t+2, Q-h0)
      if (rd) read_w(D, 1, wn);
      if (S2) stage_w(D, 0, t + 2);
      RUART_BAR();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      quad(1, 0, wn);
      RUART_BAR();
      // phase 2: (rows h1, cols h1); prefetch (t+2, P-h0)
      if (rd) read_a(D, 1, 0, 4);
      if (S2) stage_a(D, 0, t + 2);
      RUART_BAR();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      quad(1, 1, wn);
      if (N2) {
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");    // (t+1, Q-h0) has landed; the five half-tiles issued after it may be in flight
      } else if (N1) {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");     // tail: only (t+1, P-h1) is younger
      }
      RUART_BAR();
      // phase 3: (rows h1, cols h0) - operands in registers; prefetch (t+2, Q-h1); read (t+1, Q-h0) for the next tile's phase 0
      if (N2) {
        if (S2) stage_w(D, 1, t + 2);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // K-tile t+1 complete; the 3 youngest half-tiles stay in flight
      } else if (N1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      RUART_BAR();
      if (N1 && rd) read_w(D ^ 1, 0, wn);
      quad(0, 1, wc);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (free: issued 16 MFMAs ago) the Q-h0 slot is restaged two barriers on
      RUART_BAR();
    };
    if constexpr (D == 0) run(wfa, wfb); else run(wfb, wfa);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using Tt = std::true_type;
  using Ff = std::false_type;

  const int nt = T / BK;                         // even, >= 2
  stage_w(0, 0, 0);
  stage_a(0, 0, 0);
  stage_w(0, 1, 0);
  stage_a(0, 1, 0);
  stage_w(1, 0, 1);
  stage_a(1, 0, 1);
  stage_w(1, 1, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  RUART_BAR();
  read_w(0, 0, wfa);
  if (wave >= 4 && !(ab & 4)) RUART_BAR();       // stagger: waves 4-7 run one barrier behind
  int t = 0;
  for (; t + 2 < nt; t += 2) {
    tile(I0{}, Tt{}, Tt{}, t);
    tile(I1{}, Tt{}, Tt{}, t + 1);
  }
  tile(I0{}, Tt{}, Ff{}, t);
  tile(I1{}, Ff{}, Ff{}, t + 1);
  if (wave < 4 && !(ab & 4)) RUART_BAR();
  RUART_BAR();

  // accumulators -> LDS [m][n] per wave -> 16-byte fp32 stores
  constexpr int ERS = 272;
  char* my = smem + wave * (32 * ERS);
  const int rrow = lane >> 4, rcol = (lane & 15) * 4;
  const int ncol = n0 + (rcol >> 5) * 128 + wn * 32 + (rcol & 31);      // accumulator column block i: half i >> 1, sub-tile i & 1
#pragma unroll
  for (int hh = 0; hh < 4; ++hh) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        *reinterpret_cast<f32x4_t*>(my + (j * 16 + fr) * ERS + (i * 16 + fq * 4) * 4) = acc[i][hh * 2 + j];
    const int mrow = m0 + wm * 128 + hh * 32 + rrow;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr)
      store4(C + (size_t)(mrow + rr * 4) * ldc + ncol, *reinterpret_cast<const f32x4_t*>(my + (rr * 4 + rrow) * ERS + rcol * 4));
  }
}

template <typename T16>
static void launch_tn(const void* P, int ldp, const void* Q, int ldq, float* part, int ldc, int M, int N, int T, int tchunk, hipStream_t s) {
  constexpr int lds = 2 * 2 * BM4 * BK * 2;
  auto kern = gemm_16_tn_256p8<T16>;
  static bool done = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds), true);
  (void)done;
  const int nz = (T + tchunk - 1) / tchunk;
  hipLaunchKernelGGL(kern, dim3((M / BM4) * (N / BN4), nz), dim3(512), lds, s, (const T16*)P, ldp, (const T16*)Q, ldq, part, ldc, M, N, T, 0,
                     tchunk);
}

extern "C" int ruart_gemm_16_tn_splitk(const void* P, int ldp, const void* Q, int ldq, float* part, int ldc, int M, int N, int T, int tchunk,
                                       int in_dtype, void* stream) {
  RUART_ENTRY();
  if (M <= 0 || N <= 0 || T <= 0 || M % BM4 || N % BN4 || T % 128 || tchunk <= 0 || tchunk % 128 || (ldp & 7) || (ldq & 7) || (ldc & 3) ||
      ldp < M || ldq < N || ldc < N || !P || !Q || !part)
    return (int)hipErrorInvalidValue;
  if (in_dtype == RUART_DT_BF16)
    launch_tn<bf16_t>(P, ldp, Q, ldq, part, ldc, M, N, T, tchunk, (hipStream_t)stream);
  else if (in_dtype == RUART_DT_F16)
    launch_tn<f16_t>(P, ldp, Q, ldq, part, ldc, M, N, T, tchunk, (hipStream_t)stream);
  else
    return (int)hipErrorInvalidValue;
  RUART_CHECK_LAUNCH();
  return 0;
}
