// Round 6 probe: do vector loads of DIFFERENT cache policies return in issue order on gfx950?
//
// The pooling kernel removed in commit 728930f issued one default-policy load (a word's (mu, rstd) pairs), then 8-12 non-temporal row
// loads, then `s_waitcnt vmcnt(N)` with N = the number of nt loads, then v_readlane of the first load's register.  That wait is correct
// if and only if loads return in issue order across cache policies.  This program issues exactly that sequence from inline asm (so the
// compiler cannot add a wait), with the first load's destination POISONED beforehand and copied out right behind the counted wait:
//   form A: default-policy load from a COLD 1 GiB buffer (HBM miss), then N nt loads from a 1 MiB buffer every wave re-reads (L2 hits);
//   form B: the policies swapped (nt load cold, default-policy loads hot).
// A poisoned capture = the wait let the wave through before the oldest load had landed = out-of-order return.  Run alone and beside a
// second stream that streams 2 GiB through the memory system.  Build + run: tools/r06_probe.sh; output profiles/r06_load_order_probe.log.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define POISON 0xDEADBEEFu

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// cold[i] == i ^ 0x5A5A5A5A for every dword i of the big buffer; hot[] is arbitrary
template <int FORM>
__global__ __launch_bounds__(256) void probe(const unsigned* __restrict__ cold, unsigned cold_dwords, const u32x4* __restrict__ hot,
                                             unsigned hot_vecs, int iters, unsigned salt, unsigned long long* __restrict__ counts) {
  const unsigned gtid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long bad_early = 0, bad_late = 0, sink = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned idx = hash32(gtid * 977u + it * 7919u + salt) % cold_dwords;                 // a cold line per lane
    const unsigned* pc = cold + idx;
    const u32x4* ph = hot + (hash32(gtid + it * 31u + salt) % (hot_vecs - 8 * 64)) ;            // eight hot 16-byte pieces, 1 KiB apart
    unsigned first = POISON, cap;
    u32x4 r0, r1, r2, r3, r4, r5, r6, r7;
    if (FORM == 0) {
      asm volatile(
          "s_nop 4\n\t"
          "global_load_dword %0, %10, off\n\t"
          "global_load_dwordx4 %2, %11, off nt\n\t"
          "global_load_dwordx4 %3, %11, off offset:1024 nt\n\t"
          "global_load_dwordx4 %4, %11, off offset:2048 nt\n\t"
          "global_load_dwordx4 %5, %11, off offset:3072 nt\n\t"
          "global_load_dwordx4 %6, %12, off nt\n\t"
          "global_load_dwordx4 %7, %12, off offset:1024 nt\n\t"
          "global_load_dwordx4 %8, %12, off offset:2048 nt\n\t"
          "global_load_dwordx4 %9, %12, off offset:3072 nt\n\t"
          "s_waitcnt vmcnt(8)\n\t"
          "v_mov_b32 %1, %0\n\t"
          "s_waitcnt vmcnt(0)\n\t"
          : "+v"(first), "=&v"(cap), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
          : "v"(pc), "v"(ph), "v"(ph + 256)
          : "memory");
    } else {
      asm volatile(
          "s_nop 4\n\t"
          "global_load_dword %0, %10, off nt\n\t"
          "global_load_dwordx4 %2, %11, off\n\t"
          "global_load_dwordx4 %3, %11, off offset:1024\n\t"
          "global_load_dwordx4 %4, %11, off offset:2048\n\t"
          "global_load_dwordx4 %5, %11, off offset:3072\n\t"
          "global_load_dwordx4 %6, %12, off\n\t"
          "global_load_dwordx4 %7, %12, off offset:1024\n\t"
          "global_load_dwordx4 %8, %12, off offset:2048\n\t"
          "global_load_dwordx4 %9, %12, off offset:3072\n\t"
          "s_waitcnt vmcnt(8)\n\t"
          "v_mov_b32 %1, %0\n\t"
          "s_waitcnt vmcnt(0)\n\t"
          : "+v"(first), "=&v"(cap), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
          : "v"(pc), "v"(ph), "v"(ph + 256)
          : "memory");
    }
    const unsigned want = idx ^ 0x5A5A5A5Au;
    bad_early += cap != want;            // what the register held right behind the counted wait
    bad_late += first != want;           // ... and after vmcnt(0) (a wrong value here would be a wrong load, not an early read)
    sink += r0[0] + r1[1] + r2[2] + r3[3] + r4[0] + r5[1] + r6[2] + r7[3];
  }
  if (bad_early) atomicAdd(&counts[0], bad_early);
  if (bad_late) atomicAdd(&counts[1], bad_late);
  if (sink == 0x123456789ull) counts[3] = sink;      // (keeps the hot loads alive)
  atomicAdd(&counts[2], (unsigned long long)iters);
}

__global__ void fill_cold(unsigned* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i ^ 0x5A5A5A5Au;
}
__global__ void hog(float4* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = p[i];
    v.x += 1.0f;
    p[i] = v;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main() {
  const size_t cold_dwords = (size_t)1 << 28;          // 1 GiB
  const size_t hot_vecs = (size_t)1 << 16;             // 1 MiB of 16-byte pieces
  const size_t hog_vecs = (size_t)1 << 27;             // 2 GiB
  unsigned* cold; u32x4* hot; float4* hogbuf; unsigned long long* counts;
  CK(hipMalloc(&cold, cold_dwords * 4));
  CK(hipMalloc(&hot, hot_vecs * 16));
  CK(hipMalloc(&hogbuf, hog_vecs * 16));
  CK(hipMalloc(&counts, 64));
  CK(hipMemset(hot, 1, hot_vecs * 16));
  CK(hipMemset(hogbuf, 0, hog_vecs * 16));
  hipLaunchKernelGGL(fill_cold, dim3(4096), dim3(256), 0, 0, cold, cold_dwords);
  CK(hipDeviceSynchronize());
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  const int grid = 256 * 8, iters = 64;
  int rc = 0;
  for (int with_hog = 0; with_hog < 2; ++with_hog)
    for (int form = 0; form < 2; ++form) {
      unsigned long long h[4] = {0, 0, 0, 0}, tot[3] = {0, 0, 0};
      for (int rep = 0; rep < 20; ++rep) {
        CK(hipMemsetAsync(counts, 0, 64, s0));
        if (with_hog) hipLaunchKernelGGL(hog, dim3(2048), dim3(256), 0, s1, hogbuf, hog_vecs);
        if (form == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, s0, cold, (unsigned)cold_dwords, hot, (unsigned)hot_vecs, iters, 1000u * rep + 17u, counts);
        else hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, s0, cold, (unsigned)cold_dwords, hot, (unsigned)hot_vecs, iters, 1000u * rep + 17u, counts);
        CK(hipStreamSynchronize(s0));
        CK(hipStreamSynchronize(s1));
        CK(hipMemcpy(h, counts, 32, hipMemcpyDeviceToHost));
        tot[0] += h[0]; tot[1] += h[1]; tot[2] += h[2];
      }
      printf("form %c (%s first, then 8 %s loads, s_waitcnt vmcnt(8))%s: %llu sequences, first load not landed behind the wait: %llu, wrong after vmcnt(0): %llu\n",
             form ? 'B' : 'A', form ? "nt load of a cold line" : "default-policy load of a cold line", form ? "default-policy hot" : "nt hot",
             with_hog ? " beside a 2 GiB read-modify-write stream" : " alone", tot[2], tot[0], tot[1]);
      if (tot[0] || tot[1]) rc = 1;
    }
  printf(rc ? "RESULT: out-of-order return observed\n" : "RESULT: every first load had landed behind the counted wait: loads of both cache policies return in issue order\n");
  return 0;
}
