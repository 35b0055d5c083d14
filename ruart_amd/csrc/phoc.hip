// PHOC (pyramidal histogram of characters) table builder: one 604-float row per word.
//
//   ruart_phoc_table    Utils/cphoc.c:12-113 (build_phoc) applied to a whole vocabulary at once - the table the reference's
//                       preprocessing fills word by word (Utils/CoQAUtils.py:75-87) and SDNet looks up as `phoc_embed`.
//
// Row layout (cphoc.c:24-103): 36 unigrams [a-z0-9] x the 2 + 3 + 4 + 5 = 14 regions of pyramid levels 2..5 (level-major, then
// region, then character: 504 entries), then the 50 most frequent English bigrams x the 2 regions of level 2 (100 entries).
// Character i of an n-character word occupies [i/n, (i+1)/n]; it is counted in a region when at least half of it lies inside:
// (min(hi, r1) - max(lo, r0)) / (hi - lo) >= 0.5, evaluated in fp32 with exactly the reference's operation order (IEEE
// subtract, subtract, divide) - membership of characters that sit on a region boundary depends on those roundings.
//
// HBM-bound byte work: the row is assembled in LDS (one wave per word, lane = character position) and written out once as
// 151 float4 per word; nothing else leaves the CU.
#include "common.h"
#include "ruart_hip.h"

namespace {

constexpr int kPhocDim = 604, kUni = 36, kUniRows = 14, kBi = 50;

__device__ __constant__ unsigned short kBigrams[kBi] = {
    // "th","he","in","er","an","re","es","on","st","nt","en","at","ed","nd","to","or","ea","ti","ar","te","ng","al","it","as","is",
    // "ha","et","se","ou","of","le","sa","ve","ro","ra","ri","hi","ne","me","de","co","ta","ec","si","ll","so","na","li","la","el"
    // packed as first_char | second_char << 8
    't' | 'h' << 8, 'h' | 'e' << 8, 'i' | 'n' << 8, 'e' | 'r' << 8, 'a' | 'n' << 8, 'r' | 'e' << 8, 'e' | 's' << 8, 'o' | 'n' << 8,
    's' | 't' << 8, 'n' | 't' << 8, 'e' | 'n' << 8, 'a' | 't' << 8, 'e' | 'd' << 8, 'n' | 'd' << 8, 't' | 'o' << 8, 'o' | 'r' << 8,
    'e' | 'a' << 8, 't' | 'i' << 8, 'a' | 'r' << 8, 't' | 'e' << 8, 'n' | 'g' << 8, 'a' | 'l' << 8, 'i' | 't' << 8, 'a' | 's' << 8,
    'i' | 's' << 8, 'h' | 'a' << 8, 'e' | 't' << 8, 's' | 'e' << 8, 'o' | 'u' << 8, 'o' | 'f' << 8, 'l' | 'e' << 8, 's' | 'a' << 8,
    'v' | 'e' << 8, 'r' | 'o' << 8, 'r' | 'a' << 8, 'r' | 'i' << 8, 'h' | 'i' << 8, 'n' | 'e' << 8, 'm' | 'e' << 8, 'd' | 'e' << 8,
    'c' | 'o' << 8, 't' | 'a' << 8, 'e' | 'c' << 8, 's' | 'i' << 8, 'l' | 'l' << 8, 's' | 'o' << 8, 'n' | 'a' << 8, 'l' | 'i' << 8,
    'l' | 'a' << 8, 'e' | 'l' << 8};

__device__ __forceinline__ int unigram_index(unsigned char c) {
  if (c >= 'a' && c <= 'z') return c - 'a';
  if (c >= '0' && c <= '9') return 26 + (c - '0');
  return -1;
}

// at least half of [lo, hi] inside region `region` of `level`?  fp32, reference operation order
__device__ __forceinline__ bool half_inside(float lo, float hi, int region, int level) {
  const float r0 = __fdiv_rn((float)region, (float)level), r1 = __fdiv_rn((float)(region + 1), (float)level);
  const float o0 = fmaxf(lo, r0), o1 = fminf(hi, r1);
  return __fdiv_rn(__fsub_rn(o1, o0), __fsub_rn(hi, lo)) >= 0.5f;
}

__global__ __launch_bounds__(256) void phoc_table_kernel(const unsigned char* __restrict__ chars, const int* __restrict__ offsets,
                                                         int n_words, float* __restrict__ out, int ldo, int* __restrict__ status) {
  __shared__ __attribute__((aligned(16))) float rows[4][kPhocDim];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + wave;
  float* row = rows[wave];
  for (int i = lane; i < kPhocDim; i += 64) row[i] = 0.f;
  __syncthreads();
  if (w < n_words) {
    const int beg = offsets[w], n = offsets[w + 1] - beg;
    const unsigned char* word = chars + beg;
    const float fn = (float)n;
    for (int i = lane; i < n; i += 64) {
      const float lo = __fdiv_rn((float)i, fn), hi = __fdiv_rn((float)(i + 1), fn);
      const int ci = unigram_index(word[i]);
      if (ci < 0) {
        if (status) atomicCAS(status, 0, w + 1);           // the reference raises on a character outside [a-z0-9]
        continue;
      }
      int base = 0;
      for (int level = 2; level < 6; ++level) {
        for (int region = 0; region < level; ++region)
          if (half_inside(lo, hi, region, level)) row[(base + region) * kUni + ci] = 1.f;
        base += level;
      }
      if (i + 1 < n) {
        const unsigned short pair = (unsigned short)(word[i] | (word[i + 1] << 8));
        int bi = -1;
        for (int k = 0; k < kBi; ++k)
          if (kBigrams[k] == pair) {
            bi = k;
            break;
          }
        if (bi >= 0) {
          const float hi2 = __fdiv_rn((float)(i + 2), fn);
          for (int region = 0; region < 2; ++region)
            if (half_inside(lo, hi2, region, 2)) row[kUniRows * kUni + region * kBi + bi] = 1.f;
        }
      }
    }
  }
  __syncthreads();
  if (w < n_words) {
    float* dst = out + (size_t)w * ldo;
    for (int i = lane; i < kPhocDim / 4; i += 64) store4(dst + i * 4, *reinterpret_cast<const f32x4_t*>(row + i * 4));
  }
}

}  // namespace

extern "C" int ruart_phoc_table(const unsigned char* chars, const int* offsets, int n_words, float* out, int ldo, int* status,
                                void* stream) {
  RUART_ENTRY();
  if (n_words <= 0 || ldo < kPhocDim || (ldo & 3)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(phoc_table_kernel, dim3(ceil_div(n_words, 4)), dim3(256), 0, (hipStream_t)stream, chars, offsets, n_words, out,
                     ldo, status);
  RUART_CHECK_LAUNCH();
  return 0;
}
