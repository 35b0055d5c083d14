// Round 6 probe: is v_cvt_scalef32_pk_fp8_f32 (gfx950) bit-for-bit the sequence the split stores use today -
//   e4m3( clamp( x * 2^s, -448, 448 ) )   via v_mul_f32 + v_med3_f32 + v_cvt_pk_fp8_f32
// - when its scale operand is 2^-s and the clamp is applied to x at 448 / 2^s?  (First run, without a clamp in front of it: the scaled
// conversion does NOT saturate - an overflow gives the NaN code 0x7f where the clamped sequence gives 0x7e = 448.)  Sweeps every f32 bit pattern of a coarse grid plus edge cases (overflow, subnormals of the target
// format, halfway cases, NaN / inf) for s = 0 and s = 11.  Build + run: hipcc --offload-arch=gfx950 tools/r06_cvt_probe.hip -o build/r06_cvt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef short s2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned ref_pack2(float a, float b, float scale) {
  const float x = __builtin_amdgcn_fmed3f(a * scale, -448.0f, 448.0f), y = __builtin_amdgcn_fmed3f(b * scale, -448.0f, 448.0f);
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(x, y, w, false);
  return (unsigned)w & 0xffffu;
}
// the candidate: clamp in the UNscaled domain (448 / 2^s is exact), the scale folded into the conversion - no v_mul_f32
__device__ __forceinline__ unsigned new_pack2(float a, float b, float inv_scale) {
  const float lim = 448.0f * inv_scale;
  const float x = __builtin_amdgcn_fmed3f(a, -lim, lim), y = __builtin_amdgcn_fmed3f(b, -lim, lim);
  s2 w = {0, 0};
  w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, x, y, inv_scale, false);
  return (unsigned)__builtin_bit_cast(int, w) & 0xffffu;
}
__global__ void sweep(unsigned long long* out, unsigned* first_bad) {
  // bit patterns: every 2^7-th f32 (33.5 M patterns) plus their neighbours +-1
  const unsigned long long n = 1ull << 25;
  unsigned long long bad0 = 0, bad11 = 0, nan_diff = 0;
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
    for (int d = -1; d <= 1; ++d) {
      const unsigned bits = (unsigned)(i << 7) + (unsigned)d;
      const float x = __builtin_bit_cast(float, bits);
      const float y = -x * 0.37f;
      const unsigned r0 = ref_pack2(x, y, 1.0f), n0 = new_pack2(x, y, 1.0f);
      const unsigned r1 = ref_pack2(x, y, 2048.0f), n1 = new_pack2(x, y, 1.0f / 2048.0f);
      const bool isn = x != x;
      if (r0 != n0) { if (isn) ++nan_diff; else { ++bad0; if (atomicCAS(&first_bad[0], 0u, 1u) == 0u) { first_bad[1] = bits; first_bad[2] = r0; first_bad[3] = n0; first_bad[4] = 0; } } }
      if (r1 != n1) { if (isn) ++nan_diff; else { ++bad11; if (atomicCAS(&first_bad[8], 0u, 1u) == 0u) { first_bad[9] = bits; first_bad[10] = r1; first_bad[11] = n1; first_bad[12] = 11; } } }
    }
  }
  atomicAdd(&out[0], bad0);
  atomicAdd(&out[1], bad11);
  atomicAdd(&out[2], nan_diff);
}
int main() {
  unsigned long long* out; unsigned* fb;
  hipMalloc(&out, 64); hipMalloc(&fb, 128);
  hipMemset(out, 0, 64); hipMemset(fb, 0, 128);
  hipLaunchKernelGGL(sweep, dim3(4096), dim3(256), 0, 0, out, fb);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
  unsigned long long h[3]; unsigned f[32];
  hipMemcpy(h, out, 24, hipMemcpyDeviceToHost); hipMemcpy(f, fb, 128, hipMemcpyDeviceToHost);
  printf("patterns 3 x 2^25 (x, and y = -0.37 x): differences shift 0: %llu, shift 11: %llu, NaN-input differences: %llu\n", h[0], h[1], h[2]);
  for (int k = 0; k < 2; ++k) if (f[8 * k]) { float x; unsigned b = f[8 * k + 1]; memcpy(&x, &b, 4); printf("  first difference (shift %u): x = %g (0x%08x): mul+med3+cvt 0x%04x, cvt_scalef32 0x%04x\n", f[8 * k + 4], x, b, f[8 * k + 2], f[8 * k + 3]); }
  return 0;
}
