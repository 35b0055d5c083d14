// Round 6 probe: what does the chip sustain on a PURE stream of matrix instructions at its power cap, per instruction type?  One workgroup of
// 512 threads per CU slot (2 waves per SIMD), every wave issues independent MFMAs on RANDOM register operands for a fixed count; per variant:
// time, TFLOP/s, the shader clock held (s_memtime / s_memrealtime).  Socket power is sampled from outside (tools/r06_mfma_power.sh).
//   variants: f16 16x16x32, f16 32x32x16, bf16 16x16x32, fp8 (e4m3, block-scaled) 16x16x128, fp8 32x32x64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(512, 2) void spin(const int* __restrict__ seed, float* __restrict__ sink, int iters, unsigned long long* __restrict__ clk) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  // random operand bits per lane (finite: exponent fields masked to moderate values)
  i32x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    unsigned x = seed[(tid * 16 + i) & 0xfffff], y = seed[(tid * 16 + 8 + i) & 0xfffff];
    if (V <= 2) { x = (x & 0x83ff83ffu) | 0x38003800u; y = (y & 0x83ff83ffu) | 0x38003800u; }       // f16 / bf16: |v| in [0.5, 1) x sign
    else { x = (x & 0x87878787u) | 0x30303030u; y = (y & 0x87878787u) | 0x30303030u; }               // e4m3: moderate exponents
    a[i] = (int)x;
    b[i] = (int)y;
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float out = 0.f;
  if (V == 0 || V == 2) {                      // 16x16x32: eight independent accumulators
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const i32x4 a0 = {a[0], a[1], a[2], a[3]}, a1 = {a[4], a[5], a[6], a[7]}, b0 = {b[0], b[1], b[2], b[3]}, b1 = {b[4], b[5], b[6], b[7]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (V == 0) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, (i & 1) ? a1 : a0), __builtin_bit_cast(f16x8, (i & 2) ? b1 : b0), acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, (i & 1) ? a0 : a1), __builtin_bit_cast(f16x8, (i & 2) ? b0 : b1), acc[i], 0, 0, 0);
        } else {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (i & 1) ? a1 : a0), __builtin_bit_cast(bf16x8, (i & 2) ? b1 : b0), acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (i & 1) ? a0 : a1), __builtin_bit_cast(bf16x8, (i & 2) ? b0 : b1), acc[i], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) out += acc[i][0] + acc[i][3];
  } else if (V == 1) {                         // f16 32x32x16: four independent accumulators (64 VGPRs)
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const i32x4 a0 = {a[0], a[1], a[2], a[3]}, a1 = {a[4], a[5], a[6], a[7]}, b0 = {b[0], b[1], b[2], b[3]}, b1 = {b[4], b[5], b[6], b[7]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, (i & 1) ? a1 : a0), __builtin_bit_cast(f16x8, (i & 2) ? b1 : b0), acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, (i & 1) ? a0 : a1), __builtin_bit_cast(f16x8, (i & 2) ? b0 : b1), acc[i], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out += acc[i][0] + acc[i][15];
  } else if (V == 3) {                         // fp8 16x16x128, scale 2^0
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4((i & 1) ? a : b, (i & 2) ? b : a, acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) out += acc[i][0] + acc[i][3];
  } else {                                     // fp8 32x32x64
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4((i & 1) ? a : b, (i & 2) ? b : a, acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out += acc[i][0] + acc[i][15];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (out == 12345.678f) sink[tid] = out;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 4.0;
  const int only = argc > 2 ? atoi(argv[2]) : -1;
  int* seed; float* sink; unsigned long long* clk;
  const int grid = 256 * 1;                     // one 512-thread workgroup per CU = 2 waves per SIMD
  CK(hipMalloc(&seed, (1 << 20) * 4)); CK(hipMalloc(&sink, (size_t)grid * 512 * 4)); CK(hipMalloc(&clk, grid * 16));
  std::vector<int> h(1 << 20);
  srand(7);
  for (auto& v : h) v = (rand() << 16) ^ rand();
  CK(hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const char* names[5] = {"f16 16x16x32", "f16 32x32x16", "bf16 16x16x32", "fp8 16x16x128 (scaled)", "fp8 32x32x64 (scaled)"};
  // MACs per wave and iteration: 16x16x32 x 16 = 131072; 32x32x16 x 8 = 131072; 16x16x128 x 8 = 262144; 32x32x64 x 4 = 262144
  const double macs[5] = {131072., 131072., 131072., 262144., 262144.};
  for (int v = 0; v < 5; ++v) {
    if (only >= 0 && v != only) continue;
    const int iters = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double total_ms = 0; int launches = 0; float last = 0;
    while (total_ms < secs * 1e3) {
      CK(hipEventRecord(e0));
      switch (v) {
        case 0: hipLaunchKernelGGL(spin<0>, dim3(grid), dim3(512), 0, 0, seed, sink, iters, clk); break;
        case 1: hipLaunchKernelGGL(spin<1>, dim3(grid), dim3(512), 0, 0, seed, sink, iters, clk); break;
        case 2: hipLaunchKernelGGL(spin<2>, dim3(grid), dim3(512), 0, 0, seed, sink, iters, clk); break;
        case 3: hipLaunchKernelGGL(spin<3>, dim3(grid), dim3(512), 0, 0, seed, sink, iters, clk); break;
        default: hipLaunchKernelGGL(spin<4>, dim3(grid), dim3(512), 0, 0, seed, sink, iters, clk); break;
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&last, e0, e1));
      total_ms += last; ++launches;
    }
    std::vector<unsigned long long> c(grid * 2);
    CK(hipMemcpy(c.data(), clk, grid * 16, hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0;
    for (int i = 0; i < grid; ++i) { cyc += c[2 * i]; rt += c[2 * i + 1]; }
    const double ghz = cyc / (rt * 10.0);      // s_memrealtime ticks at 100 MHz
    const double flops = 2.0 * macs[v] * iters * (double)grid * 8;
    printf("%-24s last launch %.2f ms = %.0f TFLOP/s, shader clock %.3f GHz, MFMA pipe busy %.1f %% of its cycles  (%d launches)\n", names[v], last,
           flops / (last * 1e-3) / 1e12, ghz, 100.0 * (v >= 3 ? 8.0 * 32 : 16.0 * 16) * iters * (v == 1 ? 1.0 : 1.0) / (cyc / grid) * 2, launches);
    fflush(stdout);
  }
  return 0;
}
