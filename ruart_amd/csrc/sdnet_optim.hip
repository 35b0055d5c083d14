// Optimizer step of the hot path in three launches: global gradient norm + clip coefficient (Models/SDNetTrainer.py:366,
// torch.nn.utils.clip_grad_norm_) and the Adamax update (:367, torch.optim.Adamax) over ALL trainable tensors at once.
// torch's own path is ~20 multi-tensor launches and three passes over the gradients (norm, scale, update); here the clip
// coefficient stays on the device and is applied inside the update, and embedding rows that the trainer re-pins after every
// step (rows >= tune_partial, :369-373) are left out of the update (their gradients still count in the norm, as in the
// reference).  Tensors are addressed through device tables of pointers; work is cut into chunks of 8192 elements
// (tensor index, start, count), one workgroup per chunk.  All sums run in a fixed order: deterministic.
#include "common.h"
#include "ruart_hip.h"

#define OPT_CHUNK 8192

__global__ __launch_bounds__(256) void gradnorm_partial_kernel(const float* const* __restrict__ grads, const int* __restrict__ c_tensor,
                                                               const int* __restrict__ c_start, const int* __restrict__ c_count,
                                                               float* __restrict__ partial) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  const float* g = grads[c_tensor[c]] + c_start[c];
  const int n = c_count[c];
  float s = 0.f;
  for (int i = threadIdx.x * 4; i < n; i += 1024) {
    if (i + 3 < n) {
      const f32x4_t v = *reinterpret_cast<const f32x4_t*>(g + i);
      s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    } else {
      for (int j = i; j < n; ++j) s += g[j] * g[j];
    }
  }
  s = block_sum<4>(s, red);
  if (threadIdx.x == 0) partial[c] = s;
}

// out[0] = total norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))
__global__ __launch_bounds__(256) void gradnorm_final_kernel(const float* __restrict__ partial, int n, float max_norm,
                                                             const float* __restrict__ extra_sq, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = block_sum<4>(s, red);
  if (threadIdx.x == 0) {
    if (extra_sq) s += extra_sq[0];           // squared norm of gradients that are not in the chunk list (dp.GradSync.pinned_sq)
    const float norm = sqrtf(s);
    out[0] = norm;
    out[1] = fminf(1.0f, max_norm / (norm + 1e-6f));
  }
}

// torch.optim.Adamax (single-tensor form):  m += (1 - b1) (g - m);  u = max(b2 u, |g| + eps);  p -= clr m / u
__global__ __launch_bounds__(256) void adamax_update_kernel(float* const* __restrict__ params, const float* const* __restrict__ grads,
                                                            float* const* __restrict__ exp_avg, float* const* __restrict__ exp_inf,
                                                            const int* __restrict__ c_tensor, const int* __restrict__ c_start,
                                                            const int* __restrict__ c_count, const float* __restrict__ coef_ptr,
                                                            const float* __restrict__ clr_t, float one_minus_b1, float b2, float eps) {
  const int c = blockIdx.x, t = c_tensor[c], s0 = c_start[c], n = c_count[c];
  const float coef = coef_ptr ? coef_ptr[1] : 1.0f;
  const float clr = clr_t[t];                   // lr / (1 - beta1^step) with the tensor's OWN step count, as torch keeps it
  float* p = params[t] + s0;
  const float* g = grads[t] + s0;
  float* m = exp_avg[t] + s0;
  float* u = exp_inf[t] + s0;
  for (int i = threadIdx.x * 4; i < n; i += 1024) {
    if (i + 3 < n) {
      f32x4_t pv = *reinterpret_cast<f32x4_t*>(p + i), mv = *reinterpret_cast<f32x4_t*>(m + i), uv = *reinterpret_cast<f32x4_t*>(u + i);
      const f32x4_t gv = *reinterpret_cast<const f32x4_t*>(g + i);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gr = gv[r] * coef;
        mv[r] = mv[r] + one_minus_b1 * (gr - mv[r]);
        uv[r] = fmaxf(uv[r] * b2, fabsf(gr) + eps);
        pv[r] = pv[r] - clr * (mv[r] / uv[r]);
      }
      *reinterpret_cast<f32x4_t*>(p + i) = pv;
      *reinterpret_cast<f32x4_t*>(m + i) = mv;
      *reinterpret_cast<f32x4_t*>(u + i) = uv;
    } else {
      for (int j = i; j < n; ++j) {
        const float gr = g[j] * coef;
        const float mj = m[j] + one_minus_b1 * (gr - m[j]);
        const float uj = fmaxf(u[j] * b2, fabsf(gr) + eps);
        m[j] = mj;
        u[j] = uj;
        p[j] = p[j] - clr * (mj / uj);
      }
    }
  }
}

extern "C" int ruart_grad_norm_clip(const float* const* grads, const int* c_tensor, const int* c_start, const int* c_count, int n_chunks,
                                    float max_norm, float* partial, float* norm_coef, const float* extra_sq, void* stream) {
  RUART_ENTRY();
  if (n_chunks <= 0 || !grads || !partial || !norm_coef) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(gradnorm_partial_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, grads, c_tensor, c_start, c_count,
                     partial);
  hipLaunchKernelGGL(gradnorm_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n_chunks, max_norm, extra_sq, norm_coef);
  RUART_CHECK_LAUNCH();
  return 0;
}

extern "C" int ruart_adamax_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_inf,
                                 const int* c_tensor, const int* c_start, const int* c_count, int n_chunks, const float* norm_coef,
                                 const float* clr, float beta1, float beta2, float eps, void* stream) {
  RUART_ENTRY();
  if (n_chunks <= 0 || !clr || !params || !grads || !exp_avg || !exp_inf) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(adamax_update_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_inf, c_tensor,
                     c_start, c_count, norm_coef, clr, 1.0f - beta1, beta2, eps);
  RUART_CHECK_LAUNCH();
  return 0;
}
