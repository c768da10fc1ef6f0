// Likelihood-side kernels: Hessian square root of softmax cross-entropy, loss sums.
// Replaces GGNInterface._get_functional_hessian (laplace/curvature/curvature.py:366-373) and the
// `factor * lossfunc(f, y)` evaluations (curvature.py:408,417; curvlinops.py:106).
#include "lk_common.h"

namespace lk {

// One wave per sample (C <= 64 fast path keeps the row in registers; larger C loops).
// S[c][n][j] = (j == c ? sqrt(p_c) : 0) - p_j * sqrt(p_c)
__global__ __launch_bounds__(256) void softmax_hess_sqrt_kernel(const float* __restrict__ f,
                                                                const int64_t* __restrict__ y, int B, int C,
                                                                float* __restrict__ S,
                                                                float* __restrict__ partials) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;
  float nll = 0.f;
  if (n < B) {
    const float* fr = f + (int64_t)n * C;
    float m = -INFINITY;
    for (int j = lane; j < C; j += 64) m = fmaxf(m, fr[j]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    float z = 0.f;
    for (int j = lane; j < C; j += 64) z += expf(fr[j] - m);
    z = wave_sum(z);
    const float logz = logf(z) + m;
    if (y != nullptr && lane == 0) {
      const int64_t lab = y[n];  // labels outside [0, C) (e.g. an ignore_index of -100) contribute nothing
      if (lab >= 0 && lab < C) nll = logz - fr[lab];
    }
    for (int c = 0; c < C; ++c) {
      const float pc = expf(fr[c] - logz);
      const float spc = sqrtf(pc);
      float* out = S + ((int64_t)c * B + n) * C;
      for (int j = lane; j < C; j += 64) {
        const float pj = expf(fr[j] - logz);
        out[j] = (j == c ? spc : 0.f) - pj * spc;
      }
    }
  }
  if (partials != nullptr && y != nullptr) {  // fixed-order reduction: per-block partial, summed by loss_finish_kernel
    const float tot = block_sum_256(nll, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
  }
}

// Rank-revealing root with C-1 columns (the softmax Hessian has rank C-1): the closed-form Cholesky factor
//   L[j][j] = sqrt(p_j s_{j+1} / s_j),  L[i][j] = -p_i sqrt(p_j / (s_j s_{j+1}))  (i > j),  s_j = sum_{k>=j} p_k,
// L L^T = diag(p) - p p^T exactly; one reverse pass fewer than the symmetric root.  S[c][n][i] = L_n[i][c].
__global__ __launch_bounds__(256) void softmax_hess_chol_kernel(const float* __restrict__ f,
                                                                const int64_t* __restrict__ y, int B, int C,
                                                                float* __restrict__ S,
                                                                float* __restrict__ partials) {
  extern __shared__ float dyn[];  // per wave: p[C], s[C+1]
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* p = dyn + wave * (2 * C + 1);
  float* sfx = p + C;
  const int n = blockIdx.x * 4 + wave;
  float nll = 0.f;
  if (n < B) {
    const float* fr = f + (int64_t)n * C;
    float m = -INFINITY;
    for (int j = lane; j < C; j += 64) m = fmaxf(m, fr[j]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    float z = 0.f;
    for (int j = lane; j < C; j += 64) z += expf(fr[j] - m);
    z = wave_sum(z);
    const float logz = logf(z) + m;
    if (y != nullptr && lane == 0) {
      const int64_t lab = y[n];  // labels outside [0, C) (e.g. an ignore_index of -100) contribute nothing
      if (lab >= 0 && lab < C) nll = logz - fr[lab];
    }
    for (int j = lane; j < C; j += 64) p[j] = expf(fr[j] - logz);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {  // suffix sums of positive numbers (C is small)
      float acc = 0.f;
      sfx[C] = 0.f;
      for (int j = C - 1; j >= 0; --j) {
        acc += p[j];
        sfx[j] = acc;
      }
    }
    __builtin_amdgcn_wave_barrier();
    for (int c = 0; c < C - 1; ++c) {
      const float pc = p[c], sc = sfx[c], sn = sfx[c + 1];
      const bool ok = (sn > 0.f) && (sc > 0.f);
      // factored so that nothing under/overflows for nearly one-hot rows: p_i <= s_{c+1} <= s_c <= 1
      const float r0 = ok ? sqrtf(pc / sc) : 0.f;   // <= 1
      const float dg = r0 * sqrtf(sn);
      const float rs = ok ? rsqrtf(sn) : 0.f;
      float* out = S + ((int64_t)c * B + n) * C;
      for (int i = lane; i < C; i += 64) out[i] = (i < c) ? 0.f : (i == c ? dg : -(p[i] * rs) * r0);
    }
  }
  if (partials != nullptr && y != nullptr) {  // fixed-order reduction: per-block partial, summed by loss_finish_kernel
    const float tot = block_sum_256(nll, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
  }
}

__global__ __launch_bounds__(256) void sq_err_sum_kernel(const float* __restrict__ f, const float* __restrict__ y,
                                                         int64_t numel,
                                                         float* __restrict__ partials) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
    const float d = f[i] - y[i];
    s += d * d;
  }
  const float tot = block_sum_256(s, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// loss_accum[0] += scale * sum_i partials[i], summed in index order by one wave (run-to-run deterministic; the
// per-block partials themselves are fixed-order tree sums)
__global__ __launch_bounds__(64) void loss_finish_kernel(const float* __restrict__ partials, int n, float scale,
                                                         float* __restrict__ loss_accum) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += partials[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) loss_accum[0] += scale * s;
}

}  // namespace lk

using namespace lk;

extern "C" size_t lk_loss_workspace_bytes(int64_t B) { return (size_t)((B + 3) / 4 + 1) * sizeof(float); }

static int finish_loss(const float* partials, int64_t n, float scale, float* loss_accum, void* stream) {
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partials, (int)n, scale, loss_accum);
  return check_launch("loss_finish_kernel");
}

extern "C" int lk_softmax_hess_sqrt_f32(const float* f, const int64_t* y, int64_t B, int64_t C, float* S,
                                        float* loss_accum, float* ws, void* stream) {
  LK_REQUIRE(f && S && B >= 0 && C >= 1 && B < (1ll << 31) && C < (1 << 24), "lk_softmax_hess_sqrt_f32: bad arguments");
  const bool want_loss = loss_accum != nullptr && y != nullptr;
  LK_REQUIRE(!want_loss || ws, "lk_softmax_hess_sqrt_f32: the loss needs a workspace of lk_loss_workspace_bytes(B)");
  if (B == 0) return LK_OK;
  hipLaunchKernelGGL(softmax_hess_sqrt_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, f, y,
                     (int)B, (int)C, S, want_loss ? ws : nullptr);
  int rc = check_launch("softmax_hess_sqrt_kernel");
  if (rc == LK_OK && want_loss) rc = finish_loss(ws, (B + 3) / 4, 1.f, loss_accum, stream);
  return rc;
}

extern "C" int lk_softmax_hess_chol_f32(const float* f, const int64_t* y, int64_t B, int64_t C, float* S,
                                        float* loss_accum, float* ws, void* stream) {
  // 4 waves x (2C + 1) floats of dynamic LDS must stay inside the 64 KiB a launch may request by default
  LK_REQUIRE(f && S && B >= 0 && C >= 2 && B < (1ll << 31) && C <= LK_SOFTMAX_CHOL_MAX_C,
             "lk_softmax_hess_chol_f32: bad arguments (2 <= C <= %d)", LK_SOFTMAX_CHOL_MAX_C);
  const bool want_loss = loss_accum != nullptr && y != nullptr;
  LK_REQUIRE(!want_loss || ws, "lk_softmax_hess_chol_f32: the loss needs a workspace of lk_loss_workspace_bytes(B)");
  if (B == 0) return LK_OK;
  const size_t lds = (size_t)4 * (2 * C + 1) * sizeof(float);
  hipLaunchKernelGGL(softmax_hess_chol_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), lds, (hipStream_t)stream, f, y,
                     (int)B, (int)C, S, want_loss ? ws : nullptr);
  int rc = check_launch("softmax_hess_chol_kernel");
  if (rc == LK_OK && want_loss) rc = finish_loss(ws, (B + 3) / 4, 1.f, loss_accum, stream);
  return rc;
}

extern "C" int lk_sq_err_sum_f32(const float* f, const float* y, int64_t numel, float scale, float* loss_accum,
                                 float* ws, void* stream) {
  LK_REQUIRE(f && y && loss_accum && ws && numel >= 0, "lk_sq_err_sum_f32: bad arguments");
  if (numel == 0) return LK_OK;
  int64_t blocks = (numel + 255) / 256;
  if (blocks > 1024) blocks = 1024;  // <= 1024 partials: ws of 4 KiB (lk_loss_workspace_bytes(4096))
  hipLaunchKernelGGL(sq_err_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, f, y, numel, ws);
  int rc = check_launch("sq_err_sum_kernel");
  if (rc == LK_OK) rc = finish_loss(ws, blocks, scale, loss_accum, stream);
  return rc;
}
