// KronDecomposed.logdet with analytic derivatives, and the structure-exploiting GLM predictive
// variances (no [B,C,P] Jacobian is materialised).
// Replaces laplace/utils/matrix.py:381-404 (logdet), :406-461 (_bmm / inv_square_form as used by
// KronLaplace.functional_variance, laplace/baselaplace.py:1834-1835) and DiagLaplace.functional_variance
// (baselaplace.py:2113-2115) for nn.Linear layers.
#include "lk_common.h"

namespace lk {

// one wave per row i of the eigenvalue outer product; rows' partial sums go to ws, reduced in fp64 below
__global__ __launch_bounds__(256) void logdet_rows_kernel(const float* __restrict__ l1, int n1,
                                                          const float* __restrict__ l2, int n2,
                                                          const float* __restrict__ delta, int damping,
                                                          float* __restrict__ row_log, float* __restrict__ row_dd,
                                                          float* __restrict__ d_l1) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n1) return;
  const float d = delta[0];
  const float sd = damping ? sqrtf(d) : 0.f;
  const float a = damping ? l1[i] + sd : l1[i];
  float slog = 0.f, sdd = 0.f, sdl = 0.f;
  if (n2 == 0) {
    if (lane == 0) {
      const float v = l1[i] + d;
      slog = logf(v);
      sdd = 1.f / v;
      sdl = 1.f / v;
    }
  } else {
    for (int j = lane; j < n2; j += 64) {
      const float b = damping ? l2[j] + sd : l2[j];
      const float v = damping ? a * b : a * b + d;
      slog += logf(v);
      const float inv = 1.f / v;
      sdd += inv;
      sdl += b * inv;
    }
  }
  slog = wave_sum(slog);
  sdd = wave_sum(sdd);
  sdl = wave_sum(sdl);
  if (lane == 0) {
    row_log[i] = slog;
    row_dd[i] = sdd;
    if (d_l1 != nullptr) d_l1[i] += sdl;
  }
}

__global__ __launch_bounds__(256) void logdet_cols_kernel(const float* __restrict__ l1, int n1,
                                                          const float* __restrict__ l2, int n2,
                                                          const float* __restrict__ delta,
                                                          float* __restrict__ d_l2) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= n2) return;
  const float d = delta[0], b = l2[j];
  float s = 0.f;
  for (int i = lane; i < n1; i += 64) s += l1[i] / (l1[i] * b + d);
  s = wave_sum(s);
  if (lane == 0) d_l2[j] += s;
}

__global__ __launch_bounds__(256) void logdet_final_kernel(const float* __restrict__ row_log,
                                                           const float* __restrict__ row_dd, int n1,
                                                           float* __restrict__ out, float* __restrict__ d_delta) {
  __shared__ double red[2][256];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n1; i += 256) {
    a += (double)row_log[i];
    b += (double)row_dd[i];
  }
  red[0][threadIdx.x] = a;
  red[1][threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] += (float)red[0][0];
    if (d_delta != nullptr) d_delta[0] += (float)red[1][0];
  }
}

// ---- predictive variance of one nn.Linear layer ------------------------------------------------------
// fvar[n][c][k] += sum_o u[c][n][o] u[k][n][o] * wgt[n][o],  wgt[n][o] = sum_i v[n][i]^2 * W(o,i)
//   MODE 0 (Kron):  W(o,i) = 1 / (l1[o]*l2[i] + delta)         u,v = eigenbasis projections
//   MODE 1 (diag):  W(o,i) = var_w[o][i]                         u,v = raw grads / activations
// plus (optional) bias term  sum_o ub[c][n][o] ub[k][n][o] * bw[o]   with bw[o] = 1/(lb[o]+delta_b) or var_b[o].
// One workgroup per sample; v^2 staged in LDS; one wave per output unit o for the weight reduction.
template <int MODE>
__global__ __launch_bounds__(256) void quadform_linear_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                              const float* __restrict__ w0, const float* __restrict__ w1,
                                                              const float* __restrict__ delta, int B, int Cc, int Do,
                                                              int Di, const float* __restrict__ ub,
                                                              const float* __restrict__ lb,
                                                              const float* __restrict__ delta_b,
                                                              float* __restrict__ fvar) {
  extern __shared__ float dyn[];  // [Di] v^2, then [Do] wgt, then [Do] bias weight
  float* v2 = dyn;
  float* wgt = dyn + Di;
  float* bw = wgt + Do;
  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < Di; i += 256) {
    const float t = v[(int64_t)n * Di + i];
    v2[i] = t * t;
  }
  const float d = (MODE == 0) ? delta[0] : 0.f;
  if (ub != nullptr)
    for (int o = tid; o < Do; o += 256) bw[o] = (MODE == 0) ? 1.f / (lb[o] + delta_b[0]) : lb[o];
  __syncthreads();
  for (int o = wave; o < Do; o += 4) {
    float s = 0.f;
    if (MODE == 0) {
      const float a = w0[o];
      for (int i = lane; i < Di; i += 64) s += v2[i] / (a * w1[i] + d);
    } else {
      const float* vr = w0 + (int64_t)o * Di;
      for (int i = lane; i < Di; i += 64) s += v2[i] * vr[i];
    }
    s = wave_sum(s);
    if (lane == 0) wgt[o] = s;
  }
  __syncthreads();
  for (int ck = tid; ck < Cc * Cc; ck += 256) {
    const int c = ck / Cc, k = ck - c * Cc;
    const float* uc = u + ((int64_t)c * B + n) * Do;
    const float* uk = u + ((int64_t)k * B + n) * Do;
    float s = 0.f;
    for (int o = 0; o < Do; ++o) s += uc[o] * uk[o] * wgt[o];
    if (ub != nullptr) {
      const float* bc = ub + ((int64_t)c * B + n) * Do;
      const float* bk = ub + ((int64_t)k * B + n) * Do;
      for (int o = 0; o < Do; ++o) s += bc[o] * bk[o] * bw[o];
    }
    fvar[((int64_t)n * Cc + c) * Cc + k] += s;
  }
}

// fvar[n][c][k] = sum_p Js[n][c][p] var[p] Js[n][k][p] ; one workgroup per (n, c, k>=c), mirrored write
__global__ __launch_bounds__(256) void diag_quadform_js_kernel(const float* __restrict__ Js,
                                                               const float* __restrict__ var, int C, int64_t P,
                                                               float* __restrict__ fvar) {
  __shared__ float red[4];
  const int n = blockIdx.z, c = blockIdx.y, k = blockIdx.x;
  if (k < c) return;
  const float* jc = Js + ((int64_t)n * C + c) * P;
  const float* jk = Js + ((int64_t)n * C + k) * P;
  float s = 0.f;
  for (int64_t p = threadIdx.x; p < P; p += 256) s += jc[p] * var[p] * jk[p];
  const float tot = block_sum_256(s, red);
  if (threadIdx.x == 0) {
    fvar[((int64_t)n * C + c) * C + k] = tot;
    fvar[((int64_t)n * C + k) * C + c] = tot;
  }
}

}  // namespace lk

using namespace lk;

extern "C" size_t lk_kron_logdet_workspace_bytes(int64_t n1) { return (size_t)(n1 > 0 ? n1 : 0) * 2 * sizeof(float); }

extern "C" int lk_kron_logdet_f32(const float* l1, int64_t n1, const float* l2, int64_t n2, const float* delta,
                                  int damping, float* out, float* d_l1, float* d_l2, float* d_delta, void* ws,
                                  size_t ws_bytes, void* stream_) {
  LK_REQUIRE(l1 && delta && out && n1 >= 1 && n2 >= 0 && (n2 == 0 || l2), "lk_kron_logdet_f32: bad arguments");
  LK_REQUIRE(!(damping && (d_l1 || d_l2 || d_delta)), "lk_kron_logdet_f32: no derivatives with damping");
  if (ws == nullptr || ws_bytes < lk_kron_logdet_workspace_bytes(n1)) {
    set_error("lk_kron_logdet_f32: workspace too small");
    return LK_EWORKSPACE;
  }
  hipStream_t stream = (hipStream_t)stream_;
  float* row_log = static_cast<float*>(ws);
  float* row_dd = row_log + n1;
  hipLaunchKernelGGL(logdet_rows_kernel, dim3((unsigned)((n1 + 3) / 4)), dim3(256), 0, stream, l1, (int)n1, l2, (int)n2,
                     delta, damping, row_log, row_dd, d_l1);
  if (d_l2 != nullptr && n2 > 0)
    hipLaunchKernelGGL(logdet_cols_kernel, dim3((unsigned)((n2 + 3) / 4)), dim3(256), 0, stream, l1, (int)n1, l2,
                       (int)n2, delta, d_l2);
  hipLaunchKernelGGL(logdet_final_kernel, dim3(1), dim3(256), 0, stream, row_log, row_dd, (int)n1, out, d_delta);
  return check_launch("lk_kron_logdet_f32");
}

template <int MODE>
static int launch_quadform_linear(const float* u, const float* v, const float* w0, const float* w1, const float* delta,
                                  int64_t B, int64_t Cc, int64_t Do, int64_t Di, const float* ub, const float* lb,
                                  const float* delta_b, float* fvar, hipStream_t stream) {
  if (B == 0) return LK_OK;
  const size_t lds = (size_t)(Di + 2 * Do) * sizeof(float);
  if (lds > 150 * 1024) {
    set_error("quadform_linear: layer too wide for the LDS-staged kernel (Di + 2*Do = %lld floats)",
              (long long)(Di + 2 * Do));
    return LK_EINVAL;
  }
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&quadform_linear_kernel<MODE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("quadform_linear: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
      return LK_ELAUNCH;
    }
  }
  hipLaunchKernelGGL((quadform_linear_kernel<MODE>), dim3((unsigned)B), dim3(256), lds, stream, u, v, w0, w1, delta,
                     (int)B, (int)Cc, (int)Do, (int)Di, ub, lb, delta_b, fvar);
  return check_launch("quadform_linear_kernel");
}

extern "C" int lk_kron_quadform_linear_f32(const float* u, const float* v, const float* l1, const float* l2,
                                           const float* delta, int64_t B, int64_t Cc, int64_t Do, int64_t Di,
                                           const float* ub, const float* lb, const float* delta_b, float* fvar,
                                           void* stream) {
  LK_REQUIRE(u && v && l1 && l2 && delta && fvar && B >= 0 && Cc >= 1 && Do >= 1 && Di >= 1,
             "lk_kron_quadform_linear_f32: bad arguments");
  LK_REQUIRE((ub == nullptr) || (lb && delta_b), "lk_kron_quadform_linear_f32: bias block needs lb and delta_b");
  return launch_quadform_linear<0>(u, v, l1, l2, delta, B, Cc, Do, Di, ub, lb, delta_b, fvar, (hipStream_t)stream);
}

extern "C" int lk_diag_quadform_linear_f32(const float* a, const float* g, const float* var_w, const float* var_b,
                                           int64_t B, int64_t Cc, int64_t Do, int64_t Di, float* fvar, void* stream) {
  LK_REQUIRE(a && g && var_w && fvar && B >= 0 && Cc >= 1 && Do >= 1 && Di >= 1,
             "lk_diag_quadform_linear_f32: bad arguments");
  return launch_quadform_linear<1>(g, a, var_w, nullptr, nullptr, B, Cc, Do, Di, var_b ? g : nullptr, var_b, nullptr,
                                   fvar, (hipStream_t)stream);
}

extern "C" int lk_diag_quadform_js_f32(const float* Js, const float* var, int64_t B, int64_t C, int64_t P, float* fvar,
                                       void* stream) {
  LK_REQUIRE(Js && var && fvar && B >= 0 && C >= 1 && P >= 1 && C <= 65535 && B <= 65535,
             "lk_diag_quadform_js_f32: bad arguments");
  if (B == 0) return LK_OK;
  hipLaunchKernelGGL(diag_quadform_js_kernel, dim3((unsigned)C, (unsigned)C, (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, Js, var, (int)C, P, fvar);
  return check_launch("diag_quadform_js_kernel");
}
