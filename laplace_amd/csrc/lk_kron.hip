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

// ---- whole-posterior logdet: every block of a KronDecomposed in one pass -------------------------------------------
// sum_b sum_ij log(s * l1_i l2_j + delta_b), with d/d delta_b and d/d s, s = the scalar the posterior precision carries
// (H * H_factor, baselaplace.py:1820).  Up to LOGDET_BLOCKS blocks travel in the kernel arguments per launch.
constexpr int LOGDET_BLOCKS = 32;
struct LogdetBlocks {
  const float* l1[LOGDET_BLOCKS];
  const float* l2[LOGDET_BLOCKS];
  int n1[LOGDET_BLOCKS];
  int n2[LOGDET_BLOCKS];
  int row0[LOGDET_BLOCKS + 1];  // first row of block b in the workspace (rows of all launches are concatenated)
  int wg0[LOGDET_BLOCKS + 1];   // first workgroup of block b within this launch
  int count;
  int block0;                   // index of this launch's first block in delta / d_delta
};

// one wave per eigenvalue l1_i of its block: partial sums over j of log, 1/(.), l1 l2/(.)
__global__ __launch_bounds__(256) void logdet_blocks_rows_kernel(LogdetBlocks tb, const float* __restrict__ delta,
                                                                 const float* __restrict__ scale,
                                                                 float* __restrict__ rows) {
  int b = 0;
  while (b + 1 < tb.count && (int)blockIdx.x >= tb.wg0[b + 1]) ++b;
  const int lane = threadIdx.x & 63;
  const int i = ((int)blockIdx.x - tb.wg0[b]) * 4 + (threadIdx.x >> 6);
  const int n1 = tb.n1[b], n2 = tb.n2[b];
  if (i >= n1) return;
  const float d = delta[tb.block0 + b];
  const float s = scale != nullptr ? scale[0] : 1.f;
  const float a = tb.l1[b][i];
  float slog = 0.f, sdd = 0.f, sds = 0.f;
  if (n2 == 0) {
    if (lane == 0) {
      const float inv = 1.f / (s * a + d);
      slog = logf(s * a + d);
      sdd = inv;
      sds = a * inv;
    }
  } else {
    const float* __restrict__ l2 = tb.l2[b];
    for (int j = lane; j < n2; j += 64) {
      const float lam = a * l2[j];
      const float v = s * lam + d;
      const float inv = 1.f / v;
      slog += logf(v);
      sdd += inv;
      sds += lam * inv;
    }
  }
  slog = wave_sum(slog);
  sdd = wave_sum(sdd);
  sds = wave_sum(sds);
  if (lane == 0) {
    float* r = rows + 3 * (size_t)(tb.row0[b] + i);
    r[0] = slog;
    r[1] = sdd;
    r[2] = sds;
  }
}

// one workgroup per block: fixed-order fp64 reduction of its rows -> per-block (log, d_delta, d_scale)
__global__ __launch_bounds__(256) void logdet_blocks_reduce_kernel(LogdetBlocks tb, const float* __restrict__ rows,
                                                                   double* __restrict__ per_block,
                                                                   float* __restrict__ d_delta) {
  __shared__ double red[3][256];
  const int b = blockIdx.x;
  double acc[3] = {0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < tb.n1[b]; i += 256) {
    const float* r = rows + 3 * (size_t)(tb.row0[b] + i);
    acc[0] += (double)r[0];
    acc[1] += (double)r[1];
    acc[2] += (double)r[2];
  }
  for (int q = 0; q < 3; ++q) red[q][threadIdx.x] = acc[q];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s)
      for (int q = 0; q < 3; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    per_block[2 * (tb.block0 + b)] = red[0][0];
    per_block[2 * (tb.block0 + b) + 1] = red[2][0];
    if (d_delta != nullptr) d_delta[tb.block0 + b] += (float)red[1][0];
  }
}

__global__ __launch_bounds__(256) void logdet_blocks_total_kernel(const double* __restrict__ per_block, int nblocks,
                                                                  float* __restrict__ out, float* __restrict__ d_scale) {
  __shared__ double red[2][256];
  double a = 0.0, c = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) {
    a += per_block[2 * b];
    c += per_block[2 * b + 1];
  }
  red[0][threadIdx.x] = a;
  red[1][threadIdx.x] = c;
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
    if (d_scale != nullptr) d_scale[0] += (float)red[1][0];
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

extern "C" size_t lk_kron_logdet_blocks_workspace_bytes(int64_t total_rows, int64_t nblocks) {
  if (total_rows < 0 || nblocks < 0) return 0;
  return align_up((size_t)total_rows * 3 * sizeof(float), 256) + (size_t)nblocks * 2 * sizeof(double);
}

extern "C" int lk_kron_logdet_blocks_f32(int64_t nblocks, const float* const* l1, const int64_t* n1,
                                         const float* const* l2, const int64_t* n2, const float* delta,
                                         const float* scale, float* out, float* d_delta, float* d_scale, void* ws,
                                         size_t ws_bytes, void* stream_) {
  LK_REQUIRE(nblocks >= 0 && out && (nblocks == 0 || (l1 && n1 && l2 && n2 && delta)),
             "lk_kron_logdet_blocks_f32: bad arguments");
  if (nblocks == 0) return LK_OK;
  int64_t total_rows = 0;
  for (int64_t b = 0; b < nblocks; ++b) {
    LK_REQUIRE(l1[b] && n1[b] >= 1 && n2[b] >= 0 && (n2[b] == 0 || l2[b]) && n1[b] < (1ll << 30) && n2[b] < (1ll << 30),
               "lk_kron_logdet_blocks_f32: bad block");
    total_rows += n1[b];
  }
  LK_REQUIRE(total_rows < (1ll << 30), "lk_kron_logdet_blocks_f32: too many eigenvalues");
  if (ws == nullptr || ws_bytes < lk_kron_logdet_blocks_workspace_bytes(total_rows, nblocks)) {
    set_error("lk_kron_logdet_blocks_f32: workspace too small");
    return LK_EWORKSPACE;
  }
  hipStream_t stream = (hipStream_t)stream_;
  float* rows = static_cast<float*>(ws);
  double* per_block =
      reinterpret_cast<double*>(static_cast<char*>(ws) + align_up((size_t)total_rows * 3 * sizeof(float), 256));
  int row = 0;
  for (int64_t b0 = 0; b0 < nblocks; b0 += LOGDET_BLOCKS) {
    LogdetBlocks tb;
    tb.count = (int)(nblocks - b0 < LOGDET_BLOCKS ? nblocks - b0 : LOGDET_BLOCKS);
    tb.block0 = (int)b0;
    int wg = 0;
    for (int b = 0; b < tb.count; ++b) {
      tb.l1[b] = l1[b0 + b];
      tb.l2[b] = l2[b0 + b];
      tb.n1[b] = (int)n1[b0 + b];
      tb.n2[b] = (int)n2[b0 + b];
      tb.row0[b] = row;
      tb.wg0[b] = wg;
      row += tb.n1[b];
      wg += (tb.n1[b] + 3) / 4;
    }
    tb.row0[tb.count] = row;
    tb.wg0[tb.count] = wg;
    hipLaunchKernelGGL(logdet_blocks_rows_kernel, dim3((unsigned)wg), dim3(256), 0, stream, tb, delta, scale, rows);
    hipLaunchKernelGGL(logdet_blocks_reduce_kernel, dim3((unsigned)tb.count), dim3(256), 0, stream, tb, rows, per_block,
                       d_delta);
  }
  hipLaunchKernelGGL(logdet_blocks_total_kernel, dim3(1), dim3(256), 0, stream, per_block, (int)nblocks, out, d_scale);
  return check_launch("lk_kron_logdet_blocks_f32");
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
