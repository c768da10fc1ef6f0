// Dense last-layer GGN and dense last-layer GLM predictive, exploiting J_n = I_C (x) [phi_n, 1].
//
// Replaces, for a Linear head, last_layer_jacobians + GGNInterface.full
// (laplace/curvature/curvature.py:131-167,375-411) and FullLaplace.functional_variance
// (laplace/baselaplace.py:1683-1684 through laplace/lllaplace.py:212-237) without ever forming the
// [B, C, P] Jacobian (90 % zeros).
//
// GGN.  With pt = [phi, 1] (D~ = D + has_bias) and softmax probabilities p:
//     H[(j,a),(k,b)] = sum_n (delta_jk p_nj - p_nj p_nk) pt_na pt_nb
//                    = blockdiag_j Gram(sqrt(p_j) . Pt)  -  Gram(Y),    Y[n][(j,a)] = p_nj pt_na
//   i.e. C small Grams + ONE Gram with K = B rows (not B*C): 2 B C^2 D~^2 flop instead of 2 B C^3 D~^2.
//   Regression (probs == NULL): H = I_C (x) Gram(Pt).  All Grams run on the exact-fp32 MFMA engine
//   of lk_gram.hip; the result is scattered from the "augmented" order (j, a) to the reference's
//   parameter order (weight [C][D] row-major, then bias [C]).
//
// Predictive.  fvar[n][c][k] = pt_n^T Sigma[(c,:),(k,:)] pt_n : for every class pair (c <= k) one
//   [128 x D~] x [D~ x D~] MFMA product per 128-sample tile with the row-dot against pt fused into the
//   epilogue; 2 C(C+1)/2 D~^2 flop per sample instead of 2 C P^2.
#include "lk_common.h"

namespace lk {

__device__ __forceinline__ float phi_aug(const float* __restrict__ phi, int64_t n, int a, int D) {
  return a < D ? phi[n * D + a] : 1.f;
}
// augmented index (j, a) -> reference parameter index
__device__ __forceinline__ int ref_index(int j, int a, int C, int D) { return a < D ? j * D + a : C * D + j; }

// mode 0: out[n][j*Dt + a] = p[n][j] * pt[n][a]   (Y)
// mode 1: out[n][a]        = sqrt(p[n][jsel]) * pt[n][a]
// mode 2: out[n][a]        = pt[n][a]
__global__ __launch_bounds__(256) void ll_build_rows_kernel(const float* __restrict__ phi,
                                                            const float* __restrict__ probs, int64_t B, int C, int D,
                                                            int Dt, int mode, int jsel, float* __restrict__ out) {
  const int64_t width = (mode == 0) ? (int64_t)C * Dt : Dt;
  const int64_t total = B * width;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t n = idx / width;
    const int col = (int)(idx - n * width);
    float v;
    if (mode == 0) {
      const int j = col / Dt, a = col - j * Dt;
      v = probs[n * C + j] * phi_aug(phi, n, a, D);
    } else if (mode == 1) {
      v = sqrtf(probs[n * C + jsel]) * phi_aug(phi, n, col, D);
    } else {
      v = phi_aug(phi, n, col, D);
    }
    out[idx] = v;
  }
}

// H[ref(j,a)][ref(k,b)] += src[(j,a)][(k,b)]  for the full augmented matrix (srcdim = C*Dt), or, with
// block >= 0, add the Dt x Dt matrix `src` into diagonal block j = block (block == -2: into every block).
__global__ __launch_bounds__(256) void ll_scatter_kernel(const float* __restrict__ src, int C, int D, int Dt,
                                                         int block, float* __restrict__ H, int64_t P) {
  if (block == -1) {
    const int64_t n = (int64_t)C * Dt, total = n * n;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
      const int r = (int)(idx / n), c = (int)(idx - (int64_t)r * n);
      const int j = r / Dt, a = r - j * Dt, k = c / Dt, b = c - k * Dt;
      H[(int64_t)ref_index(j, a, C, D) * P + ref_index(k, b, C, D)] += src[idx];
    }
  } else {
    const int64_t total = (int64_t)Dt * Dt;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
      const int a = (int)(idx / Dt), b = (int)(idx - (int64_t)a * Dt);
      const float v = src[idx];
      if (block >= 0) {
        H[(int64_t)ref_index(block, a, C, D) * P + ref_index(block, b, C, D)] += v;
      } else {
        for (int j = 0; j < C; ++j) H[(int64_t)ref_index(j, a, C, D) * P + ref_index(j, b, C, D)] += v;
      }
    }
  }
}

// ---- dense last-layer predictive -----------------------------------------------------------------
// grid = (ceil(B/128), C(C+1)/2); workgroup = 128 samples x one class pair (c <= k); 4 waves as 2x2, each owning a
// 64(n) x 64(q) block = 2x2 MFMA tiles of T = Pt Sigma_ck; q walks D~ in steps of 128 and the row-dot with Pt[n][q]
// is folded in after every q-step, so T is never written.  Operands through double-buffered LDS (p-major), the next
// chunk's global loads in flight during the MFMAs, LDS reads one k-step ahead.
constexpr int LLQ_BK = 16;
constexpr int LLQ_LD = 132;  // 128 + 4: the two half-waves of an operand read land on disjoint banks

__global__ __launch_bounds__(256) void dense_quadform_ll_kernel(const float* __restrict__ phi,
                                                                const float* __restrict__ Sigma, int64_t B, int C,
                                                                int D, int Dt, int64_t P, float* __restrict__ fvar) {
  __shared__ float sA[2][LLQ_BK][LLQ_LD];  // Pt[n][p] as [p][n]
  __shared__ float sB[2][LLQ_BK][LLQ_LD];  // Sigma[(c,p)][(k,q)] as [p][q]
  __shared__ float sR[2][128];             // cross-wave (wn) reduction
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, hi = lane >> 5, wm = wave >> 1, wn = wave & 1;
  int c = 0, rem = blockIdx.y, rowlen = C;
  while (rem >= rowlen) {
    rem -= rowlen;
    ++c;
    --rowlen;
  }
  const int k = c + rem;
  const int64_t n0 = (int64_t)blockIdx.x * 128;

  float psum[2][16];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) psum[tm][r] = 0.f;

  // staging coordinates of this thread: A element (n = e >> 4, p = e & 15), B element (p = e >> 7, q = e & 127)
  const int a_p = tid & 15, a_n = tid >> 4;  // + 16 rows of n per j
  const int b_q = tid & 127, b_p = tid >> 7;  // + 2 rows of p per j

  for (int q0 = 0; q0 < Dt; q0 += 128) {
    f32x16 acc[2][2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
    const int qg = q0 + b_q;
    const int64_t colB = qg < Dt ? ref_index(k, qg, C, D) : -1;

    float ra[8], rb[8];
    auto fetch = [&](int p0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t n = n0 + a_n + 16 * j;
        const int p = p0 + a_p;
        ra[j] = (n < B && p < Dt) ? phi_aug(phi, n, p, D) : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int p = p0 + b_p + 2 * j;
        rb[j] = (p < Dt && colB >= 0) ? Sigma[(int64_t)ref_index(c, p, C, D) * P + colB] : 0.f;
      }
    };
    fetch(0);
    int buf = 0;
    for (int p0 = 0; p0 < Dt; p0 += LLQ_BK) {
#pragma unroll
      for (int j = 0; j < 8; ++j) sA[buf][a_p][a_n + 16 * j] = ra[j];
#pragma unroll
      for (int j = 0; j < 8; ++j) sB[buf][b_p + 2 * j][b_q] = rb[j];
      __syncthreads();
      if (p0 + LLQ_BK < Dt) fetch(p0 + LLQ_BK);
      float a_cur[2], b_cur[2], a_nxt[2], b_nxt[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a_cur[t] = sA[buf][hi][wm * 64 + t * 32 + lo];
        b_cur[t] = sB[buf][hi][wn * 64 + t * 32 + lo];
      }
#pragma unroll
      for (int kk = 0; kk < LLQ_BK / 2; ++kk) {
        if (kk + 1 < LLQ_BK / 2) {
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            a_nxt[t] = sA[buf][2 * kk + 2 + hi][wm * 64 + t * 32 + lo];
            b_nxt[t] = sB[buf][2 * kk + 2 + hi][wn * 64 + t * 32 + lo];
          }
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[tm], b_cur[tn], acc[tm][tn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          a_cur[t] = a_nxt[t];
          b_cur[t] = b_nxt[t];
        }
      }
      buf ^= 1;  // the other buffer was last read two chunks ago: one barrier per chunk
    }
    __syncthreads();
    // fused row-dot with Pt[n][q]
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int q = q0 + wn * 64 + tn * 32 + lo;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t n = n0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const float e = (n < B && q < Dt) ? phi_aug(phi, n, q, D) : 0.f;
          psum[tm][r] += acc[tm][tn][r] * e;
        }
    }
  }
  // reduce over the 32 lanes sharing `hi`, then over the two column-waves
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = psum[tm][r];
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lo == 0) sR[wn][wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] = v;
    }
  __syncthreads();
  if (tid < 128) {
    const int64_t n = n0 + tid;
    if (n < B) {
      const float v = sR[0][tid] + sR[1][tid];
      fvar[(n * C + c) * C + k] = v;
      fvar[(n * C + k) * C + c] = v;
    }
  }
}

}  // namespace lk

using namespace lk;

static inline int grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

struct LLPlan {
  int Dt;
  int64_t P;
  size_t off_Y, off_Haug, off_X, off_T, off_gram, gram_bytes, total;
};

static LLPlan ll_plan(int64_t B, int64_t C, int64_t D, int has_bias) {
  LLPlan p;
  p.Dt = (int)(D + (has_bias ? 1 : 0));
  p.P = C * p.Dt;
  size_t off = 0;
  p.off_Y = off; off += align_up((size_t)B * p.P * 4, 256);
  p.off_Haug = off; off += align_up((size_t)p.P * p.P * 4, 256);
  p.off_X = off; off += align_up((size_t)B * p.Dt * 4, 256);
  p.off_T = off; off += align_up((size_t)p.Dt * p.Dt * 4, 256);
  size_t g1 = lk_gram_workspace_bytes(p.P, B), g2 = lk_gram_workspace_bytes(p.Dt, B);
  p.gram_bytes = g1 > g2 ? g1 : g2;
  p.off_gram = off; off += align_up(p.gram_bytes, 256);
  p.total = off;
  return p;
}

extern "C" size_t lk_ll_ggn_workspace_bytes(int64_t B, int64_t C, int64_t D) {
  if (B < 0 || C < 1 || D < 1) return 0;
  return ll_plan(B, C, D, 1).total;  // sized for the has_bias case
}

extern "C" int lk_ll_ggn_full_f32(const float* phi, const float* probs, int64_t B, int64_t C, int64_t D, int has_bias,
                                  float alpha, float* H, void* ws, size_t ws_bytes, void* stream_) {
  LK_REQUIRE(phi && H && B >= 0 && C >= 1 && D >= 1, "lk_ll_ggn_full_f32: bad arguments");
  LK_REQUIRE(C * (D + 1) < (1 << 20), "lk_ll_ggn_full_f32: P too large for a dense GGN");
  if (B == 0) return LK_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const LLPlan p = ll_plan(B, C, D, has_bias);
  if (ws == nullptr || ws_bytes < p.total) {
    set_error("lk_ll_ggn_full_f32: workspace too small (%zu < %zu bytes)", ws_bytes, p.total);
    return LK_EWORKSPACE;
  }
  char* base = static_cast<char*>(ws);
  float* Y = reinterpret_cast<float*>(base + p.off_Y);
  float* Haug = reinterpret_cast<float*>(base + p.off_Haug);
  float* X = reinterpret_cast<float*>(base + p.off_X);
  float* T = reinterpret_cast<float*>(base + p.off_T);
  void* gws = base + p.off_gram;
  const int Dt = p.Dt;
  int rc;
  if (probs == nullptr) {  // regression: I_C (x) Gram(Pt)
    hipLaunchKernelGGL(ll_build_rows_kernel, dim3(grid_for(B * Dt)), dim3(256), 0, stream, phi, probs, B, (int)C, (int)D,
                       Dt, 2, 0, X);
    if (hipMemsetAsync(T, 0, (size_t)Dt * Dt * 4, stream) != hipSuccess) return LK_ELAUNCH;
    rc = lk_gram_tn_f32(X, B, Dt, Dt, alpha, T, 0, gws, p.gram_bytes, stream_);
    if (rc) return rc;
    hipLaunchKernelGGL(ll_scatter_kernel, dim3(grid_for((int64_t)Dt * Dt)), dim3(256), 0, stream, T, (int)C, (int)D, Dt,
                       -2, H, (int64_t)C * Dt);
    return check_launch("lk_ll_ggn_full_f32");
  }
  // - Gram(Y)
  hipLaunchKernelGGL(ll_build_rows_kernel, dim3(grid_for(B * p.P)), dim3(256), 0, stream, phi, probs, B, (int)C, (int)D,
                     Dt, 0, 0, Y);
  if (hipMemsetAsync(Haug, 0, (size_t)p.P * p.P * 4, stream) != hipSuccess) return LK_ELAUNCH;
  rc = lk_gram_tn_f32(Y, B, p.P, p.P, -alpha, Haug, 0, gws, p.gram_bytes, stream_);
  if (rc) return rc;
  hipLaunchKernelGGL(ll_scatter_kernel, dim3(grid_for(p.P * p.P)), dim3(256), 0, stream, Haug, (int)C, (int)D, Dt, -1, H,
                     p.P);
  // + blockdiag_j Gram(sqrt(p_j) . Pt)
  for (int j = 0; j < C; ++j) {
    hipLaunchKernelGGL(ll_build_rows_kernel, dim3(grid_for(B * Dt)), dim3(256), 0, stream, phi, probs, B, (int)C, (int)D,
                       Dt, 1, j, X);
    if (hipMemsetAsync(T, 0, (size_t)Dt * Dt * 4, stream) != hipSuccess) return LK_ELAUNCH;
    rc = lk_gram_tn_f32(X, B, Dt, Dt, alpha, T, 0, gws, p.gram_bytes, stream_);
    if (rc) return rc;
    hipLaunchKernelGGL(ll_scatter_kernel, dim3(grid_for((int64_t)Dt * Dt)), dim3(256), 0, stream, T, (int)C, (int)D, Dt,
                       j, H, p.P);
  }
  return check_launch("lk_ll_ggn_full_f32");
}

// ---- last-layer Jacobians (J2): Js[n][c][:] = e_c (x) [phi_n, 1] in the reference's parameter order (weight [C][D] row-major, then
// bias [C]).  Pure store stream: one thread per 4 consecutive parameters of one (sample, output) row.
namespace lk {
__global__ __launch_bounds__(256) void jac_last_layer_kernel(const float* __restrict__ phi, int64_t rows, int C, int D, int has_bias,
                                                             int64_t P, float* __restrict__ Js) {
  const int64_t per_row = (P + 3) / 4;
  const int64_t total = rows * per_row;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / per_row;                 // (sample n, output c)
    const int64_t p0 = (i - row * per_row) * 4;
    const int64_t n = row / C;
    const int c = (int)(row - n * C);
    const int64_t lo = (int64_t)c * D, hi = lo + D;  // the weight row of output c
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t p = p0 + j;
      float x = 0.f;
      if (p >= lo && p < hi) x = phi[n * D + (p - lo)];
      else if (has_bias && p == (int64_t)C * D + c) x = 1.f;
      v[j] = x;
    }
    float* dst = Js + row * P + p0;
    if ((P & 3) == 0) {
      *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (p0 + j < P) dst[j] = v[j];
    }
  }
}
}  // namespace lk

extern "C" int lk_jac_last_layer_f32(const float* phi, int64_t B, int64_t C, int64_t D, int has_bias, float* Js, void* stream) {
  LK_REQUIRE(phi && Js && B >= 0 && C >= 1 && D >= 1, "lk_jac_last_layer_f32: bad arguments");
  if (B == 0) return LK_OK;
  const int64_t P = C * D + (has_bias ? C : 0);
  const int64_t work = B * C * ((P + 3) / 4);
  hipLaunchKernelGGL(lk::jac_last_layer_kernel, dim3(grid_for(work)), dim3(256), 0, (hipStream_t)stream, phi, B * C, (int)C, (int)D,
                     has_bias, P, Js);
  return check_launch("jac_last_layer_kernel");
}

extern "C" size_t lk_dense_quadform_ll_workspace_bytes(int64_t B, int64_t C, int64_t D) {
  (void)B; (void)C; (void)D;
  return 0;
}

extern "C" int lk_dense_quadform_ll_f32(const float* phi, const float* Sigma, int64_t B, int64_t C, int64_t D,
                                        int has_bias, float* fvar, void* ws, size_t ws_bytes, void* stream) {
  (void)ws; (void)ws_bytes;
  LK_REQUIRE(phi && Sigma && fvar && B >= 0 && C >= 1 && D >= 1, "lk_dense_quadform_ll_f32: bad arguments");
  LK_REQUIRE(C * (C + 1) / 2 <= 65535, "lk_dense_quadform_ll_f32: too many class pairs");
  if (B == 0) return LK_OK;
  const int Dt = (int)(D + (has_bias ? 1 : 0));
  dim3 grid((unsigned)((B + 127) / 128), (unsigned)(C * (C + 1) / 2));
  hipLaunchKernelGGL(dense_quadform_ll_kernel, grid, dim3(256), 0, (hipStream_t)stream, phi, Sigma, B, (int)C, (int)D, Dt,
                     (int64_t)C * Dt, fvar);
  return check_launch("dense_quadform_ll_kernel");
}
