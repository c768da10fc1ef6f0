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
//   [64 x D~] x [D~ x D~] MFMA product per 64-sample tile with the row-dot against pt fused into the
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
// grid = (ceil(B/64), C(C+1)/2); 4 waves as 2x2 over a 64(n) x 64(q) tile.
__global__ __launch_bounds__(256) void dense_quadform_ll_kernel(const float* __restrict__ phi,
                                                                const float* __restrict__ Sigma, int64_t B, int C,
                                                                int D, int Dt, int64_t P, float* __restrict__ fvar) {
  __shared__ float sA[64][17];   // pt[n][p-chunk]
  __shared__ float sB[16][65];   // Sigma[(c,p)][(k,q)] chunk
  __shared__ float sE[64][65];   // pt[n][q-chunk] for the fused row-dot
  __shared__ float sR[2][64];    // cross-wave (wn) reduction
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, hi = lane >> 5, wm = wave >> 1, wn = wave & 1;
  int c = 0, rem = blockIdx.y, rowlen = C;
  while (rem >= rowlen) {
    rem -= rowlen;
    ++c;
    --rowlen;
  }
  const int k = c + rem;
  const int64_t n0 = (int64_t)blockIdx.x * 64;

  float psum[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) psum[r] = 0.f;

  for (int q0 = 0; q0 < Dt; q0 += 64) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // epilogue operand tile
    for (int e = tid; e < 64 * 64; e += 256) {
      const int nn = e >> 6, qq = e & 63;
      const int64_t n = n0 + nn;
      sE[nn][qq] = (n < B && q0 + qq < Dt) ? phi_aug(phi, n, q0 + qq, D) : 0.f;
    }
    for (int p0 = 0; p0 < Dt; p0 += 16) {
      for (int e = tid; e < 64 * 16; e += 256) {
        const int nn = e >> 4, pp = e & 15;
        const int64_t n = n0 + nn;
        sA[nn][pp] = (n < B && p0 + pp < Dt) ? phi_aug(phi, n, p0 + pp, D) : 0.f;
      }
      for (int e = tid; e < 16 * 64; e += 256) {
        const int pp = e >> 6, qq = e & 63;
        float v = 0.f;
        if (p0 + pp < Dt && q0 + qq < Dt)
          v = Sigma[(int64_t)ref_index(c, p0 + pp, C, D) * P + ref_index(k, q0 + qq, C, D)];
        sB[pp][qq] = v;
      }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int pp = 2 * kk + hi;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[wm * 32 + lo][pp], sB[pp][wn * 32 + lo], acc, 0, 0, 0);
      }
      __syncthreads();
    }
    // fused row-dot with pt[n][q]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      psum[r] += acc[r] * sE[row][wn * 32 + lo];
    }
    __syncthreads();  // sE is rewritten by the next q-chunk
  }
  // reduce over the 32 lanes sharing `hi`, then over the two column-waves
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = psum[r];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    psum[r] = v;
  }
  if (lo == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sR[wn][wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] = psum[r];
  }
  __syncthreads();
  if (tid < 64) {
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
  dim3 grid((unsigned)((B + 63) / 64), (unsigned)(C * (C + 1) / 2));
  hipLaunchKernelGGL(dense_quadform_ll_kernel, grid, dim3(256), 0, (hipStream_t)stream, phi, Sigma, B, (int)C, (int)D, Dt,
                     (int64_t)C * Dt, fvar);
  return check_launch("dense_quadform_ll_kernel");
}
