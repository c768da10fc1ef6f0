// Batched fp32 GEMM on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) with an element-wise epilogue weight:
//     C[b] = (op(A[b]) . op(B[b])) (.) E          op = identity or transpose, E optional (row stride 0 = one row for all)
// It carries the eigenbasis algebra of KronDecomposed (laplace/utils/matrix.py:406-461, `_bmm` at exponents -1 and -1/2:
// the GLM predictive through `inv_square_form` and the posterior samples of baselaplace.py:1845-1858): per Kronecker block
//     out_r = Q1 ((Q1^T W_r Q2) (.) (l1 (x) l2 + delta)^e) Q2^T
// is four of these products, the eigenvalue weighting fused into the second one; operands are addressed in place (batch
// strides, leading dimensions), nothing is copied or transposed in memory.
// 64 x 64 output tile per workgroup, 4 waves of one 32 x 32 MFMA tile, 16-deep K chunks through LDS (the tile of a
// transposed operand is transposed while it is staged); sizes need not be multiples of anything.
#include "lk_common.h"

namespace lk {

struct GemmArgs {
  const float* A;
  const float* B;
  const float* E;
  float* C;
  int M, N, K;
  int64_t lda, ldb, ldc, lde;   // leading dimensions of the STORED matrices (row-major); lde = 0 broadcasts one row of E
  int64_t sa, sb, sc;           // batch strides (0 = shared)
  int ta, tb;                   // op(A) = A^T (stored K x M), op(B) = B^T (stored N x K)
  float alpha;
  int accumulate;
};

__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs g) {
  __shared__ float sA[16][64 + 4];  // [k][i]
  __shared__ float sB[16][64 + 4];  // [k][j]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const float* A = g.A + (int64_t)blockIdx.z * g.sa;
  const float* B = g.B + (int64_t)blockIdx.z * g.sb;
  float* C = g.C + (int64_t)blockIdx.z * g.sc;
  const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < g.K; k0 += 16) {
    // stage 16 x 64 of op(A) and op(B): 1024 elements each, 4 per thread; the fast index of the load follows memory
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = e * 256 + tid;
      {
        int kk, ii;
        if (g.ta) { ii = idx & 63; kk = idx >> 6; } else { kk = idx & 15; ii = idx >> 4; }
        const int k = k0 + kk, i = i0 + ii;
        float v = 0.f;
        if (k < g.K && i < g.M) v = g.ta ? A[(int64_t)k * g.lda + i] : A[(int64_t)i * g.lda + k];
        sA[kk][ii] = v;
      }
      {
        int kk, jj;
        if (g.tb) { kk = idx & 15; jj = idx >> 4; } else { jj = idx & 63; kk = idx >> 6; }
        const int k = k0 + kk, j = j0 + jj;
        float v = 0.f;
        if (k < g.K && j < g.N) v = g.tb ? B[(int64_t)j * g.ldb + k] : B[(int64_t)k * g.ldb + j];
        sB[kk][jj] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k2 = 0; k2 < 16; k2 += 2) {
      const float a = sA[k2 + (lane >> 5)][wi + (lane & 31)];
      const float b = sB[k2 + (lane >> 5)][wj + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int j = j0 + wj + (lane & 31);
  if (j >= g.N) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + wi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (i >= g.M) continue;
    float v = g.alpha * acc[r];
    if (g.E) v *= g.E[(int64_t)i * g.lde + j];
    float* c = C + (int64_t)i * g.ldc + j;
    *c = g.accumulate ? *c + v : v;
  }
}

// lam[i][j] = (l1[i] * l2[j] + delta)^e   (damping: ((l1[i] + sqrt(delta)) (l2[j] + sqrt(delta)))^e);  l2 == NULL: (l1[i] + delta)^e
__global__ __launch_bounds__(256) void kron_pow_kernel(const float* __restrict__ l1, int n1, const float* __restrict__ l2,
                                                       int n2, const float* __restrict__ delta, float exponent,
                                                       int damping, float* __restrict__ lam) {
  const int64_t total = (int64_t)n1 * (l2 ? n2 : 1);
  const float d = delta[0];
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    float base;
    if (!l2) {
      base = l1[e] + d;
    } else {
      const int i = (int)(e / n2), j = (int)(e % n2);
      base = damping ? (l1[i] + sqrtf(d)) * (l2[j] + sqrtf(d)) : l1[i] * l2[j] + d;
    }
    lam[e] = exponent == -1.f ? 1.f / base : (exponent == -0.5f ? rsqrtf(base) : powf(base, exponent));
  }
}

}  // namespace lk

using namespace lk;

extern "C" int lk_gemm_f32(const float* A, const float* B, const float* E, float* C, int64_t batch, int64_t M, int64_t N,
                           int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t lde, int64_t stride_a,
                           int64_t stride_b, int64_t stride_c, int trans_a, int trans_b, float alpha, int accumulate,
                           void* stream) {
  LK_REQUIRE(A && B && C && batch >= 0 && M >= 0 && N >= 0 && K >= 0, "lk_gemm_f32: bad arguments");
  LK_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31) && batch < 65536 * 32768ll, "lk_gemm_f32: extents too large");
  if (batch == 0 || M == 0 || N == 0) return LK_OK;
  GemmArgs g{A, B, E, C, (int)M, (int)N, (int)K, lda, ldb, ldc, lde, stride_a, stride_b, stride_c, trans_a, trans_b, alpha,
             accumulate};
  const int64_t gy = (M + 63) / 64;
  LK_REQUIRE(gy <= 65535, "lk_gemm_f32: too many row tiles (M <= 4.19 M rows per batch entry)");
  for (int64_t b0 = 0; b0 < batch; b0 += 65535) {  // grid.z is a 16-bit extent
    const int64_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
    GemmArgs gb = g;
    gb.A += b0 * stride_a, gb.B += b0 * stride_b, gb.C += b0 * stride_c;
    hipLaunchKernelGGL(gemm_f32_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)gy, (unsigned)nb), dim3(256), 0,
                       (hipStream_t)stream, gb);
  }
  return check_launch("gemm_f32_kernel");
}

extern "C" int lk_kron_pow_f32(const float* l1, int64_t n1, const float* l2, int64_t n2, const float* delta, float exponent,
                               int damping, float* lam, void* stream) {
  LK_REQUIRE(l1 && delta && lam && n1 >= 0 && (!l2 || n2 >= 0), "lk_kron_pow_f32: bad arguments");
  const int64_t total = n1 * (l2 ? n2 : 1);
  if (total == 0) return LK_OK;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(kron_pow_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, l1, (int)n1, l2, (int)n2,
                     delta, exponent, damping, lam);
  return check_launch("kron_pow_kernel");
}
