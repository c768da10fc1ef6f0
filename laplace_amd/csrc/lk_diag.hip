// Diagonal GGN accumulation and per-sample Jacobian assembly for nn.Linear / nn.Conv2d layers.
// Replaces GGNInterface.diag / EFInterface.diag (laplace/curvature/curvature.py:413-433,494-505) and the
// jacrev materialisation of CurvatureInterface.jacobians (curvature.py:88-129) for supported layers.
#include "lk_common.h"

namespace lk {

// h_w[o][i] += alpha * sum_n gsq[n][o] * a[n][i]^2,  gsq[n][o] = sum_c g[c][n][o]^2 ; h_b[o] += alpha*sum_n gsq
// block = 16(o) x 16(i) outputs, batch streamed through LDS in chunks of 64 samples.  HBM-bound:
// algorithmic bytes = 4*(B*Di + Cc*B*Do + 2*Do*Di).
__global__ __launch_bounds__(256) void diag_ggn_linear_kernel(const float* __restrict__ a,
                                                              const float* __restrict__ g, int B, int Cc, int Di,
                                                              int Do, float alpha, float* __restrict__ h_w,
                                                              float* __restrict__ h_b) {
  __shared__ float sg[64][17];
  __shared__ float sa[64][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i0 = blockIdx.x * 16, o0 = blockIdx.y * 16;
  float acc = 0.f, accb = 0.f;
  for (int n0 = 0; n0 < B; n0 += 64) {
    for (int e = threadIdx.x; e < 64 * 16; e += 256) {
      const int nn = e >> 4, j = e & 15;
      const int n = n0 + nn;
      float gs = 0.f, av = 0.f;
      if (n < B) {
        if (o0 + j < Do)
          for (int c = 0; c < Cc; ++c) {
            const float v = g[((int64_t)c * B + n) * Do + o0 + j];
            gs += v * v;
          }
        if (i0 + j < Di) {
          av = a[(int64_t)n * Di + i0 + j];
          av *= av;
        }
      }
      sg[nn][j] = gs;
      sa[nn][j] = av;
    }
    __syncthreads();
#pragma unroll 8
    for (int nn = 0; nn < 64; ++nn) {
      acc += sg[nn][ty] * sa[nn][tx];
      accb += sg[nn][ty];
    }
    __syncthreads();
  }
  const int o = o0 + ty, i = i0 + tx;
  if (o < Do && i < Di) h_w[(int64_t)o * Di + i] += alpha * acc;
  if (h_b != nullptr && blockIdx.x == 0 && tx == 0 && o < Do) h_b[o] += alpha * accb;
}

// Js[n][c][col0 + o*Di + i] = g[c][n][o]*a[n][i];  Js[n][c][bcol0 + o] = g[c][n][o] (bcol0 < 0: no bias)
__global__ __launch_bounds__(256) void jac_linear_kernel(const float* __restrict__ a, const float* __restrict__ g,
                                                         int B, int Cc, int Di, int Do, float* __restrict__ Js,
                                                         int64_t P, int64_t col0, int64_t bcol0) {
  const int n = blockIdx.y / Cc, c = blockIdx.y % Cc;
  const float* gr = g + ((int64_t)c * B + n) * Do;
  const float* ar = a + (int64_t)n * Di;
  float* out = Js + ((int64_t)n * Cc + c) * P;
  const int64_t total = (int64_t)Do * Di;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int o = (int)(idx / Di), i = (int)(idx - (int64_t)o * Di);
    out[col0 + idx] = gr[o] * ar[i];
  }
  if (bcol0 >= 0 && blockIdx.x == 0)
    for (int o = threadIdx.x; o < Do; o += 256) out[bcol0 + o] = gr[o];
}

// Per-sample conv weight Jacobian:  Js[n][c][col0 + o*Dk + k] = sum_l g[c][n][o][l] * patch[n][l][k],
// k = (ci, dy, dx) in F.unfold order; patches gathered on the fly from the NCHW input.
// grid = (ceil(Dk/16), ceil(Do/16), B*Cc), 16x16 threads, L streamed through LDS in chunks of 16.
struct ConvGeom {
  int Cin, H, W, OH, OW, kh, kw, sh, sw, ph, pw, dh, dw;
};
__global__ __launch_bounds__(256) void jac_conv_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                       int B, int Cc, int Do, ConvGeom cg, float* __restrict__ Js,
                                                       int64_t P, int64_t col0, int64_t bcol0) {
  __shared__ float sg[16][17];  // [o][l]
  __shared__ float sp[16][17];  // [l][k]
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int n = blockIdx.z / Cc, c = blockIdx.z % Cc;
  const int Dk = cg.Cin * cg.kh * cg.kw, L = cg.OH * cg.OW;
  const int k0 = blockIdx.x * 16, o0 = blockIdx.y * 16;
  const float* gn = g + ((int64_t)c * B + n) * Do * L;
  const float* xn = x + (int64_t)n * cg.Cin * cg.H * cg.W;
  // this thread's patch column (for the LDS fill it owns column tx)
  const int kcol = k0 + tx;
  const int ci = kcol / (cg.kh * cg.kw);
  const int dd = kcol - ci * cg.kh * cg.kw;
  const int dy = dd / cg.kw, dx = dd - dy * cg.kw;
  float acc = 0.f, bsum = 0.f;
  for (int l0 = 0; l0 < L; l0 += 16) {
    {  // sg[o = ty][l = tx]
      const int o = o0 + ty, l = l0 + tx;
      sg[ty][tx] = (o < Do && l < L) ? gn[(int64_t)o * L + l] : 0.f;
    }
    {  // sp[l = ty][k = tx]
      const int l = l0 + ty;
      float v = 0.f;
      if (l < L && kcol < Dk) {
        const int oh = l / cg.OW, ow = l - oh * cg.OW;
        const int ih = oh * cg.sh - cg.ph + dy * cg.dh, iw = ow * cg.sw - cg.pw + dx * cg.dw;
        if (ih >= 0 && ih < cg.H && iw >= 0 && iw < cg.W) v = xn[((int64_t)ci * cg.H + ih) * cg.W + iw];
      }
      sp[ty][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int l = 0; l < 16; ++l) {
      acc += sg[ty][l] * sp[l][tx];
      bsum += sg[ty][l];
    }
    __syncthreads();
  }
  const int o = o0 + ty;
  float* out = Js + ((int64_t)n * Cc + c) * P;
  if (o < Do && kcol < Dk) out[col0 + (int64_t)o * Dk + kcol] = acc;
  if (bcol0 >= 0 && blockIdx.x == 0 && tx == 0 && o < Do) out[bcol0 + o] = bsum;
}

// h[p] += alpha * sum_r Js[r][p]^2   (rows r = (n, c)); used for conv-layer diag GGN / EF
__global__ __launch_bounds__(256) void sq_colsum_kernel(const float* __restrict__ Js, int64_t rows, int64_t P,
                                                        int64_t col0, int64_t width, float alpha,
                                                        float* __restrict__ h) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= width) return;
  float s = 0.f;
  for (int64_t r = 0; r < rows; ++r) {
    const float v = Js[r * P + col0 + p];
    s += v * v;
  }
  h[p] += alpha * s;
}

}  // namespace lk

using namespace lk;

extern "C" int lk_diag_ggn_linear_f32(const float* a, const float* g, int64_t B, int64_t Cc, int64_t Di, int64_t Do,
                                      float alpha, float* h_w, float* h_b, void* stream) {
  LK_REQUIRE(a && g && h_w && B >= 0 && Cc >= 1 && Di >= 1 && Do >= 1, "lk_diag_ggn_linear_f32: bad arguments");
  LK_REQUIRE(Do <= 65535 * 16, "lk_diag_ggn_linear_f32: Do too large");
  dim3 grid((unsigned)((Di + 15) / 16), (unsigned)((Do + 15) / 16));
  hipLaunchKernelGGL(diag_ggn_linear_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, g, (int)B, (int)Cc, (int)Di,
                     (int)Do, alpha, h_w, h_b);
  return check_launch("diag_ggn_linear_kernel");
}

extern "C" int lk_jac_linear_f32(const float* a, const float* g, int64_t B, int64_t Cc, int64_t Di, int64_t Do,
                                 float* Js, int64_t P, int64_t col0, int64_t bcol0, void* stream) {
  LK_REQUIRE(a && g && Js && B >= 0 && Cc >= 1 && Di >= 1 && Do >= 1 && col0 >= 0 && col0 + Do * Di <= P,
             "lk_jac_linear_f32: bad arguments");
  LK_REQUIRE(B * Cc <= 65535, "lk_jac_linear_f32: B*C too large for grid.y (chunk the batch)");
  if (B == 0) return LK_OK;
  int64_t bx = (Do * Di + 255) / 256;
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(jac_linear_kernel, dim3((unsigned)bx, (unsigned)(B * Cc)), dim3(256), 0, (hipStream_t)stream, a,
                     g, (int)B, (int)Cc, (int)Di, (int)Do, Js, P, col0, bcol0);
  return check_launch("jac_linear_kernel");
}

extern "C" int lk_jac_conv_f32(const float* x_nchw, const float* g, int64_t B, int64_t Cc, int64_t Cin, int64_t H,
                               int64_t W, int64_t Do, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                               float* Js, int64_t P, int64_t col0, int64_t bcol0, void* stream) {
  LK_REQUIRE(x_nchw && g && Js && B >= 0 && Cc >= 1 && Cin >= 1 && Do >= 1, "lk_jac_conv_f32: bad arguments");
  LK_REQUIRE(B * Cc <= 65535, "lk_jac_conv_f32: B*C too large for grid.z (chunk the batch)");
  if (B == 0) return LK_OK;
  ConvGeom cg;
  cg.Cin = (int)Cin; cg.H = (int)H; cg.W = (int)W; cg.kh = kh; cg.kw = kw; cg.sh = sh; cg.sw = sw;
  cg.ph = ph; cg.pw = pw; cg.dh = dh; cg.dw = dw;
  cg.OH = (int)((H + 2 * ph - dh * (kh - 1) - 1) / sh + 1);
  cg.OW = (int)((W + 2 * pw - dw * (kw - 1) - 1) / sw + 1);
  LK_REQUIRE(cg.OH > 0 && cg.OW > 0, "lk_jac_conv_f32: empty output");
  const int64_t Dk = Cin * kh * kw;
  LK_REQUIRE(col0 >= 0 && col0 + Do * Dk <= P, "lk_jac_conv_f32: column range outside Js");
  dim3 grid((unsigned)((Dk + 15) / 16), (unsigned)((Do + 15) / 16), (unsigned)(B * Cc));
  hipLaunchKernelGGL(jac_conv_kernel, grid, dim3(256), 0, (hipStream_t)stream, x_nchw, g, (int)B, (int)Cc, (int)Do, cg,
                     Js, P, col0, bcol0);
  return check_launch("jac_conv_kernel");
}

extern "C" int lk_sq_colsum_f32(const float* Js, int64_t rows, int64_t P, int64_t col0, int64_t width, float alpha,
                                float* h, void* stream) {
  LK_REQUIRE(Js && h && rows >= 0 && width >= 0 && col0 >= 0 && col0 + width <= P, "lk_sq_colsum_f32: bad arguments");
  if (width == 0) return LK_OK;
  hipLaunchKernelGGL(sq_colsum_kernel, dim3((unsigned)((width + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Js,
                     rows, P, col0, width, alpha, h);
  return check_launch("sq_colsum_kernel");
}
