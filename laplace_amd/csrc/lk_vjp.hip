// Element-wise vector-Jacobian products of the seed-batched reverse sweep (laplace_amd/sweep.py).
//
// The reference obtains output gradients from stock autograd, one activation / BatchNorm backward kernel per
// seed and layer (laplace/curvature/curvlinops.py:87-106 via curvlinops' hooks).  Here every seed travels in one
// cotangent of batch S*B; the VJP of `activation(BatchNorm_eval(.))` is a per-sample multiplier M[b,c,i]
// (ReLU mask or f'(y)) times a per-channel scale, applied to all S seeds in ONE pass:
//     out[s][b][c][i] = (g[s][b][c][i] + g2[s][b][c][i]) * M[b][c][i] * scale[c]
// (g2: the second branch of a residual connection, summed on the fly instead of in a separate pass).
// HBM-bound: algorithmic bytes = 4*S*per per input (g, g2) + 4*S*per (write out) + per*(1|4) (M, read once per
// thread and reused for all S seeds from registers).
#include "lk_common.h"

namespace lk {

template <bool MFLOAT>
__device__ __forceinline__ float4 load_mult4(const void* m, int64_t e) {
  if (MFLOAT) return *reinterpret_cast<const float4*>(static_cast<const float*>(m) + e);
  const uchar4 u = *reinterpret_cast<const uchar4*>(static_cast<const unsigned char*>(m) + e);
  return make_float4(u.x ? 1.f : 0.f, u.y ? 1.f : 0.f, u.z ? 1.f : 0.f, u.w ? 1.f : 0.f);
}

template <bool MFLOAT>
__global__ __launch_bounds__(256) void vjp_scale_mask_vec_kernel(const float* __restrict__ g, const float* __restrict__ g2,
    const void* __restrict__ m,
                                                                 const float* __restrict__ scale, int S, int64_t per,
                                                                 int C, int HW, float* __restrict__ out) {
  const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= per) return;
  float4 mult = m ? load_mult4<MFLOAT>(m, e) : make_float4(1.f, 1.f, 1.f, 1.f);
  if (scale) {
    const float sc = scale[(e / HW) % C];  // HW % 4 == 0: the four lanes share a channel
    mult.x *= sc, mult.y *= sc, mult.z *= sc, mult.w *= sc;
  }
  const float4* gp = reinterpret_cast<const float4*>(g + e);
  float4* op = reinterpret_cast<float4*>(out + e);
  const int64_t step = per / 4;
  if (g2) {
    const float4* hp = reinterpret_cast<const float4*>(g2 + e);
#pragma unroll 3
    for (int s = 0; s < S; ++s) {
      float4 v = gp[(int64_t)s * step];
      const float4 w = hp[(int64_t)s * step];
      v.x = (v.x + w.x) * mult.x, v.y = (v.y + w.y) * mult.y, v.z = (v.z + w.z) * mult.z, v.w = (v.w + w.w) * mult.w;
      op[(int64_t)s * step] = v;
    }
    return;
  }
#pragma unroll 3
  for (int s = 0; s < S; ++s) {
    float4 v = gp[(int64_t)s * step];
    v.x *= mult.x, v.y *= mult.y, v.z *= mult.z, v.w *= mult.w;
    op[(int64_t)s * step] = v;
  }
}

template <bool MFLOAT>
__global__ __launch_bounds__(256) void vjp_scale_mask_kernel(const float* __restrict__ g, const float* __restrict__ g2,
    const void* __restrict__ m,
                                                             const float* __restrict__ scale, int S, int64_t per,
                                                             int C, int HW, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= per) return;
  float mult = 1.f;
  if (m) mult = MFLOAT ? static_cast<const float*>(m)[e] : (static_cast<const unsigned char*>(m)[e] ? 1.f : 0.f);
  if (scale) mult *= scale[(e / HW) % C];
  for (int s = 0; s < S; ++s) {
    float v = g[(int64_t)s * per + e];
    if (g2) v += g2[(int64_t)s * per + e];
    out[(int64_t)s * per + e] = v * mult;
  }
}

// ---- forward of the sweep: eval-mode BatchNorm as a per-channel affine map, optional ReLU and its mask, one pass --------
//   y[b][c][i] = act(x[b][c][i] * scale[c] + shift[c]),   mask = y > 0   (scale = gamma / sqrt(var + eps),
//   shift = beta - mean * scale).  Replaces three stock launches per layer (MIOpen's BatchNorm inference kernel runs at
//   a third of the HBM rate on these shapes, then clamp_min, then the compare that produces the mask).
template <bool RELU>
__global__ __launch_bounds__(256) void bn_act_fwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             const float* __restrict__ addend, int64_t total, int C, int HW,
                                                             float* __restrict__ y, unsigned char* __restrict__ mask) {
  const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= total) return;
  const int c = (int)((e / HW) % C);  // HW % 4 == 0: the four lanes share a channel
  const float sc = scale[c], sh = shift[c];
  float4 v = *reinterpret_cast<const float4*>(x + e);
  v.x = v.x * sc + sh, v.y = v.y * sc + sh, v.z = v.z * sc + sh, v.w = v.w * sc + sh;
  if (addend) {  // the other branch of a residual connection
    const float4 a = *reinterpret_cast<const float4*>(addend + e);
    v.x += a.x, v.y += a.y, v.z += a.z, v.w += a.w;
  }
  if (RELU) {
    v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
    if (mask) *reinterpret_cast<uchar4*>(mask + e) = make_uchar4(v.x > 0.f, v.y > 0.f, v.z > 0.f, v.w > 0.f);
  }
  *reinterpret_cast<float4*>(y + e) = v;
}

template <bool RELU>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const float* __restrict__ addend,
                                                         int64_t total, int C, int HW, float* __restrict__ y,
                                                         unsigned char* __restrict__ mask) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int c = (int)((e / HW) % C);
  float v = x[e] * scale[c] + shift[c];
  if (addend) v += addend[e];
  if (RELU) {
    v = fmaxf(v, 0.f);
    if (mask) mask[e] = v > 0.f;
  }
  y[e] = v;
}

}  // namespace lk

using namespace lk;

extern "C" int lk_bn_act_fwd_f32(const float* x, const float* scale, const float* shift, const float* addend, int64_t total,
                                 int64_t C, int64_t HW, int relu, float* y, unsigned char* mask, void* stream) {
  LK_REQUIRE(x && scale && shift && y && total >= 0 && C > 0 && HW > 0 && total % (C * HW) == 0 && C < (1ll << 31) &&
                 HW < (1ll << 31),
             "lk_bn_act_fwd_f32: bad arguments");
  if (total == 0) return LK_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = HW % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0 && (uintptr_t)addend % 16 == 0 &&
                   (!mask || (uintptr_t)mask % 4 == 0);
  const int64_t nb = vec ? (total / 4 + 255) / 256 : (total + 255) / 256;
  LK_REQUIRE(nb < (1ll << 31), "lk_bn_act_fwd_f32: grid too large");
  if (vec) {
    if (relu)
      hipLaunchKernelGGL(bn_act_fwd_vec_kernel<true>, dim3((unsigned)nb), dim3(256), 0, st, x, scale, shift, addend, total,
                         (int)C, (int)HW, y, mask);
    else
      hipLaunchKernelGGL(bn_act_fwd_vec_kernel<false>, dim3((unsigned)nb), dim3(256), 0, st, x, scale, shift, addend, total,
                         (int)C, (int)HW, y, mask);
  } else {
    if (relu)
      hipLaunchKernelGGL(bn_act_fwd_kernel<true>, dim3((unsigned)nb), dim3(256), 0, st, x, scale, shift, addend, total, (int)C,
                         (int)HW, y, mask);
    else
      hipLaunchKernelGGL(bn_act_fwd_kernel<false>, dim3((unsigned)nb), dim3(256), 0, st, x, scale, shift, addend, total,
                         (int)C, (int)HW, y, mask);
  }
  return check_launch("bn_act_fwd_kernel");
}

extern "C" int lk_vjp_scale_mask_f32(const float* g, const float* g2, const void* m, int m_is_float, const float* scale, int64_t S,
                                     int64_t per_sample, int64_t C, int64_t HW, float* out, void* stream) {
  LK_REQUIRE(g && out && S >= 0 && per_sample >= 0, "lk_vjp_scale_mask_f32: bad arguments");
  LK_REQUIRE(!scale || (C > 0 && HW > 0 && per_sample % (C * HW) == 0), "lk_vjp_scale_mask_f32: per_sample must be B*C*HW");
  LK_REQUIRE(S < (1 << 30) && C < (1ll << 31) && HW < (1ll << 31), "lk_vjp_scale_mask_f32: extents too large");
  if (S == 0 || per_sample == 0) return LK_OK;
  const bool vec = per_sample % 4 == 0 && (!scale || HW % 4 == 0) && ((uintptr_t)g % 16 == 0) &&
                   ((uintptr_t)g2 % 16 == 0) && ((uintptr_t)out % 16 == 0) && (!m || (uintptr_t)m % (m_is_float ? 16 : 4) == 0);
  const int Ci = scale ? (int)C : 1, HWi = scale ? (int)HW : 1;
  hipStream_t st = (hipStream_t)stream;
  if (vec) {
    const int64_t nb = (per_sample / 4 + 255) / 256;
    LK_REQUIRE(nb < (1ll << 31), "lk_vjp_scale_mask_f32: grid too large");
    if (m_is_float)
      hipLaunchKernelGGL(vjp_scale_mask_vec_kernel<true>, dim3((unsigned)nb), dim3(256), 0, st, g, g2, m, scale, (int)S,
                         per_sample, Ci, HWi, out);
    else
      hipLaunchKernelGGL(vjp_scale_mask_vec_kernel<false>, dim3((unsigned)nb), dim3(256), 0, st, g, g2, m, scale, (int)S,
                         per_sample, Ci, HWi, out);
  } else {
    const int64_t nb = (per_sample + 255) / 256;
    LK_REQUIRE(nb < (1ll << 31), "lk_vjp_scale_mask_f32: grid too large");
    if (m_is_float)
      hipLaunchKernelGGL(vjp_scale_mask_kernel<true>, dim3((unsigned)nb), dim3(256), 0, st, g, g2, m, scale, (int)S,
                         per_sample, Ci, HWi, out);
    else
      hipLaunchKernelGGL(vjp_scale_mask_kernel<false>, dim3((unsigned)nb), dim3(256), 0, st, g, g2, m, scale, (int)S,
                         per_sample, Ci, HWi, out);
  }
  return check_launch("vjp_scale_mask_kernel");
}
