// Convolution as an implicit GEMM on NHWC tensors with fp32-level products from TWO-PIECE fp16 operands.
//
// What it replaces: the C reverse passes of curvlinops' KFACLinearOperator._compute_kfac as driven by
// CurvlinopsInterface.kron (laplace/curvature/curvlinops.py:87-100) — for a convolution layer each of them is a
// backward-data convolution; the seed-batched sweep (laplace_amd/sweep.py) folds all seeds into one batch, and this
// kernel is that sweep's convolution: backward-data of stride-1 and stride-2 convs (and the forward form, same GEMM).
//
// Arithmetic.  gfx950 has no reduced-precision fast path for fp32 inputs, and the exact fp32 MFMA runs at the vector
// rate (157 TFLOP/s).  Every operand tensor is therefore scaled by a power of two and split ONCE, by its producer, into
// two fp16 planes:   x * 2^s = h + l + e,   h = fp16(x 2^s),  l = fp16(x 2^s - h),
//   |e| <= max(2^-22 |x 2^s|, 2^-25)           (fp16 keeps 11 significant bits; l is exact down to the subnormals),
// with s chosen so that the tensor's largest magnitude lands in [2^14, 2^15): a 22-bit significand for every element
// within 2^-17 of the largest one and a fixed-point floor of 2^-39 of the largest one below that.  A product is then
//   x y 2^(s+s') ~= h h' + h l' + l h'          (dropped: l l' <= 2^-22 |x y 2^(s+s')|)
// i.e. THREE v_mfma_f32_32x32x16_f16 (32 cycles each, K = 16, fp32 accumulation) in place of eight
// v_mfma_f32_32x32x2_f32 (64 cycles each): 96 instead of 512 matrix-pipe cycles per 32x32x16 block, a ceiling of
// 2500 / 3 = 833 TFLOP/s fp32-equivalent.  Parity (1e-4 of the largest element, BASELINE.json) is tested against fp64.
//
// GEMM view:  out[p][n] = sum_t sum_c in[pix(p, t)][c] * Wt[t][n][c]   (p: output pixel, t: tap, out-of-image taps = 0)
//   backward-data, stride 1:  in = cotangent of the conv output, n = conv input channel, Wt[t][ci][co] = W[co][ci][kh][kw]
//   backward-data, stride 2:  one launch per output-pixel parity class (each class sees a fixed subset of the taps)
//   forward:                  in = conv input, n = conv output channel, Wt[t][co][ci] = W[co][ci][kh][kw]
// Both operands have k (channels) contiguous in memory — what a lane of the 16-bit MFMA reads (8 consecutive k = 16 B) —
// and a tap shift moves the A operand by whole channel vectors, so every load is an aligned 16-byte load.
//
// Data path: global_load_lds (16 B per lane, straight into LDS, no VGPR round trip) into a double-buffered stage of
// [A_h | A_l | B_h | B_l], each [rows][BK] fp16 with the 16-byte slots of a row XOR-swizzled (on the SOURCE address, as
// the LDS destination of an LDS-DMA is lane-linear) so that the ds_read_b128 fragment reads are conflict-free; taps that
// fall outside the image read a zero block.  One workgroup = 4 waves, each 64 x 64 (2 x 2 MFMA tiles) of a 128 x 128
// (or 256 x 64) output tile; the epilogue un-scales, optionally accumulates into `out`, and folds max|out| into a
// device word (what the next producer needs to choose ITS scale).
#include "lk_common.h"

#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

namespace lk {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- power-of-two scaling -------------------------------------------------------------------------------------------
// exponent s such that amax * 2^s lies in [2^14, 2^15)  (clamped so that 2^s and every scaled element stay finite)
__device__ __forceinline__ int scale_exp_for(float amax) {
  int be = (int)((__float_as_uint(amax) >> 23) & 0xffu);
  if (be == 0) be = 1;  // zero / subnormal tensors: largest scale that is safe for anything below 2^-126
  int s = 14 - (be - 127);
  return s > 120 ? 120 : s;
}
__device__ __forceinline__ float exp2i(int s) {  // 2^s, -126 <= s <= 127
  return __uint_as_float((unsigned)(127 + s) << 23);
}
__device__ __forceinline__ void split2(float x, float sc, _Float16& h, _Float16& l) {
  // the scaled value is made opaque first: h and the residual must come from the SAME fp32 value (hipcc otherwise fuses
  // the residual into fp16(x * sc - h') with h' rounded from the exact product, see lk_sweep16.hip)
  float xs = x * sc;
  asm volatile("" : "+v"(xs));
  h = (_Float16)xs;
  l = (_Float16)(xs - (float)h);
}

// maximum of a 256-thread workgroup into the result word: ONE atomic per workgroup (same-address atomics serialise in L2 at
// ~5-10 ns each: one per wave of a 2048-workgroup launch was a 40-80 us floor under every measurement, whatever its size)
__device__ __forceinline__ void block_max_to_word(unsigned m, unsigned* __restrict__ out) {
  __shared__ unsigned wave_max[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
    if (m) atomicMax(out, m);
  }
}

// the same at the end of a convolution workgroup of NW waves: the waves' maxima meet in LDS, one atomic per workgroup
// (`scale`: exact power-of-two factor applied to the float the bits stand for)
template <int NW>
__device__ __forceinline__ void conv_block_max(unsigned vmax, float scale, unsigned* __restrict__ out, char* smem) {
  unsigned* conv_wave_max = reinterpret_cast<unsigned*>(smem);  // (the kernels' LDS budgets are exact: no static word beside them)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) vmax = max(vmax, (unsigned)__shfl_xor((int)vmax, off, 64));
  __syncthreads();  // every wave is done with what the arena held
  if ((threadIdx.x & 63) == 0) conv_wave_max[threadIdx.x >> 6] = vmax;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < NW; ++w) vmax = max(vmax, conv_wave_max[w]);
    if (vmax) atomicMax(out, __float_as_uint(__uint_as_float(vmax) * scale));
  }
}

// ---- max |x| of a tensor as the bit pattern of a non-negative float (order-preserving as unsigned; atomicMax is
//      order-independent, so the result is run-to-run deterministic) ----------------------------------------------------
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t n, const float* __restrict__ cscale,
                                                     int64_t inner, int C, unsigned* __restrict__ out) {
  unsigned m = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float v = x[i];
    if (cscale) v *= cscale[(i / inner) % C];
    m = max(m, __float_as_uint(v) & 0x7fffffffu);
  }
  block_max_to_word(m, out);
}
// the same without a channel scale over float4s, four of them in flight per thread (x 16-byte aligned, n % 4 == 0): the
// scalar form above streams at ~1.6 TB/s, and the stacked pixel-pair inputs it is asked to measure are up to 268 MB
__global__ __launch_bounds__(256) void absmax4_kernel(const float4* __restrict__ x, int64_t n4, unsigned* __restrict__ out) {
  unsigned m = 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  auto fold = [&](float4 v) {
    m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), __float_as_uint(v.y) & 0x7fffffffu);
    m = max(max(m, __float_as_uint(v.z) & 0x7fffffffu), __float_as_uint(v.w) & 0x7fffffffu);
  };
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 a = x[i], b = x[i + stride], c = x[i + 2 * stride], d = x[i + 3 * stride];
    fold(a), fold(b), fold(c), fold(d);
  }
  for (; i < n4; i += stride) fold(x[i]);
  block_max_to_word(m, out);
}

// copy + max|x| in one pass (float4s): the stacked pixel-pair inputs are copied into their group buffer minibatch by
// minibatch anyway; measuring them on the way saves the separate pass over the whole stack before it is split
// (`out` is NOT reset here: it runs over the minibatches of a group)
__global__ __launch_bounds__(256) void copy_absmax4_kernel(const float4* __restrict__ x, float4* __restrict__ y, int64_t n4,
                                                           unsigned* __restrict__ out) {
  unsigned m = 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 v = x[i];
    y[i] = v;
    m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), __float_as_uint(v.y) & 0x7fffffffu);
    m = max(max(m, __float_as_uint(v.z) & 0x7fffffffu), __float_as_uint(v.w) & 0x7fffffffu);
  }
  block_max_to_word(m, out);
}

// ---- fp32 [rows][C] (NHWC) -> two fp16 planes, scale from the device-side bound `amax[0]` (* bound_mul) --------------
__global__ __launch_bounds__(256) void split_f16x2_kernel(const float* __restrict__ x, int64_t n8,
                                                          const float* __restrict__ amax, float bound_mul,
                                                          _Float16* __restrict__ ph, _Float16* __restrict__ pl,
                                                          int* __restrict__ sexp) {
  const int s = scale_exp_for(amax[0] * bound_mul);
  if (blockIdx.x == 0 && threadIdx.x == 0) sexp[0] = s;
  const float sc = exp2i(s);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    f16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      _Float16 hh, ll;
      split2(v[j], sc, hh, ll);
      h[j] = hh, l[j] = ll;
    }
    reinterpret_cast<f16x8*>(ph)[i] = h;
    reinterpret_cast<f16x8*>(pl)[i] = l;
  }
}

// ---- patch matrix of a convolution as split planes: x [B][H][W][C] fp32 -> [B Ho Wo][Kp] x (h, l), ONE scale ------------------
// Row (b, oh, ow), column k = (kh kw) C + c of the unfolded input (the kernels' native column order, taps outermost), zero for the
// taps outside the image and for the padding columns k >= KH KW C up to Kp (the split Gram engine takes 64 or a multiple of 128
// columns).  What the A factors of the strided and stem convolutions are the Gram of: round 6 moved them from the exact-fp32
// MFMA kernel (lk_gram_conv_nhwc_f32: 170 - 190 us per c4 layer on the pipe that is 16 x slower) onto lk_gram_tn_f16x2 (67 - 70 us)
// behind this pass (profiles/r06_gramconv16_bench.log).  A thread = 8 consecutive columns of one row.
struct Im2colGeom {
  int B, H, W, C, KH, KW, stride, pad, Ho, Wo, Kp, n;
  FastDiv div_wo, div_howo, div_c, div_kw, div_k8;
};
__global__ __launch_bounds__(256) void im2col_split_f16x2_kernel(const float* __restrict__ x, const Im2colGeom g,
                                                                 const float* __restrict__ amax, _Float16* __restrict__ ph,
                                                                 _Float16* __restrict__ pl, int* __restrict__ sexp, int64_t total8) {
  const int s = scale_exp_for(amax[0]);
  if (blockIdx.x == 0 && threadIdx.x == 0) sexp[0] = s;
  const float sc = exp2i(s);
  const int k8n = g.Kp >> 3;
  const bool vec = (g.C & 7) == 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (int64_t)gridDim.x * 256) {
    const int row = (int)fdiv((int)i, g.div_k8), k0 = ((int)i - row * k8n) * 8;  // (total8 < 2^31: checked by the host)
    const int b = fdiv(row, g.div_howo), rem = row - b * g.Ho * g.Wo, oh = fdiv(rem, g.div_wo), ow = rem - oh * g.Wo;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (vec) {  // the 8 columns are 8 channels of ONE tap
      if (k0 < g.n) {
        const int t = fdiv(k0, g.div_c), c = k0 - t * g.C, dy = fdiv(t, g.div_kw), dx = t - dy * g.KW;
        const int ih = oh * g.stride + dy - g.pad, iw = ow * g.stride + dx - g.pad;
        if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W) {
          const float4* src = reinterpret_cast<const float4*>(x + (((int64_t)b * g.H + ih) * g.W + iw) * g.C + c);
          const float4 a = src[0], bb = src[1];
          v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = bb.x, v[5] = bb.y, v[6] = bb.z, v[7] = bb.w;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        if (k >= g.n) continue;
        const int t = fdiv(k, g.div_c), c = k - t * g.C, dy = fdiv(t, g.div_kw), dx = t - dy * g.KW;
        const int ih = oh * g.stride + dy - g.pad, iw = ow * g.stride + dx - g.pad;
        if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W) v[j] = x[(((int64_t)b * g.H + ih) * g.W + iw) * g.C + c];
      }
    }
    f16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      _Float16 hh, ll;
      split2(v[j], sc, hh, ll);
      h[j] = hh, l[j] = ll;
    }
    reinterpret_cast<f16x8*>(ph)[i] = h;
    reinterpret_cast<f16x8*>(pl)[i] = l;
  }
}

// ---- the same with ONE SCALE PER IMAGE (leading dimension): x [N][per] -> planes, sexp[N], amax[N] --------------------------
// The forward activations of a minibatch are split image by image: a ReLU network is positively homogeneous, so the
// activations of an image follow ITS magnitude through every layer, and a mask decided on values resolved only to 2^-39 of
// the minibatch's largest image flips signs that fp32 would not (measured: one G block 3e-3 off on a minibatch spanning
// 1e-3 .. 1e+3).  GEMM rows never mix images in the forward (conv_f16x2_kernel: the epilogue un-scales row by row), so the
// per-image scale costs nothing there; the reverse sweep's cotangents keep one scale per tensor (their Grams sum over
// samples).  Two launches: per-image max|x| (one atomic per workgroup into amax[n]), then the split.
__global__ __launch_bounds__(256) void absmax_images_kernel(const float* __restrict__ x, int64_t per4, unsigned* __restrict__ amax) {
  const float4* xs = reinterpret_cast<const float4*>(x) + (int64_t)blockIdx.y * per4;
  unsigned m = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per4; i += (int64_t)gridDim.x * 256) {
    const float4 v = xs[i];
    m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), __float_as_uint(v.y) & 0x7fffffffu);
    m = max(max(m, __float_as_uint(v.z) & 0x7fffffffu), __float_as_uint(v.w) & 0x7fffffffu);
  }
  block_max_to_word(m, amax + blockIdx.y);
}
__global__ __launch_bounds__(256) void split_images_f16x2_kernel(const float* __restrict__ x, int64_t per8,
                                                                 const unsigned* __restrict__ amax, _Float16* __restrict__ ph,
                                                                 _Float16* __restrict__ pl, int* __restrict__ sexp) {
  const int n = blockIdx.y;
  const int s = scale_exp_for(__uint_as_float(amax[n]));
  if (blockIdx.x == 0 && threadIdx.x == 0) sexp[n] = s;
  const float sc = exp2i(s);
  const int64_t base = (int64_t)n * per8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per8; i += (int64_t)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(x)[2 * (base + i)], b = reinterpret_cast<const float4*>(x)[2 * (base + i) + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    f16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      _Float16 hh, ll;
      split2(v[j], sc, hh, ll);
      h[j] = hh, l[j] = ll;
    }
    reinterpret_cast<f16x8*>(ph)[base + i] = h;
    reinterpret_cast<f16x8*>(pl)[base + i] = l;
  }
}

// ---- weights: W[co][ci][kh][kw] (* cscale[co]) -> planes [2][T][N][K] fp16 ---------------------------------------------
// transpose = 1 (backward-data): n = ci, k = co;  transpose = 0 (forward): n = co, k = ci.  Tap t = kh * KW + kw.
__global__ __launch_bounds__(256) void conv_prep_weights_kernel(const float* __restrict__ W, int Co, int Ci, int T,
                                                                int transpose, const float* __restrict__ cscale,
                                                                const unsigned* __restrict__ amax,
                                                                _Float16* __restrict__ planes, int* __restrict__ sexp) {
  const int s = scale_exp_for(__uint_as_float(amax[0]));
  if (blockIdx.x == 0 && threadIdx.x == 0) sexp[0] = s;
  const float sc = exp2i(s);
  const int N = transpose ? Ci : Co, K = transpose ? Co : Ci;
  const int64_t total = (int64_t)T * N * K;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % K), n = (int)((i / K) % N), t = (int)(i / ((int64_t)K * N));
    const int co = transpose ? k : n, ci = transpose ? n : k;
    float v = W[((int64_t)co * Ci + ci) * T + t];
    if (cscale) v *= cscale[co];
    _Float16 h, l;
    split2(v, sc, h, l);
    planes[i] = h;
    planes[total + i] = l;
  }
}

// ---- the implicit GEMM -------------------------------------------------------------------------------------------------
struct ConvGeom {
  int N, Hi, Wi, Ci;   // input tensor [N][Hi][Wi][Ci] (Ci = GEMM K per tap)
  int Hc, Wc;          // output class grid: GEMM rows m = (n, i, j), i < Hc, j < Wc
  int Ho, Wo, Co;      // output tensor [N][Ho][Wo][Co] (Co = GEMM N); output pixel = (i * os + oh0, j * os + ow0)
  int os, oh0, ow0;
  int im;              // input pixel of tap t = (i * im + dh[t], j * im + dw[t])
  int T;               // taps of this launch
  int dh[9], dw[9], wt[9];  // per tap: offsets and the weight slice (index into the [T_w][Co][Ci] planes)
  FastDiv div_hw, div_w;    // GEMM row m -> (n, i, j):  n = m / (Hc*Wc), i = rem / Wc
  FastDiv div_n;            // ... or, position-major (pmajor): pixel p = m / N, n = m % N
  int pmajor;               // rows ordered (pixel, image): a tile of a small map holds ONE pixel position of many images,
                            // so the taps that fall outside the image there are skipped as whole K stages
  int dense;                // the output grid is the output tensor (os = 1, Hc = Ho, Wc = Wo): output pixel = m
  int out_nchw;             // write out[n][co][pixel] (dense grids with Ho*Wo % 4 == 0 only): float4 along the pixels
  int a_nsexp;              // entries of a_sexp: 1 (one scale for the tensor) or N (one per image: the forward's activations)
  int out_planes;           // with out_nchw: emit the position-contiguous output as two fp16 planes [n][co][pixel] scaled per
                            // a_sexp entry from the bound in_amax * l1(W) (ConvVjp::out_h / out_l / out_sexp / in_amax / w_l1) —
                            // what the predictive's quadratic-form kernel stages without splitting anything itself
};

template <int BM_, int BN_, int BK_, int WM_, int WN_, int NBUF_ = 2, int WPE_ = 2>
struct ConvCfg {
  static constexpr int EPI_LDS = WM_ * WN_ * (BM_ / WM_) * (BN_ / WN_ + 4) * 4;  // staging image of the fused (VJP) epilogue
  static constexpr bool FUSABLE = NBUF_ == 2 && BK_ == 32 && WM_ * WN_ == 4;  // shapes the VJP epilogue is built for
  static constexpr int WPE = WPE_;                        // waves per SIMD the register allocation is sized for
  static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_;
  static constexpr int NBUF = NBUF_;                      // LDS stages: NBUF - 1 stages of loads are in flight
  static constexpr int Q = BK / 8;                       // 16-byte slots per row
  static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;  // MFMA tiles per wave
  static constexpr int A_PLANE = BM * BK * 2, B_PLANE = BN * BK * 2;  // bytes
  static constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;
  static constexpr int A_SLOTS = BM * Q, B_SLOTS = BN * Q;   // per plane
  static constexpr int NT = 64 * WM * WN;                 // threads: 4 waves (2 workgroups per CU) or 8 (one)
  static constexpr int A_LD = 2 * A_SLOTS / NT, B_LD = 2 * B_SLOTS / NT;  // LDS-DMA instructions per thread and stage
  static_assert(WM * WN == 4 || WM * WN == 8, "four or eight waves");
  static_assert(A_SLOTS % NT == 0 && (2 * B_SLOTS) % NT == 0 && B_SLOTS % 64 == 0, "whole instructions, waves inside a plane");
};

// swizzle of the 16-byte slot index within a row: a 16-lane group of ds_read_b128 touches 16 distinct rows (mod 16) at
// one logical slot; physical = logical ^ f(row) spreads them over all 16 slots of the 256-byte bank row
template <int Q>
__device__ __forceinline__ int swz(int row) {
  return Q == 4 ? ((row >> 2) & 3) : Q == 8 ? ((row >> 1) & 7) : Q == 2 ? ((row >> 3) & 1) : (row & (Q - 1));
}

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// Fused epilogue ("VJP form"): instead of the fp32 tensor the launch emits what the element-wise VJP kernel
// (lk_sweep16.hip: vjp_nhwc_split_kernel) would make of it,
//   o = (conv + add) * M * scale[channel]      add: split tensor (cotangent of a residual join), M: activation multiplier
// as a split tensor.  Its scale comes from a GUARANTEED bound known before the launch,
//   max|conv| <= max|in| * l1(W),   l1(W) = max_n sum_{t,k} |Wt[t][n][k]|   (a per-layer constant, device word)
// times the bounds of the other factors — loose by a few bits, which only costs fixed-point range (see the header of this
// file) — and the launch also measures max|o| itself (amax_out): the NEXT launch derives its bound from that measured
// value, so the slack does not compound along the sweep.
struct ConvVjp {
  const unsigned* in_amax;   // measured max|in| (bit pattern) or NULL: 2^(15 - in_sexp)
  const float* w_l1;         // l1(W) of the prepared weights (including their folded channel scale)
  const _Float16 *add_h, *add_l;
  const int* add_sexp;
  const void* mask;          // uint8 (mask_float = 0) or fp32 multiplier, [mask_rows][Co]; row of output pixel m: m % mask_rows
  int mask_float;
  const unsigned* mult_amax; // max|M| for fp32 multipliers (NULL: 1)
  int64_t mask_rows;
  FastDiv div_mask;          // GEMM row m (< 2^31) -> m / mask_rows without a 64-bit division in the epilogue
  const float* scale;        // [Co] or NULL
  const unsigned* scale_amax;
  _Float16 *out_h, *out_l;
  int* out_sexp;
  const _Float16 *wc_h, *wc_l;  // the same weights chunk-major, [tap][Ci / 16][Co][16] per plane (persistent window form), or NULL
  // ---- forward epilogue (conv_f16x2_kernel<CFG, true, true>, lk_conv_bn_act_nhwc_f16x2): eval-mode BatchNorm + residual add
  //      + ReLU on the accumulators, y = act(conv * fwd_scale[co] + fwd_shift[co] + fwd_addend); in_amax / a_sexp per image,
  //      w_l1 = l1 norm of the weights (the bound of the convolution's output), out_h / out_l / out_sexp[N] the split planes
  float* fwd_y;                   // fp32 NHWC result
  unsigned char* fwd_mask;        // NHWC bytes y > 0, or NULL
  const float *fwd_scale, *fwd_shift;             // [Co]
  const unsigned *fwd_scale_amax, *fwd_shift_amax;  // words
  const float* fwd_addend;        // fp32 NHWC or NULL
  const float* fwd_addend_bound;  // 1 or N floats
  int fwd_addend_nbound, fwd_in_namax, fwd_act;
  float* fwd_bound;               // [N] guaranteed bound of max|y_n| (what out_sexp[n] was derived from)
  unsigned* fwd_amax;             // [N] measured max|y_n| (bit patterns; zeroed by the caller)
};

template <typename CFG, bool FUSE, bool FWD = false>
__global__ __launch_bounds__(CFG::NT) __attribute__((amdgpu_waves_per_eu(CFG::WPE, CFG::WPE))) void conv_f16x2_kernel(const ConvGeom g, const _Float16* __restrict__ Ah,
                                                         const _Float16* __restrict__ Al, const _Float16* __restrict__ Wh,
                                                         const _Float16* __restrict__ Wl, const int* __restrict__ a_sexp,
                                                         const int* __restrict__ w_sexp, const _Float16* __restrict__ zero16,
                                                         float* __restrict__ out, int accumulate,
                                                         unsigned* __restrict__ amax_out, int nb_m, const ConvVjp fz) {
  constexpr int BM = CFG::BM, BN = CFG::BN, BK = CFG::BK, Q = CFG::Q, TM = CFG::TM, TN = CFG::TN, NT = CFG::NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware tile order: consecutive block ids run on different XCDs (id % 8); give every XCD a contiguous range of
  // (n-tile major, m-tile minor) tiles so that the blocks sharing an L2 share one weight panel and neighbouring pixels
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, x = bid % 8, j = bid / 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
  }
  const int M = g.N * g.Hc * g.Wc;
  const int KC = g.Ci / BK;
  const int tile = bid;
  const int tile_n = tile / nb_m, tile_m = tile % nb_m;

  // ---- per-thread staging context: which rows / slots this thread feeds, fixed for the whole K loop
  // A: A_LD instructions; instruction i covers slots [i*256, i*256+256) of the concatenated [plane h | plane l] image
  int64_t a_off[CFG::A_LD / 2];   // element offset of (pixel (i*im, j*im), channel 0) per handled row
  unsigned a_valid[CFG::A_LD / 2];  // bit t: tap t is inside the image (and the row exists)
  int a_lq[CFG::A_LD / 2];
  static_assert(CFG::A_LD % 2 == 0, "the two planes are fed by the same threads");
#pragma unroll
  for (int i = 0; i < CFG::A_LD / 2; ++i) {
    const int slot = i * NT + tid;  // within a plane
    const int row = slot / Q, pq = slot % Q;
    a_lq[i] = pq ^ swz<Q>(row);
    const int m = tile_m * BM + row;
    unsigned valid = 0;
    int64_t off = 0;
    if (m < M) {
      int n, rem;
      if (g.pmajor) rem = fdiv(m, g.div_n), n = m - rem * g.N;
      else n = fdiv(m, g.div_hw), rem = m - n * (g.Hc * g.Wc);
      const int ci_ = fdiv(rem, g.div_w);
      const int ih = ci_ * g.im, iw = (rem - ci_ * g.Wc) * g.im;
      off = (((int64_t)n * g.Hi + ih) * g.Wi + iw) * g.Ci;
      for (int t = 0; t < g.T; ++t) {
        const int hh = ih + g.dh[t], ww = iw + g.dw[t];
        if (hh >= 0 && hh < g.Hi && ww >= 0 && ww < g.Wi) valid |= 1u << t;
      }
    }
    a_off[i] = off, a_valid[i] = valid;
  }
  // B: instruction i covers slots [i*256, i*256+256) of the concatenated [plane h | plane l] image (a plane may be
  // smaller than one instruction: BN * Q slots)
  int64_t b_off[CFG::B_LD];
  bool b_ok[CFG::B_LD];
  bool b_low[CFG::B_LD];
#pragma unroll
  for (int i = 0; i < CFG::B_LD; ++i) {
    const int cs = i * NT + tid;
    const int slot = cs % CFG::B_SLOTS;
    const int row = slot / Q, pq = slot % Q;
    const int n = tile_n * BN + row;
    b_low[i] = cs >= CFG::B_SLOTS;
    b_ok[i] = n < g.Co;
    b_off[i] = (int64_t)n * g.Ci + (pq ^ swz<Q>(row)) * 8;
  }
  const int64_t w_tap = (int64_t)g.Co * g.Ci;

  // The scale words (and, for the fused epilogue, the bound they combine to) are read HERE, at the top: left where they
  // are used — behind the K loop — they were a chain of dependent global loads at the end of every tile.
  const int sexp_a = a_sexp[0], sexp_w = w_sexp[0];
  const float inv_a = exp2i(-sexp_a < -126 ? -126 : -sexp_a), inv_w = exp2i(-sexp_w < -126 ? -126 : -sexp_w);
  int so = 0;
  float sc_out = 1.f, inv2 = 0.f;
  static_assert(FUSE || !FWD, "the forward epilogue uses the fused epilogue's staging");
  if constexpr (FUSE && !FWD) {
    // scale of the result from the guaranteed bound (every thread computes the same few flops; one writes the word)
    float bound = fz.in_amax ? __uint_as_float(fz.in_amax[0]) : exp2i(15 - sexp_a < -126 ? -126 : (15 - sexp_a > 127 ? 127 : 15 - sexp_a));
    bound *= fz.w_l1[0];
    if (fz.add_h) {
      const int s2 = fz.add_sexp[0];
      bound += exp2i(15 - s2 < -126 ? -126 : (15 - s2 > 127 ? 127 : 15 - s2));
      inv2 = exp2i(-s2 < -126 ? -126 : -s2);
    }
    if (fz.mask && fz.mask_float && fz.mult_amax) bound *= __uint_as_float(fz.mult_amax[0]);
    if (fz.scale) bound *= __uint_as_float(fz.scale_amax[0]);
    so = scale_exp_for(bound);
    sc_out = exp2i(so);
  }

  // Taps that reach no row of this tile are dropped from the K loop (position-major tiles of small maps: a corner pixel
  // of a 4 x 4 map sees 4 of the 9 taps).  The list is uniform over the workgroup: OR of every thread's validity bits.
  unsigned long long tap_list = 0;  // 4 bits per listed tap
  int ntap = 0;
  {
    unsigned v = 0;
#pragma unroll
    for (int i = 0; i < CFG::A_LD / 2; ++i) v |= a_valid[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v |= (unsigned)__shfl_xor((int)v, off, 64);
    unsigned* red = reinterpret_cast<unsigned*>(smem);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    v = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) v |= red[w];
    v = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    __syncthreads();  // the words are read before the first stage lands on them
    for (int t = 0; t < g.T; ++t)
      if ((v >> t) & 1u) tap_list |= (unsigned long long)t << (4 * ntap), ++ntap;
  }

  const int nstage = ntap * KC;

  auto stage = [&](int s, int buf) {
    // K order: taps major, channel chunks minor.  (The other order — the nine taps of one 32-channel chunk back to back,
    // so that only a quarter-to-sixteenth-depth window has to survive in L2 between them — was measured in round 3: better
    // on the stride-2 classes, worse on the 64- and 512-channel layers, 10.66 vs 10.50 ms per step; removed in round 4.)
    const int tj = s / KC, kc = s - tj * KC;
    const int t = (int)((tap_list >> (4 * tj)) & 15ull);
    char* base = smem + buf * CFG::STAGE;
    const int64_t tap_off = ((int64_t)g.dh[t] * g.Wi + g.dw[t]) * g.Ci + kc * BK;
#pragma unroll
    for (int i = 0; i < CFG::A_LD / 2; ++i) {
      const bool ok = (a_valid[i] >> t) & 1u;
      const int64_t e = a_off[i] + tap_off + a_lq[i] * 8;
      const _Float16* sh = ok ? Ah + e : zero16;
      const _Float16* sl = ok ? Al + e : zero16;
      char* dh_ = base + (i * NT + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds((gbl_void*)sh, (lds_void*)dh_, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void*)sl, (lds_void*)(dh_ + CFG::A_PLANE), 16, 0, 0);
    }
    const int64_t wbase = (int64_t)g.wt[t] * w_tap + kc * BK;
#pragma unroll
    for (int i = 0; i < CFG::B_LD; ++i) {
      const int64_t e = wbase + b_off[i];
      const _Float16* sp = b_ok[i] ? (b_low[i] ? Wl : Wh) + e : zero16;
      char* db = base + 2 * CFG::A_PLANE + (i * NT + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds((gbl_void*)sp, (lds_void*)db, 16, 0, 0);
    }
  };

  // ---- fragment addresses: lane reads row (tile*32 + lane&31), logical slot k16*2 + (lane>>5)
  const int wm = wave / CFG::WN, wn = wave % CFG::WN;
  const int lr = lane & 31, lh = lane >> 5;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- fused (VJP) epilogue, part 1: per-lane chunk context and the requests for the addend planes / mask bytes of ALL of
  // this lane's chunks (issued right behind the K loop: their latency runs under the hand-over barrier and the staging of
  // the accumulators.  Issuing them two K stages EARLIER was measured: 9.60 vs 9.33 ms per step — the vmcnt(0) of the last
  // hand-over then waits for HBM-latency loads in front of the last MFMAs.)
  constexpr int ROWS_W = TM * 32, COLS_W = TN * 32, PITCH = COLS_W + 4, C8 = COLS_W / 8, NIT = FUSE ? ROWS_W * C8 / 64 : 1;
  const int HWc = g.Hc * g.Wc;
  int64_t e_[NIT];
  int pix_[NIT];  // output pixel index of the chunk's row (< 2^31)
  bool ok_[NIT];
  f16x8 h2_[NIT], l2_[NIT];
  uint2 mk_[NIT];
  const bool pre_mask = fz.mask && !fz.mask_float;
  // row of the multiplier for output pixel p (the multiplier is shared by the seeds): p mod mask_rows in 32-bit
  // arithmetic (a 64-bit % here was a ~100-instruction software division per chunk: 8 per tile and lane)
  const int mrows = (int)fz.mask_rows;
  auto mask_row = [&](int p) { return p - fdiv(p, fz.div_mask) * mrows; };
  auto prefetch = [&](int it) {
    const int idx = it * 64 + lane;
    const int row = idx / C8, c8 = idx - row * C8;
    const int m = tile_m * BM + wm * ROWS_W + row;
    const int col0 = tile_n * BN + wn * COLS_W + c8 * 8;
    ok_[it] = m < M && col0 < g.Co;
    int64_t opix = m;  // dense grid: the output pixel index is the GEMM row, up to the position-major order
    if (g.pmajor) {
      const int rem = fdiv(m, g.div_n);
      opix = (int64_t)(m - rem * g.N) * HWc + rem;
    }
    e_[it] = ok_[it] ? opix * g.Co + col0 : 0;
    pix_[it] = ok_[it] ? (int)opix : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) h2_[it][j] = (_Float16)0.f, l2_[it][j] = (_Float16)0.f;
    mk_[it] = make_uint2(0x01010101u, 0x01010101u);
    if constexpr (FWD) {  // (the fp32 addend's eight values ride in the registers of the two addend planes)
      if (fz.fwd_addend && ok_[it]) {
        h2_[it] = *reinterpret_cast<const f16x8*>(fz.fwd_addend + e_[it]);
        l2_[it] = *reinterpret_cast<const f16x8*>(fz.fwd_addend + e_[it] + 4);
      }
      return;
    }
    if (fz.add_h && ok_[it]) {
      h2_[it] = *reinterpret_cast<const f16x8*>(fz.add_h + e_[it]);
      l2_[it] = *reinterpret_cast<const f16x8*>(fz.add_l + e_[it]);
    }
    if (pre_mask && ok_[it])
      mk_[it] = *reinterpret_cast<const uint2*>((const unsigned char*)fz.mask + (int64_t)mask_row((int)opix) * g.Co + col0);
  };

  {
  // Software pipeline over NBUF LDS stages: the loads of stages s+1 .. s+NBUF-1 are in flight while stage s is
  // computed (an L2 hit takes ~1.6k cycles here, two to three stage times).  vmcnt counts this wave's LDS-DMA
  // instructions in issue order, every stage issues exactly LD_PER_STAGE of them, so "stage s has landed" is
  // vmcnt(LD_PER_STAGE * (NBUF - 2)); the raw s_barrier (no implicit vmcnt(0) as __syncthreads would add) then makes it
  // true for every wave and also says that everybody has finished reading the buffer stage s+NBUF-1 is loaded into.
  constexpr int NBUF = CFG::NBUF, LD_PER_STAGE = CFG::A_LD + CFG::B_LD;
  static_assert(LD_PER_STAGE * (NBUF - 2) <= 63, "vmcnt is a 6-bit counter");
  constexpr int WAIT_STEADY = 0x0f70 | ((LD_PER_STAGE * (NBUF - 2)) & 15) | (((LD_PER_STAGE * (NBUF - 2)) >> 4) << 14);
#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i)
    if (i < nstage) stage(i, i);
  int buf = 0;
  for (int s = 0; s < nstage; ++s) {
    if (s + NBUF - 1 <= nstage)  // NBUF - 2 younger stages are outstanding behind stage s
      __builtin_amdgcn_s_waitcnt(WAIT_STEADY);
    else
      __builtin_amdgcn_s_waitcnt(0x0f70);  // tail: fewer stages in flight
    __builtin_amdgcn_s_barrier();
    if (s + NBUF - 1 < nstage) stage(s + NBUF - 1, buf == 0 ? NBUF - 1 : buf - 1);
    const int cur = buf;
    buf = buf + 1 == NBUF ? 0 : buf + 1;
    const char* base = smem + cur * CFG::STAGE;
    // fragments of step k16 + 1 are requested before the MFMAs of step k16 are issued (two register sets)
    f16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
    auto load_frags = [&](int k16, int set) {
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        const int row = (wm * TM + a) * 32 + lr;
        const int off = (row * Q + ((k16 * 2 + lh) ^ swz<Q>(row))) * 16;
        ah[set][a] = *reinterpret_cast<const f16x8*>(base + off);
        al[set][a] = *reinterpret_cast<const f16x8*>(base + CFG::A_PLANE + off);
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int row = (wn * TN + b) * 32 + lr;
        const int off = (row * Q + ((k16 * 2 + lh) ^ swz<Q>(row))) * 16;
        bh[set][b] = *reinterpret_cast<const f16x8*>(base + 2 * CFG::A_PLANE + off);
        bl[set][b] = *reinterpret_cast<const f16x8*>(base + 2 * CFG::A_PLANE + CFG::B_PLANE + off);
      }
    };
    load_frags(0, 0);
#pragma unroll
    for (int k16 = 0; k16 < BK / 16; ++k16) {
      const int set = k16 & 1;
      if (k16 + 1 < BK / 16) load_frags(k16 + 1, set ^ 1);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          f32x16 c = acc[a][b];
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[set][a], bh[set][b], c, 0, 0, 0);  // small terms first
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[set][a], bl[set][b], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[set][a], bh[set][b], c, 0, 0, 0);
          acc[a][b] = c;
        }
    }
  }
  }  // (pipeline form)

  // ---- epilogue: un-scale, store NHWC (a half-wave writes 32 consecutive channels = 128 B), max|out|
  unsigned vmax = 0;
  if constexpr (FWD) {
    // ---- forward epilogue: eval-mode BatchNorm + residual add + ReLU on the accumulators — what lk_conv_nhwc_f16x2 followed
    // by lk_bn_act_fwd_nhwc_f16x2 computes, to the bit (the same fp32 operations in the same order: un-scale by two powers of
    // two, one fma with the channel's scale / shift, one add of the addend), without the fp32 round trip of the
    // convolution's output and the second launch.  Emits y (fp32 NHWC), the ReLU mask bytes, y's split planes with ONE SCALE
    // PER IMAGE from the guaranteed bound (see bn_act_fwd_nhwc_kernel), that bound, and the measured max|y_n|: the rows'
    // maxima meet in LDS (slot = image - first image of the tile; position-major tiles: slot = row), one global atomic
    // per (workgroup, image).  Staging as in the VJP epilogue below: lane = 8 consecutive channels of one pixel.
    const bool per_img = g.a_nsexp > 1;
    const int m_first = tile_m * BM;
    const int n_lo = g.pmajor ? 0 : fdiv(m_first, g.div_hw);
    unsigned* amax_slot = reinterpret_cast<unsigned*>(smem + CFG::EPI_LDS);  // BM + 1 words behind the staging image
#pragma unroll
    for (int it = 0; it < NIT; ++it) prefetch(it);
    __syncthreads();  // every wave is done with the K loop's stage buffers
    for (int i = tid; i < BM + 1; i += NT) amax_slot[i] = 0u;
    float* img = reinterpret_cast<float*>(smem) + wave * (ROWS_W * PITCH);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          img[(a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * PITCH + b * 32 + lr] = acc[a][b][r];
    __syncthreads();  // the slots are zero before anybody's maximum arrives
    auto clampe = [](int e) { return e < -126 ? -126 : (e > 127 ? 127 : e); };
    const float s_amax = __uint_as_float(fz.fwd_scale_amax[0]), t_amax = __uint_as_float(fz.fwd_shift_amax[0]);
    const float x_mul = fz.w_l1[0];
    int c_n = -1, c_so = 0;
    float c_inv = 0.f, c_sc = 0.f, c_bound = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx / C8, c8 = idx - row * C8;
      const bool ok = ok_[it];
      const int col0 = tile_n * BN + wn * COLS_W + c8 * 8;
      const int opix = pix_[it];
      const int n = fdiv(opix, g.div_hw);
      unsigned rmax = 0;
      if (ok) {
        if (n != c_n) {  // (a tile of a large map lies in one or two images)
          const int sa = a_sexp[per_img ? n : 0];
          c_inv = exp2i(clampe(-sa));
          float bx = __uint_as_float(fz.in_amax[n < fz.fwd_in_namax ? n : fz.fwd_in_namax - 1]) * x_mul;
          c_bound = __fmaf_rn(bx, s_amax, t_amax);
          if (fz.fwd_addend) c_bound += fz.fwd_addend_bound[n < fz.fwd_addend_nbound ? n : fz.fwd_addend_nbound - 1];
          c_so = scale_exp_for(c_bound);
          c_sc = exp2i(clampe(c_so));
          c_n = n;
        }
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(img + row * PITCH + c8 * 8);
        const f32x4 p1 = *reinterpret_cast<const f32x4*>(img + row * PITCH + c8 * 8 + 4);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(fz.fwd_scale + col0), s1 = *reinterpret_cast<const f32x4*>(fz.fwd_scale + col0 + 4);
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(fz.fwd_shift + col0), t1 = *reinterpret_cast<const f32x4*>(fz.fwd_shift + col0 + 4);
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = __fmaf_rn(p0[j] * c_inv * inv_w, s0[j], t0[j]);
          v[4 + j] = __fmaf_rn(p1[j] * c_inv * inv_w, s1[j], t1[j]);
        }
        if (fz.fwd_addend) {
          const f32x4 q0 = __builtin_bit_cast(f32x4, h2_[it]), q1 = __builtin_bit_cast(f32x4, l2_[it]);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += q0[j], v[4 + j] += q1[j];
        }
        const int64_t e = e_[it];
        if (fz.fwd_act == 1) {
          unsigned m0 = 0, m1 = 0;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            v[j] = fmaxf(v[j], 0.f);
            if (v[j] > 0.f) (j < 4 ? m0 : m1) |= 1u << (8 * (j & 3));
          }
          if (fz.fwd_mask) *reinterpret_cast<uint2*>(fz.fwd_mask + e) = make_uint2(m0, m1);
        }
        *reinterpret_cast<f32x4*>(fz.fwd_y + e) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(fz.fwd_y + e + 4) = f32x4{v[4], v[5], v[6], v[7]};
        f16x8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          rmax = max(rmax, __float_as_uint(v[j]) & 0x7fffffffu);
          float xs = v[j] * c_sc;
          asm volatile("" : "+v"(xs));  // h and the residual from the SAME fp32 value (see split2)
          const _Float16 hh = (_Float16)xs;
          h[j] = hh;
          l[j] = (_Float16)(xs - (float)hh);
        }
        if (fz.out_h) {
          *reinterpret_cast<f16x8*>(fz.out_h + e) = h;
          *reinterpret_cast<f16x8*>(fz.out_l + e) = l;
        }
        if (col0 == 0 && opix == n * HWc) fz.out_sexp[n] = c_so, fz.fwd_bound[n] = c_bound;
      }
      // the C8 lanes of a row sit next to each other: their maxima meet in the row's first lane, which books the row
#pragma unroll
      for (int off = 1; off < C8; off <<= 1) rmax = max(rmax, (unsigned)__shfl_xor((int)rmax, off, 64));
      if (c8 == 0 && rmax) {
        const int slot = g.pmajor ? (wm * ROWS_W + row) : (n - n_lo);
        atomicMax(amax_slot + slot, rmax);
      }
    }
    __syncthreads();
    for (int i = tid; i < BM + 1; i += NT) {
      const unsigned v = amax_slot[i];
      if (!v) continue;
      int n = n_lo + i;
      if (g.pmajor) {
        const int m = m_first + i;
        n = m - fdiv(m, g.div_n) * g.N;
      }
      atomicMax(fz.fwd_amax + n, v);
    }
    return;
  }
  if constexpr (FUSE) {
    if (blockIdx.x == 0 && tid == 0) fz.out_sexp[0] = so;
    // the wave's 64 x 64 (32 x 32, ...) block goes through LDS: MFMA layout (lane = channel, registers = pixels) ->
    // lane = 8 consecutive channels of one pixel, i.e. 16-byte loads of the addend / mask and 16-byte stores of each plane
    static_assert(CFG::EPI_LDS == CFG::WM * CFG::WN * ROWS_W * PITCH * 4 && CFG::EPI_LDS <= 80 * 1024, "staging image: two workgroups per CU");
    // The addend planes and the mask bytes of ALL of this lane's chunks are requested first: their latency (an HBM miss
    // each) then runs under the hand-over barrier and the staging of the accumulators instead of once per chunk.
#pragma unroll
    for (int it = 0; it < NIT; ++it) prefetch(it);
    __syncthreads();  // every wave is done with the K loop's stage buffers
    float* img = reinterpret_cast<float*>(smem) + wave * (ROWS_W * PITCH);
    const float inv = inv_a * inv_w;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          img[(a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * PITCH + b * 32 + lr] = acc[a][b][r] * inv;
    // (LDS operations of one wave execute in order: no barrier between its own writes and reads)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx / C8, c8 = idx - row * C8;
      if (!ok_[it]) continue;
      const int col0 = tile_n * BN + wn * COLS_W + c8 * 8;
      const f32x4 p0 = *reinterpret_cast<const f32x4*>(img + row * PITCH + c8 * 8);
      const f32x4 p1 = *reinterpret_cast<const f32x4*>(img + row * PITCH + c8 * 8 + 4);
      float v[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
      const int64_t e = e_[it];
      if (fz.add_h) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += ((float)h2_[it][j] + (float)l2_[it][j]) * inv2;
      }
      float mult[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) mult[j] = sc_out;
      if (fz.mask) {
        if (fz.mask_float) {
          const int64_t em = (int64_t)mask_row(pix_[it]) * g.Co + col0;
          const f32x4 a = *reinterpret_cast<const f32x4*>((const float*)fz.mask + em);
          const f32x4 b = *reinterpret_cast<const f32x4*>((const float*)fz.mask + em + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) mult[j] *= a[j], mult[4 + j] *= b[j];
        } else {
          const uint2 u = mk_[it];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (!((u.x >> (8 * j)) & 0xffu)) mult[j] = 0.f;
            if (!((u.y >> (8 * j)) & 0xffu)) mult[4 + j] = 0.f;
          }
        }
      }
      if (fz.scale) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(fz.scale + col0), b = *reinterpret_cast<const f32x4*>(fz.scale + col0 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) mult[j] *= a[j], mult[4 + j] *= b[j];
      }
      f16x8 h, l;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float xs = v[j] * mult[j];
        asm volatile("" : "+v"(xs));  // h and the residual from the SAME fp32 value (see split2)
        const _Float16 hh = (_Float16)xs;
        h[j] = hh;
        l[j] = (_Float16)(xs - (float)hh);
        vmax = max(vmax, __float_as_uint(xs) & 0x7fffffffu);
      }
      *reinterpret_cast<f16x8*>(fz.out_h + e) = h;
      *reinterpret_cast<f16x8*>(fz.out_l + e) = l;
    }
    // max|o| = max|o 2^so| * 2^-so (exact: a power of two), kept as the bit pattern of a non-negative float
    if (amax_out) conv_block_max<NT / 64>(vmax, exp2i(-so < -126 ? -126 : -so), amax_out, smem);
    return;
  }
  // A operand with one scale per image (the forward's activations, lk_split_images_f16x2 / lk_bn_act_fwd_nhwc_f16x2): GEMM
  // rows never mix images, so the accumulators are un-scaled row by row here (the K loop does not know about it)
  const bool per_img = g.a_nsexp > 1;
  auto image_inv = [&](int n) {
    const int s = a_sexp[n];
    return exp2i(-s < -126 ? -126 : -s);
  };
  if (g.out_nchw) {
    // position-contiguous output [n][co][pixel] (what the predictive's quadratic-form kernel reads): a lane owns one
    // channel, registers r = 4q .. 4q+3 are four consecutive pixels of it -> one 16-byte store
    const int HW = g.Hc * g.Wc;
    int pl_ns = -1, pl_so = 0;
    float pl_sc = 0.f;
    const float pl_l1 = g.out_planes ? fz.w_l1[0] : 0.f;
    const bool pl_pair = g.out_planes && HW % 8 == 0;  // 8-pixel runs never straddle an image: the half-wave exchange applies
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = tile_m * BM + (wm * TM + a) * 32 + 8 * q + 4 * lh;
        if (m >= M) continue;
        const int n = fdiv(m, g.div_hw), pix = m - n * HW;
        const float inv_an = per_img ? image_inv(n) : inv_a;  // (the four pixels of a register quad belong to one image: HW % 4 == 0)
        if (g.out_planes) {
          // split planes: image n (or the whole tensor) scaled from the guaranteed bound max|in_n| * l1(W); the scale words
          // are re-read only when the image changes (a tile of a large map lies in one or two images)
          const int ns = per_img ? n : 0;
          if (ns != pl_ns) {
            const int sa = a_sexp[ns];
            float bound = fz.in_amax ? __uint_as_float(fz.in_amax[ns]) : exp2i(15 - sa < -126 ? -126 : (15 - sa > 127 ? 127 : 15 - sa));
            pl_so = scale_exp_for(bound * pl_l1);
            pl_sc = exp2i(pl_so) * inv_w;
            pl_ns = ns;
          }
          // Lanes l and l + 32 hold the two 4-pixel halves of an 8-pixel run of the same channel (rows 8q + 4 lh + j): they
          // swap one half each, so that lane l stores 16 bytes of the h plane and lane l + 32 16 bytes of the l plane — one
          // store instruction of 64 x 16 B per (a, q, b) instead of two of 64 x 8 B.
          const bool pair_ok = (m - 4 * lh) + 8 <= M;  // (the partner's rows exist: M % 8 == 0 on every dense grid with HW % 8 == 0)
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            const int col = tile_n * BN + (wn * TN + b) * 32 + lr;
            f16x4 h4, l4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float xs = acc[a][b][4 * q + j] * inv_an * pl_sc;
              asm volatile("" : "+v"(xs));  // h and the residual from the SAME fp32 value (see split2)
              const _Float16 hh = (_Float16)xs;
              h4[j] = hh;
              l4[j] = (_Float16)(xs - (float)hh);
            }
            if (col == 0 && pix == 0 && (per_img || n == 0) && col < g.Co) fz.out_sexp[ns] = pl_so;
            if (pl_pair && pair_ok) {
              // lh == 0 keeps h4 and receives the partner's h4; lh == 1 keeps l4 and receives the partner's l4
              uint2 mine = lh ? __builtin_bit_cast(uint2, h4) : __builtin_bit_cast(uint2, l4);  // what the partner wants
              uint2 got;
              got.x = (unsigned)__shfl_xor((int)mine.x, 32, 64);
              got.y = (unsigned)__shfl_xor((int)mine.y, 32, 64);
              if (col < g.Co) {
                const uint2 keep = lh ? __builtin_bit_cast(uint2, l4) : __builtin_bit_cast(uint2, h4);
                const u32x4 out8 = lh ? u32x4{got.x, got.y, keep.x, keep.y} : u32x4{keep.x, keep.y, got.x, got.y};
                // (planes are CHUNK-major, [n][pixel / 16][co][16]: the quadratic-form kernel stages 16 positions of 32 rows per
                //  request — one contiguous kilobyte this way, 32 pieces of 32 bytes at stride 2 HW from [n][co][pixel])
                const int p0 = pix - 4 * lh;  // (the run starts at the lh == 0 lane's pixel)
                const int64_t e = (((int64_t)n * (HW >> 4) + (p0 >> 4)) * g.Co + col) * 16 + (p0 & 15);
                *reinterpret_cast<u32x4*>((lh ? fz.out_l : fz.out_h) + e) = out8;
              }
            } else if (col < g.Co) {
              const int64_t e = (((int64_t)n * (HW >> 4) + (pix >> 4)) * g.Co + col) * 16 + (pix & 15);
              *reinterpret_cast<f16x4*>(fz.out_h + e) = h4;
              *reinterpret_cast<f16x4*>(fz.out_l + e) = l4;
            }
          }
          continue;
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const int col = tile_n * BN + (wn * TN + b) * 32 + lr;
          if (col >= g.Co) continue;
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = acc[a][b][4 * q + j] * inv_an * inv_w;
            vmax = max(vmax, __float_as_uint(v[j]) & 0x7fffffffu);
          }
          *reinterpret_cast<f32x4*>(out + ((int64_t)n * g.Co + col) * HW + pix) = v;
        }
      }
  } else
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int m = tile_m * BM + row;
      if (m >= M) continue;
      int64_t opix = m;
      int n = 0;
      if (g.pmajor || !g.dense) {
        int rem;
        if (g.pmajor) rem = fdiv(m, g.div_n), n = m - rem * g.N;
        else n = fdiv(m, g.div_hw), rem = m - n * (g.Hc * g.Wc);
        const int ci_ = fdiv(rem, g.div_w);
        opix = ((int64_t)n * g.Ho + ci_ * g.os + g.oh0) * g.Wo + (rem - ci_ * g.Wc) * g.os + g.ow0;
      } else if (per_img) {
        n = fdiv(m, g.div_hw);
      }
      const float inv_an = per_img ? image_inv(n) : inv_a;  // one scale per image: un-scaled row by row
      float* orow = out + opix * g.Co;
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int col = tile_n * BN + (wn * TN + b) * 32 + lr;
        if (col < g.Co) {
          float v = acc[a][b][r] * inv_an * inv_w;
          if (accumulate) v += orow[col];
          orow[col] = v;
          vmax = max(vmax, __float_as_uint(v) & 0x7fffffffu);
        }
      }
    }
  }
  if (amax_out) conv_block_max<NT / 64>(vmax, 1.f, amax_out, smem);
}


// ---------------------------------------------------------------------------------------------------------------------
// Strided form: the backward-data of a stride-s convolution with ALL residue classes of the input-gradient pixels in ONE
// launch (class c = (oh0, ow0): dX[i*s + oh0, j*s + ow0] = sum over the taps of that class of g[i + dh, j + dw] W[tap]),
// optionally together with a SECOND convolution that feeds the same input — the 1 x 1 shortcut of a residual
// down-sampling block, its own cotangent and weights — and with the element-wise VJP fused into the epilogue, as in
// conv_f16x2_kernel<CFG, true>.  What it replaces in a sweep over ResNet-18 (per down-sampling block): five launches of
// 55 - 160 us that each fill the chip for ONE round (four classes of 1 / 2 / 2 / 4 taps and the shortcut, fp32 output
// written with a stride, the shortcut's class read back and rewritten) plus the element-wise VJP kernel over the result.
//   rows: (class-tile, class) interleaved, i.e. the four classes of the same cotangent pixels run next to each other and
//   share the cotangent rows in L2; within a class the row order is (image, i, j) as in the generic kernel
//   second source: its taps come first in the K loop; the two sources carry different fixed-point units
//   (2^(sexp_a + sexp_w) each), so the accumulators are multiplied by the power of two between them once (exact) when the
//   K loop passes from the second source's taps to the first's.
struct StridedGeom {
  int N, Hi, Wi, Ci;   // cotangent maps [N][Hi][Wi][Ci] of both sources (Ci = GEMM K per tap)
  int Hc, Wc;          // class grid, the same for every class (Ho % os == 0, Wo % os == 0)
  int Ho, Wo, Co, os;  // output tensor [N][Ho][Wo][Co]
  int ncls;
  int oh0[4], ow0[4];
  unsigned cls_taps[4];  // bit t: tap t belongs to the class
  unsigned second;       // bit t: tap t reads the second source (these taps are listed first)
  int T;
  int dh[12], dw[12], wt[12];
  FastDiv div_hw, div_w, div_cls;
};

struct StridedSrc {
  const _Float16 *ah, *al, *wh, *wl;  // cotangent planes, weight planes [slice][Co][Ci]
  const int *a_sexp, *w_sexp;
  const unsigned* in_amax;            // measured max|cotangent| or NULL
  const float* w_l1;
};

template <typename CFG>
__global__ __launch_bounds__(CFG::NT) __attribute__((amdgpu_waves_per_eu(CFG::WPE, CFG::WPE))) void conv_strided_f16x2_kernel(
    const StridedGeom g, const StridedSrc s1, const StridedSrc s2, const _Float16* __restrict__ zero16,
    unsigned* __restrict__ amax_out, int nb_m, const ConvVjp fz) {
  constexpr int BM = CFG::BM, BN = CFG::BN, BK = CFG::BK, Q = CFG::Q, TM = CFG::TM, TN = CFG::TN, NT = CFG::NT;
  static_assert(CFG::FUSABLE, "the strided form always runs the VJP epilogue");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {  // XCD-contiguous tile ranges (see conv_f16x2_kernel)
    const int q = nblk / 8, r = nblk % 8, x = bid % 8, j = bid / 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
  }
  const int M = g.N * g.Hc * g.Wc;  // rows of ONE class
  const int KC = g.Ci / BK;
  const int nb_mc = nb_m * g.ncls;
  const int tile_n = bid / nb_mc, tmc = bid - tile_n * nb_mc;
  const int tile_m = fdiv(tmc, g.div_cls), cls = tmc - tile_m * g.ncls;
  const int oh0 = g.oh0[cls], ow0 = g.ow0[cls];
  const unsigned ctaps = g.cls_taps[cls];
  const int HWc = g.Hc * g.Wc;

  int64_t a_off[CFG::A_LD / 2];
  unsigned a_valid[CFG::A_LD / 2];
  int a_lq[CFG::A_LD / 2];
#pragma unroll
  for (int i = 0; i < CFG::A_LD / 2; ++i) {
    const int slot = i * NT + tid;
    const int row = slot / Q, pq = slot % Q;
    a_lq[i] = pq ^ swz<Q>(row);
    const int m = tile_m * BM + row;
    unsigned valid = 0;
    int64_t off = 0;
    if (m < M) {
      const int n = fdiv(m, g.div_hw), rem = m - n * HWc;
      const int ih = fdiv(rem, g.div_w), iw = rem - ih * g.Wc;
      off = (((int64_t)n * g.Hi + ih) * g.Wi + iw) * g.Ci;
      for (int t = 0; t < g.T; ++t) {
        const int hh = ih + g.dh[t], ww = iw + g.dw[t];
        if (((ctaps >> t) & 1u) && hh >= 0 && hh < g.Hi && ww >= 0 && ww < g.Wi) valid |= 1u << t;
      }
    }
    a_off[i] = off, a_valid[i] = valid;
  }
  int64_t b_off[CFG::B_LD];
  bool b_ok[CFG::B_LD];
  bool b_low[CFG::B_LD];
#pragma unroll
  for (int i = 0; i < CFG::B_LD; ++i) {
    const int cs = i * NT + tid;
    const int slot = cs % CFG::B_SLOTS;
    const int row = slot / Q, pq = slot % Q;
    const int n = tile_n * BN + row;
    b_low[i] = cs >= CFG::B_SLOTS;
    b_ok[i] = n < g.Co;
    b_off[i] = (int64_t)n * g.Ci + (pq ^ swz<Q>(row)) * 8;
  }
  const int64_t w_tap = (int64_t)g.Co * g.Ci;

  auto clampe = [](int e) { return e < -126 ? -126 : (e > 127 ? 127 : e); };
  const int sexp_a = s1.a_sexp[0], sexp_w = s1.w_sexp[0];
  const float inv = exp2i(clampe(-sexp_a)) * exp2i(clampe(-sexp_w));
  float ratio = 1.f, inv2 = 0.f;
  int so;
  {
    float bound = (s1.in_amax ? __uint_as_float(s1.in_amax[0]) : exp2i(clampe(15 - sexp_a))) * s1.w_l1[0];
    if (g.second) {
      const int sa2 = s2.a_sexp[0], sw2 = s2.w_sexp[0];
      bound += (s2.in_amax ? __uint_as_float(s2.in_amax[0]) : exp2i(clampe(15 - sa2))) * s2.w_l1[0];
      ratio = exp2i(clampe((sexp_a + sexp_w) - (sa2 + sw2)));
    }
    if (fz.add_h) {
      const int sadd = fz.add_sexp[0];
      bound += exp2i(clampe(15 - sadd));
      inv2 = exp2i(clampe(-sadd));
    }
    if (fz.mask && fz.mask_float && fz.mult_amax) bound *= __uint_as_float(fz.mult_amax[0]);
    if (fz.scale) bound *= __uint_as_float(fz.scale_amax[0]);
    so = scale_exp_for(bound);
  }
  const float sc_out = exp2i(so);

  // taps of this class that reach a row of this tile (uniform over the workgroup), the second source's first
  unsigned long long tap_list = 0;
  int ntap = 0, n_first = 0;
  {
    unsigned v = 0;
#pragma unroll
    for (int i = 0; i < CFG::A_LD / 2; ++i) v |= a_valid[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v |= (unsigned)__shfl_xor((int)v, off, 64);
    unsigned* red = reinterpret_cast<unsigned*>(smem);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    v = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) v |= red[w];
    v = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    __syncthreads();
    for (int t = 0; t < g.T; ++t)
      if ((v >> t) & 1u) {
        tap_list |= (unsigned long long)t << (4 * ntap), ++ntap;
        if ((g.second >> t) & 1u) ++n_first;
      }
  }
  const int nstage = ntap * KC;
  const int s_switch = n_first * KC;  // first stage of the first source's taps

  auto stage = [&](int s, int buf) {
    const int tj = s / KC, kc = s - tj * KC;
    const int t = (int)((tap_list >> (4 * tj)) & 15ull);
    const bool sec = (g.second >> t) & 1u;
    const _Float16* Ah = sec ? s2.ah : s1.ah;
    const _Float16* Al = sec ? s2.al : s1.al;
    const _Float16* Wh = sec ? s2.wh : s1.wh;
    const _Float16* Wl = sec ? s2.wl : s1.wl;
    char* base = smem + buf * CFG::STAGE;
    const int64_t tap_off = ((int64_t)g.dh[t] * g.Wi + g.dw[t]) * g.Ci + kc * BK;
#pragma unroll
    for (int i = 0; i < CFG::A_LD / 2; ++i) {
      const bool ok = (a_valid[i] >> t) & 1u;
      const int64_t e = a_off[i] + tap_off + a_lq[i] * 8;
      const _Float16* sh = ok ? Ah + e : zero16;
      const _Float16* sl = ok ? Al + e : zero16;
      char* dh_ = base + (i * NT + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds((gbl_void*)sh, (lds_void*)dh_, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void*)sl, (lds_void*)(dh_ + CFG::A_PLANE), 16, 0, 0);
    }
    const int64_t wbase = (int64_t)g.wt[t] * w_tap + kc * BK;
#pragma unroll
    for (int i = 0; i < CFG::B_LD; ++i) {
      const int64_t e = wbase + b_off[i];
      const _Float16* sp = b_ok[i] ? (b_low[i] ? Wl : Wh) + e : zero16;
      char* db = base + 2 * CFG::A_PLANE + (i * NT + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds((gbl_void*)sp, (lds_void*)db, 16, 0, 0);
    }
  };

  const int wm = wave / CFG::WN, wn = wave % CFG::WN;
  const int lr = lane & 31, lh = lane >> 5;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  auto rescale = [&]() {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] *= ratio;
  };

  // ---- epilogue context (see conv_f16x2_kernel: the addend planes / mask bytes of all of a lane's chunks are requested
  // right behind the K loop)
  constexpr int ROWS_W = TM * 32, COLS_W = TN * 32, PITCH = COLS_W + 4, C8 = COLS_W / 8, NIT = ROWS_W * C8 / 64;
  int64_t e_[NIT];
  int pix_[NIT];
  bool ok_[NIT];
  f16x8 h2_[NIT], l2_[NIT];
  uint2 mk_[NIT];
  const bool pre_mask = fz.mask && !fz.mask_float;
  const int mrows = (int)fz.mask_rows;
  auto mask_row = [&](int p) { return p - fdiv(p, fz.div_mask) * mrows; };
  auto prefetch = [&](int it) {
    const int idx = it * 64 + lane;
    const int row = idx / C8, c8 = idx - row * C8;
    const int m = tile_m * BM + wm * ROWS_W + row;
    const int col0 = tile_n * BN + wn * COLS_W + c8 * 8;
    ok_[it] = m < M && col0 < g.Co;
    int opix = 0;
    if (ok_[it]) {
      const int n = fdiv(m, g.div_hw), rem = m - n * HWc;
      const int ci_ = fdiv(rem, g.div_w);
      opix = (n * g.Ho + ci_ * g.os + oh0) * g.Wo + (rem - ci_ * g.Wc) * g.os + ow0;
    }
    e_[it] = (int64_t)opix * g.Co + col0;
    pix_[it] = opix;
#pragma unroll
    for (int j = 0; j < 8; ++j) h2_[it][j] = (_Float16)0.f, l2_[it][j] = (_Float16)0.f;
    mk_[it] = make_uint2(0x01010101u, 0x01010101u);
    if (fz.add_h && ok_[it]) {
      h2_[it] = *reinterpret_cast<const f16x8*>(fz.add_h + e_[it]);
      l2_[it] = *reinterpret_cast<const f16x8*>(fz.add_l + e_[it]);
    }
    if (pre_mask && ok_[it])
      mk_[it] = *reinterpret_cast<const uint2*>((const unsigned char*)fz.mask + (int64_t)mask_row(opix) * g.Co + col0);
  };

  {
    constexpr int LD_PER_STAGE = CFG::A_LD + CFG::B_LD;
    static_assert(CFG::NBUF == 2, "two LDS stages");
    (void)LD_PER_STAGE;
    if (nstage > 0) stage(0, 0);
    int buf = 0;
    for (int s = 0; s < nstage; ++s) {
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): stage s has landed (one stage in flight)
      __builtin_amdgcn_s_barrier();
      if (s + 1 < nstage) stage(s + 1, buf ^ 1);
      if (s == s_switch && n_first) rescale();
      const char* base = smem + buf * CFG::STAGE;
      buf ^= 1;
      f16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
      auto load_frags = [&](int k16, int set) {
#pragma unroll
        for (int a = 0; a < TM; ++a) {
          const int row = (wm * TM + a) * 32 + lr;
          const int off = (row * Q + ((k16 * 2 + lh) ^ swz<Q>(row))) * 16;
          ah[set][a] = *reinterpret_cast<const f16x8*>(base + off);
          al[set][a] = *reinterpret_cast<const f16x8*>(base + CFG::A_PLANE + off);
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const int row = (wn * TN + b) * 32 + lr;
          const int off = (row * Q + ((k16 * 2 + lh) ^ swz<Q>(row))) * 16;
          bh[set][b] = *reinterpret_cast<const f16x8*>(base + 2 * CFG::A_PLANE + off);
          bl[set][b] = *reinterpret_cast<const f16x8*>(base + 2 * CFG::A_PLANE + CFG::B_PLANE + off);
        }
      };
      load_frags(0, 0);
#pragma unroll
      for (int k16 = 0; k16 < BK / 16; ++k16) {
        const int set = k16 & 1;
        if (k16 + 1 < BK / 16) load_frags(k16 + 1, set ^ 1);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            f32x16 c = acc[a][b];
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[set][a], bh[set][b], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[set][a], bl[set][b], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[set][a], bh[set][b], c, 0, 0, 0);
            acc[a][b] = c;
          }
      }
    }
    if (n_first && s_switch == nstage) rescale();  // (a tile that only the second source reaches)
  }

  unsigned vmax = 0;
  if (blockIdx.x == 0 && tid == 0) fz.out_sexp[0] = so;
  static_assert(CFG::EPI_LDS == CFG::WM * CFG::WN * ROWS_W * PITCH * 4 && CFG::EPI_LDS <= 80 * 1024, "staging image: two workgroups per CU");
#pragma unroll
  for (int it = 0; it < NIT; ++it) prefetch(it);
  __syncthreads();
  float* img = reinterpret_cast<float*>(smem) + wave * (ROWS_W * PITCH);
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        img[(a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * PITCH + b * 32 + lr] = acc[a][b][r] * inv;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = it * 64 + lane;
    const int row = idx / C8, c8 = idx - row * C8;
    if (!ok_[it]) continue;
    const int col0 = tile_n * BN + wn * COLS_W + c8 * 8;
    const f32x4 p0 = *reinterpret_cast<const f32x4*>(img + row * PITCH + c8 * 8);
    const f32x4 p1 = *reinterpret_cast<const f32x4*>(img + row * PITCH + c8 * 8 + 4);
    float v[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
    const int64_t e = e_[it];
    if (fz.add_h) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += ((float)h2_[it][j] + (float)l2_[it][j]) * inv2;
    }
    float mult[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) mult[j] = sc_out;
    if (fz.mask) {
      if (fz.mask_float) {
        const int64_t em = (int64_t)mask_row(pix_[it]) * g.Co + col0;
        const f32x4 a = *reinterpret_cast<const f32x4*>((const float*)fz.mask + em);
        const f32x4 b = *reinterpret_cast<const f32x4*>((const float*)fz.mask + em + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) mult[j] *= a[j], mult[4 + j] *= b[j];
      } else {
        const uint2 u = mk_[it];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (!((u.x >> (8 * j)) & 0xffu)) mult[j] = 0.f;
          if (!((u.y >> (8 * j)) & 0xffu)) mult[4 + j] = 0.f;
        }
      }
    }
    if (fz.scale) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(fz.scale + col0), b = *reinterpret_cast<const f32x4*>(fz.scale + col0 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) mult[j] *= a[j], mult[4 + j] *= b[j];
    }
    f16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float xs = v[j] * mult[j];
      asm volatile("" : "+v"(xs));
      const _Float16 hh = (_Float16)xs;
      h[j] = hh;
      l[j] = (_Float16)(xs - (float)hh);
      vmax = max(vmax, __float_as_uint(xs) & 0x7fffffffu);
    }
    *reinterpret_cast<f16x8*>(fz.out_h + e) = h;
    *reinterpret_cast<f16x8*>(fz.out_l + e) = l;
  }
  if (amax_out) conv_block_max<NT / 64>(vmax, exp2i(-so < -126 ? -126 : -so), amax_out, smem);
}


__device__ __forceinline__ int swz2(int row) { return (row >> 3) & 1; }  // two 16-byte slots per row (see swz<2>)

// ---- persistent window form ------------------------------------------------------------------------------------------
// Round 3's measurements (profiles/r03_win_ablate.txt) say what bounds a fused 64-channel launch: not the matrix
// pipe (104 us at the nominal rate, ~165 at the clock the chip sustains) and not its traffic (172 us of HBM time), but
//   (1) the fused epilogue — HBM time, 120 us — that nothing overlaps,
//   (2) ~20 us per workgroup generation: prologue, the first loads' latency in the open, barriers draining at the end,
//   (3) every load of a step waited for with vmcnt(0): the window of the NEXT chunk (an HBM miss) held up every hand-over.
// This kernel is the window form (input window of a pixel tile resident in LDS, conv_win_f16x2_kernel) restructured
// against exactly those: TWO four-wave workgroups per CU stay for the whole launch and walk through pixel tiles tile,
// tile + G, ...; the stream of K steps runs across tile boundaries (the next tile's first window and weights are requested
// during the current tile's last chunk and land under its epilogue); the window is waited for only where it is read; and
// the two workgroups of a CU drift into different phases, so that one's epilogue (memory) runs beside the other's K loop
// (matrix pipe).  (A one-workgroup form with the epilogue interleaved into the next tile's K steps was built first: two
// accumulator sets do not fit 256 registers, and hipcc spilled the second one; an eight-wave workgroup on a 512-pixel tile
// with the epilogue in the open — WinPCfg<512>, config bit 26 — halves the weight requests per MFMA and is 5 - 30 % SLOWER.)
// Round 6 (profiles/r06_winp_study.md): the K loop is one explicit instruction stream (`step`); what the launch is bound by
// is the chip's power budget — back to back these launches hold the socket at 1.28 - 1.40 kW of 1.4 and the clock at 1.78 -
// 1.85 of 2.4 GHz (profiles/r06_power_kernels.log); with only its MFMAs left (no requests, reads, waits) the same launch
// takes 77 % of its time.
// Output channels = 64 (the 64-channel layers: the HBM-bound ones); nine taps in raster order (the host sorts them), weight
// slice of tap t = wt0 + t * wtstep; multipliers: none or a byte mask.
template <int BM_, int WPE_ = 2>
struct WinPCfg {
  static constexpr int BM = BM_, BN = 64, NW = BM / 64, NT = 64 * NW, WPE = WPE_;
  static constexpr int CK = 16;                          // channels per window chunk = one k16 step
  static constexpr int PP = BM + 96;                     // window pixels: BM + 2 Wi + 2 <= PP (Wi <= 47)
  static constexpr int W_PLANE = PP * CK * 2, WIN = 2 * W_PLANE;
  static constexpr int B_TAP = BN * CK * 2, B_STEP = 6 * B_TAP;  // three taps, h + l
  static constexpr int W_INSTR = 4 * PP / 64, W_IT = (W_INSTR + NW - 1) / NW;  // LDS-DMA wave-instructions per window / per wave
  static constexpr int B_INSTR = 12, B_IT = (B_INSTR + NW - 1) / NW;           // ... per weight step
  static constexpr int EPI_OFF = 2 * WIN + 2 * B_STEP;
  static constexpr int EPI_PITCH = 68, EPI_WAVE = 8 * EPI_PITCH * 4;  // per-wave image of one 8-row slice
  static constexpr int ZERO_OFF = EPI_OFF + NW * EPI_WAVE;            // 32 zero bytes: what out-of-image taps read
  static constexpr int LDS = ZERO_OFF + 32;
  static constexpr int NSLICE = 8;                       // 64 rows per wave = 8 slices of 8 rows
  static_assert((2 * PP) % 64 == 0, "a window plane is a whole number of wave-instructions");
  static constexpr int WG_PER_CU = BM_ <= 256 ? 2 : 1;  // (512 rows: eight waves = the CU's two waves per SIMD in ONE workgroup)
  static_assert(WG_PER_CU * LDS <= 160 * 1024, "workgroups per CU");
};

struct WinPArgs {  // (a slim argument block: everything here stays in scalar registers for the whole launch)
  int M, Hi, Wi, Ci, Co, HW, n_tiles, nb_m, wt0, wtstep, stagger, halo_all, coloc;
  FastDiv div_hw, div_w, div_mask;
  const _Float16 *Ah, *Al, *Wh, *Wl;
  const int *a_sexp, *w_sexp, *add_sexp;
  const unsigned *in_amax, *scale_amax;
  const float *w_l1, *scale;
  const _Float16 *add_h, *add_l;
  const unsigned char* mask;
  int mask_rows;
  _Float16 *out_h, *out_l;
  int* out_sexp;
  unsigned* amax_out;
  // split tail (see `item` in the kernel): walk indices from split_v0 on are K slices of the split_L leftover tiles
  int split_S, split_L, split_v0;
  float* slabs;    // [split_L * split_S][16][NT][4] partial accumulators
  int* arrivals;   // [split_L] arrival counters, zero between launches
};

#ifdef LK_WINP_TRACE
// (development build, -DLK_WINP_TRACE) per workgroup and tile: s_memtime at the start of the K loop, at its end, at the end
// of the epilogue; slot 15 of a workgroup: its hardware id (XCC / SE / CU)
__device__ unsigned long long g_winp_trace[1024 * 16 * 3];
#endif

template <typename CFG>
__global__ __launch_bounds__(CFG::NT) __attribute__((amdgpu_waves_per_eu(CFG::WPE, CFG::WPE)))
void conv_winp_f16x2_kernel(const WinPArgs p) {
  constexpr int BM = CFG::BM, NW = CFG::NW, PP = CFG::PP, TM = 2, TN = 2, PITCH = CFG::EPI_PITCH, NS = CFG::NSLICE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int M = p.M, KC = p.Ci / 16, Wi = p.Wi;
  const int G = gridDim.x;
  // XCD-aware tile order for 64 output channels: consecutive block ids run on different XCDs (id % 8); every XCD gets a
  // contiguous range of the pixel tiles in flight together, so that neighbours (shared halo rows) meet in ONE L2 (334 ->
  // 321 us on the c4 launch).  With several 64-channel column tiles per pixel tile the same order puts all the readers of
  // one window into one L2 at the same moment and LOSES (256 channels: 264 -> 314 us, 512: 289 -> 385): they stay spread.
  int tile = blockIdx.x;
  if (p.Co == 64) {
    const int q = G / 8, r = G % 8, x = tile % 8, j = tile / 8;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
  }
  // tile index -> (first pixel, first output channel); false behind the last tile of this workgroup's walk.  Plain order:
  // the column tiles of a pixel tile are consecutive indices, i.e. run at the same time on DIFFERENT XCDs (index % 8), each
  // XCD always on the same column(s): its weight slice stays hot in its L2, every window is fetched by Co / 64 L2s.
  // coloc = c > 1 (host: config bits 28-29): an XCD walks c columns of every pixel tile it owns side by side — a window is
  // fetched by Co / 64 / c L2s, an L2 holds c weight slices.  x = index % 8 is the XCD (the grid is a multiple of 8).
  const int nb_n = p.Co / 64;
  auto coords = [&](int v, int& m0_, int& n0_) {
    if (p.coloc > 1) {
      const int x = v & 7, j = v >> 3, ng = nb_n / p.coloc, per = 8 / ng;
      const int pt = (j / p.coloc) * per + x / ng;
      m0_ = pt * BM, n0_ = ((x % ng) * p.coloc + j % p.coloc) * 64;
      return pt < p.nb_m;
    }
    m0_ = (v / nb_n) * BM, n0_ = (v % nb_n) * 64;
    return v < p.n_tiles;
  };
  // The walk's last round is rarely full: 1152 images at 512 channels are 576 tiles for 512 workgroups — 64 of them would
  // run a second tile while 448 idle, and the launch lasts two tiles for 1.125 tiles of work (measured: workgroups end at
  // 140 .. 270 us).  The host therefore splits the split_L leftover tiles of that round along K into split_S slices each
  // (split_L * split_S <= grid): an item of that round is (tile, chunk range), its accumulators go to a slab, the slice that
  // arrives last at the tile's counter adds the slabs IN SLICE ORDER (a fixed order: the result does not depend on who
  // arrives last) and runs the fused epilogue.  Chunk ranges are even in start and length (the LDS buffers' parity).
  // (an item's chunk range and slab index travel as ONE word — first chunk | end chunk << 8 | (slab + 1) << 16 — : the kernel
  //  sits at the register limit, and values carried around the persistent loop are booked as vector registers)
  auto item = [&](int v, int& m0_, int& n0_, int& desc_) {
    desc_ = KC << 8;
    if (p.split_S > 1 && v >= p.split_v0) {
      const int j = v - p.split_v0;
      if (j >= p.split_L * p.split_S) return false;
      // (uniform values, but integer division runs in the vector unit: back into scalar registers for the staging bases)
      const int tl = __builtin_amdgcn_readfirstlane(j / p.split_S), len = __builtin_amdgcn_readfirstlane(KC / p.split_S);
      const int sl = j - tl * p.split_S;
      desc_ = (sl * len) | ((sl * len + len) << 8) | ((j + 1) << 16);
      return coords(p.split_v0 + tl, m0_, n0_);
    }
    return coords(v, m0_, n0_);
  };
  int desc = 0, next_desc = 0, kc1 = KC, next_kc0 = 0;
  int m0, n0;
  if (!item(tile, m0, n0, desc)) return;  // (uniform)
#ifdef LK_WINP_ABLATE  // development switch, compile-time: 1 skip the epilogue (timing only).  (Round 6's finer ablations — no
  constexpr int ablate = LK_WINP_ABLATE;  // MFMAs / requests / waits / fragment reads / barriers — lived in the round-5 form of `step`)
#else
  constexpr int ablate = 0;
#endif

  // a zero block behind everything the LDS-DMA writes: what out-of-image taps read (every lane of every staging
  // instruction then reads mapped memory and lands somewhere harmless: no pointer selects, no partial instructions);
  // the folded BatchNorm scale of the epilogue behind it
  if (tid < 8) reinterpret_cast<unsigned*>(smem + CFG::ZERO_OFF)[tid] = 0u;
  __syncthreads();  // (the hand-over barriers of the K loop are raw s_barriers: they do not order this LDS write)

  // ---- scales (see conv_f16x2_kernel): the result's scale from the guaranteed bound
  const int sexp_a = p.a_sexp[0], sexp_w = p.w_sexp[0];
  const float inv = exp2i(-sexp_a < -126 ? -126 : -sexp_a) * exp2i(-sexp_w < -126 ? -126 : -sexp_w);
  float inv2 = 0.f;
  float bound = p.in_amax ? __uint_as_float(p.in_amax[0]) : exp2i(15 - sexp_a < -126 ? -126 : (15 - sexp_a > 127 ? 127 : 15 - sexp_a));
  bound *= p.w_l1[0];
  if (p.add_h) {
    const int s2 = p.add_sexp[0];
    bound += exp2i(15 - s2 < -126 ? -126 : (15 - s2 > 127 ? 127 : 15 - s2));
    inv2 = exp2i(-s2 < -126 ? -126 : -s2);
  }
  if (p.scale) bound *= __uint_as_float(p.scale_amax[0]);
  const int so = scale_exp_for(bound);
  const float sc_out = exp2i(so);
  if (blockIdx.x == 0 && tid == 0) p.out_sexp[0] = so;

  // ---- window staging: wave-instruction i = it * NW + wave covers slots [64 i, 64 i + 64) of [plane][pixel][2].
  //      Addresses are (scalar base) + (32-bit lane offset): the lane offsets are the only registers the staging keeps.
  //      (W_INSTR need not divide by the waves: a wave whose last instruction does not exist repeats its first one —
  //      same bytes to the same place — so that every wave issues W_IT instructions and the counted waits hold)
  unsigned w_off[CFG::W_IT];  // byte offset of this lane's source slot inside a plane, for the tile that is being staged
  auto w_instr = [&](int it) {
    const int i = it * NW + wave;
    return i < CFG::W_INSTR ? i : wave;
  };
  auto setup_window = [&](int m0) {
    // What of the PP-pixel window is ever READ: the tile's own BM pixels, the Wi + 1 above them unless the tile starts at an
    // image boundary, the Wi + 1 below unless it ends at one (taps never leave the image of their row: they read the zero
    // block instead) — and nothing of the padding behind 2 Wi + 2 + BM.  The staging instructions are a fixed count (the
    // counted waits rely on it), so the slots nobody reads are pointed at the nearest pixel that IS read: same cache lines
    // as their neighbours' requests, no traffic of their own.  Maps of up to 16 x 16 (a tile = whole images) fetch BM
    // pixels instead of 352; the padding alone was 9 % (32 x 32) to 24 % (4 x 4) of every window fetched.
    const int r0 = m0 - fdiv(m0, p.div_hw) * p.HW;
    const int e0 = m0 + BM >= M ? 0 : (m0 + BM) - fdiv(m0 + BM, p.div_hw) * p.HW;
    const int px_lo = p.halo_all ? 0 : (r0 == 0 ? Wi + 1 : 0);
    const int px_hi = (p.halo_all ? PP : (e0 == 0 ? BM + Wi + 1 : BM + 2 * Wi + 2)) - 1;
#pragma unroll
    for (int it = 0; it < CFG::W_IT; ++it) {
      const int i = w_instr(it);
      const int rem = (i * 64) % (2 * PP) + lane;  // (whole instructions lie inside one plane: 2 PP % 64 == 0)
      const int px = rem >> 1, pq = rem & 1;
      const int pxs = px < px_lo ? px_lo : (px > px_hi ? px_hi : px);  // (source pixel; the LDS slot stays px's)
      int raster = m0 - (Wi + 1) + pxs;
      // rows outside the tensor (and the window's padding) are only reached by taps that read the zero block instead: any
      // mapped address will do for them
      raster = raster < 0 ? 0 : (raster > M - 1 ? M - 1 : raster);
      w_off[it] = (unsigned)(raster * p.Ci + ((pq ^ swz2(px)) << 3)) * 2u;
    }
  };
  auto stage_win = [&](int kc, int buf) {
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + buf * CFG::WIN;
#pragma unroll
    for (int it = 0; it < CFG::W_IT; ++it) {
      const int i = w_instr(it);  // (scalar)
      const char* sb = reinterpret_cast<const char*>((i * 64 >= 2 * PP) ? p.Al : p.Ah) + kc * 32;  // (scalar)
      __builtin_amdgcn_global_load_lds((gbl_void*)(sb + w_off[it]), (lds_void*)(uintptr_t)(base + i * 1024), 16, 0, 0);
    }
  };
  // ---- weight staging: slots [tap j of the step][plane][n][2]; instruction i covers 32 output channels of one (tap, plane).
  //      The weights come CHUNK-major, [plane][tap][Ci / 16][64][16]: the 32 rows of an instruction are one contiguous
  //      kilobyte = 8 cache lines.  From the GEMM-natural [tap][64][Ci] layout the same instruction touches 32 lines for 32
  //      bytes each, and the requests — not the bytes — are what the CU's vector memory path runs out of: staging cost
  //      136 of 400 us with every lane of every instruction asking for a quarter of its line.
  const int w_tap_bytes = p.Co * p.Ci * 2;  // (one Co x Ci weight slice of a plane)
  unsigned b_off[CFG::B_IT];
#pragma unroll
  for (int j = 0; j < CFG::B_IT; ++j) {
    const int i = (wave * CFG::B_IT + j) % CFG::B_INSTR;
    const int sidx = i * 64 + lane;
    const int pq = sidx & 1, n = (sidx >> 1) & 63;
    b_off[j] = (unsigned)(n * 16 + ((pq ^ swz2(n)) << 3)) * 2u;
  }
  auto stage_b = [&](int kc, int r, int slot, int bn0) {  // r: compile-time after unrolling; bn0: first output channel of the tile
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + 2 * CFG::WIN + slot * CFG::B_STEP;
#pragma unroll
    for (int j = 0; j < CFG::B_IT; ++j) {
      const int i = (wave * CFG::B_IT + j) % CFG::B_INSTR;  // (scalar)
      const int jp = i >> 1;                                 // (tap of the step, plane)
      const int t = r * 3 + (jp >> 1);
      const char* sb = reinterpret_cast<const char*>((jp & 1) ? p.Wl : p.Wh) + (int64_t)(p.wt0 + t * p.wtstep) * w_tap_bytes +
                       (kc * p.Co + bn0) * 32;
      __builtin_amdgcn_global_load_lds((gbl_void*)(sb + b_off[j]), (lds_void*)(uintptr_t)(base + i * 1024), 16, 0, 0);
    }
  };

  // ---- fragment addresses: a validity bit per (row tile, tap); the bits depend on the pixel (h, w) of a row only —
  //      constant over this workgroup's tiles when its tile stride is a whole number of images and no tile is ragged
  unsigned a_valid[TM];  // bit t: tap t of this lane's row (row tile a) lies inside the image
  const int a_px0 = wave * (TM * 32) + lr;  // window pixel of row tile 0 at shift 0 (row tile a: + 32 a)
  auto setup_frag = [&](int m0) {
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int m = m0 + (wave * TM + a) * 32 + lr;
      int h = 0, w = 0;
      const bool in = m < M;
      if (in) {
        const int rem = m - fdiv(m, p.div_hw) * p.HW;
        h = fdiv(rem, p.div_w), w = rem - h * Wi;
      }
      unsigned v = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
        if (in && hh >= 0 && hh < p.Hi && ww >= 0 && ww < Wi) v |= 1u << t;
      }
      a_valid[a] = v;
    }
  };
  const bool frag_const = G % nb_n == 0 && ((int64_t)(G / nb_n) * BM) % p.HW == 0 && (int64_t)p.nb_m * BM <= M;
  int b_addr[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int n = b * 32 + lr;
    b_addr[b] = n * 32 + (lh ^ swz2(n)) * 16;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- the K loop as ONE explicit instruction stream.  A step = three taps x 16 channels = 36 MFMAs per wave, 32 cycles of
  //      the matrix pipe each, and a wave issues in order: whatever else it has to issue in a step — 24 fragment reads with
  //      their address selects, its share of the LDS-DMA requests for later steps — sits in the gaps BETWEEN the MFMAs, one
  //      item per gap, pinned by sched_barrier (blocks of reads / requests in front of blocks of MFMAs left the pipe idle for
  //      their issue time: a workgroup alone on its CU ran at half the pipe's rate, profiles/r06_winp_alone.md).  Per step:
  //        T0: tap 0 (fragments requested during the PREVIOUS step's T2)    gaps: reads of tap 1
  //        T1: tap 1                                                         gaps: reads of tap 2
  //        hand-over: own reads of this step done, vmcnt, s_barrier — the next step's operands have landed, and nobody reads
  //                   this step's weight slot any more
  //        T2: tap 2                                                         gaps: reads of the NEXT step's tap 0, the
  //                   requests for the weights of the step after next (into the slot just released), the next chunk's window
  //      The fragment sets alternate per MFMA block, i.e. per step with three blocks: set of a step's tap 0 = (kc + r) & 1 =
  //      its weight slot; the loop is unrolled over two chunks so that both are compile-time.  A tile = (pixel tile, 64 output
  //      channels).  The requests run one uniform stream across tile boundaries (the next tile's first window and first two
  //      weight steps are requested during this tile's last chunk and land under its epilogue).
  int next_m0 = -1, next_n0 = 0;
  bool fresh = false;  // the next step follows an epilogue, which has drained this wave's loads itself (see there)
  f16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
  static_assert(TM == 2 && TN == 2, "wait_set names eight registers");
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto lds_read = [&](f16x8& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr)); };
  // (the reads are asm and waited for by COUNT: hipcc books the LDS-DMA instructions between them as possible LDS traffic and
  //  would wait lgkmcnt(0) in front of every MFMA; wait_set ties the set's registers to the wait)
  auto wait_set = [&](int set) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(ah[set][0]), "+v"(ah[set][1]), "+v"(al[set][0]), "+v"(al[set][1]), "+v"(bh[set][0]), "+v"(bh[set][1]),
                   "+v"(bl[set][0]), "+v"(bl[set][1]));
  };
  // one fragment read: u = 0 - 3 window (row tile u >> 1, plane u & 1), 4 - 7 weights (column tile (u - 4) >> 1, plane u & 1);
  // tap j of step r_ whose window chunk lies at wb and whose weights lie in slot sl_
  auto read_unit = [&](int u, int r_, int j, int set, int wb, int sl_, const unsigned* av, int apx) {
    const int t = r_ * 3 + j;
    if (u < 4) {
      const int a = u >> 1;
      const int px = apx + 32 * a + (t / 3) * Wi + t % 3;  // window pixel of this lane's row at tap t (raster order)
      const int ad = wb + (px * 2 + (lh ^ ((px >> 3) & 1))) * 16;
      const bool ok = (av[a] >> t) & 1u;
      if (u & 1) lds_read(al[set][a], lds0 + (unsigned)(ok ? ad + CFG::W_PLANE : (int)CFG::ZERO_OFF));
      else lds_read(ah[set][a], lds0 + (unsigned)(ok ? ad : (int)CFG::ZERO_OFF));
    } else {
      const int b = (u - 4) >> 1;
      const unsigned pbo = lds0 + 2 * CFG::WIN + sl_ * CFG::B_STEP;
      if (u & 1) lds_read(bl[set][b], pbo + b_addr[b] + (j * 2 + 1) * CFG::B_TAP);
      else lds_read(bh[set][b], pbo + b_addr[b] + (j * 2) * CFG::B_TAP);
    }
  };
  auto mfma_k = [&](int k, int set) {  // k = 0 .. 11: small terms first (l h, h l, h h), the four accumulators in turn
    const int term = k >> 2, a = (k >> 1) & 1, b = k & 1;
    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 0 ? al[set][a] : ah[set][a], term == 1 ? bl[set][b] : bh[set][b], acc[a][b], 0, 0, 0);
  };
  auto stage_b_one = [&](int kc_, int r_, int slot_, int bn0, int j) {
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + 2 * CFG::WIN + slot_ * CFG::B_STEP;
    const int i = (wave * CFG::B_IT + j) % CFG::B_INSTR;  // (scalar)
    const int jp = i >> 1;                                 // (tap of the step, plane)
    const int t = r_ * 3 + (jp >> 1);
    const char* sb = reinterpret_cast<const char*>((jp & 1) ? p.Wl : p.Wh) + (int64_t)(p.wt0 + t * p.wtstep) * w_tap_bytes + (kc_ * p.Co + bn0) * 32;
    // (the lane offset opaque at its use: (scalar base) + (32-bit lane offset) is one instruction; hipcc otherwise folds the
    //  loop-invariant half of the sum into 64-bit lane addresses per request, keeps all of them and spills)
    unsigned off = b_off[j];
    asm volatile("" : "+v"(off));
    __builtin_amdgcn_global_load_lds((gbl_void*)(sb + off), (lds_void*)(uintptr_t)(base + i * 1024), 16, 0, 0);
  };
  auto stage_win_one = [&](int kc_, int buf, int it) {
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + buf * CFG::WIN;
    const int i = w_instr(it);  // (scalar)
    const char* sb = reinterpret_cast<const char*>((i * 64 >= 2 * PP) ? p.Al : p.Ah) + kc_ * 32;  // (scalar)
    unsigned off = w_off[it];
    asm volatile("" : "+v"(off));
    __builtin_amdgcn_global_load_lds((gbl_void*)(sb + off), (lds_void*)(uintptr_t)(base + i * 1024), 16, 0, 0);
  };
  auto step = [&](int kc, bool first, bool last_step, bool last_tile, auto r_c, auto p_c) {
    constexpr int r = decltype(r_c)::value, P = decltype(p_c)::value;  // P = (kc + r) & 1: this step's weight slot and tap-0 set
    // (opaque to the optimiser: the fragment addresses and tap-validity selects below are invariant over the K loop, and
    //  hipcc otherwise hoists all 2 x 18 of them — and their 18 lane masks — out of it and spills them)
    int wbase = (kc & 1) * CFG::WIN;
    asm volatile("" : "+s"(wbase));
    unsigned av[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      av[a] = a_valid[a];
      asm volatile("" : "+v"(av[a]));
    }
    int apx = a_px0;  // (likewise: eighteen per-tap lane offsets, invariant over the loop, would be kept — three of them in scratch)
    asm volatile("" : "+v"(apx));
    const bool more = kc + 1 < kc1;  // (another chunk of this tile behind this one)
    if (r == 0 && first) {
      // a tile's first step: everybody's requests for it have landed (behind an epilogue this wave has drained its own
      // already, see there), its tap 0 is read in the open
      // (inline asm with a memory clobber: the s_barrier builtin is no memory operation to the optimiser, which may move
      //  fragment reads above it — it did, intermittently wrong results on the device)
      if (fresh) asm volatile("s_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      fresh = false;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; ++u) read_unit(u, 0, 0, P, wbase, P, av, apx);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- T0
    wait_set(P);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      mfma_k(k, P);
      __builtin_amdgcn_sched_barrier(0);
      if (k < 8) read_unit(k, r, 1, P ^ 1, wbase, P, av, apx);
      else if (r == 1 && (more || !last_tile) && k - 7 < CFG::W_IT) stage_win_one(more ? kc + 1 : next_kc0, (kc + 1) & 1, k - 7);  // window pieces 1 .. 4
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- T1
    wait_set(P ^ 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      mfma_k(k, P ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      if (k < 8) read_unit(k, r, 2, P, wbase, P, av, apx);
      else if (r == 1 && (more || !last_tile) && k - 3 < CFG::W_IT) stage_win_one(more ? kc + 1 : next_kc0, (kc + 1) & 1, k - 3);  // pieces 5 ..
      __builtin_amdgcn_sched_barrier(0);
    }
    static_assert(CFG::W_IT <= 9, "window pieces: one in T2 of step 0, four in T0 and up to four in T1 of step 1");
    // ---- hand-over: this wave's reads of the step are done; the next step's operands have landed — at r == 1 the window
    //      requested since the last hand-over (behind the weights, in issue order) may stay in flight: it is not read before
    //      the next chunk (none is requested in the last chunk of the last tile)
    wait_set(P);
    if (r == 1 && (more || !last_tile)) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(CFG::W_IT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- T2
    // what is requested behind this hand-over: the weights of the step after next into the slot just released — step
    // (kc, 2) behind step 0, (next chunk, 0 / 1) behind steps 1 / 2 (the next TILE's first chunk behind this tile's last) —
    // and, behind step 0, the first piece of the next chunk's window
    const bool w_here = r == 0 || more, w_next = !w_here && !last_tile;
    const int w_kc = r == 0 ? kc : (more ? kc + 1 : next_kc0), w_r = r == 0 ? 2 : r - 1, w_n0 = w_here ? n0 : next_n0;
    if (r == 0 && !more && !last_tile) setup_window(next_m0);
    int wnext = ((r == 2 ? kc + 1 : kc) & 1) * CFG::WIN;  // window buffer of the next step
    asm volatile("" : "+s"(wnext));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      mfma_k(k, P);
      __builtin_amdgcn_sched_barrier(0);
      if (k < 8) {
        if (!last_step) read_unit(k, r == 2 ? 0 : r + 1, 0, P ^ 1, wnext, P ^ 1, av, apx);
      } else if (k - 8 < CFG::B_IT) {
        if (w_here || w_next) stage_b_one(w_kc, w_r, P, w_n0, k - 8);
      } else if (k == 11 && r == 0 && (more || !last_tile)) stage_win_one(more ? kc + 1 : next_kc0, (kc + 1) & 1, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    static_assert(CFG::B_IT <= 3, "weight requests of a step fit the gaps 8 .. 10 of T2");
  };

  // ---- fused epilogue of one tile: eight 8-row slices per wave, lane = (row lane >> 3 of the slice, channels (lane & 7) * 8 ..)
  const int e_row = lane >> 3, e_c0 = (lane & 7) * 8;
  float* img = reinterpret_cast<float*>(smem + CFG::EPI_OFF) + wave * (8 * PITCH);
  unsigned vmax = 0;
  // Requests: every slice's addend planes and mask bytes are in flight before the first one is used — the first half
  // already under the tile's last K chunk (the registers exist: the K loop needs ~170), the second half when the fragment
  // registers are free — so that their latency, an HBM miss each, is not paid in the open.
  constexpr int NH = 0;
  f16x8 e_h2[NS], e_l2[NS];
  uint2 e_mk[NS];
  auto epi_request = [&](int m0_t, int n0_t, int s0, int s1) {
    const int mrow0 = m0_t + wave * 64 + e_row;
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
      if (sl < s0 || sl >= s1) continue;
      const int m = mrow0 + sl * 8;
      const bool ok = m < M;
      const int64_t e = ok ? (int64_t)m * p.Co + n0_t + e_c0 : 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) e_h2[sl][j] = (_Float16)0.f, e_l2[sl][j] = (_Float16)0.f;
      e_mk[sl] = make_uint2(0x01010101u, 0x01010101u);
      // (streaming accesses: read once / written once.  The L2 is needed for the window's lines, which the four channel
      //  chunks of a tile come back to one after the other — measured: 2.3 x the operand bytes from the fabric when the
      //  epilogue's 640 MB wash through the same cache with the default policy)
      if (p.add_h && ok) {
        e_h2[sl] = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(p.add_h + e));
        e_l2[sl] = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(p.add_l + e));
      }
      if (p.mask && ok) {
        const int mr = m - fdiv(m, p.div_mask) * p.mask_rows;
        e_mk[sl] = *reinterpret_cast<const uint2*>(p.mask + (int64_t)mr * p.Co + n0_t + e_c0);
      }
    }
  };
  auto epilogue = [&](int m0_t, int n0_t) {
    epi_request(m0_t, n0_t, NH, NS);
    const int mrow0 = m0_t + wave * 64 + e_row;
    float mult[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) mult[j] = sc_out;
    if (p.scale) {
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(p.scale + n0_t + e_c0), s1 = *reinterpret_cast<const f32x4*>(p.scale + n0_t + e_c0 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) mult[j] *= s0[j], mult[4 + j] *= s1[j];
    }
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
      const int a = sl >> 2, q = sl & 3;
      // MFMA layout (lane = channel, registers = pixels) -> image [8 rows][64 channels] -> lane = 8 channels of one pixel
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) img[(i + 4 * lh) * PITCH + b * 32 + lr] = acc[a][b][4 * q + i] * inv;
      // (LDS operations of one wave execute in order: no barrier between its own writes and reads)
      const f32x4 p0 = *reinterpret_cast<const f32x4*>(img + e_row * PITCH + e_c0);
      const f32x4 p1 = *reinterpret_cast<const f32x4*>(img + e_row * PITCH + e_c0 + 4);
      const int m = mrow0 + sl * 8;
      if (m >= M) continue;
      float v[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
      if (p.add_h) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += ((float)e_h2[sl][j] + (float)e_l2[sl][j]) * inv2;
      }
      f16x8 h, l;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool keep = ((j < 4 ? e_mk[sl].x >> (8 * j) : e_mk[sl].y >> (8 * (j - 4))) & 0xffu) != 0;
        float xs = keep ? v[j] * mult[j] : 0.f;
        asm volatile("" : "+v"(xs));  // h and the residual from the SAME fp32 value (see split2)
        const _Float16 hh = (_Float16)xs;
        h[j] = hh;
        l[j] = (_Float16)(xs - (float)hh);
        vmax = max(vmax, __float_as_uint(xs) & 0x7fffffffu);
      }
      // Before this wave's FIRST store: everything it has requested so far has landed (the operands above are in
      // registers, and the next tile's first weights were requested a whole epilogue ago) — said explicitly, because
      // the next tile's first step then does not wait on vmcnt at all: the stores below would be in that count, and a
      // store's acknowledgement (an HBM round trip per tile, in the open) is nothing the K loop has to wait for.
      if (sl == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int64_t e = (int64_t)m * p.Co + n0_t + e_c0;
      __builtin_nontemporal_store(h, reinterpret_cast<f16x8*>(p.out_h + e));
      __builtin_nontemporal_store(l, reinterpret_cast<f16x8*>(p.out_l + e));
    }
    fresh = true;
  };

  // ---- a K slice of a split tile: accumulators -> slab, arrive, and the LAST slice of its tile sums the slabs in slice order
  //      (true: `acc` now holds the tile's sums, the caller runs the epilogue).  Cross-CU visibility by the book
  //      (MI355X_MICROARCH.md, inter-workgroup visibility): plain stores, workgroup barrier, one lane's agent-scope release
  //      + drained vmcnt, relaxed agent-scope arrival; the last arriver: agent-scope acquire (invalidates THIS CU's L1),
  //      workgroup barrier, plain loads.  A split item is the last of its workgroup's walk: nothing else is in flight.
  auto combine = [&](int part_) -> bool {
    const int tl = __builtin_amdgcn_readfirstlane(part_ / p.split_S);
    f32x4* slab = reinterpret_cast<f32x4*>(p.slabs) + (size_t)part_ * (16 * CFG::NT);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = acc[a][b][4 * q + i];
          slab[((a * TN + b) * 4 + q) * CFG::NT + tid] = v;
        }
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem + CFG::ZERO_OFF + 16);  // (behind the 16 zero bytes fragment reads use; re-zeroed below)
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int old = __hip_atomic_fetch_add(p.arrivals + tl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = old == p.split_S - 1;
      if (last) {
        __hip_atomic_store(p.arrivals + tl, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (every slice has arrived: clean for the next launch)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      *flag = last;
    }
    __syncthreads();
    const bool last = __builtin_amdgcn_readfirstlane(*flag) != 0;  // (uniform: the epilogue behind it keeps scalar control flow)
    __syncthreads();
    if (tid == 0) *flag = 0;
    if (!last) return false;
    // (all sixteen loads of TWO slabs in flight before the first add: read one dependent load at a time this sum took 55 us
    //  for eight slabs — three times the slice's own K loop.  split_S is even.)
    const f32x4* src = reinterpret_cast<const f32x4*>(p.slabs) + (size_t)tl * p.split_S * (16 * CFG::NT) + tid;
    f32x4 v[16], w0[16], w1[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = src[e * CFG::NT], w0[e] = src[(16 + e) * CFG::NT];
#pragma unroll
    for (int e = 0; e < 16; ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i) v[e][i] += w0[e][i];
    for (int sl = 2; sl < p.split_S; sl += 2) {
      const f32x4* s2 = src + (size_t)sl * (16 * CFG::NT);
#pragma unroll
      for (int e = 0; e < 16; ++e) w0[e] = s2[e * CFG::NT], w1[e] = s2[(16 + e) * CFG::NT];
#pragma unroll
      for (int e = 0; e < 16; ++e)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[e][i] = (v[e][i] + w0[e][i]) + w1[e][i];
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[a][b][4 * q + i] = v[(a * TN + b) * 4 + q][i];
    return true;
  };

  // prologue: the first tile's first window and first three taps
  setup_window(m0);
  setup_frag(m0);
  stage_win(__builtin_amdgcn_readfirstlane(desc & 0xff), 0);
  stage_b(__builtin_amdgcn_readfirstlane(desc & 0xff), 0, 0, n0);
  stage_b(__builtin_amdgcn_readfirstlane(desc & 0xff), 1, 1, n0);  // (the request stream runs two steps ahead of the MFMAs)
  // the second workgroup of a CU starts half a tile late (the host passes the delay in units of 64 s_sleep cycles): from
  // then on one workgroup's epilogue runs beside the other's K loop
  if (blockIdx.x >= (unsigned)(G / 2))
    for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(64);
  // (the chunk parity of the LDS buffers restarts with every tile: KC is even — checked by the host — so that the window
  //  buffer (kc & 1) and the weight slot ((kc + r) & 1) of a tile's first step are those the previous tile's last step fed)
#ifdef LK_WINP_TRACE
  int trace_it = 0;
  if (tid == 0 && blockIdx.x < 1024) {
    g_winp_trace[(blockIdx.x * 16 + 15) * 3] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
    g_winp_trace[(blockIdx.x * 16 + 15) * 3 + 1] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    g_winp_trace[(blockIdx.x * 16 + 15) * 3 + 2] = __builtin_amdgcn_s_memtime();
    g_winp_trace[(blockIdx.x * 16 + 14) * 3] = wall_clock64();  // (100 MHz, one counter for the device: slot 14 = wall start, wall end, s_memtime end)
  }
#define LK_WINP_STAMP(k) \
  if (tid == 0 && blockIdx.x < 1024 && trace_it < 15) g_winp_trace[(blockIdx.x * 16 + trace_it) * 3 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define LK_WINP_STAMP(k)
#endif
  while (true) {
    const bool last_tile = !item(tile + G, next_m0, next_n0, next_desc);
    // (uniform by construction; said explicitly — carried around the persistent loop hipcc books them as divergent, and the
    //  staging bases and LDS buffer selects below live in scalar registers)
    const int kc0 = __builtin_amdgcn_readfirstlane(desc & 0xff), part = __builtin_amdgcn_readfirstlane(desc >> 16) - 1;
    kc1 = __builtin_amdgcn_readfirstlane((desc >> 8) & 0xff), next_kc0 = __builtin_amdgcn_readfirstlane(next_desc & 0xff);
    LK_WINP_STAMP(0)
    for (int kc = kc0; kc < kc1; kc += 2) {  // (chunk ranges are even in start and length)
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      step(kc, kc == kc0, false, last_tile, I0{}, I0{});
      step(kc, false, false, last_tile, I1{}, I1{});
      step(kc, false, false, last_tile, I2{}, I0{});
      step(kc + 1, false, false, last_tile, I0{}, I1{});
      step(kc + 1, false, false, last_tile, I1{}, I0{});
      step(kc + 1, false, kc + 2 >= kc1, last_tile, I2{}, I1{});
    }
    // (the fragment sets are dead here — the next tile's first step reads its tap 0 itself — which the optimiser cannot see
    //  through the run-time `first` / `last_step`: without this it keeps 32 registers alive across the epilogue and spills)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      asm volatile("" : "=v"(ah[0][i]), "=v"(al[0][i]), "=v"(bh[0][i]), "=v"(bl[0][i]), "=v"(ah[1][i]), "=v"(al[1][i]), "=v"(bh[1][i]), "=v"(bl[1][i]));
    LK_WINP_STAMP(1)
    if (part >= 0 && !combine(part)) {
    } else if (!(ablate & 1)) epilogue(m0, n0);
    else if (acc[0][0][0] == 12345.678f) p.out_h[0] = (_Float16)1.f;
    LK_WINP_STAMP(2)
#ifdef LK_WINP_TRACE
    ++trace_it;
#endif
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    if (last_tile) break;
    tile += G;
    m0 = next_m0, n0 = next_n0, desc = next_desc;
    if (!frag_const || (desc >> 16)) setup_frag(m0);
  }
  if (p.amax_out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = max(vmax, (unsigned)__shfl_xor((int)vmax, off, 64));
    const int back = -so < -126 ? -126 : -so;
    if (lane == 0 && vmax) atomicMax(p.amax_out, __float_as_uint(__uint_as_float(vmax) * exp2i(back)));
  }
#ifdef LK_WINP_TRACE
  if (tid == 0 && blockIdx.x < 1024) {
    g_winp_trace[(blockIdx.x * 16 + 14) * 3 + 1] = wall_clock64();
    g_winp_trace[(blockIdx.x * 16 + 14) * 3 + 2] = __builtin_amdgcn_s_memtime();
  }
#endif
}

}  // namespace lk

using namespace lk;

#ifdef LK_WINP_TRACE
extern "C" int lk_winp_trace_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_winp_trace), sizeof(unsigned long long) * 1024 * 16 * 3) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int lk_absmax_f32(const float* x, int64_t n, const float* cscale, int64_t inner, int64_t C, unsigned* out,
                             void* stream) {
  LK_REQUIRE(x && out && n >= 0 && (!cscale || (inner >= 1 && C >= 1)), "lk_absmax_f32: bad arguments");
  hipError_t e = hipMemsetAsync(out, 0, sizeof(unsigned), (hipStream_t)stream);
  if (e != hipSuccess) {
    set_error("lk_absmax_f32: memset: %s", hipGetErrorString(e));
    return LK_ELAUNCH;
  }
  if (n == 0) return LK_OK;
  if (!cscale && n >= 4096 && n % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    int64_t blocks = (n / 4 + 1023) / 1024;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(absmax4_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(x), n / 4, out);
    return check_launch("absmax4_kernel");
  }
  int64_t blocks = (n + 2047) / 2048;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, cscale,
                     cscale ? inner : 1, cscale ? (int)C : 1, out);
  return check_launch("absmax_kernel");
}

extern "C" int lk_copy_absmax_f32(const float* x, float* y, int64_t n, unsigned* amax, void* stream) {
  LK_REQUIRE(x && y && amax && n >= 0 && n % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0,
             "lk_copy_absmax_f32: 16-byte aligned buffers, n % 4 == 0");
  if (n == 0) return LK_OK;
  int64_t blocks = (n / 4 + 1023) / 1024;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(copy_absmax4_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), n / 4, amax);
  return check_launch("copy_absmax4_kernel");
}

extern "C" int lk_split_f16x2(const float* x, int64_t n, const float* amax, float bound_mul, void* planes_h,
                              void* planes_l, int* sexp, void* stream) {
  LK_REQUIRE(x && amax && planes_h && planes_l && sexp && n >= 0 && n % 8 == 0, "lk_split_f16x2: bad arguments (n % 8 == 0)");
  if (n == 0) return LK_OK;
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(split_f16x2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n / 8, amax,
                     bound_mul, (_Float16*)planes_h, (_Float16*)planes_l, sexp);
  return check_launch("split_f16x2_kernel");
}

// Patch matrix of a convolution as split planes (see im2col_split_f16x2_kernel): planes [B * Ho * Wo][Kp] with one scale from
// amax[0] = max|x| (every entry of the matrix is an entry of x or zero).  Kp % 8 == 0, Kp >= KH * KW * C.
extern "C" int lk_im2col_split_f16x2(const float* x, int64_t B, int64_t H, int64_t W, int64_t C, int64_t KH, int64_t KW,
                                     int64_t stride, int64_t pad, int64_t Ho, int64_t Wo, int64_t Kp, const float* amax,
                                     void* planes_h, void* planes_l, int* sexp, void* stream) {
  LK_REQUIRE(x && amax && planes_h && planes_l && sexp && B >= 0 && H >= 1 && W >= 1 && C >= 1 && KH >= 1 && KW >= 1 && stride >= 1 &&
                 pad >= 0 && Ho >= 1 && Wo >= 1 && Kp % 8 == 0 && Kp >= KH * KW * C,
             "lk_im2col_split_f16x2: bad arguments (Kp % 8 == 0, Kp >= KH * KW * C)");
  LK_REQUIRE((Ho - 1) * stride - pad < H && (Wo - 1) * stride - pad < W, "lk_im2col_split_f16x2: output grid outside the input");
  const int64_t rows = B * Ho * Wo, total8 = rows * (Kp / 8);
  LK_REQUIRE(total8 < (1ll << 31) && B * H * W * C < (1ll << 40), "lk_im2col_split_f16x2: too large");
  LK_REQUIRE(C % 8 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) == 0, "lk_im2col_split_f16x2: x must be 16-byte aligned");
  if (total8 == 0) return LK_OK;
  Im2colGeom g;
  g.B = (int)B, g.H = (int)H, g.W = (int)W, g.C = (int)C, g.KH = (int)KH, g.KW = (int)KW, g.stride = (int)stride, g.pad = (int)pad;
  g.Ho = (int)Ho, g.Wo = (int)Wo, g.Kp = (int)Kp, g.n = (int)(KH * KW * C);
  g.div_wo = make_fastdiv((int)Wo), g.div_howo = make_fastdiv((int)(Ho * Wo)), g.div_c = make_fastdiv((int)C);
  g.div_kw = make_fastdiv((int)KW), g.div_k8 = make_fastdiv((int)(Kp / 8));
  int64_t blocks = (total8 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(im2col_split_f16x2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, g, amax,
                     (_Float16*)planes_h, (_Float16*)planes_l, sexp, total8);
  return check_launch("im2col_split_f16x2_kernel");
}

// x [N][per] fp32 -> planes with ONE SCALE PER IMAGE: sexp[n] from the image's own max|x|, which is also left in amax[n]
// (bit pattern of a float; what the next producer derives its bound from).  per % 8 == 0.
extern "C" int lk_split_images_f16x2(const float* x, int64_t N, int64_t per, void* planes_h, void* planes_l, int* sexp,
                                     unsigned* amax, void* stream) {
  LK_REQUIRE(x && planes_h && planes_l && sexp && amax && N >= 0 && per >= 0 && per % 8 == 0 && N < 65536,
             "lk_split_images_f16x2: bad arguments (per % 8 == 0, N < 65536)");
  if (N == 0) return LK_OK;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(amax, 0, sizeof(unsigned) * (size_t)N, st);
  if (e != hipSuccess) {
    set_error("lk_split_images_f16x2: memset: %s", hipGetErrorString(e));
    return LK_ELAUNCH;
  }
  if (per == 0) return LK_OK;
  int64_t bx = (per / 8 + 255) / 256;
  if (bx * N > 8192) bx = (8192 + N - 1) / N;
  hipLaunchKernelGGL(absmax_images_kernel, dim3((unsigned)bx, (unsigned)N), dim3(256), 0, st, x, per / 4, amax);
  int rc = check_launch("absmax_images_kernel");
  if (rc != LK_OK) return rc;
  hipLaunchKernelGGL(split_images_f16x2_kernel, dim3((unsigned)bx, (unsigned)N), dim3(256), 0, st, x, per / 8, amax,
                     (_Float16*)planes_h, (_Float16*)planes_l, sexp);
  return check_launch("split_images_f16x2_kernel");
}

extern "C" int lk_conv_prep_weights_f16x2(const float* W, int64_t Co, int64_t Ci, int64_t taps, int transpose,
                                          const float* cscale, unsigned* amax_ws, void* planes, int* sexp, void* stream) {
  LK_REQUIRE(W && amax_ws && planes && sexp && Co >= 1 && Ci >= 1 && taps >= 1 && taps <= 9,
             "lk_conv_prep_weights_f16x2: bad arguments");
  const int64_t total = Co * Ci * taps;
  int rc = lk_absmax_f32(W, total, cscale, Ci * taps, Co, amax_ws, stream);
  if (rc != LK_OK) return rc;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(conv_prep_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, W, (int)Co,
                     (int)Ci, (int)taps, transpose, cscale, amax_ws, (_Float16*)planes, sexp);
  return check_launch("conv_prep_weights_kernel");
}

extern "C" int lk_conv_winp_eligible(int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t T, int mask_is_float);

template <typename CFG>
static int launch_conv(const ConvGeom& g, const void* Ah, const void* Al, const void* Wh, const void* Wl, const int* a_sexp,
                       const int* w_sexp, const void* zero16, float* out, int accumulate, unsigned* amax_out,
                       hipStream_t stream, const ConvVjp* fz = nullptr) {
  const ConvVjp plain = (fz && g.out_planes) ? *fz : ConvVjp{};  // (the plain epilogue's split-planes output rides in a ConvVjp)
  if (g.out_planes) fz = nullptr;
  const int64_t M = (int64_t)g.N * g.Hc * g.Wc;
  const int nb_m = (int)((M + CFG::BM - 1) / CFG::BM), nb_n = (g.Co + CFG::BN - 1) / CFG::BN;
  const size_t lds = (size_t)CFG::NBUF * CFG::STAGE;
  if (fz && fz->fwd_y) {  // forward epilogue (BatchNorm / add / ReLU): the staging image + BM + 1 words for the images' maxima
    if constexpr (CFG::FUSABLE) {
      const size_t need = (size_t)CFG::EPI_LDS + 4 * (CFG::BM + 1);
      const size_t lds_w = lds > need ? lds : (need + 15) / 16 * 16;
      static bool attr_set_w = false;
      if (!attr_set_w) {
        (void)hipFuncSetAttribute((const void*)conv_f16x2_kernel<CFG, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w);
        attr_set_w = true;
      }
      hipLaunchKernelGGL((conv_f16x2_kernel<CFG, true, true>), dim3((unsigned)(nb_m * nb_n)), dim3(CFG::NT), lds_w, stream, g,
                         (const _Float16*)Ah, (const _Float16*)Al, (const _Float16*)Wh, (const _Float16*)Wl, a_sexp, w_sexp,
                         (const _Float16*)zero16, out, accumulate, amax_out, nb_m, *fz);
      return check_launch("conv_f16x2_kernel(bn_act)");
    } else {
      set_error("lk_conv_bn_act_nhwc_f16x2: this tile shape has no fused epilogue");
      return LK_EINVAL;
    }
  }
  if (fz) {
    if constexpr (CFG::FUSABLE) {
      const size_t lds_f = lds > (size_t)CFG::EPI_LDS ? lds : (size_t)CFG::EPI_LDS;
      static bool attr_set_f = false;
      if (!attr_set_f) {
        (void)hipFuncSetAttribute((const void*)conv_f16x2_kernel<CFG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f);
        attr_set_f = true;
      }
      hipLaunchKernelGGL((conv_f16x2_kernel<CFG, true>), dim3((unsigned)(nb_m * nb_n)), dim3(CFG::NT), lds_f, stream, g,
                         (const _Float16*)Ah, (const _Float16*)Al, (const _Float16*)Wh, (const _Float16*)Wl, a_sexp, w_sexp,
                         (const _Float16*)zero16, out, accumulate, amax_out, nb_m, *fz);
      return check_launch("conv_f16x2_kernel(vjp)");
    } else {
      set_error("lk_conv_nhwc_f16x2_vjp: this tile shape has no fused epilogue");
      return LK_EINVAL;
    }
  }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_f16x2_kernel<CFG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_f16x2_kernel<CFG, false>), dim3((unsigned)(nb_m * nb_n)), dim3(CFG::NT), lds, stream, g, (const _Float16*)Ah,
                     (const _Float16*)Al, (const _Float16*)Wh, (const _Float16*)Wl, a_sexp, w_sexp, (const _Float16*)zero16,
                     out, accumulate, amax_out, nb_m, plain);
  return check_launch("conv_f16x2_kernel");
}

static int cu_count() {
  static int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    return v;
  }();
  return n;
}

// persistent window form (fused VJP epilogue only): one workgroup per CU walks through the pixel tiles.  The taps are put
// into raster order here (the sum over taps is commutative); their weight slices must then form an arithmetic sequence
// (forward: 0, 1, ..; backward-data: 8, 7, ..) — returns false (caller takes another kernel) otherwise.
// default number of column tiles of a pixel tile that share an XCD in the persistent window form (measured per shape:
// profiles/r05_winp_coloc.md); config bits 28-29 of a launch override it
static int coloc_default(int Ci, int Co) {
  // stand-alone launches at batch 9 x 128 (profiles/r05_winp_variants.md): 128 channels 270 -> 262 us with 2, 256 channels
  // 250 -> 240 with 4, 512 channels 262 -> 257 with 4 (8 = every column on one XCD: back to 261, its L2 then streams all eight
  // weight slices); 64 channels have one column tile
  (void)Ci;
  return Co >= 256 ? 4 : (Co == 128 ? 2 : 1);
}

// Slabs and arrival counters of the split tail, one set per stream (launches on one stream are ordered; the two lanes of a
// fit run this kernel side by side on their own streams).  Allocated at a stream's first split launch — 512 slabs of one
// tile's accumulators = 32 MB (64 MB for the eight-wave form) — and kept for the life of the process.
struct WinpScratch {
  float* slabs = nullptr;
  int* arrivals = nullptr;
  size_t slab_bytes = 0;
};
static WinpScratch* winp_scratch(hipStream_t stream, size_t slab_bytes, int n_arrivals) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, WinpScratch> pool;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  WinpScratch& w = pool[{dev, stream}];
  if (w.slab_bytes < slab_bytes || !w.arrivals) {
    // (growing: the old block stays allocated — a launch in flight may still use it)
    float* sl = nullptr;
    int* ar = nullptr;
    if (hipMalloc(&sl, slab_bytes) != hipSuccess) return nullptr;
    if (!w.arrivals) {
      if (hipMalloc(&ar, sizeof(int) * n_arrivals) != hipSuccess || hipMemset(ar, 0, sizeof(int) * n_arrivals) != hipSuccess) return nullptr;
      w.arrivals = ar;
    }
    w.slabs = sl, w.slab_bytes = slab_bytes;
  }
  return &w;
}

template <typename CFG>
static bool launch_winp(const ConvGeom& g, const void* Ah, const void* Al, const void* Wh /* chunk-major */, const void* Wl, const int* a_sexp,
                        const int* w_sexp, unsigned* amax_out, hipStream_t stream, const ConvVjp* fz, int* rc, int config) {
  int wt[9];
  for (int t = 0; t < 9; ++t) wt[t] = -1;
  for (int t = 0; t < 9; ++t) {
    if (g.dh[t] < -1 || g.dh[t] > 1 || g.dw[t] < -1 || g.dw[t] > 1) return false;
    const int c = (g.dh[t] + 1) * 3 + g.dw[t] + 1;
    if (wt[c] >= 0) return false;
    wt[c] = g.wt[t];
  }
  const int step = wt[1] - wt[0];
  for (int t = 0; t < 9; ++t)
    if (wt[t] != wt[0] + t * step) return false;
  WinPArgs p;
  const int64_t M = (int64_t)g.N * g.Hi * g.Wi;
  p.M = (int)M, p.Hi = g.Hi, p.Wi = g.Wi, p.Ci = g.Ci, p.Co = g.Co, p.HW = g.Hi * g.Wi;
  p.nb_m = (int)((M + CFG::BM - 1) / CFG::BM);
  p.n_tiles = p.nb_m * (g.Co / 64);
  p.wt0 = wt[0], p.wtstep = step;
  p.div_hw = g.div_hw, p.div_w = g.div_w, p.div_mask = fz->div_mask;
  p.Ah = (const _Float16*)Ah, p.Al = (const _Float16*)Al, p.Wh = (const _Float16*)Wh, p.Wl = (const _Float16*)Wl;
  p.a_sexp = a_sexp, p.w_sexp = w_sexp, p.add_sexp = fz->add_sexp;
  p.in_amax = fz->in_amax, p.scale_amax = fz->scale_amax;
  p.w_l1 = fz->w_l1, p.scale = fz->scale;
  p.add_h = fz->add_h, p.add_l = fz->add_l;
  p.mask = (const unsigned char*)fz->mask, p.mask_rows = (int)fz->mask_rows;
  // start delay of a CU's second workgroup, x 64 s_sleep cycles.  Round 4 started it half a tile late so that one workgroup's
  // epilogue would run beside the other's K loop; measured again in round 6 (tools/winp_bench.py, profiles/r06_winp_study.md):
  // with the reads and requests in the MFMA gaps a lone workgroup no longer runs at half rate, and the delay only idles the
  // second workgroup at the start — 0 is 2 - 4 % faster on every c4 shape
  p.stagger = 0;
  if ((config >> 20) & 31) p.stagger = ((config >> 20) & 31) - 1;  // (development: bits 20-24 = delay + 1)
  p.halo_all = (config >> 30) & 1;  // (development: stage the whole PP-pixel window as round 4 did)
  p.coloc = 1;
  p.out_h = fz->out_h, p.out_l = fz->out_l, p.out_sexp = fz->out_sexp;
  p.amax_out = amax_out;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_winp_f16x2_kernel<CFG>, hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS);
    attr_set = true;
  }
  const int wg_per_cu = (config & 524288) ? 1 : CFG::WG_PER_CU;  // (development: bit 19 = one workgroup per CU)
  if (wg_per_cu == 1) p.stagger = 0;
  const int grid = p.n_tiles < wg_per_cu * cu_count() ? p.n_tiles : wg_per_cu * cu_count();  // two workgroups per CU
  {
    // columns of a pixel tile that share an XCD (see `coords` in the kernel): config bits 28-29 = log2, 0 = the default below
    const int nb_n = g.Co / 64;
    int c = 1 << ((config >> 28) & 3);
    if (c == 1) c = coloc_default(g.Ci, g.Co);
    if (c > 1 && c <= nb_n && nb_n % c == 0 && nb_n / c <= 8 && 8 % (nb_n / c) == 0 && grid % 8 == 0 && (grid / 8) % c == 0) p.coloc = c;
  }
  // split tail (see `item` in the kernel; config bit 25 switches it ON — measured 15 - 30 us slower per c4 launch than leaving
  // the last round ragged: the slab round trip costs what the balance gains, profiles/r06_winp_study.md): the leftover tiles of the last round in S slices
  // each — S the largest divisor of the chunk count with L * S <= grid and an even number (>= 2) of chunks per slice
  p.split_S = 1, p.split_L = 0, p.split_v0 = 0, p.slabs = nullptr, p.arrivals = nullptr;
  {
    const int KC = g.Ci / 16, L = p.n_tiles % grid;
    int holes = 0;
    if (p.coloc > 1) holes = p.nb_m % (8 / ((g.Co / 64) / p.coloc));  // (`coords` with co-located columns: index space = tiles only then)
    if (L > 0 && p.n_tiles > grid && !holes && (config & 33554432)) {
      int S = 1;
      for (int d = 2; d <= 8 && d <= KC && d * L <= grid; d += 2)  // (even: the combine reads slabs in pairs)
        if (KC % d == 0 && (KC / d) % 2 == 0) S = d;
      if (S > 1) {
        WinpScratch* w = winp_scratch(stream, (size_t)cu_count() * wg_per_cu * 16 * CFG::NT * sizeof(float) * 4, cu_count() * wg_per_cu);
        if (w) p.split_S = S, p.split_L = L, p.split_v0 = p.n_tiles - L, p.slabs = w->slabs, p.arrivals = w->arrivals;
      }
    }
  }
  hipLaunchKernelGGL((conv_winp_f16x2_kernel<CFG>), dim3((unsigned)grid), dim3(CFG::NT), CFG::LDS, stream, p);
  *rc = check_launch("conv_winp_f16x2_kernel");
  return true;
}

// One launch of the implicit GEMM.  `taps`: T x {dh, dw, weight slice}.
static int conv_dispatch(const void* in_h, const void* in_l, const int* in_sexp, int64_t in_nsexp, int64_t N, int64_t Hi, int64_t Wi,
                         int64_t Ci, const void* w_h, const void* w_l, const int* w_sexp, int64_t Co,
                         int64_t Hc, int64_t Wc, int64_t in_mul, int64_t Ho, int64_t Wo, int64_t out_step,
                         int64_t oh0, int64_t ow0, int64_t T, const int* taps, const void* zero16, float* out,
                         int accumulate, unsigned* amax_out, int config, void* stream, const ConvVjp* fz) {
  LK_REQUIRE(in_h && in_l && in_sexp && w_h && w_l && w_sexp && zero16 && (out || fz) && taps, "lk_conv_nhwc_f16x2: null pointer");
  LK_REQUIRE(T >= 1 && T <= 9 && Ci >= 32 && Ci % 32 == 0 && Co >= 1 && N >= 1, "lk_conv_nhwc_f16x2: Ci % 32 == 0, 1..9 taps");
  LK_REQUIRE(N * Hc * Wc < (1ll << 31) && N * Hi * Wi * Ci < (1ll << 40), "lk_conv_nhwc_f16x2: tensor too large");
  LK_REQUIRE(in_nsexp == 1 || (in_nsexp == N && (!fz || (config & 16) || fz->fwd_y)),
             "lk_conv_nhwc_f16x2: in_nsexp is 1 or N (one scale per image: plain and forward epilogues only)");
  if (Hc == 0 || Wc == 0) return LK_OK;
  ConvGeom g;
  g.a_nsexp = (int)in_nsexp;
  g.N = (int)N, g.Hi = (int)Hi, g.Wi = (int)Wi, g.Ci = (int)Ci, g.Hc = (int)Hc, g.Wc = (int)Wc, g.Ho = (int)Ho,
  g.Wo = (int)Wo, g.Co = (int)Co, g.os = (int)out_step, g.oh0 = (int)oh0, g.ow0 = (int)ow0, g.im = (int)in_mul, g.T = (int)T;
  for (int t = 0; t < 9; ++t) g.dh[t] = g.dw[t] = g.wt[t] = 0;
  for (int t = 0; t < T; ++t) g.dh[t] = taps[3 * t], g.dw[t] = taps[3 * t + 1], g.wt[t] = taps[3 * t + 2];
  g.div_hw = make_fastdiv((int)(Hc * Wc)), g.div_w = make_fastdiv((int)Wc), g.div_n = make_fastdiv((int)N);
  // small maps: rows ordered (pixel, image) so that border taps drop out of whole tiles (config bit 15 turns it off)
  g.pmajor = (Hc * Wc <= (64 << (2 * ((config >> 16) & 3))) && N >= 64 && !(config & 16) && !(config & 32768)) ? 1 : 0;
  g.dense = out_step == 1 && oh0 == 0 && ow0 == 0 && Hc == Ho && Wc == Wo;
  g.out_nchw = (config & 16) ? 1 : 0;
  g.out_planes = (fz && fz->out_h && (config & 16)) ? 1 : 0;  // (fused launches never carry bit 4: conv_vjp_impl clears it)
  LK_REQUIRE(!g.out_nchw || (g.dense && (Ho * Wo) % 4 == 0 && !accumulate),
             "lk_conv_nhwc_f16x2: position-contiguous output needs a dense grid with Ho*Wo % 4 == 0 and no accumulate");
  hipStream_t st = (hipStream_t)stream;
  // persistent window form (fused launches with 64 output channels whose caller also handed over chunk-major weights;
  // config bit 27 switches it off): see conv_winp_f16x2_kernel
  if (fz && !g.out_planes && !fz->fwd_y && fz->wc_h && !(config & 134217728) && lk_conv_winp_eligible(N, Hi, Wi, Ci, Co, T, fz->mask && fz->mask_float) &&
      in_mul == 1 && Hc == Hi && Wc == Wi && g.dense) {
    int rc = LK_OK;
    if ((config & 67108864) && launch_winp<WinPCfg<512>>(g, in_h, in_l, fz->wc_h, fz->wc_l, in_sexp, w_sexp, amax_out, st, fz, &rc, config)) return rc;
    if (launch_winp<WinPCfg<256>>(g, in_h, in_l, fz->wc_h, fz->wc_l, in_sexp, w_sexp, amax_out, st, fz, &rc, config)) return rc;
  }
#define LK_CONV_GO(...) return launch_conv<ConvCfg<__VA_ARGS__>>(g, in_h, in_l, w_h, w_l, in_sexp, w_sexp, zero16, out, accumulate, amax_out, st, fz)
  switch ((config >> 12) & 7) {  // explicit tile shape (bits 12..14; the tests walk through them); 0: chosen below
    case 1: LK_CONV_GO(64, 64, 32, 2, 2, 2, 4);
    case 2: LK_CONV_GO(128, 64, 32, 2, 2, 2, 3);
    case 3: LK_CONV_GO(64, 128, 32, 2, 2, 2, 3);
    case 4: LK_CONV_GO(128, 128, 32, 2, 2);
    case 5: LK_CONV_GO(256, 64, 32, 4, 1);
    default: break;
  }
  {
    // Tile shape by occupancy (measured on the c4 layer shapes, profiles/r02_conv_tile_sweep.json): the big tile
    // (2 workgroups per CU = 512 slots) when it fills the chip and its last round is not mostly idle; otherwise the
    // half tile (3 per CU) if that gives >= 512 tiles; otherwise 64 x 64 (the forward of the deep, small-map layers at
    // batch 128 has only 64-128 big tiles).  A pure function of the shapes: the same launch always takes the same tile.
    const int64_t M = N * Hc * Wc;
    auto tiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((Co + bn - 1) / bn); };
    const bool narrow = Co <= 64;
    const int64_t tb = narrow ? tiles(256, 64) : tiles(128, 128);
    const int64_t rounds = (tb + 511) / 512;
    const bool big_ok = tb >= 512 && tb * 100 >= rounds * 512 * 74;
    if (!big_ok) {
      const int64_t tm = narrow ? tiles(128, 64) : tiles(64, 128);
      if (tm >= 512) {
        if (narrow) LK_CONV_GO(128, 64, 32, 2, 2, 2, 3);
        LK_CONV_GO(64, 128, 32, 2, 2, 2, 3);
      }
      LK_CONV_GO(64, 64, 32, 2, 2, 2, 4);
    }
    if (narrow) LK_CONV_GO(256, 64, 32, 4, 1);
    LK_CONV_GO(128, 128, 32, 2, 2);
  }
#undef LK_CONV_GO
}

// Does a fused 3 x 3 / stride-1 launch of this shape qualify for the persistent window form (conv_winp_f16x2_kernel)?  The
// caller then prepares chunk-major weights for it (lk_conv_nhwc_f16x2_vjp_wc).
extern "C" int lk_conv_winp_eligible(int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t T, int mask_is_float) {
  return T == 9 && Wi <= 47 && Hi * Wi >= 16 && Ci % 32 == 0 && Ci >= 32 && Co >= 64 && Co % 64 == 0 && N * Hi * Wi * Ci < (1ll << 30) &&
                 N * Hi * Wi * Co < (1ll << 31) && N * Hi * Wi >= 512 && !mask_is_float
             ? 1
             : 0;
}

extern "C" int lk_conv_nhwc_f16x2(const void* in_h, const void* in_l, const int* in_sexp, int64_t in_nsexp, int64_t N, int64_t Hi,
                                  int64_t Wi, int64_t Ci, const void* w_h, const void* w_l, const int* w_sexp, int64_t Co,
                                  int64_t Hc, int64_t Wc, int64_t in_mul, int64_t Ho, int64_t Wo, int64_t out_step,
                                  int64_t oh0, int64_t ow0, int64_t T, const int* taps, const void* zero16, float* out,
                                  int accumulate, unsigned* amax_out, int config, void* stream) {
  return conv_dispatch(in_h, in_l, in_sexp, in_nsexp, N, Hi, Wi, Ci, w_h, w_l, w_sexp, Co, Hc, Wc, in_mul, Ho, Wo, out_step, oh0, ow0, T,
                       taps, zero16, out, accumulate, amax_out, config, stream, nullptr);
}

// lk_conv_nhwc_f16x2 with the output POSITION-contiguous as two fp16 planes, out_h / out_l [N][Co][Ho * Wo], instead of
// fp32: scaled per entry of in_sexp (the whole tensor, or image by image) from the guaranteed bound
//   max|out_n| <= in_amax[n] * w_l1[0]      (in_amax: in_nsexp words, bit patterns, or NULL = 2^(15 - in_sexp[n]))
// with the scale left in out_sexp[n].  Dense grid, (Ho * Wo) % 4 == 0.  What the predictive's eigenbasis rotations hand to
// lk_kron_quadform_shared_planes_f16x2: no fp32 round trip, no splitting inside the quadratic-form kernel.
extern "C" int lk_conv_nhwc_f16x2_planes(const void* in_h, const void* in_l, const int* in_sexp, int64_t in_nsexp,
                                         const void* in_amax, int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, const void* w_h,
                                         const void* w_l, const int* w_sexp, const float* w_l1, int64_t Co, int64_t Ho,
                                         int64_t Wo, int64_t in_mul, int64_t T, const int* taps, const void* zero16,
                                         void* out_h, void* out_l, int* out_sexp, int config, void* stream) {
  LK_REQUIRE(w_l1 && out_h && out_l && out_sexp, "lk_conv_nhwc_f16x2_planes: null pointer");
  LK_REQUIRE((Ho * Wo) % 16 == 0, "lk_conv_nhwc_f16x2_planes: Ho * Wo % 16 == 0 (the planes are written in chunks of 16 positions)");
  ConvVjp fz{};
  fz.in_amax = (const unsigned*)in_amax, fz.w_l1 = w_l1;
  fz.out_h = (_Float16*)out_h, fz.out_l = (_Float16*)out_l, fz.out_sexp = out_sexp;
  fz.mask_rows = 1, fz.div_mask = make_fastdiv(1);
  return conv_dispatch(in_h, in_l, in_sexp, in_nsexp, N, Hi, Wi, Ci, w_h, w_l, w_sexp, Co, Ho, Wo, in_mul, Ho, Wo, 1, 0, 0, T, taps,
                       zero16, reinterpret_cast<float*>(out_h), 0, nullptr, config | 16, stream, &fz);
}

// lk_conv_nhwc_f16x2 (dense output grid, stride `in_mul`) with the forward's BatchNorm / residual add / ReLU in its epilogue:
//   y = act(conv(in, W) * scale[co] + shift[co] + addend)         act: 0 none, 1 ReLU
// — to the bit what lk_conv_nhwc_f16x2 followed by lk_bn_act_fwd_nhwc_f16x2 (x_mul = w_l1, no x_add) computes, in one launch
// and without the fp32 round trip of the convolution's output.  in_amax: in_namax (1 or N) words with the measured max|in_n|;
// outputs: y fp32 NHWC [N][Ho][Wo][Co], mask bytes (or NULL), y_h / y_l planes (or NULL) with y_sexp[N], y_bound[N] (the
// guaranteed bound behind each scale), y_amax[N] (measured max|y_n|, zeroed by the caller).  Co % 8 == 0.
extern "C" int lk_conv_bn_act_nhwc_f16x2(const void* in_h, const void* in_l, const int* in_sexp, int64_t in_nsexp,
                                         const void* in_amax, int64_t in_namax, int64_t N, int64_t Hi, int64_t Wi, int64_t Ci,
                                         const void* w_h, const void* w_l, const int* w_sexp, const float* w_l1, int64_t Co,
                                         int64_t Ho, int64_t Wo, int64_t in_mul, int64_t T, const int* taps, const void* zero16,
                                         const float* scale, const float* shift, const void* scale_amax, const void* shift_amax,
                                         const float* addend, const float* addend_bound, int64_t addend_nbound, int act,
                                         float* y, void* mask, void* y_h, void* y_l, int* y_sexp, float* y_bound, void* y_amax,
                                         int config, void* stream) {
  LK_REQUIRE(in_amax && w_l1 && scale && shift && scale_amax && shift_amax && y && y_sexp && y_bound && y_amax,
             "lk_conv_bn_act_nhwc_f16x2: null pointer");
  LK_REQUIRE(Co % 8 == 0 && (act == 0 || act == 1), "lk_conv_bn_act_nhwc_f16x2: Co % 8 == 0, act in 0..1");
  LK_REQUIRE(in_namax == 1 || in_namax == N, "lk_conv_bn_act_nhwc_f16x2: in_amax has 1 or N words");
  LK_REQUIRE(!addend || (addend_bound && (addend_nbound == 1 || addend_nbound == N)),
             "lk_conv_bn_act_nhwc_f16x2: the addend needs its bound (1 or N floats)");
  LK_REQUIRE((y_h == nullptr) == (y_l == nullptr), "lk_conv_bn_act_nhwc_f16x2: both planes or none");
  LK_REQUIRE(N * Ho * Wo * Co < (1ll << 40), "lk_conv_bn_act_nhwc_f16x2: tensor too large");
  ConvVjp fz{};
  fz.in_amax = (const unsigned*)in_amax, fz.w_l1 = w_l1;
  fz.mask_rows = 1, fz.div_mask = make_fastdiv(1);
  fz.out_h = (_Float16*)y_h, fz.out_l = (_Float16*)y_l, fz.out_sexp = y_sexp;
  fz.fwd_y = y, fz.fwd_mask = (unsigned char*)mask;
  fz.fwd_scale = scale, fz.fwd_shift = shift;
  fz.fwd_scale_amax = (const unsigned*)scale_amax, fz.fwd_shift_amax = (const unsigned*)shift_amax;
  fz.fwd_addend = addend, fz.fwd_addend_bound = addend_bound, fz.fwd_addend_nbound = (int)(addend ? addend_nbound : 1);
  fz.fwd_in_namax = (int)in_namax, fz.fwd_act = act;
  fz.fwd_bound = y_bound, fz.fwd_amax = (unsigned*)y_amax;
  return conv_dispatch(in_h, in_l, in_sexp, in_nsexp, N, Hi, Wi, Ci, w_h, w_l, w_sexp, Co, Ho, Wo, in_mul, Ho, Wo, 1, 0, 0, T, taps,
                       zero16, y, 0, nullptr, config & ~16, stream, &fz);
}

// The same launch with the fused VJP epilogue (see ConvVjp): dense output grid (the output tensor IS the class grid), no
// accumulate, Co % 8 == 0.  Emits the split tensor out_h / out_l / out_sexp and max|result| (out_amax, zeroed by the caller).
static int conv_vjp_impl(const void* in_h, const void* in_l, const int* in_sexp, const void* in_amax, int64_t N,
                         int64_t Hi, int64_t Wi, int64_t Ci, const void* w_h, const void* w_l, const int* w_sexp,
                         const float* w_l1, int64_t Co, int64_t Ho, int64_t Wo, int64_t T, const int* taps,
                         const void* zero16, const void* add_h, const void* add_l, const int* add_sexp,
                         const void* mask, int mask_is_float, const void* mult_amax, int64_t mask_rows,
                         const float* scale, const void* scale_amax, void* out_h, void* out_l, int* out_sexp,
                         void* out_amax, int config, void* stream, const void* wc_h = nullptr, const void* wc_l = nullptr) {
  LK_REQUIRE(w_l1 && out_h && out_l && out_sexp && out_amax, "lk_conv_nhwc_f16x2_vjp: null pointer");
  LK_REQUIRE(!wc_h == !wc_l, "lk_conv_nhwc_f16x2_vjp_wc: incomplete chunk-major weights");
  LK_REQUIRE(Co % 8 == 0, "lk_conv_nhwc_f16x2_vjp: Co % 8 == 0");
  LK_REQUIRE(!add_h || (add_l && add_sexp), "lk_conv_nhwc_f16x2_vjp: incomplete addend");
  LK_REQUIRE(!mask || mask_rows > 0, "lk_conv_nhwc_f16x2_vjp: mask_rows");
  LK_REQUIRE(!scale || scale_amax, "lk_conv_nhwc_f16x2_vjp: scale needs its bound");
  ConvVjp fz{};
  fz.in_amax = (const unsigned*)in_amax, fz.w_l1 = w_l1;
  fz.add_h = (const _Float16*)add_h, fz.add_l = (const _Float16*)add_l, fz.add_sexp = add_sexp;
  fz.mask = mask, fz.mask_float = mask_is_float, fz.mult_amax = (const unsigned*)mult_amax, fz.mask_rows = mask ? mask_rows : 1;
  LK_REQUIRE(fz.mask_rows >= 1 && fz.mask_rows < (1ll << 31), "lk_conv_nhwc_f16x2_vjp: mask_rows out of range");
  fz.div_mask = make_fastdiv((int)fz.mask_rows);
  fz.scale = scale, fz.scale_amax = (const unsigned*)scale_amax;
  fz.out_h = (_Float16*)out_h, fz.out_l = (_Float16*)out_l, fz.out_sexp = out_sexp;
  fz.wc_h = (const _Float16*)wc_h, fz.wc_l = (const _Float16*)wc_l;
  return conv_dispatch(in_h, in_l, in_sexp, 1, N, Hi, Wi, Ci, w_h, w_l, w_sexp, Co, Ho, Wo, 1, Ho, Wo, 1, 0, 0, T, taps, zero16,
                       nullptr, 0, (unsigned*)out_amax, config & ~16, stream, &fz);
}

extern "C" int lk_conv_nhwc_f16x2_vjp(const void* in_h, const void* in_l, const int* in_sexp, const void* in_amax, int64_t N,
                                      int64_t Hi, int64_t Wi, int64_t Ci, const void* w_h, const void* w_l, const int* w_sexp,
                                      const float* w_l1, int64_t Co, int64_t Ho, int64_t Wo, int64_t T, const int* taps,
                                      const void* zero16, const void* add_h, const void* add_l, const int* add_sexp,
                                      const void* mask, int mask_is_float, const void* mult_amax, int64_t mask_rows,
                                      const float* scale, const void* scale_amax, void* out_h, void* out_l, int* out_sexp,
                                      void* out_amax, int config, void* stream) {
  return conv_vjp_impl(in_h, in_l, in_sexp, in_amax, N, Hi, Wi, Ci, w_h, w_l, w_sexp, w_l1, Co, Ho, Wo, T, taps, zero16, add_h,
                       add_l, add_sexp, mask, mask_is_float, mult_amax, mask_rows, scale, scale_amax, out_h, out_l, out_sexp,
                       out_amax, config, stream);
}

// lk_conv_nhwc_f16x2_vjp with the weights ALSO in chunk-major order (wc_h / wc_l: [tap][Ci / 16][Co][16] fp16 per plane, same
// scale as w_h / w_l): shapes for which lk_conv_winp_eligible says so run the persistent window form, everything else
// ignores the extra planes.
extern "C" int lk_conv_nhwc_f16x2_vjp_wc(const void* in_h, const void* in_l, const int* in_sexp, const void* in_amax, int64_t N,
                                         int64_t Hi, int64_t Wi, int64_t Ci, const void* w_h, const void* w_l, const int* w_sexp,
                                         const float* w_l1, const void* wc_h, const void* wc_l, int64_t Co, int64_t Ho, int64_t Wo,
                                         int64_t T, const int* taps, const void* zero16, const void* add_h, const void* add_l,
                                         const int* add_sexp, const void* mask, int mask_is_float, const void* mult_amax,
                                         int64_t mask_rows, const float* scale, const void* scale_amax, void* out_h, void* out_l,
                                         int* out_sexp, void* out_amax, int config, void* stream) {
  return conv_vjp_impl(in_h, in_l, in_sexp, in_amax, N, Hi, Wi, Ci, w_h, w_l, w_sexp, w_l1, Co, Ho, Wo, T, taps, zero16, add_h,
                       add_l, add_sexp, mask, mask_is_float, mult_amax, mask_rows, scale, scale_amax, out_h, out_l, out_sexp,
                       out_amax, config, stream, wc_h, wc_l);
}

// Strided form (conv_strided_f16x2_kernel): the backward-data of a stride-`os` convolution — every residue class of the
// input-gradient pixels — and optionally of a second convolution reading the same input (in2_* / w2_*: NULL without one),
// with the fused VJP epilogue, in one launch.  `taps`: T x {dh, dw, weight slice, source (0 / 1), oh0, ow0}; every one of
// the os * os classes must own at least one tap (the launch writes only pixels of listed classes), Ho % os == Wo % os == 0,
// both cotangents are [N][Hi][Wi][Ci] with Hi == Ho / os, Wi == Wo / os.
template <typename CFG>
static int launch_strided(const lk::StridedGeom& g, const lk::StridedSrc& s1, const lk::StridedSrc& s2, const void* zero16,
                          unsigned* amax_out, hipStream_t stream, const lk::ConvVjp& fz) {
  const int64_t M = (int64_t)g.N * g.Hc * g.Wc;
  const int nb_m = (int)((M + CFG::BM - 1) / CFG::BM), nb_n = (g.Co + CFG::BN - 1) / CFG::BN;
  const size_t lds0 = (size_t)CFG::NBUF * CFG::STAGE;
  const size_t lds = lds0 > (size_t)CFG::EPI_LDS ? lds0 : (size_t)CFG::EPI_LDS;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)lk::conv_strided_f16x2_kernel<CFG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((lk::conv_strided_f16x2_kernel<CFG>), dim3((unsigned)(nb_m * g.ncls * nb_n)), dim3(CFG::NT), lds, stream, g, s1,
                     s2, (const _Float16*)zero16, amax_out, nb_m, fz);
  return lk::check_launch("conv_strided_f16x2_kernel");
}

extern "C" int lk_conv_nhwc_f16x2_vjp_strided(
    const void* in_h, const void* in_l, const int* in_sexp, const void* in_amax, const void* w_h, const void* w_l,
    const int* w_sexp, const float* w_l1, const void* in2_h, const void* in2_l, const int* in2_sexp, const void* in2_amax,
    const void* w2_h, const void* w2_l, const int* w2_sexp, const float* w2_l1, int64_t N, int64_t Hi, int64_t Wi, int64_t Ci,
    int64_t Co, int64_t Ho, int64_t Wo, int64_t os, int64_t T, const int* taps, const void* zero16, const void* add_h,
    const void* add_l, const int* add_sexp, const void* mask, int mask_is_float, const void* mult_amax, int64_t mask_rows,
    const float* scale, const void* scale_amax, void* out_h, void* out_l, int* out_sexp, void* out_amax, int config,
    void* stream) {
  using namespace lk;
  (void)config;
  LK_REQUIRE(in_h && in_l && in_sexp && w_h && w_l && w_sexp && w_l1 && zero16 && taps && out_h && out_l && out_sexp && out_amax,
             "lk_conv_nhwc_f16x2_vjp_strided: null pointer");
  const bool two = in2_h != nullptr;
  LK_REQUIRE(!two || (in2_l && in2_sexp && w2_h && w2_l && w2_sexp && w2_l1), "lk_conv_nhwc_f16x2_vjp_strided: incomplete second source");
  LK_REQUIRE(os >= 1 && os <= 2 && Ho % os == 0 && Wo % os == 0 && Hi == Ho / os && Wi == Wo / os,
             "lk_conv_nhwc_f16x2_vjp_strided: stride 1 or 2, Ho = os * Hi, Wo = os * Wi");
  LK_REQUIRE(T >= 1 && T <= 12 && Ci >= 32 && Ci % 32 == 0 && Co >= 8 && Co % 8 == 0 && N >= 1,
             "lk_conv_nhwc_f16x2_vjp_strided: 1..12 taps, Ci % 32 == 0, Co % 8 == 0");
  LK_REQUIRE(N * Ho * Wo < (1ll << 31) && N * Ho * Wo * Co < (1ll << 40) && N * Hi * Wi * Ci < (1ll << 40),
             "lk_conv_nhwc_f16x2_vjp_strided: tensor too large");
  LK_REQUIRE(!add_h || (add_l && add_sexp), "lk_conv_nhwc_f16x2_vjp_strided: incomplete addend");
  LK_REQUIRE(!mask || (mask_rows > 0 && mask_rows < (1ll << 31)), "lk_conv_nhwc_f16x2_vjp_strided: mask_rows");
  LK_REQUIRE(!scale || scale_amax, "lk_conv_nhwc_f16x2_vjp_strided: scale needs its bound");
  StridedGeom g;
  g.N = (int)N, g.Hi = (int)Hi, g.Wi = (int)Wi, g.Ci = (int)Ci, g.Hc = (int)Hi, g.Wc = (int)Wi, g.Ho = (int)Ho, g.Wo = (int)Wo,
  g.Co = (int)Co, g.os = (int)os, g.ncls = 0, g.second = 0, g.T = (int)T;
  for (int c = 0; c < 4; ++c) g.oh0[c] = g.ow0[c] = 0, g.cls_taps[c] = 0;
  for (int t = 0; t < 12; ++t) g.dh[t] = g.dw[t] = g.wt[t] = 0;
  int k = 0;
  for (int pass = 1; pass >= 0; --pass)  // the second source's taps first
    for (int t = 0; t < T; ++t) {
      const int* p = taps + 6 * t;
      LK_REQUIRE(p[3] == 0 || (p[3] == 1 && two), "lk_conv_nhwc_f16x2_vjp_strided: tap source");
      if (p[3] != pass) continue;
      LK_REQUIRE(p[4] >= 0 && p[4] < os && p[5] >= 0 && p[5] < os, "lk_conv_nhwc_f16x2_vjp_strided: residue class of a tap");
      int c = 0;
      while (c < g.ncls && (g.oh0[c] != p[4] || g.ow0[c] != p[5])) ++c;
      if (c == g.ncls) g.oh0[c] = p[4], g.ow0[c] = p[5], ++g.ncls;
      g.dh[k] = p[0], g.dw[k] = p[1], g.wt[k] = p[2];
      g.cls_taps[c] |= 1u << k;
      if (pass) g.second |= 1u << k;
      ++k;
    }
  LK_REQUIRE(g.ncls == os * os, "lk_conv_nhwc_f16x2_vjp_strided: a residue class without taps");
  // heavy classes first within each group of tiles (the classes of one tile group run side by side)
  for (int a = 0; a < g.ncls; ++a)
    for (int b = a + 1; b < g.ncls; ++b)
      if (__builtin_popcount(g.cls_taps[b]) > __builtin_popcount(g.cls_taps[a])) {
        std::swap(g.cls_taps[a], g.cls_taps[b]), std::swap(g.oh0[a], g.oh0[b]), std::swap(g.ow0[a], g.ow0[b]);
      }
  g.div_hw = make_fastdiv((int)(Hi * Wi)), g.div_w = make_fastdiv((int)Wi), g.div_cls = make_fastdiv(g.ncls);
  StridedSrc s1{(const _Float16*)in_h, (const _Float16*)in_l, (const _Float16*)w_h, (const _Float16*)w_l, in_sexp, w_sexp,
                (const unsigned*)in_amax, w_l1};
  StridedSrc s2 = s1;
  if (two)
    s2 = StridedSrc{(const _Float16*)in2_h, (const _Float16*)in2_l, (const _Float16*)w2_h, (const _Float16*)w2_l, in2_sexp, w2_sexp,
                    (const unsigned*)in2_amax, w2_l1};
  ConvVjp fz{};
  fz.in_amax = nullptr, fz.w_l1 = nullptr;
  fz.add_h = (const _Float16*)add_h, fz.add_l = (const _Float16*)add_l, fz.add_sexp = add_sexp;
  fz.mask = mask, fz.mask_float = mask_is_float, fz.mult_amax = (const unsigned*)mult_amax, fz.mask_rows = mask ? mask_rows : 1;
  fz.div_mask = make_fastdiv((int)fz.mask_rows);
  fz.scale = scale, fz.scale_amax = (const unsigned*)scale_amax;
  fz.out_h = (_Float16*)out_h, fz.out_l = (_Float16*)out_l, fz.out_sexp = out_sexp;
  fz.wc_h = fz.wc_l = nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (Co <= 64) return launch_strided<ConvCfg<256, 64, 32, 4, 1>>(g, s1, s2, zero16, (unsigned*)out_amax, st, fz);
  return launch_strided<ConvCfg<128, 128, 32, 2, 2>>(g, s1, s2, zero16, (unsigned*)out_amax, st, fz);
}
