// NHWC / split-fp16 side of the seed-batched reverse sweep (laplace_amd/sweep_nhwc.py): the element-wise VJP that
// PRODUCES split tensors for the convolution kernel (lk_conv.hip), and the G-factor Gram that CONSUMES them.
//
// Reference: the per-layer output gradients of curvlinops' KFACLinearOperator._compute_kfac and their Gram
// G = sum_{n,l,c} g g^T (laplace/curvature/curvlinops.py:87-100).  Stock autograd runs one
// activation / BatchNorm backward kernel per seed and layer; here all seeds of a layer go through one pass.
#include "lk_common.h"

namespace lk {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int scale_exp_for16(float amax) {  // as in lk_conv.hip: amax * 2^s in [2^14, 2^15)
  int be = (int)((__float_as_uint(amax) >> 23) & 0xffu);
  if (be == 0) be = 1;
  int s = 14 - (be - 127);
  return s > 120 ? 120 : s;
}
__device__ __forceinline__ float exp2i16(int s) {
  s = s < -126 ? -126 : (s > 127 ? 127 : s);
  return __uint_as_float((unsigned)(127 + s) << 23);
}

// ---- element-wise VJP, NHWC, split output -------------------------------------------------------------------------------
//   out[s][e] = (g[s][e] + g2[s][e]) * M[e] * scale[e % C]        e < per = B*H*W*C, all S seeds in one pass
// g: fp32 (the convolution kernel's output) with its max|.| in a device word; g2: a split tensor (the cotangent of a
// residual join, produced by this kernel earlier); either may be absent.  The output is split with a scale derived from
// a GUARANTEED bound of max|out| — (max|g| + max|g2|) * max|M| * max|scale|, every factor a device word — so no extra
// pass over the data is needed to find the scale; a loose bound only costs fixed-point range (lk_conv.hip).
template <bool MFLOAT>
__global__ __launch_bounds__(256) void vjp_nhwc_split_kernel(
    const float* __restrict__ g, const unsigned* __restrict__ g_amax, const _Float16* __restrict__ g2h,
    const _Float16* __restrict__ g2l, const int* __restrict__ g2_sexp, const void* __restrict__ m,
    const unsigned* __restrict__ m_amax, const float* __restrict__ scale, const unsigned* __restrict__ scale_amax, int C,
    int S, int64_t per8, _Float16* __restrict__ oh, _Float16* __restrict__ ol, int* __restrict__ out_sexp) {
  float bound = 0.f, inv2 = 0.f;
  if (g) bound += __uint_as_float(g_amax[0]);
  if (g2h) {
    const int s2 = g2_sexp[0];
    bound += exp2i16(15 - s2);  // max|g2| * 2^s2 < 2^15
    inv2 = exp2i16(-s2);
  }
  if (m && MFLOAT && m_amax) bound *= __uint_as_float(m_amax[0]);
  if (scale) bound *= __uint_as_float(scale_amax[0]);
  const int so = scale_exp_for16(bound);
  if (blockIdx.x == 0 && threadIdx.x == 0) out_sexp[0] = so;
  const float sc_out = exp2i16(so);
  const int64_t e8 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e8 >= per8) return;
  float mult[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) mult[j] = sc_out;
  if (m) {
    if (MFLOAT) {
      const float4 a = reinterpret_cast<const float4*>(m)[2 * e8], b = reinterpret_cast<const float4*>(m)[2 * e8 + 1];
      mult[0] *= a.x, mult[1] *= a.y, mult[2] *= a.z, mult[3] *= a.w, mult[4] *= b.x, mult[5] *= b.y, mult[6] *= b.z, mult[7] *= b.w;
    } else {
      const uint2 u = reinterpret_cast<const uint2*>(m)[e8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (!((u.x >> (8 * j)) & 0xffu)) mult[j] = 0.f;
        if (!((u.y >> (8 * j)) & 0xffu)) mult[4 + j] = 0.f;
      }
    }
  }
  if (scale) {
    const int c0 = (int)((e8 * 8) % C);
    const float4 a = *reinterpret_cast<const float4*>(scale + c0), b = *reinterpret_cast<const float4*>(scale + c0 + 4);
    mult[0] *= a.x, mult[1] *= a.y, mult[2] *= a.z, mult[3] *= a.w, mult[4] *= b.x, mult[5] *= b.y, mult[6] *= b.z, mult[7] *= b.w;
  }
#pragma unroll 2
  for (int s = 0; s < S; ++s) {
    const int64_t i8 = (int64_t)s * per8 + e8;
    float v[8];
    if (g) {
      const float4 a = reinterpret_cast<const float4*>(g)[2 * i8], b = reinterpret_cast<const float4*>(g)[2 * i8 + 1];
      v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
    if (g2h) {
      const f16x8 h2 = reinterpret_cast<const f16x8*>(g2h)[i8], l2 = reinterpret_cast<const f16x8*>(g2l)[i8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += ((float)h2[j] + (float)l2[j]) * inv2;
    }
    f16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // The product is made opaque before it is split.  Left to itself hipcc forms the stored h from the fp32-rounded
      // product but the residual from a fused fp16(v * mult - h') with h' = fp16 of the EXACT product (v_fma_mix); the
      // two h differ at rounding ties and the pair then misses the value by a whole fp16 ulp.
      float xs = v[j] * mult[j];
      asm volatile("" : "+v"(xs));
      const _Float16 hh = (_Float16)xs;
      h[j] = hh;
      l[j] = (_Float16)(xs - (float)hh);
    }
    reinterpret_cast<f16x8*>(oh)[i8] = h;
    reinterpret_cast<f16x8*>(ol)[i8] = l;
  }
}

// ---- forward of the NHWC sweep: eval-mode BatchNorm (per-channel affine map) + residual add + activation, one pass ------
//   y[e] = act(x[e] * scale[e % C] + shift[e % C] + addend[e])          e < N * per     act: 0 none, 1 ReLU, 2 tanh
// x: fp32 NHWC [N][per], the forward convolution kernel's output.  Emits what the rest of the step needs in ONE pass: y in
// fp32 (A-factor kernels, later residual joins), the ReLU mask as NHWC bytes (the reverse sweep reads it as is), and the
// split planes of y (next convolution) with ONE SCALE PER IMAGE (see lk_split_images_f16x2), image n scaled from the
// guaranteed bound
//   max|y_n| <= bx_n max|scale| + max|shift| + bound(addend_n),     bx_n = x_amax[n] * x_mul + x_add  >= max|x_n|
// (x_amax: measured max of the convolution's INPUT image, x_mul: l1 norm of its weights, x_add: max|bias| — a bound of
// the convolution's output that needs no pass over it; tanh: 1).  The bound is loose by the l1 slack of ONE layer
// (2^7 - 2^8 for random weights), which only costs fixed-point range; it does not compound, because the kernel also
// MEASURES max|y_n| (y_amax[n], one atomic per workgroup; zeroed by the caller) and that is what the next layer starts from.
// Grid: (chunks of an image, N) — a workgroup never straddles two images.
constexpr int BN_ACT_IT = 4;
__global__ __launch_bounds__(256) void bn_act_fwd_nhwc_kernel(
    const float* __restrict__ x, const unsigned* __restrict__ x_amax, int x_namax, const float* __restrict__ x_mul,
    const float* __restrict__ x_add, const float* __restrict__ scale, const float* __restrict__ shift,
    const unsigned* __restrict__ scale_amax, const unsigned* __restrict__ shift_amax, const float* __restrict__ addend,
    const float* __restrict__ addend_bound, int addend_nbound, int act, int C, int64_t per8, float* __restrict__ y,
    unsigned char* __restrict__ mask, _Float16* __restrict__ yh, _Float16* __restrict__ yl, int* __restrict__ y_sexp,
    float* __restrict__ y_bound, unsigned* __restrict__ y_amax) {
  const int n = blockIdx.y;
  float bx = __uint_as_float(x_amax[n < x_namax ? n : x_namax - 1]);
  if (x_mul) bx *= x_mul[0];
  if (x_add) bx += x_add[0];
  float bound = bx * __uint_as_float(scale_amax[0]) + __uint_as_float(shift_amax[0]);
  if (addend) bound += addend_bound[n < addend_nbound ? n : addend_nbound - 1];
  if (act == 2) bound = 1.f;
  const int so = scale_exp_for16(bound);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    y_sexp[n] = so;
    y_bound[n] = bound;
  }
  const float sc_out = exp2i16(so);
  unsigned vmax = 0;
  // a workgroup walks BN_ACT_IT x 256 float8s of its image (the header's dependent loads and the atomic below are paid once
  // per 8192 elements: one atomic per 2048 elements — 4096 per launch into the 128 words of a minibatch, i.e. four cache
  // lines of one L2 channel — cost the layer-1 launches a third of their time)
  for (int64_t i8 = (int64_t)blockIdx.x * (256 * BN_ACT_IT) + threadIdx.x, it = 0; it < BN_ACT_IT && i8 < per8; ++it, i8 += 256) {
    const int64_t e8 = (int64_t)n * per8 + i8;
    const int c0 = (int)((i8 * 8) % C);  // (per % C == 0: an image starts at channel 0)
    const float4 s0 = *reinterpret_cast<const float4*>(scale + c0), s1 = *reinterpret_cast<const float4*>(scale + c0 + 4);
    const float4 t0 = *reinterpret_cast<const float4*>(shift + c0), t1 = *reinterpret_cast<const float4*>(shift + c0 + 4);
    const float4 a = reinterpret_cast<const float4*>(x)[2 * e8], b = reinterpret_cast<const float4*>(x)[2 * e8 + 1];
    float v[8] = {a.x * s0.x + t0.x, a.y * s0.y + t0.y, a.z * s0.z + t0.z, a.w * s0.w + t0.w,
                  b.x * s1.x + t1.x, b.y * s1.y + t1.y, b.z * s1.z + t1.z, b.w * s1.w + t1.w};
    if (addend) {
      const float4 p = reinterpret_cast<const float4*>(addend)[2 * e8], q = reinterpret_cast<const float4*>(addend)[2 * e8 + 1];
      v[0] += p.x, v[1] += p.y, v[2] += p.z, v[3] += p.w, v[4] += q.x, v[5] += q.y, v[6] += q.z, v[7] += q.w;
    }
    if (act == 1) {
      unsigned m0 = 0, m1 = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] = fmaxf(v[j], 0.f);
        if (v[j] > 0.f) (j < 4 ? m0 : m1) |= 1u << (8 * (j & 3));
      }
      if (mask) reinterpret_cast<uint2*>(mask)[e8] = make_uint2(m0, m1);
    } else if (act == 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tanhf(v[j]);
    }
    reinterpret_cast<float4*>(y)[2 * e8] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(y)[2 * e8 + 1] = make_float4(v[4], v[5], v[6], v[7]);
#pragma unroll
    for (int j = 0; j < 8; ++j) vmax = max(vmax, __float_as_uint(v[j]) & 0x7fffffffu);
    if (yh) {
      f16x8 h, l;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float xs = v[j] * sc_out;
        asm volatile("" : "+v"(xs));  // see vjp_nhwc_split_kernel: split an opaque value
        const _Float16 hh = (_Float16)xs;
        h[j] = hh;
        l[j] = (_Float16)(xs - (float)hh);
      }
      reinterpret_cast<f16x8*>(yh)[e8] = h;
      reinterpret_cast<f16x8*>(yl)[e8] = l;
    }
  }
  if (y_amax) {  // (uniform) measured max|y_n|: the waves' maxima meet in LDS, one atomic per workgroup
    __shared__ unsigned wave_max[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = max(vmax, (unsigned)__shfl_xor((int)vmax, off, 64));
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      vmax = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
      if (vmax) atomicMax(y_amax + n, vmax);
    }
  }
}

// ---- split NHWC cotangent -> fp32, position-contiguous, sample-major:  out[b][s][c][l] = x[s*B + b][l][c] -----------
// (what the Jacobian-free predictive kernels read: `u [B, C_out, Do, L]`); 32 x 32 tiles transposed through LDS.
__global__ __launch_bounds__(256) void unsplit_transpose_kernel(const _Float16* __restrict__ xh,
                                                                const _Float16* __restrict__ xl,
                                                                const int* __restrict__ sexp, int S, int B, int L, int C,
                                                                float* __restrict__ out) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z, s = n / B, b = n - s * B;
  const int c0 = blockIdx.x * 32, l0 = blockIdx.y * 32;
  const float inv = exp2i16(-sexp[0]);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int l = l0 + i, c = c0 + tx;
    float v = 0.f;
    if (l < L && c < C) {
      const int64_t e = ((int64_t)n * L + l) * C + c;
      v = ((float)xh[e] + (float)xl[e]) * inv;
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, l = l0 + tx;
    if (l < L && c < C) out[(((int64_t)b * S + s) * C + c) * L + l] = tile[tx][i];
  }
}

// ---- Gram of a split tensor:  ws partials of  X^T X,  X = planes [R][C] (rows = (seed, sample, pixel), C contiguous) -----
// k (rows) is the STRIDED direction of both MFMA operands, so fragments come out of LDS through the transposing read
// ds_read_b64_tr_b16: a 16-lane group hands in the addresses of a [4 k][16 channel] block (lane r: row k0 + r/4,
// channels c0 + 4 (r%4) ..+3) and lane r receives channel c0 + r for k0..k0+3 — two of them make the 8 consecutive k of
// one lane of v_mfma_f32_32x32x16_f16, for the A (row-channel) and the B (column-channel) operand alike.
// The stage image [rows][NB channels] is filled by LDS-DMA (lane-linear), with the 64-byte chunks of a row XOR-swizzled by
// the row so that the four rows a half-wave reads together sit in different banks.
// Workgroup = one (bi <= bj) pair of NB-channel blocks and one slice of the rows:
//   NB = 64  (C = 64):  the four waves split the k16 steps of a stage between them (the tensor is huge, the output tiny);
//   NB = 128 (C >= 128): 2 x 2 waves, each a 64 x 64 quadrant; lower-triangle quadrant / tiles of diagonal blocks idle.
// Every wave writes its partial tiles to the workspace; gram16_reduce_kernel sums them in a fixed order.
template <int NB, int BKR, int KW>  // KW = waves that split k inside the workgroup (4 or 1)
struct Gram16Cfg {
  static constexpr int PITCH = NB * 2;                   // bytes per row of a plane image
  static constexpr int PLANE = BKR * PITCH;              // bytes
  static constexpr int PANEL = 2 * PLANE;                // h + l
  static constexpr int STAGE = 2 * PANEL;                // row panel + column panel
  static constexpr int SLOTS = BKR * NB / 8;             // 16-byte slots per plane image
  static constexpr int LD = SLOTS / 256;                 // LDS-DMA instructions per thread, plane and panel
  static constexpr int TW = KW == 4 ? NB / 32 : NB / 64; // MFMA tiles per wave and side
  static_assert(SLOTS % 256 == 0, "whole instructions");
};

template <int NB>
__device__ __forceinline__ int gram16_swz(int row) {  // XOR on the 64-byte chunk index of a row
  return NB == 64 ? ((row >> 1) & 1) : (row & 3);
}

typedef __attribute__((address_space(3))) void lds_void16;
typedef __attribute__((address_space(1))) const void gbl_void16;

// TNP = true: the "pixel-pair" products of a 3x3 convolution's A factor (lk_conv3x3_pixpair_*, lk_gram.hip): the workgroup
// takes its two column panels and its output block from a table, runs over ALL rows (the images of a few stacked
// minibatches) and adds alpha 2^(-2 sexp) X_A^T X_B into the block it alone owns:  blocks[off + r * ldc + c] += ...
struct Gram16Tnp {
  const int* tiles;   // [ntiles][3] = column of the A panel, column of the B panel, output offset (floats)
  int ldc;
  const int* sexp;
  float alpha;
  float* blocks;
};

template <int NB, int BKR, int KW, bool TNP>
__global__ __launch_bounds__(256) void gram16_kernel(const _Float16* __restrict__ Xh, const _Float16* __restrict__ Xl,
                                                     int64_t R, int64_t C, int64_t rows_per_split,
                                                     const _Float16* __restrict__ zero16, float* __restrict__ ws,
                                                     int nbc, const Gram16Tnp tp) {
  using G = Gram16Cfg<NB, BKR, KW>;
  constexpr int TW = G::TW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // block pair (bi <= bj) from the linear index
  int bi = 0, bj = 0;
  int64_t colA = 0, colB = 0, out_off = 0;
  if constexpr (TNP) {
    // consecutive table entries share their A panel (same pixel, the 13 shifts): give every XCD (block id % 8) a
    // contiguous range of the table so that the sharing happens inside one L2
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    const int q = nblk / 8, r = nblk % 8, x = bid % 8, j = bid / 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    colA = tp.tiles[3 * bid], colB = tp.tiles[3 * bid + 1], out_off = tp.tiles[3 * bid + 2];
  } else {
    int p = blockIdx.x;
    while (p >= nbc - bi) p -= nbc - bi, ++bi;
    bj = bi + p;
    colA = (int64_t)bi * NB, colB = (int64_t)bj * NB;
  }
  const bool diag = !TNP && bi == bj;
  const int64_t r0 = TNP ? 0 : (int64_t)blockIdx.y * rows_per_split;
  const int64_t r1 = TNP ? R : (r0 + rows_per_split < R ? r0 + rows_per_split : R);
  const int nstage = r1 > r0 ? (int)((r1 - r0 + BKR - 1) / BKR) : 0;

  // staging context: slot -> (row, physical 16-byte slot); the source is the logical slot of that row
  int st_row[G::LD], st_col[G::LD];
#pragma unroll
  for (int i = 0; i < G::LD; ++i) {
    const int slot = i * 256 + tid;
    const int row = slot / (NB / 8), pq = slot % (NB / 8);
    st_row[i] = row;
    st_col[i] = (pq ^ (gram16_swz<NB>(row) << 2)) * 8;  // logical channel offset inside the panel
  }
  auto stage = [&](int s, int buf) {
    char* base = smem + buf * G::STAGE;
    const int64_t rb = r0 + (int64_t)s * BKR;
#pragma unroll
    for (int pnl = 0; pnl < 2; ++pnl) {
      if (pnl == 1 && diag) break;
      const int64_t cb = pnl ? colB : colA;
#pragma unroll
      for (int i = 0; i < G::LD; ++i) {
        const int64_t r = rb + st_row[i];
        const bool ok = r < r1;
        const int64_t e = r * C + cb + st_col[i];
        char* d = base + pnl * G::PANEL + (i * 256 + wave * 64) * 16;
        __builtin_amdgcn_global_load_lds((gbl_void16*)(ok ? Xh + e : zero16), (lds_void16*)d, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void16*)(ok ? Xl + e : zero16), (lds_void16*)(d + G::PLANE), 16, 0, 0);
      }
    }
  };

  // which tiles this wave owns
  const int wr = KW == 4 ? 0 : (wave >> 1), wc = KW == 4 ? 0 : (wave & 1);  // quadrant of the NB x NB block
  const bool wave_active = !(KW == 1 && diag && wr > wc);
  f32x16 acc[TW][TW];
#pragma unroll
  for (int a = 0; a < TW; ++a)
#pragma unroll
    for (int b = 0; b < TW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // fragment geometry of this lane
  const int grp = lane >> 4, r16 = lane & 15;
  const int f_row = (grp >> 1) * 8 + (r16 >> 2);        // + j * 4 + k16 * 16
  const int f_col = (grp & 1) * 16 + (r16 & 3) * 4;     // + tile * 32 (+ quadrant * 64)

  // The fragment reads are inline assembly ON PURPOSE: hipcc knows that global_load_lds writes LDS and guards every LDS read it
  // can see behind one with `s_waitcnt vmcnt(0)` — with the builtin the next stage's requests, issued right in front of this
  // stage's products, were waited for on the spot (ISA checked) and a stage cost one round trip of the request path (~2500
  // clocks) whatever its size.  The stage hand-over (vmcnt(0) + barrier at the top of the loop) is what orders them.
  auto load_frag = [&](const char* plane, int k16, int col0) -> f16x8 {
    s16x4 v[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = k16 * 16 + f_row + j * 4;
      const int col = col0 + f_col;
      const int off = row * G::PITCH + (((col >> 5) ^ gram16_swz<NB>(row)) << 6) + (col & 31) * 2;
      const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(plane + off);
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v[j]) : "v"(addr));
    }
    const f16x4 f0 = __builtin_bit_cast(f16x4, v[0]), f1 = __builtin_bit_cast(f16x4, v[1]);
    return f16x8{f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
  };

  // TNP: the block this workgroup adds into is requested NOW, so that the read of the read-modify-write runs under the K
  // loop instead of after it (the tiles are short — 8 to 16 stages — and a dependent 16 KB read at their end was a third
  // of a tile's time)
  constexpr int NOLD = TNP ? (KW == 4 ? TW * TW * 4 : TW * TW * 16) : 1;
  float old[NOLD];
  if constexpr (TNP) {
    const int lr_ = lane & 31, lh_ = lane >> 5;
    if (KW == 4) {
#pragma unroll
      for (int i = 0; i < NOLD; ++i) {
        const int e = tid + 256 * i, t = e >> 10, a = t / TW, b = t % TW, row = (e >> 5) & 31, col = e & 31;
        old[i] = tp.blocks[out_off + (int64_t)(a * 32 + row) * tp.ldc + b * 32 + col];
      }
    } else {
#pragma unroll
      for (int a = 0; a < TW; ++a)
#pragma unroll
        for (int b = 0; b < TW; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh_;
            const int col = wc * 64 + b * 32 + lr_;
            old[(a * TW + b) * 16 + r] = tp.blocks[out_off + (int64_t)row * tp.ldc + col];
          }
    }
  }
  if (nstage > 0) stage(0, 0);
  for (int s = 0; s < nstage; ++s) {
    const int buf = s & 1;
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    __syncthreads();
    if (s + 1 < nstage) stage(s + 1, buf ^ 1);
    if (!wave_active) continue;
    const char* pa = smem + buf * G::STAGE;
    const char* pb = diag ? pa : pa + G::PANEL;
#pragma unroll
    for (int kk = 0; kk < (KW == 4 ? 1 : BKR / 16); ++kk) {
      const int k16 = KW == 4 ? wave : kk;
      f16x8 ah[TW], al[TW], bh[TW], bl[TW];
#pragma unroll
      for (int a = 0; a < TW; ++a) {
        ah[a] = load_frag(pa, k16, wr * 64 + a * 32);
        al[a] = load_frag(pa + G::PLANE, k16, wr * 64 + a * 32);
      }
#pragma unroll
      for (int b = 0; b < TW; ++b) {
        bh[b] = load_frag(pb, k16, wc * 64 + b * 32);
        bl[b] = load_frag(pb + G::PLANE, k16, wc * 64 + b * 32);
      }
      if constexpr (TW == 2)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[0]), "+v"(al[0]), "+v"(bh[0]), "+v"(bl[0]), "+v"(ah[1]), "+v"(al[1]), "+v"(bh[1]), "+v"(bl[1]));
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[0]), "+v"(al[0]), "+v"(bh[0]), "+v"(bl[0]));
#pragma unroll
      for (int a = 0; a < TW; ++a)
#pragma unroll
        for (int b = 0; b < TW; ++b) {
          if (diag && wr == wc && a > b) continue;  // lower-triangle tile of a diagonal quadrant
          f32x16 c = acc[a][b];
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[b], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[b], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b], c, 0, 0, 0);
          acc[a][b] = c;
        }
    }
  }

  // partial block of this workgroup: ws[(split * npairs + pair)][NB][NB].  KW = 4: the four waves hold partial sums
  // of the SAME tiles (they split k); they are added in wave order through LDS (the stage buffers are free now).
  const int lr = lane & 31, lh = lane >> 5;
  float* blk = TNP ? nullptr : ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (NB * NB);
  if (KW == 4) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [wave][tile (a, b)][32 x 32]
#pragma unroll
    for (int a = 0; a < TW; ++a)
#pragma unroll
      for (int b = 0; b < TW; ++b) {
        if (!TNP && a > b) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
          red[((wave * TW + a) * TW + b) * 1024 + row * 32 + lr] = acc[a][b][r];
        }
      }
    __syncthreads();
    float tscale = 0.f;
    if constexpr (TNP) {
      const float inv = exp2i16(-tp.sexp[0]);
      tscale = tp.alpha * inv * inv;
    }
#pragma unroll
    for (int i = 0; i < TW * TW * 4; ++i) {
      const int e = tid + 256 * i;
      const int t = e >> 10, a = t / TW, b = t % TW;
      if (!TNP && a > b) continue;
      const float v = ((red[e] + red[TW * TW * 1024 + e]) + red[2 * TW * TW * 1024 + e]) + red[3 * TW * TW * 1024 + e];
      const int row = (e >> 5) & 31, col = e & 31;
      if constexpr (TNP) tp.blocks[out_off + (int64_t)(a * 32 + row) * tp.ldc + b * 32 + col] = old[i] + tscale * v;
      else blk[(a * 32 + row) * NB + b * 32 + col] = v;
    }
    return;
  }
  if (!wave_active) return;
  float tscale = 0.f;
  if constexpr (TNP) {
    const float inv = exp2i16(-tp.sexp[0]);
    tscale = tp.alpha * inv * inv;
  }
#pragma unroll
  for (int a = 0; a < TW; ++a)
#pragma unroll
    for (int b = 0; b < TW; ++b) {
      if (diag && wr == wc && a > b) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int col = wc * 64 + b * 32 + lr;
        if constexpr (TNP) tp.blocks[out_off + (int64_t)row * tp.ldc + col] = old[(a * TW + b) * 16 + r] + tscale * acc[a][b][r];
        else blk[row * NB + col] = acc[a][b][r];
      }
    }
}

// G[upper tiles] += alpha * 2^(-2 sexp) * sum_partials, in a FIXED order: a workgroup owns 32 consecutive elements of one
// tile row; its 8 lane groups each sum every 8th partial (coalesced 128-byte reads), then the 8 sums are added in order.
template <int NB>
__global__ __launch_bounds__(256) void gram16_reduce_kernel(const float* __restrict__ ws, int nparts, int npairs, int nbc,
                                                            int C, const int* __restrict__ sexp, float alpha,
                                                            float* __restrict__ Gm) {
  __shared__ float red[8][32];
  const int pair = blockIdx.y;
  int bi = 0, bj = 0;
  {
    int p = pair;
    while (p >= nbc - bi) p -= nbc - bi, ++bi;
    bj = bi + p;
  }
  const int el = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;  // element of the NB x NB block; the 32 of a workgroup share one tile row
  const int row = e / NB, col = e % NB;
  if (bi == bj && (row >> 5) > (col >> 5)) return;  // tile below the diagonal of a diagonal block: never computed
  float s = 0.f;
  for (int p = pl; p < nparts; p += 8) s += ws[((int64_t)p * npairs + pair) * (NB * NB) + e];
  red[pl][el] = s;
  __syncthreads();
  if (pl == 0) {
    float t = red[0][el];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][el];
    const float inv = exp2i16(-sexp[0]);
    Gm[(int64_t)(bi * NB + row) * C + bj * NB + col] += alpha * (t * inv) * inv;
  }
}

// The same reduction for the split-K slabs of lk_gram_tn_f16x2 (a few dozen partials of 16 / 64 KB each): a workgroup owns
// 256 consecutive floats of one block; wave g sums the partials p = g, g + 4, ... as float4s (1 KB contiguous per wave and
// partial, where the kernel above reads 128 bytes), then the four sums are added in wave order.  Fixed order, no atomics.
template <int NB>
__global__ __launch_bounds__(256) void gram16_reduce4_kernel(const float* __restrict__ ws, int nparts, int npairs, int nbc,
                                                             int C, const int* __restrict__ sexp, float alpha,
                                                             float* __restrict__ Gm) {
  __shared__ float4 red[4][64];
  const int pair = blockIdx.y;
  int bi = 0, bj = 0;
  {
    int p = pair;
    while (p >= nbc - bi) p -= nbc - bi, ++bi;
    bj = bi + p;
  }
  const int q = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int e = blockIdx.x * 256 + 4 * q;  // first of this thread's four elements of the NB x NB block (one row: NB % 4 == 0)
  const int row = e / NB, col = e % NB;
  const bool live = !(bi == bj && (row >> 5) > (col >> 5));  // (tiles below the diagonal of a diagonal block are never computed)
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* src = ws + (int64_t)pair * (NB * NB) + e;
  // (eight partials requested before the first is added: the 16 - 64 workgroups of this launch are latency-bound, one
  // dependent load per partial was 34 us for 512 of them; the order of the additions is unchanged)
  const int64_t pstride = (int64_t)npairs * (NB * NB);
  int p = g;
  for (; live && p + 28 < nparts; p += 32) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(src + (int64_t)(p + 4 * u) * pstride);
#pragma unroll
    for (int u = 0; u < 8; ++u) s.x += v[u].x, s.y += v[u].y, s.z += v[u].z, s.w += v[u].w;
  }
  for (; live && p < nparts; p += 4) {
    const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)p * pstride);
    s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
  }
  red[g][q] = s;
  __syncthreads();
  if (g == 0 && live) {
    float4 t = red[0][q];
#pragma unroll
    for (int k = 1; k < 4; ++k) t.x += red[k][q].x, t.y += red[k][q].y, t.z += red[k][q].z, t.w += red[k][q].w;
    const float inv = exp2i16(-sexp[0]);
    const float f = alpha * inv * inv;
    float4* dst = reinterpret_cast<float4*>(Gm + (int64_t)(bi * NB + row) * C + bj * NB + col);
    float4 o = *dst;
    o.x += f * t.x, o.y += f * t.y, o.z += f * t.z, o.w += f * t.w;
    *dst = o;
  }
}

}  // namespace lk

using namespace lk;

extern "C" int lk_vjp_nhwc_split_f16x2(const float* g, const unsigned* g_amax, const void* g2_h, const void* g2_l,
                                       const int* g2_sexp, const void* m, int m_is_float, const unsigned* m_amax,
                                       const float* scale, const unsigned* scale_amax, int64_t C, int64_t S, int64_t per,
                                       void* out_h, void* out_l, int* out_sexp, void* stream) {
  LK_REQUIRE((g || g2_h) && out_h && out_l && out_sexp && S >= 0 && per >= 0, "lk_vjp_nhwc_split_f16x2: bad arguments");
  LK_REQUIRE(!g || g_amax, "lk_vjp_nhwc_split_f16x2: the fp32 addend needs its max|.| word");
  LK_REQUIRE(!g2_h || (g2_l && g2_sexp), "lk_vjp_nhwc_split_f16x2: incomplete split addend");
  LK_REQUIRE(!scale || (scale_amax && C >= 8 && C % 8 == 0 && per % C == 0), "lk_vjp_nhwc_split_f16x2: scale needs C % 8 == 0");
  LK_REQUIRE(per % 8 == 0 && S < (1 << 30), "lk_vjp_nhwc_split_f16x2: per % 8 == 0");
  if (S == 0 || per == 0) return LK_OK;
  const int64_t per8 = per / 8, nb = (per8 + 255) / 256;
  LK_REQUIRE(nb < (1ll << 31), "lk_vjp_nhwc_split_f16x2: grid too large");
  hipStream_t st = (hipStream_t)stream;
  if (m && m_is_float)
    hipLaunchKernelGGL(vjp_nhwc_split_kernel<true>, dim3((unsigned)nb), dim3(256), 0, st, g, g_amax, (const _Float16*)g2_h,
                       (const _Float16*)g2_l, g2_sexp, m, m_amax, scale, scale_amax, (int)(scale ? C : 8), (int)S, per8,
                       (_Float16*)out_h, (_Float16*)out_l, out_sexp);
  else
    hipLaunchKernelGGL(vjp_nhwc_split_kernel<false>, dim3((unsigned)nb), dim3(256), 0, st, g, g_amax, (const _Float16*)g2_h,
                       (const _Float16*)g2_l, g2_sexp, m, m_amax, scale, scale_amax, (int)(scale ? C : 8), (int)S, per8,
                       (_Float16*)out_h, (_Float16*)out_l, out_sexp);
  return check_launch("vjp_nhwc_split_kernel");
}

extern "C" int lk_bn_act_fwd_nhwc_f16x2(const float* x, const unsigned* x_amax, int64_t x_namax, const float* x_mul,
                                       const float* x_add, const float* scale, const float* shift,
                                       const unsigned* scale_amax, const unsigned* shift_amax, const float* addend,
                                       const float* addend_bound, int64_t addend_nbound, int act, int64_t C, int64_t N,
                                       int64_t per, float* y, void* mask, void* y_h, void* y_l, int* y_sexp, float* y_bound,
                                       unsigned* y_amax, void* stream) {
  LK_REQUIRE(x && x_amax && scale && shift && scale_amax && shift_amax && y && y_sexp && y_bound && per >= 0 && N >= 0,
             "lk_bn_act_fwd_nhwc_f16x2: null pointer");
  LK_REQUIRE(C >= 8 && C % 8 == 0 && per % C == 0 && act >= 0 && act <= 2, "lk_bn_act_fwd_nhwc_f16x2: C % 8 == 0, act in 0..2");
  LK_REQUIRE(x_namax == 1 || x_namax == N, "lk_bn_act_fwd_nhwc_f16x2: x_amax has 1 or N words");
  LK_REQUIRE(!addend || (addend_bound && (addend_nbound == 1 || addend_nbound == N)),
             "lk_bn_act_fwd_nhwc_f16x2: the addend needs its bound (1 or N words)");
  LK_REQUIRE((y_h == nullptr) == (y_l == nullptr), "lk_bn_act_fwd_nhwc_f16x2: both planes or none");
  LK_REQUIRE(N <= 65535, "lk_bn_act_fwd_nhwc_f16x2: at most 65535 images");
  if (per == 0 || N == 0) return LK_OK;
  const int64_t per8 = per / 8, nb = (per8 + 256 * BN_ACT_IT - 1) / (256 * BN_ACT_IT);
  LK_REQUIRE(nb < (1ll << 31), "lk_bn_act_fwd_nhwc_f16x2: grid too large");
  hipLaunchKernelGGL(bn_act_fwd_nhwc_kernel, dim3((unsigned)nb, (unsigned)N), dim3(256), 0, (hipStream_t)stream, x, x_amax,
                     (int)x_namax, x_mul, x_add, scale, shift, scale_amax, shift_amax, addend, addend_bound,
                     (int)(addend ? addend_nbound : 1), act, (int)C, per8, y, (unsigned char*)mask, (_Float16*)y_h,
                     (_Float16*)y_l, y_sexp, y_bound, y_amax);
  return check_launch("bn_act_fwd_nhwc_kernel");
}

extern "C" int lk_unsplit_transpose_f32(const void* x_h, const void* x_l, const int* sexp, int64_t S, int64_t B, int64_t L,
                                       int64_t C, float* out, void* stream) {
  LK_REQUIRE(x_h && x_l && sexp && out && S >= 0 && B >= 0 && L >= 0 && C >= 0 && S * B <= 65535,
             "lk_unsplit_transpose_f32: bad arguments (S * B <= 65535)");
  if (S * B == 0 || L == 0 || C == 0) return LK_OK;
  hipLaunchKernelGGL(unsplit_transpose_kernel, dim3((unsigned)((C + 31) / 32), (unsigned)((L + 31) / 32), (unsigned)(S * B)),
                     dim3(256), 0, (hipStream_t)stream, (const _Float16*)x_h, (const _Float16*)x_l, sexp, (int)S, (int)B,
                     (int)L, (int)C, out);
  return check_launch("unsplit_transpose_kernel");
}

namespace {
struct Gram16Plan {
  int nb, nbc, npairs, kw, bkr;
  int64_t nsplit, rows_per_split, nparts;
};
Gram16Plan gram16_plan(int64_t R, int64_t C) {
  Gram16Plan p;
  p.nb = C == 64 ? 64 : 128;
  p.kw = C == 64 ? 4 : 1;
  p.bkr = C == 64 ? 64 : 32;
  p.nbc = (int)(C / p.nb);
  p.npairs = p.nbc * (p.nbc + 1) / 2;
  // two workgroups per CU, slices of whole stages, at least 8 stages per slice
  int64_t want = (512 + p.npairs - 1) / p.npairs;
  int64_t stages = (R + p.bkr - 1) / p.bkr;
  int64_t per = (stages + want - 1) / want;
  if (per < 8) per = 8;
  p.rows_per_split = per * p.bkr;
  p.nsplit = (R + p.rows_per_split - 1) / p.rows_per_split;
  if (p.nsplit < 1) p.nsplit = 1;
  p.nparts = p.nsplit;  // (the k-waves of a workgroup are summed inside it)
  return p;
}
}  // namespace

extern "C" size_t lk_gram_tn_f16x2_workspace_bytes(int64_t R, int64_t C) {
  if (C < 64 || (C != 64 && C % 128)) return 0;
  const Gram16Plan p = gram16_plan(R, C);
  return (size_t)p.nparts * p.npairs * p.nb * p.nb * sizeof(float);
}

// G[C][C] (upper 32x32 tiles) += alpha * X^T X for the split tensor X [R][C]; C = 64 or a multiple of 128.
extern "C" int lk_gram_tn_f16x2(const void* x_h, const void* x_l, const int* sexp, int64_t R, int64_t C, float alpha,
                                float* Gm, const void* zero16, void* ws, size_t ws_bytes, void* stream) {
  LK_REQUIRE(x_h && x_l && sexp && Gm && zero16 && ws && R >= 0, "lk_gram_tn_f16x2: null pointer");
  LK_REQUIRE(C == 64 || (C >= 128 && C % 128 == 0 && C <= 4096), "lk_gram_tn_f16x2: C must be 64 or a multiple of 128");
  if (R == 0) return LK_OK;
  const Gram16Plan p = gram16_plan(R, C);
  LK_REQUIRE(ws_bytes >= lk_gram_tn_f16x2_workspace_bytes(R, C), "lk_gram_tn_f16x2: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)p.npairs, (unsigned)p.nsplit);
  if (C == 64) {
    using Gc = Gram16Cfg<64, 64, 4>;
    hipLaunchKernelGGL((gram16_kernel<64, 64, 4, false>), grid, dim3(256), 2 * Gc::STAGE, st, (const _Float16*)x_h,
                       (const _Float16*)x_l, R, C, p.rows_per_split, (const _Float16*)zero16, (float*)ws, p.nbc, Gram16Tnp{});
  } else {
    using Gc = Gram16Cfg<128, 32, 1>;
    hipLaunchKernelGGL((gram16_kernel<128, 32, 1, false>), grid, dim3(256), 2 * Gc::STAGE, st, (const _Float16*)x_h,
                       (const _Float16*)x_l, R, C, p.rows_per_split, (const _Float16*)zero16, (float*)ws, p.nbc, Gram16Tnp{});
  }
  int rc = check_launch("gram16_kernel");
  if (rc != LK_OK) return rc;
  dim3 rgrid((unsigned)(p.nb * p.nb / 256), (unsigned)p.npairs);
  if (C == 64)
    hipLaunchKernelGGL(gram16_reduce4_kernel<64>, rgrid, dim3(256), 0, st, (const float*)ws, (int)p.nparts, p.npairs, p.nbc,
                       (int)C, sexp, alpha, Gm);
  else
    hipLaunchKernelGGL(gram16_reduce4_kernel<128>, rgrid, dim3(256), 0, st, (const float*)ws, (int)p.nparts, p.npairs, p.nbc,
                       (int)C, sexp, alpha, Gm);
  return check_launch("gram16_reduce4_kernel");
}

namespace lk {
// ---- pixel-pair blocks of a 64-channel map, all 13 shifts of a pixel in ONE workgroup -------------------------------------
// gram16_kernel<64, 64, 4, true> gives every (pixel, shift) block its own workgroup, which stages the pixel's panel and the
// shifted pixel's panel: 26 panels per pixel, 6.8 GB through L2 per launch of 8 stacked minibatches.  Here a workgroup owns a
// pixel q: per stage of 16 images it stages q's panel ONCE and the (up to) 12 other panels of its shifts — 13 panels for 13
// blocks —, and its eight waves hold the 13 blocks' accumulators: wave = (32 x 32 tile of the 64 x 64 block) x (shifts 0 .. 6 |
// 7 .. 12).  Same stage image, swizzle and transposing fragment reads as gram16_kernel; a ring of three stages (two in flight),
// the requests of a stage issued one per shift between the shifts' products.  Measured (tools/pixpair_bench.py and its
// -DLK_PIX13_ABLATE builds, B = 1024): 536 us per launch = 235 of products + 180 of staging + 107 of block updates, which add
// rather than overlap (the CUs' joint ingest from L2, ~18 TB/s, and the matrix pipe share the power budget); 5 % less than one
// workgroup per block, 0.62 against 0.66 ms per c4 step for the family.
struct Pix13Args {
  const _Float16 *xh, *xl;
  const _Float16* zero16;
  const int* slots;   // [H W][13] block slot of (pixel, shift) or -1 (lk_conv3x3_pixpair_tables)
  const int* sexp;
  float* blocks;
  float alpha;
  int R, H, W;        // images, map
};

__device__ __forceinline__ f16x8 pix13_frag(unsigned plane, int lane, int col0) {  // (assembly: see gram16_kernel's load_frag)
  const int grp = lane >> 4, r16 = lane & 15;
  const int f_row = (grp >> 1) * 8 + (r16 >> 2), col = col0 + (grp & 1) * 16 + (r16 & 3) * 4;
  s16x4 v[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = f_row + j * 4;
    const unsigned off = plane + row * 128 + (((col >> 5) ^ gram16_swz<64>(row)) << 6) + (col & 31) * 2;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v[j]) : "v"(off));
  }
  const f16x4 f0 = __builtin_bit_cast(f16x4, v[0]), f1 = __builtin_bit_cast(f16x4, v[1]);
  return f16x8{f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
}

#ifndef LK_PIX13_ABLATE  // development builds, timing only: 1 no block update, 2 no staging, 4 no fragment reads / MFMAs
#define LK_PIX13_ABLATE 0
#endif
__global__ __launch_bounds__(512) void pixpair13_kernel(const Pix13Args p) {
  constexpr int C = 64, BKR = 16, PLANE = BKR * C * 2, PANEL = 2 * PLANE, STAGE = 13 * PANEL;  // 2 KB, 4 KB, 52 KB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int q;
  {  // neighbouring pixels share panels: every XCD (block id % 8) takes a contiguous range of pixels
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int qq = nblk / 8, r = nblk % 8, x = bid % 8, j = bid / 8;
    q = (x < r ? x * (qq + 1) : r * (qq + 1) + (x - r) * qq) + j;
  }
  // shift h: (dy, dx) = (0, 0 .. 2), (1, -2 .. 2), (2, -2 .. 2) — lk_gram.hip's kHalfDy / kHalfDx
  auto panel_col = [&](int h) {  // first column of the panel of pixel q + shift h (scalar arithmetic: h is uniform)
    const int dy = h < 3 ? 0 : (h < 8 ? 1 : 2), dx = h < 3 ? h : (h < 8 ? h - 5 : h - 10);
    return (q + dy * p.W + dx) * C;
  };
  unsigned valid = 0;
#pragma unroll
  for (int h = 0; h < 13; ++h)
    if (p.slots[q * 13 + h] >= 0) valid |= 1u << h;
  valid = (unsigned)__builtin_amdgcn_readfirstlane((int)valid);
  const int g = wave >> 2;  // shift group of this wave: shifts 7 g .. 7 g + 6
  int myslot[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) myslot[i] = g * 7 + i < 13 ? __builtin_amdgcn_readfirstlane(p.slots[q * 13 + (g * 7 + i < 13 ? g * 7 + i : 0)]) : -1;
  const int64_t ld = (int64_t)p.H * p.W * C;
  const int nstage = (p.R + BKR - 1) / BKR;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // staging: round i (0 .. 6) of 512 lanes = planes 4 i .. 4 i + 3 (panels 2 i, 2 i + 1); a wave = half a plane (64 slots of 16 B)
  const int s_half = wave & 1, s_pl = wave >> 1;            // plane 4 i + s_pl: panel 2 i + (s_pl >> 1), h / l = s_pl & 1
  const int s_row = s_half * 8 + (lane >> 3);
  const int s_col = ((lane & 7) ^ (gram16_swz<64>(s_row) << 2)) * 8;
  // (every wave issues exactly SEVEN requests per stage — a panel outside the map stages zeros, the half round 6 that has no panel
  //  repeats panel 12 (same bytes, same place) — so that the counted wait below is the same number for everybody)
  auto stage_one = [&](int s, int buf, int i) {  // request i (0 .. 6) of this wave for stage s
    const int64_t r = (int64_t)s * BKR + s_row;
    const bool ok = r < p.R;
    const int64_t e0 = r * ld + s_col;
    int pnl = 2 * i + (s_pl >> 1), hl = s_pl & 1;
    if (pnl > 12) pnl = 12;
    const bool live = ok && ((valid >> pnl) & 1u);
    const _Float16* src = live ? (hl ? p.xl : p.xh) + e0 + panel_col(pnl) : p.zero16;
    const unsigned dst = lds0 + buf * STAGE + pnl * PANEL + hl * PLANE + s_half * 1024;
    __builtin_amdgcn_global_load_lds((gbl_void16*)src, (lds_void16*)(uintptr_t)dst, 16, 0, 0);
  };
  auto stage = [&](int s, int buf) {
#pragma unroll
    for (int i = 0; i < 7; ++i) stage_one(s, buf, i);
  };

  const int t_a = (wave >> 1) & 1, t_b = wave & 1;  // tile (a, b) of the 64 x 64 block
  f32x16 acc[7];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // A ring of THREE stages, two in flight: with one in flight (request behind the barrier, wait in front of the next) a stage cost
  // one round trip of the request path, ~2500 clocks, whatever its size — which is what bound the one-block-per-workgroup kernel.
  if (nstage > 0) stage(0, 0);
  if (nstage > 1) stage(1, 1);
  int buf = 0;
  for (int s = 0; s < nstage; ++s) {
    // stage s has landed (this wave's part: all but the seven younger requests; then everybody's), and nobody reads the buffer
    // that the request below overwrites any more (it was consumed one iteration ago)
    if (s + 1 < nstage) asm volatile("s_waitcnt vmcnt(7)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    // The seven requests for stage s + 2 are issued ONE per shift, between the shifts' products: a wave whose request waits for
    // room in the request path (the CU ingests ~36 B per clock: a kilobyte every ~28 clocks, eight waves queueing) issues nothing
    // else meanwhile — in a block in front of the products that wait was the stage's whole ingest time, added to the MFMAs'.
    const bool more = !(LK_PIX13_ABLATE & 2) && s + 2 < nstage;
    const int nbuf = buf == 0 ? 2 : buf - 1;
    const unsigned base = lds0 + buf * STAGE;
    buf = buf == 2 ? 0 : buf + 1;
    if (LK_PIX13_ABLATE & 4) {
      if (more) stage(s + 2, nbuf);
      continue;
    }
    f16x8 ah = pix13_frag(base, lane, t_a * 32), al = pix13_frag(base + PLANE, lane, t_a * 32);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int h = g * 7 + i;
      if (more) stage_one(s + 2, nbuf, i);
      if (myslot[i] >= 0) {  // (uniform: the shift exists and the shifted pixel is inside the map)
        f16x8 bh = pix13_frag(base + h * PANEL, lane, t_b * 32), bl = pix13_frag(base + h * PANEL + PLANE, lane, t_b * 32);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah), "+v"(al), "+v"(bh), "+v"(bl));
        f32x16 c = acc[i];
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
        acc[i] = c;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // blocks[slot][row][col] += alpha 2^(-2 sexp) acc: the block is this wave's alone (one workgroup per pixel, one wave per tile)
  const float inv = exp2i16(-p.sexp[0]);
  const float tscale = p.alpha * inv * inv;
  const int lr = lane & 31, lh = lane >> 5;
  // (the old values of FOUR blocks — then three — are requested together: one round trip of memory latency per batch, not per block)
#pragma unroll
  for (int i0 = 0; i0 < 7; i0 += 4) {
    float old[4][16];
#pragma unroll
    for (int i = i0; i < (i0 + 4 < 7 ? i0 + 4 : 7); ++i) {
      if (myslot[i] < 0 || ((LK_PIX13_ABLATE & 1) && p.alpha != 12345.f)) continue;
      const float* blk = p.blocks + (int64_t)myslot[i] * (C * C);
#pragma unroll
      for (int r = 0; r < 16; ++r) old[i - i0][r] = blk[(t_a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * C + t_b * 32 + lr];
    }
#pragma unroll
    for (int i = i0; i < (i0 + 4 < 7 ? i0 + 4 : 7); ++i) {
      if (myslot[i] < 0 || ((LK_PIX13_ABLATE & 1) && p.alpha != 12345.f)) continue;
      float* blk = p.blocks + (int64_t)myslot[i] * (C * C);
#pragma unroll
      for (int r = 0; r < 16; ++r) blk[(t_a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * C + t_b * 32 + lr] = old[i - i0][r] + tscale * acc[i][r];
    }
  }
}
}  // namespace lk

// Pixel-pair blocks of a 3x3 convolution's A factor from a split tensor (see Gram16Tnp): x [B][H][W][Cin] as planes with
// one scale, tables of lk_conv3x3_pixpair_tables (tile edge 64 for Cin % 128 != 0, else 128).
/* The same blocks for a 64-channel map with one workgroup per PIXEL (pixpair13_kernel): slots_dev = the [H W][13] slot table of
 * lk_conv3x3_pixpair_tables. */
extern "C" int lk_conv3x3_pixpair_accumulate13_f16x2(const void* x_h, const void* x_l, const int* sexp, int64_t B, int64_t H,
                                                     int64_t W, int64_t Cin, float alpha, float* blocks,
                                                     const int32_t* slots_dev, const void* zero16, void* stream) {
  LK_REQUIRE(x_h && x_l && sexp && blocks && slots_dev && zero16 && B >= 0 && H >= 1 && W >= 1,
             "lk_conv3x3_pixpair_accumulate13_f16x2: bad arguments");
  LK_REQUIRE(Cin == 64 && H * W * Cin < (1ll << 31) && B < (1ll << 31) && H * W * 13 * Cin * Cin < (1ll << 31),
             "lk_conv3x3_pixpair_accumulate13_f16x2: needs Cin == 64 (and a map within 2^31 elements)");
  if (B == 0) return LK_OK;
  Pix13Args a;
  a.xh = (const _Float16*)x_h, a.xl = (const _Float16*)x_l, a.zero16 = (const _Float16*)zero16, a.slots = slots_dev, a.sexp = sexp;
  a.blocks = blocks, a.alpha = alpha, a.R = (int)B, a.H = (int)H, a.W = (int)W;
  constexpr int LDS = 3 * 13 * 4096;  // 156 KB of the CU's 160: one workgroup per CU
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)pixpair13_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  hipLaunchKernelGGL(pixpair13_kernel, dim3((unsigned)(H * W)), dim3(512), LDS, (hipStream_t)stream, a);
  return check_launch("pixpair13_kernel");
}

extern "C" int lk_conv3x3_pixpair_accumulate_f16x2(const void* x_h, const void* x_l, const int* sexp, int64_t B, int64_t H,
                                                   int64_t W, int64_t Cin, float alpha, float* blocks,
                                                   const int32_t* tiles_dev, int64_t n_tiles, const void* zero16,
                                                   void* stream) {
  LK_REQUIRE(x_h && x_l && sexp && blocks && tiles_dev && zero16 && B >= 0 && n_tiles >= 0,
             "lk_conv3x3_pixpair_accumulate_f16x2: bad arguments");
  LK_REQUIRE(Cin >= 64 && Cin % 64 == 0 && n_tiles < (1ll << 31) && H * W * Cin < (1ll << 31),
             "lk_conv3x3_pixpair_accumulate_f16x2: needs Cin % 64 == 0");
  if (B == 0 || n_tiles == 0) return LK_OK;
  Gram16Tnp tp;
  tp.tiles = tiles_dev, tp.ldc = (int)Cin, tp.sexp = sexp, tp.alpha = alpha, tp.blocks = blocks;
  hipStream_t st = (hipStream_t)stream;
  const int64_t ld = H * W * Cin;
  if (Cin % 128) {
    using Gc = Gram16Cfg<64, 64, 4>;
    hipLaunchKernelGGL((gram16_kernel<64, 64, 4, true>), dim3((unsigned)n_tiles), dim3(256), 2 * Gc::STAGE, st,
                       (const _Float16*)x_h, (const _Float16*)x_l, B, ld, B, (const _Float16*)zero16, (float*)nullptr, 0, tp);
  } else {
    using Gc = Gram16Cfg<128, 32, 1>;
    hipLaunchKernelGGL((gram16_kernel<128, 32, 1, true>), dim3((unsigned)n_tiles), dim3(256), 2 * Gc::STAGE, st,
                       (const _Float16*)x_h, (const _Float16*)x_l, B, ld, B, (const _Float16*)zero16, (float*)nullptr, 0, tp);
  }
  return check_launch("gram16_kernel<pixel pairs>");
}
