// GLM predictive variance of a weight-sharing layer (Conv2d, Linear over a sequence) without its Jacobian.
// Replaces, for such layers, KronDecomposed._bmm / inv_square_form as used by KronLaplace.functional_variance
// (laplace/utils/matrix.py:406-461, laplace/baselaplace.py:1834-1835) and DiagLaplace.functional_variance
// (baselaplace.py:2113-2115), both of which contract a materialised [B, C, Do*Dk] Jacobian block.
//
// The per-sample Jacobian of such a layer is a sum over the L shared positions, J_c = sum_l u_c[l] v[l]^T (Do x Dk),
// so  f_var[c][k] = sum_{o,i} J_c[o,i] J_k[o,i] w[o,i]  needs, per sample, one GEMM [(C*Do) x L] . [L x Dk] whose
// (o,i) tile is held for all C outputs at once in MFMA accumulators, weighted, and folded into the C(C+1)/2 pair sums
// in registers: nothing of size Do*Dk is ever written.  u, v are the output gradients / unfolded inputs, already
// rotated into the factors' eigenbases by the caller for the Kronecker posterior (w = 1/(l1_o l2_i + delta)), raw for
// the diagonal one (w = posterior variance of weight (o,i)).
#include <stdint.h>
#include <stdlib.h>

#include "lk_common.h"

namespace lk {

constexpr int QC_KC = 16;  // positions per chunk

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct QcOperands {  // one tile's operands: sample base pointers, first row, this lane's column
  const float* un;   // u[n]: [C][Do][L] with the outputs `cs` elements apart (Do * L when u is sample-major)
  const float* vn;   // v[n]: [Dk][L]
  int o0, icol;
  unsigned cs;
};

// LDS arena of one workgroup, reinterpreted by the two tile products below
template <int CT>
struct QcLds {
  static constexpr int BYTES = CT * 6144;  // split-bf16: [2][3 pieces][CT][32 o][16 k] bf16; fp32: [2][CT][16 k][32 o] (4096 CT)
};

// ---- generic tile product (any L, any alignment): fp32 operands, v_mfma_f32_32x32x2_f32 --------------------------
// acc[c] = the 32x32 tile (rows o0.., this wave's columns icol) of  sum_l u[c][:, l] v[:, l]^T  for all CT outputs of
// one sample.  A operand (u, all outputs) through double-buffered LDS shared by the 4 waves, B operand straight from
// memory.  Used when the positions cannot be read four at a time.
template <int CT>
__device__ __forceinline__ void qc_tile_gemm(const QcOperands& t, int C, int Do, int Dk, int L, char* lds,
                                             f32x16 (&acc)[CT]) {
  constexpr int NA = 2 * CT;  // staged dwords per thread per chunk: CT * QC_KC * 32 / 256
  float(*sA)[CT][QC_KC][32] = reinterpret_cast<float(*)[CT][QC_KC][32]>(lds);
  const int tid = threadIdx.x, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  float ra[NA], rb[QC_KC / 2];
  auto fetch = [&](int l0) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int e = tid + 256 * j, o = e & 31, ll = (e >> 5) & (QC_KC - 1), c = e >> 9;
      const bool ok = c < C && l0 + ll < L && t.o0 + o < Do;
      // 32-bit offsets from the sample's (uniform) base pointer: the host checks C*L*Do and L*Dk < 2^29
      ra[j] = ok ? t.un[(unsigned)(c * t.cs + (t.o0 + o) * L + l0 + ll)] : 0.f;
    }
#pragma unroll
    for (int kk = 0; kk < QC_KC / 2; ++kk) {
      const int l = l0 + 2 * kk + hi;
      rb[kk] = (l < L && t.icol < Dk) ? t.vn[(unsigned)(t.icol * L + l)] : 0.f;
    }
  };
  fetch(0);
  int buf = 0;
  for (int l0 = 0; l0 < L; l0 += QC_KC) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int e = tid + 256 * j;
      sA[buf][e >> 9][(e >> 5) & (QC_KC - 1)][e & 31] = ra[j];
    }
    float b[QC_KC / 2];
#pragma unroll
    for (int kk = 0; kk < QC_KC / 2; ++kk) b[kk] = rb[kk];
    __syncthreads();
    if (l0 + QC_KC < L) fetch(l0 + QC_KC);
    float a_cur[CT], a_nxt[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) a_cur[c] = sA[buf][c][hi][lo];
#pragma unroll
    for (int kk = 0; kk < QC_KC / 2; ++kk) {
      if (kk + 1 < QC_KC / 2) {
#pragma unroll
        for (int c = 0; c < CT; ++c) a_nxt[c] = sA[buf][c][2 * kk + 2 + hi][lo];
      }
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[c], b[kk], acc[c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < CT; ++c) a_cur[c] = a_nxt[c];
    }
    buf ^= 1;  // the other buffer was last read two chunks ago: one barrier per chunk suffices
  }
  __syncthreads();  // every wave is done with both LDS buffers before the caller's next tile refills them
}

// ---- the tile product on the bf16 matrix cores at fp32 accuracy (L % 4 == 0) --------------------------------------
// Every operand is split ONCE into three bf16 pieces, x = h + m + l exactly (truncation keeps the subtractions exact),
// and  x y ~= h h' + h m' + m h' + m m' + h l' + l h'  (dropped terms <= 3 * 2^-24 |x y|): six
// v_mfma_f32_32x32x16_bf16 (32 cycles, 16 positions) instead of eight v_mfma_f32_32x32x2_f32 (64 cycles, 2 positions)
// per chunk and output -- 192 matrix-pipe cycles where the fp32 form needs 512.  Both operands are position-contiguous
// in memory, which is what a lane of the bf16 MFMA wants (8 consecutive k): the A chunk is staged as float4s along
// the positions, split by the staging thread and kept in LDS as [piece][output][row o][16 k]; the B operand is this
// lane's own 8 positions of column icol, split in registers.  The pipeline runs across tiles: during a tile's last
// chunk the first chunk of the NEXT tile (other rows / columns, or the next sample) is fetched; loads are issued raw
// from a clamped address and zeroed where they are consumed.
__device__ __forceinline__ void qc_split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  h = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xffff0000u;
  l = __float_as_uint(r1 - __uint_as_float(m));  // at most 8 significant bits are left: exact in bf16
}
__device__ __forceinline__ unsigned qc_pack(unsigned even, unsigned odd) { return (even >> 16) | (odd & 0xffff0000u); }

template <int CT>
struct QcStage {  // raw operands in flight: this thread's float4 slots of the A chunk, this lane's 8 positions of B
  f32x4 ra[(CT + 1) / 2];
  f32x4 rb[2];
};

// staging coordinates of thread tid: float4 slot e4 = tid + 256 j  ->  k4 = 4 (tid & 3), row o = (tid >> 2) & 31,
// output c = (tid >> 7) + 2 j
template <int CT>
__device__ __forceinline__ void qc_fetch_b6(QcStage<CT>& st, int j_lo, int j_hi, int h_lo, int h_hi, const QcOperands& t,
                                            int l0, int C, int Do, int Dk, int L) {
  const int tid = threadIdx.x, hi = (tid & 63) >> 5;
  const int k4 = 4 * (tid & 3), o = (tid >> 2) & 31, c0 = tid >> 7;
#pragma unroll
  for (int j = j_lo; j < j_hi; ++j) {
    const bool ok = t.o0 + o < Do && c0 + 2 * j < C && l0 + k4 < L;
    const unsigned off = ok ? (unsigned)((c0 + 2 * j) * t.cs + (t.o0 + o) * L + l0 + k4) : 0u;
    st.ra[j] = *reinterpret_cast<const f32x4*>(t.un + off);
  }
#pragma unroll
  for (int h = h_lo; h < h_hi; ++h) {
    const int l = l0 + 8 * hi + 4 * h;
    const bool ok = t.icol < Dk && l < L;
    st.rb[h] = *reinterpret_cast<const f32x4*>(t.vn + (ok ? (unsigned)(t.icol * L + l) : 0u));
  }
}

// On entry `st` holds chunk 0 of `cur`; on exit chunk 0 of `nxt` (if has_next).
template <int CT>
__device__ __forceinline__ void qc_tile_gemm_b6(const QcOperands& cur, const QcOperands& nxt, bool has_next, int C,
                                                int Do, int Dk, int L, char* lds, f32x16 (&acc)[CT], QcStage<CT>& st) {
  constexpr int NA4 = (CT + 1) / 2;        // float4 slots per thread per chunk: CT * 32 * 4 / 256
  constexpr int PIECE = CT * 32 * 16 * 2;  // bytes of one piece of one buffer
  const int tid = threadIdx.x, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
  const int k4 = 4 * (tid & 3), o = (tid >> 2) & 31, c0 = tid >> 7;
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  const bool okO = cur.o0 + o < Do, okI = cur.icol < Dk;
  int buf = 0;
  for (int l0 = 0; l0 < L; l0 += QC_KC) {
    char* wr = lds + buf * 3 * PIECE;
#pragma unroll
    for (int j = 0; j < NA4; ++j)
      if (c0 + 2 * j < CT) {
        const bool ok = okO && c0 + 2 * j < C && l0 + k4 < L;
        const f32x4 x = ok ? st.ra[j] : f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) qc_split3(x[q], h[q], m[q], l[q]);
        char* dst = wr + (((c0 + 2 * j) * 32 + o) * 16 + k4) * 2;
        *reinterpret_cast<u32x2*>(dst) = u32x2{qc_pack(h[0], h[1]), qc_pack(h[2], h[3])};
        *reinterpret_cast<u32x2*>(dst + PIECE) = u32x2{qc_pack(m[0], m[1]), qc_pack(m[2], m[3])};
        *reinterpret_cast<u32x2*>(dst + 2 * PIECE) = u32x2{qc_pack(l[0], l[1]), qc_pack(l[2], l[3])};
      }
    // this lane's B operand: positions l0 + 8 hi .. + 7 of column icol, as three bf16x8
    u32x4 bp[3];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const bool ok = okI && l0 + 8 * hi + 4 * hh < L;
      const f32x4 x = ok ? st.rb[hh] : f32x4{0.f, 0.f, 0.f, 0.f};
      unsigned h[4], m[4], l[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) qc_split3(x[q], h[q], m[q], l[q]);
      bp[0][2 * hh] = qc_pack(h[0], h[1]);
      bp[0][2 * hh + 1] = qc_pack(h[2], h[3]);
      bp[1][2 * hh] = qc_pack(m[0], m[1]);
      bp[1][2 * hh + 1] = qc_pack(m[2], m[3]);
      bp[2][2 * hh] = qc_pack(l[0], l[1]);
      bp[2][2 * hh + 1] = qc_pack(l[2], l[3]);
    }
    bf16x8 b[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) b[p] = __builtin_bit_cast(bf16x8, bp[p]);
    __syncthreads();
    // what travels during this chunk: the tile's next chunk, or chunk 0 of the next tile -- chosen with uniform
    // selects, not branches; the very last chunk of a workgroup re-reads its own chunk 0 for nothing
    const bool more = l0 + QC_KC < L;
    const QcOperands src = more ? cur : (has_next ? nxt : cur);
    const int lsrc = more ? l0 + QC_KC : 0;
    const char* rd = lds + buf * 3 * PIECE + (lo * 16 + 8 * hi) * 2;  // this lane's (row, k half) in output 0, piece 0
    bf16x8 a_cur[3], a_nxt[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) a_cur[p] = *reinterpret_cast<const bf16x8*>(rd + p * PIECE);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      if (c + 1 < CT) {
#pragma unroll
        for (int p = 0; p < 3; ++p) a_nxt[p] = *reinterpret_cast<const bf16x8*>(rd + p * PIECE + (c + 1) * 32 * 16 * 2);
      }
      // the next chunk's loads go out one per output between the MFMA groups
      qc_fetch_b6<CT>(st, c, c < NA4 ? c + 1 : c, c < 2 ? c : 2, (c < 2 ? c + 1 : 2) + (CT == 1 ? 1 : 0), src, lsrc, C, Do,
                      Dk, L);
      f32x16 d = acc[c];
      d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[2], b[0], d, 0, 0, 0);  // small terms first
      d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[0], b[2], d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[1], b[1], d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[1], b[0], d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[0], b[1], d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[0], b[0], d, 0, 0, 0);
      acc[c] = d;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < 3; ++p) a_cur[p] = a_nxt[p];
    }
    buf ^= 1;  // the other buffer was last read two chunks ago: one barrier per chunk suffices
  }
  __syncthreads();
}

// ---- helpers of the two-piece fp16 form on PRE-SPLIT operands (quadform_conv_planes_kernel below): x 2^s = h + l with h, l
// fp16, x y 2^(su+sv) ~= h h' + h l' + l h' -> three v_mfma_f32_32x32x16_f16 per chunk and output where the three-piece bf16
// form needs six.  (An in-flight variant of it — fp32 operands split by the staging threads — existed in rounds 2 - 4 and
// measured equal to the bf16 form: the kernel was never bound by its matrix pipe.  Removed in round 5.)
typedef _Float16 qf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 qf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int qc_scale_exp(float amax) {  // amax * 2^s in [2^14, 2^15)
  int be = (int)((__float_as_uint(amax) >> 23) & 0xffu);
  if (be == 0) be = 1;
  const int s = 14 - (be - 127);
  return s > 120 ? 120 : s;
}
__device__ __forceinline__ float qc_exp2i(int s) {
  s = s < -126 ? -126 : (s > 127 ? 127 : s);
  return __uint_as_float((unsigned)(127 + s) << 23);
}
// grid = B * split workgroups of 4 waves; workgroup (n, sp) walks the super-tiles (32 rows o) x (128 columns i)
// t = sp, sp + split, ...; wave w owns columns [32 w, 32 w + 32) of the super-tile for all CT outputs.
// ARITH: 0 = fp32 MFMA (any L), 1 = three-piece bf16 (six MFMAs)
template <int CT, int MODE, int ARITH>
__global__ __launch_bounds__(256) void quadform_conv_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                            const float* __restrict__ w0, const float* __restrict__ w1,
                                                            const float* __restrict__ delta, int C, int Do, int Dk, int L,
                                                            int split, float* __restrict__ partial,
                                                            int64_t u_sample_stride, unsigned u_class_stride) {
  constexpr bool B6 = ARITH != 0;
  constexpr int NP = CT * (CT + 1) / 2;
  __shared__ __attribute__((aligned(16))) char lds[QcLds<CT>::BYTES];
  __shared__ float sR[4][NP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
  // XCD-aware: consecutive block ids run on different XCDs (id % 8); the workgroups of one sample re-read the same
  // u[n], v[n] tile after tile, so they are all given to ONE XCD (its L2 then holds the sample's operands)
  int n = blockIdx.x / split, sp = blockIdx.x % split;
  if (gridDim.x % (8 * split) == 0) {
    const int xcd = blockIdx.x % 8, j = blockIdx.x / 8;
    n = xcd + 8 * (j / split), sp = j % split;
  }
  const int nOt = (Do + 31) / 32, nIg = (Dk + 127) / 128, ntiles = nOt * nIg;
  const float* __restrict__ un = u + (size_t)n * u_sample_stride;
  const float* __restrict__ vn = v + (size_t)n * Dk * L;
  const float dlt = MODE == 0 ? delta[0] : 0.f;

  float pair[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) pair[p] = 0.f;

  auto operands = [&](int t) { return QcOperands{un, vn, (t % nOt) * 32, (t / nOt) * 128 + wave * 32 + lo, u_class_stride}; };
  QcStage<CT> st;
  if (B6 && sp < ntiles) qc_fetch_b6<CT>(st, 0, (CT + 1) / 2, 0, 2, operands(sp), 0, C, Do, Dk, L);
  for (int t = sp; t < ntiles; t += split) {
    const QcOperands cur = operands(t);
    const int o0 = cur.o0, icol = cur.icol;
    f32x16 acc[CT];
    if constexpr (ARITH == 1)
      qc_tile_gemm_b6<CT>(cur, operands(t + split), t + split < ntiles, C, Do, Dk, L, lds, acc, st);
    else
      qc_tile_gemm<CT>(cur, C, Do, Dk, L, lds, acc);

    // weights of this lane's 16 (o, i) positions and the pair sums, four accumulator rows at a time (the
    // accumulators live in AGPRs; only 4 * CT of them are copied out at once)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      float wgt[4], a[CT][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o = o0 + j + 8 * rg + 4 * hi;
        const bool ok = o < Do && icol < Dk;
        if (MODE == 0) {
          const float d = w0[ok ? o : 0] * w1[ok ? icol : 0] + dlt;
          wgt[j] = ok ? __builtin_amdgcn_rcpf(d) : 0.f;
        } else {
          wgt[j] = ok ? w0[(unsigned)(o * Dk + icol)] : 0.f;
        }
      }
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[c][j] = acc[c][4 * rg + j];
      int p = 0;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        float sc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sc[j] = a[c][j] * wgt[j];
#pragma unroll
        for (int k = c; k < CT; ++k) {
          pair[p] += (sc[0] * a[k][0] + sc[1] * a[k][1]) + (sc[2] * a[k][2] + sc[3] * a[k][3]);
          ++p;
        }
      }
    }
  }

#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const float s = wave_sum(pair[p]);
    if (lane == 0) sR[wave][p] = s;
  }
  __syncthreads();
  if (tid < NP) partial[((size_t)n * split + sp) * NP + tid] = (sR[0][tid] + sR[1][tid]) + (sR[2][tid] + sR[3][tid]);
}

// ---- the quadratic form on PRE-SPLIT operands (round 5) ------------------------------------------------------------------
// quadform_conv_kernel spends as many vector-pipe cycles splitting its fp32 operands in flight (three bf16 pieces of 28 values
// per thread and chunk) as its matrix pipe spends on the six MFMAs per block: one wave per SIMD, nothing to hide either
// under.  Here the operands ARRIVE split — two fp16 planes each, position-contiguous, written that way by the epilogue of
// the rotation convolutions that produce them (lk_conv_nhwc_f16x2_planes: u = Q1^T g as a 1x1 convolution over the split
// cotangent, v = the unfolded activations in the A factor's eigenbasis) — so a chunk is 8-byte copies into LDS (A), two
// 16-byte loads (B) and CT x three v_mfma_f32_32x32x16_f16.  u: [C][B] images (seed-major) with ONE scale, v: [B] images with
// one scale per sample (v_nsexp = B) or one for the tensor; L % 16 == 0 (whole chunks), Do % 32 == 0.  An image is
// CHUNK-major, [L / 16][rows][16 positions] (round 6): a request — 16 positions of 32 rows — is one contiguous kilobyte.  From
// the position-contiguous [rows][L] of round 5 it was 32 pieces of 32 bytes 2 L bytes apart: every piece its own cache line, of
// which the next three chunks use the rest only if it is still cached — the 64-channel layers (L = 1024, operands of 640 MB per
// layer) ran at 62 TFLOP/s against 190 at L = 64, four times the compulsory bytes from HBM (profiles/r06_quad_layers.log).
typedef float qf32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void qc_lds_void;
typedef __attribute__((address_space(1))) const void qc_gbl_void;

// Staging: LDS-DMA (global_load_lds, 16 bytes per lane, no register round trip) into a ring of NS stages, requested NS - 1
// chunks ahead and waited for by COUNT — with one wave per SIMD (160 accumulator registers for ten outputs) nothing else
// hides a load, and the first form of this kernel, which fetched one chunk ahead through registers, ran at the latency of
// its loads (14 ms per c4 call, as the fp32-operand kernel before it: neither its MFMAs, 1.7 ms at the nominal rate, nor its
// bytes, 4.7 ms at the CU's ~12 B / clock ingest rate, were the bound).  A stage: A = [piece][output][32 rows][16 k] (one
// wave-instruction = one (piece, output): 32 rows x 32 bytes = a lane-linear kilobyte), B = [wave][piece][lane] (a lane's
// own 8 positions of its column).  The stream of chunks runs across tiles; the pair sums of a tile touch registers only.
template <int CT, int OCC = 1>
struct QcPlanesCfg {
  static constexpr int NS = OCC == 2 ? 2 : (CT <= 5 ? 6 : 4);                 // ring stages
  static constexpr int A_BYTES = 2 * CT * 1024, B_BYTES = 4 * 2 * 1024, STAGE = A_BYTES + B_BYTES;
  static constexpr int NA = (2 * CT + 3) / 4;                // A instructions per wave and chunk (padded: a repeat)
  static constexpr int LD = NA + 2;                          // LDS-DMA instructions per wave and chunk
  static constexpr int LDS = NS * STAGE;
  static_assert(LD * (NS - 2) <= 63, "vmcnt is a 6-bit counter");
};

// SUB (one-chunk tiles: L == 16, the 4 x 4 maps of the deepest layers; OCC = 2, eigenvalues in LDS): a tile is ONE chunk there
// and the pair sums — 55 x 16 positions per lane — are the kernel (profiles/r06_pmc_quad_launches.log: instruction issue,
// not the matrix pipe or bytes).  The wave's 32 x 32 region is walked as four 16 x 16 sub-tiles on v_mfma_f32_16x16x32_f16:
// 4 accumulator registers per output instead of 16 leave room for TWO-wide running pair sums beside two waves per SIMD, i.e.
// v_pk_fma_f32 on pairs of positions without the register shuffles the one-wide sums of the 32 x 32 form cost (~700 vector
// instructions per tile instead of ~1900).
template <int CT>
struct QcSubPlan {  // SUB: the classes whose products are issued during pair-sum row c (front-loaded like the rows: CT - c pairs)
  int cnt[10], first[10];
  constexpr QcSubPlan() : cnt{}, first{} {
    constexpr int base[10] = {3, 2, 2, 1, 1, 1, 0, 0, 0, 0};
    int before = 0;
    for (int c = 0; c < 10; ++c) {
      const int left = CT - before;
      cnt[c] = left <= 0 ? 0 : (base[c] < left ? base[c] : left);
      first[c] = before;
      before += cnt[c];
    }
  }
};

template <int CT, int OCC = 1, bool SUB = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void quadform_conv_planes_kernel(
    const _Float16* __restrict__ uh, const _Float16* __restrict__ ul, const int* __restrict__ u_sexp,
    const _Float16* __restrict__ vh, const _Float16* __restrict__ vl, const int* __restrict__ v_sexp, int v_nsexp,
    const float* __restrict__ w0, const float* __restrict__ w1, const float* __restrict__ delta, int B, int C, int Do, int Dk,
    int L, int split, float* __restrict__ partial, const _Float16* __restrict__ zero16, int w_in_lds) {
  using CFG = QcPlanesCfg<CT, OCC>;
  constexpr int NP = CT * (CT + 1) / 2, NS = CFG::NS;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  __shared__ float sR[4][NP];
  const int tid = threadIdx.x, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int n = blockIdx.x / split, sp = blockIdx.x % split;
  if (gridDim.x % (8 * split) == 0) {  // (all workgroups of one sample on ONE XCD: its L2 holds the sample's operands)
    const int xcd = blockIdx.x % 8, j = blockIdx.x / 8;
    n = xcd + 8 * (j / split), sp = j % split;
  }
  const int nOt = Do / 32, nIg = (Dk + 127) / 128, ntiles = nOt * nIg;
  const int NCH = L / QC_KC;
  const unsigned cs = (unsigned)B * Do * L;
  const _Float16* __restrict__ uhn = uh + (size_t)n * Do * L;
  const _Float16* __restrict__ uln = ul + (size_t)n * Do * L;
  const _Float16* __restrict__ vhn = vh + (size_t)n * Dk * L;
  const _Float16* __restrict__ vln = vl + (size_t)n * Dk * L;
  const int su = u_sexp[0], sv = v_sexp[n < v_nsexp ? n : v_nsexp - 1];
  const float un_u = qc_exp2i(-su), un_v = qc_exp2i(-sv);
  const float dlt = delta[0];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  // the eigenvalues behind the ring (host: w_in_lds when they fit): read in every tile's epilogue — from memory those loads
  // would sit in the same in-order counter as the ring's requests and drain it once per tile (a tile of a 4 x 4 map is ONE chunk)
  float* ldsw = reinterpret_cast<float*>(lds + CFG::LDS);
  if (w_in_lds) {
    for (int i = tid; i < Do; i += 256) ldsw[i] = w0[i];
    for (int i = tid; i < Dk; i += 256) ldsw[Do + i] = w1[i];
    __syncthreads();
  }

  constexpr bool ONE_WIDE = OCC == 2;
  qf32x2 pair2[ONE_WIDE ? 1 : NP];  // (two-wide running pair sums: see the tile epilogue)
  float pair1[ONE_WIDE ? NP : 1];
#pragma unroll
  for (int p = 0; p < (ONE_WIDE ? 1 : NP); ++p) pair2[p] = qf32x2{0.f, 0.f};
#pragma unroll
  for (int p = 0; p < (ONE_WIDE ? NP : 1); ++p) pair1[p] = 0.f;
  const int my_tiles = sp < ntiles ? (ntiles - sp + split - 1) / split : 0;
  const int Q = my_tiles * NCH;  // chunks of this workgroup's walk

  // ---- the producer side: chunk (tile ordinal kt, chunk ch) -> stage `st`
  int p_kt = 0, p_ch = 0, p_stage = 0;  // next chunk to request
  auto request = [&]() {
    const int t = sp + p_kt * split;
    const int o0 = (t % nOt) * 32, icol = (t / nOt) * 128 + wave * 32 + lo;
    const unsigned base = lds0 + p_stage * CFG::STAGE;
    const unsigned a_lane = (unsigned)(((p_ch * Do + o0 + (lane >> 1)) << 4) + 8 * (lane & 1));
#pragma unroll
    for (int j = 0; j < CFG::NA; ++j) {
      int a = j * 4 + wave;                 // (scalar) instruction index = piece * CT + output
      if (a >= 2 * CT) a = a % (2 * CT);    // padding (uniform counts for the counted waits): repeat one (same bytes, same place)
      const int piece = a >= CT ? 1 : 0, c = a - piece * CT;
      const _Float16* src = (c < C ? (piece ? uln : uhn) + (unsigned)(c * cs) + a_lane : zero16);
      __builtin_amdgcn_global_load_lds((qc_gbl_void*)src, (qc_lds_void*)(uintptr_t)(base + a * 1024), 16, 0, 0);
    }
    const unsigned b_lane = (unsigned)(((p_ch * Dk + icol) << 4) + 8 * hi);
#pragma unroll
    for (int piece = 0; piece < 2; ++piece) {
      const _Float16* src = icol < Dk ? (piece ? vln : vhn) + b_lane : zero16;
      __builtin_amdgcn_global_load_lds((qc_gbl_void*)src, (qc_lds_void*)(uintptr_t)(base + CFG::A_BYTES + (wave * 2 + piece) * 1024),
                                       16, 0, 0);
    }
    p_stage = p_stage + 1 == NS ? 0 : p_stage + 1;
    if (++p_ch == NCH) p_ch = 0, ++p_kt;
  };
#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (i < Q) request();

  int c_stage = 0, q = 0;
  for (int kt = 0; kt < my_tiles; ++kt) {
    const int t = sp + kt * split;
    const int o0 = (t % nOt) * 32, icol = (t / nOt) * 128 + wave * 32 + lo;
    if constexpr (SUB) {
      // (the tile's one chunk: same hand-over as in the chunk loop below)
      if (q + NS - 1 <= Q) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(CFG::LD * (NS - 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      if (q + NS - 1 < Q) request();
      const unsigned stage = lds0 + c_stage * CFG::STAGE;
      c_stage = c_stage + 1 == NS ? 0 : c_stage + 1;
      ++q;
      // fragment lane of v_mfma_f32_16x16x32_f16: row / column l16, k = 8 kg .. + 7 of 32.  The 16 positions of the chunk take
      // HALF of that depth, so the two planes ride in the other half: A' = [a_h | a_l] against B' = [b_h | b_h] and then
      // [b_l | b_l] — all four plane products (the l l term included) in two instructions instead of three of the k = 16 form.
      // The products of sub-tile s + 1 are issued INSIDE the pair sums of sub-tile s (one-wide v_fma_f32 runs beside the
      // matrix pipe — csrc/Makefile —; with two waves per SIMD a wave that only waits for its MFMAs leaves the other one issuing
      // a vector instruction every ~5 clocks instead of the pair's 2.7).
      const int l16 = lane & 15, kg = lane >> 4;
      auto lds_read128 = [&](qf16x8& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr)); };
      const float un = un_u * un_v;  // (a power of two: one factor rides on the weights, the other on the final sums — exact)
      const unsigned rb0 = stage + CFG::A_BYTES + (wave * 2) * 1024 + (l16 + 32 * (kg & 1)) * 16;  // + 256 sk
      const unsigned ra0 = stage + (kg >> 1) * (CT * 1024) + l16 * 32 + (kg & 1) * 16;              // + 512 so + 1024 c
      f32x4 accs[2][CT];
      qf16x8 bh, bl, a2[2];
      // classes whose products are issued during pair-sum row c (row c holds CT - c pairs): front-loaded like the rows
      constexpr QcSubPlan<CT> plan{};
      auto products = [&](int sub, int c) {  // class c of sub-tile `sub` (fragment set c & 1; class c + 1 requested first)
        const unsigned ra = ra0 + (sub >> 1) * 512;
        const int set = c & 1;
        if (c + 1 < CT) {
          lds_read128(a2[set ^ 1], ra + (c + 1) * 1024);
          asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a2[set]), "+v"(bh), "+v"(bl));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a2[set]), "+v"(bh), "+v"(bl));
        }
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[set], bl, d, 0, 0, 0);  // small terms first
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[set], bh, d, 0, 0, 0);
        accs[sub & 1][c] = d;
      };
      auto first_reads = [&](int sub) {
        lds_read128(bh, rb0 + (sub & 1) * 256);
        lds_read128(bl, rb0 + (sub & 1) * 256 + 1024);
        lds_read128(a2[0], ra0 + (sub >> 1) * 512);
      };
      first_reads(0);
#pragma unroll
      for (int c = 0; c < CT; ++c) products(0, c);
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        const int so = sub >> 1, sk = sub & 1;
        // this lane's four positions: rows o0 + 16 so + 4 kg .. + 3 of column (icol - lo) + 16 sk + l16
        const int oo = o0 + so * 16 + 4 * kg, col = icol - lo + sk * 16 + l16;
        const bool ok = col < Dk;
        const float wc = ldsw[Do + (ok ? col : 0)];
        const f32x4 wr = *reinterpret_cast<const f32x4*>(ldsw + oo);
        float wg[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wg[j] = ok ? __builtin_amdgcn_rcpf(wr[j] * wc + dlt) * un : 0.f;
        if (sub + 1 < 4) first_reads(sub + 1);
        int p = 0;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          if (sub + 1 < 4) {
#pragma unroll
            for (int i = 0; i < plan.cnt[c]; ++i) products(sub + 1, plan.first[c] + i);
          }
          float sw[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) sw[r] = accs[sub & 1][c][r] * wg[r];
          // (position-major: consecutive instructions belong to different sums — a wave's dependent v_fma_f32 issue ~8 clocks apart)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = c; k < CT; ++k) pair1[p + k - c] = __builtin_fmaf(sw[r], accs[sub & 1][k][r], pair1[p + k - c]);
          p += CT - c;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      continue;
    }
    f32x16 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int ch = 0; ch < NCH; ++ch, ++q) {
      // chunk q has landed (this wave's part: all but the NS - 2 younger requests), then everybody's; and nobody reads the
      // stage that the request below overwrites any more (it was consumed one iteration ago)
      if (q + NS - 1 <= Q) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(CFG::LD * (NS - 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // tail: fewer requests in flight
      if (q + NS - 1 < Q) request();
      const unsigned stage = lds0 + c_stage * CFG::STAGE;
      c_stage = c_stage + 1 == NS ? 0 : c_stage + 1;
      // Fragment reads as asm, two register sets, waited for by COUNT: left to hipcc every read sat directly in front of
      // the MFMA that consumes it behind an lgkmcnt(0) (one register set, ISA checked) — 20 exposed LDS round trips per
      // chunk on a SIMD whose only wave this is.  The reads of output c + 1 are in flight under the three MFMAs of output c.
      auto lds_read = [&](qf16x8& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr)); };
      qf16x8 bh, bl, ah[2], al[2];
      const unsigned rd = stage + lo * 32 + hi * 16;  // this lane's (row, k half) of output 0, piece h
      lds_read(bh, stage + CFG::A_BYTES + (wave * 2) * 1024 + lane * 16);
      lds_read(bl, stage + CFG::A_BYTES + (wave * 2 + 1) * 1024 + lane * 16);
      lds_read(ah[0], rd);
      lds_read(al[0], rd + CT * 1024);
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int set = c & 1;
        if (c + 1 < CT) {
          lds_read(ah[set ^ 1], rd + (c + 1) * 1024);
          lds_read(al[set ^ 1], rd + (CT + c + 1) * 1024);
          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[set]), "+v"(al[set]), "+v"(bh), "+v"(bl));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[set]), "+v"(al[set]), "+v"(bh), "+v"(bl));
        }
        f32x16 d = acc[c];
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[set], bh, d, 0, 0, 0);  // small terms first
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[set], bl, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[set], bh, d, 0, 0, 0);
        acc[c] = d;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // weights of this lane's 16 (o, i) positions and the pair sums, four accumulator rows at a time.  At 512 channels a tile is
    // ONE chunk (L = 16) and this block — 55 pair sums x 16 positions per lane — is most of the kernel: those launches ran at
    // 1.7 ms where the 64-channel ones (64 chunks per tile) take 0.43 (profiles/r06_pmc_quad_launches.log: 8 x the
    // instructions).  Written on PAIRS of positions (v_pk_mul_f32 / v_pk_fma_f32: two products per instruction) with two-wide
    // running sums: 2 instructions per (pair, four positions) instead of the 6 of `(s0 a0 + s1 a1) + (s2 a2 + s3 a3)`.
    if constexpr (ONE_WIDE) {
      // two waves per SIMD (256 registers): one-wide running sums on one-wide instructions (the library is built without the
      // packed fp32 forms — csrc/Makefile —: v_fma_f32 runs beside the partner wave's MFMAs, v_pk_fma_f32 does not), two
      // positions at a time; one factor of the operands' scale rides on the weights, the other on the final sums (powers of
      // two: exact)
      const float un = un_u * un_v;
#pragma unroll
      for (int rh = 0; rh < 8; ++rh) {
        float wg[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int oo = o0 + (2 * (rh & 1) + j) + 8 * (rh >> 1) + 4 * hi;
          const bool ok = oo < Do && icol < Dk;
          const float d = (w_in_lds ? ldsw[ok ? oo : 0] * ldsw[Do + (ok ? icol : 0)] : w0[ok ? oo : 0] * w1[ok ? icol : 0]) + dlt;
          wg[j] = ok ? __builtin_amdgcn_rcpf(d) * un : 0.f;
        }
        int p = 0;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const float s0 = acc[c][2 * rh] * wg[0], s1 = acc[c][2 * rh + 1] * wg[1];
#pragma unroll
          for (int k = c; k < CT; ++k) pair1[p + k - c] = __builtin_fmaf(s0, acc[k][2 * rh], pair1[p + k - c]);
#pragma unroll
          for (int k = c; k < CT; ++k) pair1[p + k - c] = __builtin_fmaf(s1, acc[k][2 * rh + 1], pair1[p + k - c]);
          p += CT - c;
        }
      }
    } else {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      qf32x2 wgt[2], a[CT][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int oo = o0 + j + 8 * rg + 4 * hi;
        const bool ok = oo < Do && icol < Dk;
        const float d = (w_in_lds ? ldsw[ok ? oo : 0] * ldsw[Do + (ok ? icol : 0)] : w0[ok ? oo : 0] * w1[ok ? icol : 0]) + dlt;
        wgt[j >> 1][j & 1] = ok ? __builtin_amdgcn_rcpf(d) : 0.f;
      }
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const qf32x2 r = {acc[c][4 * rg + 2 * h], acc[c][4 * rg + 2 * h + 1]};
          a[c][h] = (r * un_u) * un_v;
        }
      int p = 0;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const qf32x2 s0 = a[c][0] * wgt[0], s1 = a[c][1] * wgt[1];
#pragma unroll
        for (int k = c; k < CT; ++k) {
          pair2[p] = __builtin_elementwise_fma(s0, a[k][0], pair2[p]);
          pair2[p] = __builtin_elementwise_fma(s1, a[k][1], pair2[p]);
          ++p;
        }
      }
    }
  }
    }
  float pair[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) pair[p] = ONE_WIDE ? pair1[p] * ((SUB || OCC == 2) ? un_u * un_v : 1.f) : pair2[p][0] + pair2[p][1];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const float s = wave_sum(pair[p]);
    if (lane == 0) sR[wave][p] = s;
  }
  __syncthreads();
  if (tid < NP) partial[((size_t)n * split + sp) * NP + tid] = (sR[0][tid] + sR[1][tid]) + (sR[2][tid] + sR[3][tid]);
}

// fvar[n][c][k] (and [k][c]) += sum over the workgroups of sample n, in fixed order
__global__ __launch_bounds__(256) void quadform_conv_reduce_kernel(const float* __restrict__ partial, int64_t B, int C,
                                                                   int CT, int split, float* __restrict__ fvar) {
  const int NP = CT * (CT + 1) / 2;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= B * NP) return;
  const int64_t n = e / NP;
  int p = (int)(e % NP), c = 0, rowlen = CT;
  while (p >= rowlen) {
    p -= rowlen;
    ++c;
    --rowlen;
  }
  const int k = c + p;
  if (k >= C) return;  // padded outputs
  float s = 0.f;
  for (int sp = 0; sp < split; ++sp) s += partial[((size_t)n * split + sp) * NP + (e % NP)];
  fvar[(n * C + c) * C + k] += s;
  if (k != c) fvar[(n * C + k) * C + c] += s;
}

// Exact GGN diagonal of a weight-sharing layer: h[o][i] = sum_{n, s} (sum_l u[n][s][l][o] v[n][l][i])^2 -- the squared
// per-sample, per-seed weight Jacobian summed over the minibatch, without the [B, S, Do*Dk] Jacobian.
// grid = ntiles * nsplit workgroups; workgroup (t, sp) owns the super-tile t for the samples sp, sp + nsplit, ... and
// keeps the running sum of squares of its 32x32 wave tiles in registers.
template <int CT, bool B6>
__global__ __launch_bounds__(256) void diag_ggn_shared_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                              int B, int C, int Do, int Dk, int L, int nsplit,
                                                              float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) char lds[QcLds<CT>::BYTES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lo = lane & 31, hi = lane >> 5;
  const int t = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  const int nOt = (Do + 31) / 32;
  const int o0 = (t % nOt) * 32, icol = (t / nOt) * 128 + wave * 32 + lo;
  float hacc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
  auto operands = [&](int n) { return QcOperands{u + (size_t)n * C * Do * L, v + (size_t)n * Dk * L, o0, icol, (unsigned)(Do * L)}; };
  QcStage<CT> st;
  if (B6 && sp < B) qc_fetch_b6<CT>(st, 0, (CT + 1) / 2, 0, 2, operands(sp), 0, C, Do, Dk, L);
  for (int n = sp; n < B; n += nsplit) {
    f32x16 acc[CT];
    if (B6)
      qc_tile_gemm_b6<CT>(operands(n), operands(n + nsplit < B ? n + nsplit : n), n + nsplit < B, C, Do, Dk, L, lds, acc, st);
    else
      qc_tile_gemm<CT>(operands(n), C, Do, Dk, L, lds, acc);
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) hacc[r] += acc[c][r] * acc[c][r];
  }
  float* __restrict__ out = partial + (size_t)sp * Do * Dk;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    if (o < Do && icol < Dk) out[(size_t)o * Dk + icol] = hacc[r];
  }
}

__global__ __launch_bounds__(256) void diag_ggn_shared_reduce_kernel(const float* __restrict__ partial, int64_t width,
                                                                     int nsplit, float alpha, float* __restrict__ h) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= width) return;
  float s = 0.f;
  for (int sp = 0; sp < nsplit; ++sp) s += partial[(size_t)sp * width + e];
  h[e] += alpha * s;
}

}  // namespace lk

using namespace lk;

static int qc_b6_enabled() { return 1; }
static bool qc_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static int qc_class_tile(int64_t C) {
  static const int tiles[] = {1, 2, 3, 4, 5, 6, 8, 10};  // 12 outputs would spill accumulators
  for (int t : tiles)
    if (C <= t) return t;
  return 0;
}

static int qc_split(int64_t B, int64_t Do, int64_t Dk) {
  const int64_t ntiles = ((Do + 31) / 32) * ((Dk + 127) / 128);
  int64_t want = (2048 + B - 1) / B;
#ifdef LK_QC_MIN_SPLIT
  if (want < LK_QC_MIN_SPLIT) want = LK_QC_MIN_SPLIT;
#endif
  if (want < 1) want = 1;
  return (int)(want < ntiles ? want : ntiles);
}

extern "C" size_t lk_quadform_shared_workspace_bytes(int64_t B, int64_t C, int64_t Do, int64_t Dk) {
  const int ct = qc_class_tile(C);
  if (B < 0 || ct == 0 || Do < 1 || Dk < 1) return 0;
  return (size_t)B * qc_split(B, Do, Dk) * (ct * (ct + 1) / 2) * sizeof(float);
}

template <int MODE>
static int launch_quadform_conv(const float* u, const float* v, const float* w0, const float* w1, const float* delta,
                                int64_t B, int64_t C, int64_t Do, int64_t Dk, int64_t L, float* fvar, void* ws,
                                size_t ws_bytes, hipStream_t stream, const char* what, bool seed_major = false) {
  // u is [B][C][Do][L] (sample-major) or, seed_major, [C][B][Do][L] (how a seed-batched sweep leaves it)
  const int64_t uss = seed_major ? Do * L : C * Do * L;
  const unsigned ucs = (unsigned)(seed_major ? B * Do * L : Do * L);
  const int ct = qc_class_tile(C);
  if (ct == 0) {
    set_error("%s: more than 10 outputs are not supported by the fused kernel", what);
    return LK_EINVAL;
  }
  if (B == 0) return LK_OK;
  if (ws == nullptr || ws_bytes < lk_quadform_shared_workspace_bytes(B, C, Do, Dk)) {
    set_error("%s: workspace too small", what);
    return LK_EWORKSPACE;
  }
  const int split = qc_split(B, Do, Dk);
  float* partial = static_cast<float*>(ws);
  const dim3 grid((unsigned)(B * split));
  const bool v4 = (L % 4 == 0) && qc_aligned16(u) && qc_aligned16(v) && qc_b6_enabled();
#define LK_QC_CASE(CT)                                                                                              \
  case CT:                                                                                                          \
    if (v4)                                                                                                         \
      hipLaunchKernelGGL((quadform_conv_kernel<CT, MODE, 1>), grid, dim3(256), 0, stream, u, v, w0, w1, delta,      \
                         (int)C, (int)Do, (int)Dk, (int)L, split, partial, uss, ucs);                               \
    else                                                                                                            \
      hipLaunchKernelGGL((quadform_conv_kernel<CT, MODE, 0>), grid, dim3(256), 0, stream, u, v, w0, w1, delta,      \
                         (int)C, (int)Do, (int)Dk, (int)L, split, partial, uss, ucs);                               \
    break;
  switch (ct) {
    LK_QC_CASE(1)
    LK_QC_CASE(2)
    LK_QC_CASE(3)
    LK_QC_CASE(4)
    LK_QC_CASE(5)
    LK_QC_CASE(6)
    LK_QC_CASE(8)
    LK_QC_CASE(10)
  }
#undef LK_QC_CASE
  const int64_t total = B * (ct * (ct + 1) / 2);
  hipLaunchKernelGGL(quadform_conv_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, partial, B,
                     (int)C, ct, split, fvar);
  return check_launch(what);
}

extern "C" int lk_kron_quadform_shared_f32(const float* u, const float* v, const float* l1, const float* l2,
                                           const float* delta, int64_t B, int64_t C, int64_t Do, int64_t Dk, int64_t L,
                                           float* fvar, void* ws, size_t ws_bytes, void* stream) {
  LK_REQUIRE(u && v && l1 && l2 && delta && fvar && B >= 0 && C >= 1 && Do >= 1 && Dk >= 1 && L >= 1,
             "lk_kron_quadform_shared_f32: bad arguments");
  LK_REQUIRE(B * 64 < (1ll << 31) && C * L * Do < (1ll << 29) && L * Dk < (1ll << 29),
             "lk_kron_quadform_shared_f32: sizes out of range");
  return launch_quadform_conv<0>(u, v, l1, l2, delta, B, C, Do, Dk, L, fvar, ws, ws_bytes, (hipStream_t)stream,
                                 "lk_kron_quadform_shared_f32");
}

// lk_kron_quadform_shared_f32 for u stored SEED-major, [C][B][Do][L] — what one seed-batched reverse sweep (and the
// position-contiguous output of the rotation convolution over it) leaves in memory: no transposed copy is needed.
extern "C" int lk_kron_quadform_shared_seedmajor_f32(const float* u, const float* v, const float* l1, const float* l2,
                                                     const float* delta, int64_t B, int64_t C, int64_t Do, int64_t Dk,
                                                     int64_t L, float* fvar, void* ws, size_t ws_bytes, void* stream) {
  LK_REQUIRE(u && v && l1 && l2 && delta && fvar && B >= 0 && C >= 1 && Do >= 1 && Dk >= 1 && L >= 1,
             "lk_kron_quadform_shared_seedmajor_f32: bad arguments");
  LK_REQUIRE(B * 64 < (1ll << 31) && C * B * L * Do < (1ll << 31) && L * Dk < (1ll << 29),
             "lk_kron_quadform_shared_seedmajor_f32: sizes out of range");
  return launch_quadform_conv<0>(u, v, l1, l2, delta, B, C, Do, Dk, L, fvar, ws, ws_bytes, (hipStream_t)stream,
                                 "lk_kron_quadform_shared_seedmajor_f32", true);
}

// The Kronecker quadratic form of a weight-sharing layer on operands that arrive as fp16 planes (quadform_conv_planes_kernel):
// u_h / u_l [C][B][Do][L] with the scale u_sexp[0] (seed-major: the rotation convolution over a seed-batched sweep's cotangent),
// v_h / v_l [B][Dk][L] with v_sexp[n] per sample (v_nsexp = B) or v_sexp[0] (v_nsexp = 1) — both as lk_conv_nhwc_f16x2_planes
// leaves them.  L % 16 == 0, Do % 32 == 0, C <= 10; zero16: >= 16 zero bytes on the device (what padded outputs / columns
// stage).  fvar [B][C][C] +=.  Same workspace as lk_kron_quadform_shared_f32.
extern "C" int lk_kron_quadform_shared_planes_f16x2(const void* u_h, const void* u_l, const int* u_sexp, const void* v_h,
                                                    const void* v_l, const int* v_sexp, int64_t v_nsexp, const float* l1,
                                                    const float* l2, const float* delta, int64_t B, int64_t C, int64_t Do,
                                                    int64_t Dk, int64_t L, const void* zero16, float* fvar, void* ws,
                                                    size_t ws_bytes, void* stream_) {
  const char* what = "lk_kron_quadform_shared_planes_f16x2";
  LK_REQUIRE(u_h && u_l && u_sexp && v_h && v_l && v_sexp && l1 && l2 && delta && fvar && zero16 && B >= 0 && C >= 1 && Do >= 1 && Dk >= 1 && L >= 1,
             "lk_kron_quadform_shared_planes_f16x2: bad arguments");
  LK_REQUIRE(L % 16 == 0 && Do % 32 == 0 && (v_nsexp == 1 || v_nsexp == B), "lk_kron_quadform_shared_planes_f16x2: L % 16 == 0, Do % 32 == 0, v_nsexp in {1, B}");
  LK_REQUIRE(B * 64 < (1ll << 31) && C * B * L * Do < (1ll << 31) && L * Dk < (1ll << 29),
             "lk_kron_quadform_shared_planes_f16x2: sizes out of range");
  LK_REQUIRE(qc_aligned16(u_h) && qc_aligned16(u_l) && qc_aligned16(v_h) && qc_aligned16(v_l), "lk_kron_quadform_shared_planes_f16x2: 16-byte aligned planes");
  const int ct = qc_class_tile(C);
  if (ct == 0) {
    set_error("%s: more than 10 outputs are not supported by the fused kernel", what);
    return LK_EINVAL;
  }
  if (B == 0) return LK_OK;
  if (ws == nullptr || ws_bytes < lk_quadform_shared_workspace_bytes(B, C, Do, Dk)) {
    set_error("%s: workspace too small", what);
    return LK_EWORKSPACE;
  }
  hipStream_t stream = (hipStream_t)stream_;
  const int split = qc_split(B, Do, Dk);
  float* partial = static_cast<float*>(ws);
  const dim3 grid((unsigned)(B * split));
  const size_t w_bytes = (size_t)(Do + Dk) * sizeof(float);
  const int w_in_lds = w_bytes <= 40960 ? 1 : 0;  // (the eigenvalues behind the ring: ResNet-18's widest layer needs 20 KB)
#define LK_QP_LAUNCH(CT, OCC, SUB)                                                                                          \
  {                                                                                                                         \
    static bool attr_set = false;                                                                                           \
    if (!attr_set) {                                                                                                        \
      (void)hipFuncSetAttribute((const void*)quadform_conv_planes_kernel<CT, OCC, SUB>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                QcPlanesCfg<CT, OCC>::LDS + (OCC == 2 ? 24576 : 40960));                                      \
      attr_set = true;                                                                                                      \
    }                                                                                                                       \
    hipLaunchKernelGGL((quadform_conv_planes_kernel<CT, OCC, SUB>), grid, dim3(256),                                         \
                       (QcPlanesCfg<CT, OCC>::LDS + (w_in_lds ? w_bytes : 0)), stream, (const _Float16*)u_h, (const _Float16*)u_l,  \
                       u_sexp, (const _Float16*)v_h, (const _Float16*)v_l, v_sexp, (int)v_nsexp, l1, l2, delta, (int)B, (int)C,      \
                       (int)Do, (int)Dk, (int)L, split, partial, (const _Float16*)zero16, w_in_lds);                         \
  }
#define LK_QP_CASE(CT)                    \
  case CT:                                \
    if (sub) LK_QP_LAUNCH(CT, 2, true)    \
    else if (occ2) LK_QP_LAUNCH(CT, 2, false) \
    else LK_QP_LAUNCH(CT, 1, false)       \
    break;
  // Two workgroups per CU — two waves per SIMD — (round 6, `OCC = 2`): one wave's pair-sum arithmetic and load latency run beside
  // the other's MFMAs.  256 registers per wave: one-wide running pair sums, a two-stage ring (the partner covers what the
  // deeper ring covered); ten outputs spill 17 registers outside the chunk loop.  Measured on the c4 layers, ms per launch
  // (profiles/r06_quad_layers.log): 64 channels 0.42 -> 0.34, 128: 0.49 -> 0.38, 256: 0.65 -> 0.48, 512: 1.37 -> 1.10; a
  // predictive call 10.5 -> 8.8 ms of quadratic forms.  One per CU only where the eigenvalues do not fit beside two rings.
  const bool occ2 = w_bytes <= 24576;
  // one-chunk tiles (4 x 4 maps): the sub-tile form with two-wide pair sums (see the kernel; -DLK_QC_NO_SUB: development build)
#ifdef LK_QC_NO_SUB
  const bool sub = false;
#else
  const bool sub = occ2 && w_in_lds && L == 16;
#endif
  switch (ct) {
    LK_QP_CASE(1)
    LK_QP_CASE(2)
    LK_QP_CASE(3)
    LK_QP_CASE(4)
    LK_QP_CASE(5)
    LK_QP_CASE(6)
    LK_QP_CASE(8)
    LK_QP_CASE(10)
  }
#undef LK_QP_CASE
#undef LK_QP_LAUNCH
  const int64_t total = B * (ct * (ct + 1) / 2);
  hipLaunchKernelGGL(quadform_conv_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, partial, B,
                     (int)C, ct, split, fvar);
  return check_launch(what);
}

extern "C" int lk_diag_quadform_shared_f32(const float* u, const float* v, const float* var_w, int64_t B, int64_t C,
                                           int64_t Do, int64_t Dk, int64_t L, float* fvar, void* ws, size_t ws_bytes,
                                           void* stream) {
  LK_REQUIRE(u && v && var_w && fvar && B >= 0 && C >= 1 && Do >= 1 && Dk >= 1 && L >= 1,
             "lk_diag_quadform_shared_f32: bad arguments");
  LK_REQUIRE(B * 64 < (1ll << 31) && C * L * Do < (1ll << 29) && L * Dk < (1ll << 29) && Do * Dk < (1ll << 31),
             "lk_diag_quadform_shared_f32: sizes out of range");
  return launch_quadform_conv<1>(u, v, var_w, nullptr, nullptr, B, C, Do, Dk, L, fvar, ws, ws_bytes, (hipStream_t)stream,
                                 "lk_diag_quadform_shared_f32");
}

static int dg_split(int64_t B, int64_t Do, int64_t Dk) {
  const int64_t ntiles = ((Do + 31) / 32) * ((Dk + 127) / 128);
  int64_t want = (1024 + ntiles - 1) / ntiles;
  if (want < 1) want = 1;
  return (int)(want < B ? want : (B < 1 ? 1 : B));
}

extern "C" size_t lk_diag_ggn_shared_workspace_bytes(int64_t B, int64_t Do, int64_t Dk) {
  if (B < 0 || Do < 1 || Dk < 1) return 0;
  return (size_t)dg_split(B, Do, Dk) * Do * Dk * sizeof(float);
}

extern "C" int lk_diag_ggn_shared_f32(const float* u, const float* v, int64_t B, int64_t S, int64_t Do, int64_t Dk,
                                      int64_t L, float alpha, float* h, void* ws, size_t ws_bytes, void* stream_) {
  LK_REQUIRE(u && v && h && B >= 0 && S >= 1 && Do >= 1 && Dk >= 1 && L >= 1, "lk_diag_ggn_shared_f32: bad arguments");
  LK_REQUIRE(S * L * Do < (1ll << 29) && L * Dk < (1ll << 29) && Do * Dk < (1ll << 31),
             "lk_diag_ggn_shared_f32: sizes out of range");
  const int ct = qc_class_tile(S);
  if (ct == 0) {
    set_error("lk_diag_ggn_shared_f32: more than 10 seeds per call are not supported (chunk them)");
    return LK_EINVAL;
  }
  if (B == 0) return LK_OK;
  if (ws == nullptr || ws_bytes < lk_diag_ggn_shared_workspace_bytes(B, Do, Dk)) {
    set_error("lk_diag_ggn_shared_f32: workspace too small");
    return LK_EWORKSPACE;
  }
  hipStream_t stream = (hipStream_t)stream_;
  const int nsplit = dg_split(B, Do, Dk);
  const int64_t ntiles = ((Do + 31) / 32) * ((Dk + 127) / 128);
  LK_REQUIRE(ntiles * nsplit < (1ll << 31), "lk_diag_ggn_shared_f32: grid too large");
  float* partial = static_cast<float*>(ws);
  const dim3 grid((unsigned)(ntiles * nsplit));
  const bool v4 = (L % 4 == 0) && qc_aligned16(u) && qc_aligned16(v) && qc_b6_enabled();
#define LK_DG_CASE(CT)                                                                                               \
  case CT:                                                                                                           \
    if (v4)                                                                                                          \
      hipLaunchKernelGGL((diag_ggn_shared_kernel<CT, true>), grid, dim3(256), 0, stream, u, v, (int)B, (int)S,       \
                         (int)Do, (int)Dk, (int)L, nsplit, partial);                                                 \
    else                                                                                                             \
      hipLaunchKernelGGL((diag_ggn_shared_kernel<CT, false>), grid, dim3(256), 0, stream, u, v, (int)B, (int)S,      \
                         (int)Do, (int)Dk, (int)L, nsplit, partial);                                                 \
    break;
  switch (ct) {
    LK_DG_CASE(1)
    LK_DG_CASE(2)
    LK_DG_CASE(3)
    LK_DG_CASE(4)
    LK_DG_CASE(5)
    LK_DG_CASE(6)
    LK_DG_CASE(8)
    LK_DG_CASE(10)
  }
#undef LK_DG_CASE
  const int64_t width = Do * Dk;
  hipLaunchKernelGGL(diag_ggn_shared_reduce_kernel, dim3((unsigned)((width + 255) / 256)), dim3(256), 0, stream, partial,
                     width, nsplit, alpha, h);
  return check_launch("lk_diag_ggn_shared_f32");
}
