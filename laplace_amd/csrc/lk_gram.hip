// Gram / Kronecker-factor accumulation on gfx950:  C += alpha * X^T X  with exact-fp32 MFMA.
//
// One MFMA engine, three operand loaders (the virtual row matrix X[K][n] is never materialised for
// the NT and CONV forms):
//   MODE_TN   X[K][n] row-major (ldx)                           nn.Linear inputs / output grads
//   MODE_NT   x[seg][nb][n][L]; row k=(b,l) -> x[b][:,l]        NCHW conv output grads (one segment per
//             backward seed, passed as a pointer table so the per-seed gradients are never stacked)
//   MODE_CONV x[B][H][W][Cin] NHWC; row k=(b,oh,ow), col=(dy,dx,ci)  implicit im2col of conv inputs
//
// Work decomposition: grid.x = upper-triangular tile pairs (bi <= bj) of the n x n output, grid.y =
// split-K slices.  Every workgroup writes its partial tile to a workspace slab (deterministic, no
// atomics); gram_reduce_kernel sums the slabs, scales by alpha, accumulates into C and mirrors the
// off-diagonal tiles.  Tile configurations:
//   WIDE  192x192 tile, BK=16, 4 waves as 2x2, each wave 3x3 MFMA 32x32x2 tiles (144 acc VGPRs): 48 flop
//          per staged byte instead of 32 -- the global->LDS path (~10 B/clk/CU), not the MFMA pipe, is what
//          limits the 128-tile; used when 192 | n (every 3x3-conv A factor of a ResNet: 576 ... 4608)
//   BIG   128x128 tile, BK=16, 4 waves as 2x2, each wave 2x2 MFMA 32x32x2 tiles (64 acc VGPRs)
//   SMALL  64x64  tile (n <= 64), BK=64, 4 waves as 2x2, each wave one 32x32 tile: tiny-n / huge-K
//          factors (conv G with 64 channels) keep all four SIMDs busy through deep split-K.
// Interior tiles run a branch-free inner loop (4 LDS reads : 4 MFMAs per k-pair); only edge tiles
// take the predicated path (wave-uniform scalar predicates).
//
// Reference being replaced: the A^T A / G^T G products inside curvlinops' KFAC as consumed by
// laplace/curvature/curvlinops.py:55-108, and the einsums of laplace/curvature/curvature.py:406,409,491.
#include <cstdlib>

#include "lk_common.h"

namespace lk {

// MODE_TNP: TN loader, tiles taken from a table (colA, colB, output offset): block-sparse products into compact blocks
enum { MODE_TN = 0, MODE_NT = 1, MODE_CONV = 2, MODE_XCORR = 3, MODE_TNP = 4, MODE_NTB = 5 };
// MODE_NTB: the NT product with the operands split into three bf16 pieces in LDS and six bf16 MFMAs per fp32 product
// (see `split3` below): same loader, same accumulators, same epilogue as MODE_NT.
__host__ __device__ constexpr bool is_nt(int mode) { return mode == MODE_NT || mode == MODE_NTB; }
#ifdef LK_GRAM_TRACE  // development build (tools/gram_trace.py): per-phase cycle counts of one wave of the last launch
__device__ long long g_gram_trace[5];
#endif
constexpr int MAX_SEG = 16;

struct GramGeom {
  const float* x;
  int64_t K;    // virtual rows
  int n;        // columns (= output dim)
  int64_t ldx;  // TN
  const int* tiles;  // TNP: [ntiles][3] = column offset of the A panel, of the B panel, output offset (floats)
  int ldc;           // TNP: leading dimension of an output block
  int slab_accumulate;  // split-K slabs are persistent accumulators: `slab += partial tile` (LK_GRAM_SLABS_PERSIST)
  int L, Lp;    // NT: positions per image, padded to a multiple of BK
  int seg_nb;   // NT: images per segment
  int nseg;     // NT: number of segments (1 = plain tensor)
  const float* seg[MAX_SEG];                          // NT: segment base pointers
  int H, W, Cin, OH, OW, kw, sh, sw, ph, pw, dh, dw;  // CONV
  FastDiv div_ohw, div_ow;                             // CONV: row index -> (b, oh, ow)
  // XCORR: rectangular  R[ci][(s, cj)] = sum_{b, q in region} x[b, q, ci] * x~[b, q + shift_s, cj]   (NHWC, x~ = x
  // zero-extended); rows enumerate (b, q) over the region [reg_h0, reg_h0+reg_h) x [reg_w0, reg_w0+reg_w)
  // Up to MAX_REG regions are processed by one launch (blockIdx.z = region), each with its own output matrix.
  int nA, nB;                    // output rows (Cin) and columns (nshift * Cin)
  int nreg;
  int reg_h0[8], reg_w0[8];
  int64_t reg_K[8];              // rows of region r = B * reg_h * reg_w
  signed char sdy[25], sdx[25];  // shift table
  FastDiv div_reg[8], div_regw[8];  // row -> (b, r), r -> (rh, rw)
};
constexpr int MAX_REG = 8;

// CFG_BIG24 / CFG_BIG32: the 128x128 tile with deeper chunks (24 / 32 virtual rows per barrier instead of 16)
// CFG_SMALL16: the 64x64 tile with 16-row chunks (17 KB of LDS: many resident workgroups for the short MODE_TNP launches)
enum { CFG_SMALL = 0, CFG_BIG = 1, CFG_WIDE = 2, CFG_BIG24 = 3, CFG_BIG32 = 4, CFG_SMALL16 = 5 };
template <int CFG>
struct Cfg {
  static constexpr bool SMALL = (CFG == CFG_SMALL);
  static constexpr int TW = (SMALL || CFG == CFG_SMALL16) ? 1 : (CFG == CFG_WIDE ? 3 : 2);  // MFMA 32x32 tiles per wave
  static constexpr int WT = 32 * TW;           // wave tile edge
  static constexpr int T = 2 * WT;             // output tile edge (64 / 128 / 192), 4 waves as 2x2
  static constexpr int BK = SMALL ? 64 : (CFG == CFG_BIG24 ? 24 : (CFG == CFG_BIG32 ? 32 : 16));  // rows per chunk
  static constexpr int LDP = T + 4;            // LDS row pitch (floats), keeps 16-B alignment
  static constexpr int EPT = T * BK / 256;     // staged elements per thread per panel
};

// Linear staging index idx in [0, T*BK/VEC) -> (krow, col) of the first element.
//   TN / CONV: consecutive idx walk along a row (columns are contiguous in memory)
//   NT       : consecutive idx walk along k (positions are contiguous in memory)
template <int MODE, int VEC, int CFG>
__device__ __forceinline__ void stage_coord(int idx, int& krow, int& col) {
  using C = Cfg<CFG>;
  if (is_nt(MODE)) {
    constexpr int PER_COL = C::BK / VEC;
    col = idx / PER_COL;
    krow = (idx % PER_COL) * VEC;
  } else {
    constexpr int PER_ROW = C::T / VEC;
    krow = idx / PER_ROW;
    col = (idx % PER_ROW) * VEC;
  }
}

// Per-thread, per-panel column context, computed once before the K loop (no div/mod in the loop).
template <int MODE, int VEC, int CFG>
struct ColCtx {
  static constexpr int NL = Cfg<CFG>::EPT / VEC;
  int64_t off[NL];     // TN: column; NT: column*L; CONV: ci
  int dy[NL], dx[NL];  // CONV: input offset of the patch element, padding folded in
  bool ok[NL];
};

template <int MODE, int VEC, int CFG>
__device__ __forceinline__ void make_colctx(const GramGeom& g, int col0, int tid, bool isB,
                                            ColCtx<MODE, VEC, CFG>& cc) {
  constexpr int NL = ColCtx<MODE, VEC, CFG>::NL;
  const int ncols = (MODE == MODE_XCORR) ? (isB ? g.nB : g.nA) : g.n;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int krow, col;
    stage_coord<MODE, VEC, CFG>(tid + 256 * i, krow, col);
    const int c = col0 + col;
    cc.ok[i] = c < ncols;
    cc.dy[i] = 0;
    cc.dx[i] = 0;
    if (MODE == MODE_XCORR) {
      int sidx = 0, ch = c;
      if (isB) {
        sidx = c / g.Cin;
        ch = c - sidx * g.Cin;
        if (sidx > 24) sidx = 24;  // only reachable for padded (invalid) columns
        cc.dy[i] = g.sdy[sidx];
        cc.dx[i] = g.sdx[sidx];
      }
      cc.off[i] = ch;
    } else if (MODE == MODE_TN || MODE == MODE_TNP) {
      cc.off[i] = c;
    } else if (is_nt(MODE)) {
      cc.off[i] = (int64_t)c * g.L;
    } else {
      const int d = c / g.Cin, ci = c - d * g.Cin;
      const int dyy = d / g.kw, dxx = d - dyy * g.kw;
      cc.dy[i] = dyy * g.dh - g.ph;
      cc.dx[i] = dxx * g.dw - g.pw;
      cc.off[i] = ci;
    }
  }
}

template <int MODE, int VEC, int CFG>
__device__ __forceinline__ void load_panel(const GramGeom& g, int64_t k0, int tid, int reg,
                                           const ColCtx<MODE, VEC, CFG>& cc, float (&st)[Cfg<CFG>::EPT]) {
  constexpr int NL = ColCtx<MODE, VEC, CFG>::NL;
  // chunk-uniform part (NT: a chunk never straddles images because Lp % BK == 0)
  const float* nt_base = nullptr;
  int nt_l0 = 0;
  if (is_nt(MODE)) {
    if (k0 < g.K) {
      const int64_t b = k0 / g.Lp;
      nt_l0 = (int)(k0 - b * g.Lp);
      const int sgi = (int)(b / g.seg_nb);
      const int bb = (int)(b - (int64_t)sgi * g.seg_nb);
      nt_base = g.seg[sgi] + (int64_t)bb * g.n * g.L + nt_l0;
    }
  }
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int krow, col;
    stage_coord<MODE, VEC, CFG>(tid + 256 * i, krow, col);
    bool valid = cc.ok[i];
    const float* p = g.x;
    if (MODE == MODE_TN || MODE == MODE_TNP) {
      const int64_t k = k0 + krow;
      valid = valid && (k < g.K);
      p = g.x + k * g.ldx + cc.off[i];
    } else if (is_nt(MODE)) {
      valid = valid && (nt_base != nullptr) && (nt_l0 + krow < g.L);
      p = nt_base + cc.off[i] + krow;
    } else if (MODE == MODE_XCORR) {
      const int k = (int)k0 + krow;
      const int b = fdiv(k, g.div_reg[reg]);
      const int r = k - b * g.div_reg[reg].d;
      const int rh = fdiv(r, g.div_regw[reg]), rw = r - rh * g.div_regw[reg].d;
      const int ih = g.reg_h0[reg] + rh + cc.dy[i];
      const int iw = g.reg_w0[reg] + rw + cc.dx[i];
      valid = valid && (k < g.reg_K[reg]) && (ih >= 0) && (ih < g.H) && (iw >= 0) && (iw < g.W);
      p = g.x + (((int64_t)b * g.H + ih) * g.W + iw) * g.Cin + cc.off[i];
    } else {
      const int k = (int)k0 + krow;  // conv: K < 2^31 (checked on the host)
      const int b = fdiv(k, g.div_ohw);
      const int r = k - b * g.div_ohw.d;
      const int oh = fdiv(r, g.div_ow), ow = r - oh * g.div_ow.d;
      const int ih = oh * g.sh + cc.dy[i];
      const int iw = ow * g.sw + cc.dx[i];
      valid = valid && (k < g.K) && (ih >= 0) && (ih < g.H) && (iw >= 0) && (iw < g.W);
      p = g.x + (((int64_t)b * g.H + ih) * g.W + iw) * g.Cin + cc.off[i];
    }
    if (VEC == 4) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (valid) v = *reinterpret_cast<const f32x4*>(p);
      st[i * 4 + 0] = v.x;
      st[i * 4 + 1] = v.y;
      st[i * 4 + 2] = v.z;
      st[i * 4 + 3] = v.w;
    } else {
      st[i] = valid ? *p : 0.f;
    }
  }
}

// ---- fp32 products on the bf16 matrix cores (MODE_NTB) ------------------------------------------------------
// x = h + m + l exactly, each piece a bf16 (8 significant bits; truncation makes every subtraction exact), and
//   x y ~= h h' + h m' + m h' + m m' + h l' + l h'      (dropped: m l', l m', l l' <= 3 * 2^-24 |x y|)
// i.e. six v_mfma_f32_32x32x16_bf16 (32 cycles, K = 16) replace eight v_mfma_f32_32x32x2_f32 (64 cycles, K = 2) per
// 16 rows of a 32x32 tile at fp32-level accuracy: 192 instead of 512 matrix-pipe cycles.  The operand is split ONCE,
// by the thread that stages it; a panel holds [piece][column][k] bf16 with k contiguous, which is how the NT loader
// reads memory anyway (positions of one channel) and what a lane of the bf16 MFMA wants (8 consecutive k).
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int CFG>
struct NtbCfg {
  using C = Cfg<CFG>;
  static constexpr int BKP = C::BK == 16 ? 16 : C::BK + 8;  // k pitch (bf16): 144-B rows keep ds_read_b128 conflict-free
  static constexpr int PIECE = C::T * BKP * 2;             // bytes of one piece of a panel
  static constexpr int PANEL_F = 3 * PIECE / 4;            // panel size in floats (the LDS arena is float-typed)
};

__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  h = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xffff0000u;
  l = __float_as_uint(r1 - __uint_as_float(m));  // at most 8 significant bits are left: exact in bf16
}
__device__ __forceinline__ unsigned pack_hi16(unsigned lo_elem, unsigned hi_elem) {
  return (lo_elem >> 16) | (hi_elem & 0xffff0000u);
}

template <int VEC, int CFG>
__device__ __forceinline__ void store_panel_ntb(float* panel, int tid, const float (&st)[Cfg<CFG>::EPT]) {
  using C = Cfg<CFG>;
  using N = NtbCfg<CFG>;
  constexpr int NL = C::EPT / VEC;
  char* base = reinterpret_cast<char*>(panel);
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int krow, col;
    stage_coord<MODE_NTB, VEC, CFG>(tid + 256 * i, krow, col);
    char* dst = base + (col * N::BKP + krow) * 2;
    if (VEC == 4) {
      unsigned h[4], m[4], l[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split3(st[i * 4 + j], h[j], m[j], l[j]);
      *reinterpret_cast<u32x2*>(dst) = u32x2{pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3])};
      *reinterpret_cast<u32x2*>(dst + N::PIECE) = u32x2{pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3])};
      *reinterpret_cast<u32x2*>(dst + 2 * N::PIECE) = u32x2{pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3])};
    } else {
      unsigned h, m, l;
      split3(st[i], h, m, l);
      *reinterpret_cast<unsigned short*>(dst) = (unsigned short)(h >> 16);
      *reinterpret_cast<unsigned short*>(dst + N::PIECE) = (unsigned short)(m >> 16);
      *reinterpret_cast<unsigned short*>(dst + 2 * N::PIECE) = (unsigned short)(l >> 16);
    }
  }
}

// One chunk of the split product for this wave: per 16 rows, 3 ds_read_b128 per 32-column operand tile and 6 MFMAs
// per 32x32 output tile.  pA / pB point at this lane's (column, k-half) inside piece 0 of the panels.
template <int CFG, bool FULL>
__device__ __forceinline__ void compute_chunk_ntb(const char* __restrict__ pA, const char* __restrict__ pB,
                                                  f32x16 (&acc)[Cfg<CFG>::TW][Cfg<CFG>::TW], int am, int an) {
  using C = Cfg<CFG>;
  using N = NtbCfg<CFG>;
  constexpr int TW = C::TW;
#pragma unroll
  for (int k16 = 0; k16 < C::BK / 16; ++k16) {
    bf16x8 a[3][TW], b[3][TW];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        a[p][t] = *reinterpret_cast<const bf16x8*>(pA + p * N::PIECE + (t * 32 * N::BKP + k16 * 16) * 2);
        b[p][t] = *reinterpret_cast<const bf16x8*>(pB + p * N::PIECE + (t * 32 * N::BKP + k16 * 16) * 2);
      }
#pragma unroll
    for (int tm = 0; tm < TW; ++tm)
#pragma unroll
      for (int tn = 0; tn < TW; ++tn)
        if (FULL || (tm < am && tn < an)) {
          f32x16 c = acc[tm][tn];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][tm], b[0][tn], c, 0, 0, 0);  // small terms first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][tm], b[2][tn], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][tm], b[1][tn], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][tm], b[0][tn], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][tm], b[1][tn], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][tm], b[0][tn], c, 0, 0, 0);
          acc[tm][tn] = c;
        }
  }
}

template <int MODE, int VEC, int CFG>
__device__ __forceinline__ void store_panel(float* panel, int tid, const float (&st)[Cfg<CFG>::EPT]) {
  using C = Cfg<CFG>;
  constexpr int NL = C::EPT / VEC;
  if (MODE == MODE_NTB) {
    store_panel_ntb<VEC, CFG>(panel, tid, st);
    return;
  }
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int krow, col;
    stage_coord<MODE, VEC, CFG>(tid + 256 * i, krow, col);
    if (VEC == 4 && MODE != MODE_NT) {
      f32x4 v = {st[i * 4], st[i * 4 + 1], st[i * 4 + 2], st[i * 4 + 3]};
      *reinterpret_cast<f32x4*>(panel + krow * C::LDP + col) = v;
    } else if (VEC == 4) {  // NT: the four values are consecutive k-rows of one column
#pragma unroll
      for (int j = 0; j < 4; ++j) panel[(krow + j) * C::LDP + col] = st[i * 4 + j];
    } else {
      panel[krow * C::LDP + col] = st[i];
    }
  }
}

__device__ __forceinline__ void pair_to_tiles(int p, int nbt, int& bi, int& bj) {
  bi = 0;
  int rowlen = nbt;
  while (p >= rowlen) {
    p -= rowlen;
    ++bi;
    --rowlen;
  }
  bj = bi + p;
}

// One chunk of MFMA work for this wave.  FULL: every 32x32 sub-tile of the wave is inside the matrix
// (branch-free); otherwise (am, an) = number of active sub-tiles per dim, wave-uniform scalars.
template <int CFG, bool FULL>
__device__ __forceinline__ void compute_chunk(const float* __restrict__ pA, const float* __restrict__ pB,
                                              f32x16 (&acc)[Cfg<CFG>::TW][Cfg<CFG>::TW], int am, int an) {
  using C = Cfg<CFG>;
  constexpr int TW = C::TW;
  constexpr int KS = C::BK / 2;  // k-steps (two virtual rows each) per chunk
  // Operands are software-pipelined through two register sets: the LDS reads of k-step kk+1 are issued BEFORE the
  // MFMAs of k-step kk, so their latency hides behind TW*TW*64 cycles of matrix work instead of stalling the wave
  // at every k-step (with a single operand set the compiler re-used the same registers and waited on lgkmcnt(0)
  // right before each MFMA group).
  float a[2][TW], b[2][TW];
#pragma unroll
  for (int t = 0; t < TW; ++t) {
    a[0][t] = pA[t * 32];
    b[0][t] = pB[t * 32];
  }
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    const int cur = kk & 1, nxt = cur ^ 1;
    if (kk + 1 < KS) {
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        a[nxt][t] = pA[2 * (kk + 1) * C::LDP + t * 32];
        b[nxt][t] = pB[2 * (kk + 1) * C::LDP + t * 32];
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMA group (the scheduler sinks it otherwise)
#pragma unroll
    for (int tm = 0; tm < TW; ++tm)
#pragma unroll
      for (int tn = 0; tn < TW; ++tn)
        if (FULL || (tm < am && tn < an))
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][tm], b[cur][tn], acc[tm][tn], 0, 0, 0);
  }
}

// Whole main loop + epilogue of one wave, specialised on FULL so that interior tiles get a branch-free
// MFMA loop (the dispatch on `full` happens ONCE per wave, outside the loop).
template <int MODE, int VEC, int CFG, bool FULL>
__device__ __forceinline__ void gram_body(const GramGeom& g, float* __restrict__ smem, float* __restrict__ slab,
                                          int tid, int wm, int wn, int lo, int hi, bool diag, int colA, int colB,
                                          int c_begin, int c_end, int am, int an, float* __restrict__ Cdirect,
                                          float alpha, int reg) {
  using C = Cfg<CFG>;
  constexpr int PANEL = MODE == MODE_NTB ? NtbCfg<CFG>::PANEL_F : C::BK * C::LDP;
  constexpr int TW = C::TW;
  constexpr int NP = (C::SMALL && MODE != MODE_XCORR && MODE != MODE_TNP) ? 1 : 2;

  f32x16 acc[TW][TW];
#pragma unroll
  for (int a = 0; a < TW; ++a)
#pragma unroll
    for (int b = 0; b < TW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  ColCtx<MODE, VEC, CFG> ccA, ccB;
  make_colctx<MODE, VEC, CFG>(g, colA, tid, false, ccA);
  make_colctx<MODE, VEC, CFG>(g, colB, tid, true, ccB);

  float stA[C::EPT], stB[C::EPT];
  if (c_begin < c_end) {
    load_panel<MODE, VEC, CFG>(g, (int64_t)c_begin * C::BK, tid, reg, ccA, stA);
    if (!diag) load_panel<MODE, VEC, CFG>(g, (int64_t)c_begin * C::BK, tid, reg, ccB, stB);
    store_panel<MODE, VEC, CFG>(smem, tid, stA);
    if (!diag) store_panel<MODE, VEC, CFG>(smem + PANEL, tid, stB);
  }
  __syncthreads();

  // this lane's operand offsets inside a panel: row = hi (k parity), column = wave tile origin + lo
  const int offA = hi * C::LDP + wm * C::WT + lo;
  const int offB = hi * C::LDP + wn * C::WT + lo;

  int cur = 0;
#ifdef LK_GRAM_TRACE
  const bool tracer = (blockIdx.x == 1 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0);
  long long tr[5] = {0, 0, 0, 0, 0};
#define LK_TSTAMP(v)                      \
  __builtin_amdgcn_sched_barrier(0);      \
  const long long v = __builtin_amdgcn_s_memtime(); \
  __builtin_amdgcn_sched_barrier(0)
#else
#define LK_TSTAMP(v)
#endif
  for (int c = c_begin; c < c_end; ++c) {
    const bool more = (c + 1) < c_end;
    LK_TSTAMP(t0);
    if (more) {
      load_panel<MODE, VEC, CFG>(g, (int64_t)(c + 1) * C::BK, tid, reg, ccA, stA);
      if (!diag) load_panel<MODE, VEC, CFG>(g, (int64_t)(c + 1) * C::BK, tid, reg, ccB, stB);
    }
    LK_TSTAMP(t1);
    const float* pA = smem + cur * NP * PANEL;
    const float* pB = diag ? pA : pA + PANEL;
    if (MODE == MODE_NTB) {
      constexpr int BKP = NtbCfg<CFG>::BKP;
      compute_chunk_ntb<CFG, FULL>(reinterpret_cast<const char*>(pA) + ((wm * C::WT + lo) * BKP + 8 * hi) * 2,
                                   reinterpret_cast<const char*>(pB) + ((wn * C::WT + lo) * BKP + 8 * hi) * 2, acc, am, an);
    } else {
      compute_chunk<CFG, FULL>(pA + offA, pB + offB, acc, am, an);
    }
    LK_TSTAMP(t2);
    if (more) {
      float* nx = smem + (cur ^ 1) * NP * PANEL;
      store_panel<MODE, VEC, CFG>(nx, tid, stA);
      if (!diag) store_panel<MODE, VEC, CFG>(nx + PANEL, tid, stB);
    }
    LK_TSTAMP(t3);
    __syncthreads();
    cur ^= 1;
#ifdef LK_GRAM_TRACE
    __builtin_amdgcn_sched_barrier(0);
    const long long t4 = __builtin_amdgcn_s_memtime();
    tr[0] += t1 - t0, tr[1] += t2 - t1, tr[2] += t3 - t2, tr[3] += t4 - t3, tr[4] += 1;
#endif
  }
#ifdef LK_GRAM_TRACE
  if (tracer)
    for (int i = 0; i < 5; ++i) g_gram_trace[i] = tr[i];
#endif
#undef LK_TSTAMP

  if (Cdirect != nullptr) {
    // single split, upper-only accumulation: C += alpha * tile straight from the accumulators
    // (32 consecutive columns per lane group = 128-byte segments); no slab, no reduce launch.
    // Two phases per 32x32 sub-tile -- 16 independent loads, then 16 stores: written as `C[i] += v` in one loop the
    // compiler must assume the store aliases the next load and serialises 16 memory round trips.
    const int ldc = MODE == MODE_TNP ? g.ldc : g.n;
    const int row0 = (MODE == MODE_TNP ? 0 : colA) + wm * C::WT + 4 * hi;  // TNP: tile-local rows in a compact block
    const int col0 = (MODE == MODE_TNP ? 0 : colB) + wn * C::WT + lo;
#pragma unroll
    for (int tm = 0; tm < TW; ++tm)
#pragma unroll
      for (int tn = 0; tn < TW; ++tn) {
        float* base = Cdirect + (int64_t)(row0 + tm * 32) * ldc + col0 + tn * 32;
        const int col = col0 + tn * 32;
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          const bool ok = FULL || MODE == MODE_TNP || (row0 + tm * 32 + dr < g.n && col < g.n);
          old[r] = ok ? base[(int64_t)dr * ldc] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          const bool ok = FULL || MODE == MODE_TNP || (row0 + tm * 32 + dr < g.n && col < g.n);
          if (ok) base[(int64_t)dr * ldc] = old[r] + alpha * acc[tm][tn][r];
        }
      }
    return;
  }
  // epilogue: partial tile -> slab (overwritten, or accumulated when the slabs persist across launches)
#pragma unroll
  for (int tm = 0; tm < TW; ++tm)
#pragma unroll
    for (int tn = 0; tn < TW; ++tn) {
      float* base = slab + (wm * C::WT + tm * 32 + 4 * hi) * C::T + wn * C::WT + tn * 32 + lo;
      float old[16];
      if (g.slab_accumulate) {
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = base[((r & 3) + 8 * (r >> 2)) * C::T];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
        base[((r & 3) + 8 * (r >> 2)) * C::T] = g.slab_accumulate ? old[r] + acc[tm][tn][r] : acc[tm][tn][r];
    }
}

template <int MODE, int VEC, int CFG>
__global__ __launch_bounds__(256) void gram_kernel(GramGeom g, float* __restrict__ slabs, int nbt, int npairs,
                                                   int chunks_per_split, int nchunks, float* __restrict__ Cdirect,
                                                   float alpha) {
  using C = Cfg<CFG>;
  constexpr int TW = C::TW;
  // dynamic LDS, 2 * NP * BK * LDP floats: [buf][panel A|B]; NP = 1 for the SMALL Gram (single diagonal tile, B
  // aliases A), sized by cfg_lds_bytes() on the host
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lo = lane & 31, hi = lane >> 5;
  constexpr bool RECT = (MODE == MODE_XCORR);
  int bi, bj, split, pidx;
  int tnp_colA = 0, tnp_colB = 0, tnp_out = 0;
  if (MODE == MODE_TNP) {  // tile table: arbitrary column windows of X, output into a compact block
    tnp_colA = g.tiles[3 * blockIdx.x];
    tnp_colB = g.tiles[3 * blockIdx.x + 1];
    tnp_out = g.tiles[3 * blockIdx.x + 2];
    bi = 0, bj = 1;  // (never a diagonal tile: both panels are loaded)
    split = blockIdx.y;
    pidx = blockIdx.x;
  } else if (RECT) {  // all (row tile, column tile) pairs of the rectangular output; nbt = column tiles
    bi = blockIdx.x / nbt;
    bj = blockIdx.x - bi * nbt;
    split = blockIdx.y;
    pidx = blockIdx.x;
  } else {
    // Natural order: workgroup w runs on XCD w % 8, so every XCD sees every 8th tile pair -- all row panels and
    // one residue class of column panels (~70 panels per chunk step, comfortably inside its 4 MB L2; PMC: ~0.2-0.4
    // TB/s reach the fabric).  A contiguous-range-per-XCD remap with 8x8 super-tiles was measured 10-30 % SLOWER
    // (96 workgroups hammering the same ~20 panels), so the simple mapping stays.
    pair_to_tiles(blockIdx.x, nbt, bi, bj);
    split = blockIdx.y;
    pidx = blockIdx.x;
  }
  const bool diag = !RECT && MODE != MODE_TNP && (C::SMALL || (bi == bj));
  const int colA = MODE == MODE_TNP ? tnp_colA : bi * C::T, colB = MODE == MODE_TNP ? tnp_colB : bj * C::T;
  const int wm = wave >> 1, wn = wave & 1;

  // number of active 32x32 sub-tiles of this wave along each dim (wave-uniform scalars)
  int am = ((RECT ? g.nA : g.n) - (colA + wm * C::WT) + 31) / 32;
  int an = ((RECT ? g.nB : g.n) - (colB + wn * C::WT) + 31) / 32;
  am = am < 0 ? 0 : (am > TW ? TW : am);
  an = an < 0 ? 0 : (an > TW ? TW : an);

  // MODE_XCORR: blockIdx.z = region; regions share the split geometry of the longest one and simply run out
  // of chunks earlier (their remaining slabs stay zero)
  const int reg = RECT ? (int)blockIdx.z : 0;
  const int my_chunks = RECT ? (int)((g.reg_K[reg] + C::BK - 1) / C::BK) : nchunks;
  const int c_begin = split * chunks_per_split;
  const int c_end = min(my_chunks, c_begin + chunks_per_split);
  float* slab = slabs + (((int64_t)reg * gridDim.y + split) * npairs + pidx) * (C::T * C::T);

  if (MODE == MODE_TNP) Cdirect += tnp_out;
  // FULL = the wave's whole WT x WT patch lies inside the matrix: branch-free MFMA loop and unguarded direct
  // epilogue.  (am == TW alone is NOT enough: am counts partially covered 32x32 sub-tiles too, and an unguarded
  // `C[row][col] += 0` on the rows / columns beyond n is an out-of-bounds read-modify-write.)
  const bool inside = (colA + (wm + 1) * C::WT <= (RECT ? g.nA : g.n)) && (colB + (wn + 1) * C::WT <= (RECT ? g.nB : g.n));
  // Every wave executes the same number of barriers on either path.
  if (inside) {
    gram_body<MODE, VEC, CFG, true>(g, smem, slab, tid, wm, wn, lo, hi, diag, colA, colB, c_begin, c_end, am, an,
                                      Cdirect, alpha, reg);
  } else {
    gram_body<MODE, VEC, CFG, false>(g, smem, slab, tid, wm, wn, lo, hi, diag, colA, colB, c_begin, c_end, am, an,
                                       Cdirect, alpha, reg);
  }
}

// Sum slabs, scale, accumulate into C, mirror off-diagonal tiles.
// grid = (npairs, (T/64)^2 * (64/rpw)); each workgroup owns `rpw` rows x 64 cols of one tile.
// First level of a two-level slab reduction: group g sums slabs [g*SG, (g+1)*SG) into slab g*SG (in place).
__global__ __launch_bounds__(256) void gram_prereduce_kernel(float* __restrict__ slabs, int nslabs, int64_t slab_elems,
                                                             int SG) {
  slabs += (int64_t)blockIdx.z * nslabs * slab_elems;  // region (MODE_XCORR), 0 otherwise
  const int first = blockIdx.y * SG;
  const int last = min(nslabs, first + SG);
  for (int64_t idx = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; idx < slab_elems;
       idx += (int64_t)gridDim.x * 256 * 4) {
    f32x4 s = *reinterpret_cast<const f32x4*>(slabs + (int64_t)first * slab_elems + idx);
    for (int k = first + 1; k < last; ++k) s += *reinterpret_cast<const f32x4*>(slabs + (int64_t)k * slab_elems + idx);
    *reinterpret_cast<f32x4*>(slabs + (int64_t)first * slab_elems + idx) = s;
  }
}

__global__ __launch_bounds__(256) void gram_reduce_kernel(const float* __restrict__ slabs, int nslabs, int slab_stride,
                                                          int npairs, int T, int nbt, float alpha,
                                                          float* __restrict__ Cmat, int n, int mirror, int rpw,
                                                          int rect_rows, int nslabs_total) {
  // rect_rows > 0: rectangular output [rect_rows][n] (n = columns = leading dimension), tiles enumerated
  // row-major with nbt column tiles, no mirroring
  __shared__ float tile[64][65];
  const int tid = threadIdx.x;
  int bi, bj;
  if (rect_rows > 0) {
    bi = blockIdx.x / nbt;
    bj = blockIdx.x - bi * nbt;
  } else {
    pair_to_tiles(blockIdx.x, nbt, bi, bj);
  }
  const int nrows = rect_rows > 0 ? rect_rows : n;
  if (rect_rows > 0) {  // region (MODE_XCORR): its own slab range and its own output matrix
    slabs += (int64_t)blockIdx.z * nslabs_total * npairs * (int64_t)T * T;
    Cmat += (int64_t)blockIdx.z * rect_rows * n;
  }
  const int subs = T / 64;
  const int slices = 64 / rpw;
  const int sub = blockIdx.y / slices, slice = blockIdx.y % slices;
  const int sr = sub / subs, sc = sub % subs;
  const int cx = tid & 63, ry = tid >> 6;
  const int64_t tile_elems = (int64_t)T * T;
  const bool do_mirror = mirror && (bi != bj);
  const int row0 = sr * 64 + slice * rpw;  // first tile row of this workgroup
  if (bi * T + row0 >= nrows || bj * T + sc * 64 >= n) return;  // entirely padding
  for (int lr = ry; lr < rpw; lr += 4) {
    const float* p = slabs + (int64_t)blockIdx.x * tile_elems + (int64_t)(row0 + lr) * T + sc * 64 + cx;
    float s = 0.f;
    for (int k = 0; k < nslabs; ++k) s += p[(int64_t)k * slab_stride * npairs * tile_elems];
    s *= alpha;
    const int r = bi * T + row0 + lr, c = bj * T + sc * 64 + cx;
    if (r < nrows && c < n) Cmat[(int64_t)r * n + c] += s;
    if (do_mirror) tile[lr][cx] = s;
  }
  if (do_mirror) {
    __syncthreads();
    for (int idx = tid; idx < rpw * 64; idx += 256) {
      const int srow = idx % rpw, scol = idx / rpw;
      const int r = bj * T + sc * 64 + scol, c = bi * T + row0 + srow;
      if (r < n && c < n) Cmat[(int64_t)r * n + c] += tile[srow][scol];
    }
  }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

struct GramPlan {
  int cfg;
  int T, BK, nbt, npairs, nchunks, nsplit, chunks_per_split, nslabs, rpw;
  size_t ws_bytes;
};

static void finish_plan(GramPlan& p) {
  // Split-K selection: workgroups run in "rounds" of (256 CUs x resident workgroups per CU); choose the split
  // count that minimises  rounds x (chunks per split + fixed per-workgroup overhead)  -- a launch of 1026
  // workgroups on 768 slots costs two full rounds (PMC: 83 % CU residency before this rule).
  const int occ = p.cfg == CFG_WIDE ? 1 : (p.cfg == CFG_BIG32 ? 2 : (p.cfg == CFG_SMALL ? 4 : 3));
  const int slots = 256 * occ;
  int cap = p.nchunks / 4;
  if (cap < 1) cap = 1;
  if (cap > 1024) cap = 1024;
  const int overhead = p.cfg == CFG_SMALL ? 1 : 3;  // prologue + epilogue in units of one chunk
  long best_cost = -1;
  p.nsplit = 1;
  for (int s = 1; s <= cap; ++s) {
    const long wgs = (long)p.npairs * s;
    const long rounds = (wgs + slots - 1) / slots;
    const long cps = (p.nchunks + s - 1) / s;
    const long cost = rounds * (cps + overhead);
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      p.nsplit = s;
    }
    if (wgs > 4L * slots) break;
  }
  p.chunks_per_split = (p.nchunks + p.nsplit - 1) / p.nsplit;
  p.nsplit = (p.nchunks + p.chunks_per_split - 1) / p.chunks_per_split;
  p.nslabs = p.nsplit;
  // reduce kernel: whole 64x64 sub-tiles per workgroup when there are many tiles, 4-row slices otherwise
  const int subs = p.T / 64;
  p.rpw = (p.npairs * subs * subs >= 512) ? 64 : 4;
  p.ws_bytes = (size_t)p.nslabs * p.npairs * p.T * p.T * sizeof(float);
}

static int cfg_tile(int cfg) { return (cfg == CFG_SMALL || cfg == CFG_SMALL16) ? 64 : (cfg == CFG_WIDE ? 192 : 128); }
static int cfg_bk(int cfg) { return cfg == CFG_SMALL ? 64 : (cfg == CFG_BIG24 ? 24 : (cfg == CFG_BIG32 ? 32 : 16)); }
// dynamic LDS of one workgroup: [2 buffers][panels][BK][T + 4] floats
static size_t cfg_lds_bytes(int cfg, bool two_panels, bool ntb = false) {
  if (ntb) {  // [2 buffers][panels][3 pieces][T][BKP] bf16 (NtbCfg)
    const int bk = cfg_bk(cfg), bkp = bk == 16 ? 16 : bk + 8;
    return (size_t)2 * (two_panels ? 2 : 1) * 3 * cfg_tile(cfg) * bkp * 2;
  }
  return (size_t)2 * (two_panels ? 2 : 1) * cfg_bk(cfg) * (cfg_tile(cfg) + 4) * sizeof(float);
}
// The 128-tile's chunk depth: 16 rows per barrier (3 workgroups per CU); the deeper variants (24 / 32) measured no better
// (tools/microbench.py).
static int big_cfg(int64_t) { return CFG_BIG; }

static GramPlan make_plan(int64_t n, int64_t K, int64_t L_nt = 0) {
  GramPlan p;
  // WIDE pays where the 128-tile wastes an edge tile (n = 576 = 4.5 x 128 = 3 x 192); at n >= 1152 its single
  // resident workgroup per CU (348 registers) loses to BIG's three (measured: profiles/r01_microbench_gram_*)
  p.cfg = n <= 64 ? CFG_SMALL : ((n % 192 == 0 && n >= 576 && n <= 768) ? CFG_WIDE : big_cfg(L_nt));
  p.T = cfg_tile(p.cfg);
  p.BK = cfg_bk(p.cfg);
  p.nbt = (int)((n + p.T - 1) / p.T);
  p.npairs = p.nbt * (p.nbt + 1) / 2;
  p.nchunks = (int)((K + p.BK - 1) / p.BK);
  if (p.nchunks < 1) p.nchunks = 1;
  finish_plan(p);
  return p;
}

// rectangular output [nA][nB] (MODE_XCORR): every (row tile, column tile) pair; p.nbt = column tiles
static GramPlan make_plan_rect(int64_t nA, int64_t nB, int64_t K) {
  GramPlan p;
  p.cfg = nA <= 64 ? CFG_SMALL : big_cfg(0);
  p.T = cfg_tile(p.cfg);
  p.BK = cfg_bk(p.cfg);
  p.nbt = (int)((nB + p.T - 1) / p.T);
  p.npairs = (int)((nA + p.T - 1) / p.T) * p.nbt;
  p.nchunks = (int)((K + p.BK - 1) / p.BK);
  if (p.nchunks < 1) p.nchunks = 1;
  finish_plan(p);
  return p;
}

static int reduce_slabs(const GramPlan& p, float* slabs, int nslabs, float alpha, float* C, int n, unsigned flags,
                        hipStream_t stream);

// Kernels whose dynamic LDS exceeds the 64 KB default need the limit raised once per function.
static bool allow_big_lds(const void* fn, size_t bytes) {
  if (bytes <= 64 * 1024) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    set_error("gram: cannot raise the dynamic LDS limit to %zu bytes", bytes);
    return false;
  }
  return true;
}

template <int MODE>
static int launch_gram(const GramGeom& g, bool vec4, float alpha, float* C, unsigned flags, void* ws,
                       size_t ws_bytes, hipStream_t stream) {
  if (g.n <= 0) return LK_OK;
  // NT: g.K counts padded positions (Lp per image); the plan wants the chunk count of exactly that
  const GramPlan p = make_plan(g.n, g.K, is_nt(MODE) ? g.L : 0);
  if (ws == nullptr || ws_bytes < p.ws_bytes) {
    set_error("gram: workspace too small (%zu < %zu bytes)", ws_bytes, p.ws_bytes);
    return LK_EWORKSPACE;
  }
  float* slabs = static_cast<float*>(ws);
  dim3 grid(p.npairs, p.nsplit), block(256);
  // one split + upper-only accumulation: the kernel adds into C itself (no slab round trip)
  const bool persist = (flags & LK_GRAM_SLABS_PERSIST) != 0;
  float* Cdirect = (!persist && p.nsplit == 1 && (flags & LK_GRAM_UPPER_ONLY)) ? C : nullptr;
  GramGeom gp = g;
  gp.slab_accumulate = persist ? 1 : 0;
  const size_t lds = cfg_lds_bytes(p.cfg, !(p.cfg == CFG_SMALL), MODE == MODE_NTB);
#define LK_LAUNCH(V, S)                                                                                          \
  do {                                                                                                           \
    if (!allow_big_lds((const void*)gram_kernel<MODE, V, S>, lds)) return LK_ELAUNCH;                            \
    hipLaunchKernelGGL((gram_kernel<MODE, V, S>), grid, block, lds, stream, gp, slabs, p.nbt, p.npairs,          \
                       p.chunks_per_split, p.nchunks, Cdirect, alpha);                                           \
  } while (0)
#define LK_LAUNCH_V(S)                    \
  do {                                    \
    if (vec4) LK_LAUNCH(4, S);            \
    else LK_LAUNCH(1, S);                 \
  } while (0)
  switch (p.cfg) {
    case CFG_SMALL: LK_LAUNCH_V(CFG_SMALL); break;
    case CFG_BIG: LK_LAUNCH_V(CFG_BIG); break;
    default: LK_LAUNCH_V(CFG_WIDE); break;
  }
#undef LK_LAUNCH_V
#undef LK_LAUNCH
  int rc = check_launch("gram_kernel");
  if (rc || Cdirect != nullptr || persist) return rc;  // persistent slabs are reduced once: lk_gram_slabs_reduce_f32
  return reduce_slabs(p, slabs, p.nslabs, alpha, C, g.n, flags, stream);
}

// sum `nslabs` slabs (two-level beyond 32), scale, accumulate into C, mirror off-diagonal tiles
static int reduce_slabs(const GramPlan& p, float* slabs, int nslabs, float alpha, float* C, int n, unsigned flags,
                        hipStream_t stream) {
  int stride = 1;
  const int64_t slab_elems = (int64_t)p.npairs * p.T * p.T;
  if (nslabs > 32) {  // two-level reduction keeps every thread's serial chain short
    const int SG = 32;
    const int groups = (nslabs + SG - 1) / SG;
    int64_t bx = (slab_elems / 4 + 255) / 256;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(gram_prereduce_kernel, dim3((unsigned)bx, groups), dim3(256), 0, stream, slabs, nslabs, slab_elems,
                       SG);
    nslabs = groups;
    stride = SG;
  }
  const int subs = p.T / 64;
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(p.npairs, subs * subs * (64 / p.rpw)), dim3(256), 0, stream, slabs, nslabs,
                     stride, p.npairs, p.T, p.nbt, alpha, C, n, (flags & LK_GRAM_UPPER_ONLY) ? 0 : 1, p.rpw, 0, 0);
  return check_launch("gram_reduce_kernel");
}

// ---- shift-correlation form of the 3x3 / stride 1 / padding 1 conv A factor ------------------------------
// For q = p + d (d, e in {-1,0,1}^2 the two patch offsets, D = e - d):
//   A[(d,ci),(e,cj)] = sum_{q in grid, q - d in grid} x[ci,q] x~[cj,q+D]
//                    = R[D] - Row_{ry(d)}[D] - Col_{cx(d)}[D] + Pix_{ry,cx}[D]
// with R[D] the correlation over the WHOLE grid (only 13 of the 25 shifts are computed, R[-D] = R[D]^T) and the
// corrections correlations over one boundary row / column / corner pixel.  The 81 (d,e) blocks collapse onto 25
// shifts: 13 C^2 L multiply-adds instead of 40.5 C^2 L for the symmetric half of the im2col Gram.
struct Region {
  int h0, w0, h, w;
};
// One launch for up to MAX_REG regions; outputs R[r] = [Cin][nshift*Cin], consecutive in memory.
static int launch_xcorr(const float* x, int64_t B, int H, int W, int Cin, const Region* regs, int nreg,
                        const signed char* sdy, const signed char* sdx, int nshift, float* R, void* ws,
                        size_t ws_bytes, hipStream_t stream) {
  if (nreg < 1 || nreg > MAX_REG) {
    set_error("xcorr: bad region count %d", nreg);
    return LK_EINVAL;
  }
  GramGeom g{};
  g.x = x; g.H = H; g.W = W; g.Cin = Cin;
  g.nA = Cin; g.nB = nshift * Cin; g.n = g.nB;
  g.nreg = nreg;
  int64_t Kmax = 0;
  for (int r = 0; r < nreg; ++r) {
    g.reg_h0[r] = regs[r].h0; g.reg_w0[r] = regs[r].w0;
    g.reg_K[r] = B * regs[r].h * regs[r].w;
    g.div_reg[r] = make_fastdiv(regs[r].h * regs[r].w);
    g.div_regw[r] = make_fastdiv(regs[r].w);
    if (g.reg_K[r] > Kmax) Kmax = g.reg_K[r];
  }
  g.K = Kmax;
  for (int i = 0; i < 25; ++i) {
    g.sdy[i] = i < nshift ? sdy[i] : 0;
    g.sdx[i] = i < nshift ? sdx[i] : 0;
  }
  GramPlan p = make_plan_rect(g.nA, g.nB, Kmax);
  if (ws == nullptr || ws_bytes < p.ws_bytes * nreg) {
    set_error("xcorr: workspace too small (%zu < %zu bytes)", ws_bytes, p.ws_bytes * nreg);
    return LK_EWORKSPACE;
  }
  float* slabs = static_cast<float*>(ws);
  const bool vec4 = (Cin % 4 == 0) && aligned16(x);
  dim3 grid(p.npairs, p.nsplit, nreg), block(256);
  const size_t lds = cfg_lds_bytes(p.cfg, true);
#define LK_LAUNCH(V, S)                                                                                            \
  do {                                                                                                             \
    if (!allow_big_lds((const void*)gram_kernel<MODE_XCORR, V, S>, lds)) return LK_ELAUNCH;                        \
    hipLaunchKernelGGL((gram_kernel<MODE_XCORR, V, S>), grid, block, lds, stream, g, slabs, p.nbt, p.npairs,       \
                       p.chunks_per_split, p.nchunks, (float*)nullptr, 1.f);                                       \
  } while (0)
#define LK_LAUNCH_V(S)                    \
  do {                                    \
    if (vec4) LK_LAUNCH(4, S);            \
    else LK_LAUNCH(1, S);                 \
  } while (0)
  switch (p.cfg) {
    case CFG_SMALL: LK_LAUNCH_V(CFG_SMALL); break;
    default: LK_LAUNCH_V(CFG_BIG); break;
  }
#undef LK_LAUNCH_V
#undef LK_LAUNCH
  int rc = check_launch("gram_kernel<XCORR>");
  if (rc) return rc;
  int nslabs = p.nslabs, stride = 1;
  const int64_t slab_elems = (int64_t)p.npairs * p.T * p.T;
  if (nslabs > 32) {
    const int SG = 32;
    const int groups = (nslabs + SG - 1) / SG;
    int64_t bx = (slab_elems / 4 + 255) / 256;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(gram_prereduce_kernel, dim3((unsigned)bx, groups, nreg), dim3(256), 0, stream, slabs, nslabs,
                       slab_elems, SG);
    nslabs = groups;
    stride = SG;
  }
  const int subs = p.T / 64;
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(p.npairs, subs * subs * (64 / p.rpw), nreg), dim3(256), 0, stream, slabs,
                     nslabs, stride, p.npairs, p.T, p.nbt, 1.f, R, g.nB, 0, p.rpw, g.nA, p.nslabs);
  return check_launch("gram_reduce_kernel<rect>");
}

// half-plane of shifts for the full-grid correlation: (0,0),(0,1),(0,2),(1,-2..2),(2,-2..2)
__constant__ signed char kHalfIndex[25] = {  // index by (Dy+2)*5 + (Dx+2); -1: use the transpose of -D
    -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12};

// A[(d,ci),(e,cj)] += alpha * (R[D] - Row - Col + Pix), native (kh,kw,ci) order, n = 9*Cin
__global__ __launch_bounds__(256) void shiftcorr_assemble_kernel(const float* __restrict__ Rf,
                                                                 const float* __restrict__ strips,
                                                                 const float* __restrict__ pix, int Cin, float alpha,
                                                                 float* __restrict__ A) {
  const int n = 9 * Cin;
  const int64_t total = (int64_t)n * n;
  const int64_t blk = (int64_t)Cin * 25 * Cin;  // one strip / pixel correlation [Cin][25*Cin]
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int r = (int)(idx / n), c = (int)(idx - (int64_t)r * n);
    const int d = r / Cin, ci = r - d * Cin;
    const int e = c / Cin, cj = c - e * Cin;
    const int dy = d / 3 - 1, dx = d % 3 - 1;
    const int Dy = (e / 3 - 1) - dy, Dx = (e % 3 - 1) - dx;
    const int t = (Dy + 2) * 5 + (Dx + 2);
    const int h = kHalfIndex[t];
    float v = (h >= 0) ? Rf[(int64_t)ci * (13 * Cin) + h * Cin + cj]
                       : Rf[(int64_t)cj * (13 * Cin) + kHalfIndex[24 - t] * Cin + ci];
    // strips: 0 = top row (dy=+1), 1 = bottom row (dy=-1), 2 = left column (dx=+1), 3 = right column (dx=-1)
    const int64_t off = (int64_t)ci * (25 * Cin) + t * Cin + cj;
    const int sr = dy == 1 ? 0 : (dy == -1 ? 1 : -1);
    const int sc = dx == 1 ? 2 : (dx == -1 ? 3 : -1);
    if (sr >= 0) v -= strips[sr * blk + off];
    if (sc >= 0) v -= strips[sc * blk + off];
    if (sr >= 0 && sc >= 0) v += pix[(sr * 2 + (sc - 2)) * blk + off];
    A[idx] += alpha * v;
  }
}

// ---- pixel-pair form of the 3x3 / stride 1 / padding 1 conv A factor on SMALL maps ------------------------------
// With x_b flattened to one row [H*W*Cin] (NHWC), the pixel-pair Gram  Cp = sum_b x_b^T x_b  is LINEAR in the data, so
// it can be accumulated over all minibatches of a fit by the plain TN Gram kernel (K = batch rows only) and the patch
// Gram assembled from it once:
//   A[(d,ci),(e,cj)] = sum_{p : p+d, p+e in the grid} Cp[(p+d, ci), (p+e, cj)]          (d, e in {-1,0,1}^2)
// For a 4x4 map that is 2 * B * (16 C)^2 / 2 flop per minibatch instead of B * 16 * (9 C)^2: 5x fewer, and the
// 81-block assembly runs once per fit.  (At 8x8 the full pixel-pair Gram is as expensive as the patch Gram; only the
// shift-window pairs would be needed there.)
__global__ __launch_bounds__(256) void pixgram_assemble_kernel(const float* __restrict__ Cp, int H, int W, int Cin,
                                                               float alpha, float* __restrict__ A) {
  const int n = 9 * Cin;
  const int64_t np = (int64_t)H * W * Cin;
  const int64_t total = (int64_t)n * n;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int r = (int)(idx / n), c = (int)(idx - (int64_t)r * n);
    const int d = r / Cin, ci = r - d * Cin;
    const int e = c / Cin, cj = c - e * Cin;
    const int dy = d / 3 - 1, dx = d % 3 - 1, ey = e / 3 - 1, ex = e % 3 - 1;
    float v = 0.f;
    for (int py = 0; py < H; ++py) {
      const int ay = py + dy, by = py + ey;
      if (ay < 0 || ay >= H || by < 0 || by >= H) continue;
      for (int px = 0; px < W; ++px) {
        const int ax = px + dx, bx = px + ex;
        if (ax < 0 || ax >= W || bx < 0 || bx >= W) continue;
        v += Cp[((int64_t)(ay * W + ax) * Cin + ci) * np + (int64_t)(by * W + bx) * Cin + cj];
      }
    }
    A[idx] += alpha * v;
  }
}

// ---- banded pixel-pair form of the 3x3 / stride 1 / padding 1 conv A factor (any map size, Cin % 64 == 0) ----------
// The same linearity as above, restricted to the pixel pairs the 3x3 window can see: for every pixel q and every
// shift D of the half plane {(0,0),(0,1),(0,2),(1,-2..2),(2,-2..2)} with q+D inside the map, the Cin x Cin block
//   Blk[q, D] += sum_b x[b, q, :]^T x[b, q+D, :]
// is accumulated over the whole fit (MODE_TNP: the TN loader on the flattened NHWC images, one workgroup per
// T x T tile of a block, K = batch rows, read-modify-write straight from the accumulators), and
//   A[(d,ci),(e,cj)] = sum_{p : p+d, p+e in the map} Blk[p+d, e-d][ci, cj]      (transposed block for e-d outside
// the half plane) is assembled once per fit.  13 C^2 L multiply-adds per sample like the shift-correlation form, but no
// boundary strips, no split-K slabs and no per-minibatch assembly.
static const signed char kHalfDy[13] = {0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2};
static const signed char kHalfDx[13] = {0, 1, 2, -2, -1, 0, 1, 2, -2, -1, 0, 1, 2};

// One thread per (shift h, ci, cj) and pixel range: every block element is read exactly once; up to nine (d, e)
// patch-offset pairs share the shift D = e - d and differ only in which pixels q = p + d they may use (p itself must be in
// the map).  `blocks2` (optional): a second set of accumulators of the same geometry (the other lane of a fit with two
// minibatches in flight) — the element-wise sum of the two is what is assembled, so the lanes' blocks never need a pass of
// their own.
// Parallelism: a 64-channel layer has only 13 * 64 * 64 = 53 k (h, ci, cj) triples and 1024 pixels to sum over each —
// round 3's form (one thread per triple, one load in flight) ran at 0.2 TB/s, 0.9-1.2 ms per 32 x 32 layer.  Now the P
// waves of a workgroup share 64 triples and split the map's pixels into P contiguous ranges (fixed-order sum of the P
// partials through LDS: deterministic), and every thread requests eight block elements ahead of its chain of additions;
// the slot table and all conditions are uniform over a wave when Cin % 64 == 0.
template <bool MIRROR>
__global__ __launch_bounds__(256) void pixpair_assemble_kernel(const float* __restrict__ blocks,
                                                               const float* __restrict__ blocks2,
                                                               const int* __restrict__ slots, int H, int W, int Cin,
                                                               float alpha, float* __restrict__ A) {
  // a thread owns FOUR consecutive elements of one shift's Cin x Cin block (16-byte loads: the one-element form issued four
  // times the requests for the same bytes and ran at 1.6 TB/s); a wave one pixel range of the map
  __shared__ float4 red[4 * 9 * 64];
  const int n = 9 * Cin;
  const int64_t bsz = (int64_t)Cin * Cin;
  const int64_t total4 = 13 * bsz / 4;
  const int HW = H * W;
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6, P = blockDim.x >> 6;
  // this wave's pixels [q_lo, q_hi)
  const int per = (HW + P - 1) / P;
  const int q_lo = part * per < HW ? part * per : HW, q_hi = q_lo + per < HW ? q_lo + per : HW;
  constexpr int U = 8;
  for (int64_t base = (int64_t)blockIdx.x * 64; base < total4; base += (int64_t)gridDim.x * 64) {  // (uniform over the workgroup)
    const int64_t idx = base + lane;
    const bool live = idx < total4;
    const int64_t el = (live ? idx : base) * 4;
    const int h = (int)(el / bsz);
    const int64_t inner = el - (int64_t)h * bsz;
    const int ci = (int)(inner / Cin), cj = (int)(inner - (int64_t)ci * Cin);  // cj % 4 == 0 (Cin % 4 == 0)
    const int Dy = h < 3 ? 0 : (h < 8 ? 1 : 2);
    const int Dx = h < 3 ? h : (h < 8 ? h - 5 : h - 10);
    // patch offsets d = (dy, dx) with e = d + D still inside {-1,0,1}^2
    const int dy_lo = -1, dy_hi = 1 - Dy;
    const int dx_lo = Dx < 0 ? -1 - Dx : -1, dx_hi = Dx > 0 ? 1 - Dx : 1;
    float4 acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) acc[a][b] = make_float4(0.f, 0.f, 0.f, 0.f);
    int qy0 = q_lo / W, qx0 = q_lo - qy0 * W;  // pixel q0 (tracked incrementally: no division in the loop)
    for (int q0 = q_lo; q0 < q_hi; q0 += U) {
      float4 v[U];
      bool have[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int q = q0 + u;
        const int slot = q < q_hi ? slots[q * 13 + h] : -1;
        have[u] = slot >= 0;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (have[u]) {
          v[u] = *reinterpret_cast<const float4*>(blocks + (int64_t)slot * bsz + inner);
          if (blocks2) {
            const float4 w = *reinterpret_cast<const float4*>(blocks2 + (int64_t)slot * bsz + inner);
            v[u].x += w.x, v[u].y += w.y, v[u].z += w.z, v[u].w += w.w;
          }
        }
      }
      int qy = qy0, qx = qx0;
#pragma unroll
      for (int u = 0; u < U; ++u, ++qx) {
        if (qx == W) qx = 0, ++qy;
        if (!have[u]) continue;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            const int dy = a - 1, dx = b - 1;  // p = q - d has to be a pixel of the map
            if (dy >= dy_lo && dy <= dy_hi && dx >= dx_lo && dx <= dx_hi && qy - dy >= 0 && qy - dy < H && qx - dx >= 0 &&
                qx - dx < W)
              acc[a][b].x += v[u].x, acc[a][b].y += v[u].y, acc[a][b].z += v[u].z, acc[a][b].w += v[u].w;
          }
      }
      if (qx == W) qx = 0, ++qy;
      qy0 = qy, qx0 = qx;
    }
    if (P > 1) {  // partials of the P pixel ranges, summed in range order by the first wave
#pragma unroll
      for (int k = 0; k < 9; ++k) red[(part * 9 + k) * 64 + lane] = acc[k / 3][k % 3];
      __syncthreads();
      if (part == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          float4 t = red[k * 64 + lane];
          for (int p_ = 1; p_ < P; ++p_) {
            const float4 w = red[(p_ * 9 + k) * 64 + lane];
            t.x += w.x, t.y += w.y, t.z += w.z, t.w += w.w;
          }
          acc[k / 3][k % 3] = t;
        }
      }
      __syncthreads();
    }
    if (part != 0 || !live) continue;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int dy = a - 1, dx = b - 1;
        if (!(dy >= dy_lo && dy <= dy_hi && dx >= dx_lo && dx <= dx_hi)) continue;
        const int d = a * 3 + b, e = (dy + Dy + 1) * 3 + (dx + Dx + 1);
        const float4 t = acc[a][b];
        const float v4[4] = {alpha * t.x, alpha * t.y, alpha * t.z, alpha * t.w};
        float4* dst = reinterpret_cast<float4*>(A + (int64_t)(d * Cin + ci) * n + e * Cin + cj);
        float4 o = *dst;
        o.x += v4[0], o.y += v4[1], o.z += v4[2], o.w += v4[3];
        *dst = o;
        if (MIRROR && h != 0) {  // the mirrored block (shift -D): a column of four rows, one 4-byte access per row — left out
                                 // when the consumer reads the upper triangle only (every (d, e) block written above has e >= d)
#pragma unroll
          for (int j = 0; j < 4; ++j) A[(int64_t)(e * Cin + cj + j) * n + d * Cin + ci] += v4[j];
        }
      }
  }
}

struct ShiftCorrPlan {
  size_t off_Rf, off_strips, off_pix, off_ws, ws_each, total;
};
static ShiftCorrPlan shiftcorr_plan(int64_t B, int64_t H, int64_t W, int64_t Cin) {
  ShiftCorrPlan p;
  size_t off = 0;
  p.off_Rf = off; off += align_up((size_t)Cin * 13 * Cin * 4, 256);
  p.off_strips = off;                                  // 4 strips followed directly by the 4 corner pixels
  p.off_pix = off + (size_t)4 * Cin * 25 * Cin * 4;
  off += align_up((size_t)8 * Cin * 25 * Cin * 4, 256);
  size_t w = make_plan_rect(Cin, 13 * Cin, B * H * W).ws_bytes;
  const int64_t strip_len = H > W ? H : W;
  size_t w2 = make_plan_rect(Cin, 25 * Cin, B * strip_len).ws_bytes * 8;  // 4 strips + 4 pixels in one launch
  p.ws_each = w > w2 ? w : w2;
  p.off_ws = off; off += align_up(p.ws_each, 256);
  p.total = off;
  return p;
}

// ---- layout helpers ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, int C, int HW,
                                                           float* __restrict__ dst) {
  // one workgroup transposes a 64(c) x 64(hw) tile of image blockIdx.z through LDS
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* s = src + (int64_t)b * C * HW;
  float* d = dst + (int64_t)b * C * HW;
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, p = p0 + tx;
    tile[i][tx] = (c < C && p < HW) ? s[(int64_t)c * HW + p] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int p = p0 + i, c = c0 + tx;
    if (c < C && p < HW) d[(int64_t)p * C + c] = tile[tx][i];
  }
}

__global__ __launch_bounds__(256) void symmetrize_kernel(float* __restrict__ Cm, int n) {
  // copy upper triangle (r < c) to the lower one, 64x64 tiles through LDS; grid over tile pairs bi<=bj
  __shared__ float tile[64][65];
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bi > bj) return;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = bi * 64 + i, c = bj * 64 + tx;
    tile[i][tx] = (r < n && c < n) ? Cm[(int64_t)r * n + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int r = bj * 64 + i, c = bi * 64 + tx;  // destination (lower) element, source = tile[tx][i]
    if (r < n && c < n && r > c) Cm[(int64_t)r * n + c] = tile[tx][i];
  }
}

__global__ __launch_bounds__(256) void permute_sym_kernel(const float* __restrict__ src, int Cin, int KK,
                                                          float* __restrict__ dst, int accumulate) {
  const int n = Cin * KK;
  const int64_t total = (int64_t)n * n;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int r = (int)(idx / n), c = (int)(idx - (int64_t)r * n);
    const int ci = r / KK, d = r - ci * KK;
    const int cj = c / KK, e = c - cj * KK;
    const float v = src[(int64_t)(d * Cin + ci) * n + (e * Cin + cj)];
    dst[idx] = accumulate ? dst[idx] + v : v;
  }
}

// ---- the once-per-fit layout pass of ALL factors in one launch ----------------------------------------------------------
// What `KronAccumulator.finalize` did with ~110 small launches (symmetrise every factor, permute every conv A factor from
// the kernels' (kh, kw, ci) column order to F.unfold's (ci, kh, kw), apply the deferred BatchNorm scales to the G factors):
// 1.7 ms of a 20-minibatch fit, all of it launch gaps.  Per factor (descriptor):
//   kk <= 1:  in place,  C[r][c] = C[c][r] = s[r] s[c] C[r][c]  for r <= c        (s optional)
//   kk  > 1:  dst[(ci,d)][(cj,e)] = src_upper[(d,ci)][(e,cj)]                     (src: upper triangle valid; dst != src)
struct FinDesc {
  const float* src;
  float* dst;
  const float* scale;
  int n, cin, kk, blk0;  // blk0: first workgroup of this factor
};
constexpr int FIN_MAX = 48;
struct FinBatch {
  int count, nblk;
  FinDesc d[FIN_MAX];
};
constexpr int FIN_PERM_ELEMS = 4096;  // elements of dst per workgroup (kk > 1)

__global__ __launch_bounds__(256) void finalize_factors_kernel(const FinBatch fb) {
  __shared__ float tile[64][65];
  int k = 0;
  for (int i = 1; i < fb.count; ++i)
    if ((int)blockIdx.x >= fb.d[i].blk0) k = i;  // (uniform; blk0 ascending)
  const FinDesc& D = fb.d[k];
  const int local = (int)blockIdx.x - D.blk0;
  const int n = D.n;
  if (D.kk <= 1) {
    // tile pair (bi <= bj), pairs ordered by bj, then bi: local = bj (bj + 1) / 2 + bi
    int bj = (int)((sqrtf(8.f * (float)local + 1.f) - 1.f) * 0.5f);
    while ((bj + 1) * (bj + 2) / 2 <= local) ++bj;
    while (bj * (bj + 1) / 2 > local) --bj;
    const int bi = local - bj * (bj + 1) / 2;
    float* Cm = D.dst ? D.dst : const_cast<float*>(D.src);
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
      const int r = bi * 64 + i, c = bj * 64 + tx;
      float v = (r < n && c < n) ? D.src[(int64_t)r * n + c] : 0.f;
      if (D.scale && r < n && c < n) v *= D.scale[r] * D.scale[c];
      tile[i][tx] = v;
    }
    __syncthreads();
    if (D.scale || D.dst) {  // the upper part itself changes (or moves)
      for (int i = ty; i < 64; i += 4) {
        const int r = bi * 64 + i, c = bj * 64 + tx;
        if (r < n && c < n && r <= c) Cm[(int64_t)r * n + c] = tile[i][tx];
      }
    }
    for (int i = ty; i < 64; i += 4) {
      const int r = bj * 64 + i, c = bi * 64 + tx;  // destination (lower) element, source = tile[tx][i]
      if (r < n && c < n && r > c) Cm[(int64_t)r * n + c] = tile[tx][i];
    }
    return;
  }
  const int Cin = D.cin, KK = D.kk;
  const int64_t total = (int64_t)n * n;
  const int64_t e0 = (int64_t)local * FIN_PERM_ELEMS;
  for (int j = threadIdx.x; j < FIN_PERM_ELEMS; j += 256) {
    const int64_t idx = e0 + j;
    if (idx >= total) break;
    const int r = (int)(idx / n), c = (int)(idx - (int64_t)r * n);
    const int ci = r / KK, d = r - ci * KK;
    const int cj = c / KK, e = c - cj * KK;
    int rs = d * Cin + ci, cs = e * Cin + cj;
    if (rs > cs) {
      const int t = rs;
      rs = cs, cs = t;
    }
    D.dst[idx] = D.src[(int64_t)rs * n + cs];
  }
}

}  // namespace lk

using namespace lk;

#ifdef LK_GRAM_TRACE
extern "C" int lk_gram_trace_read(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gram_trace), 5 * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" size_t lk_gram_workspace_bytes(int64_t n, int64_t K) {
  if (n <= 0) return 0;
  return make_plan(n, K < 1 ? 1 : K).ws_bytes;
}

extern "C" size_t lk_gram_nt_workspace_bytes(int64_t nb_total, int64_t n, int64_t L) {
  if (n <= 0 || L <= 0) return 0;
  const int BK = cfg_bk(make_plan(n, 1, L).cfg);
  const int64_t Lp = (L + BK - 1) / BK * BK;
  return make_plan(n, nb_total < 1 ? Lp : nb_total * Lp, L).ws_bytes;
}

extern "C" int lk_gram_slabs_reduce_f32(float* slabs, size_t slabs_bytes, int64_t n, int64_t L_nt, float alpha, float* C,
                                        unsigned flags, void* stream) {
  LK_REQUIRE(slabs && C && n > 0 && n < (1 << 30), "lk_gram_slabs_reduce_f32: bad arguments");
  const GramPlan p = make_plan(n, 1, L_nt);  // tile geometry depends on n (and the NT chunk depth) only
  const size_t slab_bytes = (size_t)p.npairs * p.T * p.T * sizeof(float);
  const int nslabs = (int)(slabs_bytes / slab_bytes);
  LK_REQUIRE(nslabs >= 1, "lk_gram_slabs_reduce_f32: buffer smaller than one slab");
  return reduce_slabs(p, slabs, nslabs, alpha, C, (int)n, flags, (hipStream_t)stream);
}

extern "C" int lk_gram_tn_f32(const float* X, int64_t K, int64_t n, int64_t ldx, float alpha, float* C,
                              unsigned flags, void* ws, size_t ws_bytes, void* stream) {
  LK_REQUIRE(X && C && K >= 0 && n >= 0 && ldx >= n, "lk_gram_tn_f32: bad arguments");
  LK_REQUIRE(n < (1 << 30), "lk_gram_tn_f32: n too large");
  GramGeom g{};
  g.x = X; g.K = K; g.n = (int)n; g.ldx = ldx;
  const bool vec4 = (n % 4 == 0) && (ldx % 4 == 0) && aligned16(X);
  return launch_gram<MODE_TN>(g, vec4, alpha, C, flags, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int lk_gram_nt_seg_f32(const float* const* segs, int64_t nseg, int64_t nb, int64_t n, int64_t L,
                                  float alpha, float* C, unsigned flags, void* ws, size_t ws_bytes, void* stream) {
  LK_REQUIRE(segs && C && nseg >= 1 && nseg <= MAX_SEG && nb >= 0 && n >= 0 && L >= 1,
             "lk_gram_nt_seg_f32: bad arguments (at most 16 segments)");
  LK_REQUIRE(n < (1 << 30) && L < (1 << 30), "lk_gram_nt_seg_f32: dims too large");
  const int BK = cfg_bk(make_plan(n, 1, L).cfg);
  GramGeom g{};
  g.n = (int)n; g.L = (int)L;
  g.Lp = (int)((L + BK - 1) / BK * BK);
  g.seg_nb = (int)(nb > 0 ? nb : 1);
  g.nseg = (int)nseg;
  g.K = nseg * nb * g.Lp;
  bool vec4 = (L % 4 == 0);
  for (int i = 0; i < nseg; ++i) {
    LK_REQUIRE(segs[i] != nullptr, "lk_gram_nt_seg_f32: null segment");
    g.seg[i] = segs[i];
    vec4 = vec4 && aligned16(segs[i]);
  }
  g.x = segs[0];
  // fp32 products from split-bf16 MFMAs (MODE_NTB) whenever the positions can be read four at a time
  if (vec4 && BK % 16 == 0) return launch_gram<MODE_NTB>(g, vec4, alpha, C, flags, ws, ws_bytes, (hipStream_t)stream);
  return launch_gram<MODE_NT>(g, vec4, alpha, C, flags, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int lk_gram_nt_f32(const float* X, int64_t nb, int64_t n, int64_t L, float alpha, float* C,
                              unsigned flags, void* ws, size_t ws_bytes, void* stream) {
  LK_REQUIRE(X != nullptr, "lk_gram_nt_f32: bad arguments");
  const float* segs[1] = {X};
  return lk_gram_nt_seg_f32(segs, 1, nb, n, L, alpha, C, flags, ws, ws_bytes, stream);
}

extern "C" int lk_gram_conv_nhwc_f32(const float* x, int64_t B, int64_t H, int64_t W, int64_t Cin, int kh, int kw,
                                     int sh, int sw, int ph, int pw, int dh, int dw, float alpha, float* C,
                                     unsigned flags, void* ws, size_t ws_bytes, void* stream) {
  LK_REQUIRE(x && C && B >= 0 && H > 0 && W > 0 && Cin > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0,
             "lk_gram_conv_nhwc_f32: bad arguments");
  const int64_t OH = (H + 2 * ph - dh * (kh - 1) - 1) / sh + 1;
  const int64_t OW = (W + 2 * pw - dw * (kw - 1) - 1) / sw + 1;
  LK_REQUIRE(OH > 0 && OW > 0, "lk_gram_conv_nhwc_f32: empty output");
  GramGeom g{};
  g.x = x; g.n = (int)(Cin * kh * kw); g.K = B * OH * OW;
  g.H = (int)H; g.W = (int)W; g.Cin = (int)Cin; g.OH = (int)OH; g.OW = (int)OW; g.kw = kw;
  g.sh = sh; g.sw = sw; g.ph = ph; g.pw = pw; g.dh = dh; g.dw = dw;
  LK_REQUIRE(g.K < (1ll << 31) - 64, "lk_gram_conv_nhwc_f32: B*OH*OW must be < 2^31");
  g.div_ohw = make_fastdiv((int)(OH * OW));
  g.div_ow = make_fastdiv((int)OW);
  const bool vec4 = (Cin % 4 == 0) && aligned16(x);
  return launch_gram<MODE_CONV>(g, vec4, alpha, C, flags, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int lk_nchw_to_nhwc_f32(const float* src, int64_t B, int64_t C, int64_t HW, float* dst, void* stream) {
  LK_REQUIRE(src && dst && B >= 0 && C > 0 && HW > 0, "lk_nchw_to_nhwc_f32: bad arguments");
  if (B == 0) return LK_OK;
  LK_REQUIRE(B < 65536, "lk_nchw_to_nhwc_f32: batch too large for grid.z");
  dim3 grid((unsigned)((HW + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)B);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, (int)C, (int)HW, dst);
  return check_launch("nchw_to_nhwc_kernel");
}

extern "C" int lk_symmetrize_f32(float* C, int64_t n, void* stream) {
  LK_REQUIRE(C && n >= 0, "lk_symmetrize_f32: bad arguments");
  if (n == 0) return LK_OK;
  const unsigned nb = (unsigned)((n + 63) / 64);
  hipLaunchKernelGGL(symmetrize_kernel, dim3(nb, nb), dim3(256), 0, (hipStream_t)stream, C, (int)n);
  return check_launch("symmetrize_kernel");
}

extern "C" int lk_permute_sym_f32(const float* src, int64_t Cin, int64_t KK, float* dst, int accumulate,
                                  void* stream) {
  LK_REQUIRE(src && dst && Cin > 0 && KK > 0 && src != dst, "lk_permute_sym_f32: bad arguments");
  const int64_t n = Cin * KK;
  const int64_t total = n * n;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(permute_sym_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (int)Cin,
                     (int)KK, dst, accumulate);
  return check_launch("permute_sym_kernel");
}

extern "C" int lk_finalize_factors_f32(int64_t count, const float* const* src, float* const* dst, const float* const* scale,
                                       const int64_t* n, const int64_t* cin, const int64_t* kk, void* stream) {
  LK_REQUIRE(count >= 0 && (count == 0 || (src && dst && scale && n && cin && kk)), "lk_finalize_factors_f32: bad arguments");
  int64_t i = 0;
  while (i < count) {
    FinBatch fb;
    fb.count = 0, fb.nblk = 0;
    for (; i < count && fb.count < FIN_MAX; ++i) {
      LK_REQUIRE(src[i] && n[i] >= 0 && n[i] < (1ll << 24), "lk_finalize_factors_f32: bad factor");
      if (n[i] == 0) continue;
      FinDesc& D = fb.d[fb.count];
      D.src = src[i], D.dst = dst[i], D.scale = scale[i];
      D.n = (int)n[i], D.cin = (int)cin[i], D.kk = (int)kk[i], D.blk0 = fb.nblk;
      int64_t blocks;
      if (kk[i] > 1) {
        LK_REQUIRE(dst[i] && dst[i] != src[i] && cin[i] * kk[i] == n[i] && !scale[i],
                   "lk_finalize_factors_f32: a permuted factor needs dst != src, n = cin * kk and no scale");
        blocks = (n[i] * n[i] + FIN_PERM_ELEMS - 1) / FIN_PERM_ELEMS;
      } else {
        const int64_t nb = (n[i] + 63) / 64;
        blocks = nb * (nb + 1) / 2;
      }
      if ((int64_t)fb.nblk + blocks >= (1ll << 30)) break;  // (next launch)
      fb.nblk += (int)blocks;
      ++fb.count;
    }
    if (fb.count == 0) continue;
    hipLaunchKernelGGL(finalize_factors_kernel, dim3((unsigned)fb.nblk), dim3(256), 0, (hipStream_t)stream, fb);
    if (int rc = check_launch("finalize_factors_kernel")) return rc;
  }
  return LK_OK;
}

extern "C" int lk_conv3x3_pixpair_plan(int64_t H, int64_t W, int64_t Cin, int64_t* tile, int64_t* n_tiles,
                                       int64_t* n_blocks) {
  LK_REQUIRE(H >= 1 && W >= 1 && Cin >= 64 && Cin % 64 == 0 && tile && n_tiles && n_blocks,
             "lk_conv3x3_pixpair_plan: needs Cin % 64 == 0");
  int64_t nb = 0;
  for (int64_t y = 0; y < H; ++y)
    for (int64_t x = 0; x < W; ++x)
      for (int h = 0; h < 13; ++h) {
        const int64_t y2 = y + kHalfDy[h], x2 = x + kHalfDx[h];
        nb += (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W);
      }
  const int64_t T = Cin % 128 == 0 ? 128 : 64, tpp = Cin / T;
  LK_REQUIRE(nb * Cin * Cin < (1ll << 31) && H * W * Cin < (1ll << 31), "lk_conv3x3_pixpair_plan: problem too large");
  *tile = T;
  *n_blocks = nb;
  *n_tiles = nb * tpp * tpp;
  return LK_OK;
}

extern "C" int lk_conv3x3_pixpair_tables(int64_t H, int64_t W, int64_t Cin, int32_t* tiles, int32_t* slots) {
  int64_t T, nt, nb;
  if (int rc = lk_conv3x3_pixpair_plan(H, W, Cin, &T, &nt, &nb)) return rc;
  LK_REQUIRE(tiles && slots, "lk_conv3x3_pixpair_tables: null table");
  const int64_t tpp = Cin / T;
  int64_t slot = 0, k = 0;
  for (int64_t y = 0; y < H; ++y)
    for (int64_t x = 0; x < W; ++x)
      for (int h = 0; h < 13; ++h) {
        const int64_t q = y * W + x, y2 = y + kHalfDy[h], x2 = x + kHalfDx[h];
        if (!(y2 >= 0 && y2 < H && x2 >= 0 && x2 < W)) {
          slots[q * 13 + h] = -1;
          continue;
        }
        const int64_t q2 = y2 * W + x2;
        slots[q * 13 + h] = (int32_t)slot;
        for (int64_t ta = 0; ta < tpp; ++ta)
          for (int64_t tb = 0; tb < tpp; ++tb) {
            tiles[3 * k] = (int32_t)(q * Cin + ta * T);
            tiles[3 * k + 1] = (int32_t)(q2 * Cin + tb * T);
            tiles[3 * k + 2] = (int32_t)(slot * Cin * Cin + ta * T * Cin + tb * T);
            ++k;
          }
        ++slot;
      }
  return LK_OK;
}

extern "C" int lk_conv3x3_pixpair_accumulate_f32(const float* x, int64_t B, int64_t H, int64_t W, int64_t Cin, float alpha,
                                                 float* blocks, const int32_t* tiles_dev, int64_t n_tiles, void* stream) {
  LK_REQUIRE(x && blocks && tiles_dev && B >= 0 && n_tiles >= 0, "lk_conv3x3_pixpair_accumulate_f32: bad arguments");
  int64_t T, nt, nb;
  if (int rc = lk_conv3x3_pixpair_plan(H, W, Cin, &T, &nt, &nb)) return rc;
  LK_REQUIRE(nt == n_tiles, "lk_conv3x3_pixpair_accumulate_f32: table does not match the geometry");
  LK_REQUIRE(aligned16(x) && n_tiles < (1ll << 31), "lk_conv3x3_pixpair_accumulate_f32: unaligned input / too many tiles");
  if (B == 0 || n_tiles == 0) return LK_OK;
  GramGeom g{};
  g.x = x; g.K = B; g.n = (int)(H * W * Cin); g.ldx = H * W * Cin;
  g.tiles = tiles_dev; g.ldc = (int)Cin;
  const int cfg = T == 64 ? CFG_SMALL16 : CFG_BIG;
  const int nchunks = (int)((B + cfg_bk(cfg) - 1) / cfg_bk(cfg));
  const size_t lds = cfg_lds_bytes(cfg, true);
  dim3 grid((unsigned)n_tiles, 1), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (cfg == CFG_SMALL16) {
    hipLaunchKernelGGL((gram_kernel<MODE_TNP, 4, CFG_SMALL16>), grid, block, lds, st, g, (float*)nullptr, 1, (int)n_tiles,
                       nchunks, nchunks, blocks, alpha);
  } else {
    if (!allow_big_lds((const void*)gram_kernel<MODE_TNP, 4, CFG_BIG>, lds)) return LK_ELAUNCH;
    hipLaunchKernelGGL((gram_kernel<MODE_TNP, 4, CFG_BIG>), grid, block, lds, st, g, (float*)nullptr, 1, (int)n_tiles, nchunks,
                       nchunks, blocks, alpha);
  }
  return check_launch("gram_kernel<TNP>");
}

// blocks2 (optional): a second accumulator set of the same geometry; what is assembled is blocks + blocks2
extern "C" int lk_conv3x3_pixpair_assemble2_f32(const float* blocks, const float* blocks2, const int32_t* slots_dev, int64_t H,
                                                int64_t W, int64_t Cin, float alpha, float* A, int upper_only, void* stream) {
  LK_REQUIRE(blocks && slots_dev && A && H >= 1 && W >= 1 && Cin >= 1 && 9 * Cin < (1ll << 24) && H * W < (1ll << 24),
             "lk_conv3x3_pixpair_assemble_f32: bad arguments");
  LK_REQUIRE(Cin % 4 == 0 && ((reinterpret_cast<uintptr_t>(blocks) | reinterpret_cast<uintptr_t>(blocks2) | reinterpret_cast<uintptr_t>(A)) & 15) == 0,
             "lk_conv3x3_pixpair_assemble_f32: Cin % 4 == 0 and 16-byte aligned buffers");
  const int64_t total4 = 13 * Cin * Cin / 4;
  int64_t nblk = (total4 + 63) / 64;
  if (nblk > 65536) nblk = 65536;
  // waves per workgroup = pixel ranges: enough threads for the chip on the few-channel / large-map layers, at least 32
  // pixels per range
  int P = 1;
  while (P < 4 && nblk * 64 * P < (1 << 18) && H * W >= 64 * P) P *= 2;
  if (upper_only)
    hipLaunchKernelGGL(pixpair_assemble_kernel<false>, dim3((unsigned)nblk), dim3(64 * P), 0, (hipStream_t)stream, blocks, blocks2,
                       slots_dev, (int)H, (int)W, (int)Cin, alpha, A);
  else
    hipLaunchKernelGGL(pixpair_assemble_kernel<true>, dim3((unsigned)nblk), dim3(64 * P), 0, (hipStream_t)stream, blocks, blocks2,
                       slots_dev, (int)H, (int)W, (int)Cin, alpha, A);
  return check_launch("pixpair_assemble_kernel");
}

extern "C" int lk_conv3x3_pixpair_assemble_f32(const float* blocks, const int32_t* slots_dev, int64_t H, int64_t W,
                                               int64_t Cin, float alpha, float* A, void* stream) {
  return lk_conv3x3_pixpair_assemble2_f32(blocks, nullptr, slots_dev, H, W, Cin, alpha, A, 0, stream);
}

extern "C" int lk_conv3x3_pixgram_assemble_f32(const float* Cp, int64_t H, int64_t W, int64_t Cin, float alpha, float* A,
                                               void* stream) {
  LK_REQUIRE(Cp && A && H >= 1 && W >= 1 && Cin >= 1 && H * W * Cin < (1ll << 31) && 9 * Cin < (1ll << 24),
             "lk_conv3x3_pixgram_assemble_f32: bad arguments");
  const int64_t total = 81 * Cin * Cin;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(pixgram_assemble_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, Cp, (int)H, (int)W,
                     (int)Cin, alpha, A);
  return check_launch("pixgram_assemble_kernel");
}

extern "C" size_t lk_conv3x3_shiftcorr_workspace_bytes(int64_t B, int64_t H, int64_t W, int64_t Cin) {
  if (B < 0 || H < 1 || W < 1 || Cin < 1) return 0;
  return shiftcorr_plan(B < 1 ? 1 : B, H, W, Cin).total;
}

extern "C" int lk_conv3x3_shiftcorr_f32(const float* x, int64_t B, int64_t H, int64_t W, int64_t Cin, float alpha,
                                        float* C, void* ws, size_t ws_bytes, void* stream_) {
  LK_REQUIRE(x && C && B >= 0 && H >= 2 && W >= 2 && Cin >= 1, "lk_conv3x3_shiftcorr_f32: bad arguments");
  LK_REQUIRE(B * H * W < (1ll << 31) - 64 && 25 * Cin < (1 << 24), "lk_conv3x3_shiftcorr_f32: problem too large");
  if (B == 0) return LK_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const ShiftCorrPlan p = shiftcorr_plan(B, H, W, Cin);
  if (ws == nullptr || ws_bytes < p.total) {
    set_error("lk_conv3x3_shiftcorr_f32: workspace too small (%zu < %zu bytes)", ws_bytes, p.total);
    return LK_EWORKSPACE;
  }
  char* base = static_cast<char*>(ws);
  float* Rf = reinterpret_cast<float*>(base + p.off_Rf);
  float* strips = reinterpret_cast<float*>(base + p.off_strips);
  float* pix = reinterpret_cast<float*>(base + p.off_pix);
  void* gws = base + p.off_ws;
  if (hipMemsetAsync(base, 0, p.off_ws, stream) != hipSuccess) {
    set_error("lk_conv3x3_shiftcorr_f32: hipMemsetAsync failed");
    return LK_ELAUNCH;
  }
  signed char hy[13], hx[13], ay[25], ax[25];
  int k = 0;
  for (int dx = 0; dx <= 2; ++dx) { hy[k] = 0; hx[k] = (signed char)dx; ++k; }
  for (int dy = 1; dy <= 2; ++dy)
    for (int dx = -2; dx <= 2; ++dx) { hy[k] = (signed char)dy; hx[k] = (signed char)dx; ++k; }
  for (int t = 0; t < 25; ++t) { ay[t] = (signed char)(t / 5 - 2); ax[t] = (signed char)(t % 5 - 2); }
  const int h = (int)H, w = (int)W, c = (int)Cin;
  const int64_t blk = Cin * 25 * Cin;
  const Region full = {0, 0, h, w};
  int rc = launch_xcorr(x, B, h, w, c, &full, 1, hy, hx, 13, Rf, gws, p.ws_each, stream);
  if (rc) return rc;
  // the four boundary strips (top row, bottom row, left column, right column) and the four corner pixels
  // (index (row strip) * 2 + (column strip)) in ONE launch; `strips` and `pix` are adjacent in the workspace
  const Region regs[8] = {{0, 0, 1, w},     {h - 1, 0, 1, w},     {0, 0, h, 1},     {0, w - 1, h, 1},
                          {0, 0, 1, 1},     {0, w - 1, 1, 1},     {h - 1, 0, 1, 1}, {h - 1, w - 1, 1, 1}};
  if (pix != strips + 4 * blk) {
    set_error("lk_conv3x3_shiftcorr_f32: internal layout error");
    return LK_EINVAL;
  }
  rc = launch_xcorr(x, B, h, w, c, regs, 8, ay, ax, 25, strips, gws, p.ws_each, stream);
  if (rc) return rc;
  const int64_t total = 81 * Cin * Cin;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(shiftcorr_assemble_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, Rf, strips, pix, c, alpha,
                     C);
  return check_launch("shiftcorr_assemble_kernel");
}
