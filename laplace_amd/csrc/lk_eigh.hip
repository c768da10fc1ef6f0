// Symmetric eigendecomposition on gfx950: two-sided block-Jacobi.
//
// Replaces utils.symeig -> torch.linalg.eigh(M, UPLO="U") + clamp/nan_to_num
// (laplace/utils/utils.py:193-228) as called per Kronecker factor by Kron.decompose
// (laplace/utils/matrix.py:123-150).
//
// Algorithm (n padded to np = multiple of 64; 32-wide index blocks; nb = np/32 blocks):
//   sweep = nb-1 round-robin steps; in step s the nb blocks are paired into nb/2 disjoint pivots (I,J)
//     1. pivot kernel   : each pivot's 64x64 symmetric sub-matrix [A_II A_IJ; A_JI A_JJ] is diagonalised
//                         completely by a cyclic Jacobi in LDS -> orthogonal R_P (64x64) and its diagonal
//     2. update kernel  : every 64x64 tile pair (P<=Q):  A_PQ <- R_P^T A_PQ R_Q   (two exact-fp32 MFMA
//                         products; the mirrored tile is written from the same result, keeping A symmetric)
//     3. vupdate kernel : V[:, Q] <- V[:, Q] R_Q
//   a sweep in which no pivot performed a rotation sets the device-side `converged` flag; later launches
//   return immediately, so the whole solve is enqueued without a single host synchronisation.
//   Refinement: one Newton-Schulz step re-orthonormalises V, the eigenvalues are recomputed as Rayleigh
//   quotients v_i^T A v_i against the ORIGINAL matrix (three MFMA GEMMs), then rank-sorted ascending,
//   clamped, and V's columns gathered.
// Zero padding is exact: padded rows/columns never rotate (their off-diagonals are exactly 0).
//
// Many matrices (one decomposition = 43 factors for ResNet-18): every launch of the three kernels serves a ROUND --
// the current round-robin step of every matrix that is still iterating, each at its own position of its own sweep
// (`EigRound`: per slot the matrix' descriptor, its step and the first workgroup of its share of the grid).  One
// stream, one pivot + one update launch per round for all of them: measured on the c4 factors, six streams with one
// matrix each had on average 1.3-1.8 kernels in flight (profiles/r03_eig_timeline.md) -- the solves were serialised
// by the queue, not overlapped by it.
#include <chrono>
#include <cstdlib>
#include <thread>
#include <vector>

#include "lk_common.h"

namespace lk {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// x = h + l (+ <= 2^-22 |x|, 2^-25 absolute), both fp16.  The value is made opaque first: h and the residual must come
// from the SAME fp32 value (hipcc otherwise fuses the residual into fp16(x - h') with h' rounded from an exact product)
__device__ __forceinline__ void eig_split2(float x, _Float16& h, _Float16& l) {
  asm volatile("" : "+v"(x));
  h = (_Float16)x;
  l = (_Float16)(x - (float)h);
}

constexpr int EB = 32;       // index block
constexpr int EP = 64;       // pivot size (two blocks)
constexpr int ELD = 65;      // LDS pitch, conflict-free for column access

struct EigCtrl {             // lives in the workspace
  float scale;               // max(max |A_ii|, power-iteration estimate of lambda_max)
  int rotations;             // rotations performed in the current sweep
  int converged;             // set when a sweep performed none
  int sweeps;                // completed sweeps
  float bound;               // max_i sum_j |a_ij| >= lambda_max: fixes the power-of-two scale of the split-fp16 tile products
};

struct EigDesc {             // one matrix: where its buffers are (lives in its workspace, written once per solve)
  float* Aw;
  float* V;
  float* Rws;
  float* Dws;
  int* rotated;
  EigCtrl* ctrl;
  int np, nb;
};

constexpr int kEigBatch = 64;  // matrices served by one launch
struct EigRound {              // kernel argument, by value
  int njobs;
  int step[kEigBatch];         // round-robin step of slot j in ITS sweep
  int first[kEigBatch + 1];    // first workgroup of slot j (prefix sums over the slots)
  const EigDesc* desc[kEigBatch];
};

__device__ __forceinline__ int eig_slot(const EigRound& rd, int block) {  // block-uniform
  int j = 0;
  while (j + 1 < rd.njobs && block >= rd.first[j + 1]) ++j;
  return j;
}

__global__ void eig_desc_kernel(EigDesc d, EigDesc* out) { *out = d; }

// round-robin pairing of nb blocks (nb even): step s in [0, nb-1), pivot p in [0, nb/2)
__device__ __forceinline__ void pivot_blocks(int s, int p, int nb, int& I, int& J) {
  const int m = nb - 1;
  int a, b;
  if (p == 0) {
    a = m;
    b = s % m;
  } else {
    a = (s + p) % m;
    b = (s - p + m) % m;
  }
  I = a < b ? a : b;
  J = a < b ? b : a;
}

__device__ __forceinline__ int pivot_index(int local, int I, int J) {  // local 0..63 -> global row/col
  return (local < EB ? I * EB : J * EB - EB) + local;
}

__global__ __launch_bounds__(256) void eig_init_kernel(const float* __restrict__ A, int n, int np,
                                                       float* __restrict__ Aw, float* __restrict__ A0,
                                                       float* __restrict__ V, EigCtrl* ctrl) {
  const int64_t total = (int64_t)np * np;
  float mx = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int r = (int)(idx / np), c = (int)(idx - (int64_t)r * np);
    float v = 0.f;
    if (r < n && c < n) v = (r <= c) ? A[(int64_t)r * n + c] : A[(int64_t)c * n + r];  // UPLO="U"
    if (!(v == v) || fabsf(v) > 3.0e38f) v = 0.f;                                       // NaN / inf guard
    Aw[idx] = v;
    A0[idx] = v;
    V[idx] = (r == c) ? 1.f : 0.f;
    if (r == c) mx = fmaxf(mx, fabsf(v));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(reinterpret_cast<int*>(&ctrl->scale), __float_as_int(mx));
}

// ---- spectral scale: a few power iterations give lambda_max to within a few per cent -----------------
// (KFAC factors often have one dominant eigenvalue ~ n x the largest diagonal entry, so thresholds relative
// to max|a_ii| would sit far below fp32 noise and cost many useless sweeps)
__global__ __launch_bounds__(256) void eig_matvec_kernel(const float* __restrict__ A, int np,
                                                         const float* __restrict__ x, float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= np) return;
  float s = 0.f;
  for (int c = lane; c < np; c += 64) s += A[(int64_t)row * np + c] * x[c];
  s = wave_sum(s);
  if (lane == 0) y[row] = s;
}
// ctrl->bound = max_i sum_j |a_ij|  (one wave per row, fixed summation order; the max is order-independent)
__global__ __launch_bounds__(256) void eig_rowabs_kernel(const float* __restrict__ A, int np, EigCtrl* ctrl) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= np) return;
  float s = 0.f;
  for (int c = lane; c < np; c += 64) s += fabsf(A[(int64_t)row * np + c]);
  s = wave_sum(s);
  if (lane == 0 && s > 0.f && s < 3.0e38f) atomicMax(reinterpret_cast<int*>(&ctrl->bound), __float_as_int(s));
}
// x <- y / ||y||, ctrl->scale <- max(ctrl->scale, ||y||)   (one workgroup)
__global__ __launch_bounds__(256) void eig_normalize_kernel(const float* __restrict__ y, int np, float* __restrict__ x,
                                                            EigCtrl* ctrl, int init) {
  __shared__ float red[4];
  if (init) {  // deterministic pseudo-random start vector, never orthogonal to a non-negative dominant vector
    for (int i = threadIdx.x; i < np; i += 256) x[i] = 1.f + 0.25f * __sinf(0.7f * (float)i);
    return;
  }
  float s = 0.f;
  for (int i = threadIdx.x; i < np; i += 256) s += y[i] * y[i];
  const float nrm = sqrtf(block_sum_256(s, red));
  const float inv = nrm > 0.f ? 1.f / nrm : 0.f;
  for (int i = threadIdx.x; i < np; i += 256) x[i] = y[i] * inv;
  if (threadIdx.x == 0 && nrm > ctrl->scale && nrm < 3.0e38f) ctrl->scale = nrm;
}

// ---- 1. pivot solve -------------------------------------------------------------------------------
// Cyclic Jacobi on the 64x64 pivot in LDS.  One step = 32 disjoint (p,q) pairs (round-robin over the 64
// indices): (a) 32 lanes compute the rotations; (b) ONE fused phase applies J^T S J on disjoint 2x2
// blocks (block (k1,k2) = rows of pair k1 x columns of pair k2 sees exactly the row rotation k1 and
// the column rotation k2) and the column rotation on R.  Two barriers per step.
__global__ __launch_bounds__(256) void eig_pivot_kernel(EigRound rd, float tol_rel, float tol_abs, float tol_conv,
                                                        int max_inner, int cross_only) {
  const int slot = eig_slot(rd, blockIdx.x);
  const EigDesc ds = *rd.desc[slot];
  EigCtrl* ctrl = ds.ctrl;
  if (ctrl->converged) return;
  float* __restrict__ Aw = ds.Aw;
  float* __restrict__ Rws = ds.Rws;
  float* __restrict__ Dws = ds.Dws;
  int* __restrict__ rotated = ds.rotated;
  const int np = ds.np, nb = ds.nb, step = rd.step[slot];
  const int pv = blockIdx.x - rd.first[slot];  // this workgroup's pivot of the matrix
  __shared__ float S[EP][ELD];
  __shared__ float R[EP][ELD];
  __shared__ float cs[32][2];
  __shared__ unsigned char sched[EP - 1][32][2];
  __shared__ int any_rot;
  __shared__ int sweep_rot;
  __shared__ int sweep_big;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int I, J;
  pivot_blocks(step, pv, nb, I, J);
  int total_rot = 0;
  const float floor_abs = tol_abs * ctrl->scale;   // below this an off-diagonal is rounding noise: never rotate
  const float floor_conv = tol_conv * ctrl->scale; // rotations of elements below this do not count as "unconverged"

  // load the upper triangle of the pivot sub-matrix and mirror it
  for (int idx = tid; idx < EP * EP; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    const int gr = pivot_index(r, I, J), gc = pivot_index(c, I, J);
    const float v = (r <= c) ? Aw[(int64_t)gr * np + gc] : Aw[(int64_t)gc * np + gr];
    S[r][c] = v;
    R[r][c] = (r == c) ? 1.f : 0.f;
  }
  if (tid == 0) {
    sweep_rot = 0;
    sweep_big = 0;
  }
  __syncthreads();

  // quick exit: if no off-diagonal element of the pivot is above its rotation threshold, R_P = I and the 63-step
  // sweep (two barriers per step) would only re-check what this single pass establishes -- the common case in the
  // last sweeps, when most pivots are already diagonal
  {
    int any = 0;
    for (int idx = tid; idx < EP * EP; idx += 256) {
      const int r = idx >> 6, c = idx & 63;
      if (r < c) {
        const float mag = fabsf(S[r][c]);
        any |= (mag > floor_abs) && (mag > tol_rel * sqrtf(fabsf(S[r][r] * S[c][c])));
      }
    }
    if (!__syncthreads_or(any)) {
      if (tid == 0) rotated[pv] = 0;
      return;
    }
  }

  // Pair schedule, built once: sched[t][k] = (p, q), p < q, of pair k in step t.
  //  * step 0 of an outer sweep (every index block sits in exactly one pivot): the full round-robin over the 64
  //    indices, 63 steps -- this is where the diagonal blocks A_II get diagonalised, once per sweep;
  //  * every other step: only the 32 x 32 CROSS pairs (i, 32 + (i + t) mod 32), 32 steps -- annihilating A_IJ is
  //    what the block method needs from this visit, and it halves the latency chain of the solver.
  const int nsteps = (step == 0 || !cross_only) ? EP - 1 : 32;
  for (int e = tid; e < nsteps * 32; e += 256) {
    const int t = e >> 5, k = e & 31;
    int a, b;
    if (nsteps == 32) {
      a = k;
      b = 32 + ((k + t) & 31);
    } else if (k == 0) {
      a = EP - 1;
      b = t;
    } else {
      a = (t + k) % (EP - 1);
      b = (t - k + (EP - 1)) % (EP - 1);
    }
    sched[t][k][0] = (unsigned char)(a < b ? a : b);
    sched[t][k][1] = (unsigned char)(a < b ? b : a);
  }
  __syncthreads();

  // thread -> block mapping of the fused update: the 32 lanes of a half-wave share ONE row pair k1 and take the 32
  // column pairs k2 = lane.  With the LDS pitch of 65 the bank of S[p1][p2] is (p1 + p2) mod 32 and the p2 of
  // consecutive k2 are consecutive indices, so every 32-lane access group is conflict-free (the former 8 x 8
  // mapping mixed four row pairs per group: ~3-way conflicts on every access, and the LDS is what bounds this loop).
  const int k2 = tid & 31, k1base = tid >> 5;
  for (int sw = 0; sw < max_inner; ++sw) {
    for (int t = 0; t < nsteps; ++t) {
      // (a) rotation parameters of the 32 disjoint pairs of this step (first half of wave 0)
      if (tid < 64) {
        bool rot = false, big = false;
        if (tid < 32) {
          const int p = sched[t][tid][0], q = sched[t][tid][1];
          const float app = S[p][p], aqq = S[q][q], apq = S[p][q];
          float c = 1.f, s = 0.f;
          const float mag = fabsf(apq);
          if (mag > floor_abs && mag > tol_rel * __builtin_amdgcn_sqrtf(fabsf(app * aqq))) {
            // hardware reciprocal / rsqrt (1 ulp): the rotation only has to be orthogonal to ~1e-7, the eigenvalues
            // are recomputed as Rayleigh quotients and V is re-orthonormalised at the end
            const float tau = (aqq - app) * __builtin_amdgcn_rcpf(2.f * apq);
            const float tt = (tau >= 0.f ? 1.f : -1.f) * __builtin_amdgcn_rcpf(fabsf(tau) + __builtin_amdgcn_sqrtf(1.f + tau * tau));
            c = __builtin_amdgcn_rsqf(1.f + tt * tt);
            s = tt * c;
            if (!(s == s) || !(c == c)) {  // tau overflowed: the pair is (numerically) already diagonal
              c = 1.f;
              s = 0.f;
            } else {
              rot = true;
              big = mag > floor_conv;
            }
          }
          cs[tid][0] = c;
          cs[tid][1] = s;
        }
        const unsigned long long m = __ballot(rot);
        const unsigned long long mb = __ballot(big);
        if (tid == 0) {
          any_rot = (m != 0ull);
          sweep_rot += __popcll(m);
          sweep_big += __popcll(mb);
        }
      }
      __syncthreads();
      if (any_rot) {  // block-uniform
        // (b1) S <- J^T S J on 2x2 blocks: this thread's column pair k2 against four row pairs
        const float c2 = cs[k2][0], s2 = cs[k2][1];
        const int p2 = sched[t][k2][0], q2 = sched[t][k2][1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k1 = k1base + 8 * j;
          const float c1 = cs[k1][0], s1 = cs[k1][1];  // broadcast within the half-wave
          if (s1 != 0.f || s2 != 0.f) {
            const int p1 = sched[t][k1][0], q1 = sched[t][k1][1];
            const float b00 = S[p1][p2], b01 = S[p1][q2], b10 = S[q1][p2], b11 = S[q1][q2];
            const float t00 = c1 * b00 - s1 * b10, t01 = c1 * b01 - s1 * b11;
            const float t10 = s1 * b00 + c1 * b10, t11 = s1 * b01 + c1 * b11;
            float n00 = c2 * t00 - s2 * t01, n01 = s2 * t00 + c2 * t01;
            float n10 = c2 * t10 - s2 * t11, n11 = s2 * t10 + c2 * t11;
            if (k1 == k2) {  // the annihilated element is exactly zero
              n01 = 0.f;
              n10 = 0.f;
            }
            S[p1][p2] = n00;
            S[p1][q2] = n01;
            S[q1][p2] = n10;
            S[q1][q2] = n11;
          }
        }
        // (b2) R <- R J (columns), lane = row
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = wave * 8 + i;
          const float c = cs[k][0], s = cs[k][1];
          if (s != 0.f) {  // wave-uniform
            const int p = sched[t][k][0], q = sched[t][k][1];
            const float rp = R[lane][p], rq = R[lane][q];
            R[lane][p] = c * rp - s * rq;
            R[lane][q] = s * rp + c * rq;
          }
        }
      }
      __syncthreads();
    }
    const int r = sweep_rot;  // stable: written only in phase (a), last one is behind the barrier above
    __syncthreads();
    if (tid == 0) sweep_rot = 0;
    total_rot += r;
    if (r == 0) break;
  }

  // outputs: R_P (row-major 64x64), the solved pivot, and whether anything rotated at all (R_P == I otherwise,
  // which lets the tile updates of untouched pivot pairs be skipped in the late, mostly-converged sweeps)
  if (tid == 0) rotated[pv] = total_rot > 0 ? 1 : 0;
  if (total_rot == 0) return;
  // R_P for the tile updates: the two fp16 planes of R^T * 2^14 (h, l; [j][k] = R[k][j], k contiguous -- what a lane of
  // the 16-bit MFMA reads), split once here instead of by each of the ~2 n / 64 workgroups that use it
  _Float16* Rout = reinterpret_cast<_Float16*>(Rws + (int64_t)pv * EP * EP);
  for (int idx = tid; idx < EP * EP / 4; idx += 256) {
    const int j = idx >> 4, k0 = (idx & 15) * 4;
    f16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 hh, ll;
      eig_split2(R[k0 + e][j] * 16384.f, hh, ll);
      h[e] = hh, l[e] = ll;
    }
    *reinterpret_cast<f16x4*>(Rout + j * EP + k0) = h;
    *reinterpret_cast<f16x4*>(Rout + EP * EP + j * EP + k0) = l;
  }
  float* Sout = Dws + (int64_t)pv * EP * EP;  // the (nearly) diagonalised pivot itself
  for (int idx = tid; idx < EP * EP; idx += 256) Sout[idx] = S[idx >> 6][idx & 63];
  if (tid == 0 && sweep_big > 0) atomicAdd(&ctrl->rotations, sweep_big);
}

// ---- 2. tile update  A_PQ <- R_P^T A_PQ R_Q   and   3. eigenvector update  V[:, Q] <- V[:, Q] R_Q ----------------------
// The exact-fp32 MFMA (64 cycles per 32x32x2) made these products the bound of the whole solver: the n = 4608 update ran at
// 136 TFLOP/s, 87 % of that pipe's peak.  They now run on the fp16 matrix cores at fp32 level, as the convolutions do
// (lk_conv.hip): every operand is scaled by a power of two and split into two fp16 planes, x 2^s = h + l (+ <= 2^-22 |x 2^s|),
// a product is three v_mfma_f32_32x32x16_f16 (h h', h l', l h'; fp32 accumulation) -- 96 instead of 512 matrix-pipe cycles
// per 32x32x16 block.  Scales: rotations and eigenvectors (|x| <= 1) by 2^14; the matrix by 2^sa with
// bound * 2^sa in [2^13, 2^14), bound = max row sum >= lambda_max >= every entry of A, A R and R^T A R (sub-blocks of an
// orthogonal similarity), so nothing can overflow and an entry far below lambda_max keeps an ABSOLUTE accuracy of
// 2^-38 bound -- orders below the fp32 noise of the method.  The pivot tiles themselves (all large entries: the diagonal)
// are written back from the fp32 LDS solve and never pass through a split product.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int EHP = EP + 8;                    // plane pitch in halfs (144 B): 16-byte fragment reads spread over the banks
constexpr int EPLANE = EP * EHP;               // one fp16 plane of a 64 x 64 operand
struct EigTileLds {
  union {
    struct {
      _Float16 a[2][EPLANE];                   // A-operand planes (h, l): [row i][k]
      _Float16 b[2][EPLANE];                   // B-operand planes (h, l): [column j][k]
    } op;
    float out[EP][ELD];                        // result staging for coalesced stores (after the last product)
  };
};

// acc += A B for this wave's 32 x 32 quadrant, K = 64: A-operand rows wm*32 + lr, B-operand columns wn*32 + lr
__device__ __forceinline__ f32x16 quad_mm16(const EigTileLds& L, int wm, int wn, int lr, int lh) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const _Float16* pa = &L.op.a[0][(wm * 32 + lr) * EHP + lh * 8];
  const _Float16* pb = &L.op.b[0][(wn * 32 + lr) * EHP + lh * 8];
#pragma unroll
  for (int k16 = 0; k16 < EP / 16; ++k16) {
    const f16x8 ah = *reinterpret_cast<const f16x8*>(pa + k16 * 16);
    const f16x8 al = *reinterpret_cast<const f16x8*>(pa + EPLANE + k16 * 16);
    const f16x8 bh = *reinterpret_cast<const f16x8*>(pb + k16 * 16);
    const f16x8 bl = *reinterpret_cast<const f16x8*>(pb + EPLANE + k16 * 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);  // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
  }
  return acc;
}

// the planes of R^T (written by the pivot kernel) -> an operand slot; identity when the pivot did not rotate.  In two
// halves, so that the global loads of BOTH rotations of a tile are in flight before its first product starts.
struct EigRotRegs {
  f16x8 v[4];
};
__device__ __forceinline__ EigRotRegs eig_load_rot(const float* __restrict__ Rws, int pv, bool rotated, int tid) {
  EigRotRegs g;
  const _Float16* src = reinterpret_cast<const _Float16*>(Rws + (int64_t)pv * EP * EP);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = tid + 256 * e;
    const int pl = idx >> 9, j = (idx >> 3) & 63, k0 = (idx & 7) * 8;
    if (rotated) {
      g.v[e] = *reinterpret_cast<const f16x8*>(src + pl * EP * EP + j * EP + k0);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) g.v[e][i] = (pl == 0 && k0 + i == j) ? (_Float16)16384.f : (_Float16)0.f;
    }
  }
  return g;
}
__device__ __forceinline__ void eig_store_rot(_Float16 (*dst)[EPLANE], const EigRotRegs& g, int tid) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = tid + 256 * e;
    const int pl = idx >> 9, j = (idx >> 3) & 63, k0 = (idx & 7) * 8;
    *reinterpret_cast<f16x8*>(&dst[pl][j * EHP + k0]) = g.v[e];
  }
}

// a 64 x 64 fp32 block (rows gr(r), columns gc(c) of a row-major matrix of pitch np) * sc -> the A-operand planes
template <typename RowFn, typename ColFn>
__device__ __forceinline__ void eig_stage_block(EigTileLds& L, const float* __restrict__ M, int np, RowFn gr, ColFn gc, float sc,
                                                int tid) {
  float4 vv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = tid + 256 * e;
    const int r = idx >> 4, c0 = (idx & 15) * 4;  // (four consecutive columns never straddle a 32-wide index block)
    vv[e] = *reinterpret_cast<const float4*>(M + (int64_t)gr(r) * np + gc(c0));
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = tid + 256 * e;
    const int r = idx >> 4, c0 = (idx & 15) * 4;
    const float4 v = vv[e];
    f16x4 h, l;
    _Float16 hh, ll;
    eig_split2(v.x * sc, hh, ll), h[0] = hh, l[0] = ll;
    eig_split2(v.y * sc, hh, ll), h[1] = hh, l[1] = ll;
    eig_split2(v.z * sc, hh, ll), h[2] = hh, l[2] = ll;
    eig_split2(v.w * sc, hh, ll), h[3] = hh, l[3] = ll;
    *reinterpret_cast<f16x4*>(&L.op.a[0][r * EHP + c0]) = h;
    *reinterpret_cast<f16x4*>(&L.op.a[1][r * EHP + c0]) = l;
  }
}

// One launch per round: of a matrix' share of the grid, workgroups [0, ntiles) rotate the tiles of A, the remaining
// npv * (np/64) rotate V -- the two updates are independent of each other, so they share the launch and the chip.
__global__ __launch_bounds__(256) void eig_update_kernel(EigRound rd) {
  const int slot = eig_slot(rd, blockIdx.x);
  const EigDesc ds = *rd.desc[slot];
  if (ds.ctrl->converged) return;
  float* __restrict__ Aw = ds.Aw;
  float* __restrict__ V = ds.V;
  const float* __restrict__ Rws = ds.Rws;
  const float* __restrict__ Dws = ds.Dws;
  const int* __restrict__ rotated = ds.rotated;
  const int np = ds.np, nb = ds.nb, step = rd.step[slot];
  const int blk = blockIdx.x - rd.first[slot];  // this workgroup's index within the matrix' share
  __shared__ EigTileLds L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int npv = nb / 2;
  const int ntiles = npv * (npv + 1) / 2;
  if (blk >= ntiles) {  // ---- V[rb-th 64 rows, columns of pivot Q] <- (that block) R_Q
    const int v = blk - ntiles, Q = v % npv, rb = v / npv;
    if (!rotated[Q]) return;
    int IQ, JQ;
    pivot_blocks(step, Q, nb, IQ, JQ);
    const EigRotRegs rq = eig_load_rot(Rws, Q, true, tid);
    eig_stage_block(L, V, np, [&](int r) { return rb * EP + r; }, [&](int c) { return pivot_index(c, IQ, JQ); }, 16384.f, tid);
    eig_store_rot(L.op.b, rq, tid);
    __syncthreads();
    const f32x16 t = quad_mm16(L, wm, wn, lr, lh);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) L.out[wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh][wn * 32 + lr] = t[r] * (1.f / 268435456.f);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + 256 * e;
      const int r = idx >> 4, c0 = (idx & 15) * 4;
      const float4 o = make_float4(L.out[r][c0], L.out[r][c0 + 1], L.out[r][c0 + 2], L.out[r][c0 + 3]);
      *reinterpret_cast<float4*>(V + (int64_t)(rb * EP + r) * np + pivot_index(c0, IQ, JQ)) = o;
    }
    return;
  }
  // linear index over pivot pairs P <= Q
  int P = 0, rem = blk, rowlen = npv;
  while (rem >= rowlen) {
    rem -= rowlen;
    ++P;
    --rowlen;
  }
  const int Q = P + rem;
  if (!rotated[P] && !rotated[Q]) return;  // R_P = R_Q = I: the tile is unchanged
  int IP, JP, IQ, JQ;
  pivot_blocks(step, P, nb, IP, JP);
  pivot_blocks(step, Q, nb, IQ, JQ);

  if (P == Q) {  // the pivot itself: written back from the LDS solve (upper triangle mirrored)
    const float* Sp = Dws + (int64_t)P * EP * EP;
    for (int idx = tid; idx < EP * EP; idx += 256) {
      const int r = idx >> 6, c = idx & 63;
      Aw[(int64_t)pivot_index(r, IP, JP) * np + pivot_index(c, IP, JP)] = (r <= c) ? Sp[r * EP + c] : Sp[c * EP + r];
    }
    return;
  }
  // 2^sa: bound * 2^sa in [2^13, 2^14)
  int be = (int)((__float_as_uint(ds.ctrl->bound) >> 23) & 0xffu);
  if (be == 0) be = 1;
  int sa = 13 - (be - 127);
  sa = sa > 100 ? 100 : (sa < -100 ? -100 : sa);
  const float sc_a = __uint_as_float((unsigned)(127 + sa) << 23), inv_a = __uint_as_float((unsigned)(127 - sa) << 23);
  const EigRotRegs rq = eig_load_rot(Rws, Q, rotated[Q] != 0, tid);
  const EigRotRegs rp = eig_load_rot(Rws, P, rotated[P] != 0, tid);
  eig_stage_block(L, Aw, np, [&](int r) { return pivot_index(r, IP, JP); }, [&](int c) { return pivot_index(c, IQ, JQ); }, sc_a, tid);
  eig_store_rot(L.op.b, rq, tid);
  __syncthreads();
  const f32x16 t = quad_mm16(L, wm, wn, lr, lh);  // T 2^(sa+14) = (A_PQ 2^sa) (R_Q 2^14)
  __syncthreads();
  // T^T 2^sa -> the B-operand planes ([column j of T][k = row of T]: this lane holds four consecutive rows per group);
  // R_P^T -> the A-operand planes
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 hh, ll;
      eig_split2(t[4 * g + e] * (1.f / 16384.f), hh, ll);
      h[e] = hh, l[e] = ll;
    }
    const int off = (wn * 32 + lr) * EHP + wm * 32 + 8 * g + 4 * lh;
    *reinterpret_cast<f16x4*>(&L.op.b[0][off]) = h;
    *reinterpret_cast<f16x4*>(&L.op.b[1][off]) = l;
  }
  eig_store_rot(L.op.a, rp, tid);
  __syncthreads();
  const f32x16 m = quad_mm16(L, wm, wn, lr, lh);  // M 2^(sa+14) = (R_P^T 2^14) (T 2^sa)
  __syncthreads();
  const float un = inv_a * (1.f / 16384.f);
#pragma unroll
  for (int r = 0; r < 16; ++r) L.out[wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh][wn * 32 + lr] = m[r] * un;
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = tid + 256 * e;
    const int r = idx >> 4, c0 = (idx & 15) * 4;
    const float4 o = make_float4(L.out[r][c0], L.out[r][c0 + 1], L.out[r][c0 + 2], L.out[r][c0 + 3]);
    const float4 t4 = make_float4(L.out[c0][r], L.out[c0 + 1][r], L.out[c0 + 2][r], L.out[c0 + 3][r]);
    *reinterpret_cast<float4*>(Aw + (int64_t)pivot_index(r, IP, JP) * np + pivot_index(c0, IQ, JQ)) = o;
    *reinterpret_cast<float4*>(Aw + (int64_t)pivot_index(r, IQ, JQ) * np + pivot_index(c0, IP, JP)) = t4;  // mirrored tile
  }
}

__global__ void eig_sweep_end_kernel(EigRound rd) {  // one thread per matrix whose sweep ends with this round
  if ((int)threadIdx.x >= rd.njobs) return;
  EigCtrl* ctrl = rd.desc[threadIdx.x]->ctrl;
  if (ctrl->converged) return;
  if (ctrl->rotations == 0) ctrl->converged = 1;
  ctrl->rotations = 0;
  ctrl->sweeps += 1;
}

// ---- finalize: rank-sort ascending, clamp, gather eigenvector columns ----------------------------------
__global__ __launch_bounds__(256) void eig_rank_kernel(const float* __restrict__ d, int n, int* __restrict__ perm) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float li = d[i];
  int rank = 0;
  for (int j = 0; j < n; ++j) {
    const float lj = d[j];
    rank += (lj < li) || (lj == li && j < i);
  }
  perm[rank] = i;
}

__global__ __launch_bounds__(256) void eig_gather_kernel(const float* __restrict__ d, const float* __restrict__ V,
                                                         const int* __restrict__ perm, int n, int np, int clamp,
                                                         float* __restrict__ w, float* __restrict__ Q,
                                                         const EigCtrl* ctrl, int32_t* info) {
  const int64_t total = (int64_t)n * n;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int r = (int)(idx / n), k = (int)(idx - (int64_t)r * n);
    float v = V[(int64_t)r * np + perm[k]];
    if (!(v == v)) v = 0.f;
    Q[idx] = v;
    if (r == 0) {
      float l = d[perm[k]];
      if (!(l == l)) l = 0.f;
      if (clamp && l < 0.f) l = 0.f;
      w[k] = l;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && info != nullptr) {
    info[0] = ctrl->converged ? 0 : 1;
    info[1] = ctrl->sweeps;  // sweeps actually executed (diagnostic)
  }
}

// ---- refinement GEMMs:  C = alpha * op(A) * B + beta * D   (all np x np, np % 64 == 0) ----------------
template <bool TA>
__global__ __launch_bounds__(256) void eig_gemm_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                       const float* __restrict__ D, float* __restrict__ C, int np,
                                                       float alpha, float beta) {
  __shared__ float sA[TA ? 16 : 64][TA ? 65 : 17];
  __shared__ float sB[16][65];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, hi = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  // two-level accumulation: `acc` sums 256 consecutive k, `tot` sums the blocks -- a plain fp32 chain over
  // n = 4608 terms of one sign (dominant eigenvector) would cost ~1e-5 of relative accuracy in the refinement
  f32x16 acc, tot;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc[r] = 0.f;
    tot[r] = 0.f;
  }
  for (int k0 = 0; k0 < np; k0 += 16) {
    if ((k0 & 255) == 0 && k0 != 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        tot[r] += acc[r];
        acc[r] = 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + 256 * e;
      if (TA) {
        const int kk = idx >> 6, ii = idx & 63;
        sA[kk][ii] = A[(int64_t)(k0 + kk) * np + i0 + ii];
      } else {
        const int ii = idx >> 4, kk = idx & 15;
        sA[ii][kk] = A[(int64_t)(i0 + ii) * np + k0 + kk];
      }
      const int kb = idx >> 6, jj = idx & 63;
      sB[kb][jj] = B[(int64_t)(k0 + kb) * np + j0 + jj];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int k = 2 * kk + hi;
      const float a = TA ? sA[k][wm * 32 + lo] : sA[wm * 32 + lo][k];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, sB[k][wn * 32 + lo], acc, 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = i0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, col = j0 + wn * 32 + lo;
    float v = alpha * (tot[r] + acc[r]);
    if (D != nullptr) v += beta * D[(int64_t)row * np + col];
    C[(int64_t)row * np + col] = v;
  }
}

// d[i] = sum_r V[r][i] * T[r][i]   (Rayleigh quotients v_i^T A v_i)
__global__ __launch_bounds__(256) void eig_coldot_kernel(const float* __restrict__ V, const float* __restrict__ T,
                                                         int np, float* __restrict__ d) {
  __shared__ double red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  double s = 0.0;  // HBM-bound reduction: fp64 accumulation is free and keeps the Rayleigh quotient exact
  for (int r = part; r < np; r += 4) s += (double)V[(int64_t)r * np + col] * (double)T[(int64_t)r * np + col];
  red[part][threadIdx.x & 63] = s;
  __syncthreads();
  if (part == 0) {
    const int c = threadIdx.x & 63;
    d[col] = (float)((red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
  }
}

struct EigPlan {
  int np, nb, npv;
  size_t off_A, off_V, off_A0, off_T, off_R, off_D, off_perm, off_diag, off_rot, off_ctrl, off_desc, total;
};

static EigPlan eig_plan(int64_t n) {
  EigPlan p;
  p.np = (int)((n + EP - 1) / EP * EP);
  if (p.np < EP) p.np = EP;
  p.nb = p.np / EB;
  p.npv = p.nb / 2;
  size_t off = 0;
  p.off_A = off; off += align_up((size_t)p.np * p.np * 4, 256);
  p.off_V = off; off += align_up((size_t)p.np * p.np * 4, 256);
  p.off_A0 = off; off += align_up((size_t)p.np * p.np * 4, 256);
  p.off_T = off; off += align_up((size_t)p.np * p.np * 4, 256);
  p.off_R = off; off += align_up((size_t)p.npv * EP * EP * 4, 256);
  p.off_D = off; off += align_up((size_t)p.npv * EP * EP * 4, 256);
  p.off_perm = off; off += align_up((size_t)p.np * 4, 256);
  p.off_diag = off; off += align_up((size_t)p.np * 4, 256);
  p.off_rot = off; off += align_up((size_t)p.npv * 4, 256);
  p.off_ctrl = off; off += 256;
  p.off_desc = off; off += 256;
  p.total = off;
  return p;
}

}  // namespace lk

using namespace lk;

extern "C" size_t lk_syevj_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  return eig_plan(n).total;
}

namespace lk {

struct EigJob {  // one matrix in flight: buffers carved out of its workspace + the solver constants
  EigPlan p;
  const float* A;
  int64_t n;
  float *w, *Q;
  int32_t* info;
  float *Aw, *V, *A0, *T, *Rws, *Dws, *dvec;
  int *perm, *rotated;
  EigCtrl* ctrl;
  EigDesc* desc;
  int clamp;
};

static const float kTolRel = 3.0e-7f;   // ~2.5 eps: |a_pq| <= tol_rel*sqrt(|a_pp a_qq|) counts as annihilated
static const float kTolAbs = 6.0e-8f;   // x lambda_max: absolute floor for (numerically) rank-deficient factors
static const float kTolConv = 1.0e-6f;  // x lambda_max: only rotations of larger elements keep the solve "unconverged";
                                        // what is left below it is removed from the spectrum by the Rayleigh refinement
static int eig_max_inner() {
  // inner sweeps per pivot visit.  One is best on the MI355X: the outer sweep count does not change (measured on the
  // ResNet-18 KFAC factors, tools/eig_study.py: 1065 ms vs 1479 ms with three) and the pivot solve -- the latency
  // chain of the whole solver -- is three times shorter.
  return 1;
}

static int eig_cross_only() { return 1; }

static int eig_job_setup(EigJob& j, const float* A, int64_t n, float* w, float* Q, int clamp, int32_t* info, void* ws,
                         size_t ws_bytes) {
  j.p = eig_plan(n);
  if (ws == nullptr || ws_bytes < j.p.total) {
    set_error("lk_syevj_f32: workspace too small (%zu < %zu bytes)", ws_bytes, j.p.total);
    return LK_EWORKSPACE;
  }
  char* base = static_cast<char*>(ws);
  j.A = A, j.n = n, j.w = w, j.Q = Q, j.info = info, j.clamp = clamp;
  j.Aw = reinterpret_cast<float*>(base + j.p.off_A);
  j.V = reinterpret_cast<float*>(base + j.p.off_V);
  j.A0 = reinterpret_cast<float*>(base + j.p.off_A0);
  j.T = reinterpret_cast<float*>(base + j.p.off_T);
  j.Rws = reinterpret_cast<float*>(base + j.p.off_R);
  j.Dws = reinterpret_cast<float*>(base + j.p.off_D);
  j.perm = reinterpret_cast<int*>(base + j.p.off_perm);
  j.dvec = reinterpret_cast<float*>(base + j.p.off_diag);
  j.rotated = reinterpret_cast<int*>(base + j.p.off_rot);
  j.ctrl = reinterpret_cast<EigCtrl*>(base + j.p.off_ctrl);
  j.desc = reinterpret_cast<EigDesc*>(base + j.p.off_desc);
  return LK_OK;
}

static int eig_enqueue_init(const EigJob& j, hipStream_t stream) {
  const EigPlan& p = j.p;
  if (hipMemsetAsync(j.ctrl, 0, sizeof(EigCtrl), stream) != hipSuccess) {
    set_error("lk_syevj_f32: hipMemsetAsync failed");
    return LK_ELAUNCH;
  }
  EigDesc d;
  d.Aw = j.Aw, d.V = j.V, d.Rws = j.Rws, d.Dws = j.Dws, d.rotated = j.rotated, d.ctrl = j.ctrl, d.np = p.np, d.nb = p.nb;
  hipLaunchKernelGGL(eig_desc_kernel, dim3(1), dim3(1), 0, stream, d, j.desc);
  int64_t blocks = ((int64_t)p.np * p.np + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(eig_init_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, j.A, (int)j.n, p.np, j.Aw, j.A0, j.V,
                     j.ctrl);
  hipLaunchKernelGGL(eig_rowabs_kernel, dim3((p.np + 3) / 4), dim3(256), 0, stream, j.A0, p.np, j.ctrl);
  // lambda_max estimate -> ctrl->scale (x, y live in the not-yet-used T buffer)
  float* xv = j.T;
  float* yv = j.T + p.np;
  hipLaunchKernelGGL(eig_normalize_kernel, dim3(1), dim3(256), 0, stream, yv, p.np, xv, j.ctrl, 1);
  for (int it = 0; it < 8; ++it) {
    hipLaunchKernelGGL(eig_matvec_kernel, dim3((p.np + 3) / 4), dim3(256), 0, stream, j.A0, p.np, xv, yv);
    hipLaunchKernelGGL(eig_normalize_kernel, dim3(1), dim3(256), 0, stream, yv, p.np, xv, j.ctrl, 0);
  }
  return LK_OK;
}

// One round: the pivot solves and the tile / eigenvector updates of the given matrices, each at its own step; then the
// end-of-sweep bookkeeping of those whose sweep this round completes.
struct EigSlot {
  const EigJob* job;
  int step;
};
static void eig_enqueue_round(const EigSlot* slots, int nslots, hipStream_t stream) {
  EigRound pv, up, end;
  pv.njobs = up.njobs = nslots;
  end.njobs = 0;
  pv.first[0] = up.first[0] = 0;
  for (int i = 0; i < nslots; ++i) {
    const EigPlan& p = slots[i].job->p;
    pv.step[i] = up.step[i] = slots[i].step;
    pv.desc[i] = up.desc[i] = slots[i].job->desc;
    pv.first[i + 1] = pv.first[i] + p.npv;
    up.first[i + 1] = up.first[i] + p.npv * (p.npv + 1) / 2 + p.npv * (p.np / EP);
    if (slots[i].step == p.nb - 2) end.desc[end.njobs++] = slots[i].job->desc;
  }
  hipLaunchKernelGGL(eig_pivot_kernel, dim3((unsigned)pv.first[nslots]), dim3(256), 0, stream, pv, kTolRel, kTolAbs, kTolConv,
                     eig_max_inner(), eig_cross_only());
  hipLaunchKernelGGL(eig_update_kernel, dim3((unsigned)up.first[nslots]), dim3(256), 0, stream, up);
  if (end.njobs > 0) hipLaunchKernelGGL(eig_sweep_end_kernel, dim3(1), dim3(kEigBatch), 0, stream, end);
}

static void eig_enqueue_sweep(const EigJob& j, hipStream_t stream) {
  EigSlot sl{&j, 0};
  for (sl.step = 0; sl.step < j.p.nb - 1; ++sl.step) eig_enqueue_round(&sl, 1, stream);
}

// refinement: one Newton-Schulz step re-orthonormalises V (thousands of fp32 rotations leave
// ||V^T V - I|| ~ 1e-5), then the eigenvalues are recomputed as Rayleigh quotients against the ORIGINAL
// matrix, which removes the accumulated transformation error from the spectrum; then sort / clamp / gather.
static void eig_enqueue_finalize(const EigJob& j, hipStream_t stream) {
  const EigPlan& p = j.p;
  dim3 gg(p.np / 64, p.np / 64);
  hipLaunchKernelGGL((eig_gemm_kernel<true>), gg, dim3(256), 0, stream, j.V, j.V, (const float*)nullptr, j.T, p.np, 1.f, 0.f);
  hipLaunchKernelGGL((eig_gemm_kernel<false>), gg, dim3(256), 0, stream, j.V, j.T, j.V, j.Aw, p.np, -0.5f, 1.5f);  // Aw <- V2
  hipLaunchKernelGGL((eig_gemm_kernel<false>), gg, dim3(256), 0, stream, j.A0, j.Aw, (const float*)nullptr, j.T, p.np, 1.f, 0.f);
  hipLaunchKernelGGL(eig_coldot_kernel, dim3(p.np / 64), dim3(256), 0, stream, j.Aw, j.T, p.np, j.dvec);
  hipLaunchKernelGGL(eig_rank_kernel, dim3((unsigned)((j.n + 255) / 256)), dim3(256), 0, stream, j.dvec, (int)j.n, j.perm);
  int64_t gblocks = (j.n * j.n + 255) / 256;
  if (gblocks > 4096) gblocks = 4096;
  hipLaunchKernelGGL(eig_gather_kernel, dim3((unsigned)gblocks), dim3(256), 0, stream, j.dvec, j.Aw, j.perm, (int)j.n, p.np,
                     j.clamp, j.w, j.Q, j.ctrl, j.info);
}

}  // namespace lk

extern "C" int lk_syevj_f32(const float* A, int64_t n, float* w, float* Q, int clamp, int max_sweeps, int32_t* info,
                            void* ws, size_t ws_bytes, void* stream_) {
  LK_REQUIRE(A && w && Q && n >= 0 && n <= 32768, "lk_syevj_f32: bad arguments");
  if (n == 0) return LK_OK;
  hipStream_t stream = (hipStream_t)stream_;
  EigJob j;
  if (int rc = eig_job_setup(j, A, n, w, Q, clamp, info, ws, ws_bytes)) return rc;
  if (max_sweeps <= 0) max_sweeps = 24;
  if (int rc = eig_enqueue_init(j, stream)) return rc;
  // fully asynchronous: every sweep is enqueued; once the device-side `converged` flag is up the rest return at once
  for (int sweep = 0; sweep < max_sweeps; ++sweep) eig_enqueue_sweep(j, stream);
  eig_enqueue_finalize(j, stream);
  return check_launch("lk_syevj_f32");
}

// ---- many matrices: host-side scheduler -------------------------------------------------------------------------
// A KFAC posterior needs one decomposition per factor (43 for ResNet-18, n = 10 ... 4608).  All of them iterate in
// ROUNDS on streams[0]: one pivot launch and one update launch serve the current step of every matrix that is still
// running (up to kEigBatch at a time, the rest waits for a free slot), so the chip always sees the tiles of all of
// them at once instead of one matrix' latency chain per queue.  After a round that completes a matrix' sweep its
// `converged` flag is read back asynchronously into pinned memory; a matrix is never more than `kLookahead` sweeps
// ahead of its last harvested flag (its workgroups return at once after convergence, so running ahead costs launches
// only), a converged matrix is finalised at once on streams[1] (Newton-Schulz / Rayleigh GEMMs beside the rounds of
// the others) and its slot is handed to the next waiting matrix.  On return every given stream has been made to
// wait for all of it; the host has only ever slept when every running matrix was `kLookahead` sweeps ahead.
extern "C" int lk_syevj_batched_f32(int64_t count, const float* const* A, const int64_t* n, float* const* w,
                                    float* const* Q, int32_t* const* info, void* const* ws, const size_t* ws_bytes,
                                    int clamp, int max_sweeps, void* const* streams, int64_t nstreams) {
  LK_REQUIRE(count >= 0 && nstreams >= 1 && streams && (count == 0 || (A && n && w && Q && info && ws && ws_bytes)),
             "lk_syevj_batched_f32: bad arguments");
  if (count == 0) return LK_OK;
  if (max_sweeps <= 0) max_sweeps = 24;
  constexpr int kLookahead = 2;
  constexpr int kSlots = kLookahead + 1;
  std::vector<EigJob> jobs((size_t)count);
  std::vector<int> waiting;  // input order (the caller passes the largest first)
  for (int64_t i = 0; i < count; ++i) {
    LK_REQUIRE(n[i] >= 0 && n[i] <= 32768 && (n[i] == 0 || (A[i] && w[i] && Q[i])), "lk_syevj_batched_f32: bad matrix");
    if (n[i] == 0) continue;
    if (int rc = eig_job_setup(jobs[(size_t)i], A[i], n[i], w[i], Q[i], clamp, info[i], ws[i], ws_bytes[i])) return rc;
    waiting.push_back((int)i);
  }
  // lanes: the matrices are dealt (largest first, to the lane with the smaller sum of n^3) to up to two streams, each
  // iterating its share in rounds of its own -- the pivot solves of one lane (few workgroups, LDS latency) then run
  // beside the tile updates of the other (HBM-bound); the stream after the lanes takes the finalisations
  const int nlanes = nstreams >= 3 ? 2 : 1;
  hipStream_t fin = (hipStream_t)streams[nstreams > nlanes ? nlanes : nstreams - 1];
  int* hflags = nullptr;
  if (hipHostMalloc(reinterpret_cast<void**>(&hflags), sizeof(int) * kSlots * (size_t)count, hipHostMallocDefault) != hipSuccess) {
    set_error("lk_syevj_batched_f32: hipHostMalloc failed");
    return LK_ELAUNCH;
  }
  struct Run {        // a matrix that is iterating
    int job;
    int step = 0;     // its next round-robin step
    int enq = 0;      // sweeps enqueued completely
    int pending = 0;  // flag read-backs not yet harvested
    int head = 0;     // ring position of the oldest of them
    bool converged = false;
  };
  struct Readback {   // the flags read back behind one round
    hipEvent_t ev;
    std::vector<std::pair<int, int>> items;  // (job, slot of its ring)
  };
  struct Lane {
    hipStream_t stream;
    std::vector<int> waiting;
    size_t next_waiting = 0;
    std::vector<Run> active;
    std::vector<Readback> inflight;  // in stream order
    double load = 0.0;
  };
  std::vector<Lane> lanes((size_t)nlanes);
  for (int l = 0; l < nlanes; ++l) lanes[(size_t)l].stream = (hipStream_t)streams[l];
  for (int ji : waiting) {
    Lane* best = &lanes[0];
    for (Lane& L : lanes)
      if (L.load < best->load) best = &L;
    best->waiting.push_back(ji);
    best->load += (double)n[ji] * (double)n[ji] * (double)n[ji];
  }
  std::vector<hipEvent_t> spare, handed;
  int rc = LK_OK;
  auto new_event = [&](hipEvent_t* ev) {
    if (!spare.empty()) {
      *ev = spare.back();
      spare.pop_back();
      return true;
    }
    return hipEventCreateWithFlags(ev, hipEventDisableTiming) == hipSuccess;
  };
  auto fill_slots = [&](Lane& L) {
    while (rc == LK_OK && L.active.size() < (size_t)kEigBatch && L.next_waiting < L.waiting.size()) {
      Run r;
      r.job = L.waiting[L.next_waiting++];
      rc = eig_enqueue_init(jobs[(size_t)r.job], L.stream);
      L.active.push_back(r);
    }
  };
  EigSlot slots[kEigBatch];
  int slot_run[kEigBatch];
  // one pass over a lane: harvest flags, retire finished matrices, enqueue the next round; false = nothing to do now
  auto advance = [&](Lane& L) -> bool {
    hipStream_t main = L.stream;
    bool progressed = false;
    // harvest finished read-backs (in stream order)
    size_t done = 0;
    for (; done < L.inflight.size(); ++done) {
      const hipError_t q = hipEventQuery(L.inflight[done].ev);
      if (q == hipErrorNotReady) break;
      if (q != hipSuccess) {
        set_error("lk_syevj_batched_f32: %s", hipGetErrorString(q));
        rc = LK_ELAUNCH;
        return true;
      }
      for (const auto& it : L.inflight[done].items)
        for (Run& r : L.active)
          if (r.job == it.first) {
            r.converged = r.converged || hflags[(size_t)it.first * kSlots + it.second] != 0;
            ++r.head, --r.pending;
          }
      spare.push_back(L.inflight[done].ev);
      progressed = true;
    }
    L.inflight.erase(L.inflight.begin(), L.inflight.begin() + (long)done);
    // retire: converged, or out of sweeps (the gather kernel reports the device-side flag in `info`)
    bool retired = false;
    for (size_t i = 0; i < L.active.size();) {
      const Run& r = L.active[i];
      if (!(r.converged || r.enq >= max_sweeps)) {
        ++i;
        continue;
      }
      const EigJob& j = jobs[(size_t)r.job];
      if (fin != main) {
        hipEvent_t ev;
        if (!new_event(&ev) || hipEventRecord(ev, main) != hipSuccess || hipStreamWaitEvent(fin, ev, 0) != hipSuccess) {
          set_error("lk_syevj_batched_f32: event hand-over to the finalising stream failed");
          rc = LK_ELAUNCH;
          return true;
        }
        handed.push_back(ev);  // (another stream waits on it: not re-recorded, destroyed at the end)
      }
      eig_enqueue_finalize(j, fin);
      L.active.erase(L.active.begin() + (long)i);
      retired = true;
    }
    if (retired) {
      fill_slots(L);
      progressed = true;
      if (rc != LK_OK) return true;
    }
    // the next round: every running matrix that is not yet kLookahead sweeps ahead of its flags
    int ns = 0;
    for (size_t i = 0; i < L.active.size(); ++i)
      if (L.active[i].pending < kLookahead && L.active[i].enq < max_sweeps) {
        slots[ns].job = &jobs[(size_t)L.active[i].job];
        slots[ns].step = L.active[i].step;
        slot_run[ns++] = (int)i;
      }
    if (ns == 0) return progressed;
    eig_enqueue_round(slots, ns, main);
    Readback rb;
    for (int k = 0; k < ns; ++k) {
      Run& r = L.active[(size_t)slot_run[k]];
      const EigJob& j = jobs[(size_t)r.job];
      if (r.step < j.p.nb - 2) {
        ++r.step;
        continue;
      }
      r.step = 0, ++r.enq;  // the round completed this matrix' sweep: read its flag back
      const int slot = (r.head + r.pending) % kSlots;
      if (hipMemcpyAsync(&hflags[(size_t)r.job * kSlots + slot], &j.ctrl->converged, sizeof(int), hipMemcpyDeviceToHost, main) !=
          hipSuccess) {
        set_error("lk_syevj_batched_f32: flag read-back failed");
        rc = LK_ELAUNCH;
        return true;
      }
      ++r.pending;
      rb.items.emplace_back(r.job, slot);
    }
    if (!rb.items.empty()) {
      if (!new_event(&rb.ev) || hipEventRecord(rb.ev, main) != hipSuccess) {
        set_error("lk_syevj_batched_f32: event record failed");
        rc = LK_ELAUNCH;
        return true;
      }
      L.inflight.push_back(std::move(rb));
    }
    return true;
  };
  for (Lane& L : lanes) fill_slots(L);
  for (;;) {
    bool any_active = false, progressed = false;
    for (Lane& L : lanes) {
      if (L.active.empty() || rc != LK_OK) continue;
      any_active = true;
      progressed = advance(L) || progressed;
    }
    if (!any_active || rc != LK_OK) break;
    // every running matrix is kLookahead sweeps ahead of the device: nap instead of spinning
    if (!progressed) std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
  // speculative read-backs still in flight write into hflags: drain them before the pinned buffer goes away
  for (Lane& L : lanes)
    for (auto& rb : L.inflight) {
      (void)hipEventSynchronize(rb.ev);
      spare.push_back(rb.ev);
    }
  // every given stream waits for the rounds and for the finalisations
  if (rc == LK_OK) {
    std::vector<hipStream_t> producers;
    for (Lane& L : lanes) producers.push_back(L.stream);
    producers.push_back(fin);
    for (hipStream_t ps : producers) {
      hipEvent_t ev = nullptr;
      if (!new_event(&ev) || hipEventRecord(ev, ps) != hipSuccess) {
        set_error("lk_syevj_batched_f32: final event failed");
        rc = LK_ELAUNCH;
        break;
      }
      for (int64_t l = 0; l < nstreams; ++l)
        if ((hipStream_t)streams[l] != ps) (void)hipStreamWaitEvent((hipStream_t)streams[l], ev, 0);
      handed.push_back(ev);
    }
  }
  for (hipEvent_t ev : spare) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : handed) (void)hipEventDestroy(ev);
  (void)hipHostFree(hflags);
  if (rc != LK_OK) return rc;
  return check_launch("lk_syevj_batched_f32");
}
