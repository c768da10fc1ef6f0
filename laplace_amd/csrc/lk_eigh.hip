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
#include <chrono>
#include <cstdlib>
#include <thread>
#include <vector>

#include "lk_common.h"

namespace lk {

constexpr int EB = 32;       // index block
constexpr int EP = 64;       // pivot size (two blocks)
constexpr int ELD = 65;      // LDS pitch, conflict-free for column access

struct EigCtrl {             // lives in the workspace
  float scale;               // max(max |A_ii|, power-iteration estimate of lambda_max)
  int rotations;             // rotations performed in the current sweep
  int converged;             // set when a sweep performed none
  int sweeps;                // completed sweeps
};

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
__global__ __launch_bounds__(256) void eig_pivot_kernel(float* __restrict__ Aw, int np, int nb, int step,
                                                        float* __restrict__ Rws, float* __restrict__ Dws,
                                                        int* __restrict__ rotated, EigCtrl* ctrl, float tol_rel,
                                                        float tol_abs, float tol_conv, int max_inner, int cross_only) {
  if (ctrl->converged) return;
  __shared__ float S[EP][ELD];
  __shared__ float R[EP][ELD];
  __shared__ float cs[32][2];
  __shared__ unsigned char sched[EP - 1][32][2];
  __shared__ int any_rot;
  __shared__ int sweep_rot;
  __shared__ int sweep_big;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int I, J;
  pivot_blocks(step, blockIdx.x, nb, I, J);
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
      if (tid == 0) rotated[blockIdx.x] = 0;
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
  if (tid == 0) rotated[blockIdx.x] = total_rot > 0 ? 1 : 0;
  if (total_rot == 0) return;
  float* Rout = Rws + (int64_t)blockIdx.x * EP * EP;
  for (int idx = tid; idx < EP * EP; idx += 256) Rout[idx] = R[idx >> 6][idx & 63];
  float* Sout = Dws + (int64_t)blockIdx.x * EP * EP;  // the (nearly) diagonalised pivot itself
  for (int idx = tid; idx < EP * EP; idx += 256) Sout[idx] = S[idx >> 6][idx & 63];
  if (tid == 0 && sweep_big > 0) atomicAdd(&ctrl->rotations, sweep_big);
}

// ---- 2. tile update  A_PQ <- R_P^T A_PQ R_Q ----------------------------------------------------------
// out = X * Y (64x64x64) for this wave's 32x32 quadrant; X read by column-of-row (pitch ELD), Y by row
__device__ __forceinline__ f32x16 quad_mm_AB(const float (*X)[ELD], const float (*Y)[ELD], int wm, int wn, int lo,
                                             int hi) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 8
  for (int kk = 0; kk < EP / 2; ++kk) {
    const int k = 2 * kk + hi;
    const float a = X[wm * 32 + lo][k];  // A[i][k]
    const float b = Y[k][wn * 32 + lo];  // B[k][j]
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  return acc;
}
// out = X^T * Y
__device__ __forceinline__ f32x16 quad_mm_AtB(const float (*X)[ELD], const float (*Y)[ELD], int wm, int wn, int lo,
                                              int hi) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 8
  for (int kk = 0; kk < EP / 2; ++kk) {
    const int k = 2 * kk + hi;
    const float a = X[k][wm * 32 + lo];  // X^T[i][k] = X[k][i]
    const float b = Y[k][wn * 32 + lo];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  return acc;
}
__device__ __forceinline__ void quad_store(float (*Z)[ELD], const f32x16& acc, int wm, int wn, int lo, int hi) {
#pragma unroll
  for (int r = 0; r < 16; ++r) Z[wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi][wn * 32 + lo] = acc[r];
}

// One launch per round: workgroups [0, ntiles) rotate the tiles of A, the remaining npv * (np/64) rotate V
// (section 3 below) -- the two updates are independent of each other, so they share the launch and the chip.
__device__ void eig_vupdate_block(float (*X)[ELD], float (*Y)[ELD], float* __restrict__ V, int np, int nb, int step,
                                  const float* __restrict__ Rws, const int* __restrict__ rotated, int Q, int rb);

__global__ __launch_bounds__(256) void eig_update_kernel(float* __restrict__ Aw, float* __restrict__ V, int np, int nb,
                                                         int step, const float* __restrict__ Rws,
                                                         const float* __restrict__ Dws,
                                                         const int* __restrict__ rotated, const EigCtrl* ctrl,
                                                         int ntiles) {
  if (ctrl->converged) return;
  __shared__ float X[EP][ELD];
  __shared__ float Y[EP][ELD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, hi = lane >> 5, wm = wave >> 1, wn = wave & 1;
  // linear index over pivot pairs P <= Q
  const int npv = nb / 2;
  if ((int)blockIdx.x >= ntiles) {
    const int v = blockIdx.x - ntiles;
    eig_vupdate_block(X, Y, V, np, nb, step, Rws, rotated, v % npv, v / npv);
    return;
  }
  int P = 0, rem = blockIdx.x, rowlen = npv;
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
  const float* RP = Rws + (int64_t)P * EP * EP;
  const float* RQ = Rws + (int64_t)Q * EP * EP;
  const bool rotP = rotated[P] != 0, rotQ = rotated[Q] != 0;
  for (int idx = tid; idx < EP * EP; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    X[r][c] = Aw[(int64_t)pivot_index(r, IP, JP) * np + pivot_index(c, IQ, JQ)];
    Y[r][c] = rotQ ? RQ[idx] : (r == c ? 1.f : 0.f);
  }
  __syncthreads();
  f32x16 t = quad_mm_AB(X, Y, wm, wn, lo, hi);  // T = A_PQ R_Q
  __syncthreads();
  quad_store(X, t, wm, wn, lo, hi);  // X <- T
  for (int idx = tid; idx < EP * EP; idx += 256)
    Y[idx >> 6][idx & 63] = rotP ? RP[idx] : ((idx >> 6) == (idx & 63) ? 1.f : 0.f);
  __syncthreads();
  f32x16 m = quad_mm_AtB(Y, X, wm, wn, lo, hi);  // M = R_P^T T
  __syncthreads();
  quad_store(X, m, wm, wn, lo, hi);  // X <- M
  __syncthreads();
  for (int idx = tid; idx < EP * EP; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    Aw[(int64_t)pivot_index(r, IP, JP) * np + pivot_index(c, IQ, JQ)] = X[r][c];
    Aw[(int64_t)pivot_index(r, IQ, JQ) * np + pivot_index(c, IP, JP)] = X[c][r];  // mirrored tile
  }
}

// ---- 3. eigenvector update  V[:, Q] <- V[:, Q] R_Q -----------------------------------------------------
__device__ void eig_vupdate_block(float (*X)[ELD], float (*Y)[ELD], float* __restrict__ V, int np, int nb, int step,
                                  const float* __restrict__ Rws, const int* __restrict__ rotated, int Q, int rb) {
  if (!rotated[Q]) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, hi = lane >> 5, wm = wave >> 1, wn = wave & 1;
  int IQ, JQ;
  pivot_blocks(step, Q, nb, IQ, JQ);
  const float* RQ = Rws + (int64_t)Q * EP * EP;
  for (int idx = tid; idx < EP * EP; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    X[r][c] = V[(int64_t)(rb * EP + r) * np + pivot_index(c, IQ, JQ)];
    Y[r][c] = RQ[idx];
  }
  __syncthreads();
  f32x16 t = quad_mm_AB(X, Y, wm, wn, lo, hi);
  __syncthreads();
  quad_store(X, t, wm, wn, lo, hi);
  __syncthreads();
  for (int idx = tid; idx < EP * EP; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    V[(int64_t)(rb * EP + r) * np + pivot_index(c, IQ, JQ)] = X[r][c];
  }
}

__global__ void eig_sweep_end_kernel(EigCtrl* ctrl) {
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
  size_t off_A, off_V, off_A0, off_T, off_R, off_D, off_perm, off_diag, off_rot, off_ctrl, total;
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
  static int v = [] {
    int r = 1;
    if (const char* e = getenv("LK_EIG_INNER")) {  // tuning knob (tools/eig_study.py)
      const int t = atoi(e);
      if (t >= 1 && t <= 8) r = t;
    }
    return r;
  }();
  return v;
}

static int eig_cross_only() {
  static int v = [] {
    const char* e = getenv("LK_EIG_CROSS");  // tuning knob (tools/eig_study.py); default on
    return (e && atoi(e) == 0) ? 0 : 1;
  }();
  return v;
}

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
  return LK_OK;
}

static int eig_enqueue_init(const EigJob& j, hipStream_t stream) {
  const EigPlan& p = j.p;
  if (hipMemsetAsync(j.ctrl, 0, sizeof(EigCtrl), stream) != hipSuccess) {
    set_error("lk_syevj_f32: hipMemsetAsync failed");
    return LK_ELAUNCH;
  }
  int64_t blocks = ((int64_t)p.np * p.np + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(eig_init_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, j.A, (int)j.n, p.np, j.Aw, j.A0, j.V,
                     j.ctrl);
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

static void eig_enqueue_sweep(const EigJob& j, hipStream_t stream) {
  const EigPlan& p = j.p;
  const int steps = p.nb - 1;
  const int ntiles = p.npv * (p.npv + 1) / 2;
  const int nvblk = p.npv * (p.np / EP);
  const int inner = eig_max_inner();
  for (int s = 0; s < steps; ++s) {
    hipLaunchKernelGGL(eig_pivot_kernel, dim3(p.npv), dim3(256), 0, stream, j.Aw, p.np, p.nb, s, j.Rws, j.Dws, j.rotated,
                       j.ctrl, kTolRel, kTolAbs, kTolConv, inner, eig_cross_only());
    hipLaunchKernelGGL(eig_update_kernel, dim3(ntiles + nvblk), dim3(256), 0, stream, j.Aw, j.V, p.np, p.nb, s, j.Rws,
                       j.Dws, j.rotated, j.ctrl, ntiles);
  }
  hipLaunchKernelGGL(eig_sweep_end_kernel, dim3(1), dim3(1), 0, stream, j.ctrl);
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
// A KFAC posterior needs one decomposition per factor (42 for ResNet-18, n = 10 ... 4608).  Enqueuing them one after
// the other serialises the whole job on the host: a single solve is ~10^4 dependent launches, the device queue
// back-pressures the enqueuing thread, and the other streams starve.  The scheduler below keeps every stream
// exactly `kLookahead` sweeps ahead of the device: per stream it enqueues one sweep of the current matrix, an
// asynchronous read-back of its `converged` flag into pinned memory and an event, then moves on to the next
// stream; flags are polled without blocking, a converged matrix is finalised immediately (no empty sweeps) and the
// stream's next matrix starts.  The host only sleeps on an event when every stream is already `kLookahead` ahead.
extern "C" int lk_syevj_batched_f32(int64_t count, const float* const* A, const int64_t* n, float* const* w,
                                    float* const* Q, int32_t* const* info, void* const* ws, const size_t* ws_bytes,
                                    int clamp, int max_sweeps, void* const* streams, int64_t nstreams) {
  LK_REQUIRE(count >= 0 && nstreams >= 1 && streams && (count == 0 || (A && n && w && Q && info && ws && ws_bytes)),
             "lk_syevj_batched_f32: bad arguments");
  if (count == 0) return LK_OK;
  if (max_sweeps <= 0) max_sweeps = 24;
  constexpr int kLookahead = 2;
  constexpr int kSlots = kLookahead + 1;
  struct Lane {                 // one stream and the matrices queued on it (round-robin assignment, input order)
    hipStream_t stream;
    std::vector<int> jobs;
    size_t cur = 0;             // index into jobs
    bool started = false;
    int enq = 0;                // sweeps enqueued for the current matrix
    int pending_head = 0, pending = 0;  // ring of outstanding flag read-backs (<= kLookahead)
    hipEvent_t ev[kSlots];
  };
  std::vector<EigJob> jobs((size_t)count);
  for (int64_t i = 0; i < count; ++i) {
    LK_REQUIRE(n[i] >= 0 && n[i] <= 32768 && (n[i] == 0 || (A[i] && w[i] && Q[i])), "lk_syevj_batched_f32: bad matrix");
    if (n[i] == 0) continue;
    if (int rc = eig_job_setup(jobs[(size_t)i], A[i], n[i], w[i], Q[i], clamp, info[i], ws[i], ws_bytes[i])) return rc;
  }
  const int64_t nl = nstreams < count ? nstreams : count;
  std::vector<Lane> lanes((size_t)nl);
  int* hflags = nullptr;
  if (hipHostMalloc(reinterpret_cast<void**>(&hflags), sizeof(int) * kSlots * (size_t)nl, hipHostMallocDefault) != hipSuccess) {
    set_error("lk_syevj_batched_f32: hipHostMalloc failed");
    return LK_ELAUNCH;
  }
  int rc = LK_OK;
  for (int64_t l = 0; l < nl; ++l) {
    lanes[(size_t)l].stream = (hipStream_t)streams[l];
    for (int k = 0; k < kSlots; ++k)
      if (hipEventCreateWithFlags(&lanes[(size_t)l].ev[k], hipEventDisableTiming) != hipSuccess) rc = LK_ELAUNCH;
  }
  for (int64_t i = 0; i < count; ++i)
    if (n[i] > 0) lanes[(size_t)(i % nl)].jobs.push_back((int)i);

  size_t active = 0;
  for (auto& L : lanes) active += L.cur < L.jobs.size();
  while (active > 0 && rc == LK_OK) {
    bool progressed = false;
    for (size_t li = 0; li < lanes.size() && rc == LK_OK; ++li) {
      Lane& L = lanes[li];
      if (L.cur >= L.jobs.size()) continue;
      const EigJob& j = jobs[(size_t)L.jobs[L.cur]];
      if (!L.started) {
        rc = eig_enqueue_init(j, L.stream);
        L.started = true, L.enq = 0, L.pending = 0, L.pending_head = 0;
        progressed = true;
        if (rc != LK_OK) break;
      }
      // harvest finished read-backs (in order)
      bool converged = false;
      while (L.pending > 0) {
        const int slot = L.pending_head % kSlots;
        const hipError_t q = hipEventQuery(L.ev[slot]);
        if (q == hipErrorNotReady) break;
        if (q != hipSuccess) {
          set_error("lk_syevj_batched_f32: %s", hipGetErrorString(q));
          rc = LK_ELAUNCH;
          break;
        }
        converged = converged || hflags[li * kSlots + slot] != 0;
        ++L.pending_head, --L.pending;
        progressed = true;
      }
      if (rc != LK_OK) break;
      if (converged || L.enq >= max_sweeps) {
        eig_enqueue_finalize(j, L.stream);
        ++L.cur, L.started = false;
        if (L.cur >= L.jobs.size()) --active;
        progressed = true;
        continue;
      }
      if (L.pending < kLookahead) {
        eig_enqueue_sweep(j, L.stream);
        const int slot = (L.pending_head + L.pending) % kSlots;
        if (hipMemcpyAsync(&hflags[li * kSlots + slot], &j.ctrl->converged, sizeof(int), hipMemcpyDeviceToHost, L.stream) !=
                hipSuccess ||
            hipEventRecord(L.ev[slot], L.stream) != hipSuccess) {
          set_error("lk_syevj_batched_f32: flag read-back failed");
          rc = LK_ELAUNCH;
          break;
        }
        ++L.enq, ++L.pending;
        progressed = true;
      }
    }
    // every lane is kLookahead sweeps ahead of the device: nap instead of spinning (a blocking wait on ONE lane's
    // event would starve the lanes whose small matrices finish sooner)
    if (!progressed && rc == LK_OK) std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
  // speculative read-backs still in flight write into hflags: drain them before the pinned buffer goes away
  for (auto& L : lanes) {
    for (int k = 0; k < kSlots; ++k) {
      (void)hipEventSynchronize(L.ev[k]);
      (void)hipEventDestroy(L.ev[k]);
    }
  }
  (void)hipHostFree(hflags);
  if (rc != LK_OK) return rc;
  return check_launch("lk_syevj_batched_f32");
}
