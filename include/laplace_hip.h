/*
 * laplace_hip.h — C ABI of the MI355X (gfx950) curvature kernels behind
 * `laplace_amd.HipGGN`, the drop-in `laplace.curvature.CurvatureInterface` backend.
 *
 * Conventions (all entry points):
 *   - every tensor is a raw DEVICE pointer to fp32 (labels: int64) + explicit sizes; the caller
 *     (torch) owns every buffer; kernels never allocate.  Scratch comes from a caller-provided
 *     workspace sized by the matching lk_*_workspace_bytes().
 *   - `stream` is a hipStream_t passed as void*; work is enqueued asynchronously on it.  No
 *     entry point synchronises with the host.
 *   - return 0 on success, negative LK_E* on error; lk_last_error() returns a thread-local
 *     message for the last failure on the calling thread.
 *   - matrices are row-major and dense (leading dimension = number of columns) unless stated.
 *
 * Each function cites the reference code (relative to aleximmer/Laplace @ 0.2.3) it replaces.
 */
#ifndef LAPLACE_HIP_H
#define LAPLACE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LK_OK 0
#define LK_EINVAL (-1)    /* bad argument */
#define LK_EWORKSPACE (-2) /* workspace too small */
#define LK_ELAUNCH (-3)    /* HIP launch/runtime failure */
#define LK_ENOTCONV (-4)   /* eigensolver did not converge (reported through `info`) */

/* flags for the Gram (factor accumulation) family */
#define LK_GRAM_UPPER_ONLY 1u /* update only the upper block triangle; caller runs lk_symmetrize_f32 later */
#define LK_GRAM_SLABS_PERSIST 2u /* the workspace is a zero-initialised, caller-owned accumulator of split-K partial
                                    tiles that persists across launches: `slab += partial`, C is not touched; reduce
                                    once with lk_gram_slabs_reduce_f32 (a sum over minibatches is linear) */

int lk_version(void);
const char* lk_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Likelihood: square root of the softmax-CE Hessian + loss.
 * Replaces GGNInterface._get_functional_hessian (laplace/curvature/curvature.py:366-373) and the
 * `factor * lossfunc(f, y)` evaluations (curvature.py:408,417; curvlinops.py:106).
 *   S[c][n][j] = delta_jc*sqrt(p_nc) - p_nj*sqrt(p_nc)   (column c of a root of diag(p)-pp^T)
 *   loss_accum[0] += sum_n -log p_n[y_n]                  (skipped when y or loss_accum is NULL)
 * f: [B][C] logits; S: [C][B][C] (one backward seed per class, `is_grads_batched` layout).
 *
 * The loss is reduced in a FIXED order (per-workgroup partials in `ws`, then one wave sums them in index order):
 * run-to-run deterministic, no floating-point atomics.  `ws`: lk_loss_workspace_bytes(B) bytes, needed whenever the
 * loss is requested.  Labels outside [0, C) contribute nothing to the loss (CrossEntropyLoss's ignore_index).
 * ------------------------------------------------------------------------------------------- */
size_t lk_loss_workspace_bytes(int64_t B);
int lk_softmax_hess_sqrt_f32(const float* f, const int64_t* y, int64_t B, int64_t C, float* S,
                             float* loss_accum, float* ws, void* stream);

/* Same contract with the rank-revealing root: the softmax Hessian has rank C-1, and its closed-form Cholesky
 * factor  L[j][j] = sqrt(p_j s_{j+1}/s_j),  L[i][j] = -p_i sqrt(p_j/(s_j s_{j+1})) (i > j),  s_j = sum_{k>=j} p_k
 * has only C-1 non-zero columns: S is [C-1][B][C] and the fit needs one reverse pass fewer.
 * 2 <= C <= LK_SOFTMAX_CHOL_MAX_C (4 waves x (2C+1) floats of LDS per workgroup). */
#define LK_SOFTMAX_CHOL_MAX_C 2000
int lk_softmax_hess_chol_f32(const float* f, const int64_t* y, int64_t B, int64_t C, float* S,
                             float* loss_accum, float* ws, void* stream);

/* loss_accum[0] += scale * sum (f - y)^2 over `numel` elements (MSELoss(sum) * factor); fixed-order reduction
 * through `ws` (>= 4 KiB). */
int lk_sq_err_sum_f32(const float* f, const float* y, int64_t numel, float scale, float* loss_accum,
                      float* ws, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Gram / factor accumulation:  C += alpha * X^T X   (fp32 MFMA, split-K, deterministic).
 * Replaces the A^T A / G^T G accumulations inside curvlinops' KFACLinearOperator._compute_kfac
 * as consumed by CurvlinopsInterface.kron (laplace/curvature/curvlinops.py:55-108), and the
 * dense einsums of GGNInterface.full / EFInterface.full (curvature.py:406,409,491).
 *
 *  _tn  : X is [K][n] row-major with leading dimension ldx      (nn.Linear inputs / output grads)
 *  _nt  : X is [nb][n][L] row-major; C += alpha * sum_b X_b X_b^T (NCHW conv output grads).  With L % 4 == 0 and
 *         16-byte aligned X the products run on the bf16 matrix cores at fp32 accuracy: every operand is split once
 *         into three bf16 pieces (x = h + m + l exactly) and six v_mfma_f32_32x32x16_bf16 form hh' + hm' + mh' + mm' +
 *         hl' + lh' (dropped terms <= 3 * 2^-24 |x y|, accumulation in fp32); otherwise v_mfma_f32_32x32x2_f32.
 *  _conv: x is an NHWC activation [B][H][W][Cin]; the Gram matrix of the *unfolded* patch matrix
 *         (rows = (b,oh,ow), columns = (kh,kw,ci)) is accumulated WITHOUT materialising it.
 *         Column order of C is (kh,kw,ci) ("native"); lk_permute_sym_f32 converts to the
 *         reference's F.unfold order (ci,kh,kw).
 * C is [n][n] (ldc = n).  Workspace: lk_gram_workspace_bytes(n, K) for _tn (K rows) and _conv (K = B*OH*OW);
 * lk_gram_nt_workspace_bytes(nb_total, n, L) for _nt / _nt_seg (nb_total = nseg * nb images of L positions).
 * ------------------------------------------------------------------------------------------- */
size_t lk_gram_workspace_bytes(int64_t n, int64_t K);
size_t lk_gram_nt_workspace_bytes(int64_t nb_total, int64_t n, int64_t L);
/* C += alpha * sum of the persistent slabs (LK_GRAM_SLABS_PERSIST) of an n x n product; L_nt = L of the _nt launches
 * that filled them (0 for _tn / _conv); `slabs` is modified.  flags: LK_GRAM_UPPER_ONLY as for the launches. */
int lk_gram_slabs_reduce_f32(float* slabs, size_t slabs_bytes, int64_t n, int64_t L_nt, float alpha, float* C,
                             unsigned flags, void* stream);
int lk_gram_tn_f32(const float* X, int64_t K, int64_t n, int64_t ldx, float alpha, float* C,
                   unsigned flags, void* ws, size_t ws_bytes, void* stream);
int lk_gram_nt_f32(const float* X, int64_t nb, int64_t n, int64_t L, float alpha, float* C,
                   unsigned flags, void* ws, size_t ws_bytes, void* stream);
/* Same, with X given as `nseg` (<= 16) separate [nb][n][L] tensors (HOST array of device pointers):
 * the per-seed output gradients of the C backward passes are consumed where autograd left them,
 * without stacking them into one buffer. */
int lk_gram_nt_seg_f32(const float* const* segs, int64_t nseg, int64_t nb, int64_t n, int64_t L, float alpha,
                       float* C, unsigned flags, void* ws, size_t ws_bytes, void* stream);
int lk_gram_conv_nhwc_f32(const float* x, int64_t B, int64_t H, int64_t W, int64_t Cin, int kh, int kw,
                          int sh, int sw, int ph, int pw, int dh, int dw, float alpha, float* C,
                          unsigned flags, void* ws, size_t ws_bytes, void* stream);

/* Pixel-pair form of the same factor for SMALL maps (4x4 and below): the caller accumulates the pixel-pair Gram
 *   Cp[(q,ci),(q',cj)] += sum_b x[b,q,ci] x[b,q',cj]     (lk_gram_tn_f32 on the NHWC images flattened to rows, then
 * lk_symmetrize_f32) over ALL minibatches of a fit -- it is linear in the data -- and this call assembles the patch
 * Gram from it once:  A[(d,ci),(e,cj)] += alpha * sum_{p: p+d, p+e in the grid} Cp[(p+d,ci),(p+e,cj)], native
 * (kh,kw,ci) order.  Cp: [H*W*Cin][H*W*Cin], both triangles valid.  A: [9*Cin][9*Cin]. */
int lk_conv3x3_pixgram_assemble_f32(const float* Cp, int64_t H, int64_t W, int64_t Cin, float alpha, float* A,
                                    void* stream);

/* Banded pixel-pair form for ANY map size (Cin % 64 == 0): only the pixel pairs (q, q + D) a 3x3 window can see are
 * kept, D in the half plane {(0,0),(0,1),(0,2),(1,-2..2),(2,-2..2)}: `n_blocks` blocks Blk[q,D] = [Cin][Cin], stored
 * back to back, accumulated over all minibatches of a fit and assembled once:
 *   _plan      : tile edge (64 / 128), number of T x T launch tiles and of blocks for a geometry
 *   _tables    : fills HOST tables  tiles[n_tiles][3] = (column of the A panel, column of the B panel, output offset)
 *                and  slots[H*W][13] = block index of (q, D) or -1; the caller uploads them once per geometry
 *   _accumulate: Blk[q,D] += alpha * sum_b x[b,q,:]^T x[b,q+D,:]   (x: NHWC [B][H][W][Cin]; tiles_dev on the device)
 *   _assemble  : A[(d,ci),(e,cj)] += alpha * sum_{p: p+d, p+e in the map} Blk[p+d, e-d][ci][cj]  (native order)
 * Replaces, like lk_conv3x3_shiftcorr_f32, the per-minibatch A^T A of curvlinops' KFAC for such layers. */
int lk_conv3x3_pixpair_plan(int64_t H, int64_t W, int64_t Cin, int64_t* tile, int64_t* n_tiles, int64_t* n_blocks);
int lk_conv3x3_pixpair_tables(int64_t H, int64_t W, int64_t Cin, int32_t* tiles, int32_t* slots);
int lk_conv3x3_pixpair_accumulate_f32(const float* x, int64_t B, int64_t H, int64_t W, int64_t Cin, float alpha,
                                      float* blocks, const int32_t* tiles_dev, int64_t n_tiles, void* stream);
int lk_conv3x3_pixpair_assemble_f32(const float* blocks, const int32_t* slots_dev, int64_t H, int64_t W, int64_t Cin,
                                    float alpha, float* A, void* stream);
/* _assemble of blocks + blocks2 (blocks2 optional): the accumulators of the two lanes of a fit with two minibatches in
 * flight are summed on the fly, element by element, in the same order as a separate addition would.  upper_only != 0:
 * only the (d, e) blocks with e >= d are written — the upper triangle of A, which is all lk_finalize_factors_f32 and
 * lk_pack_upper_f32 read; the mirrored blocks are one 4-byte access per element and were half of the launch. */
int lk_conv3x3_pixpair_assemble2_f32(const float* blocks, const float* blocks2, const int32_t* slots_dev, int64_t H, int64_t W,
                                     int64_t Cin, float alpha, float* A, int upper_only, void* stream);
/* _accumulate on a split tensor (csrc/lk_sweep16.hip): x as two fp16 planes with one power-of-two scale (lk_split_f16x2 of
 * the NHWC images), three fp16 MFMAs per fp32 product block instead of the exact-fp32 MFMA; same tables, same blocks. */
int lk_conv3x3_pixpair_accumulate_f16x2(const void* x_h, const void* x_l, const int* sexp, int64_t B, int64_t H, int64_t W,
                                        int64_t Cin, float alpha, float* blocks, const int32_t* tiles_dev, int64_t n_tiles,
                                        const void* zero16, void* stream);
/* The same for Cin == 64 with one workgroup per pixel: the pixel's panel is staged once for its (up to) 13 shifts instead of
 * once per (pixel, shift) block — the launch was bound by what a CU ingests, not by the matrix pipe.  slots_dev: the
 * [H W][13] slot table of lk_conv3x3_pixpair_tables (-1: the shifted pixel is outside the map). */
int lk_conv3x3_pixpair_accumulate13_f16x2(const void* x_h, const void* x_l, const int* sexp, int64_t B, int64_t H, int64_t W,
                                          int64_t Cin, float alpha, float* blocks, const int32_t* slots_dev, const void* zero16,
                                          void* stream);

/* Same result as lk_gram_conv_nhwc_f32 for a 3x3 / stride 1 / padding 1 / dilation 1 convolution, through the
 * shift-correlation identity (the input grid equals the output grid, so the 81 (offset, offset) blocks of the
 * patch Gram matrix depend only on the 25 offset differences plus boundary-strip corrections):
 *   C[(d,ci),(e,cj)] += alpha * ( R[e-d] - Row_d[e-d] - Col_d[e-d] + Pix_d[e-d] )[ci][cj]
 * 13 full-grid channel correlations (R[-D] = R[D]^T) instead of the 40.5 block products of the symmetric half.
 * C is written in the native (kh,kw,ci) order, both triangles.  H, W >= 2. */
size_t lk_conv3x3_shiftcorr_workspace_bytes(int64_t B, int64_t H, int64_t W, int64_t Cin);
int lk_conv3x3_shiftcorr_f32(const float* x, int64_t B, int64_t H, int64_t W, int64_t Cin, float alpha, float* C,
                             void* ws, size_t ws_bytes, void* stream);

/* dst[b][h][w][c] = src[b][c][h][w]  (NCHW -> NHWC staging for lk_gram_conv_nhwc_f32). */
int lk_nchw_to_nhwc_f32(const float* src, int64_t B, int64_t C, int64_t HW, float* dst, void* stream);

/* Mirror the upper block triangle written under LK_GRAM_UPPER_ONLY into the lower one. */
int lk_symmetrize_f32(float* C, int64_t n, void* stream);

/* dst[(ci*KK+d), (cj*KK+e)] (+)= src[(d*Cin+ci), (e*Cin+cj)]  — native (kh,kw,ci) -> unfold (ci,kh,kw)
 * order.  accumulate != 0 adds into dst. */
int lk_permute_sym_f32(const float* src, int64_t Cin, int64_t KK, float* dst, int accumulate, void* stream);

/* The once-per-fit layout pass of `count` factors in ONE launch (what CurvlinopsInterface.kron does per minibatch with
 * torch ops on the library's factors, laplace/curvature/curvlinops.py:55-75; here once, on the accumulated sums).
 * Host arrays of length count.  Factor i is n[i] x n[i], upper triangle valid:
 *   kk[i] <= 1: mirrored (in place if dst[i] == NULL), with C[r][c] *= scale[i][r] * scale[i][c] first when scale[i] is
 *               given (the deferred BatchNorm scale of a G factor: diag(s) G diag(s));
 *   kk[i]  > 1: dst[i] = the full matrix in F.unfold's (ci, kh, kw) order from src[i] in the kernels' (kh, kw, ci) order
 *               (n = cin * kk, dst != src, no scale) -- lk_symmetrize_f32 + lk_permute_sym_f32 in one pass. */
int lk_finalize_factors_f32(int64_t count, const float* const* src, float* const* dst, const float* const* scale,
                            const int64_t* n, const int64_t* cin, const int64_t* kk, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Convolution of the seed-batched reverse sweep (csrc/lk_conv.hip): implicit GEMM on NHWC tensors, fp32-level
 * products from two-piece fp16 operands (three v_mfma_f32_32x32x16_f16 per fp32 product block).
 * Replaces the backward-data convolutions inside the C reverse passes of curvlinops' KFACLinearOperator._compute_kfac
 * as driven by CurvlinopsInterface.kron (laplace/curvature/curvlinops.py:87-100); the same GEMM is the forward conv.
 *
 * A "split tensor" is a pair of fp16 planes (h, l) of the tensor's layout plus device ints `sexp`:
 *     x * 2^sexp = h + l (+ e, |e| <= max(2^-22 |x 2^sexp|, 2^-25)),   max|x| * 2^sexp < 2^15,
 * ONE scale for the tensor (the reverse sweep's cotangents: what consumes them are sums over samples) or ONE PER IMAGE of
 * the leading dimension, sexp[n] (the forward's activations: lk_split_images_f16x2, lk_bn_act_fwd_nhwc_f16x2) — the
 * reference computes every sample in fp32 whatever else is in its minibatch (laplace/curvature/curvature.py:375-433), and
 * a ReLU mask decided on an activation resolved only relative to the LARGEST image of the minibatch is not that.
 * ------------------------------------------------------------------------------------------- */
/* out[0] = bit pattern of max_i |x[i] * cscale[(i / inner) % C]| (cscale may be NULL); deterministic. */
int lk_absmax_f32(const float* x, int64_t n, const float* cscale, int64_t inner, int64_t C, unsigned* out, void* stream);
/* fp32 -> split tensor; the scale comes from the DEVICE value amax[0] * bound_mul (any upper bound of max|x|).
 * n % 8 == 0. */
int lk_split_f16x2(const float* x, int64_t n, const float* amax, float bound_mul, void* planes_h, void* planes_l,
                   int* sexp, void* stream);
/* x [N][per] fp32 -> split tensor with one scale per image: sexp[n] from the image's own max|x_n|, which is also left in
 * amax[n] (bit pattern of a float: the measured maximum the next producer derives its bound from).  per % 8 == 0. */
int lk_split_images_f16x2(const float* x, int64_t N, int64_t per, void* planes_h, void* planes_l, int* sexp, unsigned* amax,
                          void* stream);

/* Patch matrix of a convolution (the rows F.unfold produces, laplace/curvature/curvlinops.py:55-75: the A factor of a Conv2d is the
 * Gram of its unfolded inputs) as split planes with ONE scale: x [B][H][W][C] fp32 NHWC -> planes [B * Ho * Wo][Kp], column
 * k = (kh * KW + kw) * C + c (the kernels' native order), zero outside the image and for k >= KH * KW * C (Kp: 64 or a multiple of 128,
 * what lk_gram_tn_f16x2 takes), scale from amax[0] = max|x|.  lk_gram_tn_f16x2 on the result is the A factor of a strided or stem
 * convolution on the fp16 matrix cores at fp32 level (2.6 x faster than lk_gram_conv_nhwc_f32's exact-fp32 MFMA on the c4 layers). */
int lk_im2col_split_f16x2(const float* x, int64_t B, int64_t H, int64_t W, int64_t C, int64_t KH, int64_t KW, int64_t stride,
                          int64_t pad, int64_t Ho, int64_t Wo, int64_t Kp, const float* amax, void* planes_h, void* planes_l,
                          int* sexp, void* stream);
/* W[Co][Ci][taps] (* cscale[co], e.g. a folded BatchNorm scale) -> planes[2][taps][N][K] fp16 + sexp;
 * transpose = 1 (backward-data): n = ci, k = co; 0 (forward): n = co, k = ci.  amax_ws: one device word. */
int lk_conv_prep_weights_f16x2(const float* W, int64_t Co, int64_t Ci, int64_t taps, int transpose, const float* cscale,
                               unsigned* amax_ws, void* planes, int* sexp, void* stream);
/* One launch of the implicit GEMM
 *     out[n][i*out_step + oh0][j*out_step + ow0][co] (+)= sum_t sum_c in[n][i*in_mul + dh_t][j*in_mul + dw_t][c] * Wt[wt_t][co][c]
 * for i < Hc, j < Wc (taps outside the [Hi][Wi] image contribute zero).  in: split tensor [N][Hi][Wi][Ci], Ci % 32 == 0,
 * with in_nsexp = 1 scale or in_nsexp = N scales (one per image: GEMM rows never mix images, the epilogue un-scales row by row);
 * w: split planes [.][Co][Ci]; taps: T x {dh, dw, wt} (T <= 9, host array); zero16: >= 16 zero bytes on the device;
 * out: fp32 [N][Ho][Wo][Co]; accumulate != 0 adds into out; amax_out (may be NULL): atomicMax of the bit patterns of
 * |out| (zero it first).  Stride-1 backward-data and forward convs are one launch, a stride-s backward-data is one
 * launch per output-pixel residue class.  config bit 0: 64-deep K chunks; bit 1: never use the form that keeps
 * the input patch of a tile resident in LDS across the taps (used when the output grid is the input grid);
 * bit 4: write the output position-contiguous, out[n][co][pixel] (dense grids, Ho*Wo % 4 == 0, no accumulate);
 * bits 12..14: tile shape of the generic form — 0: chosen from the shapes by occupancy (a pure function of the arguments),
 * 1: 64x64, 2: 128x64, 3: 64x128, 4: 128x128, 5: 256x64, 6: the big tile by output width only; the remaining bits select
 * development variants of the K pipeline (csrc/lk_conv.hip).  Every choice computes the same sums in the same order. */
int lk_conv_nhwc_f16x2(const void* in_h, const void* in_l, const int* in_sexp, int64_t in_nsexp, int64_t N, int64_t Hi,
                       int64_t Wi, int64_t Ci, const void* w_h, const void* w_l, const int* w_sexp, int64_t Co, int64_t Hc,
                       int64_t Wc, int64_t in_mul, int64_t Ho, int64_t Wo, int64_t out_step, int64_t oh0, int64_t ow0,
                       int64_t T, const int* taps, const void* zero16, float* out, int accumulate, unsigned* amax_out,
                       int config, void* stream);

/* lk_conv_nhwc_f16x2 with a dense, POSITION-contiguous output emitted as a split tensor: out_h / out_l [N][Co][Ho * Wo] fp16,
 * scaled per entry of in_sexp (the whole tensor, or image by image) from the guaranteed bound
 *     max|out_n| <= in_amax[n] * w_l1[0]        (in_amax: in_nsexp words, bit patterns, or NULL = 2^(15 - in_sexp[n]); w_l1:
 *                                                max_co sum_{t,c} |Wt[t][co][c]|, device word)
 * with the scale left in out_sexp[n].  (Ho * Wo) % 4 == 0; in_mul = the convolution's stride.  This is how the Kron predictive's
 * eigenbasis rotations (laplace/utils/matrix.py:406-456: Q1^T over the output cotangents as a 1x1 convolution, the unfolded
 * inputs times Q2 as a convolution whose filters are the eigenvectors) hand their results to
 * lk_kron_quadform_shared_planes_f16x2: no fp32 round trip, nothing left to split inside the quadratic-form kernel. */
int lk_conv_nhwc_f16x2_planes(const void* in_h, const void* in_l, const int* in_sexp, int64_t in_nsexp, const void* in_amax,
                              int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, const void* w_h, const void* w_l,
                              const int* w_sexp, const float* w_l1, int64_t Co, int64_t Ho, int64_t Wo, int64_t in_mul, int64_t T,
                              const int* taps, const void* zero16, void* out_h, void* out_l, int* out_sexp, int config,
                              void* stream);

/* lk_conv_nhwc_f16x2 (dense output grid; in_mul = the convolution's stride) with the forward's eval-mode BatchNorm, residual
 * add and ReLU in its epilogue — the model's forward pass inside GGNInterface / CurvlinopsInterface
 * (laplace/curvature/curvature.py:309-311 `self.model(x)`; torchvision-style conv -> bn -> (+ identity) -> relu blocks):
 *     y[n][i][j][c] = act(conv[n][i][j][c] * scale[c] + shift[c] + addend[n][i][j][c])        act: 0 none, 1 ReLU
 * To the bit what lk_conv_nhwc_f16x2 followed by lk_bn_act_fwd_nhwc_f16x2 (x_mul = w_l1, no x_add) computes — the same fp32
 * operations in the same order — in one launch and without the fp32 round trip of the convolution's output.  in_amax:
 * in_namax (1 or N) words with the measured max|in_n|; w_l1: max_co sum |W[co]| (device word).  Outputs: y fp32 NHWC, mask
 * (NHWC bytes y > 0, may be NULL), y_h / y_l (both or none: the split planes of y with one scale per image, y_sexp[N], from the
 * guaranteed bound y_bound[N] = in_amax[n] w_l1 max|scale| + max|shift| + addend_bound[n]), y_amax[N] (measured max|y_n| as bit
 * patterns; zeroed by the caller).  Co % 8 == 0; addend_bound: 1 or N floats. */
int lk_conv_bn_act_nhwc_f16x2(const void* in_h, const void* in_l, const int* in_sexp, int64_t in_nsexp, const void* in_amax,
                              int64_t in_namax, int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, const void* w_h, const void* w_l,
                              const int* w_sexp, const float* w_l1, int64_t Co, int64_t Ho, int64_t Wo, int64_t in_mul,
                              int64_t T, const int* taps, const void* zero16, const float* scale, const float* shift,
                              const void* scale_amax, const void* shift_amax, const float* addend, const float* addend_bound,
                              int64_t addend_nbound, int act, float* y, void* mask, void* y_h, void* y_l, int* y_sexp,
                              float* y_bound, void* y_amax, int config, void* stream);

/* lk_conv_nhwc_f16x2 with the element-wise VJP of the sweep fused into its epilogue (one dense launch: forward / stride-1
 * backward-data; Co % 8 == 0):
 *     o[n][i][j][c] = (conv[n][i][j][c] + add[n][i][j][c]) * M[(n,i,j) mod mask_rows][c] * scale[c]
 * emitted as a split tensor out_h / out_l / out_sexp — what lk_vjp_nhwc_split_f16x2 would make of the fp32 output, without
 * that tensor's round trip through HBM (the backward passes of laplace/curvature/curvlinops.py:87-100 through an
 * activation / folded BatchNorm / residual join).  The result's scale is derived before the launch from a guaranteed
 * bound: max|in| (in_amax: measured, device word, or NULL = 2^(15 - in_sexp)) * w_l1 (device word: max_n sum_{t,k}
 * |Wt[t][n][k]| of the prepared weights) + 2^(15 - add_sexp), times max|M| (mult_amax, fp32 multipliers only) and
 * max|scale|; out_amax (zeroed by the caller) receives the MEASURED max|o|, which the next launch takes as its in_amax.
 * add / mask / scale may be NULL.  mask: uint8 (mask_is_float = 0) or fp32 [mask_rows][Co]. */
int lk_conv_nhwc_f16x2_vjp(const void* in_h, const void* in_l, const int* in_sexp, const void* in_amax, int64_t N, int64_t Hi,
                           int64_t Wi, int64_t Ci, const void* w_h, const void* w_l, const int* w_sexp, const float* w_l1,
                           int64_t Co, int64_t Ho, int64_t Wo, int64_t T, const int* taps, const void* zero16,
                           const void* add_h, const void* add_l, const int* add_sexp, const void* mask, int mask_is_float,
                           const void* mult_amax, int64_t mask_rows, const float* scale, const void* scale_amax, void* out_h,
                           void* out_l, int* out_sexp, void* out_amax, int config, void* stream);
/* lk_conv_nhwc_f16x2_vjp with the weights ALSO handed over chunk-major (wc_h / wc_l: [tap][Ci / 16][Co][16] fp16 per plane,
 * same scale w_sexp): 3 x 3 / stride-1 launches with 64 output channels on maps of more than 64 pixels
 * (lk_conv_winp_eligible) then run the PERSISTENT window form — two workgroups per CU walk through the pixel tiles, the
 * input window of a tile resident in LDS, the next tile's operands requested under the epilogue; the chunk-major order
 * makes every weight staging instruction one contiguous kilobyte.  Other shapes ignore the extra planes. */
int lk_conv_winp_eligible(int64_t N, int64_t Hi, int64_t Wi, int64_t Ci, int64_t Co, int64_t T, int mask_is_float);
int lk_conv_nhwc_f16x2_vjp_wc(const void* in_h, const void* in_l, const int* in_sexp, const void* in_amax, int64_t N, int64_t Hi,
                              int64_t Wi, int64_t Ci, const void* w_h, const void* w_l, const int* w_sexp, const float* w_l1,
                              const void* wc_h, const void* wc_l, int64_t Co, int64_t Ho, int64_t Wo, int64_t T, const int* taps,
                              const void* zero16, const void* add_h, const void* add_l, const int* add_sexp, const void* mask,
                              int mask_is_float, const void* mult_amax, int64_t mask_rows, const float* scale,
                              const void* scale_amax, void* out_h, void* out_l, int* out_sexp, void* out_amax, int config,
                              void* stream);

/* The backward-data of a STRIDED convolution (stride `os` = 2; reference: the torch.func Jacobians of curvature.py:94-147
 * run it once per output class through autograd) with every residue class of the input-gradient pixels in ONE launch —
 * class (oh0, ow0): dX[i*os + oh0, j*os + ow0] = sum over that class's taps of g[i + dh, j + dw] W[slice] — optionally
 * together with a SECOND convolution that reads the same input (in2_* / w2_*, NULL without one: the 1 x 1 shortcut of a
 * residual down-sampling block, with its own cotangent and weights), and with the element-wise VJP fused as in
 * lk_conv_nhwc_f16x2_vjp: out = (dX + dX2 + add) * M * scale[channel] as a split tensor with its measured max|.|.
 * taps: T <= 12 rows {dh, dw, weight slice, source (0 / 1), oh0, ow0}; every one of the os*os classes needs a tap;
 * both cotangents are [N][Hi][Wi][Ci] with Hi = Ho / os, Wi = Wo / os; weights [slice][Co][Ci] per plane. */
int lk_conv_nhwc_f16x2_vjp_strided(const void* in_h, const void* in_l, const int* in_sexp, const void* in_amax, const void* w_h,
                                   const void* w_l, const int* w_sexp, const float* w_l1, const void* in2_h, const void* in2_l,
                                   const int* in2_sexp, const void* in2_amax, const void* w2_h, const void* w2_l,
                                   const int* w2_sexp, const float* w2_l1, int64_t N, int64_t Hi, int64_t Wi, int64_t Ci,
                                   int64_t Co, int64_t Ho, int64_t Wo, int64_t os, int64_t T, const int* taps,
                                   const void* zero16, const void* add_h, const void* add_l, const int* add_sexp,
                                   const void* mask, int mask_is_float, const void* mult_amax, int64_t mask_rows,
                                   const float* scale, const void* scale_amax, void* out_h, void* out_l, int* out_sexp,
                                   void* out_amax, int config, void* stream);


/* Element-wise VJP of the NHWC sweep with a split result (csrc/lk_sweep16.hip):
 *     out[s][e] = (g[s][e] + g2[s][e]) * M[e] * scale[e % C]      e < per = B*H*W*C,  s < S
 * g: fp32 addend with the bit pattern of max|g| in g_amax (both may be NULL); g2: split addend (may be NULL);
 * M: per-sample multiplier, uint8 mask or fp32 (m_amax: bit pattern of max|M| for fp32, NULL = 1), may be NULL;
 * scale: per-channel (scale_amax: bit pattern of max|scale|), may be NULL.  The output scale is derived on the device
 * from the guaranteed bound (max|g| + max|g2|) max|M| max|scale| — no pass over the data to find it. */
int lk_vjp_nhwc_split_f16x2(const float* g, const unsigned* g_amax, const void* g2_h, const void* g2_l, const int* g2_sexp,
                            const void* m, int m_is_float, const unsigned* m_amax, const float* scale,
                            const unsigned* scale_amax, int64_t C, int64_t S, int64_t per, void* out_h, void* out_l,
                            int* out_sexp, void* stream);
/* Forward of the NHWC sweep: y = act(x * scale[c] + shift[c] + addend) on fp32 NHWC tensors x [N][per], act 0 none /
 * 1 ReLU / 2 tanh (replaces BatchNorm2d-eval + add + activation, three library launches per layer), producing in the same
 * pass the ReLU mask as NHWC bytes and the split planes of y for the next convolution with ONE SCALE PER IMAGE, image n
 * scaled from the guaranteed bound
 *     y_bound[n] = bx_n max|scale| + max|shift| + addend_bound[n],   bx_n = x_amax[n] * x_mul[0] + x_add[0] >= max|x_n|   (tanh: 1)
 * x_amax: x_namax = 1 or N words (bit patterns; for a convolution's output: the measured maxima of its INPUT images, with
 * x_mul = the l1 norm of its weights and x_add = max|bias|; x_mul / x_add may be NULL = 1 / 0); addend_bound: 1 or N floats.
 * y_amax (N words, zeroed by the caller; may be NULL) receives the MEASURED max|y_n| — the next layer's x_amax, so the
 * slack of the bound does not compound.  scale_amax / shift_amax: device words with bit patterns of the maxima; addend,
 * mask, y_h / y_l may be NULL.  C % 8 == 0, per % C == 0, N <= 65535. */
int lk_bn_act_fwd_nhwc_f16x2(const float* x, const unsigned* x_amax, int64_t x_namax, const float* x_mul, const float* x_add,
                             const float* scale, const float* shift, const unsigned* scale_amax, const unsigned* shift_amax,
                             const float* addend, const float* addend_bound, int64_t addend_nbound, int act, int64_t C,
                             int64_t N, int64_t per, float* y, void* mask, void* y_h, void* y_l, int* y_sexp, float* y_bound,
                             unsigned* y_amax, void* stream);
/* y[0..n) = x[0..n) and *amax = max(*amax, max|x|) (bit pattern of a non-negative float; NOT reset: the word runs over the
 * minibatches stacked for one pixel-pair launch, which are then split with it instead of being measured in a pass of
 * their own).  16-byte aligned buffers, n % 4 == 0. */
int lk_copy_absmax_f32(const float* x, float* y, int64_t n, unsigned* amax, void* stream);

/* Row-major packed upper triangle of a symmetric n x n factor (what the ranks of a data-parallel fit exchange: half the
 * bytes of the square): packed[i n - i (i - 1) / 2 + (j - i)] = A[i][j], j >= i.  Unpacking writes the upper triangle only. */
int lk_pack_upper_f32(const float* A, int64_t n, float* packed, void* stream);
int lk_unpack_upper_f32(const float* packed, int64_t n, float* A, void* stream);

/* Eigenbasis algebra of KronDecomposed (laplace/utils/matrix.py:406-461: `_bmm` at exponents -1 / -1/2, i.e. the
 * materialised-Jacobian GLM predictive `inv_square_form` and the posterior samples of baselaplace.py:1845-1858).
 * lk_gemm_f32: batched  C[b] = alpha (op(A[b]) . op(B[b])) (.) E  (+ C[b] if accumulate) on the exact-fp32 MFMA; matrices
 * row-major with leading dimensions, batch strides (0 = shared operand), op = transpose if trans_x != 0 (A stored K x M,
 * B stored N x K), E optional M x N weight (lde = 0: one row broadcast).
 * lk_kron_pow_f32: lam[i][j] = (l1[i] l2[j] + delta)^e (damping: ((l1[i] + sqrt delta)(l2[j] + sqrt delta))^e;
 * l2 == NULL: (l1[i] + delta)^e); delta is a device scalar. */
int lk_gemm_f32(const float* A, const float* B, const float* E, float* C, int64_t batch, int64_t M, int64_t N, int64_t K,
                int64_t lda, int64_t ldb, int64_t ldc, int64_t lde, int64_t stride_a, int64_t stride_b, int64_t stride_c,
                int trans_a, int trans_b, float alpha, int accumulate, void* stream);
int lk_kron_pow_f32(const float* l1, int64_t n1, const float* l2, int64_t n2, const float* delta, float exponent,
                    int damping, float* lam, void* stream);
/* Split NHWC cotangent x [S*B][L][C] (seed-major batch) -> fp32 out[b][s][c][l]: position-contiguous and sample-major,
 * the `u` operand of lk_kron_quadform_shared_f32 / lk_diag_quadform_shared_f32. */
int lk_unsplit_transpose_f32(const void* x_h, const void* x_l, const int* sexp, int64_t S, int64_t B, int64_t L, int64_t C,
                             float* out, void* stream);
/* G[C][C] += alpha * X^T X for a split tensor X [R][C] (rows = (seed, sample, position) of an NHWC cotangent): the
 * G factor of a convolution layer (curvlinops.py:87-100).  Only the 32x32 tiles on or above the diagonal are written
 * (lk_symmetrize_f32 mirrors).  C = 64 or a multiple of 128.  Deterministic (workspace partials, fixed-order sum). */
size_t lk_gram_tn_f16x2_workspace_bytes(int64_t R, int64_t C);
int lk_gram_tn_f16x2(const void* x_h, const void* x_l, const int* sexp, int64_t R, int64_t C, float alpha, float* G,
                     const void* zero16, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Diagonal GGN / EF.  Replaces GGNInterface.diag / EFInterface.diag (curvature.py:413-433,494-505)
 * for nn.Linear layers:  h_w[o][i] += alpha * sum_n (sum_c g[c][n][o]^2) a[n][i]^2,
 *                        h_b[o]    += alpha * sum_n  sum_c g[c][n][o]^2          (h_b may be NULL)
 * a: [B][Di], g: [Cc][B][Do].
 * ------------------------------------------------------------------------------------------- */
int lk_diag_ggn_linear_f32(const float* a, const float* g, int64_t B, int64_t Cc, int64_t Di, int64_t Do,
                           float alpha, float* h_w, float* h_b, void* stream);

/* Per-sample weight Jacobians of one layer written into Js[B][C][P] at column `col0`
 * (replaces the jacrev materialisation of CurvatureInterface.jacobians, curvature.py:88-129):
 *   linear: Js[n][c][col0 + o*Di + i] = g[c][n][o] * a[n][i];  bias: Js[n][c][bcol0 + o] = g[c][n][o]
 *   conv  : Js[n][c][col0 + o*Dk + k] = sum_l g[c][n][o][l] * patches[n][l][k]   (patches NHWC-native
 *           order is converted to unfold order on the fly) */
int lk_jac_linear_f32(const float* a, const float* g, int64_t B, int64_t Cc, int64_t Di, int64_t Do,
                      float* Js, int64_t P, int64_t col0, int64_t bcol0, void* stream);
int lk_jac_conv_f32(const float* x_nchw, const float* g, int64_t B, int64_t Cc, int64_t Cin, int64_t H, int64_t W,
                    int64_t Do, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                    float* Js, int64_t P, int64_t col0, int64_t bcol0, void* stream);

/* h[p] += alpha * sum_r Js[r][col0 + p]^2 for p < width (rows r = (sample, class)); the conv-layer
 * diagonal GGN / EF is the squared per-sample weight Jacobian summed over samples. */
int lk_sq_colsum_f32(const float* Js, int64_t rows, int64_t P, int64_t col0, int64_t width, float alpha,
                     float* h, void* stream);

/* Element-wise VJP of the seed-batched reverse sweep: all S seeds of an `activation(BatchNorm_eval(.))` backward
 * in one pass (the reference gets these from stock autograd, one backward kernel per seed and layer —
 * laplace/curvature/curvlinops.py:87-106 through curvlinops' hooks):
 *   out[s][e] = (g[s][e] + g2[s][e]) * M[e] * scale[(e / HW) % C],   e < per_sample = B*C*HW
 * g2: second branch of a residual connection, summed on the fly (NULL = none; may not alias out).
 * M: per-sample multiplier (NULL = 1): bytes that are zero / non-zero (a ReLU mask, m_is_float = 0) or floats
 * (f'(y), m_is_float = 1).  scale: per-channel factor gamma/sqrt(var+eps) (NULL = 1; then C, HW are ignored). */
/* Forward counterpart in the sweep's own interpretation of the model: eval-mode BatchNorm (a per-channel affine map)
 * with an optional ReLU and the mask its VJP needs, in one pass:
 *   y[e] = act(x[e] * scale[c(e)] + shift[c(e)] + addend[e]),  mask[e] = y[e] > 0
 *   (addend: the other branch of a residual connection, NULL = none; mask may be NULL; relu = 0: no activation),
 *   scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale, e < total = B*C*HW. */
int lk_bn_act_fwd_f32(const float* x, const float* scale, const float* shift, const float* addend, int64_t total,
                      int64_t C, int64_t HW, int relu, float* y, unsigned char* mask, void* stream);
int lk_vjp_scale_mask_f32(const float* g, const float* g2, const void* m, int m_is_float, const float* scale, int64_t S,
                          int64_t per_sample, int64_t C, int64_t HW, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense last-layer GGN.  Replaces last_layer_jacobians + GGNInterface.full for a Linear head
 * (curvature.py:131-167,375-411) without materialising Js, using J_n = I_C (x) [phi_n, 1]:
 *   H[(j,a),(k,b)] += alpha * sum_n (delta_jk p_nj - p_nj p_nk) pt_na pt_nb      pt = [phi, 1]
 *                   = alpha * ( blockdiag_j Gram(sqrt(p_j).Pt) - Gram(Y) ),  Y[n][(j,a)] = p_nj pt_na
 * phi: [B][D]; probs: [B][C] softmax probabilities (NULL = regression, Lambda = I);
 * H: [P][P] in the reference's parameter order (weight [C][D] row-major, then bias [C]),
 * P = C*D (+C if has_bias).
 * ------------------------------------------------------------------------------------------- */
size_t lk_ll_ggn_workspace_bytes(int64_t B, int64_t C, int64_t D);
int lk_ll_ggn_full_f32(const float* phi, const float* probs, int64_t B, int64_t C, int64_t D, int has_bias,
                       float alpha, float* H, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Symmetric eigendecomposition (two-sided block-Jacobi, exact-fp32 MFMA tile updates).
 * Replaces utils.symeig -> torch.linalg.eigh(M, UPLO="U") as used by Kron.decompose
 * (laplace/utils/utils.py:193-228, laplace/utils/matrix.py:123-150):
 *   reads the UPPER triangle of A[n][n]; writes ascending eigenvalues w[n] clamped at >= 0 (when
 *   clamp != 0) and eigenvectors as the COLUMNS of Q[n][n] (row-major), NaNs zeroed.
 *   info (device int32[2]): info[0] = 0 converged / 1 sweeps ran out; info[1] = sweeps executed.
 * A is not modified.  Fully asynchronous on `stream`.
 * ------------------------------------------------------------------------------------------- */
size_t lk_syevj_workspace_bytes(int64_t n);
int lk_syevj_f32(const float* A, int64_t n, float* w, float* Q, int clamp, int max_sweeps, int32_t* info,
                 void* ws, size_t ws_bytes, void* stream);

/* All factors of one Kron.decompose (laplace/utils/matrix.py:123-150 loops over them one by one) in one call.
 * Matrix i (arguments as lk_syevj_f32, given as HOST arrays of `count` entries; ws[i] >=
 * lk_syevj_workspace_bytes(n[i]), private to matrix i) runs on streams[i % nstreams], matrices sharing a stream in
 * index order (pass the largest first).  A host-side scheduler keeps every stream two sweeps ahead of the device,
 * reads each solve's `converged` flag back asynchronously and finalises a matrix as soon as it has converged, so
 * no stream starves behind another one's launch queue and no empty sweeps are enqueued.
 * EXCEPTION to the no-host-synchronisation rule: the call returns when the last sweep of every matrix has
 * executed (the refinement / sort / gather of the last matrices may still be in flight on their streams). */
int lk_syevj_batched_f32(int64_t count, const float* const* A, const int64_t* n, float* const* w, float* const* Q,
                         int32_t* const* info, void* const* ws, const size_t* ws_bytes, int clamp, int max_sweeps,
                         void* const* streams, int64_t nstreams);

/* ---------------------------------------------------------------------------------------------
 * KronDecomposed.logdet (laplace/utils/matrix.py:381-404) for one two-factor block, with the
 * derivatives autograd needs for the marginal-likelihood sweep (baselaplace.py:466-485):
 *   out[0] += sum_ij log(l1_i l2_j + delta);  d_delta[0] += sum_ij 1/(l1_i l2_j + delta)
 *   d_l1[i] += sum_j l2_j/(.) ;  d_l2[j] += sum_i l1_i/(.)      (d_* may be NULL)
 * n2 == 0 means a single-factor block: sum_i log(l1_i + delta).
 * damping != 0 uses (l1+sqrt(delta)) (x) (l2+sqrt(delta)) (matrix.py:397-399; no derivatives).
 * delta is read from DEVICE memory (delta[0]).
 * ------------------------------------------------------------------------------------------- */
size_t lk_kron_logdet_workspace_bytes(int64_t n1);
int lk_kron_logdet_f32(const float* l1, int64_t n1, const float* l2, int64_t n2, const float* delta,
                       int damping, float* out, float* d_l1, float* d_l2, float* d_delta, void* ws,
                       size_t ws_bytes, void* stream);

/* The whole posterior precision in one pass: every block of a KronDecomposed (matrix.py:381-404 loops over the blocks;
 * the marginal-likelihood sweep, baselaplace.py:466-485 and marglik_training.py:303-314, calls it once per step):
 *   out[0]      += sum_b sum_ij log(s * l1[b]_i l2[b]_j + delta[b])      (n2[b] == 0: sum_i log(s * l1[b]_i + delta[b]))
 *   d_delta[b]  += sum_ij 1 / (.)                                         (may be NULL)
 *   d_scale[0]  += sum_b sum_ij l1[b]_i l2[b]_j / (.)                     (may be NULL)
 * s = scale[0] is the scalar H_factor = 1/(sigma^2 T) of `H * H_factor + delta` (baselaplace.py:1820), applied to the
 * eigenvalue PRODUCT (matrix.py:366-376 splits it as s^(1/2) per factor); scale == NULL means 1.
 * l1, n1, l2, n2 are HOST arrays of nblocks entries (device pointers / lengths); delta, scale, out, d_* are DEVICE
 * memory (delta has nblocks entries).  Three launches per 32 blocks, fixed-order fp64 reduction: deterministic. */
size_t lk_kron_logdet_blocks_workspace_bytes(int64_t total_rows /* sum_b n1[b] */, int64_t nblocks);
int lk_kron_logdet_blocks_f32(int64_t nblocks, const float* const* l1, const int64_t* n1, const float* const* l2,
                              const int64_t* n2, const float* delta, const float* scale, float* out, float* d_delta,
                              float* d_scale, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * GLM predictive variances  f_var[n] = J_n Sigma J_n^T  without materialising J (V1-V3).
 * ------------------------------------------------------------------------------------------- */
/* Kron posterior, one nn.Linear layer (KronDecomposed.inv_square_form, matrix.py:406-461, called from
 * KronLaplace.functional_variance, baselaplace.py:1834-1835).  u = g Q1 [Cc][B][Do], v = a Q2 [B][Di]
 * are the eigenbasis projections (plain GEMMs done by the caller);
 *   fvar[n][c][k] += sum_o u[c][n][o] u[k][n][o] * ( sum_i v[n][i]^2 / (l1_o l2_i + delta) )
 *                  + (bias block, if lb != NULL: ub = g Qb, sum_o ub[c][n][o] ub[k][n][o]/(lb_o + delta_b)) */
int lk_kron_quadform_linear_f32(const float* u, const float* v, const float* l1, const float* l2,
                                const float* delta, int64_t B, int64_t Cc, int64_t Do, int64_t Di,
                                const float* ub, const float* lb, const float* delta_b,
                                float* fvar, void* stream);

/* Diagonal posterior, one nn.Linear layer (DiagLaplace.functional_variance, baselaplace.py:2113-2115):
 *   fvar[n][c][k] += sum_{o,i} g[c][n][o] g[k][n][o] a[n][i]^2 var_w[o][i] + sum_o g[c][n][o] g[k][n][o] var_b[o] */
int lk_diag_quadform_linear_f32(const float* a, const float* g, const float* var_w, const float* var_b,
                                int64_t B, int64_t Cc, int64_t Do, int64_t Di, float* fvar, void* stream);

/* Weight-sharing layers (nn.Conv2d; nn.Linear applied along a sequence): the per-sample Jacobian of output c is
 * J_c = sum_l u[n][c][:][l] v[n][:][l]^T  (Do x Dk, L shared positions), and KronLaplace / DiagLaplace
 * .functional_variance (baselaplace.py:1834-1835 via matrix.py:406-461; baselaplace.py:2113-2115) contract the
 * materialised [B, C, Do*Dk] block.  These two never form it:
 *   fvar[n][c][k] += sum_{o,i} J_c[o,i] J_k[o,i] w[o,i]
 *   Kronecker posterior: u = Q1^T (grad w.r.t. the layer output), v = Q2^T (unfolded input), w = 1/(l1[o] l2[i] + delta[0])
 *   diagonal posterior:  u, v raw,                                                          w = var_w[o][i]
 * u [B][C][Do][L], v [B][Dk][L] (both position-contiguous: the NCHW layout of a convolution's output gradient and of
 * F.unfold), C <= 10 (LK_EINVAL beyond: use the generic Jacobian form).  Per sample one MFMA GEMM
 * [(C*Do) x L].[L x Dk] whose 32x32 tiles stay in accumulators for all C outputs and are folded into the C(C+1)/2 pair
 * sums in registers; workgroup partials are reduced in fixed order.  With L % 4 == 0 and 16-byte aligned u, v the
 * product runs on the bf16 matrix cores at fp32 accuracy (operands split into three bf16 pieces, six
 * v_mfma_f32_32x32x16_bf16 per fp32 product; dropped terms <= 3 * 2^-24), otherwise on v_mfma_f32_32x32x2_f32.
 * Requires C*L*Do < 2^29 and L*Dk < 2^29. */
size_t lk_quadform_shared_workspace_bytes(int64_t B, int64_t C, int64_t Do, int64_t Dk);
int lk_kron_quadform_shared_f32(const float* u, const float* v, const float* l1, const float* l2, const float* delta,
                                int64_t B, int64_t C, int64_t Do, int64_t Dk, int64_t L, float* fvar, void* ws,
                                size_t ws_bytes, void* stream);
/* lk_kron_quadform_shared_f32 for u stored seed-major, [C][B][Do][L]: what a seed-batched reverse sweep (and the rotation
 * convolution over its cotangent) leaves in memory — no transposed copy. */
int lk_kron_quadform_shared_seedmajor_f32(const float* u, const float* v, const float* l1, const float* l2, const float* delta,
                                          int64_t B, int64_t C, int64_t Do, int64_t Dk, int64_t L, float* fvar, void* ws,
                                          size_t ws_bytes, void* stream);
/* The same quadratic form on operands that ARRIVE split (round 5): u_h / u_l [C][B][Do][L] fp16 planes with the scale
 * u_sexp[0] (seed-major), v_h / v_l [B][Dk][L] with one scale per sample (v_nsexp = B) or one for the tensor (v_nsexp = 1),
 * as lk_conv_nhwc_f16x2_planes leaves them.  A chunk of 16 positions is then 8-byte copies into LDS, two 16-byte loads and
 * C x three v_mfma_f32_32x32x16_f16 — the fp32-operand forms above spend as many vector-pipe cycles splitting in flight as
 * their matrix pipe spends on the products; the operands are staged by LDS-DMA into a ring of four stages, requested three
 * chunks ahead (one wave per SIMD: nothing else hides a load).  L % 16 == 0, Do % 32 == 0, C <= 10; zero16: >= 16 zero bytes
 * on the device; same workspace. */
int lk_kron_quadform_shared_planes_f16x2(const void* u_h, const void* u_l, const int* u_sexp, const void* v_h, const void* v_l,
                                         const int* v_sexp, int64_t v_nsexp, const float* l1, const float* l2,
                                         const float* delta, int64_t B, int64_t C, int64_t Do, int64_t Dk, int64_t L,
                                         const void* zero16, float* fvar, void* ws, size_t ws_bytes, void* stream);
int lk_diag_quadform_shared_f32(const float* u, const float* v, const float* var_w, int64_t B, int64_t C, int64_t Do,
                                int64_t Dk, int64_t L, float* fvar, void* ws, size_t ws_bytes, void* stream);

/* Exact GGN / Fisher diagonal of a weight-sharing layer (GGNInterface.diag, laplace/curvature/curvature.py:413-433,
 * restricted to the layer's weight): h[o][i] += alpha * sum_{n,s} (sum_l u[n][s][o][l] v[n][i][l])^2 with
 * u [B][S][Do][L] the seed cotangents at the layer output and v [B][Dk][L] the unfolded input, S <= 10 seeds per call.
 * Same tile GEMM as lk_*_quadform_shared_f32; the squares are summed over (sample, seed) in registers, so the
 * [B, S, Do*Dk] per-sample Jacobian the as-written einsum contracts is never formed. */
size_t lk_diag_ggn_shared_workspace_bytes(int64_t B, int64_t Do, int64_t Dk);
int lk_diag_ggn_shared_f32(const float* u, const float* v, int64_t B, int64_t S, int64_t Do, int64_t Dk, int64_t L,
                           float alpha, float* h, void* ws, size_t ws_bytes, void* stream);

/* Generic streaming form over a materialised Jacobian (any layer type):
 *   fvar[n][c][k] = sum_p Js[n][c][p] var[p] Js[n][k][p] */
int lk_diag_quadform_js_f32(const float* Js, const float* var, int64_t B, int64_t C, int64_t P, float* fvar,
                            void* stream);

/* Last-layer Jacobians (CurvatureInterface.last_layer_jacobians, laplace/curvature/curvature.py:131-167, for a Linear head):
 *   Js[n][c][:] = e_c (x) [phi_n, 1]  — [B][C][P], P = C*D (+ C with a bias), parameter order weight [C][D] row-major then bias.
 * The fused consumers (lk_ll_ggn_full_f32, lk_dense_quadform_ll_f32) never form it; this is for callers of the drop-in seam that
 * ask for the Jacobian itself (la(x) of the reference's last-layer flavours). */
int lk_jac_last_layer_f32(const float* phi, int64_t B, int64_t C, int64_t D, int has_bias, float* Js, void* stream);

/* Dense last-layer posterior (FullLaplace.functional_variance with J = I (x) [phi,1],
 * baselaplace.py:1683-1684, lllaplace.py:212-237):
 *   fvar[n][c][k] = phit_n^T Sigma[(c,:),(k,:)] phit_n,   Sigma: [P][P] in the reference's parameter
 *   order (weight [C][D] row-major, then bias [C]). */
size_t lk_dense_quadform_ll_workspace_bytes(int64_t B, int64_t C, int64_t D);
int lk_dense_quadform_ll_f32(const float* phi, const float* Sigma, int64_t B, int64_t C, int64_t D, int has_bias,
                             float* fvar, void* ws, size_t ws_bytes, void* stream);

/* ---- the fit's one collective (replaces nothing in the reference, which has no multi-GPU fit: the loop of
 * laplace/baselaplace.py:969-985 sharded over ranks needs ONE sum of the accumulated factors at epoch end) -------------------
 * Thin wrappers over RCCL (ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce(ncclFloat, ncclSum, in place)),
 * bound by dlopen at the first call, for hosts that are not PyTorch; the Python host of this repository uses
 * torch.distributed's "nccl" backend (the same RCCL) and the packed upper-triangle buffer of `KronAccumulator.tensors()`.
 * `id128`: 128 bytes (ncclUniqueId) made on rank 0 and handed to the other ranks by the host's own means. */
int lk_comm_unique_id(void* id128);
int lk_comm_init_rank(void** comm, int nranks, const void* id128, int rank);
int lk_comm_destroy(void* comm);
int lk_allreduce_sum_f32(void* comm, float* buf, int64_t count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LAPLACE_HIP_H */
