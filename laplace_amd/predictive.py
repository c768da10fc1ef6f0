"""Structure-exploiting GLM predictive variance ``f_var[n] = J_n Sigma J_n^T``.

The reference materialises ``Js[B, C, P]`` (laplace/baselaplace.py:1306-1342) and contracts it with
the posterior (Kron: laplace/utils/matrix.py:406-461; diag: baselaplace.py:2113-2115; full:
:1683-1684).  Here the per-layer factors of the Jacobian — layer inputs ``a`` and output gradients
``g`` from one batched reverse pass — go straight into HIP kernels:

* nn.Linear, Kron posterior:  ``J_nc (Q1 (x) Q2) = (Q1^T g_nc) (x) (Q2^T a_n)``, so
  ``f_var[n,c,k] = sum_o u_nco u_nko * sum_i v_ni^2 / (l1_o l2_i + delta)`` — two small GEMMs
  (plumbing) + ``lk_kron_quadform_linear_f32``.
* nn.Linear, diagonal posterior: ``lk_diag_quadform_linear_f32``.
* dense last-layer posterior: ``lk_dense_quadform_ll_f32`` (``J = I (x) [phi, 1]``).
* nn.Conv2d and nn.Linear along a sequence (weight sharing over ``L`` positions): ``J_nc = sum_l g_ncl a_nl^T``,
  so ``Q1^T J_nc Q2 = (g Q1)^T (a Q2)`` — the input rotation is ONE convolution with the eigenvectors as filters
  (library conv), the output rotation a small GEMM, and ``lk_kron_quadform_shared_f32`` /
  ``lk_diag_quadform_shared_f32`` contract the per-sample ``[(C Do) x L] . [L x Dk]`` product with the posterior
  inside MFMA accumulators: no ``Do x Dk`` block per (sample, output) is ever written.  As written
  (matrix.py:406-461) the Kronecker form costs ``4 C (Do^2 Dk + Do Dk^2)`` flop per sample and layer — 484 GFLOP
  for one 512x4608 ResNet-18 layer; here ``2 L (Dk^2 + C Do^2 + C Do Dk)`` = 1.5 GFLOP.
* more than 10 outputs (the kernel's accumulator budget): pairs of 5-output blocks of the same kernel where that is
  cheaper than the as-written route — the layer's Jacobian block assembled by ``lk_jac_conv_f32`` (only that block,
  never the full ``[B, C, P]``) and contracted with the block's posterior — which serves the rest.
"""
from __future__ import annotations

import torch

from laplace_amd._lib import get_kernels
from laplace_amd.backend import shared_operands as _shared_operands


def _identity_seeds(f: torch.Tensor) -> torch.Tensor:
    B, C = f.shape
    eye = torch.eye(C, dtype=f.dtype, device=f.device)
    return eye[:, None, :].expand(C, B, C).contiguous()


def _grads(grad_fn, seeds):
    """output gradients of all taps; the NHWC sweep hands conv taps over as split tensors (no layout conversion)"""
    if getattr(grad_fn, "accepts_keep_split", False):
        return grad_fn(seeds, keep_split=True)
    return grad_fn(seeds)  # autograd-tape / last-layer grad_fn


def _fp32_only(backend):
    """The fused predictives run the fp32 kernels on the caller's own posterior tensors.  A model in another floating
    dtype (served by an fp32 twin elsewhere, backend._twin) takes the reference's route instead — Jacobians from
    ``backend.jacobians`` (which does use the twin) through the posterior in the model's dtype: NotImplementedError is
    what the callers (laplace_amd/laplace.py, dropin.py) fall back on."""
    twin = backend._twin()[0] if hasattr(backend, "_twin") else None
    if twin is not None:
        raise NotImplementedError("fused predictive computes in float32; this model is not float32")


def _as_nchw(g, B, C):
    """split NHWC cotangent -> [C, B, Do, H, W] fp32 for the routes that still read plain tensors"""
    from laplace_amd._lib import SplitTensor

    if isinstance(g, SplitTensor):
        return g.float().reshape(C, B, *g.shape[1:]).permute(0, 1, 4, 2, 3).contiguous()
    return g


def _conv_block_jacobian(tap, g, B, C):
    g = _as_nchw(g, B, C)
    K = get_kernels()
    m = tap.module
    width = m.weight.numel()
    nb = m.out_channels if tap.has_bias else 0
    Jl = torch.zeros(B, C, width + nb, dtype=torch.float32, device=g.device)
    K.jac_conv(tap.a.to(torch.float32).contiguous(), g.contiguous(), m.kernel_size, m.stride, m.padding, m.dilation,
               Jl, 0, width if tap.has_bias else -1)
    return Jl, width


def _shared_quadform(K, call, u, v, fvar, weight_sharing_only: bool):
    """``fvar += `` the weight-sharing quadratic form for ANY number of outputs C.  The kernel holds at most
    ``K.quadform_shared_max_outputs`` (10) outputs in accumulators, so more of them are covered by pairs of 5-output
    blocks: the launch on blocks (I, J) yields the I x I, J x J and I x J parts of ``f_var`` (the diagonal parts are
    taken from the first launch that produces them).  That recomputes each output's tile product ``ceil(C/5) - 1``
    times -- against the as-written block route's ``4 C (Do^2 Dk + Do Dk^2)`` flop still the cheaper way for layers
    with few positions; returns False (nothing done) where it is not, unless the layer has no other route."""
    B, C, Do, L = u.shape
    Dk = v.shape[1]
    cap = K.quadform_shared_max_outputs
    if C <= cap:
        call(u, v, fvar)
        return True
    h = max(cap // 2, 1)
    blocks = [list(range(i, min(i + h, C))) for i in range(0, C, h)]
    n_launch = len(blocks) * (len(blocks) - 1) // 2
    if not weight_sharing_only and n_launch * 2.0 * cap * Do * L * Dk > 4.0 * C * (Do * Do * Dk + Do * Dk * Dk):
        return False
    seen = set()
    for ia in range(len(blocks)):
        for ib in range(ia + 1, len(blocks)):
            A_, B_ = blocks[ia], blocks[ib]
            tmp = torch.zeros(B, len(A_) + len(B_), len(A_) + len(B_), dtype=torch.float32, device=u.device)
            call(u[:, A_ + B_].contiguous(), v, tmp)
            na = len(A_)
            fvar[:, A_[0]:A_[-1] + 1, B_[0]:B_[-1] + 1] += tmp[:, :na, na:]
            fvar[:, B_[0]:B_[-1] + 1, A_[0]:A_[-1] + 1] += tmp[:, na:, :na]
            if ia not in seen:
                fvar[:, A_[0]:A_[-1] + 1, A_[0]:A_[-1] + 1] += tmp[:, :na, :na]
                seen.add(ia)
            if ib not in seen:
                fvar[:, B_[0]:B_[-1] + 1, B_[0]:B_[-1] + 1] += tmp[:, na:, na:]
                seen.add(ib)
    return True


def glm_variance_kron(backend, x, post):
    """``(f_mu, f_var)`` under a :class:`HipKronDecomposed` posterior precision ``post``
    (= ``H * H_factor + prior_precision``), i.e. KronLaplace.functional_variance."""
    _fp32_only(backend)
    K = get_kernels()
    f, tape, grad_fn = backend._forward(x, keep_tap_splits=True)  # the eigenbasis rotations re-use the split inputs
    if tape.uncovered:
        raise NotImplementedError("fused Kron predictive needs Linear/Conv2d-only models")
    B, C = f.shape
    grads = _grads(grad_fn, _identity_seeds(f))
    fvar = torch.zeros(B, C, C, dtype=torch.float32, device=f.device)
    blk = 0
    for tap, g in zip(tape.taps, grads):
        if len(post.eigenvalues[blk]) != 2:
            # a 1x1 bias-free layer is stored as ONE merged block [G*A] (curvlinops.py:55-75): no kernel for it here
            tape.release()
            raise NotImplementedError("merged single-factor weight block: use the Jacobian route")
        (Q1, Q2), (l1, l2), delta = post.eigenvectors[blk], post.eigenvalues[blk], post.deltas[blk]
        if post.damping:
            # laplace/utils/matrix.py:397-399, 441-444: the damped block is (Q1 (l1 + sqrt d) Q1^T) x (Q2 (l2 + sqrt d) Q2^T) —
            # eigenvalues outer(l1 + sqrt d, l2 + sqrt d) instead of outer(l1, l2) + d: the same kernels with shifted
            # eigenvalues and no additive term
            sq = torch.sqrt(delta.detach().to(l1.dtype))
            l1, l2, delta = l1 + sq, l2 + sq, torch.zeros_like(delta)
        blk += 1
        Qb = lb = delta_b = None
        if tap.has_bias:
            Qb, lb, delta_b = post.eigenvectors[blk][0], post.eigenvalues[blk][0], post.deltas[blk]
            blk += 1
        d1 = delta.detach().reshape(1).contiguous()
        if tap.kind == "linear" and tap.a.ndim == 2:
            a = tap.a.to(torch.float32)
            Do = g.shape[-1]
            g2 = g.reshape(C * B, Do)
            u = (g2 @ Q1).reshape(C, B, Do).contiguous()
            v = (a @ Q2).contiguous()
            ub = (g2 @ Qb).reshape(C, B, Do).contiguous() if Qb is not None else None
            K.kron_quadform_linear(u, v, l1.contiguous(), l2.contiguous(), d1, fvar, ub,
                                   None if lb is None else lb.contiguous(),
                                   None if delta_b is None else delta_b.detach().reshape(1).contiguous())
        else:
            bnd = {}
            u, v, gsum = _shared_operands(tap, g, B, C, Q1, Q2, bounds=bnd)
            l1c, l2c = l1.contiguous(), l2.contiguous()
            if bnd.get("planes"):  # both operands arrive as split planes from the rotation convolutions: one launch
                K.kron_quadform_shared_planes(u, v, l1c, l2c, d1, fvar, C)
                done = True
            elif bnd.get("u_seed_major"):  # u is [C, B, Do, L] (C within the kernel's accumulators): one launch
                K.kron_quadform_shared(u, v, l1c, l2c, d1, fvar, seed_major=True)
                done = True
            else:
                done = _shared_quadform(K, lambda uu, vv, out: K.kron_quadform_shared(uu, vv, l1c, l2c, d1, out), u, v,
                                        fvar, weight_sharing_only=tap.kind != "conv2d")
            if done:
                if Qb is not None:
                    ub = gsum @ Qb
                    fvar += torch.einsum("cno,kno,o->nck", ub, ub, 1.0 / (lb + delta_b))
            else:  # many outputs and many positions: the layer's Jacobian block, rotated as written
                del u, v
                Jl, width = _conv_block_jacobian(tap, g, B, C)
                Do, Dk = len(l1), len(l2)
                W = Jl[:, :, :width].reshape(B * C, Do, Dk)
                M = Q1.T @ W @ Q2
                fvar += torch.einsum("ncoi,nkoi,oi->nck", M.reshape(B, C, Do, Dk), M.reshape(B, C, Do, Dk),
                                     1.0 / (torch.outer(l1, l2) + delta))
                if Qb is not None:
                    ub = Jl[:, :, width:] @ Qb
                    fvar += torch.einsum("nco,nko,o->nck", ub, ub, 1.0 / (lb + delta_b))
    tape.release()
    return f, fvar


def glm_variance_diag(backend, x, post_var: torch.Tensor):
    """``(f_mu, f_var)`` under a diagonal posterior with variances ``post_var[P]``."""
    _fp32_only(backend)
    K = get_kernels()
    f, tape, grad_fn = backend._forward(x)
    if tape.uncovered:
        raise NotImplementedError("fused diagonal predictive needs Linear/Conv2d-only models")
    B, C = f.shape
    grads = grad_fn(_identity_seeds(f))
    post_var = post_var.detach().to(torch.float32).contiguous()
    fvar = torch.zeros(B, C, C, dtype=torch.float32, device=f.device)
    for tap, g in zip(tape.taps, grads):
        m = tap.module
        n_w = m.weight.numel()
        vw = post_var[tap.w_off:tap.w_off + n_w]
        if tap.kind == "linear" and tap.a.ndim == 2:
            vb = post_var[tap.b_off:tap.b_off + m.out_features] if tap.has_bias else None
            K.diag_quadform_linear(tap.a.to(torch.float32).contiguous(), g.contiguous(), vw, vb, fvar)
        elif C <= K.quadform_shared_max_outputs or tap.kind != "conv2d":
            u, v, gsum = _shared_operands(tap, g, B, C)
            vw2 = vw.reshape(u.shape[2], v.shape[1]).contiguous()
            _shared_quadform(K, lambda uu, vv, out: K.diag_quadform_shared(uu, vv, vw2, out), u, v, fvar,
                             weight_sharing_only=True)
            if tap.has_bias:
                vb = post_var[tap.b_off:tap.b_off + u.shape[2]]
                fvar += torch.einsum("cno,kno,o->nck", gsum, gsum, vb)
        else:  # many outputs: assembling the block Jacobian once is cheaper than re-forming tile products per pair
            Jl, width = _conv_block_jacobian(tap, g, B, C)
            var = torch.cat([vw, post_var[tap.b_off:tap.b_off + m.out_channels]]) if tap.has_bias else vw
            fvar += K.diag_quadform_js(Jl, var.contiguous())
    tape.release()
    return f, fvar


def glm_variance_full_last_layer(backend, x, Sigma: torch.Tensor):
    """``(f_mu, f_var)`` for a last-layer Laplace with dense posterior covariance ``Sigma[P, P]``."""
    _fp32_only(backend)
    K = get_kernels()
    if not backend.last_layer:
        raise NotImplementedError("dense fused predictive is implemented for last-layer Laplace")
    f, tape, _ = backend._forward(x)
    tap = tape.taps[0]
    phi = tap.a.to(torch.float32).contiguous()
    fvar = K.dense_quadform_ll(phi, Sigma.detach().to(torch.float32).contiguous(), f.shape[1], tap.has_bias)
    tape.release()
    return f, fvar
