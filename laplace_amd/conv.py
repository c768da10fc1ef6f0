"""Host side of the split-fp16 convolution kernels (csrc/lk_conv.hip): tap tables and launch plans for the
backward-data and forward forms of an ``nn.Conv2d`` on NHWC tensors.

The reverse passes of curvlinops' KFAC (laplace/curvature/curvlinops.py:87-100) are, for a convolution layer,
backward-data convolutions of the output cotangent; the seed-batched sweep (laplace_amd/sweep.py) runs them for all
seeds at once through :func:`conv_backward_data`.
"""
from __future__ import annotations

import functools

import torch
from torch import nn

from laplace_amd._lib import SplitTensor, get_kernels


def supported(m: nn.Conv2d) -> bool:
    """what the implicit-GEMM kernel covers: dense (groups = 1), undilated, zero padding, at most 9 taps, both channel
    counts multiples of 32 (a lane of the 16-bit MFMA reads 8 consecutive channels; K chunks are 32 deep)"""
    return _geometry_ok(m) and m.in_channels % 32 == 0 and m.out_channels % 32 == 0


def _geometry_ok(m) -> bool:
    return (isinstance(m, nn.Conv2d) and m.groups == 1 and tuple(m.dilation) == (1, 1) and not isinstance(m.padding, str)
            and m.padding_mode == "zeros" and m.kernel_size[0] * m.kernel_size[1] <= 9 and m.stride[0] == m.stride[1])


def forward_supported(m: nn.Conv2d) -> bool:
    """forward form: additionally a thin input (an RGB stem) is zero-padded to 32 channels"""
    return _geometry_ok(m) and (m.in_channels % 32 == 0 or m.in_channels < 32) and m.out_channels % 8 == 0


class PreparedConv:
    """Split, tap-major copies of a conv weight for the two GEMM forms; rebuilt when the weight (or the per-output-channel
    scale folded into it, e.g. a deferred BatchNorm scale) changes."""

    def __init__(self, m: nn.Conv2d):
        self.m = m
        self._bwd = None  # (key, planes, sexp)
        self._bwdc = None  # (key, chunk-major copy of the backward planes)
        self._fwd = None

    def _key(self, cscale):
        w = self.m.weight
        return (w._version, w.data_ptr(), None if cscale is None else (cscale._version, cscale.data_ptr()))

    def backward_planes(self, cscale=None):
        key = self._key(cscale)
        if self._bwd is None or self._bwd[0] != key:
            W = self.m.weight.detach()
            planes, sexp = get_kernels().conv_prep_weights(W.contiguous(), True, cscale)
            # l1 bound of the backward-data GEMM (lk_conv_nhwc_f16x2_vjp): max|dX| <= max|g| * max_ci sum_{co,kh,kw} |W s_co|.
            # A per-layer constant, built with the planes (torch reductions: preparation, not the hot path).
            Wa = W.abs().float()
            if cscale is not None:
                Wa = Wa * cscale.abs().reshape(-1, 1, 1, 1)
            l1 = Wa.sum(dim=(0, 2, 3)).max().reshape(1).contiguous()
            self._bwd = (key, planes, sexp, l1)
        return self._bwd[1], self._bwd[2]

    def backward_l1(self, cscale=None):
        self.backward_planes(cscale)
        return self._bwd[3]

    def backward_planes_chunked(self, cscale=None):
        """the backward planes ``[2, T, N, K]`` once more in chunk-major order ``[2, T, K / 16, N, 16]`` (what the
        persistent window form of the fused launch stages: every 16-channel chunk of a tap is one contiguous block)"""
        planes, _ = self.backward_planes(cscale)
        key = self._bwd[0]
        if self._bwdc is None or self._bwdc[0] != key:
            two, T, N, Kd = planes.shape
            self._bwdc = (key, planes.view(two, T, N, Kd // 16, 16).permute(0, 1, 3, 2, 4).contiguous())
        return self._bwdc[1]

    @property
    def padded_in(self) -> int:
        """input channels as the forward kernel sees them (a thin stem is zero-padded to one 32-channel chunk)"""
        ci = self.m.in_channels
        return ci if ci % 32 == 0 else 32

    def forward_planes(self):
        key = self._key(None)
        if self._fwd is None or self._fwd[0] != key:
            W = self.m.weight.detach()
            if self.padded_in != W.shape[1]:
                Wp = W.new_zeros(W.shape[0], self.padded_in, *W.shape[2:])
                Wp[:, :W.shape[1]] = W
                W = Wp
            planes, sexp = get_kernels().conv_prep_weights(W.contiguous(), False, None)
            # l1 bound of the forward GEMM: max|conv(x)_n| <= max|x_n| * max_co sum_{ci,kh,kw} |W| — what the forward's fused
            # BatchNorm / activation kernel scales an image's planes from (lk_bn_act_fwd_nhwc_f16x2: x_mul), and max|bias|
            l1 = W.abs().float().sum(dim=(1, 2, 3)).max().reshape(1).contiguous()
            b = self.m.bias
            bmax = None if b is None else b.detach().abs().float().max().reshape(1).contiguous()
            self._fwd = (key, planes, sexp, l1, bmax, None if b is None else (b._version, b.data_ptr()))
        return self._fwd[1], self._fwd[2]

    def forward_l1(self):
        """(l1, max|bias| or None): device words with ``max|conv(x)_n + bias| <= max|x_n| * l1 + max|bias|``"""
        self.forward_planes()
        b = self.m.bias
        if b is not None and self._fwd[5] != (b._version, b.data_ptr()):
            self._fwd = self._fwd[:4] + (b.detach().abs().float().max().reshape(1).contiguous(), (b._version, b.data_ptr()))
        return self._fwd[3], self._fwd[4]


def backward_plan(m: nn.Conv2d, Hin: int, Win: int):
    """Launches of the backward-data of ``m`` for an input of ``Hin x Win`` pixels: one per residue class
    ``(h % s, w % s)`` of the input-gradient pixels, ``(Hc, Wc, oh0, ow0, taps)`` with ``taps = [(dh, dw, slice)]``:
    ``dX[i*s + oh0, j*s + ow0] = sum_taps g[i + dh, j + dw] W[:, :, kh, kw]``, where ``kh = oh0 + p - dh*s``."""
    return _backward_plan(int(m.stride[0]), tuple(int(p) for p in m.padding), tuple(int(k) for k in m.kernel_size), int(Hin), int(Win))


@functools.lru_cache(maxsize=512)
def _backward_plan(s, padding, kernel_size, Hin, Win):
    # (a pure function of the geometry, asked ~45 times per fit step: cached; callers must not modify the lists)
    (ph, pw), (KH, KW) = padding, kernel_size
    plans = []
    for rh in range(min(s, Hin)):
        for rw in range(min(s, Win)):
            taps = []
            for kh in range(KH):
                if (rh + ph - kh) % s:
                    continue
                for kw in range(KW):
                    if (rw + pw - kw) % s:
                        continue
                    taps.append(((rh + ph - kh) // s, (rw + pw - kw) // s, kh * KW + kw))
            plans.append(((Hin - rh + s - 1) // s, (Win - rw + s - 1) // s, rh, rw, taps))
    return plans


def conv_backward_data(prep: PreparedConv, g: SplitTensor, in_hw, cscale=None, out=None, accumulate=False, amax_out=None):
    """``dX [N, Hin, Win, Cin]`` (fp32, NHWC) of ``prep.m`` from the split output cotangent ``g [N, Hout, Wout, Cout]``.
    ``accumulate``: add into ``out`` (the other branch of a residual connection already wrote it).  ``amax_out``: device
    word that receives max |dX| (bit pattern; must be zeroed by the caller when not accumulating over launches)."""
    K = get_kernels()
    m = prep.m
    N = g.shape[0]
    Hin, Win = in_hw
    planes, sexp = prep.backward_planes(cscale)
    if out is None:
        out = torch.empty(N, Hin, Win, m.in_channels, dtype=torch.float32, device=g.planes.device)
        accumulate = False
    plans = backward_plan(m, Hin, Win)
    if not accumulate and any(not p[4] for p in plans):
        out.zero_()  # residue classes no tap reaches (1x1 stride-2: three of the four)
    for Hc, Wc, oh0, ow0, taps in plans:
        if taps and Hc and Wc:
            K.conv_nhwc_f16x2(g, planes, sexp, Hc, Wc, 1, out, m.stride[0], oh0, ow0, taps, accumulate=accumulate,
                              amax_out=amax_out)
    return out


def fused_backward_ok(m: nn.Conv2d) -> bool:
    """the backward-data of ``m`` is ONE dense launch with an 8-aligned channel count: the VJP epilogue applies"""
    return m.stride[0] == 1 and m.stride[1] == 1 and m.in_channels % 8 == 0


def conv_backward_data_vjp(prep: PreparedConv, g: SplitTensor, in_hw, cscale=None, add=None, mult=None, mult_amax=None,
                           scale=None, scale_amax=None, amax_word=None) -> SplitTensor:
    """``(dX + add) * mult * scale[channel]`` as a SplitTensor (with its measured ``amax``): :func:`conv_backward_data`
    followed by the sweep's element-wise VJP, in one launch (stride-1 convs, see :func:`fused_backward_ok`)."""
    K = get_kernels()
    m = prep.m
    Hin, Win = in_hw
    planes, sexp = prep.backward_planes(cscale)
    (Hc, Wc, oh0, ow0, taps), = backward_plan(m, Hin, Win)
    assert (Hc, Wc, oh0, ow0) == (Hin, Win, 0, 0)
    kw = {}
    if amax_word is not None:
        kw["amax_word"] = amax_word  # a zeroed device word for the measured max|result| (saves a fill launch)
    N = g.planes.shape[1]
    if planes.shape[3] % 16 == 0 and K.conv_winp_eligible(
            N, Hin, Win, planes.shape[3], planes.shape[2], len(taps), mult is not None and mult.dtype == torch.float32):
        kw["wplanes_chunked"] = prep.backward_planes_chunked(cscale)
    return K.conv_nhwc_f16x2_vjp(g, planes, sexp, prep.backward_l1(cscale), Hin, Win, taps, add=add, mult=mult,
                                 mult_amax=mult_amax, scale=scale, scale_amax=scale_amax, **kw)


def strided_fused_ok(m: nn.Conv2d, in_hw) -> bool:
    """the backward-data of ``m`` can join a strided fused launch (:func:`conv_backward_data_vjp_strided`): stride 2 on an
    even map, an 8-aligned channel count for the epilogue's 16-byte stores"""
    Hin, Win = in_hw
    return (supported(m) and m.stride[0] == 2 and m.stride[1] == 2 and Hin % 2 == 0 and Win % 2 == 0 and Hin >= 2 and Win >= 2
            and m.in_channels % 8 == 0 and m.out_channels % 32 == 0
            and (Hin + 2 * m.padding[0] - m.kernel_size[0]) // 2 + 1 == Hin // 2
            and (Win + 2 * m.padding[1] - m.kernel_size[1]) // 2 + 1 == Win // 2)


def strided_taps(descs, in_hw):
    """tap rows ``(dh, dw, weight slice, source, oh0, ow0)`` of the strided fused launch for ``descs = [(prep, g, cscale)]``
    (one or two convolutions that read the same ``in_hw`` input), or None when some residue class of the input pixels is
    reached by no tap (a lone strided 1 x 1 convolution)"""
    rows, classes = [], set()
    g0, w0 = descs[0][1], descs[0][0].m.weight
    for prep, g, _ in descs[1:]:
        # one launch = one GEMM shape: both cotangents [N, H/2, W/2, Cout] and both weights [Cout, Cin, ., .] must agree in
        # their channel counts (a 3x3 branch 32 -> 64 beside a 1x1 branch 32 -> 32 reading the same input does not fuse)
        if tuple(g.planes.shape) != tuple(g0.planes.shape) or tuple(prep.m.weight.shape[:2]) != tuple(w0.shape[:2]):
            return None
    for i, (prep, _, _) in enumerate(descs):
        for Hc, Wc, oh0, ow0, taps in backward_plan(prep.m, *in_hw):
            for dh, dw, sl in taps:
                rows.append((dh, dw, sl, i, oh0, ow0))
                classes.add((oh0, ow0))
    s = descs[0][0].m.stride[0]
    return rows if len(classes) == s * s and len(rows) <= 12 else None


def conv_backward_data_vjp_strided(descs, in_hw, add=None, mult=None, mult_amax=None, scale=None, scale_amax=None,
                                   amax_word=None) -> SplitTensor:
    """``(sum of dX over descs + add) * mult * scale[channel]`` as a SplitTensor for one or two STRIDED convolutions
    ``descs = [(prep, g, cscale)]`` that read the same input (the 3 x 3 main branch and the 1 x 1 shortcut of a residual
    down-sampling block): all residue classes, both convolutions and the sweep's element-wise VJP in one launch
    (lk_conv_nhwc_f16x2_vjp_strided) instead of five backward-data launches into an fp32 tensor and a pass over it."""
    K = get_kernels()
    Hin, Win = in_hw
    rows = strided_taps(descs, in_hw)
    assert rows is not None
    sources = []
    for prep, g, cscale in descs:
        planes, sexp = prep.backward_planes(cscale)
        sources.append((g, planes, sexp, prep.backward_l1(cscale)))
    return K.conv_nhwc_f16x2_vjp_strided(sources, Hin, Win, descs[0][0].m.stride[0], rows, add=add, mult=mult,
                                         mult_amax=mult_amax, scale=scale, scale_amax=scale_amax, amax_word=amax_word)


def conv_forward(prep: PreparedConv, x: SplitTensor, out=None, amax_out=None):
    """``y [N, Hout, Wout, Cout]`` (fp32, NHWC, no bias) of ``prep.m`` from the split input ``x [N, Hin, Win, Cin]``"""
    K = get_kernels()
    m = prep.m
    N, Hin, Win, _ = x.shape
    s, (ph, pw), (KH, KW) = m.stride[0], m.padding, m.kernel_size
    Ho, Wo = (Hin + 2 * ph - KH) // s + 1, (Win + 2 * pw - KW) // s + 1
    planes, sexp = prep.forward_planes()
    if out is None:
        out = torch.empty(N, Ho, Wo, m.out_channels, dtype=torch.float32, device=x.planes.device)
    taps = [(kh - ph, kw - pw, kh * KW + kw) for kh in range(KH) for kw in range(KW)]
    K.conv_nhwc_f16x2(x, planes, sexp, Ho, Wo, s, out, 1, 0, 0, taps, amax_out=amax_out)
    return out


def conv_forward_bn_act(prep: PreparedConv, x: SplitTensor, scale, shift, scale_amax, shift_amax, act: int, addend=None,
                        addend_bound=None, want_mask: bool = True, amax_words=None, y_out=None):
    """``act(conv(x) * scale[c] + shift[c] + addend)`` of the bias-free ``prep.m`` in ONE launch (lk_conv_bn_act_nhwc_f16x2):
    ``(y, mask, split, bound)`` as ``bn_act_forward_nhwc`` returns them for :func:`conv_forward`'s output — the same bits"""
    K = get_kernels()
    m = prep.m
    N, Hin, Win, _ = x.shape
    s, (ph, pw), (KH, KW) = m.stride[0], m.padding, m.kernel_size
    Ho, Wo = (Hin + 2 * ph - KH) // s + 1, (Win + 2 * pw - KW) // s + 1
    planes, sexp = prep.forward_planes()
    l1, bmax = prep.forward_l1()
    assert bmax is None, "conv_forward_bn_act: bias-free convolutions only"
    taps = [(kh - ph, kw - pw, kh * KW + kw) for kh in range(KH) for kw in range(KW)]
    return K.conv_bn_act_nhwc(x, planes, sexp, l1, Ho, Wo, s, taps, scale, shift, scale_amax, shift_amax, act, addend=addend,
                              addend_bound=addend_bound, want_mask=want_mask, amax_words=amax_words, y_out=y_out)


def _filter_l1(W: torch.Tensor) -> torch.Tensor:
    """device word ``max_co sum |W[co]|``: ``max|conv(x, W)_n| <= max|x_n| * l1`` (the bound the split-planes epilogue scales from)"""
    return W.abs().float().reshape(W.shape[0], -1).sum(1).max().reshape(1).contiguous()


def conv_forward_filters(m: nn.Conv2d, a: torch.Tensor, filt: torch.Tensor, key_tensor: torch.Tensor, amax_out=None,
                         xs: SplitTensor | None = None, planes: bool = False):
    """``conv2d(a, filt)`` with the geometry of ``m`` and an arbitrary filter bank ``filt [Dk, Cin, kh, kw]`` (the
    eigenvectors of an A factor: the Kron predictive's rotation of the unfolded inputs, matrix.py:406-456) on the
    implicit-GEMM kernel; returns ``[B, Dk, Ho, Wo]`` fp32 with POSITIONS contiguous — or, ``planes=True``, the same as a
    SplitTensor ``[B, Dk, Ho * Wo]`` with one scale per image (lk_conv_nhwc_f16x2_planes: what the quadratic-form kernel stages
    as it is).  The split planes of the filters are cached ON ``key_tensor`` (the eigenvector matrix they were cut from; an
    attribute of that tensor object, so the cache lives exactly as long as the decomposition — a table keyed by address
    would serve stale planes to the next decomposition allocated at the same place).  ``amax_out``: zeroed device word
    that receives max|result| (fp32 form only)."""
    K = get_kernels()
    key = (key_tensor._version, tuple(filt.shape))
    hit = getattr(key_tensor, "_lk_filter_planes", None)
    if hit is None or hit[0] != key:
        hit = (key, K.conv_prep_weights(filt.contiguous(), False, None), _filter_l1(filt))
        key_tensor._lk_filter_planes = hit
    planes_w, sexp = hit[1]
    if xs is None or tuple(xs.shape) != (a.shape[0], a.shape[2], a.shape[3], a.shape[1]):
        xh = a.permute(0, 2, 3, 1).contiguous()  # (a view when `a` is NHWC in memory already)
        xs = K.split_images_f16x2(xh)        # (``xs``: the split copy the forward pass already made of ``a``; one scale per image)
    N, Hin, Win, _ = xs.shape
    s, (ph, pw), (KH, KW) = m.stride[0], m.padding, m.kernel_size
    Ho, Wo = (Hin + 2 * ph - KH) // s + 1, (Win + 2 * pw - KW) // s + 1
    Dk = filt.shape[0]
    taps = [(kh - ph, kw - pw, kh * KW + kw) for kh in range(KH) for kw in range(KW)]
    if planes:
        return K.conv_nhwc_f16x2_planes(xs, planes_w, sexp, hit[2], Ho, Wo, s, taps, config=(K.conv_config | 2))
    out = torch.empty(N, Dk, Ho, Wo, dtype=torch.float32, device=a.device)
    # config bit 4: position-contiguous output; the wrapper reads shapes off an NHWC-shaped view of the same memory
    K.conv_nhwc_f16x2(xs, planes_w, sexp, Ho, Wo, s, out.view(N, Ho, Wo, Dk), 1, 0, 0, taps, amax_out=amax_out,
                      config=(K.conv_config | 2 | 16))
    return out


def rotate_channels(g: SplitTensor, Q: torch.Tensor, key_tensor: torch.Tensor, planes: bool = False):
    """``out[n, :, p] = Q^T g[n, p, :]`` for an NHWC split tensor ``g [N, H, W, C]`` and a square ``Q [C, C]`` — the Kron
    predictive's rotation of the output cotangents into a G factor's eigenbasis (matrix.py:406-456) — as a 1x1
    convolution on the implicit-GEMM kernel with POSITION-contiguous output ``[N, C, H*W]``: fp32, or (``planes=True``) a
    SplitTensor with ``g``'s one scale re-derived from the bound ``max|g| * l1(Q^T)`` — what the quadratic-form kernel
    reads, without the un-split / transpose copy and the library GEMM.  Filter planes cached on ``key_tensor``."""
    K = get_kernels()
    N, H, W, C = g.shape
    key = (key_tensor._version, "rot", C)
    hit = getattr(key_tensor, "_lk_rot_planes", None)
    if hit is None or hit[0] != key:
        Wq = Q.T.reshape(C, C, 1, 1).contiguous()
        hit = (key, K.conv_prep_weights(Wq, False, None), _filter_l1(Wq))
        key_tensor._lk_rot_planes = hit
    planes_w, sexp = hit[1]
    if planes:
        return K.conv_nhwc_f16x2_planes(g, planes_w, sexp, hit[2], H, W, 1, [(0, 0, 0)], config=(K.conv_config | 2))
    out = torch.empty(N, C, H * W, dtype=torch.float32, device=g.planes.device)
    K.conv_nhwc_f16x2(g, planes_w, sexp, H, W, 1, out.view(N, H, W, C), 1, 0, 0, [(0, 0, 0)], config=(K.conv_config | 2 | 16))
    return out
