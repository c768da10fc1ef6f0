"""ctypes binding of ``csrc/liblaplace_hip.so`` (the C ABI declared in include/laplace_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every hot operation is one of
our HIP entry points.  There is NO fallback: if the shared library is missing, or a tensor is
not an fp32 contiguous tensor on a ROCm device, the call raises.

The tensor-level API lives on :class:`HipKernels`; ``get_kernels()`` returns the process-wide
instance.  Tests that exercise *host logic* on a CPU-only box install a stand-in through
``set_kernels_for_testing`` (see tests/emulated_kernels.py) — the product never does.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (LK_LIB: another build of the same library, e.g. csrc/liblaplace_hip_dev.so with the kernels' development switches)
LIB_PATH = os.environ.get("LK_LIB") or os.path.join(_HERE, "csrc", "liblaplace_hip.so")

LK_GRAM_UPPER_ONLY = 1
LK_GRAM_SLABS_PERSIST = 2

_c_f32p = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_f32 = ctypes.c_float
_sz = ctypes.c_size_t
_vp = ctypes.c_void_p
_u32 = ctypes.c_uint

# name -> (restype, argtypes); the single source of truth shared with tests/test_capi_symbols.py
SIGNATURES = {
    "lk_version": (_int, []),
    "lk_last_error": (ctypes.c_char_p, []),
    "lk_im2col_split_f16x2": (_int, [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "lk_comm_unique_id": (_int, [_vp]),
    "lk_comm_init_rank": (_int, [ctypes.POINTER(ctypes.c_void_p), _int, _vp, _int]),
    "lk_comm_destroy": (_int, [_vp]),
    "lk_allreduce_sum_f32": (_int, [_vp, _vp, _i64, _vp]),
    "lk_loss_workspace_bytes": (_sz, [_i64]),
    "lk_softmax_hess_sqrt_f32": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "lk_softmax_hess_chol_f32": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "lk_sq_err_sum_f32": (_int, [_vp, _vp, _i64, _f32, _vp, _vp, _vp]),
    "lk_gram_workspace_bytes": (_sz, [_i64, _i64]),
    "lk_gram_nt_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "lk_gram_slabs_reduce_f32": (_int, [_vp, _sz, _i64, _i64, _f32, _vp, _u32, _vp]),
    "lk_gram_tn_f32": (_int, [_vp, _i64, _i64, _i64, _f32, _vp, _u32, _vp, _sz, _vp]),
    "lk_gram_nt_f32": (_int, [_vp, _i64, _i64, _i64, _f32, _vp, _u32, _vp, _sz, _vp]),
    "lk_gram_nt_seg_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _f32, _vp, _u32, _vp, _sz, _vp]),
    "lk_gram_conv_nhwc_f32": (
        _int,
        [_vp, _i64, _i64, _i64, _i64, _int, _int, _int, _int, _int, _int, _int, _int, _f32, _vp, _u32, _vp, _sz, _vp],
    ),
    "lk_conv3x3_shiftcorr_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "lk_conv3x3_shiftcorr_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _f32, _vp, _vp, _sz, _vp]),
    "lk_conv3x3_pixgram_assemble_f32": (_int, [_vp, _i64, _i64, _i64, _f32, _vp, _vp]),
    "lk_conv3x3_pixpair_plan": (_int, [_i64, _i64, _i64, _vp, _vp, _vp]),
    "lk_conv3x3_pixpair_tables": (_int, [_i64, _i64, _i64, _vp, _vp]),
    "lk_conv3x3_pixpair_accumulate_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _f32, _vp, _vp, _i64, _vp]),
    "lk_conv3x3_pixpair_accumulate_f16x2": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _f32, _vp, _vp, _i64, _vp, _vp]),
    "lk_conv3x3_pixpair_accumulate13_f16x2": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _f32, _vp, _vp, _vp, _vp]),
    "lk_conv3x3_pixpair_assemble_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _f32, _vp, _vp]),
    "lk_conv3x3_pixpair_assemble2_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _f32, _vp, _int, _vp]),
    "lk_nchw_to_nhwc_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "lk_absmax_f32": (_int, [_vp, _i64, _vp, _i64, _i64, _vp, _vp]),
    "lk_copy_absmax_f32": (_int, [_vp, _vp, _i64, _vp, _vp]),
    "lk_split_f16x2": (_int, [_vp, _i64, _vp, _f32, _vp, _vp, _vp, _vp]),
    "lk_split_images_f16x2": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "lk_conv_prep_weights_f16x2": (_int, [_vp, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp]),
    "lk_conv_nhwc_f16x2": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64,
                                  _i64, _i64, _i64, _i64, _vp, _vp, _vp, _int, _vp, _int, _vp]),
    "lk_conv_nhwc_f16x2_planes": (_int, [_vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64,
                                         _i64, _vp, _vp, _vp, _vp, _vp, _int, _vp]),
    "lk_conv_bn_act_nhwc_f16x2": (_int, [_vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64,
                                         _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _vp, _vp, _vp, _vp, _vp,
                                         _vp, _vp, _int, _vp]),
    "lk_conv_nhwc_f16x2_vjp": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64,
                                      _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _i64, _vp, _vp, _vp, _vp, _vp,
                                      _vp, _int, _vp]),
    "lk_conv_winp_eligible": (_int, [_i64, _i64, _i64, _i64, _i64, _i64, _int]),
    "lk_conv_nhwc_f16x2_vjp_wc": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64,
                                         _i64, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _int,
                                         _vp]),
    "lk_conv_nhwc_f16x2_vjp_strided": (_int, [_vp] * 16 + [_i64] * 9 + [_vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _i64, _vp, _vp, _vp,
                                               _vp, _vp, _vp, _int, _vp]),
    "lk_vjp_nhwc_split_f16x2": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "lk_bn_act_fwd_nhwc_f16x2": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _i64, _i64, _i64, _vp, _vp,
                                        _vp, _vp, _vp, _vp, _vp, _vp]),
    "lk_unsplit_transpose_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "lk_gram_tn_f16x2_workspace_bytes": (_sz, [_i64, _i64]),
    "lk_gram_tn_f16x2": (_int, [_vp, _vp, _vp, _i64, _i64, _f32, _vp, _vp, _vp, _sz, _vp]),
    "lk_gemm_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _int, _int, _f32,
                           _int, _vp]),
    "lk_kron_pow_f32": (_int, [_vp, _i64, _vp, _i64, _vp, _f32, _int, _vp, _vp]),
    "lk_pack_upper_f32": (_int, [_vp, _i64, _vp, _vp]),
    "lk_unpack_upper_f32": (_int, [_vp, _i64, _vp, _vp]),
    "lk_symmetrize_f32": (_int, [_vp, _i64, _vp]),
    "lk_permute_sym_f32": (_int, [_vp, _i64, _i64, _vp, _int, _vp]),
    "lk_finalize_factors_f32": (_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lk_diag_ggn_linear_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _f32, _vp, _vp, _vp]),
    "lk_jac_linear_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp]),
    "lk_jac_conv_f32": (
        _int,
        [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _int, _int, _int, _int, _int, _int, _int, _int, _vp, _i64, _i64, _i64, _vp],
    ),
    "lk_sq_colsum_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _f32, _vp, _vp]),
    "lk_bn_act_fwd_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _int, _vp, _vp, _vp]),
    "lk_vjp_scale_mask_f32": (_int, [_vp, _vp, _vp, _int, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "lk_ll_ggn_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "lk_ll_ggn_full_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _int, _f32, _vp, _vp, _sz, _vp]),
    "lk_syevj_workspace_bytes": (_sz, [_i64]),
    "lk_syevj_f32": (_int, [_vp, _i64, _vp, _vp, _int, _int, _vp, _vp, _sz, _vp]),
    "lk_syevj_batched_f32": (_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _vp, _i64]),
    "lk_kron_logdet_workspace_bytes": (_sz, [_i64]),
    "lk_kron_logdet_f32": (_int, [_vp, _i64, _vp, _i64, _vp, _int, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "lk_kron_logdet_blocks_workspace_bytes": (_sz, [_i64, _i64]),
    "lk_kron_logdet_blocks_f32": (_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "lk_kron_quadform_linear_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "lk_diag_quadform_linear_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "lk_quadform_shared_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "lk_kron_quadform_shared_planes_f16x2": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64,
                                                    _vp, _vp, _vp, _sz, _vp]),
    "lk_kron_quadform_shared_seedmajor_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "lk_kron_quadform_shared_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "lk_diag_quadform_shared_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "lk_diag_ggn_shared_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "lk_diag_ggn_shared_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, ctypes.c_float, _vp, _vp, _sz, _vp]),
    "lk_diag_quadform_js_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "lk_dense_quadform_ll_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "lk_jac_last_layer_f32": (_int, [_vp, _i64, _i64, _i64, _int, _vp, _vp]),
    "lk_dense_quadform_ll_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _int, _vp, _vp, _sz, _vp]),
}


class LaplaceHipError(RuntimeError):
    pass


class SplitTensor:
    """Two fp16 planes ``planes[0] + planes[1] ~= x * 2**sexp`` of a tensor (include/laplace_hip.h, lk_split_f16x2).
    ``sexp``: int32 ``[1]`` — one scale for the tensor (the reverse sweep's cotangents) — or ``[N]``, one per image of the
    leading dimension (the forward's activations: lk_split_images_f16x2, lk_bn_act_fwd_nhwc_f16x2)."""

    __slots__ = ("planes", "sexp", "amax", "chunked")

    def __init__(self, planes: torch.Tensor, sexp: torch.Tensor, amax: torch.Tensor | None = None, chunked: bool = False):
        #: ``amax``: device word(s) with the MEASURED max|x| (per scale entry) when the producer provides them (fused
        #: convolution epilogue, per-image forward); consumers that need a bound otherwise use 2**(15 - sexp)
        #: ``chunked``: the planes of a position-contiguous ``[N, D, L]`` tensor stored CHUNK-major, ``planes [2, N, L / 16, D, 16]``
        #: (lk_conv_nhwc_f16x2_planes -> lk_kron_quadform_shared_planes_f16x2: a staged block of 16 positions x 32 rows is one
        #: contiguous kilobyte); ``shape`` and ``float()`` stay the logical ``[N, D, L]``
        self.planes, self.sexp, self.amax, self.chunked = planes, sexp, amax, bool(chunked)

    @property
    def shape(self):
        if self.chunked:
            _, N, nch, D, w = self.planes.shape
            return torch.Size((N, D, nch * w))
        return self.planes.shape[1:]

    def chunk_major(self) -> "SplitTensor":
        """the same ``[N, D, L]`` tensor with chunk-major planes (a copy unless it already is; ``L % 16 == 0``)"""
        if self.chunked:
            return self
        _, N, D, L = self.planes.shape
        pl = self.planes.view(2, N, D, L // 16, 16).permute(0, 1, 3, 2, 4).contiguous()
        return SplitTensor(pl, self.sexp, self.amax, chunked=True)

    @property
    def per_image(self) -> bool:
        return self.sexp.numel() > 1

    def float(self) -> torch.Tensor:
        """fp32 reconstruction (tests / fallbacks)"""
        s = self.sexp.float()
        pl = self.planes
        if self.chunked:
            _, N, nch, D, w = pl.shape
            pl = pl.permute(0, 1, 3, 2, 4).reshape(2, N, D, nch * w)
        if s.numel() > 1:
            s = s.reshape(-1, *([1] * (pl.dim() - 2)))
        return (pl[0].float() + pl[1].float()) * torch.exp2(-s)


def load_library(path: str = LIB_PATH) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise LaplaceHipError(
            f"HIP extension not built: {path} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C laplace_amd/csrc`). There is no CPU fallback."
        )
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda idx: torch.cuda.current_stream(idx).cuda_stream)


def _ptr(t: Optional[torch.Tensor]):
    """device address for a ``c_void_p`` argument (a plain int: ctypes converts it; ~1000 of these per fit step)"""
    return None if t is None else t.data_ptr()


def _check(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise LaplaceHipError(f"{name}: expected a tensor, got {type(t)}")
    if not t.is_cuda:
        raise LaplaceHipError(f"{name}: tensor is on {t.device}; the HIP backend needs a ROCm device (no CPU path)")
    if t.dtype != dtype:
        raise LaplaceHipError(f"{name}: dtype {t.dtype} not supported (need {dtype})")
    if not t.is_contiguous():
        raise LaplaceHipError(f"{name}: tensor must be contiguous")
    return t


def _one_scale(x: "SplitTensor", what: str) -> "SplitTensor":
    """consumers whose GEMM rows or reductions run ACROSS images (the reverse sweep's kernels) take one scale per tensor"""
    if x is not None and x.sexp.numel() != 1:
        raise LaplaceHipError(f"{what}: a split tensor with one scale per image is a forward operand (lk_conv_nhwc_f16x2); "
                              "this kernel reduces across images and needs one scale for the tensor")
    return x


def is_channels_last(x) -> bool:
    """logical [B, C, H, W] tensor whose memory is dense NHWC (and not also dense NCHW)"""
    return torch.is_tensor(x) and x.dim() == 4 and not x.is_contiguous() and x.permute(0, 2, 3, 1).is_contiguous()


def keep_layout(x):
    """``x`` as the kernels' wrappers want it: dense NCHW, or left alone when it is dense NHWC (no transposing copy)"""
    return x if is_channels_last(x) else x.contiguous()


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


class HipKernels:
    """Tensor-level wrappers; every method enqueues on torch's current stream and returns at once."""

    name = "hip"
    conv_config = 2  # lk_conv_nhwc_f16x2: bit 0 = 64-deep K chunks, bit 1 = never use the patch form (default: measured faster inside
    # the step, 12.2 vs 13.0 ms), bit 2 = 16-deep chunks in four LDS stages
    softmax_chol_max_c = 2000  # LK_SOFTMAX_CHOL_MAX_C (include/laplace_hip.h): wider outputs use the symmetric root

    def __init__(self, lib: Optional[ctypes.CDLL] = None):
        self.lib = lib if lib is not None else load_library()
        self._ws: dict = {}
        # optional per-launch timing (bench.py's roofline leg): name -> list of (start_evt, end_evt, work)
        self.profile: Optional[dict] = None
        # 3x3/stride-1 conv A factors through the shift-correlation identity (lk_conv3x3_shiftcorr_f32)
        self.use_shiftcorr = True

    def _timed(self, name: str, work: float, dev, call, nbytes: float | None = None):
        """Run ``call()``; when profiling is on, bracket it with HIP events on the launch stream.  ``work``: algorithmic flop
        (matrix families) or bytes (streaming families) of the launch; ``nbytes``: algorithmic HBM bytes of a matrix-family
        launch — every operand read once, every result written once (what bench.py prices its measured traffic against)."""
        if self.profile is None:
            return call()
        st = torch.cuda.current_stream(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        rc = call()
        e1.record(st)
        self.profile.setdefault(name, []).append((e0, e1, work, nbytes))
        return rc

    # ---- plumbing -----------------------------------------------------------------------------
    def _stream(self, dev) -> int:
        """raw handle of torch's current stream on ``dev`` (the C call behind ``torch.cuda.current_stream(dev).cuda_stream``
        without the Stream object and its device look-ups: ~170 of these per fit step)"""
        idx = dev.index if isinstance(dev, torch.device) else torch.device(dev).index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)

    def _workspace(self, nbytes: int, dev) -> torch.Tensor:
        key = (dev.index, self._stream(dev))
        buf = self._ws.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=dev)
            self._ws[key] = buf
        return buf

    def _rc(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.lk_last_error()
            raise LaplaceHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")

    # ---- likelihood ---------------------------------------------------------------------------
    def softmax_hess_sqrt(self, f, y=None, loss_accum=None, cholesky=False):
        """Columns of a root of ``diag(p) - p p^T`` as backward seeds ``[S, B, C]``: the symmetric root
        (``S = C``) or, with ``cholesky=True``, the rank-revealing Cholesky root (``S = C - 1``)."""
        _check(f, "f")
        B, C = f.shape
        ws = None
        if y is not None and loss_accum is not None:
            ws = torch.empty(max(1024, (B + 3) // 4 + 1), dtype=torch.float32, device=f.device)  # per-block partials
        if cholesky and 2 <= C <= self.softmax_chol_max_c:
            S = torch.empty(C - 1, B, C, dtype=torch.float32, device=f.device)
            if y is not None:
                _check(y, "y", torch.int64)
            if loss_accum is not None:
                _check(loss_accum, "loss_accum")
            self._rc(
                self.lib.lk_softmax_hess_chol_f32(_ptr(f), _ptr(y), B, C, _ptr(S), _ptr(loss_accum), _ptr(ws),
                                                  self._stream(f.device)),
                "lk_softmax_hess_chol_f32",
            )
            return S
        S = torch.empty(C, B, C, dtype=torch.float32, device=f.device)
        if y is not None:
            _check(y, "y", torch.int64)
        if loss_accum is not None:
            _check(loss_accum, "loss_accum")
        self._rc(
            self.lib.lk_softmax_hess_sqrt_f32(_ptr(f), _ptr(y), B, C, _ptr(S), _ptr(loss_accum), _ptr(ws),
                                              self._stream(f.device)),
            "lk_softmax_hess_sqrt_f32",
        )
        return S

    def sq_err_sum(self, f, y, scale, loss_accum):
        _check(f, "f"), _check(y, "y"), _check(loss_accum, "loss_accum")
        if f.shape != y.shape:
            raise LaplaceHipError(f"sq_err_sum: shape mismatch {tuple(f.shape)} vs {tuple(y.shape)}")
        self._rc(
            self.lib.lk_sq_err_sum_f32(_ptr(f), _ptr(y), f.numel(), float(scale), _ptr(loss_accum),
                                       _ptr(torch.empty(1024, dtype=torch.float32, device=f.device)),
                                       self._stream(f.device)),
            "lk_sq_err_sum_f32",
        )

    # ---- Gram family --------------------------------------------------------------------------
    def gram_tn(self, X, alpha, out, upper_only=False):
        _check(X, "X"), _check(out, "out")
        K, n = X.shape
        assert out.shape == (n, n)
        nb = self.lib.lk_gram_workspace_bytes(n, max(K, 1))
        ws = self._workspace(nb, X.device)
        self._rc(
            self._timed("gram_tn", float(K) * n * (n + 1), X.device, lambda: self.lib.lk_gram_tn_f32(
                _ptr(X), K, n, n, float(alpha), _ptr(out), LK_GRAM_UPPER_ONLY if upper_only else 0, _ptr(ws), ws.numel(),
                self._stream(X.device))),
            "lk_gram_tn_f32",
        )
        return out

    def gram_nt_slab_bytes(self, nb_total, n, L) -> int:
        return int(self.lib.lk_gram_nt_workspace_bytes(int(nb_total), int(n), int(L)))

    def gram_slabs_reduce(self, slabs, n, L, alpha, out, upper_only=False):
        """``out += alpha * sum(persistent slabs)`` — the one-off reduction of ``gram_nt(..., persist=slabs)``."""
        _check(out, "out")
        self._rc(self.lib.lk_gram_slabs_reduce_f32(_ptr(slabs), slabs.numel(), int(n), int(L), float(alpha), _ptr(out),
                                                   LK_GRAM_UPPER_ONLY if upper_only else 0, self._stream(out.device)),
                 "lk_gram_slabs_reduce_f32")
        return out

    def gram_nt(self, X, alpha, out, upper_only=False, persist=None):
        """``out += alpha * sum_b X_b X_b^T``; ``X`` is ``[nb, n, L]`` or a list of up to 16 such tensors
        (per-seed gradients, consumed in place through a pointer table).  ``persist``: a zero-initialised uint8 buffer
        of at least ``gram_nt_slab_bytes`` that accumulates the split-K partial tiles across calls instead of ``out``
        (reduce once with :meth:`gram_slabs_reduce`; ``alpha`` is then applied there)."""
        segs = list(X) if isinstance(X, (list, tuple)) else [X]
        for t in segs:
            _check(t, "X")
        _check(out, "out")
        nbat, n, L = segs[0].shape
        assert all(t.shape == segs[0].shape for t in segs) and out.shape == (n, n)
        if len(segs) > 16:
            segs = [torch.cat(segs)]
            nbat = segs[0].shape[0]
        nb = self.lib.lk_gram_nt_workspace_bytes(len(segs) * nbat, n, L)
        dev = segs[0].device
        flags = LK_GRAM_UPPER_ONLY if upper_only else 0
        if persist is not None:
            if persist.numel() < nb:
                raise LaplaceHipError("gram_nt: persistent slab buffer too small")
            ws, flags = persist, flags | LK_GRAM_SLABS_PERSIST
        else:
            ws = self._workspace(nb, dev)
        ptrs = (ctypes.c_void_p * len(segs))(*[t.data_ptr() for t in segs])
        self._rc(
            self._timed("gram_nt", float(len(segs) * nbat * L) * n * (n + 1), dev, lambda: self.lib.lk_gram_nt_seg_f32(
                ptrs, len(segs), nbat, n, L, float(alpha), _ptr(out), flags, _ptr(ws),
                ws.numel(), self._stream(dev))),
            "lk_gram_nt_seg_f32",
        )
        return out

    is_channels_last = staticmethod(lambda x: is_channels_last(x))

    def nchw_to_nhwc(self, x, out=None):
        if self.is_channels_last(x):  # already NHWC in memory (the sweep's own forward): a view, no pass over the data
            v = x.permute(0, 2, 3, 1)
            if out is None:
                return v
            out.copy_(v)
            return out
        _check(x, "x")
        B, C, H, W = x.shape
        if out is None:
            out = torch.empty(B, H, W, C, dtype=torch.float32, device=x.device)
        else:
            _check(out, "out")
            assert tuple(out.shape) == (B, H, W, C)
        if B:
            self._rc(self.lib.lk_nchw_to_nhwc_f32(_ptr(x), B, C, H * W, _ptr(out), self._stream(x.device)),
                     "lk_nchw_to_nhwc_f32")
        return out

    # ---- split-fp16 convolution (lk_conv.hip) ---------------------------------------------------------
    def absmax(self, x, out=None):
        """device word holding the bit pattern of ``max |x|`` (a [1] float32 tensor: the bits ARE that float)"""
        _check(x, "x")
        if out is None:
            out = torch.empty(1, dtype=torch.float32, device=x.device)
        self._rc(self.lib.lk_absmax_f32(_ptr(x), x.numel(), None, 1, 1, _ptr(out), self._stream(x.device)), "lk_absmax_f32")
        return out

    #: ``False``: the stacked pixel-pair inputs are copied by the runtime and measured in a pass of their own before they
    #: are split (measured: no difference in the step; one pass less at the end of a fit)
    use_copy_absmax = True

    def copy_absmax(self, x, out, amax):
        """``out[...] = x`` and ``amax[0] = max(amax[0], max|x|)`` in one pass (lk_copy_absmax_f32); ``amax`` is not reset"""
        _check(x, "x"), _check(out, "out"), _check(amax, "amax")
        if not (x.is_contiguous() and out.is_contiguous() and x.numel() == out.numel() and x.numel() % 4 == 0
                and x.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0):
            raise LaplaceHipError("copy_absmax: contiguous 16-byte aligned tensors of the same size, numel % 4 == 0")
        self._rc(self.lib.lk_copy_absmax_f32(_ptr(x), _ptr(out), x.numel(), _ptr(amax), self._stream(x.device)), "lk_copy_absmax_f32")
        return out

    def split_f16x2(self, x, amax=None, bound_mul=1.0, out=None):
        """fp32 tensor -> :class:`SplitTensor` of the same shape (scale from ``amax[0] * bound_mul``, measured if absent).
        ``out``: fp16 ``[2, *x.shape]`` workspace for the planes (each plane contiguous), allocated if absent"""
        _check(x, "x")
        if x.numel() % 8:
            raise LaplaceHipError("split_f16x2: numel % 8 != 0")
        if amax is None:
            amax = self.absmax(x)
        planes = out if out is not None else torch.empty((2,) + tuple(x.shape), dtype=torch.float16, device=x.device)
        if (planes.dtype != torch.float16 or tuple(planes.shape) != (2,) + tuple(x.shape) or planes.device != x.device
                or not planes[0].is_contiguous() or not planes[1].is_contiguous()):
            raise LaplaceHipError("split_f16x2: out must be fp16 [2, *x.shape] with contiguous planes on x's device")
        sexp = torch.empty(1, dtype=torch.int32, device=x.device)
        self._rc(self.lib.lk_split_f16x2(_ptr(x), x.numel(), _ptr(amax), float(bound_mul), _ptr(planes[0]), _ptr(planes[1]),
                                         _ptr(sexp), self._stream(x.device)), "lk_split_f16x2")
        return SplitTensor(planes, sexp)

    #: ``False``: the A factors of strided / stem convolutions stay on the exact-fp32 MFMA kernel (lk_gram_conv_nhwc_f32)
    use_gram_conv16 = True

    def im2col_split(self, x, kernel_size, stride, padding, Kp, amax=None):
        """lk_im2col_split_f16x2: the patch matrix of a convolution over ``x`` (logical ``[B, C, H, W]`` over NHWC memory, fp32) as a
        :class:`SplitTensor` ``[B * Ho * Wo, Kp]`` with one scale — columns in the kernels' native order ``(kh, kw, ci)``, zero
        padded to ``Kp`` columns"""
        if x.dim() != 4:
            raise LaplaceHipError("im2col_split: a [B, C, H, W] tensor")
        if not self.is_channels_last(x):
            _check(x, "x")
        B, C, H, W = x.shape
        xh = self.nchw_to_nhwc(x)  # (a view when x is NHWC in memory already)
        (KH, KW), s, p = kernel_size, int(stride), int(padding)
        Ho, Wo = (H + 2 * p - KH) // s + 1, (W + 2 * p - KW) // s + 1
        if amax is None:
            amax = self.absmax(xh)
        rows = B * Ho * Wo
        planes = torch.empty((2, rows, Kp), dtype=torch.float16, device=x.device)
        sexp = torch.empty(1, dtype=torch.int32, device=x.device)
        self._rc(self._timed("im2col16", 4.0 * xh.numel() + 4.0 * rows * Kp, x.device, lambda: self.lib.lk_im2col_split_f16x2(
            _ptr(xh), B, H, W, C, KH, KW, s, p, Ho, Wo, int(Kp), _ptr(amax), _ptr(planes[0]), _ptr(planes[1]), _ptr(sexp),
            self._stream(x.device))), "lk_im2col_split_f16x2")
        return SplitTensor(planes, sexp)

    def split_images_f16x2(self, x):
        """fp32 ``[N, ...]`` -> :class:`SplitTensor` with ONE SCALE PER IMAGE (lk_split_images_f16x2): ``sexp [N]`` from
        each image's own ``max|x_n|``, which rides along as ``amax [N]``"""
        _check(x, "x")
        N = x.shape[0]
        per = x.numel() // max(N, 1)
        if per % 8:
            raise LaplaceHipError("split_images_f16x2: elements per image % 8 != 0")
        planes = torch.empty((2,) + tuple(x.shape), dtype=torch.float16, device=x.device)
        sexp = torch.empty(N, dtype=torch.int32, device=x.device)
        amax = torch.empty(N, dtype=torch.float32, device=x.device)
        for i in range(0, max(N, 1), self.MAX_IMAGES_PER_LAUNCH):  # (the image index is a grid dimension: 65535 per launch)
            n = min(self.MAX_IMAGES_PER_LAUNCH, N - i)
            self._rc(self.lib.lk_split_images_f16x2(_ptr(x[i:i + n]), n, per, _ptr(planes[0][i:i + n]), _ptr(planes[1][i:i + n]),
                                                    _ptr(sexp[i:i + n]), _ptr(amax[i:i + n]), self._stream(x.device)),
                     "lk_split_images_f16x2")
        return SplitTensor(planes, sexp, amax)

    MAX_IMAGES_PER_LAUNCH = 65535

    def conv_prep_weights(self, W, transpose, cscale=None):
        """conv weight ``[Co, Ci, KH, KW]`` -> (planes ``[2, T, N, K]`` fp16, sexp); see lk_conv_prep_weights_f16x2"""
        _check(W, "W")
        Co, Ci = W.shape[0], W.shape[1]
        T = W[0, 0].numel()
        N, Kd = (Ci, Co) if transpose else (Co, Ci)
        planes = torch.empty(2, T, N, Kd, dtype=torch.float16, device=W.device)
        sexp = torch.empty(1, dtype=torch.int32, device=W.device)
        ws = torch.empty(1, dtype=torch.int32, device=W.device)
        if cscale is not None:
            _check(cscale, "cscale")
        self._rc(self.lib.lk_conv_prep_weights_f16x2(_ptr(W), Co, Ci, T, 1 if transpose else 0, _ptr(cscale), _ptr(ws),
                                                     _ptr(planes), _ptr(sexp), self._stream(W.device)),
                 "lk_conv_prep_weights_f16x2")
        return planes, sexp

    def _zero16(self, dev):
        z = self._zeros.get(dev.index) if hasattr(self, "_zeros") else None
        if z is None:
            if not hasattr(self, "_zeros"):
                self._zeros = {}
            z = self._zeros[dev.index] = torch.zeros(64, dtype=torch.uint8, device=dev)
        return z

    @staticmethod
    def conv_valid_pairs(Hc, Wc, in_mul, Hi, Wi, taps) -> int:
        """(output pixel, tap) pairs of one image whose input pixel ``(i * in_mul + dh, j * in_mul + dw)`` lies INSIDE the
        ``Hi x Wi`` map: the multiply-adds the convolution has to do (zero-padding taps are no work, and the kernels
        skip them where a tile holds none) — the flop figure behind the roofline numbers."""
        def inside(n_out, n_in, d):
            lo = 0 if d >= 0 else (-d + in_mul - 1) // in_mul          # first i with i * in_mul + d >= 0
            hi = min(n_out - 1, (n_in - 1 - d) // in_mul) if n_in - 1 - d >= 0 else -1
            return max(hi - lo + 1, 0)

        return sum(inside(Hc, Hi, t[0]) * inside(Wc, Wi, t[1]) for t in taps)

    def conv_nhwc_f16x2(self, x, wplanes, wsexp, Hc, Wc, in_mul, out, out_step, oh0, ow0, taps, accumulate=False,
                        amax_out=None, config=None):
        """one launch of lk_conv_nhwc_f16x2; ``x``: SplitTensor [N, Hi, Wi, Ci]; ``wplanes`` [2, T, Co, Ci];
        ``out`` fp32 [N, Ho, Wo, Co]; ``taps``: list of (dh, dw, weight slice)"""
        N, Hi, Wi, Ci = x.planes.shape[1:]
        _check(out, "out")
        Co = wplanes.shape[2]
        assert wplanes.shape[3] == Ci and out.shape[0] == N and out.shape[3] == Co
        flat = (ctypes.c_int * (3 * len(taps)))(*[int(v) for t in taps for v in t])
        cfg = self.conv_config if config is None else config
        z = self._zero16(out.device)
        # algorithmic work: the fp32 multiply-adds of the convolution (each is three fp16 MFMA multiply-adds on the chip)
        work = 2.0 * N * Co * Ci * self.conv_valid_pairs(Hc, Wc, in_mul, Hi, Wi, taps) if self.profile is not None else 0.0
        self._rc(self._timed("conv16", work, out.device, lambda: self.lib.lk_conv_nhwc_f16x2(
            _ptr(x.planes[0]), _ptr(x.planes[1]), _ptr(x.sexp), x.sexp.numel(), N, Hi, Wi, Ci, _ptr(wplanes[0]), _ptr(wplanes[1]),
            _ptr(wsexp), Co, Hc, Wc, in_mul, out.shape[1], out.shape[2], out_step, oh0, ow0, len(taps), flat, _ptr(z),
            _ptr(out), 1 if accumulate else 0, _ptr(amax_out), int(cfg), self._stream(out.device))), "lk_conv_nhwc_f16x2")
        return out

    def conv_nhwc_f16x2_planes(self, x, wplanes, wsexp, w_l1, Ho, Wo, in_mul, taps, config=None):
        """lk_conv_nhwc_f16x2_planes: the convolution of SplitTensor ``x [N, Hi, Wi, Ci]`` with ``wplanes [2, T, Co, Ci]``,
        position-contiguous and ALREADY SPLIT: SplitTensor ``[N, Co, Ho * Wo]`` with the scales of ``x``'s granularity
        (one for the tensor or one per image), each from the bound ``max|x_n| * w_l1`` (no pass over the output)"""
        N, Hi, Wi, Ci = x.planes.shape[1:]
        Co = wplanes.shape[2]
        dev = x.planes.device
        assert wplanes.shape[3] == Ci and (Ho * Wo) % 16 == 0
        planes = torch.empty((2, N, (Ho * Wo) // 16, Co, 16), dtype=torch.float16, device=dev)  # chunk-major (SplitTensor.chunked)
        sexp = torch.empty(x.sexp.numel(), dtype=torch.int32, device=dev)
        flat = (ctypes.c_int * (3 * len(taps)))(*[int(v) for t in taps for v in t])
        cfg = self.conv_config if config is None else config
        amax = x.amax if (x.amax is not None and x.amax.numel() == x.sexp.numel()) else None
        work = 2.0 * N * Co * Ci * self.conv_valid_pairs(Ho, Wo, in_mul, Hi, Wi, taps) if self.profile is not None else 0.0
        self._rc(self._timed("conv16", work, dev, lambda: self.lib.lk_conv_nhwc_f16x2_planes(
            _ptr(x.planes[0]), _ptr(x.planes[1]), _ptr(x.sexp), x.sexp.numel(), _ptr(amax), N, Hi, Wi, Ci, _ptr(wplanes[0]),
            _ptr(wplanes[1]), _ptr(wsexp), _ptr(w_l1), Co, Ho, Wo, in_mul, len(taps), flat, _ptr(self._zero16(dev)),
            _ptr(planes[0]), _ptr(planes[1]), _ptr(sexp), int(cfg), self._stream(dev))), "lk_conv_nhwc_f16x2_planes")
        return SplitTensor(planes, sexp, chunked=True)

    #: ``False``: the forward's convolutions and their BatchNorm / add / ReLU stay two launches (lk_conv_nhwc_f16x2 +
    #: lk_bn_act_fwd_nhwc_f16x2) instead of one (lk_conv_bn_act_nhwc_f16x2); the results are the same to the bit
    use_conv_bn_act = True

    def conv_bn_act_nhwc(self, x, wplanes, wsexp, w_l1, Ho, Wo, in_mul, taps, scale, shift, scale_amax, shift_amax, act,
                         addend=None, addend_bound=None, want_mask=True, want_split=True, amax_words=None, config=None,
                         y_out=None):
        """(``y_out``: where ``y`` is written — an fp32 NHWC tensor of its shape, e.g. a slot of the consumer's pixel-pair stack)
        lk_conv_bn_act_nhwc_f16x2: ``act(conv(x, W) * scale[c] + shift[c] + addend)`` for the per-image SplitTensor
        ``x [N, Hi, Wi, Ci]`` (``x.amax``: the measured per-image maxima) -> ``(y, mask, split, bound)`` exactly as
        :meth:`conv_nhwc_f16x2` followed by :meth:`bn_act_forward_nhwc` returns them (``y`` fp32 NHWC ``[N, Ho, Wo, Co]``)"""
        N, Hi, Wi, Ci = x.planes.shape[1:]
        Co = wplanes.shape[2]
        dev = x.planes.device
        if x.amax is None or x.amax.numel() not in (1, N) or wplanes.shape[3] != Ci or Co % 8:
            raise LaplaceHipError("conv_bn_act_nhwc: per-image maxima of the input, matching channels, Co % 8 == 0")
        if N > self.MAX_IMAGES_PER_LAUNCH:
            raise LaplaceHipError("conv_bn_act_nhwc: at most %d images per launch" % self.MAX_IMAGES_PER_LAUNCH)
        if y_out is not None and (tuple(y_out.shape) != (N, Ho, Wo, Co) or y_out.dtype != torch.float32 or not y_out.is_contiguous()
                                  or y_out.device != dev):
            raise LaplaceHipError("conv_bn_act_nhwc: y_out must be a contiguous fp32 [N, Ho, Wo, Co] tensor on the input's device")
        y = y_out if y_out is not None else torch.empty((N, Ho, Wo, Co), dtype=torch.float32, device=dev)
        mask = torch.empty(y.shape, dtype=torch.uint8, device=dev) if (act == 1 and want_mask) else None
        planes = torch.empty((2,) + tuple(y.shape), dtype=torch.float16, device=dev) if want_split else None
        sexp = torch.empty(N, dtype=torch.int32, device=dev)
        bound = torch.empty(N, dtype=torch.float32, device=dev)
        amax = amax_words if amax_words is not None else torch.zeros(N, dtype=torch.float32, device=dev)
        if addend is not None:
            _check(addend, "addend")
            if tuple(addend.shape) != tuple(y.shape) or addend_bound is None or addend_bound.numel() not in (1, N):
                raise LaplaceHipError("conv_bn_act_nhwc: the addend has the output's shape and a bound of 1 or N entries")
        if amax.numel() != N:
            raise LaplaceHipError("conv_bn_act_nhwc: amax_words has N entries")
        flat = (ctypes.c_int * (3 * len(taps)))(*[int(v) for t in taps for v in t])
        cfg = self.conv_config if config is None else config
        work = 2.0 * N * Co * Ci * self.conv_valid_pairs(Ho, Wo, in_mul, Hi, Wi, taps) if self.profile is not None else 0.0
        self._rc(self._timed("conv16", work, dev, lambda: self.lib.lk_conv_bn_act_nhwc_f16x2(
            _ptr(x.planes[0]), _ptr(x.planes[1]), _ptr(x.sexp), x.sexp.numel(), _ptr(x.amax), x.amax.numel(), N, Hi, Wi, Ci,
            _ptr(wplanes[0]), _ptr(wplanes[1]), _ptr(wsexp), _ptr(w_l1), Co, Ho, Wo, in_mul, len(taps), flat,
            _ptr(self._zero16(dev)), _ptr(scale), _ptr(shift), _ptr(scale_amax), _ptr(shift_amax), _ptr(addend),
            _ptr(addend_bound), 1 if addend_bound is None else addend_bound.numel(), int(act), _ptr(y), _ptr(mask),
            None if planes is None else _ptr(planes[0]), None if planes is None else _ptr(planes[1]), _ptr(sexp), _ptr(bound),
            _ptr(amax), int(cfg), self._stream(dev))), "lk_conv_bn_act_nhwc_f16x2")
        return y, mask, (SplitTensor(planes, sexp, amax) if planes is not None else None), bound

    #: ``False``: fused 64-channel launches stay on the generic kernel (see :meth:`conv_winp_eligible`)
    use_winp = True

    def conv_winp_eligible(self, N, Hi, Wi, Ci, Co, T, mask_is_float=False) -> bool:
        """does a fused 3 x 3 / stride-1 launch of this shape run the persistent window form (lk_conv_nhwc_f16x2_vjp_wc)?  The
        caller then hands over chunk-major weights (``wplanes_chunked``)"""
        return bool(self.use_winp and self.lib.lk_conv_winp_eligible(int(N), int(Hi), int(Wi), int(Ci), int(Co), int(T),
                                                                     1 if mask_is_float else 0))

    def conv_nhwc_f16x2_vjp(self, x, wplanes, wsexp, w_l1, Ho, Wo, taps, add=None, mult=None, mult_amax=None, scale=None,
                            scale_amax=None, config=None, amax_word=None, wplanes_chunked=None):
        """one dense launch of the convolution with the element-wise VJP fused into its epilogue (lk_conv_nhwc_f16x2_vjp):
        ``(conv(x) + add) * mult * scale[channel]`` -> SplitTensor [N, Ho, Wo, Co] carrying its measured ``amax``.
        ``mult``: [B, Ho, Wo, Co] uint8 / bool mask or fp32 multiplier (``mult_amax``: its bound, fp32 only), shared by
        the N / B seeds; ``add``: SplitTensor of the output's shape; ``w_l1``: device word, see conv.PreparedConv."""
        N, Hi, Wi, Ci = x.planes.shape[1:]
        Co = wplanes.shape[2]
        dev = x.planes.device
        assert wplanes.shape[3] == Ci
        _one_scale(x, "conv_nhwc_f16x2_vjp"), _one_scale(add, "conv_nhwc_f16x2_vjp")
        planes = torch.empty((2, N, Ho, Wo, Co), dtype=torch.float16, device=dev)
        sexp = torch.empty(1, dtype=torch.int32, device=dev)
        amax = amax_word if amax_word is not None else torch.zeros(1, dtype=torch.float32, device=dev)
        m_is_float, mask_rows = 0, 0
        if mult is not None:
            if mult.dtype == torch.bool:
                mult = mult.view(torch.uint8)
            m_is_float = 1 if mult.dtype == torch.float32 else 0
            if (not mult.is_contiguous() or mult.dtype not in (torch.uint8, torch.float32) or mult.dim() != 4
                    or tuple(mult.shape[1:]) != (Ho, Wo, Co) or N % mult.shape[0]):
                raise LaplaceHipError("conv_nhwc_f16x2_vjp: multiplier must be a contiguous [B, Ho, Wo, Co] uint8 / float32 tensor")
            mask_rows = mult.shape[0] * Ho * Wo
        if add is not None and tuple(add.shape) != (N, Ho, Wo, Co):
            raise LaplaceHipError("conv_nhwc_f16x2_vjp: addend shape")
        flat = (ctypes.c_int * (3 * len(taps)))(*[int(v) for t in taps for v in t])
        cfg = self.conv_config if config is None else config
        z = self._zero16(dev)
        work = 2.0 * N * Co * Ci * self.conv_valid_pairs(Ho, Wo, 1, Hi, Wi, taps) if self.profile is not None else 0.0
        # algorithmic bytes: cotangent planes in, result planes out, the addend's planes, the multiplier once per sample, weights
        nbytes = (4.0 * N * Hi * Wi * Ci + 4.0 * N * Ho * Wo * Co * (2 if add is not None else 1)
                  + (mult.numel() * mult.element_size() if mult is not None else 0) + 4.0 * wplanes[0].numel())
        if wplanes_chunked is not None:
            # [2, T, Ci / 16, Co, 16]: the persistent window form where the library finds the launch eligible
            assert wplanes_chunked.shape == (2, wplanes.shape[1], Ci // 16, Co, 16) and wplanes_chunked.is_contiguous()
            # (profile family: the persistent window kernel unless config bit 27 sends the launch to the generic one)
            self._rc(self._timed("conv16" if int(cfg) & (1 << 27) else "convp16", work, dev, lambda: self.lib.lk_conv_nhwc_f16x2_vjp_wc(
                _ptr(x.planes[0]), _ptr(x.planes[1]), _ptr(x.sexp), _ptr(x.amax), N, Hi, Wi, Ci, _ptr(wplanes[0]),
                _ptr(wplanes[1]), _ptr(wsexp), _ptr(w_l1), _ptr(wplanes_chunked[0]), _ptr(wplanes_chunked[1]), Co, Ho, Wo,
                len(taps), flat, _ptr(z), None if add is None else _ptr(add.planes[0]),
                None if add is None else _ptr(add.planes[1]), None if add is None else _ptr(add.sexp), _ptr(mult), m_is_float,
                _ptr(mult_amax), mask_rows, _ptr(scale), _ptr(scale_amax), _ptr(planes[0]), _ptr(planes[1]), _ptr(sexp),
                _ptr(amax), int(cfg), self._stream(dev)), nbytes=nbytes), "lk_conv_nhwc_f16x2_vjp_wc")
            return SplitTensor(planes, sexp, amax)
        self._rc(self._timed("conv16", work, dev, lambda: self.lib.lk_conv_nhwc_f16x2_vjp(
            _ptr(x.planes[0]), _ptr(x.planes[1]), _ptr(x.sexp), _ptr(x.amax), N, Hi, Wi, Ci, _ptr(wplanes[0]), _ptr(wplanes[1]),
            _ptr(wsexp), _ptr(w_l1), Co, Ho, Wo, len(taps), flat, _ptr(z),
            None if add is None else _ptr(add.planes[0]), None if add is None else _ptr(add.planes[1]),
            None if add is None else _ptr(add.sexp), _ptr(mult), m_is_float, _ptr(mult_amax), mask_rows, _ptr(scale),
            _ptr(scale_amax), _ptr(planes[0]), _ptr(planes[1]), _ptr(sexp), _ptr(amax), int(cfg), self._stream(dev)), nbytes=nbytes),
            "lk_conv_nhwc_f16x2_vjp")
        return SplitTensor(planes, sexp, amax)

    def conv_nhwc_f16x2_vjp_strided(self, sources, Ho, Wo, os, taps, add=None, mult=None, mult_amax=None, scale=None,
                                    scale_amax=None, amax_word=None):
        """lk_conv_nhwc_f16x2_vjp_strided: the backward-data of one or two stride-``os`` convolutions that read the same
        input, every residue class in one launch, with the element-wise VJP fused: ``(dX [+ dX2] + add) * mult * scale``
        -> SplitTensor [N, Ho, Wo, Co] carrying its measured ``amax``.  ``sources``: one or two
        ``(g SplitTensor [N, Ho/os, Wo/os, Ci], wplanes [2, T_w, Co, Ci], wsexp, w_l1)``; ``taps``: rows
        ``(dh, dw, weight slice, source, oh0, ow0)``."""
        x, wplanes, wsexp, w_l1 = sources[0]
        N, Hi, Wi, Ci = x.planes.shape[1:]
        Co = wplanes.shape[2]
        dev = x.planes.device
        if len(sources) not in (1, 2) or any(tuple(s_[0].planes.shape) != tuple(x.planes.shape) or s_[1].shape[2:] != wplanes.shape[2:]
                                             for s_ in sources):
            raise LaplaceHipError("conv_nhwc_f16x2_vjp_strided: one or two sources of the same shapes")
        for s_ in sources:
            _one_scale(s_[0], "conv_nhwc_f16x2_vjp_strided")
        _one_scale(add, "conv_nhwc_f16x2_vjp_strided")
        planes = torch.empty((2, N, Ho, Wo, Co), dtype=torch.float16, device=dev)
        sexp = torch.empty(1, dtype=torch.int32, device=dev)
        amax = amax_word if amax_word is not None else torch.zeros(1, dtype=torch.float32, device=dev)
        m_is_float, mask_rows = 0, 0
        if mult is not None:
            if mult.dtype == torch.bool:
                mult = mult.view(torch.uint8)
            m_is_float = 1 if mult.dtype == torch.float32 else 0
            if (not mult.is_contiguous() or mult.dtype not in (torch.uint8, torch.float32) or mult.dim() != 4
                    or tuple(mult.shape[1:]) != (Ho, Wo, Co) or N % mult.shape[0]):
                raise LaplaceHipError("conv_nhwc_f16x2_vjp_strided: multiplier must be a contiguous [B, Ho, Wo, Co] uint8 / float32 tensor")
            mask_rows = mult.shape[0] * Ho * Wo
        if add is not None and tuple(add.shape) != (N, Ho, Wo, Co):
            raise LaplaceHipError("conv_nhwc_f16x2_vjp_strided: addend shape")
        flat = (ctypes.c_int * (6 * len(taps)))(*[int(v) for t in taps for v in t])
        work = 0.0
        if self.profile is not None:
            classes = {}
            for t in taps:
                classes.setdefault((t[4], t[5]), []).append(t)
            work = 2.0 * N * Co * Ci * sum(self.conv_valid_pairs(Ho // os, Wo // os, 1, Hi, Wi, ts) for ts in classes.values())
        src = []
        for i in range(2):
            if i < len(sources):
                g_, wp_, ws_, l1_ = sources[i]
                src += [_ptr(g_.planes[0]), _ptr(g_.planes[1]), _ptr(g_.sexp), _ptr(g_.amax), _ptr(wp_[0]), _ptr(wp_[1]), _ptr(ws_), _ptr(l1_)]
            else:
                src += [None] * 8
        self._rc(self._timed("convs16", work, dev, lambda: self.lib.lk_conv_nhwc_f16x2_vjp_strided(
            *src, N, Hi, Wi, Ci, Co, Ho, Wo, os, len(taps), flat, _ptr(self._zero16(dev)),
            None if add is None else _ptr(add.planes[0]), None if add is None else _ptr(add.planes[1]),
            None if add is None else _ptr(add.sexp), _ptr(mult), m_is_float, _ptr(mult_amax), mask_rows, _ptr(scale),
            _ptr(scale_amax), _ptr(planes[0]), _ptr(planes[1]), _ptr(sexp), _ptr(amax), int(self.conv_config),
            self._stream(dev))), "lk_conv_nhwc_f16x2_vjp_strided")
        return SplitTensor(planes, sexp, amax)

    def vjp_nhwc_split(self, g, g_amax, g2, mult, mult_amax, scale, scale_amax, S, out_shape):
        """``(g + g2) * mult * scale[channel]`` for all ``S`` seeds -> SplitTensor of ``out_shape`` ([S*B, H, W, C]).
        ``g``: fp32 NHWC with ``g_amax`` (device word, lk_absmax / conv epilogue) or None; ``g2``: SplitTensor or None;
        ``mult``: [B, H, W, C] uint8 / bool mask or fp32 multiplier (``mult_amax`` for fp32) or None."""
        dev = (g if g is not None else g2.planes).device
        _one_scale(g2, "vjp_nhwc_split")
        planes = torch.empty((2,) + tuple(out_shape), dtype=torch.float16, device=dev)
        sexp = torch.empty(1, dtype=torch.int32, device=dev)
        n = planes[0].numel()
        per = n // S
        C = out_shape[-1]
        m_is_float = 0
        if mult is not None:
            if mult.dtype == torch.bool:
                mult = mult.view(torch.uint8)
            m_is_float = 1 if mult.dtype == torch.float32 else 0
            if not mult.is_contiguous() or mult.numel() != per or mult.dtype not in (torch.uint8, torch.float32):
                raise LaplaceHipError("vjp_nhwc_split: multiplier must be a contiguous [B, H, W, C] uint8 / float32 tensor")
        if g is not None:
            _check(g, "g")
        # HBM-bound: algorithmic bytes = every addend read once, the planes written once, the multiplier once per sample
        nbytes = 4.0 * n * ((g is not None) + (g2 is not None) + 1) + (per * (4 if m_is_float else 1) if mult is not None else 0)
        self._rc(self._timed("vjp16", nbytes, dev, lambda: self.lib.lk_vjp_nhwc_split_f16x2(
            _ptr(g), _ptr(g_amax), None if g2 is None else _ptr(g2.planes[0]), None if g2 is None else _ptr(g2.planes[1]),
            None if g2 is None else _ptr(g2.sexp), _ptr(mult), m_is_float, _ptr(mult_amax), _ptr(scale), _ptr(scale_amax),
            C, S, per, _ptr(planes[0]), _ptr(planes[1]), _ptr(sexp), self._stream(dev))), "lk_vjp_nhwc_split_f16x2")
        return SplitTensor(planes, sexp)

    def bn_act_forward_nhwc(self, x, x_amax, scale, shift, scale_amax, shift_amax, act, addend=None, addend_bound=None,
                            want_mask=True, want_split=True, x_mul=None, x_add=None, amax_words=None):
        """``y = act(x * scale[c] + shift[c] + addend)`` on fp32 NHWC ``x [B, H, W, C]`` -> ``(y, mask, split, bound)``:
        ``mask`` NHWC uint8 (ReLU only), ``split`` the SplitTensor of ``y`` with ONE SCALE PER IMAGE and its measured
        per-image maxima as ``split.amax`` (``[B]``), ``bound [B]`` the guaranteed bounds the scales were derived from.
        ``x_amax``: 1 or B words bounding ``x`` per image after ``* x_mul[0] + x_add[0]`` (device words; a convolution's
        output: the measured maxima of its input images, the l1 norm of its weights, max|bias|); ``addend_bound``: 1 or B
        floats; ``amax_words``: zeroed ``[B]`` float32 buffer for the measured maxima (allocated if absent)."""
        _check(x, "x")
        C = x.shape[-1]
        B = x.shape[0]
        y = torch.empty_like(x)
        mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if (act == 1 and want_mask) else None
        planes = torch.empty((2,) + tuple(x.shape), dtype=torch.float16, device=x.device) if want_split else None
        sexp = torch.empty(B, dtype=torch.int32, device=x.device)
        bound = torch.empty(B, dtype=torch.float32, device=x.device)
        amax = amax_words if amax_words is not None else torch.zeros(B, dtype=torch.float32, device=x.device)
        if addend is not None:
            _check(addend, "addend")
        if x_amax.numel() not in (1, B) or (addend is not None and addend_bound.numel() not in (1, B)) or amax.numel() != B:
            raise LaplaceHipError("bn_act_forward_nhwc: x_amax / addend_bound have 1 or B entries, amax_words B")
        nbytes = x.numel() * (4.0 + 4.0 + (4.0 if addend is not None else 0.0) + (4.0 if want_split else 0.0)
                              + (1.0 if mask is not None else 0.0))
        per = x.numel() // max(B, 1)
        sl = lambda t, i, n: None if t is None else (t if t.numel() == 1 else t[i:i + n])  # a per-image vector, or one word

        def launch(i, n):
            return self.lib.lk_bn_act_fwd_nhwc_f16x2(
                _ptr(x[i:i + n]), _ptr(sl(x_amax, i, n)), min(x_amax.numel(), n), _ptr(x_mul), _ptr(x_add), _ptr(scale), _ptr(shift),
                _ptr(scale_amax), _ptr(shift_amax), None if addend is None else _ptr(addend[i:i + n]), _ptr(sl(addend_bound, i, n)),
                1 if addend_bound is None else min(addend_bound.numel(), n), int(act), C, n, per, _ptr(y[i:i + n]),
                None if mask is None else _ptr(mask[i:i + n]), None if planes is None else _ptr(planes[0][i:i + n]),
                None if planes is None else _ptr(planes[1][i:i + n]), _ptr(sexp[i:i + n]), _ptr(bound[i:i + n]), _ptr(amax[i:i + n]),
                self._stream(x.device))

        if B <= self.MAX_IMAGES_PER_LAUNCH:
            self._rc(self._timed("bnact16", nbytes, x.device, lambda: launch(0, B)), "lk_bn_act_fwd_nhwc_f16x2")
        else:  # (the image index is a grid dimension: 65535 per launch)
            for i in range(0, B, self.MAX_IMAGES_PER_LAUNCH):
                self._rc(launch(i, min(self.MAX_IMAGES_PER_LAUNCH, B - i)), "lk_bn_act_fwd_nhwc_f16x2")
        return y, mask, (SplitTensor(planes, sexp, amax) if planes is not None else None), bound

    def unsplit_transpose(self, x, S, B):
        """SplitTensor ``[S*B, H, W, C]`` (seed-major) -> fp32 ``[B, S, C, H*W]``"""
        N, H, W, C = x.shape
        assert N == S * B
        _one_scale(x, "unsplit_transpose")
        out = torch.empty(B, S, C, H * W, dtype=torch.float32, device=x.planes.device)
        self._rc(self.lib.lk_unsplit_transpose_f32(_ptr(x.planes[0]), _ptr(x.planes[1]), _ptr(x.sexp), S, B, H * W, C,
                                                   _ptr(out), self._stream(out.device)), "lk_unsplit_transpose_f32")
        return out

    def gram_tn_f16x2(self, x, alpha, out):
        """``out[upper tiles] += alpha * X^T X`` for a SplitTensor ``x`` viewed as ``[rows, C]``"""
        _check(out, "out")
        _one_scale(x, "gram_tn_f16x2")
        C = x.planes.shape[-1]
        R = x.planes[0].numel() // C
        nbytes = int(self.lib.lk_gram_tn_f16x2_workspace_bytes(R, C))
        if nbytes == 0 and R:
            raise LaplaceHipError(f"gram_tn_f16x2: C = {C} not supported (64 or a multiple of 128)")
        ws = self._workspace(nbytes, out.device)
        z = self._zero16(out.device)
        self._rc(self._timed("gram16", float(R) * C * (C + 1), out.device, lambda: self.lib.lk_gram_tn_f16x2(
            _ptr(x.planes[0]), _ptr(x.planes[1]), _ptr(x.sexp), R, C, float(alpha), _ptr(out), _ptr(z), _ptr(ws),
            ws.numel(), self._stream(out.device))), "lk_gram_tn_f16x2")
        return out

    # ---- packed upper triangles (lk_pack.hip) ---------------------------------------------------------------
    def pack_upper(self, A, packed):
        """upper triangle of the square ``A`` -> ``packed [n (n + 1) / 2]`` (a slice of the exchange buffer)"""
        _check(A, "A"), _check(packed, "packed")
        self._rc(self.lib.lk_pack_upper_f32(_ptr(A), A.shape[0], _ptr(packed), self._stream(A.device)), "lk_pack_upper_f32")

    def unpack_upper(self, packed, A):
        _check(A, "A"), _check(packed, "packed")
        self._rc(self.lib.lk_unpack_upper_f32(_ptr(packed), A.shape[0], _ptr(A), self._stream(A.device)), "lk_unpack_upper_f32")

    # ---- eigenbasis algebra of KronDecomposed (lk_gemm.hip) ------------------------------------------------
    def gemm(self, A, B, C, batch, M, N, Kd, lda, ldb, ldc, sa=0, sb=0, sc=0, ta=False, tb=False, E=None, lde=0,
             alpha=1.0, accumulate=False):
        """raw batched ``C[b] = alpha op(A[b]) op(B[b]) (.) E`` on device pointers / offsets of fp32 tensors (see
        lk_gemm_f32); ``A``, ``B``, ``C``, ``E`` are tensors whose ``data_ptr()`` is the first element"""
        self._rc(self.lib.lk_gemm_f32(_ptr(A), _ptr(B), _ptr(E), _ptr(C), batch, M, N, Kd, lda, ldb, ldc, lde, sa, sb, sc,
                                      1 if ta else 0, 1 if tb else 0, float(alpha), 1 if accumulate else 0,
                                      self._stream(C.device)), "lk_gemm_f32")

    def kron_pow(self, l1, l2, delta, exponent, damping=False):
        """``(l1 (x) l2 + delta) ** exponent`` as an ``[n1, n2]`` table (``l2`` None: ``[n1]``)"""
        _check(l1, "l1")
        n1 = l1.numel()
        n2 = 0 if l2 is None else l2.numel()
        lam = torch.empty((n1, n2) if l2 is not None else (n1,), dtype=torch.float32, device=l1.device)
        d = delta.detach().reshape(-1)[:1].to(torch.float32).contiguous()
        self._rc(self.lib.lk_kron_pow_f32(_ptr(l1), n1, _ptr(l2), n2, _ptr(d), float(exponent), 1 if damping else 0,
                                          _ptr(lam), self._stream(l1.device)), "lk_kron_pow_f32")
        return lam

    def kron_sandwich(self, W, col0, P, R, Q1, Q2, lam, out):
        """``out[r, col0:col0+p] = vec(Q1 ((Q1^T W_r Q2) (.) lam) Q2^T)`` for the ``R`` rows of ``W [R, P]`` (block of
        ``p = p_in * p_out`` columns at ``col0``); ``Q2`` None: the single-factor block ``((W_r Q1) (.) lam) Q1^T``.
        Operands are read and written in place (leading dimension ``P``)."""
        _check(W, "W"), _check(out, "out"), _check(Q1, "Q1"), _check(lam, "lam")
        Wv, Ov = W.reshape(-1)[col0:], out.reshape(-1)[col0:]
        if Q2 is None:
            p = Q1.shape[0]
            T = torch.empty(R, p, dtype=torch.float32, device=W.device)
            self.gemm(Wv, Q1, T, 1, R, p, p, P, p, p, E=lam, lde=0)
            self.gemm(T, Q1, Ov, 1, R, p, p, p, p, P, tb=True)
            return out
        _check(Q2, "Q2")
        pi, po = Q1.shape[0], Q2.shape[0]
        T = torch.empty(R, pi, po, dtype=torch.float32, device=W.device)
        U = torch.empty(R, pi, po, dtype=torch.float32, device=W.device)
        self.gemm(Wv, Q2, T, R, pi, po, po, po, po, po, sa=P, sc=pi * po)                       # T_r = W_r Q2
        self.gemm(Q1, T, U, R, pi, po, pi, pi, po, po, sb=pi * po, sc=pi * po, ta=True, E=lam, lde=po)  # (Q1^T T_r) . lam
        self.gemm(Q1, U, T, R, pi, po, pi, pi, po, po, sb=pi * po, sc=pi * po)                  # Q1 M_r
        self.gemm(T, Q2, Ov, R, pi, po, po, po, po, po, sa=pi * po, sc=P, tb=True)              # ... Q2^T -> out
        return out

    def gram_conv(self, x, kernel_size, stride, padding, dilation, alpha, out, upper_only=False, native=False):
        """``out += alpha * unfold(x)^T unfold(x)`` for an NCHW input ``x`` — without materialising unfold.

        ``native=True`` leaves ``out`` in the kernel's (kh, kw, ci) column order (fused accumulation);
        otherwise the result is delivered in F.unfold's (ci, kh, kw) order.
        """
        _check(out, "out")
        if not self.is_channels_last(x):
            _check(x, "x")
        B, Cin, H, W = x.shape
        kh, kw = _pair(kernel_size)
        sh, sw = _pair(stride)
        ph, pw = _pair(padding)
        dh, dw = _pair(dilation)
        n = Cin * kh * kw
        assert out.shape == (n, n)
        OH = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
        OW = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
        xh = self.nchw_to_nhwc(x)
        direct = native or kh * kw == 1
        tgt = out if direct else torch.zeros(n, n, dtype=torch.float32, device=x.device)
        if self.use_shiftcorr and (kh, kw, sh, sw, ph, pw, dh, dw) == (3, 3, 1, 1, 1, 1, 1, 1) and H * W >= 256 and B > 0:
            # 3x3 / stride 1 / pad 1 on maps >= 16x16: 13 shift correlations + boundary strips instead of 81 patch
            # blocks (measured 1.7x at 32x32, 1.2x at 16x16, a loss at 8x8: profiles/r01_microbench_gram_v4.txt)
            nb = self.lib.lk_conv3x3_shiftcorr_workspace_bytes(B, H, W, Cin)
            ws = self._workspace(nb, x.device)
            self._rc(
                self._timed("shiftcorr", float(B * H * W) * n * (n + 1), x.device, lambda: self.lib.lk_conv3x3_shiftcorr_f32(
                    _ptr(xh), B, H, W, Cin, float(alpha), _ptr(tgt), _ptr(ws), ws.numel(), self._stream(x.device))),
                "lk_conv3x3_shiftcorr_f32",
            )
            if not direct:
                self._rc(self.lib.lk_permute_sym_f32(_ptr(tgt), Cin, kh * kw, _ptr(out), 1, self._stream(x.device)),
                         "lk_permute_sym_f32")
            return out
        nb = self.lib.lk_gram_workspace_bytes(n, max(B * OH * OW, 1))
        ws = self._workspace(nb, x.device)
        self._rc(
            self._timed("gram_conv", float(B * OH * OW) * n * (n + 1), x.device, lambda: self.lib.lk_gram_conv_nhwc_f32(
                _ptr(xh), B, H, W, Cin, kh, kw, sh, sw, ph, pw, dh, dw, float(alpha), _ptr(tgt),
                LK_GRAM_UPPER_ONLY if upper_only else 0, _ptr(ws), ws.numel(), self._stream(x.device))),
            "lk_gram_conv_nhwc_f32",
        )
        if not direct:
            self._rc(self.lib.lk_permute_sym_f32(_ptr(tgt), Cin, kh * kw, _ptr(out), 1, self._stream(x.device)),
                     "lk_permute_sym_f32")
        return out

    # ---- pixel-pair form of the 3x3/s1/p1 conv A factor on small maps (see lk_gram.hip) --------------------------
    #: largest H*W for which KronAccumulator keeps a pixel-pair accumulator instead of calling gram_conv per step
    pixgram_max_hw = 16

    def pixgram_accumulate(self, x, alpha, Cp):
        """``Cp += alpha * X^T X`` with ``X`` the NHWC images flattened to rows ``[B, H*W*Cin]`` (upper tiles only)."""
        _check(Cp, "Cp")
        B = x.shape[0]
        xh = self.nchw_to_nhwc(x)
        return self.gram_tn(xh.reshape(B, -1), alpha, Cp, upper_only=True)

    def pixgram_assemble(self, Cp, H, W, Cin, alpha, A_native):
        """``A_native += alpha * assemble(Cp)`` (native (kh,kw,ci) order); symmetrises ``Cp`` in place first."""
        _check(Cp, "Cp"), _check(A_native, "A")
        self.symmetrize(Cp)
        self._rc(self.lib.lk_conv3x3_pixgram_assemble_f32(_ptr(Cp), int(H), int(W), int(Cin), float(alpha), _ptr(A_native),
                                                          self._stream(Cp.device)), "lk_conv3x3_pixgram_assemble_f32")
        return A_native

    # ---- banded pixel-pair form (any map size, Cin % 64 == 0) -------------------------------------------------------
    def pixpair_plan(self, H, W, Cin, dev):
        """(n_blocks, tiles_dev, slots_dev) for a geometry; tables are built by the library on the host and uploaded
        once per (geometry, device).  None if the geometry is not eligible (Cin % 64 != 0 or too large)."""
        key = (int(H), int(W), int(Cin), str(dev))
        cache = self.__dict__.setdefault("_pixpair_cache", {})
        if key not in cache:
            T, nt, nb = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
            plan = None
            if Cin >= 64 and Cin % 64 == 0 and self.lib.lk_conv3x3_pixpair_plan(
                    int(H), int(W), int(Cin), ctypes.byref(T), ctypes.byref(nt), ctypes.byref(nb)) == 0:
                tiles = torch.empty(nt.value, 3, dtype=torch.int32)
                slots = torch.empty(int(H) * int(W), 13, dtype=torch.int32)
                self._rc(self.lib.lk_conv3x3_pixpair_tables(int(H), int(W), int(Cin), ctypes.c_void_p(tiles.data_ptr()),
                                                            ctypes.c_void_p(slots.data_ptr())), "lk_conv3x3_pixpair_tables")
                plan = (nb.value, tiles.to(dev), slots.to(dev))
            cache[key] = plan
        return cache[key]

    def pixpair_accumulate(self, x, alpha, blocks, plan):
        """``Blk[q, D] += alpha * sum_b x[b,q,:]^T x[b,q+D,:]`` for an NCHW input ``x``."""
        return self.pixpair_accumulate_nhwc(self.nchw_to_nhwc(x), alpha, blocks, plan)

    def pixpair_accumulate_nhwc(self, xh, alpha, blocks, plan):
        """Same for an NHWC input ``xh [B, H, W, Cin]`` (several minibatches may be stacked along ``B``: the blocks
        are read-modified-written once per launch, whatever the number of rows)."""
        _check(xh, "xh"), _check(blocks, "blocks")
        B, H, W, Cin = xh.shape
        nb, tiles, _ = plan
        assert blocks.numel() == nb * Cin * Cin
        x = xh
        # HBM-bound (each block is read-modified-written once per launch): work = algorithmic BYTES
        self._rc(self._timed("pixpair", 8.0 * blocks.numel() + 4.0 * x.numel(), x.device,
                             lambda: self.lib.lk_conv3x3_pixpair_accumulate_f32(
                                 _ptr(xh), B, H, W, Cin, float(alpha), _ptr(blocks), ctypes.c_void_p(tiles.data_ptr()),
                                 tiles.shape[0], self._stream(x.device))), "lk_conv3x3_pixpair_accumulate_f32")
        return blocks

    #: ``False``: the pixel-pair products stay on the exact-fp32 MFMA kernel.  The split-fp16 kernel
    #: requests the block it read-modify-writes at the START of its (short) tile: 0.87 vs 1.39 ms per c4 step — the
    #: dependent 16 KB read at the end of every 8-stage tile, not traffic or the matrix pipe, was what bound the product
    use_pixpair16 = True
    #: 64-channel maps: one workgroup per pixel with all 13 shifts (`lk_conv3x3_pixpair_accumulate13_f16x2`: the pixel's panel is
    #: staged once instead of 13 times); ``False``: one workgroup per (pixel, shift) block as for the wider maps
    use_pixpair13 = True

    def pixpair_accumulate_split(self, xs, alpha, blocks, plan):
        """:meth:`pixpair_accumulate_nhwc` on a SplitTensor ``xs [B, H, W, Cin]`` (three fp16 MFMAs per product block)."""
        _check(blocks, "blocks")
        _one_scale(xs, "pixpair_accumulate_split")
        B, H, W, Cin = xs.shape
        nb, tiles, slots = plan
        assert blocks.numel() == nb * Cin * Cin
        dev = blocks.device
        z = self._zero16(dev)
        if Cin == 64 and self.use_pixpair13:
            self._rc(self._timed("pixpair16", 8.0 * blocks.numel() + 4.0 * xs.planes[0].numel(), dev,
                                 lambda: self.lib.lk_conv3x3_pixpair_accumulate13_f16x2(
                                     _ptr(xs.planes[0]), _ptr(xs.planes[1]), _ptr(xs.sexp), B, H, W, Cin, float(alpha), _ptr(blocks),
                                     ctypes.c_void_p(slots.data_ptr()), _ptr(z), self._stream(dev))),
                     "lk_conv3x3_pixpair_accumulate13_f16x2")
            return blocks
        self._rc(self._timed("pixpair16", 8.0 * blocks.numel() + 4.0 * xs.planes[0].numel(), dev,
                             lambda: self.lib.lk_conv3x3_pixpair_accumulate_f16x2(
                                 _ptr(xs.planes[0]), _ptr(xs.planes[1]), _ptr(xs.sexp), B, H, W, Cin, float(alpha), _ptr(blocks),
                                 ctypes.c_void_p(tiles.data_ptr()), tiles.shape[0], _ptr(z), self._stream(dev))),
                 "lk_conv3x3_pixpair_accumulate_f16x2")
        return blocks

    def pixpair_assemble(self, blocks, plan, H, W, Cin, alpha, A_native, blocks2=None, upper_only=False):
        """``A_native += alpha * assemble(blocks [+ blocks2])``: ``blocks2`` is a second accumulator set of the same geometry
        (the other lane of a fit), summed on the fly; ``upper_only``: only the upper triangle of ``A_native`` is written (what
        :meth:`finalize_factors` and the packed exchange read)"""
        _check(blocks, "blocks"), _check(A_native, "A")
        if blocks2 is not None:
            _check(blocks2, "blocks2")
            assert blocks2.numel() == blocks.numel()
        self._rc(self.lib.lk_conv3x3_pixpair_assemble2_f32(_ptr(blocks), _ptr(blocks2), ctypes.c_void_p(plan[2].data_ptr()),
                                                           int(H), int(W), int(Cin), float(alpha), _ptr(A_native),
                                                           1 if upper_only else 0, self._stream(blocks.device)),
                 "lk_conv3x3_pixpair_assemble2_f32")
        return A_native

    def permute_native_to_unfold(self, src, Cin, KK, dst, accumulate=False):
        _check(src, "src"), _check(dst, "dst")
        self._rc(self.lib.lk_permute_sym_f32(_ptr(src), Cin, KK, _ptr(dst), 1 if accumulate else 0, self._stream(src.device)),
                 "lk_permute_sym_f32")
        return dst

    def finalize_factors(self, items):
        """The once-per-fit layout pass of many factors in one launch.  ``items``: ``(src, dst, scale, cin, kk)`` per
        factor — ``kk <= 1``: mirror the upper triangle (in place when ``dst`` is None), after ``diag(scale) C diag(scale)``
        when ``scale`` is given; ``kk > 1``: ``dst`` = the full matrix in unfold order from the native-order upper triangle
        of ``src`` (what :meth:`symmetrize` + :meth:`permute_native_to_unfold` do with two launches per factor)."""
        items = [it for it in items if it[0].numel()]
        if not items:
            return
        n = len(items)
        dev = items[0][0].device
        src = (ctypes.c_void_p * n)()
        dst = (ctypes.c_void_p * n)()
        scl = (ctypes.c_void_p * n)()
        ns = (ctypes.c_int64 * n)()
        cins = (ctypes.c_int64 * n)()
        kks = (ctypes.c_int64 * n)()
        for i, (s_, d_, sc_, cin, kk) in enumerate(items):
            _check(s_, "src")
            assert s_.dim() == 2 and s_.shape[0] == s_.shape[1] and s_.device == dev
            src[i] = s_.data_ptr()
            if d_ is not None:
                _check(d_, "dst")
                assert d_.shape == s_.shape
                dst[i] = d_.data_ptr()
            if sc_ is not None:
                _check(sc_, "scale")
                assert sc_.numel() == s_.shape[0]
                scl[i] = sc_.data_ptr()
            ns[i], cins[i], kks[i] = s_.shape[0], int(cin), int(kk)
        self._rc(self.lib.lk_finalize_factors_f32(n, src, dst, scl, ns, cins, kks, self._stream(dev)), "lk_finalize_factors_f32")

    def symmetrize(self, C):
        _check(C, "C")
        self._rc(self.lib.lk_symmetrize_f32(_ptr(C), C.shape[0], self._stream(C.device)), "lk_symmetrize_f32")
        return C

    # ---- diag / Jacobians ---------------------------------------------------------------------
    def diag_ggn_linear(self, a, g, alpha, h_w, h_b=None):
        _check(a, "a"), _check(g, "g"), _check(h_w, "h_w")
        Cc, B, Do = g.shape
        Di = a.shape[1]
        assert a.shape[0] == B and h_w.numel() == Do * Di
        if h_b is not None:
            _check(h_b, "h_b")
        self._rc(
            self.lib.lk_diag_ggn_linear_f32(_ptr(a), _ptr(g), B, Cc, Di, Do, float(alpha), _ptr(h_w), _ptr(h_b),
                                            self._stream(a.device)),
            "lk_diag_ggn_linear_f32",
        )

    def jac_linear(self, a, g, Js, col0, bcol0=-1):
        _check(a, "a"), _check(g, "g"), _check(Js, "Js")
        Cc, B, Do = g.shape
        Di = a.shape[1]
        P = Js.shape[-1]
        self._rc(
            self.lib.lk_jac_linear_f32(_ptr(a), _ptr(g), B, Cc, Di, Do, _ptr(Js), P, int(col0), int(bcol0),
                                       self._stream(a.device)),
            "lk_jac_linear_f32",
        )

    def jac_conv(self, x, g, kernel_size, stride, padding, dilation, Js, col0, bcol0=-1):
        _check(x, "x"), _check(g, "g"), _check(Js, "Js")
        B, Cin, H, W = x.shape
        Cc, _, Do = g.shape[:3]
        kh, kw = _pair(kernel_size)
        sh, sw = _pair(stride)
        ph, pw = _pair(padding)
        dh, dw = _pair(dilation)
        P = Js.shape[-1]
        self._rc(
            self.lib.lk_jac_conv_f32(_ptr(x), _ptr(g), B, Cc, Cin, H, W, Do, kh, kw, sh, sw, ph, pw, dh, dw, _ptr(Js), P,
                                     int(col0), int(bcol0), self._stream(x.device)),
            "lk_jac_conv_f32",
        )

    def sq_colsum(self, Js, col0, width, alpha, h):
        _check(Js, "Js"), _check(h, "h")
        P = Js.shape[-1]
        rows = Js.numel() // P
        self._rc(self.lib.lk_sq_colsum_f32(_ptr(Js), rows, P, int(col0), int(width), float(alpha), _ptr(h),
                                           self._stream(Js.device)), "lk_sq_colsum_f32")

    def bn_act_forward(self, x, scale, shift, relu, addend=None, want_mask=True):
        """``(y, mask)``: ``y = act(x * scale[c] + shift[c] + addend)`` for ``x`` [B, C, ...]; ``mask = y > 0`` (bool) if
        ``relu``."""
        _check(x, "x")
        if addend is not None:
            _check(addend, "addend")
            if addend.shape != x.shape:
                raise ValueError("bn_act_forward: addend must have the shape of x")
        C = x.shape[1]
        hw = x.numel() // (x.shape[0] * C) if x.numel() else 1
        y = torch.empty_like(x)
        mask = torch.empty(x.shape, dtype=torch.bool, device=x.device) if (relu and want_mask) else None
        self._rc(self.lib.lk_bn_act_fwd_f32(_ptr(x), _ptr(scale.to(torch.float32).contiguous()),
                                            _ptr(shift.to(torch.float32).contiguous()), _ptr(addend), x.numel(), C, max(hw, 1),
                                            1 if relu else 0, _ptr(y), _ptr(mask), self._stream(x.device)),
                 "lk_bn_act_fwd_f32")
        return y, mask

    def vjp_scale_mask(self, g, S, mult, scale, hw, g2=None):
        """``out[s, e] = (g[s, e] + g2[s, e]) * mult[e] * scale[channel(e)]`` for the ``S`` seeds stacked in ``g``
        ([S*B, ...]); ``g2``, ``mult`` ([B, ...], bool or float32) and ``scale`` ([C]) may be None."""
        _check(g, "g")
        if g2 is not None:
            _check(g2, "g2")
            if g2.shape != g.shape:
                raise ValueError("vjp_scale_mask: g2 must have the shape of g")
        per = g.numel() // S
        out = torch.empty_like(g)
        m_is_float = 0
        if mult is not None:
            if mult.dtype == torch.bool:
                mult = mult.contiguous()
            else:
                mult, m_is_float = mult.to(torch.float32).contiguous(), 1
            if mult.numel() != per:
                raise ValueError("vjp_scale_mask: multiplier must have the per-seed shape")
        C = 1
        if scale is not None:
            scale = scale.to(torch.float32).contiguous()
            C = scale.numel()
        self._rc(self.lib.lk_vjp_scale_mask_f32(_ptr(g), _ptr(g2), _ptr(mult), m_is_float,
                                                _ptr(scale), int(S), per, C, int(hw),
                                                _ptr(out), self._stream(g.device)), "lk_vjp_scale_mask_f32")
        return out

    # ---- dense last-layer GGN -----------------------------------------------------------------
    def ll_ggn_full(self, phi, probs, has_bias, alpha, H):
        _check(phi, "phi"), _check(H, "H")
        B, D = phi.shape
        if probs is not None:
            _check(probs, "probs")
            C = probs.shape[1]
        else:
            C = (H.shape[0]) // (D + (1 if has_bias else 0))
        nb = self.lib.lk_ll_ggn_workspace_bytes(B, C, D)
        ws = self._workspace(nb, phi.device)
        Dt = D + (1 if has_bias else 0)
        # structured form: C block-diagonal Grams of sqrt(p_j) phi~ minus the Gram of Y = [p_j phi~]_j — about 2 C^2 Dt^2 flop per
        # sample against the 2 C^3 Dt^2 of J^T Lambda J as the reference writes it (curvature.py:375-411)
        self._rc(self._timed("llggn", 2.0 * B * C * C * Dt * Dt, phi.device, lambda: self.lib.lk_ll_ggn_full_f32(
            _ptr(phi), _ptr(probs), B, C, D, 1 if has_bias else 0, float(alpha), _ptr(H), _ptr(ws), ws.numel(),
            self._stream(phi.device)), nbytes=4.0 * (B * D + B * C + (C * Dt) ** 2)), "lk_ll_ggn_full_f32")
        return H

    # ---- eigensolver --------------------------------------------------------------------------
    def syevj(self, A, clamp=True, max_sweeps=0):
        _check(A, "A")
        n = A.shape[0]
        assert A.shape == (n, n)
        w = torch.empty(n, dtype=torch.float32, device=A.device)
        Q = torch.empty(n, n, dtype=torch.float32, device=A.device)
        info = torch.zeros(2, dtype=torch.int32, device=A.device)
        nb = self.lib.lk_syevj_workspace_bytes(n)
        # the solve is long-running and asynchronous: give it a private workspace
        ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=A.device)
        self._rc(
            self.lib.lk_syevj_f32(_ptr(A), n, _ptr(w), _ptr(Q), 1 if clamp else 0, int(max_sweeps), _ptr(info), _ptr(ws),
                                  ws.numel(), self._stream(A.device)),
            "lk_syevj_f32",
        )
        ws.record_stream(torch.cuda.current_stream(A.device))
        return w, Q, info

    def syevj_batched(self, mats, clamp=True, max_sweeps=0, streams=None):
        """Eigendecompose a list of symmetric matrices (largest first is best) -> list of (w, Q, info).

        ``streams``: torch side streams for the solve — with three or more, the matrices iterate in two lanes of rounds
        (``streams[0]``, ``streams[1]``: the pivot solves of one beside the tile updates of the other) and finished
        matrices are refined on ``streams[2]``; with two, one lane + the refinements; further streams are only made to
        wait.  The caller orders them against its own stream (``wait_stream`` before / after).  ``None`` = the current
        stream."""
        if not mats:
            return []
        dev = mats[0].device
        cur = torch.cuda.current_stream(dev)
        streams = list(streams) if streams else [cur]
        count = len(mats)
        outs, keep = [], []
        arr = lambda ctype, vals: (ctype * count)(*vals)  # noqa: E731
        ns = []
        for i, A in enumerate(mats):
            _check(A, "A")
            n = A.shape[0]
            assert A.shape == (n, n) and A.device == dev
            w = torch.empty(n, dtype=torch.float32, device=dev)
            Q = torch.empty(n, n, dtype=torch.float32, device=dev)
            info = torch.zeros(2, dtype=torch.int32, device=dev)
            ws = torch.empty(max(self.lib.lk_syevj_workspace_bytes(n), 1), dtype=torch.uint8, device=dev)
            outs.append((w, Q, info))
            keep.append(ws)
            ns.append(n)
        nstreams = min(len(streams), count)
        used = streams[:nstreams]
        # `info` was zero-filled on the current stream: the side streams must see that (and the inputs) first
        for st in used:
            if st != cur:
                st.wait_stream(cur)
        rc = self.lib.lk_syevj_batched_f32(
            count, arr(ctypes.c_void_p, [A.data_ptr() for A in mats]), arr(ctypes.c_int64, ns),
            arr(ctypes.c_void_p, [o[0].data_ptr() for o in outs]), arr(ctypes.c_void_p, [o[1].data_ptr() for o in outs]),
            arr(ctypes.c_void_p, [o[2].data_ptr() for o in outs]), arr(ctypes.c_void_p, [w_.data_ptr() for w_ in keep]),
            arr(ctypes.c_size_t, [w_.numel() for w_ in keep]), 1 if clamp else 0, int(max_sweeps),
            arr(ctypes.c_void_p, [st.cuda_stream for st in used]), nstreams)
        self._rc(rc, "lk_syevj_batched_f32")
        for st in used[:3]:
            if st != cur:  # allocated on `cur`, used on `st`
                for A, o, ws in zip(mats, outs, keep):
                    for t in (A, ws) + o:
                        t.record_stream(st)
        return outs

    # ---- logdet -------------------------------------------------------------------------------
    def kron_logdet(self, l1, l2, delta, damping=False, want_grads=False):
        """Returns (value[1], d_l1, d_l2, d_delta[1]); derivative tensors are None unless requested."""
        _check(l1, "l1"), _check(delta, "delta")
        n1 = l1.numel()
        n2 = 0 if l2 is None else l2.numel()
        if l2 is not None:
            _check(l2, "l2")
        dev = l1.device
        out = torch.zeros(1, dtype=torch.float32, device=dev)
        d1 = d2 = dd = None
        if want_grads:
            d1 = torch.zeros(n1, dtype=torch.float32, device=dev)
            d2 = torch.zeros(n2, dtype=torch.float32, device=dev) if n2 else None
            dd = torch.zeros(1, dtype=torch.float32, device=dev)
        nb = self.lib.lk_kron_logdet_workspace_bytes(n1)
        ws = self._workspace(nb, dev)
        self._rc(
            self.lib.lk_kron_logdet_f32(_ptr(l1), n1, _ptr(l2), n2, _ptr(delta), 1 if damping else 0, _ptr(out), _ptr(d1),
                                        _ptr(d2), _ptr(dd), _ptr(ws), ws.numel(), self._stream(dev)),
            "lk_kron_logdet_f32",
        )
        return out, d1, d2, dd

    def kron_logdet_blocks(self, blocks, deltas, scale=None, want_grads=False):
        """``blocks``: list of ``(l1,)`` / ``(l1, l2)`` eigenvalue vectors; ``deltas [len(blocks)]``; ``scale [1]`` or
        None.  Returns ``(value[1], d_deltas or None, d_scale or None)`` -- the whole posterior in three launches."""
        nb = len(blocks)
        _check(deltas, "deltas")
        if deltas.numel() != nb:
            raise ValueError("kron_logdet_blocks: one delta per block")
        dev = deltas.device
        for ls in blocks:
            for l in ls:
                _check(l, "eigenvalues")
        if scale is not None:
            _check(scale, "scale")
        PA, LA = ctypes.c_void_p * nb, ctypes.c_int64 * nb
        l1 = PA(*[ls[0].data_ptr() for ls in blocks])
        l2 = PA(*[(ls[1].data_ptr() if len(ls) == 2 else None) for ls in blocks])
        n1 = LA(*[ls[0].numel() for ls in blocks])
        n2 = LA(*[(ls[1].numel() if len(ls) == 2 else 0) for ls in blocks])
        out = torch.zeros(1, dtype=torch.float32, device=dev)
        dd = torch.zeros(nb, dtype=torch.float32, device=dev) if want_grads else None
        ds = torch.zeros(1, dtype=torch.float32, device=dev) if want_grads and scale is not None else None
        nbytes = self.lib.lk_kron_logdet_blocks_workspace_bytes(sum(n1), nb)
        ws = self._workspace(nbytes, dev)
        self._rc(
            self.lib.lk_kron_logdet_blocks_f32(nb, l1, n1, l2, n2, _ptr(deltas), _ptr(scale), _ptr(out), _ptr(dd), _ptr(ds),
                                               _ptr(ws), ws.numel(), self._stream(dev)),
            "lk_kron_logdet_blocks_f32",
        )
        return out, dd, ds

    # ---- predictive ---------------------------------------------------------------------------
    def kron_quadform_linear(self, u, v, l1, l2, delta, fvar, ub=None, lb=None, delta_b=None):
        for t, nm in ((u, "u"), (v, "v"), (l1, "l1"), (l2, "l2"), (delta, "delta"), (fvar, "fvar")):
            _check(t, nm)
        Cc, B, Do = u.shape
        Di = v.shape[1]
        if ub is not None:
            _check(ub, "ub"), _check(lb, "lb"), _check(delta_b, "delta_b")
        self._rc(
            self.lib.lk_kron_quadform_linear_f32(_ptr(u), _ptr(v), _ptr(l1), _ptr(l2), _ptr(delta), B, Cc, Do, Di, _ptr(ub),
                                                 _ptr(lb), _ptr(delta_b), _ptr(fvar), self._stream(u.device)),
            "lk_kron_quadform_linear_f32",
        )
        return fvar

    #: most outputs the fused weight-sharing predictive holds in accumulators at once
    quadform_shared_max_outputs = 10

    def kron_quadform_shared(self, u, v, l1, l2, delta, fvar, seed_major=False):
        """``u [B, C, Do, L]`` (``seed_major``: ``[C, B, Do, L]``), ``v [B, Dk, L]`` (eigenbasis projections), fp32;
        ``fvar [B, C, C] +=`` (three-piece bf16 products; operands that arrive split: :meth:`kron_quadform_shared_planes`)"""
        for t, nm in ((u, "u"), (v, "v"), (l1, "l1"), (l2, "l2"), (delta, "delta"), (fvar, "fvar")):
            _check(t, nm)
        if seed_major:
            C, B, Do, L = u.shape
        else:
            B, C, Do, L = u.shape
        Dk = v.shape[1]
        ws = self._workspace(self.lib.lk_quadform_shared_workspace_bytes(B, C, Do, Dk), u.device)
        if seed_major:
            self._rc(
                self._timed("quadconv", 2.0 * B * C * L * Do * Dk, u.device, lambda: self.lib.lk_kron_quadform_shared_seedmajor_f32(
                    _ptr(u), _ptr(v), _ptr(l1), _ptr(l2), _ptr(delta), B, C, Do, Dk, L, _ptr(fvar), _ptr(ws), ws.numel(),
                    self._stream(u.device))),
                "lk_kron_quadform_shared_seedmajor_f32",
            )
            return fvar
        self._rc(
            self._timed("quadconv", 2.0 * B * C * L * Do * Dk, u.device, lambda: self.lib.lk_kron_quadform_shared_f32(
                _ptr(u), _ptr(v), _ptr(l1), _ptr(l2), _ptr(delta), B, C, Do, Dk, L, _ptr(fvar), _ptr(ws), ws.numel(),
                self._stream(u.device))),
            "lk_kron_quadform_shared_f32",
        )
        return fvar

    #: ``False``: the Kron predictive's rotations emit fp32 and the quadratic-form kernel splits in flight (three bf16 pieces)
    use_quad_planes = True

    def kron_quadform_shared_planes(self, u, v, l1, l2, delta, fvar, C):
        """lk_kron_quadform_shared_planes_f16x2: ``u`` SplitTensor ``[C * B, Do, L]`` (seed-major, one scale), ``v`` SplitTensor
        ``[B, Dk, L]`` (one scale per sample or one) — the outputs of :meth:`conv_nhwc_f16x2_planes`; ``fvar [B, C, C] +=``"""
        for t, nm in ((l1, "l1"), (l2, "l2"), (delta, "delta"), (fvar, "fvar")):
            _check(t, nm)
        _one_scale(u, "kron_quadform_shared_planes (u)")
        CB, Do, L = u.shape
        B, Dk = v.shape[0], v.shape[1]
        if CB != C * B or v.shape[2] != L or L % 16 or Do % 32:
            raise LaplaceHipError("kron_quadform_shared_planes: u [C * B, Do, L], v [B, Dk, L], L % 16 == 0, Do % 32 == 0")
        u, v = u.chunk_major(), v.chunk_major()  # (what the rotation convolutions emit; a copy for operands split elsewhere)
        ws = self._workspace(self.lib.lk_quadform_shared_workspace_bytes(B, C, Do, Dk), fvar.device)
        # (its own profile tag: three fp16 MFMAs per product block — bench.py prices the fp32-operand forms, six bf16 MFMAs, apart)
        self._rc(self._timed("quadconv16", 2.0 * B * C * L * Do * Dk, fvar.device, lambda: self.lib.lk_kron_quadform_shared_planes_f16x2(
            _ptr(u.planes[0]), _ptr(u.planes[1]), _ptr(u.sexp), _ptr(v.planes[0]), _ptr(v.planes[1]), _ptr(v.sexp), v.sexp.numel(),
            _ptr(l1), _ptr(l2), _ptr(delta), B, C, Do, Dk, L, _ptr(self._zero16(fvar.device)), _ptr(fvar), _ptr(ws), ws.numel(),
            self._stream(fvar.device))), "lk_kron_quadform_shared_planes_f16x2")
        return fvar

    def diag_quadform_shared(self, u, v, var_w, fvar):
        """``u [B, C, Do, L]`` output gradients, ``v [B, Dk, L]`` unfolded inputs, ``var_w [Do, Dk]``."""
        for t, nm in ((u, "u"), (v, "v"), (var_w, "var_w"), (fvar, "fvar")):
            _check(t, nm)
        B, C, Do, L = u.shape
        Dk = v.shape[1]
        ws = self._workspace(self.lib.lk_quadform_shared_workspace_bytes(B, C, Do, Dk), u.device)
        self._rc(
            self.lib.lk_diag_quadform_shared_f32(_ptr(u), _ptr(v), _ptr(var_w), B, C, Do, Dk, L, _ptr(fvar), _ptr(ws),
                                                 ws.numel(), self._stream(u.device)),
            "lk_diag_quadform_shared_f32",
        )
        return fvar

    def diag_ggn_shared(self, u, v, alpha, h):
        """``h[Do*Dk] += alpha * sum_{n,s} (u[n,s] v[n]^T)^2``; ``u [B, S, Do, L]``, ``v [B, Dk, L]``, ``S <= 10``."""
        for t, nm in ((u, "u"), (v, "v"), (h, "h")):
            _check(t, nm)
        B, S, Do, L = u.shape
        Dk = v.shape[1]
        ws = self._workspace(self.lib.lk_diag_ggn_shared_workspace_bytes(B, Do, Dk), u.device)
        self._rc(
            self.lib.lk_diag_ggn_shared_f32(_ptr(u), _ptr(v), B, S, Do, Dk, L, float(alpha), _ptr(h), _ptr(ws), ws.numel(),
                                            self._stream(u.device)),
            "lk_diag_ggn_shared_f32",
        )
        return h

    def diag_quadform_linear(self, a, g, var_w, var_b, fvar):
        for t, nm in ((a, "a"), (g, "g"), (var_w, "var_w"), (fvar, "fvar")):
            _check(t, nm)
        Cc, B, Do = g.shape
        Di = a.shape[1]
        if var_b is not None:
            _check(var_b, "var_b")
        self._rc(
            self.lib.lk_diag_quadform_linear_f32(_ptr(a), _ptr(g), _ptr(var_w), _ptr(var_b), B, Cc, Do, Di, _ptr(fvar),
                                                 self._stream(a.device)),
            "lk_diag_quadform_linear_f32",
        )
        return fvar

    def diag_quadform_js(self, Js, var):
        _check(Js, "Js"), _check(var, "var")
        B, C, P = Js.shape
        fvar = torch.empty(B, C, C, dtype=torch.float32, device=Js.device)
        self._rc(self.lib.lk_diag_quadform_js_f32(_ptr(Js), _ptr(var), B, C, P, _ptr(fvar), self._stream(Js.device)),
                 "lk_diag_quadform_js_f32")
        return fvar

    def jac_last_layer(self, phi, C, has_bias):
        """Js [B, C, P] = e_c (x) [phi_n, 1] (curvature.py:131-167 for a Linear head)."""
        _check(phi, "phi")
        B, D = phi.shape
        Js = torch.empty(B, C, C * D + (C if has_bias else 0), dtype=torch.float32, device=phi.device)
        self._rc(self.lib.lk_jac_last_layer_f32(_ptr(phi), B, C, D, 1 if has_bias else 0, _ptr(Js), self._stream(phi.device)),
                 "lk_jac_last_layer_f32")
        return Js

    def dense_quadform_ll(self, phi, Sigma, C, has_bias):
        _check(phi, "phi"), _check(Sigma, "Sigma")
        B, D = phi.shape
        fvar = torch.empty(B, C, C, dtype=torch.float32, device=phi.device)
        nb = self.lib.lk_dense_quadform_ll_workspace_bytes(B, C, D)
        ws = self._workspace(nb, phi.device)
        Dt = D + (1 if has_bias else 0)
        # phi~^T Sigma_ck phi~ for the C (C + 1) / 2 class pairs: C (C + 1) Dt^2 flop per sample (as written in the reference,
        # baselaplace.py:1683-1684 through [B, C, P] Jacobians: 2 C P^2)
        self._rc(self._timed("llquad", float(B) * C * (C + 1) * Dt * Dt, phi.device, lambda: self.lib.lk_dense_quadform_ll_f32(
            _ptr(phi), _ptr(Sigma), B, C, D, 1 if has_bias else 0, _ptr(fvar), _ptr(ws), ws.numel(), self._stream(phi.device)),
            nbytes=4.0 * (B * D + (C * Dt) ** 2 + B * C * C)), "lk_dense_quadform_ll_f32")
        return fvar


_KERNELS = None


def get_kernels():
    """The process-wide kernel provider (loads the HIP library on first use; raises if absent)."""
    global _KERNELS
    if _KERNELS is None:
        _KERNELS = HipKernels()
    return _KERNELS


def set_kernels_for_testing(impl):
    """Install a stand-in provider (CPU emulation of the kernels, used only by the `not gpu` tests
    of the host logic).  Returns the previous provider."""
    global _KERNELS
    prev = _KERNELS
    _KERNELS = impl
    return prev
