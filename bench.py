"""bench.py — KFAC-GGN fit throughput of the MI355X-native curvature backend (BASELINE.json metric).

Workload (config c4, per GPU): ResNet-18 (CIFAR stem, BN affine frozen, random init), full-network
KFAC exact GGN, minibatch 128 of synthetic N(0,1) 3x32x32 images, 10 classes.  A "step" is one
minibatch through the hot path, all of it our HIP kernels: NHWC forward (implicit-GEMM convolution on split-fp16
operands + fused BatchNorm/add/ReLU), likelihood root, ONE seed-batched reverse sweep (the same convolution kernel as
backward-data, element-wise VJPs emitting split tensors), every A / G factor, accumulation.  The library (rocBLAS)
only sees the 512 x 10 head.  N > 1: one process per GPU, each rank its own K minibatches (weak scaling), one RCCL
all-reduce of the accumulated factors inside the timed region (the fit's epoch end).

Usage:  python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run)
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH = 128
CLASSES = 10
N_DATASET = 50_000  # the global N every rank passes to kron() (A factors carry 1/N)
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA
# fp32-equivalent ceilings of the split schemes: three fp16 MFMAs (two-piece fp16 operands) / six bf16 MFMAs
# (three-piece bf16 operands) per fp32 product block
PEAK_F16X2_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3.0
PEAK_BF16X3_TFLOPS = PEAK_F16_MFMA_TFLOPS / 6.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-predictive", action="store_true")
    ap.add_argument("--no-eigh", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="bound of the CPU-baseline leg")
    ap.add_argument("--no-overlap", action="store_true", help="A-factor kernels on the main stream (A/B switch)")
    ap.add_argument("--no-sweep", action="store_true", help="one autograd reverse pass per seed instead of the seed-batched sweep")
    return ap.parse_args()


def make_batches(steps, dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    X = torch.randn(BATCH, 3, 32, 32, generator=g)
    y = torch.randint(CLASSES, (BATCH,), generator=g)
    # inputs resident in HBM before the timed region; a couple of distinct batches are cycled
    xs = [(X.roll(i, 0).to(dev), y.roll(i, 0).to(dev)) for i in range(min(steps, 4))]
    return xs


def fit_steps(backend, batches, n_steps, world, overlap=True):
    """K minibatches through the fused accumulator, the fit's single all-reduce, and the one-off
    symmetrise/permute into the reference's Kron layout — i.e. everything `fit` does before decompose."""
    from laplace_amd.laplace import allreduce_curvature

    acc = backend.kron_accumulator(N_DATASET, overlap=overlap)
    for i in range(n_steps):
        X, y = batches[i % len(batches)]
        acc.add_batch(X, y)
    if world > 1:
        allreduce_curvature(acc.tensors(), mirror=False)  # packed upper triangles; finalize() mirrors
    loss, H = acc.finalize()
    return loss, H


def cpu_baseline(seconds: float):
    """The oracle's KFAC restatement (curvlinops 2.0.0 semantics through
    laplace/curvature/curvlinops.py:77-108), fp32, on this host's cores — a reported baseline only."""
    from laplace_amd.nets import ResNet18
    from oracle import curvature_oracle as co

    torch.manual_seed(711)
    model = ResNet18(CLASSES).eval()
    g = torch.Generator().manual_seed(1)
    bs = 32
    X = torch.randn(bs, 3, 32, 32, generator=g)
    y = torch.randint(CLASSES, (bs,), generator=g)
    co.kfac_ggn(model, X[:4], y[:4], N_DATASET, "classification")  # warm-up
    done, t0 = 0, time.time()
    while done < 2 * bs or time.time() - t0 < seconds:
        co.kfac_ggn(model, X, y, N_DATASET, "classification")
        done += bs
    dt = time.time() - t0
    return {"value": done / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{done} synthetic images (batch {bs}) through oracle.kfac_ggn, ResNet-18, fp32, {dt:.1f} s"}


def predictive_leg(dev):
    """Config c3: ResNet-18 last-layer dense GGN fit + GLM predictive variance (batch 512)."""
    from laplace_amd.laplace import HipLaplace
    from laplace_amd.nets import ResNet18

    torch.manual_seed(711)
    model = ResNet18(CLASSES).to(dev).eval()
    la = HipLaplace(model, "classification", "last_layer", "full", last_layer_name="fc")
    g = torch.Generator().manual_seed(3)
    X = torch.randn(512, 3, 32, 32, generator=g).to(dev)
    y = torch.randint(CLASSES, (512,), generator=g).to(dev)

    class _DS:
        def __len__(self):
            return N_DATASET

    class _Loader:
        dataset = _DS()

        def __init__(self, n):
            self.n = n

        def __iter__(self):
            return iter([(X, y)] * self.n)

    la.fit(_Loader(2), distributed=False)  # warm-up
    torch.cuda.synchronize()
    t0 = time.time()
    la.fit(_Loader(8), distributed=False)
    torch.cuda.synchronize()
    fit_rate = 8 * 512 / (time.time() - t0)
    la._glm_predictive_distribution(X)  # warm-up incl. the one-off P^3 factorisation
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        f_mu, f_var = la._glm_predictive_distribution(X)
    torch.cuda.synchronize()
    return {"workload": "ResNet-18 last-layer (P=5130) dense GGN + GLM predictive, batch 512",
            "fit_samples_per_s": fit_rate, "predictive_samples_per_s": 10 * 512 / (time.time() - t0)}


def pmc_traffic(kernel_prefix: str, kernel_suffix: str = ""):
    """HBM bytes per launch of the dominant kernel family from the committed rocprofv3 PMC passes over this very
    command (FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 correction applied: tools/pmc_traffic.py).
    Counters cannot be collected from inside the process, so the figure is read from profiles/; None if absent."""
    table, used = None, None
    for name in ("r02_pmc_traffic_bench_c4_v6", "r02_pmc_traffic_bench_c4_v5", "r02_pmc_traffic_bench_c4_v4", "r02_pmc_traffic_bench_c4", "r01_pmc_traffic_bench_c4"):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name + ".json")
        try:
            with open(path) as fh:
                table, used = json.load(fh), name
            break
        except (OSError, ValueError):
            continue
    if table is None:
        return None, None
    rows = [v for k, v in table.items() if k.startswith(kernel_prefix) and k.endswith(kernel_suffix)]
    launches = sum(r["launches"] for r in rows)
    if not launches:
        return None, None
    total = sum(r["hbm_bytes_per_launch"] * r["launches"] for r in rows)
    return total / launches, f"profiles/{used}.{{json,md}} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, 2x read correction)"


# LK_BENCH_SELFTEST=1: control-flow check of this script without a GPU (tests/test_bench_contract.py): CPU tensors,
# gloo, the kernel emulation of the test-suite and a toy model.  Its numbers mean nothing; it exists so that the
# multi-rank branches (all-reduce, sharded eigendecomposition, max-over-ranks timing) are exercised before the
# driver runs them on 8 GPUs.
SELFTEST = os.environ.get("LK_BENCH_SELFTEST") == "1"


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if SELFTEST:
        global BATCH
        BATCH = 4
        dev = torch.device("cpu")
        if world > 1:
            dist.init_process_group("gloo")
        from laplace_amd import _lib
        from tests.emulated_kernels import EmulatedKernels

        _lib.set_kernels_for_testing(EmulatedKernels())
    else:
        assert torch.cuda.is_available(), "bench.py needs a ROCm device"
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        if world > 1:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", device_id=dev)

    from laplace_amd import HipGGN
    from laplace_amd._lib import get_kernels
    from laplace_amd.nets import ResNet18

    torch.manual_seed(711)
    if SELFTEST:
        from torch import nn

        model = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 8, 3, stride=2, padding=1), nn.ReLU(),
                              nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(8, CLASSES)).eval()
    else:
        model = ResNet18(CLASSES).to(dev).eval()
    backend = HipGGN(model, "classification")
    backend.use_sweep = not args.no_sweep
    batches = make_batches(args.steps, dev, seed=100 + rank)
    K = get_kernels()

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        sync()

    # ---- warm-up ------------------------------------------------------------------------------------
    overlap = not args.no_overlap
    fit_steps(backend, batches, max(args.warmup, 1), world, overlap)
    barrier()

    # ---- timed region: exactly K steps (+ the fit's single all-reduce and layout finalisation) --------------
    barrier()
    t0 = time.perf_counter()
    loss, H = fit_steps(backend, batches, args.steps, world, overlap)
    barrier()
    dt = time.perf_counter() - t0

    # ---- roofline leg (rank 0): the same steps once more with every Gram launch bracketed by HIP events on
    # its launch stream, A-factor kernels NOT overlapped with the reverse passes so that the per-launch
    # durations are the kernels' own (the throughput above is measured without this instrumentation)
    prof = {}
    prof_steps = min(args.steps, 5)
    serial_ms = None
    if rank == 0 and not SELFTEST:
        fit_steps(backend, batches, 2, 1, overlap=False)  # (the accumulators of this leg warm up outside the clock)
        torch.cuda.synchronize()
        K.profile = prof
        t_s = time.perf_counter()
        fit_steps(backend, batches, prof_steps, 1, overlap=False)
        torch.cuda.synchronize()
        serial_ms = (time.perf_counter() - t_s) * 1e3 / prof_steps
        K.profile = None
    if world > 1:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    result = None
    if rank == 0:
        samples = world * BATCH * args.steps
        # roofline of the dominant kernel family: the implicit-im2col MFMA Gram kernel (A factors)
        def agg(name):
            evs = prof.get(name, [])
            ms = sum(e0.elapsed_time(e1) for e0, e1, _ in evs)
            work = sum(w for _, _, w in evs)
            return ms, work, len(evs)

        # kernel families of the step: (profile key, what, bound, PMC kernel-name prefix).  MFMA families are priced on
        # the symmetric-half flop K*n*(n+1) of the product they compute; the pixel-pair family computes the same A
        # factors with 13/40.5 of those multiply-adds and is bound by the read-modify-write of its blocks, so it is
        # priced on its algorithmic bytes (2 x blocks + input) against HBM.
        PEAK_HBM_GBS = 8000.0
        F16X2 = ("fp32-level products from two-piece fp16 operands: three v_mfma_f32_32x32x16_f16 per fp32 product block; "
                 "`achieved` = algorithmic fp32 flop / s, `peak` = 2500 / 3 TFLOP/s (the dense fp16 MFMA peak over the three "
                 "MFMAs of the scheme), i.e. `frac` IS the fraction of the fp16 matrix peak the kernel sustains")
        fams = {
            "conv16": ("lk::conv_f16x2_kernel: implicit-GEMM convolution on NHWC split tensors — backward-data of the "
                       "seed-batched reverse sweep (batch 9 x 128) and the forward; " + F16X2, "mfma16",
                       "lk::conv_f16x2_kernel"),
            "gram16": ("lk::gram16_kernel (+ fixed-order reduce): G factors as Grams of the NHWC split cotangents through "
                       "transposing LDS reads; symmetric-half flop K*n*(n+1); " + F16X2, "mfma16", "lk::gram16_kernel", ",0>"),
            "vjp16": ("lk::vjp_nhwc_split_kernel: element-wise VJP (mask x folded BatchNorm scale x residual add) of all "
                      "seeds, emitting split tensors", "hbm", "lk::vjp_nhwc_split_kernel"),
            "bnact16": ("lk::bn_act_fwd_nhwc_kernel: forward BatchNorm-eval + add + ReLU, emitting fp32, mask, split planes",
                        "hbm", "lk::bn_act_fwd_nhwc_kernel"),
            "gram_nt": ("lk::gram_kernel<MODE_NTB> (+ slab reduce): G factors from NCHW cotangents (graphs outside the NHWC "
                        "sweep); six bf16 MFMAs per fp32 product block", "mfma", "void lk::gram_kernel<5,"),
            "pixpair": ("lk::gram_kernel<MODE_TNP>: banded pixel-pair accumulation of the 3x3-conv A factors "
                        "(block read-modify-write)", "hbm", "void lk::gram_kernel<4,"),
            "pixpair16": ("lk::gram16_kernel<.., TNP>: banded pixel-pair accumulation of the 3x3-conv A factors from the split "
                          "images (block read-modify-write once per four stacked minibatches; three fp16 MFMAs per fp32 "
                          "product block); priced on its algorithmic bytes, 2 x blocks + input", "hbm", "lk::gram16_kernel", ",1>"),
            "gram_conv": ("lk::gram_kernel<MODE_CONV> (+ slab reduce): implicit-im2col A-factor accumulation, "
                          "exact-fp32 MFMA", "mfma", "void lk::gram_kernel<2,"),
            "shiftcorr": ("lk_conv3x3_shiftcorr_f32: shift-correlation A factors", "mfma", "void lk::gram_kernel<3,"),
            "gram_tn": ("lk::gram_kernel<MODE_TN>: Linear-layer factors", "mfma", "void lk::gram_kernel<0,"),
        }
        fam_out, dominant = {}, None
        for key, spec in fams.items():
            what, bound, prefix = spec[:3]
            suffix = spec[3] if len(spec) > 3 else ""
            ms, work, n = agg(key)
            if n == 0 or ms <= 0:
                continue
            if bound == "mfma":
                achieved, peak, unit = work / (ms * 1e-3) / 1e12, PEAK_F32_MFMA_TFLOPS, "TFLOP/s"
            elif bound == "mfma16":
                achieved, peak, unit, bound = work / (ms * 1e-3) / 1e12, PEAK_F16X2_TFLOPS, "TFLOP/s", "mfma"
            else:
                achieved, peak, unit = work / (ms * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
            traffic, traffic_src = pmc_traffic(prefix, suffix)
            fam_out[key] = {"kernel": what, "bound": bound, "achieved": achieved, "peak": peak, "unit": unit,
                            "frac": achieved / peak, "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                            "traffic_source": traffic_src, "launches": n, "avg_launch_ms": ms / n,
                            "ms_per_step": ms / prof_steps}
            if dominant is None or ms > fam_out[dominant]["ms_per_step"] * prof_steps:
                dominant = key
        roof = dict(fam_out[dominant]) if dominant else {"bound": "mfma", "achieved": 0.0, "peak": PEAK_F32_MFMA_TFLOPS,
                                                          "unit": "TFLOP/s", "frac": 0.0, "traffic": None}
        roof["family"] = dominant
        if dominant in ("conv16", "gram16"):
            # what the nominal peak means on this chip (committed probes, same box type): the matrix pipe alone, fed from
            # registers with this kernel's instruction mix, sustains 0.63-0.78 of 2.5 PFLOP/s; the vendor library's plain
            # fp16 GEMM reaches 0.43 on 8192^3 and 0.11-0.39 on the GEMM shapes these convolutions reduce to
            roof["peak_calibration"] = {
                "sustained_mfma_frac_of_nominal": [0.63, 0.78],
                "hipblaslt_fp16_gemm_frac_of_nominal": {"8192^3": 0.43, "conv_shapes_c4": [0.11, 0.19, 0.31, 0.39]},
                "source": "profiles/r02_mfma_peak_probe.txt, profiles/r02_gemm_calib_hipblaslt_fp16.json "
                          "(tools/probes/mfma_peak_probe.hip, tools/gemm_calib.py)"}
        roof["flop_convention"] = ("convolution: 2 * pixels * Cout * Cin * taps per launch (fp32 multiply-adds of the "
                                   "algorithm); Gram families: symmetric half K*n*(n+1)")
        own_ms = sum(v["ms_per_step"] for v in fam_out.values())
        breakdown = None
        if serial_ms is not None:
            # where a step goes when nothing overlaps (the instrumented pass): our kernel families, and the rest (rocBLAS
            # for the 512 x 10 head, torch element-wise glue, launch gaps).  No MIOpen kernel is left in the step.
            breakdown = {"serial_ms_per_step": serial_ms, "own_kernels_ms": own_ms, "other_ms": max(serial_ms - own_ms, 0.0),
                         "own_share": own_ms / serial_ms if serial_ms > 0 else None,
                         "note": "measured without stream overlap; the timed region overlaps A-factor and G-factor "
                                 "kernels with the reverse sweep"}
        result = {
            "metric": "KFAC-GGN fit samples/sec, ResNet-18",
            "value": samples / dt,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "c4: ResNet-18 (CIFAR stem, BN frozen) full-network KFAC exact GGN fit, "
                                   "per-GPU minibatch 128, synthetic N(0,1) 3x32x32, 10 classes, N=50000",
                       "per_gpu_batch": BATCH, "parallelism": f"dp{world}"},
            "roofline": roof,  # the family with the largest share of the WHOLE step (measured without stream overlap)
            "roofline_families": fam_out,
            "step_breakdown": breakdown,
        }
    # ---- untimed extras on rank 0 (separate line items per BASELINE.md) --------------------------------------
    if rank == 0 and world == 1 and not SELFTEST:
        # what the reference's own `fit` executes with backend=HipGGN: `self.H += backend.kron(X, y, N)` per minibatch
        # (laplace/baselaplace.py:969-985) — one Kron per batch in the public layout, then the factor-wise add
        from laplace_amd.kron import HipKron

        def literal(n):
            Hs = HipKron.init_from_model(backend.params, dev, torch.float32)
            tot = torch.zeros((), device=dev)
            for i in range(n):
                X, y = batches[i % len(batches)]
                lb, Hb = backend.kron(X, y, N=N_DATASET)
                tot = tot + lb
                Hs += Hb
            return Hs

        literal(2)
        sync()
        t0 = time.perf_counter()
        n_lit = min(args.steps, 10)
        literal(n_lit)
        sync()
        result["dropin_fit_samples_per_s"] = n_lit * BATCH / (time.perf_counter() - t0)
    if rank == 0 and world == 1:
        if not args.no_eigh:
            sync()
            t0 = time.perf_counter()
            dec = H.decompose()
            sync()
            result["eigh_ms"] = (time.perf_counter() - t0) * 1e3
            result["eigh_converged"] = all(int(i[0].item()) == 0 for i in dec._eig_info)
            if not args.no_predictive and not SELFTEST:
                # GLM predictive variance under the full-network KFAC posterior just fitted (V1 of SURVEY.md 8a):
                # one seed-batched reverse sweep + the weight-sharing quadratic-form kernel per layer
                from laplace_amd import predictive as _pred

                post = dec + torch.ones(1, device=dev)
                Xp = batches[0][0]
                _pred.glm_variance_kron(backend, Xp, post)
                sync()
                t0 = time.perf_counter()
                for _ in range(3):
                    f_mu, f_var = _pred.glm_variance_kron(backend, Xp, post)
                sync()
                pred_rate = 3 * len(Xp) / (time.perf_counter() - t0)
                pprof = {}
                K.profile = pprof
                _pred.glm_variance_kron(backend, Xp, post)
                sync()
                K.profile = None
                pred_ms = len(Xp) / pred_rate * 1e3
                pfam = {}
                for key, peak, what in (("quadconv", PEAK_BF16X3_TFLOPS, "lk::quadform_conv_kernel: per-layer quadratic form of "
                                         "the weight-sharing Jacobian, six bf16 MFMAs per fp32 product block"),
                                        ("conv16", PEAK_F16X2_TFLOPS, "lk::conv_f16x2_kernel (forward + reverse sweep of the "
                                         "10 identity seeds)")):
                    evs = pprof.get(key, [])
                    ms_k = sum(e0.elapsed_time(e1) for e0, e1, _ in evs)
                    if evs and ms_k > 0:
                        tf = sum(w for _, _, w in evs) / (ms_k * 1e-3) / 1e12
                        pfam[key] = {"kernel": what, "bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s",
                                     "frac": tf / peak, "ms_per_call": ms_k, "launches": len(evs)}
                dom = max(pfam, key=lambda k: pfam[k]["ms_per_call"]) if pfam else None
                result["predictive_kron_c4"] = {
                    "workload": "c4 posterior: ResNet-18 full-network KFAC, GLM predictive variance [B,10,10], batch 128",
                    "samples_per_s": pred_rate, "ms_per_call": pred_ms, "finite": bool(torch.isfinite(f_var).all()),
                    "roofline": dict(pfam[dom], family=dom) if dom else None, "roofline_families": pfam}
            del dec
        if not args.no_predictive and not SELFTEST:
            result["predictive"] = predictive_leg(dev)
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(0.0 if SELFTEST else args.cpu_seconds)
    if world > 1 and not args.no_eigh:
        # the factors are identical on every rank after the all-reduce: shard the eigensolves over the GPUs
        barrier()
        t0 = time.perf_counter()
        dec = H.decompose(distributed=True)
        barrier()
        t = torch.tensor([time.perf_counter() - t0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = torch.tensor([float(all(int(i[0].item()) == 0 for i in dec._eig_info))], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if rank == 0:
            result["eigh_ms"] = float(t.item()) * 1e3
            result["eigh_sharded_over_gpus"] = world
            result["eigh_converged"] = bool(ok.item())
        del dec
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
