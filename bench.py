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
import gc
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
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-predictive", action="store_true")
    ap.add_argument("--no-eigh", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip fit_50k and the c1 / c2 / c5 legs")
    ap.add_argument("--no-check", action="store_true", help="skip the serial re-run the timed factors are compared with")
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


def fit_steps(backend, batches, n_steps, world, overlap=True, serial=False):
    """K minibatches through the fused accumulator, the fit's single all-reduce, and the one-off
    symmetrise/permute into the reference's Kron layout — i.e. everything `fit` does before decompose.
    ``serial``: every scheduling feature off (`pix_group = 1`: one pixel-pair launch per minibatch, no lagged join, no side stream) —
    the independent re-run the timed factors are checked against."""
    from laplace_amd.laplace import allreduce_curvature

    if serial:
        acc = backend.kron_accumulator(N_DATASET, overlap=False)
        acc.pix_group, acc.lag_join = 1, False
        assert acc.pix_group == 1 and not acc.lag_join and not acc.overlap
    else:
        acc = backend.kron_accumulator(N_DATASET, overlap=overlap)
    for i in range(n_steps):
        X, y = batches[i % len(batches)]
        acc.add_batch(X, y)
    if world > 1:
        info = allreduce_curvature(acc.tensors(), mirror=False)  # packed upper triangles; finalize() mirrors
        fit_steps.last_exchange = info
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
    pred_rate = 10 * 512 / (time.time() - t0)
    # What of those rates is the PATH and what the backbone: the two kernels of the leg under HIP events (their own launches;
    # algorithmic flop as the structured forms count it) and the backbone forward timed by itself —
    # the 75 k / 79 k samples/s above are a ResNet-18 forward at batch 512 with a ~0.1 ms kernel behind it.
    from laplace_amd._lib import get_kernels

    K = get_kernels()
    fams = {}
    prof = {}
    K.profile = prof
    la.fit(_Loader(4), distributed=False)
    for _ in range(4):
        la._glm_predictive_distribution(X)
    torch.cuda.synchronize()
    K.profile = None
    for key, what, asw in (("llggn", "lk_ll_ggn_full_f32: dense last-layer GGN, structured (C block Grams of sqrt(p) phi~ minus "
                            "the Gram of [p_j phi~]) on the exact-fp32 MFMA Gram engine", 2.0 * CLASSES * 5130.0 ** 2),
                           ("llquad", "lk_dense_quadform_ll_f32: phi~^T Sigma_ck phi~ for all class pairs, exact-fp32 MFMA",
                            2.0 * CLASSES * 5130.0 ** 2)):
        evs = prof.get(key, [])
        ms = sum(ev[0].elapsed_time(ev[1]) for ev in evs)
        if not evs or ms <= 0:
            continue
        work = sum(ev[2] for ev in evs)
        nb = sum(ev[3] for ev in evs)
        tf = work / (ms * 1e-3) / 1e12
        ridge = PEAK_F32_MFMA_TFLOPS * 1e12 / 8e12
        fams[key] = {"kernel": what, "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_F32_MFMA_TFLOPS,
                     "bound": "hbm" if work / nb < ridge else "mfma", "flop_per_byte_algorithmic": work / nb,
                     "frac_hbm_algorithmic": nb / (ms * 1e-3) / 8e12, "avg_launch_ms": ms / len(evs), "launches": len(evs),
                     "samples_per_s_kernel_alone": 512 * len(evs) / (ms * 1e-3),
                     "as_written_in_the_reference_flop_per_sample": asw,
                     "as_written_tflops_equivalent": asw * 512 * len(evs) / (ms * 1e-3) / 1e12}
    with torch.no_grad():
        for _ in range(2):
            model(X)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(10):
            model(X)
        torch.cuda.synchronize()
    fwd_ms = (time.time() - t0) / 10 * 1e3
    return {"workload": "ResNet-18 last-layer (P=5130) dense GGN + GLM predictive, batch 512",
            "fit_samples_per_s": fit_rate, "predictive_samples_per_s": pred_rate,
            "backbone_forward_ms_per_batch_of_512": fwd_ms, "backbone_forward_samples_per_s": 512 / (fwd_ms * 1e-3),
            "roofline_families": fams,
            "note": "fit / predictive rates include the backbone forward (stock PyTorch-ROCm eager, not part of the path); the "
                    "kernels of the path are the two families, timed by HIP events on their launch stream"}


def cpu_predictive_baseline(dec, prior: float, seconds: float):
    """The reference's GLM predictive as written — per-sample Jacobians ``Js [B, 10, P]`` (curvature.py:88-129), every
    Kronecker block rotated into the eigenbasis, weighted and contracted (utils/matrix.py:406-461 via
    baselaplace.py:1834-1835) — by the oracle's restatement on this host's cores, fp32, on the c4 posterior just
    decomposed (eigenpairs copied to the host): a reported baseline only."""
    from laplace_amd.nets import ResNet18
    from oracle import curvature_oracle as co

    torch.manual_seed(711)
    model = ResNet18(CLASSES).eval()
    Qs = [[Q.float().cpu() for Q in blk] for blk in dec.eigenvectors]
    ls = [[l.float().cpu() for l in blk] for blk in dec.eigenvalues]
    g = torch.Generator().manual_seed(1)
    bs = 2
    X = torch.randn(bs, 3, 32, 32, generator=g)
    done, t0 = 0, time.time()
    while done < bs or time.time() - t0 < seconds:
        Js, _ = co.jacobians(model, X)
        fvar = co.krondecomposed_inv_square_form_blocks(Qs, ls, prior, Js)
        done += bs
    dt = time.time() - t0
    return {"value": done / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "finite": bool(torch.isfinite(fvar).all()),
            "sample": f"{done} synthetic images (batch {bs}) through oracle.jacobians + "
                      f"oracle.krondecomposed_inv_square_form_blocks, ResNet-18 KFAC posterior (P = 11.2 M), fp32, {dt:.1f} s"}


class PowerSampler:
    """rocm-smi's socket power and shader clock while a leg runs (a thread polling the tool: ~4 samples per second).  The
    fit is bound by the chip's power budget — the dominant kernel alone holds the socket at 1.28 - 1.40 kW of 1.4 and the
    clock at 1.78 - 1.85 of 2.4 GHz (profiles/r06_power_kernels.log) — so a fraction of the NOMINAL matrix peak cannot be
    read without the clock the chip actually ran at.  Absent tool: null."""

    def __init__(self):
        self.samples, self._stop, self._thread = [], False, None

    def _run(self):
        import re
        import subprocess

        while not self._stop:
            try:
                out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True,
                                     timeout=5).stdout
            except Exception:
                return
            m_clk = re.search(r"GPU\[0\].*sclk clock level.*\((\d+)Mhz\)", out)
            m_pw = re.search(r"GPU\[0\].*Power \(W\): ([0-9.]+)", out)
            if m_clk and m_pw:
                self.samples.append((float(m_clk.group(1)) / 1e3, float(m_pw.group(1))))

    def __enter__(self):
        import threading

        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        self._thread.join(timeout=10)

    def summary(self):
        s = self.samples[1:] if len(self.samples) > 2 else self.samples  # (the first sample: the ramp)
        if not s:
            return None
        clk = sorted(c for c, _ in s)
        pw = sorted(p for _, p in s)
        return {"samples": len(s), "sclk_ghz_median": clk[len(clk) // 2], "socket_w_median": pw[len(pw) // 2],
                "socket_w_max": pw[-1], "nominal_sclk_ghz": 2.4, "socket_cap_w": 1400,
                "source": "rocm-smi --showpower --showclocks polled during the leg"}


def fit_50k_leg(backend, dev):
    """The end-to-end figure BASELINE.json's c4 names: a 50 000-sample fit = 390 minibatches of 128 + one of 80, then
    finalise (pixel-pair assembly, symmetrise, permute) and the eigendecomposition of all 43 factors (one GPU; on N
    GPUs the driver's per-N runs carry the all-reduce).  Inputs resident in HBM (four distinct minibatches cycled)."""
    g = torch.Generator().manual_seed(7)
    bs = [(torch.randn(BATCH, 3, 32, 32, generator=g).to(dev), torch.randint(CLASSES, (BATCH,), generator=g).to(dev))
          for _ in range(4)]
    n_full, rest = divmod(N_DATASET, BATCH)
    last = (bs[1][0][:rest].contiguous(), bs[1][1][:rest].contiguous()) if rest else None
    gc.collect()  # (the legs before this one leave a large heap behind: the host side of a step is 60 % of its device time)
    gc.freeze()   # ... and that heap — the event lists of the instrumented passes, the other legs' models — is not this leg's
                  # to traverse: the collector's full passes over it made a step of THIS leg host-bound in long runs
    torch.cuda.synchronize()
    ms0 = torch.cuda.memory_stats(dev)
    sampler = PowerSampler()
    t0 = time.perf_counter()
    with sampler:
        acc = backend.kron_accumulator(N_DATASET)
        for i in range(n_full):
            acc.add_batch(*bs[i % 4])
        if last is not None:
            acc.add_batch(*last)
        t_host = time.perf_counter()  # (the host has enqueued every minibatch; `max_ahead` keeps it within two of the device)
        loss, H = acc.finalize()
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    dec = H.decompose()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    gc.unfreeze()
    ok = all(int(i[0].item()) == 0 for i in dec._eig_info)
    ms1 = torch.cuda.memory_stats(dev)
    alloc = {"device_mallocs_during_the_fit": ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0),
             "device_frees_during_the_fit": ms1.get("num_device_free", 0) - ms0.get("num_device_free", 0),
             "allocator_retries_so_far": ms1.get("num_alloc_retries", 0), "ooms_so_far": ms1.get("num_ooms", 0),
             "reserved_gib_before": ms0.get("reserved_bytes.all.current", 0) / 2**30,
             "reserved_gib_after": ms1.get("reserved_bytes.all.current", 0) / 2**30}
    return {"allocator": alloc, "samples": N_DATASET, "minibatches": n_full + (1 if rest else 0), "wall_s": t2 - t0, "accumulate_s": t1 - t0,
            "decompose_s": t2 - t1, "host_loop_s": t_host - t0, "samples_per_s": N_DATASET / (t2 - t0), "eigh_converged": bool(ok),
            "loss_finite": bool(torch.isfinite(loss).all()), "power": sampler.summary()}


def small_config_legs(dev):
    """The other BASELINE.json configs as extras of the same line (whole `fit` incl. decomposition, median of 3):
    c1 MLP 1-50-1 regression (diag, N = 1000, batch 100), c2 LeNet-5 KFAC (N = 10 000, batch 256) + its Kron GLM
    predictive, c5 BERT-base last-layer KFAC + 100 marginal-likelihood steps (random init, sequence 128, batch 32)."""
    from laplace_amd.laplace import HipLaplace
    from laplace_amd.nets import lenet5, mlp_1_50_1

    out = {}

    def run(model, X, y, lik, hs, bs, reps=3, **kw):
        class L(list):
            dataset = X

        loader = L([(X[i:i + bs], y[i:i + bs]) for i in range(0, len(X), bs)])
        ts, la = [], None
        for _ in range(reps + 1):
            la = HipLaplace(model, lik, "all", hs, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            la.fit(loader)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t = sorted(ts[1:])[len(ts[1:]) // 2]
        return la, {"fit_ms": t * 1e3, "samples_per_s": len(X) / t}

    torch.manual_seed(711)
    m = mlp_1_50_1().to(dev)
    X = (8 * torch.rand(1000, 1)).to(dev)
    y = (torch.sin(X) + 0.3 * torch.randn_like(X)).to(dev)
    out["c1_mlp_diag"] = run(m, X, y, "regression", "diag", 100)[1]
    out["c1_mlp_kron"] = run(m, X, y, "regression", "kron", 100)[1]
    torch.manual_seed(711)
    m = lenet5().to(dev)
    X = torch.randn(10000, 3, 32, 32, device=dev)
    y = torch.randint(0, 10, (10000,), device=dev)
    la, out["c2_lenet_kron"] = run(m, X, y, "classification", "kron", 256)
    for _ in range(2):
        la._glm_predictive_distribution(X[:256])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(0, 2048, 256):
        la._glm_predictive_distribution(X[i:i + 256])
    torch.cuda.synchronize()
    out["c2_lenet_kron"]["predictive_samples_per_s"] = 2048 / (time.perf_counter() - t0)
    del la, X, y
    try:
        out["c5_bert_base_ll_kron"] = _c5_leg(dev)
    except Exception as e:  # transformers missing / a model change: the headline number must not depend on this leg
        out["c5_bert_base_ll_kron"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def _c5_leg(dev):
    from torch import nn
    from transformers import BertConfig, BertForSequenceClassification

    from laplace_amd.laplace import HipLaplace

    class BertHead(nn.Module):  # the wrapper of the reference's HuggingFace example: dict batch in, logits out
        def __init__(self, cfg):
            super().__init__()
            self.hf = BertForSequenceClassification(cfg)

        def forward(self, data):
            return self.hf(input_ids=data["input_ids"], attention_mask=data["attention_mask"]).logits

    torch.manual_seed(711)
    cfg = BertConfig(num_labels=2)
    T, bs, n = 128, 32, 1024
    model = BertHead(cfg).to(dev).eval()
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(cfg.vocab_size, (n, T), generator=g)
    mask = torch.ones(n, T, dtype=torch.long)
    lens = torch.randint(T // 2, T + 1, (n,), generator=g)
    mask[torch.arange(T)[None, :] >= lens[:, None]] = 0
    yy = torch.randint(2, (n,), generator=g)

    class Loader(list):
        dataset = range(n)

    train = Loader([{"input_ids": ids[i:i + bs].to(dev), "attention_mask": mask[i:i + bs].to(dev), "labels": yy[i:i + bs].to(dev)}
                    for i in range(0, n, bs)])
    la = HipLaplace(model, "classification", "last_layer", "kron", last_layer_name="hf.classifier")
    la.fit(train)  # warm-up (library GEMM selection for the encoder)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    la.fit(train)
    torch.cuda.synchronize()
    t_fit = time.perf_counter() - t0
    t0 = time.perf_counter()
    la.optimize_prior_precision(pred_type="glm", method="marglik", n_steps=100, lr=0.1, prior_structure="layerwise")
    torch.cuda.synchronize()
    t_ml = time.perf_counter() - t0
    return {"fit_samples_per_s": n / t_fit, "marglik_100_steps_ms": t_ml * 1e3,
            "note": "dominated by the encoder forward (stock PyTorch-ROCm: not part of the path)"}


def pmc_traffic(kernel_prefix: str, kernel_suffix: str = "", table: str = "bench_c4", per: str = "launch"):
    """HBM bytes per launch of a kernel family from the rocprofv3 PMC passes over this very command (FETCH_SIZE /
    WRITE_SIZE in separate passes, gfx950 correction applied: tools/pmc_traffic.py).  Counters cannot be collected from
    inside the process, so the figure is read from profiles/ — but ONLY from a table stamped with the hash of the
    kernel sources it was collected on (`_meta.csrc_sha16`) equal to the sources of this run; anything else is stale
    and reported as null with the reason.  ``table``: which command the passes ran over (`bench_c4`: this one;
    `predictive_c4`: tools/kron_predictive_c4.py, `_meta.calls` predictive calls); ``per``: "launch" or "call"."""
    import glob

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from pmc_traffic import csrc_sha16

    want = csrc_sha16(ROOT)
    newest = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic_{table}*.json")), reverse=True):  # newest round first
        try:
            with open(path) as fh:
                tab = json.load(fh)
        except (OSError, ValueError):
            continue
        meta = tab.get("_meta") or {}
        newest = newest or os.path.basename(path)
        if meta.get("csrc_sha16") != want:
            continue
        # (a family may be several kernels: `kernel_prefix` = a sequence of (prefix, suffix) then)
        alts = [(kernel_prefix, kernel_suffix)] if isinstance(kernel_prefix, str) else list(kernel_prefix)
        rows = [v for k, v in tab.items() if k != "_meta" and any((k.startswith(pre) or k.startswith("void " + pre)) and k.endswith(suf)
                                                                   for pre, suf in alts)]
        launches = sum(r["launches"] for r in rows)
        if not launches:
            return None, f"profiles/{os.path.basename(path)}: no launch of this family in the PMC pass"
        total = sum(r["hbm_bytes_per_launch"] * r["launches"] for r in rows)
        src = (f"profiles/{os.path.basename(path)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, 2x read "
               f"correction; kernel sources {want}, git {meta.get('git_head')})")
        if per == "call":
            calls = meta.get("calls")
            return (total / calls, src) if calls else (None, f"profiles/{os.path.basename(path)}: no `calls` in _meta")
        return total / launches, src
    return None, (f"stale: no PMC table `{table}` under profiles/ was collected on these kernel sources ({want}); newest is "
                  f"{newest} — rerun tools/gpu_evidence.sh")


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
    gc.collect()  # (measurement hygiene: no full collection of the warm-up's garbage inside the timed steps)
    barrier()
    t0 = time.perf_counter()
    loss, H = fit_steps(backend, batches, args.steps, world, overlap)
    barrier()
    dt = time.perf_counter() - t0

    # ---- check (rank 0): the factors of the TIMED run against the same K minibatches re-run with every scheduling
    # feature off (no side stream, no lagged join, one pixel-pair launch per minibatch, per-launch reductions) — an
    # ordering bug in the overlapped schedule would produce a fast wrong number.  Each factor block relative to its
    # own largest element; the fp64 / reference-level parity of the same configuration is tests/test_gpu_timed_config.py
    check = None
    if rank == 0 and world == 1 and not args.no_check:
        loss_s, H_s = fit_steps(backend, batches, args.steps, 1, serial=True)
        worst, worst_at = 0.0, None
        for bi, (F_, S_) in enumerate(zip(H.kfacs, H_s.kfacs)):
            for fi, (a_, s_) in enumerate(zip(F_, S_)):
                r = float((a_.double() - s_.double()).abs().max() / (s_.double().abs().max() + 1e-300))
                if r >= worst:
                    worst, worst_at = r, f"block {bi} factor {fi} (n={a_.shape[0]})"
        lerr = float((loss.double() - loss_s.double()).abs() / (loss_s.double().abs() + 1e-300))
        finite = bool(all(torch.isfinite(t).all() for F_ in H.kfacs for t in F_))
        check = {"what": "timed run's factors vs the same minibatches re-run serially (pix_group = 1, lag_join = False, "
                         "no side stream); max over the 43 factors of max|a-b| / max|b| per factor",
                 "max_block_rel_err": worst, "at": worst_at, "loss_rel_err": lerr, "finite": finite,
                 "ok": bool(finite and worst < 1e-5 and lerr < 1e-6), "tol": 1e-5}
        del H_s

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
        instrumented_ms = (time.perf_counter() - t_s) * 1e3 / prof_steps
        K.profile = None
        # The serial step itself is timed WITHOUT the per-launch events: with two event records around each of its ~230
        # launches the host cannot stay ahead of the device, and what round 3 reported as "other" (2.5 ms per step) was
        # mostly the instrumented pass waiting for its own host, not device work.
        fit_steps(backend, batches, 2, 1, overlap=False)
        torch.cuda.synchronize()
        t_s = time.perf_counter()
        fit_steps(backend, batches, prof_steps, 1, overlap=False)
        torch.cuda.synchronize()
        serial_ms = (time.perf_counter() - t_s) * 1e3 / prof_steps
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
            ms = sum(ev[0].elapsed_time(ev[1]) for ev in evs)
            work = sum(ev[2] for ev in evs)
            nb = [ev[3] for ev in evs if len(ev) > 3 and ev[3] is not None]
            return ms, work, len(evs), (sum(nb) / len(nb) if nb and len(nb) == len(evs) else None)

        # kernel families of the step: (profile key, what, bound, PMC kernel-name prefix).  MFMA families are priced on
        # the symmetric-half flop K*n*(n+1) of the product they compute; the pixel-pair family computes the same A
        # factors with 13/40.5 of those multiply-adds and is bound by the read-modify-write of its blocks, so it is
        # priced on its algorithmic bytes (2 x blocks + input) against HBM.
        PEAK_HBM_GBS = 8000.0
        F16X2 = ("fp32-level products from two-piece fp16 operands: three v_mfma_f32_32x32x16_f16 per fp32 product block; "
                 "`achieved` = algorithmic fp32 flop / s, `peak` = 2500 / 3 TFLOP/s (the dense fp16 MFMA peak over the three "
                 "MFMAs of the scheme), i.e. `frac` IS the fraction of the fp16 matrix peak the kernel sustains")
        fams = {
            "convp16": ("lk::conv_winp_f16x2_kernel: persistent window form of the implicit-GEMM convolution on NHWC split "
                        "tensors — the 3x3 / stride-1 backward-data launches of the seed-batched reverse sweep (batch 9 x 128, "
                        "13 per step) with the element-wise VJP fused into the epilogue; " + F16X2, "mfma16",
                        "void lk::conv_winp_f16x2_kernel"),
            "convs16": ("lk::conv_strided_f16x2_kernel: backward-data of the three down-sampling blocks (3x3 / stride 2, all "
                        "four residue classes, + the 1x1 shortcut) with the fused VJP epilogue, one launch per block; " + F16X2,
                        "mfma16", "lk::conv_strided_f16x2_kernel"),
            "conv16": ("lk::conv_f16x2_kernel: the generic implicit-GEMM convolution — the forward pass (batch 128, eval-mode "
                       "BatchNorm / residual add / ReLU in its epilogue) and what the two fused backward forms do not cover; "
                       + F16X2, "mfma16", "lk::conv_f16x2_kernel"),
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
            "pixpair16": ("lk::gram16_kernel<.., TNP> / lk::pixpair13_kernel (64-channel maps: one workgroup per pixel, its panel staged "
                          "once for all 13 shifts): banded pixel-pair accumulation of the 3x3-conv A factors from the split "
                          "images (block read-modify-write once per eight stacked minibatches; three fp16 MFMAs per fp32 "
                          "product block); priced on its algorithmic bytes, 2 x blocks + input", "hbm",
                          (("lk::gram16_kernel", ",1>"), ("lk::pixpair13_kernel", ""))),
            "gram_conv": ("lk::gram_kernel<MODE_CONV> (+ slab reduce): implicit-im2col A-factor accumulation, "
                          "exact-fp32 MFMA", "mfma", "void lk::gram_kernel<2,"),
            "im2col16": ("lk::im2col_split_f16x2_kernel: the patch matrix of the strided / stem convolutions as split planes (its Gram — "
                         "their A factor — then runs on lk::gram16_kernel, counted under gram16)", "hbm", "lk::im2col_split_f16x2_kernel"),
            "shiftcorr": ("lk_conv3x3_shiftcorr_f32: shift-correlation A factors", "mfma", "void lk::gram_kernel<3,"),
            "gram_tn": ("lk::gram_kernel<MODE_TN>: Linear-layer factors", "mfma", "void lk::gram_kernel<0,"),
        }
        fam_out, dominant = {}, None
        for key, spec in fams.items():
            what, bound, prefix = spec[:3]
            suffix = spec[3] if len(spec) > 3 else ""
            ms, work, n, alg_bytes = agg(key)
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
            if unit == "TFLOP/s":
                # Which roof the launch sits under is COMPUTED, not declared (round 4 labelled the persistent convolution
                # "mfma" while its counters said 4.4 TB/s): flop per launch over the bytes the launch actually moved (PMC; the
                # algorithmic bytes when no counter table matches these kernel sources) against the ridge peak flop / 8 TB/s.
                f = fam_out[key]
                secs = ms / n * 1e-3
                moved = traffic if traffic is not None else alg_bytes
                f["peak_fp32_mfma"] = PEAK_F32_MFMA_TFLOPS  # the exact-fp32 matrix peak the SURVEY's roofline (8d) names
                f["frac_of_fp32_mfma_peak"] = achieved / PEAK_F32_MFMA_TFLOPS
                f["algorithmic_bytes_per_launch"] = alg_bytes
                if moved:
                    intensity = work / n / moved
                    ridge = peak * 1e12 / (PEAK_HBM_GBS * 1e9)
                    f["bound"] = "hbm" if intensity < ridge else "mfma"
                    f["bound_from"] = {"flop_per_byte": intensity, "ridge_flop_per_byte": ridge,
                                       "bytes": "pmc" if traffic is not None else "algorithmic"}
                    f["frac_hbm"] = moved / secs / (PEAK_HBM_GBS * 1e9)
                    if traffic is not None and alg_bytes:
                        f["traffic_over_algorithmic"] = traffic / alg_bytes
            if dominant is None or ms > fam_out[dominant]["ms_per_step"] * prof_steps:
                dominant = key
        roof = dict(fam_out[dominant]) if dominant else {"bound": "mfma", "achieved": 0.0, "peak": PEAK_F32_MFMA_TFLOPS,
                                                          "unit": "TFLOP/s", "frac": 0.0, "traffic": None}
        roof["family"] = dominant
        conv_keys = [k for k in ("convp16", "convs16", "conv16") if k in fam_out]
        if conv_keys:  # (all convolution launches together: the figure rounds 2 and 3 quoted for `conv16`)
            c_ms = sum(fam_out[k]["ms_per_step"] for k in conv_keys)
            c_tf = sum(fam_out[k]["achieved"] * fam_out[k]["ms_per_step"] for k in conv_keys) / c_ms
            roof["all_convolution_launches"] = {"families": conv_keys, "ms_per_step": c_ms, "achieved": c_tf,
                                                "frac": c_tf / PEAK_F16X2_TFLOPS,
                                                "launches": sum(fam_out[k]["launches"] for k in conv_keys)}
        roof["flop_convention"] = ("convolution: 2 * Cout * Cin * (output pixel, tap) pairs that fall INSIDE the image "
                                   "(zero-padding taps are not counted: the figure does not move with the padding); "
                                   "Gram families: symmetric half K*n*(n+1)")
        own_ms = sum(v["ms_per_step"] for v in fam_out.values())
        # the whole step by SURVEY.md section 8d's convention: 23.2 GFLOP per sample end to end (forward + C input-gradient
        # passes + factor products, full-GEMM count) against the exact-fp32 matrix peak that section named as the roof, and
        # against the ceiling of the scheme actually used.  Above 1.0 of the fp32 figure is not a kernel skipping work (see
        # `check` and tests/test_gpu_timed_config.py): the products run on the fp16 pipe at three MFMAs each, and the step does
        # fewer flops than that count (9 seeds instead of 10, symmetric halves, pixel-pair A factors at 13 / 40.5).
        step_tf = 23.2e9 * BATCH / (dt / args.steps) / 1e12
        whole_step = {"flop_per_sample_section_8d": 23.2e9, "achieved_tflops_by_that_count": step_tf,
                      "frac_of_fp32_mfma_peak": step_tf / PEAK_F32_MFMA_TFLOPS, "frac_of_f16x2_ceiling": step_tf / PEAK_F16X2_TFLOPS,
                      "peak_fp32_mfma": PEAK_F32_MFMA_TFLOPS, "peak_f16x2": PEAK_F16X2_TFLOPS}
        breakdown = None
        if serial_ms is not None:
            # where a step goes when nothing overlaps (the instrumented pass): our kernel families, and the rest (rocBLAS
            # for the 512 x 10 head, torch element-wise glue, launch gaps).  No MIOpen kernel is left in the step.
            breakdown = {"serial_ms_per_step": serial_ms, "own_kernels_ms": own_ms, "other_ms": max(serial_ms - own_ms, 0.0),
                         "own_share": min(own_ms / serial_ms, 1.0) if serial_ms > 0 else None,
                         "instrumented_serial_ms_per_step": instrumented_ms,
                         "note": "one stream, no overlap (incl. the once-per-fit finalisation spread over the steps); "
                                 "own_kernels_ms: sum of per-launch HIP-event times of the instrumented pass; "
                                 "serial_ms_per_step: the same schedule without the events (the instrumented pass is "
                                 "host-bound); the timed region overlaps two minibatches and the factor kernels"}
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
            "dtype": "f32 (fp16x2-split products, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "c4: ResNet-18 (CIFAR stem, BN frozen) full-network KFAC exact GGN fit, "
                                   "per-GPU minibatch 128, synthetic N(0,1) 3x32x32, 10 classes, N=50000",
                       "per_gpu_batch": BATCH, "parallelism": f"dp{world}",
                       "minibatches_in_flight": int(__import__("laplace_amd.backend", fromlist=["KronAccumulator"]).KronAccumulator.default_lanes)},
            "roofline": roof,  # the family with the largest share of the WHOLE step (measured without stream overlap)
            "roofline_families": fam_out,
            "whole_step": whole_step,
            "step_breakdown": breakdown,
            "check": check,
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
            return Hs.kfacs  # (read: the once-per-fit symmetrise / assemble / permute into the public layout is timed)

        literal(9)
        sync()
        t0 = time.perf_counter()
        n_lit = min(args.steps, 48)
        literal(n_lit)
        sync()
        result["dropin_fit_samples_per_s"] = n_lit * BATCH / (time.perf_counter() - t0)
    pred_dec = None
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
                # every family against the ceiling of the scheme the LAUNCHED kernel uses (keyed on the wrapper's profile tag)
                for key, peak, what, pmc_name in (
                        ("quadconv16", PEAK_F16X2_TFLOPS, "lk::quadform_conv_planes_kernel (lk_kron_quadform_shared_planes_f16x2): per-layer quadratic "
                         "form of the weight-sharing Jacobian on split fp16 planes, three fp16 MFMAs per fp32 product block", "lk::quadform_conv_planes"),
                        ("quadconv", PEAK_BF16X3_TFLOPS, "lk::quadform_conv_kernel (fp32 operands split in flight): six bf16 MFMAs per fp32 "
                         "product block", "lk::quadform_conv_kernel"),
                        ("conv16", PEAK_F16X2_TFLOPS, "lk::conv_f16x2_kernel / conv_winp / conv_strided (forward + reverse sweep of the 10 "
                         "identity seeds, eigenbasis rotations)", "lk::conv_")):
                    evs = pprof.get(key, []) + (pprof.get("convp16", []) + pprof.get("convs16", []) if key == "conv16" else [])
                    ms_k = sum(ev[0].elapsed_time(ev[1]) for ev in evs)
                    if evs and ms_k > 0:
                        tf = sum(ev[2] for ev in evs) / (ms_k * 1e-3) / 1e12
                        traffic, traffic_src = pmc_traffic(pmc_name, table="predictive_c4", per="call")
                        pfam[key] = {"kernel": what, "bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s",
                                     "frac": tf / peak, "ms_per_call": ms_k, "launches": len(evs), "traffic": traffic,
                                     "traffic_unit": "HBM bytes per predictive call (all launches of the family)", "traffic_source": traffic_src}
                # the family to look at first: the most time lost to the distance from its own roof
                dom = max(pfam, key=lambda k: pfam[k]["ms_per_call"] * (1.0 - pfam[k]["frac"])) if pfam else None
                result["predictive_kron_c4"] = {
                    "workload": "c4 posterior: ResNet-18 full-network KFAC, GLM predictive variance [B,10,10], batch 128",
                    "samples_per_s": pred_rate, "ms_per_call": pred_ms, "finite": bool(torch.isfinite(f_var).all()),
                    "roofline": dict(pfam[dom], family=dom) if dom else None, "roofline_families": pfam}
                result["predictive_samples_per_s"] = {"c4_kron_full_network": pred_rate}  # the metric's second half, top level
                pred_dec = dec  # (its CPU baseline runs LAST, with the other one: 128 host threads for 12 s right in front of
                                # the 50 000-sample leg, whose host side is 60 - 85 % of its device time, is no fair start)
            del dec
        if not args.no_predictive and not SELFTEST:
            result["predictive"] = predictive_leg(dev)
            result.setdefault("predictive_samples_per_s", {})["c3_dense_last_layer"] = result["predictive"]["predictive_samples_per_s"]
        if not args.no_extras and not SELFTEST:
            # twice: the first fit of this process state pays the device allocations of its working set (~280 hipMalloc calls,
            # 32 GiB: 0.05 - 0.9 s from box to box and run to run); the second runs in the pool the first left, which is
            # what every fit after the first of a process sees.  The line carries the second, and the first beside it.
            cold = fit_50k_leg(backend, dev)
            result["fit_50k"] = fit_50k_leg(backend, dev)
            result["fit_50k"]["first_fit_of_the_process"] = {k: cold[k] for k in ("wall_s", "accumulate_s", "decompose_s", "samples_per_s", "allocator")}
            pw = result["fit_50k"].get("power")
            if pw and isinstance(result.get("roofline"), dict) and result["roofline"].get("unit") == "TFLOP/s":
                # the same fraction against the matrix peak AT THE CLOCK THE CHIP RAN (the fit holds the socket at its power
                # budget: the clock is what gives) — `frac` stays the fraction of the nominal 2.4 GHz peak
                result["roofline"]["sclk_ghz_during_fit"] = pw["sclk_ghz_median"]
                result["roofline"]["socket_w_during_fit"] = pw["socket_w_median"]
                result["roofline"]["frac_of_peak_at_measured_clock"] = result["roofline"]["frac"] * 2.4 / max(pw["sclk_ghz_median"], 0.1)
            # what a fit pays besides its minibatches (the verdict of round 3 asked for it on the line): the timed K steps
            # against the steady-state rate of the 391-minibatch fit, and `finalize` alone behind a drained device
            steady = result["fit_50k"]["accumulate_s"] / result["fit_50k"]["minibatches"] * 1e3
            torch.cuda.synchronize()
            t_s = time.perf_counter()
            acc_f = backend.kron_accumulator(N_DATASET)
            for i in range(2):  # (one minibatch per lane: allocation and zeroing of the factor / pixel-pair buffers, pipeline fill)
                acc_f.add_batch(*batches[i % len(batches)])
            torch.cuda.synchronize()
            setup_ms = (time.perf_counter() - t_s) * 1e3 - 2 * steady
            for i in range(2, 8):
                acc_f.add_batch(*batches[i % len(batches)])
            torch.cuda.synchronize()
            t_f = time.perf_counter()
            acc_f.finalize()
            torch.cuda.synchronize()
            result["fit_fixed_cost"] = {"steady_ms_per_step": steady, "timed_ms_per_step": dt / args.steps * 1e3,
                                        "fixed_ms_per_fit": dt * 1e3 - args.steps * steady,
                                        "setup_ms_first_two_minibatches_minus_two_steady_steps": setup_ms,
                                        "finalize_ms_behind_a_drained_device": (time.perf_counter() - t_f) * 1e3,
                                        "note": "steady = the 391-minibatch fit's average, which runs 1 - 2 % slower than a short burst of steps "
                                                "(power / temperature): a negative fixed_ms_per_fit says the timed burst was faster than that "
                                                "average, not that a fit costs less than its minibatches"}
            result["other_configs"] = small_config_legs(dev)
        if not args.no_cpu_baseline:
            if pred_dec is not None:
                result["predictive_kron_c4"]["cpu_baseline"] = cpu_predictive_baseline(pred_dec, 1.0, args.cpu_seconds)
            result["cpu_baseline"] = cpu_baseline(0.0 if SELFTEST else args.cpu_seconds)
        pred_dec = None
    if world > 1:
        # the fit's ONE collective: message size against what the model's factor shapes say it must be (a wrong pack /
        # a missed factor shows here, not as a silently wrong posterior), and its stand-alone time on the same buffer size
        from laplace_amd.laplace import expected_exchange_bytes

        info = getattr(fit_steps, "last_exchange", None) or {}
        want = expected_exchange_bytes(model)
        buf = torch.zeros(max(want // 4, 1), dtype=torch.float32, device=dev)
        dist.all_reduce(buf)
        barrier()
        t0 = time.perf_counter()
        for _ in range(5):
            dist.all_reduce(buf)
        barrier()
        t_ar = (time.perf_counter() - t0) / 5
        if rank == 0:
            result["allreduce"] = {"bytes": info.get("bytes"), "expected_bytes": want, "ok": info.get("bytes") == want,
                                   "ms": t_ar * 1e3, "bus_GBps": 2 * (world - 1) / world * want / t_ar / 1e9,
                                   "note": "one all-reduce per fit (packed upper triangles of the 43 factors + loss)"}
            assert info.get("bytes") == want, f"curvature exchange moved {info.get('bytes')} bytes, expected {want}"
        del buf
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
