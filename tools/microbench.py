"""Kernel microbenchmarks on the MI355X (run through gpurun): Gram engine at the ResNet-18 / c4
shapes, eigensolver time + accuracy.  Writes gpurun_out/microbench.json.  Development tool."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd._lib import HipKernels  # noqa: E402

K = HipKernels()
DEV = "cuda"
out = {"gram": [], "eig": []}


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


which = sys.argv[1] if len(sys.argv) > 1 else "all"

if which in ("all", "gram"):
    B, C = 128, 10
    conv_cases = [("l1 3x3 s1", 64, 32, 1), ("l2.0 3x3 s2", 64, 32, 2), ("l2 3x3 s1", 128, 16, 1),
                  ("l3.0 3x3 s2", 128, 16, 2), ("l3 3x3 s1", 256, 8, 1), ("l4.0 3x3 s2", 256, 8, 2),
                  ("l4 3x3 s1", 512, 4, 1)]
    for name, cin, hw, s in conv_cases:
        x = torch.randn(B, cin, hw, hw, device=DEV)
        n = cin * 9
        oh = (hw + 2 - 3) // s + 1
        Kr = B * oh * oh
        A = torch.zeros(n, n, device=DEV)
        for native, sc in ((False, False), (True, False), (True, True)):
            K.use_shiftcorr = sc
            ms = timeit(lambda: K.gram_conv(x, 3, s, 1, 1, 1.0, A, upper_only=native, native=native))
            out["gram"].append({"op": "conv" + ("_fused" if native else "") + ("_shiftcorr" if sc else ""), "name": name,
                                "n": n, "K": Kr, "ms": ms, "tflops_full": 2 * Kr * n * n / ms / 1e9})
            print(out["gram"][-1], flush=True)
        K.use_shiftcorr = True
    for name, n, L in [("G l1", 64, 1024), ("G l2", 128, 256), ("G l3", 256, 64), ("G l4", 512, 16)]:
        g = torch.randn(C * B, n, L, device=DEV)
        G = torch.zeros(n, n, device=DEV)
        ms = timeit(lambda: K.gram_nt(g, 1.0, G))
        Kr = C * B * L
        out["gram"].append({"op": "nt", "name": name, "n": n, "K": Kr, "ms": ms, "tflops_full": 2 * Kr * n * n / ms / 1e9})
        print(out["gram"][-1], flush=True)
    for name, Kr, n in [("fc A", 128, 512), ("fc G", 1280, 10), ("big", 8192, 4608), ("tall", 131072, 576), ("LL Y", 512, 5130)]:
        X = torch.randn(Kr, n, device=DEV)
        Cm = torch.zeros(n, n, device=DEV)
        ms = timeit(lambda: K.gram_tn(X, 1.0, Cm))
        out["gram"].append({"op": "tn", "name": name, "n": n, "K": Kr, "ms": ms, "tflops_full": 2 * Kr * n * n / ms / 1e9})
        print(out["gram"][-1], flush=True)
        ms = timeit(lambda: torch.mm(X.T, X))
        out["gram"].append({"op": "torch.mm(rocBLAS)", "name": name, "n": n, "K": Kr, "ms": ms,
                            "tflops_full": 2 * Kr * n * n / ms / 1e9})
        print(out["gram"][-1], flush=True)

if which in ("all", "eig"):
    sizes = [64, 128, 256, 512, 576, 1152, 2304, 4608]
    if len(sys.argv) > 2:
        sizes = [int(a) for a in sys.argv[2:]]
    for n in sizes:
        torch.manual_seed(n)
        X = torch.randn(2 * n + 7, n, device=DEV, dtype=torch.float64)
        A64 = X.T @ X / (2 * n)
        A = A64.float().contiguous()
        t0 = time.time()
        w, Q, info = K.syevj(A)
        torch.cuda.synchronize()
        first = (time.time() - t0) * 1e3
        ms = timeit(lambda: K.syevj(A), reps=2, warm=0)
        wref = torch.linalg.eigvalsh(A64)
        Q64, w64 = Q.double(), w.double()
        scale = wref.abs().max().item()
        rec = ((Q64 * w64) @ Q64.T - A64).abs().max().item() / scale
        orth = (Q64.T @ Q64 - torch.eye(n, device=DEV, dtype=torch.float64)).abs().max().item()
        val = (w64 - wref).abs().max().item() / scale
        t_ref = timeit(lambda: torch.linalg.eigh(A), reps=2, warm=1)
        Ac = A.cpu()
        t0 = time.time()
        torch.linalg.eigh(Ac)
        t_cpu = (time.time() - t0) * 1e3
        rec_ = {"n": n, "ms": ms, "first_ms": first, "info": int(info[0].item()), "rec_err": rec, "orth_err": orth, "val_err": val,
                "torch_eigh_gpu_ms": t_ref, "torch_eigh_cpu_ms": t_cpu}
        out["eig"].append(rec_)
        print(rec_, flush=True)

os.makedirs("gpurun_out", exist_ok=True)
with open(f"gpurun_out/microbench_{which}.json", "w") as fh:
    json.dump(out, fh, indent=1)
