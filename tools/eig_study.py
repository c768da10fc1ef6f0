"""Eigensolver study on real c4 KFAC factors (development tool): host-enqueue vs device time of
HipKron.decompose, per-size single-stream time and sweep counts."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd._lib import get_kernels  # noqa: E402
from laplace_amd.backend import HipGGN  # noqa: E402
from laplace_amd.nets import ResNet18  # noqa: E402

dev = "cuda"
torch.manual_seed(711)
model = ResNet18().to(dev).eval()
backend = HipGGN(model, "classification")
acc = backend.kron_accumulator(50000)
for _ in range(int(os.environ.get("STEPS", "6"))):
    acc.add_batch(torch.randn(128, 3, 32, 32, device=dev), torch.randint(0, 10, (128,), device=dev))
loss, H = acc.finalize()
torch.cuda.synchronize()
K = get_kernels()
out = {"inner": os.environ.get("LK_EIG_INNER", "3")}
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dec = H.decompose(n_streams=int(os.environ.get('N_STREAMS', '6')))
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    out[f"decompose_rep{rep}"] = {"host_enqueue_ms": t_host * 1e3, "total_ms": t_all * 1e3,
                                  "converged": all(int(i[0].item()) == 0 for i in dec._eig_info),
                                  "sweeps": [int(i[1].item()) for i in dec._eig_info]}
    del dec
seen = set()
single = []
for F in H.kfacs:
    for Hi in F:
        n = Hi.shape[0]
        if Hi.ndim < 2 or n in seen or n < 500:
            continue
        seen.add(n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        w, Q, info = K.syevj(Hi.contiguous(), clamp=True)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        A64 = Hi.double()
        rec = ((Q.double() * w.double()) @ Q.double().T - A64).abs().max().item() / A64.abs().max().item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        wr, Qr = torch.linalg.eigh(Hi)
        e1.record()
        torch.cuda.synchronize()
        rec_r = ((Qr.double() * wr.double()) @ Qr.double().T - A64).abs().max().item() / A64.abs().max().item()
        lam = wr.abs().max().item()
        single.append({"rocsolver_rec_err": rec_r, "max_abs_over_lambda_max": A64.abs().max().item() / lam,"n": n, "host_ms": t_host * 1e3, "total_ms": t_all * 1e3, "sweeps": int(info[1].item()),
                       "info": int(info[0].item()), "rec_err": rec, "rocsolver_ms": e0.elapsed_time(e1)})
        print(single[-1], flush=True)
out["single"] = single
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/eig_study_inner{out['inner']}_s{os.environ.get('N_STREAMS', '6')}.json", "w"), indent=1)
