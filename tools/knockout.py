"""Marginal cost of a kernel family INSIDE the overlapped c4 step: steady-state steps with one family's launches skipped
(wrong factors, timing only).  The step is power-bound and its kernels overlap on several streams, so a family's stand-alone
kernel time says little about what removing (or fusing) it would buy.  usage: knockout.py [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd._lib import get_kernels
from laplace_amd.nets import ResNet18

torch.manual_seed(711)
model = ResNet18(10).cuda().eval()
b = HipGGN(model, "classification")
K = get_kernels()
X = torch.randn(128, 3, 32, 32, device="cuda"); y = torch.randint(10, (128,), device="cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def run():
    acc = b.kron_accumulator(50000)
    for _ in range(8):
        acc.add_batch(X, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        acc.add_batch(X, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    for sub in (acc._lane_accs or []):  # (drop the state without the once-per-fit work)
        sub._pix, sub._pix_pending = {}, {}
    acc._lane_accs = None
    return dt


cases = [("nothing", []), ("G factors (gram_split)", ["gram_tn_f16x2"]), ("pixel-pair products", ["pixpair_accumulate_split"]),
         ("fp32-MFMA A factors (gram_conv)", ["gram_conv"]), ("copies into the pixel-pair stacks (copy_absmax)", ["copy_absmax"]),
         ("patch matrices + their Grams (im2col_split)", ["im2col_split"]), ("all three", ["gram_tn_f16x2", "pixpair_accumulate_split", "gram_conv"]),
         ("nothing", [])]
base = None
for name, names in cases:
    names = [m for m in names if hasattr(K, m)]
    saved = {m: getattr(type(K), m) for m in names}
    for m in names:
        setattr(type(K), m, lambda self, *a, **k: None)
    try:
        run()
        ms = min(run() for _ in range(2))
    finally:
        for m, f in saved.items():
            setattr(type(K), m, f)
    base = base or ms
    print("without %-36s %.2f ms per step (%+.2f)" % (name + ":", ms, ms - base), flush=True)
