"""A/B of the forward's fused convolution + BatchNorm / add / ReLU launches (SplitSweep.fuse_conv_bn) on the timed
configuration: steady-state c4 fit steps (two lanes) and serial steps (one stream), alternating, in one process.
usage: fuse_ab.py [N]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd.nets import ResNet18
from laplace_amd.sweep_nhwc import SplitSweep

torch.manual_seed(711)
model = ResNet18(10).cuda().eval()
X = torch.randn(128, 3, 32, 32, device="cuda"); y = torch.randint(10, (128,), device="cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48


backend = HipGGN(model, "classification")  # (ONE backend: its lane / side streams — and their allocator pools — are reused)


def run(fuse, serial):
    SplitSweep.fuse_conv_bn = fuse
    b = backend
    acc = b.kron_accumulator(50000, overlap=not serial)
    if serial:
        acc.lanes = 1
    for _ in range(6): acc.add_batch(X, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): acc.add_batch(X, y)
    t_host = 1e3 * (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t = 1e3 * (time.perf_counter() - t0) / n
    print(f"    fuse={fuse} serial={serial}: host {t_host:.2f} ms/step, reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB, "
          f"allocated {torch.cuda.memory_allocated() / 2**30:.1f} GiB", flush=True)
    acc.finalize()
    torch.cuda.synchronize()
    return t


for rep in range(3):
    for serial in (False, True):
        a, b_ = run(True, serial), run(False, serial)
        print(f"rep {rep} {'serial' if serial else 'lanes '}: fused {a:.3f} ms/step, two launches {b_:.3f}", flush=True)
