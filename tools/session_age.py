"""Why is the SECOND process of a box session slower?  Steady-state c4 steps in blocks of 50: host enqueue time and wall
per block, free device memory at the start, reserved at the end (development tool; DESIGN section 4, 'an oddity').
usage: session_age.py [blocks] [nosync] [ahead=N]     nosync: no synchronisation between the blocks (the host runs ahead
as in a long fit); ahead=N: KronAccumulator.max_ahead (0: unbounded)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd.nets import ResNet18
torch.manual_seed(711)
model = ResNet18(10).cuda().eval()
b = HipGGN(model, "classification")
X = torch.randn(128, 3, 32, 32, device="cuda"); y = torch.randint(10, (128,), device="cuda")
free0 = torch.cuda.mem_get_info()[0] / 2**30
nosync = "nosync" in sys.argv
for a in sys.argv:
    if a.startswith("ahead="):
        from laplace_amd.backend import KronAccumulator
        KronAccumulator.max_ahead = int(a[6:])
acc = b.kron_accumulator(50000)
for _ in range(4): acc.add_batch(X, y)
torch.cuda.synchronize()
out = []
T0 = time.perf_counter()
for blk in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    t0 = time.perf_counter()
    for _ in range(50): acc.add_batch(X, y)
    t1 = time.perf_counter()
    if not nosync:
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    out.append(f"{1e3 * (t1 - t0) / 50:.2f}/{1e3 * (t2 - t0) / 50:.2f}")
torch.cuda.synchronize()
t_all = time.perf_counter()
print("enqueue/wall ms per step, blocks of 50:", " ".join(out), f"| free at start {free0:.0f} GiB, reserved {torch.cuda.memory_reserved() / 2**30:.0f} GiB",
      f"| whole run {1e3 * (t_all - T0) / (50 * len(out)):.2f} ms/step" if nosync else "", flush=True)
