"""Per-shape timing of the split-tensor Gram (lk_gram_tn_f16x2) on the c4 cotangent shapes (batch 9 x 128)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd._lib import get_kernels
K = get_kernels()
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
tot = 0
for C, hw, cnt in ((64, 1024, 5), (128, 256, 5), (256, 64, 5), (512, 16, 5)):
    R = 1152 * hw
    x = K.split_f16x2(torch.randn(R, C, device="cuda"))
    G = torch.zeros(C, C, device="cuda")
    ms = timeit(lambda: K.gram_tn_f16x2(x, 1.0, G))
    tot += cnt * ms
    print(json.dumps({"C": C, "R": R, "ms": round(ms, 4), "GBs": round(4.0 * R * C / ms / 1e6, 1),
                      "tf_half": round(R * C * (C + 1) / ms / 1e9, 1)}))
print("per step ms", round(tot, 3))
