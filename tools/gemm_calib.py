"""Calibration: what hipBLASLt sustains for plain fp16 GEMMs on this box, on the GEMM shapes the c4 convolutions reduce to
(M = pixels, N = Cin, K = Cout * taps) — the practical ceiling a hand-written fp16 MFMA kernel can be compared with.
Development tool; writes gpurun_out/gemm_calib.json."""
import json
import os
import time

import torch

dev = "cuda"
shapes = [(8192, 8192, 8192), (1152 * 1024, 64, 576), (1152 * 256, 128, 1152), (1152 * 64, 256, 2304), (1152 * 16, 512, 4608)]
rows = []
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    b = torch.randn(K, N, device=dev, dtype=torch.float16)
    for _ in range(3):
        a @ b
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        a @ b
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    tf = 2.0 * M * N * K / ms / 1e9
    rows.append({"M": M, "N": N, "K": K, "ms": ms, "fp16_tflops": tf, "frac_of_2500": tf / 2500, "as_fp16x2_equiv_tflops": tf / 3})
    print(json.dumps(rows[-1]))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/gemm_calib.json", "w"), indent=1)
