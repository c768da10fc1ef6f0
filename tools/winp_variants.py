"""Persistent window kernel, traffic variants (round 5): per c4 layer shape the fused backward-data launch as round 4 staged
it (whole 352-pixel window: config bit 30), with the unread halo / padding slots pointed at read pixels (default), and
with c = 2 / 4 column tiles of a pixel tile sharing an XCD (config bits 28-29) — time per launch and the result against
the generic kernel's.  Development tool: `python tools/winp_variants.py [--json out]`."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch import nn
from laplace_amd import conv as cv
from laplace_amd._lib import get_kernels

K = get_kernels()
dev = "cuda"
WP = 1 << 27


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


out = {}
cases = [(64, 64, 32, 1152), (128, 128, 16, 1152), (256, 256, 8, 1152), (512, 512, 4, 1152)]
for Co, Ci, H, N in cases:  # conv Ci -> Co; backward-data: cotangent [N, H, H, Co] -> [N, H, H, Ci]
    torch.manual_seed(0)
    m = nn.Conv2d(Ci, Co, 3, 1, 1, bias=False).to(dev)
    g = K.split_f16x2((torch.randn(N, H, H, Co, device=dev) * 1e-3).contiguous())
    add = K.split_f16x2((torch.randn(N, H, H, Ci, device=dev) * 1e-2).contiguous())
    mask = (torch.rand(N // 9, H, H, Ci, device=dev) > 0.5).to(torch.uint8)
    prep = cv.PreparedConv(m)
    gf = 2.0 * N * H * H * Co * Ci * 9 / 1e9
    K.conv_config = 2 | WP
    ref = cv.conv_backward_data_vjp(prep, g, (H, H), add=add, mult=mask)
    variants = [("generic", 2 | WP), ("r04 window", 2 | (1 << 30)), ("default", 2), ("coloc 2", 2 | (1 << 28)), ("coloc 4", 2 | (2 << 28)),
                ("coloc 8", 2 | (3 << 28)), ("r04 window + coloc 2", 2 | (1 << 30) | (1 << 28))]
    for rep in range(2):  # twice, interleaved: the chip's clock drifts over a run
        for name, cfg in variants:
            K.conv_config = cfg
            o = cv.conv_backward_data_vjp(prep, g, (H, H), add=add, mult=mask)
            torch.cuda.synchronize()
            d = (o.float() - ref.float()).abs().max().item() / ref.float().abs().max().item()
            same = bool(torch.equal(o.planes, ref.planes)) if name != "generic" else True
            t = timeit(lambda: cv.conv_backward_data_vjp(prep, g, (H, H), add=add, mult=mask))
            out.setdefault(f"Ci{Ci}_Co{Co}_{H}x{H}", {}).setdefault(name, []).append(t)
            print(f"Ci={Ci} Co={Co} {H}x{H} N={N} {gf:.0f} GFLOP | {name:22s} {t:7.1f} us {gf / t * 1e3:6.1f} TFLOP/s  rel diff {d:.1e} bitwise {same}", flush=True)
    K.conv_config = 2
if "--json" in sys.argv:
    with open(sys.argv[sys.argv.index("--json") + 1], "w") as fh:
        json.dump(out, fh, indent=1)
