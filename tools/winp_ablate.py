"""Ablation builds of the persistent window kernel (make variant VARIANT=ablN VARIANT_FLAGS=-DLK_WINP_ABLATE=N: 2 no MFMAs,
4 no in-loop staging, 8 loads never waited for) with one or two workgroups per CU: what a workgroup ALONE on its CU is bound
by.  512 channels, 1024 images of 4 x 4: 512 tiles = one per workgroup at two per CU, two per workgroup at one per CU."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ROOT)
    from torch import nn
    from laplace_amd import conv as cv
    from laplace_amd._lib import get_kernels
    K = get_kernels()
    Ci = Co = int(sys.argv[2]); H = int(sys.argv[3]); N = int(sys.argv[4])
    torch.manual_seed(0)
    m = nn.Conv2d(Ci, Co, 3, 1, 1, bias=False).cuda()
    g = K.split_f16x2((torch.randn(N, H, H, Co, device="cuda") * 1e-3).contiguous())
    add = K.split_f16x2((torch.randn(N, H, H, Ci, device="cuda") * 1e-2).contiguous())
    mask = (torch.rand(N, H, H, Ci, device="cuda") > 0.5).to(torch.uint8)
    prep = cv.PreparedConv(m)
    for name, cfg in (("two per CU", 2 | (1 << 25) | (1 << 20)), ("one per CU", 2 | (1 << 25) | (1 << 19))):
        K.conv_config = cfg
        for _ in range(3):
            cv.conv_backward_data_vjp(prep, g, (H, H), add=add, mult=mask)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            cv.conv_backward_data_vjp(prep, g, (H, H), add=add, mult=mask)
        e1.record(); torch.cuda.synchronize()
        print("  %s: %.1f us" % (name, e0.elapsed_time(e1) * 50), flush=True)
else:
    for shape in (("512", "4", "1024"),):
        for lib in ("", "_abl12", "_abl28", "_abl44", "_abl60"):
            env = dict(os.environ, LK_LIB=os.path.join(ROOT, "laplace_amd/csrc/liblaplace_hip%s.so" % lib))
            print("channels %s map %s images %s | build %s" % (*shape, lib or "full"), flush=True)
            subprocess.run([sys.executable, __file__, "child", *shape], env=env)
