"""How many ReLU masks of a c4 forward differ from the fp64 forward — the split-fp16 sweep's (22-bit operands) next to a stock
fp32 forward on the same device — per layer, with the error of the pre-activations that decides them.  Development tool
(round 5: why a 64-sample minibatch moves one G block by 2.5e-4 while the fp32 CPU oracle sits at 1.6e-7 from fp64)."""
import copy, json, os, sys
import torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd._lib import get_kernels
from laplace_amd.nets import ResNet18
from laplace_amd.sweep_nhwc import SplitSweep

dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(711)
m32 = ResNet18(10, act=torch.relu).eval()
m64 = copy.deepcopy(m32).double()
g = torch.Generator().manual_seed(1000)
X = torch.randn(128, 3, 32, 32, generator=g)[:B]


def bn_outputs(model, x):
    outs = {}
    hs = [mod.register_forward_hook(lambda m_, i, o, n=n: outs.__setitem__(n, o.detach())) for n, mod in model.named_modules()
          if isinstance(mod, (nn.BatchNorm2d, nn.Conv2d))]
    with torch.no_grad():
        model(x)
    for h in hs:
        h.remove()
    return outs


ref = bn_outputs(m64, X.double())                       # fp64, CPU
gpu32 = bn_outputs(copy.deepcopy(m32).to(dev), X.to(dev))  # stock fp32 on the device (library convolutions)
cpu32 = bn_outputs(m32, X)                              # stock fp32 on the CPU
mdev = copy.deepcopy(m32).to(dev)
taps = {n: mod for n, mod in mdev.named_modules() if isinstance(mod, (nn.Conv2d, nn.Linear))}
sw = SplitSweep(mdev, taps, kernels=get_kernels)
sw.forward(X.to(dev))
ours = {n: sw.taps[n]["a"] for n in taps if sw.taps[n].get("a") is not None}  # inputs of every tapped module = post-activation maps
ins64 = {}
hs = [mod.register_forward_hook(lambda m_, i, o, n=n: ins64.__setitem__(n, i[0].detach())) for n, mod in m64.named_modules() if isinstance(mod, (nn.Conv2d, nn.Linear))]
ins_g32, ins_c32 = {}, {}
m_g = copy.deepcopy(m32).to(dev)
hs += [mod.register_forward_hook(lambda m_, i, o, n=n: ins_g32.__setitem__(n, i[0].detach())) for n, mod in m_g.named_modules() if isinstance(mod, (nn.Conv2d, nn.Linear))]
hs += [mod.register_forward_hook(lambda m_, i, o, n=n: ins_c32.__setitem__(n, i[0].detach())) for n, mod in m32.named_modules() if isinstance(mod, (nn.Conv2d, nn.Linear))]
with torch.no_grad():
    m64(X.double()); m_g(X.to(dev)); m32(X)
out = {}
print(f"{'input of':26s} {'elements':>10s} | masks differing from fp64: {'ours':>6s} {'fp32 gpu':>9s} {'fp32 cpu':>9s} | max |err| / rms: ours, fp32 gpu, fp32 cpu")
for n in taps:
    if n not in ours or n == "conv1":
        continue
    w = ins64[n]
    rms = w.pow(2).mean().sqrt().item()
    row = {}
    for tag, t in (("ours", ours[n]), ("gpu32", ins_g32[n]), ("cpu32", ins_c32[n])):
        t = t.double().cpu().reshape(w.shape)
        row[tag] = {"flips": int(((t > 0) != (w > 0)).sum()), "err_over_rms": (t - w).abs().max().item() / rms}
    out[n] = dict(row, elements=w.numel())
    print(f"{n:26s} {w.numel():10d} | {row['ours']['flips']:6d} {row['gpu32']['flips']:9d} {row['cpu32']['flips']:9d} | "
          f"{row['ours']['err_over_rms']:.1e} {row['gpu32']['err_over_rms']:.1e} {row['cpu32']['err_over_rms']:.1e}")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r05_mask_flips.json", "w"), indent=1)
