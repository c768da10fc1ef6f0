"""GPU reverse passes vs CPU fp64 truth, repeated; with and without cudnn.deterministic (development tool)."""
import copy, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd.capture import Tape
from laplace_amd.nets import ResNet18

torch.manual_seed(711)
model_cpu = ResNet18(10).eval()
g = torch.Generator().manual_seed(1)
X = torch.randn(16, 3, 32, 32, generator=g)
params = [p for p in model_cpu.parameters() if p.requires_grad]

def grads(model, X, dtype):
    tape = Tape(model, [p for p in model.parameters() if p.requires_grad])
    f = tape.forward(X)
    p = torch.softmax(f.detach(), -1)
    S = torch.diag_embed(p.sqrt()) - p.unsqueeze(2) * p.sqrt().unsqueeze(1)
    out = tape.output_grads(f, S.permute(2, 0, 1).contiguous())
    names = [t.name for t in tape.taps]
    tape.release()
    return names, out

m64 = copy.deepcopy(model_cpu).double()
names, truth = grads(m64, X.double(), torch.float64)
model = copy.deepcopy(model_cpu).cuda()
Xg = X.cuda()

def report(tag):
    _, gg = grads(model, Xg, torch.float32)
    errs = [((a.double().cpu() - t).abs().max() / t.abs().max()).item() for a, t in zip(gg, truth)]
    bad = [(n, f"{e:.1e}") for n, e in zip(names, errs) if e > 1e-4]
    print(f"{tag}: worst {max(errs):.1e} bad={bad[:8]}", flush=True)

for det in (False, True):
    torch.backends.cudnn.deterministic = det
    for i in range(6):
        report(f"deterministic={det} run{i}")
        # perturb allocator state between runs
        junk = [torch.randn(1 << (20 + (i % 4)), device="cuda") for _ in range(3)]
        del junk
torch.backends.cudnn.deterministic = False
torch.backends.cudnn.benchmark = True
for i in range(4):
    report(f"benchmark=True run{i}")
