"""Which wrapper calls a steady-state c4 step makes (name, stream, bytes): where the small main-stream launches come from."""
import os, sys, collections, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd._lib import get_kernels
from laplace_amd.nets import ResNet18
torch.manual_seed(711)
model = ResNet18(10).cuda().eval()
b = HipGGN(model, "classification")
X = torch.randn(128, 3, 32, 32, device="cuda"); y = torch.randint(10, (128,), device="cuda")
acc = b.kron_accumulator(50000)
for _ in range(9): acc.add_batch(X, y)
torch.cuda.synchronize()
K = get_kernels()
main = torch.cuda.current_stream().cuda_stream
log = collections.Counter()
for name in ("absmax", "split_f16x2", "nchw_to_nhwc", "vjp_nhwc_split", "bn_act_forward_nhwc"):
    orig = getattr(K, name)
    def wrap(*a, _o=orig, _n=name, **k):
        t = a[0] if torch.is_tensor(a[0]) else None
        st = "main" if torch.cuda.current_stream().cuda_stream == main else "side"
        fr = [f for f in traceback.extract_stack()[:-1] if "laplace_amd" in f.filename and "_lib" not in f.filename]
        where = f"{fr[-1].filename.split('/')[-1]}:{fr[-1].lineno}" if fr else "?"
        log[(_n, st, where, tuple(t.shape) if t is not None else None)] += 1
        return _o(*a, **k)
    setattr(K, name, wrap)
for _ in range(8): acc.add_batch(X, y)
torch.cuda.synchronize()
for k, v in sorted(log.items(), key=lambda kv: -kv[1]): print(v / 8.0, k)
