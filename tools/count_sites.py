"""Which host-side torch calls put fill / copy / element-wise kernels into a steady-state c4 fit step: call sites inside
laplace_amd/ of the tensor factories and in-place ops, counted over ONE minibatch (real kernels).  Development tool."""
import collections, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd.nets import ResNet18

torch.manual_seed(711)
model = ResNet18(10).cuda().eval()
b = HipGGN(model, "classification")
X = torch.randn(128, 3, 32, 32, device="cuda"); y = torch.randint(10, (128,), device="cuda")
acc = b.kron_accumulator(50000)
for _ in range(4):
    acc.add_batch(X, y)
torch.cuda.synchronize()
sites = collections.Counter()


def wrap(obj, name):
    orig = getattr(obj, name)

    def f(*a, **k):
        st = traceback.extract_stack()[:-1]
        fr = [s for s in st if "/laplace_amd/" in s.filename]
        if fr:
            s = fr[-1]
            sites[(name, s.filename.split("/")[-1], s.lineno, (s.line or "")[:100])] += 1
        return orig(*a, **k)

    setattr(obj, name, f)


for n in ("zeros", "full", "zeros_like", "ones", "cat", "stack", "where", "exp2", "matmul", "mm", "addmm"):
    wrap(torch, n)
wrap(torch.nn.functional, "linear")
for n in ("zero_", "fill_", "clone", "copy_", "contiguous", "new_zeros", "to", "float", "mul_", "add_", "__mul__", "__add__",
          "__sub__", "__truediv__", "sum", "amax", "abs", "mean", "index_select", "__getitem__", "__setitem__", "matmul", "mm", "t"):
    wrap(torch.Tensor, n)
acc.add_batch(X, y)
acc.add_batch(X, y)
torch.cuda.synchronize()
for k, v in sorted(sites.items(), key=lambda kv: -kv[1]):
    if k[0] in ("__getitem__", "t"):
        continue
    print(v / 2, k)
