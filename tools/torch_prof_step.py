"""torch.profiler view of one fit step (which aten ops launch the remaining non-lk kernels). Development tool."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN  # noqa: E402
from laplace_amd.nets import ResNet18  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
model = ResNet18().to(dev).eval()
b = HipGGN(model, "classification")
acc = b.kron_accumulator(50000)
X, y = torch.randn(128, 3, 32, 32, device=dev), torch.randint(0, 10, (128,), device=dev)
for _ in range(3):
    acc.add_batch(X, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    acc.add_batch(X, y)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=50,
                                                            max_shapes_column_width=70))
for evt in prof.events():
    if evt.name in ("aten::copy_", "aten::contiguous", "aten::clone") and evt.device_time_total > 20:
        print("COPY", evt.name, evt.input_shapes, round(evt.device_time_total), [s for s in (evt.stack or [])[:6]])
