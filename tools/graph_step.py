"""Can the c4 step be captured into a HIP graph (torch.cuda.CUDAGraph) and what does replay buy?  (development probe)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laplace_amd import HipGGN
from laplace_amd.nets import ResNet18
torch.manual_seed(711)
model = ResNet18(10).cuda().eval()
b = HipGGN(model, "classification")
G = 16  # two lanes x the pixel-pair group size of 8: capture G steps so that the host-side state machine repeats
Xs = [torch.randn(128, 3, 32, 32, device="cuda") for _ in range(G)]
ys = [torch.randint(10, (128,), device="cuda") for _ in range(G)]
acc = b.kron_accumulator(50000)
for _ in range(2):
    for i in range(G): acc.add_batch(Xs[i], ys[i])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    for i in range(G): acc.add_batch(Xs[i], ys[i])
torch.cuda.synchronize()
print("eager ms/step", 1e3 * (time.perf_counter() - t0) / (5 * G))
def join_all():
    """the calling stream waits for every lane and side stream (a captured region must end with all forked work joined)"""
    cur = torch.cuda.current_stream()
    for sub in (acc._lane_accs or []):
        if sub._lane_stream is not None:
            with torch.cuda.stream(sub._lane_stream):
                sub._join_side()
            cur.wait_stream(sub._lane_stream)


g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.stream(s):
        for i in range(G): acc.add_batch(Xs[i], ys[i])  # warm-up on the capture stream
        join_all()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for i in range(G): acc.add_batch(Xs[i], ys[i])
            join_all()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    print("graph ms/step", 1e3 * (time.perf_counter() - t0) / (10 * G))
except Exception as e:
    import traceback; traceback.print_exc()
    print("capture failed:", type(e).__name__, str(e)[:300])
