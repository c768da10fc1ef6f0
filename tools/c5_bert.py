"""Config c5 on the MI355X (development tool): BERT-base (random init, `BertConfig()`), sequence length 128, batch
32, two labels, HuggingFace-style dict batches; last-layer KFAC fit, then the marginal-likelihood prior sweep
(`optimize_prior_precision`, 100 Adam steps) and the 100-point validation gridsearch over `logspace(-4, 4)`
(docs/huggingface_example.md and baselaplace.py:466-561 of the reference).  `--tiny` shrinks the encoder so the script
can be exercised on the CPU kernel emulation (`--device cpu`)."""
import argparse
import json
import os
import sys
import time

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--device", default="cuda")
ap.add_argument("--tiny", action="store_true")
ap.add_argument("--n", type=int, default=2048)
ap.add_argument("--n-val", type=int, default=512)
args = ap.parse_args()
dev = args.device
if dev == "cpu":
    from laplace_amd import _lib
    from tests.emulated_kernels import EmulatedKernels

    _lib.set_kernels_for_testing(EmulatedKernels())
from transformers import BertConfig, BertForSequenceClassification  # noqa: E402

from laplace_amd.laplace import HipLaplace  # noqa: E402


class BertHead(nn.Module):
    """The wrapper the reference's HuggingFace example uses: dict batch in, logits out."""

    def __init__(self, cfg):
        super().__init__()
        self.hf = BertForSequenceClassification(cfg)

    def forward(self, data):
        return self.hf(input_ids=data["input_ids"], attention_mask=data["attention_mask"]).logits


def sync():
    if dev != "cpu":
        torch.cuda.synchronize()


torch.manual_seed(711)
cfg = BertConfig(num_labels=2)
if args.tiny:
    cfg = BertConfig(num_labels=2, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
                     vocab_size=100, max_position_embeddings=32)
T, bs = (16, 8) if args.tiny else (128, 32)
model = BertHead(cfg).to(dev).eval()


def make(n, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(cfg.vocab_size, (n, T), generator=g)
    mask = torch.ones(n, T, dtype=torch.long)
    lens = torch.randint(T // 2, T + 1, (n,), generator=g)
    mask[torch.arange(T)[None, :] >= lens[:, None]] = 0
    y = torch.randint(2, (n,), generator=g)
    batches = [{"input_ids": ids[i:i + bs].to(dev), "attention_mask": mask[i:i + bs].to(dev), "labels": y[i:i + bs].to(dev)}
               for i in range(0, n, bs)]

    class Loader(list):
        dataset = range(n)

    return Loader(batches)


train, val = make(args.n, 1), make(args.n_val, 2)
out = {"config": {"model": "tiny-bert" if args.tiny else "bert-base (BertConfig(), random init)", "seq_len": T,
                  "batch": bs, "n_train": args.n, "n_val": args.n_val, "structure": "last_layer kron"}}
with torch.no_grad():  # encoder-only forward rate for reference
    for b in train[:2]:
        model(b)
    sync()
    t0 = time.perf_counter()
    for b in train:
        model(b)
    sync()
    out["forward_only_samples_per_s"] = args.n / (time.perf_counter() - t0)

ts = []
for rep in range(3):
    la = HipLaplace(model, "classification", "last_layer", "kron", last_layer_name="hf.classifier", prior_precision=1.0)
    sync()
    t0 = time.perf_counter()
    la.fit(train)
    sync()
    ts.append(time.perf_counter() - t0)
out["fit_samples_per_s"] = args.n / min(ts[1:])
out["fit_ms"] = min(ts[1:]) * 1e3
out["factor_shapes"] = [[list(f.shape) for f in blk] for blk in la.H_facs.kfacs]

sync()
t0 = time.perf_counter()
la.optimize_prior_precision(pred_type="glm", method="marglik", n_steps=100, lr=0.1, prior_structure="layerwise")
sync()
out["marglik_100_steps_ms"] = (time.perf_counter() - t0) * 1e3
out["marglik_prior_precision"] = float(la.prior_precision.reshape(-1)[0])
out["log_marglik"] = float(la.log_marginal_likelihood())

sync()
t0 = time.perf_counter()
la.gridsearch_prior_precision(val, grid_size=100)
sync()
out["gridsearch_100_points_ms"] = (time.perf_counter() - t0) * 1e3
out["gridsearch_prior_precision"] = float(la.prior_precision.reshape(-1)[0])

for b in val[:2]:
    la(b)
sync()
t0 = time.perf_counter()
for b in val:
    la(b)
sync()
out["predictive_samples_per_s"] = args.n_val / (time.perf_counter() - t0)
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/c5_bert.json", "w"), indent=1)
