"""Config c5 shape: HuggingFace-style dict inputs, last-layer KFAC, marginal-likelihood prior sweep
(docs/huggingface_example.md of the reference; baselaplace.py:943-974 leaves MutableMapping batches
untouched and reads labels from ``dict_key_y``).  Host logic on the kernel emulation + GPU variant."""
import pytest
import torch
from torch import nn
from torch.utils.data import DataLoader

from oracle import curvature_oracle as co


class TinyEncoderClassifier(nn.Module):
    """Embedding -> mean-pool -> tanh MLP -> Linear head; consumes {'input_ids', 'attention_mask', 'labels'}."""

    def __init__(self, vocab=50, d=24, n_labels=2):
        super().__init__()
        self.emb = nn.Embedding(vocab, d)
        self.body = nn.Sequential(nn.Linear(d, d), nn.Tanh())
        self.classifier = nn.Linear(d, n_labels)

    def forward(self, data):
        ids, mask = data["input_ids"], data["attention_mask"].unsqueeze(-1).to(self.emb.weight.dtype)
        h = (self.emb(ids) * mask).sum(1) / mask.sum(1)
        return self.classifier(self.body(h))


def _data(dev, n=24, T=7, seed=0):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(50, (n, T), generator=g)
    mask = (torch.rand(n, T, generator=g) > 0.2).long()
    mask[:, 0] = 1
    labels = torch.randint(2, (n,), generator=g)
    rows = [{"input_ids": ids[i].to(dev), "attention_mask": mask[i].to(dev), "labels": labels[i].to(dev)} for i in range(n)]

    def collate(batch):
        return {k: torch.stack([b[k] for b in batch]) for k in batch[0]}

    return rows, collate


def _run(dev):
    from laplace_amd.laplace import HipLaplace

    torch.manual_seed(711)
    model = TinyEncoderClassifier().to(dev)
    rows, collate = _data(dev)
    loader = DataLoader(rows, batch_size=8, collate_fn=collate)
    la = HipLaplace(model, "classification", "last_layer", "kron", last_layer_name="classifier", prior_precision=1.0)
    la.fit(loader)
    # oracle: KFAC of the head from the features, accumulated over the same minibatches (fp64, CPU)
    m64 = TinyEncoderClassifier().double()
    m64.load_state_dict({k: v.double().cpu() for k, v in model.state_dict().items()})
    for p in m64.parameters():
        p.requires_grad_(False)
    for p in m64.classifier.parameters():
        p.requires_grad_(True)
    acc = None
    for batch in loader:
        b64 = {k: v.cpu() for k, v in batch.items()}
        _, kf = co.kfac_ggn(m64, b64, b64["labels"], len(rows), "classification", params=list(m64.classifier.parameters()))
        acc = kf if acc is None else co.kron_add(acc, kf)
    for F_, G_ in zip(la.H_facs.kfacs, acc):
        for a, w in zip(F_, G_):
            err = (a.double().cpu() - w).abs().max().item() / (w.abs().max().item() + 1e-30)
            assert err < 1e-4, err
    before = la.log_marginal_likelihood().item()
    la.optimize_prior_precision(method="marglik", n_steps=40, lr=0.1, prior_structure="scalar")
    assert la.log_marginal_likelihood().item() >= before - 1e-3
    batch = next(iter(loader))
    f_mu, f_var = la._glm_predictive_distribution(batch)
    Qs, ls = co.kron_decompose(acc)
    Js = co.last_layer_jacobians(m64.body(_pooled(m64, {k: v.cpu() for k, v in batch.items()})), 2, True)
    want = co.functional_variance_kron(Js, Qs, ls, float(la.prior_precision))
    err = (f_var.double().cpu() - want).abs().max().item() / want.abs().max().item()
    assert err < 1e-4, err
    probs = la(batch)
    assert probs.shape == (8, 2)
    # validation gridsearch on dict batches: one cached feature pass per batch, same choice as the literal loop
    interval = torch.logspace(-2, 2, 7)
    nll = []
    for pp in interval:
        la.prior_precision = pp
        tot = 0.0
        for b in loader:
            p = la(b)
            tot += float(-torch.log(p[torch.arange(len(p)), b["labels"]]).sum())
        nll.append(tot)
    chosen = la.gridsearch_prior_precision(loader, log_prior_prec_min=-2, log_prior_prec_max=2, grid_size=7)
    assert float(chosen) == pytest.approx(float(interval[int(torch.tensor(nll).argmin())]))


def _pooled(m, data):
    ids, mask = data["input_ids"], data["attention_mask"].unsqueeze(-1).double()
    return (m.emb(ids) * mask).sum(1) / mask.sum(1)


def test_dict_inputs_last_layer_kron_on_emulation():
    from laplace_amd import _lib
    from tests.emulated_kernels import EmulatedKernels

    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    try:
        _run("cpu")
    finally:
        _lib.set_kernels_for_testing(prev)


@pytest.mark.gpu
def test_dict_inputs_last_layer_kron_gpu():
    _run("cuda")
