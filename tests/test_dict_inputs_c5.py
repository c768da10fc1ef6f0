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


# ---- c5 at its BASELINE shape: BERT-base features (BertConfig(): 12 x 768, random init), Linear(768, 2) head -------------
def _run_bert(dev, n, T, bs):
    """Last-layer KFAC + GLM predictive of `BertForSequenceClassification(BertConfig(num_labels=2))` on dict batches
    (docs/huggingface_example.md of the reference; lllaplace.py:212-237 feeds the head's input features to the backend):
    factors against the fp64 oracle on fp64 FEATURES of the same weights (the whole encoder run in fp64 on the CPU), the
    predictive variance against the oracle's last-layer Jacobians pushed through matrix.py:406-461."""
    import copy

    transformers = pytest.importorskip("transformers")
    from laplace_amd.laplace import HipLaplace

    class BertHead(nn.Module):  # the wrapper of the reference's HuggingFace example: dict batch in, logits out
        def __init__(self, cfg):
            super().__init__()
            self.hf = transformers.BertForSequenceClassification(cfg)

        def forward(self, data):
            return self.hf(input_ids=data["input_ids"], attention_mask=data["attention_mask"]).logits

    torch.manual_seed(711)
    cfg = transformers.BertConfig(num_labels=2)
    assert (cfg.hidden_size, cfg.num_hidden_layers) == (768, 12)
    model = BertHead(cfg).eval()
    m64 = copy.deepcopy(model).double().eval()
    model = model.to(dev)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(cfg.vocab_size, (n, T), generator=g)
    mask = torch.ones(n, T, dtype=torch.long)
    lens = torch.randint(T // 2, T + 1, (n,), generator=g)
    mask[torch.arange(T)[None, :] >= lens[:, None]] = 0
    yy = torch.randint(2, (n,), generator=g)

    class Loader(list):
        dataset = range(n)

    batches = [{"input_ids": ids[i:i + bs], "attention_mask": mask[i:i + bs], "labels": yy[i:i + bs]} for i in range(0, n, bs)]
    train = Loader([{k: v.to(dev) for k, v in b.items()} for b in batches])
    la = HipLaplace(model, "classification", "last_layer", "kron", last_layer_name="hf.classifier", prior_precision=1.0)
    la.fit(train)
    head64 = nn.Linear(768, 2).double()
    head64.load_state_dict(m64.hf.classifier.state_dict())
    acc, feats = None, []
    with torch.no_grad():
        for b in batches:
            feats.append(m64.hf.bert(input_ids=b["input_ids"], attention_mask=b["attention_mask"]).pooler_output)  # (dropout: eval)
    for b, phi in zip(batches, feats):
        _, kf = co.kfac_ggn(head64, phi, b["labels"], n, "classification")
        acc = kf if acc is None else co.kron_add(acc, kf)
    worst = 0.0
    for F_, G_ in zip(la.H_facs.kfacs, acc):
        for a, w in zip(F_, G_):
            assert a.shape == w.shape
            worst = max(worst, (a.double().cpu() - w).abs().max().item() / (w.abs().max().item() + 1e-30))
    assert la.H_facs.kfacs[0][1].shape[0] == 768
    assert worst < 1e-4, f"c5 factors vs the fp64 oracle on fp64 BERT features: {worst:.2e}"
    f_mu, f_var = la._glm_predictive_distribution(train[0])
    Qs, ls = co.kron_decompose(acc)
    Js = co.last_layer_jacobians(feats[0], 2, True)
    want = co.functional_variance_kron(Js, Qs, ls, float(la.prior_precision))
    err = (f_var.double().cpu() - want).abs().max().item() / want.abs().max().item()
    err_mu = (f_mu.double().cpu() - head64(feats[0]).detach()).abs().max().item() / head64(feats[0]).abs().max().item()
    assert err < 1e-4 and err_mu < 1e-4, (err, err_mu)
    before = la.log_marginal_likelihood().item()
    la.optimize_prior_precision(pred_type="glm", method="marglik", n_steps=20, lr=0.1, prior_structure="layerwise")
    assert la.log_marginal_likelihood().item() >= before - 1e-3
    try:
        from tests.parity_log import record_error

        record_error(worst), record_error(err)
    except Exception:
        pass


def test_c5_bert_base_features_last_layer_kfac_on_emulation():
    """host logic of the BASELINE c5 shape (BertConfig() encoder, Linear(768, 2) head) at a short sequence, on the emulation"""
    from laplace_amd import _lib
    from tests.emulated_kernels import EmulatedKernels

    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    try:
        _run_bert("cpu", n=8, T=16, bs=4)
    finally:
        _lib.set_kernels_for_testing(prev)


@pytest.mark.gpu
def test_c5_bert_base_features_last_layer_kfac_at_the_baseline_shape_gpu():
    """BASELINE.json c5 as quoted: sequence 128, batch 32 (two minibatches), on the device"""
    _run_bert("cuda", n=64, T=128, bs=32)
