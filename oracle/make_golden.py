"""Generate tests/golden/*.npz by running the UNMODIFIED reference (this container only).

TEST INFRASTRUCTURE.  Usage:  ``python -m oracle.make_golden``  (needs /root/reference).

What is produced, per fixture x likelihood (float64):
  * inputs: model weights ``w.*``, ``X``, ``y``
  * the reference's in-tree torch.func backend (laplace/curvature/curvature.py):
    ``Js, f`` (GGNInterface.jacobians :88-129), ``H_ggn, h_ggn, loss`` (full/diag :375-433),
    ``H_ef, h_ef, loss_ef`` (EFInterface :467-505)
  * the reference's Kron algebra (laplace/utils/matrix.py) evaluated on the oracle's KFAC
    factors: eigenvalues after ``decompose`` (:123-150), ``logdet`` (:381-404), ``bmm`` with
    exponents 1, -1, -1/2 (:406-456), ``inv_square_form`` (:458-461), ``diag`` (:490-522)
  * end to end through the reference's Laplace classes (laplace/baselaplace.py,
    laplace/lllaplace.py) with two minibatches: accumulated ``H``/``loss``, GLM predictive
    ``f_mu, f_var`` (:1306-1342), ``log_marginal_likelihood`` (:1074-1109) for diag, full, kron,
    all-weights and last-layer.  The kron flavours receive the factors from a thin adapter
    that calls the oracle's KFAC restatement (curvlinops itself is not installable here), so
    they pin everything *downstream* of the raw factors.
"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch.utils.data import DataLoader, TensorDataset

from oracle import curvature_oracle as co
from oracle.fixtures import FIXTURES, make_fixture
from oracle.ref_import import import_reference

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

PRIOR_PREC = 0.7
SIGMA_NOISE = 0.8
DELTA = 0.3
H_FACTOR = 1.7


def _np(t):
    return t.detach().cpu().numpy().astype(np.float64) if t.is_floating_point() else t.detach().cpu().numpy()


def _flatten_kfacs(prefix, kfacs, out):
    out[f"{prefix}.n_blocks"] = np.array(len(kfacs))
    for i, F_ in enumerate(kfacs):
        out[f"{prefix}.{i}.len"] = np.array(len(F_))
        for j, Hi in enumerate(F_):
            out[f"{prefix}.{i}.{j}"] = _np(Hi)


def generate(name: str, likelihood: str) -> dict:
    import_reference()
    from laplace import Laplace
    from laplace.curvature import EFInterface, GGNInterface
    from laplace.utils.matrix import Kron

    torch.set_default_dtype(torch.float64)
    model, X, y_cls, y_reg = make_fixture(name)
    y = y_cls if likelihood == "classification" else y_reg
    out: dict = {"X": _np(X), "y": _np(y)}
    for k, v in model.state_dict().items():
        out[f"w.{k}"] = _np(v)

    # ---- in-tree backend -------------------------------------------------------------
    ggn = GGNInterface(model, likelihood)
    Js, f = ggn.jacobians(X)
    loss, H = ggn.full(X, y)
    _, h = ggn.diag(X, y)
    out.update(Js=_np(Js), f=_np(f), H_ggn=_np(H), h_ggn=_np(h), loss=_np(loss))
    ef = EFInterface(model, likelihood)
    loss_ef, H_ef = ef.full(X, y)
    _, h_ef = ef.diag(X, y)
    out.update(H_ef=_np(H_ef), h_ef=_np(h_ef), loss_ef=_np(loss_ef))

    # ---- Kron algebra of the reference on the oracle's factors --------------------------
    N = X.shape[0]
    loss_k, kfacs = co.kfac_ggn(model, X, y, N, likelihood)
    _flatten_kfacs("kfac", kfacs, out)
    out["loss_kfac"] = _np(loss_k)
    kron = Kron([[Hi.clone() for Hi in F_] for F_ in kfacs])
    out["kron_diag"] = _np(kron.diag())
    dec = kron.decompose()
    for i, ls in enumerate(dec.eigenvalues):
        for j, l in enumerate(ls):
            out[f"eigvals.{i}.{j}"] = _np(l)
    post = dec * H_FACTOR + torch.tensor(DELTA)
    out["kd_logdet"] = _np(post.logdet())
    torch.manual_seed(5)
    W = torch.randn(4, 3, Js.shape[-1])
    out["W"] = _np(W)
    for tag, e in (("p1", 1.0), ("m1", -1.0), ("mh", -0.5)):
        out[f"kd_bmm_{tag}"] = _np(post.bmm(W, exponent=e))
        out[f"kd_diag_{tag}"] = _np(post.diag(exponent=e))
    out["kd_isf"] = _np(post.inv_square_form(Js))
    per_layer = torch.linspace(0.2, 1.5, len(dec))
    post_l = dec * H_FACTOR + per_layer
    out["per_layer_delta"] = _np(per_layer)
    out["kd_logdet_layer"] = _np(post_l.logdet())
    out["kd_isf_layer"] = _np(post_l.inv_square_form(Js))

    # ---- end to end through the reference's Laplace classes ----------------------------
    class OracleKronGGN(GGNInterface):
        """Adapter: the reference's KronLaplace driven by the oracle's KFAC factors."""

        def kron(self, x, y, N, **kw):
            loss, kf = co.kfac_ggn(self.model, x, y, N, self.likelihood, params=self.params)
            return loss, Kron(kf)

    loader = DataLoader(TensorDataset(X, y), batch_size=5)
    sig = SIGMA_NOISE if likelihood == "regression" else 1.0
    for sow in ("all", "last_layer"):
        for hs in ("diag", "full", "kron"):
            tag = f"la.{sow}.{hs}"
            backend = OracleKronGGN if hs == "kron" else GGNInterface
            la = Laplace(
                model, likelihood, subset_of_weights=sow, hessian_structure=hs,
                prior_precision=PRIOR_PREC, sigma_noise=sig, backend=backend,
            )
            la.fit(loader)
            out[f"{tag}.loss"] = _np(torch.as_tensor(la.loss))
            if hs == "kron":
                _flatten_kfacs(f"{tag}.H", la.H_facs.kfacs, out)
            else:
                out[f"{tag}.H"] = _np(la.H)
            f_mu, f_var = la._glm_predictive_distribution(X)
            out[f"{tag}.f_mu"] = _np(f_mu)
            out[f"{tag}.f_var"] = _np(f_var)
            out[f"{tag}.marglik"] = _np(la.log_marginal_likelihood())
            out[f"{tag}.logdet_post"] = _np(la.log_det_posterior_precision)
    return out


def main():
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    for name in FIXTURES:
        for likelihood in ("classification", "regression"):
            arrays = generate(name, likelihood)
            path = os.path.join(GOLDEN_DIR, f"{name}_{likelihood}.npz")
            np.savez_compressed(path, **arrays)
            print(f"wrote {path}: {len(arrays)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
