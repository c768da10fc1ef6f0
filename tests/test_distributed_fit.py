"""Data-parallel fit (one process per GPU in production; here world_size-2 `gloo` on CPU with the
kernel emulation): minibatches sharded by rank + one all-reduce == single-process fit."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch.utils.data import DataLoader, TensorDataset

from tests.conftest import golden_model, load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, hs, out_path, bs=2, fixture="resnetish"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from laplace_amd import _lib
        from laplace_amd.laplace import HipLaplace, ShardedLoader
        from tests.emulated_kernels import EmulatedKernels

        _lib.set_kernels_for_testing(EmulatedKernels())
        g = load_golden(fixture, "classification")
        model, X, y = golden_model(fixture, g, dtype=torch.float32)
        loader = ShardedLoader(DataLoader(TensorDataset(X, y), batch_size=bs), rank, world)
        la = HipLaplace(model, "classification", "all", hs, prior_precision=0.7)
        la.fit(loader)
        if hs == "kron":  # every rank keeps the FULL decomposition although it solved only its share
            torch.save({"l": la.H.eigenvalues, "Q": la.H.eigenvectors, "n_outputs": la.n_outputs,
                        "output_size": getattr(la.model, "output_size", None)}, f"{out_path}.eig{rank}")
        if rank == 0:
            payload = {"loss": la.loss, "n_data": la.n_data}
            if hs == "kron":
                payload["H"] = [[Hi.clone() for Hi in F] for F in la.H_facs.kfacs]
                payload["marglik"] = la.log_marginal_likelihood()
            else:
                payload["H"] = la.H.clone()
            torch.save(payload, out_path)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("hs,world,bs,fixture", [
    ("kron", 2, 2, "resnetish"), ("diag", 2, 2, "resnetish"), ("full", 2, 2, "resnetish"),
    ("kron", 3, 3, "resnetish"),   # ragged last batch (3 + 3 + 3 + 1), uneven shards (2, 1, 1)
    ("kron", 3, 5, "resnetish"),   # two batches for three ranks: rank 2 has an EMPTY shard
    # conv -> eval-BatchNorm taps: the G factors carry a DEFERRED BatchNorm scale that an empty rank never learns; it
    # must be applied before the exchange (the sharded eigendecomposition then runs on that rank's copy of the sums)
    ("kron", 3, 5, "bnres"), ("kron", 4, 4, "bnres"),
    ("full", 3, 3, "resnetish")])
def test_sharded_fit_equals_single_process(tmp_path, hs, world, bs, fixture):
    from laplace_amd import _lib
    from laplace_amd.laplace import HipLaplace
    from tests.emulated_kernels import EmulatedKernels

    out = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(world, _free_port(), hs, out, bs, fixture), nprocs=world, join=True)
    got = torch.load(out, weights_only=False)

    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    try:
        g = load_golden(fixture, "classification")
        model, X, y = golden_model(fixture, g, dtype=torch.float32)
        la = HipLaplace(model, "classification", "all", hs, prior_precision=0.7)
        la.fit(DataLoader(TensorDataset(X, y), batch_size=bs), distributed=False)
        ref_marglik = la.log_marginal_likelihood() if hs == "kron" else None
    finally:
        _lib.set_kernels_for_testing(prev)
    assert got["n_data"] == la.n_data == 10
    torch.testing.assert_close(got["loss"], la.loss, rtol=1e-5, atol=1e-6)
    if hs == "kron":
        for F_, G_ in zip(got["H"], la.H_facs.kfacs):
            for a, b in zip(F_, G_):
                torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(got["marglik"], ref_marglik, rtol=1e-5, atol=1e-5)
        # sharded eigendecomposition: identical on both ranks (it is exchanged, not recomputed), and a valid
        # decomposition of every factor
        e0 = torch.load(out + ".eig0", weights_only=False)
        e1 = torch.load(out + f".eig{world - 1}", weights_only=False)
        assert e1["n_outputs"] == e0["n_outputs"] == la.n_outputs and e1["output_size"] == la.n_outputs
        for F_, ls0, ls1, Qs0, Qs1 in zip(la.H_facs.kfacs, e0["l"], e1["l"], e0["Q"], e1["Q"]):
            for Hi, l0, l1, Q0, Q1 in zip(F_, ls0, ls1, Qs0, Qs1):
                assert torch.equal(l0, l1) and torch.equal(Q0, Q1)
                if Hi.ndim > 1:
                    torch.testing.assert_close(Q0 @ torch.diag(l0) @ Q0.T, Hi, rtol=1e-4, atol=1e-5)
    else:
        torch.testing.assert_close(got["H"], la.H, rtol=1e-5, atol=1e-6)


def test_factor_sharding_is_balanced_and_deterministic():
    from laplace_amd.kron import HipKron

    sizes = [4608, 4608, 4608, 2304, 2304, 2304, 2304, 1152, 1152, 1152, 1152, 576, 576, 576, 576, 576, 512, 512,
             512, 512, 512, 256, 256, 256, 256, 256, 128, 128, 128, 128, 128, 64, 64, 64, 64, 64, 64, 27, 10]
    for world in (1, 2, 4, 8):
        owner = HipKron.shard_factors(sizes, world)
        assert owner == HipKron.shard_factors(list(sizes), world) and set(owner) <= set(range(world))
        load = [sum(float(n) ** 3 for n, o in zip(sizes, owner) if o == r) for r in range(world)]
        # the critical path is at most one largest factor above the ideal split
        assert max(load) <= sum(load) / world + float(max(sizes)) ** 3
    owner = HipKron.shard_factors(sizes, 8)
    assert len({owner[0], owner[1], owner[2]}) == 3  # the three 4608-factors land on different GPUs


def _ddp_style_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import warnings

        from laplace_amd import _lib
        from laplace_amd.laplace import HipLaplace
        from tests.emulated_kernels import EmulatedKernels

        _lib.set_kernels_for_testing(EmulatedKernels())
        g = load_golden("mlp", "classification")
        model, X, y = golden_model("mlp", g, dtype=torch.float32)
        la = HipLaplace(model, "classification", "all", "kron", prior_precision=0.7)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            la.fit(DataLoader(TensorDataset(X, y), batch_size=5))  # every rank sees the FULL loader, as in a DDP script
        if rank == 0:
            torch.save({"loss": la.loss, "warned": any("not sharded" in str(x.message) for x in w),
                        "H": [[Hi.clone() for Hi in F] for F in la.H_facs.kfacs]}, out_path)
    finally:
        dist.destroy_process_group()


def test_unsharded_loader_under_an_initialised_process_group_stays_local(tmp_path):
    """with torch.distributed initialised but the loader NOT sharded, an automatic all-reduce would multiply
    the curvature by the world size.  The collective is opt-in: it warns and fits locally."""
    from laplace_amd import _lib
    from laplace_amd.laplace import HipLaplace
    from tests.emulated_kernels import EmulatedKernels

    out = str(tmp_path / "ddp.pt")
    mp.spawn(_ddp_style_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out, weights_only=False)
    prev = _lib.set_kernels_for_testing(EmulatedKernels())
    try:
        g = load_golden("mlp", "classification")
        model, X, y = golden_model("mlp", g, dtype=torch.float32)
        la = HipLaplace(model, "classification", "all", "kron", prior_precision=0.7)
        la.fit(DataLoader(TensorDataset(X, y), batch_size=5))
    finally:
        _lib.set_kernels_for_testing(prev)
    assert got["warned"]
    torch.testing.assert_close(got["loss"], la.loss)
    for F_, G_ in zip(got["H"], la.H_facs.kfacs):
        for a, b in zip(F_, G_):
            torch.testing.assert_close(a, b)


def test_sharded_loader_refuses_a_shuffle_that_the_ranks_cannot_reproduce():
    from laplace_amd.laplace import ShardedLoader

    ds = TensorDataset(torch.arange(12.0).reshape(12, 1), torch.zeros(12))
    with pytest.raises(ValueError, match="generator"):
        iter(ShardedLoader(DataLoader(ds, batch_size=2, shuffle=True), 0, 2))
    # a shared generator is fine: both ranks cut the same permutation, and together they see every sample once
    seen = []
    for rank in range(2):
        ld = DataLoader(ds, batch_size=2, shuffle=True, generator=torch.Generator().manual_seed(5))
        seen += [float(v) for X, _ in ShardedLoader(ld, rank, 2) for v in X.reshape(-1)]
    assert sorted(seen) == [float(i) for i in range(12)]
    assert len(list(ShardedLoader(DataLoader(ds, batch_size=2), 0, 1))) == 6


def test_expected_exchange_bytes_of_resnet18():
    """what bench.py asserts on a multi-GPU run: the fit's one all-reduce moves the packed upper triangles of the 42 dense
    factors + the loss word — 188 MB for ResNet-18 (376 MB as squares)"""
    from laplace_amd.laplace import expected_exchange_bytes
    from laplace_amd.nets import ResNet18

    m = ResNet18(10)
    want = expected_exchange_bytes(m)
    sizes = []
    for mod in m.modules():
        if isinstance(mod, torch.nn.Conv2d):
            sizes += [mod.out_channels, mod.in_channels * 9 if mod.kernel_size[0] == 3 else mod.in_channels]
        elif isinstance(mod, torch.nn.Linear):
            sizes += [mod.out_features, mod.in_features]
    assert len(sizes) == 42 and want == 4 * (sum(n * (n + 1) // 2 for n in sizes) + 1)
    assert 187e6 < want < 189e6


def _range_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from laplace_amd import HipGGN, _lib
        from laplace_amd.laplace import allreduce_curvature
        from tests.emulated_kernels import EmulatedKernels

        _lib.set_kernels_for_testing(EmulatedKernels())
        from tests.test_sweep_nhwc import _model

        model = _model(torch.relu)  # (a model the split-fp16 sweep serves)
        torch.manual_seed(11 + rank)
        X, y = torch.randn(6, 3, 8, 8), torch.randint(5, (6,))
        if rank == 1:  # only THIS rank's shard holds a minibatch that spans eight decades
            X = X.clone()
            X[0] *= 1e-4
            X[1] *= 1e4
        acc = HipGGN(model, "classification").kron_accumulator(20)
        acc.add_batch(X, y)
        try:
            allreduce_curvature(acc.tensors(), mirror=False)
            verdict = "no error"
        except RuntimeError as e:
            verdict = str(e)
        dist.barrier()
        _, kron = acc.finalize()
        torch.save((verdict, X, y, [[M.clone() for M in F] for F in kron.kfacs]), f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()


def test_a_shard_with_a_wide_range_minibatch_needs_no_verdict(tmp_path):
    """Rounds 3 - 4 fenced minibatches whose samples differ by more than 2^16 in magnitude (a refusal that had to travel
    with the all-reduce so that every rank raised).  The forward's split tensors now carry one scale per image: such a
    minibatch is computed like any other — no flag word in the exchange, no error on any rank, and the reduced factors are
    the fp64 Grams of the union of the shards (curvlinops.py:77-108 accepts any finite input)."""
    from oracle import curvature_oracle as co
    from tests.test_sweep_nhwc import _model

    out = str(tmp_path / "verdict")
    mp.spawn(_range_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0", weights_only=False), torch.load(out + ".1", weights_only=False)
    assert r0[0] == "no error" and r1[0] == "no error"
    m64 = _model(torch.relu).double()
    ref = None
    for _, X, y, _ in (r0, r1):
        _, kf = co.kfac_ggn(m64, X.double(), y, 20, "classification")
        ref = kf if ref is None else [[a + b for a, b in zip(Fa, Fb)] for Fa, Fb in zip(ref, kf)]
    for got in (r0[3], r1[3]):
        for F_, G_ in zip(got, ref):
            for a_, w_ in zip(F_, G_):
                assert (a_.double() - w_).abs().max() <= 1e-4 * w_.abs().max()
