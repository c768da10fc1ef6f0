"""The data-parallel fit on the REAL kernels: two processes (one per GPU in production; here both on cuda:0, exchanging
through `gloo`, which RCCL's one-rank-per-device rule does not forbid) each fit their shard of the minibatches on a small
ResNet whose layers take the split-fp16 sweep, the pixel-pair A factors and the strided fused launch; ONE all-reduce of
the packed upper triangles (lk_pack_upper_f32 / lk_unpack_upper_f32 on accumulators of which only the upper triangle is
valid), the layout pass, the sharded eigendecomposition — against the single-process fit of all minibatches.  The host
logic of the same path at world sizes 2 - 4 with ragged and empty shards: tests/test_distributed_fit.py (CPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _model():
    from laplace_amd.nets import BasicBlock

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 64, 3, 1, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(64)
            self.layers = nn.Sequential(BasicBlock(64, 64, 1), BasicBlock(64, 128, 2))
            self.pool = nn.AdaptiveAvgPool2d(1)
            self.fc = nn.Linear(128, 7)

        def forward(self, x):
            x = torch.relu(self.bn1(self.conv1(x)))
            return self.fc(torch.flatten(self.pool(self.layers(x)), 1))

    torch.manual_seed(21)
    m = Net().eval()
    for mod in m.modules():
        if isinstance(mod, nn.BatchNorm2d):
            mod.running_var.uniform_(0.5, 2.0), mod.weight.data.uniform_(0.5, 1.5)
            mod.weight.requires_grad_(False), mod.bias.requires_grad_(False)
    return m


def _data():
    g = torch.Generator().manual_seed(4)
    return [(torch.randn(16, 3, 8, 8, generator=g), torch.randint(7, (16,), generator=g)) for _ in range(5)]


class _Loader(list):
    dataset = list(range(80))


def _fit(loader, distributed):
    from laplace_amd.laplace import HipLaplace

    la = HipLaplace(_model().cuda(), "classification", "all", "kron", prior_precision=0.7)
    la.fit(loader, distributed=distributed)
    return la


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from laplace_amd.laplace import ShardedLoader

        torch.cuda.set_device(0)
        batches = _Loader((X.cuda(), y.cuda()) for X, y in _data())
        la = _fit(ShardedLoader(batches, rank, world), True)
        torch.save({"loss": la.loss.cpu(), "H": [[Hi.cpu() for Hi in F] for F in la.H_facs.kfacs],
                    "l": [[l.cpu() for l in ls] for ls in la.H.eigenvalues], "marglik": la.log_marginal_likelihood().cpu()},
                   f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_on_the_real_kernels_equal_the_single_process_fit(tmp_path):
    from tests.parity_log import record_error

    out = str(tmp_path / "rank")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = [torch.load(f"{out}.{r}", weights_only=False) for r in range(2)]
    ref = _fit(_Loader((X.cuda(), y.cuda()) for X, y in _data()), False)
    for r in range(2):
        assert abs(float(got[r]["loss"]) - float(ref.loss)) <= 1e-5 * abs(float(ref.loss))
        for F_, G_ in zip(got[r]["H"], ref.H_facs.kfacs):
            for a, b in zip(F_, G_):
                err = record_error(float((a.double() - b.double().cpu()).abs().max() / (b.double().abs().max() + 1e-300)))
                assert err < 1e-5, err
        assert abs(float(got[r]["marglik"]) - float(ref.log_marginal_likelihood())) <= 1e-4 * abs(float(ref.log_marginal_likelihood()))
    # the sharded eigendecomposition is exchanged, not recomputed: identical on both ranks
    for ls0, ls1 in zip(got[0]["l"], got[1]["l"]):
        for l0, l1 in zip(ls0, ls1):
            assert torch.equal(l0, l1)


def test_the_c_abi_collective_on_a_one_rank_communicator():
    """lk_comm_unique_id / lk_comm_init_rank / lk_allreduce_sum_f32 / lk_comm_destroy (include/laplace_hip.h: RCCL bound by
    dlopen, for hosts that are not PyTorch): a world of ONE rank sums to itself — the binding, the argument order and the
    stream ordering; RCCL refuses two ranks on one device, and the box has one."""
    import ctypes

    from laplace_amd._lib import load_library

    lib = load_library()
    uid = ctypes.create_string_buffer(128)
    assert lib.lk_comm_unique_id(uid) == 0, lib.lk_last_error()
    comm = ctypes.c_void_p()
    assert lib.lk_comm_init_rank(ctypes.byref(comm), 1, uid, 0) == 0, lib.lk_last_error()
    try:
        x = torch.randn(1 << 20, device="cuda")
        want = x.clone()
        st = torch.cuda.current_stream()
        assert lib.lk_allreduce_sum_f32(comm, x.data_ptr(), x.numel(), st.cuda_stream) == 0, lib.lk_last_error()
        torch.cuda.synchronize()
        assert torch.equal(x, want)
        assert lib.lk_allreduce_sum_f32(comm, None, 0, st.cuda_stream) == 0
        assert lib.lk_allreduce_sum_f32(None, x.data_ptr(), 4, st.cuda_stream) < 0 and b"bad arguments" in lib.lk_last_error()
    finally:
        assert lib.lk_comm_destroy(comm) == 0
