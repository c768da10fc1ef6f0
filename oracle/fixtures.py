"""Deterministic tiny models + data used by the golden generator and the parity tests.

TEST INFRASTRUCTURE.  Architectures follow the reference's own test fixtures
(tests/test_curv_backends_curvlinops.py:23-65: ``Linear(3,20)-Tanh-Linear(20,2)`` on
``X[10,3]`` and the conv "complex_model" on ``X[10,3,5,5]``, seed 711) plus one ResNet-shaped
conv stack (3x3 / padding 1 / stride 2 / bias-free conv) that the reference's fixtures lack but
config c4 needs, a torchvision-style BatchNorm residual block, and a Linear applied along a sequence.  Weights and data are stored inside the golden files, so RNG drift between
torch versions cannot silently change the fixtures.
"""
from __future__ import annotations

import torch
from torch import nn

FIXTURES = ("mlp", "conv", "resnetish", "bnres", "seqlin")


class _BNResBlock(nn.Module):
    """torchvision-style BasicBlock: conv-BN-ReLU(in place)-conv-BN, `out += identity`, ReLU; BatchNorm in eval mode with
    non-trivial running statistics and frozen affine parameters (the reference's KFAC covers Linear / Conv2d only)."""

    def __init__(self, c: int):
        super().__init__()
        self.conv1 = nn.Conv2d(c, c, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(c)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(c, c, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(c)

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        out += identity
        return self.relu(out)


class _MeanOverPositions(nn.Module):
    """[B, T, D] -> [B, D]"""

    def forward(self, x):
        return x.mean(1)


def build_model(name: str) -> nn.Module:
    if name == "mlp":
        return nn.Sequential(nn.Linear(3, 20), nn.Tanh(), nn.Linear(20, 2))
    if name == "conv":
        return nn.Sequential(
            nn.Conv2d(3, 4, 2, 2), nn.Flatten(), nn.Tanh(), nn.Linear(16, 20), nn.Tanh(), nn.Linear(20, 2)
        )
    if name == "resnetish":
        return nn.Sequential(
            nn.Conv2d(2, 4, 3, padding=1),
            nn.ReLU(),
            nn.Conv2d(4, 4, 3, stride=2, padding=1, bias=False),
            nn.ReLU(),
            nn.Flatten(),
            nn.Linear(36, 3),
        )
    if name == "seqlin":
        # an nn.Linear applied along a sequence (weight sharing over T positions, as in a transformer block), pooled,
        # then an ordinary head
        return nn.Sequential(nn.Linear(5, 6), nn.Tanh(), _MeanOverPositions(), nn.Linear(6, 2))
    if name == "bnres":
        model = nn.Sequential(nn.Conv2d(2, 4, 3, padding=1), _BNResBlock(4), nn.AdaptiveAvgPool2d(1), nn.Flatten(),
                              nn.Linear(4, 3))
        for m in model.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.normal_(0.0, 0.5)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0.0, 0.3)
                m.weight.requires_grad_(False)
                m.bias.requires_grad_(False)
        return model.eval()
    raise KeyError(name)


def input_shape(name: str):
    return {"mlp": (3,), "conv": (3, 5, 5), "resnetish": (2, 5, 5), "bnres": (2, 4, 4), "seqlin": (4, 5)}[name]


def n_outputs(name: str) -> int:
    return 3 if name in ("resnetish", "bnres") else 2


def make_fixture(name: str, dtype=torch.float64, batch: int = 10, seed: int = 711):
    """Fresh model + (X, y_cls, y_reg) from the reference's seed convention."""
    torch.manual_seed(seed)
    model = build_model(name).to(dtype)
    torch.manual_seed(seed)
    X = torch.randn(batch, *input_shape(name), dtype=dtype)
    C = n_outputs(name)
    y_cls = torch.randint(C, (batch,))
    y_reg = torch.randn(batch, C, dtype=dtype)
    return model, X, y_cls, y_reg


def load_state(model: nn.Module, arrays: dict, prefix: str = "w.") -> nn.Module:
    sd = {k[len(prefix):]: torch.as_tensor(v) for k, v in arrays.items() if k.startswith(prefix)}
    model.load_state_dict(sd)
    return model
