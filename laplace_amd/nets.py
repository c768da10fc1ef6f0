"""Synthetic-weight model definitions for the BASELINE.json configs (no torchvision here).

Only shapes matter: the configs use random-init weights and synthetic data.  ResNet-18 follows the
CIFAR layout named in SURVEY.md §8 (3x3 stride-1 stem, no max-pool, BasicBlock x [2,2,2,2]); its
BatchNorm affine parameters are frozen because the reference's KFAC path supports nn.Linear /
nn.Conv2d only (docs/index.md:364-366; baselaplace.py:115-125 treats frozen params as non-Laplace).
"""
from __future__ import annotations

import torch
from torch import nn


class BasicBlock(nn.Module):
    def __init__(self, cin: int, cout: int, stride: int, act=torch.relu):
        super().__init__()
        self.act = act
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        out = self.act(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.act(out + (x if self.downsample is None else self.downsample(x)))


class ResNet18(nn.Module):
    """``act`` defaults to ReLU (the benchmark model).  Tests that compare two *separately executed*
    forward/backward passes use a smooth activation: with ReLU, fp32 rounding differences between MIOpen
    solvers flip a handful of pre-activations that sit within 1e-6 of zero, which changes individual
    gradients by O(1) and is a property of the host framework, not of the curvature kernels."""

    def __init__(self, num_classes: int = 10, freeze_bn: bool = True, act=torch.relu):
        super().__init__()
        self.act = act
        self.conv1 = nn.Conv2d(3, 64, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        blocks, cin = [], 64
        for cout, stride in ((64, 1), (64, 1), (128, 2), (128, 1), (256, 2), (256, 1), (512, 2), (512, 1)):
            blocks.append(BasicBlock(cin, cout, stride, act))
            cin = cout
        self.layers = nn.Sequential(*blocks)
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512, num_classes)
        if freeze_bn:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.weight.requires_grad_(False)
                    m.bias.requires_grad_(False)

    def forward(self, x):
        x = self.act(self.bn1(self.conv1(x)))
        x = self.layers(x)
        return self.fc(torch.flatten(self.pool(x), 1))


def lenet5(num_classes: int = 10) -> nn.Sequential:
    """Config c2: LeNet-5 on 3x32x32 inputs."""
    return nn.Sequential(
        nn.Conv2d(3, 6, 5), nn.Tanh(), nn.MaxPool2d(2), nn.Conv2d(6, 16, 5), nn.Tanh(), nn.MaxPool2d(2),
        nn.Flatten(), nn.Linear(400, 120), nn.Tanh(), nn.Linear(120, 84), nn.Tanh(), nn.Linear(84, num_classes),
    )


def mlp_1_50_1() -> nn.Sequential:
    """Config c1 (examples/regression_example.py:17-21 of the reference)."""
    return nn.Sequential(nn.Linear(1, 50), nn.Tanh(), nn.Linear(50, 1))
