"""`EqualLinear` -- the only piece of the reference's StyleGAN2 package that the ReStyle encoder heads use
(reference models/stylegan2/model.py:129-158).  The reference module JIT-compiles two unrelated CUDA ops at import
(models/stylegan2/op/fused_act.py:9-15); they are not on the StyleGAN3 path and are not rebuilt,
so the `activation` branch uses the plain leaky-ReLU definition of that op (scale sqrt(2), slope 0.2).
"""
import math

import torch
from torch import nn
from torch.nn import functional as F


class EqualLinear(nn.Module):
    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def forward(self, input):  # pylint: disable=redefined-builtin
        if self.activation:
            out = F.linear(input, self.weight * self.scale)
            return F.leaky_relu(out + (self.bias * self.lr_mul).view(1, -1), negative_slope=0.2) * math.sqrt(2)
        return F.linear(input, self.weight * self.scale, bias=self.bias * self.lr_mul)

    def __repr__(self):
        return f'{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]})'
