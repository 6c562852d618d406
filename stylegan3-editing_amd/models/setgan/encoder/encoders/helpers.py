"""IR / IR-SE residual units of the ReStyle encoder backbone (ArcFace-style ResNet), module-for-module compatible
with the reference so its checkpoints load (reference models/setgan/encoder/encoders/helpers.py: get_blocks :30-55,
SEModule :57-73, bottleneck_IR :76-95, bottleneck_IR_SE :98-120).

`forward()` is the plain PyTorch definition (used on CPU and for training); on a GPU in eval mode the owning encoder
calls `forward_hip()`, which runs the two 3x3 convolutions on the matrix cores with their element-wise neighbours
fused (torch_utils/ops/plain_conv.py): BN1 as an input affine of conv1, PReLU as its epilogue, BN2 folded into conv2.
"""
import os
from collections import namedtuple

import torch
from torch.nn import AdaptiveAvgPool2d, BatchNorm2d, Conv2d, MaxPool2d, Module, PReLU, ReLU, Sequential, Sigmoid


class Flatten(Module):
    def forward(self, input):  # pylint: disable=redefined-builtin
        return input.view(input.size(0), -1)


def l2_norm(input, axis=1):  # pylint: disable=redefined-builtin
    return torch.div(input, torch.norm(input, 2, axis, True))


class Bottleneck(namedtuple('Block', ['in_channel', 'depth', 'stride'])):
    """One residual unit: input channels, output channels, stride of its second convolution."""


def get_block(in_channel, depth, num_units, stride=2):
    return [Bottleneck(in_channel, depth, stride)] + [Bottleneck(depth, depth, 1) for _ in range(num_units - 1)]


_STAGES = {50: (3, 4, 14, 3), 100: (3, 13, 30, 3), 152: (3, 8, 36, 3)}


def get_blocks(num_layers):
    if num_layers not in _STAGES:
        raise ValueError(f"Invalid number of layers: {num_layers}. Must be one of [50, 100, 152]")
    units = _STAGES[num_layers]
    widths = [(64, 64), (64, 128), (128, 256), (256, 512)]
    return [get_block(in_channel=i, depth=d, num_units=u) for (i, d), u in zip(widths, units)]


# GPU inference runs the squeeze-and-excitation tail on two HIP kernels unless SG3_SE_TORCH=1 (A/B timing)
_SE_KERNELS = os.environ.get('SG3_SE_TORCH', '0') != '1'


class SEModule(Module):
    """Squeeze-and-excitation: x * sigmoid(fc2(relu(fc1(mean_hw(x)))))."""

    def __init__(self, channels, reduction):
        super().__init__()
        self.avg_pool = AdaptiveAvgPool2d(1)
        self.fc1 = Conv2d(channels, channels // reduction, kernel_size=1, padding=0, bias=False)
        self.relu = ReLU(inplace=True)
        self.fc2 = Conv2d(channels // reduction, channels, kernel_size=1, padding=0, bias=False)
        self.sigmoid = Sigmoid()

    def gate(self, x):
        # the two 1x1 convolutions act on a [N,C,1,1] tensor: plain matrix products
        m = x.mean(dim=(2, 3))
        m = torch.relu(m @ self.fc1.weight.flatten(1).t())
        return torch.sigmoid(m @ self.fc2.weight.flatten(1).t()).unsqueeze(-1).unsqueeze(-1)

    def forward(self, x):
        return x * self.gate(x)


class _ResidualUnit(Module):
    use_se = False

    def __init__(self, in_channel, depth, stride):
        super().__init__()
        self.stride = stride
        if in_channel == depth:
            self.shortcut_layer = MaxPool2d(1, stride)
        else:
            self.shortcut_layer = Sequential(Conv2d(in_channel, depth, (1, 1), stride, bias=False), BatchNorm2d(depth))
        layers = [BatchNorm2d(in_channel),
                  Conv2d(in_channel, depth, (3, 3), (1, 1), 1, bias=False), PReLU(depth),
                  Conv2d(depth, depth, (3, 3), stride, 1, bias=False), BatchNorm2d(depth)]
        if self.use_se:
            layers.append(SEModule(depth, 16))
        self.res_layer = Sequential(*layers)
        self._packed = None

    def forward(self, x):
        return self.res_layer(x) + self.shortcut_layer(x)

    # ---- fused eval-mode path on libsg3hip ----
    def _pack(self):
        from torch_utils.ops.plain_conv import ACT_NONE, ACT_PRELU, PackedConv, bn_affine
        bn1, conv1, prelu, conv2, bn2 = (self.res_layer[i] for i in range(5))
        a1, b1 = bn_affine(bn1)
        a2, b2 = bn_affine(bn2)
        pk = dict(conv1=PackedConv(conv1.weight, in_scale=a1, in_shift=b1, act=ACT_PRELU, slope=prelu.weight, stride=1, padding=1),
                  conv2=PackedConv(conv2.weight, out_scale=a2, bias=b2, act=ACT_NONE, stride=self.stride, padding=1))
        if isinstance(self.shortcut_layer, Sequential):
            sc, sbn = self.shortcut_layer[0], self.shortcut_layer[1]
            a, b = bn_affine(sbn)
            pk['shortcut'] = PackedConv(sc.weight, out_scale=a, bias=b, act=ACT_NONE, stride=self.stride, padding=0)
        self._packed = pk

    def forward_hip(self, x):
        if self._packed is None:
            self._pack()
        pk = self._packed
        res = pk['conv2'](pk['conv1'](x))
        if 'shortcut' in pk:
            shortcut = pk['shortcut'](x)
        else:
            shortcut = x if self.stride == 1 else x[:, :, ::self.stride, ::self.stride]
        if self.use_se:
            se = self.res_layer[5]
            if res.is_cuda and res.dtype == torch.float32 and not torch.is_grad_enabled() and _SE_KERNELS:
                from torch_utils.ops import se_ops
                return se_ops.se_residual(res.contiguous(), shortcut, se.fc1.weight, se.fc2.weight)      # two launches
            return torch.addcmul(shortcut, res, se.gate(res))
        return res + shortcut


class bottleneck_IR(_ResidualUnit):  # noqa: N801  (reference class name)
    use_se = False


class bottleneck_IR_SE(_ResidualUnit):  # noqa: N801  (reference class name)
    use_se = True


class BasicBlock(Module):
    """ResNet18/34 residual block with torchvision's module names (`conv1 bn1 relu conv2 bn2 downsample`), so the `body.*` keys
    of a ReStyle ResNetBackboneEncoder checkpoint (reference restyle_psp_encoders.py:65-77 flattens resnet34's layer1..4)
    load: y = relu(bn2(conv2(relu(bn1(conv1(x))))) + (downsample(x) | x)); conv1 carries the stride."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, (3, 3), stride, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.relu = ReLU(inplace=True)
        self.conv2 = Conv2d(planes, planes, (3, 3), 1, 1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = Sequential(Conv2d(inplanes, planes, (1, 1), stride, bias=False), BatchNorm2d(planes))
        self.stride = stride
        self._packed = None

    def forward(self, x):
        out = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        return self.relu(out + (x if self.downsample is None else self.downsample(x)))

    def _pack(self):
        from torch_utils.ops.plain_conv import ACT_LRELU, ACT_NONE, PackedConv, bn_affine
        a1, b1 = bn_affine(self.bn1)
        a2, b2 = bn_affine(self.bn2)
        zero = torch.zeros([1], device=self.conv1.weight.device)            # ReLU = leaky ReLU with slope 0 in the epilogue
        pk = dict(conv1=PackedConv(self.conv1.weight, out_scale=a1, bias=b1, act=ACT_LRELU, slope=zero, stride=self.stride, padding=1),
                  conv2=PackedConv(self.conv2.weight, out_scale=a2, bias=b2, act=ACT_NONE, stride=1, padding=1))
        if self.downsample is not None:
            a, b = bn_affine(self.downsample[1])
            pk['shortcut'] = PackedConv(self.downsample[0].weight, out_scale=a, bias=b, act=ACT_NONE, stride=self.stride, padding=0)
        self._packed = pk

    def forward_hip(self, x):
        if self._packed is None:
            self._pack()
        pk = self._packed
        res = pk['conv2'](pk['conv1'](x))
        return torch.relu_(res.add_(pk['shortcut'](x) if 'shortcut' in pk else x))


def resnet34_blocks():
    """The 16 BasicBlocks of torchvision.models.resnet34's layer1..layer4 in order (3, 4, 6, 3 blocks of 64, 128, 256, 512
    channels; the first block of layers 2..4 has stride 2 and a 1x1 projection shortcut)."""
    blocks, inplanes = [], 64
    for planes, count, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)):
        for i in range(count):
            blocks.append(BasicBlock(inplanes, planes, stride if i == 0 else 1))
            inplanes = planes
    return blocks
