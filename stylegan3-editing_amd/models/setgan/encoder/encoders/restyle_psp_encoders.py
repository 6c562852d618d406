"""ReStyle-pSp encoder backbones (reference models/setgan/encoder/encoders/restyle_psp_encoders.py).

`BackboneEncoder` (:10-50): conv3x3(input_nc -> 64) + BN + PReLU at 256^2, the IR / IR-SE stages down to [N,512,16,16],
then `n_styles` GradualStyleBlock heads, stacked to [N, n_styles, 512].  Module names and therefore state_dict keys are
the reference's.  In eval mode on a GPU the backbone runs on libsg3hip's matrix-core convolution with BatchNorm / PReLU
fused; the first level of all style heads as one strip convolution on the same kernel, the deeper levels (tiny maps,
weight-bandwidth bound) as batched GEMMs over all heads on patches written by `sg3_unfold3x3s2`.
`ResNetBackboneEncoder` (:53-97): the same heads on a ResNet34 trunk (7x7 stride-2 stem without max-pool, then torchvision's
layer1..layer4 BasicBlocks flattened into `body`, so the keys are `conv1 / bn1 / relu / body.{0..15}.{conv1,bn1,conv2,bn2,
downsample.{0,1}} / styles.*`).  The reference seeds the trunk from `torchvision.models.resnet34(pretrained=True)`, which is
neither installed nor downloadable here: the trunk is built from this repository's own `BasicBlock` with default
initialisation and takes its weights from the encoder checkpoint (which holds all of them).
"""
import torch
from torch import nn
from torch.nn import BatchNorm2d, Conv2d, Module, PReLU, Sequential

from models.setgan.encoder.encoders.helpers import bottleneck_IR, bottleneck_IR_SE, get_blocks, resnet34_blocks
from models.setgan.encoder.encoders.map2style import GradualStyleBlock


# Rows (images x pixels) up to which a head level / the EqualLinear runs on sg3_head_gemm instead of torch.baddbmm (rocBLAS fp32).
# Measured at 16 heads, N = 512 (profiles/r04_head_gemm.txt): K = 4608 -- 27.3 vs 28.0 us at 8 rows (both at the weight-read
# rate, 5.5 TB/s), rocBLAS ahead from 32 rows on (its LDS-tiled A operand; ours re-reads A from L2 per column block);
# K = 512 -- 11.4 vs 21.7 us up to 64 rows, and the LeakyReLU in front of it rides along.
_HEAD_GEMM_MAX_ROWS = (16, 64)


class _StyleHeadEncoder(Module):
    """What the two backbones share: `n_styles` GradualStyleBlock heads on the trunk's [N,512,16,16] map, the packed-weight
    cache of the fused GPU path, and the dispatch between that path and the plain PyTorch definition.  Subclasses provide the
    trunk: `_trunk_torch(x)`, `_pack_trunk(pk)`, `_trunk_hip(pk, x)`."""

    def _init_heads(self, n_styles):
        self.styles = nn.ModuleList([GradualStyleBlock(512, 512, 16) for _ in range(n_styles)])
        self.style_count = n_styles
        self._packed = None
        self._packed_key = None
        # a parent's load_state_dict (pSp / e4e: net.load_state_dict(ckpt)) recurses through _load_from_state_dict and never
        # calls this module's load_state_dict override; the post hook runs for every module of the tree that is loaded
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_packed())

    def _weights_key(self):
        """(data_ptr, _version) of every parameter and buffer the packed / folded copies are derived from: in-place edits
        (optimizer.step, EMA copy_, new BatchNorm statistics) and re-bound tensors change it."""
        return tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    def _combine(self, per_style):
        """[style_count x [N,512]] -> [N, style_count, 512] (the e4e variant overrides this with w0 + deltas)."""
        return torch.stack(per_style, dim=1)

    # plain PyTorch definition (CPU, training)
    def _forward_torch(self, x):
        x = self._trunk_torch(x)
        return self._combine([style(x) for style in self.styles])

    def invalidate_packed(self):
        """Drop the packed / folded weights (call after changing parameters or BatchNorm statistics)."""
        self._packed = None
        self._packed_key = None
        for m in getattr(self, 'body', ()):
            m._packed = None

    def train(self, mode=True):
        self.invalidate_packed()
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        self.invalidate_packed()
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_packed()          # .to(device) / .float() move the parameters the packed copies came from
        return super()._apply(fn, *args, **kwargs)

    def _pack(self):
        pk = {}
        self._pack_trunk(pk)
        # Head convolutions.  Every style block starts from the same [N,512,16,16] map and halves it to 8x8, 4x4, 2x2, 1x1.
        # Level 1 of ALL heads is one stride-2 convolution with heads * 512 output channels on the split-precision matrix-core kernel
        # (bias + LeakyReLU in its epilogue): 77 GFLOP at batch 16, compute bound.  Its 8x8 output maps are too few pixels for the
        # kernel's 4-row x 32-column tile, so the N images travel side by side as ONE strip image (see _forward_kernels).
        from torch_utils.ops.plain_conv import ACT_LRELU, PackedConv
        dev = self.styles[0].convs[0].weight.device
        pk['slope'] = float(self.styles[0].convs[1].negative_slope)
        pk['head0'] = PackedConv(torch.cat([s.convs[0].weight for s in self.styles], dim=0), bias=torch.cat([s.convs[0].bias for s in self.styles], dim=0),
                                 act=ACT_LRELU, slope=torch.full([1], pk['slope'], device=dev), stride=2, padding=1)
        # From level 2 on every head has its own input: tiny maps, bounded by reading the weights (heads x 9.4 MB per level): ONE
        # batched GEMM per level over all heads, [heads, N*pixels, 9*512] x [heads, 9*512, 512], on patches written by one launch of
        # sg3_unfold3x3s2 (rows ordered tap-major, channels innermost: the weights' rows are permuted to match)
        n_levels = (len(self.styles[0].convs) // 2) - 1
        pk['tail_w'] = [torch.stack([s.convs[2 * (l + 1)].weight.permute(2, 3, 1, 0).reshape(-1, s.out_c) for s in self.styles]).contiguous()
                        for l in range(n_levels)]                                                       # [heads, 9*I, O]
        pk['tail_b'] = [torch.stack([s.convs[2 * (l + 1)].bias for s in self.styles]).unsqueeze(1) for l in range(n_levels)]
        pk['lin_w'] = torch.stack([(s.linear.weight * s.linear.scale).t() for s in self.styles]).contiguous()   # [heads, in, out]
        pk['lin_b'] = torch.stack([s.linear.bias * s.linear.lr_mul for s in self.styles]).unsqueeze(1)
        # The same operands as matrix-instruction fragments for sg3_head_gemm, which takes the GEMMs with few rows (see
        # _HEAD_GEMM_MAX_ROWS); the fp32 tensors above stay for the larger ones and for the 'fp32' repeat of a flagged forward
        from torch_utils.ops.head_gemm import PackedHeadWeights
        pk['tail_g'] = [PackedHeadWeights(w, b) if PackedHeadWeights.supports(w) else None for w, b in zip(pk['tail_w'], pk['tail_b'])]
        pk['lin_g'] = PackedHeadWeights(pk['lin_w'], pk['lin_b']) if PackedHeadWeights.supports(pk['lin_w']) else None
        self._packed = pk

    def _forward_hip(self, x):
        """Split-precision convolutions first; if any of them met an operand outside the fp16 range (flagged on the
        device) the forward is repeated on the exact fp32 kernels."""
        from torch_utils.ops import plain_conv
        plain_conv.reset_overflow(x.device)
        out = self._forward_kernels(x)
        if plain_conv.precision != 'fp32' and plain_conv.overflowed(x.device):
            saved, plain_conv.precision = plain_conv.precision, 'fp32'
            try:
                out = self._forward_kernels(x)
            finally:
                plain_conv.precision = saved
        return out

    def _forward_kernels(self, x):
        key = self._weights_key()
        if self._packed is not None and key != self._packed_key:
            self.invalidate_packed()          # the source tensors changed behind the hooks (in-place update): re-pack
        if self._packed is None:
            self._pack()
            self._packed_key = key
        pk = self._packed
        from torch_utils.ops import plain_conv
        from torch_utils.ops.unfold_ops import unfold3x3s2
        x = self._trunk_hip(pk, x.float())
        n, ci, sh, sw = (int(v) for v in x.shape)
        heads, c = len(self.styles), self.styles[0].out_c
        # The strip: image i occupies columns [period*i, period*i + sw) of a [1, C, sh, period*N] image whose other columns stay zero;
        # period is even and > sw, so every image starts on an even column (stride 2 keeps each image's own pixel parity) and finds
        # its zero padding in the gap.  Output column (period/2)*i + ox is pixel ox of image i (one junk column per image for even sw).
        period = sw + 2 - (sw & 1)
        # One strip per (batch, map shape, device), NEVER replaced while this pack lives: a captured ReStyle graph bakes the strip's
        # address and relies on its gap columns staying zero, so an eager call at another batch size (the ragged tail of a video,
        # run_on_batch(x[:2])) must not hand that block back to the allocator.  GraphedReStyleStep pins the whole pack besides.
        strip = pk.setdefault('strips', {}).get((n, ci, sh, sw, x.device))
        if strip is None:
            strip = pk['strips'][(n, ci, sh, sw, x.device)] = torch.zeros([1, ci, sh, period * n], dtype=torch.float32, device=x.device)
        strip[0].view(ci, sh, n, period)[..., :sw].copy_(x.permute(1, 2, 0, 3))
        h = pk['head0'].run(strip)                                                                    # [1, heads*C, oh, (period/2)*N]
        oh, ow = (sh + 1) // 2, (sw + 1) // 2
        src = h[0].view(heads, c, oh, n, period // 2)[..., :ow].permute(0, 3, 1, 2, 4)                # [heads, N, C, oh, ow] view
        slope = 1.0                                                                                   # level 1 leaves the kernel activated
        hh = None
        split = plain_conv.precision != 'fp32'
        for w, b, wg in zip(pk['tail_w'], pk['tail_b'], pk['tail_g']):
            cols = unfold3x3s2(src, slope)                                                         # [heads, N*oh'*ow', 9*C]
            oh, ow = (oh + 1) // 2, (ow + 1) // 2
            if split and wg is not None and wg.usable and n * oh * ow <= _HEAD_GEMM_MAX_ROWS[0]:
                hh = wg.run(cols)
            else:
                hh = torch.baddbmm(b, cols, w)                                                     # [heads, N*oh*ow, C], before its LeakyReLU
            src = hh.view(heads, n, oh, ow, c).permute(0, 1, 4, 2, 3)
            slope = pk['slope']
        if oh * ow != 1 or hh is None:
            raise RuntimeError(f'style heads: a {sh}x{sw} feature map does not reduce to 1x1 in {len(pk["tail_w"]) + 1} halvings')
        if split and pk['lin_g'] is not None and pk['lin_g'].usable and n <= _HEAD_GEMM_MAX_ROWS[1]:
            codes = pk['lin_g'].run(hh.view(heads, n, c), pk['slope'])                            # LeakyReLU + EqualLinear of every head
        else:
            last = torch.nn.functional.leaky_relu(hh.view(heads, n, c), pk['slope'])
            codes = torch.baddbmm(pk['lin_b'], last, pk['lin_w'])
        return self._combine(list(codes.unbind(0)))

    def forward(self, x):
        if x.is_cuda and not self.training and not torch.is_grad_enabled():
            return self._forward_hip(x)
        return self._forward_torch(x)


class BackboneEncoder(_StyleHeadEncoder):
    def __init__(self, num_layers, mode='ir', n_styles=18, opts=None):
        super().__init__()
        assert num_layers in [50, 100, 152], 'num_layers should be 50,100, or 152'
        assert mode in ['ir', 'ir_se'], 'mode should be ir or ir_se'
        unit = bottleneck_IR if mode == 'ir' else bottleneck_IR_SE
        input_nc = getattr(opts, 'input_nc', 6) if opts is not None else 6
        self.input_layer = Sequential(Conv2d(input_nc, 64, (3, 3), 1, 1, bias=False), BatchNorm2d(64), PReLU(64))
        self.body = Sequential(*[unit(b.in_channel, b.depth, b.stride) for stage in get_blocks(num_layers) for b in stage])
        self._init_heads(n_styles)

    def _trunk_torch(self, x):
        return self.body(self.input_layer(x))

    def _pack_trunk(self, pk):
        from torch_utils.ops.plain_conv import ACT_PRELU, PackedConv, bn_affine
        conv, bn, prelu = self.input_layer[0], self.input_layer[1], self.input_layer[2]
        a, b = bn_affine(bn)
        pk['stem'] = PackedConv(conv.weight, out_scale=a, bias=b, act=ACT_PRELU, slope=prelu.weight, stride=1, padding=1)

    def _trunk_hip(self, pk, x):
        x = pk['stem'](x)
        for unit in self.body:
            x = unit.forward_hip(x)
        return x


class ResNetBackboneEncoder(_StyleHeadEncoder):
    """ResNet34 trunk (reference :53-97): conv7x7 s2 (input_nc -> 64) + BN + PReLU at 128^2 (no max-pool), then the 3 + 4 + 6 + 3
    BasicBlocks of torchvision's resnet34 as one flat `body`, down to [N,512,16,16]."""

    def __init__(self, n_styles=18, opts=None):
        super().__init__()
        input_nc = getattr(opts, 'input_nc', 6) if opts is not None else 6
        self.conv1 = Conv2d(input_nc, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.relu = PReLU(64)
        self.body = Sequential(*resnet34_blocks())
        self._init_heads(n_styles)

    def _trunk_torch(self, x):
        return self.body(self.relu(self.bn1(self.conv1(x))))

    def _pack_trunk(self, pk):
        from torch_utils.ops.plain_conv import bn_affine
        a, b = bn_affine(self.bn1)
        # the 7x7 stride-2 stem (0.6 % of the trunk's FLOPs, 6 input channels) is the one convolution left to the library: the
        # matrix-core kernel takes 1x1 / 3x3 windows.  BatchNorm is folded into its weights, the PReLU follows as one op
        pk['stem_w'] = (self.conv1.weight.detach() * a.view(-1, 1, 1, 1)).contiguous()
        pk['stem_b'] = b.contiguous()

    def _trunk_hip(self, pk, x):
        x = torch.nn.functional.prelu(torch.nn.functional.conv2d(x, pk['stem_w'], pk['stem_b'], stride=2, padding=3), self.relu.weight)
        for block in self.body:
            x = block.forward_hip(x)
        return x

