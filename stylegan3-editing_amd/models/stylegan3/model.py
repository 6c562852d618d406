"""`SG3Generator`: loads / builds the StyleGAN3 decoder used by the encoders, PTI and the editing tools
(API of reference models/stylegan3/model.py:19-65).

Weight-layout contract kept from the reference: a `.pkl` is `pickle.load(f)['G_ema']` (a persistent object whose
embedded NVIDIA module source is exec'd against this package's `torch_utils` / `dnnlib`, so it runs on the HIP
kernels); anything else is a plain `Generator.state_dict()` loaded strictly, or -- if that fails -- without the
`synthesis.input.transform` entry (:59-65).  `config="landscape"` selects the config-T sizes (:29-40), everything
else config-R (:42-54).  `device=` replaces the reference's hard-coded `.cuda()` on the pickle branch; a generator built
from a state dict stays on the CPU until the caller moves it, as in the reference.
"""
import pickle
from enum import Enum
from pathlib import Path
from typing import Optional

import torch

from models.stylegan3.networks_stylegan3 import Generator

_COMMON = dict(z_dim=512, c_dim=0, w_dim=512, img_channels=3, magnitude_ema_beta=0.5 ** (32 / (20 * 1e3)))
# translation-equivariant config: 3x3 convs, separable 12-tap filters, two mapping layers in the landscape checkpoints
CONFIG_T = dict(_COMMON, channel_base=32768, channel_max=512, mapping_kwargs={'num_layers': 2})
# rotation-equivariant config: 1x1 convs, radial 6-tap filters, twice the channels at a quarter of the output scale
CONFIG_R = dict(_COMMON, channel_base=65536, channel_max=1024, conv_kernel=1, filter_size=6, output_scale=0.25,
                use_radial_filters=True)
_SHAPE_DEPENDENT = 'synthesis.input.transform'


class GeneratorType(str, Enum):
    ALIGNED = "aligned"
    UNALIGNED = "unaligned"

    def __str__(self):
        return str(self.value)


def _from_pickle(path, device):
    with open(path, 'rb') as fh:
        return pickle.load(fh)['G_ema'].to(device)


def _from_state_dict(net, path):
    state = torch.load(path, map_location='cpu')
    try:
        net.load_state_dict(state, strict=True)
    except RuntimeError:
        # checkpoints saved after a batched call carry a [B,3,3] transform buffer
        net.load_state_dict({k: v for k, v in state.items() if _SHAPE_DEPENDENT not in k}, strict=False)
    return net


class SG3Generator(torch.nn.Module):
    def __init__(self, checkpoint_path: Optional[Path] = None, res: int = 1024, config: str = None, device='cuda'):
        super().__init__()
        print(f"Loading StyleGAN3 generator from path: {checkpoint_path}")
        if str(checkpoint_path).endswith("pkl"):
            self.decoder = _from_pickle(checkpoint_path, device)
        else:
            self.decoder = Generator(img_resolution=res, **(CONFIG_T if config == "landscape" else CONFIG_R))
            if checkpoint_path is not None:
                _from_state_dict(self.decoder, checkpoint_path)
        print('Done!')
