"""`SG3Generator`: loads / builds the StyleGAN3 decoder used by the encoders, PTI and the editing tools
(API of reference models/stylegan3/model.py:19-65).

Weight-layout contract kept from the reference: a `.pkl` is `pickle.load(f)['G_ema']` (a persistent object whose
embedded NVIDIA module source is exec'd against this package's `torch_utils` / `dnnlib`, so it runs on the HIP
kernels); anything else is a plain `Generator.state_dict()` loaded strictly, or -- if that fails -- without the
`synthesis.input.transform` entry (:59-65).  `config="landscape"` selects the config-T sizes (:29-40), everything
else config-R (:42-54).  Unlike the reference, `device=` is explicit and the module is moved to the GPU for every
branch, not only for pickles.
"""
import pickle
from enum import Enum
from pathlib import Path
from typing import Optional

import torch

from models.stylegan3.networks_stylegan3 import Generator

CONFIG_T = dict(z_dim=512, c_dim=0, w_dim=512, img_channels=3, channel_base=32768, channel_max=512,
                magnitude_ema_beta=0.9988915792636801, mapping_kwargs={'num_layers': 2})
CONFIG_R = dict(z_dim=512, c_dim=0, w_dim=512, img_channels=3, channel_base=65536, channel_max=1024, conv_kernel=1,
                filter_size=6, magnitude_ema_beta=0.9988915792636801, output_scale=0.25, use_radial_filters=True)


class GeneratorType(str, Enum):
    ALIGNED = "aligned"
    UNALIGNED = "unaligned"

    def __str__(self):
        return str(self.value)


class SG3Generator(torch.nn.Module):
    def __init__(self, checkpoint_path: Optional[Path] = None, res: int = 1024, config: str = None, device='cuda'):
        super().__init__()
        print(f"Loading StyleGAN3 generator from path: {checkpoint_path}")
        if str(checkpoint_path).endswith("pkl"):
            with open(checkpoint_path, "rb") as f:
                self.decoder = pickle.load(f)['G_ema'].to(device)
            print('Done!')
            return
        kwargs = CONFIG_T if config == "landscape" else CONFIG_R
        self.decoder = Generator(img_resolution=res, **kwargs)
        if checkpoint_path is not None:
            self._load_checkpoint(checkpoint_path)
        print('Done!')

    def _load_checkpoint(self, checkpoint_path):
        ckpt = torch.load(checkpoint_path, map_location='cpu')
        try:
            self.decoder.load_state_dict(ckpt, strict=True)
        except RuntimeError:
            ckpt = {k: v for k, v in ckpt.items() if "synthesis.input.transform" not in k}
            self.decoder.load_state_dict(ckpt, strict=False)
