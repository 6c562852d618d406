"""Which `opts.encoder_type` values belong to which ReStyle wrapper (the table of reference utils/model_utils.py:1-5, which
`load_encoder` uses to choose between `pSp` and `e4e`).  The e4e encoders are the progressive forms of the two pSp backbones."""
_BACKBONES = ('BackboneEncoder', 'ResNetBackboneEncoder')

ENCODER_TYPES = {
    'pSp': list(_BACKBONES),
    'e4e': [name.replace('BackboneEncoder', 'ProgressiveBackboneEncoder') for name in _BACKBONES],
}
