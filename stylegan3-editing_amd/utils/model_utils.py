"""Encoder families of the two ReStyle wrappers (reference utils/model_utils.py:1-5): which `opts.encoder_type` values belong to
`pSp` and which to `e4e`; `load_encoder` picks the wrapper class from it."""
ENCODER_TYPES = {
    'pSp': ['BackboneEncoder', 'ResNetBackboneEncoder'],
    'e4e': ['ProgressiveBackboneEncoder', 'ResNetProgressiveBackboneEncoder'],
}
