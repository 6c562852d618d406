"""StyleCLIP global directions in StyleSpace (reference editing/styleclip_global_directions/global_direction.py:7-62).

delta_i_c [num_style_channels, 512] holds, per StyleSpace channel, the CLIP-space image direction that channel causes;
a text direction delta_i (unit vector) is projected on it, channels with relevance below beta are dropped, the rest is
normalised to a peak of 1 and scaled by each channel's standard deviation.

The arithmetic lives in three module-level functions (`channel_relevance`, `features_channels_to_s`,
`prompt_direction`); `StyleCLIPGlobalDirection` keeps the reference's method names on top of them.  The CLIP text
encoder is external to the synthesis hot path: pass `text_encoder` (a callable list[str] -> [n, 512] features) or
call `get_delta_s_from_delta_i` with a direction computed elsewhere.
"""
import torch


def channel_relevance(delta_i_c, delta_i, beta):
    """Projection of the text direction on every channel's image direction, thresholded at beta, peak-normalised."""
    relevance = delta_i_c @ delta_i
    kept = relevance.masked_fill(relevance.abs() < beta, 0.0)
    top = kept.abs().max()
    return kept / top if top > 0 else kept


def features_channels_to_s(channels, std, example_s):
    """Split the flat channel vector into the per-layer dict layout of `example_s`, scaled by the per-channel std."""
    sizes = [int(example_s[key].shape[1]) for key in example_s]
    parts = torch.split(channels[:sum(sizes)], sizes)
    return {key: (part * std[key]).unsqueeze(0) for key, part in zip(example_s, parts)}


def _unit(v, **kw):
    return v / v.norm(**kw)


def prompt_direction(text_encoder, templates, target, neutral):
    """Unit vector from the template-averaged embedding of `neutral` to that of `target`."""
    def embed(text):
        rows = _unit(text_encoder([t.format(text) for t in templates]), dim=-1, keepdim=True)
        return _unit(rows.mean(dim=0))
    with torch.no_grad():
        return _unit(embed(target) - embed(neutral))


class StyleCLIPGlobalDirection:

    def __init__(self, delta_i_c, s_std, text_prompts_templates, s_avg, text_encoder=None):
        self.delta_i_c, self.s_std, self.s_avg = delta_i_c, s_std, s_avg
        self.text_prompts_templates = text_prompts_templates
        self.text_encoder = text_encoder

    def get_delta_i(self, text_prompts):
        """text_prompts = [target, neutral]."""
        if self.text_encoder is None:
            raise RuntimeError('StyleCLIPGlobalDirection: no text_encoder was supplied (the CLIP model is external to this package)')
        target, neutral = text_prompts
        return prompt_direction(self.text_encoder, self.text_prompts_templates, target, neutral)

    def get_delta_s_from_delta_i(self, delta_i, beta):
        return features_channels_to_s(channel_relevance(self.delta_i_c, delta_i, beta), self.s_std, self.s_avg)

    def get_delta_s(self, neutral_text, target_text, beta):
        return self.get_delta_s_from_delta_i(self.get_delta_i([target_text, neutral_text]).float(), beta)
