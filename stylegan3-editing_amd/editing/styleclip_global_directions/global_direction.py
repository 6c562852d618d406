"""StyleCLIP global directions in StyleSpace (reference editing/styleclip_global_directions/global_direction.py:7-62).

delta_i_c [num_style_channels, 512] holds, per StyleSpace channel, the CLIP-space image direction that channel causes;
a text direction delta_i (unit vector) is projected on it, channels with relevance below beta are dropped, the rest is
normalised to a peak of 1 and scaled by each channel's standard deviation.

The CLIP text encoder is external to the synthesis hot path: pass `text_encoder` (a callable list[str] -> [n, 512]
features) or call `get_delta_s_from_delta_i` with a direction computed elsewhere.
"""
import torch


def features_channels_to_s(channels, std, example_s):
    """Split the flat channel vector into the per-layer dict layout of `example_s`, scaled by the per-channel std."""
    sizes = [int(example_s[key].shape[1]) for key in example_s]
    parts = torch.split(channels[:sum(sizes)], sizes)
    return {key: (part * std[key]).unsqueeze(0) for key, part in zip(example_s, parts)}


class StyleCLIPGlobalDirection:

    def __init__(self, delta_i_c, s_std, text_prompts_templates, s_avg, text_encoder=None):
        self.delta_i_c = delta_i_c
        self.s_std = s_std
        self.text_prompts_templates = text_prompts_templates
        self.text_encoder = text_encoder
        self.s_avg = s_avg

    def get_delta_s(self, neutral_text, target_text, beta):
        delta_i = self.get_delta_i([target_text, neutral_text]).float()
        return self.get_delta_s_from_delta_i(delta_i, beta)

    def get_delta_s_from_delta_i(self, delta_i, beta):
        r_c = torch.matmul(self.delta_i_c, delta_i)
        delta_s = torch.where(torch.abs(r_c) < beta, torch.zeros_like(r_c), r_c)
        peak = torch.abs(delta_s).max()
        if peak > 0:
            delta_s = delta_s / peak
        return features_channels_to_s(delta_s, self.s_std, self.s_avg)

    def get_delta_i(self, text_prompts):
        text_features = self._get_averaged_text_features(text_prompts)
        delta_t = text_features[0] - text_features[1]
        return delta_t / torch.norm(delta_t)

    def _get_averaged_text_features(self, text_prompts):
        if self.text_encoder is None:
            raise RuntimeError('StyleCLIPGlobalDirection: no text_encoder was supplied (the CLIP model is external to this package)')
        feats = []
        with torch.no_grad():
            for text_prompt in text_prompts:
                emb = self.text_encoder([template.format(text_prompt) for template in self.text_prompts_templates])
                emb = emb / emb.norm(dim=-1, keepdim=True)
                emb = emb.mean(dim=0)
                feats.append(emb / emb.norm())
        return torch.stack(feats, dim=1).t()
