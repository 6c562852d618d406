"""StyleSpace statistics (reference editing/styleclip_global_directions/preprocess/s_statistics.py:40-88): sample z,
map to W, convert to StyleSpace with W2S and write W.npy, S, S_1000 and s_stats = [transform, s_mean, s_std]."""
import pickle

import numpy as np
import torch


def compute_stats(G, random_state, num_images, truncation_psi=0.7, truncation_cutoff=None, batch=4096):
    device = next(G.parameters()).device
    z = np.random.RandomState(random_state).randn(num_images, G.z_dim)
    ws_first, all_s = [], {}
    with torch.no_grad():
        for b0 in range(0, num_images, batch):
            zb = torch.tensor(z[b0:b0 + batch]).to(device)
            ws = G.mapping(z=zb, c=None, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
            for layer, s in G.synthesis.W2S(ws).items():
                all_s.setdefault(layer, []).append(s.cpu().numpy())
            ws_first.append(ws[:, 0, :].cpu().numpy())
    all_s = {layer: np.concatenate(v) for layer, v in all_s.items()}
    s_mean = {layer: s.mean(axis=0) for layer, s in all_s.items()}
    s_std = {layer: s.std(axis=0) for layer, s in all_s.items()}
    ff = all_s['input']
    transform = {'theta': np.mean(np.arccos(ff[:, 0])), 'x': ff[:, 2].mean(), 'y': ff[:, 3].mean()}
    return np.concatenate(ws_first), all_s, [transform, s_mean, s_std]


def save_stats(G, random_state, num_images, truncation_psi, truncation_cutoff, output_path):
    w, all_s, s_stats = compute_stats(G, random_state, num_images, truncation_psi, truncation_cutoff)
    np.save(output_path / 'W', w)
    with open(output_path / 'S', "wb") as fp:
        pickle.dump(all_s, fp)
    with open(output_path / 'S_1000', "wb") as fp:
        pickle.dump({layer: s[:1000] for layer, s in all_s.items()}, fp)
    with open(output_path / 's_stats', "wb") as fp:
        pickle.dump(s_stats, fp)
