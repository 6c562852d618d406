"""Input-transform helpers used by the inversion path (reference utils/common.py:9-45; tensor2im / mp4 writing need PIL /
imageio and belong to the out-of-scope I/O layer).

A transform is the 3x3 matrix that `SynthesisInput` receives in `synthesis.input.transform`: the INVERSE of "rotate by
`angle` degrees, then shift by `translate`" in the generator's [-1, 1] canvas units."""
import numpy as np


def make_transform(translate, angle):
    """Forward matrix [[c, s, tx], [-s, c, ty], [0, 0, 1]] for `angle` in degrees (expression order of the reference, so
    that the float64 result is bit-identical)."""
    theta = angle / 360.0 * np.pi * 2
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[c, s, translate[0]],
                     [-s, c, translate[1]],
                     [0.0, 0.0, 1.0]])


def get_identity_transform():
    return np.linalg.inv(make_transform((0, 0), 0.))


def generate_random_transform(translate=0.3, rotate=25):
    """Uniform angle in +-rotate degrees and uniform shift in +-translate, drawn from numpy's global stream in the
    reference's order (angle, x, y)."""
    angle = np.random.uniform(-rotate, rotate)
    shift = np.random.uniform(-translate, translate, size=2)
    return np.linalg.inv(make_transform(shift, angle))
