"""Transform helpers of the reference's utils/common.py:9-45 that the inversion path uses (tensor2im / mp4 writing need
PIL / imageio and belong to the out-of-scope I/O layer)."""
import numpy as np


def make_transform(translate, angle):
    m = np.eye(3)
    s = np.sin(angle / 360.0 * np.pi * 2)
    c = np.cos(angle / 360.0 * np.pi * 2)
    m[0][0], m[0][1], m[0][2] = c, s, translate[0]
    m[1][0], m[1][1], m[1][2] = -s, c, translate[1]
    return m


def get_identity_transform():
    return np.linalg.inv(make_transform((0, 0), 0.))


def generate_random_transform(translate=0.3, rotate=25):
    rotate = np.random.uniform(low=-1 * rotate, high=rotate)
    translate = (np.random.uniform(low=-1 * translate, high=translate), np.random.uniform(low=-1 * translate, high=translate))
    return np.linalg.inv(make_transform(translate, rotate))
