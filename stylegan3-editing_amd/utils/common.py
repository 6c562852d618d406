"""Input-transform helpers used by the inversion path and the two image helpers its callers use (reference utils/common.py:9-55).
`tensor2im` needs PIL and `generate_mp4` imageio; both are imported when called (imageio is not installed in this image).

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


def tensor2im(var):
    """CHW tensor in [-1, 1] -> PIL image, values truncated to uint8 (reference :39-45)."""
    from PIL import Image
    arr = var.cpu().detach().transpose(0, 2).transpose(0, 1).numpy()
    arr = np.clip((arr + 1) / 2, 0, 1) * 255
    return Image.fromarray(arr.astype('uint8'))


def generate_mp4(out_name, images, kwargs):
    """Write frames to `<out_name>.mp4` through imageio (reference :48-52)."""
    try:
        import imageio
    except ImportError as err:
        raise RuntimeError('generate_mp4 needs the imageio package (video writing is outside this package)') from err
    writer = imageio.get_writer(str(out_name) + '.mp4', **kwargs)
    for image in images:
        writer.append_data(np.array(image))
    writer.close()
