"""Mirror of /root/reference/utils.py (soft / rescale one-liners are on the hot path; save_image is host I/O)."""
import numpy as np
import torch


def soft(x):
    """utils.py:47-48"""
    return 0.96 * x + 0.02


def data_rescale(x):
    """utils.py:51-52"""
    return x / 127.5 - 1.0


def inverse_rescale(y):
    """utils.py:55-56 (round half to even, like tf.round)"""
    return torch.round((y + 1.0) * 127.5)


def save_image(image, path=None, shape=(None, None)):
    """utils.py:6-44: uint8 grid, images tiled column-major (index -> x = i % width rows, y = i // width cols)."""
    from PIL import Image
    image = inverse_rescale(image.detach().float().cpu()).clamp(0, 255).numpy().astype(np.uint8)
    if image.ndim == 4:
        width, height = shape
        if width is None and height is None:
            height = int(np.ceil(np.sqrt(image.shape[0])))
        if width is None:
            width = int(np.ceil(np.divide(image.shape[0], height)))
        if height is None:
            height = int(np.ceil(np.divide(image.shape[0], width)))
        iw, ih, ic = image.shape[1:4]
        combined = np.zeros((width * iw, height * ih, ic), np.uint8)
        for index, img in enumerate(image):
            y, x = index // width, index % width
            combined[x * iw:(x + 1) * iw, y * ih:(y + 1) * ih, :] = img
        image = combined
    if image.shape[2] == 1:
        pil = Image.fromarray(image.reshape(image.shape[0:2]), "L")
    else:
        pil = Image.fromarray(image, "RGB")
    if path is None:
        pil.show()
    else:
        pil.save(path)
