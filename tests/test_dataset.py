"""CelebA loader mirror (dataset.py:7-49): header-less attribute file, attribute filter, `soft` on the labels,
`data_rescale` on the pixels, floor(n / batch_size) batches, OutOfRange at the end; and the synthetic stand-in."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from PIL import Image

from littlegan_amd.dataset import CelebA


def _args(tmp_path, **kw):
    d = dict(image_path=str(tmp_path / "img"), attr_path=str(tmp_path / "attr.txt"), image_ext="png", image_dim=8,
             image_channel=3, attr=[0, 2], batch_size=2, device="cpu", seed=0, synthetic=False)
    d.update(kw)
    return SimpleNamespace(**d)


def test_file_backed_batches(tmp_path):
    (tmp_path / "img").mkdir()
    rng = np.random.default_rng(0)
    pix, rows = {}, []
    for i in range(5):
        a = rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)
        name = f"{i:03d}.png"
        Image.fromarray(a, "RGB").save(tmp_path / "img" / name)
        pix[name] = a
        rows.append(f"{name} {(-1) ** i} 1 {(-1) ** (i + 1)}")
    (tmp_path / "attr.txt").write_text("\n".join(rows) + "\n")
    ds = CelebA(_args(tmp_path))
    assert not ds.synthetic and ds.batches == 2 and ds.label == [0, 2]
    it = ds.get_new_iterator()
    seen = 0
    for _ in range(ds.batches):
        img, cond = it.get_next()
        assert img.shape == (2, 8, 8, 3) and cond.shape == (2, 2) and img.dtype == torch.float32
        assert float(img.min()) >= -1.0 and float(img.max()) <= 1.0
        assert set(np.round(cond.double().numpy().ravel(), 4).tolist()) <= {0.98, -0.94}      # soft(+1), soft(-1): dataset.py:33
        seen += 1
    with pytest.raises(StopIteration):
        it.get_next()
    assert seen == 2


def test_synthetic_stand_in_is_deterministic(tmp_path):
    a = _args(tmp_path, synthetic=True, synthetic_images=8, attr=[1, 2, 3])
    d1, d2 = CelebA(a), CelebA(a)
    assert d1.synthetic and d1.batches == 4
    b1, b2 = d1.get_new_iterator().get_next(), d2.get_new_iterator().get_next()
    assert torch.equal(b1[0], b2[0]) and torch.equal(b1[1], b2[1]) and b1[1].shape == (2, 3)
