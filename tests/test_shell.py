"""Trainer shell around the hot path (SURVEY.md par. 8f-1): the image-grid writer, the fixed evaluation batch
test_data_<env>.npz, and the main.py modes that drive train / predict.  The grid test needs no GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_save_image_grid_is_column_major(tmp_path):
    """utils.py:6-44 (tiling at :28-31): image `index` goes to block row index % width, block column index // width,
    where width = ceil(n / height) and height = ceil(sqrt(n)) by default; pixel values are round((v + 1) * 127.5)."""
    from PIL import Image
    from littlegan_amd.utils import inverse_rescale, save_image
    n, s = 5, 2
    # image k is the constant grey level 10 * (k + 1) (PNG: lossless, so the file can be compared exactly)
    levels = [10 * (k + 1) for k in range(n)]
    img = torch.stack([torch.full((s, s, 3), lv / 127.5 - 1.0) for lv in levels])
    path = str(tmp_path / "grid.png")
    save_image(img, path)
    got = np.asarray(Image.open(path))
    height = int(np.ceil(np.sqrt(n)))            # 3
    width = int(np.ceil(n / height))             # 2
    assert got.shape == (width * s, height * s, 3)
    exp = np.zeros((width * s, height * s, 3), np.uint8)
    for k, lv in enumerate(levels):
        x, y = k % width, k // width
        exp[x * s:(x + 1) * s, y * s:(y + 1) * s] = lv
    assert np.array_equal(got, exp)
    # explicit shape (1, 8) as condition-sample uses it (main.py:122): one block row, 8 block columns
    img8 = torch.stack([torch.full((s, s, 3), (k + 1) / 127.5 - 1.0) for k in range(8)])
    save_image(img8, path, (1, 8))
    got = np.asarray(Image.open(path))
    assert got.shape == (s, 8 * s, 3) and [int(got[0, k * s, 0]) for k in range(8)] == list(range(1, 9))
    # single image (3-D) and the rounding rule (half to even, like tf.round)
    assert inverse_rescale(torch.tensor([-1.0, 1.0, 0.0])).tolist() == [0.0, 255.0, 128.0]   # 127.5 -> 128 (half to even)
    save_image(img[0], path)
    assert np.asarray(Image.open(path)).shape == (s, s, 3)


def _small_args(tmp_path, **kw):
    from types import SimpleNamespace
    d = dict(batch_size=4, image_channel=3, noise_dim=5, init_dim=2, conv_filter=[32, 32, 32, 32, 32], kernel_size=5,
             leaky_alpha=0.3, dropout_rate=0.5, l1_lambda=0.02, lr=5e-5, beta_1=0.5, beta_2=0.9, use_gp=False, use_clip=True,
             clip_range=0.5, use_partition=True, partition_interval=4, train_adj=True, attr=[1, 2, 3], cond_dim=3,
             mfma_dtype="f32", device="cuda", seed=3, synthetic=True, synthetic_images=16, image_dim=32, image_path=None,
             image_ext="jpg", attr_path=None, env="t", reuse=False, restore=False, no_io=False, exp_name="x", epoch=1,
             freq_gen=2, freq_test=2, test_data_dir=str(tmp_path / "td"), result_dir=str(tmp_path / "res"))
    d.update(kw)
    return SimpleNamespace(**d)


def _trainer(args):
    from littlegan_amd.dataset import CelebA
    from littlegan_amd.eager_trainer import EagerTrainer
    from littlegan_amd.model import Adjuster, Decoder, Discriminator, Encoder, Generator
    dec, enc = Decoder(args), Encoder(args)
    g = Generator(args, dec)
    d = Discriminator(args, enc)
    return EagerTrainer(args, g, d, Adjuster(args, d, g), CelebA(args))


@pytest.mark.gpu
def test_fixed_evaluation_batch_is_created_then_reused(tmp_path):
    """eager_trainer.py:65-83: first run draws {n, c, i} and writes test_data_<env>.npz; a run with reuse loads it."""
    tr = _trainer(_small_args(tmp_path))
    npz = tmp_path / "td" / "test_data_t.npz"
    assert npz.is_file()
    with np.load(npz) as f:
        data = {k: f[k] for k in ("n", "c", "i")}   # read everything before the file is rewritten below
    assert data["n"].shape == (4, 5) and data["c"].shape == (4, 3) and data["i"].shape == (4, 32, 32, 3)
    assert np.array_equal(data["n"], tr.test_noise.cpu().numpy())
    # poison the file's noise: a reuse run must take the FILE's content, a non-reuse run regenerates the batch
    np.savez_compressed(npz, n=data["n"] + 1.0, c=data["c"], i=data["i"])
    tr2 = _trainer(_small_args(tmp_path, reuse=True))
    assert np.array_equal(tr2.test_noise.cpu().numpy(), data["n"] + 1.0)
    assert np.array_equal(tr2.test_image.cpu().numpy(), data["i"])
    tr3 = _trainer(_small_args(tmp_path, reuse=False))
    assert np.array_equal(tr3.test_noise.cpu().numpy(), data["n"])
    # a file holding ANOTHER batch count is loaded as it is (the reference loads whatever the file holds, :69-76) ...
    np.savez_compressed(npz, n=data["n"][:3], c=data["c"][:3], i=data["i"][:3])
    tr4 = _trainer(_small_args(tmp_path, reuse=True))
    assert tr4.test_noise.shape[0] == 3 and np.array_equal(tr4.test_image.cpu().numpy(), data["i"][:3])
    # ... one whose shapes the models reject is regenerated (:77-83) and rewritten
    np.savez_compressed(npz, n=data["n"][:, :4], c=data["c"], i=data["i"])
    tr5 = _trainer(_small_args(tmp_path, reuse=True))
    assert np.array_equal(tr5.test_noise.cpu().numpy(), data["n"])
    with np.load(npz) as f:
        assert f["n"].shape == (4, 5)
    # the reference attribute names of SURVEY.md par. 8b exist and alias the trained weights
    D = tr.discriminator
    assert D.dense_pr.kernel.data_ptr() == D.weights[16].data_ptr() and D.dense_cond.bias.data_ptr() == D.weights[19].data_ptr()
    x = torch.randn(4, D.dense_pr.kernel.shape[0], device="cuda")
    assert torch.allclose(D.dense_cond(x), torch.sigmoid(x @ D.dense_cond.kernel + D.dense_cond.bias), atol=1e-5)


@pytest.mark.gpu
def test_main_cli_modes(tmp_path):
    """main.py:41-81,105-130 on a tiny configuration: train (2 epochs, image dumps, predict JSON, checkpoints), then
    random-sample, condition-sample, evaluate-sample and export-model restore that run and write their outputs."""
    cfgdir = tmp_path / "cfg"
    cfgdir.mkdir()
    res = tmp_path / "results"
    (cfgdir / "sample.config.json").write_text(json.dumps({"synthetic": True}))
    (cfgdir / "t.config.json").write_text(json.dumps({
        "synthetic": True, "synthetic_images": 96, "all_result_dir": str(res), "test_data_dir": str(tmp_path / "td"),
        "batch_size": 4, "epoch": 2, "freq_gen": 2, "freq_test": 4, "mfma_dtype": "bf16", "train_adj": True, "image_dim": 32,
        "init_dim": 2, "conv_filter": [64, 32, 32, 32, 32], "noise_dim": 7, "random_sample_batch": 2,
        "condition_sample_batch": 2, "evaluate_sample_size": 8, "restore": True}))
    env = dict(os.environ, LITTLEGAN_CONFIG_DIR=str(cfgdir))

    def run(mode):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), mode, "exp", "-e", "t", "--debug"], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (mode, r.stdout[-2000:], r.stderr[-2000:])
        return r.stdout

    out = run("train")
    rd = res / "exp"
    assert "Epoch: 2" in out and "LossG" in out
    assert (rd / "checkpoint" / "ckpt-2.pt").is_file() and (rd / "train" / "gen" / "1-2.jpg").is_file()
    assert (rd / "test" / "disc" / "1-4.json").is_file() and (rd / "test" / "adj" / "1-4.jpg").is_file()
    # 96 images / batch 4 = 24 batches = 12 steps per epoch; the Adjuster branch runs from step 11 (eager_trainer.py:152)
    assert (rd / "train" / "adj" / "2-12.jpg").is_file() and not (rd / "train" / "adj" / "2-10.jpg").exists()
    js = json.load(open(rd / "test" / "disc" / "1-4.json"))
    assert set(js) >= {"real_pr_mse", "real_c_mse", "fake_pr_mse", "fake_c_mse", "real_cond", "real_pr"}
    run("random-sample")
    samples = os.listdir(rd / "sample")
    assert sum(f.startswith("generator-") for f in samples) == 2 and sum(f.startswith("input_data-") for f in samples) == 2
    run("condition-sample")
    assert (rd / "sample" / "condition-gen-2.jpg").is_file()
    from PIL import Image
    assert Image.open(rd / "sample" / "condition-gen-1.jpg").size == (8 * 32, 32)   # (1, 8) grid: one block row
    run("evaluate-sample")
    assert (rd / "evaluate" / "gen" / "8.jpg").is_file() and (rd / "evaluate" / "adj" / "real_1.jpg").is_file()
    assert "exported" in run("export-model") and (rd / "model" / "model.pt").is_file()


def test_bench_finds_its_committed_traffic_file():
    """bench.py reads the dominant kernel's HBM traffic from the newest committed counter collection; a file name it cannot parse must
    not break the bench line (round 5: a second collection of a round, r5b_, did)."""
    import json
    import bench
    files = bench.traffic_files()
    assert files and all(os.path.basename(f).startswith("r") for f in files)
    rounds = [int("".join(ch for ch in os.path.basename(f)[1:].split("_")[0] if ch.isdigit())) for f in files]
    assert rounds == sorted(rounds)
    assert isinstance(json.load(open(files[-1])), dict)


def test_bench_precision_note_quotes_the_other_dtype_of_the_same_workload():
    """the bf16 headline and the exact-f32 step that meets the north star's tolerance are quoted side by side in
    the bench line (`precision`); the figure comes from the newest committed collection whose own `dtype` field is the other one."""
    import bench
    n = bench.precision_note("c3", "bf16")
    assert "1e-4" in n["this_line"] and n["other_dtype_same_workload"]["dtype"] == "f32"
    assert n["other_dtype_same_workload"]["source"].startswith("profiles/") and n["other_dtype_same_workload"]["ms_per_step"] > 30
    assert bench.precision_note("c3", "f32")["other_dtype_same_workload"]["dtype"] == "bf16"
    assert bench.precision_note("c2", "f32")["other_dtype_same_workload"] is None   # c2's plain file IS the f32 run
