"""Every kernel call of one training step, replayed against the oracle on the call's OWN inputs (tests/replay.py).

This is the tight check of the bf16 (headline C3) path: a whole-step comparison in bf16 cannot be tighter than ~1e-2
(rounding flips decorrelate two implementations within three layers, see tests/replay.py), but each call, fed with what
the kernels actually produced upstream, is held to <= 6e-4 rms where its result is stored as bf16 (only flipped roundings
remain) and <= 2e-5 rms where it is an fp32 result of bf16 operands (weight gradients, image gradients).  Reference
channel widths at 64x64 (C1 shape) and 128x128 (the C2 / C3 geometry), both dtypes, Adjuster branch on; a partition step
at the small shape.  Together with test_step_gpu (wiring: whole-step values against the oracle) this covers every
tensor the step produces."""
import numpy as np
import pytest
import torch

from oracle import np_oracle as O
from replay import OpRecorder, check_call
from test_step_gpu import build, dev_inputs, f32_round, perturbed

pytestmark = pytest.mark.gpu


def _replay(cfg, mfma, b, seed=7):
    W = perturbed(cfg, seed)
    tr = build(cfg, W, mfma)
    inp = dev_inputs(f32_round(O.make_inputs(cfg, cfg.batch_size, seed=9)))
    with OpRecorder() as r:
        tr.train_step_from_inputs(b, inp)
        torch.cuda.synchronize()
    seen = {}
    for rec in r.calls:
        tag = check_call(rec, r.packs)
        seen[rec["name"]] = seen.get(rec["name"], 0) + 1
    return seen


@pytest.mark.parametrize("mfma", ["bf16", "f32"])
@pytest.mark.parametrize("init_dim", [4, 8])
def test_every_call_of_a_full_step_matches_the_oracle(mfma, init_dim):
    seen = _replay(O.Cfg(init_dim=init_dim, cond_dim=40, batch_size=2), mfma, 11)
    # G fwd (4 convT) + Adjuster fwd (4 convT) ; D fwd on [real;fake], A's encoder on img1, D fwd on adj (3 x 4 conv) — in the bf16
    # path D on adj hands raw maps to the normalising conv where its tiling covers the level (64 x 64 images: level 2; 128 x 128: 2, 3)
    zn = seen.get("conv2d_s2_fwd_stats_zn", 0)
    assert zn == ((1 if init_dim == 4 else 2) if mfma == "bf16" else 0)
    assert seen["convT_s2_fwd_stats"] == 8 and seen["conv2d_s2_fwd_stats"] + zn == 12
    # disc tape 4 + gen tape 4 + adj tape 4 encoder levels, G tape 4 + adj tape 4 decoder levels, 2 dense norms — in the bf16 path the
    # Adjuster's decoder level 4 (no weight gradient asked: eager_trainer.py:163) leaves the norm-backward apply to the data-gradient
    # conv below it (lg_convT_s2_dgrad_bn, the 64-column level): a coefficient launch instead of instnorm_bwd, dgrad_bn instead of dgrad
    bn = seen.get("convT_s2_dgrad_bn", 0)
    assert bn == (1 if mfma == "bf16" else 0) and seen.get("instnorm_bwd_coef", 0) == bn
    assert seen["instnorm_bwd"] == 22 - bn
    assert seen["conv2d_s2_wgrad"] == 4 and seen["convT_s2_wgrad"] == 4 and seen["convT_s1_tanh_bwd"] == 2
    assert seen["conv2d_s2_dgrad"] == 3 + 4 + 4 and seen["convT_s2_dgrad"] + bn == 4 + 4


@pytest.mark.parametrize("mfma", ["bf16", "f32"])
def test_every_call_of_partition_and_plain_steps_small(mfma):
    cfg = O.Cfg(init_dim=2, conv_filter=(64, 32, 32, 64, 32), cond_dim=5, noise_dim=11, batch_size=3)
    for b in (4, 5, 10, 15):   # plain step without the Adjuster branch; the three partition groups
        _replay(cfg, mfma, b, seed=b)
