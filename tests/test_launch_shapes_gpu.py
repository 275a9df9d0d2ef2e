"""Per-op parity AT THE LAUNCH SHAPES bench.py times: batch 256 in the bf16 configuration (C3) and batch 64 in the exact-f32
configuration (C2), 128x128 images, reference channel widths.  Everything that depends on the batch — grid size and the
XCD block remap, several samples per tile on the small maps, the split-K factor and the ordered slab reduce of the weight
gradients, the moment-partials workspace, the column-sum grid of the norm backward — only takes its benchmarked value
here (the step tests run B = 2..3).

Against the fp64 numpy oracle: forward / data-gradient convs and both norm passes on SAMPLED images (all of them are
per-sample ops), weight gradients on the FULL batch.  Full-size properties on top: the B-image result equals, bit for
bit, the same kernel run on 32-image chunks (no cross-image contamination at the big grid), and the reductions over the
batch (weight / bias / gamma / beta gradients) equal the sum of their per-chunk values."""
import numpy as np
import pytest
import torch

from oracle import np_oracle as O

pytestmark = pytest.mark.gpu

ALPHA = 0.3
# (name, kind, cb, cs, small-map side at 128x128)   conv: x[B,2s,2s,cb] -> z[B,s,s,cs] ; convT: x[B,s,s,cs] -> z[B,2s,2s,cb]
LAYERS = [("enc.conv1", "conv", 3, 64, 64), ("enc.conv2", "conv", 64, 128, 32), ("enc.conv3", "conv", 128, 256, 16),
          ("enc.conv4", "conv", 256, 384, 8), ("dec.conv1", "convT", 256, 384, 8), ("dec.conv2", "convT", 128, 256, 16),
          ("dec.conv3", "convT", 64, 128, 32), ("dec.conv4", "convT", 32, 64, 64)]
# B = 512: the C3 step runs D on [new_image ; fake] and the whole Adjuster branch on [img1 ; fake] at 2 x 256 images
# (littlegan_amd/eager_trainer.py: train_step_from_inputs; reference eager_trainer.py:134-137,152-163)
CONFIGS = [pytest.param(1, 256, id="bf16-B256"), pytest.param(1, 512, id="bf16-B512"), pytest.param(0, 64, id="f32-B64")]
CHUNK = 32


@pytest.fixture(scope="module")
def ops():
    from littlegan_amd import ops as _ops
    return _ops


def _rand(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, generator=g, device="cuda", dtype=torch.float32) * scale).to(dtype)


def _f64(t):
    return t.detach().double().cpu().numpy()


def _rms(got, exp):
    got, exp = np.asarray(got, np.float64), np.asarray(exp, np.float64)
    return float(np.sqrt(((got - exp) ** 2).mean()) / (np.sqrt((exp ** 2).mean()) + 1e-30))


def _samples(B):
    return sorted({0, 1, B // 2 - 1, B // 2, B - 1, (B * 5) // 8 + 3})


def _q(x, dt):
    return O.bf16_round(x) if dt == 1 else x


@pytest.mark.parametrize("dt,B", CONFIGS)
@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
def test_forward_with_fused_moments(ops, layer, dt, B):
    name, kind, cb, cs, s = layer
    w = _rand((5, 5, cb, cs), 1, 0.05)
    cout = cs if kind == "conv" else cb
    bias, gm, bt = _rand((cout,), 2, 0.1), torch.tensor([1.3], device="cuda"), torch.tensor([-0.2], device="cuda")
    pack = ops.conv_pack(w, cb, cs, dt)
    xs = (B, 2 * s, 2 * s, cb) if kind == "conv" else (B, s, s, cs)
    x = _rand(xs, 3)
    x16 = x.to(torch.bfloat16) if (dt == 1 and cb != 3 or dt == 1 and kind == "convT") else None
    fwd = ops.conv2d_s2_fwd_stats if kind == "conv" else ops.convT_s2_fwd_stats

    def run(lo, hi):
        xa = x[lo:hi] if (x16 is None or kind == "conv" and cb == 3) else None
        z, st = fwd(xa if xa is not None else None, pack, bias, cout, dt, gm, bt, x16=None if x16 is None else x16[lo:hi],
                    z16=(dt == 1), alpha=ALPHA)
        if st is None:
            st = ops.instnorm_stats(z, gm, bt, 0, ALPHA)
        return z, st

    z, st = run(0, B)
    assert z.dtype == (torch.bfloat16 if dt == 1 else torch.float32)
    # oracle on sampled images
    idx = _samples(B)
    xq = _q(_f64(x[idx]), dt)
    wq = _q(_f64(w), dt)
    ref = O.conv2d(xq, wq, _f64(bias), 2) if kind == "conv" else O.conv2d_transpose(xq, wq, _f64(bias), 2)
    got = _f64(z[idx])
    if dt == 1:
        assert _rms(got, O.bf16_round(ref)) < 6e-4
    else:
        assert _rms(got, ref) < 1e-5 and np.abs(got - ref).max() < 1e-4 * np.abs(ref).max()
    mu = ref.reshape(len(idx), -1).mean(1)
    sg = ref.reshape(len(idx), -1).std(1)
    s_ = _f64(st[idx])
    assert np.abs(s_[:, 0] + s_[:, 4] - mu).max() < 3e-6 * max(np.abs(mu).max(), sg.max())
    assert np.abs(s_[:, 1] - sg).max() < 3e-6 * sg.max()
    assert np.abs(s_[:, 2] - 1.3 / (sg + 1e-3)).max() < 3e-6 * (1.3 / sg.min())
    # the big launch == the same kernel on 32-image chunks, bit for bit
    for lo in range(0, B, CHUNK):
        zc, stc = run(lo, lo + CHUNK)
        assert torch.equal(zc, z[lo:lo + CHUNK]), (name, lo)
        assert torch.equal(stc, st[lo:lo + CHUNK]), (name, lo, "stats")


@pytest.mark.parametrize("dt,B", CONFIGS)
@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
def test_data_gradient(ops, layer, dt, B):
    name, kind, cb, cs, s = layer
    w = _rand((5, 5, cb, cs), 11, 0.05)
    pack = ops.conv_pack(w, cb, cs, dt)
    gs = (B, s, s, cs) if kind == "conv" else (B, 2 * s, 2 * s, cb)     # gradient w.r.t. the layer OUTPUT
    dz = _rand(gs, 12)
    dz16 = dz.to(torch.bfloat16) if dt == 1 else None
    first_level = name in ("dec.conv1", "enc.conv1")   # the gradient that leaves the stack is fp32, the others are bf16 in the bf16 path
    obf = dt == 1 and not first_level

    def run(lo, hi):
        d32 = None if dt == 1 else dz[lo:hi]
        d16 = None if dz16 is None else dz16[lo:hi]
        if kind == "conv":
            return ops.conv2d_s2_dgrad(d32, pack, cb, dt, dy16=d16, out_bf16=obf)
        return ops.convT_s2_dgrad(d32, pack, cs, dt, dy16=d16, out_bf16=obf)

    g = run(0, B)
    idx = _samples(B)
    dq, wq = _q(_f64(dz[idx]), dt), _q(_f64(w), dt)
    ref = O.conv_bwd_input(dq, wq, 2, (2 * s, 2 * s)) if kind == "conv" else O.conv_fwd(dq, wq, 2)
    got = _f64(g[idx])
    if obf:
        assert _rms(got, O.bf16_round(ref)) < 6e-4
    else:
        assert _rms(got, ref) < 2e-5
    for lo in range(0, B, CHUNK):
        assert torch.equal(run(lo, lo + CHUNK), g[lo:lo + CHUNK]), (name, lo)


@pytest.mark.parametrize("dt,B", CONFIGS)
@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
def test_weight_gradient_full_batch(ops, layer, dt, B):
    """split-K over the whole batch + ordered slab reduce, against the oracle on the FULL batch"""
    name, kind, cb, cs, s = layer
    if B == 512 and kind == "convT":
        pytest.skip("no tape of the step takes decoder weight gradients at 2B (the Adjuster trains its dense + norm only)")
    big = _rand((B, 2 * s, 2 * s, cb), 21)
    small = _rand((B, s, s, cs), 22, 0.1)
    b16 = big.to(torch.bfloat16) if (dt == 1 and cb != 3) else None
    s16 = small.to(torch.bfloat16) if dt == 1 else None
    dw = torch.full((5, 5, cb, cs), 3.0, device="cuda")

    def run(lo, hi, out):
        if kind == "conv":   # x = big, dy = small
            ops.conv2d_s2_wgrad(big[lo:hi] if (dt == 0 or cb == 3) else None, small[lo:hi] if dt == 0 else None, out, False, dt,
                                x16=None if b16 is None else b16[lo:hi], dy16=None if s16 is None else s16[lo:hi])
        else:                # x = small, dy = big
            ops.convT_s2_wgrad(small[lo:hi] if dt == 0 else None, big[lo:hi] if dt == 0 else None, out, False, dt,
                               x16=None if s16 is None else s16[lo:hi], dy16=None if b16 is None else b16[lo:hi])
        return out

    run(0, B, dw)
    ref = O.conv_bwd_filter(_q(_f64(big), dt), _q(_f64(small), dt), 2, 5)
    assert _rms(_f64(dw), ref) < (3e-5 if dt == 1 else 2e-5), name
    assert np.abs(_f64(dw) - ref).max() < 3e-4 * np.abs(ref).max()
    acc = torch.zeros_like(dw, dtype=torch.float64)
    tmp = torch.empty_like(dw)
    for lo in range(0, B, CHUNK):
        acc += run(lo, lo + CHUNK, tmp).double()
    assert _rms(_f64(dw), acc.cpu().numpy()) < 2e-5, (name, "sum of chunks")


@pytest.mark.parametrize("dt,B", CONFIGS)
@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
def test_norm_apply_and_backward(ops, layer, dt, B):
    """InstanceNorm + LeakyReLU (+ skip) forward pass and the backward with the fused bias column sums, on the layer's
    output map at the benchmarked batch."""
    name, kind, cb, cs, s = layer
    C = cs if kind == "conv" else cb
    side = s if kind == "conv" else 2 * s
    shape = (B, side, side, C)
    zt = torch.bfloat16 if dt == 1 else torch.float32
    z = _rand(shape, 31, 1.7).add_(0.4).to(zt)
    gm, bt = torch.tensor([0.9], device="cuda"), torch.tensor([0.15], device="cuda")
    zf = z.float()
    st = ops.instnorm_stats(zf, gm, bt, 0, ALPHA)
    skip = _rand(shape, 32, 0.5).to(zt) if kind == "convT" and name != "dec.conv4" else None
    h16 = torch.empty(shape, dtype=torch.bfloat16, device="cuda") if dt == 1 else None
    h = ops.instnorm_apply(z, st, skip, 0, 1, ALPHA, out16=h16, want_f32=(dt == 0))
    g = _rand(shape, 33).to(zt)
    dgm, dbt, db = torch.empty(1, device="cuda"), torch.empty(1, device="cuda"), torch.empty(C, device="cuda")
    d16 = torch.empty(shape, dtype=torch.bfloat16, device="cuda") if dt == 1 else None
    dz = ops.instnorm_bwd(z, st, g, dgm, dbt, 0, 1, ALPHA, out16=d16, want_f32=(dt == 0), db=db)
    # oracle on sampled images, from the kernel's own statistics record (its parity is test_forward_with_fused_moments)
    idx = _samples(B)
    zs, gs, ss = _f64(z[idx]).reshape(len(idx), -1), _f64(g[idx]).reshape(len(idx), -1), _f64(st[idx])
    mu, sigma, a, b = (ss[:, 0] + ss[:, 4])[:, None], ss[:, 1][:, None], ss[:, 2][:, None], ss[:, 3][:, None]
    c = zs - mu
    y = a * c + b
    href = O.leaky(y, ALPHA) + (_f64(skip[idx]).reshape(len(idx), -1) if skip is not None else 0.0)
    s32 = ss.astype(np.float32)
    y32 = (s32[:, 2:3] * ((zs.astype(np.float32) - s32[:, 0:1]) - s32[:, 4:5])).astype(np.float32) + s32[:, 3:4]
    gp = np.where(y32 > 0, gs, ALPHA * gs)
    se = sigma + 1e-3
    dref = a * (gp - gp.mean(1, keepdims=True) - c * (gp * c).mean(1, keepdims=True) / (se * sigma))
    if dt == 1:
        assert _rms(_f64(h16[idx]).reshape(len(idx), -1), O.bf16_round(href)) < 6e-4
        assert _rms(_f64(d16[idx]).reshape(len(idx), -1), O.bf16_round(dref)) < 6e-4
    else:
        assert _rms(_f64(h[idx]).reshape(len(idx), -1), href) < 3e-6
        assert _rms(_f64(dz[idx]).reshape(len(idx), -1), dref) < 2e-5
    # batch reductions == sum over 32-image chunks of the same kernel (each chunk is its own launch geometry)
    sg, sb, sdb = 0.0, 0.0, torch.zeros(C, dtype=torch.float64, device="cuda")
    for lo in range(0, B, CHUNK):
        cg, cb_, cdb = torch.empty(1, device="cuda"), torch.empty(1, device="cuda"), torch.empty(C, device="cuda")
        o16 = torch.empty((CHUNK,) + shape[1:], dtype=torch.bfloat16, device="cuda") if dt == 1 else None
        dc = ops.instnorm_bwd(z[lo:lo + CHUNK], st[lo:lo + CHUNK], g[lo:lo + CHUNK], cg, cb_, 0, 1, ALPHA, out16=o16,
                              want_f32=(dt == 0), db=cdb)
        # (the per-sample sums are split over a batch-dependent number of blocks: equal up to the last bits of m1 / m2,
        #  i.e. up to a few flipped bf16 roundings)
        assert _rms(_f64(o16 if dt == 1 else dc), _f64((d16 if dt == 1 else dz)[lo:lo + CHUNK])) < (2e-4 if dt == 1 else 1e-6), (name, lo)
        sg, sb, sdb = sg + float(cg), sb + float(cb_), sdb + cdb.double()
    gabs = float(g.float().abs().sum())
    assert abs(float(dgm) - sg) <= 1e-6 * gabs and abs(float(dbt) - sb) <= 1e-6 * gabs
    assert _rms(_f64(db), sdb.cpu().numpy()) < 1e-5
    # chunk 0 against the oracle: the per-sample sums and the column sums themselves
    idx0 = list(range(4))
    z0, g0, s0 = _f64(z[idx0]).reshape(4, -1), _f64(g[idx0]).reshape(4, -1), _f64(st[idx0])
    cg, cb_, cdb = torch.empty(1, device="cuda"), torch.empty(1, device="cuda"), torch.empty(C, device="cuda")
    o16 = torch.empty((4,) + shape[1:], dtype=torch.bfloat16, device="cuda") if dt == 1 else None
    ops.instnorm_bwd(z[:4], st[:4], g[:4], cg, cb_, 0, 1, ALPHA, out16=o16, want_f32=(dt == 0), db=cdb)
    mu0, sg0, a0, b0 = (s0[:, 0] + s0[:, 4])[:, None], s0[:, 1][:, None], s0[:, 2][:, None], s0[:, 3][:, None]
    c0 = z0 - mu0
    s032 = s0.astype(np.float32)
    y032 = (s032[:, 2:3] * ((z0.astype(np.float32) - s032[:, 0:1]) - s032[:, 4:5])).astype(np.float32) + s032[:, 3:4]
    gp0 = np.where(y032 > 0, g0, ALPHA * g0)
    d0 = a0 * (gp0 - gp0.mean(1, keepdims=True) - c0 * (gp0 * c0).mean(1, keepdims=True) / ((sg0 + 1e-3) * sg0))
    assert abs(float(cg) - float((gp0 * c0 / (sg0 + 1e-3)).sum())) <= 3e-6 * float((np.abs(gp0 * c0) / (sg0 + 1e-3)).sum())
    assert abs(float(cb_) - float(gp0.sum())) <= 3e-6 * float(np.abs(gp0).sum())
    dbr = d0.reshape(-1, C).sum(0)
    assert np.abs(_f64(cdb) - dbr).max() <= 3e-6 * np.abs(d0.reshape(-1, C)).sum(0).max()


@pytest.mark.parametrize("dt,B", CONFIGS)
def test_final_layer(ops, dt, B):
    """Conv2DTranspose(3, 5, 1, same, tanh) (model.py:86-87) forward, data gradient and weight / bias gradients at 128x128."""
    H, cs = 128, 32
    w = _rand((5, 5, 3, cs), 41, 0.05)
    bias = _rand((3,), 42, 0.1)
    pack = ops.conv_pack(w, 3, cs, dt)
    x = _rand((B, H, H, cs), 43)
    x16 = x.to(torch.bfloat16) if dt == 1 else None
    y = ops.convT_s1_tanh_fwd(None if dt == 1 else x, pack, bias, 3, dt, x16=x16)
    dpre = _rand((B, H, H, 3), 44, 0.01)
    dx = torch.empty((B, H, H, cs), dtype=torch.bfloat16 if dt == 1 else torch.float32, device="cuda")
    dw, db = torch.empty(5, 5, 3, cs, device="cuda"), torch.empty(3, device="cuda")
    ops.convT_s1_tanh_bwd(None if dt == 1 else x, dpre, pack, cs, dt, dx=None if dt == 1 else dx, dx16=dx if dt == 1 else None,
                          dw=dw, db=db, x16=x16)
    idx = _samples(B)
    xq, wq, dq = _q(_f64(x[idx]), dt), _q(_f64(w), dt), _q(_f64(dpre[idx]), dt)
    assert _rms(_f64(y[idx]), np.tanh(O.conv2d_transpose(xq, wq, _f64(bias), 1))) < 2e-5
    dref = O.conv_fwd(dq, wq, 1)
    assert _rms(_f64(dx[idx]), O.bf16_round(dref) if dt == 1 else dref) < (6e-4 if dt == 1 else 2e-5)
    ref_w = O.conv_bwd_filter(_q(_f64(dpre), dt), _q(_f64(x), dt), 1, 5)
    assert _rms(_f64(dw), ref_w) < 3e-5
    dp = _f64(dpre)
    assert np.abs(_f64(db) - dp.sum((0, 1, 2))).max() <= 3e-6 * np.abs(dp).sum((0, 1, 2)).max()
    for lo in range(0, B, CHUNK):
        yc = ops.convT_s1_tanh_fwd(None if dt == 1 else x[lo:lo + CHUNK], pack, bias, 3, dt, x16=None if x16 is None else x16[lo:lo + CHUNK])
        assert torch.equal(yc, y[lo:lo + CHUNK])


def test_whole_step_forward_rows_at_batch_256():
    """The bench configuration itself (C3: 128x128, B = 256, bf16, reference widths): generated images of sampled rows of
    the batch against the bf16-emulating oracle (the forward is per-sample), and D's head probabilities on those rows (forward only;
    the full step at this batch is test_whole_step_at_launch_batch below)."""
    from test_step_gpu import build, dev_inputs, f32_round, perturbed
    cfg = O.Cfg(init_dim=8, cond_dim=40, batch_size=256)
    W = perturbed(cfg, 7)
    tr = build(cfg, W, "bf16")
    rng = np.random.default_rng(5)
    noise = rng.standard_normal((256, cfg.noise_dim)).astype(np.float32)
    cond = O.soft(2.0 * rng.integers(0, 2, (256, cfg.cond_dim)) - 1.0).astype(np.float32)
    img = tr.generator([torch.tensor(noise, device="cuda"), torch.tensor(cond, device="cuda")])
    idx = _samples(256)
    cfg_e = O.Cfg(init_dim=8, cond_dim=40, batch_size=len(idx), emulate_bf16=True)
    ref, _ = O.generator_fwd(cfg_e, W["G"], noise[idx].astype(np.float64), cond[idx].astype(np.float64))
    got = _f64(img[idx])
    assert np.abs(got - ref).max() < 2e-2 and _rms(got, ref) < 3e-3
    p = tr.discriminator.forward_packed(img)
    (pr, pc), _ = O.discriminator_fwd(cfg_e, W["D"], got)   # D on the kernel's own images (see tests/replay.py)
    gp = _f64(p[idx])
    assert np.abs(gp[:, :1] - pr).max() < 5e-3 and np.abs(gp[:, 1:] - pc).max() < 5e-3


# ---- data gradients in the FUSED form the step launches (lg_*_dgrad_nf + lg_instnorm_leaky_bwd_z16_p) -----------------------
# (name, kind of the layer whose data gradient runs, cb, cs, small side s, sums fused at these shapes)
#   conv  : dy16 [B,s,s,cs]   -> g [B,2s,2s,cb] = gradient of the encoder level below; UP contraction (conv_up3 / conv_up4 / conv_halo)
#   convT : dy16 [B,2s,2s,cb] -> g [B,s,s,cs]   = gradient of the decoder level below; DOWN contraction (conv_down3)
FUSED = [("enc.conv2", "conv", 64, 128, 32, True), ("enc.conv3", "conv", 128, 256, 16, True), ("enc.conv4", "conv", 256, 384, 8, True),
         ("dec.conv2", "convT", 128, 256, 16, True), ("dec.conv3", "convT", 64, 128, 32, True), ("dec.conv4", "convT", 32, 64, 64, True)]


def _norm_ref(zs, gs, ss):
    """oracle of the InstanceNorm + LeakyReLU backward on sampled images from the kernel's statistics records (LeakyReLU mask in
    the kernels' fp32 order, as tests/replay.py does)"""
    mu, sigma, a = (ss[:, 0] + ss[:, 4])[:, None], ss[:, 1][:, None], ss[:, 2][:, None]
    c = zs - mu
    s32 = ss.astype(np.float32)
    y32 = (s32[:, 2:3] * ((zs.astype(np.float32) - s32[:, 0:1]) - s32[:, 4:5])).astype(np.float32) + s32[:, 3:4]
    gp = np.where(y32 > 0, gs, ALPHA * gs)
    return a * (gp - gp.mean(1, keepdims=True) - c * (gp * c).mean(1, keepdims=True) / ((sigma + 1e-3) * sigma))


@pytest.mark.parametrize("B", [256, 512])
@pytest.mark.parametrize("layer", FUSED, ids=[l[0] for l in FUSED])
def test_fused_data_gradient_and_norm_backward(ops, layer, B):
    """The bf16 step never calls the plain data gradient on these layers: it calls lg_conv2d_s2_dgrad_nf / lg_convT_s2_dgrad_nf
    (the conv's store loop also adds up sum g', sum g'c of the norm backward below, [B][nparts][2] doubles in a workspace that
    scales with the batch) and then lg_instnorm_leaky_bwd_z16_p.  At B = 256 (gen tape, rows of the fake half) and B = 512 (disc
    tape, Adjuster branch): gradient and dz against the oracle on sampled images, the fused sums against the stand-alone
    first pass, chunk equality, and WHICH shapes really fuse."""
    name, kind, cb, cs, s, expect = layer
    w = _rand((5, 5, cb, cs), 51, 0.05)
    pack = ops.conv_pack(w, cb, cs, 1)
    dy = _rand((B, s, s, cs) if kind == "conv" else (B, 2 * s, 2 * s, cb), 52)
    dy16 = dy.to(torch.bfloat16)
    gshape = (B, 2 * s, 2 * s, cb) if kind == "conv" else (B, s, s, cs)
    z16 = _rand(gshape, 53, 1.7).add_(0.4).to(torch.bfloat16)      # raw conv output of the level the gradient belongs to
    gm, bt = torch.tensor([0.9], device="cuda"), torch.tensor([0.15], device="cuda")
    st = ops.instnorm_stats(z16.float(), gm, bt, 0, ALPHA)
    dgrad = ops.conv2d_s2_dgrad if kind == "conv" else ops.convT_s2_dgrad
    cout = cb if kind == "conv" else cs

    def run(lo, hi, fuse):
        r = dgrad(None, pack, cout, 1, dy16=dy16[lo:hi], out_bf16=True, fuse=(z16[lo:hi], st[lo:hi], ALPHA) if fuse else None)
        return r if fuse else (r, None)

    g, parts = run(0, B, True)
    assert (parts is not None) == expect, (name, B, None if parts is None else parts.nparts)
    idx = _samples(B)
    dq, wq = O.bf16_round(_f64(dy16[idx])), O.bf16_round(_f64(w))
    ref = O.conv_bwd_input(dq, wq, 2, (2 * s, 2 * s)) if kind == "conv" else O.conv_fwd(dq, wq, 2)
    assert _rms(_f64(g[idx]), O.bf16_round(ref)) < 6e-4
    # norm backward from the fused sums (must be consumed before the next fused launch reuses the workspace)
    C = gshape[-1]
    outs = []
    for p in ((parts, None) if parts is not None else (None,)):
        dgm, dbt, db = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda"), torch.zeros(C, device="cuda")
        dz16 = torch.empty(gshape, dtype=torch.bfloat16, device="cuda")
        ops.instnorm_bwd(z16, st, g, dgm, dbt, 0, 1, ALPHA, out16=dz16, want_f32=False, db=db, partials=p)
        outs.append((dz16, float(dgm), float(dbt), db))
    dref = _norm_ref(_f64(z16[idx]).reshape(len(idx), -1), _f64(g[idx]).reshape(len(idx), -1), _f64(st[idx]))
    assert _rms(_f64(outs[0][0][idx]).reshape(len(idx), -1), O.bf16_round(dref)) < 6e-4
    if parts is not None:
        gabs = float(g.float().abs().sum())
        assert _rms(_f64(outs[0][0]), _f64(outs[1][0])) < 2e-4          # (a few flipped bf16 roundings: sums differ in their last bits)
        assert abs(outs[0][1] - outs[1][1]) <= 1e-6 * gabs and abs(outs[0][2] - outs[1][2]) <= 1e-6 * gabs
        assert _rms(_f64(outs[0][3]), _f64(outs[1][3])) < 1e-5
    # the unfused kernel writes the same gradient; 32-image chunks of the fused launch equal the big launch bit for bit
    g_plain, _ = run(0, B, False)
    assert _rms(_f64(g[idx]), _f64(g_plain[idx])) < 1e-6 or torch.equal(g, g_plain)
    for lo in range(0, B, CHUNK):
        gc, pc = run(lo, lo + CHUNK, True)
        assert torch.equal(gc, g[lo:lo + CHUNK]), (name, lo)
        assert (pc is not None) == expect


@pytest.mark.parametrize("B", [256, 512])
def test_final_layer_fused_data_gradient(ops, B):
    """lg_convT_s1_tanh_bwd_nf as the gen tape (B = 256, with the weight gradient) and the Adjuster branch (B = 512, data
    gradient only) launch it: bf16 dx + the norm-backward sums of decoder level 4."""
    H, cs = 128, 32
    w = _rand((5, 5, 3, cs), 61, 0.05)
    pack = ops.conv_pack(w, 3, cs, 1)
    dpre = _rand((B, H, H, 3), 62, 0.01)
    z16 = _rand((B, H, H, cs), 63, 1.7).add_(0.4).to(torch.bfloat16)
    gm, bt = torch.tensor([0.9], device="cuda"), torch.tensor([0.15], device="cuda")
    st = ops.instnorm_stats(z16.float(), gm, bt, 0, ALPHA)
    x16 = _rand((B, H, H, cs), 64).to(torch.bfloat16)
    wg = B == 256
    dx = torch.empty((B, H, H, cs), dtype=torch.bfloat16, device="cuda")
    dw, db = torch.empty(5, 5, 3, cs, device="cuda"), torch.empty(3, device="cuda")
    _, parts = ops.convT_s1_tanh_bwd(None, dpre, pack, cs, 1, dx16=dx, dw=dw if wg else None, db=db if wg else None,
                                     x16=x16 if wg else None, fuse=(z16, st, ALPHA))
    assert parts is not None
    idx = _samples(B)
    dref = O.conv_fwd(O.bf16_round(_f64(dpre[idx])), O.bf16_round(_f64(w)), 1)
    assert _rms(_f64(dx[idx]), O.bf16_round(dref)) < 6e-4
    outs = []
    for p in (parts, None):
        dgm, dbt = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
        dz16 = torch.empty_like(z16)
        ops.instnorm_bwd(z16, st, dx, dgm, dbt, 0, 1, ALPHA, out16=dz16, want_f32=False, partials=p)
        outs.append((dz16, float(dgm), float(dbt)))
    nref = _norm_ref(_f64(z16[idx]).reshape(len(idx), -1), _f64(dx[idx]).reshape(len(idx), -1), _f64(st[idx]))
    assert _rms(_f64(outs[0][0][idx]).reshape(len(idx), -1), O.bf16_round(nref)) < 6e-4
    gabs = float(dx.float().abs().sum())
    assert _rms(_f64(outs[0][0]), _f64(outs[1][0])) < 2e-4
    assert abs(outs[0][1] - outs[1][1]) <= 1e-6 * gabs and abs(outs[0][2] - outs[1][2]) <= 1e-6 * gabs
    if wg:
        ref_w = O.conv_bwd_filter(O.bf16_round(_f64(dpre)), O.bf16_round(_f64(x16)), 1, 5)
        assert _rms(_f64(dw), ref_w) < 3e-5
    for lo in range(0, B, CHUNK):
        dc = torch.empty((CHUNK, H, H, cs), dtype=torch.bfloat16, device="cuda")
        ops.convT_s1_tanh_bwd(None, dpre[lo:lo + CHUNK], pack, cs, 1, dx16=dc, fuse=(z16[lo:lo + CHUNK], st[lo:lo + CHUNK], ALPHA))
        assert torch.equal(dc, dx[lo:lo + CHUNK]), lo


@pytest.mark.parametrize("init_dim,B,chunk", [pytest.param(8, 256, 32, id="C3-128px-B256"), pytest.param(16, 64, 32, id="C5geom-256px-B64"),
                                              pytest.param(16, 256, 32, id="C5-256px-B256")])
def test_whole_step_at_launch_batch(init_dim, B, chunk):
    """ONE WHOLE bf16 step (b = 11: G, D on 2B, disc tape, gen tape, Adjuster branch on 2B, three Adam applies) at the batch
    bench.py times (C3: 128x128, B = 256) and at the C5 geometry (256x256; B = 64, and the per-GPU share of C5 itself, B = 256):
      * fake / adjusted images and D's head probabilities of sampled rows against the bf16-emulating oracle (per-sample ops);
      * the three loss scalars and EVERY gradient tensor against the same step run on `chunk`-image slices of the batch with
        the same weights (losses are batch means, so the big step must equal the mean over the slices): this is the check that
        no launch of the big step (2B = 512 grids, batch-scaled workspaces, split-K factors) mixes or drops samples.
    Reference: eager_trainer.py:133-168."""
    from test_step_gpu import build, dev_inputs, f32_round, perturbed
    cfg = O.Cfg(init_dim=init_dim, cond_dim=40, batch_size=B)
    W = perturbed(cfg, 7)
    tr = build(cfg, W, "bf16")
    inp = f32_round(O.make_inputs(cfg, B, seed=21))
    d_in = dev_inputs(inp)
    fake, adj, lg, ld, la = tr.train_step_from_inputs(11, d_in)
    torch.cuda.synchronize()
    grads = tr.store.grad.clone()
    losses = np.array([lg.item(), ld.item(), la.item()])
    assert np.isfinite(losses).all() and torch.isfinite(grads).all()
    # ---- sampled rows against the emulating oracle
    idx = _samples(B) if init_dim == 8 else [0, B // 2, B - 1]   # (the fp64 oracle at 256x256 costs seconds per image)
    cfg_e = O.Cfg(init_dim=init_dim, cond_dim=40, batch_size=len(idx), emulate_bf16=True)
    ref, _ = O.generator_fwd(cfg_e, W["G"], inp["noise"][idx], inp["real_cond_2"][idx])
    got = _f64(fake[idx])
    assert np.abs(got - ref).max() < 2e-2 and _rms(got, ref) < 3e-3
    # Adjuster rows: sample i of the first half is img1[i] with cond (c2[i]+1)/2, of the second half fake[i] with (c1[i]+1)/2
    a_img = np.concatenate([inp["real_image_1"][idx], got], 0)
    a_cond = (np.concatenate([inp["real_cond_2"][idx], inp["real_cond_1"][idx]], 0) + 1.0) * 0.5
    cfg_a = O.Cfg(init_dim=init_dim, cond_dim=40, batch_size=2 * len(idx), emulate_bf16=True)
    a_ref, _ = O.adjuster_fwd(cfg_a, W, a_img, a_cond)
    a_got = np.concatenate([_f64(adj[idx]), _f64(adj[[B + i for i in idx]])], 0)
    assert np.abs(a_got - a_ref).max() < 4e-2 and _rms(a_got, a_ref) < 6e-3
    # ---- the same step on slices of the batch, same weights (lr = 0: the applies leave them alone)
    cfg_c = O.Cfg(init_dim=init_dim, cond_dim=40, batch_size=chunk)
    trc = build(cfg_c, W, "bf16")
    trc.opt_cfg = {m: (0.0, b1, b2) for m, (_, b1, b2) in trc.opt_cfg.items()}
    acc = torch.zeros_like(grads, dtype=torch.float64)
    lacc = np.zeros(3)
    n = B // chunk
    for k in range(n):
        sl = {key: v[k * chunk:(k + 1) * chunk].contiguous() for key, v in d_in.items()}
        fk, ak, lgk, ldk, lak = trc.train_step_from_inputs(11, sl)
        assert float((fk - fake[k * chunk:(k + 1) * chunk]).abs().max()) < 2e-2, ("fake rows differ from the slice run", k)
        acc += trc.store.grad.double()
        lacc += np.array([lgk.item(), ldk.item(), lak.item()])
    assert np.abs(lacc / n - losses).max() < 2e-5 * np.abs(losses).max(), (lacc / n, losses)
    mean = (acc / n)
    worst = []
    for m in "DGA":
        for i, (s, e) in enumerate(tr.store.ranges[m]):
            a, b = grads[s:e].double(), mean[s:e]
            den = float(b.pow(2).mean().sqrt()) + 1e-30
            worst.append((float((a - b).pow(2).mean().sqrt()) / den, m, i))
    worst.sort(reverse=True)
    print("whole step vs slices: worst gradient tensors", worst[:5])
    # tensors of 1 element (gamma / beta) are sums with heavy cancellation: bound them against the largest gradient instead
    gmax = float(grads.abs().max())
    numel = {m: [t[5] for t in tr.store.index if t[0] == m] for m in "DGA"}
    for r, m, i in worst:
        s, e = tr.store.ranges[m][i]
        if numel[m][i] == 1:
            assert abs(float(grads[s]) - float(mean[s])) <= 2e-3 * gmax, (m, i)
        else:
            assert r < 2e-3, (m, i, r)


@pytest.mark.parametrize("dtype,B", [pytest.param("bf16", 256, id="C3-bf16-B256"), pytest.param("f32", 64, id="C2-f32-B64")])
def test_whole_step_is_deterministic_at_launch_batch(dtype, B):
    """Round 5 (taken from the kernels to the step): two trainers built from the same weights run the same three steps
    (b = 10: a partition step, 11 and 12: full steps with the Adjuster branch and three Adam applies each) at the batch bench.py times, on
    the same inputs.  EVERY bit of the outputs, losses, gradients, weights and Adam slots must agree: a launch-to-launch difference in any
    of the step's ~190 launches (a packed-fp32 hazard; an atomic with three summands; a read of a buffer another launch still
    writes) shows here at the sizes where round 4's build showed it.  Reference: eager_trainer.py:115-169."""
    from test_step_gpu import build, dev_inputs, f32_round, perturbed
    cfg = O.Cfg(init_dim=8, cond_dim=40, batch_size=B)
    W = perturbed(cfg, 31)
    tr1, tr2 = build(cfg, W, dtype), build(cfg, W, dtype)
    for b in (10, 11, 12):
        d_in = dev_inputs(f32_round(O.make_inputs(cfg, B, seed=40 + b)))
        r1 = [None if t is None else t.clone() for t in tr1.train_step_from_inputs(b, d_in)]
        r2 = tr2.train_step_from_inputs(b, d_in)
        torch.cuda.synchronize()
        for i, (x, y) in enumerate(zip(r1, r2)):
            assert (x is None) == (y is None) and (x is None or torch.equal(x, y)), (dtype, b, "output", i)
        for name in ("grad", "flat", "m", "v"):
            assert torch.equal(getattr(tr1.store, name), getattr(tr2.store, name)), (dtype, b, name)
    assert torch.isfinite(tr1.store.flat).all()


def test_whole_f32_step_at_c2_batch():
    """The C2 configuration at its own batch (128x128, B = 64, exact-f32 MFMA, G + D step only, b = 5 is a partition step, b = 6 a full
    one): fake rows and both losses against the fp64 oracle on sampled rows / the whole batch is too slow on the CPU, so as above:
    rows against the oracle, losses and gradients against the mean of the same step on 16-image slices (bit-level differences only
    in the batch reductions)."""
    from test_step_gpu import build, dev_inputs, f32_round, perturbed
    B, chunk = 64, 16
    cfg = O.Cfg(init_dim=8, cond_dim=40, batch_size=B, train_adj=False)
    W = perturbed(cfg, 7)
    tr = build(cfg, W, "f32")
    inp = f32_round(O.make_inputs(cfg, B, seed=23))
    d_in = dev_inputs(inp)
    fake, adj, lg, ld, la = tr.train_step_from_inputs(6, d_in)
    assert adj is None and la is None
    grads = tr.store.grad.clone()
    idx = [0, B // 2 - 1, B - 1]
    cfg_s = O.Cfg(init_dim=8, cond_dim=40, batch_size=len(idx), train_adj=False)
    ref, _ = O.generator_fwd(cfg_s, W["G"], inp["noise"][idx], inp["real_cond_2"][idx])
    assert np.abs(_f64(fake[idx]) - ref).max() < 2e-5
    cfg_c = O.Cfg(init_dim=8, cond_dim=40, batch_size=chunk, train_adj=False)
    trc = build(cfg_c, W, "f32")
    trc.opt_cfg = {m: (0.0, b1, b2) for m, (_, b1, b2) in trc.opt_cfg.items()}
    acc = torch.zeros_like(grads, dtype=torch.float64)
    lacc = np.zeros(2)
    for k in range(B // chunk):
        sl = {key: v[k * chunk:(k + 1) * chunk].contiguous() for key, v in d_in.items()}
        fk, _, lgk, ldk, _ = trc.train_step_from_inputs(6, sl)
        assert float((fk - fake[k * chunk:(k + 1) * chunk]).abs().max()) < 2e-5
        acc += trc.store.grad.double()
        lacc += np.array([lgk.item(), ldk.item()])
    n = B // chunk
    assert np.abs(lacc / n - np.array([lg.item(), ld.item()])).max() < 2e-6 * max(abs(lg.item()), abs(ld.item()))
    mean = acc / n
    gmax = float(grads.abs().max())
    numel = {m: [t[5] for t in tr.store.index if t[0] == m] for m in "DG"}
    for m in "DG":
        for i, (s, e) in enumerate(tr.store.ranges[m]):
            a, b = grads[s:e].double(), mean[s:e]
            if numel[m][i] == 1:
                assert abs(float(a[0]) - float(b[0])) <= 2e-4 * gmax, (m, i)
            else:
                r = float((a - b).pow(2).mean().sqrt()) / (float(b.pow(2).mean().sqrt()) + 1e-30)
                assert r < 5e-3, (m, i, r)   # (f32: a LeakyReLU pre-activation at rounding distance from 0 may flip between the two groupings)


def test_whole_f32_step_at_c3_batch():
    """The C3 step (128x128, B = 256, G + D + Adjuster, b = 11) on the EXACT-f32 path — the configuration `bench.py --workload c3
    --dtype f32` times, i.e. the throughput number that goes with the north star's 1e-4 loss tolerance.  Fake and adjusted rows
    against the fp64 oracle on sampled rows (2e-5), the three losses against the fp64 oracle's losses of the same step on a
    32-image slice run through BOTH implementations (kernel vs oracle: 2e-5 relative, the bound of tests/test_step_gpu.py), and
    losses / every gradient tensor of the big step against the mean of the same step on 32-image slices (no launch of the 2B = 512
    grids mixes or drops samples).  Reference: eager_trainer.py:133-168."""
    from test_step_gpu import build, dev_inputs, f32_round, perturbed
    B, chunk = 256, 32
    cfg = O.Cfg(init_dim=8, cond_dim=40, batch_size=B)
    W = perturbed(cfg, 7)
    tr = build(cfg, W, "f32")
    inp = f32_round(O.make_inputs(cfg, B, seed=29))
    d_in = dev_inputs(inp)
    fake, adj, lg, ld, la = tr.train_step_from_inputs(11, d_in)
    torch.cuda.synchronize()
    grads = tr.store.grad.clone()
    losses = np.array([lg.item(), ld.item(), la.item()])
    assert np.isfinite(losses).all() and torch.isfinite(grads).all()
    idx = [0, B // 2 - 1, B - 1]
    cfg_s = O.Cfg(init_dim=8, cond_dim=40, batch_size=len(idx))
    ref, _ = O.generator_fwd(cfg_s, W["G"], inp["noise"][idx], inp["real_cond_2"][idx])
    got = _f64(fake[idx])
    assert np.abs(got - ref).max() < 2e-5
    a_img = np.concatenate([inp["real_image_1"][idx], got], 0)
    a_cond = (np.concatenate([inp["real_cond_2"][idx], inp["real_cond_1"][idx]], 0) + 1.0) * 0.5
    a_ref, _ = O.adjuster_fwd(O.Cfg(init_dim=8, cond_dim=40, batch_size=2 * len(idx)), W, a_img, a_cond)
    a_got = np.concatenate([_f64(adj[idx]), _f64(adj[[B + i for i in idx]])], 0)
    assert np.abs(a_got - a_ref).max() < 4e-5
    cfg_c = O.Cfg(init_dim=8, cond_dim=40, batch_size=chunk)
    trc = build(cfg_c, W, "f32")
    trc.opt_cfg = {m: (0.0, b1, b2) for m, (_, b1, b2) in trc.opt_cfg.items()}
    acc = torch.zeros_like(grads, dtype=torch.float64)
    lacc = np.zeros(3)
    n = B // chunk
    for k in range(n):
        sl = {key: v[k * chunk:(k + 1) * chunk].contiguous() for key, v in d_in.items()}
        fk, ak, lgk, ldk, lak = trc.train_step_from_inputs(11, sl)
        assert float((fk - fake[k * chunk:(k + 1) * chunk]).abs().max()) < 2e-5
        acc += trc.store.grad.double()
        lacc += np.array([lgk.item(), ldk.item(), lak.item()])
    assert np.abs(lacc / n - losses).max() < 4e-6 * np.abs(losses).max(), (lacc / n, losses)
    mean = acc / n
    gmax = float(grads.abs().max())
    numel = {m: [t[5] for t in tr.store.index if t[0] == m] for m in "DGA"}
    for m in "DGA":
        for i, (s, e) in enumerate(tr.store.ranges[m]):
            a, b = grads[s:e].double(), mean[s:e]
            if numel[m][i] == 1:
                assert abs(float(a[0]) - float(b[0])) <= 2e-4 * gmax, (m, i)
            else:
                r = float((a - b).pow(2).mean().sqrt()) / (float(b.pow(2).mean().sqrt()) + 1e-30)
                assert r < 5e-3, (m, i, r)
    # the three loss scalars against the fp64 oracle at the stated tolerance (2e-5 < the north star's 1e-4 relative), on the first
    # four images (the oracle's full step costs ~10 s per image here); the slice means above tie the B = 256 losses to such steps
    cfg_4 = O.Cfg(init_dim=8, cond_dim=40, batch_size=4)
    tr4 = build(cfg_4, W, "f32")
    _, _, lg4, ld4, la4 = tr4.train_step_from_inputs(11, {key: v[:4].contiguous() for key, v in d_in.items()})
    o = O.step_gradients(cfg_4, W, 11, {key: v[:4] for key, v in inp.items()})
    for gotl, key in ((lg4, "gen_loss"), (ld4, "disc_loss"), (la4, "adj_loss")):
        assert abs(gotl.item() - o[key]) < 2e-5 * abs(o[key]), (key, gotl.item(), o[key])


@pytest.mark.parametrize("layer", ["enc.conv2", "enc.conv3"])
def test_normalising_down_conv_at_the_adjuster_batch(ops, layer):
    """lg_conv2d_s2_fwd_stats_zn at the size the step launches it: D on the Adjuster's output, 2B = 512 images
    (littlegan_amd/eager_trainer.py, Discriminator.forward_packed(top_only=True); reference eager_trainer.py:158-160).  The conv fed with
    the RAW bf16 map of the level below + its statistics records == InstanceNorm + LeakyReLU apply pass followed by the conv on the
    normalised map, bit for bit (result and moments) — that second path is the one test_forward_with_fused_moments checks against the
    oracle at B = 512 — and == the same normalising conv on 32-image chunks (no cross-sample contamination: the statistics record
    is picked per item).  Samples carry different statistics."""
    _, _, cb, cs, s = next(l for l in LAYERS if l[0] == layer)
    B = 512
    zin = _rand((B, 2 * s, 2 * s, cb), 41, 1.0)
    zin = zin * (0.5 + 2.0 * torch.rand(B, 1, 1, 1, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)))
    zin = zin + _rand((B, 1, 1, 1), 5, 1.0)
    gm_in, bt_in = torch.tensor([0.9], device="cuda"), torch.tensor([0.2], device="cuda")
    gm, bt = torch.tensor([1.2], device="cuda"), torch.tensor([-0.1], device="cuda")
    zin16 = torch.empty(zin.shape, dtype=torch.bfloat16, device="cuda")
    st_in = ops.instnorm_stats(zin, gm_in, bt_in, 0, ALPHA, x16_out=zin16)
    del zin
    w = _rand((5, 5, cb, cs), 43, 0.05)
    bias = _rand((cs,), 44, 0.1)
    pack = ops.conv_pack(w, cb, cs, 1)
    assert ops.conv2d_s2_fwd_stats_zn_supported(B, 2 * s, 2 * s, cb, cs, 1)
    h16 = torch.empty_like(zin16)
    ops.instnorm_apply(zin16, st_in, None, 0, 1, ALPHA, out16=h16, want_f32=False)
    z_ref, st_ref = ops.conv2d_s2_fwd_stats(None, pack, bias, cs, 1, gm, bt, x16=h16, z16=True, alpha=ALPHA)
    st_ref = ops.stats_tensor(st_ref)
    del h16
    z, st = ops.conv2d_s2_fwd_stats_zn(zin16, st_in, ALPHA, pack, bias, cs, 1, gm, bt)
    assert "NORM" in ops.last_kernel()
    assert torch.equal(z, z_ref) and torch.equal(st, st_ref)
    for lo in range(0, B, CHUNK):
        zc, sc = ops.conv2d_s2_fwd_stats_zn(zin16[lo:lo + CHUNK].contiguous(), st_in[lo:lo + CHUNK].contiguous(), ALPHA, pack, bias, cs, 1,
                                            gm, bt)
        assert torch.equal(zc, z[lo:lo + CHUNK]) and torch.equal(sc, st[lo:lo + CHUNK]), lo


@pytest.mark.parametrize("layer,B", [("dec.conv4", 512), ("dec.conv4", 256), ("dec.conv4", 96)])
def test_backward_normalising_data_gradient_at_the_adjuster_batch(ops, layer, B):
    """lg_convT_s2_dgrad_bn at the sizes the step launches it (the Adjuster's decoder chain at 2B = 512, eager_trainer.py:158-163;
    the Generator's on partition steps at B = 256): the data gradient fed with the level's RAW pair (z16, g16) + the per-sample
    coefficients of lg_instnorm_bwd_coef == lg_instnorm_leaky_bwd_z16_p (apply pass writing dz16) followed by lg_convT_s2_dgrad_nf,
    BIT FOR BIT: gradient AND the fused sums of the level below — that second path is the one
    test_fused_data_gradient_and_norm_backward checks against the oracle.  32-image chunks equal the big launch (the coefficient record
    is picked per item; samples carry different statistics), and dz against the oracle on sampled images through the chain."""
    _, _, cb, cs, s = next(l for l in LAYERS if l[0] == layer)
    w = _rand((5, 5, cb, cs), 61, 0.05)
    pack = ops.conv_pack(w, cb, cs, 1)
    shape = (B, 2 * s, 2 * s, cb)
    z = _rand(shape, 62, 1.0) * (0.5 + 2.0 * torch.rand(B, 1, 1, 1, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)))
    z16 = (z + _rand((B, 1, 1, 1), 63, 1.0)).to(torch.bfloat16)
    del z
    g16 = _rand(shape, 64, 1.0).to(torch.bfloat16)
    gm, bt = torch.tensor([0.9], device="cuda"), torch.tensor([0.15], device="cuda")
    st = ops.instnorm_stats(z16.float(), gm, bt, 0, ALPHA)
    zl16 = _rand((B, s, s, cs), 65, 1.3).add_(0.2).to(torch.bfloat16)          # the level below: raw output + statistics
    stl = ops.instnorm_stats(zl16.float(), torch.tensor([1.1], device="cuda"), torch.tensor([-0.05], device="cuda"), 0, ALPHA)
    assert ops.convT_s2_dgrad_bn_supported(B, s, s, cb, cs, 1)
    assert not ops.convT_s2_dgrad_bn_supported(B, 32, 32, 64, 128, 1)   # the 128-column level: measured slower than apply + conv, not built

    def reference(lo, hi, parts):
        dz16 = torch.empty((hi - lo,) + shape[1:], dtype=torch.bfloat16, device="cuda")
        ops.instnorm_bwd(z16[lo:hi], st[lo:hi], g16[lo:hi], None, None, 0, 1, ALPHA, out16=dz16, want_f32=False, partials=parts)
        gl, pl = ops.convT_s2_dgrad(None, pack, cs, 1, dy16=dz16, out_bf16=True, fuse=(zl16[lo:hi], stl[lo:hi], ALPHA))
        return dz16, gl, pl

    # sums {sum g', sum g' c} per (sample, part) as a producer would leave them ([B][4][2] doubles): here from a torch reduction
    zz, gg = z16.double().reshape(B, 4, -1), g16.double().reshape(B, 4, -1)
    mu, a_, b_ = (st[:, 0].double() + st[:, 4].double()).view(B, 1, 1), st[:, 2].view(B, 1, 1), st[:, 3].view(B, 1, 1)
    c32 = (z16.float().reshape(B, 4, -1) - st[:, 0].view(B, 1, 1)) - st[:, 4].view(B, 1, 1)
    gp = torch.where(a_ * c32 + b_ > 0, gg, ALPHA * gg)
    sums = torch.stack([gp.sum(-1), (gp * (zz - mu)).sum(-1)], -1).contiguous()
    del zz, gg, c32, gp

    def partials(lo, hi):
        return ops.NormPartials(sums[lo:hi].contiguous().view(torch.uint8).reshape(-1), 4, ALPHA, (hi - lo,) + shape[1:])

    P = partials(0, B)
    dz_ref, g_ref, p_ref = reference(0, B, P)
    sums_ref = p_ref.buf[:B * p_ref.nparts * 16].clone().view(torch.float64)
    coef = ops.instnorm_bwd_coef(z16, st, P)
    g_bn, p_bn = ops.convT_s2_dgrad_bn(z16, g16, coef, ALPHA, pack, cs, fuse=(zl16, stl, ALPHA))
    assert "BWDNORM" in ops.last_kernel()
    assert torch.equal(g_bn, g_ref)
    assert p_bn.nparts == p_ref.nparts and torch.equal(p_bn.buf[:B * p_bn.nparts * 16].view(torch.float64), sums_ref)
    del g_ref, sums_ref
    # dz of the reference chain against the oracle on sampled images (so the bit-equality above is anchored)
    idx = _samples(B)
    dref = _norm_ref(_f64(z16[idx]).reshape(len(idx), -1), _f64(g16[idx]).reshape(len(idx), -1), _f64(st[idx]))
    assert _rms(_f64(dz_ref[idx]).reshape(len(idx), -1), O.bf16_round(dref)) < 6e-4
    ref = O.conv_fwd(_f64(dz_ref[idx]), O.bf16_round(_f64(w)), 2)
    assert _rms(_f64(g_bn[idx]), O.bf16_round(ref)) < 6e-4
    for lo in range(0, B, CHUNK):
        Pc = partials(lo, lo + CHUNK)
        cc = ops.instnorm_bwd_coef(z16[lo:lo + CHUNK], st[lo:lo + CHUNK].contiguous(), Pc)
        assert torch.equal(cc, coef[lo:lo + CHUNK]), lo
        gc, _ = ops.convT_s2_dgrad_bn(z16[lo:lo + CHUNK], g16[lo:lo + CHUNK], cc, ALPHA, pack, cs,
                                      fuse=(zl16[lo:lo + CHUNK], stl[lo:lo + CHUNK].contiguous(), ALPHA))
        assert torch.equal(gc, g_bn[lo:lo + CHUNK]), lo


@pytest.mark.parametrize("layer", ["enc.conv4", "dec.conv1"])
def test_odd_batch_on_the_8x8_level_takes_another_kernel(ops, layer):
    """the sample-PAIR tiling of the 8 x 8 maps (conv_down3 / conv_up4 <PAIR>) needs an even batch; an odd batch — a short
    last batch, an odd `rows=` slice — runs on conv_halo.hip, whose accumulation order differs.  So on THIS level an odd launch and its
    even + odd slices are NOT bit-equal (everywhere else the tiling depends on the map only and they are).  Pinned here: both paths
    agree with the oracle, and with each other to fp32-accumulation-order noise before the bf16 rounding (a few flipped roundings)."""
    _, kind, cb, cs, s = next(l for l in LAYERS if l[0] == layer)
    B = 5
    w = _rand((5, 5, cb, cs), 71, 0.05)
    bias = _rand((cs if kind == "conv" else cb,), 72, 0.1)
    pack = ops.conv_pack(w, cb, cs, 1)
    gm, bt = torch.tensor([1.0], device="cuda"), torch.tensor([0.0], device="cuda")
    x16 = _rand((B, 2 * s, 2 * s, cb) if kind == "conv" else (B, s, s, cs), 73).to(torch.bfloat16)
    fwd = (lambda xs: ops.conv2d_s2_fwd_stats(None, pack, bias, cs, 1, gm, bt, x16=xs, z16=True)) if kind == "conv" else \
          (lambda xs: ops.convT_s2_fwd_stats(None, pack, bias, cb, 1, gm, bt, x16=xs, z16=True))
    z, st = fwd(x16)
    k_odd = ops.last_kernel()
    z4, st4 = fwd(x16[:4].contiguous())
    k_even = ops.last_kernel()
    z1, st1 = fwd(x16[4:].contiguous())
    assert "PAIR" in k_even and "PAIR" not in k_odd, (k_even, k_odd)
    xq, wq = O.bf16_round(_f64(x16)), O.bf16_round(_f64(w))
    ref = O.conv2d(xq, wq, _f64(bias), 2) if kind == "conv" else O.conv2d_transpose(xq, wq, _f64(bias), 2)
    assert _rms(_f64(z), O.bf16_round(ref)) < 6e-4
    assert _rms(_f64(torch.cat([z4, z1], 0)), O.bf16_round(ref)) < 6e-4
    assert _rms(_f64(z), _f64(torch.cat([z4, z1], 0))) < 6e-4          # (flipped bf16 roundings only)
    sa, sb = _f64(ops.stats_tensor(st)), _f64(torch.cat([ops.stats_tensor(st4), ops.stats_tensor(st1)], 0))
    assert np.abs(sa[:, :5] - sb[:, :5]).max() < 1e-4 * (np.abs(sa[:, :5]).max() + 1.0)


@pytest.mark.parametrize("B", [64, 128])
def test_f32_8x8_level_tilings_that_depend_on_the_batch(ops, B):
    """Round 5 (conv_halo.hip, dispatch_bn2): on the exact-f32 path the 8 x 8 maps' grids are one round of blocks, and two choices now look at
    the batch — conv4 forward splits its contraction in two halves that ADD into the zeroed output while the 64-column grid has at most
    256 blocks (B = 64; not at 2B), convT1 forward takes 128-column tiles once they fill the 512 slots (2B; not at B).  Pinned: either
    side of both thresholds is deterministic over three launches (two summands commute) and meets the fp64 oracle at the f32 bound of this
    file; a launch equals its 32-image chunks bit for bit in z and in the statistics where both sit on the same side (B = 64).  At 2B they
    do not: convT1's 128-column launch and its 64-column chunks give the same z and moment records grouped differently (statistics equal
    to rounding); conv4's chunks are split and its full launch is not, and the two differ by the order of ONE addition per
    output element over a 6400-term fp32 contraction (1.3e-6 rms measured — the size of either result's own distance to the fp64 oracle; bounded here)."""
    gm, bt = torch.tensor([1.1], device="cuda"), torch.tensor([0.05], device="cuda")
    for name in ("enc.conv4", "dec.conv1"):
        _, kind, cb, cs, s = next(l for l in LAYERS if l[0] == name)
        w = _rand((5, 5, cb, cs), 91, 0.05)
        pack = ops.conv_pack(w, cb, cs, 0)
        bias = _rand((cs if kind == "conv" else cb,), 92, 0.1)
        x = _rand((B, 2 * s, 2 * s, cb) if kind == "conv" else (B, s, s, cs), 93)
        def fwd(xs):
            z, st = ops.conv2d_s2_fwd_stats(xs, pack, bias, cs, 0, gm, bt, alpha=ALPHA) if kind == "conv" else \
                    ops.convT_s2_fwd_stats(xs, pack, bias, cb, 0, gm, bt, alpha=ALPHA)
            st = ops.instnorm_stats(z, gm, bt, 0, ALPHA) if st is None else ops.stats_tensor(st)
            return z.clone(), st.clone()

        z, st = fwd(x)
        for _ in range(2):
            z2, st2 = fwd(x)
            assert torch.equal(z2, z) and torch.equal(st2, st), (name, B, "launch-to-launch")
        for lo in range(0, B, CHUNK):
            zc, stc = fwd(x[lo:lo + CHUNK].contiguous())
            if B > 64:   # the launch and its chunks sit on different sides: split | whole contraction, 128- | 64-column moment records
                if kind == "conv":
                    assert _rms(_f64(zc), _f64(z[lo:lo + CHUNK])) < 5e-6 and not torch.equal(zc, z[lo:lo + CHUNK]), (name, B, lo)
                else:
                    assert torch.equal(zc, z[lo:lo + CHUNK]), (name, B, lo)     # a tile's width does not change the conv result
                assert np.abs(_f64(stc) - _f64(st[lo:lo + CHUNK])).max() < 1e-5 * (np.abs(_f64(st)).max() + 1.0), (name, B, lo, "stats")
            else:
                assert torch.equal(zc, z[lo:lo + CHUNK]), (name, B, lo)
                assert torch.equal(stc, st[lo:lo + CHUNK]), (name, B, lo, "stats")
        n = _samples(B)[:3]
        ref = O.conv2d(_f64(x[n]), _f64(w), _f64(bias), 2) if kind == "conv" else O.conv2d_transpose(_f64(x[n]), _f64(w), _f64(bias), 2)
        assert _rms(_f64(z[n]), ref) < 1e-5 and np.abs(_f64(z[n]) - ref).max() < 1e-4 * np.abs(ref).max(), (name, B)


@pytest.mark.parametrize("B", [1, 2, 3, 5, 84, 86, 126])
def test_f32_tiling_rules_at_odd_batches_and_their_thresholds(ops, B):
    """Round 5: the exact-f32 tiling rules of conv_halo.hip look at the batch — the split contraction of conv4 forward up to 256 blocks
    (B <= 84 at 384 columns; 86 is the first unsplit batch), 128-column UP tiles from 512 blocks (B >= 128; 126 is the last narrow one) —
    and the 8 x 8 maps are tiled in sample PAIRS, so an odd batch leaves half a tile empty.  Every one of these launches against the fp64
    oracle on all its samples (the maps are small), plus the resident form of the N = 32 level at odd batches."""
    gm, bt = torch.tensor([0.9], device="cuda"), torch.tensor([0.1], device="cuda")
    cases = [("conv", 256, 384, 8), ("convT", 256, 384, 8)] + ([("convT", 32, 64, 64)] if B <= 5 else [])
    for kind, cb, cs, s in cases:
        w = _rand((5, 5, cb, cs), 95, 0.05)
        pack = ops.conv_pack(w, cb, cs, 0)
        bias = _rand((cs if kind == "conv" else cb,), 96, 0.1)
        x = _rand((B, 2 * s, 2 * s, cb) if kind == "conv" else (B, s, s, cs), 97)
        z, st = (ops.conv2d_s2_fwd_stats(x, pack, bias, cs, 0, gm, bt, alpha=ALPHA) if kind == "conv" else
                 ops.convT_s2_fwd_stats(x, pack, bias, cb, 0, gm, bt, alpha=ALPHA))
        kern = ops.last_kernel()
        st = ops.instnorm_stats(z, gm, bt, 0, ALPHA) if st is None else ops.stats_tensor(st)
        n = list(range(B)) if B <= 5 else [0, 1, B // 2, B - 2, B - 1]
        ref = O.conv2d(_f64(x[n]), _f64(w), _f64(bias), 2) if kind == "conv" else O.conv2d_transpose(_f64(x[n]), _f64(w), _f64(bias), 2)
        got = _f64(z[n])
        assert _rms(got, ref) < 1e-5 and np.abs(got - ref).max() < 1e-4 * np.abs(ref).max(), (kind, cb, cs, B, kern)
        mu, sg = ref.reshape(len(n), -1).mean(1), ref.reshape(len(n), -1).std(1)
        s_ = _f64(st[n])
        assert np.abs(s_[:, 0] + s_[:, 4] - mu).max() < 3e-6 * max(np.abs(mu).max(), sg.max()), (kind, B, kern)
        assert np.abs(s_[:, 1] - sg).max() < 3e-6 * sg.max(), (kind, B, kern)
        if kind == "convT" and cb == 32:
            assert kern == "conv_halo_kernel<f32,UP,resident>", kern


@pytest.mark.parametrize("B", [256, 512])
@pytest.mark.parametrize("layer", [l for l in LAYERS if l[2] != 3], ids=[l[0] for l in LAYERS if l[2] != 3])
def test_persistent_kernels_are_deterministic(ops, layer, B):
    """Round 4 met a build of one persistent kernel (the BWDNORM form of conv_down3 with its per-item record read by vector loads) whose
    output changed from launch to launch with two blocks per CU while every single result was within rounding of the oracle — only a
    bit-for-bit comparison shows that.  Here: forward with fused moments, fused data gradient and weight gradient of every stride-2 layer
    at the launch sizes, three launches each on the same inputs (cache-warm, back to back), bit for bit."""
    name, kind, cb, cs, s = layer
    w = _rand((5, 5, cb, cs), 81, 0.05)
    pack = ops.conv_pack(w, cb, cs, 1)
    gm, bt = torch.tensor([1.1], device="cuda"), torch.tensor([0.05], device="cuda")
    big16 = _rand((B, 2 * s, 2 * s, cb), 82).to(torch.bfloat16)
    small16 = _rand((B, s, s, cs), 83).to(torch.bfloat16)
    bias = _rand((cs if kind == "conv" else cb,), 84, 0.1)

    def fwd():
        if kind == "conv":
            z, st = ops.conv2d_s2_fwd_stats(None, pack, bias, cs, 1, gm, bt, x16=big16, z16=True)
        else:
            z, st = ops.convT_s2_fwd_stats(None, pack, bias, cb, 1, gm, bt, x16=small16, z16=True)
        return z, ops.stats_tensor(st)

    zl16 = big16 if kind == "conv" else small16            # stands for the raw output of the level the gradient belongs to
    stl = ops.instnorm_stats(zl16.float(), gm, bt, 0, ALPHA)

    def dgrad():
        if kind == "conv":
            g, p = ops.conv2d_s2_dgrad(None, pack, cb, 1, dy16=small16, out_bf16=True, fuse=(zl16, stl, ALPHA))
        else:
            g, p = ops.convT_s2_dgrad(None, pack, cs, 1, dy16=big16, out_bf16=True, fuse=(zl16, stl, ALPHA))
        return g, (p.buf[:B * p.nparts * 16].clone() if p is not None else torch.zeros(1, device="cuda"))

    def wgrad():
        dw = torch.empty(5, 5, cb, cs, device="cuda")
        if kind == "conv":
            ops.conv2d_s2_wgrad(None, None, dw, False, 1, x16=big16, dy16=small16)
        else:
            ops.convT_s2_wgrad(None, None, dw, False, 1, x16=small16, dy16=big16)
        return (dw,)

    fns = [fwd, dgrad, wgrad]
    if kind == "convT" and ops.convT_s2_dgrad_bn_supported(B, s, s, cb, cs, 1):
        # the BWDNORM form itself (lg_convT_s2_dgrad_bn; round 5: the record's fields are wave-uniform scalars): the build whose
        # packed subtraction lost m1 differed in some hundred elements on EVERY launch at a sixteenth of this batch
        g16 = _rand((B, 2 * s, 2 * s, cb), 85).to(torch.bfloat16)
        zb16 = (_rand((B, 2 * s, 2 * s, cb), 86, 1.5) + _rand((B, 1, 1, 1), 87)).to(torch.bfloat16)
        stb = ops.instnorm_stats(zb16.float(), gm, bt, 0, ALPHA)
        coef = torch.empty(B, 8, device="cuda")
        coef[:, 0], coef[:, 1], coef[:, 2], coef[:, 3] = stb[:, 0], stb[:, 4], stb[:, 2], stb[:, 3]
        coef[:, 4] = 1e-3 * _rand((B,), 88); coef[:, 5] = 1e-3 * _rand((B,), 89); coef[:, 6] = 1e-10; coef[:, 7] = -1e-10

        def dgrad_bn():
            g, p = ops.convT_s2_dgrad_bn(zb16, g16, coef, ALPHA, pack, cs, fuse=(zl16, stl, ALPHA))
            assert "BWDNORM" in ops.last_kernel()
            return g, p.buf[:B * p.nparts * 16].clone()
        fns.append(dgrad_bn)
    for fn in fns:
        first = [t.clone() for t in fn()]
        for rep in range(2):
            again = fn()
            torch.cuda.synchronize()
            for a, b in zip(first, again):
                assert torch.equal(a, b), (name, fn.__name__, rep)
