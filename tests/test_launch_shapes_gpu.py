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
CONFIGS = [pytest.param(1, 256, id="bf16-B256"), pytest.param(0, 64, id="f32-B64")]
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
    the batch against the bf16-emulating oracle (the forward is per-sample), and the losses of a full step are finite."""
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
