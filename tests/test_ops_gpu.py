"""Per-op parity: every C-ABI entry point vs the fp64 numpy oracle on the same seeded inputs.
Tolerances (max-abs error relative to max-abs of the oracle result):
  f32 path  (exact-f32 MFMA, f32 accumulate)         : 3e-5
  bf16 path (bf16 operands, f32 accumulate)           : 2e-2   (operand rounding 2^-8 per factor)
"""
import numpy as np
import pytest
import torch

from oracle import np_oracle as O

pytestmark = pytest.mark.gpu

TOL = {0: 3e-5, 1: 2e-2}


@pytest.fixture(scope="module")
def ops():
    from littlegan_amd import ops as _ops
    return _ops


def dev(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")


def rel(got, exp):
    got = got.detach().cpu().double().numpy() if torch.is_tensor(got) else np.asarray(got)
    return np.abs(got - exp).max() / (np.abs(exp).max() + 1e-30)


def r32(rng, *shape, scale=1.0):
    """random values exactly representable in f32 (so the f32 kernels see the oracle's inputs)"""
    return (rng.standard_normal(shape) * scale).astype(np.float32).astype(np.float64)


S2_CASES = [  # (B, Hs, Ws, cb, cs)
    (3, 5, 6, 32, 64), (2, 8, 9, 64, 128), (2, 4, 4, 128, 32), (2, 3, 3, 256, 384), (3, 7, 5, 3, 64), (2, 6, 6, 3, 32),
    (1, 16, 16, 32, 32), (2, 8, 16, 3, 64), (1, 16, 16, 3, 32), (2, 16, 32, 3, 64),
]


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case", S2_CASES)
def test_conv2d_s2_fwd_dgrad_wgrad(ops, case, dtype):
    B, Hs, Ws, cb, cs = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = r32(rng, B, 2 * Hs, 2 * Ws, cb)
    w = r32(rng, 5, 5, cb, cs, scale=0.1)
    b = r32(rng, cs)
    dy = r32(rng, B, Hs, Ws, cs)
    pack = ops.conv_pack(dev(w), cb, cs, dtype)
    y = ops.conv2d_s2_fwd(dev(x), pack, dev(b), cs, dtype)
    assert rel(y, O.conv2d(x, w, b, 2)) < TOL[dtype]
    dx_e, dw_e, db_e = O.conv2d_bwd(x, w, dy, 2)
    dx = ops.conv2d_s2_dgrad(dev(dy), pack, cb, dtype)
    assert rel(dx, dx_e) < TOL[dtype]
    dw = torch.full((5, 5, cb, cs), 7.0, device="cuda")
    ops.conv2d_s2_wgrad(dev(x), dev(dy), dw, False, dtype)
    assert rel(dw, dw_e) < TOL[dtype]
    ops.conv2d_s2_wgrad(dev(x), dev(dy), dw, True, dtype)  # accumulate
    assert rel(dw, 2 * dw_e) < TOL[dtype]
    db = torch.empty(cs, device="cuda")
    ops.bias_grad(dev(dy), db)
    assert rel(db, db_e) < 3e-5


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case", [c for c in S2_CASES if c[3] != 3])
def test_convT_s2_fwd_dgrad_wgrad(ops, case, dtype):
    B, Hs, Ws, cb, cs = case
    rng = np.random.default_rng(hash(case) % 2**31 + 1)
    x = r32(rng, B, Hs, Ws, cs)
    w = r32(rng, 5, 5, cb, cs, scale=0.1)  # HWOI: out=cb, in=cs
    b = r32(rng, cb)
    dy = r32(rng, B, 2 * Hs, 2 * Ws, cb)
    pack = ops.conv_pack(dev(w), cb, cs, dtype)
    y = ops.convT_s2_fwd(dev(x), pack, dev(b), cb, dtype)
    assert rel(y, O.conv2d_transpose(x, w, b, 2)) < TOL[dtype]
    dx_e, dw_e, db_e = O.conv2d_transpose_bwd(x, w, dy, 2)
    dx = ops.convT_s2_dgrad(dev(dy), pack, cs, dtype)
    assert rel(dx, dx_e) < TOL[dtype]
    dw = torch.zeros(5, 5, cb, cs, device="cuda")
    ops.convT_s2_wgrad(dev(x), dev(dy), dw, False, dtype)
    assert rel(dw, dw_e) < TOL[dtype]
    db = torch.empty(cb, device="cuda")
    ops.bias_grad(dev(dy), db)
    assert rel(db, db_e) < 3e-5


@pytest.mark.parametrize("case", [(2, 8, 16, 32, 64), (3, 4, 16, 64, 128), (2, 8, 8, 32, 64), (5, 8, 8, 64, 128), (1, 16, 32, 32, 64),
                                  (2, 12, 16, 32, 192)])
def test_f32_all_taps_weight_gradient(ops, case):
    """wgrad_at32.hip (exact-f32 path, v_mfma_f32_32x32x2_f32, all 25 taps per block): both tilings (16 x 4 strips; whole 8-column
    maps), odd batches, several (cb / 32) x (cs / 64) units, a map whose height is a multiple of 4 but not 8, in the conv and the
    transposed-conv operand order, overwrite and accumulate — against the fp64 oracle at the f32 tolerance."""
    B, Hm, Wm, cb, cs = case
    rng = np.random.default_rng(zlib_crc(case) + 17)
    big = r32(rng, B, 2 * Hm, 2 * Wm, cb)
    small = r32(rng, B, Hm, Wm, cs)
    w = r32(rng, 5, 5, cb, cs, scale=0.1)
    dw_e = O.conv2d_bwd(big, w, small, 2)[1]          # conv: x = big, dy = small
    dw = torch.full((5, 5, cb, cs), 3.0, device="cuda")
    ops.conv2d_s2_wgrad(dev(big), dev(small), dw, False, 0)
    assert ops.last_kernel().startswith("wgrad_at32_kernel"), ops.last_kernel()
    assert rel(dw, dw_e) < TOL[0]
    ops.conv2d_s2_wgrad(dev(big), dev(small), dw, True, 0)
    assert rel(dw, 2 * dw_e) < TOL[0]
    dwT_e = O.conv2d_transpose_bwd(small, w, big, 2)[1]   # transposed conv: x = small, dy = big
    dwT = torch.zeros(5, 5, cb, cs, device="cuda")
    ops.convT_s2_wgrad(dev(small), dev(big), dwT, False, 0)
    assert ops.last_kernel().startswith("wgrad_at32_kernel")
    assert rel(dwT, dwT_e) < TOL[0]


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case", [(2, 6, 10, 32), (3, 16, 16, 32), (1, 4, 4, 64)])
def test_convT_s1_tanh_fwd_bwd(ops, case, dtype):
    B, H, W, cs = case
    rng = np.random.default_rng(5)
    x = r32(rng, B, H, W, cs)
    w = r32(rng, 5, 5, 3, cs, scale=0.05)
    b = r32(rng, 3, scale=0.1)
    dpre = r32(rng, B, H, W, 3)
    pack = ops.conv_pack(dev(w), 3, cs, dtype)
    y = ops.convT_s1_tanh_fwd(dev(x), pack, dev(b), 3, dtype)
    assert rel(y, np.tanh(O.conv2d_transpose(x, w, b, 1))) < TOL[dtype]
    dx_e, dw_e, db_e = O.conv2d_transpose_bwd(x, w, dpre, 1)
    dx = torch.empty(B, H, W, cs, device="cuda")
    dw = torch.empty(5, 5, 3, cs, device="cuda")
    db = torch.empty(3, device="cuda")
    ops.convT_s1_tanh_bwd(dev(x), dev(dpre), pack, cs, dtype, dx=dx, dw=dw, db=db)
    assert rel(dx, dx_e) < TOL[dtype]
    assert rel(dw, dw_e) < 3e-5  # patch wgrad always runs on the exact f32 MFMA
    assert rel(db, db_e) < 3e-5


@pytest.mark.parametrize("pre,post,skip", [(0, 1, False), (1, 0, True), (0, 1, True), (1, 0, False)])
@pytest.mark.parametrize("shape", [(3, 4, 4, 32), (2, 8, 8, 96), (5, 24576)])
def test_instnorm_stats_apply_bwd(ops, shape, pre, post, skip):
    rng = np.random.default_rng(9)
    a = 0.3
    x = r32(rng, *shape) * 1.5 + 0.7
    g = r32(rng, *shape)
    sk = r32(rng, *shape) if skip else None
    gamma, beta = 1.3, -0.2
    xx = O.leaky(x, a) if pre else x
    y_e, cache = O.instnorm(xx, gamma, beta)
    out_e = (O.leaky(y_e, a) if post else y_e) + (sk if skip else 0.0)
    gm, bt = dev(np.array([gamma])), dev(np.array([beta]))
    stats = ops.instnorm_stats(dev(x), gm, bt, pre, a)
    B = shape[0]
    mu_e = xx.reshape(B, -1).mean(1)
    sd_e = xx.reshape(B, -1).std(1)
    assert rel(stats[:, 0], mu_e) < 1e-5 and rel(stats[:, 1], sd_e) < 1e-5
    out = ops.instnorm_apply(dev(x), stats, dev(sk) if skip else None, pre, post, a)
    assert rel(out, out_e) < 1e-5
    dy = O.leaky_bwd(y_e, g, a) if post else g
    dxx_e, dg_e, db_e = O.instnorm_bwd(cache, gamma, dy)
    dx_e = O.leaky_bwd(x, dxx_e, a) if pre else dxx_e
    dg, db = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    dx = ops.instnorm_bwd(dev(x), stats, dev(g), dg, db, pre, post, a)
    assert rel(dx, dx_e) < 2e-5
    assert abs(dg.item() - dg_e) < 2e-5 * max(1.0, abs(dg_e)) * 10
    assert abs(db.item() - db_e) < 2e-5 * max(1.0, abs(db_e)) * 10


def test_instnorm_constant_sample_is_beta(ops):
    x = torch.full((2, 4, 4, 32), 3.25, device="cuda")
    gm, bt = dev(np.array([1.7])), dev(np.array([-0.4]))
    stats = ops.instnorm_stats(x, gm, bt, 0, 0.3)
    y = ops.instnorm_apply(x, stats, None, 0, 0, 0.3)
    assert torch.allclose(y, torch.full_like(y, -0.4), atol=1e-6)


@pytest.mark.parametrize("B,K,N", [(5, 133, 24576), (3, 16, 256), (17, 40, 1024), (64, 133, 24576), (32, 16, 256), (96, 37, 1024),
                                   (256, 133, 24576), (512, 40, 24576)])  # from (64, ...) on: fp32-MFMA kernels (skinny_mfma.hip); the last two: G.dense at B = 256, A.dense at 2B = 512 (the C3 launch shapes)
def test_dense_fwd_wgrad(ops, B, K, N):
    rng = np.random.default_rng(3)
    x, w, b, dy = r32(rng, B, K), r32(rng, K, N, scale=0.1), r32(rng, N), r32(rng, B, N)
    y = ops.dense_fwd(dev(x), dev(w), dev(b))
    assert rel(y, x @ w + b) < 1e-5
    dw, db = torch.empty(K, N, device="cuda"), torch.empty(N, device="cuda")
    ops.dense_wgrad(dev(x), dev(dy), dw, db)
    assert rel(dw, x.T @ dy) < 1e-5 and rel(db, dy.sum(0)) < 1e-5


@pytest.mark.parametrize("B,K,c", [(5, 24576, 40), (4, 256, 5), (3, 1024, 7), (64, 24576, 40), (64, 1024, 5), (128, 512, 33),
                                   (256, 24576, 40), (512, 24576, 40)])  # from (64, ...) on: fp32-MFMA kernels; the last two: the C3 launch shapes (gen tape B = 256; D forward / disc tape / Adjuster branch 2B = 512)
def test_heads_fwd_dgrad_wgrad(ops, B, K, c):
    rng = np.random.default_rng(4)
    x = r32(rng, B, K)
    wpr, wc = r32(rng, K, 1, scale=0.02), r32(rng, K, c, scale=0.02)
    bpr, bc = r32(rng, 1), r32(rng, c)
    dz = r32(rng, B, 1 + c)
    p = ops.heads_fwd(dev(x), dev(wpr), dev(bpr), dev(wc), dev(bc))
    p_e = np.concatenate([O.sigmoid(x @ wpr + bpr), O.sigmoid(x @ wc + bc)], 1)
    assert rel(p, p_e) < 1e-5
    dx = ops.heads_dgrad(dev(dz), dev(wpr), dev(wc))
    assert rel(dx, dz[:, :1] @ wpr.T + dz[:, 1:] @ wc.T) < 1e-5
    dwpr, dbpr = torch.empty(K, 1, device="cuda"), torch.empty(1, device="cuda")
    dwc, dbc = torch.empty(K, c, device="cuda"), torch.empty(c, device="cuda")
    ops.heads_wgrad(dev(x), dev(dz), dwpr, dbpr, dwc, dbc)
    assert rel(dwpr, x.T @ dz[:, :1]) < 1e-5 and rel(dwc, x.T @ dz[:, 1:]) < 1e-5
    assert rel(dbpr, dz[:, 0].sum(0, keepdims=True)) < 1e-5 and rel(dbc, dz[:, 1:].sum(0)) < 1e-5


def test_skinny_mfma_weight_gradients_accumulate(ops):
    """accumulate=True on the fp32-MFMA weight-gradient kernels (aligned shapes): dw += x^T dy on top of what is there, biases too."""
    rng = np.random.default_rng(11)
    B, K, N, c = 64, 133, 1024, 40
    x, dy = r32(rng, B, K), r32(rng, B, N)
    dw0, db0 = r32(rng, K, N), r32(rng, N)
    dw, db = dev(dw0), dev(db0)
    ops.dense_wgrad(dev(x), dev(dy), dw, db, accumulate=True)
    assert rel(dw, dw0 + x.T @ dy) < 1e-5 and rel(db, db0 + dy.sum(0)) < 1e-5
    Kh = 1024
    xh, dz = r32(rng, B, Kh), r32(rng, B, 1 + c)
    w0, b0, wc0, bc0 = r32(rng, Kh, 1), r32(rng, 1), r32(rng, Kh, c), r32(rng, c)
    dwpr, dbpr, dwc, dbc = dev(w0), dev(b0), dev(wc0), dev(bc0)
    ops.heads_wgrad(dev(xh), dev(dz), dwpr, dbpr, dwc, dbc, accumulate=True)
    assert rel(dwpr, w0 + xh.T @ dz[:, :1]) < 1e-5 and rel(dwc, wc0 + xh.T @ dz[:, 1:]) < 1e-5
    assert rel(dbpr, b0 + dz[:, 0].sum(0, keepdims=True)) < 1e-5 and rel(dbc, bc0 + dz[:, 1:].sum(0)) < 1e-5


def test_bce_heads_loss(ops):
    rng = np.random.default_rng(6)
    B, c = 7, 5
    p = rng.uniform(0.02, 0.98, (B, 1 + c)).astype(np.float32).astype(np.float64)
    p[0, 0], p[1, 2] = 0.0, 1.0  # saturated: clipped, zero gradient
    t_c = O.soft(2.0 * rng.integers(0, 2, (B, c)) - 1.0).astype(np.float32).astype(np.float64)
    loss = torch.zeros(1, device="cuda")
    dz = torch.empty(B, 1 + c, device="cuda")
    ops.bce_heads_loss(dev(p), dev(t_c), O.soft(1.0), 1.0, 2.0, loss, dz, False)
    exp = O.bce_mean(O.soft(1.0), p[:, :1]) + 2.0 * O.bce_mean(t_c, p[:, 1:])
    assert abs(loss.item() - exp) < 2e-6 * abs(exp) + 1e-6
    dp = np.concatenate([O.bce_mean_bwd(O.soft(1.0), p[:, :1]), 2.0 * O.bce_mean_bwd(t_c, p[:, 1:])], 1)
    assert rel(dz, dp * p * (1 - p)) < 1e-5
    ops.bce_heads_loss(dev(p), None, O.soft(0.0), 1.0, 0.0, loss, dz, True)
    exp2 = exp + O.bce_mean(O.soft(0.0), p[:, :1])
    assert abs(loss.item() - exp2) < 2e-6 * abs(exp2) + 1e-6
    assert float(dz[:, 1:].abs().max()) == 0.0


def test_l1_tanh_loss(ops):
    rng = np.random.default_rng(7)
    t = rng.uniform(-1, 1, (3, 8, 8, 3)).astype(np.float32).astype(np.float64)
    img = np.tanh(r32(rng, 3, 8, 8, 3)).astype(np.float32).astype(np.float64)
    gin = r32(rng, 3, 8, 8, 3, scale=1e-3)
    loss = torch.zeros(1, device="cuda")
    dpre = torch.empty(3, 8, 8, 3, device="cuda")
    ops.l1_tanh_loss(dev(t), dev(img), dev(gin), dpre, loss, 0.02, False)
    assert abs(loss.item() - 0.02 * O.l1_mean(t, img)) < 1e-6
    exp = (gin + 0.02 * O.l1_mean_bwd_b(t, img)) * (1 - img * img)
    assert rel(dpre, exp) < 1e-5


def test_clip_adam(ops):
    rng = np.random.default_rng(8)
    n = 1000
    w0, g = r32(rng, n), r32(rng, n)
    st = O.AdamState(5e-5, 0.5, 0.9, 1)
    ws = [w0.copy()]
    w, m, v = dev(w0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    state = dev(np.array([0.5, 0.9]))
    for _ in range(3):
        st.apply(ws, [0], [np.clip(0.5 * g, -0.5, 0.5)])
        ops.clip_adam_update(w, dev(g), m, v, state, 5e-5, 0.5, 0.9, 1e-8, 0.5, gscale=0.5)
        ops.adam_advance(state, 0.5, 0.9)
    assert np.abs(w.cpu().numpy() - ws[0]).max() < 1e-6
    assert abs(state[0].item() - 0.5 ** 4) < 1e-7


HALO_CASES = [  # (mode, B, Hm, Wm, Cs, N): shapes the LDS halo-tile kernel covers (power-of-two maps)
    ("up", 2, 16, 16, 64, 128), ("up", 3, 8, 8, 128, 64), ("up", 2, 4, 4, 32, 32), ("up", 2, 32, 32, 64, 32),
    ("up", 2, 16, 16, 64, 3), ("down", 2, 16, 16, 64, 128), ("down", 3, 8, 8, 32, 64), ("down", 1, 32, 16, 32, 32),
    ("s1t", 2, 16, 16, 32, 3), ("s1t", 1, 32, 32, 32, 3), ("up", 5, 2, 2, 64, 64),
]


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case", HALO_CASES)
def test_halo_tile_kernel_matches_oracle(ops, case, dtype):
    """Same ABI entry points; these shapes route to conv_halo.hip (the gather kernel is covered by the odd shapes)."""
    mode, B, Hm, Wm, Cs, N = case
    import zlib
    rng = np.random.default_rng(zlib.crc32(repr(case).encode()))
    b = r32(rng, N)
    if mode == "up":      # convT fwd: x [B,Hm,Wm,Cs] -> [B,2Hm,2Wm,N]; kernel [5,5,cb=N,cs=Cs]
        x, w = r32(rng, B, Hm, Wm, Cs), r32(rng, 5, 5, N, Cs, scale=0.1)
        if N == 3:        # conv1 data-gradient form (cb == 3): dgrad entry point, no bias
            got = ops.conv2d_s2_dgrad(dev(x), ops.conv_pack(dev(w), N, Cs, dtype), N, dtype)
            exp = O.conv_bwd_input(x, w, 2, (2 * Hm, 2 * Wm))
        else:
            got = ops.convT_s2_fwd(dev(x), ops.conv_pack(dev(w), N, Cs, dtype), dev(b), N, dtype)
            exp = O.conv2d_transpose(x, w, b, 2)
    elif mode == "down":  # conv fwd: x [B,2Hm,2Wm,Cs] -> [B,Hm,Wm,N]; kernel [5,5,cb=Cs,cs=N]
        x, w = r32(rng, B, 2 * Hm, 2 * Wm, Cs), r32(rng, 5, 5, Cs, N, scale=0.1)
        got = ops.conv2d_s2_fwd(dev(x), ops.conv_pack(dev(w), Cs, N, dtype), dev(b), N, dtype)
        exp = O.conv2d(x, w, b, 2)
    else:                 # final stride-1 convT + tanh
        x, w = r32(rng, B, Hm, Wm, Cs), r32(rng, 5, 5, N, Cs, scale=0.05)
        got = ops.convT_s1_tanh_fwd(dev(x), ops.conv_pack(dev(w), N, Cs, dtype), dev(b), N, dtype)
        exp = np.tanh(O.conv2d_transpose(x, w, b, 1))
    assert rel(got, exp) < TOL[dtype]


@pytest.mark.parametrize("case", [("up", 2, 16, 16, 64, 128), ("down", 2, 16, 16, 64, 128), ("up", 2, 32, 32, 64, 32),
                                  ("down", 3, 8, 8, 32, 64)])
def test_bf16_mirror_operands_are_bit_identical(ops, case):
    """The bf16 mirrors written by the norm kernels are exactly the MFMA operand images the conv / wgrad kernels
    would round to themselves, so feeding them must not change a single bit of the result."""
    mode, B, Hm, Wm, Cs, N = case
    rng = np.random.default_rng(17)
    if mode == "up":   # convT fwd / its wgrad: small = x [B,Hm,Wm,Cs], big = dy [B,2Hm,2Wm,N]
        x, w, dy = r32(rng, B, Hm, Wm, Cs), r32(rng, 5, 5, N, Cs, scale=0.1), r32(rng, B, 2 * Hm, 2 * Wm, N)
        xd, dyd, pack = dev(x), dev(dy), ops.conv_pack(dev(w), N, Cs, 1)
        x16, dy16 = xd.to(torch.bfloat16), dyd.to(torch.bfloat16)
        b = torch.zeros(N, device="cuda")
        y0 = ops.convT_s2_fwd(xd, pack, b, N, 1)
        gm, bt = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")
        y1, st = ops.convT_s2_fwd_stats(xd, pack, b, N, 1, gm, bt, x16=x16)
        dx0, dx1 = ops.convT_s2_dgrad(dyd, pack, Cs, 1), ops.convT_s2_dgrad(dyd, pack, Cs, 1, dy16=dy16)
        dw0, dw1 = torch.empty(5, 5, N, Cs, device="cuda"), torch.empty(5, 5, N, Cs, device="cuda")
        ops.convT_s2_wgrad(xd, dyd, dw0, False, 1)
        ops.convT_s2_wgrad(xd, dyd, dw1, False, 1, x16=x16, dy16=dy16)
        zz = y1
    else:              # conv fwd: big = x [B,2Hm,2Wm,Cs], small = dy [B,Hm,Wm,N]
        x, w, dy = r32(rng, B, 2 * Hm, 2 * Wm, Cs), r32(rng, 5, 5, Cs, N, scale=0.1), r32(rng, B, Hm, Wm, N)
        xd, dyd, pack = dev(x), dev(dy), ops.conv_pack(dev(w), Cs, N, 1)
        x16, dy16 = xd.to(torch.bfloat16), dyd.to(torch.bfloat16)
        b = torch.zeros(N, device="cuda")
        y0 = ops.conv2d_s2_fwd(xd, pack, b, N, 1)
        gm, bt = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")
        y1, st = ops.conv2d_s2_fwd_stats(xd, pack, b, N, 1, gm, bt, x16=x16)
        dx0, dx1 = ops.conv2d_s2_dgrad(dyd, pack, Cs, 1), ops.conv2d_s2_dgrad(dyd, pack, Cs, 1, dy16=dy16)
        dw0, dw1 = torch.empty(5, 5, Cs, N, device="cuda"), torch.empty(5, 5, Cs, N, device="cuda")
        ops.conv2d_s2_wgrad(xd, dyd, dw0, False, 1)
        ops.conv2d_s2_wgrad(xd, dyd, dw1, False, 1, x16=x16, dy16=dy16)
        zz = y1
    assert torch.equal(y0, y1) and torch.equal(dx0, dx1)
    # the mirrors may select the all-taps weight-gradient kernel (csrc/wgrad_at.hip): same bf16 products, another fp32 summation order
    assert torch.equal(dw0, dw1) or rel(dw1, dw0.cpu().numpy()) < 2e-6
    if st is not None:  # fused moments == moments of the produced tensor
        zf = zz.double().reshape(B, -1)
        assert rel(st[:, 0], zf.mean(1).cpu().numpy()) < 1e-6 and rel(st[:, 1], zf.std(1, unbiased=False).cpu().numpy()) < 1e-6


def test_norm_kernels_write_bf16_mirrors(ops):
    rng = np.random.default_rng(3)
    x, g = dev(r32(rng, 3, 8, 8, 32)), dev(r32(rng, 3, 8, 8, 32))
    gm, bt = dev(np.array([1.2])), dev(np.array([0.1]))
    st = ops.instnorm_stats(x, gm, bt, 0, 0.3)
    y16 = torch.empty(x.shape, dtype=torch.bfloat16, device="cuda")
    y = ops.instnorm_apply(x, st, None, 0, 1, 0.3, out16=y16)
    assert torch.equal(y16, y.to(torch.bfloat16))
    d16 = torch.empty(x.shape, dtype=torch.bfloat16, device="cuda")
    d = ops.instnorm_bwd(x, st, g, None, None, 0, 1, 0.3, out16=d16)
    assert torch.equal(d16, d.to(torch.bfloat16))


def _bf16_round(a):
    """the values a bf16 RNE rounding of the fp32 array keeps, as float64"""
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32).to(torch.bfloat16).double().numpy()


@pytest.mark.parametrize("case", [(2, 16, 16, 32), (1, 32, 48, 32), (2, 16, 32, 64)])
def test_n3_tap_product_kernels_from_bf16_mirror(ops, case):
    """bf16 path of the 3-channel layers (n3_pgemm.hip): they read ONLY the bf16 mirror of the wide operand and the
    pack's fp32 weights rounded to bf16; products are exact and accumulate in fp32, so against the oracle on the SAME
    rounded operands the error is an fp32-accumulation error, not a bf16 one."""
    B, H, W, C = case
    rng = np.random.default_rng(zlib_crc(case))
    x = r32(rng, B, H, W, C)
    w = r32(rng, 5, 5, 3, C, scale=0.05)
    b = r32(rng, 3, scale=0.1)
    dpre = r32(rng, B, H, W, 3)
    assert ops.n3_m16_supported(H, W, 3, C, 1) and not ops.n3_m16_supported(H, W, 3, C, 0)
    assert not ops.n3_m16_supported(H + 2, W, 3, C, 1)
    pack = ops.conv_pack(dev(w), 3, C, 1)
    x16 = dev(x).to(torch.bfloat16)
    xr, wr = _bf16_round(x), _bf16_round(w)
    # final layer forward: tanh(convT_s1(x) + b), x given as mirror only
    y = ops.convT_s1_tanh_fwd(None, pack, dev(b), 3, 1, x16=x16)
    assert rel(y, np.tanh(O.conv2d_transpose(xr, wr, b, 1))) < 3e-5
    # final layer backward: weight gradient from the mirror, data gradient written as bf16
    dx_e, _, db_e = O.conv2d_transpose_bwd(xr, w, dpre, 1)
    dw_e = O.conv2d_transpose_bwd(xr, w, _bf16_round(dpre), 1)[1]  # the bf16 weight-gradient MFMA rounds BOTH operands
    dx16 = torch.empty(B, H, W, C, dtype=torch.bfloat16, device="cuda")
    dw = torch.empty(5, 5, 3, C, device="cuda")
    db = torch.empty(3, device="cuda")
    ops.convT_s1_tanh_bwd(None, dev(dpre), pack, C, 1, dx16=dx16, dw=dw, db=db, x16=x16)
    assert rel(dw, dw_e) < 3e-5 and rel(db, db_e) < 3e-5
    assert rel(dx16.float(), dx_e) < TOL[1]
    # Encoder.conv1 gradients: dz [B,H,W,C] given as mirror only; image x3 [B,2H,2W,3] fp32
    x3 = r32(rng, B, 2 * H, 2 * W, 3)
    w1 = r32(rng, 5, 5, 3, C, scale=0.05)
    dz = r32(rng, B, H, W, C)
    pack1 = ops.conv_pack(dev(w1), 3, C, 1)
    dz16 = dev(dz).to(torch.bfloat16)
    dzr, w1r = _bf16_round(dz), _bf16_round(w1)
    dimg = ops.conv2d_s2_dgrad(None, pack1, 3, 1, dy16=dz16)
    assert rel(dimg, O.conv2d_bwd(x3, w1r, dzr, 2)[0]) < 3e-5
    dw1 = torch.empty(5, 5, 3, C, device="cuda")
    ops.conv2d_s2_wgrad(dev(x3), None, dw1, False, 1, dy16=dz16)
    assert rel(dw1, O.conv2d_bwd(_bf16_round(x3), w1, dzr, 2)[1]) < 3e-5


@pytest.mark.parametrize("case", [(2, 16, 16, 32), (1, 80, 48, 32), (1, 144, 16, 32), (2, 16, 32, 32), (1, 80, 64, 32), (1, 48, 96, 32), (3, 80, 160, 32)])
def test_final_layer_normalises_while_staging(ops, case):
    """lg_convT_s1_tanh_fwd_z16 (n3_rows.hip): InstanceNorm + LeakyReLU applied to the raw bf16 conv output while the final
    layer stages it == the stand-alone apply pass followed by the same layer, bit for bit; and both match the oracle on the
    rounded operands.  Shapes cover one and several row blocks per strip (64 rows each) and a ragged last one; W % 32 == 0 takes the
    pixel-pair form of the kernel (two output pixels per MFMA row, 32-column strips: one strip, partly idle blocks, two blocks per row)."""
    B, H, W, C = case
    rng = np.random.default_rng(zlib_crc(case) + 1)
    z = dev(r32(rng, B, H, W, C, scale=1.3) + 0.2)
    w, b = r32(rng, 5, 5, 3, C, scale=0.05), r32(rng, 3, scale=0.1)
    gm, bt = dev(np.array([1.1], dtype=np.float32)), dev(np.array([-0.15], dtype=np.float32))
    alpha = 0.3
    z16 = torch.empty_like(z, dtype=torch.bfloat16)
    st = ops.instnorm_stats(z, gm, bt, 0, alpha, x16_out=z16)
    h16 = torch.empty_like(z16)
    ops.instnorm_apply(z16, st, None, 0, 1, alpha, out16=h16, want_f32=False)
    pack = ops.conv_pack(dev(w), 3, C, 1)
    assert ops.convT_s1_tanh_fwd_z16_supported(H, W, 3, C, 1) and not ops.convT_s1_tanh_fwd_z16_supported(H, W + 8, 3, C, 1)
    y_ref = ops.convT_s1_tanh_fwd(None, pack, dev(b), 3, 1, x16=h16)
    y = ops.convT_s1_tanh_fwd_z16(z16, st, alpha, pack, dev(b), 3, 1)
    assert torch.equal(y, y_ref)
    exp = np.tanh(O.conv2d_transpose(h16.double().cpu().numpy(), _bf16_round(w), b, 1))
    assert rel(y, exp) < 3e-5


@pytest.mark.parametrize("case", [(2, 8, 16, 32, 128), (3, 16, 16, 64, 128), (4, 8, 16, 64, 256), (2, 16, 16, 128, 384), (5, 16, 32, 32, 128),
                                  (64, 32, 32, 64, 128)])
def test_down_conv_normalises_while_staging(ops, case):
    """lg_conv2d_s2_fwd_stats_zn (NORM form of conv_down3.hip): InstanceNorm + LeakyReLU applied to the raw bf16 conv output of
    the level below while the stride-2 conv stages its halo == the stand-alone apply pass followed by the same conv, bit for bit
    (result and moments); and against the oracle on the rounded operands.  8 x 16 tiles with one and several tiles per sample,
    odd batch, several column tiles (N = 256, 384), a batch with more items than persistent blocks.  Samples get different
    statistics (per-sample offsets and scales)."""
    B, Hm, Wm, Cs, N = case
    rng = np.random.default_rng(zlib_crc(case) + 5)
    zin = r32(rng, B, 2 * Hm, 2 * Wm, Cs, scale=1.0)
    zin = zin * (0.5 + rng.random((B, 1, 1, 1), dtype=np.float32) * 2.0) + rng.standard_normal((B, 1, 1, 1)).astype(np.float32)
    w, b = r32(rng, 5, 5, Cs, N, scale=0.05), r32(rng, N, scale=0.1)
    gm, bt = dev(np.array([0.9], dtype=np.float32)), dev(np.array([0.2], dtype=np.float32))
    gm2, bt2 = dev(np.array([1.2], dtype=np.float32)), dev(np.array([-0.1], dtype=np.float32))
    alpha = 0.3
    zin16 = torch.empty(B, 2 * Hm, 2 * Wm, Cs, dtype=torch.bfloat16, device="cuda")
    st_in = ops.instnorm_stats(dev(zin), gm, bt, 0, alpha, x16_out=zin16)
    h16 = torch.empty_like(zin16)
    ops.instnorm_apply(zin16, st_in, None, 0, 1, alpha, out16=h16, want_f32=False)
    pack = ops.conv_pack(dev(w), Cs, N, 1)
    assert ops.conv2d_s2_fwd_stats_zn_supported(B, 2 * Hm, 2 * Wm, Cs, N, 1)
    z_ref, st_ref = ops.conv2d_s2_fwd_stats(None, pack, dev(b), N, 1, gm2, bt2, x16=h16, z16=True, alpha=alpha)
    assert ops.last_kernel().startswith("conv_down3_kernel")
    z, st = ops.conv2d_s2_fwd_stats_zn(zin16, st_in, alpha, pack, dev(b), N, 1, gm2, bt2)
    assert "NORM" in ops.last_kernel()
    assert torch.equal(z, z_ref) and torch.equal(st, ops.stats_tensor(st_ref))
    if B <= 8:
        exp = O.conv2d(h16.double().cpu().numpy(), _bf16_round(w), b, 2)
        assert rel(z.float(), exp) < TOL[1]   # bf16 result
        ef = exp.reshape(B, -1)
        assert rel(st[:, 0].double() + st[:, 4].double(), ef.mean(1)) < 2e-5 and rel(st[:, 1], ef.std(1)) < 2e-5


def test_down_conv_zn_declines_outside_its_tiling(ops):
    assert not ops.conv2d_s2_fwd_stats_zn_supported(2, 16, 32, 32, 64, 1)    # 64-column tiles: no normalising form
    assert not ops.conv2d_s2_fwd_stats_zn_supported(2, 12, 12, 32, 128, 1)   # map not tileable
    assert not ops.conv2d_s2_fwd_stats_zn_supported(2, 16, 16, 32, 128, 1)   # 8 x 8 result: sample-pair tiles, no normalising form (measured: loses)
    assert not ops.conv2d_s2_fwd_stats_zn_supported(2, 16, 32, 32, 128, 0)   # f32 path
    with pytest.raises(ValueError):
        ops.conv2d_s2_fwd_stats_zn(torch.zeros(2, 16, 32, 32, dtype=torch.bfloat16, device="cuda"), torch.zeros(2, 8, device="cuda"), 0.3,
                                   None, torch.zeros(64, device="cuda"), 64, 1, None, None)


@pytest.mark.parametrize("case", [(2, 8, 8, 32, 128), (4, 8, 8, 64, 256), (3, 8, 8, 32, 128), (1, 8, 16, 32, 64), (2, 16, 16, 32, 64),
                                  (1, 8, 16, 32, 128)])
def test_bf16_path_down_kernels_small_shapes(ops, case):
    """The persistent DOWN kernel (conv_down3.hip) in all its tilings at small shapes, through the bf16-activation entry
    point (bf16 source mirror -> bf16 z + fused moments): 8 x 8 maps with an even batch = sample-PAIR tiles, an odd batch falls
    back to conv_halo.hip; N = 64 = the 2 x 2-wave 64-column tiles; N = 128 / 256 = the default.  Against the oracle on the rounded
    operands; the moments against the fp32 values the kernel rounded."""
    B, Hm, Wm, Cs, N = case
    rng = np.random.default_rng(zlib_crc(case) + 7)
    x, w, b = r32(rng, B, 2 * Hm, 2 * Wm, Cs), r32(rng, 5, 5, Cs, N, scale=0.1), r32(rng, N, scale=0.2)
    gm, bt = dev(np.array([1.0], dtype=np.float32)), dev(np.array([0.0], dtype=np.float32))
    x16 = dev(x).to(torch.bfloat16)
    pack = ops.conv_pack(dev(w), Cs, N, 1)
    z16, st = ops.conv2d_s2_fwd_stats(None, pack, dev(b), N, 1, gm, bt, x16=x16, z16=True)
    assert z16.dtype == torch.bfloat16 and st is not None
    exp = O.conv2d(_bf16_round(x), _bf16_round(w), b, 2)
    assert rel(z16.float(), exp) < TOL[1]
    ef = exp.reshape(B, -1)
    assert rel(st[:, 0].double() + st[:, 4].double(), ef.mean(1)) < 2e-5 and rel(st[:, 1], ef.std(1)) < 2e-5
    # data gradient of the matching transposed conv (same DOWN contraction) with the norm-backward sums of the layer below
    g16, np_ = ops.convT_s2_dgrad(None, pack, N, 1, dy16=x16, out_bf16=True, fuse=(z16, st, 0.3))
    assert rel(g16.float(), O.conv2d(_bf16_round(x), _bf16_round(w), np.zeros(N), 2)) < TOL[1]
    # the fused sums feed the norm backward: same dz / dgamma / dbeta as the stand-alone first pass
    outs = []
    for parts in (np_, None):
        dgm, dbt = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
        dz16 = torch.empty_like(z16)
        ops.instnorm_bwd(z16, st, g16, dgm, dbt, 0, 1, 0.3, out16=dz16, want_f32=False, partials=parts)
        outs.append((dz16.float().cpu().numpy(), dgm.item(), dbt.item()))
    assert np_ is not None or (B % 2 == 1 and Hm == 8 and Wm == 8)   # only the conv_halo fallback has no fused sums
    assert np.abs(outs[0][0] - outs[1][0]).max() <= 2e-2 * np.abs(outs[1][0]).max()  # (bf16 output: one rounding step)
    assert abs(outs[0][1] - outs[1][1]) < 1e-4 * (1 + abs(outs[1][1])) and abs(outs[0][2] - outs[1][2]) < 1e-4 * (1 + abs(outs[1][2]))


@pytest.mark.parametrize("case", [(2, 8, 16, 64, 128), (1, 16, 16, 128, 256), (2, 8, 16, 128, 64), (2, 8, 16, 64, 32), (3, 16, 32, 64, 32), (5, 8, 16, 64, 32),
                                  (2, 8, 8, 64, 128), (4, 8, 8, 128, 256), (6, 8, 8, 64, 128), (3, 8, 8, 64, 128)])
def test_bf16_path_up_kernels_small_shapes(ops, case):
    """The persistent UP kernels at small shapes through the bf16-activation entry points: conv_up4.hip (N % 128 == 0: blocks
    bound to one parity class) and conv_up3.hip ((128, 64) and (64, 32)): transposed-conv forward with fused moments, and the
    conv data gradient (same contraction, no bias) with the norm-backward sums of the layer below."""
    B, Hs, Ws, Cs, N = case
    rng = np.random.default_rng(zlib_crc(case) + 9)
    x, w, b = r32(rng, B, Hs, Ws, Cs), r32(rng, 5, 5, N, Cs, scale=0.1), r32(rng, N, scale=0.2)
    gm, bt = dev(np.array([1.0], dtype=np.float32)), dev(np.array([0.0], dtype=np.float32))
    x16 = dev(x).to(torch.bfloat16)
    pack = ops.conv_pack(dev(w), N, Cs, 1)
    z16, st = ops.convT_s2_fwd_stats(None, pack, dev(b), N, 1, gm, bt, x16=x16, z16=True)
    assert z16.dtype == torch.bfloat16 and st is not None
    exp = O.conv2d_transpose(_bf16_round(x), _bf16_round(w), b, 2)
    assert rel(z16.float(), exp) < TOL[1]
    ef = exp.reshape(B, -1)
    assert rel(st[:, 0].double() + st[:, 4].double(), ef.mean(1)) < 2e-5 and rel(st[:, 1], ef.std(1)) < 2e-5
    g16, np_ = ops.conv2d_s2_dgrad(None, pack, N, 1, dy16=x16, out_bf16=True, fuse=(z16, st, 0.3))
    # (round 3: the (64, 32) layer runs one tile per step on 4-wave workgroups and produces the sums too; 8 x 8 maps with an even
    #  batch = sample-PAIR tiles of conv_up4.hip, an odd batch falls back to conv_halo.hip, the only kernel without fused sums)
    assert (np_ is not None) or (Hs == 8 and Ws == 8 and B % 2 == 1)
    assert rel(g16.float(), O.conv2d_transpose(_bf16_round(x), _bf16_round(w), np.zeros(N), 2)) < TOL[1]
    outs = []
    for parts in (np_, None):
        dgm, dbt = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
        dz16 = torch.empty_like(z16)
        ops.instnorm_bwd(z16, st, g16, dgm, dbt, 0, 1, 0.3, out16=dz16, want_f32=False, partials=parts)
        outs.append((dz16.float().cpu().numpy(), dgm.item(), dbt.item()))
    assert np.abs(outs[0][0] - outs[1][0]).max() <= 2e-2 * np.abs(outs[1][0]).max()
    assert abs(outs[0][1] - outs[1][1]) < 1e-4 * (1 + abs(outs[1][1])) and abs(outs[0][2] - outs[1][2]) < 1e-4 * (1 + abs(outs[1][2]))


@pytest.mark.parametrize("case", [(2, 16, 16, 32, 64), (2, 4, 32, 32, 64), (2, 8, 8, 64, 128), (3, 8, 8, 64, 128), (1, 16, 32, 64, 64)])
def test_all_taps_weight_gradient_small_shapes(ops, case):
    """wgrad_at.hip in its three tilings (16 x 8 strips, 32 x 4 strips, 8 x 8 sample pairs; an odd batch of 8 x 8 maps falls
    back to the per-tap kernel) from the bf16 mirrors, overwrite and accumulate, against the oracle on the rounded operands."""
    B, Hm, Wm, cb, cs = case
    rng = np.random.default_rng(zlib_crc(case) + 3)
    big, small = r32(rng, B, 2 * Hm, 2 * Wm, cb), r32(rng, B, Hm, Wm, cs)
    b16, s16 = dev(big).to(torch.bfloat16), dev(small).to(torch.bfloat16)
    exp = O.conv2d_bwd(_bf16_round(big), np.zeros((5, 5, cb, cs)), _bf16_round(small), 2)[1]
    dw = torch.full((5, 5, cb, cs), 7.0, device="cuda")
    ops.conv2d_s2_wgrad(None, None, dw, False, 1, x16=b16, dy16=s16)
    assert rel(dw, exp) < 3e-5
    pre = r32(rng, 5, 5, cb, cs)
    dw2 = dev(pre)
    ops.conv2d_s2_wgrad(None, None, dw2, True, 1, x16=b16, dy16=s16)
    assert rel(dw2, pre.astype(np.float64) + exp) < 3e-5
    # the transposed-conv form: roles of the operands swapped at the call, same contraction
    dw3 = torch.empty(5, 5, cb, cs, device="cuda")
    ops.convT_s2_wgrad(None, None, dw3, False, 1, x16=s16, dy16=b16)
    assert torch.equal(dw3, dw)


def zlib_crc(case):
    import zlib
    return zlib.crc32(repr(case).encode())


def test_norm_apply_mirror_only(ops):
    rng = np.random.default_rng(4)
    x = dev(r32(rng, 2, 8, 8, 32))
    gm, bt = dev(np.array([0.9])), dev(np.array([-0.2]))
    st = ops.instnorm_stats(x, gm, bt, 0, 0.3)
    y16a = torch.empty(x.shape, dtype=torch.bfloat16, device="cuda")
    y16b = torch.empty(x.shape, dtype=torch.bfloat16, device="cuda")
    y = ops.instnorm_apply(x, st, None, 0, 1, 0.3, out16=y16a)
    assert ops.instnorm_apply(x, st, None, 0, 1, 0.3, out16=y16b, want_f32=False) is None
    assert torch.equal(y16a, y16b) and torch.equal(y16a, y.to(torch.bfloat16))


@pytest.mark.parametrize("case", [(2, 16, 16), (1, 32, 48), (3, 16, 32)])
def test_n3_patch_kernels_bf16(ops, case):
    """bf16 path of the layers whose SOURCE is the 3-channel tensor (patch_p16_kernel): Encoder.conv1 forward with the
    fused InstanceNorm moments, and the data gradient of the final stride-1 layer (fp32 and bf16 output).  Against
    the oracle on the same bf16-rounded operands the error is fp32 accumulation only."""
    B, H, W = case
    rng = np.random.default_rng(zlib_crc(case))
    # conv1: img [B,2H,2W,3] -> z [B,H,W,64]
    img = r32(rng, B, 2 * H, 2 * W, 3)
    w1 = r32(rng, 5, 5, 3, 64, scale=0.1)
    b1 = r32(rng, 64, scale=0.1)
    pack1 = ops.conv_pack(dev(w1), 3, 64, 1)
    gm, bt = dev(np.array([1.1])), dev(np.array([0.05]))
    z, st = ops.conv2d_s2_fwd_stats(dev(img), pack1, dev(b1), 64, 1, gm, bt)
    z_e = O.conv2d(_bf16_round(img), _bf16_round(w1), b1, 2)
    assert rel(z, z_e) < 3e-5
    assert st is not None, "the patch kernel must deliver the moments (no separate statistics pass)"
    zf = z.double().reshape(B, -1)
    assert rel(st[:, 0], zf.mean(1).cpu().numpy()) < 1e-6 and rel(st[:, 1], zf.std(1, unbiased=False).cpu().numpy()) < 1e-6
    # final layer data gradient: dpre [B,H,W,3] -> dx [B,H,W,32]
    w = r32(rng, 5, 5, 3, 32, scale=0.05)
    dpre = r32(rng, B, H, W, 3)
    pack = ops.conv_pack(dev(w), 3, 32, 1)
    dx_e = O.conv2d_transpose_bwd(np.zeros((B, H, W, 32)), _bf16_round(w), _bf16_round(dpre), 1)[0]
    dx = torch.empty(B, H, W, 32, device="cuda")
    ops.convT_s1_tanh_bwd(None, dev(dpre), pack, 32, 1, dx=dx)
    assert rel(dx, dx_e) < 3e-5
    dx16 = torch.empty(B, H, W, 32, dtype=torch.bfloat16, device="cuda")
    ops.convT_s1_tanh_bwd(None, dev(dpre), pack, 32, 1, dx16=dx16)
    assert torch.equal(dx16, dx.to(torch.bfloat16))


@pytest.mark.parametrize("shape", [(3, 4, 4, 32), (2, 8, 8, 384), (5, 16, 16, 64), (2, 6, 10, 128)])
@pytest.mark.parametrize("g16", [False, True])
def test_instnorm_bwd_fused_bias_column_sums(ops, shape, g16):
    """db of lg_instnorm_leaky_bwd_db = column sums of the dx it writes (the conv bias gradient), same pass."""
    rng = np.random.default_rng(zlib_crc(shape))
    x, g = dev(r32(rng, *shape) * 1.3 + 0.2), dev(r32(rng, *shape))
    if g16:
        g = g.to(torch.bfloat16)
    gm, bt = dev(np.array([1.2])), dev(np.array([0.1]))
    st = ops.instnorm_stats(x, gm, bt, 0, 0.3)
    C = shape[-1]
    db = torch.full((C,), 7.0, device="cuda")
    d16 = torch.empty(x.shape, dtype=torch.bfloat16, device="cuda")
    dx = ops.instnorm_bwd(x, st, g, None, None, 0, 1, 0.3, out16=d16, db=db)
    dx_plain = ops.instnorm_bwd(x, st, g, None, None, 0, 1, 0.3)
    assert torch.equal(dx, dx_plain) and torch.equal(d16, dx.to(torch.bfloat16))
    exp = dx.double().reshape(-1, C).sum(0).cpu().numpy()
    scale = np.abs(dx.double().cpu().numpy()).reshape(-1, C).sum(0).max()   # the sums cancel to ~0: compare to the mass
    assert np.abs(db.cpu().numpy() - exp).max() < 2e-6 * scale
    # mirror-only output (no fp32 dx) gives the same sums
    db2 = torch.empty(C, device="cuda")
    ops.instnorm_bwd(x, st, g, None, None, 0, 1, 0.3, out16=d16, want_f32=False, db=db2)
    assert torch.equal(db, db2)


@pytest.mark.parametrize("B,K,N", [(3, 7, 64), (5, 133, 24576), (2, 40, 1024)])
def test_dense_dgrad(ops, B, K, N):
    rng = np.random.default_rng(B * 1000 + K)
    dy, w = r32(rng, B, N), r32(rng, K, N, scale=0.1)
    dx = ops.dense_dgrad(dev(dy), dev(w))
    assert rel(dx, dy @ w.T) < 3e-5


@pytest.mark.parametrize("B,ka,kc", [(1, 3, 5), (7, 93, 40), (256, 93, 40), (512, 11, 7)])
def test_step_input_forming_kernels(ops, B, ka, kc):
    """lg_concat_cols = keras concatenate([noise, cond], -1) (model.py:97-98); lg_adj_conditions = tf.concat([c2, c1], 0) and
    (that + 1) * 0.5 (eager_trainer.py:153-154): exact copies / one fp32 add and multiply, so the comparison is bit for bit."""
    rng = np.random.default_rng(B * 100 + ka)
    a, c, c1 = r32(rng, B, ka), r32(rng, B, kc), r32(rng, B, kc)
    out = ops.concat_cols(dev(a), dev(c))
    assert np.array_equal(out.cpu().numpy(), np.concatenate([a, c], -1))
    t, u = ops.adj_conditions(dev(c), dev(c1))
    t_ref = np.concatenate([c, c1], 0).astype(np.float32)
    assert np.array_equal(t.cpu().numpy(), t_ref)
    assert np.array_equal(u.cpu().numpy(), (t_ref + np.float32(1.0)) * np.float32(0.5))   # fp32 add, fp32 multiply


@pytest.mark.parametrize("case", [(4, 8, 16, 64, 128, False), (3, 16, 16, 64, 128, True), (2, 8, 16, 128, 64, True), (6, 8, 8, 64, 128, False)])
def test_deferred_moments_finished_by_the_apply_launch(ops, case):
    """lg_instnorm_leaky_apply_z16_p (finalize + apply in one launch; with a bf16 skip over two row ranges as the Adjuster's
    decoder calls it) against the two-call form on the same conv output: statistics records and activated maps bit-identical."""
    B, Hs, Ws, Cs, N, with_skip = case
    rng = np.random.default_rng(zlib_crc(case) + 13)
    x, w, b = r32(rng, B, Hs, Ws, Cs), r32(rng, 5, 5, N, Cs, scale=0.1), r32(rng, N, scale=0.2)
    gm, bt = dev(np.array([1.2], dtype=np.float32)), dev(np.array([-0.1], dtype=np.float32))
    x16 = dev(x).to(torch.bfloat16)
    pack = ops.conv_pack(dev(w), N, Cs, 1)
    z_a, st_a = ops.convT_s2_fwd_stats(None, pack, dev(b), N, 1, gm, bt, x16=x16, z16=True)
    skip = dev(r32(rng, *z_a.shape)).to(torch.bfloat16) if with_skip else None
    h_a = torch.empty_like(z_a)
    ops.instnorm_apply(z_a, st_a, skip, 0, 1, 0.3, out16=h_a, want_f32=False)
    z_b, mom = ops.convT_s2_fwd_stats(None, pack, dev(b), N, 1, gm, bt, x16=x16, z16=True, defer_stats=True)
    assert isinstance(mom, ops.Moments) and torch.equal(z_a, z_b)
    h_b = torch.empty_like(z_b)
    b1 = B // 2
    for lo, hi in ((0, b1), (b1, B)):
        ops.instnorm_apply(z_b[lo:hi], mom[lo:hi], None if skip is None else skip[lo:hi], 0, 1, 0.3, out16=h_b[lo:hi], want_f32=False)
    assert mom.covered == B
    assert torch.equal(ops.stats_tensor(mom), st_a) and torch.equal(h_a, h_b)
    # moments nobody applied: the stand-alone finalize
    _, mom2 = ops.convT_s2_fwd_stats(None, pack, dev(b), N, 1, gm, bt, x16=x16, z16=True, defer_stats=True)
    assert torch.equal(ops.stats_tensor(mom2), st_a)


def test_f32_dtype_refuses_bf16_only_operands(ops):
    """dtype f32 with only the bf16 mirror of the source (or a bf16 destination) is refused on both sides of the boundary: the exact-f32
    kernels read the fp32 tensor, and a null fp32 pointer would be dereferenced on the device (found by scripts/bench_layer.py on the f32
    path, round 3: 'Memory access fault by GPU').  The Python wrapper raises; the C entry point returns an error without launching."""
    from littlegan_amd import _lib
    B, H, cb, cs = 2, 8, 32, 64
    w = torch.randn(5, 5, cb, cs, device="cuda") * 0.05
    pack = ops.conv_pack(w, cb, cs, 0)
    dz16 = torch.randn(B, 2 * H, 2 * H, cb, device="cuda").to(torch.bfloat16)
    g16 = torch.randn(B, H, H, cs, device="cuda").to(torch.bfloat16)
    with pytest.raises(ValueError):
        ops.convT_s2_dgrad(None, pack, cs, 0, dy16=dz16, out_bf16=True)
    with pytest.raises(ValueError):
        ops.conv2d_s2_dgrad(None, pack, cb, 0, dy16=g16, out_bf16=True)
    lib = _lib.load()
    out16 = torch.empty(B, H, H, cs, dtype=torch.bfloat16, device="cuda")
    rc = lib.lg_convT_s2_dgrad_m16(0, dz16.data_ptr(), pack.data_ptr(), 0, out16.data_ptr(), B, H, H, cb, cs, 0, 0)
    assert rc != 0 and b"fp32 source" in lib.lg_last_error()
    torch.cuda.synchronize()   # nothing was launched: the device is healthy
    assert torch.isfinite(ops.conv2d_s2_fwd(torch.randn(B, 2 * H, 2 * H, cb, device="cuda"), pack, torch.zeros(cs, device="cuda"), cs, 0)).all()
