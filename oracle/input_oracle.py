"""TEST INFRASTRUCTURE ONLY — CPU restatement of the step's input side: the TF-1.15 image ops of
/root/reference/eager_trainer.py:127-131 as deterministic functions of their random draws, and the counter-based
generator (Philox4x32-10, Salmon et al. SC'11 / Random123) the device kernels use.  PARITY UNPINNED like the rest of
the oracle: TensorFlow is not available here; the op semantics are restated from TF 1.15's public behaviour
(adjust_brightness: x + delta; adjust_contrast: (x - mean_HW) * f + mean_HW per image and channel; AdjustHue: hue
rotation that preserves each pixel's min and max channel, valid on any value range).  Nothing under littlegan_amd/
imports this module."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    """ctr: 4 uint32, key: 2 uint32 -> 4 uint32 (pure-Python ints; small cases only)."""
    c = [int(v) & MASK for v in ctr]
    k0, k1 = int(key[0]) & MASK, int(key[1]) & MASK
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c[3] ^ k1) & MASK, p0 & MASK]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c


def philox_blocks(nblocks, seed, offset):
    out = np.empty((nblocks, 4), dtype=np.uint32)
    for i in range(nblocks):
        c = (offset + i) & ((1 << 64) - 1)
        out[i] = philox4x32_10([c & MASK, c >> 32, 0, 0], [seed & MASK, (seed >> 32) & MASK])
    return out


def u01(bits):
    return (np.float32(bits >> np.uint32(8)) + np.float32(1.0)) * np.float32(1.0 / 16777216.0)


def normals(nblocks, seed, offset):
    """[nblocks, 4] standard normals exactly as the device draws them (two Box-Muller pairs per block), float64 math."""
    b = philox_blocks(nblocks, seed, offset)
    u = u01(b).astype(np.float64)
    r0, r1 = np.sqrt(-2.0 * np.log(u[:, 0])), np.sqrt(-2.0 * np.log(u[:, 2]))
    t0, t1 = 2.0 * np.pi * u[:, 1], 2.0 * np.pi * u[:, 3]
    return np.stack([r0 * np.cos(t0), r0 * np.sin(t0), r1 * np.cos(t1), r1 * np.sin(t1)], 1)


def hue_rotate(img, dh):
    r, g, b = img[..., 0], img[..., 1], img[..., 2]
    vmax, vmin = img.max(-1), img.min(-1)
    rng = vmax - vmin
    safe = np.where(rng > 0, rng, 1.0)
    h = np.where(r == vmax, (g - b) / safe, np.where(g == vmax, 2.0 + (b - r) / safe, 4.0 + (r - g) / safe))
    h = np.mod(h + 6.0 * dh, 6.0)
    sect = np.floor(h).astype(int) % 6
    f = h - np.floor(h)
    up, dn = vmin + rng * f, vmax - rng * f
    tab = {0: (vmax, up, vmin), 1: (dn, vmax, vmin), 2: (vmin, vmax, up), 3: (vmin, dn, vmax), 4: (up, vmin, vmax),
           5: (vmax, vmin, dn)}
    out = np.empty_like(img)
    for k in range(3):
        out[..., k] = np.select([sect == s for s in range(6)], [tab[s][k] for s in range(6)])
    grey = ~(rng > 0)
    out[grey] = img[grey]
    return out


def augment(img, flip, db, cf, dh):
    """img [B,H,W,3] float64; flip [B] bool.  The noise term is added by the caller."""
    x = np.where(np.asarray(flip, bool)[:, None, None, None], img[:, :, ::-1, :], img) + db
    m = x.mean(axis=(1, 2), keepdims=True)
    x = (x - m) * cf + m
    return hue_rotate(x, dh) if dh != 0 else x


def step_draws(B, seed, offset, db_max=0.02, c_lo=0.75, c_hi=1.003, dh_max=0.03):
    """The scalar draws of eager_trainer.py:127-130 as lg_augment_drawn makes them: word w of the Philox window at
    `offset` -> u_w = (bits >> 8) / 2^24; returns (db, cf, dh, flip[B])."""
    nb = (B + 3 + 3) // 4
    u = (philox_blocks(nb, seed, offset).reshape(-1) >> np.uint32(8)).astype(np.float64) / 16777216.0
    return (2.0 * u[0] - 1.0) * db_max, c_lo + u[1] * (c_hi - c_lo), (2.0 * u[2] - 1.0) * dh_max, u[3:3 + B] < 0.5
