"""Synthetic PNM inputs for parity tests and bench (SURVEY.md Appendix C generator).

Test-input synthesis only (no codec logic).  md5 sums of the produced files are pinned in
tests/golden/MANIFEST.json so a numpy RNG change cannot silently alter the inputs.
"""
import numpy as np


def synth_float(w, h, seed, shift=0):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    x = x + shift
    img = 128 + 60 * np.sin(x / 17.0) * np.cos(y / 23.0) + 40 * (((x // 32) + (y // 32)) % 2) \
        + rng.normal(0, 6, (h, w))
    img += 50 * np.exp(-((x - w * 0.3) ** 2 + (y - h * 0.6) ** 2) / (2 * (w / 10) ** 2))
    return img, x, y


def synth(w, h, seed, shift=0):
    img, _, _ = synth_float(w, h, seed, shift)
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_color_k(w, h, seed=1234, shift=0):
    """Smooth-chroma colour frame (k-generator of SURVEY Appendix C)."""
    L, x, y = synth_float(w, h, seed, shift)
    R = L + 24 * np.sin(x / 61.0)
    G = L
    B = L - 24 * np.cos(y / 47.0)
    return np.clip(np.stack([R, G, B], -1), 0, 255).astype(np.uint8)


def synth_color_c(w, h, f=0):
    r = synth(w, h, 10, 3 * f)
    g = synth(w, h, 11, 3 * f)[::-1].copy()
    b = synth(w, h, 12, 3 * f)[:, ::-1].copy()
    return np.stack([r, g, b], -1)


def noise(w=512, h=384, seed=7):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = 128 + 50 * np.sin(x * y / 3000) + 30 * np.sign(np.sin(x / 9 + y / 13)) + rng.normal(0, 20, (h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


def pgm_bytes(a):
    return b'P5\n%d %d\n255\n' % (a.shape[1], a.shape[0]) + a.tobytes()


def ppm_bytes(rgb):
    return b'P6\n%d %d\n255\n' % (rgb.shape[1], rgb.shape[0]) + rgb.tobytes()


def write_pgm(path, a):
    with open(path, 'wb') as f:
        f.write(pgm_bytes(a))


def write_ppm(path, rgb):
    with open(path, 'wb') as f:
        f.write(ppm_bytes(rgb))
