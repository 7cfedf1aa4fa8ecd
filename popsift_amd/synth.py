"""Deterministic synthetic grayscale inputs.

synth(seed, W, H) -> uint8 (H, W):
  (i)   uniform noise from numpy Generator(PCG64(seed)), Gaussian-filtered with
        sigma = 1.5 px, stretched to mean 128 / std 40;
  (ii)  plus K = round(W*H/1000) Gaussian blobs, centres uniform, std log-uniform
        in [1.5, 12] px, signed amplitude uniform in +-[20, 80];
  (iii) clip to [0, 255], round.
"""
import numpy as np


def _gauss1d(sigma):
    r = int(np.ceil(4.0 * sigma))
    x = np.arange(-r, r + 1, dtype=np.float64)
    k = np.exp(-0.5 * (x / sigma) ** 2)
    return k / k.sum()


def _blur(a, sigma):
    k = _gauss1d(sigma)
    r = len(k) // 2
    p = np.pad(a, ((0, 0), (r, r)), mode="edge")
    a = sum(k[i] * p[:, i:i + a.shape[1]] for i in range(len(k)))
    p = np.pad(a, ((r, r), (0, 0)), mode="edge")
    return sum(k[i] * p[i:i + a.shape[0], :] for i in range(len(k)))


def synth(seed, W, H):
    rng = np.random.Generator(np.random.PCG64(seed))
    img = _blur(rng.random((H, W)), 1.5)
    img = (img - img.mean()) / img.std() * 40.0 + 128.0
    K = int(round(W * H / 1000.0))
    cx = rng.random(K) * W
    cy = rng.random(K) * H
    sd = np.exp(rng.uniform(np.log(1.5), np.log(12.0), K))
    amp = rng.uniform(20.0, 80.0, K) * np.where(rng.random(K) < 0.5, -1.0, 1.0)
    for k in range(K):
        r = int(np.ceil(4.0 * sd[k]))
        x0, x1 = max(0, int(cx[k]) - r), min(W, int(cx[k]) + r + 1)
        y0, y1 = max(0, int(cy[k]) - r), min(H, int(cy[k]) + r + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        gx = np.exp(-0.5 * ((np.arange(x0, x1) - cx[k]) / sd[k]) ** 2)
        gy = np.exp(-0.5 * ((np.arange(y0, y1) - cy[k]) / sd[k]) ** 2)
        img[y0:y1, x0:x1] += amp[k] * np.outer(gy, gx)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def gaussian_blob(W, H, x0, y0, std, amp=100.0, bg=64.0, dtype=np.uint8):
    """Single isotropic Gaussian blob on a flat background (analytic KAT)."""
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    img = bg + amp * np.exp(-0.5 * (((xx - x0) ** 2 + (yy - y0) ** 2) / std ** 2))
    if dtype == np.uint8:
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return (img / 256.0).astype(np.float32)
