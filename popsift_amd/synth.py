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


# Six fixed homographies standing in for the Oxford affine-covariant sequences (boat / graffiti: zoom + rotation and
# viewpoint change of increasing strength; SURVEY.md section 8(d), BASELINE.json config 5).  They act on coordinates
# normalised to the image centre and half-diagonal, so one set serves every size.
def oxford_like_homographies():
    hs = []
    for k, (zoom, rot_deg, px, py, shear) in enumerate([(1.00, 0.0, 0.00, 0.00, 0.00), (1.12, 10.0, 0.02, 0.00, 0.03),
                                                         (1.30, 25.0, 0.05, 0.03, 0.06), (1.55, 40.0, -0.08, 0.05, 0.10),
                                                         (1.90, 60.0, 0.12, -0.08, 0.15), (2.40, 85.0, -0.18, 0.12, 0.22)]):
        a = np.deg2rad(rot_deg)
        R = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])
        S = np.array([[1.0 / zoom, shear, 0.0], [0.0, 1.0 / zoom, 0.0], [0.0, 0.0, 1.0]])
        P = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [px, py, 1.0]])
        hs.append(P @ R @ S)
    return hs


def warp(img, Hn):
    """Warp a uint8 image with the normalised homography Hn (destination -> source mapping), bilinear, edge clamp."""
    H, W = img.shape
    cx, cy, sc = (W - 1) / 2.0, (H - 1) / 2.0, 0.5 * np.hypot(W, H)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    u, v = (xx - cx) / sc, (yy - cy) / sc
    d = Hn[2, 0] * u + Hn[2, 1] * v + Hn[2, 2]
    sx = (Hn[0, 0] * u + Hn[0, 1] * v + Hn[0, 2]) / d * sc + cx
    sy = (Hn[1, 0] * u + Hn[1, 1] * v + Hn[1, 2]) / d * sc + cy
    sx = np.clip(sx, 0.0, W - 1.0)
    sy = np.clip(sy, 0.0, H - 1.0)
    x0 = np.minimum(np.floor(sx).astype(np.int64), W - 2)
    y0 = np.minimum(np.floor(sy).astype(np.int64), H - 2)
    fx, fy = sx - x0, sy - y0
    a = img.astype(np.float64)
    out = (a[y0, x0] * (1 - fx) + a[y0, x0 + 1] * fx) * (1 - fy) + (a[y0 + 1, x0] * (1 - fx) + a[y0 + 1, x0 + 1] * fx) * fy
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def oxford_like_stream(seeds=range(200, 212), W=800, H=640):
    """Config-5 stand-in: every seed's image under the six homographies -> list of (seed, k, image)."""
    hs = oxford_like_homographies()
    out = []
    for s in seeds:
        base = synth(s, W, H)
        for k, h in enumerate(hs):
            out.append((s, k, base if k == 0 else warp(base, h)))
    return out
