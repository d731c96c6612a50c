"""Seeded synthetic imagery shared by the CPU and GPU tests."""
import numpy as np


def noise_image(w, h, seed=1, octaves=6, mean=110.0, sigma=40.0):
    """Band-limited noise texture (sum of bilinearly upsampled random octaves), uint8."""
    rng = np.random.RandomState(seed)
    acc = np.zeros((h, w), np.float64)
    amp = 1.0
    for o in range(octaves):
        gw, gh = max(2, w >> (octaves - o)), max(2, h >> (octaves - o))
        g = rng.randn(gh + 1, gw + 1)
        ys = np.linspace(0, gh - 1e-6, h)
        xs = np.linspace(0, gw - 1e-6, w)
        y0 = ys.astype(int)
        x0 = xs.astype(int)
        fy = (ys - y0)[:, None]
        fx = (xs - x0)[None, :]
        v = (g[y0][:, x0] * (1 - fy) * (1 - fx) + g[y0][:, x0 + 1] * (1 - fy) * fx + g[y0 + 1][:, x0] * fy * (1 - fx)
             + g[y0 + 1][:, x0 + 1] * fy * fx)
        acc += amp * v
        amp *= 0.75
    acc = (acc - acc.mean()) / acc.std()
    return np.clip(mean + sigma * acc, 0, 255).astype(np.uint8)


def warp_affine(img, M, t):
    """Sample img at (M @ [x,y] + t) bilinearly (inverse map), uint8 out."""
    h, w = img.shape
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    sx = M[0, 0] * xs + M[0, 1] * ys + t[0]
    sy = M[1, 0] * xs + M[1, 1] * ys + t[1]
    x0 = np.floor(sx).astype(int)
    y0 = np.floor(sy).astype(int)
    fx = sx - x0
    fy = sy - y0

    def tap(yy, xx):
        return img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.float64)

    v = tap(y0, x0) * (1 - fx) * (1 - fy) + tap(y0, x0 + 1) * fx * (1 - fy) + tap(y0 + 1, x0) * (1 - fx) * fy + tap(
        y0 + 1, x0 + 1) * fx * fy
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)
