"""Deterministic synthetic video: gradient + moving textured rectangles + noise."""
import numpy as np


def synth_frames(width, height, nframes, seed=0x264, scene_len=97):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width]
    vel = [(3, 1), (-2, 2), (5, -3)]
    frames = []
    scene = None
    for n in range(nframes):
        if n % scene_len == 0:
            scene = dict(g=rng.integers(0, 4), tex=[rng.integers(0, 256, (height // 3, width // 3)).astype(np.int32) for _ in vel],
                         pos=[(int(rng.integers(0, width)), int(rng.integers(0, height))) for _ in vel],
                         base=int(rng.integers(60, 160)))
        t = n % scene_len
        y = (scene["base"] + (xx * 60 // width if scene["g"] & 1 else 0 * xx) + (yy * 50 // height if scene["g"] & 2 else 20)).astype(np.int32)
        u = np.full((height // 2, width // 2), 118 + 5 * scene["g"], np.int32)
        v = np.full((height // 2, width // 2), 134 - 4 * scene["g"], np.int32)
        for (vx, vy), tex, (px, py) in zip(vel, scene["tex"], scene["pos"]):
            th, tw = tex.shape
            x0, y0 = (px + vx * t) % width, (py + vy * t) % height
            ys = (np.arange(th) + y0) % height
            xs = (np.arange(tw) + x0) % width
            y[np.ix_(ys, xs)] = 40 + tex * 150 // 255
            u[np.ix_(ys[::2] // 2, xs[::2] // 2)] = 100 + tex[::2, ::2] * 40 // 255
            v[np.ix_(ys[::2] // 2, xs[::2] // 2)] = 150 - tex[::2, ::2] * 40 // 255
        y = np.clip(y + rng.integers(-4, 5, y.shape), 16, 235).astype(np.uint8)
        u = np.clip(u + rng.integers(-2, 3, u.shape), 16, 240).astype(np.uint8)
        v = np.clip(v + rng.integers(-2, 3, v.shape), 16, 240).astype(np.uint8)
        frames.append(np.concatenate([y.ravel(), u.ravel(), v.ravel()]))
    return frames


def psnr(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    m = float((d * d).mean())
    return 99.0 if m == 0 else 10 * np.log10(255 * 255 / m)
