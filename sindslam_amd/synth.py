"""Synthetic TUM-shaped RGB-D stream.

Static scene: three textured fronto-parallel planes at z = 1.5 / 2.5 / 4.0 m plus a far opening at 8 m
(beyond DynaDetect's 6 m validity limit), seen by a camera on a 0.3 m Lissajous path with +-3 deg yaw;
two rigid ellipsoidal "walkers" (z in [1, 2] m) move independently.  Depth is z * depth_factor as u16 with
8x8 zero-hole blocks (5 %) and sigma = 0.002 z^2 noise.  Deterministic for a given seed; numpy only.

The real TUM / Bonn sequences named in BASELINE.json are not available offline; this generator stands in for
them with the same image size, intrinsics and depth factor (reference Examples/RGB-D/TUM3.yaml, Bonn.yaml).
"""
from __future__ import annotations

import numpy as np

TUM3 = dict(fx=535.4, fy=539.2, cx=320.1, cy=247.6, depth_factor=5000.0, ini_th=15, min_th=5)
BONN = dict(fx=542.822841, fy=542.576870, cx=315.593520, cy=237.756098, depth_factor=5000.0, ini_th=20, min_th=7)
D455 = dict(fx=390.2265625, fy=390.2265625, cx=328.57140625, cy=240.3284375, depth_factor=1000.0, ini_th=20, min_th=7)


def _texture(rng: np.random.Generator, size: int = 1024, octaves: int = 8) -> np.ndarray:
    """Band-limited noise, u8-range float32, mean 110, sigma 45."""
    acc = np.zeros((size, size), np.float32)
    for o in range(octaves):
        n = max(4, size >> (octaves - 1 - o))          # 8 .. size cells
        base = rng.standard_normal((n + 1, n + 1)).astype(np.float32)
        # bilinear upsample to size x size
        xs = np.linspace(0, n, size, endpoint=False, dtype=np.float32)
        i0 = np.floor(xs).astype(np.int32); f = xs - i0
        rows = base[i0] * (1 - f)[:, None] + base[i0 + 1] * f[:, None]
        up = rows[:, i0] * (1 - f)[None, :] + rows[:, i0 + 1] * f[None, :]
        acc += up * (0.55 ** (o * 0.5))
    acc -= acc.mean(); acc /= acc.std() + 1e-6
    return (110.0 + 45.0 * acc).astype(np.float32)


def _sample(tex: np.ndarray, x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """Bilinear, wrap-around."""
    n = tex.shape[0]
    x0 = np.floor(x); y0 = np.floor(y)
    fx = (x - x0).astype(np.float32); fy = (y - y0).astype(np.float32)
    x0 = x0.astype(np.int64) % n; y0 = y0.astype(np.int64) % n
    x1 = (x0 + 1) % n; y1 = (y0 + 1) % n
    return (tex[y0, x0] * (1 - fx) * (1 - fy) + tex[y0, x1] * fx * (1 - fy)
            + tex[y1, x0] * (1 - fx) * fy + tex[y1, x1] * fx * fy)


class SyntheticStream:
    def __init__(self, width: int = 640, height: int = 480, seed: int = 12345, intr: dict = TUM3,
                 motion_scale: float = 1.0):
        self.w, self.h = width, height
        s = width / 640.0
        self.fx, self.fy, self.cx, self.cy = intr["fx"] * s, intr["fy"] * s, intr["cx"] * s, intr["cy"] * s
        self.depth_factor = intr["depth_factor"]
        self.ini_th, self.min_th = intr["ini_th"], intr["min_th"]
        self.motion_scale = motion_scale
        rng = np.random.default_rng(seed)
        self.tex = [_texture(rng) for _ in range(6)]        # 4 static surfaces + 2 walkers
        self.tint = rng.uniform(0.85, 1.15, size=(6, 3)).astype(np.float32)
        self.hole_seed = seed + 1
        u, v = np.meshgrid(np.arange(width, dtype=np.float32), np.arange(height, dtype=np.float32))
        self.dx = (u - self.cx) / self.fx
        self.dy = (v - self.cy) / self.fy
        # walkers: phase offsets of their own Lissajous motion
        self.walk = [dict(z=1.3, a=0.33, b=0.62, px=0.35, py=0.25, fx_=0.071, fy_=0.043, ph=0.3),
                     dict(z=1.8, a=0.30, b=0.70, px=0.55, py=0.20, fx_=0.053, fy_=0.037, ph=1.9)]

    def pose(self, t: int):
        m = self.motion_scale
        cxw = 0.15 * m * np.sin(0.045 * t); cyw = 0.10 * m * np.sin(0.031 * t + 0.7); czw = 0.05 * m * np.sin(0.023 * t + 1.3)
        yaw = np.deg2rad(3.0) * m * np.sin(0.027 * t + 0.4)
        return np.array([cxw, cyw, czw], np.float32), np.float32(yaw)

    def frame(self, t: int):
        """-> (bgr u8 HxWx3, depth u16 HxW)"""
        c, yaw = self.pose(t)
        cs, sn = np.cos(yaw), np.sin(yaw)
        rx = cs * self.dx + sn; ry = self.dy; rz = -sn * self.dx + cs       # world ray = R_y(yaw) * (dx, dy, 1)
        gray = np.zeros((self.h, self.w), np.float32); zbuf = np.full((self.h, self.w), 1e9, np.float32)
        surf = np.zeros((self.h, self.w), np.int32)

        def plane(idx, Z, inside, texscale=220.0, ox=0.0, oy=0.0):
            tt = (Z - c[2]) / rz
            X = c[0] + tt * rx; Y = c[1] + tt * ry
            ok = inside(X, Y) & (tt > 0.1) & (tt < zbuf)
            g = _sample(self.tex[idx], (X - ox) * texscale + 300.0, (Y - oy) * texscale + 300.0)
            np.copyto(gray, g, where=ok); np.copyto(zbuf, tt.astype(np.float32), where=ok); surf[ok] = idx
            return X, Y

        plane(3, 8.0, lambda X, Y: np.ones_like(X, bool), 60.0)                       # far opening (invalid depth)
        plane(0, 4.0, lambda X, Y: ~((np.abs(X - 1.6) < 0.8) & (np.abs(Y + 0.6) < 0.6)), 120.0)   # back wall with a window
        plane(1, 2.5, lambda X, Y: X < -0.35, 170.0)                                   # mid plane on the left
        plane(2, 1.5, lambda X, Y: (Y > 0.28) & (X > -0.1), 240.0)                     # near plane bottom right
        m = self.motion_scale
        for k, wk in enumerate(self.walk):
            px = wk["px"] * np.sin(wk["fx_"] * m * t + wk["ph"]) * 1.2
            py = wk["py"] * np.sin(wk["fy_"] * m * t + 2 * wk["ph"]) * 0.5
            Z = wk["z"] + 0.1 * np.sin(0.05 * m * t + wk["ph"])
            tt = (Z - c[2]) / rz
            X = c[0] + tt * rx; Y = c[1] + tt * ry
            r2 = ((X - px) / wk["a"]) ** 2 + ((Y - py) / wk["b"]) ** 2
            bulge = 0.15 * np.sqrt(np.clip(1.0 - r2, 0.0, 1.0))
            zz = (tt - bulge).astype(np.float32)
            ok = (r2 <= 1.0) & (zz < zbuf)
            g = _sample(self.tex[4 + k], (X - px) * 260.0 + 500.0, (Y - py) * 260.0 + 500.0)
            np.copyto(gray, g, where=ok); np.copyto(zbuf, zz, where=ok); surf[ok] = 4 + k
        tint = self.tint[surf]                                                           # HxWx3
        bgr = np.clip(gray[..., None] * tint, 0, 255).astype(np.uint8)
        # depth: noise, holes, u16
        rng = np.random.default_rng(self.hole_seed * 7919 + t)
        z = zbuf + rng.standard_normal(zbuf.shape).astype(np.float32) * (0.002 * zbuf * zbuf)
        d = np.clip(np.rint(z * self.depth_factor), 0, 65535).astype(np.uint16)
        holes = rng.random((self.h // 8, self.w // 8)) < 0.05
        d[np.kron(holes, np.ones((8, 8), bool))[: self.h, : self.w]] = 0
        return np.ascontiguousarray(bgr), np.ascontiguousarray(d)

    def frames(self, start: int, count: int):
        b = np.empty((count, self.h, self.w, 3), np.uint8); d = np.empty((count, self.h, self.w), np.uint16)
        for i in range(count):
            b[i], d[i] = self.frame(start + i)
        return b, d
