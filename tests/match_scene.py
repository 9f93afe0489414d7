"""Test scenes for the projection matcher: real pairs from the synthetic stream (ground-truth poses) and a contention stress case.
Built with the ORACLE's ORB extractor / Frame steps (test infrastructure)."""
import numpy as np

import oracle_lib as O

SCALE = (1.2 ** np.arange(8)).astype(np.float32)


def _scale_factors():
    s = np.ones(8, np.float32)
    for i in range(1, 8):
        s[i] = np.float32(s[i - 1] * np.float32(1.2))       # ORBextractor.cc:420-426
    return s


def _tcw(stream, t):
    c, yaw = stream.pose(t); cs, sn = np.cos(yaw), np.sin(yaw)
    R = np.array([[cs, 0, sn], [0, 1, 0], [-sn, 0, cs]], np.float64)       # camera -> world
    T = np.eye(4, dtype=np.float64); T[:3, :3] = R.T; T[:3, 3] = -R.T @ c.astype(np.float64)
    return T.astype(np.float32)


def stream_pair(stream, t, seed=0, drop=0.05, obs=0.7):
    """(cam10, scale, Tcw_cur, Tcw_last, last, cur) for frames t-1 -> t of a SyntheticStream with TUM3-like intrinsics."""
    rng = np.random.default_rng(seed)
    bgr, depth = stream.frames(t - 1, 2)
    orb = O.ORBextractor(1500, 1.2, 8, 15, 5)
    cal = [stream.fx, stream.fy, stream.cx, stream.cy, 0, 0, 0, 0, 0, 40.0, 1.0 / stream.depth_factor]
    fr = []
    for k in range(2):
        kp, desc = orb.extract(O.bgr2gray(bgr[k]))
        fr.append((kp, desc, O.frame_post_orb(cal, kp["x"], kp["y"], depth[k])))
    (kl, dl, pl), (kc, dc, pc) = fr
    Tl, Tc = _tcw(stream, t - 1), _tcw(stream, t)
    z = pl["depth"]; fx, fy, cx, cy = [np.float32(v) for v in cal[:4]]
    xc = np.stack([(kl["x"] - cx) * z / fx, (kl["y"] - cy) * z / fy, z], 1).astype(np.float64)
    Rl, tl = Tl[:3, :3].astype(np.float64), Tl[:3, 3].astype(np.float64)
    xw = ((xc - tl) @ Rl).astype(np.float32)                                 # R^T (x - t)
    last = dict(x3Dw=xw, valid=((z > 0) & (rng.random(len(z)) > drop)).astype(np.uint8), has_obs=(rng.random(len(z)) < obs).astype(np.uint8),
                octave=kl["octave"].copy(), angle=kl["angle"].copy(), desc=dl)
    xw[z <= 0] = (0, 0, 1)
    cur = dict(un_xy=pc["keys_un"], octave=kc["octave"].copy(), angle=kc["angle"].copy(), u_right=pc["u_right"], desc=dc,
               grid_start=pc["grid_start"], grid_idx=pc["grid_idx"], taken=None)
    cam10 = np.array([cal[0], cal[1], cal[2], cal[3], 40.0, np.float32(40.0) / np.float32(cal[0]), *pc["bounds"]], np.float32)
    return cam10, _scale_factors(), Tc, Tl, last, cur


def stress_pair(seed, n_last=3000, n_cur=2500, n_codes=6):
    """Dense random points with a handful of distinct descriptors: many equal distances and heavily contended keypoints."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy, bf = 535.4, 539.2, 320.1, 247.6, 40.0
    codes = rng.integers(0, 256, (n_codes, 32)).astype(np.uint8)
    flip = lambda d: d ^ (rng.random(d.shape) < 0.02).astype(np.uint8) * rng.integers(1, 255, d.shape).astype(np.uint8)
    cur_xy = np.stack([rng.uniform(16, 624, n_cur), rng.uniform(16, 464, n_cur)], 1).astype(np.float32)
    depth = np.full((480, 640), 10000, np.uint16); depth[:, ::7] = 0
    cal = [fx, fy, cx, cy, 0, 0, 0, 0, 0, bf, 1.0 / 5000.0]
    pc = O.frame_post_orb(cal, cur_xy[:, 0], cur_xy[:, 1], depth)
    cur = dict(un_xy=pc["keys_un"], octave=rng.integers(0, 8, n_cur).astype(np.int32), angle=rng.uniform(0, 360, n_cur).astype(np.float32),
               u_right=pc["u_right"], desc=flip(codes[rng.integers(0, n_codes, n_cur)]), grid_start=pc["grid_start"], grid_idx=pc["grid_idx"],
               taken=(rng.random(n_cur) < 0.1).astype(np.uint8))
    z = rng.uniform(1.5, 2.5, n_last); u = rng.uniform(-20, 660, n_last); v = rng.uniform(-20, 500, n_last)
    xw = np.stack([(u - cx) * z / fx, (v - cy) * z / fy, z], 1).astype(np.float32)
    xw[::97, 2] *= -1                                                        # behind the camera
    last = dict(x3Dw=xw, valid=(rng.random(n_last) > 0.1).astype(np.uint8), has_obs=(rng.random(n_last) < 0.6).astype(np.uint8),
                octave=rng.integers(0, 8, n_last).astype(np.int32), angle=rng.uniform(0, 360, n_last).astype(np.float32),
                desc=flip(codes[rng.integers(0, n_codes, n_last)]))
    Tl = np.eye(4, dtype=np.float32); Tc = np.eye(4, dtype=np.float32)
    a = np.deg2rad(0.4); Tc[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32); Tc[:3, 3] = (0.004, -0.002, 0.003)
    cam10 = np.array([fx, fy, cx, cy, bf, np.float32(bf) / np.float32(fx), *pc["bounds"]], np.float32)
    return cam10, _scale_factors(), Tc, Tl, last, cur
