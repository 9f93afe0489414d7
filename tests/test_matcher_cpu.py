"""CPU: oracle restatement of ORBmatcher::SearchByProjection (oracle/matcher.hpp) against domain properties."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as O  # noqa: E402
import match_scene as S  # noqa: E402


def test_descriptor_distance_is_popcount():
    rng = np.random.default_rng(0)
    for _ in range(50):
        a, b = rng.integers(0, 256, 32).astype(np.uint8), rng.integers(0, 256, 32).astype(np.uint8)
        assert O.lib().orc_descriptor_distance(O._p(a), O._p(b)) == int(np.unpackbits(a ^ b).sum())


def test_stream_pair_matches_are_geometrically_consistent():
    from sindslam_amd.synth import SyntheticStream
    cam, sc, Tc, Tl, last, cur = S.stream_pair(SyntheticStream(seed=12345), 5)
    m, n = O.search_by_projection(cam, sc, Tc, Tl, last, cur, 15.0)
    assert n >= (m >= 0).sum() > 300                                         # plenty of real matches; overwritten unobserved points count twice
    i2 = np.nonzero(m >= 0)[0]; i = m[i2]
    assert last["valid"][i].all()
    X = last["x3Dw"][i].astype(np.float64); Xc = X @ Tc[:3, :3].T.astype(np.float64) + Tc[:3, 3]
    u, v = cam[0] * Xc[:, 0] / Xc[:, 2] + cam[2], cam[1] * Xc[:, 1] / Xc[:, 2] + cam[3]
    r = 15.0 * sc[last["octave"][i]]
    assert (np.abs(cur["un_xy"][i2, 0] - u) < r + 1e-3).all() and (np.abs(cur["un_xy"][i2, 1] - v) < r + 1e-3).all()   # inside the search window
    assert (np.abs(cur["octave"][i2] - last["octave"][i]) <= 1).all()                                                  # level window (neither forward nor backward)
    d = np.unpackbits(last["desc"][i] ^ cur["desc"][i2], axis=1).sum(1)
    assert (d <= 100).all()                                                                                            # TH_HIGH
    assert np.median(np.hypot(cur["un_xy"][i2, 0] - u, cur["un_xy"][i2, 1] - v)) < 4.0                                 # static scene points reproject onto themselves
    # orientation check off keeps a superset
    m2, n2 = O.search_by_projection(cam, sc, Tc, Tl, last, cur, 15.0, check_orientation=False)
    assert n2 >= n and ((m < 0) | (m == m2)).all()


def test_taken_keypoints_and_observation_rule():
    cam, sc, Tc, Tl, last, cur = S.stress_pair(3)
    m, n = O.search_by_projection(cam, sc, Tc, Tl, last, cur, 15.0, check_orientation=False)
    assert not (m[cur["taken"] > 0] >= 0).any()                              # keypoints holding an observed MapPoint are never reassigned
    assert n >= (m >= 0).sum() > 200                                         # re-assignments of unobserved points count twice (reference behaviour)
    # with every last point observed nothing can be overwritten: nmatches == distinct assignments
    last2 = dict(last); last2["has_obs"] = np.ones_like(last["has_obs"])
    m3, n3 = O.search_by_projection(cam, sc, Tc, Tl, last2, cur, 15.0, check_orientation=False)
    assert n3 == (m3 >= 0).sum()


def test_no_points():
    cam, sc, Tc, Tl, last, cur = S.stress_pair(4, n_last=10, n_cur=10)
    empty = {k: v[:0] for k, v in last.items()}
    m, n = O.search_by_projection(cam, sc, Tc, Tl, empty, cur, 15.0)
    assert n == 0 and (m == -1).all()
