"""GPU: sind_match_by_projection (ORBmatcher::SearchByProjection, reference src/ORBmatcher.cc:1328-1470) against the oracle, exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _matcher(cam, sc, B, cap=4096, checkOri=True):
    from sindslam_amd.matcher import ORBmatcher
    return ORBmatcher(cam[0], cam[1], cam[2], cam[3], cam[4], cam[6:10], sc, checkOri=checkOri, cap=cap, max_batch=B)


def test_stream_pairs_batched(stream):
    import oracle_lib as O
    import match_scene as S
    scenes = [S.stream_pair(stream, t, seed=t) for t in (3, 4, 9)]
    cam, sc = scenes[0][0], scenes[0][1]
    for th in (15.0, 30.0):                                                  # th and the 2*th retry of Tracking.cc:922-928
        for ori in (True, False):
            mt = _matcher(cam, sc, len(scenes), checkOri=ori)
            got = mt.SearchByProjection([(Tc, Tl, last, cur) for _, _, Tc, Tl, last, cur in scenes], th)
            for (m, n), (c, s, Tc, Tl, last, cur) in zip(got, scenes):
                mo, no = O.search_by_projection(c, s, Tc, Tl, last, cur, th, check_orientation=ori)
                assert n == no and np.array_equal(m, mo)
                assert n > 300
            mt.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_contended_keypoints_equal_distances_and_taken_flags(seed):
    import oracle_lib as O
    import match_scene as S
    cam, sc, Tc, Tl, last, cur = S.stress_pair(seed)
    mt = _matcher(cam, sc, 1)
    for th in (7.0, 15.0, 40.0):
        (m, n), = mt.SearchByProjection([(Tc, Tl, last, cur)], th)
        mo, no = O.search_by_projection(cam, sc, Tc, Tl, last, cur, th)
        assert n == no and np.array_equal(m, mo)
    assert mt.last_rounds() >= 2                                             # the sequential dependence is really exercised
    mt.close()


def test_forward_and_backward_level_windows_and_mono():
    import oracle_lib as O
    import match_scene as S
    cam, sc, Tc, Tl, last, cur = S.stress_pair(11)
    mt = _matcher(cam, sc, 1)
    for dz, mono in ((0.5, False), (-0.5, False), (0.5, True)):              # |t_z| > mb = bf / fx: forward / backward search (ORBmatcher.cc:1348-1349)
        T = Tc.copy(); T[2, 3] = -dz
        last2 = dict(last); last2["x3Dw"] = last["x3Dw"] + np.array([0, 0, dz], np.float32)
        (m, n), = mt.SearchByProjection([(T, Tl, last2, cur)], 15.0, bMono=mono)
        mo, no = O.search_by_projection(cam, sc, T, Tl, last2, cur, 15.0, mono=mono)
        assert n == no and np.array_equal(m, mo) and n > 50
    mt.close()


def test_empty_inputs_and_errors():
    import match_scene as S
    from sindslam_amd import SindError
    cam, sc, Tc, Tl, last, cur = S.stress_pair(5, n_last=50, n_cur=40)
    mt = _matcher(cam, sc, 1, cap=64)
    empty_last = {k: v[:0] for k, v in last.items()}
    (m, n), = mt.SearchByProjection([(Tc, Tl, empty_last, cur)], 15.0)
    assert n == 0 and (m == -1).all()
    big = {k: np.concatenate([v, v]) for k, v in last.items()}
    with pytest.raises(SindError):
        mt.SearchByProjection([(Tc, Tl, big, cur)], 15.0)                    # 100 points > cap 64
    bad = dict(cur); bad["grid_idx"] = cur["grid_idx"].copy(); bad["grid_idx"][0] = 1000
    with pytest.raises(SindError):
        mt.SearchByProjection([(Tc, Tl, last, bad)], 15.0)
    mt.close()
