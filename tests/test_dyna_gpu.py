"""GPU parity: DynaDetect through the C ABI vs the CPU oracle, stage by stage and end to end over several frames."""
import numpy as np
import pytest

import oracle_lib as O
from sindslam_amd.synth import TUM3

pytestmark = pytest.mark.gpu


def iou(a, b):
    u = np.logical_or(a, b).sum()
    return 1.0 if u == 0 else np.logical_and(a, b).sum() / u


def test_detect_sequence(frames):
    from sindslam_amd.dyna import DynaDetect
    bgr, depth = frames
    K = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    ref = O.DynaDetect(bgr[1], bgr[0], *K)
    gpu = DynaDetect(bgr[1], bgr[0], *K)
    for t in range(2, 6):
        rd, rl = ref.detect(bgr[t], depth[t]); r = ref.debug()
        gd, gl = gpu.DetectDynaArea(bgr[t], depth[t], t); g = gpu.debug()
        # dense flow: FP32, bit-exact with the oracle (tolerance in SURVEY §8c: 1e-3 px)
        ff = np.stack([r["flow_full"][..., 0], r["flow_full"][..., 1]])
        assert np.array_equal(g["flow_full"].view(np.uint32), ff.view(np.uint32)), (t, np.abs(g["flow_full"] - ff).max())
        assert g["info"][0] == r["info"][0], "largeMotion flag"
        # homography (same specification on both sides), residual histogram, thresholds, masks
        assert g["info"][1] == r["info"][1]
        assert np.array_equal(g["H"], r["H"]), (g["H"], r["H"])
        assert np.array_equal(g["hist"], r["hist"])
        assert np.array_equal(g["thr"], r["thr"]), (g["thr"], r["thr"])
        assert np.array_equal(g["mask_low"], r["mask_low"]) and np.array_equal(g["mask_high"], r["mask_high"])
        # depth side: integer stages exact; the k-means centres are cv::kmeans' sequential FP32 sums, reproduced bit for bit (k_km_seqsum)
        assert np.array_equal(g["total_area"], r["total_area"])
        assert np.array_equal(g["kmeans_label"], r["kmeans_label"]), f"k-means labels differ on {(g['kmeans_label'] != r['kmeans_label']).mean():.2e} of the pixels"
        assert np.array_equal(g["centers"], r["centers"]), np.abs(g["centers"] - r["centers"]).max()
        assert np.array_equal(g["grad_edge"], r["grad_edge"]), "gradient edge after OPEN"
        assert np.array_equal(g["plane_contours"], r["plane_contours"]), "PEAC plane contours"
        assert np.array_equal(g["occ2"], r["occ2"]), "plane edges"
        assert np.array_equal(g["occ1"], r["occ1"]), "depth edges"
        assert g["info"][2] == r["info"][2], "number of pieces"
        # outputs: SURVEY §8c tolerance is mask IoU >= 0.99
        assert np.array_equal(gd, rd) and np.array_equal(gl, rl), (t, iou(gd == 255, rd == 255), (gl != rl).mean())
        assert set(np.unique(gd)) <= {0, 125, 255}
        # caller-side 15x15 dilation
        assert np.array_equal(gpu.dilate15(gd), O.dilate15(gd))
    gpu.close()


def test_requires_prime():
    import ctypes as C
    from sindslam_amd._lib import lib
    h = C.c_void_p()
    assert lib().sind_dyna_create(640, 480, C.c_float(500), C.c_float(500), C.c_float(320), C.c_float(240), C.c_float(5000), 0, C.byref(h)) == 0
    buf = np.zeros((480, 640, 3), np.uint8); dep = np.zeros((480, 640), np.uint16); out = np.zeros((480, 640), np.uint8)
    rc = lib().sind_dyna_detect(h, buf.ctypes.data_as(C.c_void_p), 0, dep.ctypes.data_as(C.c_void_p), 0, out.ctypes.data_as(C.c_void_p),
                                out.ctypes.data_as(C.c_void_p), 0)
    assert rc == -4 and b"prime" in lib().sind_last_error()
    lib().sind_dyna_destroy(h)


@pytest.mark.parametrize("w,n", [(640, 7), (640, 9), (200, 7), (64, 3), (130, 15)])
def test_bit_plane_dilation_equals_oracle_morphology(w, n):
    """k_dilate_planes (the pieces' 7x7 dilation of the region-adjacency stage, on 64-pixel words) against the oracle's cv::dilate,
    including widths that are not a multiple of 64 (tail word) and blobs touching every border"""
    import ctypes as C
    from sindslam_amd._lib import check, lib, ptr
    rng = np.random.default_rng(w + n); h, planes = 57, 3
    imgs = np.zeros((planes, h, w), np.uint8)
    for p in range(planes):
        imgs[p][rng.random((h, w)) < 0.01] = 255
        imgs[p, 0, :5] = 255; imgs[p, -1, -3:] = 255; imgs[p, 10:20, 0] = 255; imgs[p, 30:40, -1] = 255
    wpr = (w + 63) // 64
    packed = np.zeros((planes, h, wpr * 64), np.uint8); packed[:, :, :w] = imgs > 0
    words = np.packbits(packed.reshape(planes, h, wpr, 64), axis=-1, bitorder="little").view(np.uint64).reshape(planes, h, wpr).copy()
    out = np.zeros_like(words)
    check(lib().sind_debug_dilate_planes(ptr(words), planes, w, h, n, 0, ptr(out)), "sind_debug_dilate_planes")
    got = np.unpackbits(out.view(np.uint8).reshape(planes, h, wpr * 8), axis=-1, bitorder="little")[:, :, :w]
    for p in range(planes):
        assert np.array_equal(got[p] * 255, O.morph(imgs[p], n, "dilate")), (p, w, n)
    assert not np.unpackbits(out.view(np.uint8).reshape(planes, h, wpr * 8), axis=-1, bitorder="little")[:, :, w:].any()      # tail bits stay clear


def test_overlapped_lean_path_equals_the_serial_debug_path(frames):
    """the drop-in's default path (depth half of a frame beside its dense flow on a second stream and host thread, no debug copies, the large-motion candidate riding along as a
    batch of two) returns what the serial path with every debug copy returns -- over frames that do and do not take the large-motion pass -- and reports its stage times"""
    from sindslam_amd.dyna import DynaDetect
    from sindslam_amd.synth import SyntheticStream
    K = (TUM3["fx"], TUM3["fy"], TUM3["cx"], TUM3["cy"], TUM3["depth_factor"])
    flagged = calm = 0
    for bgr, depth in (SyntheticStream(seed=31, motion_scale=4.0).frames(0, 6), frames):          # fast motion: every frame takes the second pass; the fixture's stream: none does
        a = DynaDetect(bgr[1], bgr[0], *K, debug=False, overlap=True); b = DynaDetect(bgr[1], bgr[0], *K, debug=True, overlap=False)
        for t in range(2, 6):
            da, la = a.DetectDynaArea(bgr[t], depth[t], t); db, lb = b.DetectDynaArea(bgr[t], depth[t], t)
            assert np.array_equal(da, db) and np.array_equal(la, lb), t
            flagged += int(b.debug()["info"][0]); calm += int(not b.debug()["info"][0])
        st = a.timing(); b.close()
        if (bgr is not frames[0]): a.close()
    assert flagged >= 1 and calm >= 1
    assert st["calls"] == 4 and st["total"] > 0 and st["dense_flow"] > 0 and st["depth_half"] > 0 and abs(st["total"] - (st["upload"] + st["dense_flow"] + st["wait_depth_half"] + st["flow_masks_and_fusion"])) < 0.05 * st["total"] + 0.05
    print("large-motion frames:", flagged, "others:", calm, st)
    a.close()
