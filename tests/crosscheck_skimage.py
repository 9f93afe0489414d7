"""Secondary cross-check of oracle primitives against scikit-image (run by tests/test_oracle_crosscheck_cpu.py under the image's
conda interpreter, which has scikit-image 0.18 but none of this repo's dependencies).  NOT the parity reference: scikit-image is an
independent implementation whose semantics coincide with OpenCV's for these cases, so agreement is evidence that the
oracle restates the OpenCV primitive correctly.

usage: python3.9 crosscheck_skimage.py <in.npz> <out.npz>
in : fast_img (u8, the padded ORB level image), otsu_imgs (k, h, w) u8, resize_src (f32), resize_shape (2,),
     brief_img (u8, blurred level image), brief_rc (n, 2) keypoint rows / columns, brief_angle (n,) radians,
     angle_img (u8, padded unblurred level image), angle_rc (n, 2)
out: score (int16 map: largest FAST-9 threshold at which the pixel is still a corner, -1 = never), otsu (k,), triangle (k,), resized,
     pattern (256, 4) scikit-image's copy of the learned rBRIEF pattern, brief (n, 32) steered-BRIEF descriptors, OpenCV bit order,
     angle (n,) intensity-centroid orientation in radians
"""
import sys

import numpy as np
from skimage.feature import corner_fast
from skimage.filters import threshold_otsu, threshold_triangle
from skimage.transform import resize

d = np.load(sys.argv[1])
out = {}

# FAST-9 score map: corner_fast(img, 9, t + 0.5) is the segment test "9 contiguous circle pixels all > p + t or all < p - t" on integers;
# the test is monotone in t, so the largest passing t is (number of passing t) - 1
img = d["fast_img"].astype(np.float64)
score = np.full(img.shape, -1, np.int16)
for t in range(0, 255):
    c = corner_fast(img, n=9, threshold=t + 0.5) > 0
    if not c.any():
        break
    score[c] = t
out["score"] = score

out["otsu"] = np.array([threshold_otsu(im) for im in d["otsu_imgs"]], np.float64)
out["triangle"] = np.array([threshold_triangle(im) for im in d["otsu_imgs"]], np.float64)

dh, dw = [int(v) for v in d["resize_shape"]]
out["resized"] = resize(d["resize_src"].astype(np.float64), (dh, dw), order=1, mode="edge", anti_aliasing=False, preserve_range=True)

# rBRIEF: scikit-image ships the same learned 256-pair pattern as OpenCV and steers it with the keypoint angle like OpenCV does
# (row offset = x sin + y cos, column offset = x cos - y sin, rounded); bit = I(p0) < I(p1)
from skimage.feature.orb_cy import _orb_loop
from skimage.feature._orb_descriptor_positions import POS
out["pattern"] = np.asarray(POS, np.int32)
desc = _orb_loop(np.ascontiguousarray(d["brief_img"].astype(np.float64)), np.ascontiguousarray(d["brief_rc"].astype(np.intp)), np.ascontiguousarray(d["brief_angle"].astype(np.float64)))
out["brief"] = np.packbits(np.asarray(desc).astype(np.uint8), axis=1, bitorder="little")

# orientation by intensity centroid over the radius-15 disc (scikit-image builds the disc from the same u_max table as OpenCV's ORB)
from skimage.feature import corner_orientations
from skimage.feature.orb import OFAST_MASK
out["angle"] = corner_orientations(d["angle_img"].astype(np.float64), np.ascontiguousarray(d["angle_rc"].astype(np.intp)), OFAST_MASK)

np.savez(sys.argv[2], **out)
