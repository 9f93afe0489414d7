"""Secondary cross-check of oracle primitives against scikit-image (run by tests/test_oracle_crosscheck_cpu.py under the image's
conda interpreter, which has scikit-image 0.18 but none of this repo's dependencies).  NOT the parity reference: scikit-image is an
independent implementation whose semantics coincide with OpenCV's for these cases, so agreement is evidence that the
oracle restates the OpenCV primitive correctly.

usage: python3.9 crosscheck_skimage.py <in.npz> <out.npz>
in : gray (u8 image), fast_img (u8, the padded ORB level image), otsu_imgs (k, h, w) u8, resize_src (f32), resize_shape (2,)
out: score (int16 map: largest FAST-9 threshold at which the pixel is still a corner, -1 = never), otsu (k,), triangle (k,), resized
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

np.savez(sys.argv[2], **out)
