"""Key-frame scenes for the mapping-consumer tests: frames of the synthetic stream with ground-truth poses, masks and labels from the
ORACLE DynaDetect (test infrastructure)."""
import numpy as np

import oracle_lib as O


def _twc(stream, t):
    c, yaw = stream.pose(t); cs, sn = np.cos(float(yaw)), np.sin(float(yaw))
    T = np.eye(4); T[:3, :3] = [[cs, 0, sn], [0, 1, 0], [-sn, 0, cs]]; T[:3, 3] = c.astype(np.float64)
    return T


def keyframe_pair(stream, t, gap=2):
    """Key frames t-gap (last) and t (current): returns the arguments of generatePointCloud + cam5."""
    bgr, depth = stream.frames(t - gap - 2, gap + 3)
    dd = O.DynaDetect(bgr[1], bgr[0], stream.fx, stream.fy, stream.cx, stream.cy, stream.depth_factor)
    masks, labels = [], []
    for k in range(2, gap + 3):
        dy, lb = dd.detect(bgr[k], depth[k]); masks.append(O.dilate15(dy)); labels.append(lb)
    Twc_cur, Twc_last = _twc(stream, t), _twc(stream, t - gap)
    pose_cur, pose_last = np.linalg.inv(Twc_cur), np.linalg.inv(Twc_last)                 # the node's vecPose holds Tcw
    rel = pose_cur @ np.linalg.inv(pose_last)                                              # pubPointCloud.cc:278
    cam5 = [stream.fx, stream.fy, stream.cx, stream.cy, stream.depth_factor]
    return cam5, dict(imgRGB=bgr[-1], imgDepth=depth[-1], imgDepthLast=depth[-1 - gap], imgDynaMask=masks[-1], imgDynaMaskLast=masks[-1 - gap],
                      imgLabel=labels[-1], poseRelative=rel, Twc=Twc_cur)
