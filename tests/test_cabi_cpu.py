"""CPU: the C-ABI library loads and exports every symbol include/sind_hip.h declares (no compute without a GPU), the
product has no CPU fallback, and nothing under sindslam_amd/ reaches into oracle/."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "sind_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sind_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from sindslam_amd._lib import lib
    names = declared_symbols()
    assert len(names) >= 30, names
    L = lib()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_error_path_without_gpu_is_loud():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sindslam_amd._lib import lib
    h = C.c_void_p()
    rc = lib().sind_flow_create(384, 288, 1, 0, C.byref(h))
    assert rc < 0 and len(lib().sind_last_error()) > 0                           # no silent CPU path
    from sindslam_amd.orb import ORBextractor
    from sindslam_amd._lib import SindError
    with pytest.raises(SindError, match="no ROCm-capable device"):            # fails loudly, never falls back to a CPU path
        ORBextractor()


def test_bad_arguments():
    from sindslam_amd._lib import lib
    h = C.c_void_p()
    assert lib().sind_flow_create(8, 8, 1, 0, C.byref(h)) == -1
    assert lib().sind_orb_create(0, C.c_float(1.2), 8, 15, 5, 0, C.byref(h)) == -1
    assert lib().sind_pipe_create(None, C.byref(h)) == -1
    assert lib().sind_frame_create(None, 640, 480, 1, 16, 0, C.byref(h)) == -1
    assert lib().sind_match_create(None, C.byref(h)) == -1
    assert lib().sind_cloud_create(C.c_double(0.0), C.c_double(1.0), C.c_double(0.0), C.c_double(0.0), C.c_double(5000.0), 640, 480, 1, 0, C.byref(h)) == -1
    lab = lib().sind_lab_build()             # the shipped library carries solver modes 0 (per-colour reference), 4 (tiled) and 5 (streaming); the other variants are lab builds
    assert lib().sind_flow_set_sor(7, 5, 64) == -1 and lib().sind_flow_set_sor_tiled(4, 5, 64, 64) == 0 and lib().sind_flow_set_sor_tiled(5, 5, 64, 64) == 0 and lib().sind_flow_set_sor_tiled(0, 5, 64, 64) == 0
    assert lib().sind_flow_set_sor(1, 5, 64) == (0 if lab else -1) and lib().sind_flow_set_sor_tiled(3, 3, 64, 48) == (0 if lab else -1) and lib().sind_flow_set_sor_tiled(4, 0, 64, 64) == (0 if lab else -1)
    assert lib().sind_flow_set_sor_tiled(3, 3, 64, 64) == -1          # 512 threads: no three-waves-per-SIMD instance
    assert lib().sind_flow_set_sor_tiled(4, 5, 64, 64) == 0           # back to the default


def test_product_never_touches_the_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "sindslam_amd")):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                t = open(os.path.join(dp, f), errors="replace").read()
                if re.search(r"oracle[/_]|liboracle|oracle_lib", t) and "never" not in t.lower():
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
