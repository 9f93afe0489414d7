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
    # solver settings are per flow handle (round 5): a null handle is an argument error, nothing is process-wide
    assert lib().sind_flow_set_sor(None, 4, 5, 64) == -1 and lib().sind_flow_set_sor_tiled(None, 4, 5, 64, 64) == -1
    assert lib().sind_flow_set_wave_solver(None, 1, 0, 0) == -1 and lib().sind_dyna_timing_fine(None, None, 0) == -1
    assert lib().sind_flow_set_solver_workgroups(None, 0) == -1 and lib().sind_flow_set_coef_kernel(None, 1) == -1
    assert lib().sind_flow_set_coarse_chain(None, 1) == -1 and lib().sind_flow_set_latency_tiles(None, 1) == -1


def test_product_never_touches_the_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "sindslam_amd")):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                t = open(os.path.join(dp, f), errors="replace").read()
                if re.search(r"oracle[/_]|liboracle|oracle_lib", t) and "never" not in t.lower():
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_missing_rccl_is_an_error_code_not_a_crash():
    """the collective's loader on a box without RCCL (SIND_RCCL_LIB points the dlopen at a file that does not exist): sind_comm_unique_id and sind_comm_create
    return SIND_E_STATE with a message -- the first version read dlerror() twice and built a std::string from NULL"""
    import subprocess, sys
    code = (
        "import ctypes as C, sys; sys.path.insert(0, %r)\n"
        "from sindslam_amd._lib import lib\n"
        "L = lib(); buf = (C.c_char * 128)(); h = C.c_void_p()\n"
        "rc1 = L.sind_comm_unique_id(buf, 128); e1 = L.sind_last_error().decode()\n"
        "rc2 = L.sind_comm_create(buf, 0, 1, 0, C.byref(h)); e2 = L.sind_last_error().decode()\n"
        "assert rc1 == -4 and rc2 == -4, (rc1, rc2)\n"
        "assert 'RCCL is not available' in e1 and 'no_such_librccl' in e1 and 'RCCL is not available' in e2, (e1, e2)\n"
        "assert L.sind_comm_unique_id(buf, 8) == -1\n"            # too small a buffer is an argument error before anything is loaded
        "print('ok')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SIND_RCCL_LIB="/nonexistent/no_such_librccl.so"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, (r.returncode, r.stdout[-300:], r.stderr[-800:])


def test_region_grow_size_gate_and_chunk_calls_without_a_gpu():
    """argument checks of the round-4 entry points that need no device: ragged steps, retained steps, state fingerprints on a null / unbuilt handle"""
    from sindslam_amd._lib import lib
    L = lib()
    for fn in ("sind_pipe_set_state_hashing", "sind_pipe_set_active_frames", "sind_pipe_reserve_retained", "sind_pipe_retain_next", "sind_pipe_release_retained",
               "sind_pipe_host_info", "sind_pipe_set_cpu_share", "sind_pipe_get_state_hashes", "sind_pipe_replay"):
        assert hasattr(L, fn), fn
    assert L.sind_pipe_set_state_hashing(None, 1) == -1 and L.sind_pipe_set_active_frames(None, None) == -1 and L.sind_pipe_reserve_retained(None, 1) == -1
    assert L.sind_pipe_retain_next(None, 0) == -1 and L.sind_pipe_host_info(None, None) == -1 and L.sind_pipe_get_state_hashes(None, None, 0) == -1
    assert L.sind_pipe_replay(None, 0, None, None, None, None, None, None, 0, None, None) == -1


def test_wave_solver_layout_covers_every_level_size():
    """k_sor_wave's strips and bands (sind_flow_wave_layout, host arithmetic): for every width up to 2000 the strips cover the level, a strip's kept columns plus its halos fit the
    128 columns a wave works on, the kept width is even (a lane's two pixels never straddle a cut); for heights and batch sizes the bands cover the rows and are never empty"""
    import ctypes as C
    import numpy as np
    from sindslam_amd._lib import lib
    out = (C.c_int * 4)()
    for w in list(range(1, 700)) + list(range(700, 2001, 7)):
        assert lib().sind_flow_wave_layout(w, 173, 170, 0, 0, out) == 0
        n, iw = out[0], out[1]
        assert n >= 1 and iw % 2 == 0 and n * iw >= w and (n - 1) * iw < w, (w, n, iw)
        for s in range(n):
            ix0 = s * iw; ix1 = min(ix0 + iw, w); ex0 = max(ix0 - 10, 0)
            right_cut = ix1 < w
            assert ix1 + (10 if right_cut else 0) <= ex0 + 128, (w, s, n, iw)            # kept columns + the halo of a cut side lie inside the wave's 128 columns
    for h in range(1, 500):
        for B, items, forced in ((1, 0, 0), (170, 0, 0), (512, 0, 0), (3, 4096, 0), (2, 0, 5)):
            assert lib().sind_flow_wave_layout(230, h, B, items, forced, out) == 0
            nb, bh = out[2], out[3]
            assert nb >= 1 and bh >= 1 and nb * bh >= h and (nb - 1) * bh < h, (h, B, nb, bh)
            if forced:
                assert nb <= forced
    assert lib().sind_flow_wave_layout(0, 10, 1, 0, 0, out) == -1 and lib().sind_flow_wave_layout(10, 10, 1, 0, 0, None) == -1
