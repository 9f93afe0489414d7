"""Builds examples/rgbd_tum_noros_shim.cpp (the reference's frame loop on the drop-in C++ classes of include/) with plain g++."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(out_path):
    lib_dir = os.path.join(ROOT, "sindslam_amd")
    if not os.path.exists(os.path.join(lib_dir, "libsind_hip.so")):
        raise RuntimeError("libsind_hip.so is missing: run __graft_entry__.build() first")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "rgbd_tum_noros_shim.cpp"),
           "-L" + lib_dir, "-lsind_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath-link,/opt/rocm/lib", "-o", out_path]
    subprocess.check_call(cmd)
    return out_path


def build_boundary(out_path):
    """tests/cpp/boundary_callsites.cpp: the reference's call sites on cv:: types, -DSIND_WITH_OPENCV, against the test-only <opencv2/core.hpp>"""
    lib_dir = os.path.join(ROOT, "sindslam_amd")
    if not os.path.exists(os.path.join(lib_dir, "libsind_hip.so")):
        raise RuntimeError("libsind_hip.so is missing: run __graft_entry__.build() first")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-DSIND_WITH_OPENCV", "-I" + os.path.join(ROOT, "tests", "opencv_mock"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "boundary_callsites.cpp"), "-L" + lib_dir, "-lsind_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath-link,/opt/rocm/lib", "-o", out_path]
    subprocess.check_call(cmd)
    return out_path


def build_seq_fake(out_path):
    """tests/cpp/seq_fake.cpp + the product's HIP-free sequence driver (csrc/host/seq.cpp, seq_net_tcp.cpp) as one shared library, plain g++: the driver's logic on the CPU"""
    host = os.path.join(ROOT, "sindslam_amd", "csrc", "host")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-fPIC", "-shared", os.path.join(ROOT, "tests", "cpp", "seq_fake.cpp"), os.path.join(host, "seq.cpp"),
           os.path.join(host, "seq_net_tcp.cpp"), "-lpthread", "-o", out_path]
    subprocess.check_call(cmd)
    return out_path
