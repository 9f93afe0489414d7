"""CPU: the header-only C++ classes with the reference's names and signatures (include/DynaDetect.h, include/ORBextractor.h) compile
warning-free with g++ and link against the C ABI -- no GPU call is made here."""
import cpp_shim


def test_shim_example_compiles_and_links(tmp_path):
    exe = cpp_shim.build(str(tmp_path / "rgbd_tum_noros_shim"))
    import subprocess
    r = subprocess.run([exe], capture_output=True, text=True)          # no arguments: prints the usage line and exits with 2
    assert r.returncode == 2 and "usage" in r.stderr


def test_opencv_branch_of_the_shims_compiles_against_the_reference_call_sites(tmp_path):
    """-DSIND_WITH_OPENCV: cv::InputArray / OutputArray overloads, std::make_shared<DynaDetect>(cv::Mat...), (*extractor)(im, cv::Mat(), keys,
    descriptors), the getters and the public mvImagePyramid (src/Frame.cc:544, 634-651) -- type-checked and linked with -Wall -Werror against
    the test-only stand-in for <opencv2/core.hpp> (tests/opencv_mock); tests/test_cpp_shim_gpu.py runs the same binary on the GPU"""
    import subprocess
    exe = cpp_shim.build_boundary(str(tmp_path / "boundary_callsites"))
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
