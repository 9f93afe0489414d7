"""CPU: the header-only C++ classes with the reference's names and signatures (include/DynaDetect.h, include/ORBextractor.h) compile
warning-free with g++ and link against the C ABI -- no GPU call is made here."""
import cpp_shim


def test_shim_example_compiles_and_links(tmp_path):
    exe = cpp_shim.build(str(tmp_path / "rgbd_tum_noros_shim"))
    import subprocess
    r = subprocess.run([exe], capture_output=True, text=True)          # no arguments: prints the usage line and exits with 2
    assert r.returncode == 2 and "usage" in r.stderr
