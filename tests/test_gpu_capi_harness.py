"""The C ABI used from a plain C++ program -- no Python, no torch in the process ("a plain C ABI ... keeps
the kernels testable from a C++ harness without Python"): tests/capi/harness.cpp is compiled against include/curla_hip.h,
linked with the in-tree libcurla_hip.so and run; it allocates with the HIP runtime, calls the entry points and checks the
results against CPU loops of its own."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_from_a_cpp_program_without_python(tmp_path):
    import __graft_entry__ as ge
    ge.build()
    exe = str(tmp_path / "capi_harness")
    lib_dir = os.path.join(ROOT, "curla_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "capi", "harness.cpp"), "-I",
                           os.path.join(ROOT, "include"), "-L", lib_dir, "-lcurla_hip", "-Wl,-rpath," + lib_dir, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "capi_harness.txt"), "w") as f:
        f.write(r.stdout + r.stderr)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout + r.stderr
    assert "channels=32" in r.stdout and "channels=16" in r.stdout and "f64 rider" in r.stdout
