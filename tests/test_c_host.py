"""The C ABI used from a C program (no Python, no PyTorch in the process): tests/c_host/abi_smoke.c is
compiled with gcc (plain C11) against include/vcnf_hip.h and libvcnf_hip.so and run as a child process."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compile(exe):
    lib_dir = os.path.join(ROOT, "vcnf_amd", "csrc")
    cmd = ["gcc", "-std=c11", "-O2", "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tests", "c_host", "abi_smoke.c"),
           "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", "-L", lib_dir, "-L", "/opt/rocm/lib",
           "-lvcnf_hip", "-lamdhip64", "-lm", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    return subprocess.run(cmd, capture_output=True, text=True)


@pytest.mark.gpu
def test_c_program_calls_the_library(hip, tmp_path):
    exe = str(tmp_path / "abi_smoke")
    done = _compile(exe)
    assert done.returncode == 0, done.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    assert "abi_smoke ok" in run.stdout


def test_c_program_compiles_against_the_header(tmp_path):
    """No GPU: the header is valid C11 and the program links against the library."""
    import vcnf_amd
    vcnf_amd.lib()                                    # makes sure the library is built
    exe = str(tmp_path / "abi_smoke")
    done = _compile(exe)
    assert done.returncode == 0, done.stderr
    assert os.path.exists(exe)
