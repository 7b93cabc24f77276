"""Build the HIP library in-tree: ``vcnf_amd/csrc/libvcnf_hip.so`` (gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so travels to the GPU box with the repository snapshot.
"""
import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libvcnf_hip.so")
SOURCES = ["rqs_kernels.hip", "affine_kernels.hip", "fused_layer.hip", "fused_layer_v2.hip", "fused_layer_v3.hip", "fused_layer_v4.hip", "fused_affine.hip", "fused_final.hip", "rqs_backward.hip"]
HEADERS = ["rqs_math.hpp", "fused_common.hpp", os.path.join("..", "..", "include", "vcnf_hip.h")]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    """Compile every HIP source into one shared library.  Returns its path."""
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
