"""Build the HIP library in-tree: ``vcnf_amd/csrc/libvcnf_hip.so`` (gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so travels to the GPU box with the repository snapshot.
"""
import os
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libvcnf_hip.so")
SOURCES = ["rqs_kernels.hip", "affine_kernels.hip", "fused_layer.hip",
           "fused_layer_v6.hip", "fused_affine.hip", "fused_final.hip", "resnet_trunk.hip", "channel_mix.hip", "conv1x1.hip", "conv3x3_1x1.hip", "linear_wgrad.hip", "resblock_ops.hip",
           "rqs_backward.hip", "gemm_probe.hip", "rqs_f64.hip", "linear_f16x3.hip", "masked_affine_stack.hip"]
# (source, extra flags, object name): fused_layer_v6.hip is compiled once per number of residual blocks.  The two
# kernels whose spline code runs beside matrix instructions are built without SLP vectorisation: packed-f32 vector
# instructions starve beside the partner wave's matrix instructions (profiles/r02_spline_eval_microbench.md)
NO_SLP = ["-fno-slp-vectorize"]
UNITS = [(s, NO_SLP if s == "fused_final.hip" else [], os.path.splitext(s)[0]) for s in SOURCES if s != "fused_layer_v6.hip"] + \
        [("fused_layer_v6.hip", ["-DVCNF_V6_NBLK=%d" % n] + NO_SLP, "fused_layer_v6_b%d" % n) for n in (2, 3, 1)] + \
        [("fused_layer_v6s.hip", ["-DVCNF_V6_NBLK=%d" % n] + NO_SLP, "fused_layer_v6s_b%d" % n) for n in (2, 3, 1)] + \
        [("fused_layer.hip", ["-DVCNF_F32_NBLK=%d" % n], "fused_layer_f32_b%d" % n) for n in (3, 1)]
HEADERS = ["rqs_math.hpp", "rqs_lean.hpp", "fused_common.hpp", "split_half.hpp", os.path.join("..", "..", "include", "vcnf_hip.h")]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + ["fused_layer_v6s.hip"] + HEADERS)


def build(force=False, verbose=False):
    """Compile every HIP source (one hipcc process per file, in parallel) and link them into one
    shared library.  Returns its path.  The object files live in a temporary directory."""
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    jobs = max(1, min(len(UNITS), int(os.environ.get("VCNF_BUILD_JOBS", os.cpu_count() or 1))))
    keep = os.environ.get("VCNF_OBJ_DIR")            # keep the object files (variant links of one unit)
    if keep:
        os.makedirs(keep, exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="vcnf_obj_") as tmp:
        if keep:
            tmp = keep
        def compile_one(unit):
            src, extra, name = unit
            obj = os.path.join(tmp, name + ".o")
            cmd = [hipcc] + flags + extra + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, cwd=CSRC, check=True)
            return obj
        with ThreadPoolExecutor(max_workers=jobs) as pool:
            objs = list(pool.map(compile_one, sorted(UNITS, key=lambda u: not u[0].startswith("fused_layer"))))
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(link), flush=True)
        subprocess.run(link, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
