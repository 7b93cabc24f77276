"""Per-phase cycle counts of the default fused RQS layer kernel v4 (fused_layer_v4.hip).

Needs a library whose v4 translation unit (two residual blocks) was compiled with -DVCNF_TIME=1: s_memtime
stamps at every workgroup barrier; wave 0 of workgroup 0 (wave group A) leaves its sums in the first output row.
    VCNF_OBJ_DIR=obj python -m vcnf_amd.build
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DVCNF_V4_NBLK=2 -DVCNF_TIME=1 -c vcnf_amd/csrc/fused_layer_v4.hip -o v4t.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o libvcnf_time.so $(ls obj/*.o | grep -v fused_layer_v4_b2.o) v4t.o
    LIBV=$PWD/libvcnf_time.so python profiles/tools/v4_phase_timing.py
"""
import os
import sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import torch
from vcnf_amd import build as B
B.LIB = os.environ['LIBV']
import vcnf_amd as nf

NAMES = ["loop top (y rows out, first-layer weights)", "x rows -> LDS", "identity half (density direction)",
         "first layer", "V1: identity half (sampling), publish", "trunk matrix steps (4)", "trunk publish steps (2)",
         "trunk gate steps (2)", "last layer: operand fetch", "last layer: first window", "round matrix steps (4)",
         "round vector steps (4)", "log-det exchange", "after the last tile", "re-alignment barriers", "barrier waits"]
torch.manual_seed(0)
lay = nf.flows.CoupledRationalQuadraticSpline(64, 2, 128, 8, num_context_channels=16).cuda().eval()
lay.prqct.fused_precision = "fp16x3"
with torch.no_grad():
    xb, cb = torch.randn(1 << 20, 64, device='cuda'), torch.randn(1 << 20, 16, device='cuda')
    tiles = (1 << 20) // 128 // 256
    for dirn in ("forward", "inverse"):
        for _ in range(2):
            y, _ = getattr(lay, dirn)(xb, context=cb)
        torch.cuda.synchronize()
        t = y[0, :16].double().cpu() / tiles
        print("%s: shader cycles per 128-sample tile (wave 0 of workgroup 0), total %d" % (dirn, round(float(t.sum()))))
        for n, v in zip(NAMES, t):
            print("    %-52s %7d" % (n, round(float(v))))
