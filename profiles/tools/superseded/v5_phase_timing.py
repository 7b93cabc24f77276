"""Per-phase cycle counts of the experimental fused RQS layer kernel v5 (fused_layer_v5.hip).

Needs a library whose v5 translation unit was compiled with -DVCNF_TIME=2 (s_memtime stamps at the phase
boundaries; wave 0 of workgroup 0 leaves its sums in the first output row):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DVCNF_TIME=2 -c vcnf_amd/csrc/fused_layer_v5.hip -o v5t.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o libvcnf_time.so <the other objects (VCNF_OBJ_DIR)> v5t.o
    VCNF_FUSED_KERNEL=v5 LIBV=$PWD/libvcnf_time.so python profiles/tools/v5_phase_timing.py
"""
import sys, os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import torch
from vcnf_amd import build as B
B.LIB = os.environ['LIBV']
import vcnf_amd as nf
torch.manual_seed(0)
lay = nf.flows.CoupledRationalQuadraticSpline(64, 2, 128, 8, num_context_channels=16).cuda().eval()
lay.prqct.fused_precision = "fp16x3"
with torch.no_grad():
    xb, cb = torch.randn(1 << 20, 64, device='cuda'), torch.randn(1 << 20, 16, device='cuda')
    for _ in range(2):
        y, _ = lay.forward(xb, context=cb)
    torch.cuda.synchronize()
    t = y[0, :12].double().cpu()
    names = ["-", "barrier", "step0", "trunk", "group0", "loop", "tail-splines", "store+next", "xload", "ident+operands", "-", "-"]
    tiles = (1 << 20) // 128 // 256
    print("shader cycles per 128-sample tile, wave 0 of workgroup 0:", {n: round(float(v) / tiles) for n, v in zip(names, t)}, "sum", round(float(t.sum()) / tiles))
    tu = y[0, 16:42].double().cpu() / tiles
    print("group-slot mini-regions (cycles per tile, 7 slots):", [round(float(v)) for v in tu])
