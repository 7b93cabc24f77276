"""Per-phase shader cycles of the fused RQS layer kernel (timing build: VCNF_V9_FLAGS="-DVCNF_TIME=1" python -m vcnf_amd.build):
wave 0 of workgroup 0 sums clock64() differences per phase over its tiles and leaves them behind the redo flags."""
import os
import sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import torch
import vcnf_amd as nf
from vcnf_amd import _lib, fused

NAMES = {0: "loop top (first-layer weights requested)", 1: "x rows -> LDS, context -> fragments", 2: "identity half + first-layer fragments",
         3: "trunk steps (21)", 4: "last layer: operand fetch", 5: "first windows (DMA wait)", 6: "fill: group 0 matrix only",
         7: "round steps", 8: "tail: last splines, log-det", 9: "flags, log_det + y stores", 15: "waiting at barriers"}
torch.manual_seed(0)
lay = nf.flows.CoupledRationalQuadraticSpline(64, 2, 128, 8, num_context_channels=16).cuda().eval()
with torch.no_grad():
    for n, p in lay.named_parameters():
        if "unnormalized_" in n:
            p.normal_(0.0, 0.5)
c = lay.prqct
B = 1 << 20
x, ctx = torch.randn(B, 64, device='cuda'), torch.randn(B, 16, device='cuda')
with torch.no_grad():
    for sampling in (False, True):
        for _ in range(3):
            fused.run(c, x, ctx, sampling)
        torch.cuda.synchronize()
        flags = _lib._redo_flags(x.device, B // 128).cpu()
        t = flags[1024:1042].tolist()
        tiles = (B // 128 + 255) // 256
        tot = t[16] * 16
        print("%s: %d tiles per workgroup, %.0f cycles per tile, in-kernel clock %.2f GHz" % (
            "sampling" if sampling else "density", tiles, tot / tiles, 0.1 * tot / max(t[17], 1)))
        for i, nm in NAMES.items():
            print("   %-45s %8.0f" % (nm, 16.0 * t[i] / tiles))
