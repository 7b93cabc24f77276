"""Phase stamps of the small-batch fused RQS layer kernel (csrc/fused_layer_v6s.hip), one 32-sample tile per workgroup.

Needs a library whose v6s translation unit (two residual blocks) was compiled with -DVCNF_TIME=1:
    VCNF_OBJ_DIR=scratch/obj python -m vcnf_amd.build
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DVCNF_V6_NBLK=2 -DVCNF_TIME=1 \
        -c vcnf_amd/csrc/fused_layer_v6s.hip -o scratch/v6s_time.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libvcnf_time.so $(ls scratch/obj/*.o | grep -v fused_layer_v6s_b2.o) scratch/v6s_time.o
    VCNF_LIB=$PWD/scratch/libvcnf_time.so python profiles/tools/v6s_phase_timing.py
Waves 0 (trunk wave) and 4 (waits through the trunk) of workgroup 0 stamp the 100 MHz wall clock and the shader clock.
"""
import os
import sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import torch
import vcnf_amd as nf

NAMES = ["kernel entry", "set-up done (index / bias / knot tables)", "first barrier passed", "x rows + context in LDS",
         "barrier", "identity half done", "barrier", "first layer + publish", "block 0 first layer + publish",
         "block 0 second layer + gate + publish", "trunk done (all barriers)", "last-layer matrix steps done",
         "splines done, log-det share written", "barrier", "outputs issued", "outputs landed"]
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # > 1: a run of layers in one launch; the stamps after the
                                                            # identity half are those of the LAST layer
from vcnf_amd import fused as fz
flows = [nf.flows.CoupledRationalQuadraticSpline(64, 2, 128, 8, reverse_mask=bool(i % 2), num_context_channels=16) for i in range(NL)]
model = nf.NormalizingFlow(nf.distributions.DiagGaussian(64), flows).cuda().eval()
lay = flows[0]
with torch.no_grad():
    xb, cb = torch.randn(B, 64, device='cuda'), torch.randn(B, 16, device='cuda')
    for dirn in ("inverse", "forward"):
        for _ in range(3):
            if NL == 1:
                y, _ = getattr(lay, dirn)(xb, context=cb)
            else:
                order = list(reversed(model.flows)) if dirn == "inverse" else list(model.flows)
                end, run, sig = fz.plan_stack(order, 0, xb, cb)
                y, _ = fz.run_stack(run, sig, xb, cb, dirn == "forward", None, 1.0)
        torch.cuda.synchronize()
        for w, row in ((0, 0), (4, 0)):
            v = y[0, 32 * (w // 4):32 * (w // 4) + 32].double().cpu()
            print("%s, wave %d: microseconds since kernel entry | shader cycles" % ("density" if dirn == "inverse" else "sampling", w))
            for i, n in enumerate(NAMES):
                print("    %-48s %8.2f us %9d" % (n, float(v[i]) / 100.0, int(v[16 + i])))
