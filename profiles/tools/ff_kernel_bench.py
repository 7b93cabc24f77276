"""Launch time of the last-layer + spline kernel (csrc/fused_final.hip) alone at config C5's layer shape (D = 1024, K = 16,
pre-split trunk rows), 524 288 samples, HIP events over 10 launches, both directions.  VCNF_LIB selects the library build."""
import os
import sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import torch
import vcnf_amd as nf
from vcnf_amd import _lib, fused_final

torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
lay = nf.flows.CoupledRationalQuadraticSpline(1024, 2, 128, 16).cuda().eval()
with torch.no_grad():
    for n, p in lay.named_parameters():
        if "final_layer" in n or "unconditional" in n:
            p.normal_(0, 0.3)
c = lay.prqct
with torch.no_grad():
    x = torch.randn(B, 1024, device='cuda')
    out = torch.empty_like(x)
    xi = x[:, c.identity_features].contiguous()
    h = _lib.resnet_trunk(xi, fused_final.packed_trunk(c), 128, 2, split=True)
    pack = fused_final.packed_weights(c)
    for sampling in (False, True):
        fn = lambda: _lib.rqs_final_fused(x, h, out, c._index32('tf'), 512, 128, pack, c._cfg(True), sampling, presplit=True)
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        flop = 3 * 2.0 * 128 * 512 * 48 * B
        print("%s %-9s %8.3f ms per launch of %d samples  (%.0f TFLOP/s of issued f16 matrix work)" % (
            os.path.basename(os.environ.get("VCNF_LIB", "libvcnf_hip.so")), "sampling" if sampling else "density", ms, B, flop / ms * 1e-9))
        if "time" in os.environ.get("VCNF_LIB", ""):
            part = fn()
            torch.cuda.synchronize()
            nwg = 128 * 4
            t = part.flatten()[:4 * 8 * nwg].view(torch.int32).cpu().view(nwg, 8, 4).long() & 0xffffffff
            t0 = int(t[:, :, 0].min())
            st, en = (t[:, :, 0] - t0).float() / 100.0, (t[:, :, 1] - t0).float() / 100.0     # microseconds
            print("   timing build: kernel window %.0f us; wave windows: start min/median/max %.0f / %.0f / %.0f us, end %.0f / %.0f / %.0f us" % (
                float(en.max()), float(st.min()), float(st.median()), float(st.max()), float(en.min()), float(en.median()), float(en.max())))
            dur = en - st
            print("   wave durations min/median/max %.0f / %.0f / %.0f us; per-workgroup spread of wave ends (max - min) median %.0f us" % (
                float(dur.min()), float(dur.median()), float(dur.max()), float((en.max(1).values - en.min(1).values).median())))
            first = st.min(1).values < 1000.0
            print("   workgroups started in the first ms: %d of %d; second-round starts min/median %.0f / %.0f us" % (
                int(first.sum()), nwg, float(st.min(1).values[~first].min()) if (~first).any() else -1,
                float(st.min(1).values[~first].median()) if (~first).any() else -1))
            cyc = t[:, :, 2].float() * 16
            print("   in-kernel clock median %.2f GHz" % float((cyc / (dur * 1e3)).median()))
