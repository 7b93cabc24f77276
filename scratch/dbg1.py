import sys, os
sys.path[:0] = ['/root/repo', '/root/repo/tests', '/root/repo/tests/golden']
import torch, numpy as np
import vcnf_amd as nf
from vcnf_amd import _lib
from helpers import fixture, T, state_for, oracle_rqs_coupling
from oracle import nets as ON
import torch.nn.functional as F
torch.manual_seed(0)
# 1. GEMM accuracy
W = torch.randn(736, 128) * 2 / np.sqrt(128); b = torch.randn(736) * 0.1; h = torch.randn(4096, 128)
ref = F.linear(h.double(), W.double(), b.double())
cpu = F.linear(h, W, b)
gpu = F.linear(h.cuda(), W.cuda(), b.cuda()).cpu()
print("GEMM err vs fp64: cpu max %.3e mean %.3e | gpu max %.3e mean %.3e" % ((cpu - ref).abs().max(), (cpu - ref).abs().mean(), (gpu - ref).abs().max(), (gpu - ref).abs().mean()))
print("matmul flags", torch.backends.cuda.matmul.allow_tf32, torch.get_float32_matmul_precision())
# 2. isolate spline kernel: feed CPU-fp32 params
fx = fixture("g5_c3_stack")
sd, _ = state_for(fx, "c3", 501, final_gain=2.0)
sub = {k: v for k, v in sd.items() if k.startswith("flows.0.")}
o32 = oracle_rqs_coupling(sub, "flows.0.prqct.", 8, 3.0, 128)
o64 = oracle_rqs_coupling({n: v.double() if v.is_floating_point() else v for n, v in sub.items()}, "flows.0.prqct.", 8, 3.0, 128)
g = torch.Generator().manual_seed(8)
x, ctx = 1.2 * torch.randn(4096, 64, generator=g), torch.randn(4096, 16, generator=g)
idf, tf = sub["flows.0.prqct.identity_features"], sub["flows.0.prqct.transform_features"]
for name, nsf_inv in (("density", False), ("sampling", True)):
    if not nsf_inv:
        w32, l32 = o32.nsf_forward(x, ctx); w64, l64 = o64.nsf_forward(x.double(), ctx.double())
        xi = x[:, idf]
    else:
        w32, l32 = o32.nsf_inverse(x, ctx); w64, l64 = o64.nsf_inverse(x.double(), ctx.double())
        xi = w32[:, idf]
    params_cpu = ON.residual_net(sub, "flows.0.prqct.transform_net.", xi, ctx)
    params64 = ON.residual_net({n: v.double() if v.is_floating_point() else v for n, v in sub.items()}, "flows.0.prqct.transform_net.", xi.double(), ctx.double())
    u = "flows.0.prqct.unconditional_transform."
    shared = tuple(sub[u + n].cuda() for n in ("unnormalized_widths", "unnormalized_heights", "unnormalized_derivatives"))
    cfg = _lib.make_cfg(8, "linear", tail_bound=3.0, wh_scale=float(1 / np.sqrt(128)))
    for pname, pp in (("cpu32 params", params_cpu), ("fp64->32 params", params64.float())):
        with torch.no_grad():
            y, ld = _lib.rqs_coupling(x.cuda(), pp.cuda(), tf.int().cuda(), idf.int().cuda(), shared, cfg, nsf_inv)
        e_b = (ld.cpu().double() - l64).abs(); e_r = (l32.double() - l64).abs()
        ey_b = (y.cpu().double() - w64).abs(); ey_r = (w32.double() - w64).abs()
        print("%s [%s]: ld err build max %.3e mean %.3e | ref32 max %.3e mean %.3e || y err build max %.3e mean %.3e | ref max %.3e mean %.3e" % (
            name, pname, e_b.max(), e_b.mean(), e_r.max(), e_r.mean(), ey_b.max(), ey_b.mean(), ey_r.max(), ey_r.mean()))
    # full product path
    m = nf.flows.CoupledRationalQuadraticSpline(64, 2, 128, 8, num_context_channels=16)
    m.load_state_dict({k[len("flows.0."):]: v for k, v in sub.items()}); m = m.cuda().eval()
    with torch.no_grad():
        y, ld = (m.forward if nsf_inv else m.inverse)(x.cuda(), context=ctx.cuda())
    e_b = (ld.cpu().double() - l64).abs()
    print("%s [product]: ld err build max %.3e mean %.3e" % (name, e_b.max(), e_b.mean()))
    pg = m.prqct._params(x.cuda(), ctx.cuda(), nsf_inv).cpu()
    print("   params err vs fp64: gpu max %.3e mean %.3e | cpu32 max %.3e mean %.3e" % ((pg.double() - params64).abs().max(), (pg.double() - params64).abs().mean(), (params_cpu.double() - params64).abs().max(), (params_cpu.double() - params64).abs().mean()))
