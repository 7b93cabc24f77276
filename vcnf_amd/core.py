"""Model container: the log_prob / sample loops over a list of flows.
Reference: normflow/core.py - NormalizingFlow.log_prob :170-183, .sample
:144-168, save/load :185-197.

Layers that expose ``inverse_into`` / ``forward_into`` add their log|det|
straight into the running ``log_q`` inside the coupling kernel (one [B] buffer
for the whole stack, no per-layer allocation or add); every other Flow goes
through the plain ``(z, log_det)`` contract.

Training objectives (forward_kld / reverse_kld / reverse_alpha_div) differentiate through
vcnf_amd.autograd (the spline VJP kernel).  The reference's
categorical-dequantisation branch is out of scope; ``categoricals`` is always
defined (None) so ``sample`` follows core.py:150-155 instead of failing with
the AttributeError the reference has at this HEAD.
"""
import torch
from torch import nn

from .flows.affine.coupling import AffineCouplingBlock, MaskedAffineFlow, AffineConstFlow, _scale_code
from . import fused_affine, fused, fused_masked
from .flows.mixing import Permute
from .flows.neural_spline.wrapper import CoupledRationalQuadraticSpline
from .fused import refresh_packed


class _PackedWeightsMixin:
    """Keeps the packed weight copies of the fused kernels honest across the events that change parameters
    behind autograd's back: train() / eval() transitions and load_state_dict() drop the caches
    (vcnf_amd.fused.refresh_packed); after a manual ``p.data`` edit call ``refresh_packed()`` yourself."""

    def _install_pack_hooks(self):
        self.register_load_state_dict_post_hook(lambda module, incompatible: refresh_packed(module) and None)

    def train(self, mode=True):
        refresh_packed(self)
        return super().train(mode)

    def refresh_packed(self):
        return refresh_packed(self)


class NormalizingFlow(_PackedWeightsMixin, nn.Module):
    def __init__(self, q0, flows, p=None, categoricals=None, catlevels=None, catvdeqs=None):
        super().__init__()
        self._install_pack_hooks()
        if categoricals is not None:
            raise NotImplementedError("variational dequantisation of categorical columns is out of scope")
        self.q0 = q0
        self.flows = nn.ModuleList(flows)
        self.p = p
        self.categoricals = None
        self.fuse_affine_stacks = True           # runs of one-kernel affine layers in a single launch (fused_affine.run_stack)
        self.fuse_rqs_stacks = True              # runs of one-kernel RQS layers in a single launch at small batches (fused.run_stack)
        self.fuse_masked_stacks = True           # runs of MaskedAffineFlow (+ MLP conditioners) / ActNorm layers in a single launch

    # ------------------------------------------------------------ density
    def log_prob(self, x, context=None):
        """log q(x) [B]: flows inverted last to first, log-dets added, base
        log-density at the end (core.py:176-183)."""
        log_q = torch.zeros(len(x), dtype=x.dtype, device=x.device)
        z = x
        order = list(reversed(self.flows))
        skip = False
        resume = 0                               # flows before this position were executed by a stack launch
        for i, flow in enumerate(order):
            if i < resume:
                continue
            if skip:
                skip = False
                continue
            ctx = {'context': context} if (context is not None and getattr(flow, 'takes_context', False)) else {}
            # a run of one-kernel affine layers (and the Permutes between them) is ONE launch
            if i >= resume and isinstance(flow, (Permute, AffineCouplingBlock)) and self.fuse_affine_stacks:
                plan = fused_affine.cached_plan(self, order, i, z, True)
                if plan is not None:
                    resume, steps, trailing = plan
                    core_ = steps[0][0].flows[1]
                    z = fused_affine.run_stack(steps, trailing, z, _scale_code(core_.scale, core_.scale_map), True,
                                               log_q, 1.0)[0]
                    continue
            # a run of MaskedAffineFlow layers with MLP conditioners / per-feature affine layers is ONE launch
            if self.fuse_masked_stacks and isinstance(flow, (MaskedAffineFlow, AffineConstFlow)):
                plan = fused_masked.cached_plan(self, order, i, z)
                if plan is not None:
                    resume, steps = plan
                    z, log_q = fused_masked.run(steps, z, True, log_q, 1.0)       # (a new log_q when autograd records)
                    continue
            # a run of one-kernel RQS coupling layers at a small batch is ONE launch (the tile stays in LDS)
            if self.fuse_rqs_stacks and type(flow) is CoupledRationalQuadraticSpline:
                plan = fused.cached_plan_stack(self, order, i, z, context)
                if plan is not None:
                    resume, run, sig = plan
                    z = fused.run_stack(run, sig, z, context, False, log_q, 1.0)[0]
                    continue
            # a Permute undone right before a one-kernel affine layer becomes that kernel's load index
            if (isinstance(flow, Permute) and i + 1 < len(order) and z.dim() == 2
                    and isinstance(order[i + 1], AffineCouplingBlock) and order[i + 1].fusable(z)):
                z = order[i + 1].run_with_permute(z, True, log_q, 1.0, in_gather=flow._idx32(True, z.device))
                skip = True
                continue
            if hasattr(flow, 'inverse_into'):
                z = flow.inverse_into(z, log_q, **ctx)
            else:
                z, log_det = flow.inverse(z, **ctx)
                log_q += log_det
        if hasattr(self.q0, 'from_noise'):
            self.q0.log_prob(z, out=log_q)
        else:
            log_q += self.q0.log_prob(z)
        return log_q

    # ------------------------------------------------------------ sampling
    def sample(self, num_samples=1, context=None):
        """(z, log q(z)) for fresh base draws (core.py:150-155, :168)."""
        z, log_q = self.q0(num_samples)
        return self._push(z, log_q, context)

    def sample_from(self, eps, context=None):
        """Same as ``sample`` with the standard-normal base draw given."""
        z, log_q = self.q0.from_noise(eps)
        return self._push(z, log_q, context)

    def _push(self, z, log_q, context):
        order = list(self.flows)
        skip = False
        resume = 0
        for i, flow in enumerate(order):
            if i < resume:
                continue
            if skip:
                skip = False
                continue
            ctx = {'context': context} if (context is not None and getattr(flow, 'takes_context', False)) else {}
            if i >= resume and isinstance(flow, (Permute, AffineCouplingBlock)) and self.fuse_affine_stacks:
                plan = fused_affine.cached_plan(self, order, i, z, False)
                if plan is not None:
                    resume, steps, trailing = plan
                    core_ = steps[0][0].flows[1]
                    z = fused_affine.run_stack(steps, trailing, z, _scale_code(core_.scale, core_.scale_map), False,
                                               log_q, -1.0)[0]
                    continue
            if self.fuse_masked_stacks and isinstance(flow, (MaskedAffineFlow, AffineConstFlow)):
                plan = fused_masked.cached_plan(self, order, i, z)
                if plan is not None:
                    resume, steps = plan
                    z, log_q = fused_masked.run(steps, z, False, log_q, -1.0)
                    continue
            if self.fuse_rqs_stacks and type(flow) is CoupledRationalQuadraticSpline:
                plan = fused.cached_plan_stack(self, order, i, z, context)
                if plan is not None:
                    resume, run, sig = plan
                    z = fused.run_stack(run, sig, z, context, True, log_q, -1.0)[0]
                    continue
            # a Permute applied right after a one-kernel affine layer becomes that kernel's store index
            if (isinstance(flow, AffineCouplingBlock) and i + 1 < len(order) and isinstance(order[i + 1], Permute)
                    and z.dim() == 2 and flow.fusable(z)):
                z = flow.run_with_permute(z, False, log_q, -1.0, out_gather=order[i + 1]._idx32(False, z.device))
                skip = True
                continue
            if hasattr(flow, 'forward_into'):
                z = flow.forward_into(z, log_q, **ctx)
            else:
                z, log_det = flow(z, **ctx)
                log_q -= log_det
        return z, log_q

    # ------------------------------------------------------------ objectives
    def _pull(self, z, context=None, trace=None):
        """log q of given points through the plain ``(z, log_det)`` contract, optionally
        recording each layer's output and log-det (the reference's ``extended`` lists)."""
        log_q = torch.zeros(len(z), dtype=z.dtype, device=z.device)
        for flow in reversed(self.flows):
            ctx = {'context': context} if (context is not None and getattr(flow, 'takes_context', False)) else {}
            z, log_det = flow.inverse(z, **ctx)
            if trace is not None:
                trace[0].append(z.detach().cpu().numpy())
                trace[1].append(torch.as_tensor(log_det).detach().cpu().numpy())
            log_q = log_q + log_det
        return log_q + self.q0.log_prob(z)

    def forward_kld(self, x, extended=False, context=None):
        """-mean log q(x) (core.py:30-65).  ``extended``: also the per-layer latents and
        log-dets as numpy arrays and the per-sample log q."""
        if not extended:
            return -torch.mean(self.log_prob(x, context))
        trace = ([], [])
        log_q = self._pull(x, context, trace)
        return -torch.mean(log_q), trace[0], trace[1], log_q

    def _frozen_log_q(self, z, context):
        """log q(z) with the parameters held constant (core.py:87-95 / :124-131): the path
        derivative estimators differentiate through z only."""
        params = [p for p in self.parameters() if p.requires_grad]
        for p in params:
            p.requires_grad = False
        try:
            return self._pull(z, context)
        finally:
            for p in params:
                p.requires_grad = True

    def reverse_kld(self, num_samples=1, beta=1., score_fn=True, extended=False, context=None):
        """mean log q(z) - beta * mean log p(z), z ~ q (core.py:67-100)."""
        z, log_q = self.q0(num_samples)
        trace = ([], []) if extended else None
        for flow in self.flows:
            ctx = {'context': context} if (context is not None and getattr(flow, 'takes_context', False)) else {}
            z, log_det = flow(z, **ctx)
            if trace is not None:
                trace[0].append(z.detach().cpu().numpy())
                trace[1].append(torch.as_tensor(log_det).detach().cpu().numpy())
            log_q = log_q - log_det
        if not score_fn:
            log_q = self._frozen_log_q(z, context)
        log_p = self.p.log_prob(z)
        loss = torch.mean(log_q) - beta * torch.mean(log_p)
        if extended:
            return loss, trace[0], trace[1], log_p, log_q
        return loss

    def reverse_alpha_div(self, num_samples=1, alpha=1, dreg=False, extended=False, context=None):
        """Alpha divergence with samples from q (core.py:101-141); ``dreg``: the doubly
        reparametrised estimator."""
        import numpy as np
        z, log_q = self.sample(num_samples, context)
        log_p = self.p.log_prob(z)
        if dreg:
            w_const = torch.exp(log_p - log_q).detach()
            log_q = self._frozen_log_q(z, context)
            w = torch.exp(log_p - log_q)
            w_alpha = w_const ** alpha
            w_alpha = w_alpha / torch.mean(w_alpha)
            weights = (1 - alpha) * w_alpha + alpha * w_alpha ** 2
            loss = -alpha * torch.mean(weights * torch.log(w))
        else:
            loss = np.sign(alpha - 1) * torch.logsumexp(alpha * (log_p - log_q), 0)
        if extended:
            return loss, [], []
        return loss

    # ------------------------------------------------------------ checkpoints
    def save(self, path):
        torch.save(self.state_dict(), path)

    def load(self, path):
        self.load_state_dict(torch.load(path, weights_only=True))


class MultiscaleFlow(_PackedWeightsMixin, nn.Module):
    """Multiscale (RealNVP / Glow) container: per level a list of flows, a Merge between
    levels and one base distribution per level.  Reference: normflow/core.py:271-399
    (sample :310-340, log_prob :342-367).  Class-conditional bases are out of scope."""

    def __init__(self, q0, flows, merges, transform=None, class_cond=True):
        super().__init__()
        self._install_pack_hooks()
        if class_cond and any(not hasattr(q, 'from_noise') for q in q0):
            raise NotImplementedError("class-conditional base distributions are out of scope; "
                                      "use DiagGaussian bases with class_cond=False")
        self.q0 = nn.ModuleList(q0)
        self.num_levels = len(self.q0)
        self.flows = nn.ModuleList([nn.ModuleList(f) for f in flows])
        self.merges = nn.ModuleList(merges)
        self.transform = transform
        self.class_cond = False

    def forward(self, x, y=None):
        return -self.log_prob(x, y)

    def forward_kld(self, x, y=None):
        """core.py:296-308: forward KL divergence estimate -mean(log_prob); differentiable through the VJP
        kernels of vcnf_amd.autograd."""
        return -torch.mean(self.log_prob(x, y))

    def log_prob(self, x, y=None):
        """core.py:348-367: optional input transform, then per level (finest last) the
        flows inverted, the split-off half scored by that level's base."""
        log_q = 0
        z = x
        if self.transform is not None:
            z, log_det = self.transform.inverse(z)
            log_q = log_q + log_det
        for i in range(self.num_levels - 1, -1, -1):
            for flow in reversed(self.flows[i]):
                z, log_det = flow.inverse(z)
                log_q = log_q + log_det
            if i > 0:
                [z, z_], log_det = self.merges[i - 1].inverse(z)
                log_q = log_q + log_det
            else:
                z_ = z
            log_q = log_q + self.q0[i].log_prob(z_)
        return log_q

    def sample(self, num_samples=1, y=None, temperature=None):
        if temperature is not None:
            self.set_temperature(temperature)
        noise = [torch.randn((num_samples,) + q.shape, dtype=q.loc.dtype, device=q.loc.device) for q in self.q0]
        out = self.sample_from(noise)
        if temperature is not None:
            self.reset_temperature()
        return out

    def sample_from(self, noise):
        """core.py:320-340 with the per-level standard-normal draws supplied."""
        z, log_q = None, None
        for i in range(self.num_levels):
            z_, log_q_ = self.q0[i].from_noise(noise[i])
            if i == 0:
                z, log_q = z_, log_q_
            else:
                log_q = log_q + log_q_
                z, log_det = self.merges[i - 1]([z, z_])
                log_q = log_q - log_det
            for flow in self.flows[i]:
                z, log_det = flow(z)
                log_q = log_q - log_det
        if self.transform is not None:
            z, log_det = self.transform(z)
            log_q = log_q - log_det
        return z, log_q

    def set_temperature(self, temperature):
        for q0 in self.q0:
            if hasattr(q0, 'temperature'):
                q0.temperature = temperature
            else:
                raise NotImplementedError('One base function does not support temperature annealed sampling')

    def reset_temperature(self):
        self.set_temperature(None)

    def save(self, path):
        torch.save(self.state_dict(), path)

    def load(self, path):
        self.load_state_dict(torch.load(path, weights_only=True))
