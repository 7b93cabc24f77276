"""Glow block: affine coupling (HIP kernel, conv conditioner) -> invertible 1x1
convolution -> ActNorm.  Reference: normflow/flows/affine/glow.py:12-74.

Inference runs the 1x1 convolution and the ActNorm as ONE per-pixel channel map (csrc/channel_mix.hip):
    sampling direction (glow.py:59-65):  z -> exp(s) * (W^-1 z) + t            M = diag(exp s) W^-1,  v = t
    density direction  (glow.py:67-73):  z -> W ((z - t) * exp(-s))            M = W diag(exp -s),    v = -W (t exp -s)
with M, v composed in fp64 from the layer parameters (W = P L U or the dense W, mixing.py:71-95) and cached per
direction, keyed on the parameters' (data_ptr, _version) like the packed conditioner weights and rewritten in place
(fused.refresh_packed invalidates it)."""
import torch
from torch import nn

from ..base import Flow
from .coupling import AffineCouplingBlock
from ..mixing import Invertible1x1Conv
from ..normalization import ActNorm
from ... import nets, _lib, autograd


class GlowBlock(Flow):
    def __init__(self, channels, hidden_channels, scale=True, scale_map='sigmoid',
                 split_mode='channel', leaky=0.0, init_zeros=True, use_lu=True, net_actnorm=False):
        super().__init__()
        n_par = 2 if scale else 1
        if split_mode == 'channel':
            widths = (channels // 2, hidden_channels, hidden_channels, n_par * ((channels + 1) // 2))
        elif split_mode == 'channel_inv':
            widths = ((channels + 1) // 2, hidden_channels, hidden_channels, n_par * (channels // 2))
        elif 'checkerboard' in split_mode:
            raise NotImplementedError('Mode ' + split_mode + ' is not built yet (SURVEY 8f row 2).')
        else:
            raise NotImplementedError('Mode ' + split_mode + ' is not implemented.')
        param_map = nets.ConvNet2d(widths, (3, 1, 3), leaky, init_zeros, actnorm=net_actnorm)
        self.flows = nn.ModuleList([AffineCouplingBlock(param_map, scale, scale_map, split_mode)])
        if channels > 1:
            self.flows.append(Invertible1x1Conv(channels, use_lu))
        self.flows.append(ActNorm((channels,) + (1, 1)))

    # ------------------------------------------------------------------ fused mixers
    def _mixer_parts(self):
        conv = next((f for f in self.flows if isinstance(f, Invertible1x1Conv)), None)
        norm = self.flows[-1]
        return conv, norm

    def _mixer_eligible(self, z):
        conv, norm = self._mixer_parts()
        if not self.fused_mixers or conv is None or z.dim() != 4 or not z.is_cuda or z.dtype != torch.float32:
            return False
        cache = self.__dict__.setdefault('_mix_cache', {})
        if not cache.get('norm_ready'):           # the first batch initialises the ActNorm on the plain path
            if not bool(norm.data_dep_init_done > 0.):
                return False
            cache['norm_ready'] = True            # (one host read per block and refresh_packed, not one per call)
        if norm.s.numel() != z.shape[1] or norm.t.numel() != z.shape[1]:
            return False
        if autograd.needs_grad(z, *conv.parameters(), *norm.parameters()):
            return False
        return bool(_lib.lib().vcnf_channel_mix_supported(z.shape[1]))

    def _mixer(self, sampling):
        """(M, v, log|det| per pixel) of conv + ActNorm in the given direction, cached."""
        conv, norm = self._mixer_parts()
        params = list(conv.parameters()) + [norm.s, norm.t]
        key = tuple((p.data_ptr(), p._version, str(p.device)) for p in params)
        cache = self.__dict__.setdefault('_mix_cache', {})
        hit = cache.get(sampling)
        if hit is not None and hit['key'] == key:
            return hit['out']
        with torch.no_grad():
            c = conv.num_channels
            if conv.use_lu:
                lower = torch.tril(conv.L, diagonal=-1).double() + torch.eye(c, dtype=torch.float64, device=conv.L.device)
                upper = torch.triu(conv.U, diagonal=1).double() + torch.diag((conv.sign_S * torch.exp(conv.log_S)).double())
                ld_w = torch.sum(conv.log_S)
            else:
                ld_w = torch.slogdet(conv.W)[1]
            s, t = norm.s.reshape(c).double(), norm.t.reshape(c).double()
            if sampling:
                # the reference inverts in fp64 and rounds W^-1 to fp32 before the convolution (mixing.py:90-95, :104-106)
                if conv.use_lu:
                    w_inv = (torch.inverse(upper) @ torch.inverse(lower)).float().double() @ conv.P.t().double()
                else:
                    w_inv = torch.inverse(conv.W.double()).float().double()
                mat = torch.exp(s).view(c, 1) * w_inv
                vec = t
                ld = torch.sum(norm.s) - ld_w
            else:
                w = (conv.P.double() @ lower @ upper) if conv.use_lu else conv.W.double()
                mat = w * torch.exp(-s).view(1, c)
                vec = -(w @ (t * torch.exp(-s)))
                ld = ld_w - torch.sum(norm.s)
            out = (mat.float().contiguous(), vec.float().contiguous(), ld.float())
            if hit is not None and hit['out'][0].device == out[0].device:
                for old, new in zip(hit['out'], out):      # in place: a captured HIP graph keeps reading these addresses
                    old.copy_(new)
                out = hit['out']
        cache[sampling] = {'key': key, 'out': out}
        return out

    def _mix(self, z, sampling):
        mat, vec, ld = self._mixer(sampling)
        return _lib.channel_mix(z, mat, vec), ld * (z.size(2) * z.size(3))

    fused_mixers = True

    def forward(self, z):
        total = torch.zeros(z.shape[0], dtype=z.dtype, device=z.device)
        if self._mixer_eligible(z):
            z, log_det = self.flows[0](z)
            total += log_det
            z, log_det = self._mix(z, True)
            return z, total + log_det
        for flow in self.flows:
            z, log_det = flow(z)
            total += log_det
        return z, total

    def inverse(self, z):
        total = torch.zeros(z.shape[0], dtype=z.dtype, device=z.device)
        if self._mixer_eligible(z):
            z, log_det = self._mix(z, False)
            total += log_det
            z, log_det = self.flows[0].inverse(z)
            return z, total + log_det
        for flow in reversed(self.flows):
            z, log_det = flow.inverse(z)
            total += log_det
        return z, total
