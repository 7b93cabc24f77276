"""Convolutional conditioner of the Glow blocks (callee of the hot path).  The 3x3 convolutions run on
PyTorch-ROCm / MIOpen; at inference the 1x1 convolution in the middle runs on csrc/conv1x1.hip fused with the bias
adds and LeakyReLUs on both sides of it (ConvNet2d.forward).  Reference: normflow/nets/cnn.py:7-46; state-dict layout
``net.{i}.weight|bias`` with convolutions at even positions."""
import torch
from torch import nn
from torch.nn import functional as F


class ConvNet2d(nn.Module):
    def __init__(self, channels, kernel_size, leaky=0.0, init_zeros=True, actnorm=False, weight_std=None):
        """``channels``: widths from input to output; ``kernel_size``: one (odd) size per
        convolution, 'same' padding.  LeakyReLU(leaky) between convolutions; the last one
        is zero-initialised when ``init_zeros`` (cnn.py:40-43)."""
        super().__init__()
        if actnorm:
            # the reference's utils.ActNorm passes an unsupported kwarg at this HEAD
            raise NotImplementedError("ConvNet2d(actnorm=True) is broken in the reference and not rebuilt")
        mods = []
        for i in range(len(kernel_size) - 1):
            conv = nn.Conv2d(channels[i], channels[i + 1], kernel_size[i], padding=kernel_size[i] // 2)
            if weight_std is not None:
                conv.weight.data.normal_(mean=0.0, std=weight_std)
            mods += [conv, nn.LeakyReLU(leaky)]
        last = nn.Conv2d(channels[-2], channels[-1], kernel_size[-1], padding=kernel_size[-1] // 2)
        if init_zeros:
            nn.init.zeros_(last.weight)
            nn.init.zeros_(last.bias)
        mods.append(last)
        self.net = nn.Sequential(*mods)

    # ------------------------------------------------------------------ inference: fused middle layer
    fused_conv1x1 = True
    # 'fp16x3': the fused kernels' split-half matrix arithmetic; 'fp32': the library's fp32 convolutions (the
    # reference's arithmetic, nets/cnn.py:20-46); None: vcnf_amd.fused.DEFAULT_PRECISION (VCNF_FUSED_PRECISION)
    fused_precision = None

    def _fusable(self, x):
        """Conv(k) , LeakyReLU, Conv(1x1), LeakyReLU, Conv(k) on a CUDA fp32 batch without autograd: the 1x1 convolution
        runs on csrc/conv1x1.hip together with the first convolution's bias, both activations and its own bias."""
        from .. import _lib, autograd
        mods = list(self.net)
        if not (self.fused_conv1x1 and len(mods) == 5 and x.dim() == 4 and x.is_cuda and x.dtype == torch.float32):
            return False
        from .. import fused
        if fused.precision_of(self) != fused.PREC_F16X3:
            return False                   # fp32 asked for: every convolution stays on the library's fp32 path
        c1, a1, c2, a2, c3 = mods
        if not (isinstance(c1, nn.Conv2d) and isinstance(c2, nn.Conv2d) and isinstance(c3, nn.Conv2d)
                and isinstance(a1, nn.LeakyReLU) and isinstance(a2, nn.LeakyReLU)):
            return False
        if c2.kernel_size != (1, 1) or c2.stride != (1, 1) or c2.groups != 1 or c2.dilation != (1, 1) or c2.padding != (0, 0):
            return False
        if torch.is_grad_enabled() and autograd.needs_grad(x, *self.parameters()):
            return False
        return bool(_lib.lib().vcnf_conv1x1_supported(c2.in_channels, c2.out_channels))

    def _packed_conv1x1(self):
        c2 = self.net[2]
        key = (c2.weight.data_ptr(), c2.weight._version, str(c2.weight.device))
        cache = self.__dict__.setdefault('_fused_conv_pack', {})
        if cache.get('key') != key:
            with torch.no_grad():
                buf = pack_conv1x1(c2.weight.detach().view(c2.out_channels, c2.in_channels))
            old = cache.get('buf')
            if old is not None and old.shape == buf.shape and old.device == buf.device:
                old.copy_(buf)           # in place: a captured HIP graph keeps reading this address
            else:
                cache['buf'] = buf
            cache['key'] = key
        return cache['buf']

    def _packed_conv3x3(self):
        c1 = self.net[0]
        key = (c1.weight.data_ptr(), c1.weight._version, str(c1.weight.device))
        cache = self.__dict__.setdefault('_fused_conv3_pack', {})
        if cache.get('key') != key:
            with torch.no_grad():
                k1 = c1.in_channels * 9
                w = c1.weight.detach().reshape(c1.out_channels, k1)          # k = ci * 9 + ky * 3 + kx
                w = F.pad(w, (0, (-k1) % 16))
                buf = pack_conv1x1(w)
            old = cache.get('buf')
            if old is not None and old.shape == buf.shape and old.device == buf.device:
                old.copy_(buf)
            else:
                cache['buf'] = buf
            cache['key'] = key
        return cache['buf']

    def _first_two_fusable(self):
        """3x3 convolution (padding 1, stride 1) into 256 channels followed by the 256 -> 256 1x1 convolution: both on
        csrc/conv3x3_1x1.hip, the hidden activation between them stays on chip."""
        from .. import _lib
        c1, c2 = self.net[0], self.net[2]
        return (self.fused_conv3x3 and c1.kernel_size == (3, 3) and c1.padding == (1, 1) and c1.stride == (1, 1)
                and c1.dilation == (1, 1) and c1.groups == 1 and c1.padding_mode == 'zeros'
                and bool(_lib.lib().vcnf_conv3x3_1x1_supported(c1.in_channels, c1.out_channels, c2.out_channels)))

    fused_conv3x3 = True

    def _packed_taps(self):
        c3 = self.net[4]
        key = (c3.weight.data_ptr(), c3.weight._version, str(c3.weight.device))
        cache = self.__dict__.setdefault('_fused_taps_pack', {})
        if cache.get('key') != key:
            with torch.no_grad():
                co = c3.out_channels
                w = c3.weight.detach().permute(2, 3, 0, 1).reshape(9 * co, c3.in_channels)     # row t * c_out + o
                buf = pack_conv1x1(w, row_blocks=(9 * co + 31) // 32)
            old = cache.get('buf')
            if old is not None and old.shape == buf.shape and old.device == buf.device:
                old.copy_(buf)
            else:
                cache['buf'] = buf
            cache['key'] = key
        return cache['buf']

    def _all_three_fusable(self):
        """... followed by a 3x3 convolution (padding 1) to at most 56 channels: its nine taps run as one more matrix
        phase of the same kernel, a second small kernel shifts and adds them (csrc/conv3x3_1x1.hip)."""
        from .. import _lib
        c1, c3 = self.net[0], self.net[4]
        return (self.fused_conv_taps and c3.kernel_size == (3, 3) and c3.padding == (1, 1) and c3.stride == (1, 1)
                and c3.dilation == (1, 1) and c3.groups == 1 and c3.padding_mode == 'zeros' and c3.in_channels == 256
                and bool(_lib.lib().vcnf_convnet3_supported(c1.in_channels, 256, c3.out_channels)))

    fused_conv_taps = True

    def forward(self, x):
        if self._fusable(x) and self._first_two_fusable() and self._all_three_fusable():
            from .. import _lib
            c1, a1, c2, a2, c3 = self.net
            return _lib.convnet3_fused(x, self._packed_conv3x3(), self._packed_conv1x1(), self._packed_taps(), c1.bias,
                                       c2.bias, c3.bias, c3.out_channels, float(a1.negative_slope), float(a2.negative_slope))
        if self._fusable(x) and self._first_two_fusable():
            from .. import _lib
            c1, a1, c2, a2, c3 = self.net
            h = _lib.conv3x3_1x1_fused(x, self._packed_conv3x3(), self._packed_conv1x1(), c1.bias, c2.bias,
                                       float(a1.negative_slope), float(a2.negative_slope))
            return c3(h)
        if self._fusable(x):
            from .. import _lib
            c1, a1, c2, a2, c3 = self.net
            h = F.conv2d(x, c1.weight, None, c1.stride, c1.padding, c1.dilation, c1.groups)     # bias applied below
            h = _lib.conv1x1_fused(h, self._packed_conv1x1(), c2.out_channels, in_bias=c1.bias, out_bias=c2.bias,
                                   in_slope=float(a1.negative_slope), out_slope=float(a2.negative_slope))
            return c3(h)
        return self.net(x)


def pack_conv1x1(w, row_blocks=8):
    """W [c_out, c_in] -> A fragments of v_mfma_f32_32x32x16_f16 for csrc/conv1x1.hip: [8 row blocks][c_in / 16]
    [hi | lo][64 lanes][8 halves] (``row_blocks`` blocks of 32 rows), lane l holding row 32 rb + l % 32, input channels
    16 ks + 8 (l / 32) + i; rows beyond c_out are zero; hi / lo = the fp16 split of fused._split_halves (w ~ hi + lo / 2048)."""
    from ..fused import _split_halves, _as_floats
    c_out, c_in = w.shape
    dev = w.device
    rb = torch.arange(row_blocks, device=dev).view(-1, 1, 1, 1)
    ks = torch.arange(c_in // 16, device=dev).view(1, -1, 1, 1)
    lane = torch.arange(64, device=dev).view(1, 1, -1, 1)
    i = torch.arange(8, device=dev).view(1, 1, 1, -1)
    shape = (row_blocks, c_in // 16, 64, 8)
    rows = (32 * rb + (lane & 31)).expand(shape)
    cols = (16 * ks + 8 * (lane >> 5) + i).expand(shape)
    ok = rows < c_out
    vals = torch.where(ok, w[torch.where(ok, rows, torch.zeros_like(rows)), cols], torch.zeros((), device=dev, dtype=w.dtype))
    hi, lo = _split_halves(vals)                                   # [row blocks, ks, 64, 8]
    return _as_floats(torch.stack([hi, lo], dim=2)).contiguous()   # [row blocks, ks, 2, 64, 8]
