"""Convolutional conditioner of the Glow blocks (callee of the hot path; the convolutions
run on PyTorch-ROCm / MIOpen).  Reference: normflow/nets/cnn.py:7-46; state-dict layout
``net.{i}.weight|bias`` with convolutions at even positions."""
from torch import nn


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

    def forward(self, x):
        return self.net(x)
