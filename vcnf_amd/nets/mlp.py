"""Conditioner MLP of the affine couplings (callee of the hot path).

Dense layers run on PyTorch-ROCm (hipBLASLt / rocBLAS -> MFMA); the module
keeps the reference's state_dict layout ``net.{i}.weight|bias`` with Linear
layers at even positions.  Reference: normflow/nets/mlp.py:7-58.
"""
from torch import nn


class MLP(nn.Module):
    def __init__(self, layers, leaky=0.0, score_scale=None, output_fn=None,
                 output_scale=None, init_zeros=False, dropout=None):
        """``layers``: widths from input to output.  LeakyReLU(leaky) between
        Linear layers, optional Dropout before the last Linear, optional zero
        initialisation of the last Linear (mlp.py:30-40)."""
        super().__init__()
        if output_fn is not None or score_scale is not None or output_scale is not None:
            # output squashing (mlp.py:41-54) is not used by any coupling conditioner
            raise NotImplementedError("MLP output_fn/score_scale/output_scale are outside the coupling hot path")
        mods = []
        for fan_in, fan_out in zip(layers[:-2], layers[1:-1]):
            mods += [nn.Linear(fan_in, fan_out), nn.LeakyReLU(leaky)]
        if dropout is not None:
            mods.append(nn.Dropout(p=dropout))
        last = nn.Linear(layers[-2], layers[-1])
        if init_zeros:
            nn.init.zeros_(last.weight)
            nn.init.zeros_(last.bias)
        mods.append(last)
        self.net = nn.Sequential(*mods)

    def forward(self, x):
        return self.net(x)
