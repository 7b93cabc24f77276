"""MADE: masked residual network whose output block for variable i depends only on variables of
lower degree - the conditioner of the autoregressive spline flows (callee of the hot path; masked
dense layers are plain GEMMs on PyTorch-ROCm).  Parameter and buffer names follow the reference
(``initial_layer``, ``blocks.{i}.linear_layers.{0,1}``, ``context_layer``, ``final_layer``; every masked
layer carries ``mask`` and ``degrees`` buffers), so its checkpoints load unchanged.
Reference: normflow/nets/made.py:15-300."""
import torch
from torch import nn
from torch.nn import functional as F


def input_degrees(features):
    return torch.arange(1, features + 1)


def _hidden_degrees(width, autoregressive_features):
    # made.py:73-76: degrees cycle through 1 .. D-1
    hi = max(1, autoregressive_features - 1)
    lo = min(1, autoregressive_features - 1)
    return torch.arange(width) % hi + lo


class MaskedLinear(nn.Linear):
    """nn.Linear whose weight is multiplied by a fixed 0/1 ``mask`` [out, in].  Hidden layers let a
    unit of degree d see inputs of degree <= d; the output layer lets the block of variable i
    (degree d_i, repeated ``multiplier`` times consecutively) see hidden units of degree < d_i."""

    def __init__(self, in_degrees, out_features, autoregressive_features, random_mask, is_output,
                 bias=True, out_degrees_=None):
        super().__init__(in_features=len(in_degrees), out_features=out_features, bias=bias)
        in_degrees = torch.as_tensor(in_degrees)
        if is_output:
            base = input_degrees(autoregressive_features) if out_degrees_ is None else torch.as_tensor(out_degrees_)
            degrees = base.repeat_interleave(out_features // autoregressive_features)        # made.py:57-61
            mask = (degrees[:, None] > in_degrees[None, :]).float()
        else:
            if random_mask:                                                                  # made.py:64-72
                low = min(int(in_degrees.min()), autoregressive_features - 1)
                degrees = torch.randint(low=low, high=autoregressive_features, size=[out_features], dtype=torch.long)
            else:
                degrees = _hidden_degrees(out_features, autoregressive_features)
            mask = (degrees[:, None] >= in_degrees[None, :]).float()
        self.register_buffer('mask', mask)
        self.register_buffer('degrees', degrees)

    def forward(self, x):
        return F.linear(x, self.weight * self.mask, self.bias)


class MaskedFeedforwardBlock(nn.Module):
    """masked Linear -> activation -> dropout, as wide as its input (made.py:83-135)."""

    def __init__(self, in_degrees, autoregressive_features, context_features=None, random_mask=False,
                 activation=F.relu, dropout_probability=0., use_batch_norm=False):
        super().__init__()
        if context_features is not None:
            raise NotImplementedError()
        width = len(in_degrees)
        self.batch_norm = nn.BatchNorm1d(width, eps=1e-3) if use_batch_norm else None
        self.linear = MaskedLinear(in_degrees, width, autoregressive_features, random_mask, is_output=False)
        self.degrees = self.linear.degrees
        self.activation = activation
        self.dropout = nn.Dropout(p=dropout_probability)

    def forward(self, inputs, context=None):
        if context is not None:
            raise NotImplementedError()
        h = inputs if self.batch_norm is None else self.batch_norm(inputs)
        return self.dropout(self.activation(self.linear(h)))


class MaskedResidualBlock(nn.Module):
    """x + masked Linear(drop(act(masked Linear(act(x))))), optionally GLU-gated by a projection of the
    context (made.py:138-212)."""

    def __init__(self, in_degrees, autoregressive_features, context_features=None, random_mask=False,
                 activation=F.relu, dropout_probability=0., use_batch_norm=False, zero_initialization=True):
        if random_mask:
            raise ValueError('Masked residual block can\'t be used with random masks.')
        super().__init__()
        width = len(in_degrees)
        if context_features is not None:
            self.context_layer = nn.Linear(context_features, width)
        self.use_batch_norm = use_batch_norm
        if use_batch_norm:
            self.batch_norm_layers = nn.ModuleList([nn.BatchNorm1d(width, eps=1e-3) for _ in range(2)])
        first = MaskedLinear(in_degrees, width, autoregressive_features, False, is_output=False)
        second = MaskedLinear(first.degrees, width, autoregressive_features, False, is_output=False)
        self.linear_layers = nn.ModuleList([first, second])
        self.degrees = second.degrees
        if not bool(torch.all(self.degrees >= torch.as_tensor(in_degrees))):
            raise RuntimeError('In a masked residual block, the output degrees can\'t be'
                               ' less than the corresponding input degrees.')
        self.activation = activation
        self.dropout = nn.Dropout(p=dropout_probability)
        if zero_initialization:
            for p in (second.weight, second.bias):
                nn.init.uniform_(p, a=-1e-3, b=1e-3)

    def forward(self, inputs, context=None):
        h = inputs
        for i in range(2):
            if self.use_batch_norm:
                h = self.batch_norm_layers[i](h)
            h = self.activation(h)
            if i == 1:
                h = self.dropout(h)
            h = self.linear_layers[i](h)
        if context is not None:
            h = F.glu(torch.cat((h, self.context_layer(context)), dim=1), dim=1)
        return inputs + h


class MADE(nn.Module):
    """masked Linear -> blocks -> masked output Linear with ``output_multiplier`` values per variable,
    laid out [variable][value] (made.py:215-300)."""

    def __init__(self, features, hidden_features, context_features=None, num_blocks=2, output_multiplier=1,
                 use_residual_blocks=True, random_mask=False, permute_mask=False, activation=F.relu,
                 dropout_probability=0., use_batch_norm=False, preprocessing=None):
        if use_residual_blocks and random_mask:
            raise ValueError('Residual blocks can\'t be used with random masks.')
        super().__init__()
        self.preprocessing = preprocessing
        degrees = input_degrees(features)
        if permute_mask:
            degrees = degrees[torch.randperm(features)]
        self.initial_layer = MaskedLinear(degrees, hidden_features, features, random_mask, is_output=False)
        if context_features is not None:
            self.context_layer = nn.Linear(context_features, hidden_features)
        block = MaskedResidualBlock if use_residual_blocks else MaskedFeedforwardBlock
        blocks, prev = [], self.initial_layer.degrees
        for _ in range(num_blocks):
            blocks.append(block(in_degrees=prev, autoregressive_features=features, context_features=context_features,
                                random_mask=random_mask, activation=activation,
                                dropout_probability=dropout_probability, use_batch_norm=use_batch_norm))
            prev = blocks[-1].degrees
        self.blocks = nn.ModuleList(blocks)
        self.final_layer = MaskedLinear(prev, features * output_multiplier, features, random_mask, is_output=True,
                                        out_degrees_=degrees)

    def forward(self, inputs, context=None):
        h = inputs if self.preprocessing is None else self.preprocessing(inputs)
        h = self.initial_layer(h)
        if context is not None:
            h = h + self.context_layer(context)
        for blk in self.blocks:
            h = blk(h, context)
        return self.final_layer(h)
