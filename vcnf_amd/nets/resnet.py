"""Residual conditioner of the RQS couplings (callee of the hot path).

Same parameter names as the reference (``initial_layer``, ``blocks.{i}.
linear_layers.{0,1}``, ``blocks.{i}.context_layer``, ``final_layer``) so its
checkpoints load unchanged.  Reference: normflow/nets/resnet.py:8-106.
The dense layers are plain GEMMs on PyTorch-ROCm (MFMA through hipBLASLt); at training batch sizes their weight /
bias gradients come from csrc/linear_wgrad.hip (autograd.linear).
"""
import torch
from torch import nn
from torch.nn import functional as F

from ..autograd import linear as _linear, residual_block as _residual_block


class ResidualBlock(nn.Module):
    """x + Linear(drop(act(Linear(act(x))))), optionally GLU-gated by a context
    projection (resnet.py:38-57)."""

    def __init__(self, features, context_features, activation=F.relu,
                 dropout_probability=0., use_batch_norm=False, zero_initialization=True):
        super().__init__()
        self.activation = activation
        self.use_batch_norm = use_batch_norm
        if use_batch_norm:
            self.batch_norm_layers = nn.ModuleList(
                [nn.BatchNorm1d(features, eps=1e-3) for _ in range(2)])
        if context_features is not None:
            self.context_layer = nn.Linear(context_features, features)
        self.linear_layers = nn.ModuleList([nn.Linear(features, features) for _ in range(2)])
        self.dropout = nn.Dropout(p=dropout_probability)
        if zero_initialization:                       # resnet.py:34-36
            for p in (self.linear_layers[1].weight, self.linear_layers[1].bias):
                nn.init.uniform_(p, -1e-3, 1e-3)

    def forward(self, inputs, context=None):
        fused = _residual_block(self, inputs, context)      # large training batches: one autograd node, fused maps
        if fused is not None:
            return fused
        h = inputs
        for i in range(2):
            if self.use_batch_norm:
                h = self.batch_norm_layers[i](h)
            h = self.activation(h)
            if i == 1:
                h = self.dropout(h)
            h = _linear(self.linear_layers[i], h)
        if context is not None:
            gate = _linear(self.context_layer, context)
            if torch.is_grad_enabled() and (h.requires_grad or gate.requires_grad) and h.shape[0] >= 32768:
                # large training batches: glu(cat(h, gate)) = h * sigmoid(gate) without materialising the concatenation
                # and its backward slices (resnet.py:54-56; same arithmetic; more, smaller launches: not for small batches)
                h = h * torch.sigmoid(gate)
            else:
                h = F.glu(torch.cat((h, gate), dim=1), dim=1)
        return inputs + h


class ResidualNet(nn.Module):
    """Linear -> num_blocks x ResidualBlock -> Linear on flat feature vectors."""

    def __init__(self, in_features, out_features, hidden_features, context_features=None,
                 num_blocks=2, activation=F.relu, dropout_probability=0.,
                 use_batch_norm=False, preprocessing=None):
        super().__init__()
        self.hidden_features = hidden_features      # read by the coupling for the 1/sqrt(H) logit scale
        self.context_features = context_features
        self.preprocessing = preprocessing
        first_in = in_features + (context_features or 0)
        self.initial_layer = nn.Linear(first_in, hidden_features)
        self.blocks = nn.ModuleList([
            ResidualBlock(hidden_features, context_features, activation=activation,
                          dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)
            for _ in range(num_blocks)])
        self.final_layer = nn.Linear(hidden_features, out_features)

    def forward(self, inputs, context=None):
        h = inputs if self.preprocessing is None else self.preprocessing(inputs)
        if context is not None:
            h = torch.cat((h, context), dim=1)
        return self.trunk(h, context)

    def hidden(self, first_in, context=None):
        """Output of the last residual block (the input of ``final_layer``) for an already
        concatenated (identity | context) input; the last layer itself can then run inside the
        spline kernel (csrc/fused_final.hip)."""
        h = _linear(self.initial_layer, first_in)
        for block in self.blocks:
            h = block(h, context=context)
        return h

    def trunk(self, first_in, context=None):
        """Everything after the (identity | context) concatenation; the RQS
        coupling calls this directly with the buffer its gather kernel wrote."""
        h = _linear(self.initial_layer, first_in)
        for block in self.blocks:
            h = block(h, context=context)
        return _linear(self.final_layer, h)


class ConvResidualBlock(nn.Module):
    """Image form of ResidualBlock: two 3x3 convolutions, optional GLU gate from a 1x1
    projection of the context image (resnet.py:109-160)."""

    def __init__(self, channels, context_channels=None, activation=F.relu,
                 dropout_probability=0., use_batch_norm=False, zero_initialization=True):
        super().__init__()
        self.activation = activation
        if context_channels is not None:
            self.context_layer = nn.Conv2d(context_channels, channels, kernel_size=1, padding=0)
        self.use_batch_norm = use_batch_norm
        if use_batch_norm:
            self.batch_norm_layers = nn.ModuleList([nn.BatchNorm2d(channels, eps=1e-3) for _ in range(2)])
        self.conv_layers = nn.ModuleList([nn.Conv2d(channels, channels, kernel_size=3, padding=1) for _ in range(2)])
        self.dropout = nn.Dropout(p=dropout_probability)
        if zero_initialization:
            for p in (self.conv_layers[1].weight, self.conv_layers[1].bias):
                nn.init.uniform_(p, -1e-3, 1e-3)

    def forward(self, inputs, context=None):
        h = inputs
        for i in range(2):
            if self.use_batch_norm:
                h = self.batch_norm_layers[i](h)
            h = self.activation(h)
            if i == 1:
                h = self.dropout(h)
            h = self.conv_layers[i](h)
        if context is not None:
            h = F.glu(torch.cat((h, self.context_layer(context)), dim=1), dim=1)
        return inputs + h


class ConvResidualNet(nn.Module):
    """1x1 conv -> num_blocks x ConvResidualBlock -> 1x1 conv: conditioner of the image-shaped
    RQS coupling (resnet.py:163-212).  Convolutions run on PyTorch-ROCm (MIOpen)."""

    def __init__(self, in_channels, out_channels, hidden_channels, context_channels=None,
                 num_blocks=2, activation=F.relu, dropout_probability=0., use_batch_norm=False):
        super().__init__()
        self.context_channels = context_channels
        self.hidden_channels = hidden_channels      # read by the coupling for the 1/sqrt(H) logit scale
        self.initial_layer = nn.Conv2d(in_channels + (context_channels or 0), hidden_channels, kernel_size=1, padding=0)
        self.blocks = nn.ModuleList([
            ConvResidualBlock(hidden_channels, context_channels, activation=activation,
                              dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)
            for _ in range(num_blocks)])
        self.final_layer = nn.Conv2d(hidden_channels, out_channels, kernel_size=1, padding=0)

    def forward(self, inputs, context=None):
        h = inputs if context is None else torch.cat((inputs, context), dim=1)
        h = self.initial_layer(h)
        for block in self.blocks:
            h = block(h, context)
        return self.final_layer(h)
