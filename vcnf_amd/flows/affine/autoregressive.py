"""Masked affine autoregressive flow (MAF): every variable is scaled and shifted by parameters a MADE network
computes from the variables of lower degree.  ``forward`` is one MADE pass + one launch of the elementwise kernel
(csrc/affine_kernels.hip::maf_affine_kernel reads the MADE output [B, D * 2] in place); ``inverse`` needs D
sequential passes (variable i can only be inverted once its predecessors are), as in the reference.
Reference: normflow/flows/affine/autoregressive.py:11-45 (Autoregressive), :48-103 (MaskedAffineAutoregressive)."""
import numpy as np
import torch
from torch.nn import functional as F

from ..base import Flow
from ... import _lib, autograd
from ...nets.made import MADE


class Autoregressive(Flow):
    """autoregressive.py:11-45: elementwise invertible map whose parameters come from ``autoregressive_net``."""
    takes_context = True

    def __init__(self, autoregressive_net):
        super().__init__()
        self.autoregressive_net = autoregressive_net

    def forward(self, inputs, context=None):
        return self._elementwise_forward(inputs, self.autoregressive_net(inputs, context))

    def inverse(self, inputs, context=None):
        outputs = torch.zeros_like(inputs)
        logabsdet = None
        for _ in range(int(np.prod(inputs.shape[1:]))):
            outputs, logabsdet = self._elementwise_inverse(inputs, self.autoregressive_net(outputs, context))
        return outputs, logabsdet

    def _output_dim_multiplier(self):
        raise NotImplementedError()

    def _elementwise_forward(self, inputs, autoregressive_params):
        raise NotImplementedError()

    def _elementwise_inverse(self, inputs, autoregressive_params):
        raise NotImplementedError()


class MaskedAffineAutoregressive(Autoregressive):
    def __init__(self, features, hidden_features, context_features=None, num_blocks=2, use_residual_blocks=True,
                 random_mask=False, activation=F.relu, dropout_probability=0., use_batch_norm=False):
        self.features = features
        made = MADE(features=features, hidden_features=hidden_features, context_features=context_features,
                    num_blocks=num_blocks, output_multiplier=self._output_dim_multiplier(),
                    use_residual_blocks=use_residual_blocks, random_mask=random_mask, activation=activation,
                    dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)
        super().__init__(made)

    def _output_dim_multiplier(self):
        return 2

    def _elementwise(self, inputs, params, inverse):
        if inputs.dim() != 2 or inputs.shape[1] != self.features:
            raise ValueError('Expected inputs [B, {}], got {}.'.format(self.features, tuple(inputs.shape)))
        if autograd.needs_grad(inputs, params):
            # differentiable path: the reference's own composition (:75-89), autograd through torch ops
            p = params.view(-1, self.features, 2)
            scale = torch.sigmoid(p[..., 0] + 2.) + 1e-3
            log_scale = torch.log(scale).sum(1)
            if inverse:
                return (inputs - p[..., 1]) / scale, -log_scale
            return scale * inputs + p[..., 1], log_scale
        return _lib.maf_affine(inputs, params, inverse)

    def _elementwise_forward(self, inputs, autoregressive_params):
        return self._elementwise(inputs, autoregressive_params, False)

    def _elementwise_inverse(self, inputs, autoregressive_params):
        return self._elementwise(inputs, autoregressive_params, True)
