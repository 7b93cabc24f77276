import torch


def sum_except_batch(x, num_batch_dims=1):
    """Reference: normflow/utils/nn.py:131-134.  Inside the coupling kernels this
    reduction is fused (LDS staging + wave shuffles); this torch form is only for
    callers that hold a materialised per-element log-det."""
    return torch.sum(x, dim=list(range(num_batch_dims, x.dim())))


class PeriodicFeatures(torch.nn.Module):
    """Replaces the input columns ``ind`` by w0 sin(scale x) + w1 cos(scale x) (+ bias), so a
    conditioner sees circular coordinates through periodic features (utils/nn.py:59-118).
    Buffers / parameters as in the reference: ``ind``, ``ind_``, ``inv_perm``, ``weights`` [n, 2],
    optional ``scale`` buffer and ``bias``.  Runs on PyTorch-ROCm: a handful of columns."""

    def __init__(self, ndim, ind, scale=1., bias=False, activation=None):
        super().__init__()
        self.ndim = ndim
        ind = torch.as_tensor(ind, dtype=torch.long)
        self.register_buffer('ind', ind)
        chosen = set(int(i) for i in ind)
        rest = torch.tensor([i for i in range(ndim) if i not in chosen], dtype=torch.long)
        self.register_buffer('ind_', rest)
        self.register_buffer('inv_perm', torch.argsort(torch.cat((ind, rest))))
        self.weights = torch.nn.Parameter(torch.ones(len(ind), 2))
        if torch.is_tensor(scale):
            self.register_buffer('scale', scale)
        else:
            self.scale = scale
        self.apply_bias = bias
        if bias:
            self.bias = torch.nn.Parameter(torch.zeros(len(ind)))
        self.activation = activation

    def forward(self, inputs):
        ang = self.scale * inputs[..., self.ind]
        feat = self.weights[:, 0] * torch.sin(ang) + self.weights[:, 1] * torch.cos(ang)
        if self.apply_bias:
            feat = feat + self.bias
        if self.activation is not None:
            feat = self.activation(feat)
        return torch.cat((feat, inputs[..., self.ind_]), -1)[..., self.inv_perm]
