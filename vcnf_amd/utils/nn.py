import torch


def sum_except_batch(x, num_batch_dims=1):
    """Reference: normflow/utils/nn.py:131-134.  Inside the coupling kernels this
    reduction is fused (LDS staging + wave shuffles); this torch form is only for
    callers that hold a materialised per-element log-det."""
    return torch.sum(x, dim=list(range(num_batch_dims, x.dim())))
