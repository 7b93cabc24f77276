"""Deterministic synthetic weights shared by the fixture generator and the tests.

The golden fixtures do not store conditioner weights (a 12-layer C3 stack is
8 MB): they store the ordered list of float state-dict entries (name, shape)
and a seed, and both sides rebuild the identical fp32 weights from numpy's
PCG64 stream.  Integer buffers (feature indices, permutations) are stored in
the fixture verbatim, since they are parity items themselves.
"""
import numpy as np
import torch


def synth_tensor(rng, name, shape, final_gain=6.0, weight_gain=1.0, other_gain=0.2):
    """One fp32 tensor for state-dict entry ``name``.  Scales are chosen so the
    splines see varied bins (final conditioner layer boosted; unconditional
    spline logits N(0, 0.5^2) instead of the reference's all-equal identity
    init, which would hide bin-search differences)."""
    shape = tuple(int(s) for s in shape)
    z = rng.standard_normal(shape).astype(np.float32)
    leaf = name.split(".")[-1]
    if "unnormalized_" in name:
        z *= 0.5
    elif leaf == "weight" and len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        gain = final_gain if "final_layer" in name else weight_gain
        z *= np.float32(gain / np.sqrt(fan_in))
    elif leaf == "bias":
        z *= 0.1
    elif leaf in ("loc", "log_scale", "s", "t"):
        z *= 0.3
    else:
        z *= np.float32(other_gain)      # e.g. the L / U entries and log_S of an LU 1x1 convolution
    return torch.from_numpy(z)


def synth_state(entries, seed, final_gain=6.0, weight_gain=1.0, other_gain=0.2):
    """``entries``: iterable of (name, shape) in state-dict order.  ``final_gain``
    scales the last conditioner layer of RQS couplings (6: wild logits, an
    ill-conditioned stress case; 1.5: well-conditioned), ``weight_gain`` every
    other weight matrix."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return {n: synth_tensor(rng, n, s, final_gain, weight_gain, other_gain) for n, s in entries}


def float_entries(module):
    return [(k, tuple(v.shape)) for k, v in module.state_dict().items() if v.is_floating_point()]


def int_buffers(module):
    return {k: v.clone() for k, v in module.state_dict().items() if not v.is_floating_point()}


def load_synth(module, seed, skip=(), final_gain=6.0, weight_gain=1.0, other_gain=0.2):
    """Overwrite every floating-point state entry of ``module`` (except names in
    ``skip``) with synthetic values; returns the entry list used."""
    ents = [(k, s) for k, s in float_entries(module) if k not in skip]
    sd = synth_state(ents, seed, final_gain, weight_gain, other_gain)
    cur = module.state_dict()
    for k, v in sd.items():
        cur[k].copy_(v.to(cur[k].dtype))
    return ents


def encode_entries(ents):
    return np.array(["%s|%s" % (n, ",".join(str(int(x)) for x in s)) for n, s in ents])


def decode_entries(arr):
    out = []
    for item in arr.tolist():
        n, s = item.split("|")
        out.append((n, tuple(int(x) for x in s.split(",")) if s else ()))
    return out
