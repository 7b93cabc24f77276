"""ctypes binding of ``libvcnf_hip.so`` (the C ABI declared in include/vcnf_hip.h)
and thin tensor-level wrappers around each entry point.

There is no CPU implementation behind these wrappers: a CPU tensor, a missing
library or a non-zero status raises.  PyTorch is used for device memory and the
current HIP stream only.
"""
import ctypes
import math
import os

import torch

from . import build as _build

_P = ctypes.c_void_p
_I32, _I64, _F32, _INT = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_int
_F64 = ctypes.c_double

OK = 0
LD_STORE, LD_ACCUM = 0, 1
TAILS_NONE, TAILS_LINEAR, TAILS_CIRCULAR = 0, 1, 2
SCALE_EXP, SCALE_SIGMOID, SCALE_SIGMOID_INV, SCALE_NONE = 0, 1, 2, 3
SCALE_MAPS = {"exp": SCALE_EXP, "sigmoid": SCALE_SIGMOID, "sigmoid_inv": SCALE_SIGMOID_INV}


class RqsCfg(ctypes.Structure):
    """struct vcnf_rqs_cfg"""
    _fields_ = [("num_bins", _I32), ("tails", _I32), ("left", _F32), ("right", _F32),
                ("bottom", _F32), ("top", _F32), ("min_bin_width", _F32),
                ("min_bin_height", _F32), ("min_derivative", _F32), ("wh_scale", _F32)]


class RqsCfg64(ctypes.Structure):
    """struct vcnf_rqs_cfg_f64"""
    _fields_ = [("num_bins", _I32), ("tails", _I32), ("left", _F64), ("right", _F64),
                ("bottom", _F64), ("top", _F64), ("min_bin_width", _F64),
                ("min_bin_height", _F64), ("min_derivative", _F64), ("wh_scale", _F64)]


class RqsStackLayer(ctypes.Structure):
    """struct vcnf_rqs_stack_layer"""
    _fields_ = [("transform_idx", ctypes.c_void_p), ("identity_idx", ctypes.c_void_p), ("wpack", ctypes.c_void_p),
                ("shared_w", ctypes.c_void_p), ("shared_h", ctypes.c_void_p), ("shared_d", ctypes.c_void_p)]


# name -> argtypes, exactly the prototypes of include/vcnf_hip.h
PROTOTYPES = {
    "vcnf_abi_version": ([], _INT),
    "vcnf_status_string": ([_INT], ctypes.c_char_p),
    "vcnf_rqs_elementwise_f32": ([_P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _I64,
                                  ctypes.POINTER(RqsCfg), _INT, _P, _P], _INT),
    "vcnf_rqs_elementwise_strided_f32": ([_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _P, _P, _I64,
                                          ctypes.POINTER(RqsCfg), _INT, _P, _P], _INT),
    "vcnf_rqs_elementwise_bwd_f32": ([_P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _I64,
                                      ctypes.POINTER(RqsCfg), _INT, _P], _INT),
    "vcnf_affine_layer_fused_supported": ([_I32, _I32, _I32, _I32], _INT),
    "vcnf_affine_stack_fused_supported": ([_I32, _I32, _I32, _I32], _INT),
    "vcnf_affine_layer_fused_pack_floats": ([_I32, _I32, _I32], _I64),
    "vcnf_affine_layer_fused_f32": ([_P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _F32, _INT, _P, _I64,
                                     _P, _P, _INT, _INT, _F32, _P], _INT),
    "vcnf_rqs_shared_f32": ([_P, _P, _P, _P, _I64, _P, _P, _P, _I64, ctypes.POINTER(RqsCfg), _INT, _P, _P], _INT),
    "vcnf_rqs_final_fused_supported": ([_I32, _I32, _I32, _I32], _INT),
    "vcnf_rqs_final_fused_pack_floats": ([_I32, _I32, _I32], _I64),
    "vcnf_rqs_final_fused_partial_rows": ([_I32, _I32], _I64),
    "vcnf_rqs_final_fused_f32": ([_P, _P, _P, _P, _I64, _I32, _P, _I32, _I32, _P, _I64, ctypes.POINTER(RqsCfg), _INT,
                                  _P, _P], _INT),
    "vcnf_rqs_packed_bwd_f32": ([_P, _P, _I64, _I64, _P, _P, _P, _P, _I64, ctypes.POINTER(RqsCfg), _INT, _P], _INT),
    "vcnf_rqs_shared_bwd_groups": ([_I64, _I64], _I64),
    "vcnf_rqs_shared_bwd_f32": ([_P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _P, _P, _I64,
                                 ctypes.POINTER(RqsCfg), _INT, _P], _INT),
    "vcnf_rqs_coupling_f32": ([_P, _P, _P, _I32, _P, _I32, _P, _P, _P, _P, _P, _I64,
                               ctypes.POINTER(RqsCfg), _INT, _INT, _F32, _P, _P], _INT),
    "vcnf_rqs_conditioner_input_f32": ([_P, _I64, _I32, _P, _I32, _P, _I32, _P, _P, _P,
                                        ctypes.POINTER(RqsCfg), _INT, _P, _P], _INT),
    "vcnf_conv1x1_supported": ([_I32, _I32], _INT),
    "vcnf_conv1x1_pack_floats": ([_I32, _I32], _I64),
    "vcnf_conv1x1_f16x3_f32": ([_P, _P, _P, _I64, _P, _P, _I64, _I32, _I32, _I64, _INT, _F32, _INT, _F32, _P, _P], _INT),
    "vcnf_linear_wgrad_supported": ([_I32, _I32], _INT),
    "vcnf_linear_wgrad_slices": ([_I64, _I32, _I32], _I64),
    "vcnf_linear_wgrad_f32": ([_P, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _INT, _P], _INT),
    "vcnf_linear_wgrad_f16x3_f32": ([_P, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _INT, _INT, _P, _P], _INT),
    "vcnf_conv3x3_1x1_supported": ([_I32, _I32, _I32], _INT),
    "vcnf_conv3x3_1x1_pack_floats": ([_I32], _I64),
    "vcnf_conv3x3_1x1_f16x3_f32": ([_P, _P, _P, _I64, _P, _I64, _P, _P, _I64, _I32, _I32, _I32, _F32, _F32, _P, _P], _INT),
    "vcnf_convnet3_supported": ([_I32, _I32, _I32], _INT),
    "vcnf_convnet3_w3_pack_floats": ([_I32], _I64),
    "vcnf_convnet3_taps_f16x3_f32": ([_P, _P, _P, _I64, _P, _I64, _P, _I64, _P, _P, _I64, _I32, _I32, _I32, _I32, _F32, _F32,
                                      _P, _P], _INT),
    "vcnf_col2im3x3_f32": ([_P, _P, _P, _I64, _I32, _I32, _I32, _P], _INT),
    "vcnf_resblock_elementwise_f32": ([_INT, _P, _P, _P, _P, _P, _I64, _P], _INT),
    "vcnf_channel_mix_supported": ([_I32], _INT),
    "vcnf_channel_mix_f32": ([_P, _P, _P, _P, _I64, _I32, _I64, _P], _INT),
    "vcnf_rqs_identity_half_supported": ([_I32, _I32], _INT),
    "vcnf_rqs_identity_half_partial_rows": ([_I32], _I64),
    "vcnf_rqs_identity_half_f32": ([_P, _P, _P, _P, _I64, _I32, _P, _I32, _P, _P, _P, ctypes.POINTER(RqsCfg), _INT, _INT,
                                    _P, _P], _INT),
    "vcnf_rqs_layer_fused_pack_floats": ([_I32, _I32, _I32, _I32], _I64),
    "vcnf_rqs_layer_fused_supported": ([_I32, _I32, _I32, _I32, _I32, _I32, _I32], _INT),
    "vcnf_rqs_layer_fused_tile_rows": ([], _I32),
    "vcnf_rqs_layer_fused_small_batch_rows": ([_I64], _I64),
    "vcnf_masked_affine_stack_supported": ([_I32, _I32], _INT),
    "vcnf_masked_affine_stack_f32": ([_P, _P, _P, _P, _I64, _I32, _I32, _INT, _INT, _F32, _P], _INT),
    "vcnf_masked_affine_stack_f64": ([_P, _P, _P, _P, _I64, _I32, _I32, _INT, _INT, _F64, _P], _INT),
    "vcnf_masked_affine_stack_bwd_f32": ([_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _INT, _P], _INT),
    "vcnf_masked_affine_stack_bwd_f64": ([_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _INT, _P], _INT),
    "vcnf_linear_f16x3_supported": ([_I32, _I32], _INT),
    "vcnf_linear_f16x3_f32": ([_P, _P, _P, _P, _I64, _I32, _I32, _I64, _I64, _INT, _INT, _P, _P, _P, _P], _INT),
    "vcnf_rqs_stack_fused_max_layers": ([], _I32),
    "vcnf_rqs_stack_fused_f32": ([_P, _P, _P, _P, _I64, ctypes.POINTER(RqsStackLayer), _I32, _I32, _I32, _I32, _I32, _I32,
                                  _I32, _I64, ctypes.POINTER(RqsCfg), _INT, _INT, _F32, _P, _P, _P, _P], _INT),
    "vcnf_rqs_layer_fused_f32": ([_P, _P, _P, _P, _I64, _P, _I32, _P, _I32, _I32, _I32, _I32, _I32, _P, _I64,
                                  _P, _P, _P, ctypes.POINTER(RqsCfg), _INT, _INT, _F32, _P, _P, _P, _P], _INT),
    "vcnf_affine_coupling_f32": ([_P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _INT, _INT,
                                  _INT, _F32, _P], _INT),
    "vcnf_resnet_trunk_supported": ([_I32, _I32, _I32], _INT),
    "vcnf_resnet_trunk_pack_floats": ([_I32, _I32, _I32], _I64),
    "vcnf_resnet_trunk_f32": ([_P, _P, _I64, _I32, _I32, _I32, _P, _I64, _P], _INT),
    "vcnf_resnet_trunk_split_f32": ([_P, _P, _I64, _I32, _I32, _I32, _P, _I64, _P, _P], _INT),
    "vcnf_rqs_final_fused_presplit_f32": ([_P, _P, _P, _P, _I64, _I32, _P, _I32, _I32, _P, _I64, ctypes.POINTER(RqsCfg), _INT,
                                           _P, _P], _INT),
    "vcnf_affine_stack_fused_f32": ([_P, _P, _P, _I64, _I32, _I32, _P, _I32, _I32, _I32, _F32, _INT, _P, _I64, _P, _I32,
                                     _INT, _INT, _F32, _P], _INT),
    "vcnf_affine_layer_fused_h3_pack_floats": ([_I32, _I32, _I32], _I64),
    "vcnf_affine_stack_fused_f16x3_f32": ([_P, _P, _P, _I64, _I32, _I32, _P, _I32, _I32, _I32, _F32, _INT, _P, _I64, _P, _I64,
                                           _P, _I32, _INT, _INT, _F32, _P, _P], _INT),
    "vcnf_maf_affine_f32": ([_P, _P, _P, _P, _I64, _I32, _INT, _INT, _F32, _P], _INT),
    "vcnf_masked_affine_f32": ([_P, _P, _P, _P, _P, _P, _I64, _I32, _INT, _INT, _F32, _P], _INT),
    "vcnf_affine_const_f32": ([_P, _P, _P, _P, _I64, _I32, _I32, _INT, _P], _INT),
    "vcnf_permute_f32": ([_P, _P, _P, _I64, _I32, _I32, _P], _INT),
    "vcnf_split_columns_f32": ([_P, _P, _P, _P, _I64, _I32, _I32, _P], _INT),
    "vcnf_merge_columns_f32": ([_P, _P, _P, _P, _I64, _I32, _I32, _P], _INT),
    "vcnf_diag_gaussian_log_prob_f32": ([_P, _P, _P, _F32, _P, _I64, _I32, _INT, _F32, _P], _INT),
    "vcnf_diag_gaussian_sample_f32": ([_P, _P, _P, _F32, _P, _P, _I64, _I32, _P], _INT),
    "vcnf_linear_probe_f32": ([_P, _P, _P, _P, _I64, _I32, _I32, _INT, _INT, _P, _P], _INT),
    "vcnf_rqs_elementwise_f64": ([_P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _I64,
                                  ctypes.POINTER(RqsCfg64), _INT, _P, _P], _INT),
    "vcnf_affine_coupling_f64": ([_P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _INT, _INT, _INT, _F64, _P], _INT),
    "vcnf_masked_affine_f64": ([_P, _P, _P, _P, _P, _P, _I64, _I32, _INT, _INT, _F64, _P], _INT),
    "vcnf_affine_const_f64": ([_P, _P, _P, _P, _I64, _I32, _I32, _INT, _P], _INT),
    "vcnf_permute_f64": ([_P, _P, _P, _I64, _I32, _I32, _P], _INT),
    "vcnf_split_columns_f64": ([_P, _P, _P, _P, _I64, _I32, _I32, _P], _INT),
    "vcnf_merge_columns_f64": ([_P, _P, _P, _P, _I64, _I32, _I32, _P], _INT),
    "vcnf_diag_gaussian_log_prob_f64": ([_P, _P, _P, _F64, _P, _I64, _I32, _INT, _F64, _P], _INT),
    "vcnf_diag_gaussian_sample_f64": ([_P, _P, _P, _F64, _P, _P, _I64, _I32, _P], _INT),
}

_LIB = None


class VcnfError(RuntimeError):
    pass


def lib_path():
    """The in-tree library; VCNF_LIB selects another build of it (kernel variants of a measurement, profiles/tools)."""
    return os.environ.get("VCNF_LIB") or _build.LIB


def lib():
    """Load (once) and return the ctypes handle.  Raises if the library was not
    built - there is no fallback path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise VcnfError("HIP library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(or python -m vcnf_amd.build); vcnf_amd has no CPU fallback" % path)
    # torch ships its own libamdhip64 (soname libamdhip64.so.7); make sure that one is
    # the HIP runtime our library binds to, so streams and pointers are shared with torch.
    hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(hip_rt):
        ctypes.CDLL(hip_rt, mode=ctypes.RTLD_GLOBAL)
    handle = ctypes.CDLL(path)
    for name, (args, res) in PROTOTYPES.items():
        fn = getattr(handle, name)      # AttributeError if the ABI is incomplete
        fn.argtypes, fn.restype = args, res
    if handle.vcnf_abi_version() != 1:
        raise VcnfError("libvcnf_hip.so ABI version mismatch")
    _LIB = handle
    return handle


def _check(status, what):
    if status != OK:
        msg = lib().vcnf_status_string(status).decode()
        if status == 4:       # splines.py:104-107 raises ValueError
            raise ValueError(msg.capitalize())
        raise VcnfError("%s: %s (status %d)" % (what, msg, status))


def _stream():
    # raw handle of the current stream of the current device (torch.cuda.current_stream() builds a Stream object
    # and resolves the device through three layers of helpers: 8 us per call, several calls per layer)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def require_device(*tensors, allow_grad=False, f64=False):
    """Every tensor must be an fp32 tensor on one HIP device.  Autograd reaches the
    kernels only through vcnf_amd.autograd (the spline VJP kernel); a plain wrapper
    called with a tensor that requires grad raises rather than silently dropping it."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise VcnfError("vcnf_amd computes on MI355X only (tensor on %s); there is no CPU path" % t.device)
        if t.is_floating_point() and t.dtype != torch.float32 and not (f64 and t.dtype == torch.float64):
            raise VcnfError("this vcnf_amd kernel is fp32%s (got %s)" % (" / fp64" if f64 else "", t.dtype))
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise VcnfError("tensors on different devices: %s vs %s" % (dev, t.device))
        if not allow_grad and torch.is_grad_enabled() and t.requires_grad:
            raise NotImplementedError(
                "vcnf_amd: this HIP kernel has no backward pass; evaluate under torch.no_grad() "
                "(differentiable: the RQS couplings via vcnf_amd.autograd, SURVEY 8f row 1)")
    return dev


def _sfx(t):
    """Entry-point suffix of a tensor's dtype: the elementwise kernels exist for fp32 and fp64."""
    return "_f64" if t.dtype == torch.float64 else "_f32"


def n_derivatives(cfg):
    """Derivative logits per element: K-1 (linear tails), K (circular), K+1 (no tails)."""
    k = cfg.num_bins
    return k - 1 if cfg.tails == TAILS_LINEAR else k if cfg.tails == TAILS_CIRCULAR else k + 1


def make_cfg(num_bins, tails, tail_bound=1.0, left=0.0, right=1.0, bottom=0.0, top=1.0,
             min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3, wh_scale=1.0):
    if tails == "linear":
        left, right, bottom, top = -tail_bound, tail_bound, -tail_bound, tail_bound
        mode = TAILS_LINEAR
    elif tails == "circular":
        left, right, bottom, top = -tail_bound, tail_bound, -tail_bound, tail_bound
        mode = TAILS_CIRCULAR
    elif tails is None:
        mode = TAILS_NONE
    else:
        # per-feature tail lists: SURVEY 8f row 4, not built
        raise RuntimeError("{} tails are not implemented.".format(tails))
    cfg = RqsCfg(int(num_bins), mode, float(left), float(right), float(bottom), float(top),
                 float(min_bin_width), float(min_bin_height), float(min_derivative), float(wh_scale))
    # the same constants at full double precision for the fp64 spline (1e-3 is not a float)
    cfg.f64 = RqsCfg64(int(num_bins), mode, float(left), float(right), float(bottom), float(top),
                       float(min_bin_width), float(min_bin_height), float(min_derivative), float(wh_scale))
    return cfg


_BAD = {}

# bench.py hook: when set to a list, rqs_coupling brackets its kernel launch with a
# pair of HIP events on the launch stream and appends (start, end, batch) to it.
EVENT_SINK = None


def device_index(device):
    """Explicit index of a CUDA device: 'cuda' without an index means the CURRENT device (the counters below
    are kept per device; a rank whose current device is not 0 must not read device 0's)."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise VcnfError("vcnf_amd runs on HIP devices only (got %s)" % dev)
    return torch.cuda.current_device() if dev.index is None else dev.index


def _counter(table, device):
    key = device_index(device)
    if key not in table:
        table[key] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", key))
    return table[key]


class _timed:
    """bench.py hook: with EVENT_SINK set to a list, brackets a kernel launch with HIP events on the launch stream
    (torch's current stream = the stream the kernel is enqueued on) and appends (start, end, tag)."""

    def __init__(self, tag):
        self.tag, self.sink = tag, EVENT_SINK

    def __enter__(self):
        if self.sink is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.sink is not None:
            self.e1.record()
            self.sink.append((self.e0, self.e1, self.tag))
        return False


def bad_discriminant_counter(device):
    """Device int32 that the inverse spline kernels bump when b^2-4ac < 0 (the
    reference asserts on the host, splines.py:164)."""
    return _counter(_BAD, device)


_SAT = {}


def saturation_counter(device):
    """Device int32 that the fp16 split-half fused layer kernel bumps (once per workgroup) when an input, a
    context value or a hidden activation was clamped at +-65504."""
    return _counter(_SAT, device)


_REDO = {}


def range_redo_counter(device):
    """Device int32: tiles (128 samples) of fused RQS layers that held a value the fp16 split-half operands cannot
    carry and were therefore evaluated by the exact fp32 kernel instead (never clamped)."""
    return _counter(_REDO, device)


def range_redo_count(device="cuda"):
    """Host read (synchronises) and reset of ``range_redo_counter``: a statistic, not an error - those tiles have
    exact fp32 results."""
    c = range_redo_counter(device)
    n = int(c.item())
    c.zero_()
    return n


def check_discriminant(device="cuda"):
    """Host check (synchronises): raises AssertionError like splines.py:164 if an
    inverse spline saw a negative discriminant since the last check."""
    c = bad_discriminant_counter(device)
    n = int(c.item())
    c.zero_()
    assert n == 0, "negative discriminant in %d workgroup(s) of an inverse RQ spline" % n


def check_saturation(device="cuda", model=None):
    """Host check (synchronises) of the fp16 split-half matrix path: number of workgroups that clamped a value
    at the fp16 range since the last check (the reference is plain fp32 and does not clamp, nets/resnet.py:92-106).
    With ``model`` given and a non-zero count every fused RQS coupling of the model is switched to the exact
    fp32 matrix path (``fused_precision = 'fp32'``), so that re-running the evaluation gives reference results;
    and the training path's dense layers go back to the library's fp32 GEMMs (``autograd.TRAIN_MATRIX_PATH``);
    without a model a non-zero count raises VcnfError."""
    c = saturation_counter(device)
    n = int(c.item())
    c.zero_()
    if n and model is None:
        raise VcnfError("fp16 split-half matrix path clamped values at +-65504 in %d workgroup(s): "
                        "set fused_precision = 'fp32' on the couplings (or pass model=...) and re-run" % n)
    if n:
        from . import autograd
        autograd.TRAIN_MATRIX_PATH = "fp32"           # training path: the library's fp32 GEMMs (csrc/linear_f16x3.hip clamps too)
        for m in model.modules():
            if hasattr(m, "fused_precision"):
                m.fused_precision = "fp32"
            if hasattr(m, "fused_conv1x1"):           # ConvNet2d: back to the library's fp32 convolutions
                m.fused_conv1x1 = False
    return n


# ---------------------------------------------------------------- wrappers
def rqs_elementwise(x, uw, uh, ud, cfg, inverse, allow_grad=False):
    """x [...]; uw, uh [..., K]; ud [..., K-1 | K+1] (last dim contiguous)."""
    dev = require_device(x, uw, uh, ud, allow_grad=allow_grad, f64=True)
    shape = x.shape
    k = cfg.num_bins
    nd = n_derivatives(cfg)
    if uw.shape != shape + (k,) or uh.shape != shape + (k,) or ud.shape != shape + (nd,):
        raise VcnfError("spline parameter shapes %s %s %s do not match inputs %s with K=%d" % (
            tuple(uw.shape), tuple(uh.shape), tuple(ud.shape), tuple(shape), k))
    xf = x.reshape(-1).contiguous()

    def rows(t, width):
        t2 = t.reshape(-1, width)
        if t2.stride(1) != 1 or (t2.shape[0] > 1 and t2.stride(0) < width):
            t2 = t2.contiguous()
        return t2, (t2.stride(0) if t2.shape[0] > 1 else width)
    w2, ldw = rows(uw, k)
    h2, ldh = rows(uh, k)
    d2, ldd = rows(ud, nd)
    y = torch.empty_like(xf)
    lad = torch.empty_like(xf)
    f64 = xf.dtype == torch.float64
    if f64 and not all(t.dtype == torch.float64 for t in (w2, h2, d2)):
        raise VcnfError("fp64 spline: inputs and logits must all be fp64")
    with torch.cuda.device(dev):
        fn = lib().vcnf_rqs_elementwise_f64 if f64 else lib().vcnf_rqs_elementwise_f32
        st = fn(_ptr(xf), _ptr(w2), _ptr(h2), _ptr(d2), ldw, ldh, ldd,
                _ptr(y), _ptr(lad), xf.numel(), ctypes.byref(cfg.f64 if f64 else cfg),
                int(bool(inverse)),
                _ptr(bad_discriminant_counter(dev)) if inverse else None, _stream())
    _check(st, "vcnf_rqs_elementwise" + _sfx(xf))
    return y.view(shape), lad.view(shape)


def rqs_elementwise_image(x, params, cfg, inverse, allow_grad=False):
    """x [B, C, *inner] with the conditioner output params [B, C*P, *inner] read in place
    (the reference reshapes/permutes it to [B, C, *inner, P]; coupling.py:148-151)."""
    dev = require_device(x, params, allow_grad=allow_grad)
    k, nd = cfg.num_bins, n_derivatives(cfg)
    p = 2 * k + nd
    b, c = x.shape[0], x.shape[1]
    inner = int(x[0, 0].numel())
    if tuple(params.shape) != (b, c * p) + tuple(x.shape[2:]):
        raise VcnfError("conditioner output %s does not match inputs %s with %d logits per element" % (
            tuple(params.shape), tuple(x.shape), p))
    x = x.contiguous()
    params = params.contiguous()
    y, lad = torch.empty_like(x), torch.empty_like(x)
    base = params.data_ptr()
    with torch.cuda.device(dev):
        st = lib().vcnf_rqs_elementwise_strided_f32(
            _ptr(x), base, base + 4 * k * inner, base + 8 * k * inner, p * inner, p * inner, p * inner,
            inner, inner, 0, _ptr(y), _ptr(lad), x.numel(), ctypes.byref(cfg), int(bool(inverse)),
            _ptr(bad_discriminant_counter(dev)) if inverse else None, _stream())
    _check(st, "vcnf_rqs_elementwise_strided_f32")
    return y, lad


def rqs_elementwise_shared(x, uw, uh, ud, cfg, inverse, allow_grad=False):
    """x [B, *shape] with one logit row per position of ``shape`` shared by the whole batch:
    uw, uh [*shape, K], ud [*shape, nd] (PiecewiseRationalQuadraticCDF, coupling.py:211-240)."""
    dev = require_device(x, uw, uh, ud, allow_grad=allow_grad)
    k, nd = cfg.num_bins, n_derivatives(cfg)
    shape = tuple(x.shape[1:])
    if tuple(uw.shape) != shape + (k,) or tuple(uh.shape) != shape + (k,) or tuple(ud.shape) != shape + (nd,):
        raise VcnfError("shared spline logits %s %s %s do not match positions %s with K=%d" % (
            tuple(uw.shape), tuple(uh.shape), tuple(ud.shape), shape, k))
    x = x.contiguous()
    uw, uh, ud = uw.contiguous(), uh.contiguous(), ud.contiguous()
    y, lad = torch.empty_like(x), torch.empty_like(x)
    period = int(x[0].numel())
    tables = torch.empty(period * 3 * (k + 1), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = lib().vcnf_rqs_shared_f32(_ptr(x), _ptr(uw), _ptr(uh), _ptr(ud), period, _ptr(tables), _ptr(y), _ptr(lad),
                                       x.numel(), ctypes.byref(cfg), int(bool(inverse)),
                                       _ptr(bad_discriminant_counter(dev)) if inverse else None, _stream())
    _check(st, "vcnf_rqs_shared_f32")
    return y, lad


def rqs_packed_bwd(x, params, gy, glad, cfg, inverse):
    """VJP of rqs_elementwise_image: x, gy [B, C, *inner]; params [B, C*P, *inner]; glad [B]
    (gradient of the per-sample log-det).  Returns (g_x, g_params) in the layouts of x / params."""
    dev = require_device(x, params, gy, glad, allow_grad=True)
    x, params = x.detach().contiguous(), params.detach().contiguous()
    gy, glad = gy.detach().contiguous(), glad.detach().contiguous()
    inner = int(x[0, 0].numel())
    gx, gp = torch.empty_like(x), torch.empty_like(params)
    with torch.cuda.device(dev):
        st = lib().vcnf_rqs_packed_bwd_f32(_ptr(x), _ptr(params), inner, int(x[0].numel()), _ptr(gy), _ptr(glad),
                                           _ptr(gx), _ptr(gp), x.numel(), ctypes.byref(cfg),
                                           int(bool(inverse)), _stream())
    _check(st, "vcnf_rqs_packed_bwd_f32")
    return gx, gp


SHARED_BWD_BINS = (4, 8, 10, 16)


def rqs_shared_bwd(x, uw, uh, ud, gy, glad, cfg, inverse):
    """VJP of rqs_elementwise_shared: x, gy [B, *shape]; glad [B]; returns (g_x, g_uw, g_uh, g_ud)."""
    dev = require_device(x, uw, uh, ud, gy, glad, allow_grad=True)
    k, nd = cfg.num_bins, n_derivatives(cfg)
    x, gy, glad = x.detach().contiguous(), gy.detach().contiguous(), glad.detach().contiguous()
    uw, uh, ud = uw.detach().contiguous(), uh.detach().contiguous(), ud.detach().contiguous()
    b, period = x.shape[0], int(x[0].numel())
    groups = int(lib().vcnf_rqs_shared_bwd_groups(b, period))
    gx = torch.empty_like(x)
    partial = torch.empty(groups, period, 2 * k + nd, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = lib().vcnf_rqs_shared_bwd_f32(_ptr(x), _ptr(uw), _ptr(uh), _ptr(ud), b, period, period,
                                           _ptr(gy), _ptr(glad), _ptr(gx), _ptr(partial), groups,
                                           ctypes.byref(cfg), int(bool(inverse)), _stream())
    _check(st, "vcnf_rqs_shared_bwd_f32")
    g = partial.sum(0)
    return gx, g[:, :k].reshape(uw.shape), g[:, k:2 * k].reshape(uh.shape), g[:, 2 * k:].reshape(ud.shape)


def rqs_elementwise_bwd(x, uw, uh, ud, gy, glad, cfg, inverse):
    """VJP of rqs_elementwise: returns (g_x, g_uw, g_uh, g_ud), shapes of the inputs."""
    dev = require_device(x, uw, uh, ud, gy, glad, allow_grad=True)
    k = cfg.num_bins
    nd = n_derivatives(cfg)
    shape = x.shape
    xf = x.detach().reshape(-1).contiguous()
    w2 = uw.detach().reshape(-1, k).contiguous()
    h2 = uh.detach().reshape(-1, k).contiguous()
    d2 = ud.detach().reshape(-1, nd).contiguous()
    gyf = gy.detach().reshape(-1).contiguous()
    glf = glad.detach().reshape(-1).contiguous()
    gx = torch.empty_like(xf)
    gw, gh, gd = torch.empty_like(w2), torch.empty_like(h2), torch.empty_like(d2)
    with torch.cuda.device(dev):
        st = lib().vcnf_rqs_elementwise_bwd_f32(_ptr(xf), _ptr(w2), _ptr(h2), _ptr(d2), k, k, nd,
                                               _ptr(gyf), _ptr(glf), _ptr(gx), _ptr(gw), _ptr(gh), _ptr(gd),
                                               xf.numel(), ctypes.byref(cfg), int(bool(inverse)), _stream())
    _check(st, "vcnf_rqs_elementwise_bwd_f32")
    return gx.view(shape), gw.view(shape + (k,)), gh.view(shape + (k,)), gd.view(shape + (nd,))


def rqs_coupling(x, params, tf_idx, id_idx, shared, cfg, inverse, logdet=None, sign=1.0):
    """x [B,D] -> (y [B,D], logdet [B]).  ``logdet`` given: accumulate sign*sum
    into it; else a fresh tensor holding sign*sum."""
    dev = require_device(x, params, logdet, *(shared or ()))
    b, d = x.shape
    x = x.contiguous()
    params = params.contiguous()
    y = torch.empty_like(x)
    mode = LD_ACCUM
    if logdet is None:
        logdet = torch.empty(b, dtype=torch.float32, device=dev)
        mode = LD_STORE
    sw, sh, sd = shared if shared is not None else (None, None, None)
    sink = EVENT_SINK
    with torch.cuda.device(dev):
        if sink is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        st = lib().vcnf_rqs_coupling_f32(_ptr(x), _ptr(params), _ptr(tf_idx), tf_idx.numel(),
                                         _ptr(id_idx), id_idx.numel(), _ptr(sw), _ptr(sh), _ptr(sd),
                                         _ptr(y), _ptr(logdet), b, ctypes.byref(cfg), int(bool(inverse)),
                                         mode, float(sign),
                                         _ptr(bad_discriminant_counter(dev)) if inverse else None, _stream())
        if sink is not None:
            ev1.record()
            sink.append((ev0, ev1, b))
    _check(st, "vcnf_rqs_coupling_f32")
    return y, logdet


def rqs_conditioner_input(x, id_idx, context, shared, cfg, apply_inverse_shared):
    dev = require_device(x, context, *(shared or ()))
    b, d = x.shape
    x = x.contiguous()
    c = 0
    if context is not None:
        context = context.contiguous()
        c = context.shape[1]
    out = torch.empty(b, id_idx.numel() + c, dtype=torch.float32, device=dev)
    sw, sh, sd = shared if shared is not None else (None, None, None)
    with torch.cuda.device(dev):
        st = lib().vcnf_rqs_conditioner_input_f32(_ptr(x), b, d, _ptr(id_idx), id_idx.numel(), _ptr(context), c,
                                                  _ptr(sw), _ptr(sh), _ptr(sd), ctypes.byref(cfg),
                                                  int(bool(apply_inverse_shared)), _ptr(out), _stream())
    _check(st, "vcnf_rqs_conditioner_input_f32")
    return out


_REDO_FLAGS = {}


def _redo_flags(dev, tiles):
    """Per-tile flag words of the split-half kernel's range check.  One grow-only buffer per (device, stream): the
    split-half launch writes it and the fp32 launch behind it on the same stream reads it, so consecutive layers can
    share it.  Under stream capture a fresh buffer is taken from the capturing allocator instead (a HIP graph keeps
    the address)."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(tiles, dtype=torch.int32, device=dev)
    key = (device_index(dev), torch.cuda.current_stream().cuda_stream)
    buf = _REDO_FLAGS.get(key)
    if buf is None or buf.numel() < tiles:
        buf = torch.empty(max(tiles, 8192), dtype=torch.int32, device=dev)
        _REDO_FLAGS[key] = buf
    return buf


def small_batch_rows(rows=None):
    """Batch size up to which the fp16 split-half fused RQS layer runs on 32-sample tiles (csrc/fused_layer_v6s.hip)
    instead of 128-sample ones; ``rows`` sets it (0: never, a huge value: always), None only queries.  Returns the
    previous value.  Process-wide."""
    return int(lib().vcnf_rqs_layer_fused_small_batch_rows(-1 if rows is None else int(rows)))


def rqs_layer_fused(x, context, tf_idx, id_idx, ctx_dim, hidden, num_blocks, precision, wpack, shared, cfg,
                    inverse, logdet=None, sign=1.0, wpack_f32=None, cfg_f32=None):
    """Whole coupling layer (conditioner included) in one kernel; see csrc/fused_layer.hip.
    precision 1 (fp16 split-half operands) with ``wpack_f32`` given is the range-safe form: the split-half launch
    leaves tiles that hold a non-finite input or a value beyond +-65504 unwritten and flags them, a launch of the
    exact fp32 kernel behind it evaluates exactly those tiles (normally none: it returns at once) - no host round
    trip, nothing clamped.  Without ``wpack_f32`` the split-half kernel clamps and counts (saturation_counter)."""
    dev = require_device(x, context, wpack, logdet, wpack_f32, *(shared or ()))
    b, d = x.shape
    x = x.contiguous()
    if context is not None:
        context = context.contiguous()
    y = torch.empty_like(x)
    mode = LD_ACCUM
    if logdet is None:
        logdet = torch.empty(b, dtype=torch.float32, device=dev)
        mode = LD_STORE
    sw, sh, sd = shared if shared is not None else (None, None, None)
    sink = EVENT_SINK
    L = lib()
    safe = precision == 1 and wpack_f32 is not None

    def launch(prec, pack, cf, sat, redo):
        return L.vcnf_rqs_layer_fused_f32(_ptr(x), _ptr(context), _ptr(y), _ptr(logdet), b,
                                          _ptr(tf_idx), tf_idx.numel(), _ptr(id_idx), id_idx.numel(),
                                          int(ctx_dim), int(hidden), int(num_blocks), int(prec),
                                          _ptr(pack), pack.numel(), _ptr(sw), _ptr(sh), _ptr(sd),
                                          ctypes.byref(cf), int(bool(inverse)), mode, float(sign),
                                          _ptr(bad_discriminant_counter(dev)) if inverse else None, sat, redo, _stream())
    with torch.cuda.device(dev):
        flags = _redo_flags(dev, (b + 31) // 32) if safe and b > 0 else None        # one flag per 32 samples
        if sink is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        st = launch(precision, wpack, cfg, _ptr(range_redo_counter(dev) if safe else saturation_counter(dev)) if precision == 1 else None,
                    _ptr(flags))
        if sink is not None:
            ev1.record()
            sink.append((ev0, ev1, b))
        if st == OK and flags is not None:
            st = launch(0, wpack_f32, cfg_f32 if cfg_f32 is not None else cfg, None, _ptr(flags))
    _check(st, "vcnf_rqs_layer_fused_f32")
    return y, logdet


def rqs_stack_fused(x, context, layers, layers_f32, pack_floats, d_t, d_id, ctx_dim, hidden, num_blocks, precision, cfg,
                    inverse, logdet=None, sign=1.0):
    """A run of RQS coupling layers of one shape in ONE launch (csrc/fused_layer_v6s.hip; exact fp32:
    csrc/fused_layer.hip): ``layers`` is a ctypes array of RqsStackLayer in application order.  precision 1 with
    ``layers_f32`` (the same layers with their fp32 packings) is the range-safe pair of launches of rqs_layer_fused:
    tiles in which any layer left the fp16 range are re-evaluated through all layers on exact fp32 instructions."""
    dev = require_device(x, context, logdet)
    b = x.shape[0]
    x = x.contiguous()
    if context is not None:
        context = context.contiguous()
    y = torch.empty_like(x)
    mode = LD_ACCUM
    if logdet is None:
        logdet = torch.empty(b, dtype=torch.float32, device=dev)
        mode = LD_STORE
    L = lib()
    safe = precision == 1 and layers_f32 is not None

    def launch(prec, lay, sat, redo):
        return L.vcnf_rqs_stack_fused_f32(_ptr(x), _ptr(context), _ptr(y), _ptr(logdet), b, lay, len(lay),
                                          int(d_t), int(d_id), int(ctx_dim), int(hidden), int(num_blocks), int(prec),
                                          int(pack_floats), ctypes.byref(cfg), int(bool(inverse)), mode, float(sign),
                                          _ptr(bad_discriminant_counter(dev)) if inverse else None, sat, redo, _stream())
    with torch.cuda.device(dev):
        flags = _redo_flags(dev, (b + 31) // 32) if safe and b > 0 else None
        sink = EVENT_SINK
        if sink is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        st = launch(precision, layers, _ptr(range_redo_counter(dev) if safe else saturation_counter(dev)) if precision == 1 else None,
                    _ptr(flags))
        if sink is not None:
            ev1.record()
            sink.append((ev0, ev1, b))
        if st == OK and flags is not None:
            st = launch(0, layers_f32, None, _ptr(flags))
    _check(st, "vcnf_rqs_stack_fused_f32")
    return y, logdet


def affine_coupling(z, param, t_off, d_t, scale_map, inverse, logdet=None, sign=1.0):
    """z [B, C, *inner] -> (out, logdet [B] or None when scale_map is NONE and no
    logdet was passed)."""
    dev = require_device(z, param, logdet, f64=True)
    if param.dtype != z.dtype or (logdet is not None and logdet.dtype != z.dtype):
        raise VcnfError("affine_coupling: mixed dtypes")
    z = z.contiguous()
    param = param.contiguous()
    b, c = z.shape[0], z.shape[1]
    inner = int(z[0, 0].numel()) if z.dim() > 2 else 1
    out = torch.empty_like(z)
    mode = LD_ACCUM
    if logdet is None:
        mode = LD_STORE
        if scale_map != SCALE_NONE:
            logdet = torch.empty(b, dtype=z.dtype, device=dev)
    with torch.cuda.device(dev):
        st = getattr(lib(), "vcnf_affine_coupling" + _sfx(z))(
            _ptr(z), _ptr(param), _ptr(out), _ptr(logdet), b, c, inner,
            int(t_off), int(d_t), int(scale_map), int(bool(inverse)), mode, float(sign), _stream())
    _check(st, "vcnf_affine_coupling" + _sfx(z))
    return out, logdet


def rqs_final_fused(x, h, out, tf_idx, d_t, hidden, wpack, cfg, inverse, partial=None, presplit=False):
    """Last conditioner layer + splines (csrc/fused_final.hip): writes out[:, tf_idx], returns the partial
    log-det rows [rows, B] (written into the first rows of ``partial`` when the caller brings the buffer)."""
    dev = require_device(x, h, out, wpack)
    x, h = x.contiguous(), h.contiguous()
    if not out.is_contiguous():
        raise VcnfError("rqs_final_fused writes into a contiguous output tensor")
    b, d = x.shape
    rows = int(lib().vcnf_rqs_final_fused_partial_rows(d_t, cfg.num_bins))
    if partial is None:
        partial = torch.empty(rows, b, dtype=torch.float32, device=dev)
    elif partial.shape[0] < rows or partial.shape[1] != b or not partial.is_contiguous():
        raise VcnfError("partial log-det buffer %s too small for %d rows of %d" % (tuple(partial.shape), rows, b))
    with torch.cuda.device(dev), _timed("rqs_final_fused"):
        fn = lib().vcnf_rqs_final_fused_presplit_f32 if presplit else lib().vcnf_rqs_final_fused_f32
        st = fn(_ptr(x), _ptr(h), _ptr(out), _ptr(partial), b, d, _ptr(tf_idx), int(d_t), int(hidden), _ptr(wpack),
                wpack.numel(), ctypes.byref(cfg), int(bool(inverse)),
                _ptr(bad_discriminant_counter(dev)) if inverse else None, _stream())
    _check(st, "vcnf_rqs_final_fused_f32")
    return partial


def identity_half_rows(d_id, shared):
    return int(lib().vcnf_rqs_identity_half_partial_rows(d_id)) if shared is not None else 0


def rqs_identity_half(x, out, id_idx, d_id, shared, cfg, inverse, partial=None, want_cond_in=True):
    """Identity features of a coupling layer in one launch (csrc/rqs_kernels.hip::rqs_identity_half_kernel):
    out[:, id_idx] = S(x[:, id_idx]) (S^-1 when ``inverse``; the plain copy without ``shared`` logits), the
    conditioner's input rows [B, d_id] (raw columns in the density direction, S^-1 of them when sampling) and the
    log-det partial rows, written into ``partial`` (rows identity_half_rows(d_id, shared))."""
    dev = require_device(x, out, id_idx, *(shared or ()))
    x = x.contiguous()
    if not out.is_contiguous():
        raise VcnfError("rqs_identity_half writes into a contiguous output tensor")
    b, d = x.shape
    cond_in = torch.empty(b, d_id, dtype=torch.float32, device=dev) if want_cond_in else None
    if shared is not None:
        sw, sh, sd = (t.detach().contiguous() for t in shared)
        if partial is None or partial.shape[0] < identity_half_rows(d_id, shared) or partial.shape[1] != b:
            raise VcnfError("rqs_identity_half needs a [rows, B] partial log-det buffer")
    else:
        sw = sh = sd = None
    with torch.cuda.device(dev), _timed("rqs_identity_half"):
        st = lib().vcnf_rqs_identity_half_f32(
            _ptr(x), _ptr(out), _ptr(cond_in) if cond_in is not None else None,
            _ptr(partial) if partial is not None else None, b, d, _ptr(id_idx), int(d_id),
            _ptr(sw) if sw is not None else None, _ptr(sh) if sh is not None else None,
            _ptr(sd) if sd is not None and sd.numel() else None,
            ctypes.byref(cfg) if cfg is not None else None, int(bool(inverse)), int(bool(inverse)),
            _ptr(bad_discriminant_counter(dev)) if inverse else None, _stream())
    _check(st, "vcnf_rqs_identity_half_f32")
    return cond_in


def resnet_trunk(x, wpack, hidden, num_blocks, split=False):
    """ResidualNet trunk (initial layer + residual blocks) in one kernel; csrc/resnet_trunk.hip.  x [B, d_in] -> h [B, hidden]."""
    dev = require_device(x, wpack)
    x = x.contiguous()
    b, d_in = x.shape
    h = torch.empty(b, hidden, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), _timed("resnet_trunk"):
        if split:       # h: per row 128 fp16 hi halves | 128 fp16 lo halves, for rqs_final_fused(..., presplit=True)
            st = lib().vcnf_resnet_trunk_split_f32(_ptr(x), _ptr(h), b, int(d_in), int(hidden), int(num_blocks), _ptr(wpack),
                                                   wpack.numel(), _ptr(saturation_counter(dev)), _stream())
        else:
            st = lib().vcnf_resnet_trunk_f32(_ptr(x), _ptr(h), b, int(d_in), int(hidden), int(num_blocks), _ptr(wpack),
                                             wpack.numel(), _stream())
    _check(st, "vcnf_resnet_trunk_f32")
    return h


def affine_layer_fused(z, wpack, cond_off, c_in, t_off, d_t, hidden, slope, scale_map, inverse, logdet=None,
                       sign=1.0, in_gather=None, out_gather=None):
    """Whole AffineCouplingBlock (MLP conditioner included) in one kernel; csrc/fused_affine.hip."""
    dev = require_device(z, wpack, logdet)
    z = z.contiguous()
    b, d = z.shape
    out = torch.empty_like(z)
    mode = LD_ACCUM
    if logdet is None:
        mode = LD_STORE
        logdet = (torch.empty if scale_map != SCALE_NONE else torch.zeros)(b, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = lib().vcnf_affine_layer_fused_f32(_ptr(z), _ptr(out), _ptr(logdet) if scale_map != SCALE_NONE else None,
                                               b, d, int(cond_off), int(c_in), int(t_off), int(d_t), int(hidden),
                                               float(slope), int(scale_map), _ptr(wpack), wpack.numel(),
                                               _ptr(in_gather), _ptr(out_gather),
                                               int(bool(inverse)), mode, float(sign), _stream())
    _check(st, "vcnf_affine_layer_fused_f32")
    return out, logdet


class AffineStackLayer(ctypes.Structure):
    """vcnf_affine_stack_layer of include/vcnf_hip.h."""
    _fields_ = [("cond_off", ctypes.c_int32), ("t_off", ctypes.c_int32), ("d_t", ctypes.c_int32),
                ("gather_before", ctypes.c_int32)]


def affine_stack_fused(z, wpack, layers, gather_after, gathers, c_in, hidden, slope, scale_map, inverse, logdet=None,
                       sign=1.0, wpack_h3=None):
    """A run of AffineCouplingBlocks + the permutations between them in one kernel; csrc/fused_affine.hip.
    ``layers``: list of (cond_off, t_off, d_t, gather_before) in execution order; ``gathers`` int32 [rows, D] or None.
    ``wpack_h3``: the split-half fragments of the conditioners' second and third dense layers - then those run on the
    fp16 split-half matrix path (range fallback to the fp32 body on the device, counted in range_redo_counter)."""
    dev = require_device(z, wpack, logdet, gathers, wpack_h3)
    z = z.contiguous()
    b, d = z.shape
    out = torch.empty_like(z)
    mode = LD_ACCUM
    if logdet is None:
        mode = LD_STORE
        logdet = (torch.empty if scale_map != SCALE_NONE else torch.zeros)(b, dtype=torch.float32, device=dev)
    arr = (AffineStackLayer * len(layers))(*[AffineStackLayer(*map(int, l)) for l in layers])
    with torch.cuda.device(dev), _timed("affine_stack_fused"):
        if wpack_h3 is not None:
            st = lib().vcnf_affine_stack_fused_f16x3_f32(
                _ptr(z), _ptr(out), _ptr(logdet) if scale_map != SCALE_NONE else None,
                b, d, len(layers), ctypes.cast(arr, ctypes.c_void_p), int(gather_after),
                int(c_in), int(hidden), float(slope), int(scale_map), _ptr(wpack), wpack.numel(),
                _ptr(wpack_h3), wpack_h3.numel(), _ptr(gathers), 0 if gathers is None else int(gathers.shape[0]),
                int(bool(inverse)), mode, float(sign), _ptr(range_redo_counter(dev)), _stream())
        else:
            st = lib().vcnf_affine_stack_fused_f32(_ptr(z), _ptr(out), _ptr(logdet) if scale_map != SCALE_NONE else None,
                                                   b, d, len(layers), ctypes.cast(arr, ctypes.c_void_p), int(gather_after),
                                                   int(c_in), int(hidden), float(slope), int(scale_map),
                                                   _ptr(wpack), wpack.numel(), _ptr(gathers),
                                                   0 if gathers is None else int(gathers.shape[0]),
                                                   int(bool(inverse)), mode, float(sign), _stream())
    _check(st, "vcnf_affine_stack_fused_f32")
    return out, logdet


def masked_affine(z, s, t, bmask, inverse, logdet=None, sign=1.0):
    dev = require_device(z, s, t, bmask, logdet, f64=True)
    if any(u is not None and u.dtype != z.dtype for u in (s, t, bmask, logdet)):
        raise VcnfError("masked_affine: mixed dtypes")
    z = z.contiguous()
    s = s.contiguous() if s is not None else None
    t = t.contiguous() if t is not None else None
    b, d = z.shape
    out = torch.empty_like(z)
    mode = LD_ACCUM
    if logdet is None:
        logdet = torch.empty(b, dtype=z.dtype, device=dev)
        mode = LD_STORE
    with torch.cuda.device(dev):
        st = getattr(lib(), "vcnf_masked_affine" + _sfx(z))(_ptr(z), _ptr(s), _ptr(t), _ptr(bmask), _ptr(out), _ptr(logdet),
                                                            b, d, int(bool(inverse)), mode, float(sign), _stream())
    _check(st, "vcnf_masked_affine" + _sfx(z))
    return out, logdet


def maf_affine(x, params, inverse):
    """Masked-affine-autoregressive elementwise map on a MADE output [B, D * 2] (csrc/affine_kernels.hip::
    maf_affine_kernel): returns (y, log_det[B])."""
    dev = require_device(x, params)
    x, params = x.contiguous(), params.contiguous()
    b, d = x.shape
    if params.shape[0] != b or params[0].numel() != 2 * d:
        raise VcnfError("maf_affine: params %s do not match inputs %s" % (tuple(params.shape), tuple(x.shape)))
    out = torch.empty_like(x)
    ld = torch.empty(b, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = lib().vcnf_maf_affine_f32(_ptr(x), _ptr(params), _ptr(out), _ptr(ld), b, d, int(bool(inverse)), LD_STORE, 1.0,
                                       _stream())
    _check(st, "vcnf_maf_affine_f32")
    return out, ld


def affine_const(z, s, t, inverse):
    dev = require_device(z, s, t, f64=True)
    if any(u is not None and u.dtype != z.dtype for u in (s, t)):
        raise VcnfError("affine_const: mixed dtypes")
    z = z.contiguous()
    b, c = z.shape[0], z.shape[1]
    inner = int(z[0, 0].numel()) if z.dim() > 2 else 1
    out = torch.empty_like(z)
    with torch.cuda.device(dev):
        st = getattr(lib(), "vcnf_affine_const" + _sfx(z))(_ptr(z), _ptr(s), _ptr(t), _ptr(out), b, c, inner,
                                                           int(bool(inverse)), _stream())
    _check(st, "vcnf_affine_const" + _sfx(z))
    return out


def conv1x1_fused(x, wpack, c_out, in_bias=None, out_bias=None, in_slope=None, out_slope=None):
    """act_out(W act_in(x + in_bias) + out_bias) for NCHW x in one pass (csrc/conv1x1.hip, fp16 split-half matrix path);
    a slope of None switches that LeakyReLU off."""
    dev = require_device(x, wpack, in_bias, out_bias)
    x = x.contiguous()
    b, c_in = x.shape[0], x.shape[1]
    inner = int(x[0, 0].numel())
    out = torch.empty((b, c_out) + tuple(x.shape[2:]), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), _timed("conv1x1_fused"):
        st = lib().vcnf_conv1x1_f16x3_f32(_ptr(x), _ptr(out), _ptr(wpack), wpack.numel(), _ptr(in_bias), _ptr(out_bias),
                                          b, int(c_in), int(c_out), inner, int(in_slope is not None),
                                          float(in_slope or 0.0), int(out_slope is not None), float(out_slope or 0.0),
                                          _ptr(saturation_counter(dev)), _stream())
    _check(st, "vcnf_conv1x1_f16x3_f32")
    return out


def masked_affine_stack(z, table, n_layers, inverse, logdet=None, sign=1.0):
    """A run of MaskedAffineFlow (+ MLP conditioners) and per-feature affine layers in one launch
    (csrc/masked_affine_stack.hip); ``table``: device int64 [n_layers, 12] as described in include/vcnf_hip.h."""
    dev = require_device(z, logdet, f64=True)
    z = z.contiguous()
    b, d = z.shape
    out = torch.empty_like(z)
    mode = LD_ACCUM
    if logdet is None:
        logdet = torch.empty(b, dtype=z.dtype, device=dev)
        mode = LD_STORE
    with torch.cuda.device(dev):
        st = getattr(lib(), "vcnf_masked_affine_stack" + _sfx(z))(_ptr(z), _ptr(out), _ptr(logdet), _ptr(table), b, int(d),
                                                                  int(n_layers), int(bool(inverse)), mode, float(sign), _stream())
    _check(st, "vcnf_masked_affine_stack" + _sfx(z))
    return out, logdet


def masked_affine_stack_bwd(z_out, g_out, g_ld, table, goff, n_layers, n_grad, inverse):
    """VJP of masked_affine_stack: (gradient of the run's input, flat parameter-gradient buffer [n_grad])."""
    dev = require_device(z_out, g_out, g_ld, f64=True, allow_grad=True)
    z_out, g_out = z_out.contiguous(), g_out.contiguous()
    if g_ld is not None:
        g_ld = g_ld.contiguous()
    b, d = z_out.shape
    g_in = torch.empty_like(z_out)
    grads = torch.zeros(n_grad, dtype=z_out.dtype, device=dev)
    with torch.cuda.device(dev):
        st = getattr(lib(), "vcnf_masked_affine_stack_bwd" + _sfx(z_out))(
            _ptr(z_out), _ptr(g_out), _ptr(g_ld), _ptr(g_in), _ptr(grads), _ptr(table), _ptr(goff), b, int(d), int(n_layers),
            int(bool(inverse)), _stream())
    _check(st, "vcnf_masked_affine_stack_bwd" + _sfx(z_out))
    return g_in, grads


def linear_f16x3(x, weight, bias=None, input_grad=False, relu_in=False, relu_out=False, mask=None, addend=None):
    """nn.Linear on the fp16 split-half matrix path at training batch sizes (csrc/linear_f16x3.hip): ``x @ weight.T +
    bias`` for weight [out, in], or with ``input_grad`` the layer's input gradient ``x @ weight`` (x = the upstream
    gradient [B, out]).  The weight is read in place (no packed copy); ``relu_in`` / ``relu_out`` apply ReLU to x as it is
    read / to the result before it is stored; ``mask`` / ``addend`` [B, n]: result = addend + result * (mask > 0) (a ReLU's
    backward and a skip connection's gradient on the way out).  Clamped values are counted in saturation_counter
    (nf.check_saturation())."""
    dev = require_device(x, weight, bias, allow_grad=True)
    x, weight = x.detach().contiguous(), weight.detach().contiguous()
    b, k = x.shape
    n_out, n_in = weight.shape
    if input_grad:
        n, ldn, ldk = n_in, 1, n_in
        assert k == n_out and bias is None
    else:
        n, ldn, ldk = n_out, n_in, 1
        assert k == n_in
    y = torch.empty(b, n, dtype=torch.float32, device=dev)
    if mask is not None:
        mask = mask.detach().contiguous()
    if addend is not None:
        addend = addend.detach().contiguous()
    with torch.cuda.device(dev), _timed("linear_f16x3"):
        st = lib().vcnf_linear_f16x3_f32(_ptr(x), _ptr(weight), _ptr(bias.detach().contiguous() if bias is not None else None),
                                         _ptr(y), b, int(k), int(n), int(ldn), int(ldk), int(bool(relu_in)), int(bool(relu_out)),
                                         _ptr(mask), _ptr(addend), _ptr(saturation_counter(dev)), _stream())
    _check(st, "vcnf_linear_f16x3_f32")
    return y


_WGRAD_WS = {}        # workspace of the partial results per (device, stream): calls on one stream are ordered, calls on
                      # different streams (two models' backward passes, side streams) must not share it


def linear_wgrad(x, dy, want_bias=True, f16x3=False, relu_x=False):
    """(dW [out, in], db [out] or None) of y = x W^T + b from x [B, in] and dy [B, out] (csrc/linear_wgrad.hip: the batch
    reduction split over the chip, deterministic; exact fp32 matrix instructions, or with ``f16x3`` the fp16 split-half
    matrix path - clamped values are counted in saturation_counter)."""
    dev = require_device(x, dy, allow_grad=True)
    x, dy = x.detach().contiguous(), dy.detach().contiguous()
    if relu_x and not f16x3:
        x = torch.relu(x)
    b, n_in = x.shape
    n_out = dy.shape[1]
    slices = int(lib().vcnf_linear_wgrad_slices(b, n_in, n_out))
    if slices < 1:
        raise VcnfError("linear_wgrad: unsupported layer shape %d -> %d" % (n_in, n_out))
    need = slices * (n_out * n_in + n_out)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(dev).cuda_stream)
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.float32, device=dev)
        _WGRAD_WS[key] = ws
    dw = torch.empty(n_out, n_in, dtype=torch.float32, device=dev)
    db = torch.empty(n_out, dtype=torch.float32, device=dev) if want_bias else None
    with torch.cuda.device(dev), _timed("linear_wgrad"):
        if f16x3:
            st = lib().vcnf_linear_wgrad_f16x3_f32(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws), ws.numel(), b, int(n_in),
                                                   int(n_out), 0, int(bool(relu_x)), _ptr(saturation_counter(dev)), _stream())
        else:
            st = lib().vcnf_linear_wgrad_f32(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws), ws.numel(), b, int(n_in),
                                             int(n_out), 0, _stream())
    _check(st, "vcnf_linear_wgrad_f32")
    return dw, db


def conv3x3_1x1_fused(x, w1pack, w2pack, b1, b2, slope1, slope2):
    """leaky(W2 leaky(conv3x3(x) + b1) + b2) for NCHW x with 256 hidden / output channels in one launch
    (csrc/conv3x3_1x1.hip); the hidden activation between the layers never reaches memory."""
    dev = require_device(x, w1pack, w2pack, b1, b2)
    x = x.contiguous()
    b, c_in, h, w = x.shape
    out = torch.empty((b, 256, h, w), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), _timed("conv3x3_1x1_fused"):
        st = lib().vcnf_conv3x3_1x1_f16x3_f32(_ptr(x), _ptr(out), _ptr(w1pack), w1pack.numel(), _ptr(w2pack), w2pack.numel(),
                                              _ptr(b1), _ptr(b2), b, int(c_in), int(h), int(w), float(slope1), float(slope2),
                                              _ptr(saturation_counter(dev)), _stream())
    _check(st, "vcnf_conv3x3_1x1_f16x3_f32")
    return out


def convnet3_fused(x, w1pack, w2pack, w3pack, b1, b2, b3, c_out, slope1, slope2):
    """Conv3x3 -> LeakyReLU -> Conv1x1 -> LeakyReLU -> Conv3x3 of the Glow conditioner (256 hidden channels) in two launches
    (csrc/conv3x3_1x1.hip): the fused kernel up to the last layer's nine tap results, then their shift-and-add."""
    dev = require_device(x, w1pack, w2pack, w3pack, b1, b2, b3)
    x = x.contiguous()
    b, c_in, h, w = x.shape
    z = torch.empty((b, 9 * c_out, h, w), dtype=torch.float32, device=dev)
    out = torch.empty((b, c_out, h, w), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        with _timed("convnet3_taps"):
            st = lib().vcnf_convnet3_taps_f16x3_f32(_ptr(x), _ptr(z), _ptr(w1pack), w1pack.numel(), _ptr(w2pack), w2pack.numel(),
                                                    _ptr(w3pack), w3pack.numel(), _ptr(b1), _ptr(b2), b, int(c_in), int(c_out),
                                                    int(h), int(w), float(slope1), float(slope2),
                                                    _ptr(saturation_counter(dev)), _stream())
        _check(st, "vcnf_convnet3_taps_f16x3_f32")
        with _timed("col2im3x3"):
            st = lib().vcnf_col2im3x3_f32(_ptr(z), _ptr(b3), _ptr(out), b, int(c_out), int(h), int(w), _stream())
    _check(st, "vcnf_col2im3x3_f32")
    return out


def resblock_op(op, a, b, c=None, two_outputs=False):
    """Fused elementwise map of csrc/resblock_ops.hip on contiguous fp32 tensors of one shape (ops 0-3, see the header)."""
    dev = require_device(a, b, c, allow_grad=True)
    a, b = a.detach().contiguous(), b.detach().contiguous()
    c = c.detach().contiguous() if c is not None else None
    out0 = torch.empty_like(a)
    out1 = torch.empty_like(a) if two_outputs else None
    with torch.cuda.device(dev):
        st = lib().vcnf_resblock_elementwise_f32(int(op), _ptr(a), _ptr(b), _ptr(c), _ptr(out0), _ptr(out1), a.numel(),
                                                 _stream())
    _check(st, "vcnf_resblock_elementwise_f32")
    return (out0, out1) if two_outputs else out0


def channel_mix(z, matrix, shift):
    """y[b, o, ...] = sum_c matrix[o, c] z[b, c, ...] + shift[o] (csrc/channel_mix.hip): invertible 1x1 convolution
    and ActNorm of a GlowBlock in one pass over the activations."""
    dev = require_device(z, matrix, shift)
    z = z.contiguous()
    b, c = z.shape[0], z.shape[1]
    if tuple(matrix.shape) != (c, c) or shift.numel() != c:
        raise VcnfError("channel_mix: matrix %s / shift %s do not match %d channels" % (
            tuple(matrix.shape), tuple(shift.shape), c))
    inner = int(z[0, 0].numel()) if z.dim() > 2 else 1
    out = torch.empty_like(z)
    with torch.cuda.device(dev), _timed("channel_mix"):
        st = lib().vcnf_channel_mix_f32(_ptr(z), _ptr(out), _ptr(matrix.contiguous()), _ptr(shift.contiguous()), b, c,
                                        inner, _stream())
    _check(st, "vcnf_channel_mix_f32")
    return out


def permute(z, idx32):
    dev = require_device(z, f64=True)
    z = z.contiguous()
    b, c = z.shape[0], z.shape[1]
    inner = int(z[0, 0].numel()) if z.dim() > 2 else 1
    out = torch.empty_like(z)
    with torch.cuda.device(dev):
        st = getattr(lib(), "vcnf_permute" + _sfx(z))(_ptr(z), _ptr(idx32), _ptr(out), b, c, inner, _stream())
    _check(st, "vcnf_permute" + _sfx(z))
    return out


def split_columns(z, idx32, first):
    """(z[:, idx[:first]], z[:, idx[first:]]) of z [B, C] as two contiguous tensors, one pass."""
    dev = require_device(z, f64=True)
    z = z.contiguous()
    b, c = z.shape
    if idx32.numel() != c or not 0 <= first <= c:
        raise VcnfError("split_columns: index of %d entries / first part of %d columns for %d columns" % (idx32.numel(), first, c))
    pa = torch.empty(b, first, dtype=z.dtype, device=dev)
    pb = torch.empty(b, c - first, dtype=z.dtype, device=dev)
    with torch.cuda.device(dev):
        st = getattr(lib(), "vcnf_split_columns" + _sfx(z))(_ptr(z), _ptr(idx32), _ptr(pa), _ptr(pb), b, c, first, _stream())
    _check(st, "vcnf_split_columns" + _sfx(z))
    return pa, pb


def merge_columns(pa, pb, idx32):
    """cat([pa, pb], 1)[:, idx] for pa [B, Ca], pb [B, Cb] as one pass (no concatenated intermediate)."""
    dev = require_device(pa, pb, f64=True)
    if pa.dtype != pb.dtype or pa.shape[0] != pb.shape[0]:
        raise VcnfError("merge_columns: parts of different dtype / batch")
    pa, pb = pa.contiguous(), pb.contiguous()
    b, first = pa.shape
    c = first + pb.shape[1]
    if idx32.numel() != c:
        raise VcnfError("merge_columns: index of %d entries for %d columns" % (idx32.numel(), c))
    out = torch.empty(b, c, dtype=pa.dtype, device=dev)
    with torch.cuda.device(dev):
        st = getattr(lib(), "vcnf_merge_columns" + _sfx(pa))(_ptr(pa), _ptr(pb), _ptr(idx32), _ptr(out), b, c, first, _stream())
    _check(st, "vcnf_merge_columns" + _sfx(pa))
    return out


def diag_gaussian_log_prob(z, loc, log_scale, temperature=None, logp=None, sign=1.0):
    dev = require_device(z, loc, log_scale, logp, f64=True)
    if any(u is not None and u.dtype != z.dtype for u in (loc, log_scale, logp)):
        raise VcnfError("diag_gaussian_log_prob: mixed dtypes")
    b = z.shape[0]
    z2 = z.reshape(b, -1).contiguous()
    mode = LD_ACCUM
    if logp is None:
        logp = torch.empty(b, dtype=z.dtype, device=dev)
        mode = LD_STORE
    lt = 0.0 if temperature is None else math.log(temperature)
    with torch.cuda.device(dev):
        st = getattr(lib(), "vcnf_diag_gaussian_log_prob" + _sfx(z))(
            _ptr(z2), _ptr(loc.contiguous()), _ptr(log_scale.contiguous()), lt, _ptr(logp), b, z2.shape[1], mode, float(sign), _stream())
    _check(st, "vcnf_diag_gaussian_log_prob" + _sfx(z))
    return logp


def diag_gaussian_sample(eps, loc, log_scale, temperature=None):
    dev = require_device(eps, loc, log_scale, f64=True)
    if loc.dtype != eps.dtype or log_scale.dtype != eps.dtype:
        raise VcnfError("diag_gaussian_sample: mixed dtypes")
    b = eps.shape[0]
    e2 = eps.reshape(b, -1).contiguous()
    z = torch.empty_like(e2)
    logp = torch.empty(b, dtype=eps.dtype, device=dev)
    lt = 0.0 if temperature is None else math.log(temperature)
    with torch.cuda.device(dev):
        st = getattr(lib(), "vcnf_diag_gaussian_sample" + _sfx(eps))(
            _ptr(e2), _ptr(loc.contiguous()), _ptr(log_scale.contiguous()), lt, _ptr(z), _ptr(logp), b, e2.shape[1], _stream())
    _check(st, "vcnf_diag_gaussian_sample" + _sfx(eps))
    return z.view(eps.shape), logp


PROBE_F32, PROBE_F16X3, PROBE_F16X3_LL = 0, 1, 2


def linear_probe(x, weight, bias, mode, relu_input=False):
    """Diagnostic (vcnf_linear_probe_f32): one dense layer evaluated with the arithmetic of one matrix path of the
    fused RQS layer kernels.  x [B, K], weight [N, K], bias [N] or None -> y [B, N]."""
    dev = require_device(x, weight, bias)
    x, weight = x.contiguous(), weight.contiguous()
    b, k = x.shape
    n = weight.shape[0]
    if weight.shape[1] != k or (bias is not None and bias.shape != (n,)):
        raise VcnfError("linear_probe: shapes %s %s" % (tuple(x.shape), tuple(weight.shape)))
    y = torch.empty(b, n, device=dev, dtype=torch.float32)
    _check(lib().vcnf_linear_probe_f32(_ptr(x), _ptr(weight), _ptr(bias.contiguous() if bias is not None else None),
                                       _ptr(y), b, k, n, int(mode), int(bool(relu_input)),
                                       _ptr(saturation_counter(dev)), _stream()), "vcnf_linear_probe_f32")
    return y
