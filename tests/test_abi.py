"""The C-ABI library loads and exports every symbol include/vcnf_hip.h declares;
argument validation (which runs on the host before any launch) returns the
documented status codes.  No GPU needed: nothing is launched."""
import ctypes
import os
import re

import pytest

import vcnf_amd
from vcnf_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vcnf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vcnf_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    names = declared_symbols()
    assert len(names) >= 11
    handle = ctypes.CDLL(_lib.lib_path())
    for n in names:
        assert hasattr(handle, n), "libvcnf_hip.so does not export %s" % n
    assert sorted(_lib.PROTOTYPES) == names, "ctypes prototypes out of sync with the header"
    assert vcnf_amd.lib().vcnf_abi_version() == 1


def test_struct_layout_matches_header():
    assert ctypes.sizeof(_lib.RqsCfg) == 40     # 2 x int32 + 8 x float


def test_status_strings():
    for code in range(0, 7):
        assert vcnf_amd.lib().vcnf_status_string(code)


def test_validation_status_codes():
    L = vcnf_amd.lib()
    cfg = _lib.make_cfg(8, "linear", tail_bound=3.0)
    fake = ctypes.c_void_p(0x1000)    # never dereferenced: validation fails first / batch == 0
    # NULL required pointer
    assert L.vcnf_rqs_coupling_f32(None, fake, fake, 1, fake, 1, None, None, None, fake, fake, 4,
                                   ctypes.byref(cfg), 0, 0, 1.0, None, None) == 1
    # d_t < 1
    assert L.vcnf_rqs_coupling_f32(fake, fake, fake, 0, fake, 1, None, None, None, fake, fake, 4,
                                   ctypes.byref(cfg), 0, 0, 1.0, None, None) == 2
    # misaligned buffer
    assert L.vcnf_rqs_coupling_f32(ctypes.c_void_p(0x1002), fake, fake, 1, fake, 1, None, None, None, fake, fake,
                                   4, ctypes.byref(cfg), 0, 0, 1.0, None, None) == 3
    # only some of the shared logits given
    assert L.vcnf_rqs_coupling_f32(fake, fake, fake, 1, fake, 1, fake, None, None, fake, fake, 4,
                                   ctypes.byref(cfg), 0, 0, 1.0, None, None) == 1
    # min_bin_width * K > 1 (splines.py:104-107)
    bad = _lib.make_cfg(8, "linear", tail_bound=3.0, min_bin_width=0.2)
    assert L.vcnf_rqs_elementwise_f32(fake, fake, fake, fake, 8, 8, 7, fake, fake, 4,
                                      ctypes.byref(bad), 0, None, None) == 4
    # empty batch is a no-op
    assert L.vcnf_rqs_coupling_f32(fake, fake, fake, 1, fake, 1, None, None, None, fake, fake, 0,
                                   ctypes.byref(cfg), 0, 0, 1.0, None, None) == 0
    assert L.vcnf_affine_coupling_f32(fake, fake, fake, fake, 0, 4, 1, 2, 2, 0, 0, 0, 1.0, None) == 0
    # unknown scale map / transformed span outside the channels
    assert L.vcnf_affine_coupling_f32(fake, fake, fake, fake, 4, 4, 1, 2, 2, 9, 0, 0, 1.0, None) == 5
    assert L.vcnf_affine_coupling_f32(fake, fake, fake, fake, 4, 4, 1, 3, 2, 0, 0, 0, 1.0, None) == 2
    assert L.vcnf_masked_affine_f32(fake, None, None, None, fake, fake, 4, 3, 0, 0, 1.0, None) == 1
    assert L.vcnf_permute_f32(fake, None, fake, 4, 3, 1, None) == 1
    assert L.vcnf_diag_gaussian_log_prob_f32(fake, fake, fake, 0.0, fake, 4, 0, 0, 1.0, None) == 2


def test_missing_library_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib._build, "LIB", str(tmp_path / "nope.so"))
    with pytest.raises(vcnf_amd.VcnfError):
        _lib.lib()


def test_validation_status_codes_of_the_later_entry_points():
    """Entry points added after the first slice: shape / enum / NULL validation happens on the host
    before anything is launched."""
    L = vcnf_amd.lib()
    cfg = _lib.make_cfg(8, "linear", tail_bound=3.0)
    circ = _lib.make_cfg(8, "circular", tail_bound=3.0)
    assert circ.tails == _lib.TAILS_CIRCULAR and _lib.n_derivatives(circ) == 8
    fake = ctypes.c_void_p(0x1000)
    # strided spline: inner / k_stride must be >= 1
    assert L.vcnf_rqs_elementwise_strided_f32(fake, fake, fake, fake, 8, 8, 7, 0, 1, 0, fake, fake, 4,
                                              ctypes.byref(cfg), 0, None, None) == 2
    assert L.vcnf_rqs_elementwise_strided_f32(fake, fake, fake, fake, 8, 8, 7, 1, 1, 0, fake, fake, 0,
                                              ctypes.byref(cfg), 0, None, None) == 0      # empty batch
    # packed VJP: NULL gradient buffer
    assert L.vcnf_rqs_packed_bwd_f32(fake, fake, 1, 32, fake, fake, None, fake, 4, ctypes.byref(cfg), 0, None) == 1
    # shared-logit VJP: groups must be what vcnf_rqs_shared_bwd_groups says, bins in {4, 8, 10, 16}
    g = L.vcnf_rqs_shared_bwd_groups(1000, 32)
    assert g == 1000 and L.vcnf_rqs_shared_bwd_groups(1 << 20, 32) == 8192
    assert L.vcnf_rqs_shared_bwd_f32(fake, fake, fake, fake, 1000, 32, 32, fake, fake, fake, fake, g + 1,
                                     ctypes.byref(cfg), 0, None) == 2
    # fused layer shape families
    assert L.vcnf_rqs_layer_fused_supported(32, 32, 16, 128, 2, 8, _lib.TAILS_LINEAR) == 1
    assert L.vcnf_rqs_layer_fused_supported(16, 16, 0, 128, 2, 8, _lib.TAILS_LINEAR) == 1
    assert L.vcnf_rqs_layer_fused_supported(32, 32, 16, 128, 2, 8, _lib.TAILS_CIRCULAR) == 0
    assert L.vcnf_rqs_layer_fused_supported(24, 24, 0, 128, 2, 8, _lib.TAILS_LINEAR) == 0
    assert L.vcnf_rqs_layer_fused_pack_floats(24, 24, 0, 2) == 0
    assert L.vcnf_rqs_layer_fused_supported(32, 32, 16, 128, 3, 8, _lib.TAILS_LINEAR) == 1
    assert L.vcnf_rqs_layer_fused_supported(32, 32, 16, 128, 4, 8, _lib.TAILS_LINEAR) == 0
    assert L.vcnf_affine_layer_fused_supported(16, 64, 32, 32) == 1
    assert L.vcnf_affine_layer_fused_supported(16, 48, 32, 32) == 0          # hidden not in {32, 64, 128}
    assert L.vcnf_affine_layer_fused_supported(65, 64, 32, 130) == 0         # conditioner input too wide
    n = L.vcnf_affine_layer_fused_pack_floats(16, 64, 32)
    assert n == 4 * 1 * 256 + 64 + 4 * 4 * 256 + 64 + 2 * 4 * 256 + 32
    # wrong packed size / transformed range outside the row / unknown scale map
    assert L.vcnf_affine_layer_fused_f32(fake, fake, fake, 4, 32, 0, 16, 16, 16, 64, 0.0, 0, fake, n - 1,
                                         None, None, 0, 0, 1.0, None) == 2
    assert L.vcnf_affine_layer_fused_f32(fake, fake, fake, 4, 32, 0, 16, 20, 16, 64, 0.0, 0, fake, n,
                                         None, None, 0, 0, 1.0, None) == 2
    assert L.vcnf_affine_layer_fused_f32(fake, fake, fake, 4, 32, 0, 16, 16, 16, 64, 0.0, 9, fake, n,
                                         None, None, 0, 0, 1.0, None) in (2, 5)
