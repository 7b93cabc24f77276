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

def test_validation_status_codes_of_the_round2_entry_points():
    """Entry points added in round 2 (identity half, trunk, channel mix, 1x1 convolution, weight gradient, MAF map, affine
    stack): shape / size / NULL validation on the host, nothing launched."""
    L = vcnf_amd.lib()
    cfg = _lib.make_cfg(16, "linear", tail_bound=3.0)
    fake = ctypes.c_void_p(0x1000)
    # identity half: bin counts / tails it is built for; d_id <= features; partial rows per 64 features
    assert L.vcnf_rqs_identity_half_supported(16, _lib.TAILS_LINEAR) == 1 and L.vcnf_rqs_identity_half_supported(16, _lib.TAILS_CIRCULAR) == 0
    assert L.vcnf_rqs_identity_half_supported(12, _lib.TAILS_LINEAR) == 0
    assert L.vcnf_rqs_identity_half_partial_rows(512) == 8 and L.vcnf_rqs_identity_half_partial_rows(65) == 2
    assert L.vcnf_rqs_identity_half_f32(fake, fake, fake, fake, 4, 8, fake, 9, fake, fake, fake, ctypes.byref(cfg), 0, 0, None, None) == 2
    assert L.vcnf_rqs_identity_half_f32(fake, fake, fake, None, 4, 8, fake, 4, fake, fake, fake, ctypes.byref(cfg), 0, 0, None, None) == 1
    assert L.vcnf_rqs_identity_half_f32(fake, fake, fake, fake, 0, 8, fake, 4, fake, fake, fake, ctypes.byref(cfg), 0, 0, None, None) == 0
    # last-layer kernel: 8, 10 or 16 bins
    assert [L.vcnf_rqs_final_fused_supported(512, 128, k, _lib.TAILS_LINEAR) for k in (8, 10, 12, 16)] == [1, 1, 0, 1]
    # trunk: hidden 128, input width multiple of 16, 1-3 blocks; packed size
    assert L.vcnf_resnet_trunk_supported(512, 128, 2) == 1 and L.vcnf_resnet_trunk_supported(500, 128, 2) == 0
    assert L.vcnf_resnet_trunk_supported(512, 64, 2) == 0 and L.vcnf_resnet_trunk_supported(512, 128, 4) == 0
    n = L.vcnf_resnet_trunk_pack_floats(512, 128, 2)
    assert n == 8 * 32 * 256 + 128 + 2 * 2 * (64 * 256 + 128)
    assert L.vcnf_resnet_trunk_f32(fake, fake, 4, 512, 128, 2, fake, n - 1, None) == 2
    assert L.vcnf_resnet_trunk_split_f32(fake, fake, 4, 512, 128, 2, fake, n - 1, None, None) == 2
    assert L.vcnf_resnet_trunk_split_f32(fake, fake, 0, 512, 128, 2, fake, n, None, None) == 0
    assert L.vcnf_rqs_final_fused_presplit_f32(fake, fake, fake, fake, 4, 1024, fake, 512, 128, fake, 7, ctypes.byref(cfg), 0, None,
                                               None) == 2          # packed size mismatch
    # channel mix: multiples of 4 up to 64
    assert [L.vcnf_channel_mix_supported(c) for c in (4, 6, 48, 64, 68)] == [1, 0, 1, 1, 0]
    assert L.vcnf_channel_mix_f32(fake, fake, fake, fake, 2, 6, 16, None) == 5
    assert L.vcnf_channel_mix_f32(fake, fake, None, fake, 2, 8, 16, None) == 1
    assert L.vcnf_channel_mix_f32(fake, fake, fake, fake, 0, 8, 16, None) == 0
    # 1x1 convolution: c_in multiple of 16 up to 256; packed size must match
    assert L.vcnf_conv1x1_supported(256, 256) == 1 and L.vcnf_conv1x1_supported(24, 256) == 0 and L.vcnf_conv1x1_supported(256, 257) == 0
    m = L.vcnf_conv1x1_pack_floats(256, 256)
    assert m == 8 * 16 * 2 * 64 * 4
    assert L.vcnf_conv1x1_f16x3_f32(fake, fake, fake, m - 4, None, None, 2, 256, 256, 16, 1, 0.0, 1, 0.0, None, None) == 2
    assert L.vcnf_conv1x1_f16x3_f32(fake, fake, fake, m, None, None, 0, 256, 256, 16, 1, 0.0, 1, 0.0, None, None) == 0
    # first two conditioner layers in one launch: c_in <= 24, 256 hidden / output channels, packed sizes
    assert L.vcnf_conv3x3_1x1_supported(6, 256, 256) == 1 and L.vcnf_conv3x3_1x1_supported(25, 256, 256) == 0
    assert L.vcnf_conv3x3_1x1_supported(6, 128, 256) == 0
    assert L.vcnf_conv3x3_1x1_pack_floats(6) == 8 * 4 * 2 * 64 * 4 and L.vcnf_conv3x3_1x1_pack_floats(24) == 8 * 14 * 2 * 64 * 4
    assert L.vcnf_conv3x3_1x1_f16x3_f32(fake, fake, fake, 8 * 4 * 2 * 64 * 4, fake, m - 4, None, None, 2, 6, 16, 16, 0.0, 0.0,
                                        None, None) == 2
    assert L.vcnf_conv3x3_1x1_f16x3_f32(fake, fake, fake, 8 * 4 * 2 * 64 * 4, fake, m, None, None, 0, 6, 16, 16, 0.0, 0.0,
                                        None, None) == 0
    # whole conditioner: tap matrix of the last layer, col2im
    assert L.vcnf_convnet3_supported(6, 256, 12) == 1 and L.vcnf_convnet3_supported(6, 256, 57) == 0
    assert L.vcnf_convnet3_w3_pack_floats(12) == 4 * 16 * 2 * 64 * 4 and L.vcnf_convnet3_w3_pack_floats(48) == 14 * 16 * 2 * 64 * 4
    assert L.vcnf_convnet3_taps_f16x3_f32(fake, fake, fake, 8 * 4 * 2 * 64 * 4, fake, m, fake, 8, None, None, 2, 6, 12, 16, 16,
                                          0.0, 0.0, None, None) == 2
    assert L.vcnf_col2im3x3_f32(fake, None, fake, 0, 12, 16, 16, None) == 0 and L.vcnf_col2im3x3_f32(None, None, fake, 2, 12, 16, 16, None) == 1
    # weight gradient: slices and workspace size
    assert L.vcnf_linear_wgrad_supported(128, 736) == 1 and L.vcnf_linear_wgrad_supported(100, 128) == 0
    s = L.vcnf_linear_wgrad_slices(131072, 128, 128)
    assert s == 512 and L.vcnf_linear_wgrad_slices(1000, 128, 128) == 4 and L.vcnf_linear_wgrad_slices(131072, 128, 736) == 85
    assert L.vcnf_linear_wgrad_f32(fake, fake, fake, fake, fake, s * (128 * 128 + 128) - 1, 131072, 128, 128, 0, None) == 2
    assert L.vcnf_linear_wgrad_f32(fake, fake, None, fake, fake, s * (128 * 128 + 128), 131072, 128, 128, 0, None) == 1
    # residual-block maps: op 0-3, op 1 needs the second output
    assert L.vcnf_resblock_elementwise_f32(4, fake, fake, fake, fake, None, 8, None) == 5
    assert L.vcnf_resblock_elementwise_f32(1, fake, fake, fake, fake, None, 8, None) == 1
    assert L.vcnf_resblock_elementwise_f32(2, fake, fake, None, fake, None, 0, None) == 0
    # MAF map: parameters must be 8-byte aligned
    assert L.vcnf_maf_affine_f32(fake, ctypes.c_void_p(0x1004), fake, fake, 4, 7, 0, 0, 1.0, None) == 3
    assert L.vcnf_maf_affine_f32(fake, fake, fake, None, 4, 7, 0, 0, 1.0, None) == 1
    # affine stack: at most 16 layers
    layers = (_lib.AffineStackLayer * 17)()
    assert L.vcnf_affine_stack_fused_f32(fake, fake, fake, 4, 32, 17, ctypes.cast(layers, ctypes.c_void_p), -1, 16, 64, 0.0, 0,
                                         fake, 100, None, 0, 0, 0, 1.0, None) in (2, 5)
