"""The C-ABI shared library loads and exports every symbol include/v2a.h declares (no compute without a GPU); the product path
refuses to run without the HIP extension / on CPU tensors."""
import ctypes
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "v2a.h")).read()
    return sorted(set(re.findall(r"\b(v2a_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from v2a_hip import _lib
    syms = header_symbols()
    assert len(syms) >= 40
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(dll, s), f"{s} declared in include/v2a.h but not exported"
    assert set(syms) == set(_lib.SIGNATURES), set(syms) ^ set(_lib.SIGNATURES)


def test_host_only_entry_points():
    from v2a_hip._lib import lib
    assert lib.v2a_opt_chunk_elems() > 0 and lib.v2a_opt_state_bytes() > 64
    assert lib.v2a_conv2d_workspace_bytes(256, 1024, 5120) > 0          # small-M / deep-K -> split-K slabs
    assert lib.v2a_conv2d_workspace_bytes(1 << 20, 128, 1152) == 0
    assert lib.v2a_groupnorm_workspace_bytes(2, 16, 256, 8) == 0 and lib.v2a_groupnorm_workspace_bytes(2, 7 * 128 * 128, 128, 32) > 0


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from v2a_hip import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.V2AError):
        _lib._load()


def test_argument_validation_returns_error_codes_without_touching_the_device():
    """Entry points reject null / inconsistent arguments with V2A_ERR_ARG (-1) before any launch (so this runs without a GPU); the
    workspace queries of the newer kernels are host arithmetic."""
    from v2a_hip._lib import lib
    ERR_ARG = -1
    assert lib.v2a_conv2d_wgrad(None, None, None, None, None, 1, 8, 8, 64, 0, 8, 8, 64, 3, 3, 1, 1, 1, 1, 1, 0, 0, None, 0, None) == ERR_ARG
    assert lib.v2a_conv2d_wgrad_h(None, None, None, None, None, 1, 8, 8, 64, 0, 8, 8, 128, 3, 3, 1, 1, 1, 1, 1, 0, 0, None, 0, None) == ERR_ARG
    assert lib.v2a_groupnorm_fwd(None, None, 0, None, None, None, None, 0, None, None, None, 1, 16, 64, 8, 1e-5, 0, None, 0, None) == ERR_ARG
    assert lib.v2a_mha_fwd(None, None, None, None, None, 1, 4, 4, 2, 16, 32, 32, 32, 0.0, 0, 0, None) == ERR_ARG
    assert lib.v2a_mha_bwd(None, None, None, None, None, None, None, None, 1, 4, 4, 2, 16, 32, 32, 32, 0.0, 0, 0, None) == ERR_ARG
    assert lib.v2a_dropout(None, None, 16, 0.1, 1, 2, None) == ERR_ARG
    assert lib.v2a_video_qsample(None, None, None, None, None, None, 2, 64, 1, None) == ERR_ARG
    assert lib.v2a_video_loss_fwd(None, None, None, None, None, None, None, None, 2, 3, 64, 3, 2, 0, 1, None, 0, None) == ERR_ARG
    assert lib.v2a_video_loss_bwd(None, None, None, None, None, None, None, None, None, 2, 3, 64, 3, 2, 0, 1, None) == ERR_ARG
    assert lib.v2a_colsum_batched(None, None, 2, 64, 32, 0, None, 0, None) == ERR_ARG
    assert lib.v2a_layernorm_bwd(None, None, None, None, None, 4, 64, 1e-5, None) == ERR_ARG
    assert lib.v2a_bcast_rows(None, None, 2, 4, 8, 1.0, None) == ERR_ARG
    assert lib.v2a_debug_force_wgrad_plan(96, 96, 1) == ERR_ARG and lib.v2a_debug_force_wgrad_plan(0, 0, 0) == 0
    # round-4 entry points: channel-window stem conv, padded NHWC4 converter, optimiser step that also writes the operand packs
    assert lib.v2a_conv2d_fwd_window_f32(None, None, None, None, None, 2, 134, 67, 8, 32, 64, 7, 1, 2, 1, 64, 64, None, 0, None) == ERR_ARG
    assert lib.v2a_nchw_to_nhwc4p(None, 0, None, 2, 128, 128, 3, 1, None) == ERR_ARG
    assert lib.v2a_opt_step_packed(None, None, 0, None, None, 1, None, 0, 0, None) == ERR_ARG
    assert lib.v2a_opt_presum(None, None, 0, 0, None, None) == ERR_ARG
    assert lib.v2a_groupnorm_takes_post(16, 256, 8) in (0, 1)
    # round 5: the post-activation addend of a GroupNorm launch is an explicit operand set of v2a_groupnorm_fwd_s (slabs without a count: rejected)
    assert lib.v2a_groupnorm_fwd_s(None, None, 0, None, None, None, None, 0, None, None, 0, None, None, 1, 16, 64, 8, 1e-5, 0, None, 0, 0, None,
                                   None, None, 3, 0, None, None, 0, None) == ERR_ARG
    # pre-split three-plane conv: null operands / a shape its 64 x 64 plan does not take
    assert lib.v2a_conv2d_fwd_p3(None, 0, None, 0, None, 0, None, None, None, None, 64, 1, 4, 1024, 0, 1024, 1, 5, 1, 1, 0, 2, 1, 4, None, None, 0, None) == ERR_ARG
    assert lib.v2a_conv2d_p3_eligible(256, 1024, 5120, 1024, 0) == 1 and lib.v2a_conv2d_p3_eligible(65536, 64, 576, 64, 0) == 0
    assert lib.v2a_conv2d_p3_eligible(256, 1024, 35, 7, 0) == 0 and lib.v2a_split3_f32(None, None, 8, 8, None) == ERR_ARG
    # host-side sizing of the split slabs / partials
    assert lib.v2a_conv2d_wgrad_workspace_bytes(65536, 64, 576) > 0
    assert lib.v2a_conv2d_wgrad_h_workspace_bytes(229376, 128, 1152) >= 128 * 1152 * 4 * 2
    assert lib.v2a_colsum_batched_workspace_bytes(2, 114688, 128) >= 2 * 128 * 8
    assert lib.v2a_video_loss_workspace_bytes(4) >= 4 * 8


def test_persistent_denoiser_structs_match_the_library():
    """v2a_hip/policy_persist.py mirrors the op / argument structs of csrc/policy_persist.hip with ctypes; the library reports their sizes."""
    from v2a_hip.policy_persist import PPOp, PPArgs, PPSrc
    from v2a_hip._lib import lib
    assert ctypes.sizeof(PPOp) == lib.v2a_policy_persist_op_bytes()
    assert ctypes.sizeof(PPArgs) == lib.v2a_policy_persist_args_bytes()
    assert ctypes.sizeof(PPOp) == 2 * ctypes.sizeof(PPSrc) + 3 * 8 + 14 * 4
    assert lib.v2a_policy_persist_waves_per_wg() in (4, 8, 16)
    # shapes the kernel takes / refuses: (B, Tin, Tout, Cin, K, stride, pad, type)
    assert lib.v2a_policy_persist_lds_bytes(1, 16, 16, 256, 5, 1, 2, 0) > 0
    assert lib.v2a_policy_persist_lds_bytes(2, 4, 4, 2048, 5, 1, 2, 0) > 0           # the largest layer input at batch 2
    assert lib.v2a_policy_persist_lds_bytes(4, 4, 4, 2048, 5, 1, 2, 0) == 0          # batch 4 does not fit the LDS
    assert lib.v2a_policy_persist_lds_bytes(1, 16, 16, 256, 7, 1, 3, 0) == 0         # no 7-tap instance
    assert lib.v2a_policy_persist_lds_bytes(1, 8, 16, 256, 4, 2, 1, 1) > 0           # Upsample1d


def test_small_map_conv_routing_is_decided_on_the_host():
    """conv_maps_x3's eligibility and slab plan are host logic (csrc/igemm_x3m.hip conv_maps_x3_eligible / _split): the four ResNet-18 layer
    shapes of the policy's encoders at batch 64 go to it, each with 256 workgroups (1 / 2 / 4 / 8 slabs over 32-channel chunks), and the
    workspace v2a_conv2d_dma_f32_workspace_bytes reports covers those slabs; launches too small to fill the chip, maps that are not
    32 / 16 / 8 / 4 wide, channel counts off the 32 / 64 grid and row counts off the 256-row tile stay on the older kernels; the debug
    switch turns the route off and back on."""
    from v2a_hip._lib import lib
    for N, S, C, slabs in ((64, 32, 64, 1), (64, 16, 128, 2), (64, 8, 256, 4), (64, 4, 512, 8)):
        assert lib.v2a_conv2d_x3m_eligible(N, S, C, C) == 1, (N, S, C)
        M = N * S * S
        need = slabs * M * C * 4 if slabs > 1 else 0
        assert lib.v2a_conv2d_dma_f32_workspace_bytes(M, C, 9 * C) >= need, (N, S, C)
    assert lib.v2a_conv2d_x3m_eligible(256, 32, 64, 64) == 1            # B = 256: 1024 tiles, one slab
    assert lib.v2a_conv2d_x3m_eligible(8, 32, 64, 64) == 0              # 32 tiles x 2 chunks: 64 workgroups
    assert lib.v2a_conv2d_x3m_eligible(64, 64, 64, 64) == 0             # 64-wide maps: the patch kernels
    assert lib.v2a_conv2d_x3m_eligible(64, 16, 120, 128) == 0 and lib.v2a_conv2d_x3m_eligible(64, 16, 128, 96) == 0
    assert lib.v2a_conv2d_x3m_eligible(3, 8, 256, 256) == 0             # 192 rows: no whole tile
    old = lib.v2a_debug_set_maps_kernel(0)
    try:
        assert lib.v2a_conv2d_x3m_eligible(64, 32, 64, 64) == 0
    finally:
        lib.v2a_debug_set_maps_kernel(old)
    assert lib.v2a_conv2d_x3m_eligible(64, 32, 64, 64) == 1
    assert lib.v2a_debug_set_smallk(-1) in (0, 1)                       # query form: no change
    assert lib.v2a_debug_set_smallk(-1) == lib.v2a_debug_set_smallk(-1)
