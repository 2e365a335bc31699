"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol declared in
include/mi_critic.h; the host wrappers validate eagerly and never fall back to the CPU."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from mutual_info_img_txt import _hip
    return _hip.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mi_critic.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lib):
    from mutual_info_img_txt import _hip
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mi_critic.h but not exported"
    assert sorted(_hip.SIGNATURES) == declared, "ctypes signature table out of sync with the header"
    assert lib.mi_abi_version() == 4


def test_workspace_queries_are_host_only(lib):
    assert lib.mi_bound_workspace_bytes(10) > 0
    assert lib.mi_pair_index_workspace_bytes(4096) >= (4096 * 4095 // 1024) * 12
    small = lib.mi_bilinear_workspace_bytes(64, 64, 128, 128, 1)
    big = lib.mi_bilinear_workspace_bytes(4096, 4096, 512, 512, 1)
    assert big > small and big >= 4096 * 4096 * 2
    assert lib.mi_concat_mlp_workspace_bytes(64, 64, 768, 768, 1024, 512, 1, 1) > 0


def test_argument_validation_without_gpu(lib):
    # null pointers are rejected before anything touches the device
    assert lib.mi_bound_fwd(None, 4, 2, 0, None, None, None, 0, None) == -1
    assert b"null" in lib.mi_last_error()
    assert lib.mi_merge_partials(None, 1, 1, 0, None, None, None) == -1


def test_round3_entry_points_validate_without_gpu(lib):
    """mi_bilinear_step / prep_local / fp8_stage / bwd_records reject bad arguments before anything touches the device,
    and mi_bilinear_raw_records is a pure host query."""
    import ctypes
    assert lib.mi_bilinear_step(None, None, None, None, 64, 128, 128, 1, 1, None, None, None, None, None, None, None, None, 0,
                                None) == -1
    assert b"null" in lib.mi_last_error()
    assert lib.mi_bilinear_prep_local(None, None, 64, 64, 128, 128, 1, None, 0, None) == -1
    assert lib.mi_bilinear_fp8_stage(None, None, None, 64, 64, 128, 128, 0, None, None, 0, None) == -1
    assert lib.mi_bilinear_bwd_records(None, None, None, None, None, 64, 64, 0, 128, 128, 1, 1, None, 0, 0, None, None, None,
                                       None, None, None, None, 0, None) == -1
    off = ctypes.c_size_t(0)
    # 512-row block of a 4096 batch at d = 512, bf16: 4 row blocks of 128 x 32 column splits (the plan fills the 256 CUs)
    # x 4 waves = 512 records of 16 bytes
    n = lib.mi_bilinear_raw_records(512, 4096, 512, 512, 1, ctypes.byref(off))
    assert n == 512 and off.value % 256 == 0
    assert off.value + 16 * n <= lib.mi_bilinear_workspace_bytes(512, 4096, 512, 512, 1)
    assert lib.mi_bilinear_raw_records(4096, 4096, 512, 512, 1, None) == 512  # 32 row blocks x 4 splits x 4 waves
    # shapes / precisions outside the fused kernels have no raw records
    assert lib.mi_bilinear_raw_records(512, 4096, 768, 768, 1, None) == 0      # width 768
    assert lib.mi_bilinear_raw_records(500, 4096, 512, 512, 1, None) == 0      # ragged row block
    assert lib.mi_bilinear_raw_records(512, 4096, 512, 512, 0, None) == 0      # f32


def test_no_cpu_fallback():
    from mutual_info_img_txt import mi_critics
    from mutual_info_img_txt._hip import MiCriticError
    from mutual_info_img_txt.model import make_mlp
    logits = torch.zeros(8, 1)
    with pytest.raises(MiCriticError):
        mi_critics.dv_bound_loss(logits, 2, "cpu")
    with pytest.raises(MiCriticError):
        mi_critics.infonce_bound_loss(logits, 2, "cpu")
    with pytest.raises(MiCriticError):
        mi_critics.fused_mi_bound(torch.zeros(4, 8), torch.zeros(4, 8), ["a", "b", "c", "d"], make_mlp(16, [8, 8]))


def test_host_logic():
    from mutual_info_img_txt import mi_critics
    from mutual_info_img_txt.model import BilinearCritic, SeparableCritic, make_mlp
    codes = mi_critics.study_id_codes(["s1", "s2", "s1", "s3"], "cpu")
    assert codes.dtype == torch.int64 and codes[0] == codes[2] and len(set(codes.tolist())) == 3
    assert mi_critics.study_id_codes(["50000002", "50000001"], "cpu").tolist() == [50000002, 50000001]  # same on every rank
    mlp = make_mlp(1536, [1024, 512])
    assert list(mlp.state_dict().keys()) == ["0.weight", "0.bias", "2.weight", "2.bias", "4.weight", "4.bias"]
    w1, b1, w2, b2, w3, b3 = mi_critics._concat_params(mlp)
    assert w1.shape == (1024, 1536) and w2.shape == (512, 1024) and w3.shape == (1, 512)
    with pytest.raises(ValueError):
        mi_critics._concat_params(make_mlp(16, [8]))  # only two hidden layers are fused
    with pytest.raises(KeyError):
        make_mlp(16, [8, 8], activation="tanh")  # same failure mode as the reference (model.py:21-23)
    with pytest.raises(ValueError):
        mi_critics._estimator_code("mine")
    assert BilinearCritic(8, 12).weight.shape == (8, 12)
    assert SeparableCritic(8, 12, 4)(torch.zeros(3, 8), torch.zeros(3, 12)).shape == (3, 3)


def test_make_mlp_matches_reference_structure(golden):
    from mutual_info_img_txt.model import make_mlp
    g = golden("g0_make_mlp.npz")
    mlp = make_mlp(1536, [1024, 512])
    assert [type(m).__name__ for m in mlp] == list(g["kinds"])
    assert list(mlp.state_dict().keys()) == list(g["keys"])
    assert [str(tuple(v.shape)) for v in mlp.state_dict().values()] == list(g["shapes"])


def test_path_queries(lib, caplog):
    """mi_bilinear_path / mi_separable_path are host-only; the binding warns ONCE per shape that leaves the fused kernels."""
    import logging
    from mutual_info_img_txt import _hip
    assert lib.mi_bilinear_path(4096, 4096, 512, 512, _hip.MI_PREC_BF16) == _hip.MI_PATH_FUSED_TAIL
    assert lib.mi_bilinear_path(4096, 4096, 512, 512, _hip.MI_PREC_F32) == _hip.MI_PATH_GENERIC
    assert lib.mi_bilinear_path(4096, 4096, 512, 512, _hip.MI_PREC_BF16X3) == _hip.MI_PATH_GEMMS
    assert lib.mi_bilinear_path(8192, 8192, 1024, 1024, _hip.MI_PREC_FP8) == _hip.MI_PATH_FP8_GEMMS
    assert lib.mi_bilinear_path(4000, 4000, 520, 520, _hip.MI_PREC_BF16) == _hip.MI_PATH_GEMMS    # width outside the fused kernel
    assert lib.mi_bilinear_path(0, 4096, 512, 512, 1) == -1
    assert lib.mi_separable_path(256, 256, 256, 256, 256, _hip.MI_PREC_BF16) >= _hip.MI_PATH_FUSED
    with caplog.at_level(logging.WARNING, logger="mutual_info_img_txt"):
        _hip._warned_paths.clear()
        assert _hip.note_path("bilinear", (4000, 4000, 520, 520), _hip.MI_PREC_BF16) == _hip.MI_PATH_GEMMS
        assert _hip.note_path("bilinear", (4000, 4000, 520, 520), _hip.MI_PREC_BF16) == _hip.MI_PATH_GEMMS
        assert _hip.note_path("bilinear", (4096, 4096, 512, 512), _hip.MI_PREC_BF16) == _hip.MI_PATH_FUSED_TAIL
    assert sum("outside the fused" in r.getMessage() for r in caplog.records) == 1

