"""The oracle (oracle/mi_oracle.py) against the fixtures produced by the reference's own functions
(tests/golden/make_goldens.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import mi_oracle as orc


def _cases(npz, depth=1):
    return sorted({"/".join(k.split("/")[:depth]) for k in npz.files})


# ---------------------------------------------------------------- G0: critic structure (model.py:18-32)
def test_make_mlp_structure(golden):
    g = golden("g0_make_mlp.npz")
    assert list(g["keys"]) == ["0.weight", "0.bias", "2.weight", "2.bias", "4.weight", "4.bias"]
    assert list(g["kinds"]) == ["Linear", "ReLU", "Linear", "ReLU", "Linear"]
    assert list(g["shapes"]) == ["(1024, 1536)", "(1024,)", "(512, 1024)", "(512,)", "(1, 512)", "(1,)"]


# ---------------------------------------------------------------- G1: bound only (mi_critics.py)
@pytest.mark.parametrize("est", orc.ESTIMATORS)
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_bound_matches_reference(golden, est, prec):
    g = golden("g1_bound.npz")
    dt = torch.float32 if prec == "f32" else torch.float64
    for tag in _cases(g):
        pos, n = int(g[f"{tag}/pos_size"]), int(g[f"{tag}/n"])
        if tag.startswith("extreme"):
            logits = orc.hash_uniform((n, 1), 777) * 160.0
        else:
            logits = orc.hash_uniform((n, 1), 100 + n) * 6.0
        lg = logits.to(dt).clone().requires_grad_(True)
        loss = orc.bound_loss(lg, pos, est)
        loss.sum().backward()
        ref = g[f"{tag}/{est}/{prec}/loss"]
        assert tuple(loss.shape) == tuple(g[f"{tag}/{est}/{prec}/loss_shape"])
        # same torch ops in the same order: bit-exact
        np.testing.assert_array_equal(loss.detach().numpy(), ref)
        np.testing.assert_array_equal(lg.grad.numpy().reshape(-1), g[f"{tag}/{est}/{prec}/grad"])
        # closed-form gradient
        cf = orc.bound_grad_logits(logits.to(dt), pos).reshape(-1).numpy()
        np.testing.assert_allclose(cf, g[f"{tag}/{est}/{prec}/grad"], rtol=1e-5 if prec == "f32" else 1e-12,
                                   atol=1e-12 if prec == "f32" else 1e-20)


def test_dv_minus_infonce_is_log_n(golden):
    g = golden("g1_bound.npz")
    for tag in _cases(g):
        pos, n = int(g[f"{tag}/pos_size"]), int(g[f"{tag}/n"])
        d = float(g[f"{tag}/infonce/f64/loss"]) - float(g[f"{tag}/dv/f64/loss"][0])
        assert abs(d - np.log(np.float32(n - pos))) < 1e-6


# ---------------------------------------------------------------- G2: row order (main_utils.py:99-108)
def test_pair_order_matches_reference(golden):
    g = golden("g2_order.npz")
    for tag in _cases(g):
        sid = [str(s) for s in g[f"{tag}/sid"]]
        i, j = orc.pair_index(sid)
        np.testing.assert_array_equal(i, g[f"{tag}/i"])
        np.testing.assert_array_equal(j, g[f"{tag}/j"])
        il, jl = orc.pair_index_loops(sid)
        np.testing.assert_array_equal(i, np.array(il))
        np.testing.assert_array_equal(j, np.array(jl))


def test_pair_order_b4_documented(golden):
    # 14 rows for sid=[a,b,b,c]
    i, j = orc.pair_index(["a", "b", "b", "c"])
    assert list(zip(i[4:], j[4:])) == [(0, 1), (2, 3), (3, 0), (0, 2), (1, 3), (2, 0), (3, 1), (0, 3), (1, 0), (3, 2)]


# ---------------------------------------------------------------- G3/G4/G5: full step
def _check_digest(g, key, grad, rtol, atol_scale):
    a = grad.detach().double().numpy()
    flat = a.reshape(-1)
    scale = max(float(np.abs(flat).max()), 1e-30)
    if key.endswith("/db3"):
        scale = 1.0  # db3 = sum of all d loss/d score = (+1) + (-1): true value 0, addends O(1)
    atol = atol_scale * scale
    np.testing.assert_allclose(flat[g[f"{key}/sample_idx"]], g[f"{key}/sample"], rtol=rtol, atol=atol)
    np.testing.assert_allclose(np.sqrt((flat ** 2).sum()), g[f"{key}/fro"], rtol=max(rtol, 1e-5),
                               atol=atol * flat.size ** 0.5)
    if a.ndim == 2:
        # sums cancel heavily: tolerance scales with the number of addends
        np.testing.assert_allclose(a.sum(1), g[f"{key}/row_sums"], rtol=rtol, atol=atol * a.shape[1] ** 0.5 * 4)
        np.testing.assert_allclose(a.sum(0), g[f"{key}/col_sums"], rtol=rtol, atol=atol * a.shape[0] ** 0.5 * 4)


FULL_CASES = ["b8_d768", "b32_d768", "b16_d128", "b32_d128", "b16_d768_dup", "b32_d128_dup", "b24_d96x160_dup"]


@pytest.mark.parametrize("tag", FULL_CASES)
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_literal_step_matches_reference(golden, tag, prec):
    g = golden("g3_full_step.npz")
    b, di, dt_, dup, salt = [int(v) for v in g[f"{tag}/meta"]]
    dtype = torch.float32 if prec == "f32" else torch.float64
    x, y, sid, params = orc.synthetic_case(b, di, dt_, salt=salt, dup=bool(dup), dtype=dtype)
    rtol, atol_s = (2e-4, 2e-6) if prec == "f32" else (1e-9, 1e-12)
    for est in orc.ESTIMATORS:
        out = orc.literal_step(x, y, sid, params, est)
        k = f"{tag}/{est}/{prec}"
        assert out["scores"].numel() == int(g[f"{k}/n_rows"])
        np.testing.assert_allclose(out["loss"].numpy(), g[f"{k}/loss"], rtol=1e-6 if prec == "f32" else 1e-12)
        assert tuple(out["loss"].shape) == ((1,) if est == "dv" else ())
        if est == "dv":
            np.testing.assert_allclose(out["scores"].numpy(), g[f"{tag}/{prec}/scores"], rtol=rtol,
                                       atol=1e-6 if prec == "f32" else 1e-13)
            sc = max(float(np.abs(g[f"{k}/dx"]).max()), 1e-30)
            np.testing.assert_allclose(out["dx"].numpy(), g[f"{k}/dx"], rtol=rtol, atol=atol_s * sc)
            np.testing.assert_allclose(out["dy"].numpy(), g[f"{k}/dy"], rtol=rtol, atol=atol_s * sc)
            for pn, gr in zip(("w1", "b1", "w2", "b2", "w3", "b3"), out["dparams"]):
                _check_digest(g, f"{k}/d{pn}", gr, rtol, atol_s * 4)
        else:
            assert float(g[f"{k}/grad_maxdiff_vs_dv"]) <= (1e-9 if prec == "f32" else 1e-15)


@pytest.mark.parametrize("tag", FULL_CASES)
def test_factorised_matrix_form_matches_reference(golden, tag):
    """The B x B factorised form (what the kernels compute) against the reference outputs, fp64 and fp32."""
    g = golden("g3_full_step.npz")
    b, di, dt_, dup, salt = [int(v) for v in g[f"{tag}/meta"]]
    for prec, dtype, tol in (("f64", torch.float64, 1e-11), ("f32", torch.float32, 3e-5)):
        x, y, sid, params = orc.synthetic_case(b, di, dt_, salt=salt, dup=bool(dup), dtype=dtype)
        for est in orc.ESTIMATORS:
            out = orc.concat_matrix_step(x, y, sid, params, est)
            k = f"{tag}/{est}/{prec}"
            np.testing.assert_allclose(out["loss"].numpy(), g[f"{k}/loss"], rtol=tol * 10, atol=tol)
            assert tuple(out["loss"].shape) == ((1,) if est == "dv" else ())
            rows = orc.matrix_to_reference_rows(out["scores"], sid).numpy()
            np.testing.assert_allclose(rows, g[f"{tag}/{prec}/scores"], rtol=tol * 10, atol=tol)
            if est == "dv":
                sc = float(np.abs(g[f"{k}/dx"]).max())
                np.testing.assert_allclose(out["dx"].numpy(), g[f"{k}/dx"], rtol=tol * 100, atol=tol * sc)
                np.testing.assert_allclose(out["dy"].numpy(), g[f"{k}/dy"], rtol=tol * 100, atol=tol * sc)
                for pn, gr in zip(("w1", "b1", "w2", "b2", "w3", "b3"), out["dparams"]):
                    _check_digest(g, f"{k}/d{pn}", gr, tol * 100, tol * 10)
        # blocked variant used by the cpu_baseline
        if prec == "f32":
            blk = orc.concat_matrix_step(x, y, sid, params, "dv", row_block=max(b // 4, 1))
            full = orc.concat_matrix_step(x, y, sid, params, "dv")
            np.testing.assert_allclose(blk["loss"].numpy(), full["loss"].numpy(), rtol=1e-5, atol=3e-6)
            np.testing.assert_allclose(blk["dx"].numpy(), full["dx"].numpy(), rtol=1e-3,
                                       atol=3e-6 * float(full["dx"].abs().max()))
            np.testing.assert_allclose(blk["dparams"][2].numpy(), full["dparams"][2].numpy(), rtol=1e-3,
                                       atol=3e-5 * float(full["dparams"][2].abs().max()))


def test_matrix_grad_closed_form():
    x, y, sid, params = orc.synthetic_case(12, 32, 48, salt=3, dup=True, dtype=torch.float64)
    s = orc.concat_scores_matrix(x, y, params).requires_grad_(True)
    loss = orc.bound_from_matrix(s, sid, "dv")
    loss.sum().backward()
    np.testing.assert_allclose(s.grad.numpy(), orc.matrix_grad_scores(s.detach(), sid).numpy(), atol=1e-15)


@pytest.mark.parametrize("est", ["dv", "infonce"])
def test_concat_closed_form_backward_equals_autograd(est):
    """``concat_step_rounded`` (the checker of the bf16 concat-MLP kernels) states the backward in closed form; with the
    identity as rounding function it must reproduce autograd through the same forward to fp64 rounding."""
    x, y, sid, params = orc.synthetic_case(48, 24, 40, h1=64, h2=32, salt=3, dup=True, dtype=torch.float64)
    a = orc.concat_matrix_step(x, y, sid, params, est)
    c = orc.concat_step_rounded(x, y, sid, params, est, row_block=16, round_fn=lambda t: t)
    np.testing.assert_allclose(c["loss"].numpy(), a["loss"].numpy(), atol=1e-13)
    for got, ref in zip([c["dx"], c["dy"], *c["dparams"]], [a["dx"], a["dy"], *a["dparams"]]):
        np.testing.assert_allclose(got.reshape(ref.shape).numpy(), ref.numpy(), atol=1e-13 * max(1.0, float(ref.abs().max())),
                                   rtol=1e-10)
    # and the rounded variant stays in its neighbourhood (bf16 operands move these heavily cancelling gradients by tens
    # of percent of max|grad|; a sign or index error would move them by > 100 %)
    r = orc.concat_step_rounded(x, y, sid, params, est, row_block=16)
    assert float((r["dx"] - a["dx"]).abs().max()) < 0.5 * float(a["dx"].abs().max())


def test_invariants_a4():
    """dv = infonce - log N_neg; constant shift invariance; duplicate ids drop two pairs."""
    x, y, sid, params = orc.synthetic_case(10, 16, 16, salt=5, dtype=torch.float64)
    s = orc.concat_scores_matrix(x, y, params)
    dv = orc.bound_from_matrix(s, sid, "dv")
    inf = orc.bound_from_matrix(s, sid, "infonce")
    assert abs(float(inf - dv[0]) - np.log(90.0)) < 1e-6
    assert abs(float(orc.bound_from_matrix(s + 3.25, sid, "infonce") - inf)) < 1e-12
    sid2 = list(sid)
    sid2[3] = sid2[7]
    assert int(orc.negative_mask(sid2).sum()) == 90 - 2
    perm = torch.randperm(10, generator=torch.Generator().manual_seed(0))
    s_p = orc.concat_scores_matrix(x[perm], y[perm], params)
    assert abs(float(orc.bound_from_matrix(s_p, [sid[p] for p in perm], "infonce") - inf)) < 1e-12


def test_all_equal_ids_is_nonfinite():
    # reference failure mode: zero negatives -> logsumexp(empty) = -inf
    logits = torch.zeros(4, 1)
    assert not torch.isfinite(orc.dv_bound_loss(logits, 4)).all()
