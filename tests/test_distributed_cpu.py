"""world_size-2 gloo test of the global-batch path: result at G ranks == single-process oracle at the
same global batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dist_worker  # noqa: E402
from oracle import mi_oracle as orc  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("estimator", ["dv", "infonce"])
def test_global_batch_two_ranks_equals_single_process(tmp_path, estimator):
    world, b_local, d = 2, 6, 5
    mp.spawn(dist_worker.run, args=(world, _free_port(), b_local, d, estimator, str(tmp_path)), nprocs=world, join=True)
    b = world * b_local
    x, y, sid, _ = orc.synthetic_case(b, d, d, h1=8, h2=8, salt=21, dup=True, dtype=torch.float64)
    w = orc.hash_uniform((d, d), 99, torch.float64)
    ref = orc.matrix_step(lambda a, c, ww: orc.bilinear_scores(a, c, ww), [x, y, w], sid, estimator)
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=True) for r in range(world)]
    for r, o in enumerate(outs):
        sl = slice(r * b_local, (r + 1) * b_local)
        np.testing.assert_allclose(o["loss"].numpy().reshape(-1), ref["loss"].numpy().reshape(-1), rtol=1e-6, atol=1e-6)
        assert tuple(o["loss"].shape) == ((1,) if estimator == "dv" else ())
        np.testing.assert_allclose(o["dx"].numpy(), ref["grads"][0][sl].numpy(), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(o["dy"].numpy(), ref["grads"][1][sl].numpy(), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(o["dw"].numpy(), ref["grads"][2].numpy(), rtol=1e-9, atol=1e-12)
    # every rank holds bit-identical loss and parameter gradient (records merged in rank order, all-reduced grads)
    assert torch.equal(outs[0]["loss"], outs[1]["loss"])
    assert torch.equal(outs[0]["dw"], outs[1]["dw"])


@pytest.mark.parametrize("staged", [False, True])
def test_concat_exchange_two_ranks_equals_single_process(tmp_path, staged):
    """The reference critic (concat-MLP) sharded over two ranks: its six parameter gradients travel in one flat
    all-reduce; `staged` drives GlobalBatchGraphStep's call sequence instead of the autograd wrapper."""
    world, b_local, d = 2, 5, 4
    mp.spawn(dist_worker.run_concat, args=(world, _free_port(), b_local, d, "dv", str(tmp_path), staged), nprocs=world,
             join=True)
    b = world * b_local
    x, y, sid, params = orc.synthetic_case(b, d, d, h1=12, h2=8, salt=23, dup=True, dtype=torch.float64)
    ref = orc.concat_matrix_step(x, y, sid, params, "dv")
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=True) for r in range(world)]
    for r, o in enumerate(outs):
        sl = slice(r * b_local, (r + 1) * b_local)
        np.testing.assert_allclose(o["loss"].numpy().reshape(-1), ref["loss"].numpy().reshape(-1), rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(o["dx"].numpy(), ref["dx"][sl].numpy(), rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(o["dy"].numpy(), ref["dy"][sl].numpy(), rtol=1e-8, atol=1e-12)
        for got, want in zip(o["dparams"], ref["dparams"]):
            np.testing.assert_allclose(got.numpy().reshape(-1), want.numpy().reshape(-1), rtol=1e-8, atol=1e-12)
    for a, c in zip(outs[0]["dparams"], outs[1]["dparams"]):
        assert torch.equal(a, c)
    assert torch.equal(outs[0]["loss"], outs[1]["loss"])


@pytest.mark.parametrize("staged", [False, True])
def test_fp8_global_scales_two_ranks_equals_single_process(tmp_path, staged):
    """BASELINE configs[4] (fp8 critic on 8 GPUs) needs per-tensor scales of the WHOLE batch: the sharded path makes them
    global with two MAX all-reduces between the stages of the fp8 preparation (distributed.fp8_global_scales).  Two gloo
    ranks with an oracle-backed ops object staged exactly like the product's against the single-process fp8 oracle; the
    largest image entry sits on rank 1 only, so rank-local scales would give different numbers."""
    world, b_local, d = 2, 8, 16
    mp.spawn(dist_worker.run_fp8, args=(world, _free_port(), b_local, d, "infonce", str(tmp_path), staged), nprocs=world,
             join=True)
    b = world * b_local
    x, y, sid, _ = orc.synthetic_case(b, d, d, h1=8, h2=8, salt=31, dup=True, dtype=torch.float64)
    x[b - 1] *= 3.0
    w = orc.hash_uniform((d, d), 77, torch.float64)
    ref = orc.bilinear_step_fp8(x, y, w, sid, "infonce")
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=True) for r in range(world)]
    for r, o in enumerate(outs):
        sl = slice(r * b_local, (r + 1) * b_local)
        np.testing.assert_allclose(o["loss"].numpy().reshape(-1), ref["loss"].numpy().reshape(-1), rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(o["dx"].numpy(), ref["dx"][sl].numpy(), rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(o["dy"].numpy(), ref["dy"][sl].numpy(), rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(o["dw"].numpy(), ref["dw"].numpy(), rtol=1e-8, atol=1e-12)
    assert torch.equal(outs[0]["loss"], outs[1]["loss"]) and torch.equal(outs[0]["dw"], outs[1]["dw"])


@pytest.mark.parametrize("staged", [False, True])
@pytest.mark.parametrize("variant", ["split", "separable"])
def test_split_forward_and_separable_two_ranks(tmp_path, variant, staged):
    """VERDICT r2 item 7.  "split": the rank-local part of the forward (ops.prep_local: T = X W) is issued between the
    START of the input all-gathers and the wait for them, then forward() goes on from it -- same numbers as the
    single-process oracle, and the call log shows prep_local before every forward.  "separable": BASELINE configs[1]'s
    critic sharded by rows, two parameter gradients in the one flat all-reduce."""
    world, b_local, d = 2, 6, 5
    mp.spawn(dist_worker.run_variant, args=(world, _free_port(), b_local, d, "infonce", str(tmp_path), variant, staged),
             nprocs=world, join=True)
    b = world * b_local
    x, y, sid, _ = orc.synthetic_case(b, d, d, h1=8, h2=8, salt=21, dup=True, dtype=torch.float64)
    if variant == "split":
        params = [orc.hash_uniform((d, d), 99, torch.float64)]
        ref = orc.matrix_step(lambda a, c, ww: orc.bilinear_scores(a, c, ww), [x, y] + params, sid, "infonce")
    else:
        params = [orc.hash_uniform((d, 6), 41, torch.float64), orc.hash_uniform((d, 6), 42, torch.float64)]
        ref = orc.matrix_step(lambda a, c, g, h: orc.separable_scores(a, c, g, h), [x, y] + params, sid, "infonce")
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=True) for r in range(world)]
    for r, o in enumerate(outs):
        sl = slice(r * b_local, (r + 1) * b_local)
        np.testing.assert_allclose(o["loss"].numpy().reshape(-1), ref["loss"].numpy().reshape(-1), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(o["dx"].numpy(), ref["grads"][0][sl].numpy(), rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(o["dy"].numpy(), ref["grads"][1][sl].numpy(), rtol=1e-8, atol=1e-12)
        for got, want in zip(o["dparams"], ref["grads"][2:]):
            np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-8, atol=1e-12)
        if variant == "split":
            want = ["prep_local:12", "forward:12"] * (2 if staged else 1)
            assert o["calls"] == want, o["calls"]
    assert torch.equal(outs[0]["loss"], outs[1]["loss"])
    for a, c in zip(outs[0]["dparams"], outs[1]["dparams"]):
        assert torch.equal(a, c)


def test_split_tail_overlaps_the_reduce_scatter(tmp_path):
    """Round 4: the reduce-scatter of dY is STARTED between the backward's first launch (statistics, dX, partial dY) and its
    dW launch (ops.merge_backward_tail / backward_dw), so that it runs beside the latter.  Two gloo ranks, oracle-backed ops
    with the product ops' method names: the call log pins the order, the results equal the single-process oracle."""
    world, b_local, d = 2, 6, 5
    mp.spawn(dist_worker.run_raw, args=(world, _free_port(), b_local, d, "infonce", str(tmp_path), True), nprocs=world, join=True)
    b = world * b_local
    x, y, sid, _ = orc.synthetic_case(b, d, d, h1=8, h2=8, salt=21, dup=True, dtype=torch.float64)
    w = orc.hash_uniform((d, d), 99, torch.float64)
    ref = orc.matrix_step(lambda a, c, ww: orc.bilinear_scores(a, c, ww), [x, y, w], sid, "infonce")
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=True) for r in range(world)]
    for r, o in enumerate(outs):
        sl = slice(r * b_local, (r + 1) * b_local)
        np.testing.assert_allclose(o["loss"].numpy().reshape(-1), ref["loss"].numpy().reshape(-1), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(o["dx"].numpy(), ref["grads"][0][sl].numpy(), rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(o["dy"].numpy(), ref["grads"][1][sl].numpy(), rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(o["dparams"][0].numpy(), ref["grads"][2].numpy(), rtol=1e-8, atol=1e-12)
        assert o["calls"] == ["forward_raw", "merge_backward_tail:6", "reduce_scatter_start", "backward_dw"] * 2, o["calls"]
    assert torch.equal(outs[0]["loss"], outs[1]["loss"]) and torch.equal(outs[0]["dparams"][0], outs[1]["dparams"][0])


@pytest.mark.parametrize("estimator", ["dv", "infonce"])
def test_raw_record_protocol_two_ranks(tmp_path, estimator):
    """The staged step's raw-record mode (ops.forward_raw / merge_backward: what the bilinear critic's fused kernels use on
    a sharded batch -- no finalize and no merge launch): every rank's several records are gathered in rank order and
    merged inside the backward.  Oracle-backed ops, three records per rank; against the single-process oracle."""
    world, b_local, d = 2, 6, 5
    mp.spawn(dist_worker.run_raw, args=(world, _free_port(), b_local, d, estimator, str(tmp_path)), nprocs=world, join=True)
    b = world * b_local
    x, y, sid, _ = orc.synthetic_case(b, d, d, h1=8, h2=8, salt=21, dup=True, dtype=torch.float64)
    w = orc.hash_uniform((d, d), 99, torch.float64)
    ref = orc.matrix_step(lambda a, c, ww: orc.bilinear_scores(a, c, ww), [x, y, w], sid, estimator)
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=True) for r in range(world)]
    for r, o in enumerate(outs):
        sl = slice(r * b_local, (r + 1) * b_local)
        # (DV: the reference takes log N_neg in float32, mi_critics.py:10)
        np.testing.assert_allclose(o["loss"].numpy().reshape(-1), ref["loss"].numpy().reshape(-1), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(o["dx"].numpy(), ref["grads"][0][sl].numpy(), rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(o["dy"].numpy(), ref["grads"][1][sl].numpy(), rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(o["dparams"][0].numpy(), ref["grads"][2].numpy(), rtol=1e-8, atol=1e-12)
        assert o["calls"] == ["forward_raw", "merge_backward:6"] * 2, o["calls"]
    assert torch.equal(outs[0]["loss"], outs[1]["loss"]) and torch.equal(outs[0]["dparams"][0], outs[1]["dparams"][0])
