"""The fused B x B stage of the bilinear critic (csrc/mi_bilinear_flash.h) against the oracle, through the C ABI.

bf16 mode tolerances (stated): against the oracle that rounds at the kernel's rounding points
(``orc.bilinear_step_rounded``): loss 2e-3 * max(1, |S|max); every gradient 1e-2 * max|grad|.
All tests need an MI355X:  python -m pytest tests -m gpu"""
import math

import numpy as np
import pytest
import torch

from oracle import mi_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from mutual_info_img_txt import _hip
    _hip.load()
    return torch.device("cuda:0")


def _critic(dev, w):
    from mutual_info_img_txt.model import BilinearCritic
    c = BilinearCritic(w.shape[0], w.shape[1])
    with torch.no_grad():
        c.weight.copy_(w)
    return c.to(dev)


def _run(dev, x, y, w, sid, est, precision="bf16"):
    from mutual_info_img_txt import _hip, mi_critics
    critic = _critic(dev, w)
    xl, yl = x.to(dev).requires_grad_(True), y.to(dev).requires_grad_(True)
    loss, stats = mi_critics.fused_mi_bound(xl, yl, sid, critic, est, precision=precision, return_stats=True)
    loss.sum().backward()
    torch.cuda.synchronize()
    return loss.detach().cpu(), _hip.stats_dict(stats), xl.grad.cpu(), yl.grad.cpu(), critic.weight.grad.cpu()


def _rel(got, ref):
    return float((got.double() - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)


@pytest.mark.parametrize("b,d", [(256, 512), (160, 256), (96, 128)])
def test_flash_structure_exact(dev, b, d):
    """S == 0 exactly (image features live in the first half of the width, text features in the second), so every
    exponential is exactly 1: U_i and V_j are integer sums of rows, which pins the transposed-read operand maps, the
    mask, the diagonal and the slab layout without any rounding in the way."""
    gen = torch.Generator().manual_seed(b + d)
    h = d // 2
    x = torch.zeros(b, d)
    y = torch.zeros(b, d)
    x[:, :h] = torch.randint(-3, 4, (b, h), generator=gen).float()
    y[:, h:] = torch.randint(-3, 4, (b, h), generator=gen).float()
    w = torch.eye(d)
    sid = torch.arange(b)
    sid[5] = sid[77 % b]
    sid[b - 1] = sid[b // 2]
    sid[1] = sid[0]
    loss, st, gx, gy, gw = _run(dev, x, y, w, sid, "infonce")
    neg = orc.negative_mask(sid).double()
    n_neg = int(neg.sum())
    assert st["n_neg"] == n_neg
    assert abs(float(loss) - math.log(n_neg)) < 1e-5
    eye = torch.eye(b, dtype=torch.float64)
    g = neg / n_neg - eye / b
    dt = g @ y.double()          # exact small rationals; the kernel rounds dT to bf16 for the next product
    dy = g.t() @ x.double()      # T == X here
    np.testing.assert_allclose(gy.numpy(), dy.numpy(), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(gx.numpy(), dt.numpy(), rtol=4e-3, atol=1e-7)     # dX = bf16(dT) W^T, W = I
    dw = x.double().t() @ orc.round_bf16(dt)
    assert _rel(gw, dw) < 5e-3


@pytest.mark.parametrize("b,dx,dy,est,dup", [(512, 512, 512, "infonce", True), (384, 200, 256, "dv", True),
                                            (64, 128, 128, "dv", False), (64, 128, 128, "dv", True),
                                            (256, 256, 256, "infonce", True), (1024, 512, 512, "dv", False),
                                            (1024, 512, 512, "dv", True)])
def test_flash_vs_rounded_oracle(dev, b, dx, dy, est, dup):
    """BASELINE configs[0] (B=64, d=128, DV), configs[1] shape (B=256, d=256, InfoNCE), configs[2] bilinear leg (B=1024,
    d=512, DV) and others, each also with SURVEY 8d's duplicated ids."""
    gen = torch.Generator().manual_seed(1000 * b + dx)
    x = torch.randn(b, dx, generator=gen)
    y = torch.randn(b, dy, generator=gen)
    w = torch.randn(dx, dy, generator=gen) * (0.25 / math.sqrt(dx))
    sid = torch.arange(b)
    if dup:
        for n in range(max(b // 8, 4)):
            sid[n] = n - (n % 2)
        sid[b - 3] = sid[b // 3]
    loss, st, gx, gy, gw = _run(dev, x, y, w, sid, est)
    o = orc.bilinear_step_rounded(x, y, w, sid, est)
    assert st["n_neg"] == int(orc.negative_mask(sid).sum())
    sc = float(o["scores"].abs().max())
    assert abs(float(loss.sum()) - float(o["loss"].sum())) < 2e-3 * max(sc, 1.0)
    assert tuple(loss.shape) == ((1,) if est == "dv" else ())
    for name, got, ref in (("dx", gx, o["dx"]), ("dy", gy, o["dy"]), ("dw", gw, o["dw"])):
        err = _rel(got, ref)
        assert err < 1e-2, (name, err)


def test_flash_reference_point_rescale(dev):
    """Force the rare branch (guide rule 26): the scores of the LAST streamed tiles are far above those of the first, so
    every wave raises its reference point mid-sweep and rescales its accumulators."""
    b, d = 512, 256
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(b, d, generator=gen)
    y = torch.randn(b, d, generator=gen)
    w = torch.eye(d) * 0.35
    # late rows of BOTH operands are scaled up: problem 0 streams the text rows, problem 1 the image rows
    x[b - 64:] *= 3.0
    y[b - 96:] *= 3.0
    sid = torch.arange(b)
    loss, st, gx, gy, gw = _run(dev, x, y, w, sid, "infonce")
    o = orc.bilinear_step_rounded(x, y, w, sid, "infonce")
    s = o["scores"]
    early = float(s[:, :32][orc.negative_mask(sid)[:, :32]].max())
    assert float(s.max()) - early > 60.0, "the case must cross the rescale threshold (kFlThr = 60)"
    assert abs(float(loss) - float(o["loss"])) < 2e-3 * float(s.abs().max())
    for name, got, ref in (("dx", gx, o["dx"]), ("dy", gy, o["dy"]), ("dw", gw, o["dw"])):
        assert _rel(got, ref) < 1e-2, name


@pytest.mark.parametrize("G", [1, 4])
def test_flash_trained_critic_distribution(dev, G):
    """The score distribution of a TRAINED critic: every diagonal positive lies more than kFlThr = 60 above
    every negative, a few ids are duplicated (those pairs are dropped, neither positive nor negative) and the FIRST
    streamed tile of most waves is a masked one (the diagonal tile, or one holding duplicates).  The unmasked maximum of
    such a tile would raise the reference point by > 60: both fix_masked_tile paths (the in-place masking and the full
    redo with a second decision) and a pending f_pend rescale are exercised.  Checked against the rounded fp64 oracle on
    the whole batch (G = 1) and as four row blocks merged in block order (G = 4)."""
    from mutual_info_img_txt import _hip
    from mutual_info_img_txt.distributed import HipBilinearOps
    b, d = 512, 256
    gen = torch.Generator().manual_seed(23)
    y = torch.randn(b, d, generator=gen)
    y = y / y.norm(dim=1, keepdim=True)
    x = y + 0.05 * torch.randn(b, d, generator=gen)      # image i matches text i
    w = torch.eye(d) * 150.0                             # positives ~ 150, negatives ~ 150 * cos(random) within +-56
    sid = torch.arange(b)
    for i in (0, 1, 2, 33, 64, 65, 300):                 # duplicates inside the first tile, inside a later one, far apart
        sid[i + 100 if i == 300 else i + 1] = sid[i]
    o = orc.bilinear_step_rounded(x, y, w, sid, "infonce")
    s, neg = o["scores"], orc.negative_mask(sid)
    assert float(torch.diagonal(s).min()) - float(s[neg].max()) > 60.0, "positives must clear every negative by kFlThr"
    if G == 1:
        loss, st, gx, gy, gw = _run(dev, x, y, w, sid, "infonce")
        assert st["n_neg"] == int(neg.sum())
    else:
        ops = HipBilinearOps()
        xd, yd, wd, sd = x.to(dev), y.to(dev), w.to(dev), sid.to(dev)
        br = b // G
        recs, saved = [], []
        for g in range(G):
            rec, sv = ops.forward(xd[g * br:(g + 1) * br].contiguous(), yd, [wd], sd[g * br:(g + 1) * br].contiguous(), sd,
                                  g * br, 1, 1, True)
            recs.append(rec)
            saved.append(sv)
        loss, stats = ops.merge(torch.stack(recs), b, 1)
        go = torch.ones(1, device=dev)
        gx, gy, gw = torch.empty_like(xd), torch.zeros_like(yd), torch.zeros_like(wd)
        for g in range(G):
            a, c, (e,) = ops.backward(saved[g], stats, go)
            gx[g * br:(g + 1) * br] = a
            gy += c
            gw += e
        assert _hip.stats_dict(stats)["n_neg"] == int(neg.sum())
        loss, gx, gy, gw = loss.cpu(), gx.cpu(), gy.cpu(), gw.cpu()
    assert abs(float(loss.sum()) - float(o["loss"])) < 2e-3 * float(s.abs().max())
    for name, got, ref in (("dx", gx, o["dx"]), ("dy", gy, o["dy"]), ("dw", gw, o["dw"])):
        assert _rel(got, ref) < 1e-2, (name, _rel(got, ref))


def test_flash_no_negatives_and_single_block(dev):
    """All ids equal: no negative pair at all -> the same non-finite loss as the reference (logsumexp of nothing)."""
    b, d = 64, 128
    gen = torch.Generator().manual_seed(9)
    x, y = torch.randn(b, d, generator=gen), torch.randn(b, d, generator=gen)
    w = torch.eye(d)
    from mutual_info_img_txt import mi_critics
    loss = mi_critics.fused_mi_bound(x.to(dev), y.to(dev), torch.zeros(b, dtype=torch.int64), _critic(dev, w), "infonce",
                                     precision="bf16")
    assert not math.isfinite(float(loss))


@pytest.mark.parametrize("b,d,G", [(512, 256, 4), (256, 128, 2), (1024, 512, 8)])
def test_flash_row_blocks_equal_full_batch(dev, b, d, G):
    """Row-block sharding through the C ABI on one GPU: per-block forward records merged in block order give the full
    batch's statistics; per-block backward outputs add up to the full batch's gradients."""
    from mutual_info_img_txt.distributed import HipBilinearOps
    from mutual_info_img_txt import _hip
    gen = torch.Generator().manual_seed(b + G)
    x = torch.randn(b, d, generator=gen)
    y = torch.randn(b, d, generator=gen)
    w = torch.randn(d, d, generator=gen) * (0.25 / math.sqrt(d))
    sid = torch.arange(b)
    sid[3] = sid[b - 5]
    ops = HipBilinearOps()
    xd, yd, wd, sd = x.to(dev), y.to(dev), w.to(dev), sid.to(dev)
    br = b // G
    recs, saved = [], []
    for g in range(G):
        rec, sv = ops.forward(xd[g * br:(g + 1) * br].contiguous(), yd, [wd], sd[g * br:(g + 1) * br].contiguous(), sd,
                              g * br, 1, 1, True)
        recs.append(rec)
        saved.append(sv)
    loss, stats = ops.merge(torch.stack(recs), b, 1)
    go = torch.ones(1, device=dev)
    gx = torch.empty_like(xd)
    gy = torch.zeros_like(yd)
    gw = torch.zeros_like(wd)
    for g in range(G):
        a, c, (e,) = ops.backward(saved[g], stats, go)
        gx[g * br:(g + 1) * br] = a
        gy += c
        gw += e
    o = orc.bilinear_step_rounded(x, y, w, sid, "infonce")
    assert _hip.stats_dict(stats)["n_neg"] == int(orc.negative_mask(sid).sum())
    assert abs(float(loss) - float(o["loss"])) < 2e-3 * float(o["scores"].abs().max())
    for name, got, ref in (("dx", gx.cpu(), o["dx"]), ("dy", gy.cpu(), o["dy"]), ("dw", gw.cpu(), o["dw"])):
        assert _rel(got, ref) < 1e-2, name
    # one block's partial dY against the oracle's row-block form
    ob = orc.bilinear_step_rounded(x, y, w, sid, "infonce", row_block=(br, 2 * br))
    _, c1, _ = ops.backward(saved[1], stats, go)
    assert _rel(c1.cpu(), ob["dy"]) < 1e-2


@pytest.mark.parametrize("b,d,G,prec", [(1024, 512, 4, "bf16"), (512, 256, 2, "bf16x3"), (256, 128, 2, "bf16")])
def test_split_forward_equals_the_one_call(dev, b, d, G, prec):
    """mi_bilinear_prep_local (bf16 X, W and T = X W of the rank's own rows -- what a sharded run issues while the text
    all-gather is in flight) followed by mi_bilinear_fwd with bit 2 of need_grad, against mi_bilinear_fwd alone: same
    kernels on the same values, so the record and all three gradients are bit-identical.  Y and the column ids are
    written only AFTER prep_local returned (it must not have read them)."""
    from mutual_info_img_txt.distributed import HipBilinearOps
    from mutual_info_img_txt.mi_critics import _precision_code
    gen = torch.Generator().manual_seed(b + d)
    x = torch.randn(b, d, generator=gen).to(dev)
    y = torch.randn(b, d, generator=gen).to(dev)
    w = (torch.randn(d, d, generator=gen) * (0.25 / math.sqrt(d))).to(dev)
    sid = torch.arange(b)
    sid[3] = sid[b - 5]
    sid = sid.to(dev)
    pc = _precision_code(prec)
    br = b // G
    g = G - 1
    xs, ss = x[g * br:(g + 1) * br].contiguous(), sid[g * br:(g + 1) * br].contiguous()
    go = torch.ones(1, device=dev)
    one = HipBilinearOps()
    rec1, sv1 = one.forward(xs, y, [w], ss, sid, g * br, 1, pc, True)
    two = HipBilinearOps()
    y_late, sid_late = torch.full_like(y, float("nan")), torch.zeros_like(sid)
    if prec != "bf16":  # the fp32-tolerance mode runs on the tiled kernels, not the fused one: declined, nothing staged
        assert two.prep_local(xs, [w], b, pc) is False and two._local_ws is None
        return
    assert two.prep_local(xs, [w], b, pc) is True
    y_late.copy_(y)
    sid_late.copy_(sid)
    rec2, sv2 = two.forward(xs, y_late, [w], ss, sid_late, g * br, 1, pc, True)
    assert two._local_ws is None
    loss, stats = one.merge(rec1.reshape(1, -1), br, 1)
    a1, c1, (e1,) = one.backward(sv1, stats, go)
    a2, c2, (e2,) = two.backward(sv2, stats, go)
    torch.cuda.synchronize()
    assert torch.equal(rec1, rec2)
    assert torch.equal(a1, a2) and torch.equal(c1, c2) and torch.equal(e1, e2)
    # a shape outside the fused kernels: prep_local declines, forward alone does everything
    odd = HipBilinearOps()
    assert odd.prep_local(xs[:, :d - 4].contiguous(), [w[:d - 4].contiguous()], b, pc) is False and odd._local_ws is None


@pytest.mark.parametrize("b,d,G", [(1024, 512, 2), (512, 256, 4), (4096, 512, 8)])  # the last: BASELINE configs[3]
def test_raw_record_row_blocks_equal_record_path(dev, b, d, G):
    """Sharded runs without the finalize / merge launches: mi_bilinear_fwd(need_grad | 8) per row block, the blocks' raw
    per-wave records concatenated in block order (= the all-gather), mi_bilinear_bwd_records per block.  Every block gets
    the same loss bits (same records, same merge order on every workgroup); loss and gradients agree with the record path
    (finalize per block, mi_merge_partials, mi_bilinear_bwd) to fp32 rounding -- the merge ORDER differs -- and with the
    rounded oracle at the usual 1e-2."""
    from mutual_info_img_txt.distributed import HipBilinearOps
    from mutual_info_img_txt import _hip
    gen = torch.Generator().manual_seed(7 * b + G)
    x = torch.randn(b, d, generator=gen)
    y = torch.randn(b, d, generator=gen)
    w = torch.randn(d, d, generator=gen) * (0.25 / math.sqrt(d))
    sid = torch.arange(b)
    sid[3] = sid[b - 5]
    xd, yd, wd, sd = x.to(dev), y.to(dev), w.to(dev), sid.to(dev)
    br = b // G
    go = torch.ones(1, device=dev)
    ops = HipBilinearOps()
    raw, saved = [], []
    for g in range(G):
        got = ops.forward_raw(xd[g * br:(g + 1) * br].contiguous(), yd, [wd], sd[g * br:(g + 1) * br].contiguous(), sd, g * br, 1, 1)
        assert got is not None
        raw.append(got[0].clone())
        saved.append(got[1])
    records = torch.cat(raw).contiguous()
    assert records.shape == (G * raw[0].shape[0], 4)
    gx, gy, gw = torch.empty_like(xd), torch.zeros_like(yd), torch.zeros_like(wd)
    losses = []
    for g in range(G):
        loss, stats, a, c, (e,) = ops.merge_backward(saved[g], records, b, 1, go)
        losses.append(loss.clone())
        gx[g * br:(g + 1) * br] = a
        gy += c
        gw += e
    for l in losses[1:]:
        assert torch.equal(l, losses[0])
    assert _hip.stats_dict(stats)["n_neg"] == int(orc.negative_mask(sid).sum())
    # the record path on the same blocks
    recs, saved2 = [], []
    for g in range(G):
        rec, sv = ops.forward(xd[g * br:(g + 1) * br].contiguous(), yd, [wd], sd[g * br:(g + 1) * br].contiguous(), sd, g * br, 1, 1, True)
        recs.append(rec)
        saved2.append(sv)
    loss2, stats2 = ops.merge(torch.stack(recs), b, 1)
    hx, hy, hw = torch.empty_like(xd), torch.zeros_like(yd), torch.zeros_like(wd)
    for g in range(G):
        a, c, (e,) = ops.backward(saved2[g], stats2, go)
        hx[g * br:(g + 1) * br] = a
        hy += c
        hw += e
    assert abs(float(losses[0]) - float(loss2)) <= 2e-6 * max(1.0, abs(float(loss2)))
    for name, got, ref in (("dx", gx, hx), ("dy", gy, hy), ("dw", gw, hw)):
        assert _rel(got.cpu(), ref.cpu()) < 1e-4, name
    o = orc.bilinear_step_rounded(x, y, w, sid, "infonce")
    assert abs(float(losses[0]) - float(o["loss"])) < 2e-3 * float(o["scores"].abs().max())
    for name, got, ref in (("dx", gx.cpu(), o["dx"]), ("dy", gy.cpu(), o["dy"]), ("dw", gw.cpu(), o["dw"])):
        assert _rel(got, ref) < 1e-2, name
    # a shape outside the fused kernels has no raw records
    assert ops.forward_raw(xd[:br, :d - 4].contiguous(), yd[:, :d - 4].contiguous(), [wd[:d - 4, :d - 4].contiguous()],
                           sd[:br].contiguous(), sd, 0, 1, 1) is None


@pytest.mark.parametrize("b,d,k,G", [(512, 256, 256, 4), (256, 128, 64, 2)])
def test_separable_row_blocks_equal_full_batch(dev, b, d, k, G):
    """The separable critic (BASELINE configs[1]) on the sharded path's ops object: per-block records merged in block
    order and per-block gradients summed against the rounded oracle of the full batch."""
    from mutual_info_img_txt.distributed import HipSeparableOps
    gen = torch.Generator().manual_seed(b + k)
    x = torch.randn(b, d, generator=gen)
    y = torch.randn(b, d, generator=gen)
    wg = torch.randn(d, k, generator=gen) * (0.5 / math.sqrt(d))
    wh = torch.randn(d, k, generator=gen) * (0.5 / math.sqrt(d))
    sid = torch.arange(b)
    sid[3] = sid[b - 5]
    ops = HipSeparableOps()
    xd, yd, gd, hd, sd = (t.to(dev) for t in (x, y, wg, wh, sid))
    br = b // G
    recs, saved = [], []
    for g in range(G):
        rec, sv = ops.forward(xd[g * br:(g + 1) * br].contiguous(), yd, [gd, hd], sd[g * br:(g + 1) * br].contiguous(), sd,
                              g * br, 1, 1, True)
        recs.append(rec)
        saved.append(sv)
    loss, stats = ops.merge(torch.stack(recs), b, 1)
    go = torch.ones(1, device=dev)
    gx, gy, gg, gh = torch.empty_like(xd), torch.zeros_like(yd), torch.zeros_like(gd), torch.zeros_like(hd)
    for g in range(G):
        a, c, (e, f) = ops.backward(saved[g], stats, go)
        gx[g * br:(g + 1) * br] = a
        gy += c
        gg += e
        gh += f
    o = orc.separable_step_rounded(x, y, wg, wh, sid, "infonce")
    assert abs(float(loss) - float(o["loss"])) < 2e-3 * float(o["scores"].abs().max())
    for name, got, ref in (("dx", gx, o["dx"]), ("dy", gy, o["dy"]), ("dwg", gg, o["dwg"]), ("dwh", gh, o["dwh"])):
        assert _rel(got.cpu(), ref) < 1e-2, name


def test_flash_unaligned_row_offset_and_flag_values(dev):
    """A row block whose offset is NOT a multiple of 32: the diagonal of the pair matrix then cuts through the 32 x 32
    tiles instead of lying on their main diagonals, the tile flags must say "general" (2) there and the kernel must take
    the exact id compares.  Three blocks [16, 80), [80, 144), [144, 208) of a batch of 256 (rows 0..15 and 208..255 are
    columns only); the merged statistics and the summed dX / dY / dW are checked against the oracle restricted to those
    rows."""
    from mutual_info_img_txt.distributed import HipBilinearOps
    from mutual_info_img_txt import _hip
    b, d, br, off0 = 256, 128, 64, 16
    gen = torch.Generator().manual_seed(99)
    x = torch.randn(b, d, generator=gen)
    y = torch.randn(b, d, generator=gen)
    w = torch.randn(d, d, generator=gen) * (0.25 / math.sqrt(d))
    sid = torch.arange(b)
    sid[40] = sid[41]
    ops = HipBilinearOps()
    xd, yd, wd, sd = x.to(dev), y.to(dev), w.to(dev), sid.to(dev)
    rows = list(range(off0, off0 + 3 * br))
    recs, saved = [], []
    for g in range(3):
        lo = off0 + g * br
        rec, sv = ops.forward(xd[lo:lo + br].contiguous(), yd, [wd], sd[lo:lo + br].contiguous(), sd, lo, 1, 1, True)
        recs.append(rec)
        saved.append(sv)
    # the oracle on the same rows: scores of rows [16, 208) against all columns
    o = orc.bilinear_step_rounded(x, y, w, sid, "infonce")
    s = o["scores"][rows]
    neg = orc.negative_mask(sid)[rows]
    lse = torch.logsumexp(s[neg], dim=0)
    pos = torch.stack([s[k, rows[k]] for k in range(len(rows))]).mean()
    loss, stats = ops.merge(torch.stack(recs), len(rows), 1)
    st = _hip.stats_dict(stats)
    assert st["n_neg"] == int(neg.sum())
    assert abs(float(loss) - float(lse - pos)) < 2e-3 * float(s.abs().max())


@pytest.mark.parametrize("b", [32, 64, 96, 128, 160, 224])
def test_flash_small_batches_d512(dev, b):
    """One to seven streamed tiles at d = 512: the pipeline's start-up and tail variants (first tile's softmax head run in
    place, last tile without a following score product), partially filled 128-row blocks, and the forward-only kernel
    (no gradient requested) against the gradient-accumulating one."""
    from mutual_info_img_txt import mi_critics
    d = 512
    gen = torch.Generator().manual_seed(b)
    x = torch.randn(b, d, generator=gen)
    y = torch.randn(b, d, generator=gen)
    w = torch.randn(d, d, generator=gen) * (0.25 / math.sqrt(d))
    sid = torch.arange(b)
    sid[b - 1] = sid[2]
    loss, st, gx, gy, gw = _run(dev, x, y, w, sid, "dv")
    o = orc.bilinear_step_rounded(x, y, w, sid, "dv")
    assert st["n_neg"] == int(orc.negative_mask(sid).sum())
    assert abs(float(loss.sum()) - float(o["loss"].sum())) < 2e-3 * max(float(o["scores"].abs().max()), 1.0)
    for name, got, ref in (("dx", gx, o["dx"]), ("dy", gy, o["dy"]), ("dw", gw, o["dw"])):
        assert _rel(got, ref) < 1e-2, name
    with torch.no_grad():
        loss_fwd = mi_critics.fused_mi_bound(x.to(dev), y.to(dev), sid, _critic(dev, w), "dv", precision="bf16")
    # same scores, same masks; the two kernels keep their own reference points, so the sums differ in rounding only
    assert abs(float(loss_fwd.sum()) - float(loss.sum())) < 1e-4 * max(abs(float(loss.sum())), 1.0)


@pytest.mark.parametrize("b,dx,dy,est,dup", [(1024, 512, 512, "infonce", True), (4096, 512, 512, "dv", False),
                                            (512, 256, 128, "dv", True), (256, 64, 256, "infonce", False),
                                            (96, 128, 128, "dv", True)])
def test_one_call_step_equals_the_two_calls(dev, b, dx, dy, est, dup):
    """mi_bilinear_step (one C-ABI call: the statistics are merged by the launch that also produces dT / grad_y / grad_x,
    no finalize launch) against mi_bilinear_fwd + mi_bilinear_bwd on the same inputs: the same kernels and the same
    merge order, so loss, statistics and every gradient must agree BIT FOR BIT; and against the rounded oracle at the
    stated bf16 tolerances.  (b = 96: a shape the two-launch tail does not take -- the step then runs the two calls.)"""
    from mutual_info_img_txt import _hip
    lib = _hip.load()
    gen = torch.Generator().manual_seed(b + 7 * dx)
    x = torch.randn(b, dx, generator=gen)
    y = torch.randn(b, dy, generator=gen)
    w = torch.randn(dx, dy, generator=gen) * (0.25 / math.sqrt(dx))
    sid = torch.arange(b)
    if dup:
        for n in range(max(b // 8, 4)):
            sid[n] = n - (n % 2)
        sid[b - 3] = sid[b // 3]
    xd, yd, wd, sd = x.to(dev), y.to(dev), w.to(dev), sid.to(dev)
    code, prec = _hip.ESTIMATORS[est], _hip.MI_PREC_BF16
    nbytes = lib.mi_bilinear_workspace_bytes(b, b, dx, dy, prec)
    go = torch.full((1,), 0.75, device=dev)

    def outputs():
        return (torch.zeros(1, device=dev), _hip.new_stats(dev), torch.zeros(_hip.RECORD_FLOATS, device=dev),
                torch.zeros_like(xd), torch.zeros_like(yd), torch.zeros_like(wd), _hip.workspace(nbytes, dev))
    l1, s1, r1, gx1, gy1, gw1, ws1 = outputs()
    _hip.call("mi_bilinear_step", dev, xd.data_ptr(), yd.data_ptr(), wd.data_ptr(), sd.data_ptr(), b, dx, dy, code, prec,
              go.data_ptr(), l1.data_ptr(), s1.data_ptr(), r1.data_ptr(), gx1.data_ptr(), gy1.data_ptr(), gw1.data_ptr(),
              ws1.data_ptr(), ws1.numel())
    l2, s2, r2, gx2, gy2, gw2, ws2 = outputs()
    _hip.call("mi_bilinear_fwd", dev, xd.data_ptr(), yd.data_ptr(), wd.data_ptr(), sd.data_ptr(), sd.data_ptr(), b, b, 0, dx, dy,
              code, prec, 1, l2.data_ptr(), s2.data_ptr(), r2.data_ptr(), None, ws2.data_ptr(), ws2.numel())
    _hip.call("mi_bilinear_bwd", dev, xd.data_ptr(), yd.data_ptr(), wd.data_ptr(), sd.data_ptr(), sd.data_ptr(), b, b, 0, dx, dy,
              prec, s2.data_ptr(), go.data_ptr(), gx2.data_ptr(), gy2.data_ptr(), gw2.data_ptr(), ws2.data_ptr(), ws2.numel(), 1)
    torch.cuda.synchronize()
    assert torch.equal(l1, l2) and torch.equal(s1, s2) and torch.equal(r1, r2)
    assert torch.equal(gx1, gx2) and torch.equal(gy1, gy2) and torch.equal(gw1, gw2)
    o = orc.bilinear_step_rounded(x, y, w, sid, est)
    sc = float(o["scores"].abs().max())
    assert abs(float(l1) - float(o["loss"].sum())) < 2e-3 * max(sc, 1.0)
    assert _hip.stats_dict(s1)["n_neg"] == int(orc.negative_mask(sid).sum())
    for name, got, ref in (("dx", gx1, o["dx"]), ("dy", gy1, o["dy"]), ("dw", gw1, o["dw"])):
        err = _rel(got.cpu() / 0.75, ref)
        assert err < 1e-2, (name, err)
