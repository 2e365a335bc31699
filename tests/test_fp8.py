"""fp8 mode of the bilinear critic (BASELINE configs[4]; csrc/mi_fp8.h) against the oracle fed identically quantised
inputs (SURVEY.md hazard H7: orc.bilinear_step_fp8), through the public Python entry point and the C ABI.

Stated tolerances: loss 2e-3 * max(1, |S|max); every gradient 2e-2 * max|grad| (the oracle models the e4m3 quantisers and
the bf16 roundings of G and dT; what is left is fp32 accumulation order, the native exponential, and the rare element of
T that sits on an e4m3 rounding boundary and comes out one step apart under the fp32- and the fp64-computed scale).
GPU tests need an MI355X: python -m pytest tests -m gpu"""
import math
import os

import numpy as np
import pytest
import torch

from oracle import mi_oracle as orc


def test_e4m3_quantiser_matches_torch_cast():
    """The oracle's explicit e4m3 rounding against torch's float8_e4m3fn cast (OCP e4m3fn, the format of gfx950)."""
    gen = torch.Generator().manual_seed(0)
    v = torch.cat([torch.randn(200000, generator=gen) * 120.0, torch.randn(50000, generator=gen) * 0.01,
                   torch.tensor([0.0, 448.0, -448.0, 447.9, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -9, 2.5 * 2.0 ** -9, 0.0156])])
    v = v.clamp(-448.0, 448.0)
    assert torch.equal(orc.quant_e4m3(v), v.to(torch.float8_e4m3fn).float())
    x = torch.randn(48, 64, generator=gen)
    s = orc.fp8_scale(x)
    assert float(s) == float(x.abs().max()) / 448.0 or abs(float(s) - float(x.abs().max()) / 448.0) < 1e-9
    assert float(orc.quant_e4m3(x / s).abs().max()) == 448.0


@pytest.fixture(scope="module")
def dev():
    from mutual_info_img_txt import _hip
    _hip.load()
    return torch.device("cuda:0")


def _dup_ids(b):
    sid = list(range(b))
    for n in range(max(b // 8, 2)):
        sid[n] = n - (n % 2)
    return sid


@pytest.mark.gpu
@pytest.mark.parametrize("b,dx,dy,est,dup", [(64, 128, 128, "dv", False), (256, 256, 256, "infonce", True),
                                            (200, 80, 48, "dv", True), (1024, 1024, 1024, "infonce", True),
                                            # BASELINE configs[4] at its own size on one GPU (B=8192, d=1024), and half of it
                                            (4096, 1024, 1024, "infonce", True), (8192, 1024, 1024, "infonce", False),
                                            (8192, 1024, 1024, "infonce", True)])
def test_fp8_bilinear_vs_quantised_oracle(dev, b, dx, dy, est, dup):
    from mutual_info_img_txt import _hip, mi_critics
    from mutual_info_img_txt.model import BilinearCritic
    gen = torch.Generator().manual_seed(7 * b + dx)
    x = torch.randn(b, dx, generator=gen)
    y = torch.randn(b, dy, generator=gen)
    w = torch.randn(dx, dy, generator=gen) * (0.3 / math.sqrt(dx))
    sid = _dup_ids(b) if dup else list(range(b))
    critic = BilinearCritic(dx, dy)
    with torch.no_grad():
        critic.weight.copy_(w)
    critic.to(dev)
    xl, yl = x.to(dev).requires_grad_(True), y.to(dev).requires_grad_(True)
    loss, scores = mi_critics.fused_mi_bound(xl, yl, sid, critic, est, precision="fp8", return_scores=True)
    loss.sum().backward()
    torch.cuda.synchronize()
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    o = orc.bilinear_step_fp8(x, y, w, sid, est)
    sc = max(float(o["scores"].abs().max()), 1.0)
    serr = float((scores.cpu().double() - o["scores"]).abs().max())
    errs = {n: float((g.cpu().double() - r).abs().max()) / float(r.abs().max())
            for n, g, r in (("dx", xl.grad, o["dx"]), ("dy", yl.grad, o["dy"]), ("dw", critic.weight.grad, o["dw"]))}
    print(f"fp8 B={b}: loss {float(loss.sum()):.6f} vs {float(o['loss'].sum()):.6f}, score err {serr:.2e} (|S|max {sc:.2f}),",
          {k: f"{v:.1e}" for k, v in errs.items()})
    assert tuple(loss.shape) == ((1,) if est == "dv" else ())
    assert abs(float(loss.sum()) - float(o["loss"].sum())) < 2e-3 * sc
    # a T element one e4m3 step apart moves a score row by ~ |t| |y| / 8: scores are checked in the mean, not the maximum
    assert float((scores.cpu().double() - o["scores"]).abs().mean()) < 1e-4 * sc
    for name, err in errs.items():
        assert err < 2e-2, (name, err)


@pytest.mark.gpu
def test_fp8_rejected_where_not_implemented(dev):
    from mutual_info_img_txt import _hip, mi_critics
    from mutual_info_img_txt.model import make_mlp
    x = torch.randn(32, 16, device=dev)
    with pytest.raises(ValueError):
        mi_critics.fused_mi_bound(x, x, list(range(32)), make_mlp(32, [64, 256]).to(dev), "dv", precision="fp8")
    lib = _hip.load()
    # widths that are not multiples of 16: MI_ESHAPE through the C ABI, reported as an exception by the binding
    xs = torch.randn(32, 24, device=dev)
    w = torch.randn(24, 24, device=dev)
    sid = torch.arange(32, device=dev)
    ws = _hip.workspace(lib.mi_bilinear_workspace_bytes(32, 32, 24, 24, _hip.MI_PREC_FP8), dev)
    loss, stats, rec = torch.zeros(1, device=dev), _hip.new_stats(dev), torch.zeros(_hip.RECORD_FLOATS, device=dev)
    with pytest.raises((ValueError, _hip.MiCriticError)):
        _hip.call("mi_bilinear_fwd", dev, xs.data_ptr(), xs.data_ptr(), w.data_ptr(), sid.data_ptr(), sid.data_ptr(), 32, 32, 0,
                  24, 24, 0, _hip.MI_PREC_FP8, 1, loss.data_ptr(), stats.data_ptr(), rec.data_ptr(), None, ws.data_ptr(),
                  ws.numel())


@pytest.mark.gpu
@pytest.mark.parametrize("b,d,G", [(512, 256, 4), (1024, 1024, 8)])
def test_fp8_row_blocks_with_global_scales(dev, b, d, G):
    """BASELINE configs[4] is an 8-GPU configuration: a rank's row block must be quantised with the scales of the WHOLE
    batch.  One GPU plays all G ranks through the real kernels and the staged preparation (mi_bilinear_fp8_stage), the MAX
    all-reduce of the four absmax values done by hand; the result must match the single-GPU fp8 step (same quantised
    operands: equal up to fp32 summation order) and the quantised oracle at the stated fp8 tolerances."""
    from mutual_info_img_txt import _hip, mi_critics
    from mutual_info_img_txt.distributed import HipBilinearOps
    from mutual_info_img_txt.model import BilinearCritic
    gen = torch.Generator().manual_seed(b + d)
    x = torch.randn(b, d, generator=gen)
    x[b - 1] *= 3.0  # the image absmax lives in the last row block only
    y = torch.randn(b, d, generator=gen)
    w = torch.randn(d, d, generator=gen) * (0.3 / math.sqrt(d))
    sid = torch.tensor(_dup_ids(b))
    est, code, prec = "infonce", _hip.ESTIMATORS["infonce"], _hip.MI_PREC_FP8
    xd, yd, wd, sd = x.to(dev), y.to(dev), w.to(dev), sid.to(dev)
    br = b // G
    ops = [HipBilinearOps() for _ in range(G)]
    amax = [torch.zeros(4, device=dev) for _ in range(G)]
    blocks = [xd[g * br:(g + 1) * br].contiguous() for g in range(G)]
    for stage in (0, 1, 2):
        if stage:
            glob = torch.stack(amax).max(dim=0).values  # what dist.all_reduce(op=MAX) leaves on every rank
            for a in amax:
                a.copy_(glob)
        for g in range(G):
            ops[g].fp8_stage(stage, blocks[g], yd, [wd], amax[g])
    assert float(amax[0][0]) == float(x.abs().max())
    outs = [ops[g].forward(blocks[g], yd, [wd], sd[g * br:(g + 1) * br].contiguous(), sd, g * br, code, prec, True) for g in range(G)]
    loss, stats = ops[0].merge(torch.stack([o[0] for o in outs]), b, code)
    go = torch.ones(1, device=dev)
    gx, gy, gw = [], torch.zeros_like(yd), torch.zeros_like(wd)
    for g in range(G):
        a, c, (e,) = ops[g].backward(outs[g][1], stats, go)
        gx.append(a)
        gy += c
        gw += e
    gx = torch.cat(gx)
    # the single-GPU step on the same batch
    critic = BilinearCritic(d, d)
    with torch.no_grad():
        critic.weight.copy_(w)
    critic.to(dev)
    xl, yl = xd.clone().requires_grad_(True), yd.clone().requires_grad_(True)
    l1 = mi_critics.fused_mi_bound(xl, yl, sd, critic, est, precision="fp8")
    l1.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(l1)) < 1e-5 * max(1.0, abs(float(l1)))
    for got, ref in ((gx, xl.grad), (gy, yl.grad), (gw, critic.weight.grad)):
        assert float((got - ref).abs().max()) <= 2e-3 * float(ref.abs().max())  # bf16 roundings of G / dT per row block
    o = orc.bilinear_step_fp8(x, y, w, sid, est)
    sc = max(float(o["scores"].abs().max()), 1.0)
    assert abs(float(loss) - float(o["loss"].sum())) < 2e-3 * sc
    for name, got, ref in (("dx", gx, o["dx"]), ("dy", gy, o["dy"]), ("dw", gw, o["dw"])):
        err = float((got.cpu().double() - ref).abs().max()) / float(ref.abs().max())
        assert err < 2e-2, (name, err)
