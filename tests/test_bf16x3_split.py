"""The arithmetic behind precision="bf16x3" (csrc/mi_gemm_bf16.h, split3_offsets), restated in torch on the CPU:
v = hi + lo + O(2^-17 |v|) with hi = bf16(v), lo = bf16(v - hi); a tripled K with the A side laid out [hi | hi | lo]
and the B side [hi | lo | hi] turns ONE bf16 GEMM into hi.hi + hi.lo + lo.hi.  Checks the layout identity exactly and
the accuracy the mode is held to on the GPU (tests/test_parity_configs.py)."""
import torch


def split(v):
    hi = v.to(torch.bfloat16).float()
    lo = (v - hi).to(torch.bfloat16).float()
    return hi, lo


def tripled(v, role):
    hi, lo = split(v)
    return torch.cat([hi, hi, lo], 1) if role == 1 else torch.cat([hi, lo, hi], 1)


def test_tripled_k_is_the_three_term_product():
    gen = torch.Generator().manual_seed(0)
    a = torch.randn(48, 80, generator=gen)
    b = torch.randn(40, 80, generator=gen)
    ah, al = split(a)
    bh, bl = split(b)
    want = ah.double() @ bh.double().t() + ah.double() @ bl.double().t() + al.double() @ bh.double().t()
    got = tripled(a, 1).double() @ tripled(b, 2).double().t()
    assert torch.equal(got, want)
    # every stored value is a bf16 value, so the bf16 MFMA multiplies it exactly
    for t in (tripled(a, 1), tripled(b, 2)):
        assert torch.equal(t, t.to(torch.bfloat16).float())


def test_accuracy_of_the_split_product():
    """What is dropped is lo.lo and the residuals of the two-part representation: ~2^-16 per product against 2^-8 for
    plain bf16 operands."""
    gen = torch.Generator().manual_seed(1)
    a = torch.randn(64, 512, generator=gen)
    b = torch.randn(64, 512, generator=gen)
    exact = a.double() @ b.double().t()
    x3 = tripled(a, 1).double() @ tripled(b, 2).double().t()
    bf = a.to(torch.bfloat16).double() @ b.to(torch.bfloat16).double().t()
    scale = float(exact.abs().max())
    e3, e1 = float((x3 - exact).abs().max()) / scale, float((bf - exact).abs().max()) / scale
    assert e3 < 2e-5 and e1 > 50 * e3, (e3, e1)
    hi, lo = split(a)
    assert float((a - hi - lo).abs().max()) <= 2.0 ** -16 * float(a.abs().max())
