"""the reference's trainer loop (main_utils.py:112-268) around the fused MI step.

CPU part: the restated pytorch-transformers 1.0.0 AdamW / WarmupLinearSchedule against the oracle's numpy restatement.
GPU part: the three-optimiser loop with small stand-in encoders through the HIP path."""
import math
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import mi_oracle as orc


def _optim():
    from mutual_info_img_txt import optimization
    return optimization


@pytest.mark.parametrize("correct_bias", [False, True])
def test_adamw_matches_restatement(correct_bias):
    opt_mod = _optim()
    gen = torch.Generator().manual_seed(0)
    p_dec = torch.randn(7, 5, generator=gen, dtype=torch.float64).requires_grad_(True)
    p_nod = torch.randn(5, generator=gen, dtype=torch.float64).requires_grad_(True)
    opt = opt_mod.AdamW([{"params": [p_dec], "weight_decay": 0.1}, {"params": [p_nod], "weight_decay": 0.0}], lr=2e-3,
                        correct_bias=correct_bias)
    ref = [[p_dec.detach().numpy().copy(), np.zeros((7, 5)), np.zeros((7, 5)), 0.1],
           [p_nod.detach().numpy().copy(), np.zeros(5), np.zeros(5), 0.0]]
    for t in range(1, 7):
        grads = [torch.randn(7, 5, generator=gen, dtype=torch.float64), torch.randn(5, generator=gen, dtype=torch.float64)]
        p_dec.grad, p_nod.grad = grads[0].clone(), grads[1].clone()
        opt.step()
        for k in range(2):
            ref[k][0], ref[k][1], ref[k][2] = orc.adamw_step(ref[k][0], grads[k].numpy(), ref[k][1], ref[k][2], t, 2e-3,
                                                             weight_decay=ref[k][3], correct_bias=correct_bias)
        np.testing.assert_allclose(p_dec.detach().numpy(), ref[0][0], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(p_nod.detach().numpy(), ref[1][0], rtol=1e-12, atol=1e-14)
    # the package's defaults, which differ from torch.optim.AdamW's
    d = opt_mod.AdamW([torch.zeros(1, requires_grad=True)]).defaults
    assert d["eps"] == 1e-6 and d["betas"] == (0.9, 0.999) and d["weight_decay"] == 0.0 and d["correct_bias"] is True


def test_warmup_linear_schedule():
    opt_mod = _optim()
    p = torch.zeros(3, requires_grad=True)
    opt = opt_mod.AdamW([p], lr=2e-5, correct_bias=False)
    total = 40
    sched = opt_mod.WarmupLinearSchedule(opt, warmup_steps=0.1 * total, t_total=total)
    lrs = []
    for step in range(total + 3):
        lrs.append(opt.param_groups[0]["lr"])
        p.grad = torch.ones(3)
        opt.step()
        sched.step()
    want = [2e-5 * orc.warmup_linear(s, 0.1 * total, total) for s in range(total + 3)]
    np.testing.assert_allclose(lrs, want, rtol=1e-12, atol=0)
    assert lrs[0] == 0.0 and max(lrs) == pytest.approx(2e-5) and lrs[-1] == 0.0


class _TextEncoder(torch.nn.Module):
    """Stand-in with the parameter names the reference's no-decay filter looks for (bias, LayerNorm.*)."""

    def __init__(self, d_in, d_out):
        super().__init__()
        self.dense = torch.nn.Linear(d_in, d_out)
        self.LayerNorm = torch.nn.LayerNorm(d_out)

    def forward(self, t):
        return self.LayerNorm(self.dense(t))


@pytest.mark.gpu
def test_three_optimizer_loop(tmp_path):
    """Reference loop order and optimisers with stand-in encoders; the MI bound rises (the loss falls) on correlated
    inputs, every module receives updates, and the text optimiser follows the warm-up-linear schedule."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mutual_info_img_txt.main_utils import MultiModalManager
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    b, d_in, d = 64, 24, 32
    img_enc = torch.nn.Sequential(torch.nn.Linear(d_in, d), torch.nn.Tanh())
    txt_enc = _TextEncoder(d_in, d)
    mgr = MultiModalManager(d_img=d, d_txt=d, critic="concat_mlp", hidden_dims=(128, 256), image_model=img_enc,
                            text_model=txt_enc)
    gen = torch.Generator().manual_seed(1)
    base = torch.randn(b, d_in, generator=gen)

    def source(step):
        noise = torch.randn(2, b, d_in, generator=gen) * 0.3
        return (base + noise[0]).to(dev), (base + noise[1]).to(dev), [str(n) for n in range(b)]

    args = types.SimpleNamespace(mi_estimator="dv", init_lr=1e-3, num_train_epochs=4, steps_per_epoch=10, precision="f32",
                                 txt_lr=2e-3)
    before = [p.detach().clone() for m in (img_enc, txt_enc, mgr.mi_discriminator) for p in m.parameters()]
    losses = mgr.train(source, dev, args)
    after = [p.detach().cpu() for m in (mgr.image_model, mgr.text_model, mgr.mi_discriminator) for p in m.parameters()]
    assert len(losses) == 4 and all(math.isfinite(v) for v in losses) and losses[-1] < losses[0]
    assert all(not torch.equal(a, c) for a, c in zip(before, after))  # all three optimisers stepped
