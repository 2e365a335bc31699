"""SURVEY.md 8f rows on the GPU: the reference-generated ResNet golden (g6) on the device, fp32 and
under the bf16 autocast policy; a 5-step trajectory of `MultiModalManager.train` against oracle step + torch.optim.Adam
on the CPU; the default-sample hand-off through the real loop; GraphedMiStep's one-forward-at-a-time guard.

Stated tolerances: ResNet fp32 on the device 1e-4 * max(1, |ref|max) (MIOpen convolutions sum in another order than the
CPU's); bf16 autocast 4e-2 * max(1, |ref|max) on z / logits (12 convolution layers of bf16 operands); trajectory: step
losses 2e-4 relative, parameter displacement after 5 Adam steps 2e-2 relative L2 (Adam's first steps are sign-like:
an element whose gradient sits inside the fp32 noise moves by +-lr either way)."""
import importlib.util
import os
import sys
import types

import numpy as np
import pandas as pd
import pytest
import torch

from oracle import mi_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    from mutual_info_img_txt import _hip
    _hip.load()
    return torch.device("cuda:0")


def _golden_helpers():
    spec = importlib.util.spec_from_file_location("make_goldens_f", os.path.join(ROOT, "tests", "golden", "make_goldens_f.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ------------------------------------------------------------------------------------------------ f2: encoder golden
@pytest.mark.parametrize("autocast", [None, torch.bfloat16])
def test_resnet_reference_golden_on_device(dev, golden, autocast):
    """g6 = outputs of the REFERENCE's ResNet256_6_2_1 class on closed-form parameters (tests/golden/make_goldens_f.py);
    the same parameters and input through this repo's class on the MI355X."""
    from mutual_info_img_txt.model import ResNet256_6_2_1
    g = golden("g6_resnet.npz")
    model = ResNet256_6_2_1(output_channels=3)
    model.load_state_dict(_golden_helpers().closed_form_state(model))
    model = model.to(dev).eval()
    x = (orc.hash_uniform((2, 1, 256, 256), 55) + 0.5).to(dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=autocast or torch.bfloat16, enabled=autocast is not None):
        y, z, y_sig, z_local, y_logits = model(x)
    tol = 1e-4 if autocast is None else 4e-2
    worst = {}
    for got, key in ((y, "y"), (z, "z"), (y_sig, "y_sigmoid"), (y_logits, "y_logits"), (z_local.float().sum(dim=(2, 3)), "z_local_sum")):
        ref = torch.from_numpy(g[key])
        worst[key] = float((got.float().cpu() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    print(f"ResNet g6 on device, autocast={autocast}: worst error / max(1,|ref|max):", {k: f"{v:.2e}" for k, v in worst.items()})
    assert z.shape == (2, 768)
    for key, err in worst.items():
        assert err <= tol, (key, err)


# ------------------------------------------------------------------------------------------------ f1: trajectory
@pytest.mark.parametrize("kind", ["concat_mlp", "bilinear"])
def test_five_step_trajectory_vs_oracle_and_adam(dev, kind, tmp_path):
    """`MultiModalManager.train` (critic only, f32, graph replay as the trainer's default) for 5 steps against the same 5
    steps on the CPU: oracle forward/backward in fp64 + torch.optim.Adam (the reference's mi_optimizer,
    main_utils.py:153)."""
    from mutual_info_img_txt.main_utils import MultiModalManager
    b, d, steps, lr = 64, 32, 5, 1e-3
    torch.manual_seed(0)
    mgr = MultiModalManager(d_img=d, d_txt=d, critic=kind, hidden_dims=(128, 256))
    init = [p.detach().clone().double() for p in mgr.mi_discriminator.parameters()]
    gen = torch.Generator().manual_seed(9)
    base = torch.randn(b, d, generator=gen)
    data = [((base + 0.4 * torch.randn(b, d, generator=gen)), (base + 0.4 * torch.randn(b, d, generator=gen))) for _ in range(steps)]
    sid = [str(50000000 + k - (k % 2 if k < 8 else 0)) for k in range(b)]  # SURVEY 8d duplicates

    def source(step):  # one step per epoch (the epoch losses ARE the step losses, main_utils.py:233,241): count calls
        k = source.n
        source.n += 1
        return data[k][0].to(dev), data[k][1].to(dev), sid
    source.n = 0
    args = types.SimpleNamespace(mi_estimator="dv", init_lr=lr, num_train_epochs=steps, steps_per_epoch=1, precision="f32",
                                 graph=True, save_directory=None)
    got_losses = mgr.train(source, dev, args)
    got = [p.detach().cpu().double() for p in mgr.mi_discriminator.parameters()]
    # CPU: fp64 oracle + Adam
    params = [p.clone().requires_grad_(True) for p in init]
    opt = torch.optim.Adam(params, lr=lr)
    want_losses = []
    for x, y in data:
        opt.zero_grad()
        if kind == "bilinear":
            s = orc.bilinear_scores(x.double(), y.double(), params[0])
        else:
            s = orc.concat_scores_matrix(x.double(), y.double(), params)
        loss = orc.bound_from_matrix(s, sid, "dv")
        loss.sum().backward()
        opt.step()
        want_losses.append(float(loss.sum()))
    print(f"{kind}: step losses {got_losses} vs {want_losses}")
    np.testing.assert_allclose(got_losses, want_losses, rtol=2e-4, atol=2e-5)
    num = sum(float(((g - w.detach()) ** 2).sum()) for g, w in zip(got, params))
    den = sum(float(((w.detach() - i) ** 2).sum()) for w, i in zip(params, init))
    print(f"{kind}: parameter displacement error {np.sqrt(num / den):.3e} (relative L2)")
    assert np.sqrt(num / den) < 2e-2


# ------------------------------------------------------------------------------------------------ f4: default sample
def test_default_sample_substitution_through_train(dev, tmp_path):
    """reference main_utils.py:195-199: the first batch's sample 0 becomes the dataset's default; a row whose image is
    unreadable then reaches the encoders as the default image WITH ITS OWN study id (model_utils.py:162-219)."""
    pytest.importorskip("transformers")
    import collections
    sys.path.insert(0, os.path.join(ROOT, "mutual-information-multimodal_amd"))
    from multi_modal import _small_bert_config
    from mutual_info_img_txt.main_utils import MultiModalManager
    from mutual_info_img_txt.model import ResNet256_6_2_1, TextBert
    Features = collections.namedtuple("Features", "report_id input_ids input_mask segment_ids")
    n, seq, size = 16, 8, 256
    gen = torch.Generator().manual_seed(3)
    rows, feats, images = [], [], {}
    for k in range(n):
        mimic_id = f"p{10000000 + k}_s{50000000 + k}_d{k:04d}"
        rows.append(mimic_id)
        feats.append(Features(str(50000000 + k), torch.randint(4, 60, (seq,), generator=gen).tolist(), [1] * seq, [0] * seq))
        images[mimic_id] = torch.rand(size, size, generator=gen).numpy().astype(np.float32) * (1 + k)
    broken = rows[11]
    state = {"armed": False}

    def image_loader(path):
        key = os.path.basename(path)
        return None if (state["armed"] and key == broken) else images[key]
    torch.manual_seed(0)
    cfg = _small_bert_config(2)
    mgr = MultiModalManager(output_channels=2, image_model=ResNet256_6_2_1(output_channels=2), text_model=TextBert(cfg),
                            bert_config=cfg, critic="bilinear", embed_proj_dim=128)
    args = types.SimpleNamespace(mi_estimator="infonce", init_lr=1e-4, num_train_epochs=1, batch_size=8, img_size=256,
                                 image_dir=str(tmp_path), dataset_metadata=pd.DataFrame({"mimic_id": rows}),
                                 image_loader=image_loader, data_loader_workers=0, save_directory=None, graph=True,
                                 precision="bf16")
    seen = []
    orig = mgr._embed

    def spy(batch, device):
        seen.append(batch)
        return orig(batch, device)
    mgr._embed = spy
    mgr.train(feats, dev, args)            # epoch 1: everything readable; sample 0 of the first batch becomes the default
    ds = mgr._loader.dataset
    assert mgr._default_set and torch.equal(ds.default_img, seen[0][0][0])
    assert mgr.scheduler.t_total == 2       # 16 samples / batch 8, one epoch: len(data_loader) steps (main_utils.py:168)
    state["armed"] = True
    del seen[:]
    losses = mgr.train(feats, dev, args)    # epoch 2: row 11's image is unreadable
    assert len(losses) == 1 and np.isfinite(losses[0])
    hit = 0
    for img, ids, masks, segs, study, img_id in seen:
        for r, name in enumerate(img_id):
            if name == broken:
                hit += 1
                assert study[r] == "50000011"                                # its own study id
                assert torch.equal(img[r], ds.default_img)                    # the default image
                assert torch.equal(ids[r], torch.as_tensor(feats[11].input_ids))  # its own report
    assert hit == 1


# ------------------------------------------------------------------------------------------------ graphed step guard
def test_graphed_step_refuses_a_backward_of_a_stale_forward(dev):
    from mutual_info_img_txt.graphed import GraphedMiStep
    from mutual_info_img_txt.model import BilinearCritic
    torch.manual_seed(0)
    critic = BilinearCritic(64, 64).to(dev)
    step = GraphedMiStep(critic, 64, 64, 64, "infonce", "f32", dev)
    x1, y1 = torch.randn(64, 64, device=dev, requires_grad=True), torch.randn(64, 64, device=dev)
    x2, y2 = torch.randn(64, 64, device=dev, requires_grad=True), torch.randn(64, 64, device=dev)
    l1 = step.loss(x1, y1)
    l2 = step.loss(x2, y2)          # overwrites the static inputs / workspace of l1
    with pytest.raises(RuntimeError, match="no longer the step's latest"):
        l1.backward()
    l2.backward()                   # the latest forward still differentiates
    assert x2.grad is not None and torch.isfinite(x2.grad).all() and x1.grad is None
    # the normal order keeps working
    l3 = step.loss(x1, y1)
    l3.backward()
    assert torch.isfinite(x1.grad).all()
