"""CPU tests of the rows SURVEY.md 8f ranks 2-4: study-id contract, encoder -> critic interface, checkpoint formats,
dataset batches, GDV metric -- against fixtures generated from the reference (tests/golden/make_goldens_f.py)."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _golden_helpers():
    spec = importlib.util.spec_from_file_location("make_goldens_f", os.path.join(ROOT, "tests", "golden", "make_goldens_f.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ------------------------------------------------------------------------------------------------ study ids (f4, H5)
def test_mimic_id_golden(golden):
    from mutual_info_img_txt.utils import MimicID
    g = golden("g5_ids.npz")
    ids = [(10000032, 50414267, "02aa804e-bde0afdd-112c0b34-7bc16630-4e384014"), ("123", "456", "abc"), (7, 8, 9)]
    for t, printed, study in zip(ids, g["printed"], g["study"]):
        m = MimicID(*t)
        assert str(m) == str(printed)
        assert MimicID.get_study_id(str(m)) == str(study)
        assert (m.subject_id, m.study_id, m.dicom_id) == tuple(map(str, t))


def test_study_id_codes_are_process_independent():
    from mutual_info_img_txt.utils import study_id_to_int64, study_ids_to_tensor
    assert study_id_to_int64("50414267") == 50414267 == study_id_to_int64(50414267) == study_id_to_int64(np.int64(50414267))
    assert study_id_to_int64(torch.tensor(12)) == 12
    a, b = study_id_to_int64("s-17/x"), study_id_to_int64("s-17/y")
    assert a != b and a >= 1 << 62 and a == study_id_to_int64("s-17/x")     # hashed: stable, outside the numeric range
    assert study_id_to_int64("007") != study_id_to_int64("7")                 # different strings are different studies
    assert study_id_to_int64("0") == 0
    ids = ["50000003", "50000001", "50000003", "abc", "abc", "50000002"]
    codes = study_ids_to_tensor(ids)
    eq = codes[:, None] == codes[None, :]
    ref = torch.tensor([[x == y for y in ids] for x in ids])
    assert torch.equal(eq, ref)
    # the same ids in another order (another rank's shard) get the same codes: no first-seen numbering
    perm = [5, 3, 0, 1]
    assert torch.equal(study_ids_to_tensor([ids[k] for k in perm]), codes[perm])


def test_pair_masking_uses_deterministic_codes():
    from oracle import mi_oracle as orc
    ids = ["11", "x", "11", "y", "x"]
    from mutual_info_img_txt.utils import study_ids_to_tensor
    codes = study_ids_to_tensor(ids)
    i1, j1 = orc.pair_index(ids)
    i2, j2 = orc.pair_index(codes.tolist())
    assert np.array_equal(i1, i2) and np.array_equal(j1, j2)


# ------------------------------------------------------------------------------------------------ GDV (f4)
def test_gdv_golden(golden):
    sys.path.insert(0, os.path.join(ROOT, "mutual-information-multimodal_amd"))
    import validate
    from oracle import mi_oracle as orc
    g = golden("g4_gdv.npz")
    for tag in ("a", "b", "c"):
        npos, nneg, d = (int(v) for v in g[f"{tag}/shape"])
        pos = orc.hash_uniform((npos, d), 31).double() * 2.0
        neg = orc.hash_uniform((nneg, d), 32).double() * 3.0 + float(g[f"{tag}/shift"])
        got = validate.gdv_calculation(pos, neg)
        assert abs(got - float(g[f"{tag}/gdv"])) <= 1e-10 * max(1.0, abs(float(g[f"{tag}/gdv"])))


# ------------------------------------------------------------------------------------------------ encoders (f2)
def test_resnet_matches_reference_golden(golden):
    from mutual_info_img_txt.model import ResNet256_6_2_1
    from oracle import mi_oracle as orc
    g = golden("g6_resnet.npz")
    model = ResNet256_6_2_1(output_channels=3)
    names = list(model.state_dict().keys())
    assert names == [str(n) for n in g["names"]]                       # checkpoint compatibility: same keys, same order
    assert [";".join(map(str, model.state_dict()[n].shape)) for n in names] == [str(s) for s in g["shapes"]]
    model.load_state_dict(_golden_helpers().closed_form_state(model))
    model.eval()
    x = orc.hash_uniform((2, 1, 256, 256), 55) + 0.5
    with torch.no_grad():
        y, z, y_sig, z_local, y_logits = model(x)
    assert z.shape == (2, 768) and tuple(z_local.shape) == tuple(int(v) for v in g["z_local_shape"])
    for got, key in ((y, "y"), (z, "z"), (y_sig, "y_sigmoid"), (y_logits, "y_logits"), (z_local.sum(dim=(2, 3)), "z_local_sum")):
        ref = torch.from_numpy(g[key])
        assert float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), key


def _small_text(oc=3):
    sys.path.insert(0, os.path.join(ROOT, "mutual-information-multimodal_amd"))
    from multi_modal import _small_bert_config
    from mutual_info_img_txt.model import TextBert
    cfg = _small_bert_config(oc, vocab_size=50, hidden=32)
    torch.manual_seed(0)
    return TextBert(cfg), cfg


def test_textbert_contract():
    pytest.importorskip("transformers")
    text, cfg = _small_text()
    text.eval()
    ids = torch.randint(0, 50, (5, 12))
    out = text(ids, token_type_ids=torch.zeros_like(ids), attention_mask=torch.ones_like(ids))
    assert out[0].shape == (5, 32) and out[1].shape == (5, 3)
    # index 0 is the pooled [CLS] AFTER dropout and feeds the classifier (reference model.py:76-80)
    assert torch.allclose(text.classifier(out[0]), out[1], atol=1e-6)
    text.train()
    torch.manual_seed(1)
    o1 = text(ids)[0]
    text.eval()
    assert not torch.allclose(o1, text(ids)[0])     # dropout is applied to the embedding the critic sees
    text.freeze_bert_encoder()
    assert not any(p.requires_grad for p in text.bert.parameters()) and text.classifier.weight.requires_grad
    text.unfreeze_bert_encode()
    assert all(p.requires_grad for p in text.bert.parameters())


def test_image_report_model_and_checkpoint_files(tmp_path):
    pytest.importorskip("transformers")
    from mutual_info_img_txt import encoders
    from mutual_info_img_txt.model import ImageReportModel, ResNet256_6_2_1, build_resnet_model
    text, cfg = _small_text()
    torch.manual_seed(2)
    joint = ImageReportModel(text_model=text, bert_config=cfg, image_model=ResNet256_6_2_1(output_channels=3)).eval()
    img = torch.rand(2, 1, 256, 256)
    ids = torch.randint(0, 50, (2, 9))
    e_img, e_txt, l_img, l_txt = joint(img, ids, torch.ones_like(ids), torch.zeros_like(ids))
    assert e_img.shape == (2, 768) and e_txt.shape == (2, 32) and l_img.shape == (2, 3) and l_txt.shape == (2, 3)
    assert e_img.dtype == torch.float32 and e_img.is_contiguous()
    assert torch.equal(e_img, joint.image_model(img)[1])                 # index 1 of the image encoder's 5-tuple
    # the reference's file names
    d = str(tmp_path)
    assert os.path.basename(joint.save_image_model(d)) == "pytorch_MI_image_model.bin"
    assert os.path.basename(joint.save_text_model(d)) == "pytorch_MI_text_model.bin"
    assert os.path.basename(joint.save_pretrained(d, epoch=3)) == "pytorch_model_epoch3.bin"
    assert os.path.basename(joint.save_pretrained(d)) == "pytorch_model.bin"
    assert os.path.isfile(os.path.join(d, "config.json"))
    keys = torch.load(os.path.join(d, "pytorch_model_epoch3.bin"), weights_only=True).keys()
    assert any(k.startswith("image_model.layer6.") for k in keys) and any(k.startswith("text_model.bert.") for k in keys)
    # image encoder back out of the JOINT checkpoint: prefix stripped, the joint model's fc head dropped
    enc, info = ResNet256_6_2_1.from_pretrained(os.path.join(d, "pytorch_model_epoch3.bin"), output_channels=3,
                                                loading_from_joint=True, output_loading_info=True)
    assert sorted(info["missing_keys"]) == ["fc1.bias", "fc1.weight"]
    assert all(k.startswith("text_model.") for k in info["unexpected_keys"])
    assert torch.equal(enc.layer6[1].conv2.weight, joint.image_model.layer6[1].conv2.weight)
    # ... and out of the image-only file, through the reference's builder, with the encoder frozen except layer6 / fc
    enc2 = build_resnet_model("resnet256_6_2_1", checkpoint_path=os.path.join(d, "pytorch_MI_image_model.bin"),
                              output_channels=3, freeze_encoder=True)
    assert torch.equal(enc2.fc1.weight, joint.image_model.fc1.weight)
    assert not enc2.layer1[0].conv1.weight.requires_grad and enc2.layer6[0].conv1.weight.requires_grad
    with pytest.raises(ValueError):
        build_resnet_model("resnet18")
    # legacy key names
    legacy = {"bn1.gamma": torch.ones(8), "bn1.beta": torch.zeros(8), "conv1.weight": torch.zeros(8, 1, 3, 3)}
    assert sorted(encoders.convert_legacy_keys(legacy)) == ["bn1.bias", "bn1.weight", "conv1.weight"]


def test_projection_heads_and_widths():
    pytest.importorskip("transformers")
    from mutual_info_img_txt.main_utils import MultiModalManager
    from mutual_info_img_txt.model import ResNet256_6_2_1
    text, cfg = _small_text(1)
    mgr = MultiModalManager(output_channels=1, image_model=ResNet256_6_2_1(output_channels=1), text_model=text,
                            bert_config=cfg, embed_proj_dim=256, critic="bilinear")
    assert mgr.model is not None and (mgr.d_img, mgr.d_txt) == (256, 256)
    assert tuple(mgr.mi_discriminator.weight.shape) == (256, 256)
    e_img, e_txt, _, _ = mgr.model.eval()(torch.rand(2, 1, 256, 256), torch.randint(0, 50, (2, 7)))
    assert e_img.shape == (2, 256) and e_txt.shape == (2, 256)
    # the reference's constructor keywords are accepted by name (main_utils.py:58-59)
    m2 = MultiModalManager(bert_pretrained_dir=None, bert_config_name=None, output_channels=4, image_model_name=None)
    assert [tuple(p.shape) for p in m2.mi_discriminator.parameters()][0] == (1024, 1536)   # make_mlp(1536, [1024, 512])


def test_training_state_roundtrip(tmp_path):
    from mutual_info_img_txt.main_utils import MultiModalManager
    torch.manual_seed(3)
    a = MultiModalManager(d_img=8, d_txt=8, critic="concat_mlp", hidden_dims=(16, 8))
    opt = torch.optim.Adam(a.mi_discriminator.parameters(), lr=1e-3)
    for p in a.mi_discriminator.parameters():
        p.grad = torch.randn_like(p)
    opt.step()
    a.training_loss = [1.5, -0.25]
    path = a.save_training_state(str(tmp_path), 2, opt)
    assert os.path.basename(path) == "mi_critic_state.pt"
    b = MultiModalManager(d_img=8, d_txt=8, critic="concat_mlp", hidden_dims=(16, 8))
    opt_b = torch.optim.Adam(b.mi_discriminator.parameters(), lr=1e-3)
    assert b.load_training_state(str(tmp_path), opt_b) == 2
    for p, q in zip(a.mi_discriminator.parameters(), b.mi_discriminator.parameters()):
        assert torch.equal(p, q)
    assert b.training_loss == [1.5, -0.25]
    sa, sb = opt.state_dict()["state"], opt_b.state_dict()["state"]
    assert all(torch.equal(sa[k]["exp_avg"], sb[k]["exp_avg"]) for k in sa)


# ------------------------------------------------------------------------------------------------ dataset contract (f4)
class _Feat:
    def __init__(self, report_id, n):
        self.report_id, self.input_ids, self.input_mask, self.segment_ids = report_id, [n] * 6, [1] * 6, [0] * 6


def test_dataset_batches_and_default_substitution():
    import pandas as pd
    from mutual_info_img_txt.model_utils import CXRImageReportDataset
    meta = pd.DataFrame({"mimic_id": ["p1_s50000001_a", "p1_s50000001_b", "p2_s50000002_c", "p3_s50000003_d"]})
    feats = [_Feat("50000001", 11), _Feat("50000002", 22)]           # no tokens for study 50000003
    images = {"p1_s50000001_a": np.full((4, 4), 2.0), "p1_s50000001_b": None, "p2_s50000002_c": np.full((4, 4), 8.0)}
    ds = CXRImageReportDataset(feats, "", meta, image_loader=lambda path: images.get(os.path.basename(path)),
                               transform=lambda im: im / 2)
    ds.set_default(np.zeros((1, 4, 4), np.float32), torch.zeros(6, dtype=torch.long), torch.ones(6, dtype=torch.long),
                   torch.zeros(6, dtype=torch.long))
    img, txt, masks, segs, study, img_id = ds[0]
    assert img.shape == (1, 4, 4) and float(img[0, 0, 0]) == 1.0 and study == "50000001" and img_id == "p1_s50000001_a"
    assert txt.tolist() == [11] * 6 and txt.dtype == torch.long
    img1, _, _, _, study1, _ = ds[1]                                   # unreadable image -> default image, own study id
    assert float(np.abs(img1).max()) == 0.0 and study1 == "50000001"
    img3, txt3, _, _, study3, _ = ds[3]                                # no tokens -> the whole default sample, own id
    assert txt3.tolist() == [0] * 6 and study3 == "50000003"
    assert len(ds) == 4
    # the two images of study 50000001 must not become each other's negatives
    from oracle import mi_oracle as orc
    i, j = orc.pair_index([ds[k][4] for k in range(4)])
    assert (0, 1) not in set(zip(i[4:].tolist(), j[4:].tolist()))
