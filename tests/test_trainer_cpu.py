"""CPU tests of the trainer's host logic: schedule length on the reference's own input path,
default-sample hand-off, resume with encoders, BERT checkpoint loading, GraphedMiStep's one-forward-at-a-time guard, and
bench.py's rank launcher.

`MultiModalManager.train` runs unchanged; only its `mi_step` (the HIP critic, which has no CPU path by design) is
replaced by the oracle's differentiable restatement of the same loss -- the checker standing in for the kernel so that
the LOOP around it (reference main_utils.py:112-268) can run without a GPU."""
import collections
import json
import os
import subprocess
import sys
import types

import numpy as np
import pandas as pd
import pytest
import torch

from oracle import mi_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Features = collections.namedtuple("Features", "report_id input_ids input_mask segment_ids")


def _small_bert_config(hidden=32, vocab=48):
    from transformers import BertConfig
    cfg = BertConfig(vocab_size=vocab, hidden_size=hidden, num_hidden_layers=1, num_attention_heads=2,
                     intermediate_size=2 * hidden, max_position_embeddings=32)
    cfg.num_classes = 2
    return cfg


def _oracle_mi_step(self, embedding_img, embedding_txt, study_id, mi_estimator="dv", precision="f32", fused=True, graph=False):
    params = [p for p in self.mi_discriminator.parameters()]
    scores = orc.concat_scores_matrix(embedding_img, embedding_txt, params)
    return orc.bound_from_matrix(scores, list(study_id), mi_estimator)


def _tiny_dataset(n=10, seq=8, size=256, missing_image=None, vocab=48):
    gen = torch.Generator().manual_seed(5)
    rows, feats, images = [], [], {}
    for k in range(n):
        study = 50000000 + k
        mimic_id = f"p{10000000 + k}_s{study}_d{k:04d}"
        rows.append(mimic_id)
        feats.append(Features(str(study), torch.randint(4, vocab, (seq,), generator=gen).tolist(), [1] * seq, [0] * seq))
        images[mimic_id] = (torch.rand(size, size, generator=gen).numpy() * (k + 1)).astype(np.float32)
    meta = pd.DataFrame({"mimic_id": rows})

    def image_loader(path):
        key = os.path.basename(path)
        if key == missing_image:
            return None
        return images[key]
    return feats, meta, image_loader, images


def _manager(critic_hidden=(16, 8)):
    from mutual_info_img_txt.main_utils import MultiModalManager
    from mutual_info_img_txt.model import ResNet256_6_2_1, TextBert
    torch.manual_seed(0)
    cfg = _small_bert_config()
    mgr = MultiModalManager(output_channels=2, image_model=ResNet256_6_2_1(output_channels=2), text_model=TextBert(cfg),
                            bert_config=cfg, critic="concat_mlp", hidden_dims=critic_hidden, embed_proj_dim=8)
    mgr.mi_step = types.MethodType(_oracle_mi_step, mgr)
    return mgr


def _args(tmp_path, meta, image_loader, epochs=2, batch=4, **kw):
    return types.SimpleNamespace(mi_estimator="dv", init_lr=1e-3, num_train_epochs=epochs, batch_size=batch, img_size=256,
                                 image_dir=str(tmp_path), dataset_metadata=meta, image_loader=image_loader,
                                 data_loader_workers=0, save_directory=str(tmp_path / "out"), graph=False, precision="f32",
                                 steps_per_epoch=20, **kw)


def test_schedule_length_follows_the_loader_on_the_reference_input_path(tmp_path):
    """reference main_utils.py:168: t_total = num_train_epochs * len(data_loader) -- NOT args.steps_per_epoch (20 here),
    which only drives callable sources.  10 samples, batch 4, drop_last -> 2 steps per epoch."""
    feats, meta, loader, _ = _tiny_dataset()
    mgr = _manager()
    args = _args(tmp_path, meta, loader, epochs=3)
    losses = mgr.train(feats, torch.device("cpu"), args)
    assert len(losses) == 3
    assert len(mgr._loader) == 2
    assert mgr.scheduler.t_total == 3 * 2 and mgr.scheduler.warmup_steps == pytest.approx(0.6)
    assert mgr.scheduler.last_epoch == 6  # one scheduler step per batch
    # at t_total the multiplier is exactly 0 and was positive before (the run did not outlive its schedule)
    assert mgr.scheduler.get_last_lr()[0] == 0.0


def test_first_batch_sample_becomes_the_default_and_replaces_an_unreadable_image(tmp_path):
    """reference main_utils.py:195-199 + model_utils.py:162-219: sample 0 of the first batch is handed to the dataset;
    a row whose image cannot be read then yields the default image with ITS OWN study id (the masking stays right)."""
    feats, meta, loader, images = _tiny_dataset(missing_image=None)
    mgr = _manager()
    args = _args(tmp_path, meta, loader, epochs=1)
    seen = []
    inner = mgr.mi_step

    def spy(img, txt, sid, *a, **k):
        seen.append(list(sid))
        return inner(img, txt, sid, *a, **k)
    mgr.mi_step = spy
    mgr.train(feats, torch.device("cpu"), args)
    ds = mgr._loader.dataset
    assert mgr._default_set and ds.default_img is not None and tuple(ds.default_img.shape) == (1, 256, 256)
    assert ds.default_tokens is not None and ds.default_tokens.dtype == torch.long
    # now break one image: the dataset substitutes the default image and keeps the row's own study id / image id
    broken = meta["mimic_id"][7]
    ds.image_loader = lambda path: None if os.path.basename(path) == broken else images[os.path.basename(path)]
    img, txt, masks, segs, study, img_id = ds[7]
    assert torch.equal(torch.as_tensor(img), torch.as_tensor(ds.default_img))
    assert study == "50000007" and img_id == broken
    assert torch.equal(txt, torch.as_tensor(feats[7].input_ids))  # its own report: only the image was missing
    # and a whole epoch over the broken dataset collates and trains (before the hand-off this crashed default_collate)
    mgr.train(feats, torch.device("cpu"), args)
    assert all(len(s) == 4 for s in seen)


def test_resume_restores_encoder_weights_and_schedule(tmp_path):
    """a resumed run must pair the restored optimiser moments with the encoder weights of the same epoch."""
    feats, meta, loader, _ = _tiny_dataset()
    torch.manual_seed(1)
    a = _manager()
    args = _args(tmp_path, meta, loader, epochs=2)
    a.train(feats, torch.device("cpu"), args)
    saved = {k: v.clone() for k, v in a.model.state_dict().items()}
    # a fresh manager (fresh random encoders) resumes from the files of epoch 2 and runs epoch 3
    torch.manual_seed(2)
    b = _manager()
    assert any(not torch.equal(v, saved[k]) for k, v in b.model.state_dict().items() if v.dtype.is_floating_point)
    args3 = _args(tmp_path, meta, loader, epochs=3, resume_from=str(tmp_path / "out"))
    captured = {}
    orig = b.load_training_state

    def spy(*aa, **kk):
        n = orig(*aa, **kk)
        captured["state"] = {k: v.clone() for k, v in b.model.state_dict().items()}
        captured["lr"] = [g["lr"] for g in aa[3].param_groups]
        captured["last_epoch"] = aa[4].last_epoch
        return n
    b.load_training_state = spy
    losses = b.train(feats, torch.device("cpu"), args3)
    assert len(losses) == 3 and losses[:2] == a.training_loss[:2]
    for k, v in saved.items():
        assert torch.equal(captured["state"][k], v), k
    assert captured["last_epoch"] == 4  # 2 epochs x 2 steps
    # the first resumed step runs at the learning rate of step 4 of a 6-step schedule
    from mutual_info_img_txt.optimization import warmup_linear_factor
    assert captured["lr"][0] == pytest.approx(2e-5 * warmup_linear_factor(4, 0.6, 6))
    # encoders attached but the joint checkpoint of that epoch is gone -> refuse
    os.remove(tmp_path / "out" / "pytorch_model_epoch3.bin")
    c = _manager()
    with pytest.raises(FileNotFoundError, match="joint checkpoint"):
        c.train(feats, torch.device("cpu"), _args(tmp_path, meta, loader, epochs=4, resume_from=str(tmp_path / "out")))


def test_resume_with_plain_encoder_modules_keeps_their_weights_in_the_state_file(tmp_path):
    from mutual_info_img_txt.main_utils import MultiModalManager
    torch.manual_seed(3)

    def build():
        m = MultiModalManager(d_img=6, d_txt=6, critic="concat_mlp", hidden_dims=(16, 8),
                              image_model=torch.nn.Linear(5, 6), text_model=torch.nn.Linear(5, 6))
        m.mi_step = types.MethodType(_oracle_mi_step, m)
        return m
    gen = torch.Generator().manual_seed(0)
    data = [(torch.randn(8, 5, generator=gen), torch.randn(8, 5, generator=gen), [str(k) for k in range(8)]) for _ in range(3)]
    args = types.SimpleNamespace(mi_estimator="infonce", init_lr=1e-2, num_train_epochs=1, save_directory=str(tmp_path),
                                 graph=False, precision="f32")
    a = build()
    a.train(data, torch.device("cpu"), args)
    b = build()
    args.num_train_epochs, args.resume_from = 1, str(tmp_path)
    b.train(data, torch.device("cpu"), args)  # nothing left to run: loads and returns
    for pa, pb in zip(a.image_model.parameters(), b.image_model.parameters()):
        assert torch.equal(pa, pb)
    for pa, pb in zip(a.text_model.parameters(), b.text_model.parameters()):
        assert torch.equal(pa, pb)


def test_textbert_from_pretrained_is_never_silently_random(tmp_path):
    from mutual_info_img_txt.model import TextBert
    cfg = _small_bert_config()
    with pytest.raises(FileNotFoundError):
        TextBert.from_pretrained(str(tmp_path), cfg)
    torch.manual_seed(4)
    src = TextBert(cfg)
    # (a) the wrapper's own keys ("bert.*", "classifier.*")
    d1 = tmp_path / "prefixed"
    d1.mkdir()
    torch.save(src.state_dict(), d1 / "pytorch_model.bin")
    torch.manual_seed(5)
    got = TextBert.from_pretrained(str(d1), cfg)
    for k, v in src.state_dict().items():
        assert torch.equal(got.state_dict()[k], v), k
    # (b) a bare BertModel checkpoint (no prefix) goes into model.bert, as BertPreTrainedModel.from_pretrained does
    d2 = tmp_path / "bare"
    d2.mkdir()
    torch.save(src.bert.state_dict(), d2 / "pytorch_model.bin")
    torch.manual_seed(6)
    got = TextBert.from_pretrained(str(d2), cfg)
    for k, v in src.bert.state_dict().items():
        assert torch.equal(got.bert.state_dict()[k], v), k
    # (c) legacy gamma/beta names (reference model.py:441-455) still load
    d3 = tmp_path / "legacy"
    d3.mkdir()
    legacy = {k.replace("LayerNorm.weight", "LayerNorm.gamma").replace("LayerNorm.bias", "LayerNorm.beta"): v
              for k, v in src.state_dict().items()}
    torch.save(legacy, d3 / "pytorch_model.bin")
    got = TextBert.from_pretrained(str(d3), cfg)
    assert torch.equal(got.bert.embeddings.LayerNorm.weight, src.bert.embeddings.LayerNorm.weight)
    # (d) a file with no BERT parameter at all is an error, not a random encoder
    d4 = tmp_path / "other"
    d4.mkdir()
    torch.save({"conv1.weight": torch.zeros(3)}, d4 / "pytorch_model.bin")
    with pytest.raises(ValueError, match="no parameter of the BERT encoder"):
        TextBert.from_pretrained(str(d4), cfg)


def test_warmup_linear_schedule_matches_transformers():
    """The package the reference pins (pytorch-transformers 1.0.0) is absent; its successor `transformers` ships the same
    schedule as get_linear_schedule_with_warmup for integer warm-up lengths: pin the restatement against it."""
    transformers = pytest.importorskip("transformers")
    from mutual_info_img_txt.optimization import AdamW, WarmupLinearSchedule
    for total, warm in ((40, 4), (100, 10), (7, 0), (12, 3)):
        p1, p2 = torch.zeros(2, requires_grad=True), torch.zeros(2, requires_grad=True)
        o1 = AdamW([p1], lr=2e-5, correct_bias=False)
        o2 = torch.optim.SGD([p2], lr=2e-5)
        s1 = WarmupLinearSchedule(o1, warmup_steps=warm, t_total=total)
        s2 = transformers.get_linear_schedule_with_warmup(o2, num_warmup_steps=warm, num_training_steps=total)
        for _ in range(total + 2):
            assert o1.param_groups[0]["lr"] == pytest.approx(o2.param_groups[0]["lr"], rel=1e-12, abs=0.0)
            p1.grad, p2.grad = torch.ones(2), torch.ones(2)
            o1.step(); o2.step(); s1.step(); s2.step()


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts two ranks itself (gloo here: no GPU needed for the
    launcher's own rendezvous test); a world size that contradicts --gpus is an error, not an N=1 line."""
    env = dict(os.environ, MI_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240, cwd="/tmp")
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert line["world"] == 2 and line["gpus_arg"] == 2 and line["sum"] == 3.0
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240, cwd="/tmp")
    assert bad.returncode != 0 and b"WORLD_SIZE=1" in bad.stderr and not bad.stdout.strip()
