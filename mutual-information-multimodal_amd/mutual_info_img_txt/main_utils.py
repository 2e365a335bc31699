"""MI355X-native counterpart of the hot-path part of the reference's ``mutual_info_img_txt/main_utils.py``.

``MultiModalManager`` keeps the reference's names for the pieces on the path:

* ``create_mi_pairs(embedding_img, embedding_txt, study_id, device)`` -- reference main_utils.py:80-110, same row order,
  built by an integer stream-compaction + one gather kernel instead of one ``torch.cat`` per row;
* ``mi_discriminator`` -- ``make_mlp(d_img + d_txt, [1024, 512])`` as at reference main_utils.py:77;
* ``mi_step(...)`` -- the body of the reference's inner loop, main_utils.py:220-226, on the fused HIP path.

``train`` is the reference's loop (main_utils.py:112-268) around that step: its three optimisers (Adam for the image
encoder, Adam for the critic, AdamW without bias correction + warm-up-linear schedule for the text encoder), its
zero_grad / backward / step order and its epoch log lines.  The encoders themselves, the datasets and the checkpoint
files are out of scope for this tier: ``train`` takes any pair of ``nn.Module`` encoders (or none: the source then
yields embeddings directly, as ``train.py --synthetic`` does).
"""
from __future__ import annotations

import logging
import time
from typing import Sequence

import torch

from . import _hip, mi_critics
from .model import BilinearCritic, SeparableCritic, make_mlp
from .optimization import AdamW, WarmupLinearSchedule


class _CreatePairsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, txt, sid):
        lib = _hip.load()
        img = _hip.f32c(img, "embedding_img")
        txt = _hip.f32c(txt, "embedding_txt")
        b, d_img = img.shape
        d_txt = txt.shape[1]
        dev = img.device
        ws = _hip.workspace(lib.mi_pair_index_workspace_bytes(b), dev)
        cap = b * b
        pair_i = torch.empty(cap, dtype=torch.int32, device=dev)
        pair_j = torch.empty(cap, dtype=torch.int32, device=dev)
        n_dev = torch.empty(1, dtype=torch.int64, device=dev)
        rowpos = torch.empty(max(b * (b - 1), 1), dtype=torch.int32, device=dev)
        _hip.call("mi_pair_index", dev, sid.data_ptr(), b, pair_i.data_ptr(), pair_j.data_ptr(), cap, n_dev.data_ptr(),
                                     rowpos.data_ptr(), ws.data_ptr(), ws.numel())
        n_rows = int(n_dev.item())  # the output shape is data dependent (host sync, as any nonzero()-like op)
        out = torch.empty(n_rows, d_img + d_txt, dtype=torch.float32, device=dev)
        _hip.call("mi_create_pairs", dev, img.data_ptr(), txt.data_ptr(), pair_i.data_ptr(), pair_j.data_ptr(), n_rows,
                                       d_img, d_txt, out.data_ptr())
        ctx.save_for_backward(rowpos)
        ctx.dims = (b, d_img, d_txt)
        ctx.pair_index = (pair_i[:n_rows], pair_j[:n_rows])
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _hip.load()
        (rowpos,) = ctx.saved_tensors
        b, d_img, d_txt = ctx.dims
        g = _hip.f32c(grad_out, "grad of mi_input")
        gi = torch.empty(b, d_img, dtype=torch.float32, device=g.device)
        gt = torch.empty(b, d_txt, dtype=torch.float32, device=g.device)
        _hip.call("mi_create_pairs_bwd", g.device, g.data_ptr(), rowpos.data_ptr(), b, d_img, d_txt, gi.data_ptr(),
                                           gt.data_ptr())
        return gi, gt, None


def pair_index(study_id: Sequence, device):
    """(pair_i, pair_j) int32 device tensors: the (image, text) index of every row of mi_input in reference order."""
    lib = _hip.load()
    sid = mi_critics.study_id_codes(study_id, device)
    b = sid.numel()
    ws = _hip.workspace(lib.mi_pair_index_workspace_bytes(b), sid.device)
    cap = b * b
    pair_i = torch.empty(cap, dtype=torch.int32, device=sid.device)
    pair_j = torch.empty(cap, dtype=torch.int32, device=sid.device)
    n_dev = torch.empty(1, dtype=torch.int64, device=sid.device)
    _hip.call("mi_pair_index", sid.device, sid.data_ptr(), b, pair_i.data_ptr(), pair_j.data_ptr(), cap, n_dev.data_ptr(), None,
                                 ws.data_ptr(), ws.numel())
    n = int(n_dev.item())
    return pair_i[:n], pair_j[:n]


class MultiModalManager:
    """Hot-path subset of the reference's MultiModalManager (main_utils.py:53-268)."""

    def __init__(self, d_img: int = 768, d_txt: int = 768, critic: str = "concat_mlp", hidden_dims=(1024, 512),
                 d_proj: int = 256, image_model=None, text_model=None):
        # optional encoders (the reference's self.model.image_model / text_model, model.py:540-555): any modules that
        # map a batch to [B, d_img] / [B, d_txt] fp32 embeddings
        self.image_model, self.text_model = image_model, text_model
        if critic == "concat_mlp":
            self.mi_discriminator = make_mlp(d_img + d_txt, list(hidden_dims))  # reference main_utils.py:77
        elif critic == "bilinear":
            self.mi_discriminator = BilinearCritic(d_img, d_txt)
        elif critic == "separable":
            self.mi_discriminator = SeparableCritic(d_img, d_txt, d_proj)
        else:
            raise ValueError(f"unknown critic {critic!r}: expected concat_mlp, bilinear or separable")
        self.critic_kind = critic
        self.logger = logging.getLogger(__name__)

    def create_mi_pairs(self, embedding_img, embedding_txt, study_id: list, device=None):
        """[N, d_img + d_txt]: B positive rows then the kept negatives, gap-major / i-minor with j = (i+gap+1) mod B,
        a pair kept iff study_id[i] != study_id[j] (reference main_utils.py:80-110)."""
        _hip.require_device(embedding_img, "embedding_img")
        sid = mi_critics.study_id_codes(study_id, embedding_img.device)
        if sid.numel() != embedding_img.shape[0] or embedding_txt.shape[0] != embedding_img.shape[0]:
            raise ValueError("embedding_img, embedding_txt and study_id must have the same length")
        return _CreatePairsFn.apply(embedding_img, embedding_txt, sid)

    def mi_step(self, embedding_img, embedding_txt, study_id, mi_estimator: str = "dv", precision: str = "bf16",
                fused: bool = True):
        """Reference main_utils.py:220-224.  fused=False runs the literal three-call sequence (pair kernel, critic
        module, bound kernel) and is only practical for small batches."""
        if fused:
            return mi_critics.fused_mi_bound(embedding_img, embedding_txt, study_id, self.mi_discriminator,
                                             mi_estimator, precision)
        if self.critic_kind != "concat_mlp":
            scores = self.mi_discriminator(embedding_img, embedding_txt)
            return mi_critics.matrix_bound_loss(scores, study_id, mi_estimator)
        mi_input = self.create_mi_pairs(embedding_img, embedding_txt, study_id, embedding_img.device)
        mi_output = self.mi_discriminator(mi_input)
        critic = {"dv": mi_critics.dv_bound_loss, "infonce": mi_critics.infonce_bound_loss}[mi_estimator]
        return critic(mi_output, len(study_id), embedding_img.device)

    def train(self, embedding_source, device, args):
        """The reference's training loop (main_utils.py:112-268) around the fused MI step.

        ``embedding_source(step) -> (img, txt, study_id)``: with encoders attached, ``img`` / ``txt`` are the encoders'
        inputs (one batch, ``drop_last`` batching as main_utils.py:127-129); without, they are the embeddings.
        Optimisers and order as the reference: Adam(image encoder, init_lr), Adam(critic, init_lr), AdamW(text encoder,
        lr 2e-5, weight decay 0.1 except bias / LayerNorm, no bias correction) with a warm-up-linear schedule over
        ``num_train_epochs * steps_per_epoch`` steps (10 % warm-up) -- main_utils.py:152-172; per step: zero_grad of
        all three, forward, ``loss.backward()``, then critic, image, text optimiser steps and the scheduler step --
        main_utils.py:205-229.  Epoch loss = sum of the step losses; the reference's two log lines per epoch."""
        logger = logging.getLogger(__name__)
        mi_critics._estimator_code(args.mi_estimator)  # eager validation (the reference fails late, main_utils.py:224)
        self.mi_discriminator = self.mi_discriminator.to(device)
        mi_optimizer = torch.optim.Adam(self.mi_discriminator.parameters(), lr=args.init_lr)  # main_utils.py:153
        img_optimizer = txt_optimizer = scheduler = None
        if self.image_model is not None:
            self.image_model = self.image_model.to(device).train()
            img_optimizer = torch.optim.Adam(self.image_model.parameters(), lr=args.init_lr)  # main_utils.py:152
        if self.text_model is not None:
            self.text_model = self.text_model.to(device).train()
            no_decay = ['bias', 'LayerNorm.bias', 'LayerNorm.weight']  # main_utils.py:158-165
            param_txt = list(self.text_model.named_parameters())
            grouped = [{'params': [p for n, p in param_txt if not any(nd in n for nd in no_decay)], 'weight_decay': 0.1},
                       {'params': [p for n, p in param_txt if any(nd in n for nd in no_decay)], 'weight_decay': 0.0}]
            txt_optimizer = AdamW(grouped, lr=getattr(args, "txt_lr", 2e-5), correct_bias=False)
            num_train_steps = int(args.num_train_epochs * args.steps_per_epoch)
            scheduler = WarmupLinearSchedule(txt_optimizer, warmup_steps=0.1 * num_train_steps, t_total=num_train_steps)
        precision = getattr(args, "precision", "bf16")
        training_loss = []
        for epoch in range(int(args.num_train_epochs)):
            start_time = time.time()
            epoch_loss = torch.zeros((), device=device)
            for step in range(int(args.steps_per_epoch)):
                img, txt, study_id = embedding_source(step)
                if img_optimizer is not None:
                    img_optimizer.zero_grad()
                if txt_optimizer is not None:
                    txt_optimizer.zero_grad()
                mi_optimizer.zero_grad()
                embedding_img = self.image_model(img) if self.image_model is not None else img
                embedding_txt = self.text_model(txt) if self.text_model is not None else txt
                loss = self.mi_step(embedding_img, embedding_txt, study_id, args.mi_estimator, precision)
                loss.sum().backward()
                mi_optimizer.step()
                if img_optimizer is not None:
                    img_optimizer.step()
                if txt_optimizer is not None:
                    txt_optimizer.step()
                    scheduler.step()
                epoch_loss += loss.detach().sum()  # device-side accumulation; one sync per epoch
            epoch_loss = float(epoch_loss.item())
            training_loss.append(epoch_loss)
            interval = time.time() - start_time
            logger.info(f"  Epoch {epoch+1} loss = {epoch_loss:.5f}")
            logger.info(f"  Epoch {epoch+1} took {interval:.3f} s")
        return training_loss
