"""MI355X-native counterpart of the hot-path part of the reference's ``mutual_info_img_txt/main_utils.py``.

``MultiModalManager`` keeps the reference's names for the pieces on the path:

* ``create_mi_pairs(embedding_img, embedding_txt, study_id, device)`` -- reference main_utils.py:80-110, same row order,
  built by an integer stream-compaction + one gather kernel instead of one ``torch.cat`` per row;
* ``mi_discriminator`` -- ``make_mlp(d_img + d_txt, [1024, 512])`` as at reference main_utils.py:77;
* ``mi_step(...)`` -- the body of the reference's inner loop, main_utils.py:220-226, on the fused HIP path.

The encoders, datasets, optimisers' bookkeeping and checkpointing of the reference trainer (main_utils.py:112-268)
are out of scope for this tier except for the synthetic-embedding loop in ``train`` used by ``train.py --synthetic``.
"""
from __future__ import annotations

import logging
import time
from typing import Sequence

import torch

from . import _hip, mi_critics
from .model import BilinearCritic, SeparableCritic, make_mlp


class _CreatePairsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, txt, sid):
        lib = _hip.load()
        img = _hip.f32c(img, "embedding_img")
        txt = _hip.f32c(txt, "embedding_txt")
        b, d_img = img.shape
        d_txt = txt.shape[1]
        dev = img.device
        st = _hip.stream_ptr()
        ws = _hip.workspace(lib.mi_pair_index_workspace_bytes(b), dev)
        cap = b * b
        pair_i = torch.empty(cap, dtype=torch.int32, device=dev)
        pair_j = torch.empty(cap, dtype=torch.int32, device=dev)
        n_dev = torch.empty(1, dtype=torch.int64, device=dev)
        rowpos = torch.empty(max(b * (b - 1), 1), dtype=torch.int32, device=dev)
        _hip.check(lib.mi_pair_index(sid.data_ptr(), b, pair_i.data_ptr(), pair_j.data_ptr(), cap, n_dev.data_ptr(),
                                     rowpos.data_ptr(), ws.data_ptr(), ws.numel(), st), "mi_pair_index")
        n_rows = int(n_dev.item())  # the output shape is data dependent (host sync, as any nonzero()-like op)
        out = torch.empty(n_rows, d_img + d_txt, dtype=torch.float32, device=dev)
        _hip.check(lib.mi_create_pairs(img.data_ptr(), txt.data_ptr(), pair_i.data_ptr(), pair_j.data_ptr(), n_rows,
                                       d_img, d_txt, out.data_ptr(), st), "mi_create_pairs")
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
        _hip.check(lib.mi_create_pairs_bwd(g.data_ptr(), rowpos.data_ptr(), b, d_img, d_txt, gi.data_ptr(),
                                           gt.data_ptr(), _hip.stream_ptr()), "mi_create_pairs_bwd")
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
    _hip.check(lib.mi_pair_index(sid.data_ptr(), b, pair_i.data_ptr(), pair_j.data_ptr(), cap, n_dev.data_ptr(), None,
                                 ws.data_ptr(), ws.numel(), _hip.stream_ptr()), "mi_pair_index")
    n = int(n_dev.item())
    return pair_i[:n], pair_j[:n]


class MultiModalManager:
    """Hot-path subset of the reference's MultiModalManager (main_utils.py:53-268)."""

    def __init__(self, d_img: int = 768, d_txt: int = 768, critic: str = "concat_mlp", hidden_dims=(1024, 512),
                 d_proj: int = 256):
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
        """Synthetic-embedding training loop with the reference's step order (main_utils.py:189-235): zero_grad,
        forward, loss.backward(), optimizer step, epoch loss = sum of step losses, and its two log lines.
        ``embedding_source(step) -> (embedding_img, embedding_txt, study_id)`` stands in for the encoders."""
        logger = logging.getLogger(__name__)
        mi_critics._estimator_code(args.mi_estimator)  # eager validation (the reference fails late, main_utils.py:224)
        self.mi_discriminator = self.mi_discriminator.to(device)
        mi_optimizer = torch.optim.Adam(self.mi_discriminator.parameters(), lr=args.init_lr)  # main_utils.py:153
        training_loss = []
        for epoch in range(int(args.num_train_epochs)):
            start_time = time.time()
            epoch_loss = torch.zeros((), device=device)
            for step in range(int(args.steps_per_epoch)):
                embedding_img, embedding_txt, study_id = embedding_source(step)
                mi_optimizer.zero_grad()
                loss = self.mi_step(embedding_img, embedding_txt, study_id, args.mi_estimator,
                                    getattr(args, "precision", "bf16"))
                loss.sum().backward()
                mi_optimizer.step()
                epoch_loss += loss.detach().sum()  # device-side accumulation; one sync per epoch
            epoch_loss = float(epoch_loss.item())
            training_loss.append(epoch_loss)
            interval = time.time() - start_time
            logger.info(f"  Epoch {epoch+1} loss = {epoch_loss:.5f}")
            logger.info(f"  Epoch {epoch+1} took {interval:.3f} s")
        return training_loss
