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
import os
import time
from typing import Sequence

import torch

from . import _hip, mi_critics
from .model import (BilinearCritic, ImageReportModel, ResNet256_6_2_1, SeparableCritic, TextBert, build_bert_model,
                    build_resnet_model, make_mlp)
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


def _is_token_features(source) -> bool:
    """The reference's ``text_token_features``: a list of objects with ``report_id`` / ``input_ids`` / ... fields."""
    return isinstance(source, (list, tuple)) and bool(source) and hasattr(source[0], "report_id")


class MultiModalManager:
    """The reference's MultiModalManager (main_utils.py:53-268) around the MI355X critic path.

    Constructor: the reference's four arguments keep their names and order (``bert_pretrained_dir``, ``bert_config_name``,
    ``output_channels``, ``image_model_name``, main_utils.py:58-59) and build the same objects: ``build_bert_model`` ->
    ``self.text_model`` / ``self.bert_config``, ``build_resnet_model`` -> ``self.image_model``, ``ImageReportModel`` ->
    ``self.model``, ``make_mlp(1536, [1024, 512])`` -> ``self.mi_discriminator``.  Keyword-only extensions: ready-made
    encoder modules (``image_model=``, ``text_model=``: anything that maps a batch to ``[B, d]`` embeddings), the critic
    kind / widths, projection heads (``embed_proj_dim``) and an autocast dtype for the encoders.  With no encoder at all
    the manager trains the critic on embeddings handed in directly (``train.py --synthetic``)."""

    def __init__(self, bert_pretrained_dir=None, bert_config_name=None, output_channels=None, image_model_name=None, *,
                 d_img: int = 768, d_txt: int = 768, critic: str = "concat_mlp", hidden_dims=(1024, 512),
                 d_proj: int = 256, image_model=None, text_model=None, bert_config=None, embed_proj_dim=None,
                 autocast_dtype=None):
        self.bert_pretrained_dir = bert_pretrained_dir
        self.bert_config_name = bert_config_name
        self.output_channels = output_channels
        self.image_model_name = image_model_name
        self.bert_config = bert_config
        if bert_pretrained_dir is not None and bert_config_name is not None and text_model is None:
            text_model, self.bert_config = build_bert_model(bert_pretrained_dir=bert_pretrained_dir,
                                                            bert_config_name=bert_config_name,
                                                            output_channels=output_channels)  # main_utils.py:65-68
        if image_model_name is not None and image_model is None:
            image_model = build_resnet_model(model_name=image_model_name,
                                             output_channels=output_channels if output_channels is not None else 4)
        self.image_model, self.text_model = image_model, text_model
        self.model = None
        if isinstance(image_model, ResNet256_6_2_1) and isinstance(text_model, TextBert):
            # the reference's joint model (main_utils.py:73-75); embeddings: z [B,768] and the pooled [CLS] after dropout
            self.model = ImageReportModel(text_model=text_model, bert_config=self.bert_config, image_model=image_model,
                                          proj_dim=embed_proj_dim, autocast_dtype=autocast_dtype)
            d_img = embed_proj_dim or 768
            d_txt = embed_proj_dim or getattr(self.bert_config, "hidden_size", 768)
        if critic == "concat_mlp":
            self.mi_discriminator = make_mlp(d_img + d_txt, list(hidden_dims))  # reference main_utils.py:77
        elif critic == "bilinear":
            self.mi_discriminator = BilinearCritic(d_img, d_txt)
        elif critic == "separable":
            self.mi_discriminator = SeparableCritic(d_img, d_txt, d_proj)
        else:
            raise ValueError(f"unknown critic {critic!r}: expected concat_mlp, bilinear or separable")
        self.critic_kind = critic
        self.d_img, self.d_txt = d_img, d_txt
        self.training_loss = []
        self._graphed = None
        self._default_set = False
        self.logger = logging.getLogger(__name__)

    def create_mi_pairs(self, embedding_img, embedding_txt, study_id: list, device=None):
        """[N, d_img + d_txt]: B positive rows then the kept negatives, gap-major / i-minor with j = (i+gap+1) mod B,
        a pair kept iff study_id[i] != study_id[j] (reference main_utils.py:80-110)."""
        _hip.require_device(embedding_img, "embedding_img")
        sid = mi_critics.study_id_codes(study_id, embedding_img.device)
        if sid.numel() != embedding_img.shape[0] or embedding_txt.shape[0] != embedding_img.shape[0]:
            raise ValueError("embedding_img, embedding_txt and study_id must have the same length")
        return _CreatePairsFn.apply(embedding_img, embedding_txt, sid)

    def mi_step(self, embedding_img, embedding_txt, study_id, mi_estimator: str = "dv", precision: str = "f32",
                fused: bool = True, graph: bool = False):
        """Reference main_utils.py:220-224.  ``fused=False`` runs the literal three-call sequence (pair kernel, critic
        module, bound kernel) and is only practical for small batches.  ``graph=True`` replays the fused step from
        hipGraphs (``graphed.GraphedMiStep``, built on first use for this batch shape; the training loop's setting)."""
        if fused and graph and self.critic_kind != "separable":
            g = self._graphed
            key = (tuple(embedding_img.shape), tuple(embedding_txt.shape), mi_estimator, precision)
            if g is None or g.key != key:
                from .graphed import GraphedMiStep
                g = GraphedMiStep(self.mi_discriminator, embedding_img.shape[0], embedding_img.shape[1],
                                  embedding_txt.shape[1], mi_estimator, precision, embedding_img.device)
                g.key = key
                self._graphed = g
            return g.loss(embedding_img, embedding_txt, study_id)
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

    # ------------------------------------------------------------------------------------------ batches
    def _build_loader(self, text_token_features, args):
        """Dataset + shuffling ``drop_last`` DataLoader of the reference (main_utils.py:123-129), built once per manager:
        the schedule length ``num_train_epochs * len(data_loader)`` (main_utils.py:168) needs it before the optimisers."""
        loader = getattr(self, "_loader", None)
        if loader is None:
            from .model_utils import build_training_imagereportset
            dataset = build_training_imagereportset(text_token_features=text_token_features, img_dir=args.image_dir,
                                                    img_size=getattr(args, "img_size", 256),
                                                    dataset_metadata=args.dataset_metadata,
                                                    image_loader=getattr(args, "image_loader", None))
            # pinned host batches as the reference (main_utils.py:127-129); GraphedMiStep captures with
            # capture_error_mode="thread_local", so the loader's pin-memory thread cannot invalidate a capture
            loader = torch.utils.data.DataLoader(dataset, batch_size=args.batch_size, shuffle=True,
                                                 num_workers=getattr(args, "data_loader_workers", 0),
                                                 pin_memory=torch.cuda.is_available(), drop_last=True)
            print(f'Total number of training image-report pairs: {len(dataset)}')
            self._loader = loader
        return loader

    def _batches(self, source, device, args):
        """One epoch of batches ``(img, txt, study_id)`` or ``(img, txt_ids, txt_masks, txt_segments, study_id, img_id)``.

        ``source`` is (a) a callable ``step -> (img, txt, study_id)`` (synthetic embeddings / tests; ``steps_per_epoch``
        of them), (b) a DataLoader-like iterable of the reference's 6-tuples (main_utils.py:192), or (c) the reference's
        ``text_token_features`` list, from which the dataset and a shuffling ``drop_last`` DataLoader are built as at
        main_utils.py:123-129 (``model_utils.build_training_imagereportset``)."""
        if callable(source):
            for step in range(int(args.steps_per_epoch)):
                yield source(step)
            return
        if _is_token_features(source):
            source = self._build_loader(source, args)
        dataset = getattr(source, "dataset", None)
        for batch in source:
            if len(batch) == 6 and not self._default_set and hasattr(dataset, "set_default"):
                # the reference hands sample 0 of the first batch to the dataset as the substitute for unreadable rows
                # (main_utils.py:195-199); host tensors, before anything moves to the device
                img, txt_ids, txt_masks, txt_segments, study_id, _ = batch
                dataset.set_default(img[0].clone().detach(), txt_ids[0].clone().detach(), txt_masks[0].clone().detach(),
                                    txt_segments[0].clone().detach(), study_id[0])
                self._default_set = True
            yield batch

    def _embed(self, batch, device):
        """Batch -> (embedding_img, embedding_txt, study_id) through whatever encoders are attached."""
        if len(batch) == 6:  # the reference's batch (main_utils.py:192, 201-204, 218-219)
            img, txt_ids, txt_masks, txt_segments, study_id, _img_id = batch
            img = img.to(device, non_blocking=True)
            txt_ids = txt_ids.to(device, non_blocking=True)
            txt_masks = txt_masks.to(device, non_blocking=True)
            txt_segments = txt_segments.to(device, non_blocking=True)
            if self.model is None:
                raise ValueError("reference-style batches need the joint ImageReportModel (image + text encoders)")
            embedding_img, embedding_txt, _logits_img, _logits_txt = self.model(img, txt_ids, txt_masks, txt_segments)
            return embedding_img, embedding_txt, list(study_id)
        img, txt, study_id = batch
        embedding_img = self.image_model(img) if self.image_model is not None and self.model is None else img
        embedding_txt = self.text_model(txt) if self.text_model is not None and self.model is None else txt
        return embedding_img, embedding_txt, study_id

    def train(self, text_token_features, device, args):
        """The reference's training loop (main_utils.py:112-268) around the fused MI step.

        First argument: see ``_batches`` (the reference passes its ``text_token_features``).  Optimisers and order as the
        reference: Adam(image encoder, init_lr), Adam(critic, init_lr), AdamW(text encoder, lr 2e-5, weight decay 0.1
        except bias / LayerNorm, no bias correction) with a warm-up-linear schedule over ``num_train_epochs *
        len(loader)`` steps, 10 % warm-up (main_utils.py:152-172); per step: zero_grad of all three, forward,
        ``loss.backward()``, then critic, image, text optimiser steps and the scheduler step (main_utils.py:205-229).
        Epoch loss = sum of the step losses.  Per epoch, with encoders attached and ``args.save_directory`` set: the
        reference's three files (``pytorch_MI_image_model.bin``, ``pytorch_MI_text_model.bin``,
        ``pytorch_model_epoch{n}.bin`` + BERT config, main_utils.py:242-245) and its log lines; additionally -- the
        reference never saves the critic -- ``mi_critic_state.pt`` with the critic, the optimisers and the schedule for
        resuming (a new file name, so the reference's loaders are unaffected).  Returns the list of epoch losses (the
        reference returns None) and keeps it in ``self.training_loss``."""
        logger = logging.getLogger(__name__)
        mi_critics._estimator_code(args.mi_estimator)  # eager validation (the reference fails late, main_utils.py:224)
        self.mi_discriminator = self.mi_discriminator.to(device)
        if self.model is not None:
            self.model = self.model.to(device)
        mi_optimizer = torch.optim.Adam(self.mi_discriminator.parameters(), lr=args.init_lr)  # main_utils.py:153
        img_optimizer = txt_optimizer = scheduler = None
        # schedule length = num_train_epochs * len(data_loader) (reference main_utils.py:168): the loader exists before the
        # optimisers, as at main_utils.py:123-129
        if _is_token_features(text_token_features):
            text_token_features = self._build_loader(text_token_features, args)
        if callable(text_token_features):
            steps_per_epoch = int(getattr(args, "steps_per_epoch", 0) or 0)
        elif hasattr(text_token_features, "__len__"):
            steps_per_epoch = len(text_token_features)
        else:
            steps_per_epoch = int(getattr(args, "steps_per_epoch", 0) or 0)
        if self.image_model is not None:
            self.image_model = self.image_model.to(device).train()
            img_params = list(self.image_model.parameters())
            if self.model is not None and self.model.proj_img is not None:
                img_params += list(self.model.proj_img.parameters()) + list(self.model.proj_txt.parameters())
            img_optimizer = torch.optim.Adam(img_params, lr=args.init_lr)  # main_utils.py:152
        if self.text_model is not None:
            self.text_model = self.text_model.to(device).train()
            no_decay = ['bias', 'LayerNorm.bias', 'LayerNorm.weight']  # main_utils.py:158-165
            param_txt = list(self.text_model.named_parameters())
            grouped = [{'params': [p for n, p in param_txt if not any(nd in n for nd in no_decay)], 'weight_decay': 0.1},
                       {'params': [p for n, p in param_txt if any(nd in n for nd in no_decay)], 'weight_decay': 0.0}]
            txt_optimizer = AdamW(grouped, lr=getattr(args, "txt_lr", 2e-5), correct_bias=False)
            num_train_steps = int(args.num_train_epochs * max(steps_per_epoch, 1))
            scheduler = WarmupLinearSchedule(txt_optimizer, warmup_steps=0.1 * num_train_steps, t_total=num_train_steps)
        self.scheduler = scheduler
        precision = getattr(args, "precision", "f32")
        use_graph = bool(getattr(args, "graph", True))
        save_dir = getattr(args, "save_directory", None)
        start_epoch = 0
        resume = getattr(args, "resume_from", None)
        if resume:
            start_epoch = self.load_training_state(resume, mi_optimizer, img_optimizer, txt_optimizer, scheduler, device)
        training_loss = self.training_loss = list(self.training_loss[:start_epoch])
        for epoch in range(start_epoch, int(args.num_train_epochs)):
            start_time = time.time()
            epoch_loss = torch.zeros((), device=device)
            for batch in self._batches(text_token_features, device, args):
                if img_optimizer is not None:
                    img_optimizer.zero_grad()
                if txt_optimizer is not None:
                    txt_optimizer.zero_grad()
                mi_optimizer.zero_grad()
                embedding_img, embedding_txt, study_id = self._embed(batch, device)
                loss = self.mi_step(embedding_img, embedding_txt, study_id, args.mi_estimator, precision, graph=use_graph)
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
            if save_dir:
                os.makedirs(save_dir, exist_ok=True)
                if self.model is not None:  # main_utils.py:242-245 and the log lines :253-255
                    image_model_file_path = self.model.save_image_model(save_dir)
                    text_model_file_path = self.model.save_text_model(save_dir)
                    checkpoint_path = self.model.save_pretrained(save_dir, epoch=epoch + 1)
                    logger.info(f"  Epoch {epoch+1} checkpoint saved in {checkpoint_path}")
                    logger.info(f"  Image model saved in {image_model_file_path}")
                    logger.info(f"  Text model saved in {text_model_file_path}")
                self.save_training_state(save_dir, epoch + 1, mi_optimizer, img_optimizer, txt_optimizer, scheduler)
        if save_dir and training_loss:
            _plot_losses(training_loss, os.path.join(save_dir, 'mutual_information_training.png'))  # main_utils.py:259-266
        return training_loss

    # ------------------------------------------------------------------------------------------ resume (new capability)
    CRITIC_STATE_FILE = 'mi_critic_state.pt'

    def save_training_state(self, save_directory, epoch, mi_optimizer, img_optimizer=None, txt_optimizer=None,
                            scheduler=None):
        """Critic, optimiser moments, schedule position and finished-epoch count.  Encoder WEIGHTS: the joint
        ``ImageReportModel`` already writes ``pytorch_model_epoch{n}.bin`` every epoch (projection heads included,
        main_utils.py:244 of the reference) and that file is what a resume reloads; encoders handed in as plain modules
        (no joint model) have no file of their own, so their state dicts go into this one."""
        os.makedirs(save_directory, exist_ok=True)
        state = {"epoch": int(epoch), "critic_kind": self.critic_kind, "training_loss": list(self.training_loss),
                 "mi_discriminator": self.mi_discriminator.state_dict(), "mi_optimizer": mi_optimizer.state_dict(),
                 "img_optimizer": None if img_optimizer is None else img_optimizer.state_dict(),
                 "txt_optimizer": None if txt_optimizer is None else txt_optimizer.state_dict(),
                 "scheduler": None if scheduler is None else {"last_epoch": scheduler.last_epoch},
                 "image_model": None, "text_model": None}
        if self.model is None:
            if self.image_model is not None:
                state["image_model"] = self.image_model.state_dict()
            if self.text_model is not None:
                state["text_model"] = self.text_model.state_dict()
        path = os.path.join(save_directory, self.CRITIC_STATE_FILE)
        torch.save(state, path)
        return path

    def load_training_state(self, path, mi_optimizer, img_optimizer=None, txt_optimizer=None, scheduler=None,
                            device=None) -> int:
        """Restore what ``save_training_state`` wrote (tensors, numbers, lists and dicts only: loaded with
        ``weights_only=True``) AND the encoder weights of the same epoch -- optimiser moments and a late-schedule
        learning rate paired with freshly initialised encoders would not be a resumed run.  Raises when encoders are
        attached and their weights of that epoch cannot be found.  Returns the number of finished epochs."""
        if os.path.isdir(path):
            path = os.path.join(path, self.CRITIC_STATE_FILE)
        state = torch.load(path, map_location=device or 'cpu', weights_only=True)
        epoch = int(state["epoch"])
        if self.model is not None:
            from .encoders import JOINT_MODEL_EPOCH_FILE
            joint = os.path.join(os.path.dirname(path), JOINT_MODEL_EPOCH_FILE.format(epoch))
            if not os.path.isfile(joint):
                raise FileNotFoundError(f"resume: encoders are attached but {joint} (the joint checkpoint of epoch {epoch}) "
                                        "is missing; refusing to pair restored optimiser state with other encoder weights")
            self.model.load_state_dict(torch.load(joint, map_location=device or 'cpu', weights_only=True))
        else:
            for name, module in (("image_model", self.image_model), ("text_model", self.text_model)):
                if module is None:
                    continue
                if state.get(name) is None:
                    raise FileNotFoundError(f"resume: {name} is attached but {path} holds no weights for it")
                module.load_state_dict(state[name])
        self.mi_discriminator.load_state_dict(state["mi_discriminator"])
        mi_optimizer.load_state_dict(state["mi_optimizer"])
        if img_optimizer is not None and state.get("img_optimizer") is not None:
            img_optimizer.load_state_dict(state["img_optimizer"])
        if txt_optimizer is not None and state.get("txt_optimizer") is not None:
            txt_optimizer.load_state_dict(state["txt_optimizer"])
        if scheduler is not None and state.get("scheduler") is not None:
            # LambdaLR.step() sets last_epoch + 1 and the lr from it: set both, so the first resumed step runs at the
            # learning rate the uninterrupted run would have used
            scheduler.last_epoch = int(state["scheduler"]["last_epoch"])
            for group, base_lr, fn in zip(scheduler.optimizer.param_groups, scheduler.base_lrs, scheduler.lr_lambdas):
                group["lr"] = base_lr * fn(scheduler.last_epoch)
        self.training_loss = list(state.get("training_loss", []))
        return epoch


def _plot_losses(training_loss, path):
    """Loss curve of the reference (main_utils.py:259-266); skipped silently where matplotlib is unavailable."""
    try:
        import matplotlib
        matplotlib.use("Agg")
        from matplotlib import pyplot as plt
    except Exception:
        return
    plt.xlabel('Epochs')
    plt.ylabel('Value for Loss')
    plt.plot(training_loss, label="train loss")
    plt.legend()
    plt.savefig(path)
    plt.clf()
