"""MI355X-native drop-in for the reference's ``mutual_info_img_txt/mi_critics.py``.

Same two callables with the same signatures and return shapes as the reference:

* ``dv_bound_loss(discriminator_logits, pos_size, device)``      -- reference mi_critics.py:3-12,  returns shape [1]
* ``infonce_bound_loss(discriminator_logits, pos_size, device)`` -- reference mi_critics.py:14-23, returns shape []

plus the fused entry point that replaces lines ``main_utils.py:220-224`` of the reference training step
(create_mi_pairs -> mi_discriminator -> mi_critic) without materialising ``mi_input`` / ``mi_output``:

* ``fused_mi_bound(embedding_img, embedding_txt, study_id, critic, estimator, ...)``

Every function runs hand-written HIP kernels through the C ABI in ``include/mi_critic.h`` (loaded with ctypes,
wrapped in ``torch.autograd.Function``).  There is no CPU path: CPU tensors raise.
"""
from __future__ import annotations

from typing import Optional, Sequence, Union

import torch

from . import _hip
from ._hip import ESTIMATORS, PRECISIONS

__all__ = ["dv_bound_loss", "infonce_bound_loss", "matrix_bound_loss", "fused_mi_bound", "study_id_codes",
           "BilinearCriticFn", "SeparableCriticFn", "ConcatMlpCriticFn"]


# ----------------------------------------------------------------------------------------------------------
# study ids: list[str] in the reference (model_utils.py:212); compared with != only (main_utils.py:105)
# ----------------------------------------------------------------------------------------------------------
def study_id_codes(study_id: Union[Sequence, torch.Tensor], device) -> torch.Tensor:
    """int64 device tensor with equal code <=> equal study id.  The codes are a pure function of each id
    (``utils.study_id_to_int64``: the numeric value of ids like "50414267", a 62-bit hash otherwise), hence identical in
    every process of a sharded run -- a first-seen numbering would not be."""
    from .utils import study_ids_to_tensor
    return study_ids_to_tensor(study_id, device)


def _estimator_code(estimator: str) -> int:
    if estimator not in ESTIMATORS:
        # the reference leaves mi_critic unbound for an unknown estimator (main_utils.py:141-144, UnboundLocalError
        # at :224); here it is rejected eagerly
        raise ValueError(f"unknown mi_estimator {estimator!r}: expected one of {sorted(ESTIMATORS)}")
    return ESTIMATORS[estimator]


def _precision_code(precision: str) -> int:
    if precision not in PRECISIONS:
        raise ValueError(f"unknown precision {precision!r}: expected one of {sorted(PRECISIONS)}")
    return PRECISIONS[precision]


def _grad_scalar(grad: torch.Tensor) -> torch.Tensor:
    return grad.reshape(-1)[:1].to(torch.float32).contiguous()


# ----------------------------------------------------------------------------------------------------------
# a3 / a4: bound on materialised logits
# ----------------------------------------------------------------------------------------------------------
class _BoundFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: torch.Tensor, pos_size: int, estimator: int):
        lib = _hip.load()
        flat = _hip.f32c(logits, "discriminator_logits").reshape(-1)
        n = flat.numel()
        dev = flat.device
        ws = _hip.workspace(lib.mi_bound_workspace_bytes(n), dev)
        stats = _hip.new_stats(dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        _hip.call("mi_bound_fwd", dev, flat.data_ptr(), n, int(pos_size), estimator, loss.data_ptr(), stats.data_ptr(),
                                    ws.data_ptr(), ws.numel())
        ctx.save_for_backward(flat, stats)
        ctx.pos_size = int(pos_size)
        ctx.in_shape = logits.shape
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        lib = _hip.load()
        flat, stats = ctx.saved_tensors
        go = _grad_scalar(grad_loss)
        grad = torch.empty_like(flat)
        _hip.call("mi_bound_bwd", flat.device, flat.data_ptr(), flat.numel(), ctx.pos_size, stats.data_ptr(), go.data_ptr(),
                                    grad.data_ptr())
        return grad.reshape(ctx.in_shape), None, None


def _bound(discriminator_logits, pos_size, estimator):
    _hip.require_device(discriminator_logits, "discriminator_logits")
    n = discriminator_logits.shape[0]
    if discriminator_logits.numel() != n:
        raise ValueError("discriminator_logits must be [N] or [N, 1] (one score per pair row)")
    if not 0 <= int(pos_size) <= n:
        raise ValueError(f"pos_size={pos_size} outside [0, {n}]")
    return _BoundFn.apply(discriminator_logits, int(pos_size), estimator)


def dv_bound_loss(discriminator_logits: torch.Tensor, pos_size: int, device=None) -> torch.Tensor:
    """Donsker-Varadhan bound, reference mi_critics.py:3-12: ``LSE(logits[pos:]) - log(N - pos) - mean(logits[:pos])``.
    ``device`` is kept for signature compatibility (the reference only uses it for the log-N constant)."""
    loss = _bound(discriminator_logits, pos_size, _hip.MI_DV)
    return loss.reshape(discriminator_logits.shape[1:])  # [N,1] -> [1] as in the reference


def infonce_bound_loss(discriminator_logits: torch.Tensor, pos_size: int, device=None) -> torch.Tensor:
    """The reference's "InfoNCE" bound, mi_critics.py:14-23: ``LSE(logits[pos:]) - mean(logits[:pos])`` (no log-N term;
    not a row-wise softmax cross-entropy).  Returns shape []."""
    return _bound(discriminator_logits, pos_size, _hip.MI_INFONCE).reshape(())


# ----------------------------------------------------------------------------------------------------------
# bound on a B x B score matrix with study-id masking
# ----------------------------------------------------------------------------------------------------------
class _MatrixBoundFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores: torch.Tensor, sid: torch.Tensor, estimator: int):
        lib = _hip.load()
        s = _hip.f32c(scores, "scores")
        b = s.shape[0]
        dev = s.device
        ws = _hip.workspace(lib.mi_matrix_bound_workspace_bytes(b), dev)
        stats = _hip.new_stats(dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        _hip.call("mi_matrix_bound_fwd", dev, s.data_ptr(), sid.data_ptr(), b, estimator, loss.data_ptr(),
                                           stats.data_ptr(), ws.data_ptr(), ws.numel())
        ctx.save_for_backward(s, sid, stats)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        lib = _hip.load()
        s, sid, stats = ctx.saved_tensors
        go = _grad_scalar(grad_loss)
        grad = torch.empty_like(s)
        _hip.call("mi_matrix_bound_bwd", s.device, s.data_ptr(), sid.data_ptr(), s.shape[0], stats.data_ptr(), go.data_ptr(),
                                           grad.data_ptr())
        return grad, None, None


def matrix_bound_loss(scores: torch.Tensor, study_id, estimator: str = "dv") -> torch.Tensor:
    """The reference loss on a [B,B] score matrix S[i,j] = critic(img_i, txt_j): positives are the diagonal, negatives
    the pairs with i != j and different study ids (main_utils.py:99-108).  Shape [1] (dv) / [] (infonce)."""
    _hip.require_device(scores, "scores")
    if scores.dim() != 2 or scores.shape[0] != scores.shape[1]:
        raise ValueError("scores must be [B, B]")
    code = _estimator_code(estimator)
    sid = study_id_codes(study_id, scores.device)
    if sid.numel() != scores.shape[0]:
        raise ValueError("study_id length must equal B")
    loss = _MatrixBoundFn.apply(scores, sid, code)
    return loss if estimator == "dv" else loss.reshape(())


# ----------------------------------------------------------------------------------------------------------
# fused critics
# ----------------------------------------------------------------------------------------------------------
class BilinearCriticFn(torch.autograd.Function):
    """loss = bound(S), S = (X W) Y^T with study-id masking; all gradients by the HIP backward."""

    @staticmethod
    def forward(ctx, x, y, w, sid, estimator: int, precision: int, want_scores: bool):
        lib = _hip.load()
        x = _hip.f32c(x, "embedding_img")
        y = _hip.f32c(y, "embedding_txt")
        w = None if w is None else _hip.f32c(w, "bilinear weight")
        b, dx = x.shape
        dy = y.shape[1]
        dev = x.device
        if w is not None:
            _hip.note_path("bilinear", (b, b, dx, dy), precision)
        ws = _hip.workspace(lib.mi_bilinear_workspace_bytes(b, b, dx, dy, precision), dev)
        stats = _hip.new_stats(dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        record = torch.empty(_hip.RECORD_FLOATS, dtype=torch.float32, device=dev)
        scores = torch.empty(b, b, dtype=torch.float32, device=dev) if want_scores else None
        need_grad = 1 if any(ctx.needs_input_grad[:3]) else 0
        _hip.call("mi_bilinear_fwd", dev, x.data_ptr(), y.data_ptr(), _hip.ptr(w), sid.data_ptr(), sid.data_ptr(), b, b, 0,
                                       dx, dy, estimator, precision, need_grad, loss.data_ptr(), stats.data_ptr(),
                                       record.data_ptr(), _hip.ptr(scores), ws.data_ptr(), ws.numel())
        ctx.save_for_backward(x, y, sid, stats, ws, *([] if w is None else [w]))
        ctx.precision = precision
        ctx.mark_non_differentiable(stats)
        if want_scores:
            ctx.mark_non_differentiable(scores)
            return loss, stats, scores
        return loss, stats, None

    @staticmethod
    def backward(ctx, grad_loss, _gs, _gsc):
        lib = _hip.load()
        x, y, sid, stats, ws, *rest = ctx.saved_tensors
        w = rest[0] if rest else None
        b, dx = x.shape
        dy = y.shape[1]
        go = _grad_scalar(grad_loss)
        gx, gy = torch.empty_like(x), torch.empty_like(y)
        gw = None if w is None else torch.empty_like(w)
        _hip.call("mi_bilinear_bwd", x.device, x.data_ptr(), y.data_ptr(), _hip.ptr(w), sid.data_ptr(), sid.data_ptr(), b, b, 0,
                                       dx, dy, ctx.precision, stats.data_ptr(), go.data_ptr(), gx.data_ptr(),
                                       gy.data_ptr(), _hip.ptr(gw), ws.data_ptr(), ws.numel(), 1)
        return gx, gy, gw, None, None, None, None


class SeparableCriticFn(torch.autograd.Function):
    """loss = bound(S), S = (X Wg)(Y Wh)^T with study-id masking; projections, fused B x B stage and all gradients by
    the HIP library (BASELINE.json configs[1])."""

    @staticmethod
    def forward(ctx, x, y, wg, wh, sid, estimator: int, precision: int):
        lib = _hip.load()
        x, y = _hip.f32c(x, "embedding_img"), _hip.f32c(y, "embedding_txt")
        wg, wh = _hip.f32c(wg, "image projection"), _hip.f32c(wh, "text projection")
        b, dx = x.shape
        dy, k = y.shape[1], wg.shape[1]
        if wg.shape[0] != dx or wh.shape != (dy, k):
            raise ValueError("projection shapes must be [d_img, d_proj] and [d_txt, d_proj]")
        dev = x.device
        _hip.note_path("separable", (b, b, dx, dy, k), precision)
        ws = _hip.workspace(lib.mi_separable_workspace_bytes(b, b, dx, dy, k, precision), dev)
        stats = _hip.new_stats(dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        record = torch.empty(_hip.RECORD_FLOATS, dtype=torch.float32, device=dev)
        need_grad = 1 if any(ctx.needs_input_grad[:4]) else 0
        _hip.call("mi_separable_fwd", dev, x.data_ptr(), y.data_ptr(), wg.data_ptr(), wh.data_ptr(), sid.data_ptr(),
                  sid.data_ptr(), b, b, 0, dx, dy, k, estimator, precision, need_grad, loss.data_ptr(), stats.data_ptr(),
                  record.data_ptr(), ws.data_ptr(), ws.numel())
        ctx.save_for_backward(x, y, wg, wh, sid, stats, ws)
        ctx.precision = precision
        ctx.mark_non_differentiable(stats)
        return loss, stats

    @staticmethod
    def backward(ctx, grad_loss, _gs):
        x, y, wg, wh, sid, stats, ws = ctx.saved_tensors
        b, dx = x.shape
        dy, k = y.shape[1], wg.shape[1]
        go = _grad_scalar(grad_loss)
        gx, gy, gg, gh = (torch.empty_like(t) for t in (x, y, wg, wh))
        _hip.call("mi_separable_bwd", x.device, x.data_ptr(), y.data_ptr(), wg.data_ptr(), wh.data_ptr(), sid.data_ptr(),
                  sid.data_ptr(), b, b, 0, dx, dy, k, ctx.precision, stats.data_ptr(), go.data_ptr(), gx.data_ptr(),
                  gy.data_ptr(), gg.data_ptr(), gh.data_ptr(), ws.data_ptr(), ws.numel(), 1)
        return gx, gy, gg, gh, None, None, None


class ConcatMlpCriticFn(torch.autograd.Function):
    """loss = bound(S), S[i,j] = MLP([x_i ; y_j]) with the reference critic make_mlp(d,[h1,h2]) (model.py:18-32)."""

    @staticmethod
    def forward(ctx, x, y, w1, b1, w2, b2, w3, b3, sid, estimator: int, precision: int, want_scores: bool):
        lib = _hip.load()
        x = _hip.f32c(x, "embedding_img")
        y = _hip.f32c(y, "embedding_txt")
        params = [_hip.f32c(p, f"critic param {n}") for n, p in enumerate((w1, b1, w2, b2, w3, b3))]
        b, dx = x.shape
        dy = y.shape[1]
        h1, h2 = params[0].shape[0], params[2].shape[0]
        dev = x.device
        need_grad = 1 if any(ctx.needs_input_grad[:8]) else 0
        ws = _hip.workspace(lib.mi_concat_mlp_workspace_bytes(b, b, dx, dy, h1, h2, precision, need_grad), dev)
        stats = _hip.new_stats(dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        record = torch.empty(_hip.RECORD_FLOATS, dtype=torch.float32, device=dev)
        scores = torch.empty(b, b, dtype=torch.float32, device=dev)
        _hip.call("mi_concat_mlp_fwd", dev, x.data_ptr(), y.data_ptr(), *[p.data_ptr() for p in params], sid.data_ptr(),
                                         sid.data_ptr(), b, b, 0, dx, dy, h1, h2, estimator, precision, need_grad,
                                         loss.data_ptr(), stats.data_ptr(), record.data_ptr(), scores.data_ptr(),
                                         ws.data_ptr(), ws.numel())
        ctx.save_for_backward(x, y, *params, sid, stats, scores, ws)
        ctx.precision = precision
        ctx.mark_non_differentiable(stats, scores)
        return loss, stats, scores

    @staticmethod
    def backward(ctx, grad_loss, _gs, _gsc):
        lib = _hip.load()
        x, y, w1, b1, w2, b2, w3, b3, sid, stats, scores, ws = ctx.saved_tensors
        b, dx = x.shape
        dy = y.shape[1]
        h1, h2 = w1.shape[0], w2.shape[0]
        go = _grad_scalar(grad_loss)
        grads = [torch.empty_like(t) for t in (x, y, w1, b1, w2, b2, w3, b3)]
        _hip.call("mi_concat_mlp_bwd", x.device, x.data_ptr(), y.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                         b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), sid.data_ptr(), sid.data_ptr(), b,
                                         b, 0, dx, dy, h1, h2, ctx.precision, stats.data_ptr(), go.data_ptr(),
                                         scores.data_ptr(), *[g.data_ptr() for g in grads], ws.data_ptr(), ws.numel())
        return (*grads, None, None, None, None)


def _concat_params(critic):
    """(W1,b1,W2,b2,W3,b3) of an nn.Sequential built by make_mlp(input_dim,[h1,h2]) (reference model.py:18-32)."""
    mods = list(critic)
    lin = [m for m in mods if isinstance(m, torch.nn.Linear)]
    act = [m for m in mods if not isinstance(m, torch.nn.Linear)]
    if len(lin) != 3 or len(mods) != 5 or not all(isinstance(a, torch.nn.ReLU) for a in act) or lin[2].out_features != 1:
        raise ValueError("the fused concat-MLP path supports make_mlp(input_dim, [h1, h2]) critics "
                         "(Linear-ReLU-Linear-ReLU-Linear(->1)); use create_mi_pairs + the critic module otherwise")
    return lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias, lin[2].weight, lin[2].bias


def fused_mi_bound(embedding_img: torch.Tensor, embedding_txt: torch.Tensor, study_id, critic, estimator: str = "dv",
                   precision: str = "f32", return_scores: bool = False, return_stats: bool = False):
    """Fused replacement of the reference lines main_utils.py:220-224:

        mi_input  = self.create_mi_pairs(embedding_img, embedding_txt, study_id, device)
        mi_output = self.mi_discriminator(mi_input)
        loss      = mi_critic(mi_output, args.batch_size, device)

    ``critic`` is the reference's ``make_mlp(d_img+d_txt,[h1,h2])`` nn.Sequential, or a ``BilinearCritic`` /
    ``SeparableCritic`` from ``mutual_info_img_txt.model`` (extensions).  Returns the loss (shape [1] for "dv",
    [] for "infonce", as the reference) and optionally the [B,B] score matrix S[i,j] = critic(img_i, txt_j).

    ``precision`` defaults to "f32": the reference critic is fp32 throughout and this is the mode whose results match it
    within the stated fp32 tolerances.  For the concat-MLP and separable critics it means exact fp32
    products on the fp32-input MFMA; for ``BilinearCritic`` (sizes multiples of 8) it runs the "bf16x3" scheme below, which
    meets the same tolerances at several times the speed -- pass "f32_exact" to insist on exact fp32 products.  "bf16" (bf16 MFMA operands, fp32
    accumulate) is the fast mode; it moves the gradients of this heavily cancelling loss by up to a few percent of
    max|grad| against the fp32 reference and must be asked for explicitly.  "bf16x3" (BilinearCritic only) splits every
    operand into two bf16 parts and spends three bf16 MFMAs per product: fp32-grade gradients at several times the speed
    of "f32".  "fp8" (BilinearCritic only) quantises the embeddings, the weight and T = x W to e4m3 with per-tensor scales
    and runs both forward products on the fp8 MFMA (BASELINE configs[4]); its results match an oracle fed the same
    quantised values, not the fp32 reference.  Embeddings in float64 are cast to float32 (the reference would run them in fp64; this path computes in
    fp32).
    """
    from . import model as _model  # local import: model.py imports nothing from here

    _hip.require_device(embedding_img, "embedding_img")
    _hip.require_device(embedding_txt, "embedding_txt")
    if embedding_img.dtype == torch.float64:
        embedding_img = embedding_img.float()
    if embedding_txt.dtype == torch.float64:
        embedding_txt = embedding_txt.float()
    code = _estimator_code(estimator)
    prec = _precision_code(precision)
    if isinstance(critic, _model.BilinearCritic) and embedding_img.dim() == 2 and embedding_txt.dim() == 2:
        # "f32" on the bilinear critic: fp32-grade results from three bf16 MFMAs per product (_hip.resolve_precision)
        prec = _hip.resolve_precision(precision, True, (embedding_img.shape[0], embedding_img.shape[1], embedding_txt.shape[1]))
    if embedding_img.dim() != 2 or embedding_txt.dim() != 2 or embedding_img.shape[0] != embedding_txt.shape[0]:
        raise ValueError("embedding_img / embedding_txt must be [B, d_img] / [B, d_txt]")
    sid = study_id_codes(study_id, embedding_img.device)
    if sid.numel() != embedding_img.shape[0]:
        raise ValueError("study_id length must equal the batch size")
    if prec in (_hip.MI_PREC_BF16X3, _hip.MI_PREC_FP8) and not isinstance(critic, _model.BilinearCritic):
        raise ValueError(f'precision="{precision}" is implemented for BilinearCritic only')
    if prec in (_hip.MI_PREC_F16, _hip.MI_PREC_F16X3) and isinstance(critic, (_model.BilinearCritic, _model.SeparableCritic)):
        raise ValueError(f'precision="{precision}" is the fp16-operand mode of the make_mlp critic (its generated operand '
                         'relu(U_i + V_j) is formed by packed fp16 arithmetic); use "bf16" for this critic')
    if isinstance(critic, _model.BilinearCritic):
        loss, stats, scores = BilinearCriticFn.apply(embedding_img, embedding_txt, critic.weight, sid, code, prec,
                                                     bool(return_scores))
    elif isinstance(critic, _model.SeparableCritic):
        if return_scores:  # per-pair scores are a diagnostic output: eager projections + the bilinear form with W = None
            a = critic.project_img(embedding_img)
            c = critic.project_txt(embedding_txt)
            loss, stats, scores = BilinearCriticFn.apply(a, c, None, sid, code, prec, True)
        else:
            loss, stats = SeparableCriticFn.apply(embedding_img, embedding_txt, critic.wg, critic.wh, sid, code, prec)
            scores = None
    else:
        w1, b1, w2, b2, w3, b3 = _concat_params(critic)
        # "f32" on the reference's critic: fp32-grade results from the two-part fp16 scheme (_hip.resolve_precision)
        prec = _hip.resolve_precision(precision, False, concat_hidden=(w1.shape[0], w2.shape[0]))
        if w1.shape[1] != embedding_img.shape[1] + embedding_txt.shape[1]:
            raise ValueError(f"critic expects {w1.shape[1]} inputs, embeddings give "
                             f"{embedding_img.shape[1]} + {embedding_txt.shape[1]}")
        loss, stats, scores = ConcatMlpCriticFn.apply(embedding_img, embedding_txt, w1, b1, w2, b2, w3.reshape(-1), b3,
                                                      sid, code, prec, bool(return_scores))
    loss = loss if estimator == "dv" else loss.reshape(())
    out = [loss]
    if return_scores:
        out.append(scores)
    if return_stats:
        out.append(stats)
    return out[0] if len(out) == 1 else tuple(out)
