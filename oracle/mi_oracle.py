"""CPU oracle for the MI-critic hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (torch CPU / numpy, fp32 or fp64) of the reference algorithm for the one
path this repository accelerates.  It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under ``mutual-information-multimodal_amd/``
imports it, and the product path raises when the HIP library is missing instead of falling back to it.

Parity status: PINNED.  ``tests/golden/make_goldens.py`` runs the reference's own functions (in the build
container only, where ``/root/reference`` exists) and commits their outputs as fixtures under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function below against those fixtures.
The reference holds no golden vectors of its own.

Reference code restated here (file:line relative to the reference root):

* ``pair_index``        <- MultiModalManager.create_mi_pairs, mutual_info_img_txt/main_utils.py:80-110
* ``create_mi_pairs``   <- same, main_utils.py:93 (positives) and :99-108 (negatives, gap-major / i-minor)
* ``mlp_forward``       <- make_mlp, mutual_info_img_txt/model.py:18-32 (Linear/ReLU/Linear/ReLU/Linear)
* ``dv_bound_loss``     <- mutual_info_img_txt/mi_critics.py:3-12
* ``infonce_bound_loss``<- mutual_info_img_txt/mi_critics.py:14-23
* ``literal_step``      <- the call site main_utils.py:220-226 (pairs -> critic -> bound -> backward)

The factorised forms (``concat_scores_matrix``, ``bound_from_matrix``) use the identity
``W1 [x_i ; y_j] + b1 = W1x x_i + (W1y y_j + b1)`` and are checked against the literal
form in the tests.  ``bilinear_scores`` / ``separable_scores`` are extensions that have no reference code:
their scorer parity is "unpinned by the reference"; the bound/masking applied on top of them is pinned.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

ESTIMATORS = ("dv", "infonce")


# ----------------------------------------------------------------------------------------------------
# a1: pair enumeration (main_utils.py:88-110)
# ----------------------------------------------------------------------------------------------------
def sid_to_int(study_id: Sequence) -> np.ndarray:
    """Map arbitrary hashable study ids (strings in the reference, model_utils.py:212) to int64 codes so
    that equality is preserved.  The reference only ever compares ids with ``!=`` (main_utils.py:105)."""
    table: Dict[object, int] = {}
    out = np.empty(len(study_id), dtype=np.int64)
    for n, s in enumerate(study_id):
        if torch.is_tensor(s):
            s = s.item()
        out[n] = table.setdefault(s, len(table))
    return out


def pair_index(study_id: Sequence) -> Tuple[np.ndarray, np.ndarray]:
    """(I, J) index arrays of the rows of the reference's ``mi_input`` in reference order.

    Rows 0..B-1 are the positives (r, r) (main_utils.py:93).  Then for gap in 0..B-2, for i in 0..B-1:
    j = i+gap+1 if that is < B else i+gap+1-B; the row is emitted iff study_id[i] != study_id[j]
    (main_utils.py:99-108)."""
    sid = sid_to_int(study_id)
    b = len(sid)
    pos = np.arange(b, dtype=np.int64)
    if b < 2:
        return pos, pos.copy()
    gap = np.arange(b - 1, dtype=np.int64)[:, None]
    i = np.arange(b, dtype=np.int64)[None, :]
    j = (i + gap + 1) % b
    i = np.broadcast_to(i, j.shape)
    keep = sid[i] != sid[j]
    return np.concatenate([pos, i[keep]]), np.concatenate([pos, j[keep]])


def pair_index_loops(study_id: Sequence) -> Tuple[List[int], List[int]]:
    """Pure-Python double loop, line by line as main_utils.py:99-108 (small cases only)."""
    b = len(study_id)
    ii = list(range(b))
    jj = list(range(b))
    for gap in range(b - 1):
        for i in range(b):
            j = i + (gap + 1) if i + (gap + 1) < b else i + (gap + 1) - b
            if study_id[i] != study_id[j]:
                ii.append(i)
                jj.append(j)
    return ii, jj


def create_mi_pairs(embedding_img: torch.Tensor, embedding_txt: torch.Tensor, study_id: Sequence) -> torch.Tensor:
    """[N, d_img + d_txt] critic input in reference row order (main_utils.py:80-110), built with one gather
    instead of one ``torch.cat`` per row."""
    i, j = pair_index(study_id)
    i = torch.from_numpy(i)
    j = torch.from_numpy(j)
    return torch.cat((embedding_img[i], embedding_txt[j]), 1)


# ----------------------------------------------------------------------------------------------------
# a2: concat-MLP critic (model.py:18-32)
# ----------------------------------------------------------------------------------------------------
def mlp_forward(rows: torch.Tensor, params: Sequence[torch.Tensor]) -> torch.Tensor:
    """nn.Sequential(Linear, ReLU, Linear, ReLU, Linear) with params = (W1,b1,W2,b2,W3,b3) in PyTorch
    [out,in] layout.  F.linear is used on purpose (SURVEY.md hazard H3a)."""
    w1, b1, w2, b2, w3, b3 = params
    h = F.relu(F.linear(rows, w1, b1))
    h = F.relu(F.linear(h, w2, b2))
    return F.linear(h, w3, b3)


# ----------------------------------------------------------------------------------------------------
# a3 / a4: bound reductions (mi_critics.py:3-12, 14-23)
# ----------------------------------------------------------------------------------------------------
def dv_bound_loss(discriminator_logits: torch.Tensor, pos_size: int) -> torch.Tensor:
    """mi_critics.py:3-12.  Result shape [1] for [N,1] logits.  The log-N constant is float32 in the
    reference (``.float()``, mi_critics.py:10) whatever the logits dtype."""
    size = discriminator_logits.shape[0]
    pos_energy = torch.mean(discriminator_logits[:pos_size])
    lse = torch.logsumexp(discriminator_logits[pos_size:], dim=0)
    neg_energy = lse - torch.log(torch.tensor(size - pos_size).float())
    return neg_energy - pos_energy


def infonce_bound_loss(discriminator_logits: torch.Tensor, pos_size: int) -> torch.Tensor:
    """mi_critics.py:14-23.  Result shape [] for [N,1] logits (mean of the 1-element logsumexp)."""
    pos_energy = torch.mean(discriminator_logits[:pos_size])
    lse = torch.logsumexp(discriminator_logits[pos_size:], dim=0)
    neg_energy = torch.mean(lse)
    return neg_energy - pos_energy


def bound_loss(logits: torch.Tensor, pos_size: int, estimator: str) -> torch.Tensor:
    if estimator == "dv":
        return dv_bound_loss(logits, pos_size)
    if estimator == "infonce":
        return infonce_bound_loss(logits, pos_size)
    raise ValueError(f"unknown estimator {estimator!r}")


def bound_grad_logits(logits: torch.Tensor, pos_size: int) -> torch.Tensor:
    """Closed form of d loss / d logits: -1/B on positives, softmax over negatives."""
    flat = logits.reshape(-1)
    g = torch.empty_like(flat)
    g[:pos_size] = -1.0 / pos_size
    g[pos_size:] = torch.softmax(flat[pos_size:], dim=0)
    return g.reshape(logits.shape)


# ----------------------------------------------------------------------------------------------------
# a5: the literal call site (main_utils.py:220-226)
# ----------------------------------------------------------------------------------------------------
def literal_step(x: torch.Tensor, y: torch.Tensor, study_id: Sequence, params: Sequence[torch.Tensor],
                 estimator: str, pos_size: Optional[int] = None):
    """pairs -> mi_discriminator -> mi_critic -> backward, exactly as the reference strings them together.
    Returns dict(scores [N], loss, dx, dy, dparams (6 tensors))."""
    x = x.detach().clone().requires_grad_(True)
    y = y.detach().clone().requires_grad_(True)
    params = [p.detach().clone().requires_grad_(True) for p in params]
    mi_input = create_mi_pairs(x, y, study_id)
    mi_output = mlp_forward(mi_input, params)
    loss = bound_loss(mi_output, len(study_id) if pos_size is None else pos_size, estimator)
    loss.sum().backward()
    return {
        "scores": mi_output.detach().reshape(-1),
        "loss": loss.detach(),
        "dx": x.grad,
        "dy": y.grad,
        "dparams": [p.grad for p in params],
    }


# ----------------------------------------------------------------------------------------------------
# B x B matrix forms (what the HIP kernels compute)
# ----------------------------------------------------------------------------------------------------
def negative_mask(study_id: Sequence) -> torch.Tensor:
    """[B,B] bool: True where (i,j) is a negative row of the reference: i != j and sid_i != sid_j."""
    sid = torch.from_numpy(sid_to_int(study_id))
    return sid[:, None] != sid[None, :]  # i == j implies equal ids, so the diagonal is False


def bound_from_matrix(scores: torch.Tensor, study_id: Sequence, estimator: str) -> torch.Tensor:
    """The reference loss evaluated on a [B,B] score matrix S[i,j] = critic(x_i, y_j): positives are the
    diagonal, negatives the masked off-diagonal entries.  Equal to ``bound_loss`` on the reference-ordered
    rows because logsumexp/mean do not depend on row order."""
    b = scores.shape[0]
    neg = negative_mask(study_id)
    pos_energy = torch.diagonal(scores).mean()
    lse = torch.logsumexp(scores[neg], dim=0)
    if estimator == "dv":
        n_neg = int(neg.sum())
        return (lse - torch.log(torch.tensor(n_neg).float()) - pos_energy).reshape(1)
    if estimator == "infonce":
        return lse - pos_energy
    raise ValueError(f"unknown estimator {estimator!r}")


def matrix_to_reference_rows(scores: torch.Tensor, study_id: Sequence) -> torch.Tensor:
    i, j = pair_index(study_id)
    return scores[torch.from_numpy(i), torch.from_numpy(j)]


def concat_scores_matrix(x, y, params, round_fn=None) -> torch.Tensor:
    """Factorised concat-MLP critic: S[i,j] = MLP([x_i ; y_j]) via U_i + V_j.

    ``round_fn`` (optional) is applied at the points where the 16-bit MFMA path rounds: to
    H1 = relu(U_i + V_j) and to W2.  With round_fn=None this is plain fp32/fp64."""
    w1, b1, w2, b2, w3, b3 = params
    dx = x.shape[1]
    u = F.linear(x, w1[:, :dx])
    v = F.linear(y, w1[:, dx:], b1)
    h1 = F.relu(u[:, None, :] + v[None, :, :])  # [B,B,h1]
    w2e = w2
    if round_fn is not None:
        h1 = round_fn(h1)
        w2e = round_fn(w2)
    h2 = F.relu(F.linear(h1, w2e, b2))
    return F.linear(h2, w3, b3).squeeze(-1)


def bilinear_scores(x, y, w, round_fn=None) -> torch.Tensor:
    """Extension (no reference code): S = (x W) y^T.  round_fn mimics 16-bit operand rounding."""
    if round_fn is None:
        return (x @ w) @ y.t()
    t = round_fn(round_fn(x) @ round_fn(w))
    return t @ round_fn(y).t()


def bilinear_step_rounded(x, y, w, study_id: Sequence, estimator: str, row_block: Optional[Tuple[int, int]] = None):
    """Extension (no reference code).  fp64 forward + backward of S = (x W) y^T under the reference bound with the
    rounding points of the 16-bit MFMA path (mi_bilinear_flash.h): x, y, w and T = x W are rounded to bf16; the
    exponentials P = exp(S - lse) are rounded to bf16 before the two B x B gradient contractions; dT is rounded to
    bf16 before dW = x^T dT and dX = dT w^T.  (The kernel exponentiates against a per-wave reference point instead of
    lse; bf16 rounding is scale invariant up to the position of the binade boundaries, which the tolerance covers.)
    ``row_block`` = (r0, r1): only the image rows [r0, r1) contribute (the sharded case): dY is then the partial sum of
    that row block, dX / dW likewise, and the statistics are still those of the full matrix."""
    xb, yb, wb = round_bf16(x.double()), round_bf16(y.double()), round_bf16(w.double())
    tb = round_bf16(xb @ wb)
    s = tb @ yb.t()
    b = s.shape[0]
    neg = negative_mask(study_id)
    lse = torch.logsumexp(s[neg], dim=0)
    loss = bound_from_matrix(s, study_id, estimator)
    p = round_bf16(torch.where(neg, torch.exp(s - lse), torch.zeros_like(s)))
    r0, r1 = (0, b) if row_block is None else row_block
    rows = slice(r0, r1)
    eye = torch.eye(b, dtype=s.dtype)[rows]
    dt = round_bf16(p[rows] @ yb - (eye @ yb) / b)
    dy = p[rows].t() @ tb[rows] - (eye.t() @ tb[rows]) / b
    return {"scores": s, "loss": loss, "dx": dt @ wb.t(), "dy": dy, "dw": xb[rows].t() @ dt, "dt": dt}


# ----------------------------------------------------------------------------------------------------
# fp8 critic (BASELINE configs[4]; SURVEY.md hazard H7: parity is defined on identically quantised inputs)
# ----------------------------------------------------------------------------------------------------
E4M3_MAX = 448.0


def quant_e4m3(v: torch.Tensor) -> torch.Tensor:
    """Round to the OCP e4m3fn grid (what gfx950's v_cvt_pk_fp8_f32 produces for |v| <= 448): 3 mantissa bits, normal
    exponents -6 .. 8, subnormal step 2^-9, round-to-nearest-even, saturating at +-448.  Returns the VALUES in v's dtype
    (every e4m3 value is exact in bf16, fp32 and fp64)."""
    a = v.abs().double()
    e = torch.floor(torch.log2(torch.clamp(a, min=2.0 ** -20)))
    e = torch.clamp(e, min=-6.0, max=8.0)
    step = torch.pow(torch.tensor(2.0, dtype=torch.float64), e - 3.0)
    q = torch.round(a / step) * step          # torch.round is round-half-to-even; a / step is exact (power of two)
    q = torch.clamp(q, max=E4M3_MAX)
    return (torch.sign(v).double() * q).to(v.dtype)


def fp8_scale(t: torch.Tensor) -> torch.Tensor:
    """Per-tensor scale absmax / 448 in fp32 arithmetic, as the library computes it."""
    return (t.float().abs().max() / torch.tensor(E4M3_MAX, dtype=torch.float32)).float()


def bilinear_step_fp8(x, y, w, study_id: Sequence, estimator: str, row_block: Optional[Tuple[int, int]] = None):
    """Extension (no reference code).  fp64 forward + backward of S = (x W) y^T under the reference bound with the
    quantisation and rounding points of the library's fp8 mode (csrc/mi_bilinear.hip, MI_PREC_FP8):

      x, y, w -> e4m3 with per-tensor scales absmax / 448 (scales and the divisions in fp32);
      T = x_q w_q exactly (fp8 MFMA products are exact, fp32 accumulation) -> e4m3 with its own per-tensor scale;
      S = T_q y_q^T; bound and masking as the reference;
      backward on the dequantised operands (straight-through for the quantisers): G = dL/dS rounded to bf16,
      dT = G y_q rounded to bf16 before dW = x_q^T dT and dX = dT w_q^T;  dY = G^T T_q.

    ``row_block = (offset, rows)`` restricts the image rows (a rank's share of a sharded batch): statistics stay global,
    dx / dw / dy are the block's terms."""
    f32 = torch.float32
    sx, sy, sw = fp8_scale(x), fp8_scale(y), fp8_scale(w)
    qx = quant_e4m3((x.to(f32) / sx)).double()
    qy = quant_e4m3((y.to(f32) / sy)).double()
    qw = quant_e4m3((w.to(f32) / sw)).double()
    sx, sy, sw = sx.double(), sy.double(), sw.double()
    t = (sx * sw) * (qx @ qw)
    st = fp8_scale(t)
    qt = quant_e4m3((t.to(f32) / st)).double()
    st = st.double()
    s = (st * sy) * (qt @ qy.t())
    loss = bound_from_matrix(s, study_id, estimator)
    g = matrix_grad_scores(s, study_id)
    off, rows = row_block if row_block is not None else (0, x.shape[0])
    gb = round_bf16(g[off:off + rows])
    dt = round_bf16(sy * (gb @ qy))
    return {"scores": s, "loss": loss, "dx": sw * (dt @ qw.t()), "dy": st * (gb.t() @ qt[off:off + rows]),
            "dw": sx * (qx[off:off + rows].t() @ dt), "scales": (float(sx), float(sy), float(sw), float(st))}


def separable_step_rounded(x, y, wg, wh, study_id: Sequence, estimator: str):
    """Extension (no reference code).  fp64 forward + backward of S = (x Wg)(y Wh)^T under the reference bound with the
    rounding points of the 16-bit path (mi_bilinear.hip, separable form): x, y, Wg, Wh and the projections A = x Wg,
    C = y Wh are rounded to bf16; the exponentials P = exp(S - lse) are rounded to bf16 before the two B x B
    contractions; dA and dC are rounded to bf16 before the projection gradients."""
    xb, yb, gb, hb = (round_bf16(t.double()) for t in (x, y, wg, wh))
    a = round_bf16(xb @ gb)
    c = round_bf16(yb @ hb)
    s = a @ c.t()
    b = s.shape[0]
    neg = negative_mask(study_id)
    lse = torch.logsumexp(s[neg], dim=0)
    loss = bound_from_matrix(s, study_id, estimator)
    p = round_bf16(torch.where(neg, torch.exp(s - lse), torch.zeros_like(s)))
    da = round_bf16(p @ c - c / b)
    dc = round_bf16(p.t() @ a - a / b)
    return {"scores": s, "loss": loss, "dx": da @ gb.t(), "dy": dc @ hb.t(), "dwg": xb.t() @ da, "dwh": yb.t() @ dc}


def separable_scores(x, y, wg, wh) -> torch.Tensor:
    """Extension (no reference code): S = (x Wg)(y Wh)^T."""
    return (x @ wg) @ (y @ wh).t()


def round_bf16(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(t.dtype)


def matrix_step(scores_fn, leaves: Sequence[torch.Tensor], study_id: Sequence, estimator: str):
    """Autograd through a [B,B] score function; returns scores, loss and grads of ``leaves``."""
    leaves = [t.detach().clone().requires_grad_(True) for t in leaves]
    s = scores_fn(*leaves)
    loss = bound_from_matrix(s, study_id, estimator)
    loss.sum().backward()
    return {"scores": s.detach(), "loss": loss.detach(), "grads": [t.grad for t in leaves]}


def concat_matrix_step(x, y, study_id, params, estimator: str, round_fn=None, row_block: int = 0):
    """Full fwd+bwd of the factorised concat-MLP critic.  ``row_block`` > 0 processes i-rows in blocks to
    bound memory (two passes: LSE first, then gradients) -- used by bench.py's cpu_baseline at large B."""
    if row_block <= 0 or row_block >= x.shape[0]:
        out = matrix_step(lambda a, b, *p: concat_scores_matrix(a, b, p, round_fn), [x, y, *params],
                          study_id, estimator)
        return {"scores": out["scores"], "loss": out["loss"], "dx": out["grads"][0], "dy": out["grads"][1],
                "dparams": out["grads"][2:]}
    return _concat_matrix_step_blocked(x, y, study_id, params, estimator, row_block, round_fn)


def _concat_matrix_step_blocked(x, y, study_id, params, estimator, rb, round_fn=None):
    b = x.shape[0]
    neg = negative_mask(study_id)
    n_neg = int(neg.sum())
    with torch.no_grad():
        blocks = [concat_scores_matrix(x[s:s + rb], y, params, round_fn) for s in range(0, b, rb)]
        s_all = torch.cat(blocks, 0)
        lse = torch.logsumexp(s_all[neg], dim=0)
        pos = torch.diagonal(s_all).mean()
    loss = lse - pos
    if estimator == "dv":
        loss = (loss - math.log(float(n_neg))).reshape(1)
    g = torch.where(neg, torch.exp(s_all - lse), torch.zeros_like(s_all))
    g = g - torch.eye(b, dtype=s_all.dtype) / b
    xl = x.detach().clone().requires_grad_(True)
    yl = y.detach().clone().requires_grad_(True)
    pl = [p.detach().clone().requires_grad_(True) for p in params]
    for s in range(0, b, rb):
        sb = concat_scores_matrix(xl[s:s + rb], yl, pl, round_fn)
        (sb * g[s:s + rb]).sum().backward()
    return {"scores": s_all, "loss": loss, "dx": xl.grad, "dy": yl.grad, "dparams": [p.grad for p in pl]}


def _scores_pass(block_scores, b, row_block, r0, r1, scores_outside, progress, dtype):
    """[B, B] scores by row blocks; rows outside [r0, r1) from ``scores_outside`` when given."""
    if scores_outside is None:
        lo, hi = 0, b
        s_all = torch.empty(b, b, dtype=dtype)
    else:
        lo, hi = r0, r1
        s_all = scores_outside.to(dtype).clone()
    for s in range(lo, hi, row_block):
        e = min(s + row_block, hi)
        s_all[s:e] = block_scores(s)[:e - s]
        if progress is not None and ((s - lo) // row_block) % 16 == 0:
            progress(f"oracle scores: row {s} of [{lo}, {hi})")
    return s_all


def concat_step_rounded(x, y, study_id, params, estimator: str, row_block: int = 64, round_fn=None, rows=None,
                        progress=None, scores_outside=None):
    """fp64 forward + backward of the factorised concat-MLP critic with ALL rounding points of the 16-bit MFMA path
    (csrc/mi_concat_fwd_dma.h, csrc/mi_concat_bwd.h), the backward in closed form instead of autograd:

      forward   H1 = bf16(relu(U_i + V_j)), W2 -> bf16 (as ``concat_scores_matrix(round_fn=...)``);
      dU / dV   E[p, k] = sum_n M[p, n] bf16(w3[n] W2[n, k])               (prep_w2w_kernel rounds the PRODUCT)
      dW2 ...   D[n, k] = sum_p M[p, n] bf16(g_p relu(U_i[k] + V_j[k]))    (concat_bwd_dw2_kernel rounds the PRODUCT)
      everything else (first layer, bound, g, the finishing sums) exact.

    With round_fn = identity this equals ``concat_matrix_step`` (checked in tests/test_oracle_golden.py), which pins the
    closed form; with round_fn = round_bf16 (default) it is what the bf16 kernels compute up to fp32 accumulation order.
    Rows are processed ``row_block`` at a time ([rb, B, h1] fp64 temporaries).  ``rows = (r0, r1)``: only the image rows
    [r0, r1) contribute to the gradients (a rank's share of a sharded batch: dx rows of the block, the block's partial dy
    and parameter gradients; scores, loss and g are those of the whole batch).  ``progress`` (optional callable) is called
    with a short string now and then (long runs on a harness that kills silent commands).  ``scores_outside`` ([B, B],
    with ``rows``): the scores of the rows OUTSIDE [r0, r1) are taken from it instead of being recomputed (three quarters
    of the host time at B = 4096; the caller checks those scores elsewhere); the block's own rows are always computed."""
    rf = round_bf16 if round_fn is None else round_fn
    w1, b1, w2, b2, w3, b3 = [p.detach() for p in params]
    x, y = x.detach(), y.detach()
    b, dx = x.shape
    w3v = w3.reshape(-1)
    u = F.linear(x, w1[:, :dx])
    v = F.linear(y, w1[:, dx:], b1)
    w2r = rf(w2)
    w2w = rf(w2 * w3v[:, None])  # [h2, h1]

    def block(s):
        pre = u[s:s + row_block, None, :] + v[None, :, :]
        h1 = F.relu(pre)
        z2 = F.linear(rf(h1), w2r, b2)
        return pre, h1, z2

    r0, r1 = (0, b) if rows is None else rows
    s_all = _scores_pass(lambda s: F.linear(F.relu(block(s)[2]), w3, b3).squeeze(-1), b, row_block, r0, r1, scores_outside,
                         progress, x.dtype)
    loss = bound_from_matrix(s_all, study_id, estimator)
    g = matrix_grad_scores(s_all, study_id)
    du = torch.zeros_like(u)
    dv = torch.zeros_like(v)
    dmat = torch.zeros_like(w2)
    mvec = torch.zeros_like(b2)
    for s in range(r0, r1, row_block):
        e = min(s + row_block, r1)
        pre, h1, z2 = block(s)
        pre, h1, z2 = pre[:e - s], h1[:e - s], z2[:e - s]
        gb = g[s:e, :, None]
        m = (z2 > 0).to(x.dtype)
        dh1 = gb * (pre > 0).to(x.dtype) * (m @ w2w)
        du[s:e] = dh1.sum(1)
        dv += dh1.sum(0)
        hf = rf(h1 * gb)
        dmat += torch.einsum("ijn,ijk->nk", m, hf)
        mvec += (gb * m).sum((0, 1))
        if progress is not None and ((s - r0) // row_block) % 8 == 0:
            progress(f"oracle gradients: row {s} of [{r0}, {r1})")
    dparams = [torch.cat([du.t() @ x, dv.t() @ y], 1), dv.sum(0), w3v[:, None] * dmat, w3v * mvec,
               ((w2 * dmat).sum(1) + b2 * mvec).reshape(w3.shape), g[r0:r1].sum().reshape(b3.shape)]
    return {"scores": s_all, "loss": loss, "dx": (du @ w1[:, :dx])[r0:r1], "dy": dv @ w1[:, dx:], "dparams": dparams}


def round_f16(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.float16).to(t.dtype)


def f16_pow2_scale(a: float, lo_exp: int) -> float:
    """2^e with 2^e * a in [2^lo_exp, 2^(lo_exp + 1)) for finite a > 0, else 1 (csrc/mi_concat_bwd.h f16_pow2_scale)."""
    if not (a > 0.0) or math.isinf(a):
        return 1.0
    _, e = math.frexp(a)
    return 2.0 ** (lo_exp + 1 - e)


def concat_step_f16(x, y, study_id, params, estimator: str, row_block: int = 64, rows=None, progress=None,
                    scores_outside=None):
    """fp64 forward + backward of the factorised concat-MLP critic with the rounding points of the library's fp16 mode
    (csrc/mi_concat_f16.h, MI_PREC_F16), the backward in closed form:

      scales    s_uv, s_w, s_ww: powers of two from the absmax of U, V, W2, w3 (f16_pow2_scale); s_g: the power of two that
                puts the batch's largest |g| = max(exp(neg_max - lse), 1 / B) into [2^13, 2^14)
      forward   Uh = fp16(U s_uv), Vh = fp16(V s_uv); h = fp16(clamp(Uh_i + Vh_j, 0, 1)) (packed fp16 add with the clamp
                modifier = relu); W2h = fp16(W2 s_w);  Z2 = (h W2h^T) / (s_uv s_w) + b2;  scores from Z2 unrounded
      dU / dV   E[p, k] = sum_n M[p, n] fp16(w3[n] W2[n, k] s_ww) / s_ww; relu' of layer 1 decided on Uh_i + Vh_j > 0
      dW2 ...   D[n, k] = sum_p M[p, n] fp16(fp16(g_p s_g) h_pk) / (s_uv s_g)
      everything else (first layer, bound, g, the finishing sums) exact.

    The absmax values are taken from fp32 U, V as the library does (its first layer is an exact-fp32 GEMM).  ``rows`` /
    ``progress`` as in ``concat_step_rounded``; with ``rows`` the scale s_uv is taken from the U rows of the block, as a
    rank of a sharded run does."""
    w1, b1, w2, b2, w3, b3 = [p.detach() for p in params]
    x, y = x.detach(), y.detach()
    b, dx = x.shape
    w3v = w3.reshape(-1)
    u = F.linear(x, w1[:, :dx])
    v = F.linear(y, w1[:, dx:], b1)
    f32max = lambda t: float(t.float().abs().max())  # noqa: E731
    r0, r1 = (0, b) if rows is None else rows
    s_uv = f16_pow2_scale(float(torch.tensor(f32max(u[r0:r1]), dtype=torch.float32) + torch.tensor(f32max(v), dtype=torch.float32)), -2)
    s_w = f16_pow2_scale(f32max(w2), 13)
    s_ww = f16_pow2_scale(float(torch.tensor(f32max(w2), dtype=torch.float32) * torch.tensor(f32max(w3v), dtype=torch.float32)), 13)
    uh, vh = round_f16(u * s_uv), round_f16(v * s_uv)
    w2h = round_f16(w2 * s_w)
    w2w = round_f16(w2 * w3v[:, None] * s_ww) / s_ww  # [h2, h1]

    def block(s):
        pre = uh[s:s + row_block, None, :] + vh[None, :, :]
        h = round_f16(torch.clamp(pre, 0.0, 1.0))
        z2 = F.linear(h, w2h) / (s_uv * s_w) + b2
        return pre, h, z2

    s_all = _scores_pass(lambda s: F.linear(F.relu(block(s)[2]), w3, b3).squeeze(-1), b, row_block, r0, r1, scores_outside,
                         progress, x.dtype)
    loss = bound_from_matrix(s_all, study_id, estimator)
    g = matrix_grad_scores(s_all, study_id)
    negm = negative_mask(study_id)
    g_max = max(float(g[negm].max()) if bool(negm.any()) else 0.0, 1.0 / b)
    s_g = f16_pow2_scale(float(torch.tensor(g_max, dtype=torch.float32)), 13)
    du = torch.zeros_like(u)
    dv = torch.zeros_like(v)
    dmat = torch.zeros_like(w2)
    mvec = torch.zeros_like(b2)
    for s in range(r0, r1, row_block):
        e = min(s + row_block, r1)
        pre, h, z2 = block(s)
        pre, h, z2 = pre[:e - s], h[:e - s], z2[:e - s]
        gb = g[s:e, :, None]
        m = (z2 > 0).to(x.dtype)
        dh1 = gb * (pre > 0).to(x.dtype) * (m @ w2w)
        du[s:e] = dh1.sum(1)
        dv += dh1.sum(0)
        hf = round_f16(round_f16(gb * s_g) * h)
        dmat += torch.einsum("ijn,ijk->nk", m, hf) / (s_uv * s_g)
        mvec += (gb * m).sum((0, 1))
        if progress is not None and ((s - r0) // row_block) % 8 == 0:
            progress(f"oracle gradients: row {s} of [{r0}, {r1})")
    dparams = [torch.cat([du.t() @ x, dv.t() @ y], 1), dv.sum(0), w3v[:, None] * dmat, w3v * mvec,
               ((w2 * dmat).sum(1) + b2 * mvec).reshape(w3.shape), g[r0:r1].sum().reshape(b3.shape)]
    return {"scores": s_all, "loss": loss, "dx": (du @ w1[:, :dx])[r0:r1], "dy": dv @ w1[:, dx:], "dparams": dparams,
            "scales": (s_uv, s_w, s_ww, s_g)}


def concat_relu_flip_budget(x, y, params, margin: float, round_fn=None):
    """How far can the choice of relu'(0) move the gradients?  A second-layer pre-activation within the forward's own
    rounding noise of zero comes out on either side in two correct implementations (the 16-bit path rounds H1 to bf16:
    an fp32 and an fp64 first layer round a few H1 elements to neighbouring bf16 values, which moves Z2 by up to
    ulp(H1) * |W2| ~ 1e-4), and its sign bit M switches a whole term of the backward on or off.  For the POSITIVE pairs
    (weight 1/B each; a negative pair weighs ~1/B^2 and is invisible at the test tolerances) this returns elementwise
    bounds on the movement of dx, dy, dW2 and db2 if every unit n of pair (i, i) with |Z2| < ``margin`` flipped:

        dU_i[k] moves by g_ii A[i, k] w3[n] W2[n, k]  ->  dx row i, dy row i through W1
        D[n, k] moves by g_ii H1[i, k]                ->  dW2 row n;   db2[n] by g_ii w3[n]        (g_ii = -1/B)

    Test infrastructure: the tests add these budgets to their absolute tolerance and print how many units are affected."""
    rf = (lambda t: t) if round_fn is None else round_fn
    w1, b1, w2, b2, w3, _ = [p.detach() for p in params]
    b, dx = x.shape
    w3v = w3.reshape(-1)
    pre = F.linear(x, w1[:, :dx]) + F.linear(y, w1[:, dx:], b1)  # pair (i, i)
    h1 = F.relu(pre)
    z2 = F.linear(rf(h1), rf(w2), b2)
    amb = z2.abs() < margin                                     # [B, h2]
    w2w = w2 * w3v[:, None]                                     # [h2, h1]
    bdx, bdy = torch.zeros_like(x), torch.zeros_like(y)
    for i, n in torch.nonzero(amb).tolist():                    # one exact movement vector per candidate flip
        du = (pre[i] > 0).to(x.dtype) * w2w[n] / b
        bdx[i] += (du @ w1[:, :dx]).abs()
        bdy[i] += (du @ w1[:, dx:]).abs()
    ambf = amb.to(x.dtype)
    return {"n_units": int(amb.sum()), "dx": bdx, "dy": bdy, "dw2": w3v.abs()[:, None] * (ambf.t() @ h1) / b,
            "db2": w3v.abs() * ambf.sum(0) / b}


# ----------------------------------------------------------------------------------------------------
# closed-form backward of the factorised concat-MLP critic -- used to check autograd-free
# ----------------------------------------------------------------------------------------------------
def matrix_grad_scores(scores: torch.Tensor, study_id: Sequence) -> torch.Tensor:
    """G[i,j] = d loss / d S[i,j]: exp(S - LSE_neg) on negatives, -1/B on the diagonal, 0 on dropped pairs."""
    b = scores.shape[0]
    neg = negative_mask(study_id)
    lse = torch.logsumexp(scores[neg], dim=0)
    g = torch.where(neg, torch.exp(scores - lse), torch.zeros_like(scores))
    return g - torch.eye(b, dtype=scores.dtype) / b


# ----------------------------------------------------------------------------------------------------
# deterministic, RNG-free synthetic data (bit-reproducible anywhere: integer hash -> exact float division)
# ----------------------------------------------------------------------------------------------------
def hash_uniform(shape: Sequence[int], salt: int, dtype=torch.float32) -> torch.Tensor:
    """Values in [-0.5, 0.5) from a 32-bit integer hash of the flat index; exact in fp32 and fp64."""
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    h = (idx * np.uint64(2654435761) + np.uint64(salt) * np.uint64(40503) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    h = (h >> np.uint64(8)).astype(np.float64)  # 24 bits -> exact in fp32
    v = h / float(1 << 24) - 0.5
    return torch.from_numpy(v.reshape(tuple(shape))).to(dtype)


def synthetic_case(b: int, d_img: int, d_txt: int, h1: int = 1024, h2: int = 512, salt: int = 0,
                   dup: bool = False, dtype=torch.float32):
    """Closed-form inputs + critic parameters (make_mlp(d_img+d_txt,[h1,h2]) shapes, model.py:18-32)."""
    x = hash_uniform((b, d_img), salt * 16 + 1, dtype) * 2.0
    y = hash_uniform((b, d_txt), salt * 16 + 2, dtype) * 2.0
    d = d_img + d_txt
    w1 = hash_uniform((h1, d), salt * 16 + 3, dtype) * (2.0 / math.sqrt(d))
    b1 = hash_uniform((h1,), salt * 16 + 4, dtype) * (2.0 / math.sqrt(d))
    w2 = hash_uniform((h2, h1), salt * 16 + 5, dtype) * (2.0 / math.sqrt(h1))
    b2 = hash_uniform((h2,), salt * 16 + 6, dtype) * (2.0 / math.sqrt(h1))
    w3 = hash_uniform((1, h2), salt * 16 + 7, dtype) * (2.0 / math.sqrt(h2)) * 8.0
    b3 = hash_uniform((1,), salt * 16 + 8, dtype) * (2.0 / math.sqrt(h2))
    sid = [str(50000000 + n) for n in range(b)]
    if dup:  # SURVEY.md 8d "duplicates variant": sid_i = i - (i mod 2) for i < max(B/8, 4)
        for n in range(max(b // 8, min(b, 4))):
            sid[n] = str(50000000 + n - (n % 2))
    return x, y, sid, [w1, b1, w2, b2, w3, b3]


# ------------------------------------------------------------------------------------------------ trainer pieces (f1)
def adamw_step(p, g, m, v, t, lr, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
    """One step of pytorch-transformers==1.0.0 ``optimization.AdamW`` (the optimiser of the reference's text encoder,
    main_utils.py:166-168; the package is absent here, this restates its published algorithm) on numpy float64 arrays.
    ``t`` is the 1-based step count.  Returns the new (p, m, v)."""
    b1, b2 = betas
    m = b1 * m + (1.0 - b1) * g
    v = b2 * v + (1.0 - b2) * g * g
    step = lr
    if correct_bias:
        step = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    p = p - step * m / (np.sqrt(v) + eps)
    if weight_decay > 0.0:
        p = p - lr * weight_decay * p
    return p, m, v


def warmup_linear(step, warmup_steps, t_total):
    """Multiplier of pytorch-transformers==1.0.0 ``WarmupLinearSchedule`` (main_utils.py:170-172)."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(t_total - step) / float(max(1.0, t_total - warmup_steps)))
