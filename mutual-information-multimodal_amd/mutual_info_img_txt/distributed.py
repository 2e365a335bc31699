"""Global-batch negatives across the GPUs of one node (one process per GPU, torch.distributed over RCCL/xGMI).

The reference is single-process; "global batch" is defined as the reference loss at B = B_global
(SURVEY.md 8e).  The B x B score matrix is sharded by row blocks: rank g owns samples [g*B/G, (g+1)*B/G).

forward : all-gather the text embeddings and study ids (the only bulk exchange: B/G x d floats per rank), score the
          local row block against all columns with the fused kernel, all-gather one 32-byte partial record per rank
          (max, rescaled sum, positive sum, negative count) and merge the records IN RANK ORDER on every rank, so all
          ranks hold bit-identical statistics and loss.
backward: local kernel -> dX of the row block is complete; dY is a partial over this row block for ALL columns ->
          reduce-scatter; critic-parameter gradients -> all-reduce (sum: the loss is one global scalar).

The local compute is an ``ops`` object.  The product default runs the HIP kernels through the C ABI; the CPU tests
(gloo, world_size 2) inject an oracle-backed ops object to check the exchange logic without a GPU.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from . import _hip


# ----------------------------------------------------------------------------------------------------------
# local ops through the C ABI (the product path)
# ----------------------------------------------------------------------------------------------------------
class HipBilinearOps:
    """S = (X W) Y^T row block; params = [W]."""
    n_params = 1

    def __init__(self):
        self._fp8_ws = None    # workspace of a staged fp8 preparation, handed on to forward()
        self._local_ws = None  # workspace in which prep_local() prepared the rank's own part, handed on to forward()
        self._local_key = None

    def prep_local(self, x, params, b, precision) -> bool:
        """The part of the forward that needs neither the gathered text embeddings nor the gathered ids -- bf16 copies of
        X and W, T = X W (mi_bilinear_prep_local) -- issued while the all-gather is in flight.  False where the shape or
        precision does not take the fused kernels (forward() then does everything, as before)."""
        lib = _hip.load()
        (w,) = params
        br, dx = x.shape
        dy = w.shape[1]
        if precision == _hip.MI_PREC_FP8:
            return False
        ws = _hip.workspace(lib.mi_bilinear_workspace_bytes(br, b, dx, dy, precision), x.device)
        with torch.cuda.device(x.device):
            rc = lib.mi_bilinear_prep_local(x.data_ptr(), w.data_ptr(), br, b, dx, dy, precision, ws.data_ptr(), ws.numel(),
                                            torch.cuda.current_stream(x.device).cuda_stream)
        if rc == _hip.MI_ESHAPE:
            return False
        _hip.check(rc, "mi_bilinear_prep_local")
        self._local_ws = ws
        self._local_key = (x.data_ptr(), w.data_ptr(), br, b, dx, dy, precision)
        return True

    def _take_local_ws(self, x, w, br, b, dx, dy, precision):
        """The workspace prep_local() filled -- only for the call it was made for (same tensors, shapes, precision)."""
        ws, self._local_ws = self._local_ws, None
        if ws is not None and self._local_key != (x.data_ptr(), w.data_ptr(), br, b, dx, dy, precision):
            return None  # a prep_local() whose forward never came: not this call's
        return ws

    def fp8_stage(self, stage, x, y_all, params, amax):
        """One stage of the fp8 mode's preparation (mi_bilinear_fp8_stage): ``amax`` (4 floats on the device: x, y, W, T)
        is MAX-all-reduced by the caller between the stages, so that every rank quantises with the whole batch's scales."""
        lib = _hip.load()
        (w,) = params
        br, dx = x.shape
        b, dy = y_all.shape
        if stage == 0:
            self._fp8_ws = _hip.workspace(lib.mi_bilinear_workspace_bytes(br, b, dx, dy, _hip.MI_PREC_FP8), x.device)
        ws = self._fp8_ws
        _hip.call("mi_bilinear_fp8_stage", x.device, x.data_ptr(), y_all.data_ptr(), w.data_ptr(), br, b, dx, dy, int(stage),
                  amax.data_ptr(), ws.data_ptr(), ws.numel())

    def forward(self, x, y_all, params, sid_rows, sid_all, row_offset, estimator, precision, need_grad):
        lib = _hip.load()
        (w,) = params
        br, dx = x.shape
        b, dy = y_all.shape
        dev = x.device
        staged = precision == _hip.MI_PREC_FP8 and self._fp8_ws is not None
        if staged:
            ws, self._fp8_ws = self._fp8_ws, None
            need_grad = int(bool(need_grad)) | 2  # bit 1: the fp8 operands are staged in this workspace
        else:
            ws = self._take_local_ws(x, w, br, b, dx, dy, precision)
            if ws is not None:
                need_grad = int(bool(need_grad)) | 4  # bit 2: prep_local() already ran in this workspace
            else:
                ws = _hip.workspace(lib.mi_bilinear_workspace_bytes(br, b, dx, dy, precision), dev)
        stats = _hip.new_stats(dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        record = torch.empty(_hip.RECORD_FLOATS, dtype=torch.float32, device=dev)
        _hip.call("mi_bilinear_fwd", dev, x.data_ptr(), y_all.data_ptr(), w.data_ptr(), sid_rows.data_ptr(),
                                       sid_all.data_ptr(), br, b, row_offset, dx, dy, estimator, precision,
                                       int(need_grad), loss.data_ptr(), stats.data_ptr(), record.data_ptr(), None,
                                       ws.data_ptr(),
                                       ws.numel())
        return record, (x, y_all, w, sid_rows, sid_all, row_offset, precision, ws)

    def forward_raw(self, x, y_all, params, sid_rows, sid_all, row_offset, estimator, precision):
        """Forward WITHOUT the finalize launch (mi_bilinear_fwd, need_grad bit 3): returns the fused kernel's per-wave
        records ([n, 4] float32, a view of the workspace) for the caller to all-gather, or None where the shape does not take
        that path.  `merge_backward` then merges the gathered records inside the backward's first launch."""
        lib = _hip.load()
        (w,) = params
        br, dx = x.shape
        b, dy = y_all.shape
        off = ctypes.c_size_t(0)
        n = lib.mi_bilinear_raw_records(br, b, dx, dy, precision, ctypes.byref(off))
        if n == 0:
            return None
        dev = x.device
        flags = 1 | 8
        ws = self._take_local_ws(x, w, br, b, dx, dy, precision)
        if ws is not None:
            flags |= 4
        else:
            ws = _hip.workspace(lib.mi_bilinear_workspace_bytes(br, b, dx, dy, precision), dev)
        stats = _hip.new_stats(dev)  # (untouched by this call; the C ABI wants a valid pointer)
        _hip.call("mi_bilinear_fwd", dev, x.data_ptr(), y_all.data_ptr(), w.data_ptr(), sid_rows.data_ptr(),
                  sid_all.data_ptr(), br, b, row_offset, dx, dy, estimator, precision, flags, None, stats.data_ptr(), None,
                  None, ws.data_ptr(), ws.numel())
        records = ws[off.value:off.value + 16 * n].view(torch.float32).view(n, 4)
        return records, (x, y_all, w, sid_rows, sid_all, row_offset, precision, ws)

    def merge_backward(self, saved, records_all, n_pos, estimator, grad_out, out=None):
        """mi_bilinear_bwd_records: merge of the gathered raw records (rank order), loss, statistics and all gradients in
        the backward's two launches."""
        x, y_all, w, sid_rows, sid_all, row_offset, precision, ws = saved
        br, dx = x.shape
        b, dy = y_all.shape
        dev = x.device
        if out is None:
            gx, gy, gw = torch.empty_like(x), torch.empty_like(y_all), torch.empty_like(w)
        else:
            gx, gy, (gw,) = out
        stats = _hip.new_stats(dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        _hip.call("mi_bilinear_bwd_records", dev, x.data_ptr(), y_all.data_ptr(), w.data_ptr(), sid_rows.data_ptr(),
                  sid_all.data_ptr(), br, b, row_offset, dx, dy, precision, estimator, records_all.data_ptr(),
                  records_all.shape[0], n_pos, grad_out.data_ptr(), loss.data_ptr(), stats.data_ptr(), gx.data_ptr(),
                  gy.data_ptr(), gw.data_ptr(), ws.data_ptr(), ws.numel())
        return loss, stats, gx, gy, [gw]

    def merge_backward_tail(self, saved, records_all, n_pos, estimator, grad_out, out=None):
        """The first launch of merge_backward only (mi_bilinear_bwd_records with grad_w = NULL): statistics, loss, grad_x and
        the partial grad_y.  The caller starts the reduce-scatter of grad_y and then calls ``backward_dw``."""
        x, y_all, w, sid_rows, sid_all, row_offset, precision, ws = saved
        br, dx = x.shape
        b, dy = y_all.shape
        dev = x.device
        if out is None:
            gx, gy = torch.empty_like(x), torch.empty_like(y_all)
        else:
            gx, gy, _ = out
        stats = _hip.new_stats(dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        _hip.call("mi_bilinear_bwd_records", dev, x.data_ptr(), y_all.data_ptr(), w.data_ptr(), sid_rows.data_ptr(),
                  sid_all.data_ptr(), br, b, row_offset, dx, dy, precision, estimator, records_all.data_ptr(),
                  records_all.shape[0], n_pos, grad_out.data_ptr(), loss.data_ptr(), stats.data_ptr(), gx.data_ptr(),
                  gy.data_ptr(), None, ws.data_ptr(), ws.numel())
        return loss, stats, gx, gy

    def backward_dw(self, saved, out=None):
        """dW = X^T dT (mi_bilinear_bwd_dw) from the workspace merge_backward_tail left."""
        x, y_all, w, _sr, _sa, _ro, precision, ws = saved
        br, dx = x.shape
        b, dy = y_all.shape
        gw = torch.empty_like(w) if out is None else out[2][0]
        _hip.call("mi_bilinear_bwd_dw", x.device, br, b, dx, dy, precision, gw.data_ptr(), ws.data_ptr(), ws.numel())
        return [gw]

    def merge(self, records, n_pos, estimator):
        lib = _hip.load()
        dev = records.device
        stats = _hip.new_stats(dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        _hip.call("mi_merge_partials", dev, records.data_ptr(), records.shape[0], n_pos, estimator, loss.data_ptr(),
                                         stats.data_ptr())
        return loss, stats

    def backward(self, saved, stats, grad_out, out=None):
        """``out`` = (grad_x, grad_y_partial, [grad_params...]) preallocated buffers (e.g. views of one flat all-reduce
        buffer); allocated here when None."""
        lib = _hip.load()
        x, y_all, w, sid_rows, sid_all, row_offset, precision, ws = saved
        br, dx = x.shape
        b, dy = y_all.shape
        if out is None:
            gx, gy, gw = torch.empty_like(x), torch.empty_like(y_all), torch.empty_like(w)
        else:
            gx, gy, (gw,) = out
        _hip.call("mi_bilinear_bwd", x.device, x.data_ptr(), y_all.data_ptr(), w.data_ptr(), sid_rows.data_ptr(),
                                       sid_all.data_ptr(), br, b, row_offset, dx, dy, precision, stats.data_ptr(),
                                       grad_out.data_ptr(), gx.data_ptr(), gy.data_ptr(), gw.data_ptr(), ws.data_ptr(),
                                       ws.numel(), 1)
        return gx, gy, [gw]


class HipSeparableOps:
    """S = (X Wg)(Y Wh)^T row block (BASELINE.json configs[1]); params = [Wg, Wh].  Every rank projects ALL text rows
    (B d k flops, small beside the B^2 stage); d(Wh) and dY are partials over the row block like the bilinear dY."""
    n_params = 2

    def forward(self, x, y_all, params, sid_rows, sid_all, row_offset, estimator, precision, need_grad):
        lib = _hip.load()
        wg, wh = params
        br, dx = x.shape
        b, dy = y_all.shape
        k = wg.shape[1]
        dev = x.device
        ws = _hip.workspace(lib.mi_separable_workspace_bytes(br, b, dx, dy, k, precision), dev)
        stats = _hip.new_stats(dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        record = torch.empty(_hip.RECORD_FLOATS, dtype=torch.float32, device=dev)
        _hip.call("mi_separable_fwd", dev, x.data_ptr(), y_all.data_ptr(), wg.data_ptr(), wh.data_ptr(), sid_rows.data_ptr(),
                  sid_all.data_ptr(), br, b, row_offset, dx, dy, k, estimator, precision, int(bool(need_grad)),
                  loss.data_ptr(), stats.data_ptr(), record.data_ptr(), ws.data_ptr(), ws.numel())
        return record, (x, y_all, wg, wh, sid_rows, sid_all, row_offset, precision, ws)

    def backward(self, saved, stats, grad_out, out=None):
        x, y_all, wg, wh, sid_rows, sid_all, row_offset, precision, ws = saved
        br, dx = x.shape
        b, dy = y_all.shape
        k = wg.shape[1]
        if out is None:
            gx, gy, gg, gh = (torch.empty_like(t) for t in (x, y_all, wg, wh))
        else:
            gx, gy, (gg, gh) = out
        _hip.call("mi_separable_bwd", x.device, x.data_ptr(), y_all.data_ptr(), wg.data_ptr(), wh.data_ptr(),
                  sid_rows.data_ptr(), sid_all.data_ptr(), br, b, row_offset, dx, dy, k, precision, stats.data_ptr(),
                  grad_out.data_ptr(), gx.data_ptr(), gy.data_ptr(), gg.data_ptr(), gh.data_ptr(), ws.data_ptr(), ws.numel(),
                  1)
        return gx, gy, [gg, gh]

    merge = HipBilinearOps.merge


class HipConcatMlpOps:
    """S[i,j] = MLP([x_i ; y_j]) row block; params = [W1, b1, W2, b2, w3 (flat), b3]."""
    n_params = 6

    def forward(self, x, y_all, params, sid_rows, sid_all, row_offset, estimator, precision, need_grad):
        lib = _hip.load()
        br, dx = x.shape
        b, dy = y_all.shape
        h1, h2 = params[0].shape[0], params[2].shape[0]
        dev = x.device
        ws = _hip.workspace(lib.mi_concat_mlp_workspace_bytes(br, b, dx, dy, h1, h2, precision, int(need_grad)), dev)
        stats = _hip.new_stats(dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        record = torch.empty(_hip.RECORD_FLOATS, dtype=torch.float32, device=dev)
        scores = torch.empty(br, b, dtype=torch.float32, device=dev)
        _hip.call("mi_concat_mlp_fwd", dev, x.data_ptr(), y_all.data_ptr(), *[p.data_ptr() for p in params],
                                         sid_rows.data_ptr(), sid_all.data_ptr(), br, b, row_offset, dx, dy, h1, h2,
                                         estimator, precision, int(need_grad), loss.data_ptr(), stats.data_ptr(),
                                         record.data_ptr(), scores.data_ptr(), ws.data_ptr(), ws.numel())
        return record, (x, y_all, list(params), sid_rows, sid_all, row_offset, precision, scores, ws)

    merge = HipBilinearOps.merge

    def backward(self, saved, stats, grad_out, out=None):
        lib = _hip.load()
        x, y_all, params, sid_rows, sid_all, row_offset, precision, scores, ws = saved
        br, dx = x.shape
        b, dy = y_all.shape
        h1, h2 = params[0].shape[0], params[2].shape[0]
        if out is None:
            gx, gy = torch.empty_like(x), torch.empty_like(y_all)
            gp = [torch.empty_like(p) for p in params]
        else:
            gx, gy, gp = out
        _hip.call("mi_concat_mlp_bwd", x.device, x.data_ptr(), y_all.data_ptr(), *[p.data_ptr() for p in params],
                                         sid_rows.data_ptr(), sid_all.data_ptr(), br, b, row_offset, dx, dy, h1, h2,
                                         precision, stats.data_ptr(), grad_out.data_ptr(), scores.data_ptr(),
                                         gx.data_ptr(), gy.data_ptr(), *[g.data_ptr() for g in gp], ws.data_ptr(),
                                         ws.numel())
        return gx, gy, gp


# ----------------------------------------------------------------------------------------------------------
# collectives (RCCL on GPUs; gloo in the CPU tests)
# ----------------------------------------------------------------------------------------------------------
def _all_gather_rows(t: torch.Tensor, group, async_op: bool = False):
    world = dist.get_world_size(group)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    work = dist.all_gather_into_tensor(out, t.contiguous(), group=group, async_op=async_op)
    return (out, work) if async_op else out


def gather_under_local_work(ops, x, params, b, precision, gathers):
    """Overlap: ``gathers`` = callables that START the all-gathers of the text
    embeddings and ids (async: on RCCL's stream) and return their work handles; the part of the forward that depends on
    this rank's own rows only (``ops.prep_local``: conversions of X and W, T = X W) is then issued on the compute stream,
    and only after that the compute stream is made to wait for the gathers.  Ops objects without ``prep_local`` (the
    concat-MLP critic, whose first layer's text half needs the gathered rows) just wait."""
    works = [g() for g in gathers]
    if hasattr(ops, "prep_local"):
        ops.prep_local(x, params, b, precision)
    for w in works:
        if w is not None:
            w.wait()


def _reduce_scatter_rows(t: torch.Tensor, group) -> torch.Tensor:
    """Sum over ranks of t [G*r, ...], return this rank's block [r, ...]."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    rows = t.shape[0] // world
    if dist.get_backend(group) == "gloo":  # gloo has no reduce_scatter: all-reduce and slice (CPU tests only)
        t = t.contiguous()
        dist.all_reduce(t, group=group)
        return t[rank * rows:(rank + 1) * rows].clone()
    out = torch.empty((rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.reduce_scatter_tensor(out, t.contiguous(), group=group)
    return out


def _reduce_scatter_rows_start(t: torch.Tensor, group):
    """Start the reduce-scatter of t [G*r, ...] (async: on the backend's own stream); returns a function that waits for it
    and returns this rank's block.  What the caller launches in between runs beside the collective."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    rows = t.shape[0] // world
    t = t.contiguous()
    if dist.get_backend(group) == "gloo":  # gloo has no reduce_scatter: all-reduce and slice (CPU tests only)
        work = dist.all_reduce(t, group=group, async_op=True)

        def finish():
            work.wait()
            return t[rank * rows:(rank + 1) * rows].clone()
        return finish
    out = torch.empty((rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    work = dist.reduce_scatter_tensor(out, t, group=group, async_op=True)

    def finish():
        work.wait()
        return out
    return finish


def fp8_global_scales(ops, x, y_all, params, group) -> None:
    """fp8 mode (BASELINE configs[4]) on a sharded batch: per-tensor scales absmax / 448 of the WHOLE batch.  The local
    image rows (and the local rows of T = x W) differ between ranks, so the four absmax values are MAX-all-reduced
    between the stages of the preparation; the text embeddings are already all-gathered and W is replicated (their
    maxima are equal everywhere -- reducing them too keeps it one 16-byte collective).  Two collectives, forward only."""
    amax = torch.zeros(4, dtype=torch.float32, device=x.device)
    ops.fp8_stage(0, x, y_all, params, amax)
    dist.all_reduce(amax, op=dist.ReduceOp.MAX, group=group)
    ops.fp8_stage(1, x, y_all, params, amax)
    dist.all_reduce(amax, op=dist.ReduceOp.MAX, group=group)
    ops.fp8_stage(2, x, y_all, params, amax)


def flat_views(like: Sequence[torch.Tensor]):
    """One contiguous buffer and per-tensor views into it: the parameter gradients of a step travel in ONE all-reduce
    (six separate latency-bound collectives for the concat-MLP critic otherwise)."""
    if not like:
        return None, []
    flat = torch.empty(sum(t.numel() for t in like), dtype=like[0].dtype, device=like[0].device)
    views, off = [], 0
    for t in like:
        views.append(flat[off:off + t.numel()].view(t.shape))
        off += t.numel()
    return flat, views


def _supports_out(ops) -> bool:
    import inspect
    try:
        return "out" in inspect.signature(ops.backward).parameters
    except (TypeError, ValueError):
        return False


class GlobalBatchCriticFn(torch.autograd.Function):
    """loss over the GLOBAL batch from this rank's row block.  Returns (loss[1], stats)."""

    @staticmethod
    def forward(ctx, ops, group, estimator: int, precision: int, sid_rows, x, y, *params):
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        br = x.shape[0]
        need_grad = any(t.requires_grad for t in (x, y) + tuple(params))
        x = x.contiguous()
        params_c = [p.contiguous() for p in params]
        got = {}

        def start(name, t):
            def go():
                got[name], work = _all_gather_rows(t, group, async_op=True)
                return work
            return go
        gather_under_local_work(ops, x, params_c, world * br, precision, [start("y", y), start("sid", sid_rows)])
        y_all, sid_all = got["y"], got["sid"]
        if precision == _hip.MI_PREC_FP8:
            if not hasattr(ops, "fp8_stage"):
                raise ValueError('precision="fp8" on a sharded batch needs an ops object with fp8_stage (bilinear critic)')
            fp8_global_scales(ops, x, y_all, params_c, group)
        record, saved = ops.forward(x, y_all, params_c, sid_rows.contiguous(), sid_all, rank * br, estimator, precision,
                                    need_grad)
        records = _all_gather_rows(record.reshape(1, -1), group)  # [G, 8], rank order
        loss, stats = ops.merge(records, world * br, estimator)
        ctx.ops, ctx.group, ctx.saved = ops, group, saved
        ctx.param_like = [p.detach() for p in params]
        ctx.y_all_like, ctx.x_shape = y_all, tuple(x.shape)
        ctx.save_for_backward(stats)
        ctx.mark_non_differentiable(stats)
        return loss, stats

    @staticmethod
    def backward(ctx, grad_loss, _gstats):
        (stats,) = ctx.saved_tensors
        go = grad_loss.reshape(-1)[:1].to(torch.float32).contiguous()
        params = ctx.param_like
        flat, views = flat_views(params)
        if flat is not None and _supports_out(ctx.ops):
            y_all_like = ctx.y_all_like
            gx = torch.empty(ctx.x_shape, dtype=y_all_like.dtype, device=y_all_like.device)
            gy_partial = torch.empty_like(y_all_like)
            ctx.ops.backward(ctx.saved, stats, go, out=(gx, gy_partial, views))
            gparams = views
        else:  # ops objects without preallocated outputs (the oracle-backed ones of the CPU tests)
            gx, gy_partial, gparams = ctx.ops.backward(ctx.saved, stats, go)
            if flat is not None:
                for v, g in zip(views, gparams):
                    v.copy_(g)
                gparams = views
        gy = _reduce_scatter_rows(gy_partial, ctx.group)
        if flat is not None:
            dist.all_reduce(flat, group=ctx.group)  # one collective for all parameter gradients
        return (None, None, None, None, None, gx, gy, *gparams)


def _sharded_precision(precision: str, critic: str, x, y, params=None) -> int:
    """The same name -> code resolution as on one GPU (_hip.resolve_precision): "f32" on the bilinear critic is the bf16x3
    scheme where every size is a multiple of 8 (fp32 tolerances at a sixth of the time), exact fp32 products otherwise and
    under "f32_exact" -- so that a precision name means the same numerics and speed on one GPU and on a sharded batch."""
    from .mi_critics import _precision_code
    if critic == "bilinear" and x.dim() == 2 and y.dim() == 2:
        return _hip.resolve_precision(precision, True, (x.shape[0], x.shape[1], y.shape[1]))
    if critic == "concat_mlp" and params is not None and len(params) == 6:
        return _hip.resolve_precision(precision, False, concat_hidden=(params[0].shape[0], params[2].shape[0]))
    return _precision_code(precision)


def global_batch_mi_bound(embedding_img, embedding_txt, study_id_codes, critic_params: Sequence[torch.Tensor],
                          estimator: str = "infonce", precision: str = "bf16", critic: str = "bilinear", group=None,
                          ops=None, return_stats: bool = False):
    """Reference loss at B = world_size * local_batch with this rank's [B/G, d] embeddings.
    ``study_id_codes``: int64 tensor [B/G] (codes must be consistent across ranks, e.g. the integer study ids).
    Every rank returns the same loss; ``.backward()`` leaves the gradient of the global loss w.r.t. the local
    embeddings and the (all-reduced) gradient w.r.t. the critic parameters."""
    from .mi_critics import _estimator_code, _precision_code
    if ops is None:
        ops = {"bilinear": HipBilinearOps, "separable": HipSeparableOps, "concat_mlp": HipConcatMlpOps}[critic]()
    est = _estimator_code(estimator)
    prec = _sharded_precision(precision, critic, embedding_img, embedding_txt, critic_params)
    loss, stats = GlobalBatchCriticFn.apply(ops, group, est, prec, study_id_codes, embedding_img, embedding_txt,
                                            *critic_params)
    loss = loss if estimator == "dv" else loss.reshape(())
    return (loss, stats) if return_stats else loss


class GlobalBatchGraphStep:
    """Forward + backward of the global-batch bound for FIXED shapes and storage, with the two compute sections of a
    step replayed from hipGraphs and the collectives issued eagerly between them:

        all-gather Y, ids (graph 0 under them: bf16 X, W and T = X W of the rank's own rows) |
        graph 1: rest of the local forward -> partial record | all-gather records |
        graph 2: rank-ordered merge + local backward | reduce-scatter dY, all-reduce d(params)

    Bilinear critic on the fused kernels ("raw-record mode"): graph 1 ends with the fused B x B kernel (no finalize
    launch), the all-gather moves every rank's per-wave records (16 bytes each, a few KB per rank) and the first launch
    of graph 2 merges ALL of them in the gathered order on every workgroup -- no merge launch either.

    Why: at global batch 4096 a rank's kernels take tens of microseconds, and launching them one by one from Python
    (plus an autograd graph) costs several times that.  The graphs hold exactly the C-ABI calls of
    ``GlobalBatchCriticFn``; no collective is captured (RCCL calls stay ordinary stream work between two replays).

    ``x``, ``y`` [B/G, d], ``sid`` int64 [B/G] and ``params`` are read in place on every ``step()``: update their
    contents (``copy_``), not the objects.  ``step()`` returns the loss ([1]); gradients are in ``grad_x``, ``grad_y``,
    ``grad_params`` (overwritten by every step)."""

    def __init__(self, x, y, sid, params: Sequence[torch.Tensor], estimator: str = "infonce", precision: str = "bf16",
                 critic: str = "bilinear", group=None, ops=None, capture: bool = True, overlap_reduce_scatter=None):
        from .mi_critics import _estimator_code, _precision_code
        self.group = group
        # step_eager(): start the reduce-scatter of dY between the backward's two launches (it then runs beside dW).  OFF by
        # default: with direct calls the step is bound by the HOST (one-rank RCCL rehearsal, B = 4096: 0.196 ms per step
        # for 0.12 ms of kernels), and the extra C call + async handle cost 13 us there (0.208 ms) -- a GPU-side overlap
        # buys nothing until the host keeps ahead.  MI_DIST_RS_OVERLAP=1 or the argument turn it on.
        self.overlap_reduce_scatter = (bool(os.environ.get("MI_DIST_RS_OVERLAP")) if overlap_reduce_scatter is None
                                       else bool(overlap_reduce_scatter))
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.ops = ops if ops is not None else {"bilinear": HipBilinearOps, "separable": HipSeparableOps, "concat_mlp": HipConcatMlpOps}[critic]()
        self.est, self.prec = _estimator_code(estimator), _sharded_precision(precision, critic, x, y, params)
        self.x, self.y, self.sid = x.detach(), y.detach(), sid
        self.params = [p.detach() for p in params]
        for t in (self.x, self.y, self.sid, *self.params):
            if not t.is_contiguous():
                raise ValueError("GlobalBatchGraphStep needs contiguous tensors (they are read in place)")
        br = x.shape[0]
        dev = x.device
        on_gpu = dev.type == "cuda"
        self.y_all = torch.empty((self.world * br,) + tuple(y.shape[1:]), dtype=y.dtype, device=dev)
        self.sid_all = torch.empty(self.world * br, dtype=sid.dtype, device=dev)
        self.records = torch.empty(self.world, _hip.RECORD_FLOATS, dtype=x.dtype if not on_gpu else torch.float32, device=dev)
        self.grad_out = torch.ones(1, dtype=x.dtype if not on_gpu else torch.float32, device=dev)
        # the parameter gradients of a step live in one flat buffer: ONE all-reduce per step
        self.grad_flat, self.grad_params = flat_views(self.params)
        self._out = None
        if _supports_out(self.ops):
            self.grad_x = torch.empty_like(self.x)
            self.grad_y_partial = torch.empty_like(self.y_all)
            self._out = (self.grad_x, self.grad_y_partial, self.grad_params)
        self.graph_fwd = self.graph_bwd = self.graph_local = self.graph_full = None
        # raw-record mode (ops.forward_raw / merge_backward): the per-wave records of every rank are gathered and merged
        # inside the backward's first launch -- no finalize and no merge launch.  MI_DIST_NO_RAW=1: A/B switch.
        self._raw = (hasattr(self.ops, "forward_raw") and self.prec != _hip.MI_PREC_FP8
                     and not os.environ.get("MI_DIST_NO_RAW"))
        self.records_raw = None
        # MI_DIST_NO_OVERLAP=1: A/B switch of the measurements in profiles/README.md (the local part then runs inside forward)
        self._has_local = hasattr(self.ops, "prep_local") and not os.environ.get("MI_DIST_NO_OVERLAP")
        self._gather_inputs(local=False)
        if self.prec == _hip.MI_PREC_FP8:
            if not hasattr(self.ops, "fp8_stage"):
                raise ValueError('precision="fp8" on a sharded batch needs the bilinear critic')
            capture = False  # the fp8 preparation holds collectives (global scales): issued eagerly
        if not (capture and on_gpu):
            return  # eager: the same calls issued one by one (the gloo tests on the CPU, `bench.py --graph off`)
        torch.cuda.synchronize()
        # warm-up on a side stream (module loading, attribute calls), then capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._prep_local()
            self._forward()
            self._gather_records()
            self._merge_backward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if capture == "full":
            # ONE graph for the whole step, the five RCCL collectives captured with the kernels (RCCL calls are
            # capturable stream work): the host issues a single launch per step instead of ~12 calls.  Opt-in: the
            # capture of collectives could only be rehearsed with one rank on this project's one-GPU boxes.
            self._reduce_scatter_out = torch.empty_like(self.y) if dist.get_backend(group) != "gloo" else None
            with torch.cuda.stream(side):   # one eager pass of exactly the captured sequence (RCCL channel set-up)
                self._whole_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            dist.barrier(group=self.group)
            self.graph_full = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_full):
                self._whole_step()
            self._remember_captured()
            self._captured["grad_y"] = self.grad_y
            return
        pool = torch.cuda.graph_pool_handle()
        if self._has_local:
            # the rank-local part as its own graph, replayed while the input all-gathers are in flight; its workspace is
            # the one the forward graph goes on with (both captured from one pool, the tensor kept in self.saved)
            local = torch.cuda.CUDAGraph()
            with torch.cuda.graph(local, pool=pool):
                self._prep_local()
            if getattr(self.ops, "_local_ws", None) is not None:
                self.graph_local = local
        self.graph_fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_fwd, pool=pool):
            self._forward()
        # the record gathered from every rank must exist before the second capture reads it
        self._gather_records()
        self.graph_bwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_bwd, pool=pool):
            self._merge_backward()
        self._remember_captured()

    _CAPTURED_ATTRS = ("record", "saved", "loss", "stats", "grad_x", "grad_y_partial")

    def _remember_captured(self):
        """The tensors the graphs read and write.  step_eager() on this object rebinds record / saved / loss ... to a
        fresh eager workspace; a later replay must gather and return the CAPTURED ones again (in raw-record
        mode the backward graph otherwise merged the records of an older forward)."""
        self._captured = {k: getattr(self, k) for k in self._CAPTURED_ATTRS if hasattr(self, k)}

    def _restore_captured(self):
        for k, v in getattr(self, "_captured", {}).items():
            setattr(self, k, v)

    def _prep_local(self):
        if self._has_local:
            self.ops.prep_local(self.x, self.params, self.y_all.shape[0], self.prec)

    def _gather_inputs(self, local: bool = True):
        """The two input all-gathers, with the rank-local part of the forward issued while they are in flight.  Not
        captured: in the graphed mode the local part is its own small graph (graph_local) replayed at the same place."""
        works = [dist.all_gather_into_tensor(self.y_all, self.y, group=self.group, async_op=True),
                 dist.all_gather_into_tensor(self.sid_all, self.sid, group=self.group, async_op=True)]
        if local and self.graph_local is not None:
            self.graph_local.replay()
        elif local:
            self._prep_local()
        for w in works:
            w.wait()

    def _forward(self):
        if self._raw:
            got = self.ops.forward_raw(self.x, self.y_all, self.params, self.sid, self.sid_all,
                                       self.rank * self.x.shape[0], self.est, self.prec)
            if got is not None:
                self.record, self.saved = got  # [n, 4]: gathered as it is
                if self.records_raw is None:
                    self.records_raw = torch.empty((self.world * self.record.shape[0], 4), dtype=self.record.dtype,
                                                   device=self.record.device)
                return
            self._raw = False  # the shape does not take that path: the record protocol below, from now on
        if self.prec == _hip.MI_PREC_FP8:  # (two tiny MAX all-reduces inside: this section is never captured in that mode)
            fp8_global_scales(self.ops, self.x, self.y_all, self.params, self.group)
        self.record, self.saved = self.ops.forward(self.x, self.y_all, self.params, self.sid, self.sid_all,
                                                   self.rank * self.x.shape[0], self.est, self.prec, True)

    def _gather_records(self):
        if self._raw:
            dist.all_gather_into_tensor(self.records_raw, self.record, group=self.group)
        else:
            dist.all_gather_into_tensor(self.records, self.record.reshape(1, -1).to(self.records.dtype), group=self.group)

    def _merge_backward(self):
        if self._raw:
            self.loss, self.stats, gx, gy, gp = self.ops.merge_backward(
                self.saved, self.records_raw, self.world * self.x.shape[0], self.est, self.grad_out, out=self._out)
            if self._out is None:
                self.grad_x, self.grad_y_partial = gx, gy
                for v, g in zip(self.grad_params, gp):
                    v.copy_(g)
            return
        self.loss, self.stats = self.ops.merge(self.records, self.world * self.x.shape[0], self.est)
        if self._out is not None:
            self.ops.backward(self.saved, self.stats, self.grad_out, out=self._out)
        else:
            self.grad_x, self.grad_y_partial, gp = self.ops.backward(self.saved, self.stats, self.grad_out)
            for v, g in zip(self.grad_params, gp):
                v.copy_(g)

    def _exchange_gradients(self):
        self.grad_y = _reduce_scatter_rows(self.grad_y_partial, self.group)
        if self.grad_flat is not None:
            dist.all_reduce(self.grad_flat, group=self.group)

    def _whole_step(self):
        self._gather_inputs()
        self._forward()
        self._gather_records()
        self._merge_backward()
        if self._reduce_scatter_out is not None:
            dist.reduce_scatter_tensor(self._reduce_scatter_out, self.grad_y_partial, group=self.group)
            self.grad_y = self._reduce_scatter_out
        else:
            self.grad_y = _reduce_scatter_rows(self.grad_y_partial, self.group)
        if self.grad_flat is not None:
            dist.all_reduce(self.grad_flat, group=self.group)

    def step(self):
        """all-gather Y, ids | local forward | all-gather records | merge + local backward | reduce-scatter dY, ONE
        all-reduce of the flat parameter-gradient buffer.  Five collectives per step, in this order on every rank."""
        if self.graph_full is not None:
            self._restore_captured()
            self.graph_full.replay()
            return self.loss
        if self.graph_fwd is None:
            return self.step_eager()
        self._restore_captured()
        self._gather_inputs()
        self.graph_fwd.replay()
        self._gather_records()
        self.graph_bwd.replay()
        self._exchange_gradients()
        return self.loss

    def step_eager(self):
        self._gather_inputs()
        self._forward()
        self._gather_records()
        if self._raw and hasattr(self.ops, "merge_backward_tail") and self.overlap_reduce_scatter:
            # the reduce-scatter of dY needs the backward's FIRST launch only: started (on the backend's stream) before the
            # dW launch, it runs beside it; the parameter all-reduce follows dW.  Same five collectives, same order on
            # every rank.
            got = self.ops.merge_backward_tail(self.saved, self.records_raw, self.world * self.x.shape[0], self.est,
                                               self.grad_out, out=self._out)
            self.loss, self.stats = got[0], got[1]
            if self._out is None:
                self.grad_x, self.grad_y_partial = got[2], got[3]
            finish = _reduce_scatter_rows_start(self.grad_y_partial, self.group)
            gp = self.ops.backward_dw(self.saved, out=self._out)
            if self._out is None:
                for v, g in zip(self.grad_params, gp):
                    v.copy_(g)
            if self.grad_flat is not None:
                dist.all_reduce(self.grad_flat, group=self.group)
            self.grad_y = finish()
            return self.loss
        self._merge_backward()
        self._exchange_gradients()
        return self.loss
