"""hipGraph replay of the fused MI step (single GPU), as a product API.

The fused step is a handful of short kernels: at B = 4096 they take ~0.15 ms on the GPU while launching them one by one
through Python, ctypes and the autograd engine takes longer than that on the host.  ``GraphedMiStep`` captures the C-ABI
launches of one forward and of one backward into two hipGraphs, ONCE, for fixed shapes and fixed storage:

* nothing of autograd is inside a capture (a backward captured through ``loss.backward()`` drags ``AccumulateGrad``
  nodes of whatever stream created the leaves into the capture; that is how round 1's benchmark managed to abort inside
  ``capture_end``).  The captures hold exactly the library calls of ``mi_critics.BilinearCriticFn`` /
  ``ConcatMlpCriticFn``;
* ``step()`` replays both graphs on the static buffers (benchmarks, custom loops);
* ``loss(embedding_img, embedding_txt, study_id)`` returns a loss that is connected to autograd through a thin
  ``torch.autograd.Function``: forward = copy the embeddings into the static buffers + replay graph 1, backward = copy
  ``grad_output`` + replay graph 2 + hand the gradient buffers to autograd.  This is what ``MultiModalManager.train``
  uses, so the trainer's step costs what the benchmark measures plus two small copies.

The critic's parameters are read IN PLACE on every replay (optimizers update them in place): do not replace the
``nn.Parameter`` objects' storage after building the step (``.to()`` the critic first).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import _hip, mi_critics
from .mi_critics import _concat_params, _estimator_code, _precision_code


class GraphedMiStep:
    def __init__(self, critic, batch_size: int, d_img: int, d_txt: int, estimator: str = "dv", precision: str = "f32",
                 device=None, capture: bool = True, boundary: str = "f32"):
        """``boundary="bf16"`` (bilinear critic in "bf16" precision on the fused kernels only): the step takes bfloat16
        embeddings -- what the encoders emit under autocast -- and returns their gradients in bfloat16
        (``mi_bilinear_step_bf16``: the same bits as the fp32 boundary fed the same values, without the 25 MB of
        conversion traffic per step at B = 4096, d = 512).  The critic's parameters and their gradients stay float32."""
        from . import model as _model
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise _hip.MiCriticError("GraphedMiStep needs a ROCm device (no CPU fallback)")
        self.lib = _hip.load()
        self.est, self.prec = _estimator_code(estimator), _precision_code(precision)
        self.estimator = estimator
        self.b, self.dx, self.dy = int(batch_size), int(d_img), int(d_txt)
        dev = self.device
        if isinstance(critic, _model.BilinearCritic):
            self.kind = "bilinear"
            self.params: List[torch.Tensor] = [critic.weight]
            self.prec = _hip.resolve_precision(precision, True, (self.b, self.dx, self.dy))  # "f32" -> bf16x3 here
        elif isinstance(critic, _model.SeparableCritic):
            self.kind = "separable"  # S = (X Wg)(Y Wh)^T: projections, B x B stage and gradients by mi_separable_fwd/bwd
            self.params = [critic.wg, critic.wh]
            self.k = int(critic.wg.shape[1])
        elif critic is None:
            if d_img != d_txt:
                raise ValueError("critic=None is the separable form S = X Y^T: widths must agree")
            self.kind = "bilinear"
            self.params = []
        else:
            self.kind = "concat_mlp"
            w1, b1, w2, b2, w3, b3 = _concat_params(critic)
            self.params = [w1, b1, w2, b2, w3, b3]
            self.prec = _hip.resolve_precision(precision, False, concat_hidden=(w1.shape[0], w2.shape[0]))  # "f32" -> f16x3
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError("critic parameters must be contiguous float32 tensors on the step's device")
        if boundary not in ("f32", "bf16"):
            raise ValueError('boundary must be "f32" or "bf16"')
        self.boundary = boundary
        if boundary == "bf16":
            if self.kind != "bilinear" or not self.params or self.prec != _hip.MI_PREC_BF16 or \
                    self.lib.mi_bilinear_path(self.b, self.b, self.dx, self.dy, self.prec) != _hip.MI_PATH_FUSED_TAIL:
                raise ValueError('boundary="bf16" needs a BilinearCritic in precision "bf16" on a shape of the fused '
                                 'kernels (batch % 128 == 0, d_txt in {128, 256, 512}, d_img % 32 == 0)')
        io_dtype = torch.bfloat16 if boundary == "bf16" else torch.float32
        # static inputs
        self.x = torch.zeros(self.b, self.dx, dtype=io_dtype, device=dev)
        self.y = torch.zeros(self.b, self.dy, dtype=io_dtype, device=dev)
        self.sid = torch.arange(self.b, dtype=torch.int64, device=dev)
        self.grad_out = torch.ones(1, dtype=torch.float32, device=dev)
        # static outputs
        self.loss_buf = torch.zeros(1, dtype=torch.float32, device=dev)
        self.stats = _hip.new_stats(dev)
        self.record = torch.zeros(_hip.RECORD_FLOATS, dtype=torch.float32, device=dev)
        self.grad_x = torch.zeros_like(self.x)
        self.grad_y = torch.zeros_like(self.y)
        self.grad_params = [torch.zeros_like(p) for p in self.params]
        self.path = None
        if self.kind == "bilinear":
            if self.params:
                self.path = _hip.note_path("bilinear", (self.b, self.b, self.dx, self.dy), self.prec)
            nbytes = self.lib.mi_bilinear_workspace_bytes(self.b, self.b, self.dx, self.dy, self.prec)
        elif self.kind == "separable":
            if tuple(self.params[0].shape) != (self.dx, self.k) or tuple(self.params[1].shape) != (self.dy, self.k):
                raise ValueError("projection shapes must be [d_img, d_proj] and [d_txt, d_proj]")
            self.path = _hip.note_path("separable", (self.b, self.b, self.dx, self.dy, self.k), self.prec)
            nbytes = self.lib.mi_separable_workspace_bytes(self.b, self.b, self.dx, self.dy, self.k, self.prec)
        else:
            self.h1, self.h2 = self.params[0].shape[0], self.params[2].shape[0]
            if self.params[0].shape[1] != self.dx + self.dy:
                raise ValueError("critic input width does not match d_img + d_txt")
            nbytes = self.lib.mi_concat_mlp_workspace_bytes(self.b, self.b, self.dx, self.dy, self.h1, self.h2, self.prec, 1)
            self.scores = torch.zeros(self.b, self.b, dtype=torch.float32, device=dev)
        self.ws = _hip.workspace(nbytes, dev)
        self.graph_fwd: Optional[torch.cuda.CUDAGraph] = None
        self.graph_bwd: Optional[torch.cuda.CUDAGraph] = None
        self.graph_step: Optional[torch.cuda.CUDAGraph] = None
        # the static inputs, the workspace and the statistics belong to ONE forward at a time: every forward bumps the
        # generation, and a backward whose forward is no longer the latest one raises instead of silently differentiating
        # another batch (gradient accumulation over two losses needs two GraphedMiStep objects, or the eager path)
        self.generation = 0
        if capture:
            self._capture()

    # ------------------------------------------------------------------------------------------ raw C-ABI calls
    def _w3_flat(self):
        return self.params[4].reshape(-1)  # [1, h2] -> [h2], a view of the parameter's storage

    def _fwd(self):
        p = self.params
        if self.boundary == "bf16":
            raise _hip.MiCriticError('boundary="bf16" has the one-call step only (step() / step_eager() / loss())')
        if self.kind == "bilinear":
            _hip.call("mi_bilinear_fwd", self.device, self.x.data_ptr(), self.y.data_ptr(), p[0].data_ptr() if p else None,
                      self.sid.data_ptr(), self.sid.data_ptr(), self.b, self.b, 0, self.dx, self.dy, self.est, self.prec, 1,
                      self.loss_buf.data_ptr(), self.stats.data_ptr(), self.record.data_ptr(), None, self.ws.data_ptr(),
                      self.ws.numel())
        elif self.kind == "separable":
            _hip.call("mi_separable_fwd", self.device, self.x.data_ptr(), self.y.data_ptr(), p[0].data_ptr(), p[1].data_ptr(),
                      self.sid.data_ptr(), self.sid.data_ptr(), self.b, self.b, 0, self.dx, self.dy, self.k, self.est,
                      self.prec, 1, self.loss_buf.data_ptr(), self.stats.data_ptr(), self.record.data_ptr(),
                      self.ws.data_ptr(), self.ws.numel())
        else:
            _hip.call("mi_concat_mlp_fwd", self.device, self.x.data_ptr(), self.y.data_ptr(), p[0].data_ptr(),
                      p[1].data_ptr(), p[2].data_ptr(), p[3].data_ptr(), p[4].data_ptr(), p[5].data_ptr(),
                      self.sid.data_ptr(), self.sid.data_ptr(), self.b, self.b, 0, self.dx, self.dy, self.h1, self.h2,
                      self.est, self.prec, 1, self.loss_buf.data_ptr(), self.stats.data_ptr(), self.record.data_ptr(),
                      self.scores.data_ptr(), self.ws.data_ptr(), self.ws.numel())

    def _bwd(self):
        p, g = self.params, self.grad_params
        if self.kind == "bilinear":
            _hip.call("mi_bilinear_bwd", self.device, self.x.data_ptr(), self.y.data_ptr(), p[0].data_ptr() if p else None,
                      self.sid.data_ptr(), self.sid.data_ptr(), self.b, self.b, 0, self.dx, self.dy, self.prec,
                      self.stats.data_ptr(), self.grad_out.data_ptr(), self.grad_x.data_ptr(), self.grad_y.data_ptr(),
                      g[0].data_ptr() if g else None, self.ws.data_ptr(), self.ws.numel(), 1)
        elif self.kind == "separable":
            _hip.call("mi_separable_bwd", self.device, self.x.data_ptr(), self.y.data_ptr(), p[0].data_ptr(), p[1].data_ptr(),
                      self.sid.data_ptr(), self.sid.data_ptr(), self.b, self.b, 0, self.dx, self.dy, self.k, self.prec,
                      self.stats.data_ptr(), self.grad_out.data_ptr(), self.grad_x.data_ptr(), self.grad_y.data_ptr(),
                      g[0].data_ptr(), g[1].data_ptr(), self.ws.data_ptr(), self.ws.numel(), 1)
        else:
            _hip.call("mi_concat_mlp_bwd", self.device, self.x.data_ptr(), self.y.data_ptr(), p[0].data_ptr(),
                      p[1].data_ptr(), p[2].data_ptr(), p[3].data_ptr(), p[4].data_ptr(), p[5].data_ptr(),
                      self.sid.data_ptr(), self.sid.data_ptr(), self.b, self.b, 0, self.dx, self.dy, self.h1, self.h2,
                      self.prec, self.stats.data_ptr(), self.grad_out.data_ptr(), self.scores.data_ptr(),
                      self.grad_x.data_ptr(), self.grad_y.data_ptr(), *[t.data_ptr() for t in g], self.ws.data_ptr(),
                      self.ws.numel())

    def _step(self):
        """Forward + backward as ONE C-ABI call where the library has one (bilinear critic: mi_bilinear_step, four launches
        and no finalize kernel); the two calls otherwise."""
        p, g = self.params, self.grad_params
        if self.boundary == "bf16":
            _hip.call("mi_bilinear_step_bf16", self.device, self.x.data_ptr(), self.y.data_ptr(), p[0].data_ptr(),
                      self.sid.data_ptr(), self.b, self.dx, self.dy, self.est, self.grad_out.data_ptr(),
                      self.loss_buf.data_ptr(), self.stats.data_ptr(), self.record.data_ptr(), self.grad_x.data_ptr(),
                      self.grad_y.data_ptr(), 1, g[0].data_ptr(), self.ws.data_ptr(), self.ws.numel())
        elif self.kind == "bilinear":
            _hip.call("mi_bilinear_step", self.device, self.x.data_ptr(), self.y.data_ptr(), p[0].data_ptr() if p else None,
                      self.sid.data_ptr(), self.b, self.dx, self.dy, self.est, self.prec, self.grad_out.data_ptr(),
                      self.loss_buf.data_ptr(), self.stats.data_ptr(), self.record.data_ptr(), self.grad_x.data_ptr(),
                      self.grad_y.data_ptr(), g[0].data_ptr() if g else None, self.ws.data_ptr(), self.ws.numel())
        elif self.kind == "separable":
            _hip.call("mi_separable_step", self.device, self.x.data_ptr(), self.y.data_ptr(), p[0].data_ptr(), p[1].data_ptr(),
                      self.sid.data_ptr(), self.b, self.dx, self.dy, self.k, self.est, self.prec, self.grad_out.data_ptr(),
                      self.loss_buf.data_ptr(), self.stats.data_ptr(), self.record.data_ptr(), self.grad_x.data_ptr(),
                      self.grad_y.data_ptr(), g[0].data_ptr(), g[1].data_ptr(), self.ws.data_ptr(), self.ws.numel())
        else:
            self._fwd()
            self._bwd()

    def _capture(self):
        # warm-up on a side stream: module loading and the one-time kernel-attribute calls must not fall into a capture
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(2):
                if self.boundary != "bf16":
                    self._fwd()
                    self._bwd()
                self._step()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        # thread_local: another thread's allocator traffic (a DataLoader's pin-memory thread) must not invalidate a capture
        pool = None
        if self.boundary != "bf16":
            self.graph_fwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_fwd, capture_error_mode="thread_local"):
                self._fwd()
            self.graph_bwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_bwd, pool=self.graph_fwd.pool(), capture_error_mode="thread_local"):
                self._bwd()
            pool = self.graph_fwd.pool()
        # forward + backward as ONE graph for step(): a replay costs ~10 us of fixed overhead, more than the launches of a
        # short backward
        self.graph_step = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_step, pool=pool, capture_error_mode="thread_local"):
            self._step()

    # ------------------------------------------------------------------------------------------ replay
    def forward(self) -> torch.Tensor:
        """Loss of the static inputs (shape [1]; a view of the step's own buffer)."""
        self.generation += 1
        if self.graph_fwd is not None:
            self.graph_fwd.replay()
        else:
            self._fwd()
        return self.loss_buf

    def backward(self) -> None:
        """Gradients of ``grad_out * loss`` into ``grad_x``, ``grad_y``, ``grad_params`` (overwritten)."""
        if self.graph_bwd is not None:
            self.graph_bwd.replay()
        else:
            self._bwd()

    def step(self) -> torch.Tensor:
        """Forward + backward of the static inputs with grad_out as it stands (1 unless changed): one graph replay."""
        self.generation += 1
        if self.graph_step is not None:
            self.graph_step.replay()
        else:
            self._step()
        return self.loss_buf

    def step_eager(self) -> torch.Tensor:
        """The same C-ABI calls issued one by one (no replay overhead; needs a host that keeps ahead of ~10 us kernels)."""
        self.generation += 1
        self._step()
        return self.loss_buf

    def set_inputs(self, embedding_img: torch.Tensor, embedding_txt: torch.Tensor, study_id=None) -> None:
        self.x.copy_(embedding_img.detach())
        self.y.copy_(embedding_txt.detach())
        if study_id is not None:
            self.sid.copy_(mi_critics.study_id_codes(study_id, self.device))

    def loss(self, embedding_img: torch.Tensor, embedding_txt: torch.Tensor, study_id=None) -> torch.Tensor:
        """Autograd-connected loss of one batch: shape [1] for "dv", [] for "infonce", as the reference."""
        if tuple(embedding_img.shape) != (self.b, self.dx) or tuple(embedding_txt.shape) != (self.b, self.dy):
            raise ValueError(f"GraphedMiStep was built for [{self.b},{self.dx}] / [{self.b},{self.dy}] embeddings")
        if study_id is not None:
            self.sid.copy_(mi_critics.study_id_codes(study_id, self.device))
        out = _GraphedFn.apply(self, embedding_img, embedding_txt, *self.params)
        return out if self.estimator == "dv" else out.reshape(())


class _GraphedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, step: GraphedMiStep, x, y, *params):
        step.x.copy_(x)
        step.y.copy_(y)
        if step.boundary == "bf16":
            step.step()  # one call computes the loss AND the gradients of 1 * loss; backward() scales them
        else:
            step.forward()
        ctx.step = step
        ctx.generation = step.generation
        return step.loss_buf.clone()

    @staticmethod
    def backward(ctx, grad_loss):
        step = ctx.step
        if ctx.generation != step.generation:
            raise RuntimeError(
                "GraphedMiStep: this loss's forward is no longer the step's latest one (another loss() / forward() / step() "
                "ran on the same GraphedMiStep before this backward): its static inputs, workspace and statistics now "
                "belong to the later batch.  Call backward() before the next forward, or use one GraphedMiStep per loss "
                "that is alive at the same time (or mi_critics.fused_mi_bound, which keeps per-call state).")
        if step.boundary == "bf16":
            go = grad_loss.reshape(-1)[:1].float()
            return (None, (step.grad_x.float() * go).to(step.grad_x.dtype), (step.grad_y.float() * go).to(step.grad_y.dtype),
                    *[g * go for g in step.grad_params])
        step.grad_out.copy_(grad_loss.reshape(-1)[:1])
        step.backward()
        # clones: autograd may keep (or accumulate into) what it is handed, and the buffers are rewritten next step
        grads = (None, step.grad_x.clone(), step.grad_y.clone(), *[g.clone() for g in step.grad_params])
        step.grad_out.fill_(1.0)  # raw step() calls always differentiate 1 * loss
        return grads
