"""Worker of tests/test_distributed_gpu.py: TWO ranks sharing the box's one GPU, process group on gloo (RCCL refuses two
ranks on one device), the product's HIP ops (HipBilinearOps / HipSeparableOps / HipConcatMlpOps) underneath.  What no other
test runs together: the real kernels on a rank's row block + real inter-process collectives with world_size > 1 -- the
sharded forward, raw-record gather, merged backward, reduce-scatter of dY beside the dW launch, flat all-reduce -- through
the autograd entry (global_batch_mi_bound), the eager step and the graphed step.  Each rank checks ITS rows against the
single-GPU step on the full batch (same kernels, same rounding of operands; summation order differs)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mutual-information-multimodal_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    from mutual_info_img_txt import distributed as mid, mi_critics
    from mutual_info_img_txt.model import BilinearCritic, SeparableCritic, make_mlp
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda:0")
    cases = [("bilinear", "bf16", 512, 256, "infonce"), ("bilinear", "bf16", 256, 128, "dv"),
             ("bilinear", "f32", 128, 64, "infonce"), ("bilinear", "fp8", 256, 128, "infonce"),
             ("separable", "bf16", 256, 256, "infonce"), ("concat_mlp", "f16", 128, 64, "dv"),
             ("concat_mlp", "f32", 128, 32, "infonce")]
    for kind, prec, b, d, est in cases:
        torch.manual_seed(100 + b + d)
        if kind == "bilinear":
            critic = BilinearCritic(d, d)
        elif kind == "separable":
            critic = SeparableCritic(d, d, d)
        else:
            critic = make_mlp(2 * d, [128, 256])
        critic = critic.to(dev)
        gen = torch.Generator().manual_seed(b * 7 + d)
        x, y = torch.randn(b, d, generator=gen).to(dev), torch.randn(b, d, generator=gen).to(dev)
        sid = torch.arange(b)
        sid[5] = sid[4]
        sid[b - 1] = sid[1]  # an equal pair across the two ranks' row blocks
        sid = sid.to(dev)
        params = [p.detach() for p in critic.parameters()]
        # single-GPU reference on the full batch
        xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
        for p in critic.parameters():
            p.grad = None
        ref = mi_critics.fused_mi_bound(xr, yr, sid, critic, est, precision=prec)
        ref.sum().backward()
        ref_p = [p.grad.clone() for p in critic.parameters()]
        br = b // world
        rows = slice(rank * br, (rank + 1) * br)
        # summation order is all that differs from one GPU -- except: fp8 (T is requantised per rank under the MAX-reduced
        # scales) and the separable critic in bf16 (a rank rounds ITS partial dC to bf16 before dY = dC Wh^T; one GPU rounds
        # the full sum once): those two compare at their operand precision
        tol = 2e-2 if prec == "fp8" else (6e-3 if (kind == "separable" and prec == "bf16") else 3e-5)

        def check(tag, loss, gx, gy, gp):
            assert abs(float(loss.sum()) - float(ref.sum())) <= tol * max(1.0, abs(float(ref.sum()))), (kind, prec, tag, float(loss.sum()), float(ref.sum()))
            for name, got, want in [("dx", gx, xr.grad[rows]), ("dy", gy, yr.grad[rows])] + \
                                   [(f"dp{n}", g.reshape(w.shape), w) for n, (g, w) in enumerate(zip(gp, ref_p))]:
                scale = float(want.abs().max()) + 1e-30
                err = float((got - want).abs().max()) / scale
                assert err <= tol, (kind, prec, tag, name, err)

        # (1) autograd entry
        xl, yl = x[rows].clone().requires_grad_(True), y[rows].clone().requires_grad_(True)
        pl = [p.clone().requires_grad_(True) for p in params]
        loss = mid.global_batch_mi_bound(xl, yl, sid[rows].contiguous(), pl, est, prec, critic=kind)
        loss.sum().backward()
        check("autograd", loss.detach(), xl.grad, yl.grad, [p.grad for p in pl])
        # (2) the step object: eager launches, then graph replays (twice, new inputs in place for the second)
        step = mid.GlobalBatchGraphStep(x[rows].contiguous(), y[rows].contiguous(), sid[rows].contiguous(), params, est, prec,
                                        critic=kind)
        l1 = step.step_eager().clone()
        check("step_eager", l1, step.grad_x, step.grad_y, step.grad_params)
        step.overlap_reduce_scatter = True   # the split backward: dY's reduce-scatter started before the dW launch
        l1b = step.step_eager().clone()
        check("step_eager, reduce-scatter beside dW", l1b, step.grad_x, step.grad_y, step.grad_params)
        step.overlap_reduce_scatter = False
        l2 = step.step().clone()
        check("step (replay)", l2, step.grad_x, step.grad_y, step.grad_params)
        l3 = step.step().clone()
        check("step (second replay)", l3, step.grad_x, step.grad_y, step.grad_params)
        assert float(l2) == float(l3)
        torch.cuda.synchronize()
        dist.barrier()
        if rank == 0:
            print(f"two-rank gpu ok: {kind} {prec} B={b} d={d} {est}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
