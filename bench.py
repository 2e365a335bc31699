#!/usr/bin/env python3
"""Benchmark of the MI-critic hot path: one critic forward + backward over a synthetic batch per step.

  python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: launched by torch.distributed.run, one rank/GPU)

Headline workload (BASELINE.json configs[3], the configuration its metric is quoted on; it fits one GPU):
  global-batch "InfoNCE" (reference semantics, mi_critics.py:14-23), bilinear critic S = (X W) Y^T, B_global = 4096,
  d = 512, bf16 MFMA operands / fp32 accumulate, unique study ids, inputs resident in HBM.  With N GPUs the SAME global
  batch is sharded by row blocks (strong scaling) and the text embeddings are all-gathered over RCCL.
Secondary workload in the same JSON line ("secondary"): the reference's own critic, make_mlp(2d,[1024,512]) concat-MLP.

One JSON line on stdout (rank 0) with the driver's contract plus "roofline" (dominant kernel, HIP-event timed through
the library's profiling hook) and "cpu_baseline" (the oracle timed on this box's host cores, rank 0, N = 1 only).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "mutual-information-multimodal_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}  # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--batch", type=int, default=4096, help="GLOBAL batch")
    p.add_argument("--dim", type=int, default=512)
    p.add_argument("--critic", default="bilinear", choices=["bilinear", "concat_mlp"])
    p.add_argument("--estimator", default="infonce", choices=["dv", "infonce"])
    p.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                   help="replay the step from hipGraphs (auto = on).  On one GPU the whole step is one graph; sharded "
                        "over GPUs the two compute sections are graphs and the RCCL collectives stay eager between "
                        "them.  Eagerly the bilinear step is bound by the host's launch rate, not by the GPU; for the "
                        "concat-MLP step (51 ms of kernels) it makes no measurable difference")
    p.add_argument("--no-secondary", action="store_true")
    p.add_argument("--secondary-steps", type=int, default=5)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=15.0)
    p.add_argument("--profile-steps", type=int, default=5)
    return p.parse_args()


# ------------------------------------------------------------------------------------------------------------
def make_inputs(batch, d_img, d_txt, seed, rank, world, device):
    """X, Y ~ N(0,1) from a seeded generator; every rank draws the full batch and keeps its row block, so
    an N-GPU run sees exactly the single-GPU data."""
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, d_img, generator=gen)
    y = torch.randn(batch, d_txt, generator=gen)
    br = batch // world
    sl = slice(rank * br, (rank + 1) * br)
    sid = torch.arange(batch, dtype=torch.int64)
    return x[sl].to(device), y[sl].to(device), sid[sl].to(device), (x, y)


def make_critic(kind, d_img, d_txt, seed, device):
    from mutual_info_img_txt.model import BilinearCritic, make_mlp
    torch.manual_seed(seed)
    if kind == "bilinear":
        m = BilinearCritic(d_img, d_txt)
    else:
        m = make_mlp(d_img + d_txt, [1024, 512])  # reference main_utils.py:77 with 2d inputs
    return m.to(device)


def critic_params(kind, critic):
    from mutual_info_img_txt.mi_critics import _concat_params
    if kind == "bilinear":
        return [critic.weight]
    w1, b1, w2, b2, w3, b3 = _concat_params(critic)
    return [w1, b1, w2, b2, w3.reshape(-1), b3]


def algorithmic_flops(kind, b, d_img, d_txt, h1=1024, h2=512):
    """fwd + bwd, no recompute credit."""
    if kind == "bilinear":
        return 6.0 * b * b * d_txt + 6.0 * b * d_img * d_txt
    return 6.0 * b * b * h1 * h2 + 6.0 * b * b * h2 + 12.0 * b * (d_img + d_txt) / 2 * h1


def kernel_flops(name, br, b, d_img, d_txt, h1=1024, h2=512):
    """Algorithmic flops of ONE launch of a named kernel (0 for HBM-bound helper kernels)."""
    if name.startswith("bilinear"):
        if "fused S |" in name:  # scores + both B x B gradient contractions in one launch (no recompute credit)
            return 6.0 * br * b * d_txt
        if "fused S + LSE" in name:
            return 2.0 * br * b * d_txt
        if "from the fused sums" in name:
            return 0.0
        if "dW" in name and "|" in name:  # two-problem launch: dW = X^T dT and dX = dT W^T
            return 4.0 * br * d_img * d_txt
        if "|" in name:  # two-problem launch: dT = G Y and dY = G^T T
            return 4.0 * br * b * d_txt
        if any(t in name for t in ("score+LSE", "bilinear G", "dT = G Y", "dY = G^T T")):
            return 2.0 * br * b * d_txt
        if "prep" in name or "slabs" in name:
            return 0.0
        return 2.0 * br * d_img * d_txt
    if name in ("concat_fwd_kernel", "concat_bwd_duv_kernel", "concat_bwd_dw2_kernel"):
        return 2.0 * br * b * h1 * h2
    if name.startswith("concat U") or name.startswith("concat dX") or name.startswith("concat dW1x"):
        return 2.0 * br * d_img * h1
    if name.startswith("concat V") or name.startswith("concat dY") or name.startswith("concat dW1y"):
        return 2.0 * b * d_txt * h1
    return 0.0


class Stepper:
    """One critic forward + backward through the public (autograd) API; optionally replayed from a hipGraph."""

    def __init__(self, kind, args, rank, world, device, group):
        from mutual_info_img_txt import mi_critics
        from mutual_info_img_txt.distributed import global_batch_mi_bound
        self.kind, self.world, self.device = kind, world, device
        d = args.dim
        self.x, self.y, self.sid, self.full = make_inputs(args.batch, d, d, 3, rank, world, device)
        self.critic = make_critic(kind, d, d, 3, device)
        # built per use: a cached w3.reshape(-1) keeps an autograd view (and w3's AccumulateGrad node, bound to the
        # stream of this constructor) alive, which breaks the capture of a later backward
        self.params_fn = lambda: critic_params(kind, self.critic)
        self.x.requires_grad_(True)
        self.y.requires_grad_(True)
        self.args = args
        self.graph = None
        self.staged = None
        self.dist_mode = not (world == 1 and not os.environ.get("MI_BENCH_FORCE_DIST"))
        if world == 1 and not os.environ.get("MI_BENCH_FORCE_DIST"):
            self._loss = lambda: mi_critics.fused_mi_bound(self.x, self.y, self.sid, self.critic, args.estimator,
                                                           precision=args.precision)
        else:
            self._loss = lambda: global_batch_mi_bound(self.x, self.y, self.sid, self.params_fn(), args.estimator,
                                                       args.precision, critic=kind, group=group)
        self.loss = None

    def eager(self):
        self.x.grad = None
        self.y.grad = None
        for p in self.critic.parameters():
            p.grad = None
        self.loss = self._loss()
        self.loss.sum().backward()

    def try_capture(self):
        self.eager()
        torch.cuda.synchronize()
        self.loss = None
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self.eager()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        # drop the last autograd graph: its AccumulateGrad nodes are bound to the warm-up stream, and a backward that
        # reuses them inside the capture synchronises with that (non-capturing) stream and takes the process down
        self.loss = None
        self.x.grad = None
        self.y.grad = None
        for p in self.critic.parameters():
            p.grad = None
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.loss = self._loss()
            self.loss.sum().backward()
        self.graph = g

    def try_staged(self, group):
        """Sharded step with its two compute sections replayed from hipGraphs and the collectives eager between them."""
        from mutual_info_img_txt.distributed import GlobalBatchGraphStep
        self.staged = GlobalBatchGraphStep(self.x.detach(), self.y.detach(), self.sid,
                                           [p.detach() for p in self.params_fn()], self.args.estimator,
                                           self.args.precision, critic=self.kind, group=group)

    def step(self):
        if self.staged is not None:
            self.loss = self.staged.step()
        elif self.graph is not None:
            self.graph.replay()
        else:
            self.eager()


def timed_run(stepper, steps, warmup, world):
    for _ in range(warmup):
        stepper.step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        stepper.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=stepper.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def profile_kernels(stepper, steps):
    from mutual_info_img_txt import _hip
    graph, stepper.graph = stepper.graph, None  # events need eager launches
    stepper.eager()
    torch.cuda.synchronize()
    with _hip.kernel_profile() as prof:
        for _ in range(steps):
            stepper.eager()
    stepper.graph = graph
    return prof.by_name()


# profiling-hook kernel name -> substring of the rocprofv3 kernel name in profiles/*_pmc_traffic.json
PMC_KERNEL_OF = {
    "bilinear fused S | P Y | P^T T": "bilinear_flash_kernel",
    "bilinear dT = G Y | dY = G^T T": "gemm_bf16_pipe_kernel",
    "bilinear G": "gemm_bf16_big_kernel<mi::EpiGradScore2>",
    "bilinear score+LSE": "gemm_bf16_big_kernel<mi::EpiScoreLse2>",
    "concat_fwd_kernel": "concat_fwd_dma_kernel",
    "concat_bwd_duv_kernel": "concat_bwd_duv_kernel",
    "concat_bwd_dw2_kernel": "concat_bwd_dw2_kernel",
}


def measured_traffic(name, b, d):
    """HBM bytes per launch of a kernel from the newest committed PMC pass (tools/profile_round.sh; FETCH_SIZE and
    WRITE_SIZE in separate rocprofv3 --pmc runs, gfx950 corrections applied there).  Only valid for the configuration
    those passes ran (B=4096, d=512, one GPU); None otherwise -- bench.py itself cannot run the profiler."""
    if (b, d) != (4096, 512) or name not in PMC_KERNEL_OF:
        return None
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            table = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    for key, val in table.items():
        if PMC_KERNEL_OF[name] in key:
            return {"bytes": round(val["total_bytes"]), "fetch": round(val["fetch_bytes"]),
                    "write": round(val["write_bytes"]), "source": os.path.basename(files[-1])}
    return None


def roofline_of(kernels, br, b, d, precision):
    if not kernels:
        return None
    name = max(kernels, key=lambda k: kernels[k]["ms_total"])
    k = kernels[name]
    fl = kernel_flops(name, br, b, d, d)
    if fl <= 0:
        return {"kernel": name, "bound": "hbm", "achieved": None, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": None,
                "traffic": None, "avg_us": k["ms_avg"] * 1e3}
    achieved = fl / (k["ms_avg"] * 1e-3) / 1e12
    peak = PEAK_TFLOPS[precision]
    tr = measured_traffic(name, b, d) if br == b else None
    return {"kernel": name, "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": tr["bytes"] if tr else None, "traffic_detail": tr,
            "avg_us": round(k["ms_avg"] * 1e3, 2), "flops_per_launch": fl}


def cpu_baseline(kind, args):
    """The oracle (a CPU port of the reference algorithm, pinned to the reference by tests/golden) timed on this box's
    host cores on a bounded sample of the same workload."""
    from oracle import mi_oracle as orc
    # host cores actually available to this process (a 1-GPU box shares a 256-thread host: its CPU share is 16)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, int(os.environ.get("MI_BENCH_CPU_THREADS", "16"))))
    torch.set_num_threads(threads)
    d = args.dim
    if kind == "bilinear":
        b = args.batch
        gen = torch.Generator().manual_seed(3)
        x, y = torch.randn(b, d, generator=gen), torch.randn(b, d, generator=gen)
        w = torch.randn(d, d, generator=gen) / math.sqrt(d)
        sid = list(range(b))
        fn = lambda: orc.matrix_step(lambda a, c, ww: orc.bilinear_scores(a, c, ww), [x, y, w], sid, args.estimator)  # noqa: E731
        sample = f"full workload B={b} d={d} fp32, oracle matrix_step"
    else:
        b = min(args.batch, 256)  # the CPU needs ~4 s per step at B=256; larger B is infeasible in-budget
        x, y, sid, params = orc.synthetic_case(b, d, d, salt=1)
        fn = lambda: orc.concat_matrix_step(x, y, sid, params, args.estimator)  # noqa: E731
        sample = f"B={b} of {args.batch} (pairs/s falls further with B), d={d}, fp32, oracle concat_matrix_step"
    fn()
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        el = time.perf_counter() - t0
        if el > args.cpu_seconds or n >= 200:
            break
    return {"value": round(b * n / el, 2), "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{sample}; {n} steps in {el:.1f} s"}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the MI critic path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    group = None
    if world > 1 or os.environ.get("MI_BENCH_FORCE_DIST"):  # the latter: rehearse the RCCL path with one rank
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        group = dist.group.WORLD
    if args.batch % world:
        raise SystemExit("--batch must be divisible by the number of GPUs")
    from mutual_info_img_txt import _hip
    _hip.load()

    def run(kind, steps, warmup):
        st = Stepper(kind, args, rank, world, device, group)
        graph_used = False
        want_graph = args.graph in ("on", "auto")
        if want_graph:
            try:
                if st.dist_mode:
                    st.try_staged(group)
                    graph_used = "staged"
                else:
                    st.try_capture()
                    graph_used = True
            except Exception as e:  # capture is an optimisation, not a requirement
                if args.graph == "on":
                    raise
                st.graph = None
                st.staged = None
                torch.cuda.synchronize()
                if rank == 0:
                    print(f"[bench] hipGraph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
        elapsed = timed_run(st, steps, warmup, world)
        kernels = profile_kernels(st, args.profile_steps)
        loss = float(st.loss.detach().float().sum().item())
        return st, elapsed, kernels, graph_used, loss

    st, elapsed, kernels, graph_used, loss = run(args.critic, args.steps, args.warmup)
    b, d, br = args.batch, args.dim, args.batch // world
    ms = elapsed / args.steps * 1e3
    flops = algorithmic_flops(args.critic, b, d, d)
    out = {
        "metric": "img-txt pairs/sec (MI critic fwd+bwd), global-batch InfoNCE",
        "value": round(b / (ms * 1e-3), 1),
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms, 5),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": args.precision,
        "data": "synthetic",
        "config": {"workload": f"BASELINE configs[3]: global-batch {args.estimator} (reference semantics), "
                               f"{args.critic} critic fwd+bwd, B_global={b}, d={d}",
                   "global_batch": b, "embed_dim": d, "critic": args.critic, "estimator": args.estimator,
                   "parallelism": f"row-block sharding x{world}, RCCL all-gather of text embeddings" if world > 1 else "single GPU",
                   "hip_graph": bool(graph_used),
                   "graph_mode": {True: "whole step", "staged": "compute sections; collectives eager between them",
                                  False: "none"}[graph_used]},
        "loss": loss,
        "step_algorithmic_tflops": round(flops / (ms * 1e-3) / 1e12, 2),
        "step_frac_of_peak": round(flops / (ms * 1e-3) / 1e12 / (PEAK_TFLOPS[args.precision] * world), 5),
        "roofline": roofline_of(kernels, br, b, d, args.precision),
        "kernels_us": {k: round(v["ms_avg"] * 1e3, 2) for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms_total"])},
    }
    del st
    torch.cuda.empty_cache()
    if not args.no_secondary:
        other = "concat_mlp" if args.critic == "bilinear" else "bilinear"
        try:
            st2, el2, k2, g2, loss2 = run(other, args.secondary_steps, 2)
            ms2 = el2 / args.secondary_steps * 1e3
            fl2 = algorithmic_flops(other, b, d, d)
            out["secondary"] = {
                "workload": f"{other} critic fwd+bwd, B_global={b}, d={d}" + (" (the reference's mi_discriminator)" if other == "concat_mlp" else ""),
                "value": round(b / (ms2 * 1e-3), 1), "unit": "pairs/s", "ms_per_step": round(ms2, 4),
                "steps": args.secondary_steps, "hip_graph": g2, "loss": loss2,
                "step_algorithmic_tflops": round(fl2 / (ms2 * 1e-3) / 1e12, 2),
                "step_frac_of_peak": round(fl2 / (ms2 * 1e-3) / 1e12 / (PEAK_TFLOPS[args.precision] * world), 5),
                "roofline": roofline_of(k2, br, b, d, args.precision),
                "kernels_us": {k: round(v["ms_avg"] * 1e3, 2) for k, v in sorted(k2.items(), key=lambda kv: -kv[1]["ms_total"])},
            }
            del st2
        except Exception as e:
            out["secondary"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.critic, args)
        if "secondary" in out and "error" not in out["secondary"]:
            out["secondary"]["cpu_baseline"] = cpu_baseline("concat_mlp" if args.critic == "bilinear" else "bilinear", args)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
