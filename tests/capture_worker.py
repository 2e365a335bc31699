"""Child process of tests/test_autograd_capture.py: captures the PUBLIC autograd path (mi_critics.fused_mi_bound forward +
loss.backward()) into a hipGraph with PyTorch's whole-network recipe, replays it on new inputs and compares with eager.

Run in a child because a capture that goes wrong aborts the process instead of raising (round 1: a backward captured
while AccumulateGrad nodes created on ANOTHER stream were alive died inside capture_end).  The recipe that avoids it:
warm up forward + backward on a side stream, set the grads to None, capture on that same stream (torch.cuda.graph's own),
keep the inputs static.  Product code that wants replay should use mutual_info_img_txt.graphed.GraphedMiStep, which
keeps autograd out of the capture altogether."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mutual-information-multimodal_amd"))

import torch  # noqa: E402

from mutual_info_img_txt import mi_critics  # noqa: E402
from mutual_info_img_txt.model import BilinearCritic, make_mlp  # noqa: E402


def run(kind: str, precision: str) -> None:
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    b, d = (256, 128) if kind == "bilinear" else (96, 64)
    critic = (BilinearCritic(d, d) if kind == "bilinear" else make_mlp(2 * d, [128, 256])).to(dev)
    sid = torch.arange(b, dtype=torch.int64)
    sid[5] = sid[4]
    sid = sid.to(dev)  # a device tensor: a host list would be copied to the device inside the capture (not capturable)
    sx = torch.randn(b, d, device=dev, requires_grad=True)   # static inputs: refilled in place before a replay
    sy = torch.randn(b, d, device=dev, requires_grad=True)
    leaves = [sx, sy, *critic.parameters()]

    def step():
        loss = mi_critics.fused_mi_bound(sx, sy, sid, critic, "infonce", precision=precision)
        loss.backward()
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            for t in leaves:
                t.grad = None
            step()
    torch.cuda.current_stream().wait_stream(side)
    for t in leaves:
        t.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_loss = step()
    static_grads = [t.grad for t in leaves]

    for trial in range(2):
        nx, ny = torch.randn(b, d, device=dev), torch.randn(b, d, device=dev)
        with torch.no_grad():
            sx.copy_(nx)
            sy.copy_(ny)
        graph.replay()
        torch.cuda.synchronize()
        got = [float(static_loss)] + [g.clone() for g in static_grads]
        ex, ey = nx.clone().requires_grad_(True), ny.clone().requires_grad_(True)
        for p in critic.parameters():
            p.grad = None
        loss = mi_critics.fused_mi_bound(ex, ey, sid, critic, "infonce", precision=precision)
        loss.backward()
        ref = [float(loss), ex.grad, ey.grad] + [p.grad for p in critic.parameters()]
        assert got[0] == ref[0], (kind, trial, got[0], ref[0])
        for g, r in zip(got[1:], ref[1:]):
            assert torch.equal(g, r), (kind, trial)
        # restore the captured grad buffers as the leaves' grads (the eager pass replaced the parameters' ones)
        for t, g in zip(leaves, static_grads):
            t.grad = g
    print(f"capture ok: {kind} {precision}")


if __name__ == "__main__":
    run(sys.argv[1], sys.argv[2])
