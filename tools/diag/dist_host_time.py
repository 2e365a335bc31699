"""Diagnostic: where the HOST time of a sharded step goes (one-rank RCCL rehearsal).  Times every phase of
GlobalBatchGraphStep.step_eager() with perf_counter (no synchronisation inside a step: these are issue costs).
   python tools/diag/dist_host_time.py [B] [d]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mutual-information-multimodal_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
from mutual_info_img_txt import distributed as mid

b = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl")
x, y = torch.randn(b, d, device=dev), torch.randn(b, d, device=dev)
w = torch.randn(d, d, device=dev) / d ** 0.5
sid = torch.arange(b, device=dev)
st = mid.GlobalBatchGraphStep(x, y, sid, [w], "infonce", "bf16", critic="bilinear", capture=False)
phases = ["_gather_inputs", "_forward", "_gather_records", "_merge_backward", "_exchange_gradients"]
acc = {p: 0.0 for p in phases}
for it in range(300):
    if it == 100:
        torch.cuda.synchronize(); acc = {p: 0.0 for p in phases}; t_all = time.perf_counter()
    for p in phases:
        t0 = time.perf_counter()
        getattr(st, p)()
        acc[p] += time.perf_counter() - t0
torch.cuda.synchronize()
total = (time.perf_counter() - t_all) / 200
print(f"B={b} d={d}: {total*1e6:.1f} us per step (host issue + final drain)")
for p in phases:
    print(f"  {p:22s} {acc[p]/200*1e6:7.1f} us")
dist.destroy_process_group()
