"""Diagnostic: s_memtime stamps of the fused bilinear kernel (needs lib_stamps/, `make STAMPS=1`):
   MI_STAMP_KERNEL="fused S" python tools/diag/stamps_flash.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
os.environ.setdefault("MI_CRITIC_LIB", os.path.join(ROOT, "mutual-information-multimodal_amd", "lib_stamps", "libmi_critic_hip.so"))
os.environ.setdefault("MI_STAMP_KERNEL", "fused S")
sys.path.insert(0, os.path.join(ROOT, "mutual-information-multimodal_amd"))
sys.path.insert(0, ROOT)
import ctypes
import numpy as np
import torch
from mutual_info_img_txt import mi_critics, _hip
from mutual_info_img_txt.model import BilinearCritic

dev = torch.device("cuda:0")
b, d = int(os.environ.get("B", 4096)), int(os.environ.get("D", 512))
x = torch.randn(b, d, device=dev, requires_grad=True)
y = torch.randn(b, d, device=dev, requires_grad=True)
sid = torch.arange(b, device=dev)
critic = BilinearCritic(d, d).to(dev)
lib = _hip.load()
lib.mi_debug_set_stamps.argtypes = [ctypes.c_void_p]
buf = torch.zeros(8192 * 16, dtype=torch.int64, device=dev)
lib.mi_debug_set_stamps(buf.data_ptr())

def step():
    x.grad = None; y.grad = None
    for p in critic.parameters(): p.grad = None
    loss = mi_critics.fused_mi_bound(x, y, sid, critic, "infonce", precision="bf16")
    loss.sum().backward()

for _ in range(5): step()
torch.cuda.synchronize()
buf.zero_()
step()
torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(-1, 16)
s = s[s[:, 0] != 0]
print("workgroups stamped:", len(s))
t0 = s[:, 0].min()
def seg(name, a, c):
    d_ = s[:, c] - s[:, a]
    print(f"{name:34s} median {np.median(d_):9.0f}  p10 {np.percentile(d_,10):9.0f}  p90 {np.percentile(d_,90):9.0f}")
seg("prologue (entry -> first barrier)", 0, 1)
seg("  Q loads issued", 0, 11)
seg("  id flags (byte loads + ballot)", 11, 12)
seg("  24 LDS-DMA pieces issued", 12, 13)
seg("  id copy, accumulator init", 13, 14)
seg("  s_waitcnt vmcnt(0)", 14, 15)
seg("  barrier", 15, 1)
seg("tile 0 scores + pipelined loop", 1, 8)
seg("last tile (softmax + output product)", 8, 9)
seg("epilogue (records + slab stores)", 9, 10)
seg("whole workgroup", 0, 10)
print("--- one pipelined iteration (t = nt/2)")
seg("wait + barrier", 2, 3)
seg("scores(t+1) | softmax(t) | output(t)", 3, 4)
print(f"kernel span {s[:,10].max()-s[:,0].min()} ticks")
# who is slow?  (the launch ends with its slowest workgroup).  Workgroup L runs on XCD L % 8; with the row-block mapping
# unit = L % 8 + 8 * ((L // 8) // 4), split = (L // 8) % 4, problem 0 for the first b / 128 units
full = buf.cpu().numpy().reshape(-1, 16)
idx = np.nonzero(full[:, 0] != 0)[0]
loop = full[idx, 8] - full[idx, 1]
whole = full[idx, 10] - full[idx, 0]
n_rb = b // 128
for name, key in (("XCD", idx % 8), ("split", (idx // 8) % 4), ("problem", ((idx % 8) + 8 * ((idx // 8) // 4)) // n_rb)):
    print(f"loop cycles by {name}:", {int(k): int(np.median(loop[key == k])) for k in np.unique(key)})
order = np.argsort(whole)[::-1][:8]
print("slowest workgroups (L, xcd, split, whole, loop):", [(int(idx[o]), int(idx[o] % 8), int((idx[o] // 8) % 4), int(whole[o]), int(loop[o])) for o in order])
g = full[idx]
has = g[:, 5] != 0
if has.any():
    d56 = g[has, 6] - g[has, 5]
    print(f"general mask path of wave 0 (slots 5 -> 6): {int(has.sum())} workgroups, median {np.median(d56):.0f} cycles, max {d56.max()}")
    print("loop of workgroups whose wave 0 took it vs not:", int(np.median(loop[has])), int(np.median(loop[~has])))
    if (g[has, 7] != 0).all():
        print(f"   positives block (5 -> 7) median {np.median(g[has, 7] - g[has, 5]):.0f}, id compares (7 -> 6) median {np.median(g[has, 6] - g[has, 7]):.0f}")
