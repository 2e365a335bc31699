"""Diagnostic: per-phase s_memtime medians of one stamped bf16 GEMM launch (needs lib_stamps/, make STAMPS=1)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
os.environ["MI_CRITIC_LIB"] = os.path.join(ROOT, "mutual-information-multimodal_amd", "lib_stamps", "libmi_critic_hip.so")
sys.path.insert(0, os.path.join(ROOT, "mutual-information-multimodal_amd"))
sys.path.insert(0, ROOT)
import ctypes
import numpy as np
import torch
from mutual_info_img_txt import mi_critics, _hip
from mutual_info_img_txt.model import BilinearCritic

dev = torch.device("cuda:0")
b, d = 4096, 512
x = torch.randn(b, d, device=dev, requires_grad=True)
y = torch.randn(b, d, device=dev, requires_grad=True)
sid = torch.arange(b, device=dev)
critic = BilinearCritic(d, d).to(dev)
lib = _hip.load()
lib.mi_debug_set_stamps.argtypes = [ctypes.c_void_p]
buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
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
s = buf.cpu().numpy().reshape(-1, 8)
s = s[s[:, 0] != 0]
print("kernel filter:", os.environ.get("MI_STAMP_KERNEL"), "workgroups stamped:", len(s))
t0 = s[:, 0].min()
names = ["entry->tile0 landed", "tile0->nt/4", "nt/4->3nt/4", "3nt/4->loop end", "epilogue"]
for i, n in enumerate(names):
    dlt = s[:, i + 1] - s[:, i]
    print(f"{n:24s} median {np.median(dlt):9.0f}  p10 {np.percentile(dlt,10):9.0f}  p90 {np.percentile(dlt,90):9.0f} cycles")
print(f"start skew (entry - first entry): median {np.median(s[:,0]-t0):.0f} max {np.max(s[:,0]-t0):.0f}")
print(f"whole WG: median {np.median(s[:,5]-s[:,0]):.0f}; kernel span {s[:,5].max()-t0} ticks of s_memtime")
if s[:, 6].any():
    print("epilogue halves (slots 4 -> 6 -> 7 -> 5):",
          f"first half {np.median(s[:,6]-s[:,4]):.0f}, second half {np.median(s[:,7]-s[:,6]):.0f}, tail {np.median(s[:,5]-s[:,7]):.0f} cycles")
