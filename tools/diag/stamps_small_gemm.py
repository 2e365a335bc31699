"""Diagnostic: s_memtime stamps of the 4-stage short-K GEMM launch selected by MI_STAMP_KERNEL (default: the backward's
dW | dX launch).  Needs lib_stamps/ (make STAMPS=1).  Slots: 0 entry, 1 second k-step reached, 2 loop end, 3 block end."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
os.environ["MI_CRITIC_LIB"] = os.path.join(ROOT, "mutual-information-multimodal_amd", "lib_stamps", "libmi_critic_hip.so")
os.environ.setdefault("MI_STAMP_KERNEL", "bilinear dW")
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
    loss = mi_critics.fused_mi_bound(x, y, sid, critic, "infonce", precision="bf16")
    loss.sum().backward()

for _ in range(5): step()
torch.cuda.synchronize()
buf.zero_()
step()
torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(-1, 8)
s = s[s[:, 0] != 0]
print("kernel filter:", os.environ["MI_STAMP_KERNEL"], "blocks stamped:", len(s))
for i, n in enumerate(["entry -> first k-step done", "remaining k-steps", "epilogue"]):
    dlt = s[:, i + 1] - s[:, i]
    print(f"   {n:28s} median {np.median(dlt):8.0f}  p10 {np.percentile(dlt,10):8.0f}  p90 {np.percentile(dlt,90):8.0f}")
print(f"   whole block                  median {np.median(s[:,3]-s[:,0]):8.0f}  p90 {np.percentile(s[:,3]-s[:,0],90):8.0f}")
