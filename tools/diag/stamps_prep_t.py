"""Diagnostic: s_memtime stamps of the fused conversion + T = X W launch (needs lib_stamps/, make STAMPS=1).
Slots: 0 entry, 1 first tile in LDS, 2 first k-step's MFMAs issued, 3 first k-step's barrier passed, 4 loop end, 5 block end."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
os.environ["MI_CRITIC_LIB"] = os.path.join(ROOT, "mutual-information-multimodal_amd", "lib_stamps", "libmi_critic_hip.so")
os.environ["MI_STAMP_KERNEL"] = "bilinear prep + T"
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
n_t = (b // 128) * (d // 128)
live = s[:, 0] != 0
t0 = s[live, 0].min()
T = s[:n_t]
C = s[n_t:][live[n_t:]]
print("blocks stamped:", int(live.sum()), "T tiles:", n_t, "conversion blocks:", len(C))
print(f"kernel span: {s[live, 5].max() - t0} cycles")
print(f"T tiles : start median {np.median(T[:,0]-t0):.0f} max {np.max(T[:,0]-t0):.0f};  end median {np.median(T[:,5]-t0):.0f} max {np.max(T[:,5]-t0):.0f}")
for i, n in enumerate(["entry -> first tile in LDS", "first k-step MFMAs", "first k-step store+barrier", "remaining k-steps", "epilogue"]):
    dlt = T[:, i + 1] - T[:, i]
    print(f"   {n:28s} median {np.median(dlt):8.0f}  p10 {np.percentile(dlt,10):8.0f}  p90 {np.percentile(dlt,90):8.0f}")
dur = C[:, 5] - C[:, 0]
print(f"conversion blocks: duration median {np.median(dur):.0f} p90 {np.percentile(dur,90):.0f};  start median {np.median(C[:,0]-t0):.0f} p90 {np.percentile(C[:,0]-t0,90):.0f} max {np.max(C[:,0]-t0):.0f};  last end {np.max(C[:,5]-t0):.0f}")
hist, edges = np.histogram(C[:, 0] - t0, bins=8)
print("conversion block start histogram:", list(zip(edges[:-1].astype(int).tolist(), hist.tolist())))
# (s_memtime values are only comparable within one workgroup: durations above, no absolute timeline)
