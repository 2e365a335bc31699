#!/usr/bin/env python3
"""Per-kernel times of the concat-MLP step (the reference's critic) in the given precision modes, one process, interleaved
rounds (guide rule 24).  usage: concat_time.py [B] [d] [modes, comma separated] [rounds]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mutual-information-multimodal_amd"))
import torch  # noqa: E402

from mutual_info_img_txt import _hip  # noqa: E402
from mutual_info_img_txt.graphed import GraphedMiStep  # noqa: E402
from mutual_info_img_txt.model import make_mlp  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
modes = (sys.argv[3] if len(sys.argv) > 3 else "bf16,f16").split(",")
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
torch.manual_seed(0)
mlp = make_mlp(2 * d, [1024, 512]).to(dev)
gen = torch.Generator().manual_seed(3)
x, y = torch.randn(b, d, generator=gen).to(dev), torch.randn(b, d, generator=gen).to(dev)
steps = {m: GraphedMiStep(mlp, b, d, d, "infonce", m, dev, capture=False) for m in modes}
for s in steps.values():
    s.set_inputs(x, y, torch.arange(b))
    s.step_eager()
torch.cuda.synchronize()
agg = {m: {} for m in modes}
for r in range(rounds):
    for m in modes:
        with _hip.kernel_profile() as prof:
            steps[m].step_eager()
        for name, v in prof.by_name().items():
            agg[m].setdefault(name, []).append(v["ms_total"])
for m in modes:
    tot = 0.0
    print(f"== {m}: loss {float(steps[m].loss_buf):.6f}")
    for name, v in sorted(agg[m].items(), key=lambda kv: -min(kv[1])):
        tot += min(v)
        if min(v) > 0.02:
            print(f"   {name:44s} min {min(v):8.3f} ms  median {sorted(v)[len(v) // 2]:8.3f}")
    print(f"   sum of kernel minima: {tot:.3f} ms")
