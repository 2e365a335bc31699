#!/usr/bin/env python3
"""Where do the bf16 concat-MLP gradients differ from the fully rounded oracle at the reference's widths?
Prints, per gradient, the error relative to max|ref| and the index pattern of the worst elements."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mutual-information-multimodal_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import mi_oracle as orc  # noqa: E402
from test_parity_configs import GRAD_NAMES, _concat_all_grads, _dup_ids  # noqa: E402

b, d, h1, h2, rb = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (512, 768, 1024, 512, 64))]
dev = torch.device("cuda:0")
x, y, _, params = orc.synthetic_case(b, d, d, h1=h1, h2=h2, salt=b // 8)
sid = _dup_ids(b)
loss16, g16 = _concat_all_grads(dev, x, y, sid, params, (h1, h2), "dv", "bf16")
loss32, g32 = _concat_all_grads(dev, x, y, sid, params, (h1, h2), "dv", "f32")
p64 = [p.double() for p in params]
o = orc.concat_step_rounded(x.double(), y.double(), sid, p64, "dv", row_block=rb)
ofwd = orc.concat_matrix_step(x.double(), y.double(), sid, p64, "dv", round_fn=orc.round_bf16, row_block=rb)
oex = orc.concat_matrix_step(x.double(), y.double(), sid, p64, "dv", row_block=rb)
print("loss bf16 kernel", float(loss16), "rounded oracle", float(o["loss"]), "f32 kernel", float(loss32), "exact", float(oex["loss"]))
refs = [o["dx"], o["dy"]] + list(o["dparams"])
refs_f = [ofwd["dx"], ofwd["dy"]] + list(ofwd["dparams"])
refs_e = [oex["dx"], oex["dy"]] + list(oex["dparams"])
for name, got, g3, ref, rfw, rex in zip(GRAD_NAMES, g16, g32, refs, refs_f, refs_e):
    ref, rfw, rex = ref.reshape(got.shape), rfw.reshape(got.shape), rex.reshape(got.shape)
    sc = float(ref.abs().max())
    err = (got.double() - ref).abs() / sc
    print(f"{name:4s} max|ref| {sc:.3e}  kernel16-vs-rounded {float(err.max()):.2e}  rounded-vs-fwdrounded "
          f"{float((ref - rfw).abs().max()) / sc:.2e}  rounded-vs-exact {float((ref - rex).abs().max()) / sc:.2e}  "
          f"kernel32-vs-exact {float((g3.double() - rex).abs().max()) / sc:.2e}  rms err {float(err.pow(2).mean().sqrt()):.2e}")
    if got.dim() == 2 and float(err.max()) > 1e-3:
        flat = torch.topk(err.flatten(), 8).indices
        print("     worst:", [(int(f // got.shape[1]), int(f % got.shape[1]), f"{float(err.flatten()[f]):.1e}") for f in flat])
        print("     mean err by row block of 8ths:", [f"{float(c.mean()):.1e}" for c in err.chunk(8, 0)])
        print("     mean err by col block of 8ths:", [f"{float(c.mean()):.1e}" for c in err.chunk(8, 1)])
