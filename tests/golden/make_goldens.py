#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE'S OWN FUNCTIONS.

Runs only in the build container (needs /root/reference).  The reference never travels to the GPU box, so
the outputs are committed as small .npz fixtures next to this script.

How the reference is executed (nothing is copied into this repository):

* ``mutual_info_img_txt/mi_critics.py`` (dv_bound_loss :3-12, infonce_bound_loss :14-23) only imports torch
  and is imported as-is from /root/reference.
* ``make_mlp`` (mutual_info_img_txt/model.py:18-32) and ``MultiModalManager.create_mi_pairs``
  (mutual_info_img_txt/main_utils.py:80-110) live in modules whose top-level imports need packages that are
  absent here (torchvision, pytorch_transformers, cv2, pytorch_grad_cam).  Their function definitions are
  located with ``ast`` in the reference's source files at generation time, compiled from that AST and executed
  unmodified in a namespace that only holds ``torch`` / ``torch.nn`` -- the reference's own code runs, the
  module-level imports it does not touch are skipped.

Inputs are the closed-form cases of ``oracle.mi_oracle.synthetic_case`` (RNG-free, bit-reproducible), so a
fixture stores outputs only.  Each case is also run in fp64 (the "G5 twins") for tolerance calibration.
"""
from __future__ import annotations

import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MI_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import mi_oracle as orc  # noqa: E402


def _extract_function(path: str, name: str, cls: str | None = None):
    """Compile ONE function definition found in the reference source file ``path`` and return it."""
    with open(path, "r") as fh:
        tree = ast.parse(fh.read(), filename=path)
    body = tree.body
    if cls is not None:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    node = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == name)
    mod = ast.Module(body=[node], type_ignores=[])
    ns = {"torch": torch, "nn": torch.nn}
    exec(compile(mod, path, "exec"), ns)
    return ns[name]


def load_reference():
    if not os.path.isdir(REF):
        raise SystemExit(f"reference not found at {REF}: goldens can only be regenerated in the build container")
    sys.path.insert(0, REF)
    from mutual_info_img_txt import mi_critics as ref_critics  # the reference module itself
    make_mlp = _extract_function(os.path.join(REF, "mutual_info_img_txt", "model.py"), "make_mlp")
    create_mi_pairs = _extract_function(os.path.join(REF, "mutual_info_img_txt", "main_utils.py"),
                                        "create_mi_pairs", cls="MultiModalManager")
    return ref_critics, make_mlp, create_mi_pairs


def g1_bound(ref_critics):
    """G1: bound only, on closed-form logits."""
    out = {}
    cases = [(4, 14), (8, 64), (64, 4096), (32, 1016), (5, 6)]
    for b, n in cases:
        logits = (orc.hash_uniform((n, 1), 100 + n) * 6.0)
        out.update(_bound_case(ref_critics, f"b{b}_n{n}", logits, b))
    # extreme magnitudes: LSE stability
    logits = orc.hash_uniform((300, 1), 777) * 160.0  # +-80
    out.update(_bound_case(ref_critics, "extreme_b16_n300", logits, 16))
    return out


def _bound_case(ref_critics, tag, logits, pos):
    res = {f"{tag}/pos_size": np.int64(pos), f"{tag}/n": np.int64(logits.shape[0])}
    dev = torch.device("cpu")
    for est, fn in (("dv", ref_critics.dv_bound_loss), ("infonce", ref_critics.infonce_bound_loss)):
        for dt, name in ((torch.float32, "f32"), (torch.float64, "f64")):
            lg = logits.to(dt).clone().requires_grad_(True)
            loss = fn(lg, pos, dev)
            loss.sum().backward()
            res[f"{tag}/{est}/{name}/loss"] = loss.detach().numpy()
            res[f"{tag}/{est}/{name}/loss_shape"] = np.array(loss.shape, dtype=np.int64)
            res[f"{tag}/{est}/{name}/grad"] = lg.grad.numpy().reshape(-1)
    return res


def g2_order(create_mi_pairs):
    """G2: row order, recovered from the reference's output by tagging each embedding with its index."""
    out = {}
    cases = {
        "b4_dups": ["a", "b", "b", "c"],
        "b8_unique": [str(n) for n in range(8)],
        "b6_alldup": ["z"] * 6,
        "b7_mixed": ["1", "2", "1", "3", "3", "3", "4"],
        "b1": ["only"],
    }
    for tag, sid in cases.items():
        b = len(sid)
        x = torch.arange(b, dtype=torch.float32).reshape(b, 1)
        y = torch.arange(b, dtype=torch.float32).reshape(b, 1) + 1000.0
        rows = create_mi_pairs(None, x, y, sid, torch.device("cpu"))
        out[f"{tag}/i"] = rows[:, 0].numpy().astype(np.int64)
        out[f"{tag}/j"] = (rows[:, 1].numpy() - 1000.0).astype(np.int64)
        out[f"{tag}/sid"] = np.array(sid)
    return out


def _digest(t: torch.Tensor):
    """Compact digest of a large gradient: fixed sampled entries, row/col sums, Frobenius norm."""
    a = t.detach().double().numpy()
    flat = a.reshape(-1)
    idx = (np.arange(64, dtype=np.int64) * 7919 + 13) % flat.size
    d = {"sample_idx": idx, "sample": flat[idx], "fro": np.array(np.sqrt((flat ** 2).sum())),
         "sum": np.array(flat.sum())}
    if a.ndim == 2:
        d["row_sums"] = a.sum(1)
        d["col_sums"] = a.sum(0)
    return d


def g3_full_step(ref_critics, make_mlp, create_mi_pairs):
    """G3/G4/G5: full step through the reference's own pair builder, make_mlp critic and bound."""
    out = {}
    cases = [
        ("b8_d768", 8, 768, 768, False), ("b32_d768", 32, 768, 768, False),
        ("b16_d128", 16, 128, 128, False), ("b32_d128", 32, 128, 128, False),
        ("b16_d768_dup", 16, 768, 768, True),
        ("b32_d128_dup", 32, 128, 128, True), ("b24_d96x160_dup", 24, 96, 160, True),
    ]
    dev = torch.device("cpu")
    for salt, (tag, b, di, dt_, dup) in enumerate(cases):
        for dtype, name in ((torch.float32, "f32"), (torch.float64, "f64")):
            x, y, sid, params = orc.synthetic_case(b, di, dt_, salt=salt, dup=dup, dtype=dtype)
            mlp = make_mlp(di + dt_, [1024, 512]).to(dtype)
            with torch.no_grad():
                for p, v in zip(mlp.parameters(), params):
                    p.copy_(v)
            for est, fn in (("dv", ref_critics.dv_bound_loss), ("infonce", ref_critics.infonce_bound_loss)):
                xl = x.clone().requires_grad_(True)
                yl = y.clone().requires_grad_(True)
                mlp.zero_grad()
                mi_input = create_mi_pairs(None, xl, yl, sid, dev)
                mi_output = mlp(mi_input)
                loss = fn(mi_output, b, dev)
                loss.sum().backward()
                k = f"{tag}/{est}/{name}"
                out[f"{k}/loss"] = loss.detach().numpy()
                out[f"{k}/n_rows"] = np.int64(mi_input.shape[0])
                if est == "dv":
                    out[f"{tag}/{name}/scores"] = mi_output.detach().numpy().reshape(-1)
                grads = [xl.grad.clone(), yl.grad.clone()] + [p.grad.clone() for p in mlp.parameters()]
                if est == "dv":
                    dv_grads = grads
                    out[f"{k}/dx"] = xl.grad.numpy()
                    out[f"{k}/dy"] = yl.grad.numpy()
                    for pn, p in zip(("w1", "b1", "w2", "b2", "w3", "b3"), mlp.parameters()):
                        for dk, dv in _digest(p.grad).items():
                            out[f"{k}/d{pn}/{dk}"] = dv
                else:  # the two estimators differ by a constant: store only how far the gradients are apart
                    out[f"{k}/grad_maxdiff_vs_dv"] = np.array(
                        max(float((a - b_).abs().max()) for a, b_ in zip(grads, dv_grads)))
        out[f"{tag}/meta"] = np.array([b, di, dt_, int(dup), salt], dtype=np.int64)
    return out


def g0_make_mlp(make_mlp):
    """Structure of the reference critic: layer types and state-dict keys (model.py:18-32)."""
    mlp = make_mlp(1536, [1024, 512])
    keys = list(mlp.state_dict().keys())
    shapes = [tuple(v.shape) for v in mlp.state_dict().values()]
    kinds = [type(m).__name__ for m in mlp]
    return {"keys": np.array(keys), "shapes": np.array([str(s) for s in shapes]), "kinds": np.array(kinds)}


def main():
    torch.set_num_threads(8)
    ref_critics, make_mlp, create_mi_pairs = load_reference()
    np.savez_compressed(os.path.join(HERE, "g0_make_mlp.npz"), **g0_make_mlp(make_mlp))
    np.savez_compressed(os.path.join(HERE, "g1_bound.npz"), **g1_bound(ref_critics))
    np.savez_compressed(os.path.join(HERE, "g2_order.npz"), **g2_order(create_mi_pairs))
    np.savez_compressed(os.path.join(HERE, "g3_full_step.npz"), **g3_full_step(ref_critics, make_mlp, create_mi_pairs))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
