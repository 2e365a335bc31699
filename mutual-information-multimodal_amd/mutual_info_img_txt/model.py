"""Critic modules of the MI path.  Mirrors the part of the reference's ``mutual_info_img_txt/model.py`` that the hot
path uses: ``make_mlp`` (reference model.py:18-32).  The image / text encoders (reference model.py:54-595) stay
ordinary PyTorch-ROCm modules and are out of scope.

``BilinearCritic`` and ``SeparableCritic`` are extensions named by BASELINE.json (no reference code)."""
import math

import torch
import torch.nn as nn

# the encoder -> critic interface and the checkpoint formats live in encoders.py; re-exported here because the reference
# keeps them in model.py (`from mutual_info_img_txt.model import build_resnet_model, build_bert_model, ImageReportModel`)
from .encoders import (BasicBlock, ImageReportModel, ProjectionHead, ResNet256_6_2_1, TextBert,  # noqa: F401
                       build_bert_model, build_resnet256_6_2_1, build_resnet_model, conv1x1, conv3x3,
                       convert_legacy_keys, image_state_from_checkpoint)


def make_mlp(input_dim, hidden_dims: list, output_dim=1, activation='relu'):
    """Same contract as the reference's make_mlp (model.py:18-32): an nn.Sequential of Linear/ReLU pairs followed by a
    final Linear, default PyTorch initialisation, state-dict keys 0.weight, 0.bias, 2.weight, ...  Unknown activation
    names raise KeyError as in the reference."""
    act = {'relu': nn.ReLU}[activation]
    dims = [input_dim] + list(hidden_dims)
    layers = []
    for fan_in, fan_out in zip(dims[:-1], dims[1:]):
        layers.append(nn.Linear(fan_in, fan_out))
        layers.append(act())
    layers.append(nn.Linear(dims[-1], output_dim))
    return nn.Sequential(*layers)


class BilinearCritic(nn.Module):
    """S[i,j] = x_i^T W y_j  (extension; BASELINE.json headline critic)."""

    def __init__(self, d_img: int, d_txt: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(d_img, d_txt))
        nn.init.normal_(self.weight, std=1.0 / math.sqrt(d_img))  # W ~ N(0, 1/d)

    def forward(self, embedding_img, embedding_txt):
        """Reference-style eager scores (plain torch ops); the fused path is mi_critics.fused_mi_bound."""
        return (embedding_img @ self.weight) @ embedding_txt.t()


class SeparableCritic(nn.Module):
    """S[i,j] = (x_i Wg) . (y_j Wh)  (extension; BASELINE.json configs[1]).  Through ``mi_critics.fused_mi_bound`` the
    projections, the B x B contraction, the bound and every gradient run on the HIP library (``mi_separable_fwd/bwd``);
    ``forward`` / ``project_*`` below are the eager torch form, kept for reference-style use."""

    def __init__(self, d_img: int, d_txt: int, d_proj: int):
        super().__init__()
        self.wg = nn.Parameter(torch.empty(d_img, d_proj))
        self.wh = nn.Parameter(torch.empty(d_txt, d_proj))
        nn.init.normal_(self.wg, std=1.0 / math.sqrt(d_img))
        nn.init.normal_(self.wh, std=1.0 / math.sqrt(d_txt))

    def project_img(self, x):
        return x @ self.wg

    def project_txt(self, y):
        return y @ self.wh

    def forward(self, embedding_img, embedding_txt):
        return self.project_img(embedding_img) @ self.project_txt(embedding_txt).t()
