"""Embedding-separability metric of the reference's ``validate.py`` (:16-49): the generalised discrimination value (GDV)
between the image embeddings of two classes.  It is the acceptance check named in SURVEY.md 8f rank 4: embeddings learned
through the MI355X critic path must score like those of the reference.

``gdv_calculation(positive_embeddings, negative_embeddings)`` restates the reference formula on torch tensors (any device;
float64 internally), including its quirks:

* each class is z-scored with ITS OWN mean / population standard deviation per feature (``StandardScaler`` fitted per
  class, validate.py:16-21; zero-variance features are left at 0 as sklearn's scaler does);
* the "mean" intra-class distance divides the sum of all pairwise Euclidean distances (both orders) by ``T (T - 1) / 2``
  with ``T = rows * columns`` -- the number of matrix ENTRIES, not of samples -- and the inter-class one by
  ``T_pos * T_neg`` (validate.py:23-35);
* result = ``((intra_pos + intra_neg) / 2 - inter) / sqrt(n_pos + n_neg)`` (validate.py:37-49).

The reference module trains a classifier at import time (validate.py:52-172); only the metric is reproduced.
"""
import math

import torch


def z_scored_transform(source_tensor):
    x = torch.as_tensor(source_tensor, dtype=torch.float64)
    mean = x.mean(dim=0, keepdim=True)
    std = x.std(dim=0, unbiased=False, keepdim=True)
    std = torch.where(std == 0, torch.ones_like(std), std)  # sklearn: zero-variance features are only centred
    return (x - mean) / std


def mean_intra_class_distance(items):
    items = torch.as_tensor(items, dtype=torch.float64)
    total_items = items.shape[0] * items.shape[1]
    return float(torch.cdist(items, items).sum()) * 2 / (total_items * (total_items - 1))


def mean_inter_class_distance(source, dest):
    source = torch.as_tensor(source, dtype=torch.float64)
    dest = torch.as_tensor(dest, dtype=torch.float64)
    return float(torch.cdist(source, dest).sum()) / ((source.shape[0] * source.shape[1]) * (dest.shape[0] * dest.shape[1]))


def gdv_calculation(positive_embeddings, negative_embeddings):
    pos = z_scored_transform(positive_embeddings)
    neg = z_scored_transform(negative_embeddings)
    intra = (mean_intra_class_distance(pos) + mean_intra_class_distance(neg)) / 2
    inter = mean_inter_class_distance(pos, neg)
    return (intra - inter) / math.sqrt(len(positive_embeddings) + len(negative_embeddings))
