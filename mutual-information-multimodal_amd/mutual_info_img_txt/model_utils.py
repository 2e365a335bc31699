"""Dataset contract of the MI training loop (reference model_utils.py:92-230).

Only what the hot path's CALLER depends on is reproduced -- the batch layout and the study-id semantics:

* a metadata table (CSV path or DataFrame) with a ``mimic_id`` column; ``study_id = MimicID.get_study_id(mimic_id)``, a
  string (reference model_utils.py:117-120); ``create_mi_pairs`` masks pairs with equal study ids (main_utils.py:105);
* ``__getitem__`` -> ``(img [1,H,W] float32, txt_ids, txt_masks, txt_segments, study_id, img_id)`` (model_utils.py:212),
  text tensors looked up by study id in the tokenised features (``report_id``, ``input_ids``, ``input_mask``,
  ``segment_ids``: model_utils.py:111-113);
* a sample whose image or tokens are missing is replaced by the cached DEFAULT sample but keeps ITS OWN study id
  (model_utils.py:138-148, 162-219), so the masking stays right.

Image decoding (OpenCV in the reference) is out of scope and pluggable: ``image_loader(path) -> 2-D array or None``; the
default loader reads ``.npy`` files, tests pass arrays directly.  JPEG loading, tokenisation and the augmentation
pipeline stay with the user's PyTorch-ROCm stack.
"""
from __future__ import annotations

import logging
import os
from typing import Callable, Optional, Sequence

import numpy as np
import torch

from .utils import MimicID


def _default_image_loader(path: str):
    if os.path.isfile(path + ".npy"):
        return np.load(path + ".npy", allow_pickle=False)
    if os.path.isfile(path) and path.endswith(".npy"):
        return np.load(path, allow_pickle=False)
    return None


class CXRImageReportDataset(torch.utils.data.Dataset):
    def __init__(self, text_token_features: Sequence, img_dir: str, dataset_metadata, data_key: str = 'mimic_id',
                 transform: Optional[Callable] = None, image_loader: Optional[Callable] = None):
        import pandas as pd
        self.all_txt_tokens = {f.report_id: f.input_ids for f in text_token_features}
        self.all_txt_masks = {f.report_id: f.input_mask for f in text_token_features}
        self.all_txt_segments = {f.report_id: f.segment_ids for f in text_token_features}
        meta = dataset_metadata if hasattr(dataset_metadata, "loc") else pd.read_csv(dataset_metadata)
        meta = meta.reset_index(drop=True).copy()
        meta['study_id'] = [MimicID.get_study_id(m) for m in meta['mimic_id']]
        self.dataset_metadata = meta
        self.img_dir = img_dir
        self.data_key = data_key
        self.transform = transform
        self.image_loader = image_loader or _default_image_loader
        self.image_ids = meta[data_key]
        self.default_img = None
        self.default_tokens = None
        self.default_token_masks = None
        self.default_token_segments = None
        self.logger = logging.getLogger(__name__)

    def set_default(self, img, tokens, token_masks, token_segments, study_id=None):
        """The sample substituted for unreadable ones (the reference captures it from the first batch,
        main_utils.py:195-199)."""
        self.default_img, self.default_tokens = img, tokens
        self.default_token_masks, self.default_token_segments = token_masks, token_segments

    def __len__(self):
        return len(self.image_ids)

    def __getitem__(self, idx):
        img_id, study_id = self.dataset_metadata.loc[idx, [self.data_key, 'study_id']]
        try:
            def text(table, default):
                v = table[study_id]
                return default if v is None else torch.as_tensor(v, dtype=torch.long)
            txt = text(self.all_txt_tokens, self.default_tokens)
            txt_masks = text(self.all_txt_masks, self.default_token_masks)
            txt_segments = text(self.all_txt_segments, self.default_token_segments)
            img = None
            try:
                img = self.image_loader(os.path.join(self.img_dir, str(img_id)))
                if img is not None:
                    if self.transform is not None:
                        img = self.transform(img)
                    # a tensor, like the default sample taken from a collated batch: default_collate cannot stack a
                    # batch that mixes arrays and tensors
                    img = torch.from_numpy(np.ascontiguousarray(np.expand_dims(np.asarray(img, dtype=np.float32), axis=0)))
            except Exception as e:  # an unreadable image falls back to the default one
                self.logger.error(f"Exception loading image for study_id={study_id}, img_id={img_id}: {e!r}")
                img = None
            if img is None:
                img = self.default_img
            return img, txt, txt_masks, txt_segments, study_id, img_id
        except Exception as e:  # missing tokens: the whole default sample, with this row's own study id
            self.logger.error(f"Exception raise for study_id={study_id}, img_id={img_id}: {e!r}")
            return (self.default_img, self.default_tokens, self.default_token_masks, self.default_token_segments,
                    study_id, img_id)


def max_normalise(img):
    """The last step of the reference's transform (main_utils.py:42-43): divide by max(1e-3, max)."""
    img = np.asarray(img, dtype=np.float32)
    return img / max(1e-3, float(img.max()))


def build_training_imagereportset(text_token_features, img_dir, img_size: int, dataset_metadata='../data/training.csv',
                                  random_degrees=(-20, 20), random_translate=(0.1, 0.1), image_loader=None):
    """Reference main_utils.py:28-50.  The random-affine augmentation needs torchvision, which is not part of this
    image: images are centre-cropped to ``img_size`` and max-normalised; pass your own ``transform`` through
    ``CXRImageReportDataset`` for augmentation."""
    def transform(img):
        img = np.asarray(img)
        h, w = img.shape[-2:]
        top, left = max((h - img_size) // 2, 0), max((w - img_size) // 2, 0)
        return max_normalise(img[..., top:top + img_size, left:left + img_size])
    return CXRImageReportDataset(text_token_features=text_token_features, img_dir=img_dir,
                                 dataset_metadata=dataset_metadata, transform=transform, image_loader=image_loader)
