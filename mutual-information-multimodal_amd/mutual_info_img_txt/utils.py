"""Study-id contract of the MI path (hazard H5).

Mirrors the part of the reference's ``mutual_info_img_txt/utils.py`` that the hot path depends on:

* ``MimicID`` (reference utils.py:3-18): ``p{subject}_s{study}_{dicom}`` identifiers; ``MimicID.get_study_id`` is what the
  reference's dataset uses to produce the ``study_id`` strings that ``create_mi_pairs`` compares (model_utils.py:157-158,
  main_utils.py:105).

and adds what a sharded run needs: a DETERMINISTIC ``study id -> int64`` code that is the same in every process.  (A
first-seen numbering, as a single process could use, would give the same study different codes on different ranks and
mask the wrong pairs after the all-gather.)
"""
from __future__ import annotations

import hashlib
from typing import Iterable, Sequence, Union

import torch

_INT64_MAX = (1 << 63) - 1


class MimicID:
    """Identifier triple of a MIMIC-CXR image, printed as ``p{subject_id}_s{study_id}_{dicom_id}`` (reference
    utils.py:3-14).  The three parts are kept as strings, as in the reference."""

    subject_id = ''
    study_id = ''
    dicom_id = ''

    def __init__(self, subject_id, study_id, dicom_id):
        self.subject_id = str(subject_id)
        self.study_id = str(study_id)
        self.dicom_id = str(dicom_id)

    def __str__(self):
        return f"p{self.subject_id}_s{self.study_id}_{self.dicom_id}"

    @staticmethod
    def get_study_id(mimic_id: str):
        """The study part of a printed id (reference utils.py:16-18): second ``_`` field without its leading ``s``."""
        return mimic_id.split('_')[1][1:]


def study_id_to_int64(study_id) -> int:
    """Process-independent int64 code of one study id; equal ids <=> equal codes.

    * integers (Python / numpy / 0-d tensors) map to themselves;
    * strings of decimal digits -- what ``MimicID.get_study_id`` yields for MIMIC-CXR (e.g. ``"50414267"``) -- map to
      their value when it fits in 62 bits;
    * anything else maps to the first 8 bytes of BLAKE2b over its ``str``, with the top bit pattern ``01`` so that a
      hashed code can never collide with a parsed one (parsed codes are < 2**62).  Two DIFFERENT non-numeric ids collide
      with probability ~n^2 / 2**63 per batch, i.e. never in practice; numeric ids are exact.
    """
    if torch.is_tensor(study_id):
        study_id = study_id.item()
    if isinstance(study_id, bool):
        return int(study_id)
    if isinstance(study_id, int) or hasattr(study_id, "__index__"):
        v = int(study_id)
        if -(1 << 63) <= v <= _INT64_MAX:
            return v
    text = study_id if isinstance(study_id, str) else str(study_id)
    if text.isascii() and text.isdigit() and len(text) <= 18 and (text == "0" or text[0] != "0"):
        return int(text)  # < 10**18 < 2**62; leading zeros are kept distinct by the hash branch below
    digest = hashlib.blake2b(text.encode("utf-8"), digest_size=8).digest()
    h = int.from_bytes(digest, "little") & ((1 << 62) - 1)
    return h | (1 << 62)


def study_ids_to_tensor(study_id: Union[Sequence, Iterable, torch.Tensor], device=None) -> torch.Tensor:
    """int64 tensor of ``study_id_to_int64`` codes (a tensor input is only cast and moved)."""
    if torch.is_tensor(study_id):
        return study_id.to(device=device, dtype=torch.int64).contiguous()
    return torch.tensor([study_id_to_int64(s) for s in study_id], dtype=torch.int64, device=device)
