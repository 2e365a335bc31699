"""Entry point kept from the reference (multi_modal.py:15-67): ``train_mutual_information(args, device)``.

The reference builds the ResNet/BERT encoders, tokenises MIMIC-CXR reports and calls MultiModalManager.train; none of
that is available offline and the encoders are out of scope (they stay ordinary PyTorch-ROCm modules).  This build
trains the MI critic on synthetic embeddings of the configured width through the fused HIP path."""
import logging
import os

import torch

from mutual_info_img_txt.main_utils import MultiModalManager


def synthetic_embedding_source(args, device):
    gen = torch.Generator(device="cpu").manual_seed(int(getattr(args, "seed", 0)))
    b, d_img, d_txt = args.batch_size, args.embed_dim_img, args.embed_dim_txt
    study_id = [str(n) for n in range(b)]
    # a fixed pool of correlated pairs so that the bound can actually rise during training
    base = torch.randn(b, min(d_img, d_txt), generator=gen)

    def source(step):
        noise_i = torch.randn(b, d_img, generator=gen) * 0.5
        noise_t = torch.randn(b, d_txt, generator=gen) * 0.5
        img = noise_i
        txt = noise_t
        k = base.shape[1]
        img[:, :k] += base
        txt[:, :k] += base
        return img.to(device), txt.to(device), study_id
    return source


def train_mutual_information(args, device):
    os.makedirs(args.save_directory, exist_ok=True)
    # same file name / mode as the reference (multi_modal.py:27-30); force=True so that an already configured root
    # logger (a test runner, a notebook) does not silently swallow the training log
    logging.basicConfig(filename=os.path.join(args.save_directory, 'training_MI.log'), level=logging.INFO, filemode='w',
                        force=True)
    logging.getLogger(__name__).info(f"args: {vars(args)}")
    manager = MultiModalManager(d_img=args.embed_dim_img, d_txt=args.embed_dim_txt, critic=args.critic)
    losses = manager.train(synthetic_embedding_source(args, device), device, args)
    return manager, losses
