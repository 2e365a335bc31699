"""Entry point kept from the reference (train.py:21-36 ``train_MI_models``).  ``python train.py --synthetic`` trains
the MI critic on synthetic embeddings on the ROCm device; the real-data path of the reference (MIMIC-CXR JPEGs, a
private BERT checkpoint) is not available offline."""
import argparse
import os

import torch

from multi_modal import train_mutual_information


def construct_training_parameters(argv=None):
    """The flags of the reference parser (helpers.py:84-144) that the hot path reads, plus the synthetic-mode ones."""
    p = argparse.ArgumentParser()
    p.add_argument('--batch_size', default=64, type=int)                       # helpers.py:110
    p.add_argument('--mi_estimator', default='dv', type=str)                   # helpers.py:122-124
    p.add_argument('--init_lr', default=1e-4, type=float)                      # helpers.py:125
    p.add_argument('--num_train_epochs', default=1, type=int)
    p.add_argument('--save_directory', default='save_dir/mm_synthetic', type=str)
    p.add_argument('--synthetic', action='store_true')
    p.add_argument('--critic', default='concat_mlp', choices=['concat_mlp', 'bilinear', 'separable'])
    p.add_argument('--embed_dim_img', default=128, type=int)
    p.add_argument('--embed_dim_txt', default=128, type=int)
    p.add_argument('--steps_per_epoch', default=20, type=int)
    p.add_argument('--precision', default='f32', choices=['bf16', 'f32', 'bf16x3'])  # the reference is fp32; bf16 is the fast mode
    p.add_argument('--synthetic_encoders', action='store_true')
    p.add_argument('--img_size', default=256, type=int)                        # helpers.py:130
    p.add_argument('--output_channels', default=1, type=int)                   # helpers.py:131
    p.add_argument('--embed_proj_dim', default=None, type=int)
    p.add_argument('--no_graph', dest='graph', action='store_false')
    p.add_argument('--seed', default=0, type=int)
    return p.parse_args(argv)


def train_MI_models(argv=None):
    args = construct_training_parameters(argv)
    if args.mi_estimator not in ('dv', 'infonce'):
        raise ValueError(f"unknown --mi_estimator {args.mi_estimator!r}")
    if not torch.cuda.is_available():
        raise RuntimeError("the MI critic path needs an MI355X (ROCm) device; there is no CPU fallback")
    device = torch.device('cuda')
    args.save_directory = os.path.join(args.save_directory, f'mm_{args.mi_estimator}_epoch{args.num_train_epochs}')
    train_mutual_information(args, device)
    losses = train_mutual_information.last_manager.training_loss
    for n, l in enumerate(losses):
        print(f'Epoch {n+1} finished! Epoch loss: {l:.5f}')
    return losses


if __name__ == '__main__':
    train_MI_models()
