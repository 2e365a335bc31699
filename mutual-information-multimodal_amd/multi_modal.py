"""Entry point kept from the reference (multi_modal.py:15-67): ``train_mutual_information(args, device)`` builds the
``MultiModalManager``, trains it and returns ``model_manager.image_model``.

Three modes, by what ``args`` holds:

* ``args.synthetic`` (``train.py --synthetic``): the MI critic on synthetic embeddings of the configured widths -- no
  encoders, returns None for the image model (BASELINE configs[0]: B=64, d=128, DV);
* ``args.synthetic_encoders``: the reference's encoders (ResNet256_6_2_1 + a small randomly initialised BERT) on
  synthetic images / tokens -- the whole reference step (encoders -> fused critic -> three optimisers -> checkpoints)
  without MIMIC-CXR or the private BERT checkpoint;
* otherwise the reference's own arguments (``bert_pretrained_dir``, ``bert_config_name``, ``output_channels``,
  ``image_model_name``, ``image_dir``, ``dataset_metadata``, ``text_token_features`` or a ``data_loader``): real data.
  Tokenisation of raw reports (reference multi_modal.py:44-45, model_utils.py:341-544) is the caller's job.
"""
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


def synthetic_encoder_batches(args, vocab_size=64, seq_len=16):
    """A list of reference-style batches (img, txt_ids, txt_masks, txt_segments, study_id, img_id) with a shared latent
    between an image's blob position and its report's first tokens."""
    gen = torch.Generator(device="cpu").manual_seed(int(getattr(args, "seed", 0)))
    b, size = args.batch_size, int(getattr(args, "img_size", 256))
    batches = []
    for step in range(int(args.steps_per_epoch)):
        key = torch.randint(0, vocab_size - 4, (b,), generator=gen)
        img = torch.rand(b, 1, size, size, generator=gen) * 0.1
        for n in range(b):
            r = int(key[n]) % 8 * (size // 8)
            img[n, 0, r:r + size // 8, :] += 0.8
        ids = torch.randint(4, vocab_size, (b, seq_len), generator=gen)
        ids[:, 0] = 1
        ids[:, 1] = key + 4
        study = [str(50000000 + step * b + n) for n in range(b)]
        batches.append((img, ids, torch.ones(b, seq_len, dtype=torch.long), torch.zeros(b, seq_len, dtype=torch.long),
                        study, [f"img{step}_{n}" for n in range(b)]))
    return batches


def _small_bert_config(output_channels, vocab_size=64, hidden=64):
    from transformers import BertConfig
    cfg = BertConfig(vocab_size=vocab_size, hidden_size=hidden, num_hidden_layers=2, num_attention_heads=4,
                     intermediate_size=4 * hidden, max_position_embeddings=64)
    cfg.num_classes = output_channels
    return cfg


def train_mutual_information(args, device):
    os.makedirs(args.save_directory, exist_ok=True)
    # same file name / mode as the reference (multi_modal.py:27-30); force=True so that an already configured root
    # logger (a test runner, a notebook) does not silently swallow the training log
    logging.basicConfig(filename=os.path.join(args.save_directory, 'training_MI.log'), level=logging.INFO, filemode='w',
                        format='%(asctime)s - %(name)s %(message)s', datefmt='%m-%d %H:%M', force=True)
    logging.getLogger(__name__).info(f"args: {args}")
    critic = getattr(args, "critic", "concat_mlp")
    if getattr(args, "synthetic", False):
        model_manager = MultiModalManager(d_img=args.embed_dim_img, d_txt=args.embed_dim_txt, critic=critic)
        source = synthetic_embedding_source(args, device)
    elif getattr(args, "synthetic_encoders", False):
        from mutual_info_img_txt.model import ResNet256_6_2_1, TextBert
        oc = int(getattr(args, "output_channels", 1) or 1)
        cfg = _small_bert_config(oc)
        model_manager = MultiModalManager(output_channels=oc, image_model=ResNet256_6_2_1(output_channels=oc),
                                          text_model=TextBert(cfg), bert_config=cfg, critic=critic,
                                          embed_proj_dim=getattr(args, "embed_proj_dim", None))
        source = synthetic_encoder_batches(args, vocab_size=cfg.vocab_size)
    else:
        model_manager = MultiModalManager(bert_pretrained_dir=args.bert_pretrained_dir,
                                          bert_config_name=args.bert_config_name,
                                          output_channels=args.output_channels,
                                          image_model_name=args.image_model_name, critic=critic,
                                          embed_proj_dim=getattr(args, "embed_proj_dim", None))
        source = getattr(args, "data_loader", None) or args.text_token_features
    print("Start training for ImageTextModelManager")
    model_manager.train(source, device=device, args=args)
    print("Finish training for ImageTextModelManager")
    train_mutual_information.last_manager = model_manager  # the reference only returns the image model (multi_modal.py:67)
    return model_manager.image_model
