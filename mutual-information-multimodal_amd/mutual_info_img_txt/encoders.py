"""Encoder -> critic interface of the MI path and its on-disk formats (rank 3).

The encoders are NOT part of the accelerated hot path: they stay ordinary PyTorch-ROCm modules (north_star).  What this
module pins is the CONTRACT between them and the critic, with the same public names as the reference's ``model.py`` so
that checkpoints and call sites carry over:

* ``ResNet256_6_2_1`` (reference model.py:272-369): 1 x 256 x 256 input, conv 3x3 (1 -> 8) + six stride-2 stages of two
  basic blocks (8, 16, 32, 64, 128, 192 channels) + 2x2 average pooling -> ``z`` [B, 768] -> ``fc1``.  ``forward`` returns
  the reference's 5-tuple ``(softmax, z, sigmoid, z_local, logits)``; the MI critic consumes index 1 (model.py:543).
  Parameter names equal the reference's (``conv1``, ``bn1``, ``layer1..6.{0,1}.{conv1,bn1,conv2,bn2,downsample.{0,1}}``,
  ``fc1``), so reference checkpoints load.
* ``TextBert`` (reference model.py:54-88): BERT -> pooled [CLS] -> dropout -> ``pooled_output`` [B, hidden] (index 0, the
  critic's text embedding: AFTER dropout, model.py:76-80) and classifier logits (index 1).  Built on ``transformers``'
  ``BertModel`` (the reference's ``pytorch_transformers`` 1.0.0 is absent; parameter names under ``bert.`` are the same).
* ``ImageReportModel`` (reference model.py:529-595): ``forward(img, txt_ids, txt_masks, txt_segments)`` ->
  ``(embedding_img, embedding_txt, logits_img, logits_txt)``; ``save_image_model`` / ``save_text_model`` /
  ``save_pretrained(dir, epoch)`` write the reference's file names (``pytorch_MI_image_model.bin``,
  ``pytorch_MI_text_model.bin``, ``pytorch_model[_epoch{n}].bin`` + the BERT config json).
* loading (reference model.py:408-497): legacy ``gamma`` / ``beta`` key names, ``image_model.`` prefix stripping when
  loading the image encoder out of a joint checkpoint (its ``fc`` head is dropped), optional freezing of everything but
  ``layer6`` / ``fc``.

Extensions (no reference code): ``ProjectionHead`` maps the 768-d embeddings to the critic widths BASELINE.json names
(256 / 512 / 1024); ``ImageReportModel(..., proj_dim=...)`` applies one per modality; ``autocast_dtype=torch.bfloat16``
runs the encoders under autocast and hands fp32 embeddings to the critic (the fused critic path takes fp32 inputs).
"""
from __future__ import annotations

import json
import logging
import os
from typing import Dict, Iterable, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

IMAGE_MODEL_FILE = 'pytorch_MI_image_model.bin'      # reference model.py:559
TEXT_MODEL_FILE = 'pytorch_MI_text_model.bin'        # reference model.py:566
JOINT_MODEL_FILE = 'pytorch_model.bin'               # reference model.py:587
JOINT_MODEL_EPOCH_FILE = 'pytorch_model_epoch{}.bin'  # reference model.py:589-590
BERT_CONFIG_FILE = 'config.json'                     # what BertConfig.save_pretrained writes (model.py:584)


# ----------------------------------------------------------------------------------------------------------
# image encoder
# ----------------------------------------------------------------------------------------------------------
def conv3x3(in_planes, out_planes, stride=1, groups=1, dilation=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=dilation, groups=groups, bias=False,
                     dilation=dilation)


def conv1x1(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


class BasicBlock(nn.Module):
    """Two 3x3 conv + norm layers with an identity (or 1x1-projected) shortcut (reference model.py:121-152)."""

    def __init__(self, inplanes, planes, stride=1, downsample=None, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = norm_layer(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + shortcut)


class ResNet256_6_2_1(nn.Module):
    STAGE_PLANES = (8, 16, 32, 64, 128, 192)

    def __init__(self, block=BasicBlock, blocks_per_layers=(2, 2, 2, 2, 2, 2), output_channels=4,
                 norm_layer=nn.BatchNorm2d, zero_init_residual=False):
        super().__init__()
        self._norm_layer = norm_layer
        self.inplanes = 8
        self.conv1 = nn.Conv2d(1, self.inplanes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn1 = norm_layer(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        for n, (planes, count) in enumerate(zip(self.STAGE_PLANES, blocks_per_layers), start=1):
            setattr(self, f'layer{n}', self._make_layer(block, planes, count, stride=2))
        self.avgpool = nn.AvgPool2d((2, 2))
        self.fc1 = nn.Linear(768, output_channels)
        self.softmax = nn.Softmax(dim=1)
        self.sigmoid = nn.Sigmoid()
        for m in self.modules():  # reference model.py:313-320
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)

    def _make_layer(self, block, planes, num_of_blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes, stride), self._norm_layer(planes))
        layers = [block(self.inplanes, planes, stride=stride, downsample=downsample, norm_layer=self._norm_layer)]
        self.inplanes = planes
        layers += [block(planes, planes, norm_layer=self._norm_layer) for _ in range(1, num_of_blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.relu(self.bn1(self.conv1(x)))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        z_local = self.layer5(x)
        z = torch.flatten(self.avgpool(self.layer6(z_local)), 1)  # [B, 192 * 2 * 2] = [B, 768]
        y_logits = self.fc1(z)
        return self.softmax(y_logits), z, self.sigmoid(y_logits), z_local, y_logits

    def save_pretrained(self, save_directory, epoch=-1):
        os.makedirs(save_directory, exist_ok=True)
        name = JOINT_MODEL_FILE if epoch == -1 else JOINT_MODEL_EPOCH_FILE.format(epoch)
        path = os.path.join(save_directory, name)
        torch.save(getattr(self, 'module', self).state_dict(), path)
        return path

    @classmethod
    def from_pretrained(cls, pretrained_model_path, block=BasicBlock, blocks_per_layers=(2, 2, 2, 2, 2, 2),
                        output_channels=4, loading_from_joint=False, freeze_encoder=False, state_dict=None,
                        output_loading_info=False, **kwargs):
        model = cls(block, blocks_per_layers, output_channels=output_channels, **kwargs)
        if state_dict is None:
            state_dict = torch.load(pretrained_model_path, map_location='cpu', weights_only=True)
        state_dict = image_state_from_checkpoint(state_dict, loading_from_joint)
        info = model.load_state_dict(state_dict, strict=False)
        log = logging.getLogger(__name__)
        if info.missing_keys:
            log.info("Weights of %s not initialized from pretrained model: %s", cls.__name__, info.missing_keys)
        if info.unexpected_keys:
            log.info("Weights from pretrained model not used in %s: %s", cls.__name__, info.unexpected_keys)
        if freeze_encoder:  # reference model.py:490-495: only layer6 and the fc head stay trainable
            for n, p in model.named_parameters():
                if 'layer6' not in n and 'fc' not in n:
                    p.requires_grad = False
        if output_loading_info:
            return model, {"missing_keys": list(info.missing_keys), "unexpected_keys": list(info.unexpected_keys),
                           "error_msgs": []}
        return model


def convert_legacy_keys(state_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """``gamma`` -> ``weight`` and ``beta`` -> ``bias`` in parameter names (old TF-style checkpoints; reference
    model.py:431-444 and the same convention in pytorch-transformers' loader).  As in the reference, when a key holds
    both words only the ``beta`` rename is applied (its second ``if`` overwrites the first)."""
    out = {}
    for key, value in state_dict.items():
        new_key = key
        if 'gamma' in key:
            new_key = key.replace('gamma', 'weight')
        if 'beta' in key:
            new_key = key.replace('beta', 'bias')
        out[new_key] = value
    return out


def image_state_from_checkpoint(state_dict: Dict[str, torch.Tensor], loading_from_joint: bool = False):
    """State dict for ``ResNet256_6_2_1`` out of a reference checkpoint: legacy key names converted; with
    ``loading_from_joint`` the ``image_model.`` prefix of a joint ``ImageReportModel`` checkpoint is stripped and the
    joint model's ``image_model.fc*`` head is dropped (reference model.py:446-460)."""
    state_dict = convert_legacy_keys(state_dict)
    if not loading_from_joint:
        return state_dict
    out = {}
    for key, value in state_dict.items():
        if key.startswith('image_model.'):
            if 'image_model.fc' in key:
                continue
            out[key[len('image_model.'):]] = value
        else:
            out[key] = value  # the reference leaves the other keys in place (reported as unexpected on load)
    return out


def build_resnet256_6_2_1(block=BasicBlock, blocks_per_layers=(2, 2, 2, 2, 2, 2), pretrained=False,
                          pretrained_model_path=None, output_channels=4, loading_from_joint=False, freeze_encoder=False,
                          **kwargs):
    if pretrained:
        return ResNet256_6_2_1.from_pretrained(pretrained_model_path, block, blocks_per_layers, output_channels,
                                               loading_from_joint=loading_from_joint, freeze_encoder=freeze_encoder,
                                               **kwargs)
    return ResNet256_6_2_1(block, blocks_per_layers, output_channels=output_channels, **kwargs)


def build_resnet_model(model_name, checkpoint_path=None, output_channels=4, loading_from_joint=False,
                       freeze_encoder=False):
    """Reference model.py:513-526.  Only 'resnet256_6_2_1' exists there (any other name yields an unbound local)."""
    if model_name != 'resnet256_6_2_1':
        raise ValueError(f"unknown image_model_name {model_name!r}: the reference only builds 'resnet256_6_2_1'")
    return build_resnet256_6_2_1(output_channels=output_channels, pretrained=checkpoint_path is not None,
                                 pretrained_model_path=checkpoint_path, loading_from_joint=loading_from_joint,
                                 freeze_encoder=freeze_encoder)


# ----------------------------------------------------------------------------------------------------------
# text encoder
# ----------------------------------------------------------------------------------------------------------
def _transformers():
    try:
        import transformers  # noqa: F401
        from transformers import BertConfig, BertModel
        return BertConfig, BertModel
    except Exception as e:  # pragma: no cover - depends on the image
        raise ImportError("the text encoder needs the `transformers` package (the reference's pytorch_transformers 1.0.0 is "
                          "not available offline)") from e


class TextBert(nn.Module):
    """BERT -> pooled [CLS] -> dropout -> (pooled_output, logits, ...) as reference model.py:54-81."""

    def __init__(self, config):
        super().__init__()
        _, BertModel = _transformers()
        self.config = config
        self.num_classes = config.num_classes
        self.bert = BertModel(config)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.classifier = nn.Linear(config.hidden_size, config.num_classes)

    def forward(self, input_ids, token_type_ids=None, attention_mask=None, labels=None):
        outputs = self.bert(input_ids, token_type_ids=token_type_ids, attention_mask=attention_mask, return_dict=False)
        pooled_output = self.dropout(outputs[1])  # the critic's text embedding is taken AFTER dropout
        logits = self.classifier(pooled_output)
        return (pooled_output, logits) + tuple(outputs[2:])

    def freeze_bert_encoder(self):
        for p in self.bert.parameters():
            p.requires_grad = False

    def unfreeze_bert_encode(self):  # (sic) the reference's spelling, model.py:86
        for p in self.bert.parameters():
            p.requires_grad = True

    @classmethod
    def from_pretrained(cls, pretrained_dir, config):
        """Weights from ``<pretrained_dir>/pytorch_model.bin`` (reference model.py:104 through
        ``BertPreTrainedModel.from_pretrained``, which raises when the file is absent and redirects a bare ``BertModel``
        checkpoint into ``model.bert`` by its base-model prefix).  A missing file or a checkpoint that fills no ``bert.*``
        parameter is an error: random initialisation is an explicit choice (``TextBert(config)``), never a fall-back."""
        model = cls(config)
        path = os.path.join(pretrained_dir, JOINT_MODEL_FILE)
        if not os.path.isfile(path):
            raise FileNotFoundError(f"TextBert.from_pretrained: no weights file {path} (construct TextBert(config) "
                                    "directly for a randomly initialised text encoder)")
        state = convert_legacy_keys(torch.load(path, map_location='cpu', weights_only=True))
        own = model.state_dict()
        if not any(k.startswith('bert.') for k in state):
            # a bare BertModel checkpoint (keys "embeddings...", "encoder...", "pooler..."): load it into model.bert
            bare = {k: v for k, v in state.items() if 'bert.' + k in own}
            if bare:
                state = {**{'bert.' + k: v for k, v in bare.items()}, **{k: v for k, v in state.items() if k not in bare}}
        loaded = [k for k in state if k.startswith('bert.') and k in own]
        if not loaded:
            raise ValueError(f"TextBert.from_pretrained: {path} holds no parameter of the BERT encoder "
                             f"(first keys: {list(state)[:5]})")
        info = model.load_state_dict(state, strict=False)
        missing_bert = [k for k in info.missing_keys if k.startswith('bert.') and not k.endswith('position_ids')]
        log = logging.getLogger(__name__)
        if missing_bert:
            log.warning("TextBert.from_pretrained: %d BERT parameters keep their random initialisation, e.g. %s",
                        len(missing_bert), missing_bert[:5])
        log.info("TextBert.from_pretrained: loaded %d bert.* tensors; missing %s unexpected %s", len(loaded),
                 info.missing_keys, info.unexpected_keys)
        return model


def build_bert_model(bert_pretrained_dir, bert_config_name, output_channels):
    """Reference model.py:91-105: BERT config json in ``bert_pretrained_dir`` -> (TextBert, config)."""
    BertConfig, _ = _transformers()
    config_path = os.path.join(bert_pretrained_dir, bert_config_name)
    with open(config_path) as f:
        print('BERT config:', json.load(f))
    config = BertConfig.from_json_file(config_path)
    config.num_classes = output_channels
    return TextBert.from_pretrained(bert_pretrained_dir, config=config), config


# ----------------------------------------------------------------------------------------------------------
# joint model, projection heads
# ----------------------------------------------------------------------------------------------------------
class ProjectionHead(nn.Module):
    """Extension: linear map of an encoder embedding to the critic width (BASELINE configs 2-5 use 256 / 512 / 1024)."""

    def __init__(self, d_in: int, d_out: int, normalize: bool = False):
        super().__init__()
        self.linear = nn.Linear(d_in, d_out)
        self.normalize = normalize

    def forward(self, x):
        x = self.linear(x)
        return F.normalize(x, dim=1) * (x.shape[1] ** 0.5) if self.normalize else x


class ImageReportModel(nn.Module):
    def __init__(self, text_model, bert_config, image_model, proj_dim: Optional[int] = None, autocast_dtype=None):
        super().__init__()
        self.text_model = text_model
        self.bert_config = bert_config
        self.image_model = image_model
        self.autocast_dtype = autocast_dtype
        if proj_dim:
            d_txt = getattr(bert_config, 'hidden_size', 768)
            self.proj_img = ProjectionHead(768, proj_dim)
            self.proj_txt = ProjectionHead(d_txt, proj_dim)
        else:
            self.proj_img = self.proj_txt = None

    def forward(self, img, txt_ids, txt_masks=None, txt_segments=None):
        use_amp = self.autocast_dtype is not None and img.is_cuda
        with torch.autocast(device_type='cuda', dtype=self.autocast_dtype, enabled=use_amp):
            outputs_img = self.image_model.forward(img)
            outputs_txt = self.text_model.forward(input_ids=txt_ids, attention_mask=txt_masks,
                                                  token_type_ids=txt_segments)
        embedding_img = outputs_img[1]   # z [B, 768]                         (reference model.py:543)
        embedding_txt = outputs_txt[0]   # pooled [CLS] after dropout [B, hid] (reference model.py:552)
        if self.proj_img is not None:
            embedding_img = self.proj_img(embedding_img.float())
            embedding_txt = self.proj_txt(embedding_txt.float())
        # the fused critic takes contiguous fp32 embeddings
        return (embedding_img.float().contiguous(), embedding_txt.float().contiguous(), outputs_img[-1], outputs_txt[1])

    def save_image_model(self, save_directory):
        path = os.path.join(save_directory, IMAGE_MODEL_FILE)
        torch.save(self.image_model.state_dict(), path)
        return path

    def save_text_model(self, save_directory):
        path = os.path.join(save_directory, TEXT_MODEL_FILE)
        torch.save(self.text_model.state_dict(), path)
        return path

    def save_pretrained(self, save_directory, epoch=-1):
        os.makedirs(save_directory, exist_ok=True)
        model_to_save = getattr(self, 'module', self)
        if hasattr(model_to_save.bert_config, 'save_pretrained'):
            model_to_save.bert_config.save_pretrained(save_directory)
        name = JOINT_MODEL_FILE if epoch == -1 else JOINT_MODEL_EPOCH_FILE.format(epoch)
        path = os.path.join(save_directory, name)
        torch.save(model_to_save.state_dict(), path)
        return path
