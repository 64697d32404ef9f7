"""Independent pin of the oracle's ResNet-50 body (`oracle/gln.py` resnet_body / bottleneck / frozen_bn).

torchvision -- whose `resnet50` the reference imports (`/root/reference/cvpce/models/proposals.py:176-181` via
`torchvision.models.detection.retinanet`) -- is not installed in this image and the reference holds no fixture for it, so the oracle's
restatement of that body is checked here against a THIRD implementation of the same published architecture: Hugging Face
`transformers.ResNetModel` (bottleneck layers, depths 3-4-6-3, stride on the 3x3 conv = "v1.5", 7x7/2 stem + 3x3/2 max-pool),
at reduced width so that the fixture stays small.  BatchNorm in eval mode with random statistics = FrozenBatchNorm2d with the same
buffers (both eps 1e-5).

    python tests/golden/make_thirdparty.py        # writes tests/golden/resnet_body_hf.pt  (needs `transformers`; ~30 s)

The fixture holds the weights under torchvision's key names, the input and the four stage outputs C2..C5; the test that reads it
(`tests/test_oracle_golden.py::test_resnet_body_matches_third_party_implementation`) imports neither transformers nor this script."""
import os
import re

import torch


def main():
    from transformers import ResNetConfig, ResNetModel
    torch.manual_seed(1234)
    cfg = ResNetConfig(num_channels=3, embedding_size=16, hidden_sizes=[32, 64, 128, 256], depths=[3, 4, 6, 3], layer_type='bottleneck',
                       hidden_act='relu', downsample_in_first_stage=False, downsample_in_bottleneck=False)
    m = ResNetModel(cfg).eval()
    with torch.no_grad():
        for name, mod in m.named_modules():
            if isinstance(mod, torch.nn.BatchNorm2d):       # non-trivial affine + statistics
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.normal_(0, 0.2)
                mod.running_mean.normal_(0, 0.3)
                mod.running_var.uniform_(0.4, 1.6)
        x = torch.randn(2, 3, 96, 128)
        out = m(x, output_hidden_states=True)
    hs = out.hidden_states                                   # (stem output, stage 1 .. stage 4)
    assert len(hs) == 5
    sd = {}
    for k, v in m.state_dict().items():
        if k.endswith('num_batches_tracked'):
            continue
        k = k.replace('embedder.embedder.convolution', 'conv1').replace('embedder.embedder.normalization', 'bn1')
        mm = re.match(r'encoder\.stages\.(\d)\.layers\.(\d+)\.(shortcut|layer\.(\d))\.(convolution|normalization)\.(.*)', k)
        if mm:
            stage, layer, which, kk, kind, leaf = mm.groups()
            base = f'layer{int(stage) + 1}.{layer}.'
            if which == 'shortcut':
                k = base + ('downsample.0.' if kind == 'convolution' else 'downsample.1.') + leaf
            else:
                k = base + (f'conv{int(kk) + 1}.' if kind == 'convolution' else f'bn{int(kk) + 1}.') + leaf
        sd['backbone.body.' + k] = v.clone()
    fx = {'state_dict': sd, 'x': x, 'stages': [h.clone() for h in hs[1:]],
          'made_by': f'transformers.ResNetModel (transformers {__import__("transformers").__version__}), widths 32-64-128-256, seed 1234'}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'resnet_body_hf.pt')
    torch.save(fx, path)
    print('wrote', path, os.path.getsize(path), 'bytes;', len(sd), 'tensors; stage shapes', [tuple(h.shape) for h in hs[1:]])


if __name__ == '__main__':
    main()
