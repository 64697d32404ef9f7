"""The eval commands end to end on a tiny on-disk dataset: files -> readers -> HIP path -> metrics.
Checks that each command runs, prints the reference's summary lines, and agrees with the harness called directly."""
import json
import os
import re

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def world(cuda, tmp_path_factory):
    from PIL import Image
    from cvpce_amd import synthetic
    root = tmp_path_factory.mktemp('cli')
    to_img = lambda t: Image.fromarray((t.clamp(0, 1) * 255).round().to(torch.uint8).permute(1, 2, 0).numpy())
    # checkpoints in the reference's formats
    det = synthetic.synthetic_gln(seed=0, detections_per_img=60)
    enc = synthetic.synthetic_macvgg(seed=1)
    torch.save({'model_state_dict': det.state_dict()}, root / 'gln.tar')
    torch.save({'model_state_dict': enc.state_dict()}, root / 'dihe.tar')
    # GP training tree: 6 products
    gal = (synthetic.gallery_images(6, seed=5) + 1) / 2
    names = []
    for i in range(6):
        d = root / 'Training' / 'Food' / f'Cat{i % 2}'
        d.mkdir(parents=True, exist_ok=True)
        to_img(gal[i]).save(d / f'{10 + i}.png')
        names.append(f'Food/Cat{i % 2}/{10 + i}')
    (root / 'Training' / 'Background').mkdir()
    to_img(gal[0]).save(root / 'Training' / 'Background' / 'bg.png')
    # GP-180-like test images: products pasted on a shelf (png bytes under the reference's .jpg names)
    (root / 'Testing' / 'store1' / 'images').mkdir(parents=True)
    (root / 'ann').mkdir(); (root / 'plano').mkdir()
    for img_id, picks in ((15, (3, 1, 4)), (16, (0, 5, 2))):
        shelf = torch.full((3, 300, 840), 0.3)
        rows = []
        for j, i in enumerate(picks):
            x = 10 + j * 270
            shelf[:, 20:276, x:x + 256] = gal[i]
            rows.append(f'{names[i]}.jpg, {x}, 20, {x + 256}, 276')
        to_img(shelf).save(root / 'Testing' / 'store1' / 'images' / f'store1_{img_id}.jpg', format='PNG')
        (root / 'ann' / f's1_{img_id}.csv').write_text('\n'.join(rows) + '\n')
        graph = [{'ogg': j, 'n': -1, 's': -1, 'e': j + 1 if j < 2 else -1, 'w': j - 1, 'ne': -1, 'nw': -1, 'se': -1, 'sw': -1} for j in range(3)]
        objects = [{'width': 256, 'height': 256, 'img_path': names[i] + '.jpg'} for i in picks]
        (root / 'plano' / f's1_{img_id}.json').write_text(json.dumps({'graph': graph, 'objects': objects}))
    # SKU-110K-like set
    (root / 'sku').mkdir()
    rows = []
    for k in range(2):
        img = synthetic.shelf_image(30 + k, 384, 512)
        to_img(img).save(root / 'sku' / f'test_{k}.png')
        rows += [f'test_{k}.png,{20 + 60 * b},{30},{70 + 60 * b},{120},object,512,384' for b in range(4)]
    (root / 'sku' / 'annotations.csv').write_text('\n'.join(rows) + '\n')
    (root / 'out').mkdir()
    return root


def _run(args):
    from click.testing import CliRunner
    from cvpce_amd.cli import cli
    r = CliRunner().invoke(cli, [str(a) for a in args], catch_exceptions=False)
    assert r.exit_code == 0, r.output
    return r.output


def test_gln_eval_and_detect(world):
    from cvpce_amd import datautils, proposals_eval
    out = _run(['gln', 'eval', '--dataset', 'sku110k', '--imgs', world / 'sku', '--annotations', world / 'sku' / 'annotations.csv',
                '-t', '0.5', '-t', '0.75', '--no-plots', world / 'gln.tar'])
    ap = float(re.search(r'--> AP ([0-9.e-]+)', out).group(1))
    ar = float(re.search(r'--> AR300 ([0-9.e-]+)', out).group(1))
    data = datautils.SKU110KDataset(str(world / 'sku'), str(world / 'sku' / 'annotations.csv'))
    direct = proposals_eval.evaluate_gln(str(world / 'gln.tar'), data, thresholds=[0.5, 0.75], trim_module_prefix=False)
    assert abs(ap - sum(float(direct[t]['ap']) for t in (0.5, 0.75)) / 2) < 1e-6
    assert abs(ar - sum(float(direct[t]['ar_300']) for t in (0.5, 0.75)) / 2) < 1e-6
    assert '0.5:\t' in out and '0.75:\t' in out
    out = _run(['gln', 'detect', '--conf-thresh', '0.3', '--save', world / 'out' / 'det.png', world / 'gln.tar', world / 'sku' / 'test_0.png'])
    n = int(re.search(r'--> (\d+) detections', out).group(1))
    assert n == len([l for l in out.splitlines() if re.match(r'^[-0-9. ]+$', l)]) and (world / 'out' / 'det.png').exists()


def test_dihe_prebuild_index_and_eval(world):
    out = _run(['dihe', 'prebuild-index', '--img-dir', world / 'Training', '--out-dir', world / 'out', world / 'dihe.tar'])
    idx = torch.load(world / 'out' / 'classifier_index.pkl', weights_only=False)
    assert idx['embedding'].shape == (6, 1024) and sorted(idx['annotations']) == sorted(f'Food/Cat{i % 2}/{10 + i}' for i in range(6))
    out = _run(['dihe', 'eval', '--img-dir', world / 'Training', '--test-imgs', world / 'Testing', '--annotations', world / 'ann',
                '--enc-weights', world / 'dihe.tar', '--knn', '1', '--knn', '3'])
    acc = eval(re.search(r'--> accuracy (\{.*\})', out).group(1))
    assert acc == {1: 1.0, 3: 1.0}          # exact 256x256 pastes of the gallery products: every crop is its own product


def test_eval_product_detection_and_planograms(world):
    if not (world / 'out' / 'classifier_index.pkl').exists():
        _run(['dihe', 'prebuild-index', '--img-dir', world / 'Training', '--out-dir', world / 'out', world / 'dihe.tar'])
    common = ['--img-dir', world / 'Training', '--test-imgs', world / 'Testing']
    out = _run(['eval-product-detection', *common, '--annotations', world / 'ann', '--load-classifier-index',
                world / 'out' / 'classifier_index.pkl', world / 'gln.tar', world / 'dihe.tar'])
    assert re.search(r'--> mAP [0-9.e-]+', out) and re.search(r'--> mAR300 [0-9.e-]+', out)
    out = _run(['eval-planograms', *common, '--test-annotations', world / 'ann', '--planograms', world / 'plano', '--verbose',
                world / 'gln.tar', world / 'dihe.tar'])
    assert out.count('Detected accuracy:') == 2 and re.search(r'--> Mean accuracy [0-9.e-]+', out) and re.search(r'--> MSE: [0-9.e-]+', out)
