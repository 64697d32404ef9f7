"""Dataset readers + planogram reader + CLI surface against what the REFERENCE's own index builders
returned on the same files (tests/golden/datasets.pt, made by tests/golden/make_golden.py section 5)."""
import os
import re

import pytest
import torch


@pytest.fixture(scope='module')
def ds(golden_dir, tmp_path_factory):
    g = torch.load(os.path.join(golden_dir, 'datasets.pt'), weights_only=False)
    root = tmp_path_factory.mktemp('datasets')
    for rel, text in g['files'].items():
        full = root / rel
        full.parent.mkdir(parents=True, exist_ok=True)
        full.write_text(text)
    return g, str(root)


def _same_entries(mine, ref, keys):
    assert len(mine) == len(ref)
    for a, b in zip(mine, ref):
        for k in keys:
            if torch.is_tensor(b[k]):
                assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), k
            else:
                assert a[k] == b[k], (k, a[k], b[k])


def test_sku110k_index(ds):
    from cvpce_amd import datautils
    g, root = ds
    idx = datautils.SKU110KDataset.build_index(os.path.join(root, 'sku/annotations.csv'), ['train_882.jpg'])
    _same_entries(idx, g['sku110k'], ('image_name', 'image_width', 'image_height', 'boxes', 'labels'))
    assert [e['image_name'] for e in idx] == ['test_0.jpg', 'test_1.jpg', 'test_2.jpg']      # order of first appearance
    with pytest.raises(NotImplementedError):
        datautils.SKU110KDataset(root, os.path.join(root, 'sku/annotations.csv'), include_gaussians=True)


def test_gpbaseline_index(ds):
    from cvpce_amd import datautils
    g, root = ds
    idx = datautils.GPBaselineDataset.build_index(os.path.join(root, 'imgs'), os.path.join(root, 'base/gt.csv'))
    mine = [{**e, 'image_path': os.path.relpath(e['image_path'], root)} for e in idx]
    _same_entries(mine, g['gpbaseline'], ('image_path', 'boxes', 'labels'))


@pytest.mark.parametrize('key,only,skip', [('all', None, None), ('only', ['s2_3.csv', 's3_111.csv'], None), ('skip', None, ['s1_15.csv'])])
def test_gp180_index(ds, key, only, skip):
    from cvpce_amd import datautils
    g, root = ds
    ts = datautils.GroceryProductsTestSet('TESTIMGS', os.path.join(root, 'ann'), only=only, skip=skip)
    _same_entries(ts.index, g['gp180'][key], ('id', 'path', 'anns', 'boxes'))
    assert ts.int_to_ann == sorted(set(a for e in ts.index for a in e['anns']))
    assert all(ts.int_to_ann[i] == a for a, i in ts.ann_to_int.items())


def test_gp180_validation_split_semantics(ds):
    """ints instead of lists: keep / drop the first k annotations of every image (cli/dihe.py:291-296)."""
    from cvpce_amd import datautils
    g, root = ds
    val = datautils.GroceryProductsTestSet('T', os.path.join(root, 'ann'), only=2)
    test = datautils.GroceryProductsTestSet('T', os.path.join(root, 'ann'), skip=2)
    assert (val.toskip, val.tokeep, test.toskip, test.tokeep) == (0, 2, 2, 9999)
    e = val.index[0]
    assert e['anns'][val.toskip:val.tokeep] == e['anns'][:2] and e['anns'][test.toskip:test.tokeep] == e['anns'][2:]


def test_gp_training_tree(ds):
    from cvpce_amd import datautils
    g, root = ds
    skip = re.compile('|'.join(f'({s})' for s in (r'^Background.*$', r'^.*/[Oo]riginals?$')))
    for key, only in (('all_clean', None), ('only_food_clean', ['Food'])):
        p, c, a = datautils.GroceryProductsDataset.build_index([os.path.join(root, 'gp/Training')], skip, only)
        mine = sorted(zip([os.path.relpath(x, root) for x in p], c, a))
        assert mine == [tuple(t) for t in g['gp_walk'][key]]
    p, c, a = datautils.GroceryProductsDataset.build_index_from_file([os.path.join(root, 'gp')], skip, None)
    assert list(zip([os.path.relpath(x, root) for x in p], c, a)) == [tuple(t) for t in g['gp_walk']['from_file']]
    # a file name without an extension makes the reference's walker crash (golden None); here it is skipped with a note
    assert g['gp_walk']['all'] is None
    open(os.path.join(root, 'gp/Training/Drinks/Juice/noextension'), 'w').write('x')
    try:
        p, c, a = datautils.GroceryProductsDataset.build_index([os.path.join(root, 'gp/Training')], skip, None)
        assert sorted(zip([os.path.relpath(x, root) for x in p], c, a)) == [tuple(t) for t in g['gp_walk']['all_clean']]
    finally:
        os.remove(os.path.join(root, 'gp/Training/Drinks/Juice/noextension'))


def test_gp_gallery_tensorize(tmp_path):
    """tensorize (datautils.py:397-415): longer side -> 256 (PIL bilinear), [-1,1], zero padding right / bottom."""
    from PIL import Image
    from cvpce_amd import datautils
    (tmp_path / 'Training' / 'Food' / 'Tea').mkdir(parents=True)
    rng = torch.Generator().manual_seed(0)
    tall = (torch.rand(300, 120, 3, generator=rng) * 255).to(torch.uint8).numpy()
    wide = (torch.rand(90, 400, 3, generator=rng) * 255).to(torch.uint8).numpy()
    Image.fromarray(tall).save(tmp_path / 'Training/Food/Tea/1.png')
    Image.fromarray(wide).save(tmp_path / 'Training/Food/Tea/2.png')
    data = datautils.GroceryProductsDataset([str(tmp_path / 'Training')], include_annotations=True)
    assert data.annotations == ['Food/Tea/1', 'Food/Tea/2'] and len(data) == 2
    t, t2, cats, ann = data[0]
    assert t.shape == (3, 256, 256) and cats == ['Food', 'Tea'] and ann == 'Food/Tea/1' and t2 is t
    w = round(256 * 120 / 300)
    assert float(t.min()) >= -1 and float(t.max()) <= 1 and torch.all(t[:, :, w:] == 0) and not torch.all(t[:, :, :w] == 0)
    ref = datautils.pil_to_tensor(Image.fromarray(tall).resize((w, 256), Image.BILINEAR)) * 2 - 1
    assert torch.equal(t[:, :, :w], ref)
    t, _, _, _ = data[1]
    h = round(256 * 90 / 400)
    assert torch.all(t[:, h:, :] == 0) and t.shape == (3, 256, 256)
    assert data.index_for_ann('Food/Tea/2') == 1 and data.index_for_ann('nope') is None


@pytest.mark.parametrize('name', ['s1_15.json', 's2_3.json', 's3_111.json'])
def test_tonioni_planogram(ds, name):
    from cvpce_amd import planogram_adapters
    g, root = ds
    boxes, labels, graph = planogram_adapters.read_tonioni_planogram(os.path.join(root, 'plano', name))
    want = g['tonioni'][name]
    assert boxes.dtype == torch.float32 and torch.equal(boxes, want['boxes'])
    assert labels == want['labels']
    assert sorted((int(n), d['label']) for n, d in graph.nodes(data=True)) == [tuple(t) for t in want['nodes']]
    assert sorted((int(a), int(b), d['dir']) for a, b, d in graph.edges(data=True)) == [tuple(t) for t in want['edges']]
    assert all(set(d) == {'label'} for _, d in graph.nodes(data=True))


def test_planogram_testset_and_internal(ds):
    from cvpce_amd import datautils
    g, root = ds
    ps = datautils.PlanogramTestSet('TESTIMGS', os.path.join(root, 'ann'), os.path.join(root, 'plano'))
    assert [e['id'] for e in ps.index] == [('1', '15'), ('2', '3'), ('3', '111')]
    for e in ps.index:
        want = g['tonioni'][f's{e["id"][0]}_{e["id"][1]}.json']
        assert torch.equal(e['plano']['boxes'], want['boxes']) and e['plano']['labels'] == want['labels']
        assert e['plano']['actual_accuracy'] == 1.0
    internal = datautils.InternalPlanoSet.build_index(os.path.join(root, 'internal'))
    mine = [{**e, 'img': os.path.relpath(e['img'], root)} for e in internal]
    _same_entries(mine, g['internal'], ('img', 'anns', 'boxes', 'actual_accuracy'))


def test_image_loading_roundtrip(tmp_path):
    from PIL import Image
    from cvpce_amd import datautils
    a = (torch.rand(40, 60, 3, generator=torch.Generator().manual_seed(3)) * 255).to(torch.uint8)
    Image.fromarray(a.numpy()).save(tmp_path / 'test_0.png')
    (tmp_path / 'ann.csv').write_text('test_0.png,1,2,30,35,object,60,40\n')
    data = datautils.SKU110KDataset(str(tmp_path), str(tmp_path / 'ann.csv'))
    img, entry = data[0]
    assert img.shape == (3, 40, 60) and img.dtype == torch.float32
    assert torch.equal(img, a.permute(2, 0, 1).float() / 255) and entry['boxes'].tolist() == [[1, 2, 30, 35]]
    assert data.index_for_name('test_0.png') == 0 and data.index_for_name('x') is None


# ---- CLI surface: same command names, arguments and option names / defaults as the reference (cli/gln.py:230-252,
# 275-281; cli/dihe.py:257-286,382-401; cli/eval.py:12-41,168-206) ----
CLI_SURFACE = {
    ('gln', 'eval'): (['state_file'], {'dataset': 'sku110k', 'batch_size': 1, 'dataloader_workers': 4, 'metric_workers': 8,
                                       'iou_threshold': (0.5,), 'coco': False, 'trim_module_prefix': False, 'plots': True,
                                       'plot_res_reduction': 200, 'imgs': None, 'annotations': None}),
    ('gln', 'detect'): (['state_file', 'image_file'], {'conf_thresh': 0.5, 'save': None}),
    ('dihe', 'eval'): ([], {'model': 'vgg16', 'resnet_layers': (2, 3), 'batch_norm': False, 'batch_size': 8, 'dataloader_workers': 8,
                            'enc_weights': None, 'only': 'none', 'knn': (1,), 'img_dir': None, 'test_imgs': None, 'annotations': None}),
    ('dihe', 'prebuild-index'): (['dihe_state'], {'datatype': 'gp', 'img_dir': None, 'out_dir': None}),
    ('eval-product-detection',): (['gln_state', 'dihe_state'], {'iou_threshold': (0.5,), 'coco': False, 'load_classifier_index': None,
                                                                 'img_dir': None, 'test_imgs': None, 'annotations': None}),
    ('eval-planograms',): (['gln_state', 'dihe_state'], {'datatype': 'gp', 'load_classifier_index': None, 'verbose': False,
                                                         'img_dir': None, 'test_imgs': None, 'test_annotations': None, 'planograms': None}),
}


@pytest.mark.parametrize('path', list(CLI_SURFACE))
def test_cli_surface(path):
    import click
    from cvpce_amd.cli import cli
    cmd = cli
    for name in path:
        cmd = cmd.commands[name]
    args, opts = CLI_SURFACE[path]
    assert [p.name for p in cmd.params if isinstance(p, click.Argument)] == args
    got = {p.name: p.default for p in cmd.params if isinstance(p, click.Option)}
    assert set(got) == set(opts), set(got) ^ set(opts)
    for k, v in opts.items():
        if v is not None:
            assert (tuple(got[k]) if isinstance(v, tuple) else got[k]) == v, (k, got[k], v)


def test_cli_help_runs():
    from click.testing import CliRunner
    from cvpce_amd.cli import cli
    r = CliRunner().invoke(cli, ['--help'])
    assert r.exit_code == 0 and all(c in r.output for c in ('gln', 'dihe', 'eval-product-detection', 'eval-planograms'))
    for sub in (['gln', 'eval'], ['gln', 'detect'], ['dihe', 'eval'], ['dihe', 'prebuild-index'], ['eval-product-detection'], ['eval-planograms']):
        assert CliRunner().invoke(cli, sub + ['--help']).exit_code == 0
