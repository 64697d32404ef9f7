"""`cvpce` command line, evaluation surface only (reference: cvpce/cli/__init__.py, cli/gln.py:230-307,
cli/dihe.py:257-309,382-423, cli/eval.py:12-71,168-240): the same command names, arguments, options and defaults as the
reference for

    cvpce gln eval | gln detect | dihe eval | dihe prebuild-index | eval-product-detection | eval-planograms

Training, hyper-parameter search, dataset visualisation and plotting commands are out of scope.  Options that only steer
the reference's DataLoader / matplotlib (`--dataloader-workers`, `--plots`,
`--plot-res-reduction`) are accepted and ignored: images are read in-process and every metric is printed.

    python -m cvpce_amd.cli --help
"""
import os

import click
import torch

from .. import classification_eval, datautils, detection_eval, production, proposals_eval
from ..defaults import (GP_ANN_DIR, GP_PLANO_DIR, GP_TEST_DIR, GP_TEST_VALIDATION_SET_SIZE, GP_TRAIN_FOLDERS, OUT_DIR,
                        SKU110K_ANNOTATION_FILE, SKU110K_IMG_DIR, SKU110K_SKIP)
from ..models import classification, proposals

MODEL_STATE_DICT_KEY = 'model_state_dict'          # proposals_training.py / classification_training.py checkpoint keys
EMBEDDER_STATE_DICT_KEY = 'model_state_dict'

_DIR = dict(exists=True, file_okay=False, dir_okay=True, readable=True)
_FILE = dict(exists=True, file_okay=True, dir_okay=False, readable=True)


EVAL_WINDOW = 8     # images read ahead and run through the kernels together by the evaluation commands


def _coco_or(thresholds, coco):
    return [f.item() for f in torch.linspace(.5, .95, 10)] if coco else list(thresholds)


def _load_encoder(dihe_state):
    encoder = classification.macvgg_embedder(model='vgg16', pretrained=False).cuda()
    state = torch.load(dihe_state, map_location='cpu')
    encoder.load_state_dict(state[EMBEDDER_STATE_DICT_KEY])
    return encoder.eval().requires_grad_(False)


@click.group()
def cli():
    """Computer vision based planogram compliance evaluation -- MI355X inference / evaluation commands."""


# ------------------------------------------------------------------ gln ------------------------------------------------
@cli.group()
def gln():
    """Product proposal generation (GLN)."""


@gln.command()
@click.option('--dataset', type=click.Choice(('sku110k', 'gp180', 'gpbaseline')), default='sku110k', show_default=True,
              help='Dataset type that --imgs and --annotations points to')
@click.option('--imgs', type=click.Path(**_DIR), default=SKU110K_IMG_DIR, show_default=True, help='Path to image dir')
@click.option('--annotations', type=click.Path(exists=True), default=SKU110K_ANNOTATION_FILE, show_default=True,
              help='Path to annotations used for testing')
@click.option('--batch-size', type=int, default=1, show_default=True, help='Batch size')
@click.option('--dataloader-workers', type=int, default=4, show_default=True, help='(ignored) Number of data loading processes')
@click.option('--metric-workers', type=int, default=8, show_default=True, help='Number of metric calculating processes (0 = in-process)')
@click.option('--iou-threshold', '-t', type=float, multiple=True, default=(0.5,), show_default=True,
              help='IoU thresholds to calculate metrics for')
@click.option('--coco/--no-coco', default=False, show_default=True,
              help="Whether to use COCO IoU thresholds instead of whatever's set in --iou-threshold")
@click.option('--trim-module-prefix/--no-trim-module-prefix', default=False, show_default=True,
              help='Trim "module." prefix from loaded model')
@click.option('--plots/--no-plots', default=True, show_default=True, help='(ignored) precision-recall curves are not drawn')
@click.option('--plot-res-reduction', type=int, default=200, show_default=True, help='(ignored)')
@click.argument('state-file', type=click.Path(**_FILE))
def eval(dataset, imgs, annotations, batch_size, dataloader_workers, metric_workers, iou_threshold, coco,
         trim_module_prefix, plots, plot_res_reduction, state_file):
    """Evaluate product proposal generation performance.

    STATE_FILE should point to some GLN weights."""
    if dataset == 'sku110k':
        data = datautils.SKU110KDataset(imgs, annotations, skip=SKU110K_SKIP, include_gaussians=False, flip_chance=0)
    elif dataset == 'gp180':
        data = datautils.GroceryProductsTestSet(imgs, annotations, retinanet_annotations=True)
    else:
        data = datautils.GPBaselineDataset(imgs, annotations)
    thresholds = _coco_or(iou_threshold, coco)
    evaluation = proposals_eval.evaluate_gln(state_file, data, thresholds=thresholds, batch_size=batch_size,
                                             num_metric_processes=metric_workers, trim_module_prefix=trim_module_prefix)
    ap = ar = 0
    for t in thresholds:
        print(f'{t}:\t{evaluation[t]}')
        ap += evaluation[t]['ap']
        ar += evaluation[t]['ar_300']
    print(f'--> AP {ap / len(thresholds)}')
    print(f'--> AR300 {ar / len(thresholds)}')


@gln.command()
@click.option('--conf-thresh', type=float, default=0.5, show_default=True, help='Confidence threhsold for detections')
@click.option('--save', type=click.Path(writable=True), help='Path to save the resulting image to')
@click.argument('state-file', type=click.Path(**_FILE))
@click.argument('image-file', type=click.Path(**_FILE))
def detect(conf_thresh, save, state_file, image_file):
    """Detect products; prints the boxes (x1 y1 x2 y2) and, with --save, writes the image with the boxes drawn."""
    from PIL import Image, ImageDraw
    model = proposals.gln(pretrained_backbone=False).cuda()
    model.load_state_dict(torch.load(state_file, map_location='cpu')[MODEL_STATE_DICT_KEY])
    model.eval()
    generator = production.ProposalGenerator(model, confidence_threshold=conf_thresh)
    pil_img = Image.open(image_file).convert('RGB')
    with torch.no_grad():
        detections = generator.generate_proposals(datautils.pil_to_tensor(pil_img)).cpu()
    for x1, y1, x2, y2 in detections.tolist():
        print(f'{x1:.1f} {y1:.1f} {x2:.1f} {y2:.1f}')
    print(f'--> {len(detections)} detections')
    if save is not None:
        draw = ImageDraw.Draw(pil_img)
        for box in detections.tolist():
            draw.rectangle(box, outline=(0, 255, 0), width=3)
        pil_img.save(save)


# ------------------------------------------------------------------ dihe -----------------------------------------------
@cli.group()
def dihe():
    """Product classification (DIHE embedder)."""


@dihe.command(name='eval')
@click.option('--img-dir', type=click.Path(**_DIR), multiple=True, default=GP_TRAIN_FOLDERS, show_default=True,
              help='Path to GP training image root')
@click.option('--test-imgs', type=click.Path(**_DIR), default=GP_TEST_DIR, show_default=True, help='Path to GP test image root')
@click.option('--annotations', type=click.Path(**_DIR), default=GP_ANN_DIR, show_default=True, help='Path to GP-180 annotation root')
@click.option('--model', type=click.Choice(('vgg16', 'resnet50')), default='vgg16', show_default=True, help='Base for the encoder model')
@click.option('--resnet-layers', type=int, multiple=True, default=[2, 3], show_default=True, help='(resnet50 only)')
@click.option('--batch-norm/--no-batch-norm', default=False, show_default=True, help="Use/don't use batch normalization in encoder")
@click.option('--batch-size', type=int, default=8, show_default=True, help='Batch size')
@click.option('--dataloader-workers', type=int, default=8, show_default=True, help='(ignored) Number of data loading processes')
@click.option('--enc-weights', help='Path to encoder weights')
@click.option('--only', type=click.Choice(('none', 'test', 'val')), default='none', show_default=True,
              help='Use all images (none) or only the test- or validation split')
@click.option('--knn', type=int, multiple=True, default=(1,), show_default=True,
              help='Consider classification correct if the correct class is among this many closest neighbours')
def dihe_eval(img_dir, test_imgs, annotations, model, resnet_layers, batch_norm, batch_size, dataloader_workers, enc_weights,
              only, knn):
    """Evaluate classification performance."""
    if enc_weights is None:
        raise click.UsageError('--enc-weights is required: ImageNet weights cannot be downloaded here')
    if model == 'vgg16':
        encoder = classification.macvgg_embedder(model='vgg16_bn' if batch_norm else 'vgg16', pretrained=False).cuda()
    else:
        encoder = classification.macresnet_encoder(pretrained=False, desc_layers=list(resnet_layers)).cuda()
    encoder.load_state_dict(torch.load(enc_weights, map_location='cpu')[EMBEDDER_STATE_DICT_KEY])
    encoder.eval().requires_grad_(False)
    sampleset = datautils.GroceryProductsDataset(img_dir, include_annotations=True)
    only_list = skip_list = None
    if only == 'test':
        skip_list = GP_TEST_VALIDATION_SET_SIZE
    elif only == 'val':
        only_list = GP_TEST_VALIDATION_SET_SIZE
    testset = datautils.GroceryProductsTestSet(test_imgs, annotations, only=only_list, skip=skip_list)
    accuracy = classification_eval.eval_dihe(encoder, sampleset, testset, batch_size, 0, k=knn, verbose=True)
    print(f'--> accuracy {accuracy}')


@dihe.command()
@click.option('--img-dir', type=click.Path(**_DIR), multiple=True, default=GP_TRAIN_FOLDERS, show_default=True,
              help='Path to training image root')
@click.option('--datatype', type=click.Choice(('gp', 'internal')), default='gp', show_default=True,
              help='Dataset type; gp for Grocery Product, internal for our internal dataset')
@click.option('--out-dir', type=click.Path(exists=True, file_okay=False, dir_okay=True, writable=True), default=OUT_DIR,
              show_default=True, help='Output directory for embedded images')
@click.argument('dihe-state', type=click.Path(**_FILE))
def prebuild_index(img_dir, datatype, out_dir, dihe_state):
    """Pre-embed images.

    Passes all the images in the training set through the encoder and saves the resulting embedding vectors
    (classifier_index.pkl, usable with --load-classifier-index)."""
    sampleset = (datautils.GroceryProductsDataset(img_dir, include_annotations=True) if datatype == 'gp'
                 else datautils.InternalTrainSet(img_dir[0], include_annotations=True))
    classifier = production.Classifier(_load_encoder(dihe_state), sampleset, verbose=True)
    classifier.save_index(os.path.join(out_dir, 'classifier_index.pkl'))


# ------------------------------------------------------------------ whole pipeline -------------------------------------
@cli.command()
@click.option('--img-dir', type=click.Path(**_DIR), multiple=True, default=GP_TRAIN_FOLDERS, show_default=True,
              help='Path to GP training image root')
@click.option('--test-imgs', type=click.Path(**_DIR), default=GP_TEST_DIR, show_default=True, help='Path to GP test image root')
@click.option('--annotations', type=click.Path(**_DIR), default=GP_ANN_DIR, show_default=True, help='Path to GP-180 annotation root')
@click.option('--iou-threshold', '-t', type=float, multiple=True, default=(0.5,), show_default=True,
              help='IoU thresholds to calculate metrics for')
@click.option('--coco/--no-coco', default=False, show_default=True,
              help="Whether to use COCO IoU thresholds instead of whatever's set in --iou-threshold")
@click.option('--load-classifier-index', type=click.Path(),
              help='Load pre-embedded images from given path instead of calculating the embedding on the fly')
@click.argument('gln-state', type=click.Path(**_FILE))
@click.argument('dihe-state', type=click.Path(**_FILE))
def eval_product_detection(img_dir, test_imgs, annotations, iou_threshold, coco, load_classifier_index, gln_state, dihe_state):
    """Evaluate product detection performance."""
    sampleset = datautils.GroceryProductsDataset(img_dir, include_annotations=True)
    testset = datautils.GroceryProductsTestSet(test_imgs, annotations, retinanet_annotations=True)
    thresholds = _coco_or(iou_threshold, coco)
    proposal_generator = proposals_eval.load_gln(gln_state, False, detections_per_img=200)
    proposal_generator.requires_grad_(False)
    res, all_res = detection_eval.evaluate_detections(proposal_generator, _load_encoder(dihe_state), testset, sampleset,
                                                      thresholds=thresholds, load_classifier_index=load_classifier_index)
    mam = detection_eval.mean_average_metrics(res, thresholds)
    m_ap = m_ar = 0
    for t in thresholds:
        print(t, all_res[t])
        print(t, mam[t])
        m_ap += mam[t]['map']
        m_ar += mam[t]['mar300']
    print(f'--> mAP {m_ap / len(thresholds)}')
    print(f'--> mAR300 {m_ar / len(thresholds)}')


@cli.command()
@click.option('--img-dir', type=click.Path(**_DIR), multiple=True, default=GP_TRAIN_FOLDERS, show_default=True,
              help='Path to training image root')
@click.option('--test-imgs', type=click.Path(**_DIR), default=GP_TEST_DIR, show_default=True, help='Path to test image root')
@click.option('--test-annotations', type=click.Path(**_DIR), default=GP_ANN_DIR, show_default=True, help='Path to annotation root')
@click.option('--planograms', type=click.Path(**_DIR), default=GP_PLANO_DIR, show_default=True, help='Path to planograms root')
@click.option('--datatype', type=click.Choice(('gp', 'internal')), default='gp', show_default=True,
              help='Dataset type; gp for Grocery Product, internal for our internal dataset')
@click.option('--load-classifier-index', type=click.Path(),
              help='Load pre-embedded images from given path instead of calculating the embedding on the fly')
@click.option('--verbose/--no-verbose', default=False, show_default=True, help='Print more information')
@click.argument('gln-state', type=click.Path(**_FILE))
@click.argument('dihe-state', type=click.Path(**_FILE))
def eval_planograms(img_dir, test_imgs, test_annotations, planograms, datatype, load_classifier_index, verbose, gln_state,
                    dihe_state):
    """Evaluate planogram compliance evaluation."""
    if datatype == 'gp':
        planoset = datautils.PlanogramTestSet(test_imgs, test_annotations, planograms)
        sampleset = datautils.GroceryProductsDataset(img_dir, include_annotations=True)
    else:
        planoset = datautils.InternalPlanoSet(planograms)
        sampleset = datautils.InternalTrainSet(img_dir[0], include_annotations=True)
    proposal_generator = proposals_eval.load_gln(gln_state, False)
    proposal_generator.requires_grad_(False)
    generator = production.ProposalGenerator(proposal_generator)
    classifier = production.Classifier(_load_encoder(dihe_state), sampleset, batch_size=8, load=load_classifier_index)
    evaluator = production.PlanogramEvaluator(generator, classifier, production.PlanogramComparator())
    total_a = total_e = 0.0
    # the reference evaluates one image at a time (cvpce/cli/eval.py:224-230); here EVAL_WINDOW images are read ahead and detected /
    # embedded / matched together (PlanogramEvaluator.evaluate_batch: same per-image results, one pass of the kernels per window)
    for s in range(0, len(planoset), EVAL_WINDOW):
        data = [planoset[i] for i in range(s, min(s + EVAL_WINDOW, len(planoset)))]
        pairs = [((d[0], d[3]) if datatype == 'gp' else d) for d in data]
        accs = evaluator.evaluate_batch([img for img, _ in pairs], [plano for _, plano in pairs])
        for i, ((img, plano), acc) in enumerate(zip(pairs, accs), start=s):
            acc = float(acc)
            err = acc - plano['actual_accuracy']
            if verbose:
                print(f'Detected accuracy: {acc:.3f}, Actual accuracy: {plano["actual_accuracy"]:.3f}, Error: {err:.3f}, SE: {err ** 2:.3f}')
            elif i % 10 == 0:
                print(i)
            total_e += err ** 2
            total_a += acc
    print(f'--> Mean accuracy {total_a / len(planoset)}')
    print(f'--> MSE: {total_e / len(planoset)}')


if __name__ == '__main__':
    cli()
