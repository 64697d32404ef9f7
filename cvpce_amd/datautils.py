"""Counterpart of /root/reference/cvpce/datautils.py: the hot-path member (`resize_for_classification`, on the GPU) and the
host-side dataset readers of the eval commands (below)."""
import torch


CLASSIFICATION_IMAGE_SIZE = 256


def resize_for_classification(img):
    """datautils.py:234-239: pad (3,h,w) to a square with 0.5 (top-left anchored), bilinear -> 256x256.

    Runs the K9 HIP gather kernel (the whole image is the crop box)."""
    if not img.is_cuda:
        raise RuntimeError('resize_for_classification runs on an MI355X (HIP) device only (no CPU fallback)')
    _, h, w = img.shape
    box = torch.tensor([[0.0, 0.0, float(w), float(h)]], device=img.device)
    from . import ops          # the only kernel call in this module: the readers below are host-only
    return ops.crop_resize(img.to(torch.float32).contiguous(), box, CLASSIFICATION_IMAGE_SIZE, mode=0)[0]


# ---------------------------------------------------------------------------------------------------------------------
# Dataset readers of the eval commands.  Host-side, CPU: csv / json / os / PIL like the reference.
# Only what inference needs: index building + image loading; the training-time augmentations (gaussian target maps,
# flips, random crops, alpha masks) are not rebuilt and raise if asked for.
# `to_tensor` / `resize` / `pad` are torchvision 0.9 functions in the reference (PARITY UNPINNED); restated:
#   to_tensor: uint8 HWC -> float32 CHW / 255;  resize: PIL bilinear to (h, w);  pad: constant fill right / bottom.
# ---------------------------------------------------------------------------------------------------------------------
import csv as _csv
import json as _json
import os as _os
import re as _re


def pil_to_tensor(img):
    """torchvision.transforms.functional.to_tensor for 8-bit PIL images: (C,H,W) float32 in [0,1]."""
    import numpy as np
    if img.mode not in ('L', 'RGB', 'RGBA'):
        img = img.convert('RGB')
    a = np.asarray(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(a.copy()).permute(2, 0, 1).to(torch.float32) / 255.0


def _open_image(path):
    from PIL import Image
    return Image.open(path)


class SKU110KDataset:
    """datautils.py:130-189, evaluation use (cli/gln.py:264): 8-column CSV `name,x1,y1,x2,y2,class,width,height`, one
    row per box, images grouped in order of first appearance.  Items: (image (3,H,W) float, entry dict)."""

    def __init__(self, img_dir_path, annotation_file_path, skip=(), include_gaussians=False, flip_chance=0, **_unused):
        if include_gaussians or flip_chance:
            raise NotImplementedError('training-time targets / augmentation are out of scope (DESIGN.md 6)')
        self.img_dir = img_dir_path
        self.index = self.build_index(annotation_file_path, skip)

    @staticmethod
    def build_index(annotation_file_path, skip=()):
        per_image = {}
        with open(annotation_file_path, 'r') as f:
            for row in _csv.reader(f):
                if len(row) != 8:
                    print(f'Malformed annotation row: {row}, skipping')
                    continue
                name, x1, y1, x2, y2, _, width, height = row
                if name in skip:
                    continue
                entry = per_image.setdefault(name, {'image_name': name, 'image_width': int(width), 'image_height': int(height), 'boxes': []})
                entry['boxes'].append([int(x1), int(y1), int(x2), int(y2)])
        index = list(per_image.values())
        for entry in index:
            entry['labels'] = torch.zeros(len(entry['boxes']), dtype=torch.long)
            entry['boxes'] = torch.tensor(entry['boxes'])
        return index

    def index_for_name(self, name):
        return next((i for i, e in enumerate(self.index) if e['image_name'] == name), None)

    def __len__(self):
        return len(self.index)

    def __getitem__(self, i):
        entry = dict(self.index[i])
        try:
            return pil_to_tensor(_open_image(_os.path.join(self.img_dir, entry['image_name']))), entry
        except OSError:
            print(f'WARNING: Malformed image: {entry["image_name"]} - returning image 0 ({self.index[0]["image_name"]}) instead.')
            return self[0]


class GPBaselineDataset:
    """datautils.py:191-232: one 6-column CSV with a header row, image files under <root>/storeN/images/."""

    def __init__(self, img_dir_path, annotation_file_path):
        self.index = self.build_index(img_dir_path, annotation_file_path)

    @staticmethod
    def build_index(image_dir_path, annotation_file_path):
        name_re = _re.compile(r'^(store\d)\_\d+.jpg$')
        per_image = {}
        with open(annotation_file_path, 'r') as f:
            for i, row in enumerate(_csv.reader(f)):
                if i == 0:
                    continue
                if len(row) != 6:
                    print(f'Malformed annotation row: {row}, skipping')
                    continue
                name, x1, y1, x2, y2, _ = row
                if name not in per_image:
                    m = name_re.match(name)
                    if m is None:
                        print(f'Malformed annotation row: {row}, skipping')
                        continue
                    per_image[name] = {'image_path': _os.path.join(image_dir_path, m.group(1), 'images', name), 'boxes': []}
                per_image[name]['boxes'].append([int(x1), int(y1), int(x2), int(y2)])
        index = list(per_image.values())
        for entry in index:
            entry['labels'] = torch.zeros(len(entry['boxes']), dtype=torch.long)
            entry['boxes'] = torch.tensor(entry['boxes'])
        return index

    def __len__(self):
        return len(self.index)

    def __getitem__(self, i):
        entry = self.index[i]
        return pil_to_tensor(_open_image(entry['image_path'])), entry


class GroceryProductsTestSet:
    """datautils.py:623-688: GP-180, one 5-column CSV `Category/.../name.jpg, x1, y1, x2, y2` per test image, named
    s<store>_<image>.csv.  `only` / `skip` are lists of file names, or ints = keep the first `only` / drop the first
    `skip` annotations of every image (the reference's validation split).  Files are visited in sorted order (the
    reference takes os.scandir order, which is unspecified) and the label ids of `retinanet_annotations` are the sorted
    annotation strings (the reference enumerates a set)."""

    def __init__(self, image_dir, ann_dir, only=None, skip=None, retinanet_annotations=False):
        self.image_dir = image_dir
        self.toskip = skip if type(skip) == int else 0
        self.tokeep = only if type(only) == int else 9999
        self.index = self.build_index(ann_dir, None if type(only) == int else only, None if type(skip) == int else skip)
        self.int_to_ann = sorted({ann for e in self.index for ann in e['anns']})
        self.ann_to_int = {ann: i for i, ann in enumerate(self.int_to_ann)}
        self.retinanet_annotations = retinanet_annotations

    def get_image_path(self, store, image):
        return _os.path.join(self.image_dir, f'store{store}', 'images', f'store{store}_{image}.jpg')

    def build_index(self, ann_dir, only=None, skip=None):
        file_re = _re.compile(r'^s(\d+)_(\d+)\.csv$')
        ann_re = _re.compile(r'^(.+)\.jpg')
        index = []
        for name in sorted(_os.listdir(ann_dir)):
            full = _os.path.join(ann_dir, name)
            if not _os.path.isfile(full) or (only is not None and name not in only) or (skip is not None and name in skip):
                continue
            m = file_re.match(name)
            if m is None:
                continue
            anns, boxes = [], []
            with open(full, 'r') as f:
                for row in _csv.reader(f, skipinitialspace=True):
                    if len(row) != 5:
                        print(f'Malformed annotation row in file {name}: {row}; skipping')
                        continue
                    ann, x1, y1, x2, y2 = row
                    am = ann_re.match(ann)
                    if am is None:      # (the reference prints and then fails on the next line; skipping is the stated intent)
                        print(f'Non-conforming annotation in file {name}: {ann}; skipping')
                        continue
                    anns.append(am.group(1))
                    boxes.append([int(x1), int(y1), int(x2), int(y2)])
            index.append({'id': (m.group(1), m.group(2)), 'path': self.get_image_path(m.group(1), m.group(2)),
                          'anns': anns, 'boxes': torch.tensor(boxes)})
        return index

    def get_index_for(self, store, image):
        target = self.get_image_path(store, image)
        return next((i for i, e in enumerate(self.index) if e['path'] == target), None)

    def __len__(self):
        return len(self.index)

    def __getitem__(self, i):
        e = self.index[i]
        img = pil_to_tensor(_open_image(e['path']))
        anns, boxes = e['anns'][self.toskip:self.tokeep], e['boxes'][self.toskip:self.tokeep]
        if self.retinanet_annotations:
            return img, {'labels': torch.tensor([self.ann_to_int[a] for a in anns], dtype=torch.long), 'boxes': boxes}
        return img, anns, boxes


class PlanogramTestSet(GroceryProductsTestSet):
    """datautils.py:692-707: GP-180 test images + their Tonioni planograms (s<store>_<image>.json)."""

    def __init__(self, image_dir, ann_dir, plano_dir, only=None, skip=None):
        self.plano_dir = plano_dir
        super().__init__(image_dir, ann_dir, only, skip)

    def build_index(self, ann_dir, only=None, skip=None):
        from . import planogram_adapters
        index = super().build_index(ann_dir, only, skip)
        for e in index:
            s, i = e['id']
            boxes, labels, g = planogram_adapters.read_tonioni_planogram(_os.path.join(self.plano_dir, f's{s}_{i}.json'))
            e['plano'] = {'boxes': boxes, 'labels': labels, 'graph': g, 'actual_accuracy': 1.0}
        return index

    def __getitem__(self, i):
        img, anns, boxes = super().__getitem__(i)
        return img, anns, boxes, self.index[i]['plano']


class InternalPlanoSet:
    """datautils.py:709-750: index.json -> [{image, planogram, correct, facings}], planogram = [{code, box}] with the
    y axis pointing up (flipped here to detector coordinates)."""

    def __init__(self, dir):
        self.index = self.build_index(dir)

    @staticmethod
    def build_index(dir):
        with open(_os.path.join(dir, 'index.json'), 'r') as f:
            listing = _json.load(f)
        res = []
        for obj in listing:
            with open(_os.path.join(dir, obj['planogram']), 'r') as f:
                plano = _json.load(f)
            boxes = torch.tensor([e['box'] for e in plano], dtype=torch.float)
            top = boxes[:, 3].max()
            y1, y2 = top - boxes[:, 3], top - boxes[:, 1]
            boxes[:, 1], boxes[:, 3] = y1, y2
            res.append({'img': _os.path.join(dir, obj['image']), 'anns': [e['code'] for e in plano], 'boxes': boxes,
                        'actual_accuracy': obj['correct'] / obj['facings']})
        return res

    def __len__(self):
        return len(self.index)

    def __getitem__(self, i):
        e = self.index[i]
        return pil_to_tensor(_open_image(e['img'])), {'labels': e['anns'], 'boxes': e['boxes'], 'actual_accuracy': e['actual_accuracy']}


class GroceryProductsDataset:
    """datautils.py:301-451, gallery use (`include_annotations=True`, cli/eval.py:45): walks the GP training tree
    (<root>/<category>/.../<name>.<ext>), skipping hierarchies that match `skip`; annotation = 'Category/.../name'.
    Items: (image, image, categories[, annotation]) with the image resized so that its longer side is 256, scaled to
    [-1,1] and padded right / bottom with 0 to 256x256 (`tensorize`, :397-415).  Directory entries are visited in sorted
    order (the reference takes os.scandir order).  Random crops and masks are training-time and not rebuilt."""

    IGNORED_FILES = ('.DS_Store', 'index.txt', 'TrainingClassesIndex.mat', 'classes.csv', 'Thumbs.db')

    def __init__(self, image_roots, skip=(r'^Background.*$', r'^.*/[Oo]riginals?$'), only=None, random_crop=False,
                 resize=True, test_can_load=False, include_annotations=False, include_masks=False, index_from_file=False,
                 **_unused):
        if include_masks:
            raise NotImplementedError('training-time masks are out of scope (DESIGN.md 6)')
        if isinstance(image_roots, str):
            image_roots = (image_roots,)
        skip_re = _re.compile('|'.join(f'({s})' for s in skip))
        build = self.build_index_from_file if index_from_file else self.build_index
        self.paths, self.categories, self.annotations = build(image_roots, skip_re, only, test_can_load)
        self.resize = resize
        self.include_annotations = include_annotations

    @classmethod
    def build_index(cls, image_roots, skip, only=None, test_can_load=False):
        stem_re = _re.compile(r'^(.+)\.\w+$')
        paths, categories, annotations = [], [], []
        for root in image_roots:
            stack = [(root, [])]
            while stack:
                cur, hier = stack.pop()
                if skip.match('/'.join(hier)) is not None:
                    continue
                if only is not None and hier and hier[0] not in only:
                    continue
                for name in sorted(_os.listdir(cur)):
                    full = _os.path.join(cur, name)
                    if _os.path.isdir(full) and not _os.path.islink(full):
                        stack.append((full, hier + [name]))
                    elif _os.path.isfile(full):
                        if name in cls.IGNORED_FILES or skip.match('/'.join(hier + [name])):
                            continue
                        if test_can_load:
                            try:
                                _open_image(full)
                            except OSError:
                                continue
                        m = stem_re.match(name)
                        if m is None:
                            print(f'Nonconforming filename: {name}, skipping')
                            continue
                        paths.append(full)
                        categories.append(hier)
                        annotations.append('/'.join(hier + [m.group(1)]))
        return paths, categories, annotations

    @staticmethod
    def build_index_from_file(dataset_roots, skip, only=None, test_can_load=False, index_filename='TrainingFiles.txt'):
        paths, categories, annotations = [], [], []
        for root in dataset_roots:
            with open(_os.path.join(root, index_filename), 'r') as f:
                for line in f:
                    parts = line.strip().split('/')
                    if len(parts) < 2:
                        continue
                    hier = parts[1:-1]          # drop the leading "Training" folder and the file name
                    if only is not None and hier[0] not in only:
                        continue
                    if skip.match('/'.join(hier)) is not None:
                        continue
                    paths.append(_os.path.join(root, *parts))
                    categories.append(hier)
                    annotations.append('/'.join(parts[1:]))
        return paths, categories, annotations

    def index_for_ann(self, ann):
        return next((i for i, a in enumerate(self.annotations) if a == ann), None)

    def tensorize(self, img, tanh=False):
        from PIL import Image
        if not self.resize:
            return pil_to_tensor(img)
        S = CLASSIFICATION_IMAGE_SIZE
        h, w = (S, round(S * img.width / img.height)) if img.height > img.width else (round(S * img.height / img.width), S)
        t = pil_to_tensor(img.resize((w, h), Image.BILINEAR))[:3]
        if tanh:
            t = t * 2 - 1
        out = torch.full((t.shape[0], S, S), 0.0 if tanh else 0.5)
        out[:, :h, :w] = t
        return out

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, i):
        img = _open_image(self.paths[i])
        if img.mode != 'RGB':
            img = img.convert('RGB')
        t = self.tensorize(img, True)
        if self.include_annotations:
            return t, t, self.categories[i], self.annotations[i]
        return t, t, self.categories[i]


class InternalTrainSet(GroceryProductsDataset):
    """datautils.py:453-482: product renders with an alpha channel; the annotation is the numeric product code at the
    end of the path, transparent pixels become white."""

    def __init__(self, root, skip=(r'^Unknown.*$',), resize=True, include_annotations=False, **_unused):
        super().__init__([root], skip=skip, resize=resize, include_annotations=include_annotations)
        code_re = _re.compile(r'^(.+/)*(\d+)')
        self.annotations = [code_re.match(a).group(2) for a in self.annotations]

    def index_for_ann(self, ann):
        best = None
        for i, a in enumerate(self.annotations):
            if a != ann:
                continue
            if 'front' in self.categories[i]:
                return i
            if 'back' in self.categories[i] or best is None:
                best = i if ('back' in self.categories[i] or best is None) else best
        return best

    def __getitem__(self, i):
        from PIL import Image
        img = _open_image(self.paths[i]).convert('RGBA')
        white = Image.new('RGBA', img.size, (255, 255, 255, 255))
        a = pil_to_tensor(img)
        rgb = a[:3].clone()
        rgb[:, a[3] == 0] = 1.0
        img = Image.fromarray((rgb.permute(1, 2, 0) * 255).round().to(torch.uint8).numpy(), 'RGB')
        del white
        t = self.tensorize(img, True)
        if self.include_annotations:
            return t, t, self.categories[i], self.annotations[i]
        return t, t, self.categories[i]
