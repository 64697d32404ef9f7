"""Where the eval commands look for data when no path is given (same locations and names as the reference's
cvpce/defaults.py: a `data/` and an `out/` directory next to the package directory)."""
import os

_PKG = os.path.dirname(os.path.abspath(__file__))


def rel_path(*parts):
    """Path relative to the package directory."""
    return os.path.join(_PKG, *parts)


def _data(*parts):
    return rel_path('..', 'data', *parts)


# --- SKU-110K (detector evaluation) ---
SKU110K_IMG_DIR = _data('SKU110K_fixed', 'images')
SKU110K_ANNOTATION_FILE = _data('SKU110K_fixed', 'annotations', 'annotations_val.csv')
# images the reference leaves out: files that do not decode or decode corrupted, images missing most of their boxes,
# and two very poor photos (cvpce/defaults.py:13-18); kept as (split, ids) and expanded to file names
_SKU110K_EXCLUDED = (
    ('test', (274,)),
    ('train', (882, 924, 4222, 5822, 789, 5007, 6090, 7576, 104, 890, 1296, 3029, 3530, 3622, 4899, 6216, 7880, 701, 6566)),
)
SKU110K_SKIP = [f'{split}_{i}.jpg' for split, ids in _SKU110K_EXCLUDED for i in ids]

# --- Grocery Products / GP-180 (classification, product detection, planograms) ---
GP_TRAIN_FOLDERS = (_data('Grocery_products', 'Training'),)
GP_TEST_DIR = _data('Grocery_products', 'Testing')
GP_ANN_DIR = _data('Planogram_Dataset', 'annotations')
GP_PLANO_DIR = _data('Planogram_Dataset', 'planograms')
GP_BASELINE_ANN_FILE = _data('Baseline', 'Grocery_products_coco_gt_object.csv')
# validation split of GP-180: these shelf images, or (as an int) the first N annotations of every image
_GP_VALIDATION_IMAGES = ((1, 15), (2, 3), (2, 30), (2, 143), (2, 157), (3, 111), (3, 260), (5, 55))
GP_TEST_VALIDATION_SET = [f's{store}_{image}.csv' for store, image in _GP_VALIDATION_IMAGES]
GP_PLANO_VALIDATION_SET = [f's{store}_{image}.json' for store, image in _GP_VALIDATION_IMAGES]
GP_TEST_VALIDATION_SET_SIZE = 2

OUT_DIR = rel_path('..', 'out')
