"""Default dataset / model locations of the eval commands (reference: cvpce/defaults.py): paths relative to the
package directory, `../data/...` and `../models`."""
from os import path

_HERE = path.dirname(path.abspath(__file__))


def rel_path(*parts):
    return path.join(_HERE, *parts)


DATA_DIR = ('..', 'data')
SKU110K_IMG_DIR = rel_path(*DATA_DIR, 'SKU110K_fixed', 'images')
SKU110K_ANNOTATION_FILE = rel_path(*DATA_DIR, 'SKU110K_fixed', 'annotations', 'annotations_val.csv')
# images the reference excludes from SKU-110K (corrupted files, missing ground truth, very poor images): defaults.py:13-18
SKU110K_SKIP = [
    'test_274.jpg', 'train_882.jpg', 'train_924.jpg', 'train_4222.jpg', 'train_5822.jpg',
    'train_789.jpg', 'train_5007.jpg', 'train_6090.jpg', 'train_7576.jpg',
    'train_104.jpg', 'train_890.jpg', 'train_1296.jpg', 'train_3029.jpg', 'train_3530.jpg', 'train_3622.jpg',
    'train_4899.jpg', 'train_6216.jpg', 'train_7880.jpg',
    'train_701.jpg', 'train_6566.jpg',
]
GP_ROOT = (*DATA_DIR, 'Grocery_products')
GP_TRAIN_FOLDERS = (rel_path(*GP_ROOT, 'Training'),)
GP_TEST_DIR = rel_path(*GP_ROOT, 'Testing')
GP_ANN_DIR = rel_path(*DATA_DIR, 'Planogram_Dataset', 'annotations')
GP_BASELINE_ANN_FILE = rel_path(*DATA_DIR, 'Baseline', 'Grocery_products_coco_gt_object.csv')
GP_PLANO_DIR = rel_path(*DATA_DIR, 'Planogram_Dataset', 'planograms')
GP_TEST_VALIDATION_SET = ['s1_15.csv', 's2_3.csv', 's2_30.csv', 's2_143.csv', 's2_157.csv', 's3_111.csv', 's3_260.csv', 's5_55.csv']
GP_TEST_VALIDATION_SET_SIZE = 2
GP_PLANO_VALIDATION_SET = [f'{s.split(".")[0]}.json' for s in GP_TEST_VALIDATION_SET]
OUT_DIR = rel_path('..', 'out')
