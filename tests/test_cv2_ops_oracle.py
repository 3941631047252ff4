"""CPU: sanity properties of the numpy restatement of OpenCV's 8-bit resize / warp arithmetic (oracle/cv2_ops.py).
cv2 is absent here, so these are properties the real functions are known to have, not comparisons with cv2
(the file header says PARITY UNPINNED)."""
import numpy as np

from oracle import cv2_ops as CV


def test_identity_resizes_return_the_image():
    img = np.random.default_rng(0).integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(CV.resize_cubic_u8(img, (37, 53)), img)
    assert np.array_equal(CV.resize_linear_u8(img, (37, 53)), img)


def test_constant_images_stay_constant():
    img = np.full((40, 60, 3), 137, np.uint8)
    for fn in (CV.resize_cubic_u8, CV.resize_linear_u8):
        for hw in ((17, 23), (96, 131), (40, 200)):
            assert (fn(img, hw) == 137).all()


def test_cubic_resize_overshoots_and_saturates():
    img = np.zeros((8, 64, 3), np.uint8)
    img[:, 32:] = 255                                   # a hard edge: the cubic kernel's negative lobes clip at 0 / 255
    out = CV.resize_cubic_u8(img, (8, 200))
    assert out.min() == 0 and out.max() == 255
    lin = CV.resize_linear_u8(img, (8, 200))
    assert (np.diff(lin[0, :, 0].astype(int)) >= 0).all()      # linear interpolation is monotone across the edge


def test_remap_table_blocks_sum_to_one():
    tab = CV.cubic_remap_table()
    assert tab.shape == (1024, 4, 4) and (tab.reshape(1024, 16).sum(1) == 32768).all()
    assert tab[0, 1, 1] == 32767 and tab[0, 2, 2] == 1 and tab[0].sum() == 32768     # integer position: centre tap (int16 max) + 1


def test_identity_warp_and_integer_shift():
    img = np.random.default_rng(1).integers(0, 256, (30, 40, 3), dtype=np.uint8)
    eye = np.eye(3)
    assert np.array_equal(CV.warp_perspective_cubic_u8(img, eye, (40, 30)), img)
    shift = np.array([[1, 0, 5], [0, 1, 3], [0, 0, 1]], float)           # destination (x, y) reads source (x + 5, y + 3)
    out = CV.warp_perspective_cubic_u8(img, shift, (20, 10))
    assert np.array_equal(out, img[3:13, 5:25])
    far = CV.warp_perspective_cubic_u8(img, np.array([[1, 0, 100], [0, 1, 0], [0, 0, 1]], float), (4, 30))
    assert np.array_equal(far, np.repeat(img[:, -1:], 4, axis=1))        # BORDER_REPLICATE


def test_rotate_crop_and_resize_norm():
    img = np.random.default_rng(2).integers(0, 256, (200, 300, 3), dtype=np.uint8)
    quad = np.array([[40, 50], [240, 50], [240, 90], [40, 90]], np.float32)
    crop = CV.get_rotate_crop_image(img, quad)
    assert crop.shape == (40, 200, 3) and np.array_equal(crop, img[50:90, 40:240])
    tall = CV.get_rotate_crop_image(img, np.array([[10, 10], [30, 10], [30, 110], [10, 110]], np.float32))
    assert tall.shape == (20, 100, 3) and np.array_equal(tall, np.rot90(img[10:110, 10:30]))
    x = CV.resize_norm_img(crop, 320 / 48)
    assert x.shape == (3, 48, 320) and x.dtype == np.float32
    assert float(np.abs(x[:, :, 240:]).max()) == 0.0 and -1.0 <= x.min() and x.max() <= 1.0
    wide = CV.resize_norm_img(img[:20, :300], 320 / 48)                  # 48 * 15 > 320: capped, no padding
    assert wide.shape == (3, 48, 320) and float(np.abs(wide[:, :, -1]).max()) > 0


def test_layout_preprocess_shapes_and_norm():
    img = np.random.default_rng(3).integers(0, 256, (120, 90, 3), dtype=np.uint8)
    x = CV.layout_preprocess(img, 64)
    assert x.shape == (1, 3, 64, 64) and x.dtype == np.float32 and 0.0 <= x.min() and x.max() <= 1.0
    y = CV.layout_preprocess(img, 64, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
    assert np.allclose(y[0, 0], (x[0, 0] - 0.485) / 0.229, atol=1e-6)
