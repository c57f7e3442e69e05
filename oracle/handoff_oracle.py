"""handoff_oracle.py -- CPU restatement of the reference's image hand-off.  TEST INFRASTRUCTURE ONLY:
imported by tests/ (and nothing else); the product path is csrc/nsr_handoff.hip.

What it restates (reference file:line):
  * to8b                       RH:14           (255*np.clip(x,0,1)).astype(np.uint8)
  * PNG round trip             RN:245-250 -> NM:793 cv2.imread: lossless, RGB in the file, BGR in memory
  * get_annotation / find_bbox NM:786-797      gray = cv2.cvtColor(img, COLOR_RGB2GRAY) on that BGR array,
                                               cv2.threshold(gray, 1, 255, THRESH_BINARY),
                                               cv2.connectedComponentsWithStats (8-connectivity, row 0 = zero pixels),
                                               stats[stats[:,4].argsort()][:-1]
  * the box actually used      NM:691-692, NM:817-818   argmax of w*h over the remaining rows

PARITY UNPINNED for the OpenCV pieces: `opencv-python` is an un-pinned requirement (requirements.txt:9) and is
not installed in this image, so cv2 itself cannot be run here.  Its published algorithm is restated instead:
  * RGB2GRAY for uint8 (OpenCV 4.x, modules/imgproc/src/color_rgb.simd.hpp, RGB2Gray<uchar>):
        (c0*RY15 + c1*GY15 + c2*BY15 + (1 << 14)) >> 15,  RY15 = 9798, GY15 = 19235, BY15 = 3735
    with c0 the first channel of the array handed in -- blue here, because the reference passes imread's BGR.
  * connectedComponentsWithStats: labels in raster order of each component's first pixel, 8-connectivity,
    stats columns [left, top, width, height, area]; cross-checked below against scipy.ndimage.label.
  * numpy's argsort on the area column is treated as stable (it is for the <= 16 rows that occur in practice:
    insertion sort); equal areas are vanishingly rare and the rule is written down so the GPU can match it.
to8b and the PNG round trip are pinned (numpy semantics; PNG files against imageio 2.9.0 -- the library the reference
writes and reads them with -- via tests/golden/io_*.png, oracle/gen_io_golden.py).  connected_components_with_stats is
pinned to a second implementation of the same documented contract: scikit-image 0.18.3 measure.label(connectivity=2) +
regionprops on 24 masks (tests/golden/g12_io.npz, tests/test_data_readers.py).  What stays a restated formula is the
8-bit RGB2GRAY fixed-point conversion: nothing in this image implements OpenCV's.  It is unpinned AND shown not to matter:
the reference only thresholds the grey image at > 1 (NM:795), and over all 2^24 colours the restated formula, OpenCV 3.x's
14-bit variant and the float formula under either rounding rule give the same mask bit
(tests/test_handoff.py::test_mask_is_insensitive_to_the_unpinned_gray_arithmetic), hence the same components and boxes.
"""
import numpy as np
from scipy import ndimage


def to8b(x):
    """RH:14."""
    with np.errstate(invalid="ignore"):
        return (255 * np.clip(np.asarray(x, np.float32), 0, 1)).astype(np.uint8)


def gray_as_reference(rgb8):
    """cv2.cvtColor(cv2.imread(png), COLOR_RGB2GRAY) for a PNG that holds `rgb8` (NM:793-794)."""
    c = rgb8.astype(np.int64)
    b, g, r = c[..., 2], c[..., 1], c[..., 0]            # imread order: channel 0 = blue
    return ((b * 9798 + g * 19235 + r * 3735 + (1 << 14)) >> 15).astype(np.uint8)


def mask_as_reference(rgb8):
    """cv2.threshold(gray, 1, 255, THRESH_BINARY) (NM:795)."""
    return np.where(gray_as_reference(rgb8) > 1, 255, 0).astype(np.uint8)


def connected_components_with_stats(mask):
    """cv2.connectedComponentsWithStats(mask) for a 2-D uint8 mask: (n_labels, labels, stats[n,5])."""
    fg = mask != 0
    labels, n = ndimage.label(fg, structure=np.ones((3, 3), int))      # raster order of first pixel, 8-connectivity
    H, W = mask.shape
    stats = np.zeros((n + 1, 5), np.int64)
    ys, xs = np.nonzero(~fg)
    if len(ys):
        stats[0] = [xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1, len(ys)]
    for lab, sl in enumerate(ndimage.find_objects(labels), start=1):
        stats[lab] = [sl[1].start, sl[0].start, sl[1].stop - sl[1].start, sl[0].stop - sl[0].start,
                      int((labels[sl] == lab).sum())]
    return n + 1, labels, stats


def find_bbox(mask):
    """NM:786-789: rows sorted by area, the largest dropped."""
    _, _, stats = connected_components_with_stats(mask)
    stats = stats[np.argsort(stats[:, 4], kind="stable")]
    return stats[:-1]


def get_annotation(rgb8):
    """NM:791-797 + the selection at NM:691-692: returns (bbox XYWH or None, n_rows, mask)."""
    mask = mask_as_reference(rgb8)
    rows = find_bbox(mask)[:, :-1]
    if rows.shape[0] == 0:
        return None, 0, mask                        # the reference raises here (np.argmax of an empty array)
    best = rows[int(np.argmax(rows[:, -2] * rows[:, -1]))]
    return best, rows.shape[0], mask
