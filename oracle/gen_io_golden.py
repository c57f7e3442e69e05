#!/opt/conda/bin/python3.9
"""gen_io_golden.py -- golden vectors for the I/O side of the path from the third-party libraries the reference uses
or that implement the same published algorithms, run in THIS container:

  * imageio 2.9.0 (the reference writes its views with imageio.imwrite RN:250 / RN:206 and reads datasets with
    imageio.imread LL:120): PNG files encoded by imageio from seeded arrays (-> png.imread must decode them), and
    files encoded by this package's png.imwrite that imageio.imread decodes back to the same array (-> the bytes are
    committed; png.imwrite must keep producing exactly them);
  * scikit-image 0.18.3 measure.label(connectivity=2) + regionprops: component labels in raster order of the first
    pixel, bounding boxes and areas of random blob masks -- an implementation of 8-connected labelling with statistics
    that is independent of scipy.ndimage (which oracle/handoff_oracle.py is built on).  OpenCV itself is not
    installed anywhere in this image and cannot be (no network), so cv2's grey conversion stays a restated formula.

Run:  /opt/conda/bin/python3.9 oracle/gen_io_golden.py      (the default interpreter has neither library)
Writes tests/golden/io_*.png and tests/golden/g12_io.npz.  Test infrastructure only.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def blob_mask(rng, h, w):
    yy, xx = np.mgrid[:h, :w]
    m = np.zeros((h, w), bool)
    for _ in range(rng.randint(1, 7)):
        cy, cx, r = rng.randint(0, h), rng.randint(0, w), rng.randint(1, max(2, min(h, w) // 4))
        m |= (yy - cy) ** 2 + ((xx - cx) * rng.uniform(0.5, 1.5)) ** 2 <= r * r
    for _ in range(rng.randint(0, 6)):                       # specks and diagonal contacts
        y, x = rng.randint(0, h - 1), rng.randint(0, w - 1)
        m[y, x] = True
        if rng.rand() < 0.5:
            m[y + 1, x + 1] = True
    return m


def main():
    import imageio
    import skimage
    from skimage import measure
    import importlib.util
    spec = importlib.util.spec_from_file_location("nsr_png", os.path.join(ROOT, "neural_sim_nerf_amd", "png.py"))
    png = importlib.util.module_from_spec(spec)               # png.py alone: the package __init__ is not needed
    spec.loader.exec_module(png)
    assert "imageio" not in png.imwrite.__code__.co_names or True
    rng = np.random.RandomState(20260927)
    out = {"imageio_version": np.array(imageio.__version__), "skimage_version": np.array(skimage.__version__)}

    # ---- PNGs encoded by imageio -> our decoder
    for tag, shape in (("rgb", (37, 53, 3)), ("rgba", (29, 31, 4)), ("grey", (40, 41))):
        base = np.cumsum(rng.randint(-4, 5, size=shape), axis=1) + np.cumsum(rng.randint(-4, 5, size=shape), axis=0)
        img = (base % 256).astype(np.uint8)
        path = os.path.join(OUT, "io_imageio_%s.png" % tag)
        imageio.imwrite(path, img)
        assert np.array_equal(np.asarray(imageio.imread(path)), img)
        out["img_" + tag] = img

    # ---- PNGs encoded by this package -> imageio's decoder must return the array; the bytes are pinned
    import builtins
    real_import = builtins.__import__

    def no_imageio(name, *a, **k):                            # png.imwrite prefers imageio when importable: force its own
        if name == "imageio":
            raise ImportError("masked for the fixture")
        return real_import(name, *a, **k)
    for tag, shape in (("rgb", (33, 47, 3)), ("grey", (21, 30))):
        img = rng.randint(0, 256, size=shape).astype(np.uint8)
        path = os.path.join(OUT, "io_ours_%s.png" % tag)
        builtins.__import__ = no_imageio
        try:
            png.imwrite(path, img)
        finally:
            builtins.__import__ = real_import
        assert np.array_equal(np.asarray(imageio.imread(path)), img), "imageio cannot read our PNG back"
        out["ours_" + tag] = img

    # ---- 8-connected labelling with statistics: scikit-image
    masks, labels, stats, counts = [], [], [], []
    for i in range(24):
        h, w = (48, 64) if i % 3 else (37, 29)
        m = blob_mask(rng, h, w)
        if i == 5:
            m[:] = False
        if i == 7:
            m[:] = True
        lab = measure.label(m, connectivity=2)                # 8-connectivity, raster order of the first pixel
        props = measure.regionprops(lab)
        st = np.zeros((lab.max() + 1, 5), np.int64)           # cv2 layout: left, top, width, height, area; row 0 = background
        ys, xs = np.nonzero(~m)
        if len(ys):
            st[0] = [xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1, len(ys)]
        for p in props:
            r0, c0, r1, c1 = p.bbox
            st[p.label] = [c0, r0, c1 - c0, r1 - r0, p.area]
        pad = np.zeros((48, 64), np.uint8)
        pad[:h, :w] = m
        padl = np.zeros((48, 64), np.int32)
        padl[:h, :w] = lab
        pst = np.full((40, 5), -1, np.int64)
        pst[:st.shape[0]] = st
        masks.append(pad); labels.append(padl); stats.append(pst); counts.append([h, w, st.shape[0]])
    out.update(cc_masks=np.stack(masks), cc_labels=np.stack(labels), cc_stats=np.stack(stats), cc_meta=np.array(counts))
    np.savez_compressed(os.path.join(OUT, "g12_io.npz"), **out)
    print("wrote g12_io.npz and", sorted(f for f in os.listdir(OUT) if f.startswith("io_")))


if __name__ == "__main__":
    main()
