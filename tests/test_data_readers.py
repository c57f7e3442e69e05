"""f-4: on-disk readers of the render path's inputs -- the PNG decoder (all five scan-line filters) and
load_LINEMOD_data (LL:97-163) on a synthetic dataset written in the reference's format; when the reference checkout
is present the reader is pinned against the reference's own function on the same files."""
import json
import os
import struct
import subprocess
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_png_with_filters(path, img, filters):
    """Encode uint8 [H,W,C] with scan-line filter filters[y % len] (0 None, 1 Sub, 2 Up, 3 Average, 4 Paeth)."""
    h, w, ch = img.shape
    flat = img.reshape(h, w * ch).astype(np.int32)
    raw = bytearray()
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):
        ft = filters[y % len(filters)]
        cur = flat[y]
        a = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
        c = np.concatenate([np.zeros(ch, np.int32), prev[:-ch]])
        if ft == 0:
            pred = np.zeros_like(cur)
        elif ft == 1:
            pred = a
        elif ft == 2:
            pred = prev
        elif ft == 3:
            pred = (a + prev) >> 1
        else:
            p = a + prev - c
            pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        raw.append(ft)
        raw += bytes(((cur - pred) & 255).astype(np.uint8))
        prev = cur
    chunk = lambda tag, data: struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)))
        z = zlib.compress(bytes(raw), 6)
        f.write(chunk(b"IDAT", z[:len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b""))


@pytest.mark.parametrize("ch", [1, 2, 3, 4])
def test_png_decoder_undoes_every_filter(tmp_path, ch):
    sys.path.insert(0, ROOT)
    from neural_sim_nerf_amd import png
    rng = np.random.RandomState(ch)
    img = rng.randint(0, 256, size=(23, 31, ch)).astype(np.uint8)
    p = str(tmp_path / "f.png")
    _write_png_with_filters(p, img, [0, 1, 2, 3, 4, 4, 3, 1])
    got = png.imread(p)
    assert np.array_equal(got, img if ch > 1 else img[..., 0])
    assert png.imsize(p) == (31, 23)
    if ch != 2:                                                    # and what the package itself writes (no grey+alpha writer)
        png.imwrite(p, img if ch > 1 else img[..., 0])
        assert np.array_equal(png.imread(p), img if ch > 1 else img[..., 0])
    with pytest.raises(ValueError):
        open(p, "wb").write(b"not a png")
        png.imread(p)


def _make_dataset(root, n=(5, 4, 6), hw=(12, 10)):
    rng = np.random.RandomState(0)
    K = [[133.3, 0.0, 5.2], [0.0, 134.1, 6.1], [0.0, 0.0, 1.0]]
    for s, cnt in zip(("train", "val", "test"), n):
        frames = []
        os.makedirs(os.path.join(root, s), exist_ok=True)
        for i in range(cnt):
            img = rng.randint(0, 256, size=(hw[0], hw[1], 4)).astype(np.uint8)
            path = os.path.join(root, s, "r_%d.png" % i)
            _write_png_with_filters(path, img, [1, 4, 2])
            pose = np.eye(4)
            pose[:3, :4] = rng.standard_normal((3, 4))
            frames.append({"file_path": path, "transform_matrix": pose.tolist(),
                           "intrinsic_matrix": [[v + (0.5 if s == "test" else 0.0) for v in row] for row in K]})
        json.dump({"near": 0.8 + 0.01 * len(s), "far": 1.4 + 0.02 * len(s), "frames": frames},
                  open(os.path.join(root, "transforms_%s.json" % s), "w"))


def test_load_linemod_data_semantics(tmp_path):
    sys.path.insert(0, ROOT)
    from neural_sim_nerf_amd import data
    _make_dataset(str(tmp_path))
    imgs, poses, render_poses, hwf, K, i_split, near, far = data.load_LINEMOD_data(str(tmp_path), half_res=False, testskip=2)
    assert imgs.shape == (5 + 2 + 3, 12, 10, 4) and imgs.dtype == np.float32 and imgs.max() <= 1.0
    assert poses.shape == (10, 4, 4) and poses.dtype == np.float32
    assert [len(x) for x in i_split] == [5, 2, 3] and i_split[2][0] == 7
    assert tuple(render_poses.shape) == (40, 4, 4)
    assert hwf[:2] == [12, 10] and hwf[2] == K[0][0] == 133.3 + 0.5        # focal/K from the LAST split's first frame
    assert near == pytest.approx(min(0.85, 0.84) - 1) and far == pytest.approx(max(1.5, 1.48) + 1)
    # testskip = 0 means "no skipping" (LL:111)
    assert data.load_LINEMOD_data(str(tmp_path), testskip=0, load_images=False)[1].shape[0] == 15
    # poses / intrinsics only
    none, poses2, _, hwf2, K2, _, _, _ = data.load_LINEMOD_data(str(tmp_path), testskip=2, load_images=False)
    assert none is None and np.array_equal(poses2, poses) and hwf2 == hwf and K2 == K
    # half_res: factor 2 on H, W, focal and the first two rows of K; 2x2 block mean of the images
    h_imgs, _, _, h_hwf, h_K, _, _, _ = data.load_LINEMOD_data(str(tmp_path), half_res=True, testskip=2)
    assert h_hwf == [6, 5, hwf[2] / 2] and h_K[0] == [v / 2 for v in K[0]] and h_K[2] == K[2]
    assert h_imgs.shape == (10, 6, 5, 4)
    assert np.allclose(h_imgs[3, 2, 1], imgs[3, 4:6, 2:4].reshape(4, 4).mean(0), atol=1e-6)


@pytest.mark.skipif(not os.path.isdir("/root/reference/optimization"), reason="needs the reference checkout")
def test_load_linemod_data_matches_reference(tmp_path):
    """Pinned against LL:97-163 itself on the same files.  Two shims, both outside the arithmetic under test: imageio
    (absent here) is replaced by this package's PNG decoder, and `pose_spherical` by the reference's own
    `pose_spherical_nograd` -- the published call `pose_spherical(angle, -30.0, 1.01)` raises on a python float
    (LL:63), so render_poses cannot be produced by the reference as written."""
    _make_dataset(str(tmp_path))
    code = r'''
import sys, types, json, numpy as np
sys.path.insert(0, %r)
from neural_sim_nerf_amd import png, data
for m in ("imageio", "cv2"):
    sys.modules[m] = types.ModuleType(m)
sys.modules["imageio"].imread = png.imread
sys.path.insert(0, "/root/reference/optimization")
import utils.load_LINEMOD_noscale as LL
try:
    LL.load_LINEMOD_data(%r, False, 2)
    raised = False
except AttributeError:
    raised = True
LL.pose_spherical = LL.pose_spherical_nograd
ref = LL.load_LINEMOD_data(%r, False, 2)
got = data.load_LINEMOD_data(%r, False, 2)
ok = raised
ok &= np.array_equal(ref[0], got[0]) and ref[0].dtype == got[0].dtype
ok &= np.array_equal(ref[1], got[1]) and np.array_equal(ref[2].numpy(), got[2].numpy())
ok &= list(ref[3]) == list(got[3]) and ref[4] == got[4]
ok &= all(np.array_equal(a, b) for a, b in zip(ref[5], got[5])) and ref[6] == got[6] and ref[7] == got[7]
print("ok" if ok else "mismatch", raised)
''' % (ROOT, str(tmp_path), str(tmp_path), str(tmp_path))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1] == "ok True", (out.stdout, out.stderr[-2000:])


def test_imwrite_many_equals_imwrite(tmp_path):
    sys.path.insert(0, ROOT)
    from neural_sim_nerf_amd import png
    rng = np.random.RandomState(2)
    imgs = [rng.randint(0, 256, size=(20, 30, 3)).astype(np.uint8) for _ in range(9)]
    png.imwrite_many([str(tmp_path / ("m%d.png" % i)) for i in range(9)], imgs, threads=4)
    for i, im in enumerate(imgs):
        png.imwrite(str(tmp_path / ("s%d.png" % i)), im)
        assert open(tmp_path / ("m%d.png" % i), "rb").read() == open(tmp_path / ("s%d.png" % i), "rb").read()
        assert np.array_equal(png.imread(str(tmp_path / ("m%d.png" % i))), im)
    with pytest.raises(ValueError):
        png.imwrite_many(["a.png"], [])


# ------------------------------------------------------------------------------------------------------
# golden vectors from imageio 2.9.0 and scikit-image 0.18.3 (oracle/gen_io_golden.py, run with /opt/conda/bin/python3.9)
# ------------------------------------------------------------------------------------------------------
GOLD = os.path.join(ROOT, "tests", "golden")


def test_png_io_against_imageio_golden(tmp_path, monkeypatch):
    """The reference writes views with imageio.imwrite (RN:250) and reads datasets with imageio.imread (LL:120).
    PNGs ENCODED BY imageio must decode to the arrays it was given; PNGs encoded by this package -- which imageio
    decoded back correctly when the fixture was generated -- must still come out byte for byte."""
    sys.path.insert(0, ROOT)
    import builtins
    from neural_sim_nerf_amd import png
    g = np.load(os.path.join(GOLD, "g12_io.npz"))
    assert str(g["imageio_version"]) == "2.9.0"
    for tag in ("rgb", "rgba", "grey"):
        assert np.array_equal(png.imread(os.path.join(GOLD, "io_imageio_%s.png" % tag)), g["img_" + tag]), tag
    real_import = builtins.__import__

    def no_imageio(name, *a, **k):          # exercise the package's own encoder even where imageio is installed
        if name == "imageio":
            raise ImportError("masked")
        return real_import(name, *a, **k)
    monkeypatch.setattr(builtins, "__import__", no_imageio)
    for tag in ("rgb", "grey"):
        p = str(tmp_path / ("o_%s.png" % tag))
        png.imwrite(p, g["ours_" + tag])
        assert open(p, "rb").read() == open(os.path.join(GOLD, "io_ours_%s.png" % tag), "rb").read(), tag


def test_connected_components_against_skimage_golden():
    """oracle/handoff_oracle.connected_components_with_stats (scipy.ndimage underneath) against scikit-image's
    measure.label(connectivity=2) + regionprops on 24 blob masks incl. an empty and a full one: the same labels (raster
    order of each component's first pixel), boxes and areas -- cv2.connectedComponentsWithStats' documented contract
    from a second, independent implementation.  (The device path is then held bit-equal to this oracle by the GPU
    tests.)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import handoff_oracle as HO
    g = np.load(os.path.join(GOLD, "g12_io.npz"))
    assert str(g["skimage_version"]) == "0.18.3"
    for i in range(g["cc_masks"].shape[0]):
        h, w, n = g["cc_meta"][i]
        mask = (g["cc_masks"][i][:h, :w] * 255).astype(np.uint8)
        n_labels, labels, stats = HO.connected_components_with_stats(mask)
        assert n_labels == n, i
        assert np.array_equal(labels, g["cc_labels"][i][:h, :w]), i
        assert np.array_equal(stats, g["cc_stats"][i][:n]), i
