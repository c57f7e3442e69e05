"""Image hand-off: to8b + get_annotation/find_bbox (RH:14, NM:786-797) on the GPU, bit-exact
against oracle/handoff_oracle.py; CPU tests pin the oracle's own restatement of the OpenCV pieces."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import handoff_oracle as HO          # noqa: E402


def blobs(rng, H, W, n, rmax=9, value=True):
    """A random RGB image: n filled discs/rectangles of random colour on black."""
    img = np.zeros((H, W, 3), np.uint8)
    yy, xx = np.mgrid[:H, :W]
    for _ in range(n):
        cy, cx, r = rng.randint(0, H), rng.randint(0, W), rng.randint(1, rmax + 1)
        col = rng.randint(0, 256, 3).astype(np.uint8)
        if rng.rand() < 0.5:
            sel = (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
        else:
            sel = (abs(yy - cy) <= r) & (abs(xx - cx) <= rng.randint(1, rmax + 1))
        img[sel] = col
    return img


def flood_labels(fg):
    """Brute-force 8-connected labelling in raster order (independent of scipy)."""
    H, W = fg.shape
    lab = np.zeros((H, W), int)
    n = 0
    for y in range(H):
        for x in range(W):
            if fg[y, x] and not lab[y, x]:
                n += 1
                stack = [(y, x)]
                lab[y, x] = n
                while stack:
                    cy, cx = stack.pop()
                    for dy in (-1, 0, 1):
                        for dx in (-1, 0, 1):
                            ny, nx = cy + dy, cx + dx
                            if 0 <= ny < H and 0 <= nx < W and fg[ny, nx] and not lab[ny, nx]:
                                lab[ny, nx] = n
                                stack.append((ny, nx))
    return n, lab


# ---------------------------------------------------------------------------------------------------------
# CPU: the oracle's restatement
# ---------------------------------------------------------------------------------------------------------
def test_oracle_gray_formula():
    assert 9798 + 19235 + 3735 == 1 << 15                       # OpenCV's 15-bit coefficients sum to one
    px = np.array([[[255, 255, 255], [0, 0, 0], [1, 1, 1], [2, 2, 2], [255, 0, 0], [0, 255, 0], [0, 0, 255]]], np.uint8)
    g = HO.gray_as_reference(px)[0]
    # the reference feeds BGR into RGB2GRAY: the *red* channel of the render gets the blue weight 0.114
    assert list(g) == [255, 0, 1, 2, 29, 150, 76]
    assert list(HO.mask_as_reference(px)[0]) == [255, 0, 0, 255, 255, 255, 255]      # gray > 1


def test_mask_is_insensitive_to_the_unpinned_gray_arithmetic():
    """f-3's one unpinned piece is cv2's 8-bit RGB2GRAY arithmetic (opencv-python is an un-pinned requirement and is not
    installable here).  The reference only ever looks at `gray > 1` (cv2.threshold(gray, 1, 255, THRESH_BINARY), NM:795), so
    what matters is on which side of 1.5 a colour's luma falls.  EXHAUSTIVELY over all 2^24 colours: the restated OpenCV 4.x
    fixed-point formula (15-bit coefficients, +2^14, >> 15), OpenCV 3.x's (14-bit, +2^13, >> 14), and the textbook float
    formula 0.299 c0 + 0.587 c1 + 0.114 c2 in float64 and float32 with round-half-even and with round-half-up give grey
    values that differ on tens of thousands of colours -- and the SAME mask bit on every one of them.  So no image exists
    on which any of these cv2 candidates would produce another mask, component set or box than the oracle's."""
    v = np.arange(256, dtype=np.int64)
    c0, c1, c2 = np.meshgrid(v, v, v, indexing="ij")            # c0 = first channel of the array handed to cvtColor (blue: NM:793)
    rgb8 = np.stack([c2, c1, c0], -1).astype(np.uint8)          # ... of a PNG holding (r, g, b) = (c2, c1, c0)
    gray = HO.gray_as_reference(rgb8.reshape(-1, 1, 3)).reshape(256, 256, 256).astype(np.int64)
    assert np.array_equal(gray, (c0 * 9798 + c1 * 19235 + c2 * 3735 + (1 << 14)) >> 15)
    x64 = 0.299 * c0 + 0.587 * c1 + 0.114 * c2
    x32 = (np.float32(0.299) * c0.astype(np.float32) + np.float32(0.587) * c1.astype(np.float32)) + np.float32(0.114) * c2.astype(np.float32)
    variants = {"opencv 3.x fixed point (>> 14)": (c0 * 4899 + c1 * 9617 + c2 * 1868 + (1 << 13)) >> 14,
                "float64, round half even": np.rint(x64).astype(np.int64),
                "float64, round half up": np.floor(x64 + 0.5).astype(np.int64),
                "float32, round half even": np.rint(x32).astype(np.int64),
                "float32, round half up": np.floor(x32 + np.float32(0.5)).astype(np.int64)}
    for name, g in variants.items():
        assert (g != gray).sum() > 1000, name                   # the grey values themselves do differ ...
        assert not ((g > 1) != (gray > 1)).any(), name          # ... the thresholded mask never does


def test_oracle_to8b_truncates():
    x = np.array([-0.5, 0.0, 0.5, 1.0, 1.5, 254.999 / 255, 0.999999, 1 / 255, np.nextafter(np.float32(1 / 255), 0)], np.float32)
    assert list(HO.to8b(x)) == [0, 0, 127, 255, 255, 254, 254, 1, 0]


def test_oracle_components_match_flood_fill():
    rng = np.random.RandomState(0)
    for trial in range(20):
        H, W = rng.randint(4, 24), rng.randint(4, 24)
        fg = rng.rand(H, W) < rng.uniform(0.2, 0.6)
        n, lab = flood_labels(fg)
        n2, lab2, stats = HO.connected_components_with_stats(fg.astype(np.uint8) * 255)
        assert n2 == n + 1 and np.array_equal(lab, lab2)                   # same partition, same raster label order
        for k in range(1, n + 1):
            ys, xs = np.nonzero(lab == k)
            assert list(stats[k]) == [xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1, len(ys)]
        assert stats[0, 4] == (~fg).sum()


def test_oracle_selection_rules():
    img = np.zeros((20, 30, 3), np.uint8)
    img[2:5, 3:9] = 200          # 3x6 block, area 18
    img[10:18, 20:22] = 200      # 8x2 block, area 16
    img[0, 29] = 200             # single pixel
    bbox, rows, mask = HO.get_annotation(img)
    assert rows == 3 and list(bbox) == [3, 2, 6, 3]           # background (largest area) dropped, largest w*h kept
    # an object larger than the background: the *object* is dropped and the background row wins (reference quirk)
    img2 = np.full((10, 10, 3), 200, np.uint8)
    img2[0, :3] = 0
    bbox2, rows2, _ = HO.get_annotation(img2)
    assert rows2 == 1 and list(bbox2) == [0, 0, 3, 1]
    empty = np.zeros((8, 8, 3), np.uint8)
    assert HO.get_annotation(empty)[:2] == (None, 0)


def test_dataset_dicts_records():
    """Host logic of the hand-off: the detectron2-style records built from an annotation dict (no GPU needed)."""
    import torch
    sys.path.insert(0, ROOT)
    from neural_sim_nerf_amd import handoff
    imgs = torch.arange(2 * 4 * 5 * 3, dtype=torch.uint8).reshape(2, 4, 5, 3)
    ann = {"bbox": torch.tensor([[1, 0, 3, 2], [0, 1, 2, 2]], dtype=torch.int32), "count": torch.tensor([1, 3], dtype=torch.int32),
           "mask": torch.zeros((2, 4, 5), dtype=torch.uint8)}
    recs = handoff.dataset_dicts(imgs, ann, category_id=15, first_image_id=10)
    assert [r["image_id"] for r in recs] == [10, 11] and recs[0]["height"] == 4 and recs[0]["width"] == 5
    assert recs[1]["annotations"][0] == {"bbox": [0, 1, 2, 2], "bbox_mode": 1, "category_id": 15,
                                         "segmentation_mask": recs[1]["annotations"][0]["segmentation_mask"]}
    assert np.array_equal(recs[0]["image"], imgs[0].numpy()[..., ::-1])          # BGR, like read_image(format="BGR")
    ann["count"][0] = 0
    with pytest.raises(ValueError):                                                # the reference's np.argmax([]) error
        handoff.dataset_dicts(imgs, ann, category_id=15)


# ---------------------------------------------------------------------------------------------------------
# GPU: the product path against the oracle, bit-exact
# ---------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def util():
    import torch                                                           # noqa: F401
    from neural_sim_nerf_amd.run_nerf_noscale import _util_model
    return _util_model()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 3, 4, 5, 255, 1024, 400 * 400 * 3 + 1])
def test_to8b_bit_exact(util, n):
    rng = np.random.RandomState(n)
    x = rng.uniform(-0.2, 1.2, n).astype(np.float32)
    k = rng.randint(0, 256, n)
    edge = (k / 255.0).astype(np.float32)
    x[::3] = edge[::3]                                                     # exact and just-below quantisation steps
    x[1::7] = np.nextafter(edge[1::7], np.float32(0))
    if n > 4:
        x[4] = np.nan
    got = util.to8b(x).cpu().numpy()
    assert got.dtype == np.uint8 and np.array_equal(got, HO.to8b(x))


@pytest.mark.gpu
def test_to8b_unaligned_view(util):
    import torch
    x = torch.rand(1001, device=util.device)
    got = util.to8b(x[1:]).cpu().numpy()                                    # data pointer 4 B off a 16 B boundary
    assert np.array_equal(got, HO.to8b(x[1:].cpu().numpy()))


def _check_images(util, imgs):
    bbox, count, mask = util.find_bbox(imgs, with_mask=True)
    bbox, count, mask = bbox.cpu().numpy(), count.cpu().numpy(), mask.cpu().numpy()
    for i, im in enumerate(imgs):
        want, rows, m = HO.get_annotation(im)
        assert np.array_equal(mask[i], m), "mask %d" % i
        assert count[i] == rows, "rows %d: %d vs %d" % (i, count[i], rows)
        assert list(bbox[i]) == ([0, 0, 0, 0] if want is None else [int(v) for v in want]), "bbox %d" % i


@pytest.mark.gpu
def test_find_bbox_random_blobs(util):
    rng = np.random.RandomState(1)
    imgs = np.stack([blobs(rng, 100, 100, rng.randint(1, 12)) for _ in range(37)])          # 37 > the batch of 16
    _check_images(util, imgs)


@pytest.mark.gpu
def test_find_bbox_edge_cases(util):
    H, W = 48, 64
    empty = np.zeros((H, W, 3), np.uint8)
    full = np.full((H, W, 3), 255, np.uint8)
    big = full.copy(); big[0, :5] = 0                               # object larger than the background
    diag = empty.copy()
    for k in range(20):
        diag[10 + k, 5 + k] = 255                                   # one 8-connected diagonal line
    anti = empty.copy()
    for k in range(20):
        anti[10 + k, 40 - k] = 255                                  # anti-diagonal (NE/SW links)
    border = empty.copy(); border[:, 0] = 255; border[H - 1, :] = 255; border[0, W - 1] = 90
    dim = empty.copy(); dim[5:9, 5:9] = (1, 1, 1); dim[20:30, 20:30] = (2, 2, 2); dim[35:40, 10:30] = (9, 0, 0)   # gray 1 / 2 / 0
    ties = empty.copy(); ties[2:6, 2:6] = 255; ties[2:6, 20:24] = 255; ties[20:24, 2:6] = 255          # equal areas and boxes
    spiral = empty.copy()
    spiral[4:44, 4] = 255; spiral[43, 4:60] = 255; spiral[8:44, 59] = 255; spiral[8, 10:60] = 255; spiral[8:38, 10] = 255
    checker = empty.copy(); checker[::2, ::2] = 255; checker[1::2, 1::2] = 255                          # one diagonal-linked net
    _check_images(util, np.stack([empty, full, big, diag, anti, border, dim, ties, spiral, checker]))


@pytest.mark.gpu
def test_find_bbox_non_square_and_single(util):
    rng = np.random.RandomState(3)
    _check_images(util, blobs(rng, 33, 257, 6)[None])
    _check_images(util, blobs(rng, 400, 400, 9, rmax=60)[None])


@pytest.mark.gpu
def test_render_path_inmemory_equals_png_round_trip(util, tmp_path):
    """The in-memory hand-off delivers the bytes the PNGs hold and the box the reference derives from them."""
    import torch
    from neural_sim_nerf_amd import handoff, png, synthetic as S
    from neural_sim_nerf_amd.run_nerf_helpers import NeRF
    from neural_sim_nerf_amd.run_nerf_noscale import render_path
    sd_c = S.synth_weights(0)
    sd_f = S.synth_weights(1000, fine_of=sd_c)
    nets = []
    for sd in (sd_c, sd_f):
        m = NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
        m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
        nets.append(m.cuda())
    kw = dict(network_fn=nets[0], network_fine=nets[1], N_samples=64, N_importance=128, perturb=0., use_viewdirs=True,
              white_bkgd=False, raw_noise_std=0., ndc=False, lindisp=False, near=S.YCBV_NEAR, far=S.YCBV_FAR,
              network_query_fn=None)
    poses = torch.as_tensor(S.sweep_poses(3, seed=1))
    side = 48
    K = S.scaled_K(400.0 / side)
    hwf = [side, side, K[0][0]]
    rgbs, _ = render_path(None, poses, hwf, K, 4096, kw, savedir=str(tmp_path), object_id=2)
    images, ann = handoff.render_path_inmemory(poses, hwf, K, kw, with_mask=True)
    images = images.cpu().numpy()
    for i in range(3):
        file_img = png.imread(os.path.join(str(tmp_path), "2", "%03d.png" % i))
        assert np.array_equal(images[i], file_img)
        want, rows, m = HO.get_annotation(file_img)
        assert np.array_equal(ann["mask"][i].cpu().numpy(), m) and int(ann["count"][i]) == rows
        if want is not None:
            assert list(ann["bbox"][i].cpu().numpy()) == [int(v) for v in want]
    if int(ann["count"].min()) > 0:
        recs = handoff.dataset_dicts(ann["images"], ann, category_id=2)
        assert recs[0]["image"].shape == (side, side, 3) and np.array_equal(recs[0]["image"][..., ::-1], images[0])
        assert recs[1]["annotations"][0]["bbox_mode"] == 1 and len(recs) == 3
