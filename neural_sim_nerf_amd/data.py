"""On-disk inputs of the render path: the camera/bounds file the reference reads at start-up.

    load_data_param    LL:166-199   <basedir>/nerf_traindata_info.json -> [H, W, focal], K, near, far
    load_LINEMOD_data  LL:97-163    <basedir>/transforms_{train,val,test}.json (+ the frames' PNGs) ->
                                    imgs, poses, render_poses, [H, W, focal], K, i_split, near, far

`half_res` divides H, W and the first two rows of K by FOUR (LL:185-192, not two), and the bounds are widened by
0.5 on both sides (LL:197-198).  K stays a nested list of python floats, which is what `render`/`get_rays` expect
(RH:160 feeds those scalars into fp32 tensor ops)."""
import json
import os

import numpy as np


def load_data_param(basedir, half_res=False, testskip=1):
    with open(os.path.join(basedir, "nerf_traindata_info.json"), "r") as fp:
        meta = json.load(fp)
    frame = meta["frames"][0]
    H, W = meta["H"], meta["W"]
    K = [list(row) for row in frame["intrinsic_matrix"]]
    focal = float(K[0][0])
    if half_res:
        scale = 4
        K[0] = [v / scale for v in K[0]]
        K[1] = [v / scale for v in K[1]]
        H, W, focal = H // scale, W // scale, focal / scale
    return [H, W, focal], K, meta["near"] - 0.5, meta["far"] + 0.5


def load_LINEMOD_data(basedir, half_res=False, testskip=1, load_images=True):
    """LL:97-163: the BlenderProc/LINEMOD-style dataset the NeRFs were trained on (the reference's train() reads it,
    RN:510; the bilevel loop only needs load_data_param).

    Per split (train, val, test; val/test sub-sampled by `testskip` unless it is 0, LL:111-114) the frames'
    `transform_matrix` [4,4] -> poses (float32) and `file_path` -> RGBA images / 255 (float32, all 4 channels kept,
    LL:122).  H, W come from the first image, focal and K from the LAST split's first frame (`meta` is the loop
    variable left over at LL:134-135), near/far = min/max over train and test widened by ONE (LL:162-163, not the 0.5
    of load_data_param).  half_res halves H, W, focal and the first two rows of K (factor 2 here, 4 in
    load_data_param) and area-averages the images 2x2 (cv2.INTER_AREA with an integer factor is the block mean).
    render_poses: 40 poses at elevation -30 on the radius-1.01 circle (LL:138); the reference builds them with the
    autograd variant `pose_spherical`, which raises on python floats (LL:63 `phi.requires_grad_()`), so its train path
    cannot run as published -- the no-grad construction (LL:89-94) is used here.

    load_images=False skips decoding (poses / intrinsics only): imgs is then None and H, W are taken from the JSON
    ('h'/'w' or 'H'/'W' keys) or from the first PNG's header."""
    import torch
    from . import png
    from .pose import pose_spherical_nograd
    splits = ["train", "val", "test"]
    metas = {}
    for s in splits:
        with open(os.path.join(basedir, "transforms_{}.json".format(s)), "r") as fp:
            metas[s] = json.load(fp)
    all_imgs, all_poses, counts = [], [], [0]
    first_file = None
    for s in splits:
        meta = metas[s]
        skip = 1 if (s == "train" or testskip == 0) else testskip
        imgs, poses = [], []
        for frame in meta["frames"][::skip]:
            fname = frame["file_path"]
            if not os.path.isabs(fname) and not os.path.exists(fname):
                fname = os.path.join(basedir, fname)               # the reference relies on the cwd; accept both
            first_file = first_file or fname
            if load_images:
                imgs.append(png.imread(fname))
            poses.append(np.array(frame["transform_matrix"]))
        poses = np.array(poses).astype(np.float32).reshape(-1, 4, 4)
        counts.append(counts[-1] + poses.shape[0])
        if load_images:
            all_imgs.append((np.array(imgs) / 255.).astype(np.float32))
        all_poses.append(poses)
    i_split = [np.arange(counts[i], counts[i + 1]) for i in range(3)]
    poses = np.concatenate(all_poses, 0)
    imgs = np.concatenate(all_imgs, 0) if load_images else None
    if load_images:
        H, W = imgs[0].shape[:2]
    else:
        m0 = metas["train"]
        H, W = (m0.get("H", m0.get("h")), m0.get("W", m0.get("w")))
        if H is None or W is None:
            W, H = png.imsize(first_file)
    K = [list(row) for row in meta["frames"][0]["intrinsic_matrix"]]       # `meta` = the last split, as in LL:134-135
    focal = float(K[0][0])
    render_poses = torch.stack([pose_spherical_nograd(angle, -30.0, 1.01) for angle in np.linspace(-180, 180, 40 + 1)[:-1]], 0)
    if half_res:
        K[0] = [v / 2 for v in K[0]]
        K[1] = [v / 2 for v in K[1]]
        H, W, focal = H // 2, W // 2, focal / 2
        if load_images:
            n, c = imgs.shape[0], imgs.shape[-1]
            blocks = imgs[:, :2 * H, :2 * W].reshape(n, H, 2, W, 2, c)
            imgs = ((blocks[:, :, 0, :, 0] + blocks[:, :, 0, :, 1]) + (blocks[:, :, 1, :, 0] + blocks[:, :, 1, :, 1])) * np.float32(0.25)
            imgs = imgs.astype(np.float64)                          # LL:149 allocates the half-res stack with np.zeros (float64)
    near = min(metas["train"]["near"], metas["test"]["near"]) - 1
    far = max(metas["train"]["far"], metas["test"]["far"]) + 1
    return imgs, poses, render_poses, [H, W, focal], K, i_split, near, far
