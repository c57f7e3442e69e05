"""On-disk inputs of the render path: the camera/bounds file the reference reads at start-up.

    load_data_param  LL:166-199   <basedir>/nerf_traindata_info.json -> [H, W, focal], K, near, far

`half_res` divides H, W and the first two rows of K by FOUR (LL:185-192, not two), and the bounds are widened by
0.5 on both sides (LL:197-198).  K stays a nested list of python floats, which is what `render`/`get_rays` expect
(RH:160 feeds those scalars into fp32 tensor ops)."""
import json
import os


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
