"""Image hand-off from the renderer to the detector's loader without the PNG round trip.

The reference writes every view to `<savedir>/<object_id>/{i:03d}.png` (RN:245-250) and the detector side reads
the files back to build its dataset (NM:655-700 createCocoJSONFromSynthetics, NM:799-830 get_ycbv_dicts): per image
`get_annotation` (NM:791-797) thresholds a grey version, runs cv2.connectedComponentsWithStats and keeps one
XYWH box.  Here the float render is quantised (to8b, RH:14), thresholded, labelled and boxed on the GPU
(csrc/nsr_handoff.hip); nothing touches the disk and one device->host copy moves K boxes instead of K images.

    images, ann = render_path_inmemory(render_poses, hwf, K, render_kwargs_test)
    dicts = dataset_dicts(images, ann, category_id=2)        # detectron2 "dataset dict" records with the image inline

`render_path` (run_nerf_noscale.py) is unchanged and still writes the PNGs the unmodified reference reads.
"""
import numpy as np
import torch

from . import run_nerf_noscale as _rn


def annotate(rgb, model=None, with_mask=False):
    """float [K,H,W,3] in [0,1] (or uint8 RGB) -> dict(images uint8 [K,H,W,3] RGB on device, bbox [K,4] int32 XYWH,
    count [K] int32[, mask [K,H,W] uint8]).  count == 0 marks an image the reference would raise on (no component
    survives find_bbox's `[:-1]`, NM:788-789)."""
    if model is None:
        dev = rgb.device if torch.is_tensor(rgb) and rgb.is_cuda else None
        model = _rn._util_model(dev)
    if torch.is_tensor(rgb) and rgb.dtype == torch.uint8 or (isinstance(rgb, np.ndarray) and rgb.dtype == np.uint8):
        img8 = torch.as_tensor(rgb, dtype=torch.uint8, device=model.device)
    else:
        img8 = model.to8b(rgb)
    if img8.dim() == 3:
        img8 = img8[None]
    res = model.find_bbox(img8, with_mask=with_mask)
    out = dict(images=img8, bbox=res[0], count=res[1])
    if with_mask:
        out["mask"] = res[2]
    return out


def render_path_inmemory(render_poses, hwf, K, render_kwargs, render_factor=0, with_mask=False):
    """render_path (RN:213-255) without the files: returns (images uint8 [K,H,W,3] device tensor -- the bytes the
    PNGs would hold -- and the annotation dict of annotate())."""
    H, W, _ = _rn._scaled_hw(hwf, render_factor)
    kw = dict(render_kwargs)
    near, far = kw.pop("near", 0.), kw.pop("far", 1.)
    if kw.pop("ndc", True):
        raise NotImplementedError("render_path_inmemory: ndc=True is not supported")
    if not kw.pop("use_viewdirs", False):
        raise NotImplementedError("render_path_inmemory: use_viewdirs=False is not supported")
    _rn._check_kwargs(kw)
    n_imp = kw.get("N_importance", 0)
    model = _rn._model_for(kw["network_fn"], kw.get("network_fine", None) if n_imp > 0 else None, n_imp, kw)
    poses = torch.as_tensor(render_poses, dtype=torch.float32)
    with torch.no_grad():
        out = model.render_views(poses[:, :3, :4].to(model.device), H, W, K, near, far)
        ann = annotate(out["rgb_map"].reshape(-1, H, W, 3), model=model, with_mask=with_mask)
    return ann["images"], ann


def dataset_dicts(images, ann, category_id, first_image_id=0, file_prefix="inmemory"):
    """The records get_ycbv_dicts (NM:799-830) builds from the PNG directory, with the image inline instead of a
    path: `image` is the uint8 BGR HWC array detectron2's read_image(format="BGR") (detection_utils.py:169-188)
    would return for the file.  bbox_mode 1 = BoxMode.XYWH_ABS.  Raises like the reference (np.argmax of an empty
    array, NM:818) when an image has no component left."""
    bbox = ann["bbox"].cpu().numpy()
    count = ann["count"].cpu().numpy()
    imgs = images.cpu().numpy()
    masks = ann["mask"].cpu().numpy() if "mask" in ann else None
    records = []
    for i in range(imgs.shape[0]):
        if count[i] == 0:
            raise ValueError("attempt to get argmax of an empty sequence (image %d has no foreground component)" % i)
        obj = {"bbox": [int(v) for v in bbox[i]], "bbox_mode": 1, "category_id": int(category_id)}
        if masks is not None:
            obj["segmentation_mask"] = masks[i]
        records.append({"file_name": "%s/%03d.png" % (file_prefix, i), "image_id": first_image_id + i,
                        "height": int(imgs.shape[1]), "width": int(imgs.shape[2]),
                        "image": np.ascontiguousarray(imgs[i][..., ::-1]), "annotations": [obj]})
    return records
