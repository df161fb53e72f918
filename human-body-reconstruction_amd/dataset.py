"""Dataset ingest: the two on-disk formats the reference's trainers read, and the all-rays
pre-materialisation of train_hash2.py:74-96 moved onto the GPU.

* `NeRF_DATA`      - Blender synthetic format (reference dataset.py:9-44): `camera_angle_x`, `frames[*].file_path`
                     (+'.png'), `transform_matrix`, `rotation`; focal = W / (2 tan(camera_angle_x / 2)) (:26).
* `NeRF_DATA_NEW`  - colmap2nerf format (reference dataset_new.py:9-44): `fl_x, fl_y, cx, cy, w, h`, `file_path` with
                     extension, `sharpness`.

Images are decoded with PIL (cv2/torchvision are not in this image).  The reference does cv2.imread (3-channel BGR,
alpha ignored) -> BGR2RGB -> ToTensor; `Image.convert("RGB")` likewise drops alpha without compositing and PNG is
lossless, so the tensors are identical: [3,H,W] float32 in [0,1].  (Decode parity is unpinned by a reference run -
cv2 is absent - but there is no arithmetic in it.)
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from .helper import get_od


def _frame_file(json_path: str, file_path: str, suffix: str) -> str:
    """Image path of a frame: the json's directory joined with `file_path` minus everything up to and including its
    first '.' (so './train/r_0' -> '<dir>/train/r_0'), plus `suffix` - the reference's rule (dataset.py:22,35),
    including its requirement that json_path contains a '/'."""
    return json_path[:json_path.rfind('/')] + file_path[file_path.find('.') + 1:] + suffix


def _read_rgb(filename: str) -> torch.Tensor:
    from PIL import Image
    if not os.path.exists(filename):
        raise AssertionError(f"image file missing: {filename}")
    with Image.open(filename) as im:
        a = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return torch.from_numpy(a).permute(2, 0, 1).float().div(255.0)


class _TransformsJson(Dataset):
    """A `transforms*.json` scene: `frames[*]` = {file_path, transform_matrix, <extra>}.  Subclasses say where the
    intrinsics come from, which suffix image files carry and which per-frame extra is returned third.  The public
    attributes (`path, data, camera_angle_x, dataset, image_transforms, H, W, focal1, focal2, cx, cy`) are the ones
    the reference's trainers read (train_hash2.py:58-72)."""
    suffix = ""
    extra_key = None

    def __init__(self, json_path, transforms=None):
        super().__init__()
        if not os.path.exists(json_path):
            raise AssertionError(f"scene description missing: {json_path}")
        self.path, self.image_transforms = json_path, transforms
        with open(json_path) as fh:
            self.data = json.load(fh)
        self.dataset = self.data["frames"]
        self.camera_angle_x = torch.tensor(self.data["camera_angle_x"])
        self.H, self.W, self.focal1, self.focal2, self.cx, self.cy = self._camera()

    def _camera(self):
        raise NotImplementedError

    def frame_path(self, idx: int) -> str:
        return _frame_file(self.path, self.dataset[idx]["file_path"], self.suffix)

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, idx):
        fr = self.dataset[idx]
        image = _read_rgb(self.frame_path(idx))
        if self.image_transforms:
            image = self.image_transforms(image)
        return image, torch.Tensor(fr["transform_matrix"]), fr.get(self.extra_key, 0.0)


class NeRF_DATA(_TransformsJson):
    """Blender synthetic format (reference dataset.py:9-44): size from the first PNG, one focal length from
    `camera_angle_x` (:26), principal point at the image centre; third item = `rotation`."""
    suffix = ".png"
    extra_key = "rotation"

    def _camera(self):
        _, H, W = _read_rgb(self.frame_path(0)).shape
        focal = W / (2 * torch.tan(self.camera_angle_x / 2))
        return int(H), int(W), focal, focal, W / 2, H / 2


class NeRF_DATA_NEW(_TransformsJson):
    """colmap2nerf format (reference dataset_new.py:9-44): `h, w, fl_x, fl_y, cx, cy` from the json, `file_path`
    already carries its extension; third item = `sharpness`."""
    extra_key = "sharpness"

    def _camera(self):
        d = self.data
        return d["h"], d["w"], d["fl_x"], d["fl_y"], d["cx"], d["cy"]


def intrinsics(ds) -> torch.Tensor:
    """K as train_hash2.py:67-72 builds it.  (The reference fills an INTEGER identity matrix, truncating the focal
    length; that is an artefact of np.array([[1,0,0],...]) being int64 - kept here for drop-in parity.)"""
    K = torch.from_numpy(np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1]]))
    K[0, 0] = ds.focal1
    K[1, 1] = ds.focal2
    K[0, 2] = ds.cx
    K[1, 2] = ds.cy
    return K


@torch.no_grad()
def materialise_rays(ds, K: torch.Tensor, device, images_per_batch: int = 50):
    """All training rays at once (train_hash2.py:74-96), but built and kept on `device`: returns
    (rays_o [M,3], rays_d [M,3], dir_norms [M,1], gts [M,3]) with M = len(ds)*H*W.  64 M lego rays are 2.6 GB -
    trivial next to 288 GB of HBM - and remove the reference's 2.3 GB host staging + per-step H2D copies."""
    H, W = int(ds.H), int(ds.W)
    Kd = K.to(device)
    O, D, Nn, G = [], [], [], []
    for i0 in range(0, len(ds), images_per_batch):
        items = [ds[i] for i in range(i0, min(len(ds), i0 + images_per_batch))]
        image = torch.stack([it[0] for it in items]).to(device)
        c2w = torch.stack([it[1] for it in items]).to(device)
        o, d, n = get_od(H, W, Kd, c2w)
        O.append(o.reshape(-1, 3)); D.append(d.reshape(-1, 3)); Nn.append(n.reshape(-1, 1))
        G.append(image.permute(0, 2, 3, 1).reshape(-1, 3))
    return torch.cat(O), torch.cat(D), torch.cat(Nn), torch.cat(G)
