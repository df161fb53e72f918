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


def _resolve(json_path: str, file_path: str, suffix: str) -> str:
    # dataset.py:22,35: directory of the json + file_path from its first '.' onward (drops a leading '.')
    return json_path[:json_path.rfind('/')] + file_path[file_path.find('.') + 1:] + suffix


def _read_rgb(filename: str) -> torch.Tensor:
    from PIL import Image
    assert os.path.exists(filename), "The file {} does not exist".format(filename)
    with Image.open(filename) as im:
        a = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return torch.from_numpy(a).permute(2, 0, 1).float().div(255.0)


class NeRF_DATA(Dataset):
    def __init__(self, json_path, transforms=None):
        super().__init__()
        assert os.path.exists(json_path), "The path {} does not exist".format(json_path)
        self.path = json_path
        with open(json_path, "r") as f:
            self.data = json.load(f)
        self.camera_angle_x = torch.tensor(self.data["camera_angle_x"])
        self.dataset = self.data["frames"]
        self.image_transforms = transforms
        first = _read_rgb(_resolve(self.path, self.dataset[0]["file_path"], ".png"))
        self.H, self.W = int(first.shape[1]), int(first.shape[2])
        focal = self.W / (2 * torch.tan(self.camera_angle_x / 2))
        self.focal1 = focal
        self.focal2 = focal
        self.cx = self.W / 2
        self.cy = self.H / 2

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, idx):
        fr = self.dataset[idx]
        image = _read_rgb(_resolve(self.path, fr["file_path"], ".png"))
        if self.image_transforms:
            image = self.image_transforms(image)
        return image, torch.Tensor(fr["transform_matrix"]), fr.get("rotation", 0.0)


class NeRF_DATA_NEW(Dataset):
    def __init__(self, json_path, transforms=None):
        super().__init__()
        assert os.path.exists(json_path), "The path {} does not exist".format(json_path)
        self.path = json_path
        with open(json_path, "r") as f:
            self.data = json.load(f)
        self.camera_angle_x = torch.tensor(self.data["camera_angle_x"])
        self.dataset = self.data["frames"]
        self.image_transforms = transforms
        self.H, self.W = self.data["h"], self.data["w"]
        self.focal1 = self.data["fl_x"]
        self.focal2 = self.data["fl_y"]
        self.cx = self.data["cx"]
        self.cy = self.data["cy"]

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, idx):
        fr = self.dataset[idx]
        image = _read_rgb(_resolve(self.path, fr["file_path"], ""))
        if self.image_transforms:
            image = self.image_transforms(image)
        return image, torch.Tensor(fr["transform_matrix"]), fr.get("sharpness", 0.0)


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
