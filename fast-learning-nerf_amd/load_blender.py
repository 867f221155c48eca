"""Blender (NeRF-synthetic) dataset reader: the reference's load_blender.py (:29-89) behind the same names.
Host-side I/O only (json + PNG via PIL; the reference reads through imageio).  `half_res` is the reference's
cv2.INTER_AREA resize at exactly 2x, i.e. the mean of every 2x2 block (OpenCV is not available here: parity unpinned
for that branch, exact for even image sizes by OpenCV's documented semantics)."""
import json
import os

import numpy as np
import torch


def _rot_x(phi):
    c, s = np.cos(phi), np.sin(phi)
    return torch.tensor([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], dtype=torch.float64).float()


def _rot_y(th):
    c, s = np.cos(th), np.sin(th)
    return torch.tensor([[c, 0, -s, 0], [0, 1, 0, 0], [s, 0, c, 0], [0, 0, 0, 1]], dtype=torch.float64).float()


def pose_spherical(theta, phi, radius):
    """Camera-to-world of a camera on a sphere (load_blender.py:29-34): translate along z, tilt by phi, pan by theta
    (degrees), then the fixed axis swap into Blender's convention.  fp32 matrix products like the reference."""
    c2w = torch.eye(4)
    c2w[2, 3] = radius
    c2w = _rot_x(phi / 180. * np.pi) @ c2w
    c2w = _rot_y(theta / 180. * np.pi) @ c2w
    swap = torch.tensor([[-1., 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]])
    return swap @ c2w


def _imread(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im)


def load_blender_data(basedir, half_res=False, testskip=1):
    """-> imgs [n,H,W,4] float32 in [0,1] (RGBA kept), poses [n,4,4] float32, render_poses [40,4,4] (torch),
    [H, W, focal], i_split = [train, val, test] index arrays.  `testskip` thins val/test only (0 = keep all)."""
    imgs_all, poses_all, counts = [], [], [0]
    meta = None
    for split in ('train', 'val', 'test'):
        with open(os.path.join(basedir, 'transforms_{}.json'.format(split)), 'r') as fp:
            meta = json.load(fp)
        skip = 1 if (split == 'train' or testskip == 0) else testskip
        frames = meta['frames'][::skip]
        imgs = np.array([_imread(os.path.join(basedir, f['file_path'] + '.png')) for f in frames])
        imgs_all.append((imgs / 255.).astype(np.float32))
        poses_all.append(np.array([f['transform_matrix'] for f in frames]).astype(np.float32))
        counts.append(counts[-1] + len(frames))
    i_split = [np.arange(counts[i], counts[i + 1]) for i in range(3)]
    imgs = np.concatenate(imgs_all, 0)
    poses = np.concatenate(poses_all, 0)
    H, W = imgs[0].shape[:2]
    focal = .5 * W / np.tan(.5 * float(meta['camera_angle_x']))   # the LAST split's angle, like the reference
    render_poses = torch.stack([pose_spherical(a, -30.0, 4.0) for a in np.linspace(-180, 180, 40 + 1)[:-1]], 0)
    if half_res:
        H, W, focal = H // 2, W // 2, focal / 2.
        x = imgs[:, :2 * H, :2 * W].astype(np.float64)
        imgs = (0.25 * (x[:, 0::2, 0::2] + x[:, 0::2, 1::2] + x[:, 1::2, 0::2] + x[:, 1::2, 1::2]))   # float64 like the reference's np.zeros buffer
    return imgs, poses, render_poses, [H, W, focal], i_split
