"""NeRF++ scene-directory reader behind the reference's names (nerf++-ours/data_loader_split.py:27-106 +
the per-view sampler of nerf_sample_ray_split.py:36-173):

    <basedir>/<scene>/<split>/{rgb, intrinsics, pose[, mask, min_depth]}/<name>.{png|jpg|txt}   (+ max_depth.txt)

`load_data_split(basedir, scene, split, skip, try_load_min_depth, only_img_files)` returns one `RaySamplerSingleImage`
per view.  Camera convention: OpenCV / COLMAP (x right, y down, z into the scene), 4x4 intrinsics and
camera-to-world matrices stored as 16 whitespace-separated numbers; pixel centres at +0.5.  Like the reference's
driver this build loads every view at HALF resolution (`resolution_level=2`, data_loader_split.py:102).

Differences by construction: images are read with PIL; rays are generated on the GPU by `fastnerf_pp_gen_rays`
(get_rays_single_image, nerf_sample_ray_split.py:10-34) when first asked for, instead of being held as host arrays for
every view.  Parity note: the reference shrinks images with cv2.resize (INTER_AREA / INTER_NEAREST / INTER_LINEAR);
OpenCV is neither vendored nor pinned there and is absent here.  For the integer shrink factors this reader is used
with, INTER_AREA is the mean over each factor x factor block and INTER_NEAREST picks the block's top-left pixel --
restated from OpenCV's documented behaviour and marked "parity unpinned" (the goldens were recorded with the same
restatement standing in for cv2).  Golden: tests/golden/g17_pp_loader.npz (oracle/make_golden_pp_loader.py)."""
import glob
import os
from collections import OrderedDict

import numpy as np
import torch


def find_files(folder, exts):
    """Sorted files of `folder` matching any of the glob patterns `exts` ([] when the folder does not exist)."""
    if not os.path.isdir(folder):
        return []
    return sorted(f for pattern in exts for f in glob.glob(os.path.join(folder, pattern)))


def read_matrix_txt(path):
    """A 4x4 float32 matrix stored as 16 numbers."""
    with open(path) as f:
        values = [float(tok) for tok in f.read().split()]
    if len(values) != 16:
        raise ValueError('{}: expected 16 numbers, found {}'.format(path, len(values)))
    return np.asarray(values, dtype=np.float32).reshape(4, 4)


def _read_image(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im).astype(np.float32) / 255.


def shrink_area(img, factor):
    """INTER_AREA for an integer factor: block means."""
    if factor == 1:
        return img
    h, w = img.shape[0] // factor, img.shape[1] // factor
    blocks = img[:h * factor, :w * factor].reshape((h, factor, w, factor) + img.shape[2:])
    return blocks.mean(axis=(1, 3), dtype=np.float32)


def shrink_linear(img, factor):
    """cv2.resize(..., INTER_LINEAR) to 1/factor of the size for an integer factor (nerf_sample_ray_split.py:82): destination
    pixel x samples the source at factor * (x + 0.5) - 0.5 (half-pixel centres) -- for an even factor the midpoint of the two
    central source pixels of its block in each direction (their 2 x 2 mean), for an odd one the block's central pixel.
    OpenCV is not vendored or pinned by the reference: restated from its documented sampling rule, parity unpinned."""
    if factor == 1:
        return img
    if img.shape[0] % factor or img.shape[1] % factor:
        # the reference resizes to (W // lvl, H // lvl) with a NON-integer scale src / dst here (general bilinear sampling at
        # (x + 0.5) * src / dst - 0.5); the block formulas below would silently crop instead and return other depths
        raise NotImplementedError('shrink_linear: image size {}x{} is not a multiple of the resolution level {} (non-integer '
                                  'INTER_LINEAR scales are not restated)'.format(img.shape[0], img.shape[1], factor))
    h, w = img.shape[0] // factor, img.shape[1] // factor
    c = (factor - 1) // 2
    if factor % 2:
        return np.ascontiguousarray(img[c::factor, c::factor][:h, :w])
    a = img[c::factor][:h].astype(np.float32)
    b = img[c + 1::factor][:h].astype(np.float32)
    rows = 0.5 * a + 0.5 * b
    return (0.5 * rows[:, c::factor][:, :w] + 0.5 * rows[:, c + 1::factor][:, :w]).astype(np.float32)


def shrink_nearest(img, factor):
    return img[::factor, ::factor][:img.shape[0] // factor, :img.shape[1] // factor]


class RaySamplerSingleImage:
    """One posed view: image (flattened [H*W,3]), optional mask / per-pixel near depth, and its rays."""

    def __init__(self, H, W, intrinsics, c2w, img_path=None, resolution_level=1, mask_path=None, min_depth_path=None,
                 max_depth=None, device='cuda'):
        self.W_orig, self.H_orig = W, H
        self.intrinsics_orig = intrinsics
        self.c2w_mat = c2w
        self.img_path, self.mask_path, self.min_depth_path, self.max_depth = img_path, mask_path, min_depth_path, max_depth
        self.device = device
        self.resolution_level = -1
        self.set_resolution_level(resolution_level)

    def set_resolution_level(self, resolution_level):
        if resolution_level == self.resolution_level:
            return
        lvl = self.resolution_level = resolution_level
        self.W, self.H = self.W_orig // lvl, self.H_orig // lvl
        self.intrinsics = np.array(self.intrinsics_orig, copy=True)
        self.intrinsics[:2, :3] /= lvl                      # focal lengths and principal point scale with the image
        self.img = self.mask = self.min_depth = None
        if self.img_path is not None:
            self.img = shrink_area(_read_image(self.img_path)[..., :3], lvl).reshape(-1, 3)
        if self.mask_path is not None:
            self.mask = shrink_nearest(_read_image(self.mask_path), lvl).reshape(-1)
        if self.min_depth_path is not None:
            md = _read_image(self.min_depth_path) * self.max_depth + 1e-4
            self.min_depth = np.ascontiguousarray(shrink_linear(md, lvl)).reshape(-1)
        self._rays = None

    def _ray_tensors(self):
        if self._rays is None:
            from . import ops
            ro, rd = ops.pp_gen_rays(self.H, self.W, self.intrinsics, self.c2w_mat, device=self.device)
            depth = float(np.linalg.inv(np.asarray(self.c2w_mat, dtype=np.float64))[2, 3])
            self._rays = (ro, rd, torch.full((self.H * self.W,), depth, device=ro.device, dtype=torch.float32))
        return self._rays

    @property
    def rays_o(self):
        return self._ray_tensors()[0]

    @property
    def rays_d(self):
        return self._ray_tensors()[1]

    @property
    def depth(self):
        return self._ray_tensors()[2]

    def get_img(self):
        return None if self.img is None else self.img.reshape(self.H, self.W, 3)

    def _near(self, index=None):
        if self.min_depth is not None:
            md = torch.from_numpy(self.min_depth if index is None else self.min_depth[index])
        else:
            md = torch.full((self.H * self.W if index is None else len(index),), 1e-4, dtype=torch.float32)
        return md

    def get_all(self):
        """Every ray of the view: OrderedDict(ray_o, ray_d, depth, rgb, mask, min_depth) of tensors (None where absent)."""
        ro, rd, depth = self._ray_tensors()
        return OrderedDict([('ray_o', ro), ('ray_d', rd), ('depth', depth),
                            ('rgb', None if self.img is None else torch.from_numpy(self.img)),
                            ('mask', None if self.mask is None else torch.from_numpy(self.mask)),
                            ('min_depth', self._near())])

    def select_indices(self, N_rand, center_crop=False):
        """Flat pixel indices of one random batch, drawn without replacement with numpy's global generator exactly
        like the reference (nerf_sample_ray_split.py:118-137): from the central half-size window when center_crop."""
        if not center_crop:
            return np.random.choice(self.H * self.W, size=(N_rand,), replace=False)
        hh, hw = self.H // 2, self.W // 2
        qh, qw = hh // 2, hw // 2
        cols, rows = np.meshgrid(np.arange(hw - qw, hw + qw), np.arange(hh - qh, hh + qh))
        cols, rows = cols.reshape(-1), rows.reshape(-1)
        pick = np.random.choice(cols.shape[0], size=(N_rand,), replace=False)
        return rows[pick] * self.W + cols[pick]

    def random_sample(self, N_rand, center_crop=False):
        idx = self.select_indices(N_rand, center_crop)
        ro, rd, depth = self._ray_tensors()
        dev_idx = torch.from_numpy(idx).to(ro.device)
        return OrderedDict([('ray_o', ro[dev_idx]), ('ray_d', rd[dev_idx]), ('depth', depth[dev_idx]),
                            ('rgb', None if self.img is None else torch.from_numpy(self.img[idx])),
                            ('mask', None if self.mask is None else torch.from_numpy(self.mask[idx])),
                            ('min_depth', self._near(idx)), ('img_name', self.img_path)])


def load_data_split(basedir, scene, split, skip=1, try_load_min_depth=True, only_img_files=False, device='cuda'):
    root = os.path.join(basedir.rstrip('/'), scene, split)
    images = find_files(os.path.join(root, 'rgb'), ['*.png', '*.jpg'])
    if only_img_files:
        return images
    intrinsics = find_files(os.path.join(root, 'intrinsics'), ['*.txt'])[::skip]
    poses = find_files(os.path.join(root, 'pose'), ['*.txt'])[::skip]
    n_views = len(poses)

    def per_view(files, what):
        """Every `skip`-th file of an optional per-view folder, or a list of Nones."""
        if not files:
            return [None] * n_views
        files = files[::skip]
        if len(files) != n_views:
            raise AssertionError('{}: {} {} files for {} poses'.format(root, len(files), what, n_views))
        return files
    images = per_view(images, 'rgb')
    masks = per_view(find_files(os.path.join(root, 'mask'), ['*.png', '*.jpg']), 'mask')
    near_maps = per_view(find_files(os.path.join(root, 'min_depth'), ['*.png', '*.jpg']) if try_load_min_depth else [],
                         'min_depth')
    # every split is assumed to have the training images' size
    first_train = find_files(os.path.join(basedir.rstrip('/'), scene, 'train', 'rgb'), ['*.png', '*.jpg'])[0]
    H, W = _read_image(first_train).shape[:2]
    try:
        with open(os.path.join(root, 'max_depth.txt')) as f:
            max_depth = float(f.readline().strip())
    except (OSError, ValueError):
        max_depth = None
    return [RaySamplerSingleImage(H=H, W=W, intrinsics=read_matrix_txt(intrinsics[i]), c2w=read_matrix_txt(poses[i]),
                                  img_path=images[i], mask_path=masks[i], min_depth_path=near_maps[i], max_depth=max_depth,
                                  resolution_level=2, device=device) for i in range(n_views)]
