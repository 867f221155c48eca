"""LLFF (forward-facing / 360-degree) capture reader behind the reference's entry point
`load_llff_data(basedir, factor, recenter, bd_factor, spherify, path_zflat)` (nerf-ours/load_llff.py:244-316),
returning the same five things: images [n,H,W,3], poses [n,3,5] (c2w | (H,W,focal) column), bds [n,2],
render_poses [m,3,5] (all float32) and the index of the hold-out view.

Written from the camera geometry, batched over views (no per-view Python loops):

  * `poses_bounds.npy` rows are 15 + 2 numbers: a 3x5 matrix whose first four columns are the camera-to-world
    transform in LLFF's (down, right, back) axis order and whose last column is (H, W, focal), then near / far depth.
    NeRF's axis order is (right, up, back): new x = old y, new y = -old x.
  * a "frame" is an orthonormal basis built from a viewing axis and an up hint (`frames_from`);
  * the capture's mean frame (`mean_frame`) recentres the poses, anchors the spiral render path and picks the
    hold-out view (closest camera to the mean position);
  * 360-degree captures are normalised to the unit sphere around the least-squares focus point of the optical
    axes and get a circular render path (`normalise_to_sphere`).

Host-side numpy only.  Images are read with PIL; down-scaled folders (`images_<factor>`) must already exist --
the reference shells out to ImageMagick's `mogrify` to create them (load_llff.py:7-57), which is outside this build.
Golden vectors recorded from the reference: tests/golden/g15_loaders.npz (oracle/make_golden_loaders.py)."""
import os

import numpy as np

_IMG_EXT = ('JPG', 'jpg', 'png')


def _list_images(folder):
    return [os.path.join(folder, f) for f in sorted(os.listdir(folder)) if f.endswith(_IMG_EXT)]


def _read_rgb(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im)


def read_capture(basedir, factor=None, width=None, height=None):
    """-> c2w_llff [n,3,4] (LLFF axis order), hwf [3] of the images actually loaded, bounds [n,2], images [n,H,W,3] in [0,1].
    `factor` (or a target width / height) selects the `images_<suffix>` folder and rescales the focal length."""
    table = np.load(os.path.join(basedir, 'poses_bounds.npy'))
    mats = table[:, :15].reshape(-1, 3, 5)
    bounds = table[:, 15:17].copy()
    full_h, full_w = _read_rgb(_list_images(os.path.join(basedir, 'images'))[0]).shape[:2]
    if factor is not None:
        suffix, scale = '_{}'.format(factor), float(factor)
    elif height is not None:
        scale = full_h / float(height)
        suffix = '_{}x{}'.format(int(full_w / scale), height)
    elif width is not None:
        scale = full_w / float(width)
        suffix = '_{}x{}'.format(width, int(full_h / scale))
    else:
        suffix, scale = '', 1.0
    folder = os.path.join(basedir, 'images' + suffix)
    if not os.path.isdir(folder):
        raise FileNotFoundError(folder + ' does not exist (create the down-scaled copies first; the reference runs '
                                'ImageMagick mogrify for this, load_llff.py:7-57)')
    files = _list_images(folder)
    if len(files) != mats.shape[0]:
        raise ValueError('Mismatch between imgs {} and poses {}'.format(len(files), mats.shape[0]))
    pixels = np.stack([_read_rgb(f)[..., :3] for f in files], 0) / 255.
    hwf = np.array([pixels.shape[1], pixels.shape[2], mats[0, 2, 4] / scale], dtype=np.float64)
    return mats[:, :, :4].copy(), hwf, bounds, pixels


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def frames_from(axis, up_hint, origin):
    """Right-handed camera frames [..,3,4] = (right | up | back | origin) whose back axis is `axis` (normalised) and whose up
    vector is the component of `up_hint` orthogonal to it."""
    back = _unit(np.asarray(axis, dtype=np.float64))
    right = _unit(np.cross(np.broadcast_to(up_hint, back.shape), back))
    up = _unit(np.cross(back, right))
    return np.stack([right, up, back, np.broadcast_to(origin, back.shape)], -1)


def mean_frame(c2w):
    """One frame for the whole capture: mean camera position, summed back axes, summed up axes."""
    return frames_from(_unit(c2w[:, :, 2].sum(0)), c2w[:, :, 1].sum(0), c2w[:, :, 3].mean(0))


def _homogeneous(m34):
    m = np.zeros(m34.shape[:-2] + (4, 4), dtype=m34.dtype)
    m[..., :3, :] = m34
    m[..., 3, 3] = 1.0
    return m


def in_frame(c2w, frame):
    """The poses expressed in `frame`'s coordinates: frame^-1 @ c2w."""
    return (np.linalg.inv(_homogeneous(frame)) @ _homogeneous(c2w))[..., :3, :]


def spiral_path(frame, up, radii, focus_depth, z_rate, turns, n_views):
    """Cameras on an elliptical spiral around `frame`'s origin (extent `radii` per axis, z oscillating at `z_rate` times the
    angular rate), every one looking at the point `focus_depth` in front of the frame."""
    theta = np.linspace(0., 2. * np.pi * turns, n_views + 1)[:-1]
    local = np.stack([np.cos(theta), -np.sin(theta), -np.sin(theta * z_rate), np.ones_like(theta)], -1) * np.append(radii, 1.)
    position = local @ frame.T                                    # [m,4] @ [4,3]
    target = frame @ np.array([0., 0., -focus_depth, 1.])
    return frames_from(position - target, up, position)


def normalise_to_sphere(c2w, bounds, n_views=120):
    """360-degree captures: move the origin to the point closest (least squares) to all optical axes, turn the mean
    camera offset into +z, scale the RMS camera distance to 1 and put `n_views` cameras on the circle at the cameras'
    mean height, looking inwards.  -> (poses, ring, scaled bounds)."""
    axis, origin = c2w[:, :, 2], c2w[:, :, 3]
    # distance^2 of x to the line (o, d) is |P (x - o)|^2 with P = I - d d^T; minimise the mean over the cameras
    proj = np.eye(3) - axis[:, :, None] * axis[:, None, :]
    focus = np.linalg.solve((proj.transpose(0, 2, 1) @ proj).mean(0), (proj @ origin[:, :, None]).mean(0))[:, 0]
    zdir = _unit((origin - focus).mean(0))
    xdir = _unit(np.cross([.1, .2, .3], zdir))
    ydir = _unit(np.cross(zdir, xdir))
    local = in_frame(c2w, np.stack([xdir, ydir, zdir, focus], 1))
    shrink = 1. / np.sqrt(np.mean(np.sum(np.square(local[:, :, 3]), -1)))
    local[:, :, 3] *= shrink
    height = local[:, 2, 3].mean()
    ring_radius = np.sqrt(1. - height ** 2)
    angle = np.linspace(0., 2. * np.pi, n_views)
    spot = np.stack([ring_radius * np.cos(angle), ring_radius * np.sin(angle), np.full_like(angle, height)], -1)
    back = _unit(spot)
    right = _unit(np.cross(back, [0., 0., -1.]))
    ring = np.stack([right, _unit(np.cross(back, right)), back, spot], -1)
    return local, ring, (bounds * shrink).astype(bounds.dtype)


def _with_hwf(c2w, hwf):
    return np.concatenate([c2w, np.broadcast_to(np.asarray(hwf).reshape(3, 1), c2w.shape[:-1] + (1,))], -1)


def load_llff_data(basedir, factor=8, recenter=True, bd_factor=.75, spherify=False, path_zflat=False):
    c2w, hwf, bounds, images = read_capture(basedir, factor=factor)
    # (down, right, back) -> (right, up, back); the reference continues in float32 from here on
    c2w = np.stack([c2w[:, :, 1], -c2w[:, :, 0], c2w[:, :, 2], c2w[:, :, 3]], -1).astype(np.float32)
    bounds = bounds.astype(np.float32)
    hwf = hwf.astype(np.float32)
    if bd_factor is not None:                       # nearest scene depth -> 1 / bd_factor
        shrink = 1. / (bounds.min() * bd_factor)
        c2w[:, :, 3] *= shrink
        bounds = (bounds * shrink).astype(np.float32)
    if recenter:
        c2w = in_frame(c2w, mean_frame(c2w)).astype(np.float32)
    if spherify:
        c2w, path, bounds = normalise_to_sphere(c2w, bounds)
    else:
        centre = mean_frame(c2w)
        up = _unit(c2w[:, :, 1].sum(0))
        near_depth, far_depth = bounds.min() * .9, bounds.max() * 5.
        blend = .75                                  # focus depth: harmonic blend of the depth range
        focus_depth = 1. / ((1. - blend) / near_depth + blend / far_depth)
        radii = np.percentile(np.abs(c2w[:, :, 3]), 90, 0)
        n_views, turns = 120, 2
        if path_zflat:                               # planar path slightly behind the mean camera
            centre[:, 3] += -near_depth * .1 * centre[:, 2]
            radii[2] = 0.
            n_views, turns = n_views // 2, 1         # (the reference leaves a float count here, which numpy's linspace rejects)
        path = spiral_path(centre, up, radii, focus_depth, .5, turns, n_views)
    poses = _with_hwf(c2w, hwf).astype(np.float32)
    render_poses = _with_hwf(path, hwf).astype(np.float32)
    holdout = np.argmin(np.sum(np.square(mean_frame(poses[:, :, :4])[:, 3] - poses[:, :, 3]), -1))
    return images.astype(np.float32), poses, bounds, render_poses, holdout
