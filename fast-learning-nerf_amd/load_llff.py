"""LLFF (forward-facing / 360) dataset reader: the reference's load_llff.py (:60-316) behind the same names.
Host-side numpy only.  Differences by construction: images are read with PIL; down-scaled image folders
(`images_<factor>`) must already exist -- the reference shells out to ImageMagick's `mogrify` to create them
(`_minify`, :7-57), which is outside this build."""
import os

import numpy as np


def _imread(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im)


def _is_img(f):
    return f.endswith('JPG') or f.endswith('jpg') or f.endswith('png')


def _load_data(basedir, factor=None, width=None, height=None, load_imgs=True):
    """poses_bounds.npy -> poses [3,5,n] (columns 0..3 = c2w in LLFF's axis order, column 4 = (H, W, focal)),
    bds [2,n], imgs [H,W,3,n] in [0,1] (load_llff.py:60-113)."""
    arr = np.load(os.path.join(basedir, 'poses_bounds.npy'))
    poses = arr[:, :-2].reshape([-1, 3, 5]).transpose([1, 2, 0])
    bds = arr[:, -2:].transpose([1, 0])
    full = os.path.join(basedir, 'images')
    sh0 = _imread(os.path.join(full, sorted(f for f in os.listdir(full) if _is_img(f))[0])).shape
    sfx = ''
    if factor is not None:
        sfx = '_{}'.format(factor)
    elif height is not None:
        factor = sh0[0] / float(height)
        width = int(sh0[1] / factor)
        sfx = '_{}x{}'.format(width, height)
    elif width is not None:
        factor = sh0[1] / float(width)
        height = int(sh0[0] / factor)
        sfx = '_{}x{}'.format(width, height)
    else:
        factor = 1
    imgdir = os.path.join(basedir, 'images' + sfx)
    if not os.path.exists(imgdir):
        raise FileNotFoundError(imgdir + ' does not exist (create the down-scaled copies first; the reference runs '
                                'ImageMagick mogrify for this, load_llff.py:7-57)')
    files = [os.path.join(imgdir, f) for f in sorted(os.listdir(imgdir)) if _is_img(f)]
    if poses.shape[-1] != len(files):
        raise ValueError('Mismatch between imgs {} and poses {}'.format(len(files), poses.shape[-1]))
    sh = _imread(files[0]).shape
    poses[:2, 4, :] = np.array(sh[:2]).reshape([2, 1])
    poses[2, 4, :] = poses[2, 4, :] * 1. / factor
    if not load_imgs:
        return poses, bds
    imgs = np.stack([_imread(f)[..., :3] / 255. for f in files], -1)
    return poses, bds, imgs


def normalize(x):
    return x / np.linalg.norm(x)


def viewmatrix(z, up, pos):
    """Camera frame looking along z with the given up hint: columns (right, true up, z, pos)."""
    z = normalize(z)
    right = normalize(np.cross(up, z))
    return np.stack([right, normalize(np.cross(z, right)), z, pos], 1)


def ptstocam(pts, c2w):
    return np.matmul(c2w[:3, :3].T, (pts - c2w[:3, 3])[..., np.newaxis])[..., 0]


def poses_avg(poses):
    """Mean position, summed viewing direction and summed up vector -> one [3,5] pose (hwf of pose 0)."""
    center = poses[:, :3, 3].mean(0)
    return np.concatenate([viewmatrix(normalize(poses[:, :3, 2].sum(0)), poses[:, :3, 1].sum(0), center), poses[0, :3, -1:]], 1)


def render_path_spiral(c2w, up, rads, focal, zdelta, zrate, rots, N):
    rads = np.array(list(rads) + [1.])
    hwf = c2w[:, 4:5]
    out = []
    for theta in np.linspace(0., 2. * np.pi * rots, N + 1)[:-1]:
        c = np.dot(c2w[:3, :4], np.array([np.cos(theta), -np.sin(theta), -np.sin(theta * zrate), 1.]) * rads)
        z = normalize(c - np.dot(c2w[:3, :4], np.array([0, 0, -focal, 1.])))
        out.append(np.concatenate([viewmatrix(z, up, c), hwf], 1))
    return out


def recenter_poses(poses):
    """Express every pose in the frame of the average pose (load_llff.py:166-178)."""
    out = poses + 0
    bottom = np.reshape([0, 0, 0, 1.], [1, 4])
    c2w = np.concatenate([poses_avg(poses)[:3, :4], bottom], -2)
    p44 = np.concatenate([poses[:, :3, :4], np.tile(bottom[None], [poses.shape[0], 1, 1])], -2)
    out[:, :3, :4] = (np.linalg.inv(c2w) @ p44)[:, :3, :4]
    return out


def spherify_poses(poses, bds):
    """360-degree captures (load_llff.py:184-241): recenter on the point closest to all optical axes, scale the mean
    camera distance to 1, and lay a 120-view circle at the cameras' mean height."""
    def to44(p):
        return np.concatenate([p, np.tile(np.reshape(np.eye(4)[-1, :], [1, 1, 4]), [p.shape[0], 1, 1])], 1)
    rays_d, rays_o = poses[:, :3, 2:3], poses[:, :3, 3:4]
    A = np.eye(3) - rays_d * np.transpose(rays_d, [0, 2, 1])
    b = -A @ rays_o
    center = np.squeeze(-np.linalg.inv((np.transpose(A, [0, 2, 1]) @ A).mean(0)) @ b.mean(0))
    up = (poses[:, :3, 3] - center).mean(0)
    v0 = normalize(up)
    v1 = normalize(np.cross([.1, .2, .3], v0))
    v2 = normalize(np.cross(v0, v1))
    c2w = np.stack([v1, v2, v0, center], 1)
    reset = np.linalg.inv(to44(c2w[None])) @ to44(poses[:, :3, :4])
    rad = np.sqrt(np.mean(np.sum(np.square(reset[:, :3, 3]), -1)))
    sc = 1. / rad
    reset[:, :3, 3] *= sc
    bds *= sc
    rad *= sc
    zh = np.mean(reset[:, :3, 3], 0)[2]
    radcircle = np.sqrt(rad ** 2 - zh ** 2)
    ring = []
    for th in np.linspace(0., 2. * np.pi, 120):
        origin = np.array([radcircle * np.cos(th), radcircle * np.sin(th), zh])
        z = normalize(origin)
        x = normalize(np.cross(z, np.array([0, 0, -1.])))
        ring.append(np.stack([x, normalize(np.cross(z, x)), z, origin], 1))
    ring = np.stack(ring, 0)
    hwf = poses[0, :3, -1:]
    ring = np.concatenate([ring, np.broadcast_to(hwf, ring[:, :3, -1:].shape)], -1)
    reset = np.concatenate([reset[:, :3, :4], np.broadcast_to(hwf, reset[:, :3, -1:].shape)], -1)
    return reset, ring, bds


def load_llff_data(basedir, factor=8, recenter=True, bd_factor=.75, spherify=False, path_zflat=False):
    """-> images [n,H,W,3], poses [n,3,5], bds [n,2], render_poses [m,3,5] (all float32), i_test (load_llff.py:244-316)."""
    poses, bds, imgs = _load_data(basedir, factor=factor)
    # LLFF stores (down, right, back); NeRF wants (right, up, back): swap the first two axes, negate the new second
    poses = np.concatenate([poses[:, 1:2, :], -poses[:, 0:1, :], poses[:, 2:, :]], 1)
    poses = np.moveaxis(poses, -1, 0).astype(np.float32)
    images = np.moveaxis(imgs, -1, 0).astype(np.float32)
    bds = np.moveaxis(bds, -1, 0).astype(np.float32)
    sc = 1. if bd_factor is None else 1. / (bds.min() * bd_factor)
    poses[:, :3, 3] *= sc
    bds *= sc
    if recenter:
        poses = recenter_poses(poses)
    if spherify:
        poses, render_poses, bds = spherify_poses(poses, bds)
    else:
        c2w = poses_avg(poses)
        up = normalize(poses[:, :3, 1].sum(0))
        close_depth, inf_depth = bds.min() * .9, bds.max() * 5.
        dt = .75
        focal = 1. / ((1. - dt) / close_depth + dt / inf_depth)   # "focus depth" of the spiral
        zdelta = close_depth * .2
        rads = np.percentile(np.abs(poses[:, :3, 3]), 90, 0)
        n_views, n_rots = 120, 2
        if path_zflat:
            c2w[:3, 3] = c2w[:3, 3] + (-close_depth * .1) * c2w[:3, 2]
            rads[2] = 0.
            n_rots, n_views = 1, n_views // 2   # (the reference leaves a float here, which current numpy rejects in linspace)
        render_poses = render_path_spiral(c2w, up, rads, focal, zdelta, zrate=.5, rots=n_rots, N=n_views)
    render_poses = np.array(render_poses).astype(np.float32)
    c2w = poses_avg(poses)
    i_test = np.argmin(np.sum(np.square(c2w[:3, 3] - poses[:, :3, 3]), -1))
    return images.astype(np.float32), poses.astype(np.float32), bds, render_poses, i_test
