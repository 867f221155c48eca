#!/usr/bin/env python3
"""Golden vectors for the dataset readers, recorded from the REFERENCE's load_blender.py / load_llff.py
(build container only; imageio is stubbed with a PIL reader).  The fixture holds the synthetic datasets themselves
(images, json fields, poses_bounds) and the loaders' outputs -> tests/golden/g15_loaders.npz."""
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import install_stubs, OUT  # noqa: E402

REF = '/root/reference/nerf-ours'


def write_blender(d, rng):
    from PIL import Image
    rec = {}
    for split, n in (('train', 3), ('val', 4), ('test', 5)):
        frames = []
        os.makedirs(os.path.join(d, split), exist_ok=True)
        imgs = rng.integers(0, 256, size=(n, 8, 6, 4), dtype=np.uint8)
        mats = rng.normal(size=(n, 4, 4))
        for i in range(n):
            Image.fromarray(imgs[i], 'RGBA').save(os.path.join(d, split, 'r_%d.png' % i))
            frames.append({'file_path': './%s/r_%d' % (split, i), 'transform_matrix': mats[i].tolist()})
        ang = 0.6 + 0.1 * len(frames)
        json.dump({'camera_angle_x': ang, 'frames': frames}, open(os.path.join(d, 'transforms_%s.json' % split), 'w'))
        rec['blender.%s.imgs' % split] = imgs
        rec['blender.%s.mats' % split] = mats
        rec['blender.%s.angle' % split] = np.float64(ang)
    return rec


def write_llff(d, rng, n=7):
    from PIL import Image
    os.makedirs(os.path.join(d, 'images'), exist_ok=True)
    os.makedirs(os.path.join(d, 'images_2'), exist_ok=True)
    full = rng.integers(0, 256, size=(n, 12, 16, 3), dtype=np.uint8)
    half = rng.integers(0, 256, size=(n, 6, 8, 3), dtype=np.uint8)
    for i in range(n):
        Image.fromarray(full[i], 'RGB').save(os.path.join(d, 'images', 'im_%02d.png' % i))
        Image.fromarray(half[i], 'RGB').save(os.path.join(d, 'images_2', 'im_%02d.png' % i))
    # plausible forward-facing rig: small rotations around identity, cameras near the origin, hwf column, bounds
    pb = np.zeros((n, 17))
    for i in range(n):
        q, _ = np.linalg.qr(np.eye(3) + 0.1 * rng.normal(size=(3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        pose = np.concatenate([q, 0.3 * rng.normal(size=(3, 1)), np.array([[12.], [16.], [20.]])], 1)
        pb[i, :15] = pose.reshape(-1)
        pb[i, 15:] = [1.5 + rng.random(), 8. + 4 * rng.random()]
    np.save(os.path.join(d, 'poses_bounds.npy'), pb)
    return {'llff.full': full, 'llff.half': half, 'llff.poses_bounds': pb}


def main():
    install_stubs()
    from PIL import Image
    import imageio
    imageio.imread = lambda f, **kw: np.asarray(Image.open(f))
    sys.path.insert(0, REF)
    import load_blender as LB
    import load_llff as LL
    rng = np.random.default_rng(2024)
    rec = {}
    with tempfile.TemporaryDirectory() as tmp:
        bd = os.path.join(tmp, 'blender'); os.makedirs(bd)
        rec.update(write_blender(bd, rng))
        for skip in (1, 2, 0):
            imgs, poses, render_poses, hwf, i_split = LB.load_blender_data(bd, half_res=False, testskip=skip)
            p = 'blender.out%d.' % skip
            rec[p + 'imgs'] = imgs; rec[p + 'poses'] = poses; rec[p + 'render_poses'] = render_poses.numpy()
            rec[p + 'hwf'] = np.array(hwf, dtype=np.float64)
            for k, s in enumerate(i_split):
                rec[p + 'split%d' % k] = s
        ld = os.path.join(tmp, 'llff'); os.makedirs(ld)
        rec.update(write_llff(ld, rng))
        cases = {'a': dict(factor=2, recenter=True, bd_factor=.75, spherify=False),
                 'b': dict(factor=2, recenter=False, bd_factor=None, spherify=False),
                 'c': dict(factor=None, recenter=True, bd_factor=.75, spherify=True)}
        for name, kw in cases.items():
            images, poses, bds, render_poses, i_test = LL.load_llff_data(ld, **kw)
            p = 'llff.out_%s.' % name
            rec[p + 'images'] = images; rec[p + 'poses'] = poses; rec[p + 'bds'] = bds
            rec[p + 'render_poses'] = render_poses; rec[p + 'i_test'] = np.int64(i_test)
    np.savez_compressed(os.path.join(OUT, 'g15_loaders.npz'), **rec)
    print('wrote', len(rec), 'arrays')


if __name__ == '__main__':
    main()
