#!/usr/bin/env python3
"""G17: nerf++ scene-directory reader (data_loader_split.py:27-106 + RaySamplerSingleImage, nerf_sample_ray_split.py:36-173)
recorded from the REFERENCE (build container only).  A tiny scene (train: 3 views of 8x12 with masks, test: 2 views) is
written into a temp dir; the fixture carries the raw files' contents (so the test can re-create the directory), and what
the reference's samplers report: sizes, scaled intrinsics, the half-resolution image / mask, rays of every view, and the
pixel indices of seeded random_sample calls (plain and center-crop).

cv2 is absent: `cv2.resize` is stood in for by block means (INTER_AREA) / top-left picks (INTER_NEAREST) for the integer
factor 2 this path uses -- the same restatement the product documents as "parity unpinned"; imageio by a PIL reader."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, install_stubs  # noqa: E402

REF = '/root/reference/nerf++-ours'


def main():
    install_stubs()
    from PIL import Image
    cv2 = sys.modules['cv2']
    cv2.INTER_NEAREST, cv2.INTER_LINEAR = 0, 1

    def resize(img, wh, interpolation=None):
        w, h = wh
        f = img.shape[0] // h
        assert f * h == img.shape[0] and f * w == img.shape[1]
        if interpolation == cv2.INTER_AREA:
            return img.reshape((h, f, w, f) + img.shape[2:]).mean(axis=(1, 3), dtype=np.float32)
        return img[::f, ::f]
    cv2.resize = resize
    sys.modules['imageio'].imread = lambda p: np.asarray(Image.open(p))
    sys.path.insert(0, REF)
    import data_loader_split as D
    rng = np.random.RandomState(17)
    rec = {}
    with tempfile.TemporaryDirectory() as base:
        for split, n in (('train', 3), ('test', 2)):
            for sub in ('rgb', 'intrinsics', 'pose') + (('mask',) if split == 'train' else ()):
                os.makedirs(os.path.join(base, 'scene', split, sub))
            for i in range(n):
                img = rng.randint(0, 256, (8, 12, 3)).astype(np.uint8)
                Image.fromarray(img, 'RGB').save(os.path.join(base, 'scene', split, 'rgb', '%03d.png' % i))
                K = np.eye(4); K[0, 0] = K[1, 1] = 20.0 + i; K[0, 2] = 6.0; K[1, 2] = 4.0
                th = 0.3 * i + (0.1 if split == 'test' else 0.0)
                c2w = np.eye(4)
                c2w[:3, :3] = [[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]]
                c2w[:3, 3] = [0.1 * i, -0.05, 0.2 - 0.1 * i]
                np.savetxt(os.path.join(base, 'scene', split, 'intrinsics', '%03d.txt' % i), K.reshape(1, 16))
                np.savetxt(os.path.join(base, 'scene', split, 'pose', '%03d.txt' % i), c2w.reshape(1, 16))
                rec['%s.img%d' % (split, i)], rec['%s.K%d' % (split, i)], rec['%s.c2w%d' % (split, i)] = img, K, c2w
                if split == 'train':
                    m = (rng.rand(8, 12) > 0.5).astype(np.uint8) * 255
                    Image.fromarray(m, 'L').save(os.path.join(base, 'scene', split, 'mask', '%03d.png' % i))
                    rec['train.mask%d' % i] = m
        for split, n in (('train', 3), ('test', 2)):
            samplers = D.load_data_split(base + '/', 'scene', split, skip=1)
            assert len(samplers) == n
            for i, s in enumerate(samplers):
                p = '%s.out%d.' % (split, i)
                rec[p + 'HW'] = np.array([s.H, s.W])
                rec[p + 'intrinsics'] = s.intrinsics
                rec[p + 'img'] = s.img
                if s.mask is not None:
                    rec[p + 'mask'] = s.mask
                rec[p + 'rays_o'], rec[p + 'rays_d'], rec[p + 'depth'] = s.rays_o, s.rays_d, s.depth
                al = s.get_all()
                rec[p + 'all_min_depth'] = al['min_depth'].numpy()
            np.random.seed(5)
            a = samplers[0].random_sample(7, center_crop=False)
            b = samplers[1].random_sample(4, center_crop=True)
            rec['%s.rand_rgb' % split], rec['%s.rand_ray_d' % split] = a['rgb'].numpy(), a['ray_d'].numpy()
            rec['%s.crop_rgb' % split], rec['%s.crop_ray_d' % split] = b['rgb'].numpy(), b['ray_d'].numpy()
        rec['skip2_count'] = np.array(len(D.load_data_split(base, 'scene', 'train', skip=2)))
        rec['only_img_files'] = np.array([os.path.basename(f) for f in D.load_data_split(base, 'scene', 'test', only_img_files=True)])
    np.savez_compressed(os.path.join(OUT, 'g17_pp_loader.npz'), **rec)
    print('wrote g17_pp_loader.npz', len(rec), 'arrays')


if __name__ == '__main__':
    main()
