"""Dataset readers vs the reference's own outputs (G15): the fixture carries the synthetic datasets
(pixels, json fields, poses_bounds), the test writes them to disk and reads them back through our loaders."""
import json
import os

import numpy as np
import pytest

PIL = pytest.importorskip('PIL.Image')


@pytest.fixture(scope='module')
def g(golden_dir):
    return np.load(os.path.join(golden_dir, 'g15_loaders.npz'))


@pytest.fixture(scope='module')
def loaders():
    import fastnerf
    from fastnerf import load_blender, load_llff
    return load_blender, load_llff


def test_blender_reader(g, loaders, tmp_path):
    LB, _ = loaders
    d = str(tmp_path / 'blender')
    for split in ('train', 'val', 'test'):
        os.makedirs(os.path.join(d, split))
        imgs, mats = g['blender.%s.imgs' % split], g['blender.%s.mats' % split]
        frames = []
        for i in range(imgs.shape[0]):
            PIL.fromarray(imgs[i], 'RGBA').save(os.path.join(d, split, 'r_%d.png' % i))
            frames.append({'file_path': './%s/r_%d' % (split, i), 'transform_matrix': mats[i].tolist()})
        json.dump({'camera_angle_x': float(g['blender.%s.angle' % split]), 'frames': frames},
                  open(os.path.join(d, 'transforms_%s.json' % split), 'w'))
    for skip in (1, 2, 0):
        imgs, poses, render_poses, hwf, i_split = LB.load_blender_data(d, half_res=False, testskip=skip)
        p = 'blender.out%d.' % skip
        assert imgs.dtype == np.float32 and np.array_equal(imgs, g[p + 'imgs'])
        assert poses.dtype == np.float32 and np.array_equal(poses, g[p + 'poses'])
        assert np.abs(render_poses.numpy() - g[p + 'render_poses']).max() < 1e-6
        assert np.allclose(np.array(hwf, dtype=np.float64), g[p + 'hwf'], rtol=0, atol=1e-12)
        for k in range(3):
            assert np.array_equal(i_split[k], g[p + 'split%d' % k])
    # half_res = mean of 2x2 blocks (cv2.INTER_AREA at factor 2; OpenCV absent here, parity unpinned)
    imgs_h, _, _, hwf_h, _ = LB.load_blender_data(d, half_res=True, testskip=1)
    full = g['blender.out1.imgs']
    assert imgs_h.shape == (full.shape[0], 4, 3, 4) and hwf_h[:2] == [4, 3]
    assert np.abs(imgs_h[0, 0, 0] - full[0, :2, :2].reshape(4, 4).mean(0)).max() < 1e-7
    assert abs(hwf_h[2] - g['blender.out1.hwf'][2] / 2) < 1e-12


def test_llff_reader(g, loaders, tmp_path):
    _, LL = loaders
    d = str(tmp_path / 'llff')
    os.makedirs(os.path.join(d, 'images'))
    os.makedirs(os.path.join(d, 'images_2'))
    for i in range(g['llff.full'].shape[0]):
        PIL.fromarray(g['llff.full'][i], 'RGB').save(os.path.join(d, 'images', 'im_%02d.png' % i))
        PIL.fromarray(g['llff.half'][i], 'RGB').save(os.path.join(d, 'images_2', 'im_%02d.png' % i))
    np.save(os.path.join(d, 'poses_bounds.npy'), g['llff.poses_bounds'])
    cases = {'a': dict(factor=2, recenter=True, bd_factor=.75, spherify=False),
             'b': dict(factor=2, recenter=False, bd_factor=None, spherify=False),
             'c': dict(factor=None, recenter=True, bd_factor=.75, spherify=True)}
    for name, kw in cases.items():
        images, poses, bds, render_poses, i_test = LL.load_llff_data(d, **kw)
        p = 'llff.out_%s.' % name
        assert images.dtype == np.float32 and np.array_equal(images, g[p + 'images'])
        for got, key in ((poses, 'poses'), (bds, 'bds'), (render_poses, 'render_poses')):
            ref = g[p + key]
            assert got.shape == ref.shape and got.dtype == ref.dtype, key
            assert np.abs(got - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), (name, key, np.abs(got - ref).max())
        assert int(i_test) == int(g[p + 'i_test'])
    with pytest.raises(FileNotFoundError):
        LL.load_llff_data(d, factor=8)          # images_8 was never made (the reference would shell out to mogrify)
    # path_zflat: 60-view single-rotation path at z = -0.1 * close depth
    _, _, _, rp, _ = LL.load_llff_data(d, factor=2, path_zflat=True)
    assert rp.shape == (60, 3, 5)


def test_load_dataset_branches(g, loaders, tmp_path):
    """run_nerf.load_dataset = the data branch of the reference's train() (run_nerf.py:160-241)."""
    import fastnerf
    LB, LL = loaders
    d = str(tmp_path / 'blender')
    for split in ('train', 'val', 'test'):
        os.makedirs(os.path.join(d, split))
        imgs, mats = g['blender.%s.imgs' % split], g['blender.%s.mats' % split]
        frames = []
        for i in range(imgs.shape[0]):
            PIL.fromarray(imgs[i], 'RGBA').save(os.path.join(d, split, 'r_%d.png' % i))
            frames.append({'file_path': './%s/r_%d' % (split, i), 'transform_matrix': mats[i].tolist()})
        json.dump({'camera_angle_x': float(g['blender.%s.angle' % split]), 'frames': frames},
                  open(os.path.join(d, 'transforms_%s.json' % split), 'w'))
    args = fastnerf.run_nerf.make_args(dataset_type='blender', datadir=d, half_res=False, testskip=1, white_bkgd=True)
    ds = fastnerf.run_nerf.load_dataset(args)
    rgba = g['blender.out1.imgs']
    assert np.array_equal(ds['images'], rgba[..., :3] * rgba[..., -1:] + (1. - rgba[..., -1:]))
    assert ds['poses'].shape == (12, 3, 4) and (ds['near'], ds['far']) == (2., 6.)
    assert [len(ds[k]) for k in ('i_train', 'i_val', 'i_test')] == [3, 4, 5]
    H, W, focal = ds['hwf']
    assert (H, W) == (8, 6) and np.allclose(ds['K'], [[focal, 0, 3.], [0, focal, 4.], [0, 0, 1]])
    with pytest.raises(ValueError):
        fastnerf.run_nerf.load_dataset(fastnerf.run_nerf.make_args(dataset_type='deepvoxels', datadir=d))
