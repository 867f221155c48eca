"""nerf++ scene-directory reader (fastnerf.data_loader_split) vs G17, recorded from the reference's
data_loader_split.py:27-106 + RaySamplerSingleImage (oracle/make_golden_pp_loader.py).  Host side here; the rays are
checked on the GPU in tests/test_gpu_nerfpp.py::test_pp_loader_rays."""
import os

import numpy as np
import pytest
import torch
from PIL import Image


@pytest.fixture(scope='module')
def g(golden_dir):
    return np.load(os.path.join(golden_dir, 'g17_pp_loader.npz'))


def write_scene(g, base):
    for split, n in (('train', 3), ('test', 2)):
        for sub in ('rgb', 'intrinsics', 'pose') + (('mask',) if split == 'train' else ()):
            os.makedirs(os.path.join(base, 'scene', split, sub))
        for i in range(n):
            Image.fromarray(g['%s.img%d' % (split, i)], 'RGB').save(os.path.join(base, 'scene', split, 'rgb', '%03d.png' % i))
            np.savetxt(os.path.join(base, 'scene', split, 'intrinsics', '%03d.txt' % i), g['%s.K%d' % (split, i)].reshape(1, 16))
            np.savetxt(os.path.join(base, 'scene', split, 'pose', '%03d.txt' % i), g['%s.c2w%d' % (split, i)].reshape(1, 16))
            if split == 'train':
                Image.fromarray(g['train.mask%d' % i], 'L').save(os.path.join(base, 'scene', split, 'mask', '%03d.png' % i))


def test_directory_reader(g, tmp_path):
    from fastnerf.data_loader_split import RaySamplerSingleImage, find_files, load_data_split, read_matrix_txt
    base = str(tmp_path)
    write_scene(g, base)
    for split, n in (('train', 3), ('test', 2)):
        samplers = load_data_split(base + '/', 'scene', split, skip=1, device='cpu')      # (no rays are generated here)
        assert len(samplers) == n and all(isinstance(s, RaySamplerSingleImage) for s in samplers)
        for i, s in enumerate(samplers):
            p = '%s.out%d.' % (split, i)
            assert [s.H, s.W] == g[p + 'HW'].tolist() == [4, 6] and (s.H_orig, s.W_orig) == (8, 12)      # half resolution
            assert s.intrinsics.dtype == np.float32 and np.array_equal(s.intrinsics, g[p + 'intrinsics'])
            assert s.img.dtype == np.float32 and np.abs(s.img - g[p + 'img']).max() < 1e-7
            assert np.array_equal(s.get_img(), s.img.reshape(4, 6, 3))
            if split == 'train':
                assert np.array_equal(s.mask, g[p + 'mask'])
            else:
                assert s.mask is None
            assert s.min_depth is None and s.max_depth is None and s.img_path.endswith('%03d.png' % i)
            assert np.array_equal(s._near().numpy(), g[p + 'all_min_depth'])
        # seeded batches: the same numpy draws as the reference -> the same pixels (colours compared; rays on the GPU)
        np.random.seed(5)
        ia = samplers[0].select_indices(7, center_crop=False)
        ib = samplers[1].select_indices(4, center_crop=True)
        assert np.abs(samplers[0].img[ia] - g['%s.rand_rgb' % split]).max() < 1e-7
        assert np.abs(samplers[1].img[ib] - g['%s.crop_rgb' % split]).max() < 1e-7
        rows, cols = ib // 6, ib % 6
        assert ((rows >= 1) & (rows < 3) & (cols >= 2) & (cols < 4)).all()            # the central half-size window
    assert len(load_data_split(base, 'scene', 'train', skip=2, device='cpu')) == int(g['skip2_count']) == 2
    files = load_data_split(base, 'scene', 'test', only_img_files=True)
    assert [os.path.basename(f) for f in files] == g['only_img_files'].tolist()
    assert find_files(os.path.join(base, 'nope'), ['*.txt']) == []
    with pytest.raises(ValueError):
        bad = os.path.join(base, 'bad.txt')
        open(bad, 'w').write('1 2 3')
        read_matrix_txt(bad)


def test_min_depth_half_resolution_is_the_bilinear_sample_at_pixel_centres():
    """load_data_split builds its samplers at resolution level 2, and a scene with a min_depth folder must load there.
    cv2.resize(INTER_LINEAR) samples destination pixel x at 2x + 0.5 in the source: the mean of the 2 x 2 block (restated from
    OpenCV's documented rule; parity unpinned -- OpenCV is neither vendored nor pinned by the reference)."""
    import numpy as np
    from fastnerf.data_loader_split import shrink_linear
    rng = np.random.default_rng(0)
    img = rng.random((10, 14)).astype(np.float32)
    out = shrink_linear(img, 2)
    assert out.shape == (5, 7) and out.dtype == np.float32
    ref = 0.25 * (img[0::2, 0::2] + img[1::2, 0::2] + img[0::2, 1::2] + img[1::2, 1::2])
    assert np.abs(out - ref).max() < 1e-6
    assert np.array_equal(shrink_linear(img, 1), img)
    o3 = shrink_linear(img[:9, :12], 3)                      # odd factor: the block's central pixel
    assert np.array_equal(o3, img[1:9:3, 1:12:3])
    o4 = shrink_linear(rng.random((8, 8)).astype(np.float32), 4)
    assert o4.shape == (2, 2)
    # a size that is not a multiple of the level means a non-integer INTER_LINEAR scale in the reference -- refused loudly,
    # never cropped silently
    import pytest
    with pytest.raises(NotImplementedError):
        shrink_linear(img[:9, :11], 2)
