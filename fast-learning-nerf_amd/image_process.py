"""Mirror of {nerf-ours,nerf++-ours}/image_process.py's ImageProcessor pieces that the quadtree's
`prob=True` ray picks use: local-variance "sharpness" map, its normalisation to a
sampling probability and the weighted pixel draw.  Host-side (numpy), like the reference.

Parity note: `get_sharp_img` is built on cv2.blur / cv2.cvtColor in the reference; OpenCV is neither
vendored nor pinned there and is absent here, so this map is restated from OpenCV's documented
semantics (3x3 normalised box filter, BORDER_REFLECT_101; BGR2GRAY = 0.114 B + 0.587 G + 0.299 R) and
is "parity unpinned".  Everything downstream of the map (to_prob_v2, sample_pixels, the tree) is pinned
by goldens that take the map as an input fixture; callers may pass their own maps."""
import numpy as np
import torch


def box_blur3(img):
    pad = np.pad(img, ((1, 1), (1, 1)) + ((0, 0),) * (img.ndim - 2), mode='reflect')
    out = np.zeros_like(img)
    for dx in range(3):
        for dy in range(3):
            out = out + pad[dx:dx + img.shape[0], dy:dy + img.shape[1]]
    return out / 9.0


class ImageProcessor:
    def __init__(self, images, scale=50, sharp_imgs=None):
        self.scale = scale
        self.n_images, self.h, self.w = len(images), images[0].shape[0], images[0].shape[1]
        self.images = images
        self.images_np = np.stack([np.asarray(im) for im in images], 0)
        self.sharp_imgs = list(sharp_imgs) if sharp_imgs is not None else \
            [self.get_sharp_img(self.images_np[i]) for i in range(self.n_images)]

    def get_sharp_img(self, img):
        """image_process.py:26-39."""
        e_square = box_blur3(img ** 2)
        square_e = box_blur3(img) ** 2
        sharp = np.sqrt(np.abs(e_square - square_e))
        bgr = sharp[:, :, [2, 1, 0]]
        return 0.114 * bgr[..., 0] + 0.587 * bgr[..., 1] + 0.299 * bgr[..., 2]

    def to_prob_v2(self, gray_img):
        """image_process.py:58-72."""
        raw_shape = gray_img.shape
        g = np.asarray(gray_img, dtype=np.float64).flatten() + 1e-6
        g_min = 0.01 * np.mean(g)
        g_max = np.max(g)
        g = np.clip(g, g_min, g_max)
        g = (g - 0) / (g_max - 0)
        return np.reshape(g / np.sum(g), raw_shape)

    def sample_pixels(self, image, sample_num=320000):
        """image_process.py:74-93: weighted draw with numpy's global RNG."""
        prob = self.to_prob_v2(image)
        h, w = prob.shape
        idx = np.random.choice(h * w, sample_num, p=prob.reshape(-1))
        sx = np.floor(idx / w)
        sy = idx - sx * w
        return torch.stack([torch.LongTensor(sx), torch.LongTensor(sy)], 1)
