"""CPU oracle for the quadtree ray-selection path -- TEST INFRASTRUCTURE ONLY.

Restates nerf-ours/tree.py (QuadTreeNode / QuadTree / QuadTreeManager
gen_rays_v3_multiThread + adjust_tree_multiThread) with plain Python floats so
that every float comparison the reference performs (`area == minArea`,
`area > minArea + 0.01`, ceil/ceil(-0.01) pixel ranges) has the same value.
Only tests / smoke / the bench's cpu_baseline may import it.

Parity status: PINNED by `tests/golden/tree_*.npz`, produced by
`oracle/make_golden.py` from the reference itself (leaf lists, per-leaf ray
counts, seeded `result_leaf_id`, before/after leaf lists of seeded adjust
sequences).

Representation: a tree is the DFS-ordered list of its leaves.  Splitting a
leaf replaces it in place by its four children in the reference's child order
(tree.py:61-72: (x0,y0,mx,my), (mx,y0,x1,my), (x0,my,mx,y1), (mx,my,x1,y1);
x = row axis, y = column axis), which is exactly what the reference's
get_children() DFS (tree.py:679-686) enumerates.
"""
import math

import numpy as np
import torch


def split_box(b):
    """subdivide_once (tree.py:57-72)."""
    x0, y0, x1, y1 = b
    mx = (x0 + x1) / 2
    my = (y0 + y1) / 2
    return [(x0, y0, mx, my), (mx, y0, x1, my), (x0, my, mx, y1), (mx, my, x1, y1)]


def box_area(b):
    """QuadTreeNode.area (tree.py:74-76)."""
    return (b[2] - b[0]) * (b[3] - b[1])


def block_error(b, img):
    """QuadTreeNode.get_error (tree.py:29-55): sum over channels of the
    population variance of the block's pixels."""
    x0, x1 = math.ceil(b[0]), math.floor(b[2])
    y0, y1 = math.ceil(b[1]), math.floor(b[3])
    px = np.asarray(img)[x0:x1, y0:y1, :]
    tot = 0.0
    for c in range(3):
        ch = px[:, :, c]
        avg = np.mean(ch)
        tot = tot + np.square(np.subtract(ch, avg)).mean()
    return tot


class Tree:
    """QuadTree (tree.py:82-99) reduced to its leaf list + minArea."""

    def __init__(self, H, W, max_depth, image=None, thres=0.0):
        self.H, self.W = H, W
        self.leaves = []
        self._build((0, 0, H, W), 1, max_depth, image, thres)
        self.minArea = H * W / (4 ** (max_depth - 1))

    def _build(self, b, depth, max_depth, image, thres):
        # recursive_subdivide (tree.py:655-676)
        if depth >= max_depth:
            self.leaves.append(b)
            return
        if image is not None and block_error(b, image) < thres:
            self.leaves.append(b)
            return
        for c in split_box(b):
            self._build(c, depth + 1, max_depth, image, thres)


def leaf_ray_num(tree, b, ray_num_per_pixel):
    """tree.py:578-581."""
    if box_area(b) > tree.minArea + 0.01:
        return 10
    return int(box_area(b) * ray_num_per_pixel)


def leaf_pixel_range(b):
    """tree.py:598-599: row range [lo,hi), col range [lo,hi)."""
    return (math.ceil(b[0]), math.ceil(b[2]), math.ceil(b[1]), math.ceil(b[3] - 0.01))


class Manager:
    """QuadTreeManager (tree.py:159-193, 377-428, 533-557) without rays: it
    produces pixel picks (img, row, col) + leaf tags; ray/rgb gathering is a
    plain index and is done by the caller."""

    def __init__(self, H, W, n_images, max_depth):
        self.h, self.w, self.n_images = H, W, n_images
        self.epoch_size = n_images * H * W
        self.trees = [Tree(H, W, max_depth) for _ in range(n_images)]
        self.cur_level = max_depth
        self.result_leaf_id = None

    def gen_pixels(self, down_scale=1, last_epoch=False):
        """gen_rays_v3_multiThread(prob=False) (tree.py:377-428) + sub-thread
        (tree.py:569-626), ThreadPool(1) => serial, torch global CPU RNG."""
        ray_num_per_image = self.epoch_size / self.n_images / down_scale
        ray_num_per_pixel = ray_num_per_image / self.h / self.w
        trees = [Tree(self.h, self.w, 1) for _ in range(self.n_images)] if last_epoch else self.trees
        pix, tags = [], []
        for ti, tr in enumerate(trees):
            for li, b in enumerate(tr.leaves):
                n = leaf_ray_num(tr, b, ray_num_per_pixel)
                r0, r1, c0, c1 = leaf_pixel_range(b)
                xs = torch.randint(r0, r1, (n,))
                ys = torch.randint(c0, c1, (n,))
                pix.append(torch.stack([torch.full((n,), ti, dtype=torch.int64), xs, ys], 1))
                tags.append(torch.tensor([[ti, li]], dtype=torch.float32).repeat([n, 1]))
        pix = torch.cat(pix, 0)
        tags = torch.cat(tags, 0)
        perm = torch.randperm(pix.shape[0])
        self.result_leaf_id = tags[perm]
        return pix[perm]

    def adjust(self, rgb_gt, rgb_pred, thres):
        """adjust_tree_multiThread + adjust_tree_subThread (tree.py:533-557,
        629-652): split a finest leaf when max |gt-pred| over its rays and the
        three channels exceeds thres."""
        loss = torch.abs(rgb_gt - rgb_pred)
        for ti, tr in enumerate(self.trees):
            rows = torch.where(self.result_leaf_id[:, 0] == ti)
            leaves = self.result_leaf_id[rows][:, 1]
            li_loss = loss[rows]
            min_area = tr.minArea
            new_leaves = []
            for li, b in enumerate(tr.leaves):
                sel = torch.where(leaves == li)
                if li_loss[sel].max() > thres and box_area(b) == min_area:
                    new_leaves.extend(split_box(b))
                    if tr.minArea == min_area:
                        tr.minArea /= 4
                else:
                    new_leaves.append(b)
            tr.leaves = new_leaves
        self.cur_level += 1

    def adjust_from_table(self, table, thres):
        """Same decision driven by a precomputed per-(image, leaf) max table
        (what the device path produces); leaves with no ray keep -inf."""
        for ti, tr in enumerate(self.trees):
            min_area = tr.minArea
            new_leaves = []
            for li, b in enumerate(tr.leaves):
                if float(table[ti][li]) > thres and box_area(b) == min_area:
                    new_leaves.extend(split_box(b))
                    if tr.minArea == min_area:
                        tr.minArea /= 4
                else:
                    new_leaves.append(b)
            tr.leaves = new_leaves
        self.cur_level += 1

    def leaf_array(self, ti):
        return np.array(self.trees[ti].leaves, dtype=np.float64).reshape(-1, 4)


# --------------------------------------------------------------------------
# nerf++-ours fork: variance-weighted picks (prob=True)
# --------------------------------------------------------------------------
def to_prob_v2(gray_img):
    """ImageProcessor.to_prob_v2 (nerf++-ours/image_process.py:58-72): sampling probability of every pixel of a block of
    the local-variance map -- values + 1e-6, clipped from below at 1 % of their mean, scaled by the maximum, normalised."""
    raw_shape = np.shape(gray_img)
    g = np.asarray(gray_img, dtype=np.float64).flatten() + 1e-6
    g_min = 0.01 * np.mean(g)
    g_max = np.max(g)
    g = np.clip(g, g_min, g_max)
    g = (g - 0) / (g_max - 0)
    return np.reshape(g / np.sum(g), raw_shape)


def leaf_pick_split(ray_num, rand_samp):
    """nerf++-ours/tree.py:566-568: (weighted picks, uniform picks) of a leaf."""
    n1 = int(ray_num * (1 - rand_samp))
    return n1, ray_num - n1


def expected_pixel_counts(trees, H, W, sharp_imgs, ray_num_per_pixel, prob, rand_samp):
    """Expected number of picks per pixel of ONE epoch of gen_rays_v3_1_subThread (nerf++-ours/tree.py:548-607): per leaf
    `leaf_ray_num` picks; with prob=True the first int(n (1 - rand)) of them are np.random.choice draws over the block
    [int(x0):int(x1), int(y0):int(y1)] of the variance map with probability to_prob_v2 (image_process.py:74-93), the rest
    (all of them with prob=False) torch.randint over rows [ceil(x0), ceil(x1)) x columns [ceil(y0), ceil(y1 - 0.01)).
    -> float64 [n_images, H, W]."""
    out = np.zeros((len(trees), H, W), dtype=np.float64)
    for ti, tr in enumerate(trees):
        for b in tr.leaves:
            n = leaf_ray_num(tr, b, ray_num_per_pixel)
            n1, n2 = leaf_pick_split(n, rand_samp) if prob else (0, n)
            if n1 > 0:
                x0, y0, x1, y1 = int(b[0]), int(b[1]), int(b[2]), int(b[3])
                out[ti, x0:x1, y0:y1] += n1 * to_prob_v2(np.asarray(sharp_imgs[ti])[x0:x1, y0:y1])
            r0, r1, c0, c1 = leaf_pixel_range(b)
            out[ti, r0:r1, c0:c1] += n2 / float((r1 - r0) * (c1 - c0))
    return out
