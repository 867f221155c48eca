"""Quadtree ray selection -- mirror of nerf-ours/tree.py on a native backend.

The reference keeps a Python object graph per image and scans every ray of the epoch once per leaf
on the CPU (tree.py:629-652).  Here the trees are DFS leaf arrays in libfastnerf.so (host C++), the
per-(image, leaf) max |gt-pred| is reduced on the device while training runs
(fastnerf_mse_leafmax), and rays are generated on the fly from (image,row,col) picks instead of
gathering from [n,H,W,3] host arrays.  Leaf enumeration, per-leaf ray counts, pixel ranges and the
split rule are bit-exact with the reference (tests/test_tree_*.py).

API kept (everything run_nerf.py touches):
  QuadTreeManager(H, W, K, images, poses, mseThres, max_depth) (tree.py:161),
  gen_rays_v3_multiThread (:377-428), adjust_tree_multiThread (:533-557), attributes h, w, n_images, images,
  epoch_size, cur_level, result_leaf_id, origins, dirs, and -- as settable object VIEWS over the native leaf
  arrays -- `quadTrees` (list of QuadTree, run_nerf.py:342,544) and `childrens` (list of lists of leaf
  QuadTreeNode, run_nerf.py:343);
  QuadTreeNode (:17-79), QuadTree (:82-99), recursive_subdivide (:655-676), get_children (:679-686);
  save_quadtrees / load_quadtrees: the `treeDivide_{epoch:04d}.pkl` files of run_nerf.py:338-345,542-544,
  written so that the reference's own `pickle.load` accepts them and reading the reference's files.
"""
import ctypes as C
import io
import math
import pickle
import sys
import types

import numpy as np
import torch

from . import ops
from ._lib import check, lib


class QuadTreeNode:
    """tree.py:17-79: a float box (x = row axis, y = column axis) and its four children (or [])."""

    def __init__(self, x0, y0, x1, y1):
        self.x0, self.y0, self.x1, self.y1 = x0, y0, x1, y1
        self.children = []

    @property
    def area(self):
        return (self.x1 - self.x0) * (self.y1 - self.y0)

    def box(self):
        return (self.x0, self.y0, self.x1, self.y1)

    def get_error(self, img):
        """tree.py:29-55: sum over the three channels of the population variance of the block's pixels
        (rows ceil(x0)..floor(x1), columns ceil(y0)..floor(y1))."""
        r0, r1 = math.ceil(self.x0), math.floor(self.x1)
        c0, c1 = math.ceil(self.y0), math.floor(self.y1)
        px = img[r0:r1, c0:c1, :]
        if torch.is_tensor(px):
            px = px.detach().cpu().numpy()
        total = 0.0
        for c in range(3):
            ch = px[:, :, c]
            total = total + np.square(np.subtract(ch, np.mean(ch))).mean()
        return total

    def subdivide_once(self):
        """tree.py:57-72: midpoint split; child order (x0,y0,mx,my), (mx,y0,x1,my), (x0,my,mx,y1), (mx,my,x1,y1)."""
        mx, my = (self.x0 + self.x1) / 2, (self.y0 + self.y1) / 2
        self.children = [QuadTreeNode(self.x0, self.y0, mx, my), QuadTreeNode(mx, self.y0, self.x1, my),
                         QuadTreeNode(self.x0, my, mx, self.y1), QuadTreeNode(mx, my, self.x1, self.y1)]

    def __str__(self):
        return "({:.1f}, {:.1f}), ({:.1f}, {:.1f})".format(self.x0, self.y0, self.x1, self.y1)


def recursive_subdivide(node, thres, image, cur_depth, max_depth):
    """tree.py:655-676: split while depth < max_depth and the block's colour variance is >= thres."""
    if cur_depth >= max_depth or node.get_error(image) < thres:
        return
    node.subdivide_once()
    for child in node.children:
        recursive_subdivide(child, thres, image, cur_depth + 1, max_depth)


def get_children(node):
    """tree.py:679-686: the leaves below `node` in depth-first child order (= the leaf ids)."""
    out, stack = [], [node]
    while stack:
        n = stack.pop()
        if n.children:
            stack.extend(reversed(n.children))
        else:
            out.append(n)
    return out


class QuadTree:
    """tree.py:82-99.  `QuadTree(image, stdThres, max_depth)` builds the initial tree like the reference
    (uniform when stdThres == 0); `QuadTree.from_leaves` rebuilds the node graph from a DFS leaf array."""

    def __init__(self, image, stdThres, max_depth):
        self.H, self.W = image.shape[:2]
        self.threshold = stdThres
        self.image = image
        self.root = QuadTreeNode(0, 0, self.H, self.W)
        recursive_subdivide(self.root, self.threshold, self.image, 1, max_depth)
        self.minArea = self.H * self.W / (4 ** (max_depth - 1))

    @classmethod
    def from_leaves(cls, H, W, boxes, minArea, image=None, threshold=0.0):
        """Inverse of get_children for trees made of midpoint splits: boxes [n,4] in DFS order."""
        t = cls.__new__(cls)
        t.H, t.W, t.threshold, t.image, t.minArea = H, W, threshold, image, minArea
        boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
        t.root = QuadTreeNode(0, 0, H, W)
        pos = 0
        stack = [(t.root, 0)]
        while stack:
            node, depth = stack.pop()
            if pos >= boxes.shape[0]:
                raise ValueError('leaf list ends before the tree is complete')
            b = boxes[pos]
            if node.x0 == b[0] and node.y0 == b[1] and node.x1 == b[2] and node.y1 == b[3]:
                pos += 1
                continue
            if depth > 24:
                raise ValueError('leaf list is not the DFS enumeration of a midpoint quadtree')
            node.subdivide_once()
            stack.extend((c, depth + 1) for c in reversed(node.children))
        if pos != boxes.shape[0]:
            raise ValueError('leaf list has {} entries beyond the tree'.format(boxes.shape[0] - pos))
        return t

    def leaf_array(self):
        return np.array([n.box() for n in get_children(self.root)], dtype=np.float64).reshape(-1, 4)


# ---- treeDivide_*.pkl (run_nerf.py:338-345, 542-544) -------------------------------------------------
# The reference pickles `treeManager.quadTrees`, i.e. instances of ITS classes tree.QuadTree / tree.QuadTreeNode
# (pickle records "module name + class name" and restores __dict__ without calling __init__).  Files written here
# name the same two classes, so the reference's plain `pickle.load` rebuilds its own objects from them; files
# written by the reference are read by mapping those two names onto the classes above.  The per-tree image copy the
# reference drags along (7.7 MB per 800x800 view) is not written: nothing on the path reads QuadTree.image after
# construction.
_REF_MODULE = 'tree'


def _reference_classes():
    mod = sys.modules.get(_REF_MODULE)
    if mod is not None and hasattr(mod, 'QuadTree') and hasattr(mod, 'QuadTreeNode'):
        return mod.QuadTree, mod.QuadTreeNode, None
    stub = types.ModuleType(_REF_MODULE)
    T = type('QuadTree', (), {'__module__': _REF_MODULE})
    N = type('QuadTreeNode', (), {'__module__': _REF_MODULE})
    stub.QuadTree, stub.QuadTreeNode = T, N
    return T, N, stub


def save_quadtrees(trees, path):
    """Write a list of QuadTree (e.g. `treeManager.quadTrees`) as `treeDivide_XXXX.pkl`."""
    T, N, stub = _reference_classes()

    def conv(node):
        out = N.__new__(N)
        out.__dict__.update(x0=node.x0, y0=node.y0, x1=node.x1, y1=node.y1, children=[conv(c) for c in node.children])
        return out
    objs = []
    for t in trees:
        o = T.__new__(T)
        o.__dict__.update(H=t.H, W=t.W, threshold=t.threshold, image=None, root=conv(t.root), minArea=t.minArea)
        objs.append(o)
    prev = sys.modules.get(_REF_MODULE)
    if stub is not None:
        sys.modules[_REF_MODULE] = stub
    try:
        data = pickle.dumps(objs, protocol=2)
    finally:
        if stub is not None:
            if prev is None:
                sys.modules.pop(_REF_MODULE, None)
            else:
                sys.modules[_REF_MODULE] = prev
    with open(path, 'wb') as f:
        f.write(data)


class _TreeUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if name == 'QuadTree' and module.split('.')[-1] == 'tree':
            return QuadTree
        if name == 'QuadTreeNode' and module.split('.')[-1] == 'tree':
            return QuadTreeNode
        return super().find_class(module, name)


def load_quadtrees(path):
    """Read a `treeDivide_XXXX.pkl` written by save_quadtrees OR by the reference (run_nerf.py:542-544)."""
    with open(path, 'rb') as f:
        trees = _TreeUnpickler(io.BytesIO(f.read())).load()
    for t in trees:
        if not isinstance(t, QuadTree):
            raise TypeError('{} does not hold a list of QuadTree'.format(path))
    return trees


class QuadTreeManager:
    def __init__(self, H, W, K, images, poses, mseThres=0.1, max_depth=5, device='cuda', criterion='max',
                 sharp_imgs=None):
        """criterion: 'max' (nerf-ours, tree.py:642) or 'mean' (nerf++-ours fork, tree.py:622).
        sharp_imgs: optional per-image variance maps for prob=True picks (see image_process.py)."""
        assert criterion in ('max', 'mean')
        self.criterion = criterion
        self._sharp_in = sharp_imgs
        self.processor = None
        self.h, self.w = int(H), int(W)
        self.K = np.asarray(K, dtype=np.float64)
        self.n_images = int(poses.shape[0])
        self.device = torch.device(device)
        self.images = torch.as_tensor(images, dtype=torch.float32)
        self.poses = torch.as_tensor(poses, dtype=torch.float32)[:, :3, :4].contiguous()
        self.epoch_size = self.n_images * self.h * self.w
        self.cur_level = max_depth
        self._t = lib().fastnerf_tree_create(self.h, self.w, self.n_images, int(max_depth))
        if not self._t:
            raise RuntimeError('fastnerf_tree_create failed')
        self._views = None           # cached QuadTree views of the native leaf arrays (invalidated by every change)
        self._version = 0            # bumped by every change of the trees (keys the cached weighted-pick tables)
        self._wcache = None
        if mseThres != 0.0:
            # variance-gated initial subdivision (tree.py:101-156 with get_error; the reference driver always passes
            # 0.0): built by the Python mirror above, then handed to the native arrays like a loaded pickle
            imgs_np = self.images.cpu().numpy()
            self.quadTrees = [QuadTree(imgs_np[i], mseThres, max_depth) for i in range(self.n_images)]
        self._leaf_id = None         # [N,2] float32 (image, leaf), as the reference stores it (built lazily from the tags)
        self.result_leaf_tag = None  # [N,2] int32 on the device (what the kernels consume)

        self._dev_images = None
        self._dev_poses = None

    def __del__(self):
        try:
            if getattr(self, '_t', None):
                lib().fastnerf_tree_destroy(self._t)
                self._t = None
        except Exception:
            pass

    @property
    def result_leaf_id(self):
        if self._leaf_id is None and self.result_leaf_tag is not None:
            self._leaf_id = self.result_leaf_tag.float()
        return self._leaf_id

    @result_leaf_id.setter
    def result_leaf_id(self, v):
        self._leaf_id = v

    # ---- tree state ---------------------------------------------------------------------------
    def num_leaves(self, i):
        return check(lib().fastnerf_tree_num_leaves(self._t, i), 'fastnerf_tree_num_leaves')

    def max_leaves(self):
        return check(lib().fastnerf_tree_max_leaves(self._t), 'fastnerf_tree_max_leaves')

    def min_area(self, i):
        return float(lib().fastnerf_tree_min_area(self._t, i))

    def leaves(self, i):
        """[n_leaves,4] float64 (x0,y0,x1,y1) in the reference's get_children() order."""
        n = self.num_leaves(i)
        out = np.empty((n, 4), dtype=np.float64)
        check(lib().fastnerf_tree_get_leaves(self._t, i, out.ctypes.data), 'fastnerf_tree_get_leaves')
        return out

    # `quadTrees` / `childrens` (tree.py:183-193) are object views of the native state: reading builds (and caches)
    # QuadTree / QuadTreeNode graphs from the leaf arrays; ASSIGNING (what run_nerf.py:342-343 does after
    # pickle.load) writes the leaves + minArea back into the native arrays.  Mutating a returned node in place does
    # not reach the native side -- assign the list back.
    def _tree_views(self):
        if self._views is None:
            self._views = [QuadTree.from_leaves(self.h, self.w, self.leaves(i), self.min_area(i))
                           for i in range(self.n_images)]
        return self._views

    @property
    def quadTrees(self):
        return self._tree_views()

    @quadTrees.setter
    def quadTrees(self, trees):
        trees = list(trees)
        if len(trees) != self.n_images:
            raise ValueError('expected {} trees, got {}'.format(self.n_images, len(trees)))
        self.import_leaves([(np.array([[n.x0, n.y0, n.x1, n.y1] for n in get_children(t.root)], dtype=np.float64),
                             t.minArea) for t in trees])

    @property
    def childrens(self):
        return [get_children(t.root) for t in self._tree_views()]

    @childrens.setter
    def childrens(self, lists):
        lists = list(lists)
        if len(lists) != self.n_images:
            raise ValueError('expected {} leaf lists, got {}'.format(self.n_images, len(lists)))
        self.import_leaves([(np.array([[n.x0, n.y0, n.x1, n.y1] for n in leaves], dtype=np.float64).reshape(-1, 4),
                             self.min_area(i)) for i, leaves in enumerate(lists)])

    def export_leaves(self):
        return [(self.leaves(i), self.min_area(i)) for i in range(self.n_images)]

    def import_leaves(self, state):
        for i, (boxes, min_area) in enumerate(state):
            b = np.ascontiguousarray(boxes, dtype=np.float64)
            check(lib().fastnerf_tree_set_leaves(self._t, i, b.shape[0], b.ctypes.data, float(min_area)),
                  'fastnerf_tree_set_leaves')
        self._views = None
        self._version += 1

    def save_trees(self, path):
        """run_nerf.py:542-544: `pickle.dump(treeManager.quadTrees, f)` in the reference's own class names."""
        save_quadtrees(self.quadTrees, path)

    def load_trees(self, path, cur_level=None):
        """run_nerf.py:339-345."""
        self.quadTrees = load_quadtrees(path)
        if cur_level is not None:
            self.cur_level = cur_level

    def leaf_plan(self, i, ray_num_per_pixel, last_epoch=False):
        """[n,5] int32: ray count, row_lo, row_hi, col_lo, col_hi (tree.py:578-581,598-599)."""
        n = 1 if last_epoch else self.num_leaves(i)
        out = np.empty((n, 5), dtype=np.int32)
        check(lib().fastnerf_tree_leaf_plan(self._t, i, float(ray_num_per_pixel), int(bool(last_epoch)),
                                            out.ctypes.data), 'fastnerf_tree_leaf_plan')
        return out

    # ---- rays ---------------------------------------------------------------------------------
    def _dev(self):
        if self._dev_images is None:
            self._dev_images = self.images.to(self.device)
            self._dev_poses = self.poses.to(self.device)
        return self._dev_images, self._dev_poses

    def _all_rays(self):
        """[n,H,W,3] origins / directions of every pixel of every view (tree.py:170-180).  Nothing on the hot path
        needs them (rays are generated from the picks); they exist for the reference driver's warm-up, which indexes
        `treeManager.origins[i][rows, cols]` (run_nerf.py:386-388).  Built once on first use, on the device:
        2 x 768 MB at 100 x 800 x 800."""
        if getattr(self, '_rays_cache', None) is None:
            rays = [ops.gen_rays(self.h, self.w, self.K, self.poses[i]) for i in range(self.n_images)]
            self._rays_cache = (torch.stack([r[0] for r in rays], 0), torch.stack([r[1] for r in rays], 0))
        return self._rays_cache

    @property
    def origins(self):
        return self._all_rays()[0]

    @property
    def dirs(self):
        return self._all_rays()[1]

    def gather(self, pix):
        """pix [N,3] int64/int32 (image,row,col) -> rays_o, rays_d, rgb on the device."""
        imgs, poses = self._dev()
        p = pix.to(self.device)
        ro, rd = ops.gen_rays_pixels(p.int().contiguous(), poses, self.K)
        rgb = imgs[p[:, 0].long(), p[:, 1].long(), p[:, 2].long()]
        return ro, rd, rgb.contiguous()

    def gen_pixels(self, down_scale=1, last_epoch=False, compat_rng=True, prob=False, rand=1.0):
        """Pixel picks + leaf tags of one epoch (host side of tree.py:377-428 / 569-626).
        Returns pix [N,3] int64 (image,row,col), already shuffled; sets result_leaf_id.
        prob=True: int(n*(1-rand)) picks per leaf are drawn with probability proportional to the local
        variance map (np.random.choice, then the uniform torch.randint picks: the reference's call
        order), nerf++-ours/tree.py:566-578."""
        ray_num_per_image = self.epoch_size / self.n_images / down_scale
        ray_num_per_pixel = ray_num_per_image / self.h / self.w
        plans = [self.leaf_plan(i, ray_num_per_pixel, last_epoch) for i in range(self.n_images)]
        if prob:
            if self.processor is None:
                from .image_process import ImageProcessor
                self.processor = ImageProcessor([self.images[i].cpu().numpy() for i in range(self.n_images)], scale=0,
                                                sharp_imgs=self._sharp_in)
        if prob and not compat_rng:
            return self._gen_pixels_prob_device(plans, last_epoch, rand)
        if compat_rng:
            pix, tags = [], []
            for ti, plan in enumerate(plans):
                boxes = None if (last_epoch or not prob) else self.leaves(ti)
                for li in range(plan.shape[0]):
                    n, r0, r1, c0, c1 = (int(v) for v in plan[li])
                    if prob:
                        x0, y0, x1, y1 = (0.0, 0.0, float(self.h), float(self.w)) if last_epoch else boxes[li]
                        n1 = int(n * (1 - rand))
                        n2 = n - n1
                        block = self.processor.sharp_imgs[ti][int(x0):int(x1), int(y0):int(y1)]
                        sel = self.processor.sample_pixels(block, n1) + torch.LongTensor([int(x0), int(y0)])
                        pix.append(torch.cat([torch.full((n1, 1), ti, dtype=torch.int64), sel], 1))
                        xs = torch.randint(r0, r1, (n2,))
                        ys = torch.randint(c0, c1, (n2,))
                        pix.append(torch.stack([torch.full((n2,), ti, dtype=torch.int64), xs, ys], 1))
                        tags.append(torch.tensor([[ti, li]], dtype=torch.float32).repeat([n, 1]))
                        continue
                    xs = torch.randint(r0, r1, (n,))
                    ys = torch.randint(c0, c1, (n,))
                    pix.append(torch.stack([torch.full((n,), ti, dtype=torch.int64), xs, ys], 1))
                    tags.append(torch.tensor([[ti, li]], dtype=torch.float32).repeat([n, 1]))
            pix = torch.cat(pix, 0)
            tags = torch.cat(tags, 0)
            perm = torch.randperm(pix.shape[0])
            pix, tags = pix[perm], tags[perm]
            self.result_leaf_id = tags
            self._tags_i32 = tags.to(torch.int32)
        else:
            dev = self.device
            allp = torch.from_numpy(np.concatenate(plans, 0)).to(dev).long()           # [L,5]
            img = torch.from_numpy(np.concatenate([np.full(p.shape[0], i) for i, p in enumerate(plans)])).to(dev)
            leaf = torch.from_numpy(np.concatenate([np.arange(p.shape[0]) for p in plans])).to(dev)
            idx = torch.repeat_interleave(torch.arange(allp.shape[0], device=dev), allp[:, 0])
            n = idx.shape[0]
            lo_r, hi_r, lo_c, hi_c = allp[idx, 1], allp[idx, 2], allp[idx, 3], allp[idx, 4]
            xs = lo_r + torch.floor(torch.rand(n, device=dev, dtype=torch.float64) * (hi_r - lo_r)).long()
            ys = lo_c + torch.floor(torch.rand(n, device=dev, dtype=torch.float64) * (hi_c - lo_c)).long()
            pix = torch.stack([img[idx], xs, ys], 1)
            tags = torch.stack([img[idx], leaf[idx]], 1)
            perm = torch.randperm(n, device=dev)
            pix, tags = pix[perm], tags[perm]
            self._tags_i32 = tags.to(torch.int32)
            self.result_leaf_id = self._tags_i32.float()
        self.result_pix = pix
        return pix

    def _gen_pixels_prob_device(self, plans, last_epoch, rand):
        """prob=True picks without the per-leaf host loop: the same distribution as
        nerf++-ours/tree.py:566-578 + image_process.py:58-93 -- per leaf int(n*(1-rand)) draws with probability
        proportional to clip(var + 1e-6, 0.01 * mean_leaf, max_leaf) over the leaf's block
        [int(x0):int(x1), int(y0):int(y1)] and n - that uniform draws -- evaluated for all leaves of all images at
        once with tensor ops on the manager's device (segmented inverse-CDF over the pixels sorted by leaf)."""
        dev = self.device
        H, W, nI = self.h, self.w, self.n_images
        sharp = torch.stack([torch.as_tensor(np.asarray(self.processor.sharp_imgs[i]), dtype=torch.float64)
                             for i in range(nI)], 0).to(dev)                      # [nI,H,W]
        # global leaf id per pixel: paint the leaf rectangles with a 2-D difference array (they partition the image)
        boxes, base = [], [0]
        for i in range(nI):
            b = np.array([[0.0, 0.0, float(H), float(W)]]) if last_epoch else self.leaves(i)
            boxes.append(np.concatenate([np.full((b.shape[0], 1), i), np.floor(b)], 1))
            base.append(base[-1] + b.shape[0])
        bx = torch.from_numpy(np.concatenate(boxes, 0)).to(dev).long()            # [L,5] img,x0,y0,x1,y1 (int())
        L = bx.shape[0]
        gid = torch.arange(1, L + 1, device=dev, dtype=torch.int64)
        diff = torch.zeros(nI, H + 1, W + 1, device=dev, dtype=torch.int64)
        for (r, c, sgn) in ((1, 2, 1), (1, 4, -1), (3, 2, -1), (3, 4, 1)):
            diff.index_put_((bx[:, 0], bx[:, r], bx[:, c]), sgn * gid, accumulate=True)
        leaf_of = diff.cumsum(1).cumsum(2)[:, :H, :W].reshape(-1) - 1          # [nI*H*W] global leaf id, -1 = uncovered
        g = sharp.reshape(-1) + 1e-6
        ok = leaf_of >= 0
        lid = torch.where(ok, leaf_of, torch.zeros_like(leaf_of))
        cnt = torch.zeros(L, device=dev, dtype=torch.float64).index_add_(0, lid[ok], torch.ones_like(g[ok]))
        ssum = torch.zeros(L, device=dev, dtype=torch.float64).index_add_(0, lid[ok], g[ok])
        gmin = 0.01 * ssum / cnt.clamp(min=1.0)
        wgt = torch.where(ok, torch.maximum(g, gmin[lid]), torch.zeros_like(g))   # the /max and /sum factors cancel
        order = torch.argsort(torch.where(ok, leaf_of, torch.full_like(leaf_of, L)), stable=True)
        cum = torch.cumsum(wgt[order], 0)
        seg_end = torch.cumsum(cnt.long(), 0)                                      # pixels of leaves 0..l
        seg_beg = seg_end - cnt.long()
        tot_before = torch.where(seg_beg > 0, cum[(seg_beg - 1).clamp(min=0)], torch.zeros_like(cum[:1]).expand(L))
        tot = cum[(seg_end - 1).clamp(min=0)] - tot_before
        # per-leaf counts (reference: n1 = int(n * (1 - rand)) weighted, n2 = n - n1 uniform)
        allp = torch.from_numpy(np.concatenate(plans, 0)).to(dev).long()          # [L,5] n, r0, r1, c0, c1
        n_all = allp[:, 0]
        n1 = torch.floor(n_all.double() * (1.0 - rand)).long()
        n1 = torch.where(cnt > 0, n1, torch.zeros_like(n1))   # (a leaf whose integer block is empty only gets uniform picks)
        n2 = n_all - n1
        img_of_leaf = bx[:, 0]
        leaf_local = torch.arange(L, device=dev) - torch.as_tensor(base[:-1], device=dev)[img_of_leaf]
        # weighted picks
        i1 = torch.repeat_interleave(torch.arange(L, device=dev), n1)
        u = torch.rand(i1.shape[0], device=dev, dtype=torch.float64)
        target = tot_before[i1] + u * tot[i1]
        pos = torch.searchsorted(cum, target, right=True)
        pos = torch.minimum(torch.maximum(pos, seg_beg[i1]), seg_end[i1] - 1)
        flat = order[pos]
        pr = (flat // W) % H
        pc = flat % W
        pix1 = torch.stack([img_of_leaf[i1], pr, pc], 1)
        # uniform picks (same integer ranges as the non-prob sampler)
        i2 = torch.repeat_interleave(torch.arange(L, device=dev), n2)
        xs = allp[i2, 1] + torch.floor(torch.rand(i2.shape[0], device=dev, dtype=torch.float64) * (allp[i2, 2] - allp[i2, 1])).long()
        ys = allp[i2, 3] + torch.floor(torch.rand(i2.shape[0], device=dev, dtype=torch.float64) * (allp[i2, 4] - allp[i2, 3])).long()
        pix2 = torch.stack([img_of_leaf[i2], xs, ys], 1)
        pix = torch.cat([pix1, pix2], 0)
        li = torch.cat([i1, i2], 0)
        tags = torch.stack([img_of_leaf[li], leaf_local[li]], 1)
        perm = torch.randperm(pix.shape[0], device=dev)
        pix, tags = pix[perm], tags[perm]
        self._tags_i32 = tags.to(torch.int32)
        self.result_leaf_id = self._tags_i32.float()
        self.result_pix = pix
        return pix

    # ---- epoch ray generation in one device launch (csrc/rays.hip epoch_rays_kernel) ------------------------------
    def epoch_plan(self, down_scale=1, last_epoch=False):
        """Host plan of the epoch: [L,7] int32 rows (image, leaf, count, row_lo, row_hi, col_lo, col_hi) of all trees,
        and the total ray count (tree.py:572-581, 598-599)."""
        ray_num_per_pixel = (self.epoch_size / self.n_images / down_scale) / self.h / self.w
        n_rays = C.c_int64(0)
        L = check(lib().fastnerf_tree_epoch_plan(self._t, float(ray_num_per_pixel), int(bool(last_epoch)), None, C.byref(n_rays)),
                  'fastnerf_tree_epoch_plan')
        plan = np.empty((L, 7), dtype=np.int32)
        check(lib().fastnerf_tree_epoch_plan(self._t, float(ray_num_per_pixel), int(bool(last_epoch)), plan.ctypes.data,
                                             C.byref(n_rays)), 'fastnerf_tree_epoch_plan')
        return plan, int(n_rays.value)

    def _weighted_tables(self, last_epoch):
        """Per-leaf inverse-CDF tables of the clipped-variance weights (image_process.py:58-72 per leaf block
        [int(x0):int(x1), int(y0):int(y1)]): pixels sorted by leaf, running sum of clip(var + 1e-6, 0.01 * leaf mean, .)
        (the /max and /sum normalisations cancel in the inverse CDF).  Cached until the trees change."""
        key = (self._version, bool(last_epoch))
        if self._wcache is not None and self._wcache[0] == key:
            return self._wcache[1]
        if self.processor is None:
            from .image_process import ImageProcessor
            self.processor = ImageProcessor([self.images[i].cpu().numpy() for i in range(self.n_images)], scale=0,
                                            sharp_imgs=self._sharp_in)
        dev = self.device
        H, W, nI = self.h, self.w, self.n_images
        sharp = torch.stack([torch.as_tensor(np.asarray(self.processor.sharp_imgs[i]), dtype=torch.float64)
                             for i in range(nI)], 0).to(dev)
        boxes = []
        for i in range(nI):
            b = np.array([[0.0, 0.0, float(H), float(W)]]) if last_epoch else self.leaves(i)
            boxes.append(np.concatenate([np.full((b.shape[0], 1), i), np.floor(b)], 1))
        bx = torch.from_numpy(np.concatenate(boxes, 0)).to(dev).long()            # [L,5] img, int(x0), int(y0), int(x1), int(y1)
        L = bx.shape[0]
        gid = torch.arange(1, L + 1, device=dev, dtype=torch.int64)
        diff = torch.zeros(nI, H + 1, W + 1, device=dev, dtype=torch.int64)       # paint the leaf rectangles (2-D difference array)
        for (r, c, sgn) in ((1, 2, 1), (1, 4, -1), (3, 2, -1), (3, 4, 1)):
            diff.index_put_((bx[:, 0], bx[:, r], bx[:, c]), sgn * gid, accumulate=True)
        leaf_of = diff.cumsum(1).cumsum(2)[:, :H, :W].reshape(-1) - 1
        g = sharp.reshape(-1) + 1e-6
        ok = leaf_of >= 0
        lid = torch.where(ok, leaf_of, torch.zeros_like(leaf_of))
        cnt = torch.zeros(L, device=dev, dtype=torch.float64).index_add_(0, lid[ok], torch.ones_like(g[ok]))
        ssum = torch.zeros(L, device=dev, dtype=torch.float64).index_add_(0, lid[ok], g[ok])
        gmin = 0.01 * ssum / cnt.clamp(min=1.0)
        wgt = torch.where(ok, torch.maximum(g, gmin[lid]), torch.zeros_like(g))
        order = torch.argsort(torch.where(ok, leaf_of, torch.full_like(leaf_of, L)), stable=True)
        cum = torch.cumsum(wgt[order], 0).contiguous()
        seg_end = torch.cumsum(cnt.long(), 0).contiguous()
        seg_beg = (seg_end - cnt.long()).contiguous()
        tabs = dict(order=order.int().contiguous(), cum=cum, seg_beg=seg_beg, seg_end=seg_end, npix=cnt.long())
        self._wcache = (key, tabs)
        return tabs

    def gen_rays_device(self, down_scale=1, last_epoch=False, prob=False, rand=1.0, seed=None, shuffle=True, want_pix=False, shard=None):
        """The whole epoch in one launch: per-leaf counts -> prefix sums -> Philox pixel draws -> rays from the poses ->
        colour gather -> (image, leaf) tags, already in shuffled order.  Same distribution as tree.py:377-428 / 569-626
        (and nerf++-ours/tree.py:548-607 with prob=True: int(n * (1 - rand)) variance-weighted picks per leaf, the rest
        uniform); the seed comes from torch's global CPU generator unless given.  Returns (rays_o, rays_d, rgb) on the
        device and sets result_leaf_tag (int32) / result_leaf_id (float32, lazily) / result_pix (when want_pix).
        shard=(rank, world, batch): only the rows this rank steps -- rows rank :: world of every batch of `batch` consecutive epoch rows
        (parallel.shard_global_rows lists them) -- bit-identical to those rows of the unsharded call with the same seed; the epoch's total
        row count is left in self.epoch_rows."""
        plan, N = self.epoch_plan(down_scale, last_epoch)
        self.epoch_rows = N
        rk, world, batch = (0, 1, max(N, 1)) if shard is None else (int(shard[0]), int(shard[1]), int(shard[2]))
        n_out = int(lib().fastnerf_epoch_shard_rows(N, batch, rk, world))
        if n_out < 0:
            raise ValueError('gen_rays_device: bad shard %r' % (shard,))
        dev = self.device
        imgs, poses = self._dev()
        offs = np.zeros(plan.shape[0] + 1, dtype=np.int64)
        np.cumsum(plan[:, 2], out=offs[1:])
        plan_d = torch.from_numpy(plan).to(dev)
        offs_d = torch.from_numpy(offs).to(dev)
        wt = [None] * 5
        if prob:
            tabs = self._weighted_tables(last_epoch)
            n1 = np.floor(plan[:, 2].astype(np.float64) * (1.0 - rand)).astype(np.int32)
            n1_d = torch.from_numpy(n1).to(dev)
            n1_d = torch.where(tabs['npix'] > 0, n1_d, torch.zeros_like(n1_d)).contiguous()   # empty integer block: uniform only
            wt = [n1_d, tabs['seg_beg'], tabs['seg_end'], tabs['order'], tabs['cum']]
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        f32 = dict(device=dev, dtype=torch.float32)
        ro, rd, rgb = torch.empty(n_out, 3, **f32), torch.empty(n_out, 3, **f32), torch.empty(n_out, 3, **f32)
        tag = torch.empty(n_out, 2, device=dev, dtype=torch.int32)
        pix = torch.empty(n_out, 3, device=dev, dtype=torch.int32) if want_pix else None
        K = self.K
        check(lib().fastnerf_epoch_rays_shard(N, plan.shape[0], ops.ptr(plan_d), ops.ptr(offs_d), ops.ptr(imgs.contiguous()), ops.ptr(poses.contiguous()),
                                              self.n_images, self.h, self.w, float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]),
                                              int(seed), int(bool(shuffle)), *[ops.ptr(t) for t in wt], batch, rk, world, ops.ptr(ro), ops.ptr(rd),
                                              ops.ptr(rgb), ops.ptr(tag), ops.ptr(pix), ops.stream()), 'fastnerf_epoch_rays_shard')
        self.result_leaf_tag = tag
        self._tags_i32 = tag
        self._leaf_id = None
        self.result_pix = pix
        return ro, rd, rgb

    def gen_rays_v3_multiThread(self, down_scale=16, prob=True, randSamp_proc=0.95, debug=False, last_epoch=False,
                                compat_rng=True, shard=None):
        """tree.py:377-428.  compat_rng=True draws pixels with torch's global CPU generator in the
        reference's exact call order (per image, per leaf: randint rows, randint cols; then one
        randperm), so a seeded run selects identical pixels; compat_rng=False generates the whole epoch
        in one device launch (gen_rays_device: same distribution).  Returns (origins, dirs, rgb) on the device."""
        if not compat_rng and self.device.type == 'cuda':
            return self.gen_rays_device(down_scale, last_epoch, prob=prob, rand=randSamp_proc, shard=shard)
        if shard is not None:
            raise ValueError('shard= needs the device generator (compat_rng=False on a GPU): the reference-order host picks exist as a whole epoch only')
        pix = self.gen_pixels(down_scale, last_epoch, compat_rng, prob=prob, rand=randSamp_proc)
        self.result_leaf_tag = self._tags_i32.to(self.device).contiguous()
        return self.gather(pix)

    def gen_rays_v3_1(self, down_scale=16, debug=False, last_epoch=False):
        """tree.py:309-375: the single-thread twin of gen_rays_v3_multiThread(prob=False) -- the same draws in the same order
        (oracle/fuzz_tree_vs_reference.py checks both against the reference)."""
        return self.gen_rays_v3_multiThread(down_scale, prob=False, debug=debug, last_epoch=last_epoch, compat_rng=True)

    # ---- adjustment -----------------------------------------------------------------------------
    def adjust_tree_from_table(self, table, thres=0.001):
        """Split rule of tree.py:629-652 driven by the reduced per-(image, leaf) table.
        table: [n_images, max_leaves] float32 values or their int32 bit patterns."""
        t = table
        if t.dtype in (torch.int32, torch.uint32):
            t = t.view(torch.float32)
        t = t.detach().float().cpu().contiguous()
        assert t.shape[0] == self.n_images
        tot = check(lib().fastnerf_tree_adjust(self._t, t.data_ptr(), int(t.shape[1]), float(thres)),
                    'fastnerf_tree_adjust')
        self._views = None
        self._version += 1
        self.cur_level += 1
        return int(tot)

    def adjust_tree_from_sumcount(self, sums, counts, thres):
        """nerf++ fork's MEAN rule (nerf++-ours/tree.py:609-632) from per-(image, leaf) fp64 sums of
        |gt-pred| and ray counts."""
        s = sums.detach().double().cpu().contiguous().view(self.n_images, -1)
        c = counts.detach().to(torch.int32).cpu().contiguous().view(self.n_images, -1)
        tot = check(lib().fastnerf_tree_adjust_mean(self._t, s.data_ptr(), c.data_ptr(), int(s.shape[1]), float(thres)),
                    'fastnerf_tree_adjust_mean')
        self._views = None
        self._version += 1
        self.cur_level += 1
        return int(tot)

    def adjust_tree(self, rgb_gt, rgb_pred, thres=0.01, debug=False):
        """tree.py:493-531: the older single-thread adjustment.  It splits a bottom-level leaf when the MEAN absolute error of
        its rays exceeds `thres` (the multiThread version the driver calls uses the max); the reduction runs on the device."""
        dev = self.device
        gt = torch.as_tensor(rgb_gt, dtype=torch.float32).to(dev).contiguous()
        pred = torch.as_tensor(rgb_pred, dtype=torch.float32).to(dev).contiguous()
        ml = self.max_leaves()
        sums = torch.zeros(self.n_images * ml, device=dev, dtype=torch.float64)
        counts = torch.zeros(self.n_images * ml, device=dev, dtype=torch.int32)
        ops.leaf_sumcount(pred, gt, self.result_leaf_tag, ml, sums, counts)
        return self.adjust_tree_from_sumcount(sums, counts, thres)

    def adjust_tree_multiThread(self, rgb_gt, rgb_pred, thres=0.001, debug=False):
        """tree.py:533-557 with the reference's arguments: the epoch's gt / predicted colours in
        the order of self.result_leaf_id.  The segmented max runs on the device."""
        dev = self.device
        gt = torch.as_tensor(rgb_gt, dtype=torch.float32).to(dev).contiguous()
        pred = torch.as_tensor(rgb_pred, dtype=torch.float32).to(dev).contiguous()
        ml = self.max_leaves()
        if self.criterion == 'mean':
            sums = torch.zeros(self.n_images * ml, device=dev, dtype=torch.float64)
            counts = torch.zeros(self.n_images * ml, device=dev, dtype=torch.int32)
            ops.leaf_sumcount(pred, gt, self.result_leaf_tag, ml, sums, counts)
            tot = self.adjust_tree_from_sumcount(sums, counts, thres)
            print('After sudivide, there are {} child nodes'.format(tot))
            return tot
        table = torch.zeros(self.n_images * ml, device=dev, dtype=torch.int32)
        ops.mse_leafmax(pred, None, gt, want_grads=False, leaf_tag=self.result_leaf_tag, max_leaves=ml, table=table)
        tot = self.adjust_tree_from_table(table.view(self.n_images, ml), thres)
        print('After sudivide, there are {} child nodes'.format(tot))
        return tot
