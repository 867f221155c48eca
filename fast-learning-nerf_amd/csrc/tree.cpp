// tree.cpp -- host-side quadtree for adaptive ray selection (no device work).
//
// Native replacement for the Python object graph of nerf-ours/tree.py (QuadTreeNode / QuadTree /
// get_children / adjust_tree_subThread).  A tree is stored as the DFS-ordered array of its leaf
// boxes: splitting a leaf replaces it in place by its four children in the reference's child
// order (tree.py:61-72), which is exactly the enumeration get_children() (tree.py:679-686)
// produces, so leaf ids are identical.  All geometry is IEEE double evaluated with the same
// expressions as the reference's Python floats so that `area == minArea` (tree.py:645) and
// `area > minArea + 0.01` (tree.py:578) take the same branches.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>
#include "../../include/fastnerf.h"

namespace fn { void set_error(const char* fmt, ...); }

struct Box { double x0, y0, x1, y1; };
static inline double area(const Box& b) { return (b.x1 - b.x0) * (b.y1 - b.y0); }
static inline void split(const Box& b, Box out[4]) {
  const double mx = (b.x0 + b.x1) / 2, my = (b.y0 + b.y1) / 2;
  out[0] = {b.x0, b.y0, mx, my};
  out[1] = {mx, b.y0, b.x1, my};
  out[2] = {b.x0, my, mx, b.y1};
  out[3] = {mx, my, b.x1, b.y1};
}

struct OneTree {
  std::vector<Box> leaves;
  double min_area;
};

struct fn_tree {
  int H, W, n_images;
  std::vector<OneTree> trees;
};

static void build(std::vector<Box>& out, const Box& b, int depth, int max_depth) {
  if (depth >= max_depth) { out.push_back(b); return; }   // recursive_subdivide, thres 0.0 (tree.py:655-676)
  Box c[4];
  split(b, c);
  for (int i = 0; i < 4; ++i) build(out, c[i], depth + 1, max_depth);
}

static double pow4(int e) { double p = 1.0; for (int i = 0; i < e; ++i) p *= 4.0; return p; }

extern "C" fn_tree* fastnerf_tree_create(int H, int W, int n_images, int max_depth) {
  if (H <= 0 || W <= 0 || n_images <= 0 || max_depth < 1 || max_depth > 12) {
    fn::set_error("fastnerf_tree_create: bad argument");
    return nullptr;
  }
  fn_tree* t = new (std::nothrow) fn_tree();
  if (!t) return nullptr;
  t->H = H; t->W = W; t->n_images = n_images;
  OneTree proto;
  build(proto.leaves, Box{0.0, 0.0, (double)H, (double)W}, 1, max_depth);
  proto.min_area = ((double)H * (double)W) / pow4(max_depth - 1);  // tree.py:94
  t->trees.assign(n_images, proto);
  return t;
}

extern "C" void fastnerf_tree_destroy(fn_tree* t) { delete t; }

extern "C" int fastnerf_tree_num_leaves(const fn_tree* t, int image) {
  if (!t || image < 0 || image >= t->n_images) { fn::set_error("fastnerf_tree_num_leaves: bad argument"); return -1; }
  return (int)t->trees[image].leaves.size();
}

extern "C" int fastnerf_tree_max_leaves(const fn_tree* t) {
  if (!t) return -1;
  size_t m = 0;
  for (const auto& tr : t->trees) m = tr.leaves.size() > m ? tr.leaves.size() : m;
  return (int)m;
}

extern "C" double fastnerf_tree_min_area(const fn_tree* t, int image) {
  if (!t || image < 0 || image >= t->n_images) return -1.0;
  return t->trees[image].min_area;
}

extern "C" int fastnerf_tree_get_leaves(const fn_tree* t, int image, double* out_host) {
  if (!t || image < 0 || image >= t->n_images || !out_host) { fn::set_error("fastnerf_tree_get_leaves: bad argument"); return -1; }
  const auto& lv = t->trees[image].leaves;
  memcpy(out_host, lv.data(), lv.size() * sizeof(Box));
  return (int)lv.size();
}

extern "C" int fastnerf_tree_set_leaves(fn_tree* t, int image, int n_leaves, const double* boxes_host, double min_area) {
  if (!t || image < 0 || image >= t->n_images || n_leaves < 1 || !boxes_host) { fn::set_error("fastnerf_tree_set_leaves: bad argument"); return -1; }
  auto& tr = t->trees[image];
  tr.leaves.resize(n_leaves);
  memcpy(tr.leaves.data(), boxes_host, (size_t)n_leaves * sizeof(Box));
  tr.min_area = min_area;
  return 0;
}

extern "C" int fastnerf_tree_leaf_plan(const fn_tree* t, int image, double ray_num_per_pixel, int last_epoch,
                                       int32_t* out_host) {
  if (!t || image < 0 || image >= t->n_images || !out_host) { fn::set_error("fastnerf_tree_leaf_plan: bad argument"); return -1; }
  const Box root{0.0, 0.0, (double)t->H, (double)t->W};
  const Box* lv;
  size_t n;
  double min_area;
  if (last_epoch) {  // fresh depth-1 tree: one leaf, minArea = H*W (tree.py:390-400)
    lv = &root; n = 1; min_area = (double)t->H * (double)t->W;
  } else {
    lv = t->trees[image].leaves.data(); n = t->trees[image].leaves.size(); min_area = t->trees[image].min_area;
  }
  for (size_t i = 0; i < n; ++i) {
    const Box& b = lv[i];
    const double a = area(b);
    const int cnt = (a > min_area + 0.01) ? 10 : (int)(a * ray_num_per_pixel);  // tree.py:578-581
    int32_t* o = out_host + i * 5;
    o[0] = cnt;
    o[1] = (int32_t)ceil(b.x0);           // tree.py:598
    o[2] = (int32_t)ceil(b.x1);
    o[3] = (int32_t)ceil(b.y0);           // tree.py:599
    o[4] = (int32_t)ceil(b.y1 - 0.01);
  }
  return (int)n;
}

// The per-leaf plans of EVERY tree in one call, concatenated in (image, leaf) order, for the device-side epoch ray
// generator (fastnerf_epoch_rays): rows of 7 int32 = image, leaf, count, row_lo, row_hi, col_lo, col_hi.  out_host may be
// NULL to query the sizes.  Returns the number of rows; *n_rays_host = sum of the counts.
extern "C" int64_t fastnerf_tree_epoch_plan(const fn_tree* t, double ray_num_per_pixel, int last_epoch, int32_t* out_host,
                                            int64_t* n_rays_host) {
  if (!t) { fn::set_error("fastnerf_tree_epoch_plan: bad argument"); return -1; }
  int64_t rows = 0, rays = 0;
  std::vector<int32_t> tmp;
  for (int img = 0; img < t->n_images; ++img) {
    const size_t n = last_epoch ? 1 : t->trees[img].leaves.size();
    tmp.resize(n * 5);
    if (fastnerf_tree_leaf_plan(t, img, ray_num_per_pixel, last_epoch, tmp.data()) < 0) return -1;
    for (size_t i = 0; i < n; ++i) {
      if (out_host) {
        int32_t* o = out_host + (rows + (int64_t)i) * 7;
        o[0] = img; o[1] = (int32_t)i;
        for (int k = 0; k < 5; ++k) o[2 + k] = tmp[i * 5 + k];
      }
      rays += tmp[i * 5];
    }
    rows += (int64_t)n;
  }
  if (n_rays_host) *n_rays_host = rays;
  return rows;
}

extern "C" int64_t fastnerf_tree_adjust(fn_tree* t, const float* table_host, int max_leaves, double thres) {
  if (!t || !table_host || max_leaves < 1) { fn::set_error("fastnerf_tree_adjust: bad argument"); return -1; }
  // torch compares the float32 loss against the Python float after casting it to float32
  const float thr = (float)thres;
  int64_t total = 0;
  for (int img = 0; img < t->n_images; ++img) {
    OneTree& tr = t->trees[img];
    if ((int)tr.leaves.size() > max_leaves) { fn::set_error("fastnerf_tree_adjust: table too narrow"); return -1; }
    const double min_before = tr.min_area;
    std::vector<Box> next;
    next.reserve(tr.leaves.size() * 2);
    for (size_t li = 0; li < tr.leaves.size(); ++li) {
      const Box& b = tr.leaves[li];
      const float mx = table_host[(size_t)img * max_leaves + li];
      if (mx > thr && area(b) == min_before) {   // tree.py:642-645
        Box c[4];
        split(b, c);
        for (int k = 0; k < 4; ++k) next.push_back(c[k]);
        if (tr.min_area == min_before) tr.min_area /= 4;  // tree.py:649-650
      } else {
        next.push_back(b);
      }
    }
    tr.leaves.swap(next);
    total += (int64_t)tr.leaves.size();
  }
  return total;
}


// nerf++ fork (nerf++-ours/tree.py:609-632): same walk, criterion `leaf_loss.mean() > thres` with the mean
// over the leaf's rays x 3 channels.  sum_host / count_host: [n_images, max_leaves] (fp64 sums of
// |gt-pred| over rays and channels, ray counts).  Leaves without rays are left alone.
extern "C" int64_t fastnerf_tree_adjust_mean(fn_tree* t, const double* sum_host, const int32_t* count_host,
                                             int max_leaves, double thres) {
  if (!t || !sum_host || !count_host || max_leaves < 1) { fn::set_error("fastnerf_tree_adjust_mean: bad argument"); return -1; }
  const float thr = (float)thres;
  int64_t total = 0;
  for (int img = 0; img < t->n_images; ++img) {
    OneTree& tr = t->trees[img];
    if ((int)tr.leaves.size() > max_leaves) { fn::set_error("fastnerf_tree_adjust_mean: table too narrow"); return -1; }
    const double min_before = tr.min_area;
    std::vector<Box> next;
    next.reserve(tr.leaves.size() * 2);
    for (size_t li = 0; li < tr.leaves.size(); ++li) {
      const Box& b = tr.leaves[li];
      const int32_t cnt = count_host[(size_t)img * max_leaves + li];
      const float mean = cnt > 0 ? (float)(sum_host[(size_t)img * max_leaves + li] / (3.0 * (double)cnt)) : 0.0f;
      if (cnt > 0 && mean > thr && area(b) == min_before) {
        Box c[4];
        split(b, c);
        for (int k = 0; k < 4; ++k) next.push_back(c[k]);
        if (tr.min_area == min_before) tr.min_area /= 4;
      } else {
        next.push_back(b);
      }
    }
    tr.leaves.swap(next);
    total += (int64_t)tr.leaves.size();
  }
  return total;
}
