// Host-side builder of the per-mesh bounding-volume hierarchies the depth renderer traverses (smj_render.hip).
//
// Layout, chosen for a stack-free traversal on the GPU: the triangles of a mesh are sorted along a Morton curve of
// their centroids and cut into leaves of four; the leaves are the last level of a COMPLETE binary tree stored as a
// 1-based heap (children of n are 2n and 2n+1), padded with empty nodes to a power of two.  "Next subtree" is then pure
// index arithmetic (sibling = n ^ 1, parent = n >> 1), so a ray needs no per-thread stack; every inner node carries the
// axis that separates its children so that a ray can visit the nearer child first (node[...][3], see below).
#pragma once
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <utility>
#include <vector>

struct SmjBvhMesh {
  int nodebase;  // heap node n of this mesh lives at node[nodebase + n]
  int tribase;   // first packed triangle of this mesh
  int leaf0;     // heap index of the first leaf (= number of leaves, a power of two)
  int ntri;      // packed (padded) triangle count = 4 * leaf0
};

struct SmjBvhSet {
  std::vector<float> node;  // [nnode][8]: lo.xyz, 0, hi.xyz, 0   (empty node: lo = +big, hi = -big)
  std::vector<float> tri;   // [ntri][12]: v0.xyz, 0, e1.xyz, 0, e2.xyz, 0  (padding triangles are all zero)
  std::vector<SmjBvhMesh> mesh;
};

static inline uint32_t smj_morton_spread(uint32_t x) {
  x &= 1023u;
  x = (x | (x << 16)) & 0x030000FFu;
  x = (x | (x << 8)) & 0x0300F00Fu;
  x = (x | (x << 4)) & 0x030C30C3u;
  x = (x | (x << 2)) & 0x09249249u;
  return x;
}

// verts: float[nv][3]; faces: int[nf][3] (indices into verts)
static inline void smj_bvh_add_mesh(SmjBvhSet& set, const float* verts, int nv, const int* faces, int nf) {
  (void)nv;
  SmjBvhMesh m{};
  const int nleaf_real = std::max(1, (nf + 3) / 4);
  int leaf0 = 1;
  while (leaf0 < nleaf_real) leaf0 <<= 1;
  m.leaf0 = leaf0;
  m.ntri = 4 * leaf0;
  m.nodebase = (int)(set.node.size() / 8);
  m.tribase = (int)(set.tri.size() / 12);
  // Morton order of the centroids
  float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
  std::vector<float> cen(3 * (size_t)nf);
  for (int f = 0; f < nf; f++)
    for (int k = 0; k < 3; k++) {
      const float c = (verts[3 * faces[3 * f] + k] + verts[3 * faces[3 * f + 1] + k] + verts[3 * faces[3 * f + 2] + k]) / 3.f;
      cen[3 * f + k] = c;
      lo[k] = std::min(lo[k], c);
      hi[k] = std::max(hi[k], c);
    }
  std::vector<std::pair<uint32_t, int>> order(nf);
  for (int f = 0; f < nf; f++) {
    uint32_t code = 0;
    for (int k = 0; k < 3; k++) {
      const float ext = hi[k] - lo[k];
      const float u = ext > 0 ? (cen[3 * f + k] - lo[k]) / ext : 0.f;
      const uint32_t q = (uint32_t)std::min(1023.f, std::max(0.f, u * 1024.f));
      code |= smj_morton_spread(q) << k;
    }
    order[f] = {code, f};
  }
  std::sort(order.begin(), order.end());
  // packed triangles
  set.tri.resize(set.tri.size() + 12 * (size_t)m.ntri, 0.f);
  float* T = set.tri.data() + 12 * (size_t)m.tribase;
  for (int i = 0; i < nf; i++) {
    const int f = order[i].second;
    const float* a = verts + 3 * faces[3 * f];
    const float* b = verts + 3 * faces[3 * f + 1];
    const float* c = verts + 3 * faces[3 * f + 2];
    for (int k = 0; k < 3; k++) {
      T[12 * i + k] = a[k];
      T[12 * i + 4 + k] = b[k] - a[k];
      T[12 * i + 8 + k] = c[k] - a[k];
    }
  }
  // boxes, bottom-up
  const int nnode = 2 * leaf0;
  set.node.resize(set.node.size() + 8 * (size_t)nnode, 0.f);
  float* N = set.node.data() + 8 * (size_t)m.nodebase;
  for (int n = 0; n < nnode; n++)
    for (int k = 0; k < 3; k++) { N[8 * n + k] = 3e38f; N[8 * n + 4 + k] = -3e38f; }
  for (int i = 0; i < nf; i++) {
    const int n = leaf0 + i / 4;
    for (int k = 0; k < 3; k++) {
      const float a = T[12 * i + k], b = a + T[12 * i + 4 + k], c = a + T[12 * i + 8 + k];
      N[8 * n + k] = std::min(N[8 * n + k], std::min(a, std::min(b, c)));
      N[8 * n + 4 + k] = std::max(N[8 * n + 4 + k], std::max(a, std::max(b, c)));
    }
  }
  for (int n = leaf0 - 1; n >= 1; n--) {
    const float* L = N + 8 * (2 * n);
    const float* Rr = N + 8 * (2 * n + 1);
    for (int k = 0; k < 3; k++) {
      N[8 * n + k] = std::min(L[k], Rr[k]);
      N[8 * n + 4 + k] = std::max(L[4 + k], Rr[4 + k]);
    }
    // visiting order hint: the axis along which the two children are furthest apart, and whether the left child is the
    // one with the larger coordinate (code = axis + 4 * swapped); a ray enters the child on its own side first
    int axis = 0, swapped = 0;
    float sep = -1.f;
    if (L[0] <= L[4] && Rr[0] <= Rr[4])
      for (int k = 0; k < 3; k++) {
        const float dc = (Rr[k] + Rr[4 + k]) - (L[k] + L[4 + k]);
        if (fabsf(dc) > sep) { sep = fabsf(dc); axis = k; swapped = dc < 0; }
      }
    N[8 * n + 3] = (float)(axis + 4 * swapped);
  }
  set.mesh.push_back(m);
}
