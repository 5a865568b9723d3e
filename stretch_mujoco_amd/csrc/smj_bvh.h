// Host-side builder of the per-mesh bounding-volume hierarchies the depth renderer traverses (smj_render.hip).
//
// Layout, chosen for a stack-free traversal on the GPU: the triangles of a mesh are partitioned by recursive object splits
// (see below) into leaves of up to SMJ_BVH_LEAF; the leaves are the last level of a COMPLETE binary tree stored as a
// 1-based heap (children of n are 2n and 2n+1), padded with empty nodes to a power of two.  "Next subtree" is then pure
// index arithmetic (sibling = n ^ 1, parent = n >> 1), so a ray needs no per-thread stack.  An inner node stores the boxes of
// BOTH its children (64 bytes): one visit tests the two boxes, orders them by entry distance and skips a missed child without
// ever fetching it; leaves have no node record of their own.
#pragma once
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <utility>
#include <vector>

#define SMJ_BVH_LEAF 2   // triangle slots per leaf (the traversal in smj_render.hip unrolls over them)

struct SmjBvhMesh {
  int nodebase;  // inner heap node n (1 <= n < leaf0) of this mesh lives at node[nodebase + n] (16 floats each)
  int tribase;   // first packed triangle of this mesh
  int leaf0;     // heap index of the first leaf (= number of leaves, a power of two)
  int ntri;      // packed (padded) triangle count = SMJ_BVH_LEAF * leaf0
};

struct SmjBvhSet {
  std::vector<float> node;  // [inner node][16]: child 2n lo.xyz,0 hi.xyz,0, child 2n+1 lo.xyz,0 hi.xyz,0  (empty child: lo = +big, hi = -big)
  std::vector<float> tri;   // [ntri][12]: v0.xyz, 0, e1.xyz, 0, e2.xyz, 0  (padding triangles are all zero)
  std::vector<SmjBvhMesh> mesh;
  std::vector<std::vector<int>> order;   // per mesh: its faces in leaf order (spatially coherent runs: the rasteriser's meshlets)
};

// verts: float[nv][3]; faces: int[nf][3] (indices into verts)
static inline void smj_bvh_add_mesh(SmjBvhSet& set, const float* verts, int nv, const int* faces, int nf) {
  (void)nv;
  SmjBvhMesh m{};
  const int nleaf_real = std::max(1, (nf + SMJ_BVH_LEAF - 1) / SMJ_BVH_LEAF);
  int leaf0 = 1;
  while (leaf0 < nleaf_real) leaf0 <<= 1;
  m.leaf0 = leaf0;
  m.ntri = SMJ_BVH_LEAF * leaf0;
  m.nodebase = (int)(set.node.size() / 16);
  m.tribase = (int)(set.tri.size() / 12);
  // Triangle order: recursive object split.  The node that covers leaves [l0, l1) splits its triangles by the binned
  // surface-area heuristic -- any plane whose two sides fit the capacity of the two halves of the leaves is admissible --
  // and hands each side to one half of its leaves.  The tree stays complete (index arithmetic instead of child pointers),
  // but its boxes follow the geometry far better than a Morton curve cut into equal runs: fewer node visits per ray.
  // Leaves may be partly filled; the empty slots are degenerate (all-zero) triangles that no ray hits.
  std::vector<float> cen(3 * (size_t)nf), tlo(3 * (size_t)nf), thi(3 * (size_t)nf);
  for (int f = 0; f < nf; f++)
    for (int k = 0; k < 3; k++) {
      const float a = verts[3 * faces[3 * f] + k], b = verts[3 * faces[3 * f + 1] + k], c = verts[3 * faces[3 * f + 2] + k];
      cen[3 * f + k] = (a + b + c) / 3.f;
      tlo[3 * f + k] = std::min(a, std::min(b, c));
      thi[3 * f + k] = std::max(a, std::max(b, c));
    }
  std::vector<int> idx(nf), slot(nf);   // slot[f] = position of triangle f in the packed array
  for (int f = 0; f < nf; f++) idx[f] = f;
  struct Job { int t0, t1, l0, l1; };
  std::vector<Job> stack;
  stack.push_back({0, nf, 0, leaf0});
  while (!stack.empty()) {
    const Job j = stack.back();
    stack.pop_back();
    const int n = j.t1 - j.t0, nl = j.l1 - j.l0;
    if (nl == 1) {
      for (int i = 0; i < n; i++) slot[idx[j.t0 + i]] = SMJ_BVH_LEAF * j.l0 + i;
      continue;
    }
    const int cap = SMJ_BVH_LEAF * (nl / 2);                                   // triangle capacity of each half
    const int lmin = std::max(n - cap, n > 1 ? 1 : 0), lmax = std::min(cap, n > 1 ? n - 1 : n);   // admissible left counts
    int left = std::min(cap, std::max(n - cap, (n + 1) / 2));        // default: as even as the capacities allow
    if (n > 1 && lmin <= lmax) {
      // binned surface-area heuristic: 32 bins of the centroid range per axis, every bin boundary whose left count is
      // admissible is a candidate; cost = area(left) * count(left) + area(right) * count(right)
      constexpr int NB = 32;
      float clo[3] = {3e38f, 3e38f, 3e38f}, chi[3] = {-3e38f, -3e38f, -3e38f};
      for (int i = j.t0; i < j.t1; i++)
        for (int k = 0; k < 3; k++) { clo[k] = std::min(clo[k], cen[3 * idx[i] + k]); chi[k] = std::max(chi[k], cen[3 * idx[i] + k]); }
      int best_axis = -1, best_left = left, best_bin = 0;
      float best_cost = 3e38f;
      for (int ax = 0; ax < 3; ax++) {
        const float ext = chi[ax] - clo[ax];
        if (!(ext > 0.f)) continue;
        int cnt[NB] = {0};
        float blo[NB][3], bhi[NB][3];
        for (int b = 0; b < NB; b++)
          for (int k = 0; k < 3; k++) { blo[b][k] = 3e38f; bhi[b][k] = -3e38f; }
        auto bin_of = [&](int f) { return std::min(NB - 1, std::max(0, (int)((cen[3 * f + ax] - clo[ax]) / ext * NB))); };
        for (int i = j.t0; i < j.t1; i++) {
          const int f = idx[i], b = bin_of(f);
          cnt[b]++;
          for (int k = 0; k < 3; k++) { blo[b][k] = std::min(blo[b][k], tlo[3 * f + k]); bhi[b][k] = std::max(bhi[b][k], thi[3 * f + k]); }
        }
        float rarea[NB];   // area of bins b..NB-1
        {
          float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
          for (int b = NB - 1; b >= 0; b--) {
            for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], blo[b][k]); hi[k] = std::max(hi[k], bhi[b][k]); }
            const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
            rarea[b] = (hi[0] >= lo[0]) ? 2.f * (dx * dy + dy * dz + dz * dx) : 0.f;
          }
        }
        float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
        int nleft = 0;
        for (int b = 0; b + 1 < NB; b++) {   // split after bin b
          nleft += cnt[b];
          for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], blo[b][k]); hi[k] = std::max(hi[k], bhi[b][k]); }
          if (nleft < lmin || nleft > lmax) continue;
          const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
          const float la = (hi[0] >= lo[0]) ? 2.f * (dx * dy + dy * dz + dz * dx) : 0.f;
          const float cost = la * nleft + rarea[b + 1] * (n - nleft);
          if (cost < best_cost) { best_cost = cost; best_axis = ax; best_left = nleft; best_bin = b + 1; }
        }
      }
      if (best_axis >= 0) {
        // partition by bin membership (same rule as the counting pass), stable in the triangle index
        const int ax = best_axis;
        const float ext = chi[ax] - clo[ax];
        auto bin_of = [&](int f) { return std::min(NB - 1, std::max(0, (int)((cen[3 * f + ax] - clo[ax]) / ext * NB))); };
        std::stable_partition(idx.begin() + j.t0, idx.begin() + j.t1, [&](int f) { return bin_of(f) < best_bin; });
        left = best_left;
      } else {
        // no admissible plane (degenerate centroids or capacity limits): even split along the widest axis
        int ax = 0;
        for (int k = 1; k < 3; k++)
          if (chi[k] - clo[k] > chi[ax] - clo[ax]) ax = k;
        std::nth_element(idx.begin() + j.t0, idx.begin() + j.t0 + left, idx.begin() + j.t1,
                         [&](int a, int b) { return cen[3 * a + ax] < cen[3 * b + ax] || (cen[3 * a + ax] == cen[3 * b + ax] && a < b); });
      }
    }
    const int lmid = j.l0 + nl / 2;
    stack.push_back({j.t0, j.t0 + left, j.l0, lmid});
    stack.push_back({j.t0 + left, j.t1, lmid, j.l1});
  }
  // packed triangles
  set.tri.resize(set.tri.size() + 12 * (size_t)m.ntri, 0.f);
  float* T = set.tri.data() + 12 * (size_t)m.tribase;
  for (int f = 0; f < nf; f++) {
    const int i = slot[f];
    const float* a = verts + 3 * faces[3 * f];
    const float* b = verts + 3 * faces[3 * f + 1];
    const float* c = verts + 3 * faces[3 * f + 2];
    for (int k = 0; k < 3; k++) {
      T[12 * i + k] = a[k];
      T[12 * i + 4 + k] = b[k] - a[k];
      T[12 * i + 8 + k] = c[k] - a[k];
    }
  }
  // boxes, bottom-up, then packed into the parents
  const int nnode = 2 * leaf0;
  std::vector<float> N(8 * (size_t)nnode);
  for (int n = 0; n < nnode; n++)
    for (int k = 0; k < 4; k++) { N[8 * n + k] = k < 3 ? 3e38f : 0.f; N[8 * n + 4 + k] = k < 3 ? -3e38f : 0.f; }
  for (int f = 0; f < nf; f++) {
    const int i = slot[f], n = leaf0 + i / SMJ_BVH_LEAF;
    for (int k = 0; k < 3; k++) {
      const float a = T[12 * i + k], b = a + T[12 * i + 4 + k], c = a + T[12 * i + 8 + k];
      N[8 * n + k] = std::min(N[8 * n + k], std::min(a, std::min(b, c)));
      N[8 * n + 4 + k] = std::max(N[8 * n + 4 + k], std::max(a, std::max(b, c)));
    }
  }
  for (int n = leaf0 - 1; n >= 1; n--)
    for (int k = 0; k < 3; k++) {
      N[8 * n + k] = std::min(N[8 * (2 * n) + k], N[8 * (2 * n + 1) + k]);
      N[8 * n + 4 + k] = std::max(N[8 * (2 * n) + 4 + k], N[8 * (2 * n + 1) + 4 + k]);
    }
  set.node.resize(set.node.size() + 16 * (size_t)leaf0, 0.f);
  float* P = set.node.data() + 16 * (size_t)m.nodebase;
  for (int n = 1; n < leaf0; n++)
    for (int k = 0; k < 8; k++) { P[16 * n + k] = N[8 * (2 * n) + k]; P[16 * n + 8 + k] = N[8 * (2 * n + 1) + k]; }
  for (int k = 0; k < 8; k++) P[k] = N[8 + k];   // slot 0 (no heap node 0): the root's own box, for hosts that want it
  set.mesh.push_back(m);
  std::vector<int> ord(nf);
  for (int f = 0; f < nf; f++) ord[f] = f;
  std::sort(ord.begin(), ord.end(), [&](int a, int b) { return slot[a] < slot[b]; });
  set.order.push_back(std::move(ord));
}
