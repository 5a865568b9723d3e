// Test harness for the host-side BVH builder (stretch_mujoco_amd/csrc/smj_bvh.h): random meshes, structural invariants, and a
// brute-force nearest-hit comparison through a host mirror of the GPU traversal's box logic (all leaves whose boxes a ray meets).
#include "smj_bvh.h"

#include <cstdio>
#include <cstdlib>
#include <map>
#include <array>

static bool tri_hit(const float* T, const float* o, const float* d, float& t) {   // Moeller-Trumbore, both faces
  const float* v0 = T; const float* e1 = T + 4; const float* e2 = T + 8;
  const float p[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
  const float det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
  if (fabsf(det) < 1e-12f) return false;
  const float s[3] = {o[0] - v0[0], o[1] - v0[1], o[2] - v0[2]};
  const float u = (s[0] * p[0] + s[1] * p[1] + s[2] * p[2]) / det;
  if (u < 0 || u > 1) return false;
  const float q[3] = {s[1] * e1[2] - s[2] * e1[1], s[2] * e1[0] - s[0] * e1[2], s[0] * e1[1] - s[1] * e1[0]};
  const float v = (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]) / det;
  if (v < 0 || u + v > 1) return false;
  t = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) / det;
  return t > 0;
}
static bool box_hit(const float* N, const float* o, const float* d) {
  if (N[0] > N[4]) return false;   // empty (padding) node
  float t0 = 0, t1 = 3e38f;
  for (int k = 0; k < 3; k++) {
    if (d[k] == 0) { if (o[k] < N[k] || o[k] > N[4 + k]) return false; continue; }
    float a = (N[k] - o[k]) / d[k], b = (N[4 + k] - o[k]) / d[k];
    if (a > b) std::swap(a, b);
    t0 = std::max(t0, a); t1 = std::min(t1, b);
  }
  return t0 <= t1 * (1 + 1e-5f) + 1e-6f;
}
// box of heap node n: stored in its parent (child 2p at floats 0..7, child 2p+1 at 8..15); the root's own box sits in slot 0
static const float* node_box(const SmjBvhSet& s, const SmjBvhMesh& m, int n) {
  return n == 1 ? &s.node[16 * (size_t)m.nodebase] : &s.node[16 * (size_t)(m.nodebase + n / 2) + 8 * (n & 1)];
}
static float walk(const SmjBvhSet& s, const SmjBvhMesh& m, int n, const float* o, const float* d) {
  const float* N = node_box(s, m, n);
  if (!box_hit(N, o, d)) return 3e38f;
  if (n >= m.leaf0) {
    float best = 3e38f, t;
    for (int k = 0; k < SMJ_BVH_LEAF; k++)
      if (tri_hit(&s.tri[12 * (size_t)(m.tribase + SMJ_BVH_LEAF * (n - m.leaf0) + k)], o, d, t)) best = std::min(best, t);
    return best;
  }
  return std::min(walk(s, m, 2 * n, o, d), walk(s, m, 2 * n + 1, o, d));
}

int main() {
  srand(7);
  for (int trial = 0; trial < 30; trial++) {
    const int nv = 20 + rand() % 1500, nf = 1 + rand() % 4000;
    std::vector<float> v(3 * nv);
    std::vector<int> f(3 * nf);
    for (auto& x : v) x = (rand() % 2000) / 1000.f - 1.f + ((trial % 3 == 1 && rand() % 9 == 0) ? 4.f : 0.f);
    if (trial % 5 == 0) for (int i = 0; i < nv; i++) v[3 * i + 2] = 0.25f;   // all centroids share one coordinate
    for (int i = 0; i < nf; i++) {
      int a = rand() % nv, b = rand() % nv, c = rand() % nv;
      if (trial % 4 == 2) { b = (a + 1) % nv; c = (a + 2) % nv; }   // small triangles
      f[3 * i] = a; f[3 * i + 1] = b; f[3 * i + 2] = c;
    }
    SmjBvhSet s;
    if (trial % 2) smj_bvh_add_mesh(s, v.data(), nv, f.data(), std::min(nf, 7));   // a second mesh in the same set: bases
    smj_bvh_add_mesh(s, v.data(), nv, f.data(), nf);
    const SmjBvhMesh& m = s.mesh.back();
    if (m.ntri != SMJ_BVH_LEAF * m.leaf0 || (m.leaf0 & (m.leaf0 - 1)) || SMJ_BVH_LEAF * m.leaf0 < nf) { printf("layout\n"); return 1; }
    // every input triangle appears exactly once; the other slots are all-zero
    std::map<std::array<float, 9>, int> want;
    for (int i = 0; i < nf; i++) {
      std::array<float, 9> k;
      for (int q = 0; q < 3; q++) { const float a = v[3 * f[3 * i] + q]; k[q] = a; k[3 + q] = v[3 * f[3 * i + 1] + q] - a; k[6 + q] = v[3 * f[3 * i + 2] + q] - a; }
      want[k]++;
    }
    int zeros = 0;
    for (int i = 0; i < m.ntri; i++) {
      const float* T = &s.tri[12 * (size_t)(m.tribase + i)];
      std::array<float, 9> k = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
      auto it = want.find(k);
      if (it != want.end() && it->second > 0) { it->second--; continue; }
      bool z = true;
      for (int q = 0; q < 12; q++) z &= T[q] == 0.f;
      if (!z) { printf("unknown triangle in slot %d\n", i); return 1; }
      zeros++;
    }
    for (auto& kv : want) if (kv.second != 0) { printf("triangle lost\n"); return 1; }
    if (zeros != m.ntri - nf) { printf("padding count\n"); return 1; }
    // boxes: triangles inside their leaf, children inside their parent, order hint in range
    for (int i = 0; i < m.ntri; i++) {
      const float* T = &s.tri[12 * (size_t)(m.tribase + i)];
      bool z = true;
      for (int q = 0; q < 12; q++) z &= T[q] == 0.f;
      if (z) continue;
      const float* N = node_box(s, m, m.leaf0 + i / SMJ_BVH_LEAF);
      for (int q = 0; q < 3; q++) {
        const float a = T[q], b = a + T[4 + q], c = a + T[8 + q];
        if (std::min(a, std::min(b, c)) < N[q] || std::max(a, std::max(b, c)) > N[4 + q]) { printf("triangle outside its leaf box\n"); return 1; }
      }
    }
    for (int n = 2; n < 2 * m.leaf0; n++) {
      const float* C = node_box(s, m, n);
      const float* P = node_box(s, m, n / 2);
      if (C[0] > C[4]) continue;
      for (int q = 0; q < 3; q++) if (C[q] < P[q] || C[4 + q] > P[4 + q]) { printf("child box outside parent\n"); return 1; }
    }
    // nearest hit through the hierarchy == brute force over all triangles
    for (int r = 0; r < 60; r++) {
      float o[3], d[3];
      for (int q = 0; q < 3; q++) { o[q] = (rand() % 6000) / 1000.f - 3.f; d[q] = (rand() % 2000) / 1000.f - 1.f; }
      float brute = 3e38f, t;
      for (int i = 0; i < m.ntri; i++)
        if (tri_hit(&s.tri[12 * (size_t)(m.tribase + i)], o, d, t)) brute = std::min(brute, t);
      const float w = walk(s, m, 1, o, d);
      if (w != brute) { printf("nearest hit differs: %g vs %g\n", w, brute); return 1; }
    }
  }
  printf("ok\n");
  return 0;
}
