// Host-side builder of the MESHLETS the depth rasteriser works on (smj_render.hip, smj_meshlet_kernel): runs of up to 128
// triangles of a mesh in BVH leaf order (spatially coherent, smj_bvh.h) with their own list of up to 192 vertices, a bounding
// sphere and a normal cone.  One wavefront takes one meshlet: it transforms each vertex ONCE (lane = vertex), then sets up its
// triangles from the projected vertices in LDS (lane = triangle) -- against three vertex transforms per triangle when
// triangles are streamed on their own -- and a meshlet that is off screen, out of range or turned away is dropped whole.
#pragma once
#include <math.h>
#include <stdint.h>

#include <map>
#include <vector>

#define SMJ_MESHLET_TRIS 128
#define SMJ_MESHLET_VERTS 192   // (128 triangles of a closed surface share ~70-100 vertices; 192 keeps the wave's LDS slice at 2.3 KB)

struct SmjMeshlet {     // 12 words
  int vbase, nvert, tbase, ntri;
  float cen[3], rad;    // bounding sphere (mesh frame)
  float axis[3], cosc;  // normal cone: every face normal within acos(cosc) of axis; cosc <= 0: no cone test
};
struct SmjMeshletSet {
  std::vector<float> vert;       // float4 per meshlet vertex (xyz, 0)
  std::vector<uint32_t> tri;     // local vertex indices a | b << 8 | c << 16
  std::vector<SmjMeshlet> let;
  std::vector<int> mesh_first, mesh_count;   // per render mesh: its meshlets
};

static inline void smj_meshlets_add_mesh(SmjMeshletSet& set, const float* verts, const int* faces, const std::vector<int>& order) {
  set.mesh_first.push_back((int)set.let.size());
  size_t i = 0;
  const size_t nf = order.size();
  while (i < nf) {
    SmjMeshlet m{};
    m.vbase = (int)(set.vert.size() / 4);
    m.tbase = (int)set.tri.size();
    std::map<int, int> local;
    std::vector<int> gl;
    float nsum[3] = {0, 0, 0};
    std::vector<float> normals;
    while (i < nf && m.ntri < SMJ_MESHLET_TRIS) {
      const int f = order[i];
      int need = 0;
      for (int k = 0; k < 3; k++) need += local.count(faces[3 * f + k]) ? 0 : 1;
      if ((int)gl.size() + need > SMJ_MESHLET_VERTS) break;
      uint32_t packed = 0;
      for (int k = 0; k < 3; k++) {
        const int v = faces[3 * f + k];
        auto it = local.find(v);
        if (it == local.end()) { it = local.emplace(v, (int)gl.size()).first; gl.push_back(v); }
        packed |= (uint32_t)it->second << (8 * k);
      }
      set.tri.push_back(packed);
      m.ntri++;
      const float* a = verts + 3 * faces[3 * f]; const float* b = verts + 3 * faces[3 * f + 1]; const float* c = verts + 3 * faces[3 * f + 2];
      const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
      float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
      for (int k = 0; k < 3; k++) nsum[k] += n[k];     // area-weighted
      const float len = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      if (len > 0.f) { for (int k = 0; k < 3; k++) normals.push_back(n[k] / len); }
      i++;
    }
    m.nvert = (int)gl.size();
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    for (int v : gl)
      for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], verts[3 * v + k]); hi[k] = fmaxf(hi[k], verts[3 * v + k]); }
    for (int k = 0; k < 3; k++) m.cen[k] = 0.5f * (lo[k] + hi[k]);
    float r2 = 0.f;
    for (int v : gl) {
      float d2 = 0.f;
      for (int k = 0; k < 3; k++) { const float d = verts[3 * v + k] - m.cen[k]; d2 += d * d; }
      r2 = fmaxf(r2, d2);
      set.vert.push_back(verts[3 * v]); set.vert.push_back(verts[3 * v + 1]); set.vert.push_back(verts[3 * v + 2]); set.vert.push_back(0.f);
    }
    m.rad = sqrtf(r2) * (1.f + 1e-5f) + 1e-7f;
    const float nl = sqrtf(nsum[0] * nsum[0] + nsum[1] * nsum[1] + nsum[2] * nsum[2]);
    m.cosc = -1.f;
    if (nl > 0.f && !normals.empty()) {
      float mind = 1.f;
      for (int k = 0; k < 3; k++) m.axis[k] = nsum[k] / nl;
      for (size_t q = 0; q < normals.size(); q += 3) mind = fminf(mind, normals[q] * m.axis[0] + normals[q + 1] * m.axis[1] + normals[q + 2] * m.axis[2]);
      m.cosc = mind - 1e-4f;   // (a degenerate face has no normal and cannot be hit: it does not widen the cone)
    }
    set.let.push_back(m);
  }
  set.mesh_count.push_back((int)set.let.size() - set.mesh_first.back());
}

// A box geom as one meshlet (8 vertices, 12 outward-facing triangles, geom frame): the fixtures of a kitchen are boxes, and a box
// that is rasterised with the meshes costs the per-pixel kernel nothing.  Returns the meshlet's index.
static inline int smj_meshlets_add_box(SmjMeshletSet& set, const float* half) {
  SmjMeshlet m{};
  m.vbase = (int)(set.vert.size() / 4);
  m.tbase = (int)set.tri.size();
  m.nvert = 8; m.ntri = 12;
  for (int i = 0; i < 8; i++) {
    set.vert.push_back((i & 1) ? half[0] : -half[0]); set.vert.push_back((i & 2) ? half[1] : -half[1]);
    set.vert.push_back((i & 4) ? half[2] : -half[2]); set.vert.push_back(0.f);
  }
  // faces -x +x -y +y -z +z, counter-clockwise seen from outside
  static const int F[12][3] = {{0, 4, 6}, {0, 6, 2}, {1, 3, 7}, {1, 7, 5}, {0, 1, 5}, {0, 5, 4}, {2, 6, 7}, {2, 7, 3}, {0, 2, 3}, {0, 3, 1}, {4, 5, 7}, {4, 7, 6}};
  for (int f = 0; f < 12; f++) set.tri.push_back((uint32_t)F[f][0] | ((uint32_t)F[f][1] << 8) | ((uint32_t)F[f][2] << 16));
  m.cen[0] = m.cen[1] = m.cen[2] = 0.f;
  m.rad = sqrtf(half[0] * half[0] + half[1] * half[1] + half[2] * half[2]) * (1.f + 1e-5f) + 1e-7f;
  m.cosc = -1.f;   // faces point everywhere: no cone
  set.let.push_back(m);
  return (int)set.let.size() - 1;
}
