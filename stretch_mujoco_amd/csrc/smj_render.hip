// Depth cameras of the batched Stretch simulator on gfx950.
//
// Replaces mujoco.Renderer.update_scene + render with depth enabled (reference: mujoco_server_camera_manager.py:127-143)
// and the post-processing of StretchCameras.post_processing_callback (enums/stretch_cameras.py:87-102 ->
// utils.limit_depth_distance, utils.py:87-91).  [MJ] the depth image of mujoco.Renderer is the distance along the optical
// axis (not along the ray), the camera looks down its -Z with +Y up, the projection is a pure-fovy pinhole when
// cam_sensorsize is 0 (it is: enums/stretch_cameras.py leaves sensor_size None), fragments nearer than znear*extent or
// farther than zfar*extent are clipped, back faces are culled.
//
// One thread per pixel ray, a 16x16 pixel tile per workgroup (coherent rays -> coherent BVH walks), blockIdx.y = env.
// The tile first stages the world poses of the visible geoms in LDS; every ray then runs over those geoms with a
// bounding-sphere reject and, for meshes, a stack-free walk of the mesh BVH (heap layout, see smj_bvh.h) in the mesh frame.
#include "smj_render.h"
#include "smj_bvh.h"   // SMJ_BVH_LEAF
#include "smj_meshlet.h"   // SMJ_MESHLET_TRIS / _VERTS

#include <stdio.h>
#include <stdlib.h>

namespace {

// Work counters of the depth kernel (a tools-only build, -DSMJ_DEPTH_STATS: the image then holds a per-ray count instead of the
// depth -- 0: geoms in the tile's list, 1: geoms past the bounding-sphere reject, 2: mesh walks, 3: inner-node visits, 4: leaves)
#ifdef SMJ_DEPTH_STATS
#define STAT(k) (stat_cnt[k] += 1.f)
#define STAT_DECL float stat_cnt[5] = {0, 0, 0, 0, 0};
#define STAT_ARG , float* stat_cnt
#define STAT_PASS , stat_cnt
#else
#define STAT(k) ((void)0)
#define STAT_DECL
#define STAT_ARG
#define STAT_PASS
#endif

constexpr int TILE_RAY = 16;    // pixels per side of a workgroup's tile when every mesh is ray cast (32 was measured slower: coarser culling outweighs the shared staging)
constexpr int TILE_RASTER = 64; // ... when the meshes and boxes are in the z-buffer already: the few primitives left make the tile's set-up (a chain of
                                // dependent global loads) the cost of the per-pixel kernel, so it is shared by 16 times as many pixels

struct RGeom {
  float pos[3], mat[9], cen[3], rbound, size[3];
  int type, rmesh, gid;
};

enum { RT_PLANE = 0, RT_SPHERE = 2, RT_CAPSULE = 3, RT_ELLIPSOID = 4, RT_CYLINDER = 5, RT_BOX = 6, RT_MESH = 7 };

__device__ __forceinline__ float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }   // 1-ulp reciprocal (the rasteriser's paths: depths are compared at 1e-4)
__device__ __forceinline__ void mulT(float* r, const float* m, const float* v) {   // r = m' v
  r[0] = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  r[1] = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  r[2] = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
}
__device__ __forceinline__ void mul(float* r, const float* m, const float* v) {    // r = m v
  r[0] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  r[1] = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  r[2] = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
}

// nearest hit of the ray o + t d with the mesh, t in [tnear, best); returns the new best.  CULL: front faces only (cameras,
// GL_CULL_FACE); otherwise both sides ([MJ] mj_rayMesh, lidar)
template <bool CULL>
__device__ __forceinline__ float ray_leaf(const float4* T, const float* o, const float* d, float tnear, float best) {
#pragma unroll
  for (int k = 0; k < SMJ_BVH_LEAF; k++) {
    const float4 v0 = T[3 * k], e1 = T[3 * k + 1], e2 = T[3 * k + 2];
    const float p[3] = {d[1] * e2.z - d[2] * e2.y, d[2] * e2.x - d[0] * e2.z, d[0] * e2.y - d[1] * e2.x};
    const float det = e1.x * p[0] + e1.y * p[1] + e1.z * p[2];
    if (CULL ? det > 1e-30f : fabsf(det) > 1e-30f) {
      const float id = 1.f / det;
      const float tv[3] = {o[0] - v0.x, o[1] - v0.y, o[2] - v0.z};
      const float u = (tv[0] * p[0] + tv[1] * p[1] + tv[2] * p[2]) * id;
      const float q[3] = {tv[1] * e1.z - tv[2] * e1.y, tv[2] * e1.x - tv[0] * e1.z, tv[0] * e1.y - tv[1] * e1.x};
      const float v = (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]) * id;
      if (u >= 0.f && v >= 0.f && u + v <= 1.f) {
        const float t = (e2.x * q[0] + e2.y * q[1] + e2.z * q[2]) * id;
        if (t >= tnear && t < best) best = t;
      }
    }
  }
  return best;
}
// slab test of one box against the ray, clipped to [tnear, best]; returns the entry distance or -1 (miss / empty box).
// NaN (0 * inf, a ray lying in a face plane of the box) compares false and reads as a miss of that slab only.
__device__ __forceinline__ float ray_box(const float4 lo, const float4 hi, const float* o, const float* inv, float tnear, float best) {
  float t0 = tnear, t1 = best;
  {
    const float a = (lo.x - o[0]) * inv[0], b = (hi.x - o[0]) * inv[0];
    t0 = fmaxf(t0, fminf(a, b)); t1 = fminf(t1, fmaxf(a, b));
  }
  {
    const float a = (lo.y - o[1]) * inv[1], b = (hi.y - o[1]) * inv[1];
    t0 = fmaxf(t0, fminf(a, b)); t1 = fminf(t1, fmaxf(a, b));
  }
  {
    const float a = (lo.z - o[2]) * inv[2], b = (hi.z - o[2]) * inv[2];
    t0 = fmaxf(t0, fminf(a, b)); t1 = fminf(t1, fmaxf(a, b));
  }
  return (t0 <= t1 && lo.x <= hi.x) ? t0 : -1.f;   // padding boxes are stored inverted: the slab test alone would accept them
}
template <bool CULL>
__device__ float ray_mesh(const DevRender& R, int rmesh, const float* o, const float* d, float tnear, float best STAT_ARG) {
  const int4 mi = R.mesh[rmesh];
  const float4* node = R.node + 4 * (long)mi.x;
  const float4* tri = R.tri + 3 * (long)mi.y;
  const int leaf0 = mi.z;
  if (leaf0 == 1) return ray_leaf<CULL>(tri, o, d, tnear, best);   // a mesh of one leaf has no inner node
  const float inv[3] = {1.f / d[0], 1.f / d[1], 1.f / d[2]};   // +-inf for axis-parallel rays: the slab test copes
  // Stack-free walk of the complete tree with a TRAIL register: bit `level` says whether the sibling of the node on the current
  // path at that level still has to be visited (0) or not (1: already visited, or its box was missed when the parent was
  // tested).  An inner node tests the boxes of both children, stored in the node itself, and enters the nearer hit one first.
  int n = 1, level = 0;
  unsigned trail = 0;
  while (true) {
    // n is an inner node
    STAT(3);
    const float4 l0 = node[4 * n], h0 = node[4 * n + 1], l1 = node[4 * n + 2], h1 = node[4 * n + 3];
    const float e0 = ray_box(l0, h0, o, inv, tnear, best), e1 = ray_box(l1, h1, o, inv, tnear, best);
    bool down = false;
    if (e0 >= 0.f || e1 >= 0.f) {
      const bool both = e0 >= 0.f && e1 >= 0.f;
      const int first = (e0 >= 0.f && (e1 < 0.f || e0 <= e1)) ? 0 : 1;
      n = 2 * n + first;
      level++;
      trail = both ? (trail & ~(1u << level)) : (trail | (1u << level));
      down = true;
    }
    while (true) {
      if (down) {
        if (n < leaf0) break;   // an inner node: test its children next
        STAT(4);
        best = ray_leaf<CULL>(tri + 3 * SMJ_BVH_LEAF * (long)(n - leaf0), o, d, tnear, best);
        down = false;
      }
      // next subtree: the sibling if it is still pending, otherwise climb
      if (level == 0) return best;
      if (!(trail & (1u << level))) { trail |= 1u << level; n ^= 1; down = true; continue; }
      n >>= 1;
      level--;
    }
  }
  return best;
}

// entering intersection of the ray with a primitive in its own frame (outside faces only), or -1
__device__ float ray_prim(int type, const float* size, const float* lp, const float* lv, float tnear) {
  if (type == RT_PLANE) {
    if (lv[2] > -1e-15f) return -1.f;
    const float x = -lp[2] / lv[2];
    if (x < tnear) return -1.f;
    const float px = lp[0] + x * lv[0], py = lp[1] + x * lv[1];
    if ((size[0] <= 0 || fabsf(px) <= size[0]) && (size[1] <= 0 || fabsf(py) <= size[1])) return x;
    return -1.f;
  }
  if (type == RT_SPHERE) {
    const float a = dot3(lv, lv), b = dot3(lv, lp), c = dot3(lp, lp) - size[0] * size[0];
    const float det = b * b - a * c;
    if (det < 1e-15f) return -1.f;
    const float x = (-b - sqrtf(det)) / a;
    return x >= tnear ? x : -1.f;
  }
  if (type == RT_CYLINDER) {
    float best = -1.f;
    const float a = lv[0] * lv[0] + lv[1] * lv[1], b = lv[0] * lp[0] + lv[1] * lp[1], c = lp[0] * lp[0] + lp[1] * lp[1] - size[0] * size[0];
    const float det = b * b - a * c;
    if (a > 1e-15f && det >= 1e-15f) {
      const float x = (-b - sqrtf(det)) / a;
      if (x >= tnear && fabsf(lp[2] + x * lv[2]) <= size[1]) best = x;
    }
    if (fabsf(lv[2]) > 1e-15f) {
      const float sg = lv[2] < 0 ? 1.f : -1.f;   // the cap facing the ray
      const float x = (sg * size[1] - lp[2]) / lv[2];
      if (x >= tnear) {
        const float px = lp[0] + x * lv[0], py = lp[1] + x * lv[1];
        if (px * px + py * py <= size[0] * size[0] && (best < 0 || x < best)) best = x;
      }
    }
    return best;
  }
  if (type == RT_BOX) {
    float best = -1.f;
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
      if (fabsf(lv[ax]) < 1e-15f) continue;
      const float sg = lv[ax] < 0 ? 1.f : -1.f;   // the face of this axis that faces the ray
      const float x = (sg * size[ax] - lp[ax]) / lv[ax];
      if (x < tnear) continue;
      const int a1 = (ax + 1) % 3, a2 = (ax + 2) % 3;
      if (fabsf(lp[a1] + x * lv[a1]) <= size[a1] && fabsf(lp[a2] + x * lv[a2]) <= size[a2] && (best < 0 || x < best)) best = x;
    }
    return best;
  }
  if (type == RT_CAPSULE) {
    float best = -1.f;
    const float a = lv[0] * lv[0] + lv[1] * lv[1], b = lv[0] * lp[0] + lv[1] * lp[1], c = lp[0] * lp[0] + lp[1] * lp[1] - size[0] * size[0];
    const float det = b * b - a * c;
    if (a > 1e-15f && det >= 1e-15f) {
      const float x = (-b - sqrtf(det)) / a;
      if (x >= tnear && fabsf(lp[2] + x * lv[2]) <= size[1]) best = x;
    }
    for (int sg = -1; sg <= 1; sg += 2) {
      const float q[3] = {lp[0], lp[1], lp[2] - sg * size[1]};
      const float aa = dot3(lv, lv), bb = dot3(lv, q), cc = dot3(q, q) - size[0] * size[0];
      const float dd = bb * bb - aa * cc;
      if (dd < 1e-15f) continue;
      const float x = (-bb - sqrtf(dd)) / aa;
      if (x >= tnear && sg * (lp[2] + x * lv[2]) >= size[1] && (best < 0 || x < best)) best = x;
    }
    return best;
  }
  if (type == RT_ELLIPSOID) {
    const float s[3] = {1.f / size[0], 1.f / size[1], 1.f / size[2]};
    const float v[3] = {lv[0] * s[0], lv[1] * s[1], lv[2] * s[2]}, p[3] = {lp[0] * s[0], lp[1] * s[1], lp[2] * s[2]};
    const float a = dot3(v, v), b = dot3(v, p), c = dot3(p, p) - 1.f;
    const float det = b * b - a * c;
    if (det < 1e-15f) return -1.f;
    const float x = (-b - sqrtf(det)) / a;
    return x >= tnear ? x : -1.f;
  }
  return -1.f;
}


// nearest intersection (either side) of a ray with a primitive in its own frame, or -1.  [MJ] mj_rayGeom
__device__ float ray_quad(float a, float b, float c) {
  float det = b * b - a * c;
  if (det < 1e-15f) return -1.f;
  det = sqrtf(det);
  const float x0 = (-b - det) / a, x1 = (-b + det) / a;
  if (x0 >= 0) return x0;
  if (x1 >= 0) return x1;
  return -1.f;
}
__device__ float ray_prim_any(int type, const float* size, const float* lp, const float* lv) {
  if (type == RT_PLANE) {
    if (lv[2] > -1e-15f) return -1.f;
    const float x = -lp[2] / lv[2];
    if (x < 0) return -1.f;
    const float px = lp[0] + x * lv[0], py = lp[1] + x * lv[1];
    if ((size[0] <= 0 || fabsf(px) <= size[0]) && (size[1] <= 0 || fabsf(py) <= size[1])) return x;
    return -1.f;
  }
  if (type == RT_SPHERE) return ray_quad(dot3(lv, lv), dot3(lv, lp), dot3(lp, lp) - size[0] * size[0]);
  if (type == RT_CYLINDER) {
    float best = -1.f;
    const float a = lv[0] * lv[0] + lv[1] * lv[1], b = lv[0] * lp[0] + lv[1] * lp[1], c = lp[0] * lp[0] + lp[1] * lp[1] - size[0] * size[0];
    if (a > 1e-15f) {
      const float x = ray_quad(a, b, c);
      if (x >= 0 && fabsf(lp[2] + x * lv[2]) <= size[1]) best = x;
    }
    if (fabsf(lv[2]) > 1e-15f)
      for (int sg = -1; sg <= 1; sg += 2) {
        const float x = (sg * size[1] - lp[2]) / lv[2];
        if (x >= 0) {
          const float px = lp[0] + x * lv[0], py = lp[1] + x * lv[1];
          if (px * px + py * py <= size[0] * size[0] && (best < 0 || x < best)) best = x;
        }
      }
    return best;
  }
  if (type == RT_BOX) {
    float best = -1.f;
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
      if (fabsf(lv[ax]) < 1e-15f) continue;
      for (int sg = -1; sg <= 1; sg += 2) {
        const float x = (sg * size[ax] - lp[ax]) / lv[ax];
        if (x < 0) continue;
        const int a1 = (ax + 1) % 3, a2 = (ax + 2) % 3;
        if (fabsf(lp[a1] + x * lv[a1]) <= size[a1] && fabsf(lp[a2] + x * lv[a2]) <= size[a2] && (best < 0 || x < best)) best = x;
      }
    }
    return best;
  }
  if (type == RT_CAPSULE) {   // the cylinder's side between the caps, then the two end spheres beyond them
    float best = -1.f;
    const float a = lv[0] * lv[0] + lv[1] * lv[1], b = lv[0] * lp[0] + lv[1] * lp[1], c = lp[0] * lp[0] + lp[1] * lp[1] - size[0] * size[0];
    if (a > 1e-15f) {
      float det = b * b - a * c;
      if (det >= 1e-15f) {
        det = sqrtf(det);
        for (int sgn = -1; sgn <= 1; sgn += 2) {
          const float x = (-b + sgn * det) / a;
          if (x >= 0 && fabsf(lp[2] + x * lv[2]) <= size[1] && (best < 0 || x < best)) best = x;
        }
      }
    }
    for (int sg = -1; sg <= 1; sg += 2) {
      const float q[3] = {lp[0], lp[1], lp[2] - sg * size[1]};
      const float aa = dot3(lv, lv), bb = dot3(lv, q), cc = dot3(q, q) - size[0] * size[0];
      float dd = bb * bb - aa * cc;
      if (dd < 1e-15f) continue;
      dd = sqrtf(dd);
      for (int sgn = -1; sgn <= 1; sgn += 2) {
        const float x = (-bb + sgn * dd) / aa;
        if (x >= 0 && sg * (lp[2] + x * lv[2]) >= size[1] && (best < 0 || x < best)) best = x;
      }
    }
    return best;
  }
  if (type == RT_ELLIPSOID) {
    const float sc[3] = {1.f / size[0], 1.f / size[1], 1.f / size[2]};
    const float v[3] = {lv[0] * sc[0], lv[1] * sc[1], lv[2] * sc[2]}, q[3] = {lp[0] * sc[0], lp[1] * sc[1], lp[2] * sc[2]};
    return ray_quad(dot3(v, v), dot3(v, q), dot3(q, q) - 1.f);
  }
  return -1.f;
}

// 2-D lidar.  Replaces the 360 rangefinder sensors mj_step evaluates (mujoco_server_sensor_manager.py:77-83 reads them):
// [MJ] mj_sensorPos -> mj_ray from each site along its +Z against every geom (all groups, alpha > 0, both sides), the site's
// own body excluded, -1 when nothing is hit, clipped to the sensor cutoff.  One workgroup per env, one thread per ray; geoms
// welded to the laser were ray-cast once by the model compiler (lidar_static), the others are staged in LDS in chunks.
__global__ __launch_bounds__(384) void smj_lidar_kernel(const DevRender R, const float* __restrict__ xpose, long ld, float* __restrict__ lidar, long lidar_ld) {
  __shared__ RGeom geoms[SMJ_RGEOM_MAX];
  const int env = blockIdx.x, tid = threadIdx.x;
  float pnt[3] = {0, 0, 0}, vec[3] = {0, 0, 1}, best = -1.f;
  const bool active = tid < R.nlidar;
  if (active) {
    const int sid = R.lidar_site[tid], b = R.site_bodyid[sid];
    float bp[3], bm[9];
    for (int k = 0; k < 3; k++) bp[k] = xpose[(12 * b + k) * ld + env];
    for (int k = 0; k < 9; k++) bm[k] = xpose[(12 * b + 3 + k) * ld + env];
    float w[3];
    mul(w, bm, R.site_pos + 3 * sid);
    for (int k = 0; k < 3; k++) pnt[k] = bp[k] + w[k];
    const float lz[3] = {R.site_mat[9 * sid + 2], R.site_mat[9 * sid + 5], R.site_mat[9 * sid + 8]};
    mul(vec, bm, lz);
    best = R.lidar_static[tid];
  }
  // the scan plane in the world frame (R.lidar_plane_*): a geom whose bounding sphere cannot reach it is not staged at all -- in the
  // empty scene 90 of the 96 run-time geoms (the arm, the wrist, the head) sit above the plane of a lidar 17 cm off the floor
  __shared__ int nkeep;
  float pp[3] = {0, 0, 0}, pn[3] = {0, 0, 0};
  const bool cull = R.lidar_plane_body >= 0;
  if (cull) {
    const int b = R.lidar_plane_body;
    float bp[3], bm[9], w[3];
    for (int k = 0; k < 3; k++) bp[k] = xpose[(12 * b + k) * ld + env];
    for (int k = 0; k < 9; k++) bm[k] = xpose[(12 * b + 3 + k) * ld + env];
    mul(w, bm, R.lidar_plane_p);
    for (int k = 0; k < 3; k++) pp[k] = bp[k] + w[k];
    mul(pn, bm, R.lidar_plane_n);
  }
  for (int base = 0; base < R.nlgeom; base += SMJ_RGEOM_MAX) {
    const int cnt_all = min(SMJ_RGEOM_MAX, R.nlgeom - base);
    __syncthreads();
    if (tid == 0) nkeep = 0;
    __syncthreads();
    if (tid < cnt_all) {
      const int g = R.lgeom[base + tid], b = R.geom_bodyid[g];
      float bp[3], bm[9];
      for (int k = 0; k < 3; k++) bp[k] = xpose[(12 * b + k) * ld + env];
      for (int k = 0; k < 9; k++) bm[k] = xpose[(12 * b + 3 + k) * ld + env];
      float w[3], cen[3];
      mul(w, bm, R.geom_bcenter + 3 * g);
      for (int k = 0; k < 3; k++) cen[k] = bp[k] + w[k];
      const int type = R.geom_type[g];
      const float rb = R.geom_rbound[g];
      bool keep = true;
      if (cull && type != RT_PLANE) {
        const float e[3] = {cen[0] - pp[0], cen[1] - pp[1], cen[2] - pp[2]};
        const float h = fabsf(dot3(e, pn)), reach = sqrtf(dot3(e, e)) + rb;   // a ray leaves the plane by at most slack + slope x length
        keep = h <= rb + R.lidar_plane_slack + R.lidar_plane_slope * reach;
      }
      if (keep) {
        RGeom& G = geoms[atomicAdd(&nkeep, 1)];   // (order does not matter: a ray keeps the nearest hit)
        mul(w, bm, R.geom_pos + 3 * g);
        for (int k = 0; k < 3; k++) { G.pos[k] = bp[k] + w[k]; G.cen[k] = cen[k]; }
        const float* lm = R.geom_mat + 9 * g;
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) G.mat[3 * i + j] = bm[3 * i] * lm[j] + bm[3 * i + 1] * lm[3 + j] + bm[3 * i + 2] * lm[6 + j];
        G.type = type;
        G.rmesh = R.geom_rmeshid[g];
        G.rbound = rb;
        for (int k = 0; k < 3; k++) G.size[k] = R.geom_size[3 * g + k];
      }
    }
    __syncthreads();
    const int cnt = nkeep;
    if (!active) continue;
    for (int i = 0; i < cnt; i++) {
      const RGeom& G = geoms[i];
      const float lim = best >= 0 ? best : 3.0e38f;
      if (G.type != RT_PLANE) {   // bounding sphere (vec is a unit vector)
        const float oc[3] = {G.cen[0] - pnt[0], G.cen[1] - pnt[1], G.cen[2] - pnt[2]};
        const float b = dot3(oc, vec), r = G.rbound;
        if (dot3(oc, oc) - b * b > r * r || b + r < 0.f || b - r > lim) continue;
      }
      const float dif[3] = {pnt[0] - G.pos[0], pnt[1] - G.pos[1], pnt[2] - G.pos[2]};
      float lp[3], lv[3];
      mulT(lp, G.mat, dif);
      mulT(lv, G.mat, vec);
      if (G.type == RT_MESH) {
        if (G.rmesh >= 0) {
          STAT_DECL
          const float x = ray_mesh<false>(R, G.rmesh, lp, lv, 0.f, lim STAT_PASS);
          if (x < lim) best = x;
        }
      } else {
        const float x = ray_prim_any(G.type, G.size, lp, lv);
        if (x >= 0 && x < lim) best = x;
      }
    }
  }
  if (active) {
    if (R.lidar_cutoff > 0 && best > R.lidar_cutoff) best = R.lidar_cutoff;
    lidar[(long)tid * lidar_ld + env] = best;
  }
}

// ------------------------------------------------------------------------------------------------ depth cameras
// Per-env staging pass (one workgroup per env, lane = visible geom), run once per render instead of once per tile:
// world poses of the geoms and of the camera, each geom's conservative SCREEN RECTANGLE (bounds of x/depth and y/depth over
// its bounding sphere, tightened by the box of a mesh) and the front-to-back order (distance from the camera to the
// bounding sphere).  Geoms that cannot be seen at all -- behind the near plane, beyond the depth range, filtered by the
// layer mode -- are dropped here.  The tile kernel then keeps a geom iff its rectangle meets the tile: four compares
// instead of a cone test plus an eight-corner frustum test, no sorting, and only the kept geoms are staged in LDS.
// Everything here is a superset test: the image does not depend on it.
//
// Workspace per env (floats): header cpos[3], cmat[9], count; then one 32-float slot per surviving geom in front-to-back
// order: RGeom (pos 3, mat 9, cen 3, rbound, size 3, type, rmesh, geom id = 22 words), rect x0 x1 y0 y1.
constexpr int WS_HDR = 16, WS_SLOT = 48, WS_RECT = 24, WS_RCG = 28, WS_TCG = 37, WS_LP = 40;   // slot: RGeom words 0..21, rect, geom -> camera rotation / translation, camera position in the geom frame
constexpr int WS_IDX = WS_HDR + SMJ_RGEOM_MAX * WS_SLOT;   // [SMJ_RGEOM_MAX] slot of visible-geom table entry i, or -1 (the rasteriser's look-up)
// Triangles the rasteriser hands to the per-pixel kernel (boxes beyond 16384 pixels, triangles cut by the near plane): per env a
// count and up to HLCAP entries of HLW words -- v0, e1, e2 in the camera frame, the pixel box u0 | u1 << 16, w0 | w1 << 16, and the
// screen polygon of the visible part (vertex count, then x y pairs in pixels) for the per-tile cull
constexpr int HLCAP = 256, HLW = 20;   // (one per thread of the per-pixel kernel's list builder)
constexpr int WS_HL = WS_IDX + SMJ_RGEOM_MAX;
constexpr int WS_STRIDE = WS_HL + 4 + HLCAP * HLW;

// mode 0: all visible geoms.  mode 1: only the geoms rigidly attached to the camera's body, with that body at the
// identity (no state is read) -- the camera-static layer.  mode 2: all other geoms.
__global__ __launch_bounds__(128) void smj_depth_prepass(const DevRender R, const float* __restrict__ xpose, long ld, int cam,
                                                         float max_depth, float* __restrict__ ws, int mode, int nocull) {
  __shared__ float cpos[3], cmat[9], gkey[SMJ_RGEOM_MAX];
  const int env = blockIdx.x, tid = threadIdx.x;
  float* W = ws + (long)env * WS_STRIDE;
  auto body_pose = [&](int b, float* bp, float* bm) {
    if (mode == 1) {   // the static layer is a property of the model: camera body at the identity, no state involved
      for (int k = 0; k < 3; k++) bp[k] = 0.f;
      for (int k = 0; k < 9; k++) bm[k] = (k % 4 == 0) ? 1.f : 0.f;
    } else {
      for (int k = 0; k < 3; k++) bp[k] = xpose[(12 * b + k) * ld + env];
      for (int k = 0; k < 9; k++) bm[k] = xpose[(12 * b + 3 + k) * ld + env];
    }
  };
  if (tid == 127) {
    float bp[3], bm[9], w[3];
    body_pose(R.cam_bodyid[cam], bp, bm);
    mul(w, bm, R.cam_pos + 3 * cam);
    for (int k = 0; k < 3; k++) cpos[k] = bp[k] + w[k];
    const float* lm = R.cam_mat + 9 * cam;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) cmat[3 * i + j] = bm[3 * i] * lm[j] + bm[3 * i + 1] * lm[3 + j] + bm[3 * i + 2] * lm[6 + j];
  }
  RGeom G{};
  if (tid < R.nrgeom) {
    const int g = R.rgeom[tid];
    float bp[3], bm[9], w[3];
    body_pose(R.geom_bodyid[g], bp, bm);
    const float lp[3] = {R.geom_pos[3 * g], R.geom_pos[3 * g + 1], R.geom_pos[3 * g + 2]};
    const float lc[3] = {R.geom_bcenter[3 * g], R.geom_bcenter[3 * g + 1], R.geom_bcenter[3 * g + 2]};
    mul(w, bm, lp);
    for (int k = 0; k < 3; k++) G.pos[k] = bp[k] + w[k];
    mul(w, bm, lc);
    for (int k = 0; k < 3; k++) G.cen[k] = bp[k] + w[k];
    const float* lm = R.geom_mat + 9 * g;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) G.mat[3 * i + j] = bm[3 * i] * lm[j] + bm[3 * i + 1] * lm[3 + j] + bm[3 * i + 2] * lm[6 + j];
    G.type = R.geom_type[g];
    G.rmesh = R.geom_rmeshid[g];
    G.rbound = R.geom_rbound[g];
    for (int k = 0; k < 3; k++) G.size[k] = R.geom_size[3 * g + k];
  }
  __syncthreads();
  const float tfar = (max_depth > 0.f && max_depth < R.zfar) ? max_depth : R.zfar;   // hits beyond the limit become 0 anyway
  const float BIG = 3.0e38f;
  float key = BIG, rect[4] = {-BIG, BIG, -BIG, BIG};
  if (tid < R.nrgeom) {
    const float dc[3] = {G.cen[0] - cpos[0], G.cen[1] - cpos[1], G.cen[2] - cpos[2]};
    if (G.type == RT_PLANE) key = -1.f;
    else {
      float pc[3];
      mulT(pc, cmat, dc);   // bounding-sphere centre in the camera frame (x right, y up, looking down -z)
      const float D = -pc[2], r = G.rbound * (1.f + 1e-5f) + 1e-6f;
      key = sqrtf(dot3(dc, dc)) - G.rbound;
      if (D + r < R.znear) key = BIG;                       // entirely behind the near plane: t >= znear never reaches it
      else if (D - r > tfar * (1.f + 1e-5f)) key = BIG;     // entirely beyond the depth range
      else if (D - r > 1e-4f && !nocull) {
        // bounds of x / depth and y / depth over the sphere: tangents from the eye to the circle (X, D, r) in the x-z plane
        const float den = D * D - r * r;
        for (int a = 0; a < 2; a++) {
          const float X = pc[a], sq = r * sqrtf(fmaxf(0.f, X * X + den));
          float lo = (X * D - sq) / den, hi = (X * D + sq) / den;
          const float e = 1e-5f * (1.f + fabsf(lo) + fabsf(hi));
          rect[2 * a] = lo - e; rect[2 * a + 1] = hi + e;
        }
        {
          // the geom's box (geom frame): long thin parts and big flat fixtures have large bounding spheres but small projections
          const float* bb = R.geom_aabb + 6 * R.rgeom[tid];
          float lo[2] = {BIG, BIG}, hi[2] = {-BIG, -BIG}, dmin = BIG;
          bool ok = true;
          for (int cidx = 0; cidx < 8; cidx++) {
            const float pad = 1e-5f;   // the box was taken before the vertices were rounded to fp32
            const float lc[3] = {bb[0] + ((cidx & 1) ? bb[3] + pad : -bb[3] - pad), bb[1] + ((cidx & 2) ? bb[4] + pad : -bb[4] - pad),
                                 bb[2] + ((cidx & 4) ? bb[5] + pad : -bb[5] - pad)};
            float w[3], q[3];
            mul(w, G.mat, lc);
            const float dw[3] = {G.pos[0] + w[0] - cpos[0], G.pos[1] + w[1] - cpos[1], G.pos[2] + w[2] - cpos[2]};
            mulT(q, cmat, dw);
            const float zd = -q[2];
            if (zd < 1e-3f) { ok = false; break; }   // a corner at or behind the eye plane: no valid projection, keep the sphere's
            dmin = fminf(dmin, zd);
            for (int a = 0; a < 2; a++) { const float v = q[a] / zd; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
          }
          if (ok) {
            for (int a = 0; a < 2; a++) {
              const float e = 1e-5f * (1.f + fabsf(lo[a]) + fabsf(hi[a]));
              rect[2 * a] = fmaxf(rect[2 * a], lo[a] - e); rect[2 * a + 1] = fminf(rect[2 * a + 1], hi[a] + e);
            }
            if (dmin > tfar * (1.f + 1e-5f) + 1e-5f) key = BIG;
          }
        }
      }
    }
    if (mode != 0 && key < 1.0e38f) {
      const bool attached = R.geom_bodyid[R.rgeom[tid]] == R.cam_bodyid[cam];
      if (attached != (mode == 1)) key = BIG;
    }
    gkey[tid] = key;
  }
  __syncthreads();
  if (tid < R.nrgeom) {
    int rank = 0, kept = 0;
    for (int j = 0; j < R.nrgeom; j++) {
      rank += (gkey[j] < key) || (gkey[j] == key && j < tid);
      kept += gkey[j] < 1.0e38f;
    }
    if (key < 1.0e38f) {
      float* S = W + WS_HDR + (long)rank * WS_SLOT;
      for (int k = 0; k < 3; k++) { S[k] = G.pos[k]; S[12 + k] = G.cen[k]; S[16 + k] = G.size[k]; }
      for (int k = 0; k < 9; k++) S[3 + k] = G.mat[k];
      S[15] = G.rbound;
      S[19] = __int_as_float(G.type); S[20] = __int_as_float(G.rmesh); S[21] = __int_as_float(R.rgeom[tid]);
      for (int k = 0; k < 4; k++) S[WS_RECT + k] = rect[k];
      // for the rasteriser: geom frame -> camera frame (x right, y up, looking down -z), and the camera in the geom frame
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) S[WS_RCG + 3 * i + j] = cmat[i] * G.mat[j] + cmat[3 + i] * G.mat[3 + j] + cmat[6 + i] * G.mat[6 + j];
      const float gc[3] = {G.pos[0] - cpos[0], G.pos[1] - cpos[1], G.pos[2] - cpos[2]}, cg[3] = {-gc[0], -gc[1], -gc[2]};
      mulT(S + WS_TCG, cmat, gc);
      mulT(S + WS_LP, G.mat, cg);
    }
    reinterpret_cast<int*>(W + WS_IDX)[tid] = key < 1.0e38f ? rank : -1;
    if (tid == 0) {
      for (int k = 0; k < 3; k++) W[k] = cpos[k];
      for (int k = 0; k < 9; k++) W[3 + k] = cmat[k];
      W[12] = __int_as_float(kept);
      reinterpret_cast<int*>(W + WS_HL)[0] = 0;
    }
  }
}

// COLOR: the RGB stand-in -- besides the nearest depth the ray keeps WHICH geom gave it; writes that geom's 8-bit albedo
// (rgb_out) and, if asked, its id (gid_out) instead of the depth.  A second instantiation: the depth cameras' loop stays as it is.
template <bool COLOR, bool RASTER>
__global__ __launch_bounds__(256) void smj_depth_kernel(const DevRender R, const float* __restrict__ ws, int width, int height,
                                                        float tan_half_fovy, float max_depth, float* __restrict__ out,
                                                        const float* __restrict__ layer, int mode,
                                                        unsigned char* __restrict__ rgb_out, int* __restrict__ gid_out) {
  __shared__ RGeom geoms[SMJ_RGEOM_MAX];
  __shared__ float cpos[3], cmat[9];
  __shared__ int wcount[2];
  __shared__ float htri[RASTER ? HLCAP : 1][9];   // the rasteriser's handed-over triangles that meet this tile (camera frame)
  __shared__ int hcount[4];
  constexpr int TILE = RASTER ? TILE_RASTER : TILE_RAY;
  const int env = blockIdx.y;
  const int tiles_x = (width + TILE - 1) / TILE;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int tid = threadIdx.x;
  const float* W = ws + (long)env * WS_STRIDE;
  const float aspect = (float)width / (float)height;
  const float tfar = (max_depth > 0.f && max_depth < R.zfar) ? max_depth : R.zfar;   // hits beyond the limit become 0 anyway
  if (tid < 3) cpos[tid] = W[tid];
  else if (tid < 12) cmat[tid - 3] = W[tid];
  // Per-tile geom list: the staged geoms (front to back) whose screen rectangle meets this tile, compacted in order.  The
  // nearest hit is found early and the per-ray bounding-sphere reject then drops most of what lies behind it.
  const int nk = __float_as_int(W[12]);
  int keep = 0;
  if (tid < nk) {
    const float* S = W + WS_HDR + (long)tid * WS_SLOT + WS_RECT;
    const float x0 = ((float)(tx * TILE) / width * 2.f - 1.f) * tan_half_fovy * aspect;
    const float x1 = ((float)(tx * TILE + TILE) / width * 2.f - 1.f) * tan_half_fovy * aspect;
    const float y1 = (1.f - (float)(ty * TILE) / height * 2.f) * tan_half_fovy;
    const float y0 = (1.f - (float)(ty * TILE + TILE) / height * 2.f) * tan_half_fovy;
    keep = S[1] >= x0 && S[0] <= x1 && S[3] >= y0 && S[2] <= y1;
    if (RASTER) {   // meshes -- and boxes, one meshlet each -- are in the z-buffer already (smj_meshlet_kernel)
      const int gt = __float_as_int(S[19 - WS_RECT]);
      if (gt == RT_MESH || (gt == RT_BOX && R.raster_boxes)) keep = 0;
    }
  }
  const unsigned long long bal = __ballot(keep);
  const int wv0 = tid >> 6, ln0 = tid & 63;
  if (wv0 < 2 && ln0 == 0) wcount[wv0] = __popcll(bal);
  __syncthreads();
  if (keep) {
    const int at = (wv0 ? wcount[0] : 0) + __popcll(bal & ((1ull << ln0) - 1ull));
    const float* S = W + WS_HDR + (long)tid * WS_SLOT;
    RGeom& G = geoms[at];
    for (int k = 0; k < 3; k++) { G.pos[k] = S[k]; G.cen[k] = S[12 + k]; G.size[k] = S[16 + k]; }
    for (int k = 0; k < 9; k++) G.mat[k] = S[3 + k];
    G.rbound = S[15];
    G.type = __float_as_int(S[19]); G.rmesh = __float_as_int(S[20]); G.gid = __float_as_int(S[21]);
  }
  const int nkeep_total = wcount[0] + wcount[1];
  int nhuge = 0;
  if (RASTER) {
    const int* H = reinterpret_cast<const int*>(W + WS_HL);
    const int nh = min(H[0], HLCAP);
    int hk = 0;
    if (tid < nh) {
      const int bu = H[4 + tid * HLW + 9], bw = H[4 + tid * HLW + 10];
      hk = (bu >> 16) >= tx * TILE && (bu & 0xffff) < tx * TILE + TILE && (bw >> 16) >= ty * TILE && (bw & 0xffff) < ty * TILE + TILE;
      if (hk) {
        // the tile's pixel centres against the convex screen polygon: out if they are all beyond one of its edges (a long thin
        // triangle meets few of the tiles of its box)
        const float* P = W + WS_HL + 4 + tid * HLW + 12;
        const int nvp = H[4 + tid * HLW + 11];
        const float rx0 = (float)(tx * TILE) + 0.5f, rx1 = (float)(tx * TILE + TILE) - 0.5f, ry0 = (float)(ty * TILE) + 0.5f, ry1 = (float)(ty * TILE + TILE) - 0.5f;
        float area = 0.f;
        for (int q = 0; q < nvp; q++) { const int q2 = q + 1 < nvp ? q + 1 : 0; area += P[2 * q] * P[2 * q2 + 1] - P[2 * q2] * P[2 * q + 1]; }
        const float sg = area >= 0.f ? 1.f : -1.f;
        for (int q = 0; q < nvp; q++) {
          const int q2 = q + 1 < nvp ? q + 1 : 0;
          const float ex = P[2 * q2] - P[2 * q], ey = P[2 * q2 + 1] - P[2 * q + 1], lim = -0.05f * (fabsf(ex) + fabsf(ey));
          const float e00 = sg * (ex * (ry0 - P[2 * q + 1]) - ey * (rx0 - P[2 * q])), e10 = sg * (ex * (ry0 - P[2 * q + 1]) - ey * (rx1 - P[2 * q]));
          const float e01 = sg * (ex * (ry1 - P[2 * q + 1]) - ey * (rx0 - P[2 * q])), e11 = sg * (ex * (ry1 - P[2 * q + 1]) - ey * (rx1 - P[2 * q]));
          if (e00 < lim && e10 < lim && e01 < lim && e11 < lim) hk = 0;
        }
      }
    }
    const unsigned long long hb = __ballot(hk);
    if (ln0 == 0) hcount[wv0] = __popcll(hb);
    __syncthreads();
    if (hk) {
      int at = __popcll(hb & ((1ull << ln0) - 1ull));
      for (int q = 0; q < wv0; q++) at += hcount[q];
      for (int k = 0; k < 9; k++) htri[at][k] = W[WS_HL + 4 + tid * HLW + k];
    }
    nhuge = hcount[0] + hcount[1] + hcount[2] + hcount[3];
  }
  __syncthreads();
  // A workgroup renders a TILE x TILE pixel tile with its 256 threads in (TILE/16)^2 rounds; each wavefront covers an 8x8
  // pixel square (not a 16x4 strip): neighbouring rays share more of their walk.  The staging and culling above are paid
  // once per tile, which is why the tile is larger than one round.
  const int wv = tid >> 6, ln = tid & 63;
  const float tnear = R.znear;
  const int ng = nkeep_total;
  for (int rd = 0; rd < (TILE / 16) * (TILE / 16); rd++) {
    const int u = tx * TILE + (rd % (TILE / 16)) * 16 + (wv & 1) * 8 + (ln & 7), v = ty * TILE + (rd / (TILE / 16)) * 16 + (wv >> 1) * 8 + (ln >> 3);
    if (u >= width || v >= height) continue;
    // pixel centre -> ray in the camera frame (x right, y up, looking down -z); parameter t = distance along the optical axis
    const float xn = RASTER ? ((u + 0.5f) * (2.f * frcp((float)width)) - 1.f) * tan_half_fovy * aspect : ((u + 0.5f) / width * 2.f - 1.f) * tan_half_fovy * aspect;
    const float yn = RASTER ? (1.f - (v + 0.5f) * (2.f * frcp((float)height))) * tan_half_fovy : (1.f - (v + 0.5f) / height * 2.f) * tan_half_fovy;
    const float dc[3] = {xn, yn, -1.f};
    float d[3], o[3] = {cpos[0], cpos[1], cpos[2]};
    mul(d, cmat, dc);
    const float dd = dot3(d, d), dl = sqrtf(dd);
    float best = tfar * (1.f + 1e-6f);
    int hit = -1;
    STAT_DECL
    if (mode == 2) best = fminf(best, layer[(long)v * width + u]);
    if (RASTER) best = fminf(best, out[mode == 1 ? (long)v * width + u : ((long)env * height + v) * width + u]);   // nearest mesh hit of this pixel
    for (int i = 0; i < ng; i++) {
      const RGeom& G = geoms[i];
      STAT(0);
      if (G.type != RT_PLANE) {   // bounding sphere
        const float oc[3] = {G.cen[0] - o[0], G.cen[1] - o[1], G.cen[2] - o[2]};
        const float b = dot3(oc, d), r = G.rbound;
        if (dot3(oc, oc) * dd - b * b > r * r * dd) continue;
        if (b + r * dl < tnear * dd || b - r * dl > best * dd) continue;
      }
      STAT(1);
      const float dif[3] = {o[0] - G.pos[0], o[1] - G.pos[1], o[2] - G.pos[2]};
      if (RASTER && G.type == RT_PLANE && G.size[0] <= 0.f && G.size[1] <= 0.f) {
        // the floor: an unbounded plane needs only the third row of the ray in its frame (normal = the frame's z axis)
        const float nz[3] = {G.mat[2], G.mat[5], G.mat[8]};
        const float den = dot3(nz, d);
        if (den < -1e-15f) {
          const float x = -dot3(nz, dif) * frcp(den);
          if (x >= tnear && x < best) best = x;
        }
        continue;
      }
      float lp[3], lv[3];
      mulT(lp, G.mat, dif);
      mulT(lv, G.mat, d);
      if (!RASTER && G.type == RT_MESH) {
        if (G.rmesh >= 0) {
          STAT(2);
          const float nb = ray_mesh<true>(R, G.rmesh, lp, lv, tnear, best STAT_PASS);
          if (COLOR && nb < best) hit = G.gid;
          best = nb;
        }
      } else {
        const float x = ray_prim(G.type, G.size, lp, lv, tnear);
        if (x >= 0 && x < best) { best = x; if (COLOR) hit = G.gid; }
      }
    }
    if (RASTER) {   // the large / near-plane-cut triangles of the meshes: the ray (origin 0, direction dc) in the camera frame
      for (int i = 0; i < nhuge; i++) {
        const float* t = htri[i];
        const float p[3] = {dc[1] * t[8] - dc[2] * t[7], dc[2] * t[6] - dc[0] * t[8], dc[0] * t[7] - dc[1] * t[6]};
        const float det = t[3] * p[0] + t[4] * p[1] + t[5] * p[2];
        if (det > 1e-30f) {
          const float id = 1.f / det;
          const float tv[3] = {-t[0], -t[1], -t[2]};
          const float uu = (tv[0] * p[0] + tv[1] * p[1] + tv[2] * p[2]) * id;
          const float q[3] = {tv[1] * t[5] - tv[2] * t[4], tv[2] * t[3] - tv[0] * t[5], tv[0] * t[4] - tv[1] * t[3]};
          const float vv = (dc[0] * q[0] + dc[1] * q[1] + dc[2] * q[2]) * id;
          if (uu >= 0.f && vv >= 0.f && uu + vv <= 1.f) {
            const float tt = (t[6] * q[0] + t[7] * q[1] + t[8] * q[2]) * id;   // dc[2] = -1: the ray parameter is the depth
            if (tt >= tnear && tt < best) best = tt;
          }
        }
      }
    }
    if (COLOR) {
      const long px = ((long)env * height + v) * width + u;
      if (best > tfar) hit = -1;
      if (gid_out) gid_out[px] = hit;
      float c[3] = {169.f / 255.f, 224.f / 255.f, 1.f};   // nothing up to the far plane: the sky of docs/getting_started.ipynb cell 14
      if (hit >= 0) for (int k = 0; k < 3; k++) c[k] = fminf(1.f, fmaxf(0.f, R.geom_rgba[4 * hit + k]));
      for (int k = 0; k < 3; k++) rgb_out[3 * px + k] = (unsigned char)(c[k] * 255.f + 0.5f);
      continue;
    }
    float z = best;
    if (mode == 1) { out[(long)v * width + u] = z; continue; }   // raw nearest depth (or just beyond the far plane)
#ifdef SMJ_DEPTH_STATS
    if (R.stat_select >= 0) { out[((long)env * height + v) * width + u] = stat_cnt[R.stat_select]; continue; }
#endif
    if (z > tfar) z = (max_depth > 0.f) ? 0.f : R.zfar;   // nothing in range: the far plane, which limit_depth_distance zeroes
    if (max_depth > 0.f && z > max_depth) z = 0.f;
    out[((long)env * height + v) * width + u] = z;
  }
}

// ------------------------------------------------------------------------------------------------ mesh rasteriser
// The robot's visual meshes are 196 k triangles: casting a ray per pixel walks ~8 BVH nodes per ray for ~1 triangle test (and a
// wave pays for its slowest ray).  Here the meshes are RASTERISED into the depth image with atomicMin, and the per-pixel kernel
// only resolves the primitives (floor, fixtures) against that z-buffer:
//  * one WAVEFRONT per meshlet (smj_meshlet.h: <= 128 triangles in BVH leaf order, <= 192 own vertices, bounding sphere, normal
//    cone).  A meshlet that is out of the depth range, off screen or turned away from the camera is dropped whole -- tested 32
//    meshlets at a time, lane = meshlet;
//  * lane = vertex: each vertex goes to the camera frame and the screen ONCE, into the wave's LDS slice;
//  * lane = triangle: screen box from the projected vertices, back faces out by the sign of the screen area.  Only a few per cent
//    of the triangles have a pixel centre in their box; those are compacted into the wave's LDS queue as barycentric edge
//    functions (relative to the box corner) + 1 / depth at the vertices (perspective-correct: 1 / depth is affine on screen);
//  * lane = candidate pixel: the queue's candidate counts are prefix-summed and every lane takes one candidate at a time (binary
//    search of the prefix for its triangle), so small and medium triangles fill the lanes whatever the mix;
//  * a triangle cut by the near plane has no projection as a whole (the shell of the head around the head camera): its part beyond
//    the plane is one or two triangles, whose screen box is what the per-pixel kernel gets together with the uncut triangle
//    (camera frame, the env's workspace) -- it tests such triangles like primitives, per tile; so it does the handful of
//    triangles with more than 16384 candidate pixels.  (List full: the cut triangles are rasterised here after all.)
// Same image as the ray caster (pixel centres, front faces only, depth along the optical axis, t >= znear) up to fp32 rounding
// and the tie-break at shared edges: tests/test_gpu_depth.py compares the two.
__device__ __forceinline__ float rl(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
constexpr int RASTER_MAX_BOX = 16384;   // candidate pixels a wave still takes itself (256 rounds of 64); larger boxes go to the per-pixel kernel
#ifdef SMJ_ZMIN_XCD
#define ZMIN(p, t) __hip_atomic_fetch_min(reinterpret_cast<unsigned*>(p), __float_as_uint(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#else
#define ZMIN(p, t) atomicMin(reinterpret_cast<unsigned*>(p), __float_as_uint(t))
#endif
struct MlEntry { float la[3], lb[3], lc[3], w[3]; int u0, w0, nu, npx; };
#ifdef SMJ_MESHLET_STATS   // tools build: work counters and shader cycles per phase, summed over the waves of a dispatch
__device__ unsigned long long g_mlstat[16];
#define MLC(i, v) mlc[i] += (unsigned long long)(v)
#define MLT(i) { const long long t1_ = __builtin_readcyclecounter(); mlc[i] += (unsigned long long)(t1_ - mlt); mlt = t1_; }
#else
#define MLC(i, v)
#define MLT(i)
#endif   // lambda_k(col, row) = la[k] col + lb[k] row + lc[k]

#ifndef SMJ_ML_WAVES
#define SMJ_ML_WAVES 4
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SMJ_ML_WAVES, 8))) void smj_meshlet_kernel(const DevRender R, float* __restrict__ ws, int width, int height,
                                                          float tan_half_fovy, float tfar_eps, float* __restrict__ zbuf, int single_image) {
  __shared__ float vx[4][SMJ_MESHLET_VERTS], vy[4][SMJ_MESHLET_VERTS], vd[4][SMJ_MESHLET_VERTS];
  __shared__ MlEntry queue[4][64];
  __shared__ int prefix[4][65];
  // grid = (env, split): the workgroups of one env have linear ids env + nenv * split -- with nenv a multiple of 8 they land on the
  // same XCD (round-robin dispatch), so that env's image and workspace stay in one L2
  const int env = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float* W = ws + (long)env * WS_STRIDE;
  const int* slot_of = reinterpret_cast<const int*>(W + WS_IDX);
  int* HL = reinterpret_cast<int*>(W + WS_HL);
  float* zimg = zbuf + (single_image ? 0 : (long)env * height * width);
  const float aspect = (float)width / (float)height, tnear = R.znear, tx = tan_half_fovy * aspect, ty = tan_half_fovy;
  const float sx = 0.5f * width / tx, sy = 0.5f * height / ty, isx = 1.f / sx, isy = 1.f / sy;
  // (1-ulp reciprocals throughout: screen boxes carry their own margin, depths are compared at 1e-4)
  const float dn = tnear * (1.f - 1e-3f);
  float* X = vx[wv]; float* Y = vy[wv]; float* Dp = vd[wv];
  MlEntry* Q = queue[wv];
  int* pre = prefix[wv];
#ifdef SMJ_MESHLET_STATS
  unsigned long long mlc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long mlt = __builtin_readcyclecounter();
  const long long mlt0 = mlt;
#endif
  // lane = meshlet: the env's (geom, meshlet) items are dealt out to its waves round-robin (neighbouring meshlets of a mesh share
  // their fate -- visible or not -- so a stride of one wave per item balances the waves), and a wave culls 64 of its items in one
  // pass of lane-parallel loads and tests: bounding sphere against depth range and frustum, normal cone against the camera.  The
  // survivors are then taken one after the other by the whole wave, the vertices and triangle words of the NEXT survivor being
  // fetched while the current one is worked on.
  const int nw = gridDim.y * 4, wid = blockIdx.y * 4 + wv;
  for (int c0 = 0; c0 < R.nmlist; c0 += 64 * nw) {
    const int item = c0 + lane * nw + wid;
    MLC(0, 1);
    int4 r0 = make_int4(0, 0, 0, 0);
    int alive_l = 0;
    float rcgl[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tcgl[3] = {0, 0, 0};
    if (item < R.nmlist) {
      MLC(1, 1);
      const int2 it = R.mlitem[item];                  // (visible-geom table entry, meshlet)
      const int sl = slot_of[it.x];
      if (sl >= 0) {                                   // else: dropped by the staging pass for this env (out of range, other layer)
        const float* S = W + WS_HDR + (long)sl * WS_SLOT;
        float lp[3];
        for (int k = 0; k < 9; k++) rcgl[k] = S[WS_RCG + k];
        for (int k = 0; k < 3; k++) { tcgl[k] = S[WS_TCG + k]; lp[k] = S[WS_LP + k]; }
        r0 = R.mlrec[3 * it.y];
        const float4 r1 = reinterpret_cast<const float4*>(R.mlrec)[3 * it.y + 1], r2 = reinterpret_cast<const float4*>(R.mlrec)[3 * it.y + 2];
        const float cen[3] = {r1.x, r1.y, r1.z}, rad = r1.w;
        float cc[3];
        mul(cc, rcgl, cen);
        const float ccx = cc[0] + tcgl[0], ccy = cc[1] + tcgl[1], D = -(cc[2] + tcgl[2]);
        alive_l = !(D + rad < tnear || D - rad > tfar_eps);
        // (a point at depth d is on screen iff |x| <= d tx, |y| <= d ty; the sphere's points have depth <= D + rad)
        if (ccx - rad > (D + rad) * tx || -ccx - rad > (D + rad) * tx || ccy - rad > (D + rad) * ty || -ccy - rad > (D + rad) * ty) alive_l = 0;
        if (r2.w > 0.f) {
          // every face normal within theta of the axis; u = direction camera -> centre, phi its angle to the axis.  A face at x is
          // turned away iff n . (x - camera) >= 0; n . (x - camera) >= dist cos(phi + theta) - rad when phi + theta < 90 deg
          const float uv[3] = {cen[0] - lp[0], cen[1] - lp[1], cen[2] - lp[2]};
          const float dist = sqrtf(uv[0] * uv[0] + uv[1] * uv[1] + uv[2] * uv[2]);
          const float cf = (r2.x * uv[0] + r2.y * uv[1] + r2.z * uv[2]) * frcp(fmaxf(dist, 1e-20f));
          const float sf = sqrtf(fmaxf(0.f, 1.f - cf * cf)), sc = sqrtf(fmaxf(0.f, 1.f - r2.w * r2.w));
          const float val = cf * r2.w - sf * sc;
          if (cf > 0.f && val > 0.f && val * dist >= rad) alive_l = 0;
        }
      }
    }
    unsigned long long alive = __ballot(alive_l);
    MLC(2, __popcll(alive));
    MLT(10)
    float4 pv[3];
    unsigned pt[SMJ_MESHLET_TRIS / 64];
#define ML_FETCH(ML) { \
      const int vb_ = __builtin_amdgcn_readlane(r0.x, ML), nv_ = __builtin_amdgcn_readlane(r0.y, ML), tb_ = __builtin_amdgcn_readlane(r0.z, ML), nt_ = __builtin_amdgcn_readlane(r0.w, ML); \
      _Pragma("unroll") for (int u = 0; u < 3; u++) { const int j = lane + 64 * u; pv[u] = R.mlvert[vb_ + (j < nv_ ? j : nv_ - 1)]; } \
      _Pragma("unroll") for (int u = 0; u < SMJ_MESHLET_TRIS / 64; u++) { const int k = lane + 64 * u; pt[u] = R.mltri[tb_ + (k < nt_ ? k : nt_ - 1)]; } }
    if (alive) { const int f_ = __ffsll((long long)alive) - 1; ML_FETCH(f_) }
    while (alive) {
    const int ml = __ffsll((long long)alive) - 1;
    alive &= alive - 1;
    const int nvert = __builtin_amdgcn_readlane(r0.y, ml), ntri = __builtin_amdgcn_readlane(r0.w, ml);
    float rcg[9], tcg[3];
#pragma unroll
    for (int k = 0; k < 9; k++) rcg[k] = rl(rcgl[k], ml);
#pragma unroll
    for (int k = 0; k < 3; k++) tcg[k] = rl(tcgl[k], ml);
    float4 cvv[3];
    unsigned ctt[SMJ_MESHLET_TRIS / 64];
#pragma unroll
    for (int u = 0; u < 3; u++) cvv[u] = pv[u];
#pragma unroll
    for (int u = 0; u < SMJ_MESHLET_TRIS / 64; u++) ctt[u] = pt[u];
    if (alive) { const int n_ = __ffsll((long long)alive) - 1; ML_FETCH(n_) }
    // lane = vertex: camera frame, then the screen (pixel coordinates); a vertex nearer than the near plane keeps its camera x, y
#pragma unroll
    for (int u = 0; u < 3; u++) {
      const int j = lane + 64 * u;
      if (j < nvert) {
        const float pvv[3] = {cvv[u].x, cvv[u].y, cvv[u].z};
        float cv[3];
        mul(cv, rcg, pvv);
        const float cxx = cv[0] + tcg[0], cyy = cv[1] + tcg[1], D = -(cv[2] + tcg[2]);
        if (D >= dn) {
          const float inv = frcp(D);
          X[j] = cxx * inv * sx + 0.5f * width; Y[j] = 0.5f * height - cyy * inv * sy;
        } else { X[j] = cxx; Y[j] = cyy; }
        Dp[j] = D;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    MLT(11)
#pragma unroll
    for (int round = 0; round < SMJ_MESHLET_TRIS / 64; round++) {
      if (round * 64 >= ntri) break;                  // uniform
      const int k = round * 64 + lane;
      // the lane's triangle as up to two SCREEN triangles (pixel coordinates + depth): itself, or -- cut by the near plane, where it
      // has no projection as a whole -- the one or two triangles of its part beyond the plane
      float qx[4] = {0, 0, 0, 0}, qy[4] = {0, 0, 0, 0}, qd[4] = {1, 1, 1, 1};
      int nsub = 0;
      if (k < ntri) {
        const unsigned t = ctt[round];
        const int ia = t & 255, ib = (t >> 8) & 255, ic = (t >> 16) & 255;
        float px[3] = {X[ia], X[ib], X[ic]}, py[3] = {Y[ia], Y[ib], Y[ic]}, pd[3] = {Dp[ia], Dp[ib], Dp[ic]};
        const float dmin = fminf(pd[0], fminf(pd[1], pd[2])), dmax = fmaxf(pd[0], fmaxf(pd[1], pd[2]));
        if (!(dmax < tnear * (1.f - 1e-4f) || dmin > tfar_eps * (1.f + 1e-4f))) {   // else: every hit fails t >= znear / loses to the initial depth
          if (dmin < dn) {
            // back to the camera frame (vertices nearer than the plane were left there); front faces only
            float cx[3], cy[3];
            for (int q = 0; q < 3; q++) {
              if (pd[q] >= dn) { cx[q] = (px[q] - 0.5f * width) * isx * pd[q]; cy[q] = (0.5f * height - py[q]) * isy * pd[q]; }
              else { cx[q] = px[q]; cy[q] = py[q]; }
            }
            const float e1[3] = {cx[1] - cx[0], cy[1] - cy[0], -(pd[1] - pd[0])}, e2[3] = {cx[2] - cx[0], cy[2] - cy[0], -(pd[2] - pd[0])};
            const float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            if (-(cx[0] * n[0] + cy[0] * n[1] - pd[0] * n[2]) > 0.f) {
              // Sutherland-Hodgman against depth >= dn, by cases: rotate the vertices so that vertex 0 is the lone one on its side
              const int in0 = pd[0] >= dn, in1 = pd[1] >= dn, in2 = pd[2] >= dn, nin = in0 + in1 + in2;
              const int lone = nin == 1 ? (in0 ? 0 : in1 ? 1 : 2) : (!in0 ? 0 : !in1 ? 1 : 2);
              float rx[3], ry[3], rd[3];
#pragma unroll
              for (int q = 0; q < 3; q++) {
                const int src = (q + lone) % 3;
                rx[q] = src == 0 ? cx[0] : src == 1 ? cx[1] : cx[2]; ry[q] = src == 0 ? cy[0] : src == 1 ? cy[1] : cy[2]; rd[q] = src == 0 ? pd[0] : src == 1 ? pd[1] : pd[2];
              }
              // the plane's crossings of the two edges at the lone vertex: 0 -> 1 and 2 -> 0
              const float s01 = (dn - rd[0]) * frcp(rd[1] - rd[0]), s20 = (dn - rd[2]) * frcp(rd[0] - rd[2]);
              const float ax = rx[0] + s01 * (rx[1] - rx[0]), ay = ry[0] + s01 * (ry[1] - ry[0]);
              const float bx = rx[2] + s20 * (rx[0] - rx[2]), by = ry[2] + s20 * (ry[0] - ry[2]);
              float vx4[4], vy4[4], vd4[4];
              if (nin == 1) {   // the lone vertex is beyond the plane: (V0, A, B)
                vx4[0] = rx[0]; vy4[0] = ry[0]; vd4[0] = rd[0]; vx4[1] = ax; vy4[1] = ay; vd4[1] = dn; vx4[2] = bx; vy4[2] = by; vd4[2] = dn;
                vx4[3] = 0; vy4[3] = 0; vd4[3] = 1;
                nsub = 1;
              } else {          // the lone vertex is cut off: (A, V1, V2, B)
                vx4[0] = ax; vy4[0] = ay; vd4[0] = dn; vx4[1] = rx[1]; vy4[1] = ry[1]; vd4[1] = rd[1]; vx4[2] = rx[2]; vy4[2] = ry[2]; vd4[2] = rd[2];
                vx4[3] = bx; vy4[3] = by; vd4[3] = dn;
                nsub = 2;
              }
#pragma unroll
              for (int q = 0; q < 4; q++) {
                const float inv = frcp(vd4[q]);
                qx[q] = vx4[q] * inv * sx + 0.5f * width; qy[q] = 0.5f * height - vy4[q] * inv * sy; qd[q] = vd4[q];
              }
              // These are few but large on screen (a few centimetres from the camera): the per-pixel kernel takes the uncut triangle,
              // tile by tile inside the cut part's box (its t >= znear test does the cutting) -- measured 7 ms per render cheaper
              // than shading them here, where one wave would sit on each for hundreds of rounds.  Only if the env's list is full do
              // the cut triangles stay in this wave (nsub).
              const float eps = 0.05f;
              const float bx0 = fminf(fminf(qx[0], qx[1]), nsub == 2 ? fminf(qx[2], qx[3]) : qx[2]), bx1 = fmaxf(fmaxf(qx[0], qx[1]), nsub == 2 ? fmaxf(qx[2], qx[3]) : qx[2]);
              const float by0 = fminf(fminf(qy[0], qy[1]), nsub == 2 ? fminf(qy[2], qy[3]) : qy[2]), by1 = fmaxf(fmaxf(qy[0], qy[1]), nsub == 2 ? fmaxf(qy[2], qy[3]) : qy[2]);
              const int hu0 = (int)ceilf(fmaxf(bx0 - 0.5f - eps, 0.f)), hu1 = (int)floorf(fminf(bx1 - 0.5f + eps, (float)(width - 1)));
              const int hw0 = (int)ceilf(fmaxf(by0 - 0.5f - eps, 0.f)), hw1 = (int)floorf(fminf(by1 - 0.5f + eps, (float)(height - 1)));
              if (hu0 > hu1 || hw0 > hw1) nsub = 0;
              else {
                const int at = atomicAdd(HL, 1);
                if (at < HLCAP) {
                  float* E = W + WS_HL + 4 + at * HLW;
                  E[0] = cx[0]; E[1] = cy[0]; E[2] = -pd[0];
                  E[3] = e1[0]; E[4] = e1[1]; E[5] = e1[2];
                  E[6] = e2[0]; E[7] = e2[1]; E[8] = e2[2];
                  reinterpret_cast<int*>(E)[9] = hu0 | (hu1 << 16); reinterpret_cast<int*>(E)[10] = hw0 | (hw1 << 16);
                  reinterpret_cast<int*>(E)[11] = nsub + 2;
                  for (int q = 0; q < 4; q++) { E[12 + 2 * q] = qx[q]; E[13 + 2 * q] = qy[q]; }
                  nsub = 0;
                }
              }
            }
          } else {
            // twice the screen area, y up: positive = counter-clockwise = facing the camera
            const float area = (px[1] - px[0]) * (py[0] - py[2]) - (px[2] - px[0]) * (py[0] - py[1]);
            if (area > 0.f) {
              nsub = 1;
              for (int q = 0; q < 3; q++) { qx[q] = px[q]; qy[q] = py[q]; qd[q] = pd[q]; }
            }
          }
        }
      }
      MLC(3, __popcll(__ballot(k < ntri))); MLC(4, __popcll(__ballot(nsub > 0)));
      MLT(12)
      for (int sub = 0; sub < 2; sub++) {
      if (__ballot(nsub > sub) == 0) break;            // uniform (the second pass: only waves that hold a quadrilateral)
      int u0 = 0, w0 = 0, nu = 0, npx = 0, lost = 0, npx_lost = 0;
      float px[3] = {qx[0], sub ? qx[2] : qx[1], sub ? qx[3] : qx[2]}, py[3] = {qy[0], sub ? qy[2] : qy[1], sub ? qy[3] : qy[2]};
      float pd[3] = {qd[0], sub ? qd[2] : qd[1], sub ? qd[3] : qd[2]};
      if (nsub > sub) {
        const float eps = 0.02f;   // projection rounding is ~1e-4 of a pixel
        u0 = (int)ceilf(fmaxf(fminf(px[0], fminf(px[1], px[2])) - 0.5f - eps, 0.f));
        const int u1 = (int)floorf(fminf(fmaxf(px[0], fmaxf(px[1], px[2])) - 0.5f + eps, (float)(width - 1)));
        w0 = (int)ceilf(fmaxf(fminf(py[0], fminf(py[1], py[2])) - 0.5f - eps, 0.f));
        const int w1 = (int)floorf(fminf(fmaxf(py[0], fmaxf(py[1], py[2])) - 0.5f + eps, (float)(height - 1)));
        if (u0 <= u1 && w0 <= w1) {
          nu = u1 - u0 + 1;
          npx = nu * (w1 - w0 + 1);
          if (npx > RASTER_MAX_BOX) {
            // too many candidates for one wave: the per-pixel kernel takes the triangle (camera frame), tile by tile
            float cx[3], cy[3];
            for (int q = 0; q < 3; q++) { cx[q] = (px[q] - 0.5f * width) * isx * pd[q]; cy[q] = (0.5f * height - py[q]) * isy * pd[q]; }
            const int at = atomicAdd(HL, 1);
            if (at < HLCAP) {
              float* E = W + WS_HL + 4 + at * HLW;
              E[0] = cx[0]; E[1] = cy[0]; E[2] = -pd[0];
              E[3] = cx[1] - cx[0]; E[4] = cy[1] - cy[0]; E[5] = -(pd[1] - pd[0]);
              E[6] = cx[2] - cx[0]; E[7] = cy[2] - cy[0]; E[8] = -(pd[2] - pd[0]);
              reinterpret_cast<int*>(E)[9] = u0 | (u1 << 16); reinterpret_cast<int*>(E)[10] = w0 | (w1 << 16);
              reinterpret_cast<int*>(E)[11] = 3;
              for (int q = 0; q < 3; q++) { E[12 + 2 * q] = px[q]; E[13 + 2 * q] = py[q]; }
            } else {   // the list is full: this wave shades the triangle itself, below (keep what that needs in the lane)
              lost = 1;
              for (int q = 0; q < 3; q++) { px[q] = cx[q]; py[q] = cy[q]; }
              npx_lost = npx;
            }
            npx = 0;
          }
        }
      }
      // hand-over list full (not seen with the Stretch meshes: boxes beyond 16384 pixels are a handful per image): one such
      // triangle at a time, lanes = the pixels of its box, the per-pixel kernel's ray / triangle test in the camera frame
      for (unsigned long long lm = __ballot(lost); lm;) {
        const int l = __ffsll((long long)lm) - 1;
        lm &= lm - 1;
        float tv0[3], te1[3], te2[3];
        tv0[0] = rl(px[0], l); tv0[1] = rl(py[0], l); tv0[2] = -rl(pd[0], l);
        te1[0] = rl(px[1], l) - tv0[0]; te1[1] = rl(py[1], l) - tv0[1]; te1[2] = -rl(pd[1], l) - tv0[2];
        te2[0] = rl(px[2], l) - tv0[0]; te2[1] = rl(py[2], l) - tv0[1]; te2[2] = -rl(pd[2], l) - tv0[2];
        const int bu0 = __builtin_amdgcn_readlane(u0, l), bw0 = __builtin_amdgcn_readlane(w0, l), bnu = __builtin_amdgcn_readlane(nu, l);
        const int bn = __builtin_amdgcn_readlane(npx_lost, l);
        for (int i = lane; i < bn; i += 64) {
          const int row = (int)(((float)i + 0.5f) / (float)bnu), col = i - row * bnu;
          const float dc[3] = {(((float)(bu0 + col) + 0.5f) / width * 2.f - 1.f) * tx, (1.f - ((float)(bw0 + row) + 0.5f) / height * 2.f) * ty, -1.f};
          const float p[3] = {dc[1] * te2[2] - dc[2] * te2[1], dc[2] * te2[0] - dc[0] * te2[2], dc[0] * te2[1] - dc[1] * te2[0]};
          const float det = te1[0] * p[0] + te1[1] * p[1] + te1[2] * p[2];
          if (det > 1e-30f) {
            const float id = 1.f / det;
            const float tv[3] = {-tv0[0], -tv0[1], -tv0[2]};
            const float uu = (tv[0] * p[0] + tv[1] * p[1] + tv[2] * p[2]) * id;
            const float q[3] = {tv[1] * te1[2] - tv[2] * te1[1], tv[2] * te1[0] - tv[0] * te1[2], tv[0] * te1[1] - tv[1] * te1[0]};
            const float vv = (dc[0] * q[0] + dc[1] * q[1] + dc[2] * q[2]) * id;
            if (uu >= 0.f && vv >= 0.f && uu + vv <= 1.f) {
              const float tt = (te2[0] * q[0] + te2[1] * q[1] + te2[2] * q[2]) * id;
              if (tt >= tnear) ZMIN(zimg + (long)(bw0 + row) * width + bu0 + col, tt);
            }
          }
        }
      }
      const unsigned long long have = __ballot(npx > 0);
      if (have == 0) continue;       // wave-uniform
      const int n = __popcll(have);
      if (npx > 0) {
        // barycentric coordinates as affine functions of (col, row) relative to the box corner's pixel centre (y down); 1 / depth
        MlEntry& E = Q[__popcll(have & ((1ull << lane) - 1ull))];
        const float ox = (float)u0 + 0.5f, oy = (float)w0 + 0.5f;
        const float x0 = px[0] - ox, y0 = py[0] - oy, x1 = px[1] - ox, y1 = py[1] - oy, x2 = px[2] - ox, y2 = py[2] - oy;
        const float A = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0), ia = frcp(A);
        // lambda_0 = E_12 / A, lambda_1 = E_20 / A, lambda_2 = E_01 / A;  E_ij(p) = (xj - xi)(py - yi) - (yj - yi)(px - xi)
        E.la[0] = -(y2 - y1) * ia; E.lb[0] = (x2 - x1) * ia; E.lc[0] = ((y2 - y1) * x1 - (x2 - x1) * y1) * ia;
        E.la[1] = -(y0 - y2) * ia; E.lb[1] = (x0 - x2) * ia; E.lc[1] = ((y0 - y2) * x2 - (x0 - x2) * y2) * ia;
        E.la[2] = -(y1 - y0) * ia; E.lb[2] = (x1 - x0) * ia; E.lc[2] = ((y1 - y0) * x0 - (x1 - x0) * y0) * ia;
        E.w[0] = frcp(pd[0]); E.w[1] = frcp(pd[1]); E.w[2] = frcp(pd[2]);
        E.u0 = u0; E.w0 = w0; E.nu = nu; E.npx = npx;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      int incl = lane < n ? Q[lane].npx : 0;
      // inclusive prefix sum over the wave on the DPP data path: within the 16-lane rows (row_shr 1, 2, 4, 8; lanes without a source add
      // 0), then the row totals across (row_bcast 15 into rows 1 and 3, row_bcast 31 into rows 2 and 3)
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false);
      if (lane == 0) pre[0] = 0;     // pre[j] = candidate pixels of entries 0..j-1
      pre[lane + 1] = incl;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int total = pre[n];
      MLC(5, n); MLC(6, total);
      MLT(13)
      for (int pidx = lane; pidx < total; pidx += 64) {
        int lo = 0, hi = n;          // the entry j with pre[j] <= pidx < pre[j + 1]
#pragma unroll
        for (int st = 0; st < 6; st++) {
          const int mid = (lo + hi) >> 1;
          if (hi - lo > 1) { if (pre[mid] <= pidx) lo = mid; else hi = mid; }
        }
        const MlEntry& E = Q[lo];
        const int i = pidx - pre[lo];
        const int row = (int)(((float)i + 0.5f) * __builtin_amdgcn_rcpf((float)E.nu));   // i / nu (the half keeps the product away from an integer)
        const int col = i - row * E.nu;
        const float fc = (float)col, fr = (float)row;
        const float l0 = E.la[0] * fc + E.lb[0] * fr + E.lc[0], l1 = E.la[1] * fc + E.lb[1] * fr + E.lc[1], l2 = E.la[2] * fc + E.lb[2] * fr + E.lc[2];
        if (l0 >= -2e-5f && l1 >= -2e-5f && l2 >= -2e-5f) {
          const float t = frcp(l0 * E.w[0] + l1 * E.w[1] + l2 * E.w[2]);
          if (t >= tnear) ZMIN(zimg + (long)(E.w0 + row) * width + E.u0 + col, t);   // t > 0: the bit patterns order like the values
        }
        MLC(7, __popcll(__ballot(l0 >= -2e-5f && l1 >= -2e-5f && l2 >= -2e-5f)));
      }
      MLT(14)
      __builtin_amdgcn_wave_barrier();   // the queue is rewritten by the next pass
      }
    }
    __builtin_amdgcn_wave_barrier();     // ... and the vertices by the next meshlet
    }
#undef ML_FETCH
  }
#ifdef SMJ_MESHLET_STATS
  mlc[15] = (unsigned long long)(__builtin_readcyclecounter() - mlt0);
  mlc[9] = 1;
  if (lane == 0) for (int i = 0; i < 16; i++) atomicAdd(&g_mlstat[i], mlc[i]);
#endif
}

__global__ void smj_fill_kernel(float* __restrict__ p, long n, float v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

}  // namespace

void smj_launch_lidar(const DevRender& r, const float* xpose, long ld, int num_envs, float* lidar, long lidar_ld, hipStream_t stream) {
  hipLaunchKernelGGL(smj_lidar_kernel, dim3(num_envs), dim3(384), 0, stream, r, xpose, ld, lidar, lidar_ld);
}
size_t smj_depth_workspace_bytes(int num_envs) { return sizeof(float) * (size_t)WS_STRIDE * (size_t)(num_envs + 1); }
void smj_launch_depth(const DevRender& r, const float* xpose, long ld, int num_envs, int cam, int width, int height,
                      float fovy_deg, float max_depth, float* out, const float* layer, int mode, float* workspace, hipStream_t stream) {
  const int tiles = ((width + TILE_RAY - 1) / TILE_RAY) * ((height + TILE_RAY - 1) / TILE_RAY);
  const int tiles_r = ((width + TILE_RASTER - 1) / TILE_RASTER) * ((height + TILE_RASTER - 1) / TILE_RASTER);
  const float th = tanf(fovy_deg * 3.14159265358979323846f / 360.f);
  const int nenv = mode == 1 ? 1 : num_envs;
  float* ws = mode == 1 ? workspace + (size_t)WS_STRIDE * num_envs : workspace;   // the static layer stages into the last block
  const float md = mode == 1 ? 0.f : max_depth;
  static const int nocull = getenv("SMJ_DEPTH_NOCULL") ? 1 : 0;   // debug: every geom in every tile (the image must not change)
  DevRender rr = r;
  rr.stat_select = -1;
#ifdef SMJ_DEPTH_STATS
  if (getenv("SMJ_DEPTH_STAT")) rr.stat_select = atoi(getenv("SMJ_DEPTH_STAT"));
#endif
  hipLaunchKernelGGL(smj_depth_prepass, dim3(nenv), dim3(128), 0, stream, rr, xpose, ld, cam, md, ws, mode, nocull);
  if (r.raster && r.nmlist > 0 && rr.stat_select < 0) {
    // meshes by rasterisation into the image (z-buffer, atomicMin), then the per-pixel kernel resolves the primitives against it
    const float tfar = (md > 0.f && md < r.zfar) ? md : r.zfar;
    const long npx = (long)nenv * width * height;
    hipLaunchKernelGGL(smj_fill_kernel, dim3(2048), dim3(256), 0, stream, out, npx, tfar * (1.f + 1e-6f));
    hipLaunchKernelGGL(smj_meshlet_kernel, dim3(nenv, r.raster_splits), dim3(256), 0, stream, rr, ws, width, height, th, tfar * (1.f + 1e-6f), out, mode == 1 ? 1 : 0);
    hipLaunchKernelGGL((smj_depth_kernel<false, true>), dim3(tiles_r, nenv), dim3(256), 0, stream, rr, ws, width, height, th, md, out, layer, mode,
                       (unsigned char*)nullptr, (int*)nullptr);
#ifdef SMJ_MESHLET_STATS
    if (mode != 1) {
      (void)hipStreamSynchronize(stream);
      unsigned long long h[16], z[16] = {0};
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_mlstat), sizeof(h));
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mlstat), z, sizeof(z));
      const char* nm[16] = {"groups", "meshlets", "alive", "triangles", "tri_front", "queued", "candidates", "fragments", "-", "waves", "cyc_cull", "cyc_vertex", "cyc_setup", "cyc_queue", "cyc_shade", "cyc_total"};
      fprintf(stderr, "meshlet stats cam %d, per env:", cam);
      for (int i = 0; i < 16; i++) if (i != 8) fprintf(stderr, " %s %.1f", nm[i], (double)h[i] / nenv);
      fprintf(stderr, "\n");
    }
#endif
    if (getenv("SMJ_DEPTH_DEBUG")) {   // tools: how many triangles the rasteriser handed over, per env
      (void)hipStreamSynchronize(stream);
      int cnt[64]; double sum = 0; int mx = 0; const int ne = nenv < 64 ? nenv : 64;
      for (int e = 0; e < ne; e++) { (void)hipMemcpy(&cnt[e], ws + (size_t)e * WS_STRIDE + WS_HL, 4, hipMemcpyDeviceToHost); sum += cnt[e]; mx = cnt[e] > mx ? cnt[e] : mx; }
      fprintf(stderr, "depth cam %d mode %d: handed-over triangles per env: mean %.1f max %d (first %d envs)\n", cam, mode, sum / ne, mx, ne);
    }
    return;
  }
  hipLaunchKernelGGL((smj_depth_kernel<false, false>), dim3(tiles, nenv), dim3(256), 0, stream, rr, ws, width, height, th, md, out, layer, mode,
                     (unsigned char*)nullptr, (int*)nullptr);
}
void smj_launch_rgb(const DevRender& r, const float* xpose, long ld, int num_envs, int cam, int width, int height, float fovy_deg,
                    unsigned char* rgb, int* gid, float* workspace, hipStream_t stream) {
  const int tiles = ((width + TILE_RAY - 1) / TILE_RAY) * ((height + TILE_RAY - 1) / TILE_RAY);
  const float th = tanf(fovy_deg * 3.14159265358979323846f / 360.f);
  hipLaunchKernelGGL(smj_depth_prepass, dim3(num_envs), dim3(128), 0, stream, r, xpose, ld, cam, 0.f, workspace, 0, 0);
  hipLaunchKernelGGL((smj_depth_kernel<true, false>), dim3(tiles, num_envs), dim3(256), 0, stream, r, workspace, width, height, th, 0.f,
                     (float*)nullptr, (const float*)nullptr, 0, rgb, gid);
}
