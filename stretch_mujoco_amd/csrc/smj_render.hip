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

#include <stdlib.h>

namespace {

constexpr int TILE = 16;   // pixels per side of a workgroup's tile (32 was measured slower: coarser culling outweighs the shared staging)

struct RGeom {
  float pos[3], mat[9], cen[3], rbound, size[3];
  int type, rmesh, gid;
};

enum { RT_PLANE = 0, RT_SPHERE = 2, RT_CAPSULE = 3, RT_ELLIPSOID = 4, RT_CYLINDER = 5, RT_BOX = 6, RT_MESH = 7 };

__device__ __forceinline__ float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
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
__device__ float ray_mesh(const DevRender& R, int rmesh, const float* o, const float* d, float tnear, float best) {
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
  for (int base = 0; base < R.nlgeom; base += SMJ_RGEOM_MAX) {
    const int cnt = min(SMJ_RGEOM_MAX, R.nlgeom - base);
    __syncthreads();
    if (tid < cnt) {
      const int g = R.lgeom[base + tid], b = R.geom_bodyid[g];
      float bp[3], bm[9];
      for (int k = 0; k < 3; k++) bp[k] = xpose[(12 * b + k) * ld + env];
      for (int k = 0; k < 9; k++) bm[k] = xpose[(12 * b + 3 + k) * ld + env];
      RGeom& G = geoms[tid];
      float w[3];
      mul(w, bm, R.geom_pos + 3 * g);
      for (int k = 0; k < 3; k++) G.pos[k] = bp[k] + w[k];
      mul(w, bm, R.geom_bcenter + 3 * g);
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
          const float x = ray_mesh<false>(R, G.rmesh, lp, lv, 0.f, lim);
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
constexpr int WS_HDR = 16, WS_SLOT = 32, WS_RECT = 24;
constexpr int WS_STRIDE = WS_HDR + SMJ_RGEOM_MAX * WS_SLOT;

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
        if (G.type == RT_MESH) {
          // a mesh's box (geom frame): long thin parts have large bounding spheres but small projections
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
    }
    if (tid == 0) {
      for (int k = 0; k < 3; k++) W[k] = cpos[k];
      for (int k = 0; k < 9; k++) W[3 + k] = cmat[k];
      W[12] = __int_as_float(kept);
    }
  }
}

// COLOR: the RGB stand-in -- besides the nearest depth the ray keeps WHICH geom gave it; writes that geom's 8-bit albedo
// (rgb_out) and, if asked, its id (gid_out) instead of the depth.  A second instantiation: the depth cameras' loop stays as it is.
template <bool COLOR>
__global__ __launch_bounds__(256) void smj_depth_kernel(const DevRender R, const float* __restrict__ ws, int width, int height,
                                                        float tan_half_fovy, float max_depth, float* __restrict__ out,
                                                        const float* __restrict__ layer, int mode,
                                                        unsigned char* __restrict__ rgb_out, int* __restrict__ gid_out) {
  __shared__ RGeom geoms[SMJ_RGEOM_MAX];
  __shared__ float cpos[3], cmat[9];
  __shared__ int wcount[2];
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
    const float xn = ((u + 0.5f) / width * 2.f - 1.f) * tan_half_fovy * aspect;
    const float yn = (1.f - (v + 0.5f) / height * 2.f) * tan_half_fovy;
    const float dc[3] = {xn, yn, -1.f};
    float d[3], o[3] = {cpos[0], cpos[1], cpos[2]};
    mul(d, cmat, dc);
    const float dd = dot3(d, d), dl = sqrtf(dd);
    float best = tfar * (1.f + 1e-6f);
    int hit = -1;
    if (mode == 2) best = fminf(best, layer[(long)v * width + u]);
    for (int i = 0; i < ng; i++) {
      const RGeom& G = geoms[i];
      if (G.type != RT_PLANE) {   // bounding sphere
        const float oc[3] = {G.cen[0] - o[0], G.cen[1] - o[1], G.cen[2] - o[2]};
        const float b = dot3(oc, d), r = G.rbound;
        if (dot3(oc, oc) * dd - b * b > r * r * dd) continue;
        if (b + r * dl < tnear * dd || b - r * dl > best * dd) continue;
      }
      const float dif[3] = {o[0] - G.pos[0], o[1] - G.pos[1], o[2] - G.pos[2]};
      float lp[3], lv[3];
      mulT(lp, G.mat, dif);
      mulT(lv, G.mat, d);
      if (G.type == RT_MESH) {
        if (G.rmesh >= 0) {
          const float nb = ray_mesh<true>(R, G.rmesh, lp, lv, tnear, best);
          if (COLOR && nb < best) hit = G.gid;
          best = nb;
        }
      } else {
        const float x = ray_prim(G.type, G.size, lp, lv, tnear);
        if (x >= 0 && x < best) { best = x; if (COLOR) hit = G.gid; }
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
    if (z > tfar) z = (max_depth > 0.f) ? 0.f : R.zfar;   // nothing in range: the far plane, which limit_depth_distance zeroes
    if (max_depth > 0.f && z > max_depth) z = 0.f;
    out[((long)env * height + v) * width + u] = z;
  }
}

}  // namespace

void smj_launch_lidar(const DevRender& r, const float* xpose, long ld, int num_envs, float* lidar, long lidar_ld, hipStream_t stream) {
  hipLaunchKernelGGL(smj_lidar_kernel, dim3(num_envs), dim3(384), 0, stream, r, xpose, ld, lidar, lidar_ld);
}
size_t smj_depth_workspace_bytes(int num_envs) { return sizeof(float) * (size_t)WS_STRIDE * (size_t)(num_envs + 1); }
void smj_launch_depth(const DevRender& r, const float* xpose, long ld, int num_envs, int cam, int width, int height,
                      float fovy_deg, float max_depth, float* out, const float* layer, int mode, float* workspace, hipStream_t stream) {
  const int tiles = ((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
  const float th = tanf(fovy_deg * 3.14159265358979323846f / 360.f);
  const int nenv = mode == 1 ? 1 : num_envs;
  float* ws = mode == 1 ? workspace + (size_t)WS_STRIDE * num_envs : workspace;   // the static layer stages into the last block
  const float md = mode == 1 ? 0.f : max_depth;
  static const int nocull = getenv("SMJ_DEPTH_NOCULL") ? 1 : 0;   // debug: every geom in every tile (the image must not change)
  hipLaunchKernelGGL(smj_depth_prepass, dim3(nenv), dim3(128), 0, stream, r, xpose, ld, cam, md, ws, mode, nocull);
  hipLaunchKernelGGL(smj_depth_kernel<false>, dim3(tiles, nenv), dim3(256), 0, stream, r, ws, width, height, th, md, out, layer, mode,
                     (unsigned char*)nullptr, (int*)nullptr);
}
void smj_launch_rgb(const DevRender& r, const float* xpose, long ld, int num_envs, int cam, int width, int height, float fovy_deg,
                    unsigned char* rgb, int* gid, float* workspace, hipStream_t stream) {
  const int tiles = ((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
  const float th = tanf(fovy_deg * 3.14159265358979323846f / 360.f);
  hipLaunchKernelGGL(smj_depth_prepass, dim3(num_envs), dim3(128), 0, stream, r, xpose, ld, cam, 0.f, workspace, 0, 0);
  hipLaunchKernelGGL(smj_depth_kernel<true>, dim3(tiles, num_envs), dim3(256), 0, stream, r, workspace, width, height, th, 0.f,
                     (float*)nullptr, (const float*)nullptr, 0, rgb, gid);
}
