// Depth-camera renderer of the batched Stretch simulator (smj_render.hip): one thread per pixel ray, cast against the
// geoms a MuJoCo camera draws (geom groups 0-2, alpha > 0), mesh geoms through per-mesh BVHs (smj_bvh.h).
#pragma once
#include <hip/hip_runtime.h>

#define SMJ_RGEOM_MAX 128

struct DevRender {
  int nrgeom, ncam, nbody;
  float znear, zfar;               // absolute clip distances = vis.map.znear/zfar * stat.extent
  const int* rgeom;                // [nrgeom] geom ids a camera can see
  const int* geom_type;
  const int* geom_bodyid;
  const int* geom_rmeshid;
  const float* geom_pos;           // [ngeom][3]  (fused-body frame)
  const float* geom_mat;           // [ngeom][9]
  const float* geom_size;          // [ngeom][3]
  const float* geom_rbound;        // [ngeom]
  const float* geom_bcenter;       // [ngeom][3]  bounding-sphere centre, body frame
  const float* geom_aabb;          // [ngeom][6]  box in the geom frame: centre, half sizes
  const float* geom_rgba;          // [ngeom][4]  colour of the RGB stand-in (geom / material rgba), or null
  const int* cam_bodyid;           // [ncam]
  const float* cam_pos;            // [ncam][3]
  const float* cam_mat;            // [ncam][9]
  const float4* node;              // BVH inner nodes, four float4 each: the boxes of both children (smj_bvh.h)
  const float4* tri;               // packed triangles, three float4 per triangle
  const int4* mesh;                // per render mesh: nodebase, tribase, leaf0, ntri
  // 2-D lidar (rangefinder sites on the laser body)
  int nlidar, nlgeom;
  float lidar_cutoff;
  const int* lgeom;                // [nlgeom] geoms tested at run time (everything not welded to the laser, alpha > 0)
  const int* lidar_site;           // [nlidar] site ids, ray order
  const int* site_bodyid;
  const float* site_pos;           // [nsite][3]
  const float* site_mat;           // [nsite][9]; the ray runs along the site's +Z
  const float* lidar_static;       // [nlidar] distance to the geoms welded to the laser (ray-cast by the model compiler), -1 none
  // the scan plane, when every ray starts on one body and runs in one plane of it (setup_render checks): body, a point and the unit
  // normal in the body frame, the rays' largest offset from the plane at their origins and their largest slope out of it -- the
  // lidar kernel drops, per env, the geoms whose bounding sphere cannot reach the plane (a superset test: ranges do not depend on it)
  int lidar_plane_body;            // -1: no such plane, no cull
  float lidar_plane_p[3], lidar_plane_n[3], lidar_plane_slack, lidar_plane_slope;
  // mesh rasteriser (smj_meshlet_kernel; tables built at smj_create from smj_meshlet.h): meshlet vertices (float4), triangles (three
  // local vertex indices packed in 32 bits), records (three 16-byte words per meshlet: vbase nvert tbase ntri | sphere | cone) and
  // the work list over the mesh / box geoms of the visible-geom table: (table entry, first meshlet, count <= 32, 0).  raster = 0: ray cast the meshes (BVHs)
  const float4* mlvert;
  const unsigned* mltri;
  const int4* mlrec;
  const int2* mlitem;   // (visible-geom table entry, meshlet) for every meshlet of every mesh / box geom a camera can see; nmlist of them
  int nmlist, raster, raster_splits, raster_boxes;   // raster_boxes: box geoms are in the work list too (one meshlet each)
  int stat_select;                 // tools-only build (-DSMJ_DEPTH_STATS): >= 0 writes that work counter instead of the depth
};

// lidar: [nlidar][ld] batch-major ranges, -1 = no hit, clipped to the sensor cutoff
void smj_launch_lidar(const DevRender& r, const float* xpose, long ld, int num_envs, float* lidar, long lidar_ld, hipStream_t stream);
// xpose: [nbody*12][ld] batch-major body poses (xpos 3 + xmat 9) written by the step kernel (SMJ_READ_POSES).
// mode 0: everything.  mode 1: render only the geoms rigidly attached to the camera's body, for env 0, raw depth into
// out[height][width] (the camera-static layer: it does not depend on the state).  mode 2: skip those geoms and start every
// ray from layer[height][width].
// workspace: smj_depth_workspace_bytes(num_envs) of device memory, scratch of the per-env staging pass (smj_render.hip).
size_t smj_depth_workspace_bytes(int num_envs);
// RGB stand-in: per pixel the albedo (geom rgba, 8 bit, unlit) of the first geom the ray meets, sky (169, 224, 255) where it meets
// none; rgb [num_envs][height][width][3] bytes, gid (optional) [num_envs][height][width] the geom ids (-1 = none).
void smj_launch_rgb(const DevRender& r, const float* xpose, long ld, int num_envs, int cam, int width, int height, float fovy_deg,
                    unsigned char* rgb, int* gid, float* workspace, hipStream_t stream);
void smj_launch_depth(const DevRender& r, const float* xpose, long ld, int num_envs, int cam, int width, int height,
                      float fovy_deg, float max_depth, float* out, const float* layer, int mode, float* workspace, hipStream_t stream);
