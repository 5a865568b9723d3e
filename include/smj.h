/* smj.h -- C-ABI of the batched Stretch physics path (libsmj.so, HIP / gfx950).
 *
 * This is the drop-in boundary ("lower seam").  In the reference the seam is the set of
 * pybind11 calls into the `mujoco` package made by stretch_mujoco/mujoco_server.py; each entry point below
 * cites the reference call it replaces.  All buffers are plain device pointers owned by the CALLER (PyTorch
 * tensors on the ROCm device); the library never allocates or frees memory it is handed.  Every call returns
 * 0 on success and a negative code on error (see smj_last_error); nothing throws or aborts across the ABI.
 * Launches are asynchronous on the stream passed by the caller.  One context per device; a context is not
 * thread-safe, distinct contexts are independent.
 */
#ifndef SMJ_H
#define SMJ_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct smj_ctx smj_ctx;

/* Buffer slots for smj_bind.  Arrays are fp32 (int32 where noted), batch-major [dim][ld], ld >= num_envs. */
enum smj_slot {
  SMJ_SLOT_QPOS = 0,      /* [nq][B]   MjData.qpos                                                        */
  SMJ_SLOT_QVEL = 1,      /* [nv][B]   MjData.qvel                                                        */
  SMJ_SLOT_CTRL = 2,      /* [nu][B]   MjData.ctrl   (written by push_command, mujoco_server.py:527-578)   */
  SMJ_SLOT_WARMSTART = 3, /* [nv][B]   MjData.qacc_warmstart                                               */
  SMJ_SLOT_NSTEP = 4,     /* int32 [B] steps since reset; MjData.time = nstep * opt.timestep               */
  SMJ_SLOT_ACT_LENGTH = 5,/* [nu][B]   MjData.actuator_length   (pull_status, mujoco_server.py:475-504)    */
  SMJ_SLOT_ACT_VELOCITY = 6, /* [nu][B] MjData.actuator_velocity                                           */
  SMJ_SLOT_BASE_POSE = 7, /* [3][B]    x, y, theta of base_link (BaseController.get_base_pose, :124-129)   */
  SMJ_SLOT_GYRO = 8,      /* [3][B]    sensor base_gyro   (mujoco_server_sensor_manager.py:85)             */
  SMJ_SLOT_ACCEL = 9,     /* [3][B]    sensor base_accel                                                   */
  SMJ_SLOT_LIDAR = 10,    /* [nlidar][B] sensors base_lidar000.. (mujoco_server_sensor_manager.py:77-83)   */
  SMJ_SLOT_INFO = 11,     /* int32 [4][B] nefc, ncon, solver iterations, flags (sticky, OR-ed by every step until the caller
                             clears them: bit 0 = constraint rows beyond the kernel variant's capacity were degraded / dropped,
                             bit 1 = contacts beyond capacity dropped, bit 2 = non-finite state reset to qpos0, bit 3 = the
                             pipelined dispatch gave up waiting for the env's previous chunk: the env ran fewer steps than
                             asked for (NSTEP tells how many) -- never observed, it bounds a wait that would otherwise hang); bits 8..14: which capacity
                             bits 0 / 1 refer to in the satellite builds (diagnostic: items, coupled satellites, satellite-satellite rows,
                             broadphase lists, dense rows, rows, contacts)  */
  SMJ_SLOT_DEBUG = 12,    /* [SMJ_DEBUG_FLOATS][B] optional stage dumps for parity tests (may stay unbound) */
  SMJ_SLOT_PROF = 13,     /* [32][B] optional per-stage shader-cycle counters and event counts of a launch (filled by the
                             standard kernel variant only: binding it selects that variant's profiling build)           */
  SMJ_SLOT_XPOSE = 14,    /* [nbody*12][B] world pose (xpos 3 + xmat 9, row major) of every fused body at the last step:
                             what Renderer.update_scene reads from MjData (mujoco_server_camera_manager.py:135); input of
                             smj_render_depth                                                                            */
  SMJ_SLOT_BASECTL = 15,  /* [8][B] optional: relative base move in flight, per env (BaseController, mujoco_server.py:93-176): row 0 mode
                             (0 none, 1 translate-by, 2 rotate-by, 3 velocity), rows 1-3 start pose x, y, theta, row 4 increment,
                             rows 5-6 v, omega.  Written by the host when a base command is pushed (push_command, :541,:566-568);
                             smj_step runs BaseController.update() inside the kernel after every physics step */
  SMJ_SLOT_COUNT = 16
};

enum smj_dim {
  SMJ_DIM_NQ = 0, SMJ_DIM_NV = 1, SMJ_DIM_NU = 2, SMJ_DIM_NBODY = 3, SMJ_DIM_NLIDAR = 4, SMJ_DIM_NKEY = 5,
  SMJ_DIM_NUM_ENVS = 6, SMJ_DIM_DEBUG_FLOATS = 7, SMJ_DIM_NEFC_MAX = 8, SMJ_DIM_NCON_MAX = 9, SMJ_DIM_NCAM = 10,
  SMJ_DIM_NV_MAX = 11,   /* dof capacity of the kernel variant chosen for this model: 32 (standard) or 64 (big) */
  SMJ_DIM_NSAT_MAX = 12, /* satellite capacity of the variant (0: a build without satellites; the debug dump then ends with 6 floats
                            of qacc per satellite slot) */
  SMJ_DIM_COUNT = 13
};

/* readout flags for smj_step */
enum { SMJ_READ_IMU = 1, SMJ_READ_LIDAR = 2, SMJ_READ_POSES = 4 };

/* Replaces MjModel.from_xml_path + MjData(model) (mujoco_server.py:252,258): `blob` is the compiled model
 * produced by stretch_mujoco_amd.mjcf_compiler / model_fuse (SMJB format, model_blob.py). */
int smj_create(const void* blob, size_t nbytes, int num_envs, int device, smj_ctx** out);
int smj_destroy(smj_ctx* ctx);

/* Attach a caller-owned device buffer to a slot (the MjData field views the reference reads/writes). */
int smj_bind(smj_ctx* ctx, int slot, void* dev_ptr, long ld);

/* Copy of model constants the host glue needs: qpos0[nq], key_ctrl[nkey][nu], jnt ranges are read from the blob
 * on the Python side; this returns dimensions only (model.nq etc.). */
int smj_dims(const smj_ctx* ctx, int* out /* SMJ_DIM_COUNT ints */);

/* mj_resetData for the envs whose mask byte is non-zero (mask_dev == NULL: all).  qpos <- qpos0 (per-env start
 * pose applied by the caller afterwards, cf. change_start_pose, mujoco_server.py:206-229), qvel, ctrl,
 * warmstart, nstep <- 0. */
int smj_reset(smj_ctx* ctx, const uint8_t* mask_dev, void* stream);

/* `nsteps` x mj_step (mujoco_server.py:378) for every env with ctrl held constant, one wavefront per env.
 * On return (stream order) the bound ACT_LENGTH / ACT_VELOCITY / BASE_POSE hold what MjData holds after the last mj_step --
 * the values of its forward pass, one step behind qpos, exactly what pull_status reads (mujoco_server.py:465-515) -- and,
 * if requested in read_flags, GYRO/ACCEL (+ LIDAR) hold the sensor values of the last step.  With BASECTL bound the relative
 * base moves advance inside the launch (BaseController.update after every step) and CTRL's wheel entries are written back. */
int smj_step(smj_ctx* ctx, int nsteps, unsigned read_flags, void* stream);

/* One BaseController.update() (mujoco_server.py:110-122) on the bound arrays: reads BASE_POSE and BASECTL, writes the wheel
 * entries of CTRL and the controller mode.  smj_step does this after every physics step inside the kernel; this entry runs
 * the same arithmetic once, for the tick that follows a freshly pushed command and for the parity tests (golden sequences
 * of the reference's controller on scripted poses). */
int smj_base_controller_tick(smj_ctx* ctx, void* stream);

/* Solver / collision options (mjOption fields): "iterations", "tolerance", "warmstart", "pgs_fixed_iter", "qcqp_exact" (PGS, default 0:
 * the friction QCQP of an elliptic contact finds mju_QCQP's root through the secular form, started at the contact's multiplier of
 * the previous sweep; 1: mju_QCQP's own iteration from 0, cap 20 -- the two differ where that cap is hit),
 * "pgs_two_waves" (PGS on the 16-satellite build, default 1: two wavefronts per env -- the second sweeps the satellites' constraint
 * islands beside the first one's sweeps of the dense system; 0: one wavefront), "newton_two_waves" (Newton on the 16-satellite build,
 * default 1 (= 3: both satellite builds; tools: 5 the 16-satellite build only, 2 the 32-satellite one only): two wavefronts per env -- the second takes the moving-moving pairs of the collision stage and the satellites' lane-serial
 * stages (forward pass, Newton blocks, integration) beside the first one's work; the same states bit for bit as 0: one wavefront),
 * "balance_min" (default 1: launches of at least this many
 * steps are dispatched longest-env-first),
 * "pgs_island_stop" (PGS on the satellite builds, default 1: a satellite's constraint island whose own scaled improvement fell below
 * tolerance / 64 stops sweeping while the rest goes on; 0: every island sweeps until the whole system stops, as mj_solPGS without islands),
 * "grad_noise" (Newton, default 4e-6: the loop also stops when every gradient component is below grad_noise * (|M a| + |qfrc| + |J' f|) of
 * its dof -- the rounding of the gradient's own terms in fp32; 0 = MuJoCo's scale * |grad| < tolerance test only),
 * "manifold_cache" (default on for models with free objects: a convex pair whose two bodies stand within 2e-5 of the poses its manifold
 * was built at keeps it, carried along to first order in the motion since), "pgs_dual_warmstart" (PGS, default on: the sweeps may start
 * from the previous step's constraint forces, matched row by row, when that start has the lower dual cost; not MuJoCo's rule, same fixed point),
 * "max_contacts_per_pair", "solver" (0 PGS, 2 Newton), "convex_pairs", "multiccd" (mjENBL_MULTICCD, stretch.xml:8; default on), "escalate" (default on: an env whose step needs more constraint rows / contacts
 * than the standard kernel variant holds is finished by the tall variant instead of being flagged), "balance" (default on:
 * workgroups are dispatched in the order of the envs' shader time in the previous dispatch, longest first), "chunk" (default 0 = one dispatch; k > 0:
 * smj_step sends its n steps out as dispatches of this many steps on the staged state, each with a fresh order; 0 = one dispatch),
 * "pipeline" (default 5; batches of more than "pipeline_min_envs" envs -- default 511 --: the n steps of a call are cut into chunks of this many steps and the
 * grid holds one workgroup per (chunk, env) that waits on the env's progress counter instead of a barrier between chunks --
 * scheduling only, results are bit-identical; 0 = one workgroup per env per call), "pollers" (default 2: workgroups of the
 * tall variant that run beside the standard kernel on a second stream, finish the current chunk of an env that ran out of
 * rows and hand it back -- they leave at once unless one of the last 8 calls had such envs; -n: n pollers that always stay; 0 = such envs are
 * finished after the standard kernel), "pipeline_big" (the same chunking for the 38- / 50-column variants),
 * "primary_rows" (0 = the variant's own limit; tests lower it to force hand-overs to the larger variant),
 * "sep_cache" (default 1: a convex pair found disjoint keeps the separating direction and tests it first on the next steps --
 * a proof of disjointness whatever the entry holds, so results do not depend on the option),
 * "depth_raster" (default 1: smj_render_depth draws meshes and boxes with the meshlet rasteriser and resolves the remaining
 * primitives per pixel; 0: per-pixel ray cast of every geom through the mesh BVHs -- the same image up to fp32 rounding at
 * silhouette pixels), "depth_raster_splits" (default 8: workgroups per env of the rasteriser). */
int smj_set_option(smj_ctx* ctx, const char* name, double value);

/* Depth image of camera `camera_id` (index into the model's cameras, stretch.xml order: d405_rgb, d405_depth,
 * d435i_camera_rgb, d435i_camera_depth, nav_camera_rgb) for every env, from the body poses in SMJ_SLOT_XPOSE (written by
 * the last smj_step that had SMJ_READ_POSES set).  Replaces Renderer.update_scene + Renderer.render with depth enabled
 * (mujoco_server_camera_manager.py:127-143) followed by StretchCameras.post_processing_callback
 * (enums/stretch_cameras.py:87-102): out_dev is fp32 [num_envs][height][width], metres along the optical axis, row 0 at the
 * top; values beyond max_depth are 0 (utils.limit_depth_distance, utils.py:87-91); max_depth <= 0 returns the raw render
 * (far plane where nothing is hit).  fovy_deg is the vertical field of view set_camera_params writes into the model
 * (mujoco_server_camera_manager.py:185-215). */
int smj_render_depth(smj_ctx* ctx, int camera_id, int width, int height, float fovy_deg, float max_depth, void* out_dev,
                     void* stream);

/* RGB stand-in for the reference's colour cameras (cam_d405_rgb, cam_d435i_rgb, cam_nav_rgb; Renderer.render without depth,
 * mujoco_server_camera_manager.py:127-143): per pixel the 8-bit albedo -- the geom's rgba, its material's where the MJCF names
 * one -- of the first geom the pixel ray meets (same rays, geoms and front-face rule as smj_render_depth), (169, 224, 255)
 * where it meets none.  No lighting, no textures, no shadows: NOT MuJoCo's OpenGL image, a placeholder with the reference's
 * shapes, resolutions and intrinsics.  rgb_dev: uint8 [num_envs][height][width][3]; gid_dev (optional, may be null): int32
 * [num_envs][height][width], the geom ids (-1 = none). */
int smj_render_rgb(smj_ctx* ctx, int camera_id, int width, int height, float fovy_deg, void* rgb_dev, void* gid_dev, void* stream);

/* Multi-GPU: envs shard across one process per GPU with no exchange while stepping -- the reference itself is
 * one env per process with no coupling (stretch_mujoco_simulator.py:102-118).  The only collective of the path gathers the
 * per-env returns of all ranks, rank-major, with RCCL (ncclAllGather over xGMI), issued on the caller's stream.
 * smj_comm_init joins this context to a communicator of `world` ranks: rank 0 creates the ncclUniqueId and publishes it
 * atomically in the file `id_path` (a path all ranks of the node can read, unique per job); the other ranks wait for it up to
 * `timeout_s` seconds.  librccl is bound at run time (dlopen; a copy already mapped by PyTorch is reused; the environment variable
 * SMJ_RCCL_LIB names another library file -- a site's own RCCL build, or the tests' stub), so a single-GPU
 * process never loads it.  Without smj_comm_init (or with world == 1) smj_allgather_returns is a device copy. */
int smj_comm_init(smj_ctx* ctx, int rank, int world, const char* id_path, double timeout_s);
int smj_allgather_returns(smj_ctx* ctx, const float* send_dev /* [count] */, float* recv_dev /* [world*count] */, int count,
                          void* stream);
int smj_comm_destroy(smj_ctx* ctx);

const char* smj_last_error(const smj_ctx* ctx);
const char* smj_version(void);

#ifdef __cplusplus
}
#endif
#endif
