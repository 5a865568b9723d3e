/* TEST INFRASTRUCTURE ONLY -- fp64 CPU restatement of the physics step ("oracle").
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product path (stretch_mujoco_amd/) never links, imports or calls it.
 *
 * PARITY UNPINNED: the arithmetic restated here lives in third-party mujoco==3.2.6
 * (reference pyproject.toml:12), which is absent from /root/reference and from this
 * image.  The one reference call site is stretch_mujoco/mujoco_server.py:378
 * (`mj_step(self.mjmodel, self.mjdata)`).  Each function below names the MuJoCo stage
 * it restates; no MuJoCo could be run against it here.
 * What does pin it are the MuJoCo outputs the reference itself holds in print (docs/getting_started.ipynb): the joint
 * status at t = 8.26 s (cell 20: lift / arm to 1.4e-5, wrist and head joints to 5e-7, the lift's creep rate to 1 %), the
 * head_tilt limit stop (cell 23, to 2e-9) and 30 depth pixels of both depth cameras in the default scene (cell 14, to the
 * printed 1e-3) -- tests/test_oracle_physics.py, tests/test_depth_oracle.py -- and, since round 5, the IMAGES it stores: the five camera
 * frames of cell 15 (the wrist depth map: 53 400 MuJoCo pixels to 0.3 grey levels on average), the nav frame of cell 23 and the lidar
 * figure of cell 18 ray by ray, all of them at the one base pose cell 20 prints (tests/test_notebook_images.py; golden data decoded by
 * tools/gen_notebook_images_golden.py / gen_lidar_golden.py).  They cover the quasi-static chain (kinematics, gravity compensation,
 * actuators, equality constraints, friction loss, limits, wheel contacts, implicitfast), the camera and rangefinder models and the
 * scene geometry; fast contact dynamics, multiccd and box-box manifolds stay unpinned.  Options that are NOT MuJoCo's (default off
 * here): qcqp_cap, pgs_dual_warmstart, manifold_keep (round 6: the twin of the kernels' contact-manifold cache -- tests bound the kernels
 * against it and it against the unmodified restatement, separately).
 */
#ifndef SMJ_ORACLE_H
#define SMJ_ORACLE_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct smjo_model smjo_model;
typedef struct smjo_data smjo_data;

smjo_model* smjo_load(const void* blob, size_t nbytes);
void smjo_free_model(smjo_model* m);
smjo_data* smjo_make_data(const smjo_model* m);
void smjo_free_data(smjo_data* d);

void smjo_reset(const smjo_model* m, smjo_data* d);          /* qpos=qpos0, qvel=0, ctrl=0, time=0 */
void smjo_forward(const smjo_model* m, smjo_data* d);        /* mj_forward */
/* tests: the next forward pass takes these contacts instead of running its collision stage (one-shot) */
void smjo_set_contacts(smjo_data* d, int n, const double* con /* n x (dist, pos[3], normal[3], geom1, geom2) */);
/* option "manifold_keep" (NOT MuJoCo; the twin of the kernels' manifold cache): the kept manifolds as smjo_mc_words() doubles */
int smjo_mc_words(void);
void smjo_mc_get(const smjo_data* d, double* buf);
void smjo_mc_set(smjo_data* d, const double* buf);
void smjo_step(const smjo_model* m, smjo_data* d);           /* mj_step (implicitfast + PGS) */
void smjo_step_n(const smjo_model* m, smjo_data* d, int n);
void smjo_sensors(const smjo_model* m, smjo_data* d, int with_lidar); /* gyro, accel, lidar into d */

/* options: name in {"iterations","tolerance","warmstart","pgs_fixed_iter","max_contacts_per_pair", "qcqp_cap" (process-wide; 20 = MuJoCo)} */
int smjo_set_option(smjo_model* m, const char* name, double value);

/* array access for tests: returns pointer (double*) or NULL; *n receives element count.
 * int-typed arrays are exposed through smjo_get_int. */
double* smjo_get(smjo_data* d, const char* name, int* n);
int* smjo_get_int(smjo_data* d, const char* name, int* n);
int smjo_dim(const smjo_model* m, const char* name);
/* floating-point operations counted since the last reset (vector primitives + the dense loops of every stage) */
long long smjo_flops(int reset);

/* depth image float[H][W] of camera `cam` from the poses of the last smjo_forward / smjo_step; -1 if the blob has no
 * render tables.  max_depth <= 0: raw render (far plane where nothing is hit). */
int smjo_render_depth(smjo_model* m, const smjo_data* d, int cam, int W, int H, double fovy_deg, double max_depth, float* out);
int smjo_render_geomid(smjo_model* m, const smjo_data* d, int cam, int W, int H, double fovy_deg, int* gid, unsigned char* rgb);

#ifdef __cplusplus
}
#endif
#endif
