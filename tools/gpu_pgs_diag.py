"""PGS sweep profile on the device (profiling build of the standard variant): cycles per step in scalar rows / elliptic blocks / QCQP.
   python tools/gpu_pgs_diag.py   (under PGS the Newton counter slots carry: n_update = scalar-row cycles, n_grad = block cycles,
   n_xa = cycles inside the QCQP, n_hmfma = QCQP Newton iterations, n_gauss_jordan = scalar rows visited, n_solve = blocks visited;
   satellite builds: sat_h = cycles of the satellite islands' lanes)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpu_diag
scene = sys.argv[1] if len(sys.argv) > 1 else None   # (a scene whose variant has a profiling build: SMJ_LIB_PATH=.../libsmj_bigprof.so)
for rnd in (False, True):
    gpu_diag.profile(1024, rnd, solver="pgs", scene=scene)
