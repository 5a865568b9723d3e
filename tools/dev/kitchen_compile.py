import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from stretch_mujoco_amd import mjcf_compiler as C, model_fuse as F, model_blob as B
from stretch_mujoco_amd.robocasa_import import convert_kitchen_xml
from kitchen_robocasa_fixture import kitchen_xml
st='/root/reference/stretch_mujoco/models/stretch.xml'
xml, stats = kitchen_xml(); print(stats)
kx, pose = convert_kitchen_xml(xml, st); print('pose', pose)
t=time.time(); m=C.compile_string(kx); print('compile', time.time()-t, 's', dict(zip("nq nv nu nbody njnt ngeom".split(), [int(x) for x in m["dims"][:6]])), 'npair', int(m['dims'][12]))
t=time.time(); f=F.prepare_for_kernels(m, satellites=True); print('fuse', time.time()-t, 'nsat', f['k_nsat'], 'main', f['k_main_dims'])
