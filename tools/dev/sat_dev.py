"""Development probe: the satellite build of the kernel (lane emulator) against the fp64 oracle on the same model."""
import sys, numpy as np, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from stretch_mujoco_amd import mjcf_compiler as C, model_fuse as F, model_blob as B
from oracle.oracle import Oracle
from emul.emul import Emul
st='/root/reference/stretch_mujoco/models/stretch.xml'
scene=sys.argv[1] if len(sys.argv)>1 else 'scene'
xml = C.scene_table_xml(st) if scene=='scene' else C.kitchen_standin_xml(st, free_objects=True)
m=C.compile_string(xml)
f=F.prepare_for_kernels(m, satellites=("nosat" not in sys.argv))
blob=B.dumps(f)
o=Oracle(blob); o.set_option('solver',2)
nq,nv,nu=o.dim('nq'),o.dim('nv'),o.dim('nu')
print('nq nv nu',nq,nv,nu,'nsat',f['k_nsat'])
e=Emul(blob, dict(nq=nq,nv=nv,nu=nu,nlidar=360), num_envs=1, debug=True)
e.set_option('solver',2)
HOME=[0,0,0.6,0.1,0,0,0,0,0,0]
o.arr('ctrl')[:nu]=HOME; e.ctrl[:,0]=HOME
nsteps=int(sys.argv[2]) if len(sys.argv)>2 else 50
sync = len(sys.argv)>3 and sys.argv[3]=='sync'
e.qpos[:,0]=o.arr('qpos'); e.qvel[:,0]=o.arr('qvel'); e.warm[:,0]=o.arr('qacc_warmstart')
for k in range(nsteps):
    if sync:
        e.qpos[:,0]=o.arr('qpos'); e.qvel[:,0]=o.arr('qvel'); e.warm[:,0]=o.arr('qacc_warmstart')
    e.step(1); o.step(1)
    dq=np.abs(e.qpos[:,0]-o.arr('qpos')); dv=np.abs(e.qvel[:,0]-o.arr('qvel'))
    if k<5 or k%10==0 or dq.max()>1e-3:
        print(k,'dq',dq.max(),dq.argmax(),'dv',dv.max(),dv.argmax(),'nefc',e.info[0,0],o.nefc,'ncon',e.info[1,0],o.ncon,'it',e.info[2,0],'fl',e.info[3,0])
    if not np.isfinite(e.qpos).all(): print('NaN'); break
