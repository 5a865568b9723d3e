import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from stretch_mujoco_amd import mjcf_compiler as C, model_fuse as F, model_blob as B
from stretch_mujoco_amd.robocasa_import import convert_kitchen_xml
from kitchen_robocasa_fixture import kitchen_xml
import rollout_common as rc
st='/root/reference/stretch_mujoco/models/stretch.xml'
xml, stats = kitchen_xml()
kx, pose = convert_kitchen_xml(xml, st)
m=C.compile_string(kx)
m["qpos0"][0:3]=pose["pos"]; m["qpos0"][3:7]=pose["quat"]
f=F.prepare_for_kernels(m, satellites=True)
blob=B.dumps(f); model=B.loads(blob)
print('blob bytes', len(blob))
Bn=int(sys.argv[1]); W=int(sys.argv[2])
t=time.time()
be=rc.EmulBackend(blob, Bn, variant='sat32')
rel,events=rc.state_synchronised(be, blob, model, Bn, W, seed=5)
print('time', time.time()-t)
for ev in events: print({k:(round(v,4) if isinstance(v,float) else v) for k,v in ev.items()})
print('rel p50 %.2e p99 %.2e max %.2e'%(np.percentile(rel,50),np.percentile(rel,99),rel.max()), 'events',len(events))
c=rc.state_synchronised.contacts
print('contacts',c['n'],'mismatched steps',c['mismatched_steps'])
