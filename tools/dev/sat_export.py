import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from stretch_mujoco_amd import mjcf_compiler as C, model_fuse as F, model_blob as B
from kitchen_export_fixture import KITCHEN_EXPORT
from stretch_mujoco_amd.robocasa_import import convert_kitchen_xml
import rollout_common as rc
st='/root/reference/stretch_mujoco/models/stretch.xml'
kx, pose = convert_kitchen_xml(KITCHEN_EXPORT, st)
f=F.prepare_for_kernels(C.compile_string(kx), satellites=True)
print('nsat',f['k_nsat'],'main',f['k_main_dims'],'sat_i',f['k_sat_i'][:,:6].tolist())
blob=B.dumps(f); model=B.loads(blob)
Bn=int(sys.argv[1]); W=int(sys.argv[2])
be=rc.EmulBackend(blob, Bn)
rel,events=rc.state_synchronised(be, blob, model, Bn, W, seed=5)
for ev in events: print({k:(round(v,4) if isinstance(v,float) else v) for k,v in ev.items()})
print('rel p50 %.2e p99 %.2e max %.2e'%(np.percentile(rel,50),np.percentile(rel,99),rel.max()), 'events',len(events))
c=rc.state_synchronised.contacts
print('contacts',c['n'],'mismatched steps',c['mismatched_steps'])
