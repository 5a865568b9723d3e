"""Development probe: state-synchronised bench workload, satellite emulator build vs the oracle."""
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from stretch_mujoco_amd import mjcf_compiler as C, model_fuse as F, model_blob as B
import rollout_common as rc
st='/root/reference/stretch_mujoco/models/stretch.xml'
scene=sys.argv[1]; B_=int(sys.argv[2]); W=int(sys.argv[3])
xml = C.scene_table_xml(st) if scene=='scene' else C.kitchen_standin_xml(st, free_objects=True)
f=F.prepare_for_kernels(C.compile_string(xml), satellites=("nosat" not in sys.argv))
blob=B.dumps(f); model=B.loads(blob)
be=rc.EmulBackend(blob, B_)
rel,events=rc.state_synchronised(be, blob, model, B_, W, seed=5)
for ev in events: print({k:(round(v,4) if isinstance(v,float) else v) for k,v in ev.items()})
print('rel p50 %.2e p99 %.2e max %.2e'%(np.percentile(rel,50),np.percentile(rel,99),rel.max()), 'events',len(events), 'unexplained',[e for e in events if not e['explained']])
c=rc.state_synchronised.contacts
print('contacts',c['n'],'mismatched steps',c['mismatched_steps'])

import ctypes
L=be.e.L; L.emul_ext_steps.restype=ctypes.c_long
print('steps with a dense extension:', L.emul_ext_steps())
