import sys, time, numpy as np, json
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from stretch_mujoco_amd import mjcf_compiler as C, model_fuse as F, model_blob as B
from stretch_mujoco_amd.robocasa_import import convert_kitchen_xml
from kitchen_robocasa_fixture import kitchen_xml
from oracle.oracle import Oracle
st='/root/reference/stretch_mujoco/models/stretch.xml'
xml, stats = kitchen_xml()
kx, pose = convert_kitchen_xml(xml, st)
m=C.compile_string(kx)
m["qpos0"][0:3]=pose["pos"]; m["qpos0"][3:7]=pose["quat"]
f=F.prepare_for_kernels(m, satellites=True)
blob=B.dumps(f)
names=json.loads(bytes(f['names_json']).decode())
o=Oracle(blob); o.set_option('solver',2)
o.arr('ctrl')[:10]=[0,0,0.6,0.1,0,0,0,0,0,0]
si=f['k_sat_i']; gn=names['geom']
def contacts(tag):
    c=o.arr('contact').reshape(o.ncon,-1)
    gid=[tuple(int(v) for v in c[k,-2:].copy().view(np.int32)[1:3]) for k in range(o.ncon)]
    print(tag, 'ncon', o.ncon, 'nefc', o.nefc)
    for k,(a,b) in enumerate(gid):
        if c[k,0] < -1e-3 or tag=='final': print('   ', gn[a] or a, gn[b] or b, 'dist %.5f'%c[k,0])
o.forward(); contacts('initial (deep ones)')
for k in range(0,1001,200):
    if k: o.step(200)
    q=o.arr('qpos'); v=o.arr('qvel')
    print('step',k,'ncon',o.ncon,'nefc',o.nefc, 'free obj z:', [round(float(q[s[2]+2]),4) for s in si if s[4]==6], 'artic q', [round(float(q[s[2]]),4) for s in si if s[4]==1], '|v|max', float(np.abs(v[26:]).max()))
contacts('final')
