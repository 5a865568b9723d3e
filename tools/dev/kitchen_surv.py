import sys, os, numpy as np, json, subprocess
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from stretch_mujoco_amd import model_blob as B
from oracle.oracle import Oracle
from emul.emul import Emul
blob=open('/root/repo/stretch_mujoco_amd/models/stretch_kitchen_robocasa.smjb','rb').read()
f=B.loads(blob); names=json.loads(bytes(f['names_json']).decode())['geom']
o=Oracle(blob); o.set_option('solver',2); o.arr('ctrl')[:10]=[0,0,0.6,0.1,0,0,0,0,0,0]; o.step(500)
e=Emul(blob, dict(nq=o.dim('nq'),nv=o.dim('nv'),nu=10,nlidar=360), num_envs=1, variant='sat32'); e.set_option('solver',2)
e.qpos[:,0]=o.arr('qpos'); e.qvel[:,0]=o.arr('qvel'); e.warm[:,0]=o.arr('qacc_warmstart'); e.ctrl[:,0]=[0,0,0.6,0.1,0,0,0,0,0,0]
os.environ['SMJ_SAT_TRACE']='1'
r,w=os.pipe(); old=os.dup(2); os.dup2(w,2)
e.step(1)
os.dup2(old,2); os.close(w)
out=os.read(r,1<<20).decode()
line=[l for l in out.split('\n') if l.startswith('static survivors')][0]
ps=[int(x) for x in line.split(':')[1].split()]
g1=f['pair_geom1']; g2=f['pair_geom2']
from collections import Counter
print(len(ps)); 
for p in ps: print(names[g1[p]] or g1[p], '|', names[g2[p]] or g2[p])
