import sys, time; import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
import numpy as np, gmupt_pkg, oracle_lib as O, parity_util as PU
g = gmupt_pkg.load()
dev = g.capi.Device(0)
# detmath
rng = np.random.default_rng(0)
for fn,name,x,y in [(0,'sin',rng.uniform(-300,300,100000),None),(1,'cos',rng.uniform(-300,300,100000),None),(2,'log2',rng.uniform(1e-30,2,100000),None),(3,'exp2',rng.uniform(-140,20,100000),None),(4,'pow',rng.uniform(0,1,100000),np.full(100000,1/2.2)),(6,'rng',rng.integers(0,1<<22,100000).astype(np.float32),rng.uniform(0,1,100000))]:
    a = O.detmath(fn,x,y); b = dev.detmath(fn,x,y)
    print(name, 'bit-equal' if np.array_equal(a.view(np.uint32), b.view(np.uint32)) else 'MISMATCH %d'%(a.view(np.uint32)!=b.view(np.uint32)).sum())
scene = g.scenes.build_scene(g.scenes.cornell_mesh())
W,H,P = 64,36,4096
orc,hip,ocam,hcam,sb = PU.make_pair(g, dev, scene, W,H,P)
for it in range(12):
    PU.step_both(orc,hip,ocam,hcam)
    bad = PU.compare_state(orc,hip,P,P)
    qa = orc.counters().copy(); qb = hip.counters()
    fa = orc.framebuffer(); fb = hip.framebuffer()
    print(it, 'state mismatches', [(b[0],b[1]) for b in bad], 'qc', qa.tolist(), qb.tolist(), 'fb equal', np.array_equal(fa.view(np.uint32), fb.view(np.uint32)))
    if bad: print(bad[:3]); break
print(hip.stats().as_dict())
