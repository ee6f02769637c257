import os, sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import gmupt_pkg, oracle_lib as O, parity_util as PU
pkg = gmupt_pkg.load()
scene = pkg.scenes.build_scene(pkg.scenes.random_triangles_mesh(2000, seed=1))
os.environ["GMUPT_TRAVERSAL"] = "wide"
dev = pkg.capi.Device(0)
W, H, P = 32, 18, 1024
orc, hip, ocam, hcam, sb = PU.make_pair(pkg, dev, scene, W, H, P, collect_stats=True)
PU.step_both(orc, hip, ocam, hcam)
a = orc.path_state(); b = hip.read_path_state()
hd_o = O.state_field(a, P, "hitDistance").view(np.float32)[:, 0]; hd_h = O.state_field(b, P, "hitDistance").view(np.float32)[:, 0]
tri_o = O.state_field(a, P, "triangle"); tri_h = O.state_field(b, P, "triangle")
bad = np.nonzero(hd_o.view(np.uint32) != hd_h.view(np.uint32))[0]
print("rays", P, "hitDistance differs on", len(bad))
for i in bad[:12]:
    print(i, "oracle t", hd_o[i], "tri", tri_o[i].tolist(), "| hip t", hd_h[i], "tri", tri_h[i].tolist())
so, sh = orc.stats(), hip.stats()
print("oracle ext tris", so.extTris, "leaves", so.extLeaves, "hip tris", sh.ext_tris, "leaves", sh.ext_leaves, "inner", sh.ext_inner, "redo", sh.cast_redo_rays)
# where are the oracle's hit references within their leaves?
nodes = scene["nodes"]; leaves = nodes[nodes["isLeaf"] != 0]
