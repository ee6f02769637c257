import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, gmupt_pkg
g = gmupt_pkg.load(); capi = g.capi
scene = g.scenes.build_scene(g.scenes.spheres_mesh(202, 3, seed=1234))
dev = capi.Device(0); sb = capi.SceneBuffers(dev, scene)
r = capi.Renderer(dev, 1920, 1080, tile=(0, 0), collect_stats=True); r.bind_scene(sb)
cam = capi.Camera(1920, 1080); cam.set_pose(*scene["camera"])
for _ in range(270): cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
r.reset_stats()
for _ in range(5): cam.update(0.0); r.set_camera(cam.buffer); r.iterate()
st = r.stats(); h = np.array(list(st.ext_depth_hist), dtype=np.float64)
print("inner/ray", st.ext_inner / st.ext_rays, "tris/ray", st.ext_tris / st.ext_rays)
print("visits per ray by depth:", np.round(h / st.ext_rays, 2).tolist())
print("cumulative fraction:", np.round(np.cumsum(h) / h.sum(), 3).tolist())
nodes = scene["nodes"]; inner = nodes["isLeaf"] == 0
depth = np.zeros(len(nodes), int)
for i in range(len(nodes)):
    if inner[i]: depth[nodes["left"][i]] = depth[i] + 1; depth[nodes["right"][i]] = depth[i] + 1
print("inner nodes per depth:", np.bincount(depth[inner]).tolist())
