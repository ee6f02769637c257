"""Generates tests/golden/*.npz from the CPU oracle (oracle/) and the host builder.

These fixtures are SELF-GENERATED regression pins, not reference-derived vectors: the reference ships no tests, no
golden images and cannot be run here ("parity unpinned").  They freeze the oracle's behaviour so that a
change to oracle/ or to the builder that alters results is caught, and they travel to the GPU box where the HIP path is
compared against them as well.
"""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gmupt_pkg  # noqa: E402
import oracle_lib as O  # noqa: E402

pkg = gmupt_pkg.load()
OUT = os.path.join(R, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def render(scene, W, H, P, iters, **kw):
    orc = O.Renderer(scene, W, H, P, **kw)
    cam = O.Camera(W, H); cam.set_pose(*scene["camera"]); cam.buffer.lightCount = scene["light_count"]
    seeds = []
    for _ in range(iters):
        cam.update(); orc.set_camera(cam.buffer); orc.iterate()
        seeds.append((cam.buffer.randomSeed[0], cam.buffer.randomSeed[1]))
    fb = orc.framebuffer().copy(); qc = orc.counters().copy(); st = orc.path_state().copy()
    s = orc.stats()
    orc.close()
    return fb, qc, st, np.array(seeds, np.float32), s


def main():
    cases = {
        "cornell_64x36_p4096_i24": (pkg.scenes.cornell_mesh(), 64, 36, 4096, 24, {}),
        "spheres12_48x27_p2048_i160": (pkg.scenes.spheres_mesh(n_spheres=12, subdiv=2, seed=7, floor_quads=4), 48, 27, 2048, 160, {}),
        "textured_48x27_p2048_i30": (pkg.scenes.textured_mesh(), 48, 27, 2048, 30, {}),
        "soup2000_32x18_p1024_i16_budget": (pkg.scenes.random_triangles_mesh(2000, seed=1), 32, 18, 1024, 64, {"path_budget": 32 * 18 * 4}),
    }
    for name, (mesh, W, H, P, iters, kw) in cases.items():
        scene = pkg.scenes.build_scene(mesh)
        fb, qc, st, seeds, s = render(scene, W, H, P, iters, **kw)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), framebuffer=fb, counters=qc,
                            state_crc=np.array([np.bitwise_xor.reduce(st.view(np.uint32).astype(np.uint64) * (np.arange(st.size // 4, dtype=np.uint64) % 1000003 + 1))], np.uint64),
                            seeds=seeds, nodes=scene["nodes"], tris=scene["tris"], sah=np.float32(scene["sah"]),
                            paths_ended=np.uint64(s.pathsEnded), ext_inner=np.uint64(s.extInner), ext_tris=np.uint64(s.extTris),
                            width=W, height=H, pool=P, iters=iters, path_budget=kw.get("path_budget", 0))
        print(name, "nodes", scene["nodes"].shape[0], "refs", scene["tris"].shape[0], "ended", s.pathsEnded, "spp mean", fb[..., 3].view(np.uint32).mean())


if __name__ == "__main__":
    main()
