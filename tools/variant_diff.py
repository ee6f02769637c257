"""Debug aid: first iteration at which a traversal variant (GMUPT_TRAVERSAL=<mode>) departs from the oracle, and in which fields."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, gmupt_pkg
import oracle_lib as O, parity_util as U
os.environ["GMUPT_TRAVERSAL"] = sys.argv[1] if len(sys.argv) > 1 else "cast0"
g = gmupt_pkg.load(); dev = g.capi.Device(0)
scene = g.scenes.build_scene(g.scenes.cornell_mesh() if len(sys.argv) < 3 else g.scenes.spheres_mesh(12, 2, seed=3))
W, H, P = 64, 36, 4096
orc, hip, ocam, hcam, sb = U.make_pair(g, dev, scene, W, H, P)
for it in range(6):
    U.step_both(orc, hip, ocam, hcam)
    bad = U.compare_state(orc, hip, P, P)
    print("iteration", it, "mismatching fields:", [(b[0], b[1], b[2]) for b in bad][:8])
    if bad:
        name, n, idx, a, b = bad[0]
        print("   first:", name, "slot/comp", idx, "oracle", hex(int(a)), "hip", hex(int(b)))
        hd_o = O.state_field(orc.path_state(), P, "hitDistance", P).view(np.float32)[:8, 0]; hd_h = O.state_field(hip.read_path_state(), P, "hitDistance", P).view(np.float32)[:8, 0]
        print("   hitDistance oracle", hd_o.tolist()); print("   hitDistance hip   ", hd_h.tolist())
        break
