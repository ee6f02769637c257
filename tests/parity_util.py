"""Shared helpers of the GPU parity tests: run the oracle and the HIP renderer side by side on the same inputs."""
import numpy as np

import oracle_lib as O


def make_pair(pkg, device, scene, width, height, pool, live=0, tile=None, path_budget=0, max_depth=0, full=None, collect_stats=False, threads=8):
    """(oracle renderer, HIP renderer, oracle camera, HIP camera) for one configuration."""
    fw, fh = full if full else (width, height)
    orc = O.Renderer(scene, width, height, pool, live=live, tile=tile, path_budget=path_budget, max_depth=max_depth, threads=threads)
    sb = pkg.capi.SceneBuffers(device, scene)
    hip = pkg.capi.Renderer(device, width, height, pool_paths=pool, live_paths=live, tile=tile, path_budget=path_budget,
                            max_depth=max_depth, collect_stats=collect_stats)
    hip.bind_scene(sb)
    ocam = O.Camera(fw, fh); ocam.set_pose(*scene["camera"])
    hcam = pkg.capi.Camera(fw, fh); hcam.set_pose(*scene["camera"])
    ocam.buffer.lightCount = scene["light_count"]
    hcam.buffer.lightCount = scene["light_count"]
    return orc, hip, ocam, hcam, sb


def step_both(orc, hip, ocam, hcam):
    ocam.update(); hcam.update(0.0)
    assert bytes(ocam.buffer) == bytes(hcam.buffer), "host camera streams diverged"
    orc.set_camera(ocam.buffer); hip.set_camera(hcam.buffer)
    orc.iterate(); hip.iterate()


def compare_state(orc, hip, pool, live, fields=None, where=""):
    """Bitwise comparison of the live slots of every path-state field; returns the list of mismatching field names."""
    a = orc.path_state(); b = hip.read_path_state()
    bad = []
    for name in (fields or O.STATE_FIELDS.keys()):
        fa = O.state_field(a, pool, name, live); fb = O.state_field(b, pool, name, live)
        if not np.array_equal(fa, fb):
            idx = np.argwhere(fa != fb)
            bad.append((name, len(idx), idx[0].tolist(), fa[tuple(idx[0])], fb[tuple(idx[0])]))
    return bad


def max_rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(a), 1e-12))) if a.size else 0.0
