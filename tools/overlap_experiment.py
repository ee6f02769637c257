"""Would a two-half software pipeline pay?  Two independent renderers (each: half the pool, half the frame, own stream) stepped alternately, so that
the logic / material kernels of one can run beside the ray cast of the other, against one renderer with the whole pool.  Throughput only (no parity)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, gmupt_pkg
g = gmupt_pkg.load(); capi = g.capi
scene = g.scenes.build_scene(g.scenes.spheres_mesh(202, 3, seed=1234))
dev = capi.Device(0); sb = capi.SceneBuffers(dev, scene)
W, H = 1920, 1080
def make(pool, y0, rows):
    r = capi.Renderer(dev, W, rows, pool_paths=pool, tile=(0, y0)); r.bind_scene(sb)
    c = capi.Camera(W, H); c.set_pose(*scene["camera"]); c.buffer.lightCount = scene["light_count"]
    return r, c
def run(rs, prewarm, steps):
    def it(n):
        for _ in range(n):
            for r, c in rs:
                c.update(0.0); r.set_camera(c.buffer); r.iterate()
    it(prewarm)
    for r, _ in rs: r.synchronize(); r.reset_stats()
    t = time.perf_counter(); it(steps)
    for r, _ in rs: r.synchronize()
    dt = time.perf_counter() - t
    paths = sum(r.stats().paths_completed for r, _ in rs); seg = sum(r.stats().segments for r, _ in rs)
    return paths / dt / 1e6, seg / dt / 1e6, dt / steps * 1e3
pre, steps = 2010, 402
one = [make(1 << 21, 0, H)]
print("one renderer, pool 2^21:        %.3f Mpaths/s %.1f Mseg/s %.4f ms per round" % run(one, pre, steps)); one[0][0].close()
half = [make(1 << 20, 0, H // 2)]
print("one renderer, pool 2^20 (half): %.3f Mpaths/s %.1f Mseg/s %.4f ms per round" % run(half, pre, steps)); half[0][0].close()
two = [make(1 << 20, 0, H // 2), make(1 << 20, H // 2, H // 2)]
print("two renderers, 2 x 2^20:        %.3f Mpaths/s %.1f Mseg/s %.4f ms per round" % run(two, pre, steps))
for r, _ in two: r.close()
four = [make(1 << 19, k * (H // 4), H // 4) for k in range(4)]
print("four renderers, 4 x 2^19:       %.3f Mpaths/s %.1f Mseg/s %.4f ms per round" % run(four, pre, steps))
