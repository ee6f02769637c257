#!/usr/bin/env python3
"""CPU estimate for the wide (4-ary) collapse of the binary SBVH: nodes visited per ray, binary against wide.

The reference's walk never prunes a box against the current hit (extensionRayCast.hlsl:79-94,132-159), so the set of leaves a ray
reaches is "every leaf whose box and all of whose ancestors' boxes are hit" -- and a child box lies inside its parent's box, so a
hit child implies a hit parent: testing only SOME ancestors reaches the same leaves.  This script collapses the binary tree into
4-wide nodes (always opening the inner slot with the largest surface area) and counts, for random rays, the binary inner nodes
visited, the wide nodes visited, box tests on either side, the deepest stack, and checks that both walks reach the same leaves.

  python tools/wide_estimate.py [--spheres 202] [--subdiv 3] [--rays 20000]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gmupt_pkg  # noqa: E402


def slab(mn, mx, o, inv):
    with np.errstate(invalid="ignore", over="ignore"):
        f = (mx - o) * inv
        n = (mn - o) * inv
        t1 = np.fmin(np.fmin(np.fmax(f[..., 0], n[..., 0]), np.fmax(f[..., 1], n[..., 1])), np.fmax(f[..., 2], n[..., 2]))
        t0 = np.fmax(np.fmax(np.fmin(f[..., 0], n[..., 0]), np.fmin(f[..., 1], n[..., 1])), np.fmin(f[..., 2], n[..., 2]))
        return (t1 >= t0) & (t1 > 0)


def collapse(nodes, width=4):
    """-> list of slot lists (binary node ids) per wide node, wide index of every binary inner node that is a wide root"""
    area = lambda i: (lambda d: d[0] * d[1] + d[1] * d[2] + d[2] * d[0])(nodes["max"][i].astype(np.float64) - nodes["min"][i])
    wide_of = {}
    slots_of = []
    todo = [0]
    while todo:
        n = todo.pop()
        wide_of[n] = len(slots_of)
        slots = [int(nodes["left"][n]), int(nodes["right"][n])]
        while len(slots) < width:
            inner = [(area(s), -k, k) for k, s in enumerate(slots) if not nodes["isLeaf"][s]]
            if not inner:
                break
            _, _, k = max(inner)
            s = slots[k]
            slots[k:k + 1] = [int(nodes["left"][s]), int(nodes["right"][s])]
        slots_of.append(slots)
        for s in slots:
            if not nodes["isLeaf"][s]:
                todo.append(s)
    return slots_of, wide_of


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spheres", type=int, default=202)
    ap.add_argument("--subdiv", type=int, default=3)
    ap.add_argument("--rays", type=int, default=4000)
    ap.add_argument("--width", type=int, default=4)
    args = ap.parse_args()
    pkg = gmupt_pkg.load()
    scene = pkg.scenes.build_scene(pkg.scenes.spheres_mesh(args.spheres, args.subdiv, seed=1234))
    nodes = scene["nodes"]
    N = nodes.shape[0]
    n_inner = int((nodes["isLeaf"] == 0).sum())
    slots_of, wide_of = collapse(nodes, args.width)
    nslots = np.array([len(s) for s in slots_of])
    print("binary: %d nodes, %d inner; wide: %d nodes, slots/node mean %.2f (hist %s)" % (N, n_inner, len(slots_of), nslots.mean(), np.bincount(nslots).tolist()))

    # static stack bound of the wide walk
    order = sorted(wide_of, reverse=True)   # children have larger binary ids than parents
    occ = {}
    for n in order:
        s = slots_of[wide_of[n]]
        occ[n] = max((len(s) - 1 + (occ[c] if not nodes["isLeaf"][c] else 0)) for c in s)
    depth = np.zeros(N, np.int32)
    for i in range(N):
        if not nodes["isLeaf"][i]:
            depth[nodes["left"][i]] = depth[i] + 1; depth[nodes["right"][i]] = depth[i] + 1
    print("binary depth %d; static bound of the wide stack %d entries" % (depth.max(), occ[0]))

    rng = np.random.default_rng(1)
    lo, hi = nodes["min"][0], nodes["max"][0]
    R = args.rays
    o = rng.uniform(lo + 0.05 * (hi - lo), hi - 0.05 * (hi - lo), (R, 3)).astype(np.float32)
    d = rng.normal(size=(R, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    inv = (np.float32(1.0) / d).astype(np.float32)
    mn, mx = nodes["min"], nodes["max"]
    tot_bi, tot_wi, tot_bt, tot_wt, max_b, max_w, tot_leaf = 0, 0, 0, 0, 0, 0, 0
    for r in range(R):
        # binary
        leaves_b = set(); stack = [0]; vis = 0; ms = 0
        if not slab(mn[0], mx[0], o[r], inv[r]):
            stack = []
        while stack:
            ms = max(ms, len(stack))
            n = stack.pop()
            if nodes["isLeaf"][n]:
                leaves_b.add(n); continue
            vis += 1
            for c in (int(nodes["left"][n]), int(nodes["right"][n])):
                if slab(mn[c], mx[c], o[r], inv[r]):
                    stack.append(c)
        tot_bi += vis; tot_bt += 2 * vis; max_b = max(max_b, ms)
        # wide
        leaves_w = set(); stack = [0]; visw = 0; tests = 0; ms = 0
        if not slab(mn[0], mx[0], o[r], inv[r]):
            stack = []
        while stack:
            ms = max(ms, len(stack))
            n = stack.pop()
            if nodes["isLeaf"][n]:
                leaves_w.add(n); continue
            visw += 1
            s = slots_of[wide_of[n]]
            tests += len(s)
            hit = slab(mn[s], mx[s], o[r], inv[r])
            for c, h in zip(s, hit):
                if h:
                    stack.append(c)
        assert leaves_b == leaves_w, "the two walks reach different leaves"
        tot_wi += visw; tot_wt += tests; max_w = max(max_w, ms); tot_leaf += len(leaves_b)
    print("per ray: binary inner %.2f (box tests %.1f), wide nodes %.2f (box tests %.1f, slots incl. empty %.1f), leaves %.2f; deepest stack binary %d, wide %d"
          % (tot_bi / R, tot_bt / R, tot_wi / R, tot_wt / R, args.width * tot_wi / R, tot_leaf / R, max_b, max_w))


if __name__ == "__main__":
    main()


def top_walk_estimate(spheres=202, subdiv=3, rays=3000, top=512):
    """Phase split of the wide walk: all visits of the LDS-resident top first (no vector memory), then the rest.  Per ray: top-node visits,
    entries the top walk leaves behind (global inner nodes, leaves), deepest stack of top nodes."""
    pkg = gmupt_pkg.load()
    scene = pkg.scenes.build_scene(pkg.scenes.spheres_mesh(spheres, subdiv, seed=1234))
    nodes = scene["nodes"]
    slots_of, wide_of = collapse(nodes, 4)
    area = lambda i: (lambda d: d[0] * d[1] + d[1] * d[2] + d[2] * d[0])(nodes["max"][i].astype(np.float64) - nodes["min"][i])
    in_top = set(); frontier = [(area(0), 0)]
    while frontier and len(in_top) < top:
        frontier.sort(); a, n = frontier.pop(); in_top.add(n)
        for c in slots_of[wide_of[n]]:
            if not nodes["isLeaf"][c]:
                frontier.append((area(c), c))
    rng = np.random.default_rng(1)
    lo, hi = nodes["min"][0], nodes["max"][0]
    o = rng.uniform(lo + 0.05 * (hi - lo), hi - 0.05 * (hi - lo), (rays, 3)).astype(np.float32)
    d = rng.normal(size=(rays, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    inv = (np.float32(1.0) / d).astype(np.float32)
    mn, mx = nodes["min"], nodes["max"]
    tv, gi, gl, depth, allv = [], [], [], [], []
    for r in range(rays):
        stack = [0] if slab(mn[0], mx[0], o[r], inv[r]) else []
        visits = 0; ginner = 0; gleaf = 0; md = 0
        while stack:
            md = max(md, len(stack)); n = stack.pop(); visits += 1
            s = slots_of[wide_of[n]]
            hit = slab(mn[s], mx[s], o[r], inv[r])
            for c, h in zip(s, hit):
                if not h: continue
                if nodes["isLeaf"][c]: gleaf += 1
                elif c in in_top: stack.append(c)
                else: ginner += 1
        tv.append(visits); gi.append(ginner); gl.append(gleaf); depth.append(md)
    tv, gi, gl, depth = map(np.array, (tv, gi, gl, depth))
    left = gi + gl
    print("top of %d wide nodes: per ray %.2f top visits; left behind: %.2f global inner + %.2f leaves (max %d, 99.9%% %d); deepest top stack %d"
          % (top, tv.mean(), gi.mean(), gl.mean(), left.max(), int(np.percentile(left, 99.9)), depth.max()))


if __name__ == "__main__" and "--top-walk" in sys.argv:
    pass
