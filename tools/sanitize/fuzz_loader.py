"""Mutation fuzzing of the host-side scene ingestion under AddressSanitizer + UBSan (CPU only).

Writes a valid textured scene as .gltf (+ .bin + .png) and .glb, a .gmesh dump and a .params file, derives mutants (byte flips, truncations,
inserted garbage, swapped JSON numbers) and feeds them to tools/sanitize/loader_fuzz.cpp: every mutant must be either loaded or rejected with
an exception -- never a crash, an out-of-range index or a sanitizer report.  Usage: python tools/sanitize/fuzz_loader.py [mutants-per-file]
"""
import os, re, subprocess, sys, tempfile
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import gmupt_pkg


def main(per_file=120):
    pkg = gmupt_pkg.load()
    tmp = tempfile.mkdtemp(prefix="gmupt_fuzz_")
    exe = os.path.join(tmp, "loader_fuzz")
    host = os.path.join(ROOT, "gmu-path-tracer_amd", "host")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-ffp-contract=off", "-o", exe,
                    os.path.join(HERE, "loader_fuzz.cpp")] + [os.path.join(host, f) for f in ("GltfLoader.cpp", "MeshData.cpp", "TextureLoader.cpp", "AvirResize.cpp", "SceneParams.cpp")], check=True)
    written = pkg.scenes.save_gltf(pkg.scenes.textured_mesh(), os.path.join(tmp, "s.gltf"), texture_scale={(0, 0): 2})
    pkg.scenes.gltf_to_glb(os.path.join(tmp, "s.gltf"), os.path.join(tmp, "packed.glb"))
    pkg.scenes.save_gmesh(pkg.scenes.cornell_mesh(), os.path.join(tmp, "c.gmesh"), os.path.join(tmp, "c.params"))
    seeds = [f for f in sorted(os.listdir(tmp)) if f.endswith((".gltf", ".glb", ".gmesh", ".params", ".png", ".bin")) and os.path.isfile(os.path.join(tmp, f))]
    rng = np.random.default_rng(7)
    total = 0
    for name in seeds:
        data = bytearray(open(os.path.join(tmp, name), "rb").read())
        targets = []
        stem, ext = os.path.splitext(name)
        for k in range(per_file):
            m = bytearray(data)
            kind = k % 5
            if kind == 0 and len(m) > 1:                                   # flip a few bytes
                for _ in range(int(rng.integers(1, 6))): m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
            elif kind == 1 and len(m) > 1: m = m[: int(rng.integers(0, len(m)))]           # truncate
            elif kind == 2: pos = int(rng.integers(0, len(m) + 1)); m[pos:pos] = bytes(rng.integers(0, 256, int(rng.integers(1, 40))).astype(np.uint8))   # insert garbage
            elif kind == 3 and ext in (".gltf", ".params"):                 # replace a number by an extreme one
                nums = list(re.finditer(rb"-?\d+(\.\d+)?", bytes(m)))
                if nums:
                    n = nums[int(rng.integers(0, len(nums)))]
                    m[n.start():n.end()] = [b"-1", b"4294967295", b"1e308", b"99999999999999999999", b"0", b"-0.0", b"nan"][int(rng.integers(0, 7))]
            else:                                                           # zero a block
                a = int(rng.integers(0, max(len(m), 1))); m[a: a + int(rng.integers(1, 64))] = bytes(min(int(rng.integers(1, 64)), max(len(m) - a, 0)))
            # a mutant of the .bin / .png belongs to a copy of the scene that references it
            if ext in (".bin", ".png"):
                d = os.path.join(tmp, "m_%s_%d" % (name.replace(".", "_"), k)); os.makedirs(d, exist_ok=True)
                for other in seeds:
                    if other.endswith((".gltf", ".bin", ".png")): open(os.path.join(d, other), "wb").write(open(os.path.join(tmp, other), "rb").read())
                open(os.path.join(d, name), "wb").write(bytes(m))
                targets.append(os.path.join(d, "s.gltf"))
                if ext == ".png": targets.append(os.path.join(d, name))
            else:
                path = os.path.join(tmp, "m_%s_%d%s" % (stem, k, ext)); open(path, "wb").write(bytes(m))
                if ext == ".gltf":                                            # the mutated document next to the intact .bin / .png files
                    pass
                targets.append(path)
        for i in range(0, len(targets), 60):
            r = subprocess.run([exe] + targets[i:i + 60], capture_output=True, text=True, timeout=600, cwd=tmp)
            total += len(targets[i:i + 60])
            if r.returncode != 0:
                print(r.stdout[-2000:]); print(r.stderr[-6000:])
                raise SystemExit("loader fuzz FAILED on a mutant of %s (kept in %s)" % (name, tmp))
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    print("loader fuzz: %d mutants of %d seed files, no crash, no sanitizer report" % (total, len(seeds)))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 120)
