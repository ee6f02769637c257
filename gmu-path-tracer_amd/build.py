"""Builds libgmupt.so (HIP kernels + C-ABI + host classes) for gfx950 with hipcc, in-tree.

The float flags are part of the numerical contract (see csrc/detmath.hpp and DESIGN.md):
no FMA contraction, correctly rounded fp32 divide/sqrt, fp32 denormals kept.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libgmupt.so")

DEVICE_SOURCES = ["csrc/pt_kernels.hip", "csrc/pt_traverse.hip", "csrc/pt_traverse_wide.hip", "csrc/pt_traverse_variants.hip", "csrc/gmupt_capi.hip"]   # pt_traverse_variants.hip is empty without -DGMUPT_VARIANTS
HOST_SOURCES = ["host/sbvh_builder.cpp", "host/Camera.cpp", "host/TextureLoader.cpp", "host/AvirResize.cpp"]
HEADERS = ["csrc/pt_traverse_common.hpp", "csrc/pt_traverse_deferred.hpp", "csrc/pt_kernel_util.hpp", "host/MeshData.hpp", "host/BVHWrapper.hpp", "csrc/pt_device.hpp", "csrc/detmath.hpp", "host/sbvh_builder.hpp", "host/Camera.hpp", "host/TextureLoader.hpp", "host/png_reader.hpp", "host/Constants.hpp", "../include/gmupt.h"]

FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero",
    "-fno-fast-math", "-Wall", "-Wno-unused-function",
]


# Test-only builds of the same sources (tests/test_parity_gpu.py): the traversal ladder kept for A/B timing, and a build whose
# two-level rank computation has one block per group (so that a small pool reaches the many-group paths of k_logic / k_material),
# and a build of the wide ray cast whose stacks overflow on ordinary scenes.
TEST_BUILDS = {"variants": ["-DGMUPT_VARIANTS"], "scan1": ["-DGMUPT_SCAN_GROUP=1"],
               "wides8": ["-DGMUPT_WIDE_STACK=8", "-DGMUPT_WIDE_TOP=64", "-DGMUPT_WIDE_PARK=4"]}   # wide ray cast with a tiny LDS share per lane: stacks run full, rays are parked for the exact walk all the time
EXPERIMENT_BUILDS = {"wxcd": ["-DGMUPT_WIDE_XCD_EXPERIMENT=1"], "wsg0": ["-DGMUPT_WIDE_SIGNED=0"], "mrg0": ["-DGMUPT_MATERIAL_REGROUP=0"], "wq0": ["-DGMUPT_WIDE_QUADPK=0"], "wqt0": ["-DGMUPT_WIDE_QUADTRI=0"]}   # name -> extra flags of A/B timing builds (tools/ only, never loaded by tests), e.g. {"wg1024": ["-DGMUPT_DEF_BLOCK=1024", "-DGMUPT_TOP_NODES=512"]}


def lib_path(name=None):
    return LIB if not name else os.path.join(HERE, "libgmupt_%s.so" % name)


def needs_build(name=None):
    lib = lib_path(name)
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = DEVICE_SOURCES + HOST_SOURCES + HEADERS + ["build.py"]
    return any(os.path.getmtime(os.path.join(HERE, d)) > t for d in deps if os.path.exists(os.path.join(HERE, d)))


def build(force=False, verbose=False, name=None):
    """Builds libgmupt.so (name=None) or one of TEST_BUILDS (libgmupt_<name>.so); returns the path."""
    lib = lib_path(name)
    if not force and not needs_build(name):
        return lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("GMUPT_EXTRA_FLAGS", "").split() + ({**TEST_BUILDS, **EXPERIMENT_BUILDS}[name] if name else [])
    cmd = [hipcc] + FLAGS + extra + ["-x", "hip"] + [os.path.join(HERE, s) for s in DEVICE_SOURCES + HOST_SOURCES] + ["-o", lib]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=HERE)
    return lib


def build_all(force=False, verbose=False):
    """The shipped library and the test builds, compiled concurrently."""
    from concurrent.futures import ThreadPoolExecutor
    names = [None] + sorted(TEST_BUILDS)
    with ThreadPoolExecutor(len(names)) as ex:
        return list(ex.map(lambda n: build(force=force, verbose=verbose, name=n), names))


if __name__ == "__main__":
    if "--experiments" in sys.argv:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(4) as ex:
            print("\n".join(ex.map(lambda n: build(force=True, name=n), sorted(EXPERIMENT_BUILDS))))
    elif "--all" in sys.argv:
        print("\n".join(build_all(force="--force" in sys.argv, verbose=True)))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
