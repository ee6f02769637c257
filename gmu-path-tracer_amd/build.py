"""Builds libgmupt.so (HIP kernels + C-ABI + host classes) for gfx950 with hipcc, in-tree.

The float flags are part of the numerical contract (see csrc/detmath.hpp and DESIGN.md):
no FMA contraction, correctly rounded fp32 divide/sqrt, fp32 denormals kept.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libgmupt.so")

DEVICE_SOURCES = ["csrc/pt_kernels.hip", "csrc/pt_traverse.hip", "csrc/pt_traverse_variants.hip", "csrc/gmupt_capi.hip"]
HOST_SOURCES = ["host/sbvh_builder.cpp", "host/Camera.cpp", "host/TextureLoader.cpp"]
HEADERS = ["csrc/pt_traverse_common.hpp", "csrc/pt_kernel_util.hpp", "host/MeshData.hpp", "host/BVHWrapper.hpp", "csrc/pt_device.hpp", "csrc/detmath.hpp", "host/sbvh_builder.hpp", "host/Camera.hpp", "host/TextureLoader.hpp", "host/png_reader.hpp", "host/Constants.hpp", "../include/gmupt.h"]

FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero",
    "-fno-fast-math", "-Wall", "-Wno-unused-function",
]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = DEVICE_SOURCES + HOST_SOURCES + HEADERS + ["build.py"]
    return any(os.path.getmtime(os.path.join(HERE, d)) > t for d in deps if os.path.exists(os.path.join(HERE, d)))


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("GMUPT_EXTRA_FLAGS", "").split()
    cmd = [hipcc] + FLAGS + extra + ["-x", "hip"] + [os.path.join(HERE, s) for s in DEVICE_SOURCES + HOST_SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=HERE)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
