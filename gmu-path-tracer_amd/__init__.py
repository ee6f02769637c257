"""MI355X-native wavefront path tracer behind the GMU-Path-Tracer Renderer/Scene/Camera API.

Package layout (only what the hot path needs):
  csrc/   HIP kernels (pt_kernels.hip), deterministic math, C-ABI implementation (gmupt_capi.hip)
  host/   C++17 host side mirroring the reference classes: Camera, SBVH builder + flatten, Renderer/Scene
  capi.py ctypes binding of include/gmupt.h (plumbing for tests / bench; no compute in Python)
  scenes.py  seeded synthetic scenes of the BASELINE configurations
  tiles.py   framebuffer tile split across ranks + RCCL gather (torch.distributed)
  progressive.py  headless progressive front-end: camera / light / resolution events, periodic tile gather
The directory name contains a hyphen, so import it through gmupt_pkg.load() at the repository root.
"""
from . import build  # noqa: F401
from . import capi  # noqa: F401
from . import scenes  # noqa: F401
from . import tiles  # noqa: F401
from . import progressive  # noqa: F401
