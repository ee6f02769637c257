"""Headless progressive front-end: what Window::loop + GUI do around Renderer::update()/draw() in the reference
(Source/Window.cpp:60-90, Source/GUI.cpp:43-131), without a window.

A session iterates continuously; between frames it applies the events the reference takes from its GUI -- camera motion (with the
accumulation-reset hysteresis of Camera.cpp:72-83, implemented in the host camera), light edits (buffer re-upload + restart,
GUI.cpp:125-130), resolution switches (Renderer.cpp:408-413) -- and every `preview_every` frames it gathers the row-band tiles of all
ranks to rank 0 (tiles.gather_tiles: RCCL over xGMI with the "nccl" backend, gloo in the CPU tests) and hands the frame to a callback.
The renderer / camera objects are the C-ABI wrappers of capi.py (or anything with the same methods).
"""
import numpy as np

from . import tiles


def to_rgba8(frame):
    """The 8-bit image Renderer::captureScreen writes (Source/Renderer.cpp:383-394): uint8 = float * 255 truncated, alpha 255."""
    out = np.empty(frame.shape, np.uint8)
    out[..., :3] = (frame[..., :3] * 255.0).astype(np.uint8)
    out[..., 3] = 255
    return out


class ProgressiveSession:
    def __init__(self, renderer, camera, width, height, rank=0, world=1, dist=None, preview_every=32, on_preview=None):
        self.renderer, self.camera = renderer, camera
        self.width, self.height = width, height
        self.rank, self.world, self.dist = rank, world, dist
        self.preview_every, self.on_preview = preview_every, on_preview
        self.frames = 0
        self.previews = 0

    # ---- events (applied before the next frame, like the GUI callbacks of the reference)
    def move_camera(self, mouse_dx=0.0, mouse_dy=0.0, w=False, s=False, a=False, d=False):
        self.camera.set_input(mouse_dx, mouse_dy, w, s, a, d)

    def set_lights(self, light_buffer, lights, count):
        """GUI light editor: re-upload the light array, set lightCount, restart the accumulation (GUI.cpp:125-130)."""
        light_buffer.update(lights)
        self.camera.buffer.lightCount = count
        self.camera.reset_accumulation()

    def resize(self, width, height, rows=None):
        """Resolution switch (Renderer.cpp:146-150,408-413): new accumulation target, camera vectors for the new aspect, restart."""
        self.width, self.height = width, height
        self.renderer.resize(width, rows if rows is not None else height)
        self.camera.update_resolution(width, height)

    # ---- frames
    def frame(self, dt=0.0):
        self.camera.update(dt)                       # Renderer::update: camera vectors, iterationCounter, randomSeed
        self.renderer.set_camera(self.camera.buffer)
        self.renderer.iterate()                      # Renderer::draw
        self.frames += 1
        if self.preview_every and self.frames % self.preview_every == 0:
            return self.preview()
        return None

    def run(self, frames, dt=0.0):
        last = None
        for _ in range(frames):
            out = self.frame(dt)
            last = out if out is not None else last
        return last

    def preview(self):
        """Gathers the tiles; rank 0 gets the assembled float frame (and calls on_preview), the other ranks None."""
        import torch
        local = torch.from_numpy(np.ascontiguousarray(self.renderer.framebuffer()))
        if self.dist is not None and self.world > 1 and self.dist.get_backend() == "nccl":
            local = local.cuda()
        frame = tiles.gather_tiles(local, self.width, self.height, self.rank, self.world, self.dist if self.world > 1 else None)
        frame = frame.cpu().numpy() if frame is not None else None
        self.previews += 1
        if frame is not None and self.on_preview is not None:
            self.on_preview(self.frames, frame)
        return frame
