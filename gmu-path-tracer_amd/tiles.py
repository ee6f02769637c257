"""Framebuffer tile split across ranks and the gather at accumulate time.

Pixels are independent, so the path shards with no data-path collective: every rank owns a private pipeline
(pool, queues, counters, accumulation tile) and renders a horizontal row band of the frame with the GLOBAL
camera (gmupt_renderer_desc.tile_*).  The only exchange is one gather of the disjoint tiles to rank 0 --
torch.distributed.gather (backend "nccl" = RCCL over xGMI on MI355X, "gloo" in the CPU tests); no reduction.
"""
import numpy as np


def row_bands(height, world_size):
    """[(y0, rows)] for each rank: contiguous row bands whose sizes differ by at most one row."""
    base, extra = divmod(height, world_size)
    out, y = [], 0
    for r in range(world_size):
        rows = base + (1 if r < extra else 0)
        out.append((y, rows))
        y += rows
    return out


def tile_budget(width, rows, spp):
    return width * rows * spp


def assemble(tiles, width, height, world_size):
    """Stacks per-rank row bands (each rows x width x 4 float32) into the full frame."""
    bands = row_bands(height, world_size)
    frame = np.zeros((height, width, 4), dtype=np.float32)
    for (y0, rows), t in zip(bands, tiles):
        frame[y0:y0 + rows] = np.asarray(t).reshape(-1, width, 4)[:rows]
    return frame


def gather_tiles(local_tile, width, height, rank, world_size, dist=None, device=None, force_collective=False, pad_rows=None):
    """Gathers the per-rank tiles on rank 0.  local_tile: torch tensor (rows, width, 4) float32.

    Bands may differ by one row, so every rank pads to the largest band.  Rank 0 returns the assembled frame as a torch tensor
    (height x width x 4) ON THE DEVICE OF THE TILES -- no host copy happens here, the read-back is the caller's business
    (frame.cpu().numpy()); the other ranks return None.
    A single rank needs no exchange and returns its tile -- unless force_collective is set (the RCCL test: the same gather call on a
    one-rank nccl group); pad_rows pads every band to at least that many rows (the same test, to run the padded branch there).
    """
    import torch
    if dist is None or (world_size == 1 and not force_collective):
        return local_tile.reshape(height, width, 4)
    bands = row_bands(height, world_size)
    max_rows = max(r for _, r in bands)
    if pad_rows is not None and pad_rows > max_rows:
        max_rows = pad_rows
    if local_tile.shape[0] == max_rows:
        padded = local_tile.contiguous()
    else:
        padded = torch.zeros((max_rows, width, 4), dtype=torch.float32, device=local_tile.device)
        padded[:local_tile.shape[0]] = local_tile
    gathered = [torch.empty_like(padded) for _ in range(world_size)] if rank == 0 else None
    dist.gather(padded, gathered, dst=0)
    if rank != 0:
        return None
    return torch.cat([g[:rows] for g, (_, rows) in zip(gathered, bands)], dim=0)
