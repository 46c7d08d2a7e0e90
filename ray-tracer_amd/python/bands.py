"""Row-interleaved 8-row tile bands: band b of the image goes to rank b % N.
Each rank stores its bands back to back (RTX_F_BANDS_COMPACT); buffers are padded to the same
number of rows so the all-gather sends equal counts."""
import numpy as np

TILE = 8


def n_bands(height):
    return (height + TILE - 1) // TILE


def rows_padded(height, world_size):
    """rows every rank's compact buffer holds (max over ranks, whole bands)"""
    return (n_bands(height) + world_size - 1) // world_size * TILE


def rank_rows(height, rank, world_size):
    """image rows owned by `rank`, in the order they sit in its compact buffer"""
    rows = []
    for b in range(rank, n_bands(height), world_size):
        rows.extend(range(b * TILE, min(b * TILE + TILE, height)))
    return np.asarray(rows, dtype=np.int64)


def deinterleave(gathered, height):
    """gathered: (N, rows_padded, W, C) array (numpy or torch) as produced by the all-gather;
    returns the (height, W, C) image."""
    n = gathered.shape[0]
    out = None
    for r in range(n):
        rows = rank_rows(height, r, n)
        part = gathered[r][:len(rows)]
        if out is None:
            out = part.new_zeros((height,) + tuple(part.shape[1:])) if hasattr(part, "new_zeros") \
                else np.zeros((height,) + part.shape[1:], part.dtype)
        out[rows] = part
    return out
