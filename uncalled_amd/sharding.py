"""Read sharding across GPUs: reads are independent units (each Mapper touches only read-local state and the
read-only index, mapper.hpp:80-85), so N ranks take contiguous slices balanced by sample count and never
exchange anything on the data path."""
import numpy as np


def shard_bounds(offsets, world):
    """offsets: uint64[n+1] cumulative sample offsets -> int64[world+1] read indices; shard r = [b[r], b[r+1])."""
    off = np.asarray(offsets, dtype=np.uint64)
    n = off.size - 1
    total = float(off[-1] - off[0])
    targets = off[0] + (np.arange(1, world, dtype=np.float64) * (total / world)).astype(np.uint64)
    cuts = np.searchsorted(off[1:], targets, side="left") + 1 if n else np.zeros(world - 1, dtype=np.int64)
    b = np.concatenate(([0], np.minimum(cuts, n), [n])).astype(np.int64)
    return np.maximum.accumulate(b)


def shard(offsets, rank, world):
    """-> (first_read, last_read_exclusive) of this rank."""
    b = shard_bounds(offsets, world)
    return int(b[rank]), int(b[rank + 1])
