"""Hash-sharding of the group index space across the GPUs of a node.

Groups are independent (PaxosManager.java:3170-3171), so every GPU holds only its shard's SoA rows
and no collective is needed on the decide path: the host routes each record to
`fmix32(gidx) % n_shards` and rewrites the global group index into the shard-local dense index.
"""
from __future__ import annotations

import numpy as np

from .streams import shard_of


class ShardMap:
    """global gidx <-> (shard, local gidx) for a fixed global group-index space."""

    def __init__(self, num_groups_global: int, n_shards: int):
        self.G = int(num_groups_global)
        self.n = int(n_shards)
        g = np.arange(self.G, dtype=np.int32)
        self.shard = shard_of(g, self.n)
        self.local = np.zeros(self.G, np.int32)
        self.counts = np.bincount(self.shard, minlength=self.n).astype(np.int64)
        self.globals_of = []
        for s in range(self.n):
            idx = np.nonzero(self.shard == s)[0].astype(np.int32)
            self.local[idx] = np.arange(idx.shape[0], dtype=np.int32)
            self.globals_of.append(idx)

    def route(self, cols, shard: int):
        """The sub-batch of `cols` (first column = global gidx) that belongs to `shard`, in the
        original record order, with gidx rewritten to the shard-local index.  Out-of-range gidx
        are kept on shard 0 as -1 (dropped there with GPX_S_NOGROUP)."""
        g = np.asarray(cols[0], np.int32)
        ok = (g >= 0) & (g < self.G)
        sh = np.zeros(g.shape[0], np.int32)
        sh[ok] = self.shard[g[ok]]
        sel = sh == shard
        out = [np.ascontiguousarray(np.asarray(c)[sel]) for c in cols]
        gl = np.full(out[0].shape[0], -1, np.int32)
        oks = ok[sel]
        gl[oks] = self.local[out[0][oks]]
        out[0] = gl
        return out, np.nonzero(sel)[0]

    def to_global(self, shard: int, local_gidx):
        return self.globals_of[shard][np.asarray(local_gidx, np.int64)]
