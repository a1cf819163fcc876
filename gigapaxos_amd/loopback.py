"""In-process replica cluster wired over engines — no sockets.

Host-side plumbing that plays the role of PaxosManager.send / sendOrLoopback
(PaxosManager.java:2098-2128) and PaxosPacketBatcher (PaxosPacketBatcher.java:121-209)
between several engine handles in one process: one engine per emulated replica (node id),
whole rounds of PROPOSE -> ACCEPT xK -> ACCEPT_REPLY xK -> DECISION -> BATCHED_COMMIT xK ->
in-order execution moved as SoA batches.  It is what BASELINE config #1 (loopback, 1 group,
3 replicas) and config #2 (10k groups, full pipeline) run on.

Message order mirrors the reference's loopback short-circuit (SHORT_CIRCUIT_LOCAL,
PaxosManager.java:2116-2128): the coordinator's own copy of a multicast is delivered
first, then the other members in ascending node-id order.
"""
from __future__ import annotations

import numpy as np

from ._abi import Engine, S_OK, D_DECISION, hri_create


class LoopbackCluster:
    def __init__(self, lib, node_ids, num_groups, window=8, max_batch=1 << 20, coordinator=None,
                 rows_fn=hri_create):
        self.node_ids = sorted(int(x) for x in node_ids)
        self.k = len(self.node_ids)
        self.G = num_groups
        self.engines = {
            nid: Engine(lib, nid, num_groups, kmax=self.k, window=window, max_batch=max_batch)
            for nid in self.node_ids
        }
        gidx = np.arange(num_groups, dtype=np.int32)
        members = np.tile(np.array(self.node_ids, np.int32), (num_groups, 1))
        # coordinator(g): default = all groups coordinated by node_ids[0] unless a per-group
        # array is given (host computes roundRobinCoordinator from the group NAME, PISM:2251-2256)
        if coordinator is None:
            coordinator = np.full(num_groups, self.node_ids[0], np.int32)
        self.coordinator = np.asarray(coordinator, np.int32)
        for nid, e in self.engines.items():
            rows = rows_fn(num_groups, self.k, self.coordinator)
            st = e.create_groups(gidx, members, self.k, rows)
            assert (st == S_OK).all()
        # executed slots per replica: list of (gidx, first, count) arrays in arrival order
        self.exec_log = {nid: [] for nid in self.node_ids}
        self.decision_log = []

    def close(self):
        for e in self.engines.values():
            e.close()

    def _delivery_order(self, coord):
        return [coord] + [n for n in self.node_ids if n != coord]

    def round(self, gidx, is_stop=None):
        """One consensus round: every listed group proposes one request at its coordinator.
        Returns the decisions (as (g, slot, bnum, bcoord, median, kind) rows)."""
        gidx = np.asarray(gidx, np.int32)
        all_dec = []
        for coord in self.node_ids:
            sel = self.coordinator[gidx] == coord
            if not sel.any():
                continue
            g = gidx[sel]
            stp = None if is_stop is None else np.asarray(is_stop, np.uint8)[sel]
            ce = self.engines[coord]
            slot, bnum, bcoord, median, st = ce.propose(g, stp)
            ok = st == S_OK
            g, slot, bnum, bcoord, median = g[ok], slot[ok], bnum[ok], bcoord[ok], median[ok]
            aflags = None if stp is None else stp[ok]
            order = self._delivery_order(coord)
            # ACCEPT multicast -> ACCEPT_REPLY per acceptor (loopback first)
            votes = []
            for nid in order:
                (rb, rc, rmax, rfl, ast), runs = self.engines[nid].accept(g, bnum, bcoord, slot, median, aflags)
                self._log_runs(nid, runs)
                okk = ast == S_OK
                votes.append((g[okk], rb[okk], rc[okk], slot[okk], np.full(okk.sum(), nid, np.int32), rmax[okk]))
            # replies arrive at the coordinator acceptor by acceptor
            dec = []
            for v in votes:
                d = ce.accept_reply(*v)
                dec.append(d.as_tuple_array())
            dec = np.concatenate(dec) if dec else np.zeros((0, 6), np.int32)
            all_dec.append(dec)
            dd = dec[dec[:, 5] == D_DECISION]
            # DECISION -> BATCHED_COMMIT multicast to the group (loopback first)
            for nid in order:
                st2, runs = self.engines[nid].commit(dd[:, 0], dd[:, 2], dd[:, 3], dd[:, 1], dd[:, 4])
                self._log_runs(nid, runs)
        out = np.concatenate(all_dec) if all_dec else np.zeros((0, 6), np.int32)
        self.decision_log.append(out)
        return out

    def _log_runs(self, nid, runs):
        if runs.gidx.shape[0]:
            self.exec_log[nid].append(runs.as_tuple_array())

    def executed(self, nid):
        """All exec runs of replica nid as one (m,3) array in arrival order."""
        if not self.exec_log[nid]:
            return np.zeros((0, 3), np.int32)
        return np.concatenate(self.exec_log[nid])
