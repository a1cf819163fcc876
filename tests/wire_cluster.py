"""A replica cluster that talks in WIRE FRAMES only (test driver).

Every hop of one consensus round crosses the byte formats a gigapaxos node puts on the wire:

  client REQUEST frames -> coordinator: gpx_wire_decode -> gpx_propose_batch
  -> ACCEPT frames (the host stamps slot / ballot onto the request bytes, AcceptPacket.toBytes)
  -> every replica: gpx_wire_decode -> gpx_accept_batch -> gpx_wire_pack_accept_replies
  -> BATCHED_ACCEPT_REPLY frames -> coordinator: gpx_wire_decode -> gpx_accept_reply_batch
  -> gpx_wire_pack_commits -> BATCHED_COMMIT frames -> the other replicas: gpx_wire_decode ->
  gpx_commit_batch; the coordinator's own copy of a decision short-circuits as a full DECISION
  (SHORT_CIRCUIT_LOCAL, PaxosManager.java:2116-2128).

Run once over the HIP library and once over the oracle; the tests compare every frame and every
execution log."""
import numpy as np

from gigapaxos_amd import Engine, hri_create, S_OK, D_DECISION, C_HASVALUE
from gigapaxos_amd import wire as W


class WireCluster:
    def __init__(self, lib, node_ids, names, coordinator_of, window=8, max_batch=1 << 16):
        self.ids = sorted(node_ids)
        self.k = len(self.ids)
        self.names = names
        G = len(names)
        self.G = G
        self.coord = np.asarray(coordinator_of, np.int32)
        self.eng, self.wire = {}, {}
        members = np.tile(np.array(self.ids, np.int32), (G, 1))
        for nid in self.ids:
            e = Engine(lib, nid, G, kmax=self.k, window=window, max_batch=max_batch)
            we = W.WireEngine(e)
            assert (e.create_groups(np.arange(G), members, self.k, hri_create(G, self.k, self.coord)) == S_OK).all()
            assert (we.bind(names, np.arange(G)) == S_OK).all()
            self.eng[nid], self.wire[nid] = e, we
        self.exec_log = {nid: [] for nid in self.ids}
        self.trace = []  # every frame that crossed the wire, in order

    def close(self):
        for e in self.eng.values():
            e.close()

    def round(self, groups, rnd, value_len=64):
        """One request per listed group, sent to the group's coordinator."""
        groups = np.asarray(groups, np.int32)
        inbox_acc = {nid: [] for nid in self.ids}
        for c in self.ids:
            mine = groups[self.coord[groups] == c]
            if mine.size == 0:
                continue
            req_frames = [W.request(self.names[g], 0, (rnd << 32) | int(g), bytes([g & 0xFF]) * value_len)
                          for g in mine]
            self.trace += req_frames
            d = self.wire[c].decode(req_frames)
            assert (d.f_status == W.W_OK).all()
            rq = d.requests
            slot, bnum, bcoord, med, st = self.eng[c].propose(rq["gidx"], rq["is_stop"])
            assert (st == S_OK).all()
            for i in range(rq["gidx"].shape[0]):
                g = int(rq["gidx"][i])
                acc = W.accept(self.names[g], 0, int(rq["req_id"][i]), int(slot[i]), int(bnum[i]), int(bcoord[i]),
                               int(med[i]), c, bytes([g & 0xFF]) * value_len)
                for nid in [c] + [n for n in self.ids if n != c]:
                    inbox_acc[nid].append(acc)
        inbox_bar = {nid: [] for nid in self.ids}
        for nid in self.ids:
            if not inbox_acc[nid]:
                continue
            self.trace += inbox_acc[nid]
            d = self.wire[nid].decode(inbox_acc[nid])
            assert (d.f_status == W.W_OK).all()
            a = d.accepts
            (rb, rc, rm, rf, st), runs = self.eng[nid].accept(a["gidx"], a["bnum"], a["bcoord"], a["slot"],
                                                              a["median_cp"], a["flags"])
            assert runs.gidx.shape[0] == 0
            frames, fg, fd, ub, _ = self.wire[nid].pack_accept_replies(a["gidx"], a["slot"], rb, rc, rm, st,
                                                                       sender=a["sender"], req_id=a["req_id"])
            assert not ub.any()
            for f, dest in zip(frames, fd):
                inbox_bar[int(dest)].append(f)
        inbox_bc = {nid: [] for nid in self.ids}
        decisions = {}
        for c in self.ids:
            if not inbox_bar[c]:
                continue
            self.trace += inbox_bar[c]
            d = self.wire[c].decode(inbox_bar[c])
            assert (d.f_status == W.W_OK).all()
            v = d.votes
            dec = self.eng[c].accept_reply(v["gidx"], v["bnum"], v["bcoord"], v["slot"], v["acceptor"], v["max_cp"])
            decisions[c] = dec
            frames, fg, _ = self.wire[c].pack_commits(dec)
            for nid in self.ids:
                if nid != c:
                    inbox_bc[nid] += frames
            # local short circuit: the coordinator handles its own decisions as full DECISIONs
            sel = dec.kind == D_DECISION
            st, runs = self.eng[c].commit(dec.gidx[sel], dec.bnum[sel], dec.bcoord[sel], dec.slot[sel],
                                          dec.median_cp[sel], np.full(int(sel.sum()), C_HASVALUE, np.uint8))
            assert (st == S_OK).all()
            self.exec_log[c].append(runs.as_tuple_array())
        for nid in self.ids:
            if not inbox_bc[nid]:
                continue
            self.trace += inbox_bc[nid]
            d = self.wire[nid].decode(inbox_bc[nid])
            assert (d.f_status == W.W_OK).all()
            cm = d.commits
            st, runs = self.eng[nid].commit(cm["gidx"], cm["bnum"], cm["bcoord"], cm["slot"], cm["median_cp"],
                                            cm["kind"])
            assert (st == S_OK).all()
            self.exec_log[nid].append(runs.as_tuple_array())
        return decisions

    def executed(self, nid):
        return np.concatenate(self.exec_log[nid]) if self.exec_log[nid] else np.zeros((0, 3), np.int32)
