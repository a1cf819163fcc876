"""A Python reading of the reference's byte constructors - written from the Java, not from oracle/gpx_wire_oracle.inc -
for checking what gpx_wire_decode makes of a burst of frames, well-formed or damaged:

  PaxosPacketDemultiplexerFast.toPaxosPacket   paxosutil/PaxosPacketDemultiplexerFast.java:66-103
  PaxosPacket(ByteBuffer)                      paxospackets/PaxosPacket.java:443-458
  RequestPacket(ByteBuffer), isStopRequest     paxospackets/RequestPacket.java:956-1020, 1069-1080
  ProposalPacket / PValuePacket / AcceptPacket paxospackets/ProposalPacket.java:76-80, PValuePacket.java:126-133,
                                               AcceptPacket.java:87-91
  BatchedCommit(ByteBuffer)                    paxospackets/BatchedCommit.java:156-170
  AcceptReplyPacket / BatchedAcceptReply       paxospackets/AcceptReplyPacket.java:150-168, BatchedAcceptReply.java:103-117
  PaxosManager.handlePaxosPacket's demux       PaxosManager.java:1148-1194 (instance by name, version equal)

What the Java does with a frame it cannot parse is throw (BufferUnderflowException, NegativeArraySizeException,
NullPointerException on a null paxosID) and drop it: status W_MALFORMED here.  A known type without a byte
constructor falls through the switch (`assert (false)`, the packet stays null): W_UNSUPPORTED."""
import struct

W_OK, W_NOGROUP, W_VERSION, W_MALFORMED, W_UNSUPPORTED = 0, 1, 2, 3, 4
PAXOS_PACKET, REQUEST, ACCEPT, BATCHED_ACCEPT_REPLY, BATCHED_COMMIT = 90, 1, 3, 34, 35
# PaxosPacket.PaxosPacketType (PaxosPacket.java:202-297): the ints getPaxosPacketType() knows
PACKET_TYPES = {1, 2, 3, 4, 5, 6, 7, 8, 9, 13, 21, 23, 32, 33, 34, 35, 36, 37, 90, 9999}
# the engine's own limit on a slot list that is NOT ascending (include/gpx_wire.h GPX_W_MAX_UNSORTED); not the Java's
MAX_UNSORTED = 1024


class Thrown(Exception):
    """any of the three exceptions above"""


class ByteBuffer:
    """java.nio.ByteBuffer.wrap(bytes): big-endian relative gets"""

    def __init__(self, b):
        self.b, self.pos = bytes(b), 0

    def _take(self, n):
        if n < 0:
            raise Thrown("NegativeArraySizeException")      # new byte[n]
        if len(self.b) - self.pos < n:
            raise Thrown("BufferUnderflowException")
        out = self.b[self.pos:self.pos + n]
        self.pos += n
        return out

    def get(self):
        return struct.unpack(">b", self._take(1))[0]

    def get_short(self):
        return struct.unpack(">h", self._take(2))[0]

    def get_int(self):
        return struct.unpack(">i", self._take(4))[0]

    def get_long(self):
        return struct.unpack(">q", self._take(8))[0]

    def get_bytes(self, n):
        return self._take(n)

    def rewind(self):
        self.pos = 0


class Packet:
    pass


def paxos_packet(bbuf, p):
    bbuf.get_int()                                  # packet type
    p.packet_type = bbuf.get_int()
    p.version = bbuf.get_int()
    id_len = bbuf.get()                             # a byte: 128 .. 255 are negative lengths
    id_bytes = bbuf.get_bytes(id_len)
    p.paxos_id = id_bytes if len(id_bytes) > 0 else None


def request_packet(bbuf, p):
    paxos_packet(bbuf, p)
    p.request_id = bbuf.get_long()
    p.stop = bbuf.get() == 1
    bbuf.get_bytes(4), bbuf.get_short()             # client address
    bbuf.get_bytes(4), bbuf.get_short()             # listen address
    bbuf.get_int(), bbuf.get_long()                 # entryReplica, entryTime
    bbuf.get(), bbuf.get_int()                      # shouldReturnRequestValue, forwardCount
    bbuf.get()                                      # broadcasted
    digest_length = bbuf.get_int()
    if digest_length > 0:
        bbuf.get_bytes(digest_length)
    bbuf.get_bytes(bbuf.get_int())                  # requestValue
    bbuf.get_bytes(bbuf.get_int())                  # responseValue
    p.batched = None
    num_batched = bbuf.get_int()
    if num_batched == 0:
        return
    if num_batched < 0:
        raise Thrown("NegativeArraySizeException")  # new RequestPacket[numBatched]
    p.batched = []
    for _ in range(num_batched):
        element = bbuf.get_bytes(bbuf.get_int())
        q = Packet()
        request_packet(ByteBuffer(element), q)      # new RequestPacket(element): a buffer of its own
        p.batched.append(q)


def is_stop_request(p):
    return p.stop or any(is_stop_request(q) for q in (p.batched or ()))


def accept_packet(bbuf, p):
    request_packet(bbuf, p)
    p.slot = bbuf.get_int()                         # ProposalPacket
    p.ballot = (bbuf.get_int(), bbuf.get_int())     # PValuePacket
    bbuf.get()                                      # recovery
    p.median = bbuf.get_int()
    bbuf.get()                                      # noCoalesce
    p.sender = bbuf.get_int()                       # AcceptPacket


def _ascending(seq):
    return all(a < b for a, b in zip(seq, seq[1:]))


def batched_commit(bbuf, p):
    paxos_packet(bbuf, p)
    p.ballot = (bbuf.get_int(), bbuf.get_int())
    p.median = bbuf.get_int()
    listed = [bbuf.get_int() for _ in range(bbuf.get_int())]
    p.slots = sorted(set(listed))                   # TreeSet<Integer>
    p.unsorted_long = not _ascending(listed) and len(listed) > MAX_UNSORTED
    for _ in range(bbuf.get_int()):                 # group
        bbuf.get_int()


def batched_accept_reply(bbuf, p):
    paxos_packet(bbuf, p)
    if p.paxos_id is None:
        raise Thrown("NullPointerException")        # this.getPaxosID().getBytes(CHARSET)
    p.acceptor = bbuf.get_int()
    p.ballot = (bbuf.get_int(), bbuf.get_int())
    bbuf.get_int()                                  # slotNumber
    p.maxcp = bbuf.get_int()
    bbuf.get_long()                                 # requestID
    bbuf.get()                                      # digest request
    listed = []
    for _ in range(bbuf.get_int()):
        listed.append(bbuf.get_int())
        bbuf.get_long()
    p.slots = sorted(set(listed))                   # TreeMap<Integer, Long> keys
    p.unsorted_long = not _ascending(listed) and len(listed) > MAX_UNSORTED


def to_paxos_packet(frame):
    """-> (status, type or -1, packet or None)"""
    try:
        bbuf = ByteBuffer(frame)
        t = bbuf.get_int()
        t = bbuf.get_int() if t == PAXOS_PACKET else None
        if t is None or t not in PACKET_TYPES:
            return W_MALFORMED, -1, None            # fatal(bytes)
        bbuf.rewind()
        p = Packet()
        if t == REQUEST:
            request_packet(bbuf, p)
        elif t == ACCEPT:
            accept_packet(bbuf, p)
        elif t == BATCHED_COMMIT:
            batched_commit(bbuf, p)
        elif t == BATCHED_ACCEPT_REPLY:
            batched_accept_reply(bbuf, p)
        else:
            return W_UNSUPPORTED, t, None
        if getattr(p, "unsorted_long", False):
            return W_MALFORMED, -1, None
        return W_OK, t, p
    except Thrown:
        return W_MALFORMED, -1, None


def decode(frames, instances):
    """instances: paxosID bytes -> (row, version) of the instances that exist.  -> (per frame (status, row, type),
    votes, commits, accepts, requests) with the records in frame order and, within a frame, in the order the
    reference's handlers walk them (TreeMap / TreeSet order)."""
    per_frame, votes, commits, accepts, requests = [], [], [], [], []
    for i, f in enumerate(frames):
        st, t, p = to_paxos_packet(f)
        g = -1
        if st == W_OK:
            inst = instances.get(p.paxos_id) if p.paxos_id is not None else None
            if inst is None:
                st = W_NOGROUP
            else:
                g = inst[0]
                if inst[1] != p.version:
                    st = W_VERSION
        if st == W_OK:
            if t == BATCHED_ACCEPT_REPLY:
                votes += [(g, p.ballot[0], p.ballot[1], s, p.acceptor, p.maxcp, i) for s in p.slots]
            elif t == BATCHED_COMMIT:
                commits += [(g, p.ballot[0], p.ballot[1], s, p.median, 0, i) for s in p.slots]
            elif t == ACCEPT:
                accepts.append((g, p.ballot[0], p.ballot[1], p.slot, p.median, int(is_stop_request(p)), p.sender, p.request_id, i))
            else:
                requests.append((g, int(is_stop_request(p)), p.request_id, i))
        per_frame.append((st, g, t))
    return per_frame, votes, commits, accepts, requests


def check_decode(we, frames, instances, tag=""):
    """gpx_wire_decode of `frames` by the library behind `we` against decode() above; returns the number of frames
    of each status"""
    d = we.decode(frames)
    per_frame, votes, commits, accepts, requests = decode(frames, instances)
    got = list(zip(d.f_status.tolist(), d.f_gidx.tolist(), d.f_type.tolist()))
    for i, (a, b) in enumerate(zip(got, per_frame)):
        assert a == b, f"{tag} frame {i} ({len(frames[i])} bytes, {frames[i][:24].hex()}...): {a} != {b}"
    cols = {"votes": ("gidx", "bnum", "bcoord", "slot", "acceptor", "max_cp", "frame"),
            "commits": ("gidx", "bnum", "bcoord", "slot", "median_cp", "kind", "frame"),
            "accepts": ("gidx", "bnum", "bcoord", "slot", "median_cp", "flags", "sender", "req_id", "frame"),
            "requests": ("gidx", "is_stop", "req_id", "frame")}
    for cls, want in (("votes", votes), ("commits", commits), ("accepts", accepts), ("requests", requests)):
        have = getattr(d, cls)
        rows = list(zip(*[have[k].tolist() for k in cols[cls]]))
        assert rows == want, f"{tag} {cls}: {len(rows)} records against {len(want)}"
    assert d.counts == {"n_votes": len(votes), "n_commits": len(commits), "n_accepts": len(accepts),
                        "n_requests": len(requests), "n_bad_frames": sum(st != W_OK for st, _, _ in per_frame)}
    hist = [0] * 5
    for st, _, _ in per_frame:
        hist[st] += 1
    return hist


# ---- the sending side: PaxosPacketBatcher's coalescing and the two toBytes -----------------------------------------
#   PaxosPacketBatcher.enqueueImpl(BatchedCommit / AcceptReplyPacket)   PaxosPacketBatcher.java:121-156
#   allCoalescableDecisions / allPositiveAcceptReplies                 PaxosPacketBatcher.java:438-455
#   BatchedCommit(PValuePacket, group), addCommit / addBatchedCommit   paxospackets/BatchedCommit.java:62-69, 97-119
#   BatchedAcceptReply(AcceptReplyPacket), addAcceptReply              paxospackets/BatchedAcceptReply.java:52-57, 179-184
#   PaxosPacket.toBytes, BatchedCommit.toBytes, AcceptReplyPacket.toBytes, BatchedAcceptReply.toBytes
#                                   PaxosPacket.java:460-475, BatchedCommit.java:184-215, AcceptReplyPacket.java:174-184,
#                                   BatchedAcceptReply.java:119-173
# The Java's dequeue order is HashMap iteration order (none); the order of the frames is the library's contract
# (include/gpx_wire.h): commits by first row, accept replies by row ascending with a row's ballots in
# first-appearance order.  Every decision is enqueued as a task of its own here (one fold per row).

def _i32(x):
    return ((x + 2**31) % 2**32) - 2**31


def _header_bytes(ptype, version, name):
    return struct.pack(">iiib", PAXOS_PACKET, ptype, version, len(name)) + name    # (byte) paxosIDBytes.length, <= 127 here


def pack_commits(rows, info, my_id, decision_kind):
    """rows: (g, slot, bnum, bcoord, median, kind); info[g] = (paxosID bytes, version, members) of a named instance,
    absent otherwise -> (frames, their rows)"""
    commits = {}                                     # (paxosID row, ballot) -> [median, TreeSet]; dicts keep insertion order
    for g, slot, bnum, bcoord, median, kind in rows:
        if kind != decision_kind:                    # isCoalescable: decisions only
            continue
        c = commits.get((g, bnum, bcoord))
        if c is None:
            commits[(g, bnum, bcoord)] = [median, {slot}]
        else:
            if _i32(median - c[0]) > 0:              # Java int subtraction
                c[0] = median
            c[1].add(slot)
    frames, fg = [], []
    for (g, bnum, bcoord), (median, slots) in commits.items():
        if g not in info:
            continue
        name, version, members = info[g]
        group = sorted(m for m in members if m != my_id)            # Util.arrayToIntSet(Util.filter(recipients, myID))
        out = _header_bytes(BATCHED_COMMIT, version, name) + struct.pack(">iiii", bnum, bcoord, median, len(slots))
        out += b"".join(struct.pack(">i", s) for s in sorted(slots))
        out += struct.pack(">i", len(group)) + b"".join(struct.pack(">i", m) for m in group)
        frames.append(out)
        fg.append(g)
    return frames, fg


def pack_accept_replies(rows, info, my_id, ok_status=0):
    """rows: (g, slot, bnum, bcoord, maxcp, status, sender, req_id) = the replies of one batch of ACCEPTs in their order
    -> (frames, their rows, their destinations, per reply: 1 = exists but leaves as a plain ACCEPT_REPLY)"""
    replies, unbatched = {}, []
    for g, slot, bnum, bcoord, maxcp, status, sender, req_id in rows:
        exists = status == ok_status and g >= 0
        if not (exists and g in info and bcoord == sender):         # allPositiveAcceptReplies; the row must have a name
            unbatched.append(int(exists))
            continue
        unbatched.append(0)
        ar_map = replies.setdefault(g, {})
        bar = ar_map.get((bnum, bcoord))
        if bar is None:
            ar_map[(bnum, bcoord)] = [slot, maxcp, req_id, {slot: req_id}]   # new BatchedAcceptReply(acceptReply)
        else:
            bar[3][slot] = req_id                                            # slots.put
    frames, fg, fd = [], [], []
    for g in sorted(replies):
        name, version, _ = info[g]
        for (bnum, bcoord), (first_slot, maxcp, first_req, slots) in replies[g].items():
            assert len(replies[g]) <= 4 and sum(len(b[3]) for b in replies[g].values()) <= 256, "beyond one pass of the engine"
            out = _header_bytes(BATCHED_ACCEPT_REPLY, version, name)
            out += struct.pack(">iiiiiqb", my_id, bnum, bcoord, first_slot, maxcp, first_req, 0)
            out += struct.pack(">i", len(slots)) + b"".join(struct.pack(">iq", s, slots[s]) for s in sorted(slots))
            frames.append(out)
            fg.append(g)
            fd.append(bcoord)
    return frames, fg, fd, unbatched
