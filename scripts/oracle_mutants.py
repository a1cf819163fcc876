#!/usr/bin/env python3
"""Would the second readings of the Java notice a wrong oracle?  Builds deliberately broken copies of the CPU oracle
(one small change each: a comparison flipped, a rule dropped, a field swapped), points the tests at each copy
(GPX_ORACLE_SO, tests/oracle_binding.py) and runs ONLY the tests that compare the oracle with an independent Python
reading of the reference (no hand-computed known answers, no golden fixtures): a mutant that survives marks a rule
the readings do not pin.  CPU only; test infrastructure like everything around oracle/.

  python scripts/oracle_mutants.py [-k substring | --skip N] > profiles/rNN_oracle_mutants.txt"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = {"cpp": os.path.join(ROOT, "oracle", "gpx_oracle.cpp"), "inc": os.path.join(ROOT, "oracle", "gpx_wire_oracle.inc")}

# the model-based tests, fastest first (pytest -x stops at the first failure)
READINGS = [
    "tests/test_host_rows_oracle.py::test_request_batcher_random_bursts_against_java_reading",
    "tests/test_host_rows_oracle.py::test_election_scan_random_groups_against_java_reading",
    "tests/test_election_oracle.py::test_election_begin_sequences_against_java_reading",
    "tests/test_wire_model.py",
    "tests/test_oracle_kat.py::test_pcs_main_accept_reply_tail_every_coin",
    "tests/test_oracle_kat.py::test_pcs_accept_reply_tail_with_checkpoint_slots_enumerated",
    "tests/test_oracle_kat.py::test_pcs_accept_replies_in_any_order_against_java_reading",
    "tests/test_oracle_kat.py::test_acceptor_side_at_the_int_wrap_against_java_reading",
    "tests/test_oracle_kat.py::test_whole_round_across_the_int_wrap",
    "tests/test_oracle_kat.py::test_whole_round_with_unusual_group_sizes",
    "tests/test_oracle_kat.py::test_pause_and_hot_restore_between_rounds_against_java_reading",
    "tests/test_oracle_kat.py::test_view_change_after_lossy_rounds_against_java_reading",
    "tests/test_oracle_kat.py::test_whole_round_against_the_two_java_readings_together",
    "tests/test_oracle_kat.py::test_acceptor_side_long_random_sequences_against_java_reading",
    "tests/test_oracle_kat.py::test_acceptor_side_enumerated_against_java_reading",
]

# (what is broken, file, exact text, replacement)
MUTANTS = [
    ("Ballot.compareTo ignores the coordinator id", "cpp", "    return jsub(coord, b.coord);\n  }\n  bool equals", "    return 0;\n  }\n  bool equals"),
    ("WaitforUtility.getIndex returns the FIRST match", "cpp", "      if ((*members)[i] == node) index = (int)i;", "      if ((*members)[i] == node && index < 0) index = (int)i;"),
    ("heardFromMajority: >= instead of >", "cpp", "return heardCount > (int)members->size() / 2;", "return heardCount >= (int)members->size() / 2;"),
    ("a repeated vote counts again", "cpp", "      if (!responded[index]) {\n        changed = true;\n        heardCount++;\n      }", "      changed = true;\n      heardCount++;"),
    ("getMedianMinus of an even group takes the upper median", "cpp", "copy.size() % 2 == 0 ? copy.size() / 2 - 1 : copy.size() / 2;", "copy.size() / 2;"),
    ("recordSlotNumber: <= instead of the plain <", "cpp", "        if (nodeSlotNumbers[i] < maxCP) nodeSlotNumbers[i] = maxCP;", "        if (nodeSlotNumbers[i] <= maxCP + 1) nodeSlotNumbers[i] = maxCP;"),
    ("recordSlotNumber wraparound-aware (jsub) instead of the plain <", "cpp", "        if (nodeSlotNumbers[i] < maxCP) nodeSlotNumbers[i] = maxCP;", "        if (jsub(nodeSlotNumbers[i], maxCP) < 0) nodeSlotNumbers[i] = maxCP;"),
    ("propose after a stop is not refused", "cpp", "    if (prev != myProposals.end() && prev->second.stop) return 1;", "    if (false && prev != myProposals.end() && prev->second.stop) return 1;"),
    ("acceptor accepts only strictly higher ballots", "cpp", "    if (accept.ballot.compareTo(getBallot()) >= 0) {", "    if (accept.ballot.compareTo(getBallot()) > 0) {"),
    ("accept at the GC slot itself is stored", "cpp", "      if (jsub(accept.slot, acceptedGCSlot) > 0) {", "      if (jsub(accept.slot, acceptedGCSlot) >= 0) {"),
    ("garbage collection may pass the next slot", "cpp", "    if (jsub(_slot, gcSlot) <= 0) gcSlot = jsub(_slot, 1);", "    if (false) gcSlot = jsub(_slot, 1);"),
    ("isPreemptable: >= instead of >", "cpp", "return b.compareTo(myBallot) > 0; } /* :271-279 */", "return b.compareTo(myBallot) >= 0; } /* :271-279 */"),
    ("carry-over keeps the FIRST pvalue of a slot, not the highest ballot", "cpp", "      if (ex == carryoverProposals.end() || b.compareTo(ex->second.ballot) > 0)", "      if (ex == carryoverProposals.end())"),
    ("toLog ignores the not-logged-again rule", "cpp", "                 (!havePrev || prevBallot.compareTo(accept.ballot) < 0);", "                 true;"),
    ("a pause does not look at pending decisions", "cpp", "      bool caughtUp = g->paxosState.committedRequests.empty() &&", "      bool caughtUp = true &&"),
    ("a pause does not look at outstanding proposals", "cpp", "                      (!g->coordinator || g->coordinator->myProposals.empty());", "                      true;"),
    ("HotRestoreInfo of a coordinator-less instance: nextProposalSlot 0", "cpp", "    r->next_proposal_slot = -1; /* getNextProposalSlotIfActive", "    r->next_proposal_slot = 0; /* getNextProposalSlotIfActive"),
    ("BatchedCommit median: last row's instead of the fold", "inc", "      if (jsub(d_median_cp[i], bc.median) > 0) bc.median = d_median_cp[i];", "      bc.median = d_median_cp[i];"),
    ("PREEMPTED rows are packed like decisions", "inc", "    if (d_kind[i] != GPX_D_DECISION) continue; /* allCoalescableDecisions", "    if (false) continue; /* allCoalescableDecisions"),
    ("a repeated slot keeps the FIRST request id", "inc", "          bar->slots[slot[i]] = req_id ? req_id[i] : 0; /* addAcceptReply: slots.put */", "          bar->slots.emplace(slot[i], req_id ? req_id[i] : 0); /* addAcceptReply: slots.put */"),
    ("NACKs are coalesced too", "inc", "      const bool coalescable = has_reply && named && (!sender || sender[i] == r_bcoord[i]);", "      const bool coalescable = has_reply && named;"),
    ("paxosID length is read as unsigned", "inc", "  const int8_t idLen = (int8_t)b.get();", "  const int32_t idLen = (int32_t)b.get();"),
    ("a null paxosID is fine in an accept reply", "inc", "        if (!pkt.hdr.has_id) throw JNullPointer();", "        if (false) throw JNullPointer();"),
    ("a negative digest length throws instead of being skipped", "inc", "  if (digestLength > 0) b.getBytes(digestLength);", "  if (digestLength != 0) b.getBytes(digestLength);"),
    ("nested stop requests are not seen", "inc", "    r.stop_any = r.stop_any || nested.stop_any;", "    r.stop_any = r.stop_any;"),
    ("version mismatch is ignored", "inc", "      else if (grp->version != pkt.hdr.version) {", "      else if (false) {"),
    ("election: long dead alone is enough (the node need not be down)", "inc", "      else if (!nodeUp && longDead)", "      else if (longDead)"),
    ("election: my own ballot does not count as mine", "inc", "      if (cur.coord == e->cfg.my_id)\n        why = GPX_RUN_MINE;\n      else if", "      if (false)\n        why = GPX_RUN_MINE;\n      else if"),
    # second batch
    ("garbage collection of decisions also drops the decision AT the slot", "cpp", "      if (jsub(slot, it->first) > 0)\n        it = committedRequests.erase(it);", "      if (jsub(slot, it->first) >= 0)\n        it = committedRequests.erase(it);"),
    ("reconstructDecision does not compare the accept's ballot with the commit's", "cpp", "      if (a != acceptedProposals.end() && a->second.ballot.equals(c->second.ballot)) {", "      if (a != acceptedProposals.end()) {"),
    ("a placeholder replaces a decision that has its value", "cpp", "      if (c == committedRequests.end() || !c->second.hasValue)\n        committedRequests[decision->slot] = *decision;", "      committedRequests[decision->slot] = *decision;"),
    ("a stopped instance keeps its pending decisions", "cpp", "    if (stopped) committedRequests.clear();", "    if (false) committedRequests.clear();"),
    ("accepts stay in memory although they come from disk", "cpp", "    if (haveNext && fromDisk) acceptedProposals.erase(next->slot);", "    if (false) acceptedProposals.erase(next->slot);"),
    ("a decision below the next slot is stored", "cpp", "    if (jsub(decision->slot, _slot) >= 0) {\n      auto c = committedRequests.find(decision->slot);", "    if (true) {\n      auto c = committedRequests.find(decision->slot);"),
    ("canIgnorePrepareReply: <= instead of <", "cpp", "    if (b.compareTo(myBallot) < 0) return true;\n    bool member = false;", "    if (b.compareTo(myBallot) <= 0) return true;\n    bool member = false;"),
    ("a second PREPARE_REPLY of the same acceptor is recorded again", "cpp", "    if (!member || (idx >= 0 && waitforMyBallot->responded[idx])) return true;", "    if (!member) return true;"),
    ("getMinSlot recorded with a plain <", "cpp", "      if (members[i] == acceptor && jsub(nodeSlotNumbers[i], minSlot) < 0) nodeSlotNumbers[i] = minSlot;", "      if (members[i] == acceptor && nodeSlotNumbers[i] > minSlot) nodeSlotNumbers[i] = minSlot;"),
    ("holes among the carried pvalues are not filled with no-ops", "cpp", "      } else if (pa == preActives.end()) {\n        ProposalState ps{false, WaitforUtility(&members)};\n        ps.kind = GPX_E_NOOP;\n        myProposals.emplace(cur, ps);", "      } else if (pa == preActives.end()) {"),
    ("duplicates of carried requests are proposed again", "cpp", "        if (!dup) myProposals.emplace(cur, pa->second);", "        myProposals.emplace(cur, pa->second);"),
    ("no stop is appended when a stop is not last", "cpp", "      if (last != myProposals.end() && !last->second.stop) {", "      if (false) {"),
    ("a coordinator with proposals left resigns on a higher ballot", "cpp", "    if (c && Ballot{bnum[i], bcoord[i]}.compareTo(c->myBallot) > 0 && c->preemptedFully())", "    if (c && Ballot{bnum[i], bcoord[i]}.compareTo(c->myBallot) > 0)"),
    ("RequestBatcher: >= instead of > on the size limit", "inc", "        if (nsize > max_size) break;", "        if (nsize >= max_size) break;"),
    ("RequestBatcher: >= instead of > on the byte limit", "inc", "        if (nbytes > max_bytes) break;", "        if (nbytes >= max_bytes) break;"),
    ("dequeueImpl: the bound is tested after the add", "inc", "    while (lengthEstimate < max_payload && !pending.empty()) {", "    while (lengthEstimate <= max_payload && !pending.empty()) {"),
    ("process(): >= MIN_PP_BATCH_SIZE tasks are regrouped", "inc", "(int32_t)pkts.size() > min_batch) {", "(int32_t)pkts.size() >= min_batch) {"),
    ("shouldSync: > instead of >= threshold", "inc", "    should_sync[i] = ((gap >= threshold) ||", "    should_sync[i] = ((gap > threshold) ||"),
    ("a meta-commit without its accept is not missing", "inc", "          (!c->second.hasValue && !a.acceptedProposals.count(s)))", "          false)"),
    ("handlePrepare adopts an equal ballot as an upgrade (logs again)", "inc", "    if (prep.compareTo(prev) > 0) {\n      a.ballotNum = prep.num;", "    if (prep.compareTo(prev) >= 0) {\n      a.ballotNum = prep.num;"),
    ("prepare replies carry accepted pvalues below firstUndecidedSlot", "inc", "        if (jsub(kv.first, first_slot[i]) < 0) continue;", "        if (false) continue;"),
    ("a BatchedCommit's slots keep their wire order", "inc", "          if (i > 0 && !(prev < s)) ascending = false;\n          prev = s;\n          pkt.c_slots.insert(s);", "          if (i > 0 && !(prev < s)) ascending = false;\n          prev = s;\n          if (pkt.c_slots.empty() || *pkt.c_slots.rbegin() < s) pkt.c_slots.insert(s);"),
    # third batch
    ("the first int of a frame is not checked", "inc", "    if (first != GPX_WT_PAXOS_PACKET) return GPX_W_MALFORMED; /* type == null -> fatal */", "    if (false) return GPX_W_MALFORMED; /* type == null -> fatal */"),
    ("PREEMPTED (13) is not a packet type", "inc", "case 9: case 13: case 21:", "case 9: case 21:"),
    ("a BatchedCommit's group includes this node", "inc", "      if (m != e->cfg.my_id) grp.insert(m);", "      grp.insert(m);"),
    ("a BatchedAcceptReply carries the LAST reply's maxCheckpointedSlot", "inc", "          bar->slots[slot[i]] = req_id ? req_id[i] : 0; /* addAcceptReply: slots.put */", "          { bar->slots[slot[i]] = req_id ? req_id[i] : 0; bar->maxcp = r_maxcp[i]; } /* addAcceptReply: slots.put */"),
    ("getNextCoordinator: the coordinator itself is next", "inc", "        next = g->members[(q + 1) % g->members.size()];", "        next = g->members[q % g->members.size()];"),
    ("getMaxCommittedSlot of a stopped instance looks at its decisions", "inc", "    if (!a.stopped && !a.committedRequests.empty()) {", "    if (!a.committedRequests.empty()) {"),
    ("getMissingCommittedSlots ignores the size limit", "inc", "    for (int32_t s = slot; jsub(s, maxc) < 0 && jsub(s, limit) < 0 && j < 64;", "    for (int32_t s = slot; jsub(s, maxc) < 0 && j < 64;"),
    ("the ACCEPT's sender is read before its median", "inc", "        pkt.median = b.getInt();\n        b.get(); /* noCoalesce */\n        pkt.sender = b.getInt();", "        pkt.sender = b.getInt();\n        b.get(); /* noCoalesce */\n        pkt.median = b.getInt();"),
    # fourth batch: plain comparisons where the Java subtracts (visible only where slots cross Integer.MAX_VALUE)
    ("a decision is stored iff its slot >= the next slot, compared plainly", "cpp", "    if (jsub(decision->slot, _slot) >= 0) {\n      auto c = committedRequests.find(decision->slot);", "    if (decision->slot >= _slot) {\n      auto c = committedRequests.find(decision->slot);"),
    ("the GC slot is clamped to the next slot with a plain compare", "cpp", "    if (jsub(_slot, gcSlot) <= 0) gcSlot = jsub(_slot, 1);", "    if (_slot <= gcSlot) gcSlot = jsub(_slot, 1);"),
    ("accepted pvalues are collected with a plain compare", "cpp", "        if (jsub(it->first, gcSlot) <= 0)\n          it = acceptedProposals.erase(it);", "        if (it->first <= gcSlot)\n          it = acceptedProposals.erase(it);"),
    ("an ACCEPT is stored iff its slot > the GC slot, compared plainly", "cpp", "      if (jsub(accept.slot, acceptedGCSlot) > 0) {", "      if (accept.slot > acceptedGCSlot) {"),
    ("the next proposal slot does not wrap (saturates)", "cpp", "    nextProposalSlotNumber = (int32_t)((uint32_t)nextProposalSlotNumber + 1u);\n    ProposalState ps{stop, WaitforUtility(&members)};", "    nextProposalSlotNumber = nextProposalSlotNumber == INT32_MAX ? INT32_MAX : nextProposalSlotNumber + 1;\n    ProposalState ps{stop, WaitforUtility(&members)};"),
    # fifth batch: makeCoordinator
    ("makeCoordinator replaces a coordinator of the SAME ballot too", "inc", "    if (!c || c->myBallot.compareTo(nb) < 0) {", "    if (!c || c->myBallot.compareTo(nb) <= 0) {"),
    ("the PREPARE of an ACTIVE coordinator of the same ballot is sent again", "inc", "    } else if (c->myBallot.compareTo(nb) == 0 && !c->active) {", "    } else if (c->myBallot.compareTo(nb) == 0) {"),
    ("poke: any outstanding proposal, not the acceptor's next slot", "inc", "      auto p = c->myProposals.find(s); /* isCommandering(slot) */", "      auto p = c->myProposals.begin(); /* isCommandering(slot) */"),
]


def main():
    pick = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == "-k" else None
    skip = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[1] == "--skip" else 0     # --skip N: leave out the first N
    text = {k: open(p).read() for k, p in SRC.items()}
    work = tempfile.mkdtemp(prefix="gpx_mutants_")
    os.makedirs(os.path.join(work, "include"))
    for f in ("gpx.h", "gpx_wire.h"):
        shutil.copy(os.path.join(ROOT, "include", f), os.path.join(work, "include", f))
    os.makedirs(os.path.join(work, "oracle"))
    killed = survived = 0
    print("# scripts/oracle_mutants.py: one deliberate fault in the oracle per line; the tests run are ONLY those that hold")
    print("# the oracle to an independent Python reading of the Java (tests/round_model.py, acc_enum_common.py,")
    print("# pcs_enum_common.py, wire_model.py, test_host_rows_oracle.py).  killed = some reading noticed.")
    for what, f, old, new in MUTANTS[skip:]:
        if pick and pick not in what:
            continue
        assert text[f].count(old) == 1, f"mutation site not unique / not found: {what}"
        for k in SRC:
            with open(os.path.join(work, "oracle", os.path.basename(SRC[k])), "w") as fh:
                fh.write(text[k].replace(old, new) if k == f else text[k])
        so = os.path.join(work, "libmutant.so")
        cc = subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-w", "-shared", "-o", so, os.path.join(work, "oracle", "gpx_oracle.cpp")],
                            stderr=subprocess.PIPE, text=True)
        if cc.returncode != 0:
            print(f"DOES NOT COMPILE  {what}: {cc.stderr[:200]}")
            continue
        t = time.time()
        env = dict(os.environ, GPX_ORACLE_SO=so)
        try:
            r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + READINGS, cwd=ROOT, env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
            out, rc = r.stdout, r.returncode
        except subprocess.TimeoutExpired as ex:
            out, rc = (ex.stdout or b"").decode() if isinstance(ex.stdout, bytes) else (ex.stdout or ""), 1
            out += "\nFAILED (timeout: the mutant hangs a test)"
        first = next((l for l in out.splitlines() if l.startswith("FAILED") or l.startswith("ERROR")), "")
        if rc != 0:
            killed += 1
            print(f"killed    {what:78s} {time.time() - t:5.0f} s  {first[:150]}")
        else:
            survived += 1
            print(f"SURVIVED  {what:78s} {time.time() - t:5.0f} s")
        sys.stdout.flush()
    print(f"# {killed} killed, {survived} survived")
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
