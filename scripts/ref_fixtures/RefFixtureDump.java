/*
 * RefFixtureDump — runs a recorded coordinator stream through the REFERENCE's own classes and dumps
 * what they decide, so that tests/golden/ref_*.npz pin the oracle to the Java (SURVEY.md §8c).
 *
 * Lives in the reference's package on purpose: PaxosCoordinator's entry points are protected /
 * package-private.  It is compiled NEXT TO the reference's jar by scripts/make_ref_fixtures.sh on a box
 * that has a JDK (none exists in the build image or on the GPU box, so this file has never been
 * compiled: treat a compile error as a bug of this harness, not of the engine).
 *
 * Replays exactly the calls PaxosInstanceStateMachine makes on this path:
 *   PaxosCoordinator.hotRestore(null, HotRestoreInfo.createHRI(..))      PISM:677-690
 *   PaxosCoordinator.propose(c, members, request)                         PISM:833-851
 *   PaxosCoordinator.handleAcceptReply(c, members, reply)                 PISM:1248-1357
 *   PaxosCoordinator.isPreemptedFully(c, reply) -> coordinator = null     PISM:1361-1364
 *
 * Stream file (big-endian int32, written by scripts/ref_fixtures/make_stream.py):
 *   G K me R  members[K]
 *   R x { nProp  gidx[nProp]   nVotes  gidx[] bnum[] bcoord[] slot[] acceptor[] maxcp[] }
 * Output file (big-endian int32):
 *   R x { nProp x (slot bnum bcoord medianCp ok)   nDec  nDec x (voteIndex gidx slot bnum bcoord medianCp kind) }
 *   kind: 1 = DECISION, 2 = PREEMPTED (GPX_D_*); decisions in ARRIVAL order (the test regroups them by
 *   gidx with a stable sort, the engine's output order contract).
 */
package edu.umass.cs.gigapaxos;

import java.io.BufferedInputStream;
import java.io.BufferedOutputStream;
import java.io.DataInputStream;
import java.io.DataOutputStream;
import java.io.FileInputStream;
import java.io.FileOutputStream;

import edu.umass.cs.gigapaxos.paxospackets.AcceptPacket;
import edu.umass.cs.gigapaxos.paxospackets.AcceptReplyPacket;
import edu.umass.cs.gigapaxos.paxospackets.PValuePacket;
import edu.umass.cs.gigapaxos.paxospackets.PaxosPacket;
import edu.umass.cs.gigapaxos.paxospackets.RequestPacket;
import edu.umass.cs.gigapaxos.paxosutil.Ballot;
import edu.umass.cs.gigapaxos.paxosutil.HotRestoreInfo;

public class RefFixtureDump {
	private static int[] readInts(DataInputStream in, int n) throws Exception {
		int[] a = new int[n];
		for (int i = 0; i < n; i++)
			a[i] = in.readInt();
		return a;
	}

	public static void main(String[] args) throws Exception {
		DataInputStream in = new DataInputStream(new BufferedInputStream(new FileInputStream(args[0]), 1 << 20));
		DataOutputStream out = new DataOutputStream(new BufferedOutputStream(new FileOutputStream(args[1]), 1 << 20));
		int G = in.readInt(), K = in.readInt(), me = in.readInt(), R = in.readInt();
		int[] members = readInts(in, K);
		PaxosCoordinator[] coord = new PaxosCoordinator[G];
		for (int g = 0; g < G; g++)
			coord[g] = PaxosCoordinator.hotRestore(null, HotRestoreInfo.createHRI("g" + g, members, me));
		for (int r = 0; r < R; r++) {
			int nProp = in.readInt();
			int[] pg = readInts(in, nProp);
			for (int i = 0; i < nProp; i++) {
				RequestPacket req = new RequestPacket(((long) r << 32) | (long) i, "v", false);
				AcceptPacket a = PaxosCoordinator.propose(coord[pg[i]], members, req);
				if (a != null) {
					out.writeInt(a.slot);
					out.writeInt(a.ballot.ballotNumber);
					out.writeInt(a.ballot.coordinatorID);
					out.writeInt(a.getMedianCheckpointedSlot());
					out.writeInt(1);
				} else
					for (int q = 0; q < 5; q++)
						out.writeInt(0);
			}
			int nVotes = in.readInt();
			int[] vg = readInts(in, nVotes), vb = readInts(in, nVotes), vc = readInts(in, nVotes);
			int[] vs = readInts(in, nVotes), va = readInts(in, nVotes), vm = readInts(in, nVotes);
			java.util.ArrayList<int[]> dec = new java.util.ArrayList<int[]>();
			for (int i = 0; i < nVotes; i++) {
				AcceptReplyPacket reply = new AcceptReplyPacket(va[i], new Ballot(vb[i], vc[i]), vs[i], vm[i]);
				PValuePacket d = PaxosCoordinator.handleAcceptReply(coord[vg[i]], members, reply);
				if (PaxosCoordinator.isPreemptedFully(coord[vg[i]], reply))
					coord[vg[i]] = null;
				if (d != null)
					dec.add(new int[] { i, vg[i], d.slot, d.ballot.ballotNumber, d.ballot.coordinatorID,
							d.getMedianCheckpointedSlot(),
							d.getType() == PaxosPacket.PaxosPacketType.DECISION ? 1 : 2 });
			}
			out.writeInt(dec.size());
			for (int[] d : dec)
				for (int x : d)
					out.writeInt(x);
		}
		out.close();
		in.close();
	}
}
