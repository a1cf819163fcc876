/* Test infrastructure: builds the C++ host layer (gigapaxos_amd/host) against the CPU oracle instead
 * of libgpx_hip.so by renaming the C-ABI symbols it uses (force-included before include/gpx.h, so the
 * declarations are renamed too).  The product build never sees this file. */
#define gpx_engine_create orc_engine_create
#define gpx_engine_destroy orc_engine_destroy
#define gpx_last_error orc_last_error
#define gpx_rows_alloc orc_rows_alloc
#define gpx_rows_free orc_rows_free
#define gpx_group_create orc_group_create
#define gpx_group_retire orc_group_retire
#define gpx_names_bind orc_names_bind
#define gpx_names_unbind orc_names_unbind
#define gpx_wire_decode orc_wire_decode
#define gpx_propose_batch orc_propose_batch
#define gpx_accept_batch orc_accept_batch
#define gpx_accept_reply_batch orc_accept_reply_batch
#define gpx_commit_batch orc_commit_batch
#define gpx_wire_pack_accept_replies orc_wire_pack_accept_replies
#define gpx_wire_pack_commits orc_wire_pack_commits
#define gpx_propose_batch_h orc_propose_batch_h
#define gpx_election_scan orc_election_scan
#define gpx_election_begin orc_election_begin
#define gpx_prepare_batch orc_prepare_batch
#define gpx_prepare_reply_batch orc_prepare_reply_batch
#define gpx_request_batch orc_request_batch
#define gpx_gap_scan orc_gap_scan
#define gpx_poke_scan orc_poke_scan
#define gpx_group_snapshot orc_group_snapshot
