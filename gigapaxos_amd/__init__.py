"""gigapaxos_amd — MI355X-native batched accept/decide engine for gigapaxos's
PaxosInstanceStateMachine hot path (see docs/HISTORY.md).  The compute path is the
hand-written HIP library gigapaxos_amd/csrc/libgpx_hip.so behind include/gpx.h."""
from ._abi import (  # noqa: F401
    Engine, GpxLib, GpxError, load_hip, hri_create, hri_initial, make_hri, HRI_DTYPE,
    hri_to_string, hri_from_string,
    S_OK, S_NOGROUP, S_STOPPED, S_WINDOW, S_FORWARD, S_REFUSED, S_EXISTS, S_BUSY, S_UNORDERED,
    ORDERED_PROPOSE, ORDERED_ACCEPT, ORDERED_COMMIT, ORDERED_REPLY_RUNS, TRY_REPLY_RUNS, LAZY_OUTPUTS,
    D_DECISION, D_PREEMPTED, R_TOLOG, R_STORED, A_STOP, C_HASVALUE, C_STOP,
    F_ACCEPTS_FROM_DISK, RETIRE_PAUSE, RETIRE_KILL,
)
