import sys, numpy as np
sys.path.insert(0, '/root/repo')
from gigapaxos_amd import load_hip, Engine, hri_create
from tests.oracle_binding import load_oracle
for lib in (load_hip(), load_oracle()):
    e = Engine(lib, 100, 4, kmax=3, window=64)
    mem = np.tile(np.array([100,104,105],np.int32),(4,1))
    e.create_groups(np.arange(4), mem, 3, hri_create(4,3,100))
    print(e.commit([1],[0],[100],[14],[15],[0]))
    print(e.dump(1).tolist())
    print(e.accept([1],[0],[100],[14],[14],[0]))
    print(e.dump(1).tolist())
    # several accepts in one batch
    print(e.accept([1,1,1],[0,0,0],[100,100,100],[6,14,13],[8,14,38],[0,0,0]))
    print(e.dump(1).tolist())
