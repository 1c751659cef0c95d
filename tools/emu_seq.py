import sys, time, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from autoforce_amd import _lib
from autoforce_amd.workloads import lips
numbers, pos, cell, pbc = lips(16, seed=0)
N, m = len(numbers), 512
mdl = bench.build_model(0, numbers, pos, cell, pbc, m)
lib = _lib.load(); h = mdl.handle
dev = torch.device('cuda', 0)
pos_d = torch.from_numpy(pos).to(dev); cell_d = torch.from_numpy(cell).to(dev)
packed = torch.zeros(int(lib.sgpr_packed_len(N)), dtype=torch.float64, device=dev)
sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
def run(world, graph, n=300):
    _lib.check(lib.sgpr_set_option(h, b"graph", graph))
    _lib.check(lib.sgpr_bind_system(h, N, _lib.ptr(_lib.i32(numbers)), _lib.ptr(_lib.i32(pbc.astype(np.int32))), 0, world))
    for _ in range(20):
        _lib.check(lib.sgpr_step_dev_next(h, pos_d.data_ptr(), cell_d.data_ptr(), packed.data_ptr(), pos_d.data_ptr(), sp))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        _lib.check(lib.sgpr_step_dev_next(h, pos_d.data_ptr(), cell_d.data_ptr(), packed.data_ptr(), pos_d.data_ptr(), sp))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
seq = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]]
for w, g in seq:
    print(f"world {w} graph {g}: {run(w, g):.1f} us", flush=True)
