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
NEXT = True   # the last kernel bins the (same) next frame: 5 launches per step
worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
for world in worlds:
    for graph in ((0, 1) if len(worlds) > 1 else (0,)):
        _lib.check(lib.sgpr_set_option(h, b"graph", graph))
        _lib.check(lib.sgpr_bind_system(h, N, _lib.ptr(_lib.i32(numbers)), _lib.ptr(_lib.i32(pbc.astype(np.int32))), 0, world))
        for _ in range(20):
            _lib.check(lib.sgpr_step_dev_next(h, pos_d.data_ptr(), cell_d.data_ptr(), packed.data_ptr(), pos_d.data_ptr() if NEXT else None, sp))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            _lib.check(lib.sgpr_step_dev_next(h, pos_d.data_ptr(), cell_d.data_ptr(), packed.data_ptr(), pos_d.data_ptr() if NEXT else None, sp))
        torch.cuda.synchronize()
        rb = C.c_int64(0); _lib.check(lib.sgpr_get_list_rebuilds(h, C.addressof(rb)))
        print(f"world={world} rank0 share, graph={graph}: {(time.perf_counter()-t0)/300*1e6:.1f} us/step (list rebuilds so far: {rb.value})")
    if world > 1:
        mdl.profile(True)
        acc = {}
        for _ in range(30):
            _lib.check(lib.sgpr_step_dev_next(h, pos_d.data_ptr(), cell_d.data_ptr(), packed.data_ptr(), pos_d.data_ptr() if NEXT else None, sp))
            torch.cuda.synchronize()
            for k, v in mdl.stage_times().items():
                acc[k] = acc.get(k, 0.0) + v / 30
        mdl.profile(False)
        print("   stages (us, incl. ~3 us marker each):", {k: round(v * 1e3, 1) for k, v in acc.items()})
