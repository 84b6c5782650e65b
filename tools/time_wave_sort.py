import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlap_amd import ops
lib, h = ops._handle(torch.device("cuda", 0))
rng = np.random.RandomState(0)
for n, kind in [(38, "ties"), (380, "ties"), (380, "distinct"), (380, "allequal"), (500, "ties")]:
    narr = 2048
    arrays = []
    for a in range(narr):
        if kind == "ties": k = np.concatenate([np.ones(n // 2), rng.rand(n - n // 2)])[rng.permutation(n)]
        elif kind == "distinct": k = rng.rand(n)
        else: k = np.ones(n)
        arrays.append(k)
    offs = np.zeros(narr + 1, dtype=np.int32); offs[1:] = np.cumsum([len(a) for a in arrays])
    keys = torch.from_numpy(np.concatenate(arrays)).cuda(); offs_t = torch.from_numpy(offs).cuda()
    out = torch.empty(int(offs[-1]), dtype=torch.int32, device="cuda")
    lib.rlap_debug_wave_sort(h, keys.data_ptr(), offs_t.data_ptr(), narr, 0, out.data_ptr())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lib.rlap_debug_wave_sort(h, keys.data_ptr(), offs_t.data_ptr(), narr, 0, out.data_ptr())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"n={n} {kind}: {narr} arrays on 2048 single-wave workgroups: {dt*1e3:.2f} ms total")
