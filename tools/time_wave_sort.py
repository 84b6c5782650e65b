import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlap_amd import ops
lib, h = ops._handle(torch.device("cuda", 0))
rng = np.random.RandomState(0)
for n, kind in [(38, "ties"), (100, "ties"), (100, "dupids"), (128, "allequal"), (200, "ties"), (380, "ties"), (380, "distinct"), (380, "allequal"), (500, "ties")]:
    narr = 2048
    arrays = []
    for a in range(narr):
        if kind == "ties": k = np.concatenate([np.ones(n // 2), rng.rand(n - n // 2)])[rng.permutation(n)]
        elif kind == "distinct": k = rng.rand(n)
        elif kind == "dupids": k = rng.randint(0, 4 * n, size=n).astype(float)
        else: k = np.ones(n)
        arrays.append(k)
    offs = np.zeros(narr + 1, dtype=np.int32); offs[1:] = np.cumsum([len(a) for a in arrays])
    keys = torch.from_numpy(np.concatenate(arrays)).cuda(); offs_t = torch.from_numpy(offs).cuda()
    out = torch.empty(int(offs[-1]), dtype=torch.int32, device="cuda")
    res = {}
    for flag in (0, 8):
        lib.rlap_debug_wave_sort(h, keys.data_ptr(), offs_t.data_ptr(), narr, flag, out.data_ptr())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lib.rlap_debug_wave_sort(h, keys.data_ptr(), offs_t.data_ptr(), narr, flag, out.data_ptr())
        torch.cuda.synchronize(); res[flag] = (time.perf_counter() - t0, out.clone())
    same = bool(torch.equal(res[0][1], res[8][1]))
    print(f"n={n} {kind}: {narr} arrays on 2048 single-wave workgroups: wave_std_sort {res[0][0]*1e3:.2f} ms, level-synchronous {res[8][0]*1e3:.2f} ms, same permutation: {same}")
