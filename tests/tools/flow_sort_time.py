"""Time of ONE long-column sort of the dataflow kernel by length and form (test hook rlap_debug_wave_sort, desc bits 5/6/7):
256 arrays of one length, one workgroup each -- the launch lasts as long as one sort."""
import os, sys, time
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
import numpy as np, torch
from rlap_amd import ops
lib, h = ops._handle(torch.device("cuda", 0))
rng = np.random.RandomState(0)
for n in (300, 512, 600, 900, 1737, 3000, 5000):
    for kind, mk in (("ties", lambda: np.concatenate([np.ones(n // 2), rng.rand(n - n // 2)])[rng.permutation(n)]), ("distinct", lambda: rng.permutation(n).astype(float))):
        arrays = [mk() for _ in range(256)]
        offs = np.zeros(257, dtype=np.int32); offs[1:] = np.cumsum([n] * 256)
        keys = torch.from_numpy(np.concatenate(arrays)).cuda(); offs_t = torch.from_numpy(offs).cuda()
        out = torch.empty(int(offs[-1]), dtype=torch.int32, device="cuda")
        for desc, name in ((32, "auto"), (160, "index"), (96, "global")):
            lib.rlap_debug_wave_sort(h, keys.data_ptr(), offs_t.data_ptr(), 256, desc | 256, out.data_ptr())
            lib.rlap_debug_wave_sort(h, keys.data_ptr(), offs_t.data_ptr(), 256, desc | 256, out.data_ptr())
            ticks = out.cpu().numpy()[offs[:-1]]
            print(f"n={n} {kind} {name}: {ticks.mean() / 100:.1f} us per sort in the kernel (min {ticks.min() / 100:.1f}, max {ticks.max() / 100:.1f})", flush=True)
