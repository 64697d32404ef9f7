"""dev: phase durations of the fused VGG stem from in-kernel s_memtime stamps (side library built with CVPCE_DBG=128)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cvpce_amd import ops, synthetic, _lib
dev = torch.device('cuda')
enc = synthetic.synthetic_macvgg(seed=1).to(dev)
eng = enc.engine()
x = torch.load('/tmp/real_stem_in.pt').to(dev)
for _ in range(20):
    ops.vgg_stem(x, eng.stem)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (2 * 16 * 6))()
_lib.lib.cvpce_debug_stem_stamps.restype = ctypes.c_int
assert _lib.lib.cvpce_debug_stem_stamps(buf) == 0
t = torch.tensor(list(buf), dtype=torch.int64).view(2, 16, 6)
names = ['phase1 conv1_1', 'barrier1 wait', 'phase2 conv1_2', 'epilogue pool+store', 'barrier2 wait']
for team in range(2):
    d = (t[team, 4:15, 1:] - t[team, 4:15, :-1]).float()
    per_tile = (t[team, 5:15, 0] - t[team, 4:14, 0]).float()
    print(f'team {team}: cycles per tile {per_tile.mean():.0f}  | ' + '  '.join(f'{n} {v:.0f}' for n, v in zip(names, d.mean(0).tolist())))
print('offset team1 - team0 at tile 6 (cycles):', int(t[1, 6, 0] - t[0, 6, 0]))
