import sys, torch
sys.path.insert(0, '.')
from neuroclear_amd import ops
v = torch.arange(2*3*4, dtype=torch.float32).reshape(1,1,2,3,4)
for axis, idx in ((0,1),(1,2),(2,3)):
    m = ops.volume_slice(v.cuda(), axis, idx).cpu()
    print(axis, idx, m.flatten().tolist(), v.select(axis+2, idx).flatten().tolist())
    m2 = ops.volume_mip(v.cuda(), axis, 0, [2,3,4][axis]).cpu()
    print('  mip', m2.flatten().tolist(), v.max(axis+2)[0].flatten().tolist())
