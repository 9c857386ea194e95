import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.nn.functional as F
from speinet_amd import ops, pack
from speinet_amd.ops import FMap
def rnd(seed, *shape, scale=1.0):
    return torch.from_numpy((np.random.RandomState(seed).randn(*shape) * scale).astype(np.float32))
ops.set_precision("bf16x3")
for (cin,cout,k,stride,h,w) in [(32,64,5,2,40,60),(512,256,1,1,33,7),(64,64,5,1,13,17)]:
    x = rnd(1,1,cin,h,w); wt = rnd(2,cout,cin,k,k,scale=1.0/np.sqrt(cin*k*k)); b = rnd(3,cout,scale=0.1)
    ref = F.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=k//2).float()
    out = ops.igemm(FMap.from_nchw(x.cuda()), pack.conv_w(wt).cuda(), b.cuda(), cout, ksize=k, stride=stride).nchw().cpu()
    err = (out-ref).abs()
    print((cin,cout,k,stride,h,w), 'max err', err.max().item(), 'ref max', ref.abs().max().item())
    bad = (err > 1e-3)
    print(' bad frac', bad.float().mean().item(), 'bad channels', bad.any(dim=(0,2,3)).nonzero().flatten().tolist()[:40])
    print(' bad rows', bad.any(dim=(0,1,3)).nonzero().flatten().tolist()[:40], 'bad cols', bad.any(dim=(0,1,2)).nonzero().flatten().tolist()[:40])
