import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from speinet_amd.loss import Loss
from speinet_amd.speinet import default_args
from speinet_amd.swint import SPEINet
from speinet_amd.synth import synth_frames, synth_state_dict
dev = "cuda:0"
args = default_args(); args.n_sequence = 3
net = SPEINet(n_sequence=3, args=args)
net.load_state_dict(synth_state_dict(net.state_dict(), seed=0), strict=True)
net = net.to(dev).train(); net.train_precision = "bf16x3"
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
loss_fn = Loss("1*L1+2*HEM", device=dev)
B = 8
x = synth_frames(B, 200, 200, seed=7)[:, :3].contiguous().to(dev)
gt = synth_frames(B, 200, 200, seed=8)[:, 1].contiguous().to(dev)
def step():
    out = net(x); opt.zero_grad(); loss = loss_fn(out, gt); loss.backward(); opt.step()
step(); step(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    step(); torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.count)
for e in rows[:60]:
    print(f"{e.count:6d} {e.cpu_time_total/1e3:9.2f} ms  {e.key[:110]}")
