import sys, os, json, torch
sys.path.insert(0, os.getcwd())
from speinet_amd.speinet import SPEINet, default_args
from speinet_amd.synth import state_dict_template, synth_frames, synth_state_dict
dev="cuda:0"
net=SPEINet(args=default_args()); net.load_state_dict(synth_state_dict(state_dict_template(), seed=0)); net=net.to(dev).eval()
net.precision, net.corr_precision, net.use_graph, net.streams = "f16","top2",True,2
x=[synth_frames(1,720,1280,seed=1234+i).to(dev) for i in range(2)]
outs={}
for fa in (True, False, True, False):
    net.knobs={"fuse_apply": fa}
    with torch.no_grad():
        for i in range(4): net(x[i%2], routing=[False])
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(12): o=net(x[i%2], routing=[False])
        e1.record(); torch.cuda.synchronize()
    outs[fa]=o.clone()
    print(f"fuse_apply={fa}: {e0.elapsed_time(e1)/12:.2f} ms/frame")
print("max |diff|", (outs[True]-outs[False]).abs().max().item())
