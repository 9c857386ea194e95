#!/usr/bin/env python3
"""G18: the REFERENCE's `model/swint.py` (the `--model swint` variant of main_swint.py) run on the CPU, synthetic name-keyed
weights (seed 0): n_sequence 3 on a 40x60 window and on 100x100, n_sequence 1 on 40x60.  Also writes the variant's
state_dict key inventory (tests/golden/state_dict_keys_swint.txt, n_sequence 3).

Run:  python tests/golden/make_golden_swint.py      (needs /root/reference; writes tests/golden/g18_swint_*.npz)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, save, template_args      # noqa: E402


def main():
    from speinet_amd.synth import synth_frames, synth_state_dict
    import_reference()
    import model.swint as mw                    # the reference module itself (uses `util.utils`, pure torch/numpy)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    for n_seq, cases in ((3, (("g18_swint_40x60", 181, 2, 40, 60), ("g18_swint_100x100", 182, 1, 100, 100))),
                         (1, (("g18_swint_n1_40x60", 183, 1, 40, 60),))):
        args = template_args()
        args.n_sequence = n_seq
        net = mw.SPEINet(in_channels=3, n_sequence=n_seq, out_channels=3, n_resblock=3, n_feat=32, device="cpu", args=args)
        net.load_state_dict(synth_state_dict(net.state_dict(), seed=0), strict=True)
        net.eval()
        if n_seq == 3:
            with open(os.path.join(HERE, "state_dict_keys_swint.txt"), "w") as f:
                for k, v in net.state_dict().items():
                    f.write(f"{k}\t{','.join(map(str, v.shape))}\t{str(v.dtype).replace('torch.', '')}\n")
        for name, seed, b, h, w in cases:
            x = synth_frames(b, h, w, seed=seed)[:, :max(n_seq, 1)]
            with torch.no_grad():
                save(name, seed=seed, n_sequence=n_seq, out=net(x))


if __name__ == "__main__":
    main()
