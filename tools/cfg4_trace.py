"""dev: a few cfg-4 RAFT-loop passes for `rocprofv3 --kernel-trace --stats` (where does an iteration's time go?)
python tools/cfg4_trace.py [mixed]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
from tools.bench_configs import raft_inputs
dev = torch.device("cuda")
f1, f2, nets, inps = raft_inputs(8, 136, 240, dev, 400)
torch.manual_seed(4)
loop = S.RAFTStereoLoop([128, 128, 128], mixed_precision="mixed" in sys.argv).eval().to(dev)
with torch.no_grad():
    for _ in range(3):
        loop(f1, f2, [n.clone() for n in nets], inps, iters=16, test_mode=True, use_graph=False)
torch.cuda.synchronize()
