"""dev: a few passes of the cfg-5 chain (bench_configs.cfg5) for `rocprofv3 --kernel-trace --stats`.  python tools/cfg5_trace.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tools import bench_configs as BC
dev = torch.device("cuda")
print(BC.cfg5(dev, 0, 1, 3, 2, False)["value"])
