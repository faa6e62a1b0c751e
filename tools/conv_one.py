"""dev: a few launches of the gru08 convz|convr shape (gate epilogues, as the loop runs it) for PMC passes.
python tools/conv_one.py [bf16x6]"""
import os, sys, torch, torch.nn as nn
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
from stereoformer_amd import conv_split as C
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
srcs = [cl(torch.randn(8, 128, 136, 240, device="cuda")) for _ in range(3)]
ctx = [cl(torch.randn(8, 128, 136, 240, device="cuda")) for _ in range(2)]
convs = [nn.Conv2d(384, 128, 3, padding=1).cuda() for _ in range(2)]
prec = "bf16x6" if "bf16x6" in sys.argv else "highest"
with torch.no_grad():
    for _ in range(4):
        C.conv_nhwc(srcs, convs, epilogue=[("sigmoid", ctx[0]), ("sigmoid_mul", ctx[1], srcs[0])], precision=prec)
torch.cuda.synchronize()
