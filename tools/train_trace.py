"""dev: the cfg-2 training step (bench_configs.cfg2_train) a few times, for `rocprofv3 --kernel-trace --stats`.
python tools/train_trace.py [B]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stereoformer_amd as S
dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
C, H, W, D, G = 32, 540, 960, 192, 8
g = torch.Generator(device=dev).manual_seed(200)
L = torch.randn(B, C, H, W, device=dev, generator=g)
R = torch.roll(L, 9, 3) + 0.1 * torch.randn(B, C, H, W, device=dev, generator=g)
lk = S.LocalCostVolume(2, 20)


def step():
    with torch.enable_grad():
        Lt, Rt = L.detach().requires_grad_(True), R.detach().requires_grad_(True)
        _, dsp, mean = S.build_gwc_volume_with_disparity(Lt, Rt, D, G, return_mean=True)
        out = lk(mean, dsp.unsqueeze(1), True)
        return torch.autograd.grad(out.mean(), (Lt, Rt))


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print("train step ms:", (time.perf_counter() - t0) / 10 * 1e3)
