#!/usr/bin/env python3
"""Headline benchmark: stereo pairs/s on SceneFlow-shaped synthetic pairs (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

N=1 workload = BASELINE.json configs[1]: 960x540, D=192, group-wise (G=8) cost volume +
group-mean + soft-argmin + LocalCostVolume(r=2,S=20) refinement, batch 8 per GPU, C=32
feature channels, fp32.  A "step" is one pass of that chain over one batch of 8 pairs with
the features already resident in HBM; every intermediate the reference API materialises
(the [8,8,192,540,960] group-wise volume, 25.5 GB, and the [8,192,540,960] mean volume) is
materialised here too.

N>1: the pairs shard over ranks with no data-path collective ("weak" scaling: 8 pairs per
GPU); RCCL carries only the barrier / max-over-ranks timing and one final gather of the
disparity shards.  `python bench.py --gpus N` from a bare shell starts its own N ranks
(`torch.distributed.run`, one process per GPU) as a CHILD process before anything in this
process touches the GPU, relays rank 0's JSON line and exits with the child's code; when
WORLD_SIZE is already set (the launcher started us) it runs as one rank.

The JSON line carries `roofline` (the dominant kernel: the fused volume stage, HBM-write
bound; every per-step HIP-event time of it is kept), `cpu_baseline` (the C oracle, OpenMP,
on a bounded sample) and `other_configs` (BASELINE configs[2..4], compact, sharded the same way).
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU, C, H, W, D, G = 8, 32, 540, 960, 192, 8
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the compact cfg-3/4/5 entries (headline line only)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="run the N-rank code path with every rank on cuda:0 over gloo (1-GPU boxes; timings are "
                         "NOT a valid measurement -- the ranks share one GPU; N <= 6)")
    ap.add_argument("--rehearse-on-cpu", action="store_true",
                    help="launcher / sharding / reduction plumbing only, over gloo, no GPU work at all (CPU tests)")
    ap.add_argument("--fp16-features", action="store_true",
                    help="hand the features over in fp16 (SURVEY 8(f) N2: widened on the way into LDS, fp32 arithmetic and "
                         "outputs); NOT the headline configuration -- the reference's features are fp32")
    ap.add_argument("--no-placement", action="store_true",
                    help="let every step's outputs come from the caching allocator's first allocation instead of buffers "
                         "placed with stereoformer_amd.hbm (see DESIGN.md section 5: the launch runs at one of two stable "
                         "speeds depending on the physical memory behind its 28.7 GB of outputs)")
    ap.add_argument("--reread", action="store_true",
                    help="group-mean + soft-argmin by re-reading the 25.5 GB volume (the pre-fusion chain)")
    return ap.parse_args(argv)


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks as a child `torch.distributed.run` (never an exec,
    and before this process has made any GPU call), relay its output, return its exit code."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:          # rank 0 prints the one JSON line; relay everything as it comes
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def synth_pairs(batch, seed, device):
    """L = randn L2-normalised over C, scaled by 0.5*sqrt(C); R = L shifted by a
    smooth disparity field in [0, 0.8 D] + 0.05 noise."""
    import torch

    g = torch.Generator(device=device).manual_seed(seed)
    L = torch.randn(batch, C, H, W, device=device, generator=g)
    L = L / L.norm(dim=1, keepdim=True).clamp_min(1e-6) * (0.5 * C ** 0.5)
    ys = torch.linspace(0, 1, H, device=device).view(1, 1, H, 1)
    xs = torch.linspace(0, 1, W, device=device).view(1, 1, 1, W)
    d_true = 0.8 * D * (0.5 + 0.25 * torch.sin(6.28 * xs) * torch.cos(3.14 * ys) + 0.25 * ys)
    d_int = d_true.round().long().clamp(0, D - 1).expand(batch, C, H, W)
    idx = (torch.arange(W, device=device).view(1, 1, 1, W) + d_int).clamp(max=W - 1)
    # R[w - d] = L[w]  <=>  R[v] = L[v + d(v)] (approximately; exact matching is not needed)
    R = torch.gather(L, 3, idx) + 0.05 * torch.randn(batch, C, H, W, device=device, generator=g)
    return L.contiguous(), R.contiguous()


def usable_cpus():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p))))
    except Exception:
        pass
    return n


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(seed):
    """The C oracle (a port, OpenMP) on ONE pair of the same workload, same chain: at every usable core (the reported
    `value`) and at 8 threads (to sit beside BASELINE.md's 8-thread figure of the real reference, 0.17 pairs/s)."""
    import numpy as np
    import torch

    import oracle

    cores = usable_cpus()
    path = None
    try:  # -march=native build on this host; falls back to the portable library
        path = oracle.build(force=True, native=True, out=os.path.join("/tmp", f"libstereo_oracle_native_{os.getpid()}.so"))
        lib = oracle.load(path)
    except Exception:
        lib = oracle.load()
    Lt, Rt = synth_pairs(1, seed, "cpu")
    L, R = Lt.numpy(), Rt.numpy()
    refined = None

    def chain_seconds(threads, budget_s, runs):
        nonlocal refined
        oracle.set_threads(threads, lib=lib)
        times = []
        t_all = time.perf_counter()
        for _ in range(runs):
            t0 = time.perf_counter()
            volg = oracle.gwc_volume(L, R, D, G, lib=lib)
            mean = oracle.group_mean(volg, lib=lib)
            disp = oracle.soft_argmin(mean, lib=lib)
            refined = oracle.local_lookup(mean, disp[:, None], 2.0, 20, True, lib=lib)
            times.append(time.perf_counter() - t0)
            del volg, mean
            if time.perf_counter() - t_all > budget_s:
                break
        return times

    times = chain_seconds(cores, 15, 3)
    times8 = chain_seconds(min(8, cores), 8, 2) if cores != 8 else times
    oracle.set_threads(cores, lib=lib)
    if path and os.path.exists(path):
        os.remove(path)
    best = min(times)
    # parity of the measured GPU chain against the oracle on this very pair (checker use of the oracle)
    import stereoformer_amd as S
    dev = torch.device("cuda", torch.cuda.current_device())
    with torch.no_grad():
        volg, gd, gmean = S.build_gwc_volume_with_disparity(Lt.to(dev), Rt.to(dev), D, G, return_mean=True)
        gref = S.LocalCostVolume(2, 20)(gmean, gd.unsqueeze(1), True).cpu().numpy()
    err = np.abs(gref - refined)
    return {"value": round(1.0 / best, 4), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"1 pair 960x540 D=192 G=8 (gwc -> group-mean -> soft-argmin -> local lookup), "
                      f"best of {len(times)} runs, C oracle with OpenMP",
            "ms_per_pair": round(best * 1e3, 1), "cpu_model": cpu_model(),
            "at_8_threads": {"value": round(1.0 / min(times8), 4), "ms_per_pair": round(min(times8) * 1e3, 1),
                             "threads": min(8, cores),
                             "beside": "BASELINE.md section 2: the real reference (PyTorch CPU, 8 threads, build container) "
                                       "runs this chain at ~0.17 pairs/s"},
            "gpu_vs_oracle_disparity": {"max_abs": float(err.max()), "mean_abs": float(err.mean()), "bar": 1e-3}}


def spread(ms):
    """Every per-step HIP-event time of one launch, summarised so a single run shows ramp vs steady state."""
    n = len(ms)
    k = min(5, n)
    return {"min": round(min(ms), 4), "median": round(statistics.median(ms), 4), "max": round(max(ms), 4),
            "mean": round(sum(ms) / n, 4), "first5_mean": round(sum(ms[:k]) / k, 4),
            "last5_mean": round(sum(ms[-k:]) / k, 4), "per_step": [round(x, 3) for x in ms]}


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args, argv))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

    if args.rehearse_on_cpu:
        # plumbing only: launcher -> ranks -> sharding -> max-over-ranks -> gather -> rank 0's line.  No GPU, no kernels.
        from stereoformer_amd import shard
        if world > 1:
            dist.init_process_group("gloo")
        sl = shard.pair_slice(world * B_PER_GPU, rank, world)
        local = torch.arange(sl.start, sl.stop, dtype=torch.float32).view(-1, 1)
        t0 = time.perf_counter()
        time.sleep(0.01 * (rank + 1))
        elapsed = time.perf_counter() - t0
        per_rank = shard.all_ranks(elapsed)
        elapsed_max = shard.max_over_ranks(elapsed)
        gathered = shard.gather_pairs(local, world * B_PER_GPU)
        if rank == 0:
            print(json.dumps({"metric": "plumbing rehearsal (no GPU work)", "value": None, "unit": "pairs/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "data": "REHEARSAL on CPU over gloo: launcher and sharding plumbing only; not a measurement",
                              "per_rank_ms": [round(x * 1e3, 3) for x in per_rank],
                              "max_rank_ms": round(elapsed_max * 1e3, 3),
                              "pairs_gathered": int(gathered.shape[0]),
                              "checksum": float(gathered.sum().item())}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 and args.rehearse_on_one_gpu:
        dist.init_process_group("gloo")
    elif world > 1:
        dist.init_process_group("nccl", device_id=device)

    import stereoformer_amd as S
    from stereoformer_amd import shard

    S._native.load()
    lookup = S.LocalCostVolume(radius=2, sample_points=20)
    L, R = synth_pairs(B_PER_GPU, 1234 + rank, device)
    if args.fp16_features:
        L, R = L.half(), R.half()
    feat_bytes = L.element_size()

    # Output buffers of the volume stage: owned by the caller for the whole run and overwritten by every step (what the caching
    # allocator does anyway), allocated through stereoformer_amd.hbm: the launch's speed is a stable property of where in the
    # device's memory its 25.5-GB volume lies, so the volume is the fastest window of a chunk-mapped arena (up to 160 GB beyond its
    # size, probed coarse-to-fine with the launch itself; every chunk outside the window is released again before the timed
    # region -- reported under `placement`).
    placed, placement = None, None
    if not args.no_placement and not args.reread:
        tries = 2 if args.rehearse_on_one_gpu else 8
        try:
            placed, placement = S.hbm.place_volume_stage_outputs(L, R, D, G, max_tries=tries, arena=not args.rehearse_on_one_gpu)
        except Exception as e:   # placement is an optimisation of where the outputs lie: never a reason not to measure
            placed, placement = None, {"mode": "none", "error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.empty_cache()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    ev_read = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    ev_look = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None):
        with torch.no_grad():
            if args.reread:   # the pre-fusion chain: volume kernel, then a reader over the 25.5 GB volume
                if i is not None:
                    ev[i][0].record()
                volg = S.build_gwc_volume(L, R, D, G)
                if i is not None:
                    ev[i][1].record()
                    ev_read[i][0].record()
                disp, mean = S.group_mean_soft_argmin(volg, return_mean=True)
                if i is not None:
                    ev_read[i][1].record()
            else:
                # ONE kernel for the volume stage: group-wise volume (written once, never read back) + its group mean
                # (== the all-channel correlation volume, identity I1) + the soft-argmin disparity
                if i is not None:
                    ev[i][0].record()
                volg, disp, mean = S.build_gwc_volume_with_disparity(L, R, D, G, return_mean=True, out=placed)
                if i is not None:
                    ev[i][1].record()
            if i is not None:
                ev_look[i][0].record()
            out = lookup(mean, disp.unsqueeze(1), True)
            if i is not None:
                ev_look[i][1].record()
            return out

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    sync_all()
    elapsed_local = time.perf_counter() - t0
    per_rank = shard.all_ranks(elapsed_local, device=device)
    elapsed = shard.max_over_ranks(elapsed_local, device=device)

    build_steps = [a.elapsed_time(b) for a, b in ev]
    build_ms = sum(build_steps) / args.steps
    read_ms = sum(a.elapsed_time(b) for a, b in ev_read) / args.steps if args.reread else None
    look_steps = [a.elapsed_time(b) for a, b in ev_look]

    # one RCCL gather of the disparity shards (the only data collective; outside the timed region)
    gathered = shard.gather_pairs(out, world * B_PER_GPU)
    assert gathered.shape[0] == world * B_PER_GPU
    checksum = float(gathered.double().sum().item())
    del gathered

    # secondary, untimed-in-`value` numbers: the fused no-materialisation path
    fused_ms = None
    other = {}
    if rank == 0:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        S.corr_soft_argmin(L, R, D)
        e0.record()
        for _ in range(5):
            S.corr_soft_argmin(L, R, D)
        e1.record()
        torch.cuda.synchronize()
        fused_ms = e0.elapsed_time(e1) / 5
        with torch.no_grad():
            def t5(fn):
                fn()
                e0.record()
                for _ in range(5):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                return round(e0.elapsed_time(e1) / 5, 4)
            if not args.reread:   # the shortcut the fused kernel takes, verified on this very batch: vol.mean(1) == mean
                volg, _, mean_f = S.build_gwc_volume_with_disparity(L, R, D, G, return_mean=True)
                other["identity_I1_max_abs(vol.mean(1) - mean)"] = max(
                    float((volg[b_, :, :, r_].mean(0) - mean_f[b_, :, r_]).abs().max().item())
                    for b_ in (0, B_PER_GPU - 1) for r_ in (0, H // 2, H - 1))
                del volg, mean_f
            if not args.fp16_features:   # N2: the same launch fed fp16 features (half the read side)
                Lh, Rh = L.half(), R.half()
                # ... into the SAME (placed) output buffers as the timed fp32 launch, and the fp32 launch re-timed right beside
                # it: the launch's speed follows the memory behind its outputs, so only same-buffer times compare
                other["volume_stage_fused_fp16_features"] = t5(
                    lambda: S.build_gwc_volume_with_disparity(Lh, Rh, D, G, return_mean=True, out=placed))
                other["volume_stage_fused_fp32_features_same_buffers"] = t5(
                    lambda: S.build_gwc_volume_with_disparity(L, R, D, G, return_mean=True, out=placed))
                other["volume_stage_fused_fp16_features_default_allocation"] = t5(
                    lambda: S.build_gwc_volume_with_disparity(Lh, Rh, D, G, return_mean=True))
                other["fused_corr_soft_argmin_no_volume_fp16_features"] = t5(lambda: S.corr_soft_argmin(Lh, Rh, D))
                del Lh, Rh
            other["gwc_volume_alone"] = t5(lambda: S.build_gwc_volume(L, R, D, G))
            other["mean_volume_soft_argmin_from_features"] = t5(lambda: S.corr_volume_with_disparity(L, R, D))
            if not args.reread:
                volg = S.build_gwc_volume(L, R, D, G)
                other["group_mean_soft_argmin_reread"] = t5(lambda: S.group_mean_soft_argmin(volg, return_mean=True))
                del volg
    del out, placed
    res = None
    if rank == 0:
        n_pix = B_PER_GPU * H * W
        gwc_bytes = 2 * n_pix * C * feat_bytes + n_pix * G * D * 4  # A3 = 26.54 GB
        if not args.reread:
            gwc_bytes += n_pix * D * 4 + n_pix * 4         # + mean volume (8(d): A1's N*D*4) + disparity = 29.74 GB
        traffic, traffic_source = None, None  # HBM bytes per launch: NOT observed in this run -- read from the committed PMC passes
        try:
            pmc_path = os.path.join("profiles", "pmc_traffic_latest.json")
            pmc = json.load(open(os.path.join(ROOT, pmc_path)))
            pref = "corr_volume_mfma_kernel"
            cands = [v for k, v in pmc["kernels"].items() if k.startswith(pref) and v.get("hbm_write_bytes", 0) > 2e10
                     and (("true" in k.split(",")[1]) != bool(args.reread))]
            if cands:
                traffic = cands[0]["hbm_bytes_per_launch"]
                traffic_source = (f"{pmc_path} (builder's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, "
                                  "FETCH_SIZE x2 per the gfx950 guide; NOT collected in this run)")
        except Exception:
            pass
        achieved = gwc_bytes / (build_ms * 1e-3) / 1e9
        pairs = world * B_PER_GPU * args.steps
        res = {
            "metric": "stereo pairs/s, SceneFlow 960x540 D=192 (gwc volume + soft-argmin + local lookup)",
            "value": round(pairs / elapsed, 2),
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "ms_per_pair": round(elapsed / args.steps * 1e3 / B_PER_GPU, 4),
            "per_rank_ms_per_step": [round(x / args.steps * 1e3, 3) for x in per_rank],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if not args.fp16_features else "f32 arithmetic and outputs, fp16 feature operands (N2; not the headline)",
            "data": "synthetic" if not args.rehearse_on_one_gpu else "synthetic (REHEARSAL: all ranks share cuda:0 over gloo; not a measurement)",
            "config": {"workload": "cfg-2: SceneFlow 960x540 D=192 group-wise (G=8,C=32) LocalCostVolume + soft-argmin, "
                                   "batch=8 per GPU, all reference-API intermediates materialised"
                                   + (" (group mean by re-reading the volume)" if args.reread else
                                      " (volume stage in one kernel: group mean + disparity from the feature tiles, identity I1; the volume is written "
                                      "once and never read back)"),
                       "pairs_per_gpu": B_PER_GPU, "total_pairs_per_step": world * B_PER_GPU,
                       "parallelism": f"pairs sharded over {world} GPU(s), no data collective"},
            "roofline": {"bound": "hbm",
                         "kernel": ("gwc volume kernel (A3 group-wise volume build)" if args.reread else
                                    "corr_volume_mfma_kernel<7,ARGMIN,8,3,CPG2=2> (A3 group-wise volume + group mean + A4 "
                                    "soft-argmin in one launch)"), "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes": gwc_bytes, "avg_launch_ms": round(build_ms, 4),
                         "launch_ms": spread(build_steps),
                         "frac_at_min_launch": round(gwc_bytes / (min(build_steps) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "frac_at_max_launch": round(gwc_bytes / (max(build_steps) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "kernels_ms": dict({("gwc_build" if args.reread else "volume_stage_fused"): round(build_ms, 4),
                                "local_lookup": round(sum(look_steps) / len(look_steps), 4),
                                "fused_corr_soft_argmin_no_volume": None if fused_ms is None else round(fused_ms, 4)},
                               **({"group_mean_soft_argmin_reread": round(read_ms, 4)} if args.reread else {}), **other),
            "checksum": checksum,
            "placement": placement if placement is not None else "caching allocator's own allocation (--no-placement)",
        }
    del L, R
    torch.cuda.empty_cache()

    others = None
    if not args.no_other_configs and not args.reread:
        from tools import bench_configs
        others = bench_configs.run_all(device, rank, world, cpu=(world == 1 and not args.no_cpu_baseline),
                                       rehearsal=args.rehearse_on_one_gpu)
    if rank == 0:
        res["other_configs"] = others
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(1234)
            except Exception as e:  # the headline line must still come out
                res["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
