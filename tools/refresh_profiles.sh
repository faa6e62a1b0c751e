#!/bin/bash
# Regenerates everything under profiles/ that DESIGN.md quotes (run on the GPU box from the repo root):
#   bash tools/refresh_profiles.sh r02
# rocprofv3 passes: kernel-trace/stats alone; each PMC group in its own pass (never combined with a trace domain).
set -u
R=${1:-r02}
O=gpurun_out/refresh_$R
P=$O/profiles   # only gpurun_out/ travels back from the GPU box: copy $P/* into profiles/ afterwards
mkdir -p $O $P
# the bench line first, on a fresh box (after ~1 min of sustained convolution load the same launch runs ~8-15 % slower)
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err && cp $O/bench.json $P/${R}_bench_line.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_under_rocprof.log 2>&1
cp $O/stats/run_kernel_stats.csv $P/${R}_bench_kernel_stats.csv 2>/dev/null
cp $O/bench_under_rocprof.log $P/${R}_bench_under_rocprof.log 2>/dev/null
for v in f32 f16; do
  X=""; [ $v = f16 ] && X="--fp16-features"
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_$v -o run --output-format csv -- python3 bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 1 $X > $O/pmc_fetch_$v.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_$v -o run --output-format csv -- python3 bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 1 $X > $O/pmc_write_$v.log 2>&1
  python3 tools/pmc_summary.py $O/pmc_fetch_$v corr_volume_mfma group_mean lookup cross_attention > $O/fetch_$v.json
  python3 tools/pmc_summary.py $O/pmc_write_$v corr_volume_mfma group_mean lookup cross_attention > $O/write_$v.json
done
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_mfma -o run --output-format csv -- python3 tools/kernel_bench.py --no-graph --only A7_ > $O/pmc_mfma_a7.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_mfma8 -o run --output-format csv -- python3 tools/kernel_bench.py --no-graph --only A8_ > $O/pmc_mfma_a8.log 2>&1
timeout -k 10 600 python3 tools/kernel_bench.py > $O/kernel_bench.log 2>&1 && cp gpurun_out/kernel_bench.json $P/${R}_kernel_bench.json
python3 - $O $R <<'PY'
import json, sys
o, r = sys.argv[1], sys.argv[2]
out = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on `bench.py --no-cpu-baseline --no-other-configs "
                 "--steps 3 --warmup 1` (fp32 features) and the same with --fp16-features; FETCH_SIZE in KiB x2 (gfx950 correction, "
                 "MI355X_MICROARCH.md), WRITE_SIZE in KiB", "kernels": {}, "kernels_fp16_features": {}}
for v, key in (("f32", "kernels"), ("f16", "kernels_fp16_features")):
    f, w = json.load(open(f"{o}/fetch_{v}.json")), json.load(open(f"{o}/write_{v}.json"))
    for k in f:
        e = {"hbm_read_bytes": f[k].get("hbm_read_bytes"), "hbm_write_bytes": w.get(k, {}).get("hbm_write_bytes"),
             "launches": f[k]["launches"]}
        if e["hbm_read_bytes"] is not None and e["hbm_write_bytes"] is not None:
            e["hbm_bytes_per_launch"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
        out[key][k] = e
json.dump(out, open(o + "/profiles/%s_pmc_traffic.json" % r, "w"), indent=1)
json.dump(out, open(o + "/profiles/pmc_traffic_latest.json", "w"), indent=1)
PY
python3 tools/pmc_summary.py $O/pmc_mfma cross_attention > $O/mfma7.json
python3 tools/pmc_summary.py $O/pmc_mfma8 allpairs > $O/mfma8.json
python3 - $O $R <<'PY'
import json, sys
o, r = sys.argv[1], sys.argv[2]
k = json.load(open(o + "/mfma7.json")); k.update(json.load(open(o + "/mfma8.json")))
json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES "
                     "SQ_BUSY_CYCLES on tools/kernel_bench.py --only A7_ / A8_ (own passes)",
           "note": "mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)",
           "kernels": k}, open(o + "/profiles/%s_pmc_mfma.json" % r, "w"), indent=1)
PY
cat $O/bench.json | cut -c1-600
