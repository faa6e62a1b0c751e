#!/bin/bash
# Round-3 additions to tools/refresh_profiles.sh (run on the GPU box from the repo root): bash tools/refresh_r03.sh
# Everything lands under gpurun_out/refresh_r03/profiles: copy into profiles/ afterwards.
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/refresh_r03
P=$O/profiles
mkdir -p $P
cd $ROOT
timeout -k 10 500 python3 bench.py > $O/bench.json 2> $O/bench.err && cp $O/bench.json $P/r03_bench_line.json
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_under_rocprof.log 2>&1)
cp $O/stats/run_kernel_stats.csv $P/r03_bench_kernel_stats.csv 2>/dev/null
cp $O/bench_under_rocprof.log $P/r03_bench_under_rocprof.log 2>/dev/null
bash tools/conv_pmc.sh r03 > $O/conv_pmc.log 2>&1
cp gpurun_out/conv_pmc_r03/pmc_conv.json $P/r03_pmc_conv.json 2>/dev/null
cp gpurun_out/conv_pmc_r03/pmc_conv.json $P/pmc_conv_latest.json 2>/dev/null
bash tools/a7_pmc.sh > $O/a7_pmc.log 2>&1
cp gpurun_out/a7_pmc/pmc_a7.json $P/r03_pmc_a7.json 2>/dev/null
cp gpurun_out/a7_pmc/pmc_a7.json $P/pmc_a7_latest.json 2>/dev/null
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/cfg4 -o run --output-format csv -- python3 $ROOT/tools/cfg4_trace.py > $O/cfg4_trace.log 2>&1)
python3 - $O <<'PY' > $P/r03_cfg4_loop_kernel_trace.txt
import csv, collections, re, sys
o = sys.argv[1]
csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(o + "/cfg4/run_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "allpairs_lds" in r["Kernel_Name"]]
last = rows[idx[-1]:]
agg = collections.defaultdict(lambda: [0.0, 0])
for r in last:
    k = re.sub(r"\(anonymous namespace\)::|at::native::", "", r["Kernel_Name"])[:110]
    agg[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    agg[k][1] += 1
wall = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e6
print("rocprofv3 --kernel-trace of tools/cfg4_trace.py (third pass of the loop = steady state; eager, so the motion encoder's side\n"
      "stream overlaps the coarse GRUs and kernel times of overlapped launches are inflated), summarised per kernel name")
print(f"one pass of RAFTStereoLoop (16 iterations, 8 pairs at 1/4 of 544x960, NHWC, eager): wall {wall:.1f} ms, {len(last)} kernels, "
      f"sum of kernel times {sum(v[0] for v in agg.values()):.1f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:20]:
    print("  %8.3f ms %4d launches  %s" % (v[0], v[1], k))
PY
bash tools/a7_sq_pmc.sh > $O/a7_sq_pmc.log 2>&1
cp gpurun_out/a7_sq/pmc_a7_sq.json $P/r03_pmc_a7_sq.json 2>/dev/null
timeout -k 10 200 python3 tools/place_mapped_probe.py > $P/r03_place_mapped_probe.txt 2>&1
bash tools/a1bwd_pmc.sh corr_bwd_fused=1 full > $O/a1bwd_pmc.log 2>&1
cp gpurun_out/a1bwd_pmc/pmc_a1bwd.json $P/r03_pmc_a1bwd.json 2>/dev/null
(for o in corr_bwd_fused=1 corr_bwd_fused=0 corr_bwd_fused=2 corr_bwd_fused=3 "corr_bwd_fused=1 B=2"; do timeout -k 10 120 python3 tools/a1bwd_ab.py $o 2>&1 | tail -1; done) > $P/r03_a1bwd_ab.txt
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/train -o run --output-format csv -- python3 $ROOT/tools/train_trace.py > $O/train_trace.log 2>&1)
python3 - $O <<'PY' > $P/r03_train_step_kernel_stats.txt
import csv, sys
o = sys.argv[1]
print("rocprofv3 --kernel-trace --stats of tools/train_trace.py: the cfg-2 training step (2 pairs, forward + backward), 13 steps; per kernel")
print(open(o + "/train_trace.log").read().strip().splitlines()[-1])
for r in list(csv.DictReader(open(o + "/train/run_kernel_stats.csv")))[:9]:
    print("  %-90s calls %3s  avg %9.1f us  %5s %%" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
PY
timeout -k 10 900 python3 tools/kernel_bench.py > $O/kernel_bench.log 2>&1 && cp gpurun_out/kernel_bench.json $P/r03_kernel_bench.json
timeout -k 10 200 python3 tools/conv_audit.py > $P/r03_conv_error_audit.txt 2>&1
timeout -k 10 200 python3 tools/conv_ab.py 2>&1 | grep -v amdgpu > $P/r03_conv_ab.txt
timeout -k 10 200 python3 tools/a7_v9_ab.py 2>&1 | grep -v amdgpu > $P/r03_a7_ab.txt
cut -c1-700 $O/bench.json
