#!/bin/bash
# Issue-side counters of the attention launches of cfg-3 / cfg-5 (run on the GPU box from the repo root): bash tools/a7_sq_pmc.sh
# Two counter groups, each its own rocprofv3 pass (never combined with a trace domain); writes gpurun_out/a7_sq/pmc_a7_sq.json.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/a7_sq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G1="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
G2="SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
for cfg in cfg3 cfg5; do
  i=0
  for grp in "$G1" "$G2"; do
    i=$((i + 1))
    timeout -k 10 200 rocprofv3 --pmc $grp -d $O/${cfg}_p$i -o run --output-format csv -- python3 $ROOT/tools/a7_one.py $cfg > $O/${cfg}_p$i.log 2>&1
  done
done
cd $ROOT
python3 - $O <<'PY'
import csv, json, sys, collections
o = sys.argv[1]
csv.field_size_limit(1 << 30)
out = {"source": "rocprofv3 --pmc, two passes per shape (tools/a7_sq_pmc.sh) on tools/a7_one.py: cfg-3 [8,32,540,960] D=192 out + disparity; "
                 "cfg-5 [4,256,96,312] D=64 disparity only; per launch (sums over all SEs / XCDs)", "kernels": {}}
for cfg in ("cfg3", "cfg5"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for i in (1, 2):
        try:
            rows = list(csv.DictReader(open(f"{o}/{cfg}_p{i}/run_counter_collection.csv")))
        except FileNotFoundError:
            continue
        for r in rows:
            k = r["Kernel_Name"]
            if "cross_attention" not in k:
                continue
            k = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[(k, i)].add(r["Dispatch_Id"])
    for k, c in agg.items():
        n = max(len(launches[(k, 1)]), len(launches[(k, 2)]), 1)
        e = {name: v / n for name, v in sorted(c.items())}
        e["launches"] = n
        if e.get("GRBM_GUI_ACTIVE"):
            # MFMA-busy cycles are summed over the 4 SIMDs of 256 CUs; GRBM_GUI_ACTIVE over the 8 XCDs
            e["mfma_busy_fraction"] = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (e["GRBM_GUI_ACTIVE"] / 8 * 1024)
        if e.get("SQ_INSTS_MFMA"):
            e["valu_per_mfma"] = e.get("SQ_INSTS_VALU", 0.0) / e["SQ_INSTS_MFMA"]
            e["salu_per_mfma"] = e.get("SQ_INSTS_SALU", 0.0) / e["SQ_INSTS_MFMA"]
        out["kernels"][cfg + ": " + k] = e
json.dump(out, open(o + "/pmc_a7_sq.json", "w"), indent=1)
for k, e in out["kernels"].items():
    print(k, {n: round(v, 3) if isinstance(v, float) else v for n, v in e.items() if n in
              ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "mfma_busy_fraction", "valu_per_mfma", "salu_per_mfma", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "GRBM_GUI_ACTIVE")})
PY
