#!/bin/bash
# PMC passes of the dominant convolution launch of the RAFT loop (run on the GPU box from the repo root): bash tools/conv_pmc.sh r03
# Each counter group in its own rocprofv3 pass, never combined with a trace domain; the program directly after `--`.
# Writes gpurun_out/conv_pmc_<R>/{pmc_conv.json}: copy it to profiles/<R>_pmc_conv.json and profiles/pmc_conv_latest.json.
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/conv_pmc_$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T=$ROOT/tools
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VMEM_RD TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp -d $O/p$i -o run --output-format csv -- python3 $T/conv_one.py > $O/p$i.log 2>&1
  python3 $T/pmc_summary.py $O/p$i conv_split > $O/p$i.json
done
cd $ROOT
python3 - $O <<'PY'
import json, sys
o = sys.argv[1]
k = {}
for i in range(1, 6):
    for name, e in json.load(open(f"{o}/p{i}.json")).items():
        k.setdefault(name, {}).update({c: v for c, v in e.items() if c not in ("launches",)})
out = {"source": "rocprofv3 --pmc, five passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA "
                 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU | SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT "
                 "SQ_LDS_IDX_ACTIVE ... | SQ_INSTS_VMEM_RD TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum) on "
                 "tools/conv_one.py: convz | convr of the 1/4-scale ConvGRU with its gate epilogues, [8,384,136,240] -> 256 "
                 "channels, per launch; FETCH_SIZE in KiB x2 (gfx950 correction, MI355X_MICROARCH.md), WRITE_SIZE in KiB",
       "kernels": k}
for name, e in k.items():
    if "hbm_read_bytes" in e and "hbm_write_bytes" in e:
        out["traffic_bytes_per_launch"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
        out["kernel"] = name
json.dump(out, open(o + "/pmc_conv.json", "w"), indent=1)
print(json.dumps({n: {c: e.get(c) for c in ("hbm_read_bytes", "hbm_write_bytes", "mfma_busy_fraction", "clock_GHz_under_pmc", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "avg_ns_under_pmc")} for n, e in k.items()}, indent=1))
PY
