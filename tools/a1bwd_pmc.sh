#!/bin/bash
# Counters of the single-read A1 backward at cfg-2 size (run on the GPU box from the repo root): bash tools/a1bwd_pmc.sh [opt=val] [full]
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/a1bwd_pmc
mkdir -p $O
rm -rf $O/p*
cd /tmp && export TMPDIR=/tmp
i=0
GROUPS_=("SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS")
if [ "${2:-}" = "full" ]; then GROUPS_+=("FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD"); fi
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp -d $O/p$i -o run --output-format csv -- python3 $ROOT/tools/a1bwd_ab.py $1 > $O/p$i.log 2>&1
  python3 $ROOT/tools/pmc_summary.py $O/p$i corr_volume_bwd > $O/p$i.json
done
cd $ROOT
python3 - $O <<'PY'
import json, sys, glob
o = sys.argv[1]
m = {}
for p in sorted(glob.glob(o + "/p*.json")):
    for k, e in json.load(open(p)).items():
        m.setdefault(k, {}).update(e)
json.dump(m, open(o + "/pmc_a1bwd.json", "w"), indent=1)
print(json.dumps(m, indent=1))
PY
