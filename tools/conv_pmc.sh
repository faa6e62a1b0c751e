#!/bin/bash
# PMC passes of the dominant convolution launch (run on the GPU box from the repo root): bash tools/conv_pmc.sh r03
# Each counter group in its own rocprofv3 pass, never combined with a trace domain; the program directly after `--`.
R=${1:-r03}
O=$PWD/gpurun_out/conv_pmc_$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/tools
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VMEM_RD TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp -d $O/p$i -o run --output-format csv -- python3 $T/conv_one.py > $O/p$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $O conv_split > $O/summary.json
cat $O/summary.json
