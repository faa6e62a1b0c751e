#!/bin/bash
# HBM traffic of the attention launches of cfg-3 / cfg-5 (run on the GPU box from the repo root): bash tools/a7_pmc.sh
# Separate FETCH_SIZE / WRITE_SIZE passes; writes gpurun_out/a7_pmc/pmc_a7.json (copy to profiles/pmc_a7_latest.json).
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/a7_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in cfg3 cfg5; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c -d $O/${cfg}_$c -o run --output-format csv -- python3 $ROOT/tools/a7_one.py $cfg > $O/${cfg}_$c.log 2>&1
    python3 $ROOT/tools/pmc_summary.py $O/${cfg}_$c cross_attention > $O/${cfg}_$c.json
  done
done
cd $ROOT
python3 - $O <<'PY'
import json, sys
o = sys.argv[1]
out = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on tools/a7_one.py [cfg5]; FETCH_SIZE in KiB x2 (gfx950 "
                 "correction, MI355X_MICROARCH.md), WRITE_SIZE in KiB; per launch", "kernels": {}}
for cfg in ("cfg3", "cfg5"):
    f, w = json.load(open(f"{o}/{cfg}_FETCH_SIZE.json")), json.load(open(f"{o}/{cfg}_WRITE_SIZE.json"))
    for k in f:
        e = {"hbm_read_bytes": f[k].get("hbm_read_bytes"), "hbm_write_bytes": w.get(k, {}).get("hbm_write_bytes")}
        if None not in e.values():
            e["traffic_bytes_per_launch"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
        out["kernels"][cfg + ": " + k] = e
json.dump(out, open(o + "/pmc_a7.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
PY
