"""Summarise a rocprofv3 --pmc ... --output-format json pass of tools/arena_pmc.py: per arena (3 launches each) the launch time and,
per counter, the spread of its PER-INSTANCE values (16 L2 channels x 8 XCDs = 128 instances).
    python tools/arena_pmc_summary.py gpurun_out/.../run_results.json"""
import json
import sys

import numpy as np

r = json.load(open(sys.argv[1]))["rocprofiler-sdk-tool"][0]
names = {c["id"]["handle"]: c["name"] for c in r["counters"]}
ksym = {k["kernel_id"]: k.get("formatted_kernel_name", k.get("kernel_name", "")) for k in r["kernel_symbols"]}
recs = r["callback_records"].get("counter_collection") or r["buffer_records"].get("counter_collection")
disp = []
for rec in recs:
    dd = rec["dispatch_data"]
    if "corr_volume_mfma" not in ksym.get(dd["dispatch_info"]["kernel_id"], ""):
        continue
    per = {}
    for x in rec["records"]:
        per.setdefault(names[x["counter_id"]["handle"]], []).append(x["value"])
    disp.append(((dd["end_timestamp"] - dd["start_timestamp"]) / 1e6, per))
for i, (ms, per) in enumerate(disp):
    if i % 3 == 0:
        continue   # warm-up launch of the arena
    line = f"arena {i // 3} launch {i % 3}: {ms:.3f} ms"
    for n, v in per.items():
        a = np.array(v).reshape(-1)
        line += f" | {n}: n={a.size} sum {a.sum():.4g} max/mean {a.max() / a.mean():.2f} min/mean {a.min() / a.mean():.2f} cv {a.std() / a.mean():.3f}"
        if a.size == 128:
            x = a.reshape(8, 16) if "--xcc-major" in sys.argv else a.reshape(16, 8).T
            line += f" per-XCD sums/mean {np.round(x.sum(1) / x.sum(1).mean(), 2).tolist()}"
    print(line)
