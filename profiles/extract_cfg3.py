"""rocprofv3 passes of `tests/tools/gpu_cfg.py cfg3 2048 8` (profiles/collect_cfg3.sh) -> profiles/<tag>_pmc_configs2.json:
per kernel calls / average duration, FETCH_SIZE / WRITE_SIZE KiB per launch (separate --pmc passes), and for the dominant kernel the
HBM bytes per launch with the gfx950 FETCH correction of /opt/skills/guides/MI355X_MICROARCH.md ((2 x FETCH + WRITE) KiB)."""
import json, sqlite3, sys

stats_db, fetch_db, write_db, tag = sys.argv[1:5]
# (gpu_cfg.py runs the pipeline 1 + 1 + 3 times: the counters are averaged per dispatch, `calls` counts all five runs)
out = {"tag": tag, "command": "python tests/tools/gpu_cfg.py cfg3 2048 8", "docs": 2048, "runs_of_the_pipeline": 5, "kernels": {}}
c = sqlite3.connect(stats_db)
for name, calls, tot, avg, pct in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    if name.startswith("k_"):
        out["kernels"][name] = {"calls": calls, "avg_ms": round(avg / 1e3, 4), "total_ms": round(tot / 1e3, 3), "pct": round(pct, 2)}
for db, cn in ((fetch_db, "FETCH_SIZE"), (write_db, "WRITE_SIZE")):
    c = sqlite3.connect(db)
    for name, n, avg in c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name", (cn,)):
        if name in out["kernels"]:
            out["kernels"][name][cn + "_KiB_per_launch"] = round(avg, 1)
dom = max(out["kernels"], key=lambda k: out["kernels"][k]["total_ms"])
out["dominant_kernel"] = dom
k = out["kernels"][dom]
if "FETCH_SIZE_KiB_per_launch" in k and "WRITE_SIZE_KiB_per_launch" in k:
    out["hbm_bytes_per_launch_raw"] = int((k["FETCH_SIZE_KiB_per_launch"] + k["WRITE_SIZE_KiB_per_launch"]) * 1024)
    out["hbm_bytes_per_launch"] = int((2 * k["FETCH_SIZE_KiB_per_launch"] + k["WRITE_SIZE_KiB_per_launch"]) * 1024)
    out["launches_per_batch"] = 2
    out["note"] = "per launch = one of the context's two streams (1,024 of the 2,048 documents); (2*FETCH_SIZE + WRITE_SIZE)*1024, the gfx950 FETCH correction"
json.dump(out, open(f"profiles/{tag}_pmc_configs2.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1)[:2000])
