"""rocprofv3 passes of ONE other_configs entry (profiles/collect_other.sh) -> its record in profiles/<tag>_pmc_other.json: per kernel
calls / average duration (kernel trace pass), FETCH_SIZE / WRITE_SIZE KiB per launch (separate --pmc passes).  bench.py prices the
entry's dominant stage with (2 x FETCH + WRITE) KiB x dispatches — the gfx950 FETCH correction of /opt/skills/guides/MI355X_MICROARCH.md."""
import json, os, sqlite3, sys

tag, name, stats_db, fetch_db, write_db, log = sys.argv[1:7]
path = f"profiles/{tag}_pmc_other.json"
allrec = json.load(open(path)) if os.path.exists(path) else {}
line = {}
for l in open(log, errors="replace"):
    if l.startswith('{"name"'):
        cand = json.loads(l)
        if not line or cand.get("name") == name:
            line = cand
# (other_configs._run runs the pipeline 1 + reps + 1 times — parity, timed repetitions, the stage-timing pass: `pipeline_runs` of the tool's line)
rec = {"command": f'python tests/tools/gpu_one_config.py "{name}"', "docs": line.get("docs"), "runs_of_the_pipeline": line.get("pipeline_runs") or 5, "stage_ms": line.get("stage_ms"),
       "docs_per_s": line.get("docs_per_s"), "kernels": {}}
c = sqlite3.connect(stats_db)
for kn, calls, tot, avg, pct in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    if kn.startswith("k_"):
        rec["kernels"][kn] = {"calls": calls, "avg_ms": round(avg / 1e3, 4), "total_ms": round(tot / 1e3, 3), "pct": round(pct, 2)}
for db, cn in ((fetch_db, "FETCH_SIZE"), (write_db, "WRITE_SIZE")):
    c = sqlite3.connect(db)
    for kn, n, avg in c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name", (cn,)):
        if kn in rec["kernels"]:
            rec["kernels"][kn][cn + "_KiB_per_launch"] = round(avg, 1)
if rec["kernels"]:
    rec["dominant_kernel"] = max(rec["kernels"], key=lambda k: rec["kernels"][k]["total_ms"])
allrec[name] = rec
json.dump(allrec, open(path, "w"), indent=1, sort_keys=True)
print(name, rec.get("dominant_kernel"), rec.get("docs_per_s"))
