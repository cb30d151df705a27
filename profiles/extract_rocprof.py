"""Summarise rocprofv3 (ROCm 7.2, rocpd sqlite output) runs of bench.py into the files committed here.

  python profiles/extract_rocprof.py gpurun_out/prof_stats/r01_results.db gpurun_out/prof_fetch/r01_results.db \
         gpurun_out/prof_write/r01_results.db r01

The three databases come from three separate runs of the SAME command, as the HBM section of
/opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass):
  rocprofv3 --kernel-trace --stats -d ... -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-end-to-end
  rocprofv3 --pmc FETCH_SIZE      -d ... -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-end-to-end
  rocprofv3 --pmc WRITE_SIZE      -d ... -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-end-to-end
FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction from the guide: FETCH_SIZE under-reports wide coalesced
reads by 2x; the corrected figure doubles it (other access widths are uncalibrated — both figures are kept).
"""
import json, sqlite3, sys


def main():
    stats_db, fetch_db, write_db, tag = sys.argv[1:5]
    out = {"tag": tag, "kernels": {}}
    c = sqlite3.connect(stats_db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = ["| kernel | calls | total ms | avg ms | % |", "|---|---:|---:|---:|---:|"]
    for name, calls, tot, avg, pct in rows:
        out["kernels"][name] = {"calls": calls, "avg_ms": round(avg / 1e3, 4), "total_ms": round(tot / 1e3, 3), "pct": round(pct, 2)}
        lines.append(f"| {name} | {calls} | {tot / 1e3:.3f} | {avg / 1e3:.4f} | {pct:.2f} |")
    # bench.py ends with two passes in which the context's streams run one after the other (lm_set_profiling(1)): the last
    # 2 x n_streams dispatches of a pipeline kernel are launches with nothing beside them — the figure bench.py's roofline uses
    # (kernel_ms / kernel_ms_alone); the average over ALL dispatches mixes set-up, overlapped timed steps and those passes
    alone_lines = ["", "## the serialized passes at the end of the command (last 4 dispatches per kernel: 2 passes x 2 streams, nothing beside them)", "",
                   "| kernel | dispatches | avg ms | min ms | max ms |", "|---|---:|---:|---:|---:|"]
    try:
        per = {}
        for name, st, en in c.execute("select name, start, end from kernels order by start"):
            per.setdefault(name, []).append((en - st) / 1e6)
        for name, v in per.items():
            if not name.startswith("k_") or len(v) < 8:
                continue
            last = v[-4:]
            out["kernels"].setdefault(name, {})["alone_avg_ms"] = round(sum(last) / len(last), 4)
            alone_lines.append(f"| {name} | {len(last)} | {sum(last) / len(last):.4f} | {min(last):.4f} | {max(last):.4f} |")
    except Exception as ex:   # older rocpd schema: the summary table alone
        alone_lines.append(f"(per-dispatch table unavailable: {ex})")
    for db, cn in ((fetch_db, "FETCH_SIZE"), (write_db, "WRITE_SIZE")):
        c = sqlite3.connect(db)
        q = "select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name"
        for name, n, avg in c.execute(q, (cn,)):
            out["kernels"].setdefault(name, {})[cn + "_KiB_per_launch"] = round(avg, 1)
    # the integrate kernel that ran (plain documents take the sweep instantiation by default, lm_pipeline.h)
    dom = next((k for k in ("k_integrate_span_plain_sweep", "k_integrate_span_plain", "k_integrate_span", "k_integrate") if k in out["kernels"]), "k_integrate")
    out["dominant_kernel"] = dom
    k = out["kernels"].get(dom, {})
    if "FETCH_SIZE_KiB_per_launch" in k and "WRITE_SIZE_KiB_per_launch" in k:
        raw = (k["FETCH_SIZE_KiB_per_launch"] + k["WRITE_SIZE_KiB_per_launch"]) * 1024
        cor = (2 * k["FETCH_SIZE_KiB_per_launch"] + k["WRITE_SIZE_KiB_per_launch"]) * 1024
        out["hbm_bytes_per_launch_raw"] = int(raw)
        out["hbm_bytes_per_launch"] = int(cor)
        out["note"] = "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 for the integrate kernel (gfx950 FETCH_SIZE correction); raw = uncorrected"
    json.dump(out, open(f"profiles/{tag}_pmc_integrate.json", "w"), indent=1, sort_keys=True)
    open(f"profiles/{tag}_kernel_stats.md", "w").write(
        f"# rocprofv3 --kernel-trace --stats — bench.py --steps 3 --warmup 1 (configs[1], 10k docs, 1 MI355X)\n\n" + "\n".join(lines + alone_lines) + "\n")
    print(json.dumps(out, indent=1)[:1500])


if __name__ == "__main__":
    main()
