"""GPU box: ONE entry of bench.py's other_configs in a process of its own (profiles/collect_other.sh runs it under rocprofv3).
   python tests/tools/gpu_one_config.py "configs[3]" """
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["LM_BENCH_ONLY"] = sys.argv[1]
os.environ["LM_BENCH_NO_POOL"] = "1"
import bench
out = bench.other_configs(0, os.cpu_count() or 8)
for name, e in out.items():
  print(json.dumps({"name": name, "docs": e.get("docs"), "docs_per_s": e.get("docs_per_s"), "ms_per_batch": e.get("ms_per_batch"), "stage_ms": e.get("stage_ms"), "error": e.get("error"), "pipeline_runs": e.get("pipeline_runs_in_this_process"), "state_documents": e.get("state_documents"), "gpu_over_cpu": e.get("gpu_over_cpu")}))
