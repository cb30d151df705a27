"""GPU box: the resident-document entries of bench.py alone (configs[1]-incremental, configs[4]-resident).
   python tests/tools/gpu_resident.py [n_docs] [--quick]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10000
print(json.dumps(bench.resident_configs(0, bench.host_cores()[0], n_docs=n, full="--quick" not in sys.argv), indent=1))
