"""Register / scratch / LDS use of every kernel of the product library (no GPU needed): hipcc device-only compile, then the
code object's notes.   python tests/tools/kres.py [pattern] [-DNAME ...]"""
import os, re, subprocess, sys, tempfile
repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
csrc = os.path.join(repo, "loro_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin/"
pat = re.compile(next((a for a in sys.argv[1:] if not a.startswith("-D")), "."))
defs = [a for a in sys.argv[1:] if a.startswith("-D")]
work = tempfile.mkdtemp(prefix="kres_")
obj, co = os.path.join(work, "k.o"), os.path.join(work, "k.co")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "--offload-device-only", "-c"] + defs +
                      ["-o", obj, os.path.join(csrc, "lm_hip.cpp")], cwd=csrc, stderr=subprocess.DEVNULL)
subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + obj, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
notes = subprocess.check_output([LLVM + "llvm-readelf", "--notes", co], text=True)
cur = {}
rows = []
for l in notes.split("\n"):
    m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)", l)
    if not m:
        continue
    k, v = m.group(1), m.group(2).strip()
    if k == "agpr_count" and cur.get("name"):
        pass
    if k in ("name", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count", "max_flat_workgroup_size"):
        if k == "name" and v.startswith("k_") and "name" in cur and "vgpr_count" in cur:
            rows.append(cur); cur = {}
        if k == "name" and not v.startswith("k_"):
            continue
        cur[k] = v
    if k == "wavefront_size" and cur.get("name") and "vgpr_count" in cur:
        rows.append(cur); cur = {}
if cur.get("name") and "vgpr_count" in cur:
    rows.append(cur)
print(f"{'kernel':36s} vgpr sgpr scratch  lds  vspill sspill")
for r in sorted(rows, key=lambda r: r["name"]):
    if pat.search(r["name"]):
        print(f"{r['name']:36s} {r.get('vgpr_count','?'):>4s} {r.get('sgpr_count','?'):>4s} {r.get('private_segment_fixed_size','?'):>7s} {r.get('group_segment_fixed_size','?'):>5s} {r.get('vgpr_spill_count','?'):>6s} {r.get('sgpr_spill_count','?'):>6s}")
