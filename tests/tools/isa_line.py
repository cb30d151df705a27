"""Ad-hoc: the gfx950 instructions the compiler attributes to given source lines of lm_k_integrate_span.h inside one kernel
(other instantiations of the span body emptied for a quick compile).   python tests/tools/isa_line.py KERNEL LINE [LINE...]"""
import os, re, shutil, subprocess, sys, tempfile
repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
kernel, want = sys.argv[1], [int(x) for x in sys.argv[2:]]
work = tempfile.mkdtemp(prefix="isa_line_")
fast = os.path.join(work, "loro_amd", "csrc")
shutil.copytree(os.path.join(repo, "loro_amd", "csrc"), fast, ignore=shutil.ignore_patterns("*.so"))
os.symlink(os.path.join(repo, "include"), os.path.join(work, "include"))
fn = os.path.join(fast, "lm_k_integrate_span.h")
src = open(fn).read()
src = re.sub(r"^(LM_KERNEL[^\n]*void (k_integrate_span\w*)\([^{]*\{\n)([^\n]*integrate_span_body<[^\n]*\n)", lambda m: m.group(1) + (m.group(3) if m.group(2) == kernel else "\n"), src, flags=re.M)
open(fn, "w").write(src)
out = os.path.join(work, "k.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "--offload-device-only", "-S", "-gline-tables-only", "-o", out, "lm_hip.cpp"], cwd=fast, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
fileno = None
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"lm_k_integrate_span.h"', l)
    if m: fileno = int(m.group(1))
start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
cur = None
for i in range(start, end):
    l = lines[i]
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        new = (int(m.group(1)), int(m.group(2)))
        if new != cur and new[0] == fileno and new[1] in want: print(f"--- line {new[1]} (asm line {i})")
        cur = new
        continue
    if cur and cur[0] == fileno and cur[1] in want and (re.match(r"\s+([sv]_|global_|ds_|buffer_|flat_|scratch_)", l) or l.startswith(".LBB")):
        print(l.rstrip())
print("asm:", out)
