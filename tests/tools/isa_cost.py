"""Static-ISA x dynamic-line-count cost model for the integrate kernel (no GPU needed).

The pipeline is instruction-issue bound, so the number of instructions a wave executes tracks the kernel's
duration.  Estimate per instruction of the compiled kernel:
    executions = c(innermost source line) x  Π over its inline chain of  c(call-site line) / Σ c(all call sites of that callee)
with c(line) = wave-level executions of the line for ONE configs[1] document, from the kernel-logic harness built with gcov
(lane-fiber counts / 64).  Instructions and inline chains come from the gfx950 code object (`hipcc -g`, llvm-objdump,
llvm-symbolizer --inlines).  Lane-divergent loops are weighted by their mean trip count (the hardware pays the maximum),
so the absolute figure is a lower bound; it is meant for comparing builds (NEXT.md §5 holds the check against the round-2
GPU A/B measurements).

  python tests/tools/isa_cost.py [repo_dir] [kernel] [--top N] [--define LM_LOC16]      # default: this repo, k_integrate_span"""
import os, re, subprocess, sys, tempfile
from collections import defaultdict

defines = []
if "--define" in sys.argv:
    i = sys.argv.index("--define")
    defines = ["-D" + sys.argv[i + 1]]
    del sys.argv[i:i + 2]
args = [a for a in sys.argv[1:] if not a.startswith("--")]

top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 20
args = [a for a in args if not a.isdigit() or a != str(top)] if "--top" in sys.argv else args
repo = os.path.abspath(args[0]) if args else os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
kernel = args[1] if len(args) > 1 else "k_integrate_span"
work = tempfile.mkdtemp(prefix="isa_cost_")
csrc = os.path.join(repo, "loro_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin/"

# ---- 1. c(line): gcov build of the kernel-logic harness, one configs[1] document
subprocess.check_call(["g++", "-O0", "-g", "--coverage", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-DLM_EMU_TRACE"] + defines + [
                       "-o", os.path.join(work, "libloroemu_cov.so"), os.path.join(repo, "tests", "emu", "lm_emu.cpp")], cwd=work)
subprocess.check_call([sys.executable, "-c", f"""
import sys, os
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, 'tests'))
from loro_amd._cabi import Binding, Context
from loro_amd import workload
docs = [workload.Cfg2Template(50000, 25000, seed=0, commit_every=10, fuse=True).stamp(0)]
if os.environ.get('ISA_DOC') == 'cfg3':   # one LWW Map document of configs[2]'s block shape (4 peers x 2,500 writes on 1,024 keys): ISA_DOC=cfg3 ... k_map_fused
    docs = [workload.cfg3_doc(0, n_peers=4, n_writes=2500, n_keys=1024, combined=True)]
if os.environ.get('ISA_DOC') == 'cfg4':   # one configs[3] document (mixed List / Map / Text, 4 peers, pairwise syncs): ISA_DOC=cfg4 ... k_integrate_span
    import _fuzz
    docs = [_fuzz.blobs_of(_fuzz.random_session(1000, n_peers=4, n_steps=1000, kinds=('text', 'list', 'map'), sync_prob=0.02, styles=True))]
with Context(Binding(os.path.join({work!r}, 'libloroemu_cov.so'), 'lmemu_')) as c:
    assert c.merge_batch(docs)[0][0] == 0
"""], cwd=work, stderr=subprocess.DEVNULL, env=dict(os.environ, **({"LM_PLAIN": "1"} if kernel.endswith("_plain") else ({"LM_PLAIN": "2"} if kernel.endswith("_plain_sweep") else {}))))
gcda = [f for f in os.listdir(work) if f.endswith(".gcda")]
subprocess.check_call(["gcov", "-o", work, os.path.join(work, gcda[0])], cwd=work, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
counts = defaultdict(dict)   # header -> line -> wave-level executions
for f in os.listdir(work):
    if f.startswith("lm_") and f.endswith(".h.gcov"):
        for l in open(os.path.join(work, f), errors="replace"):
            m = re.match(r"\s*(\d+)\*?:\s*(\d+):", l)
            if m:
                counts[f[:-5]][int(m.group(2))] = int(m.group(1)) / 64.0
c = lambda fl: counts.get(fl[0], {}).get(fl[1], 0.0)

# ---- 2. the kernel's instructions and their inline chains
obj, co = os.path.join(work, "k.o"), os.path.join(work, "k.co")
if "--fast" in sys.argv:
    # compile a copy of csrc in which every OTHER instantiation of the span body (and the element-granular kernel) is emptied:
    # the target kernel's code is the same, the compile takes a fraction of the time.  Line numbers are kept.
    import shutil
    fast = os.path.join(work, "loro_amd", "csrc")
    shutil.copytree(csrc, fast, ignore=shutil.ignore_patterns("*.so"))
    os.symlink(os.path.join(repo, "include"), os.path.join(work, "include"))
    for fn, pat in (("lm_k_integrate_span.h", r"^(LM_KERNEL[^\n]*void (k_integrate_span\w*)\([^{]*\{\n)([^\n]*integrate_span_body<[^\n]*\n)"), ):
        src = open(os.path.join(fast, fn)).read()
        src = re.sub(pat, lambda m: m.group(1) + (m.group(3) if m.group(2) == kernel else "\n"), src, flags=re.M)
        open(os.path.join(fast, fn), "w").write(src)
    src = open(os.path.join(fast, "lm_k_integrate.h")).read()
    src = re.sub(r"(LM_KERNEL[^\n]*void k_integrate\([^{]*\{\n)", r"\1  return;\n" if False else r"\1", src)   # (left as is: its body is a template-free kernel)
    open(os.path.join(fast, "lm_k_integrate.h"), "w").write(src)
    csrc_c = fast
else:
    csrc_c = csrc
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "--offload-device-only", "-c", "-gline-tables-only"] + defines + [
                       "-o", obj, os.path.join(csrc_c, "lm_hip.cpp")], cwd=csrc_c, stderr=subprocess.DEVNULL)
subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + obj, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
dis = subprocess.check_output([LLVM + "llvm-objdump", "-d", "--disassemble-symbols=" + kernel, co], text=True)
ins = []   # (address, opcode)
for l in dis.split("\n"):
    m = re.match(r"\s+(\S+).*//\s*([0-9A-Fa-f]+):", l)
    if m and re.match(r"([sv]_|global_|ds_|buffer_|flat_|scratch_)", m.group(1)):
        ins.append((int(m.group(2), 16), m.group(1)))
sym = subprocess.check_output([LLVM + "llvm-symbolizer", "--obj=" + co, "--inlines"], input="\n".join(hex(a) for a, _ in ins) + "\n", text=True)
stacks = []
for blk in sym.strip().split("\n\n"):
    ls = blk.strip().split("\n")
    fr = []
    for i in range(0, len(ls) - 1, 2):
        m = re.match(r"(.*):(\d+):\d+$", ls[i + 1])
        fr.append((re.sub(r"\(.*", "", ls[i]), os.path.basename(m.group(1)) if m else "?", int(m.group(2)) if m else 0))
    stacks.append(fr)
assert len(stacks) == len(ins), (len(stacks), len(ins))
# call sites of every inlined callee: callee function -> {(caller file, line)}
sites = defaultdict(set)
for fr in stacks:
    for i in range(1, len(fr)):
        sites[fr[i - 1][0]].add((fr[i][1], fr[i][2]))
site_sum = {f: sum(c(s) for s in ss) for f, ss in sites.items()}

tot = sc = ve = sp = 0.0
by_line = defaultdict(lambda: [0.0, 0])
by_path = defaultdict(float)
by_func = defaultdict(float)   # inclusive of inlined callees
by_op = defaultdict(float)
by_line_f = defaultdict(lambda: [0.0, 0])   # --ops REGEX: the by-line table restricted to matching opcodes
ops_re = re.compile(os.environ['ISA_OPS']) if os.environ.get('ISA_OPS') else None
for (addr, op), fr in zip(ins, stacks):
    if not fr:
        continue
    w = c((fr[0][1], fr[0][2]))
    for i in range(1, len(fr)):
        s = site_sum.get(fr[i - 1][0], 0.0)
        w = w * (c((fr[i][1], fr[i][2])) / s) if s > 0 else 0.0
    tot += w
    if op.startswith("s_"): sc += w
    else: ve += w
    if op in ("v_readlane_b32", "v_writelane_b32"): sp += w
    by_op[op] += w
    if ops_re and ops_re.match(op):
        by_line_f[(fr[0][1], fr[0][2])][0] += w; by_line_f[(fr[0][1], fr[0][2])][1] += 1
    by_line[(fr[0][1], fr[0][2])][0] += w; by_line[(fr[0][1], fr[0][2])][1] += 1
    for fn in {f[0] for f in fr}:
        by_func[fn] += w
    # the kernel-body line the instruction descends from (which part of the row loop pays)
    by_path[(fr[-2][1], fr[-2][2]) if len(fr) >= 2 else (fr[-1][1], fr[-1][2])] += w
meta = subprocess.check_output([LLVM + "llvm-readelf", "--notes", co], text=True)
spill = re.search(r"\.name:\s+%s\n.*?\.sgpr_spill_count:\s+(\d+)" % kernel, meta, re.S)
print(f"{kernel} @ {repo}: {len(ins)} instructions, SGPR spills {spill.group(1) if spill else '?'}")
print(f"  estimated executions for one configs[1] document: total {tot / 1e6:.3f} M = scalar {sc / 1e6:.3f} M + vector/memory {ve / 1e6:.3f} M "
      f"(spill traffic v_readlane/v_writelane {sp / 1e3:.1f} k)")
print("  by source line:")
for (f, ln), (w, n) in sorted(by_line.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"    {w / 1e3:9.1f} k  {n:4d} instr  {f}:{ln}")
if ops_re:
    print("  by source line, opcodes matching", ops_re.pattern, ":")
    for (f, ln), (w, n) in sorted(by_line_f.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"    {w / 1e3:9.1f} k  {n:4d} instr  {f}:{ln}")
print("  by opcode:")
for op_, w in sorted(by_op.items(), key=lambda kv: -kv[1])[:top]:
    print(f"    {w / 1e3:9.1f} k  {op_}")
print("  by function, inclusive of what is inlined into it:")
for fn, w in sorted(by_func.items(), key=lambda kv: -kv[1])[:top]:
    print(f"    {w / 1e3:9.1f} k  {fn}")
print("  by the kernel-body line they descend from:")
for (f, ln), w in sorted(by_path.items(), key=lambda kv: -kv[1])[:12]:
    print(f"    {w / 1e3:9.1f} k  {f}:{ln}")
