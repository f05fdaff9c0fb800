#!/usr/bin/env python3
"""Static instruction mix of the gfx950 kernels in libblight_mi355x.so (llvm-objdump of the embedded code object): per kernel the
number of VALU / SALU / vector-memory / scalar-memory / LDS / branch / s_waitcnt instructions in the binary.  Static counts: loops
count once (the dynamic figure is SQ_INSTS_VALU of the PMC pass); they are what `tools/pmc_summarize.py`'s per-wave numbers are
compared with.  usage: tools/isa_counts.py [regex] > profiles/rNN_isa_counts.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
flt = re.compile(sys.argv[1] if len(sys.argv) > 1 else r"k_query|k_bin_|k_gather|k_read_|k_superkmer|k_mphf_group|k_mphf_level|k_fill_pos|k_expand")
lib = os.path.join(ROOT, "blight_amd", "libblight_mi355x.so")
with tempfile.TemporaryDirectory() as td:
    # the fat binary lives in .hip_fatbin: unbundle the gfx950 code object
    fat = os.path.join(td, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    # one bundle per translation unit, back to back in the section
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)] + [len(blob)]
    dis = ""
    for i in range(len(starts) - 1):
        part, co = os.path.join(td, "part%d.bin" % i), os.path.join(td, "gfx950_%d.co" % i)
        open(part, "wb").write(blob[starts[i]:starts[i + 1]])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        dis += subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], stdout=subprocess.PIPE, text=True).stdout
rows, cur = [], None
for line in dis.splitlines():
    m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
        name = name.replace("(anonymous namespace)::", "").replace("blmx::", "").replace("void ", "")
        cur = {"name": re.sub(r"\(.*", "", name), "valu": 0, "salu": 0, "vmem": 0, "smem": 0, "lds": 0, "branch": 0, "wait": 0, "all": 0}
        rows.append(cur)
        continue
    m = re.match(r"^\s+([a-z_0-9]+)", line)
    if not m or cur is None:
        continue
    op = m.group(1)
    cur["all"] += 1
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        cur["vmem"] += 1
    elif op.startswith("ds_"):
        cur["lds"] += 1
    elif op.startswith("v_"):
        cur["valu"] += 1
    elif op.startswith(("s_load", "s_buffer_load")):
        cur["smem"] += 1
    elif op.startswith(("s_cbranch", "s_branch")):
        cur["branch"] += 1
    elif op.startswith("s_waitcnt"):
        cur["wait"] += 1
    elif op.startswith("s_"):
        cur["salu"] += 1
print("%-64s %6s %6s %5s %5s %5s %6s %5s %6s" % ("kernel (static counts)", "valu", "salu", "vmem", "smem", "lds", "branch", "wait", "all"))
for r in rows:
    if "rocprim" in r["name"] or not flt.search(r["name"]) or r["all"] < 8:
        continue
    print("%-64s %6d %6d %5d %5d %5d %6d %5d %6d" % (r["name"][:64], r["valu"], r["salu"], r["vmem"], r["smem"], r["lds"], r["branch"], r["wait"], r["all"]))
