#!/usr/bin/env python3
"""codeobj_notes.py -- per-kernel resource figures of the gfx950 code objects inside libbspatom.so, from the code-object notes
(llvm-readelf --notes): registers, spills, scratch (private segment), LDS.  Used by tests/test_host_cpu.py to keep scratch out
of the hot-path kernels, and by hand:

    python tools/codeobj_notes.py [--scratch-only] [path/to/lib.so]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
FIELDS = (".private_segment_fixed_size", ".vgpr_spill_count", ".sgpr_spill_count", ".vgpr_count", ".agpr_count", ".sgpr_count",
          ".group_segment_fixed_size", ".max_flat_workgroup_size")


def kernels(lib):
    """{demangled kernel name: {field: int}} over every device code object of `lib` (one bundle per translation unit)"""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib])
        data = open(fat, "rb").read()
        pos = [m.start() for m in re.finditer(re.escape(MAGIC), data)] + [len(data)]
        for i in range(len(pos) - 1):
            b = os.path.join(td, "b%d.bin" % i); co = os.path.join(td, "b%d.co" % i)
            open(b, "wb").write(data[pos[i]:pos[i + 1]])
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + b,
                                   "--targets=" + TARGET, "--output=" + co], stderr=subprocess.DEVNULL)
            notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
            cur = None
            # the metadata is YAML: a kernel's map starts at '- .agpr_count:' (keys sorted) and holds '.name:'
            for blk in re.split(r"\n\s*- (?=\.agpr_count:|\.args:)", notes):
                if ".name:" not in blk or ".kernarg_segment_size" not in blk:
                    continue
                name = re.search(r"\n\s*\.name:\s*(\S+)", "\n" + blk)
                if not name:
                    continue
                cur = {}
                for f in FIELDS:
                    m = re.search(r"(?:^|\n)\s*" + re.escape(f) + r":\s*(\d+)", blk)
                    cur[f.lstrip(".")] = int(m.group(1)) if m else None
                out[name.group(1)] = cur
    names = list(out)
    import shutil
    filt = shutil.which("c++filt") or (os.path.join(LLVM, "llvm-cxxfilt") if os.path.exists(os.path.join(LLVM, "llvm-cxxfilt")) else None)
    dem = subprocess.run([filt] + names, capture_output=True, text=True).stdout.split("\n") if filt else names
    # "void bsp::(anonymous namespace)::tsqr_tree_kernel<2>(bsp::(anonymous namespace)::TsqrArgs)" -> "bsp::tsqr_tree_kernel<2>"
    short = lambda d: d.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    return {(short(dem[i]) if i < len(dem) and dem[i] else n): out[n] for i, n in enumerate(names)}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bspatom_amd", "libbspatom.so")
    ks = kernels(lib)
    for n in sorted(ks):
        v = ks[n]
        if "--scratch-only" in sys.argv and not v["private_segment_fixed_size"]:
            continue
        print("%-110s vgpr %3s agpr %3s scratch %5s B  vgpr spills %3s sgpr spills %3s  LDS %6s" %
              (n[:110], v["vgpr_count"], v["agpr_count"], v["private_segment_fixed_size"], v["vgpr_spill_count"], v["sgpr_spill_count"],
               v["group_segment_fixed_size"]))
