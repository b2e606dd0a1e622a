#!/usr/bin/env python3
"""Prints VGPR / SGPR / scratch / LDS per kernel of a built libbpp_amd.so (reads the gfx950 code objects).
usage: python tools/kernel_resources.py [path/to/libbpp_amd.so]"""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin/"
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bulletproofsplus_amd", "libbpp_amd.so")
with tempfile.TemporaryDirectory() as d:
    fb = os.path.join(d, "fatbin")
    subprocess.run([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fb, so], check=True)
    data = open(fb, "rb").read()
    # the section concatenates one offload bundle per translation unit
    starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
    rows = []
    for i, st in enumerate(starts):
        chunk = data[st:starts[i + 1] if i + 1 < len(starts) else len(data)]
        f = os.path.join(d, "b%d" % i)
        open(f, "wb").write(chunk)
        co = os.path.join(d, "co%d" % i)
        r = subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + f,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
        if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        cur = {}
        for line in notes.splitlines():
            m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip()
            if k in ("name", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size"):
                cur[k] = v
            if k == "wavefront_size":
                if "name" in cur and "vgpr_count" in cur:
                    rows.append(cur)
                cur = {}
    seen = set()
    print("%-72s %5s %5s %8s %6s" % ("kernel", "vgpr", "sgpr", "scratch", "lds"))
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        if name in seen:
            continue
        seen.add(name)
        print("%-72s %5s %5s %8s %6s" % (name[:72], r.get("vgpr_count"), r.get("sgpr_count"),
                                          r.get("private_segment_fixed_size"), r.get("group_segment_fixed_size")))
