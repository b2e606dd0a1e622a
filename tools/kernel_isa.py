#!/usr/bin/env python3
"""Opcode histogram of one kernel of a translation unit, compiled for gfx950 (no GPU needed).
usage: python tools/kernel_isa.py bulletproofsplus_amd/csrc/tu_verify_bls.hip k_fixed_msm [extra hipcc flags...]
Prints, for every kernel whose mangled name contains the pattern: instruction count by opcode (loop bodies are counted
once, as they appear in the text), VGPRs, scratch bytes."""
import collections, os, re, subprocess, sys, tempfile

src, pat = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                    "-o", out, src] + extra, check=True, stderr=subprocess.DEVNULL)
    txt = open(out).read()
for m in re.finditer(r"^(\w*%s\w*):\s*; @" % re.escape(pat), txt, re.M):
    name = m.group(1)
    body = txt[m.end():txt.index("s_endpgm", m.end())]
    c = collections.Counter()
    for line in body.split("\n"):
        line = line.strip()
        if not line or line[0] in ";." or line.endswith(":"):
            continue
        c[line.split()[0]] += 1
    meta = re.search(r"\.name:\s+%s\b.*?\.vgpr_count:\s+(\d+)" % re.escape(name), txt, re.S)
    scr = re.search(r"\.name:\s+%s\b.*?\.private_segment_fixed_size:\s+(\d+)" % re.escape(name), txt, re.S)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print(dem[:100])
    print("  total %d  vgpr %s  scratch %s" % (sum(c.values()), meta.group(1) if meta else "?", scr.group(1) if scr else "?"))
    print("  " + ", ".join("%s %d" % kv for kv in c.most_common(24)))
