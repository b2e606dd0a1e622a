#!/usr/bin/env python3
"""Copies the summaries of a tools/collect_evidence.sh run into profiles/ under round-tagged names and derives the PMC
figures (mean per dispatch of the two dominant kernels).  usage: python tools/evidence_to_profiles.py gpurun_out/r3ev r03"""
import csv
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles")


def last_json(path):
    return json.loads([l for l in open(path).read().strip().splitlines() if l.startswith("{")][-1])


for leg in ("headline", "latency_combined", "grouped", "pipelined", "c3", "prove_serialized_production", "msm"):
    if not os.path.exists(os.path.join(src, leg, "k_kernel_stats.csv")):
        continue
    shutil.copy(os.path.join(src, leg, "k_kernel_stats.csv"), os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, leg)))
    json.dump(last_json(os.path.join(src, leg + ".json")), open(os.path.join(dst, "%s_bench_under_rocprof_%s.json" % (tag, leg)), "w"))


def pmc(dirname, kernel_substr):
    """mean counter value and duration per dispatch of the kernels whose name contains kernel_substr"""
    vals, dur = [], []
    for r in csv.DictReader(open(os.path.join(src, dirname, "p_counter_collection.csv"))):
        if kernel_substr in r["Kernel_Name"]:
            vals.append(float(r["Counter_Value"]))
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    return (sum(vals) / len(vals), sum(dur) / len(dur), len(vals)) if vals else (None, None, 0)


out = {"note": "rocprofv3 --pmc <counter> --kernel-trace --output-format csv, one counter per pass (tools/collect_evidence.sh part 2), "
               "means per dispatch.  FETCH_SIZE / WRITE_SIZE are in KB.  MI355X_MICROARCH.md: FETCH_SIZE counts 128-byte requests of "
               "wide coalesced streams at 64 bytes (double it for those); other access widths are uncalibrated -- calibrated here on "
               "known byte counts: k_fixed_msm must gather batch x 2050 x 15 entries of 96 B and RAW FETCH_SIZE reads exactly that "
               "(no doubling for the 16-byte-per-lane LDS-DMA gathers of 96-byte entries); k_pip_chunks gathers W x items points "
               "of 96 B the same way.  effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time."}
for name, dirpfx, kern, kb_expected in (("k_fixed_msm<Bls12381, 0>", "pmc_headline_", "k_fixed_msm<bpp::Bls12381, 0>", 8192 * 2050 * 15 * 96 / 1024.0),
                                        ("k_pip_chunks<Bls12381> at N = 2^22", "pmc_msm_", "k_pip_chunks<bpp::Bls12381>", 8 * (2 << 22) * 96 / 1024.0)):
    f, fd, fn = pmc(dirpfx + "FETCH_SIZE", kern)
    w, wd, wn = pmc(dirpfx + "WRITE_SIZE", kern)
    g, gd, gn = pmc(dirpfx + "GRBM_GUI_ACTIVE", kern)
    out[name] = {"dispatches": fn, "FETCH_SIZE_KB_mean": f, "WRITE_SIZE_KB_mean": w, "GRBM_GUI_ACTIVE_mean": g,
                 "kernel_ms_under_pmc": fd, "hbm_bytes_per_launch": (f + w) * 1024 if f is not None and w is not None else None,
                 "gather_bytes_expected": kb_expected * 1024,
                 "effective_clock_GHz": (g / 8 / (gd * 1e-3) / 1e9) if g else None}
json.dump(out, open(os.path.join(dst, "%s_pmc.json" % tag), "w"), indent=1)
# keep the counter rows of the dominant kernels (the raw CSVs hold every dispatch of the run)
for d in sorted(os.listdir(src)):
    if d.startswith("pmc_") and os.path.isdir(os.path.join(src, d)):
        rows = list(csv.reader(open(os.path.join(src, d, "p_counter_collection.csv"))))
        keep = [rows[0]] + [r for r in rows[1:] if "k_fixed_msm" in r[8] or "k_pip_" in r[8] or "k_container" in r[8] or "k_records" in r[8]]
        csv.writer(open(os.path.join(dst, "%s_%s.csv" % (tag, d)), "w", newline="")).writerows(keep)
for f, name in (("ubench.json", "ubench_%s.json"), ("window_sweep.json", "%s_window_sweep.json"), ("sustained.json", "%s_sustained_200_steps.json"),
                ("graph_memset_probe.jsonl", "%s_graph_memset_probe_raw_hip.jsonl"), ("graph_memset_probe_torch.jsonl", "%s_graph_memset_probe_torch.jsonl")):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, name % tag))
print(json.dumps(out, indent=1)[:2500])
