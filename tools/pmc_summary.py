#!/usr/bin/env python3
"""Per-kernel mean of every counter in rocprofv3 counter_collection CSVs (one or more --pmc passes).
Usage: python tools/pmc_summary.py gpurun_out/prof4 [kernel-substring]
       python tools/pmc_summary.py gpurun_out/prof4 --traffic-json profiles/pmc_traffic.json [--provider fast]
The second form writes the HBM traffic per launch of the fused spectrum kernel (FETCH_SIZE doubled per
MI355X_MICROARCH.md for 16-byte-per-lane streaming reads, WRITE_SIZE as is; both in KiB) together with its provenance:
the kernel symbol, the counter means, and the sha256 of the kernel sources the profiled library was built from
(tools/kernel_hash.py) -- bench.py quotes the figure only while that hash matches the tree it runs from."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
traffic_json = None
if len(sys.argv) > 3 and sys.argv[2] == "--traffic-json":
    traffic_json = sys.argv[3]
want = sys.argv[2] if len(sys.argv) > 2 and not traffic_json else ""
acc = defaultdict(lambda: defaultdict(list))
for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    per_dispatch = defaultdict(float)
    names = {}
    for row in csv.DictReader(open(path)):
        key = (row["Dispatch_Id"], row["Counter_Name"])
        per_dispatch[key] += float(row["Counter_Value"])  # summed over XCDs / dimensions
        names[row["Dispatch_Id"]] = row["Kernel_Name"]
    for (disp, counter), v in per_dispatch.items():
        acc[names[disp]][counter].append(v)
for kernel, counters in acc.items():
    if want not in kernel:
        continue
    print(kernel[:120])
    for counter, vals in sorted(counters.items()):
        vals = vals[len(vals) // 4:]  # drop warm-up dispatches
        print(f"    {counter:24s} mean {sum(vals)/len(vals):16.1f}   (n={len(vals)})")

if traffic_json:
    import json
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from kernel_hash import kernel_sources_sha256
    provider = sys.argv[5] if len(sys.argv) > 5 and sys.argv[4] == "--provider" else "generic"
    # --cycles N: the profiled run was cycle-batched, its full launches carry N compute cycles each (bench.py quotes the
    # figure only for a run with the same launch form)
    cycles = int(sys.argv[7]) if len(sys.argv) > 7 and sys.argv[6] == "--cycles" else 1
    # StoreAmplitudeRangeT<..> or, with the Spectrogram's row indices as a side output, StoreAmplitudeRangeSideT<..>
    tag = "T<true>" if provider == "fast" else "T<false>"
    # the fused 4096-point side kernel of any form (round 5: fft_quad_kernel; pipelined; one wavefront per transform)
    forms = ("fft_quad_kernel<", "fft_pipe_kernel<4096", "fft_wave4096_kernel")
    pick = [k for k in acc if any(f in k for f in forms) and "LoadCF32TimesWindow" in k
            and "StoreAmplitudeRange" in k and tag in k]
    pick.sort(key=lambda k: [f in k for f in forms].index(True))
    kname_tag = next((f for f in forms if pick and f in pick[0]), forms[0])
    if not pick or "FETCH_SIZE" not in acc[pick[0]] or "WRITE_SIZE" not in acc[pick[0]]:
        sys.exit("no FETCH_SIZE / WRITE_SIZE pass for the fused spectrum kernel (" + provider + ") under " + root)
    k = pick[0]
    # full launches only: a cycle-batched run also holds a few shorter launches (the settle cycle, span heads and tails)
    # (within 5 % of the median of the launches that are at least half the largest: one cold outlier must not become "the" launch)
    def full(c):
        big = sorted(x for x in acc[k][c] if x >= 0.5 * max(acc[k][c]))
        med = big[len(big) // 2]
        return [x for x in big if abs(x - med) <= 0.05 * med]
    mean = lambda c: sum(full(c)) / len(full(c))
    fetch_kib, write_kib = mean("FETCH_SIZE"), mean("WRITE_SIZE")
    rec = {"spectrum_fused_hbm_bytes_per_launch": int(round((2.0 * fetch_kib + write_kib) * 1024.0)),
           "source": "pmc-file", "written_by": "tools/pmc_summary.py --traffic-json", "passes_dir": root,
           "kernel": k[:200], "fetch_size_kib_mean": fetch_kib, "write_size_kib_mean": write_kib,
           "launches": len(full("FETCH_SIZE")), "cycles_per_launch": cycles,
           "correction": "FETCH_SIZE x 2 (gfx950 tallies 128-B requests of a streaming read at 64 B, "
                         "MI355X_MICROARCH.md section HBM; the doubled value reproduces the known 32 MiB input + tables to "
                         "0.1 %), WRITE_SIZE as reported",
           "kernel_sources_sha256": kernel_sources_sha256()}
    # the rocprofv3 --kernel-trace --stats average of the same kernel, when that pass sits beside the PMC passes
    # (<root>/trace/**/kernel_stats.csv): bench.py derives roofline.frac_rocprof from it under the same hash guard
    for path in sorted(glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)):
        for row in csv.DictReader(open(path)):
            if row["Name"][:150] == k[:150] or (kname_tag in row["Name"] and tag in row["Name"] and "LoadCF32TimesWindow" in row["Name"]):
                rec["rocprofv3_kernel_us_mean"] = float(row["AverageNs"]) / 1e3
                rec["rocprofv3_kernel_calls"] = int(row["Calls"])
    doc = {}
    if os.path.exists(traffic_json):
        try:
            doc = json.load(open(traffic_json))
        except ValueError:
            doc = {}
        if "spectrum_fused_hbm_bytes_per_launch" in doc:  # the one-provider layout of earlier rounds
            doc = {}
    # one record per (provider, launch form); the bare provider key keeps the form with the most cycles per launch
    doc[f"{provider}@{cycles}"] = rec
    if provider not in doc or int(doc[provider].get("cycles_per_launch", 1)) <= cycles:
        doc[provider] = rec
    with open(traffic_json, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", traffic_json, provider, rec["spectrum_fused_hbm_bytes_per_launch"], "bytes per launch")
