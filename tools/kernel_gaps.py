#!/usr/bin/env python3
"""Gaps between the kernels of the headline step from a `rocprofv3 --kernel-trace --output-format csv` directory: the fused
spectrum kernel -> the span Spectrogram inside a region, and the Spectrogram -> the next region's fused kernel (the host's
turn-around: completion, Python, the next graph launch, dispatch).  Usage: kernel_gaps.py <trace dir>.
Round 5 (bench.py --steps 20 --warmup 5 under the profiler): fused 200.2 us, span 31.8 us, fused -> span 0.0 us (back to
back), span -> next fused 23.4-26.0 us."""
import csv
import glob
import statistics as st
import sys


def main(root: str) -> None:
    f = glob.glob(root + "/*/*kernel_trace.csv")[0]
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
    gaps, spans, fused, between = [], [], [], []
    for a, b in zip(rows, rows[1:]):
        if "fft_quad" in a[2] and "spectrogram_index_span" in b[2]:
            gaps.append(b[0] - a[1])
            fused.append(a[1] - a[0])
            spans.append(b[1] - b[0])
        if "spectrogram_index_span" in a[2] and "fft_quad" in b[2]:
            between.append(b[0] - a[1])
    print("pairs", len(gaps), "| fused us", st.median(fused) / 1e3, "| span us", st.median(spans) / 1e3, "| fused -> span us",
          st.median(gaps) / 1e3, "| span -> next fused us: median", st.median(between) / 1e3, "min", min(between) / 1e3)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
