"""tools/pmc_summary.py --traffic-json: the per-launch HBM traffic must come from the FULL launches of a cycle-batched run
(a few shorter ones -- the settle cycle, span heads and tails -- and one cold outlier must not move it), with the launch
form recorded beside it."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = ("void jst::dev::fft_pipe_kernel<4096, true, true, jst::dev::LoadCF32TimesWindow, "
          "jst::dev::StoreAmplitudeRangeSideT<true> >(jst::dev::FftLayout)")


def _write(path, counter, values):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        for i, v in enumerate(values):
            for part in range(8):  # summed over XCDs
                w.writerow([i + 1, KERNEL, counter, v / 8.0])


def test_traffic_takes_the_full_launches(tmp_path):
    full_fetch, full_write = 262500.0, 327680.0
    fetch = [full_fetch / 16, full_fetch * 15 / 16, full_fetch * 1.22] + [full_fetch * (1 + 0.001 * (i % 3)) for i in range(40)]
    write = [full_write / 16, full_write * 15 / 16, full_write * 1.25] + [full_write] * 40
    _write(str(tmp_path / "pmc_fetch" / "x_counter_collection.csv"), "FETCH_SIZE", fetch)
    _write(str(tmp_path / "pmc_write" / "x_counter_collection.csv"), "WRITE_SIZE", write)
    out = str(tmp_path / "traffic.json")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), str(tmp_path), "--traffic-json", out,
                           "--provider", "fast", "--cycles", "16"], stdout=subprocess.DEVNULL)
    rec = json.load(open(out))["fast"]
    assert rec["cycles_per_launch"] == 16 and rec["launches"] == 40
    expect = (2.0 * sum(fetch[3:]) / 40 + full_write) * 1024.0
    assert abs(rec["spectrum_fused_hbm_bytes_per_launch"] - expect) <= 1.0
    assert len(rec["kernel_sources_sha256"]) == 64
