"""tools/pmc_summary.py --traffic-json: the per-launch HBM traffic must come from the FULL launches of a cycle-batched run
(a few shorter ones -- the settle cycle, span heads and tails -- and one cold outlier must not move it), with the launch
form recorded beside it."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = ("void jst::dev::fft_quad_kernel<true, jst::dev::RealOperand<jst::dev::LoadCF32TimesWindow>, "
          "jst::dev::StoreAmplitudeRangeSideT<true> >(jst::dev::FftLayout, HIP_vector_type<float, 2u> const*, ...)")


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


def test_one_record_per_launch_form(tmp_path):
    """A second pass with another launch form (cycles per launch) lands beside the first: "<provider>@<cycles>"; the bare
    provider key keeps the form with the most cycles per launch (bench.py looks the run's own form up first)."""
    out = str(tmp_path / "traffic.json")
    for cycles, scale in ((16, 1.0), (32, 2.0), (16, 1.0)):
        root = tmp_path / f"form{cycles}"
        _write(str(root / "pmc_fetch" / "x_counter_collection.csv"), "FETCH_SIZE", [262500.0 * scale] * 12)
        _write(str(root / "pmc_write" / "x_counter_collection.csv"), "WRITE_SIZE", [327680.0 * scale] * 12)
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), str(root), "--traffic-json", out,
                               "--provider", "fast", "--cycles", str(cycles)], stdout=subprocess.DEVNULL)
    doc = json.load(open(out))
    assert set(doc) == {"fast", "fast@16", "fast@32"}
    assert doc["fast"]["cycles_per_launch"] == 32 and doc["fast@16"]["cycles_per_launch"] == 16
    assert doc["fast@32"]["spectrum_fused_hbm_bytes_per_launch"] == 2 * doc["fast@16"]["spectrum_fused_hbm_bytes_per_launch"]


def test_bench_picks_the_ring_period_from_the_run_length():
    """bench.py: as many ring slots as the run has steps, between 16 and 32 (a timed region holds a whole period; the driver's
    --steps 20 is one 20-cycle launch per unit)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "args.slots = min(32, max(16, args.steps))" in src
