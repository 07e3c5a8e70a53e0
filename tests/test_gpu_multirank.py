"""The multi-rank launch paths, exercised unattended: `python bench.py --gpus 2` re-executes itself
under torch.distributed.run (two ranks; on a one-GPU box they share the device and the control plane runs on gloo),
and tools/bench_c5_multi.py under torchrun with two ranks.  Asserted: the JSON contract of the bench line (n_gpus,
roofline, cpu_baseline, parity) and that config 5's cross-rank PSD mean does not overwrite a rank's own IIR state."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _last_json(text: str) -> dict:
    lines = [ln for ln in text.strip().splitlines() if ln.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_bench_two_ranks_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                          "--no-alt"], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _last_json(out.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 20 and line["warmup"] == 5 and line["scaling"] == "weak"
    assert line["unit"] == "MS/s" and line["value"] > 0 and line["higher_is_better"] is True
    assert line["config"]["workload"].startswith("configs[1]")
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and 0.0 < roof["frac"] < 1.0 and roof["kernel_ms"] > 0 and "step_frac" in roof
    assert "kernel_ms_method" in roof
    cpu = line["cpu_baseline"]   # present for N > 1 too (measured on rank 0 after the process group is gone)
    assert cpu and cpu["value"] > 0 and cpu["cores"] == 1 and cpu["configs0"]["value"] > 0
    par = line["parity"]
    assert par["checked"] and par["bit_exact"] and par["output_rows"] >= 64 * 16 and par["cycles"] >= 16
    # the optional exchange step: timed outside the region by the LIBRARY's communicator when RCCL is the backend (one
    # GPU per rank); on this box the two ranks share a device, the control plane is gloo and the key is null
    assert "collective" in line["config"]
    coll = line["config"]["collective"]
    assert coll is None or (coll["rccl_ranks"] == 2 and coll["uses_rccl"] and coll["sum_of_counts_exact"]
                            and coll["allreduce_us"]["u32_counts_4MiB"] > 0 and coll["in_timed_region"] is False)


def test_config5_two_ranks_psd_reduce_keeps_rank_state():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = _env()
    env["JST_BENCH_BACKEND"] = "gloo"   # two ranks on one device: control plane on gloo, trace through the host
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "tools", "bench_c5_multi.py"), "--cycles", "50", "--interval", "25"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _last_json(out.stdout)
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["backend"] == "gloo"
    assert line["merged_trace_is_mean_of_rank_traces"] is True


def test_bench_single_gpu_line_carries_the_other_configs():
    """configs[2..4] ride in the driver-run command's own JSON line, each with its time per cycle, a
    roofline fraction on SURVEY 8(d)'s bytes and a parity stamp against the oracle on the timed tensors; plus the
    round-1 definition of the headline (provider generic, per-cycle launches) and the rocprofv3-derived fraction key."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "32", "--warmup", "16",
                          "--no-cpu-baseline", "--no-host-fed"], cwd=ROOT, env=_env(), capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _last_json(out.stdout)
    assert line["n_gpus"] == 1 and line["parity"]["bit_exact"]
    cfgs = line["configs"]
    assert len(cfgs) == 3
    for rec in cfgs:
        assert "error" not in rec and "skipped" not in rec, rec
        assert rec["parity"]["checked"] and rec["parity"]["bit_exact"], rec
        assert 0.0 < rec["roofline"]["frac"] < 1.0
    assert cfgs[0]["ms_per_cycle"] > 0 and cfgs[1]["x_realtime"] > 1 and cfgs[2]["us_per_cycle"] > 0
    # config 5 runs on a resident ring with the runtime's defaults: cycle-batched; the launch-per-cycle form rides along
    assert cfgs[2]["cycle_batched"] is True and cfgs[2]["launch_per_cycle"]["bit_exact"]
    assert cfgs[2]["us_per_cycle"] < cfgs[2]["launch_per_cycle"]["us_per_cycle"]
    assert line["value_generic_per_cycle"]["value"] > 0
    assert "frac_rocprof" in line["roofline"] and "rocprofv3_kernel_us" in line["roofline"]
    # round 6: the stamp of config 5 covers a cycle-batched span (every slot of the output ring), the 8-stream form rides along
    assert cfgs[2]["parity"]["batched_span_bit_exact"] is True and cfgs[2]["parity"]["slots_compared"] == 16
    assert cfgs[2]["streams_8"]["transforms_per_cycle"] == 128 and cfgs[2]["streams_8"]["bit_exact_rows_0_3_and_124_127"]
    # config 3's provider fast runs on the matrix cores, within north_star's tolerance of the bit-exact chain
    assert cfgs[0]["fast"]["parity"]["within_1e-5"]
    # the collective executes at N = 1 too: a real one-rank RCCL communicator
    coll = line["config"]["collective"]
    assert coll is None or (coll["rccl_ranks"] == 1 and coll["uses_rccl"] and coll["sum_of_counts_exact"] and coll["allreduce_us"]["u32_counts_4MiB"] > 0)
    # the reference's own scheduler in charge of the same workload, device-resident (where the patched reference was built)
    ref = line["reference_driven"]
    assert ref["available"] is False or (ref["parity"]["spectrogram_bit_exact"] and ref["parity"]["output_within_1e-5"]
                                         and "+indices" in ref["deferred_spans"]["units"] and ref["deferred_spans_sustained"]["us_per_cycle"] > 0)
