"""CPU: the Filter block's plan as the library computes it (jst_filter_plan -> CalculateFilterPlan in
csrc/modules/filter_modules.cc) against an independent restatement of the reference's CalculateCandidatePlan
(src/domains/dsp/filter/block_impl.cc:40-168) kept here as test infrastructure, over a parameter sweep, plus the
reference's bypass and error cases."""
import math

import numpy as np
import pytest


def reference_plan(sample_rate, bandwidth, center, taps, heads, signal_size):
    """block_impl.cc:40-168, written out with Python floats (F64) and exact integers."""
    sr = float(np.float32(sample_rate))  # the block config holds F32 (filter/block.hh)
    bw = float(np.float32(bandwidth))
    plan = {"padSize": taps - 1, "convolutionSize": taps + signal_size - 1, "resample": False,
            "resamplerOffsets": [], "resamplerSize": 0, "resampledSampleRate": 0.0}
    conv = plan["convolutionSize"]
    ratio = sr / bw if bw != 0.0 else math.inf
    if not math.isfinite(ratio) or ratio <= 0 or ratio >= 2.0 ** 64 or ratio != math.floor(ratio):
        return plan
    r = int(ratio)
    if plan["padSize"] % r != 0 or conv % r != 0:
        return plan
    offsets = [0] * heads
    per_bin = sr / float(conv)
    for head in range(heads):
        ct = float(np.float32(center[head])) if head < len(center) else 0.0
        if ct == 0.0:
            continue
        center_bin = ct / per_bin
        rounded = math.floor(abs(center_bin) + 0.5) * math.copysign(1.0, center_bin)  # std::round: half away from zero
        start = -rounded
        if start < 0.0:
            rem = int(-start) % conv
            offsets[head] = 0 if rem == 0 else conv - rem
        else:
            offsets[head] = int(math.fmod(start, float(conv)))
    plan.update(resamplerOffsets=offsets, resamplerSize=conv // r, padSize=plan["padSize"] // r,
                resampledSampleRate=float(np.float32(sr / float(r))), resample=True)
    return plan


def test_plan_integers_of_the_baseline_configs(js):
    p = js.filter_plan(20e6, 2e6, [0.0], 251, 1, 159750)
    assert (p["convolutionSize"], p["resample"], p["resamplerSize"], p["padSize"]) == (160000, True, 16000, 25)
    p = js.filter_plan(20e6, 2e6, [0.0, 3.0e6, -5.0e6], 101, 3, 900)
    assert p["resamplerOffsets"] == [0, 850, 250] and p["resamplerSize"] == 100
    assert not js.filter_plan(2e6, 0.7e6, [0.0], 65, 1, 960)["resample"]     # non-integer ratio
    assert not js.filter_plan(20e6, 2e6, [0.0], 101, 1, 905)["resample"]     # conv % 10 != 0
    assert not js.filter_plan(20e6, 2e6, [0.0], 100, 1, 901)["resample"]     # (taps - 1) % 10 != 0


def test_plan_sweep_against_the_restatement(js):
    rng = np.random.default_rng(7)
    cases = 0
    for _ in range(4000):
        ratio = int(rng.choice([1, 2, 4, 5, 8, 10, 16, 25, 100]))
        bw = float(rng.choice([1e5, 2e5, 2.5e5, 1e6, 2e6]))
        sr = bw * ratio if rng.random() < 0.85 else bw * (ratio + 0.37)
        taps = int(rng.integers(1, 40)) * ratio + 1 if rng.random() < 0.8 else int(rng.integers(2, 300))
        signal = int(rng.integers(1, 500)) * ratio if rng.random() < 0.8 else int(rng.integers(1, 5000))
        heads = int(rng.integers(1, 5))
        center = [float(rng.choice([0.0, 1.0, -1.0, 0.5, -0.5, 0.25]) * rng.integers(0, 40) * bw / 4)
                  for _ in range(int(rng.integers(0, heads + 2)))]
        got = js.filter_plan(sr, bw, center, taps, heads, signal)
        ref = reference_plan(sr, bw, center, taps, heads, signal)
        assert got == ref, (sr, bw, center, taps, heads, signal, got, ref)
        cases += got["resample"]
    assert cases > 1000  # the sweep does reach the resampling branch


def test_plan_errors_like_the_block(js):
    with pytest.raises(js.JetstreamError, match="exceeds the supported range"):
        js.filter_plan(2e6, 1e6, [0.0], 2 ** 63, 1, 2 ** 63 + 5)
    # zero bandwidth / negative ratio: the block bypasses resampling, it does not fail
    assert not js.filter_plan(2e6, 0.0, [0.0], 11, 1, 100)["resample"]
    assert not js.filter_plan(2e6, -1e6, [0.0], 11, 1, 100)["resample"]
