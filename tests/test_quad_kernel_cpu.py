"""CPU checks of the round-5 4096-point kernel (cyberether_amd/csrc/kernels/fft_quad.hh) that need no GPU:
  * the in-place slot addressing and the LDS-DMA piece map, replayed by tools/quad_index_model.py against numpy.fft;
  * the bank behaviour the header claims for its strides;
  * in the kernel's gfx950 ISA, the `s_waitcnt vmcnt(n)` that waits for the LDS-DMA pieces counts exactly the stores issued
    behind the last piece, and nothing else sits between them (tools/check_quad_isa.py) -- a miscount would be a data race
    that a passing GPU test does not rule out."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_in_place_chain_is_a_dft():
    import quad_index_model as m
    rng = np.random.default_rng(5)
    x = rng.standard_normal(m.N) + 1j * rng.standard_normal(m.N)
    assert np.abs(m.transform_in_place(x) - np.fft.fft(x)).max() < 1e-9
    # every input element is brought in by exactly one lane of one piece, as the first or second of an aligned pair
    seen = np.zeros(m.N, int)
    for q in range(m.PIECES):
        for lane in range(64):
            n = m.piece_source(q, lane)
            if n is not None:
                assert n % 2 == 0 and m.phys(n) == 128 * q + 2 * lane and m.phys(n + 1) == m.phys(n) + 1
                seen[n] += 1
                seen[n + 1] += 1
    assert (seen == 1).all()
    assert max(m.phys(n) for n in range(m.N)) < m.PIECES * 128


def test_bank_behaviour_of_the_strides():
    import quad_index_model as m
    ways = m.bank_conflicts()
    assert ways["pass0"] == (1, 1) and ways["pass1"] == (1, 1) and ways["pass2"] == (1, 1)
    assert ways["pass3_read"][0] == 2  # the price of an even n3 stride (16-byte aligned LDS-DMA pairs)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_piece_wait_counts_the_stores_behind_the_last_piece(tmp_path):
    import check_quad_isa
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = tmp_path / "quad.s"
    csrc = os.path.join(ROOT, "cyberether_amd", "csrc")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++20", "-ffp-contract=off", "-S", "--cuda-device-only",
                    "-I", os.path.join(csrc, "kernels"), "-I", csrc, "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tools", "ubench", "quad_bench.hip"), "-o", str(out)], check=True, capture_output=True)
    r = check_quad_isa.check(out.read_text())
    assert r["vmcnt"] == r["stores_behind_last_piece"] == 14, r
    assert not r["other_vmem_between"], r
