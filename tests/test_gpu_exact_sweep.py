"""GPU: exhaustive sweeps (all 2^32 float bit patterns) of the one-float functions behind the exact
Amplitude -> Range epilogue and behind the fast provider's Spectrogram bin guard, run inside the shipped
library through the C ABI (jst_probe_exact_sweep, csrc/kernels/exact_sweep.hip).

Why it is exhaustive: everything from the power p = re^2 + im^2 on is a function of ONE float.  The main-path
epilogue (device_math.hh / libm_float.hh) replaces the compiler's general sqrt and divide expansions and the
published tanhf class ladder by shorter sequences; each is admissible only if it returns the bits of the general
form on every float, and the general form is what the oracle / glibc pin (tests/test_libm_float.py on the host,
tests/test_gpu_elementwise.py and the chain tests on the device).  Bit-exact: 0 mismatches allowed."""
import ctypes as C
import math

import pytest

pytestmark = pytest.mark.gpu


def sweep(js, which, coeff=0.0, scale=0.0, offset=0.0, height=0.0):
    bad, visited, first = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
    r = js._lib.jst_probe_exact_sweep(which, coeff, scale, offset, height, C.byref(bad), C.byref(visited), C.byref(first))
    assert r == 0, js.last_error()
    return bad.value, visited.value, first.value


# (amplitude coeff, range scale, range offset): the bench configuration (N = 4096, -100..0 dB), the 65536-point
# configuration (-130..-10 dB), a narrow high range, and one that puts every power into the middle tanh classes
PARAMS = [
    (20.0 * math.log10(1.0 / 4096.0), 1.0 / 100.0, 1.0),
    (20.0 * math.log10(1.0 / 65536.0), 1.0 / 120.0, 130.0 / 120.0),
    (20.0 * math.log10(1.0 / 8.0), 1.0 / 40.0, 30.0 / 40.0),
    (0.0, 1.0 / 300.0, 0.5),
]


@pytest.fixture(scope="module")
def js():
    import cyberether_amd.jetstream as js
    return js


def test_main_path_sqrt_is_correctly_rounded_on_its_whole_domain(js):
    bad, visited, first = sweep(js, 0)
    assert visited == 0x71800000 - 0x0d800000 + 1  # every float of [2^-100, 2^100]
    assert bad == 0, f"{bad} mismatches, first at bits {first:#x}"


def test_main_path_tanhf_equals_the_class_ladder_on_every_float(js):
    bad, visited, first = sweep(js, 1)
    assert visited == 2 * (0x40f00000 - 0x32800000)  # 2^-26 <= |x| < 7.5, both signs
    assert bad == 0, f"{bad} mismatches, first at bits {first:#x}"


@pytest.mark.parametrize("coeff,scale,offset", PARAMS)
def test_amplitude_range_from_every_power(js, coeff, scale, offset):
    bad, _, first = sweep(js, 2, coeff, scale, offset)
    assert bad == 0, f"{bad} mismatches, first at power bits {first:#x}"


def test_amplitude_from_every_power(js):
    bad, visited, first = sweep(js, 3, PARAMS[0][0])
    assert visited == 0x71800000 - 0x0d800000 + 1
    assert bad == 0, f"{bad} mismatches, first at power bits {first:#x}"


@pytest.mark.parametrize("height", [256.0, 64.0, 512.0, 1000.0, 2048.0])
@pytest.mark.parametrize("coeff,scale,offset", PARAMS[:3])
def test_fast_provider_bins_are_exact_with_the_guard_on_every_power(js, coeff, scale, offset, height):
    """the guard width was empirical; this is the proof by exhaustion, per height."""
    bad, kept_fast, first = sweep(js, 4, coeff, scale, offset, height)
    assert bad == 0, f"{bad} powers land in another Spectrogram bin, first at bits {first:#x}"
    assert kept_fast > 1 << 20  # the guard did not simply send everything down the exact path


def test_without_the_guard_bins_do_move(js):
    bad, _, _ = sweep(js, 5, *PARAMS[0], 256.0)
    assert bad > 0  # the sweep can tell


@pytest.mark.parametrize("coeff,scale,offset", PARAMS)
def test_lean_fast_value_deviates_less_than_4e7_on_every_power(js, coeff, scale, offset):
    """Round 3's lean fast form (folded cubic as four fused multiply-adds, logistic through v_exp / v_rcp): its largest
    deviation from the exact provider over EVERY power of its domain (measured: 2.4e-7 .. 3.6e-7 over the four parameter
    sets).  The bin guard's width (height x 7.5e-7) covers this bound plus the two roundings of the value x height product
    (1.2e-7) with margin, and is itself proven per height by the exhaustive bin sweep above; BASELINE's float tolerance
    is 1e-5."""
    bad, visited, dev_bits = sweep(js, 6, coeff, scale, offset)
    # round 4: the lean form's domain is every power whose magnitude is a positive normal float (one v_cmp_class on
    # sqrt(p)) -- at least the round-3 domain [2^-100, 2^100], at most every positive normal power
    assert 0x71800000 - 0x0d800000 + 1 <= visited <= 0x7f800000 - 0x00000001 and bad == 0
    import struct
    dev = struct.unpack("<f", struct.pack("<I", dev_bits))[0]
    print(f"largest |lean - exact| = {dev:.3e} (coeff {coeff:.2f}, scale {scale:.4f})")
    assert 0.0 < dev <= 4.0e-7, dev


def test_libm_pinned_on_this_box():
    """The bit-exact float comparisons of this suite take their STRICT branch only where the host libm is the one the parity is
    stated against (glibc 2.35 x86-64, tests/golden/libm_pin.json); elsewhere tests/util.py: assert_bit_equal falls back to
    1e-5 of the peak with a warning.  This test FAILS on such a box, so that a green GPU run records which branch ran."""
    from util import libm_pinned
    assert libm_pinned(), "host libm differs from the pinned glibc 2.35: float parity was judged at 1e-5 of the peak, not bit for bit"
