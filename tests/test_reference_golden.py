"""CPU: the oracle (oracle/jst_oracle.c) reproduces, BIT FOR BIT, the outputs the reference itself produced for every
frozen case of tests/golden/reference_vectors.npz (tools/make_reference_vectors.py ran them on the reference compiled in
place), and those outputs satisfy what the reference's own tests assert about them (transcribed in
tests/reference_cases.py: filter_engine/block_tests.cc:584-724, filter/block_tests.cc:343-400,
fm/module_tests.cc:204-483).  Runs without the reference tree -- this is the pin that travels."""
import numpy as np
import pytest

import reference_cases as rc


@pytest.mark.parametrize("name", rc.names())
def test_golden_outputs_meet_the_reference_tests_expectations(name):
    rc.expectations(name, rc.load()[name]["outs"])


@pytest.mark.parametrize("name", rc.names())
def test_oracle_reproduces_the_reference_outputs(oracle, name):
    case = rc.load()[name]
    got = rc.run_oracle(oracle, case)
    for c, (g, want) in enumerate(zip(got, case["outs"])):
        g = np.asarray(g)
        assert g.shape == want.shape, (name, c, g.shape, want.shape)
        assert np.array_equal(np.ascontiguousarray(g).view(np.uint32), np.ascontiguousarray(want).view(np.uint32)), \
            f"{name} cycle {c}: oracle differs from the reference ({case['source']})"
    rc.expectations(name, got)


def test_every_transcribed_reference_vector_has_a_case():
    have = set(rc.names())
    for needed in ("filter_engine_center_pos", "filter_engine_center_neg", "filter_engine_center_wrapped",
                   "filter_engine_head_centers", "filter_block_head_centers", "fm_narrow_deemphasis",
                   "fm_wide_stereo_multiplex", "fm_wide_tone_separation", "fm_nonfinite_none", "fm_nonfinite_50us",
                   "fm_cross_submission", "config4_chain", "spectrum_engine_c1_c2_rows"):
        assert needed in have
