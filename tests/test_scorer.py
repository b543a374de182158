"""The scorer restatement against the reference's evaluate_sinr / evaluate_bler outputs (tests/golden/eval.npz)."""
import numpy as np
import pytest

from conftest import load_golden
from sig_sdp_mmw_amd import scorer
from sig_sdp_mmw_amd.graphs import journal_graph


@pytest.mark.parametrize("name", ["env75", "env108"])
def test_sinr_and_bler_match_reference(name):
    g = load_golden("eval")
    state, geo = journal_graph(int(g[name + "_cell_size"]), 75e-4, int(g[name + "_seed"]), return_geometry=True)
    rx = scorer.receive_power(geo["sta_locs"], geo["ap_locs"])
    for suffix, z, Z in (("", g[name + "_z_vec"], int(g[name + "_Z"])), ("_bad", np.arange(rx.shape[0]) % 3, 3)):
        sinr = scorer.evaluate_sinr(rx, z, Z)
        np.testing.assert_allclose(sinr, g[name + "_sinr" + suffix], rtol=1e-10)
        bler = scorer.evaluate_bler(rx, z, Z)
        np.testing.assert_allclose(bler, g[name + "_bler" + suffix], rtol=1e-8, atol=1e-300)
    assert np.sum(g[name + "_sinr_bad"] == 1e-3) > 0  # the bad colouring really exercises the collision rule
