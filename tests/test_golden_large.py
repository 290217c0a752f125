"""CPU checks of the large-N fixtures (tests/golden/large_N*.npz): their inputs are exactly what eqf_vio_amd/synth.py generates today (so the
GPU tests that drive the generator and the fixtures' expected outputs talk about the same stream), and the recorded outputs are sane.
(Re-running the structured oracle at these sizes takes minutes per frame: that is what the fixtures are for; tests/golden/make_golden_large.py
regenerates them.)"""
import numpy as np
import pytest

from helpers import load_golden


@pytest.mark.parametrize("N", [2000, 4000])
def test_large_fixture_inputs_are_the_generators_and_outputs_are_sane(N):
    from eqf_vio_amd import synth

    d, settings = load_golden(f"large_N{N}")
    st = synth.make_stream(N, seed=int(d["seed"][0]), duration=float(d["duration"][0]))
    nf = len(d["vision_stamps"])
    assert nf >= 2 and np.array_equal(st.imu, d["imu"]) and np.array_equal(st.vision_stamps[:nf], d["vision_stamps"])
    assert np.array_equal(st.ids, d["ids"]) and np.array_equal(st.bearings[:nf], d["bearings"])
    ref = synth.template_settings_dict()
    for k, v in settings.items():
        assert np.allclose(np.asarray(ref[k], dtype=float), np.asarray(v, dtype=float)), k
    fr = d["frames"]
    assert np.all(np.abs(np.linalg.norm(fr[:, 0:4], axis=1) - 1.0) < 1e-12)          # unit attitude quaternions
    assert np.all(fr[1:, 16] < fr[:-1, 16]) and np.all(fr[1:, 17] < fr[:-1, 17])      # |Sigma|_F and trace go down while landmarks converge
    n = 11 + 3 * N
    assert d["sample_rows"].max() < n and d["sample_cols"].max() < n and d["sigma_samples"].shape == (nf, 600)
    diag = d["sample_rows"] == d["sample_cols"]
    assert diag.sum() >= 64 and np.all(d["sigma_samples"][:, diag] > 0)                # sampled variances are positive
    for f in range(nf):
        B = d["sigma_base"][f]
        assert np.abs(B - B.T).max() <= 1e-9 * np.abs(B).max() and np.all(np.linalg.eigvalsh(0.5 * (B + B.T)) > 0)
