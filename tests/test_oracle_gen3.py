"""Property tests for the gen-3 (receive) part of the oracle.  The reference
holds no test or stored output for it (parity unpinned), so these pin
the restatement through physics it must obey."""
import numpy as np
import pytest

from beifong_amd import capi, scenes
from tests.oracle_lib import OracleScene


def test_pulse_gates_returns_to_range_bins():
    """wignertransmitter 'pulse' with tau = 2 dr / c: a return from range R lands
    in ADC bin ~ R / dr.  The antenna is 0.3 m above the ground plane, so nothing
    arrives before bin 3 (rectangle.cpp:132-220, wignertransmitter.cpp:111-146,
    :422-425 retarded time).  The Wigner gain is signed, and the pulse train is
    periodic (fmodulo with 1/prf = T), so bin 0 may hold a faint wrapped return."""
    sd, lp = scenes.bus_receive(n_tris=2000, n_paths=100000)
    h, _, st = OracleScene(sd).render(lp, threads=8)
    y = h.reshape(256, 3)[:, 0]
    assert np.all(y[1:3] == 0) and np.abs(y[3:8]).sum() > 0
    assert abs(y[0]) < 1e-3 * np.abs(y).max()
    assert st.n_invalid == 0


def test_area_transmitter_is_time_invariant():
    """areatransmitter has no signal model: receive time is uniform, so the
    weight channel is flat and Y does not depend on the bin (within MC noise)."""
    sd, lp = scenes.bus_receive(n_tris=2000, n_paths=200000, transmitter="area")
    h, _, _ = OracleScene(sd).render(lp, threads=8)
    h = h.reshape(256, 3)
    w = h[:, 2]
    assert abs(w.mean() - 200000 / 256) < 1 and w.std() < 4 * np.sqrt(w.mean())
    y = h[:, 0] / w
    assert y.std() / y.mean() < 0.5


def test_weight_channel_counts_every_sample():
    sd, lp = scenes.bus_receive(n_tris=2000, n_paths=50000)
    h, _, st = OracleScene(sd).render(lp, threads=4)
    assert h.reshape(256, 3)[:, 2].sum() == 50000 - st.n_invalid


def test_serial_stream_matches_per_path_streams_statistically():
    """rng_mode 1 = the reference's literal single PCG32 stream (integrator.cpp:
    219-231); rng_mode 0 = per-path streams (what the HIP path runs).  Same
    estimator, so per-bin means agree within Monte-Carlo noise."""
    sd, lp = scenes.trans_rad(spp=400000)
    o = OracleScene(sd)
    a, ra, _ = o.render(lp, rng_mode=0, threads=8, records=True)
    b, _, _ = o.render(lp, rng_mode=1)
    n = lp.n_paths
    bins_a = a[5:].reshape(50, 3)[:, 1] / n
    bins_b = b[5:].reshape(50, 3)[:, 1] / n
    # per-bin standard error from the per-path records
    idx = np.floor(ra["aux"] / np.float32(0.5e-9)).astype(int)
    for k in range(50):
        sel = ra["L"][idx == k] / np.pi          # records hold the sensor-weighted radiance (x pi)
        var = (np.sum(sel.astype(np.float64) ** 2) / n - (np.sum(sel) / n) ** 2) / n
        assert abs(bins_a[k] - bins_b[k]) <= 6 * np.sqrt(2 * max(var, 0)) + 1e-9, k


def test_phase_integrator_leaves_base_channels_alone_and_bins_by_last_segment():
    """receive o phase o pathtimefrequency (phase.cpp is built at HEAD).  The S{k}.Y channels follow
    Y, A, W; they never change those.  The phase a path reports is that of its LAST traced segment
    (spawn_ray resets ray.phase, interaction.h:61-64), -2 pi t / K with K = half the band WIDTH
    (ray.h:92, Q4): with a band so wide that K = 1 km every finite phase is a small negative number,
    wraps to just under 2 pi and must land in the last bin; paths whose last ray escaped carry
    phase -inf -> fmod = NaN -> no bin."""
    P = 8
    sd, lp = scenes.bus_receive(n_tris=2000, n_paths=60000)
    base, _, _ = OracleScene(sd).render(lp, threads=8)
    sd.physics.lambda_min_nm = 0.0
    sd.physics.lambda_max_nm = 2.0e12            # (max - min) / 2 * 1e-9 = 1000 m
    sd.finalize()
    ref3, _, _ = OracleScene(sd).render(lp, threads=8)      # other band: other wavelengths, other paths
    lp.phase_bins = P
    h, _, st = OracleScene(sd).render(lp, threads=8)
    h = h.reshape(256, 3 + P)
    assert np.array_equal(h[:, :3], ref3.reshape(256, 3))
    assert base.shape == ref3.shape
    s = h[:, 3:]
    assert np.all(s[:, :P - 1] == 0)
    assert s[:, P - 1].sum() > 0
    # only paths that ended on a surface (Russian roulette / absorbed) have a finite last segment:
    # far fewer than the valid ones
    assert s[:, P - 1].sum() < 4.0 * np.abs(h[:, 0]).sum() + 1e30


def test_phase_bins_cover_the_circle_for_a_narrow_band():
    """With the fork's default band K is ~1 mm, so segment lengths of metres spread the phases of the
    paths that end on a surface over all bins."""
    P = 16
    sd, lp = scenes.bus_receive(n_tris=2000, n_paths=400000, transmitter="area")   # no signal gating: L > 0
    lp.phase_bins = P
    h, _, _ = OracleScene(sd).render(lp, threads=8)
    s = h.reshape(256, 3 + P)[:, 3:].sum(0)
    assert np.all(s > 0)            # few, heavy-tailed contributions: coverage only, no flatness claim


def test_iq_mode_is_a_phasor_sum_of_the_raw_contributions():
    """BF_MODE_RECEIVE_IQ (physical mode; no reference counterpart): every contribution c of a path becomes
    c * exp(-j 2 pi L / lambda) with L its own optical length.  (1) The weight channel and the sample
    bookkeeping are those of RECEIVE_RAW.  (2) |I + jQ| <= Y per path, with equality when the path
    has a single contribution (depth limit 2, tiny transmitter: almost every path).  (3) As lambda
    grows beyond every path length the phasors align: I -> Y, Q -> 0."""
    sd, lp = scenes.bus_receive(n_tris=2000, n_paths=40000, transmitter="area")
    lp.max_depth = 2
    o = OracleScene(sd)
    raw, rr, _ = o.render(lp, records=True, threads=8)
    lp.mode = capi.BF_MODE_RECEIVE_IQ
    iq, ri, st = o.render(lp, records=True, threads=8)
    raw, iq = raw.reshape(256, 3), iq.reshape(256, 3)
    assert np.array_equal(raw[:, 2], iq[:, 2])
    y = rr["L"].astype(np.float64)
    mag = np.hypot(ri["L"].astype(np.float64), ri["aux"].astype(np.float64))
    lit = y > 0
    assert lit.sum() > 1000
    assert np.all(mag[lit] <= y[lit] * (1 + 1e-5))
    assert np.mean(np.abs(mag[lit] - y[lit]) <= 1e-5 * y[lit]) > 0.99
    assert np.all(mag[~lit] == 0)
    # the phases are spread: the coherent sum per bin is well below the incoherent one
    assert np.hypot(iq[:, 0], iq[:, 1]).sum() < 0.5 * raw[:, 0].sum()
    # lambda -> infinity
    sd.physics.lambda_min_nm = 0.9e18
    sd.physics.lambda_max_nm = 1.1e18
    sd.finalize()
    o = OracleScene(sd)
    lp.mode = capi.BF_MODE_RECEIVE_RAW
    raw, _, _ = o.render(lp, threads=8)
    lp.mode = capi.BF_MODE_RECEIVE_IQ
    iq, _, _ = o.render(lp, threads=8)
    raw, iq = raw.reshape(256, 3), iq.reshape(256, 3)
    assert np.allclose(iq[:, 0], raw[:, 0], rtol=1e-5, atol=1e-9 * raw[:, 0].max())
    assert np.abs(iq[:, 1]).max() <= 1e-5 * raw[:, 0].max()


def test_doppler_hook_value_and_default():
    """Shape::doppler (src/librender/shape.cpp:388): 2 dot(si.wi, velocity * to_local(si.p)) / MTS_C * wavelength.  With the
    identity velocity dot(wi, to_local(p)) = -d . p (orthonormal frame), so a path whose first hit is p (ray direction d)
    gets d_lambda = -2 (d . p) / c * lambda at that hit: checked through the frequency row the sample lands in for a
    one-bounce-deep render (max_depth = 1: nothing but the first intersection contributes a shift)."""
    from beifong_amd import capi, scenes
    from tests.oracle_lib import OracleScene
    sd, lp = scenes.bus_receive(n_tris=2000, n_paths=4000, t_bins=8)
    sd.sensor.f_bins = 64
    c, lmin, lmax = sd.physics.c, sd.physics.lambda_min_nm, sd.physics.lambda_max_nm
    sd.sensor.f_bandwidth = c / (lmin * 1e-9)
    sd.finalize()
    lp.bins_y = 64
    lp.max_depth = 1
    o = OracleScene(sd)
    h0, _, _ = o.render(lp, threads=4)
    lp.flags = capi.BF_FLAG_DOPPLER
    h1, _, _ = o.render(lp, threads=4)
    w0 = h0.reshape(64, 8, 3)[:, :, 2].sum(1)
    w1 = h1.reshape(64, 8, 3)[:, :, 2].sum(1)
    assert w0.sum() > 0 and not np.array_equal(w0, w1)
    # the rays leave the radar at (0, 0, 0.3) along +x and hit points with d . p > 0: every shift is negative, i.e. towards
    # SHORTER wavelengths = higher frequency rows (or out of the ADC at its top)
    centre0 = (np.arange(64) * w0).sum() / w0.sum()
    centre1 = (np.arange(64) * w1).sum() / max(w1.sum(), 1)
    assert w1.sum() <= w0.sum() and (w1.sum() == 0 or centre1 > centre0)


def test_mix_resample_bins_the_beat_frequency():
    """receive_type "mix_resample" (integrator.cpp:1588-1603): tf[1] = |c / lambda_after - f_rx|.  At the reference's HEAD
    the wavelength never changes, the beat is 0 and SignalBlock::put drops every sample (lo = ceil(0 - 1) = -1); with the
    Doppler hook on, the beat rows are the raw rows folded about the receive frequency: with f_rx on the edge between raw
    rows 31 | 32 and rows of equal width, W_mix[k] = W_raw[32 + k] + W_raw[31 - k]."""
    from beifong_amd import capi, scenes
    from tests.oracle_lib import OracleScene
    lam = 8.0e6                                                   # nm; a band 2e-6 wide: f_rx is one frequency to 1e-6
    sd, lp = scenes.bus_receive(n_tris=2000, n_paths=6000, t_bins=4, lambda_band_nm=(lam * (1 - 1e-6), lam * (1 + 1e-6)))
    f0 = sd.physics.c / (lam * 1e-9)
    vel = np.eye(4, dtype=np.float32) * np.float32(2.0)
    vel[3, 3] = 1.0
    for s_ in sd.shapes:
        if s_.type == capi.BF_SHAPE_MESH:
            s_.velocity = (capi.M16)(*vel.reshape(-1).tolist())
    sd.sensor.f_bins = 64
    sd.sensor.f_bandwidth = 2.0 * f0
    sd.finalize()
    lp.bins_y = 64
    lp.flags = capi.BF_FLAG_DOPPLER
    raw, _, _ = OracleScene(sd).render(lp, threads=4)
    w_raw = raw.reshape(64, 4, 3)[:, :, 2].sum(1)
    assert w_raw.sum() == lp.n_paths or w_raw.sum() > 0.9 * lp.n_paths
    assert (w_raw[:31].sum() + w_raw[33:].sum()) > 0                # the hook moves samples off the receive frequency
    sd.sensor.f_bins = 32
    sd.sensor.f_bandwidth = f0
    sd.finalize()
    lp.bins_y = 32
    lp.flags = capi.BF_FLAG_DOPPLER | capi.BF_FLAG_MIX_RESAMPLE
    mix, _, st = OracleScene(sd).render(lp, threads=4)
    w_mix = mix.reshape(32, 4, 3)[:, :, 2].sum(1)
    fold = w_raw[32:] + w_raw[31::-1]
    # a path with NO shift (it missed the moving mesh) has beat exactly 0 and is dropped: those sit in raw row 31 (f_rx is
    # at or just below the edge) -- fold row 0 counts them, the mix row 0 does not
    assert np.array_equal(w_mix[1:], fold[1:]) and 0 < w_mix[0] <= fold[0]
    # without the hook: nothing lands in the ADC
    lp.flags = capi.BF_FLAG_MIX_RESAMPLE
    none, _, st0 = OracleScene(sd).render(lp, threads=4)
    assert not none.any()
    # render modes do not take the flag; the Wigner receiver has a local oscillator of its own under it (round 4), except a
    # "pulse" that is a delta signal (an uninitialised frequency in the reference)
    sdw, lpw = scenes.bus_receive(n_tris=500, n_paths=16, t_bins=4, receiver="wigner")
    sdw.sensor.rx_signal_type, sdw.sensor.rx_sig_is_delta = capi.BF_SIGNAL_PULSE, 1
    sdw.finalize()
    lpw.flags = capi.BF_FLAG_MIX_RESAMPLE
    with pytest.raises(Exception):
        OracleScene(sdw).render(lpw, threads=1)


@pytest.mark.parametrize("plate_x", [3.0, 5.0, 9.0])
def test_fmcw_dechirp_known_answer_the_beat_row_is_the_range(plate_x):
    """A PHYSICAL pin of the fork-specific de-chirp path (nothing of the reference's pins it): a resample_freq chirp transmitter,
    the Wigner receiver's local oscillator under "mix_resample" (wignertransmitter.cpp:152-168, 211-221, 430-441;
    wignerreceiver.cpp:149-189; integrator.cpp:1588-1603) and a plate at range r — the beat B / T * 2 r / c puts the plate's
    energy into ADC row ceil(64 r / 12.8 - 1) (slant ranges r .. r + 0.02 m), whenever the return is received."""
    sd, lp = scenes.fmcw_plate(plate_x=plate_x, r_max=12.8, n_paths=1 << 17)
    h, _, st = OracleScene(sd).render(lp, threads=8)
    energy = np.abs(h.reshape(64, 8, 3)[:, :, 0]).sum(1)
    assert energy.sum() > 0
    pos = 64.0 * plate_x / 12.8
    rows = (int(np.ceil(pos - 1.0)), int(np.ceil(pos - 1.0)) + 1)
    assert int(np.argmax(energy)) in rows, (rows, np.flatnonzero(energy))
    assert energy[rows[0]:rows[1] + 1].sum() > 0.95 * energy.sum()
    # every row of the receive window shows the same range: the beat does not depend on the receive time
    per_t = np.abs(h.reshape(64, 8, 3)[:, :, 0])
    for k in range(8):
        if per_t[:, k].sum() > 0 and k >= 2:            # (the first returns wrap into the previous chirp: other rows)
            assert int(np.argmax(per_t[:, k])) in rows
