"""The C++ host layer above the C ABI (correrender_amd/csrc/host): mirror of the reference's Calculator /
VolumeData / CorrelationCalculator surface.  `settings` runs on CPU; `compute` needs the GPU and is compared with the
oracle evaluated with the parameters the reference's calculateCpu would derive (CorrelationCalculator.cpp:781-866)."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from correrender_amd import default_kraskov_k, synth
from parity import assert_bit_exact, assert_close
import oracle_lib

ROOT = Path(__file__).resolve().parent.parent
EXE = ROOT / "correrender_amd" / "host_adapter_test"


def test_settings_defaults_naming_and_clamping():
    r = subprocess.run([str(EXE), "settings"], capture_output=True, text=True)
    assert r.returncode == 0 and "SETTINGS-OK" in r.stdout, r.stderr


def _run_compute(tmp_path, data, devices=None):
    """data: float32 [nfields, ts, es, zs, ys, xs]; devices: value of the adapter's "devices" setting (a device group)"""
    nf, ts, es, zs, ys, xs = data.shape
    tmp_path.mkdir(exist_ok=True)
    inp = tmp_path / "in.bin"
    with open(inp, "wb") as f:
        np.array([xs, ys, zs, ts, es, nf], np.int32).tofile(f)
        np.ascontiguousarray(data, np.float32).tofile(f)
    cmd = [str(EXE), "compute", str(inp), str(tmp_path)] + ([devices] if devices else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "COMPUTE-OK" in r.stdout, r.stdout + r.stderr
    return lambda tag: np.fromfile(tmp_path / f"{tag}.bin", np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_calculate_cpu_on_a_device_group_is_bit_identical(tmp_path, devices):
    """The adapter's `devices` setting: calculateCpu over a crf_group (z-slabs, one worker per slot; two / three slots
    rehearsed on the box's one GPU) -- every scripted scenario, all seven measures, the SEPARATE / time-lag /
    SEPARATE_SYMMETRIC modes -- must reproduce the single-context fields bit for bit."""
    xs, ys, zs, cs = 24, 16, 7, 32
    a = synth.box_ensemble(xs, ys, zs, cs, seed=51)
    b = synth.box_ensemble(xs, ys, zs, cs, seed=52)
    data = np.stack([np.stack([a * 0.5 + 1.0, a]), np.stack([b, b * 2.0])])
    single = _run_compute(tmp_path / "single", data)
    group = _run_compute(tmp_path / "group", data, devices)
    tags = ["m_pearson", "m_spearman", "m_kendall", "m_mi_binned", "m_mi_kraskov", "m_binned_mi_correlation_coefficient",
            "m_kmi_correlation_coefficient", "pearson_ref123", "kraskov2_k3_ref123", "spearman_separate",
            "binned_separate", "pearson_separate_lag0", "kendall_symmetric", "binned_symmetric"]
    for tag in tags:
        assert_bit_exact(group(tag), single(tag), f"adapter on devices={devices}: {tag}")


@pytest.mark.gpu
def test_calculate_cpu_ensemble_mode(tmp_path, oracle):
    xs, ys, zs, cs = 24, 16, 8, 32
    a = synth.box_ensemble(xs, ys, zs, cs, seed=31)
    b = synth.box_ensemble(xs, ys, zs, cs, seed=32)
    # two time steps (t=1 is the one evaluated), two fields
    data = np.stack([np.stack([a * 0.5 + 1.0, a]), np.stack([b, b * 2.0])])      # [nf=2, ts=2, es, z, y, x]
    out = _run_compute(tmp_path, data)
    ens = data[0, 1]
    centre = (xs // 2, ys // 2, zs // 2)
    refc = ens[:, centre[2], centre[1], centre[0]].copy()
    mm = oracle.minmax(ens)
    k = default_kraskov_k(cs)
    assert_bit_exact(out("m_pearson"), oracle.field(oracle_lib.PEARSON, ens, refc), "adapter pearson")
    assert_bit_exact(out("m_spearman"), oracle.field(oracle_lib.SPEARMAN, ens, refc), "adapter spearman")
    assert_bit_exact(out("m_kendall"), oracle.field(oracle_lib.KENDALL, ens, refc), "adapter kendall")
    assert_close(out("m_mi_binned"), oracle.field(oracle_lib.MI_BINNED, ens, refc, num_bins=80, minmax_ref=mm), "binned")
    assert_close(out("m_binned_mi_correlation_coefficient"),
                 oracle.field(oracle_lib.BINNED_MI_CC, ens, refc, num_bins=80, minmax_ref=mm), "binned cc")
    assert_close(out("m_mi_kraskov"), oracle.field(oracle_lib.MI_KRASKOV, ens, refc, k=k), "kraskov")
    assert_close(out("m_kmi_correlation_coefficient"), oracle.field(oracle_lib.KMI_CC, ens, refc, k=k), "kmi cc")
    ref123 = ens[:, 3, 2, 1].copy()
    assert_bit_exact(out("pearson_ref123"), oracle.field(oracle_lib.PEARSON, ens, ref123), "moved reference point")
    assert_close(out("kraskov2_k3_ref123"), oracle.field(oracle_lib.MI_KRASKOV, ens, ref123, k=3, estimator=2), "ksg2")
    # SEPARATE mode: reference vector from field 2 at the same time step; binned ranges per field (:820-846)
    ens2 = data[1, 1]
    ref_sep = ens2[:, 3, 2, 1].copy()
    assert_bit_exact(out("spearman_separate"), oracle.field(oracle_lib.SPEARMAN, ens, ref_sep), "separate spearman")
    assert_close(out("binned_separate"), oracle.field(oracle_lib.MI_BINNED, ens, ref_sep, num_bins=80,
                                                      minmax_ref=oracle.minmax(ens2), minmax_query=mm), "separate binned")
    ref_lag = data[1, 0][:, 3, 2, 1].copy()                                      # time-lag: field 2 at time step 0
    assert_bit_exact(out("pearson_separate_lag0"), oracle.field(oracle_lib.PEARSON, ens, ref_lag), "time lag")
    # sibling ensemble calculators (EnsembleMean / Spread / SetPredicate / DKL mirrors) on field 1 at the same time step
    assert_bit_exact(out("ensemble_mean"), oracle.ensemble_stat(0, ens), "adapter ensemble mean")
    assert_bit_exact(out("ensemble_spread"), oracle.ensemble_stat(1, ens), "adapter ensemble spread")
    assert_bit_exact(out("set_predicate"), oracle.set_predicate(0, 0.25, cs // 2, cs // 2, ens), "adapter set predicate")
    assert_close(out("dkl_knn"), oracle.dkl(1, ens, k=default_kraskov_k(cs)), "adapter DKL k-NN")
    assert_close(out("dkl_binned"), oracle.dkl(0, ens, num_bins=16), "adapter DKL binned")
    # SEPARATE_SYMMETRIC: field 1 (reference side) vs field 2 (query side) at every voxel
    assert_bit_exact(out("kendall_symmetric"), oracle.symmetric_field(oracle_lib.KENDALL, ens, ens2), "symmetric kendall")
    assert_close(out("binned_symmetric"), oracle.symmetric_field(oracle_lib.MI_BINNED, ens, ens2, num_bins=80,
                                                                 minmax_ref=mm, minmax_query=oracle.minmax(ens2)),
                 "symmetric binned")


@pytest.mark.gpu
def test_calculate_cpu_time_mode(tmp_path, oracle):
    """es = 1, ts = cs: the member axis is time (CorrelationCalculator.cpp:85-91,131-138)."""
    xs, ys, zs, cs = 16, 12, 6, 20
    a = synth.box_ensemble(xs, ys, zs, cs, seed=41)
    data = a[None, :, None]                                                      # [nf=1, ts=cs, es=1, z, y, x]
    out = _run_compute(tmp_path, data)
    refc = a[:, zs // 2, ys // 2, xs // 2].copy()
    assert_bit_exact(out("m_pearson"), oracle.field(oracle_lib.PEARSON, a, refc), "time-mode pearson")
    assert_bit_exact(out("m_kendall"), oracle.field(oracle_lib.KENDALL, a, refc), "time-mode kendall")
    assert_close(out("m_mi_kraskov"), oracle.field(oracle_lib.MI_KRASKOV, a, refc, k=default_kraskov_k(cs)), "time-mode ksg")


@pytest.mark.gpu
def test_calculate_cpu_divergent_field_range(tmp_path, oracle):
    """A field named "Helicity" is divergent in the reference: getMinMaxScalarFieldValue centres its range at zero
    (VolumeData.cpp:616-621, 1661-1666), which changes the binned-MI normalisation (CorrelationCalculator.cpp:820-846)."""
    xs, ys, zs, cs = 16, 12, 6, 16
    a = synth.box_ensemble(xs, ys, zs, cs, seed=61)
    h = synth.normal_ensemble(xs, ys, zs, cs, seed=62) + 0.75          # asymmetric range
    data = np.stack([a[None], a[None] * 2.0, h[None]])                 # [nf=3, ts=1, es, z, y, x]
    out = _run_compute(tmp_path, data)
    mn, mx = oracle.minmax(h)
    m = max(abs(mn), abs(mx))
    ref = h[:, 3, 2, 1].copy()
    want = oracle.field(oracle_lib.MI_BINNED, h, ref, num_bins=80, minmax_ref=(-m, m))
    assert_close(out("binned_helicity"), want, "binned MI on a divergent field")
    plain = oracle.field(oracle_lib.MI_BINNED, h, ref, num_bins=80, minmax_ref=(mn, mx))
    assert not np.array_equal(want, plain)                             # the symmetrised range matters
