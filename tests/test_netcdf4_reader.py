"""NetCDF-4 input files (HDF5 containers -- the format the reference's generator writes,
scripts/generate_synth_box_ensembles.py:151-158) through the built-in decoder of the host layer
(correrender_amd/csrc/host/Hdf5Reader.cpp behind NetCdfLoader), with NO netcdf-c / libhdf5 at run time.

The fixtures under tests/golden/netcdf4/ were written by the real libhdf5 1.10.6 through h5py, following the netCDF-4
on-disk conventions (tests/golden/make_netcdf4_fixtures.py, which documents them); <builder>.npz holds the arrays each
file contains.  Three HDF5 format generations per layout: superblock 0 ("earliest" bounds: what current netcdf-c
writes), superblock 2 (v1.8 bounds) and superblock 3 (v1.10: version-4 layout messages, fixed-array chunk indexes)."""
import os
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from parity import assert_bit_exact
import oracle_lib

ROOT = Path(__file__).resolve().parent.parent
EXE = ROOT / "correrender_amd" / "host_adapter_test"
FIX = ROOT / "tests" / "golden" / "netcdf4"
GENERATIONS = ["earliest", "v18", "latest"]


def _run(mode, path, out_dir, env=None):
    clean = {k: v for k, v in os.environ.items() if k not in ("CRF_LIBNETCDF", "CRF_NETCDF_BACKEND")}
    return subprocess.run([str(EXE), mode, str(path), str(out_dir)], capture_output=True, text=True, timeout=300,
                          env=dict(clean, **(env or {})))


def _parse(stdout):
    meta = {"fields": [], "warnings": []}
    for line in stdout.splitlines():
        if line.startswith("grid "):
            t = line.split()
            meta["grid"] = tuple(int(x) for x in t[1:4])
            meta["ts"], meta["es"] = int(t[5]), int(t[7])
        elif line.startswith("field "):
            meta["fields"].append(line[6:])
        elif line.startswith("warning "):
            meta["warnings"].append(line[8:])
    return meta


def _load(tmp_path, name):
    r = _run("netcdf", FIX / name, tmp_path)
    assert r.returncode == 0 and "NETCDF-OK" in r.stdout, r.stdout + r.stderr
    return _parse(r.stdout), (lambda field, t, e: np.fromfile(tmp_path / f"{field}_t{t}_e{e}.bin", np.float32))


@pytest.mark.parametrize("generation", GENERATIONS)
def test_the_generators_layout(tmp_path, generation):
    """member x lev x lat x lon float32, contiguous, dimensions without coordinate variables: exactly what
    generate_synth_box_ensembles.py creates."""
    meta, vol = _load(tmp_path, f"generator_layout_{generation}.nc")
    want = np.load(FIX / "generator_layout.npz")["data"]
    m, z, y, x = want.shape
    assert meta["grid"] == (x, y, z) and meta["es"] == m and meta["ts"] == 1
    assert meta["fields"] == ["data"] and not meta["warnings"]          # member / lev / lat / lon: the reference's names
    for e in range(m):
        np.testing.assert_array_equal(vol("data", 0, e), want[e].reshape(-1))


@pytest.mark.parametrize("generation", GENERATIONS)
def test_chunked_compressed_variables_fill_values_doubles_and_dense_attributes(tmp_path, generation):
    """Chunked storage with shuffle + deflate, chunk shapes that do not divide the grid, float64 data, _FillValue /
    missing_value -> NaN, standard_name as the field name, coordinate variables, a variable with more than eight
    attributes (dense attribute storage), `ensemble` as the member axis."""
    meta, vol = _load(tmp_path, f"chunked_deflate_{generation}.nc")
    arrays = np.load(FIX / "chunked_deflate.npz")
    e_, z, y, x = arrays["t"].shape
    assert meta["grid"] == (x, y, z) and meta["es"] == e_ and meta["ts"] == 1 and not meta["warnings"]
    assert meta["fields"] == ["air_temperature", "air_pressure", "q", "r"]       # creation order, like netcdf-c's varids
    t = arrays["t"].copy()
    t[t == np.float32(-999.0)] = np.nan
    p = arrays["p"].astype(np.float32)                                            # NC_DOUBLE is converted to float
    p[arrays["p"] == -5.0e3] = np.nan
    assert np.isnan(t).sum() == 2 and np.isnan(p).sum() == 1
    for e in range(e_):
        np.testing.assert_array_equal(vol("air_temperature", 0, e), t[e].reshape(-1))
        np.testing.assert_array_equal(vol("air_pressure", 0, e), p[e].reshape(-1))
        np.testing.assert_array_equal(vol("q", 0, e), arrays["q"][e].reshape(-1))
        np.testing.assert_array_equal(vol("r", 0, e), arrays["r"][e].reshape(-1))


@pytest.mark.parametrize("generation", GENERATIONS)
def test_time_axis_and_a_root_group_with_dense_link_storage(tmp_path, generation):
    """Fifteen links in the root group (fractal heap + v2 B-tree), a `time` leading axis with a coordinate variable,
    variable-length string attributes in the global heap, a variable on another grid that must be ignored."""
    meta, vol = _load(tmp_path, f"time_axis_many_variables_{generation}.nc")
    arrays = np.load(FIX / "time_axis_many_variables.npz")
    t_, z, y, x = arrays["var00"].shape
    assert meta["grid"] == (x, y, z) and meta["ts"] == t_ and meta["es"] == 1
    assert meta["fields"] == [f"quantity_{i}" for i in range(9)]
    assert not meta["warnings"]                                                   # z / y / x and time are known names
    for i in range(9):
        want = arrays[f"var{i:02d}"].copy()
        if i == 4:
            want[want == np.float32(1e20)] = np.nan
            assert np.isnan(want).sum() == 1
        for t in range(t_):
            np.testing.assert_array_equal(vol(f"quantity_{i}", t, 0), want[t].reshape(-1))


def test_plain_hdf5_with_old_style_groups(tmp_path):
    """No creation-order tracking: version-1 object headers, the root group as a symbol table (v1 B-tree + local heap)."""
    meta, vol = _load(tmp_path, "plain_hdf5_old_style_earliest.nc")
    want = np.load(FIX / "plain_hdf5_old_style.npz")["data"]
    m, z, y, x = want.shape
    assert meta["grid"] == (x, y, z) and meta["es"] == m and meta["fields"] == ["data"]
    for e in range(m):
        np.testing.assert_array_equal(vol("data", 0, e), want[e].reshape(-1))


def test_damaged_files_are_rejected_not_crashed_on(tmp_path):
    """Truncation and random byte damage: an error message (exit code 2) or, when the damage misses every structure,
    a normal result -- never a crash or a hang.  Without netcdf-c the message names both ways out."""
    raw = (FIX / "chunked_deflate_earliest.nc").read_bytes()
    (tmp_path / "trunc.nc").write_bytes(raw[:len(raw) // 2])
    r = _run("netcdf", tmp_path / "trunc.nc", tmp_path)
    assert r.returncode == 2 and "built-in HDF5 decoder" in r.stderr and "netcdf-c" in r.stderr, r.stderr
    rng = np.random.default_rng(1)
    for source in ["chunked_deflate_earliest.nc", "time_axis_many_variables_v18.nc", "generator_layout_latest.nc"]:
        raw = bytearray((FIX / source).read_bytes())
        for trial in range(12):
            bad = bytearray(raw)
            for _ in range(8):
                bad[int(rng.integers(8, len(bad)))] = int(rng.integers(0, 256))
            (tmp_path / "fuzz.nc").write_bytes(bytes(bad))
            r = _run("netcdf", tmp_path / "fuzz.nc", tmp_path)
            assert r.returncode in (0, 2), f"{source} trial {trial}: rc={r.returncode}\n{r.stderr[-400:]}"


def test_netcdf_c_takes_over_when_asked_for(tmp_path):
    """CRF_LIBNETCDF / CRF_NETCDF_BACKEND=library put the netcdf-c binding first (here: absent -> its message)."""
    r = _run("netcdf", FIX / "generator_layout_earliest.nc", tmp_path, env={"CRF_NETCDF_BACKEND": "library"})
    assert r.returncode == 2 and "netcdf-c library" in r.stderr and "not found" in r.stderr


@pytest.mark.skipif(not Path("/opt/conda/bin/python3.9").exists(), reason="needs the image's h5py (python3.9 under /opt/conda)")
def test_freshly_written_files_read_back(tmp_path):
    """Where the image's h5py / libhdf5 exist: regenerate every fixture with the real library now and read the new files
    (guards against the committed fixtures and the generator drifting apart)."""
    script = tmp_path / "make.py"
    script.write_text((ROOT / "tests" / "golden" / "make_netcdf4_fixtures.py").read_text().replace(
        'OUT = Path(__file__).resolve().parent / "netcdf4"', f'OUT = Path({str(tmp_path / "fresh")!r})').replace(
        "Path(__file__).resolve().parents[2]", f"Path({str(ROOT)!r})"))
    r = subprocess.run(["/opt/conda/bin/python3.9", str(script)], capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and "No module named" in r.stderr:
        pytest.skip("h5py is not importable on this machine")
    assert r.returncode == 0, r.stderr
    for nc in sorted((tmp_path / "fresh").glob("*.nc")):
        out = tmp_path / "out"
        shutil.rmtree(out, ignore_errors=True)
        out.mkdir()
        rr = _run("netcdf", nc, out)
        assert rr.returncode == 0 and "NETCDF-OK" in rr.stdout, nc.name + ": " + rr.stderr
        builder = nc.stem.rsplit("_", 1)[0]
        arrays = np.load(tmp_path / "fresh" / f"{builder}.npz")
        first = sorted(arrays.files)[0] if builder != "chunked_deflate" else "q"
        got = np.fromfile(sorted(out.glob("*_t0_e0.bin"))[0], np.float32)
        assert got.size == arrays[first][0].size


@pytest.mark.gpu
def test_generator_format_file_to_correlation_field(tmp_path, oracle):
    """A file in the generator's format (NETCDF4_CLASSIC layout, written by libhdf5) -> NetCdfLoader (built-in decoder)
    -> createVolumeData() -> CorrelationCalculator::calculateCpu -> HIP kernel: bit-exact Pearson vs the oracle."""
    r = _run("netcdf_compute", FIX / "box_ensemble_earliest.nc", tmp_path)
    assert r.returncode == 0 and "NETCDF-COMPUTE-OK" in r.stdout, r.stdout + r.stderr
    ens = np.load(FIX / "box_ensemble.npz")["data"]
    cs, zs, ys, xs = ens.shape
    from correrender_amd import synth
    np.testing.assert_array_equal(ens, synth.box_ensemble(xs, ys, zs, cs, seed=77))   # the fixture IS the synthetic ensemble
    ref = ens[:, zs // 2, ys // 2, xs // 2].copy()
    assert_bit_exact(np.fromfile(tmp_path / "pearson.bin", np.float32), oracle.field(oracle_lib.PEARSON, ens, ref),
                     "NetCDF-4 file -> Pearson field")
