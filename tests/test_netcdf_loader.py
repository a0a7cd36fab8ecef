"""On-disk input format: the C++ NetCDF classic reader of the host layer
(correrender_amd/csrc/host/NetCdfLoader.cpp, mirroring src/Loaders/NetCdfLoader.cpp) against files written by an
independent implementation (scipy.io.netcdf_file): CDF-1 and CDF-2, fixed and record (UNLIMITED) leading dimension,
float and double data, fill values, the dimension-name conventions, and rejection of what it cannot read."""
import subprocess
from pathlib import Path

import numpy as np
import pytest
from scipy.io import netcdf_file

from correrender_amd import synth
from parity import assert_bit_exact
import oracle_lib

EXE = Path(__file__).resolve().parent.parent / "correrender_amd" / "host_adapter_test"


def _write(path, data, *, version=1, lead="member", names=("lev", "lat", "lon"), record=False, dtype="f",
           var="data", attrs=None, extra=None):
    f = netcdf_file(str(path), "w", version=version)
    n, zs, ys, xs = data.shape
    f.createDimension(lead, None if record else n)
    for nm, ln in zip(names, (zs, ys, xs)):
        f.createDimension(nm, ln)
    v = f.createVariable(var, dtype, (lead,) + tuple(names))
    for k, val in (attrs or {}).items():
        setattr(v, k, val)
    v[:] = data
    for nm, arr in (extra or {}).items():
        w = f.createVariable(nm, "f", (lead,) + tuple(names))
        w[:] = arr
    # a coordinate variable and a 1-D non-field variable, which must not be taken for fields
    lon = f.createVariable(names[2], "f", (names[2],))
    lon[:] = np.arange(xs, dtype=np.float32)
    f.close()


def _run(mode, path, out_dir, env=None):
    import os
    r = subprocess.run([str(EXE), mode, str(path), str(out_dir)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, **(env or {})))
    return r


@pytest.fixture(scope="module")
def fake_netcdf(tmp_path_factory):
    """The test double of netcdf-c's C API (tests/fake_libnetcdf.c) as a shared object for CRF_LIBNETCDF."""
    out = tmp_path_factory.mktemp("fake") / "libnetcdf_fake.so"
    src = Path(__file__).resolve().parent / "fake_libnetcdf.c"
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-o", str(out), str(src)], check=True)
    return out


def _describe(path, dims, variables):
    """Side-car description read by the test double: dims = [(name, len)], variables = [(name, type, dimids, opts, array)]."""
    lines = ["dims %d %s" % (len(dims), " ".join(f"{n} {l}" for n, l in dims))]
    for name, nctype, dimids, opts, arr in variables:
        lines.append("var %s %d %d %s %s" % (name, nctype, len(dimids), " ".join(map(str, dimids)), " ".join(opts)))
        lines.append(" ".join(repr(float(x)) for x in np.asarray(arr, np.float64).reshape(-1)))
    Path(str(path) + ".fake").write_text("\n".join(lines) + "\n")


def test_netcdf4_goes_through_the_netcdf_c_binding(tmp_path, fake_netcdf):
    """An HDF5-signature file is handed to netcdf-c (dlopen of $CRF_LIBNETCDF): dimension conventions, standard_name,
    fill value -> NaN, double -> float and the per-member hyperslab reads are the loader's; the library (here its test
    double) only serves nc_* calls."""
    rng = np.random.default_rng(8)
    data = rng.standard_normal((4, 3, 2, 5))
    data[2, 1, 0, 3] = -9999.0
    other = rng.standard_normal((4, 3, 2, 5)).astype(np.float32)
    path = tmp_path / "n4.nc"
    path.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    _describe(path, [("members", 4), ("lev", 3), ("lat", 2), ("lon", 5)],
              [("lon", 5, [3], [], np.arange(5)),
               ("data", 6, [0, 1, 2, 3], ["standard_name=geopotential", "fill=-9999.0"], data),
               ("other", 5, [0, 1, 2, 3], [], other)])
    r = _run("netcdf", path, tmp_path, env={"CRF_LIBNETCDF": str(fake_netcdf)})
    assert r.returncode == 0 and "NETCDF-OK" in r.stdout, r.stdout + r.stderr
    meta = _parse(r.stdout)
    assert meta["grid"] == (5, 2, 3) and meta["es"] == 4 and meta["ts"] == 1
    assert meta["fields"] == ["geopotential", "other"] and not meta["warnings"]
    want = data.astype(np.float32)
    want[2, 1, 0, 3] = np.nan
    for e in range(4):
        np.testing.assert_array_equal(np.fromfile(tmp_path / f"geopotential_t0_e{e}.bin", np.float32), want[e].reshape(-1))
        np.testing.assert_array_equal(np.fromfile(tmp_path / f"other_t0_e{e}.bin", np.float32), other[e].reshape(-1))
    # a library without the expected entry points is reported as such
    bad = tmp_path / "empty.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(bad), "-x", "c", "/dev/null"], check=True)
    r = _run("netcdf", path, tmp_path, env={"CRF_LIBNETCDF": str(bad)})
    assert r.returncode == 2 and "lacks the expected nc_* entry points" in r.stderr


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


@pytest.mark.parametrize("version,record,dtype", [(1, False, "f"), (2, False, "f"), (1, True, "f"), (2, True, "d"),
                                                  (1, False, "d")])
def test_member_volumes_round_trip(tmp_path, version, record, dtype):
    rng = np.random.default_rng(version * 10 + record)
    data = rng.standard_normal((5, 3, 4, 7)).astype(np.float32)
    _write(tmp_path / "a.nc", data, version=version, record=record, dtype=dtype)
    r = _run("netcdf", tmp_path / "a.nc", tmp_path)
    assert r.returncode == 0 and "NETCDF-OK" in r.stdout, r.stdout + r.stderr
    meta = _parse(r.stdout)
    assert meta["grid"] == (7, 4, 3) and meta["ts"] == 1 and meta["es"] == 5
    assert meta["fields"] == ["data"] and not meta["warnings"]
    for e in range(5):
        got = np.fromfile(tmp_path / f"data_t0_e{e}.bin", np.float32)
        np.testing.assert_array_equal(got, data[e].reshape(-1))


def test_time_axis_fill_values_standard_name_and_second_record_variable(tmp_path):
    rng = np.random.default_rng(3)
    data = rng.standard_normal((4, 2, 3, 5)).astype(np.float32)
    data[1, 0, 1, 2] = -999.0
    other = rng.standard_normal((4, 2, 3, 5)).astype(np.float32)
    _write(tmp_path / "t.nc", data, lead="time", names=("z", "y", "x"), record=True,
           attrs={"standard_name": "air_temperature", "missing_value": np.float32(-999.0)}, extra={"other": other})
    r = _run("netcdf", tmp_path / "t.nc", tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    meta = _parse(r.stdout)
    assert meta["grid"] == (5, 3, 2) and meta["ts"] == 4 and meta["es"] == 1
    assert meta["fields"] == ["air_temperature", "other"]      # standard_name replaces the variable name
    want = data.copy()
    want[1, 0, 1, 2] = np.nan
    for t in range(4):                                           # two interleaved record variables
        np.testing.assert_array_equal(np.fromfile(tmp_path / f"air_temperature_t{t}_e0.bin", np.float32),
                                      want[t].reshape(-1))
        np.testing.assert_array_equal(np.fromfile(tmp_path / f"other_t{t}_e0.bin", np.float32), other[t].reshape(-1))


def test_unknown_leading_dimension_is_time_with_warning(tmp_path):
    data = np.zeros((3, 2, 2, 2), np.float32)
    _write(tmp_path / "u.nc", data, lead="realization", names=("level", "rlat", "rlon"))
    meta = _parse(_run("netcdf", tmp_path / "u.nc", tmp_path).stdout)
    assert meta["ts"] == 3 and meta["es"] == 1
    assert any("Assuming time" in w for w in meta["warnings"]) and any("positionally" in w for w in meta["warnings"])


def test_rejects_what_it_cannot_read(tmp_path):
    # NetCDF-4 / CDF-5 without the netcdf-c library (this image has none): a message that says so and how to go on
    (tmp_path / "h.nc").write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    r = _run("netcdf", tmp_path / "h.nc", tmp_path, env={"CRF_LIBNETCDF": str(tmp_path / "no_such_lib.so")})
    assert r.returncode == 2 and "NetCDF-4" in r.stderr and "nccopy" in r.stderr and "not found" in r.stderr
    (tmp_path / "c5.nc").write_bytes(b"CDF\x05" + b"\0" * 64)
    assert "CDF-5" in _run("netcdf", tmp_path / "c5.nc", tmp_path).stderr
    (tmp_path / "x.nc").write_bytes(b"not netcdf at all")
    assert "not a NetCDF file" in _run("netcdf", tmp_path / "x.nc", tmp_path).stderr
    good = tmp_path / "g.nc"
    _write(good, np.zeros((2, 2, 2, 2), np.float32))
    raw = good.read_bytes()
    (tmp_path / "trunc.nc").write_bytes(raw[:len(raw) - 40])
    assert "truncated" in _run("netcdf", tmp_path / "trunc.nc", tmp_path).stderr
    assert _run("netcdf", tmp_path / "missing.nc", tmp_path).returncode == 2


@pytest.mark.gpu
def test_netcdf_to_correlation_field(tmp_path, oracle):
    """File -> NetCdfLoader -> VolumeData -> CorrelationCalculator::calculateCpu -> HIP kernel, vs the oracle."""
    xs, ys, zs, cs = 16, 12, 8, 20
    ens = synth.box_ensemble(xs, ys, zs, cs, seed=77)
    _write(tmp_path / "box.nc", ens, version=2)
    r = _run("netcdf_compute", tmp_path / "box.nc", tmp_path)
    assert r.returncode == 0 and "NETCDF-COMPUTE-OK" in r.stdout, r.stdout + r.stderr
    ref = ens[:, zs // 2, ys // 2, xs // 2].copy()
    assert_bit_exact(np.fromfile(tmp_path / "pearson.bin", np.float32), oracle.field(oracle_lib.PEARSON, ens, ref),
                     "NetCDF -> Pearson field")
