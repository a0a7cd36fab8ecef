#!/opt/conda/bin/python3.9
"""Writes the NetCDF-4 (HDF5) fixtures of tests/golden/netcdf4/ with the REAL HDF5 library.

The reference's generator (scripts/generate_synth_box_ensembles.py:151-158) writes
    Dataset(outfile, mode='w', format='NETCDF4_CLASSIC'); dimensions member, lev, lat, lon;
    createVariable('data', float32, ('member', 'lev', 'lat', 'lon'))
through netCDF4-python -> netcdf-c -> libhdf5.  Neither netCDF4-python nor netcdf-c exists in this image, but libhdf5
1.10.6 and h5py 3.3 do (under /opt/conda, python3.9 only -- run this script with /opt/conda/bin/python3.9).  The files
are therefore produced by the same HDF5 library netcdf-c sits on, following the netCDF-4 on-disk conventions netcdf-c
documents ("NetCDF-4/HDF5 file format", docs/file_format_specifications.md) and h5netcdf implements on h5py:
  * creation order of links and attributes tracked (netcdf-c requires it: H5Pset_link_creation_order /
    H5Pset_attr_creation_order) -- which makes libhdf5 write version-2 object headers with link messages even under the
    "earliest" format bounds;
  * every dimension is an HDF5 dimension scale (H5DSset_scale): a dataset named like the dimension with attributes
    CLASS = "DIMENSION_SCALE" and NAME = "This is a netCDF dimension but not a netCDF variable.%10d" when no coordinate
    variable of that name exists, plus _Netcdf4Dimid;
  * the variable's dimensions are attached with H5DSattach_scale (DIMENSION_LIST: variable-length object references in
    the global heap; REFERENCE_LIST on the scales);
  * root attributes _NCProperties and, for NETCDF4_CLASSIC, _nc3_strict.
Variants cover what real data sets do on top of that: chunked + shuffle + deflate storage, float64 data, _FillValue,
coordinate variables, a `time` leading axis, `standard_name`, more than eight variables in the root group (dense link
storage: fractal heap + v2 B-tree), and the three format generations libhdf5 can write (earliest: superblock 0 / v1
B-trees; v18: superblock 2; latest of 1.10: superblock 3 / version-4 layout messages).

Outputs: tests/golden/netcdf4/<builder>_<generation>.nc and <builder>.npz (the arrays the files hold, for the tests).
The arrays are deterministic; the files carry HDF5 modification times, so regenerated files differ in a few bytes.
"""
import sys
from pathlib import Path

import h5py
import numpy as np

OUT = Path(__file__).resolve().parent / "netcdf4"
NOT_A_VARIABLE = b"This is a netCDF dimension but not a netCDF variable."


def ncproperties():
    return np.string_(f"version=2,netcdf=4.8.1,hdf5={h5py.version.hdf5_version}")


def add_dimension(f, name, size, dimid, coordinate=None):
    """A netCDF-4 dimension: a dimension-scale dataset.  coordinate: values of a coordinate variable of the same name."""
    if coordinate is None:
        ds = f.create_dataset(name, shape=(size,), dtype=">f4", track_order=True)
        ds.make_scale(NOT_A_VARIABLE + b"%10d" % size)
    else:
        ds = f.create_dataset(name, data=coordinate, track_order=True)
        ds.make_scale(name)
    ds.attrs["_Netcdf4Dimid"] = np.int32(dimid)
    return ds


def add_variable(f, name, data, scales, **kw):
    ds = f.create_dataset(name, data=data, track_order=True, **kw)
    for i, s in enumerate(scales):
        ds.dims[i].attach_scale(s)
    return ds


def box_like(shape, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal(shape).astype(np.float32)


def write(name, libver, builder):
    """<builder>_<name>.nc for one format generation; the arrays (identical for every generation) once, as <builder>.npz"""
    path = OUT / f"{builder.__name__}_{name}.nc"
    with h5py.File(path, "w", libver=libver, track_order=True) as f:
        f.attrs["_NCProperties"] = ncproperties()
        arrays = builder(f)
    np.savez(OUT / f"{builder.__name__}.npz", **arrays)
    print(f"{path.name}: {path.stat().st_size} bytes")


def generator_layout(f):
    """Exactly the generator's file: member x lev x lat x lon float32, contiguous, no attributes."""
    f.attrs["_nc3_strict"] = np.int32(1)
    m, z, y, x = 5, 3, 4, 6
    dims = [add_dimension(f, n, s, i) for i, (n, s) in enumerate([("member", m), ("lev", z), ("lat", y), ("lon", x)])]
    data = box_like((m, z, y, x), 1)
    add_variable(f, "data", data, dims)
    return {"data": data}


def box_ensemble(f):
    """The generator's layout holding the synthetic box ensemble the GPU tests use (correrender_amd/synth.py restates the
    generator's recipe): file -> NetCdfLoader -> VolumeData -> calculateCpu must give the oracle's Pearson field."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("synth", Path(__file__).resolve().parents[2] / "correrender_amd" / "synth.py")
    synth = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(synth)
    f.attrs["_nc3_strict"] = np.int32(1)
    xs, ys, zs, cs = 12, 8, 6, 16
    data = synth.box_ensemble(xs, ys, zs, cs, seed=77)
    dims = [add_dimension(f, n, s, i) for i, (n, s) in enumerate([("member", cs), ("lev", zs), ("lat", ys), ("lon", xs)])]
    add_variable(f, "data", data, dims)
    return {"data": data}


def chunked_deflate(f):
    """Compressed storage (what real ensemble data sets use), _FillValue, float64 data, coordinate variables, several
    fields with standard_name, `ensemble` as the member axis name."""
    f.attrs["_nc3_strict"] = np.int32(1)
    e, z, y, x = 4, 5, 7, 9
    lat = np.linspace(40.0, 50.0, y)
    lon = np.linspace(0.0, 16.0, x)
    dims = [add_dimension(f, "ensemble", e, 0), add_dimension(f, "lev", z, 1),
            add_dimension(f, "lat", y, 2, coordinate=lat), add_dimension(f, "lon", x, 3, coordinate=lon)]
    t = box_like((e, z, y, x), 2)
    t[1, 2, 3, 4] = np.float32(-999.0)
    t[3, 0, 0, 0] = np.float32(-999.0)
    ds = add_variable(f, "t", t, dims, chunks=(1, 2, 4, 5), compression="gzip", compression_opts=4, shuffle=True,
                      fillvalue=np.float32(-999.0))
    ds.attrs["_FillValue"] = np.array([-999.0], dtype=np.float32)
    ds.attrs["standard_name"] = np.string_("air_temperature")
    ds.attrs["units"] = np.string_("K")
    p = box_like((e, z, y, x), 3).astype(np.float64) * 1e3
    ds = add_variable(f, "p", p, dims, chunks=(2, 5, 7, 9), compression="gzip", compression_opts=1)
    ds.attrs["long_name"] = np.string_("pressure")
    # more than eight attributes: dense attribute storage (fractal heap + v2 B-tree) -- CF metadata easily gets there
    for i, (k, v) in enumerate([("units", "Pa"), ("cell_methods", "time: mean"), ("grid_mapping", "rotated_pole"),
                                ("coordinates", "lon lat"), ("institution", "test"), ("source", "synthetic"),
                                ("comment", "dense attribute storage"), ("history", "none")]):
        ds.attrs[k] = np.string_(v)
    ds.attrs["standard_name"] = np.string_("air_pressure")
    ds.attrs["missing_value"] = np.float64(-5.0e3)
    p[0, 0, 0, 1] = -5.0e3
    ds[...] = p
    q = box_like((e, z, y, x), 4)
    add_variable(f, "q", q, dims, chunks=(4, 1, 7, 9), shuffle=True)      # chunked, shuffle only, partial edge chunks none
    r = box_like((e, z, y, x), 5)
    add_variable(f, "r", r, dims, chunks=(3, 2, 3, 4))                     # chunked, unfiltered, edge chunks on every axis
    return {"t": t, "p": p, "q": q, "r": r, "lat": lat, "lon": lon}


def time_axis_many_variables(f):
    """A `time` leading axis and more than eight links in the root group (dense link storage: fractal heap + v2 B-tree
    under creation-order tracking), variable-length string attributes (global heap)."""
    t, z, y, x = 3, 2, 3, 5
    dims = [add_dimension(f, "time", t, 0, coordinate=np.arange(t, dtype=np.float64) * 3600.0), add_dimension(f, "z", z, 1),
            add_dimension(f, "y", y, 2), add_dimension(f, "x", x, 3)]
    arrays = {}
    for i in range(9):
        name = f"var{i:02d}"
        a = box_like((t, z, y, x), 10 + i)
        ds = add_variable(f, name, a, dims)
        ds.attrs["standard_name"] = f"quantity_{i}"                       # str -> variable-length string
        if i == 4:
            ds.attrs["missing_value"] = np.float64(1e20)
            a[2, 1, 2, 4] = np.float32(1e20)
            ds[...] = a
        arrays[name] = a
    # a variable on another grid: must be ignored by the loader (its trailing dimension lengths do not match)
    add_dimension(f, "station", 4, 4)
    f.create_dataset("station_height", data=np.arange(4, dtype=np.float32), track_order=True)
    return arrays


def plain_hdf5_old_style(path):
    """No creation-order tracking, earliest format: version-1 object headers, symbol-table groups (v1 B-tree + local
    heap) -- what older netcdf-c / plain h5py files look like.  Dimension names come from the dimension scales."""
    with h5py.File(path, "w", libver="earliest") as f:
        m, z, y, x = 3, 2, 4, 5
        scales = []
        for i, (n, s) in enumerate([("members", m), ("zs", z), ("ys", y), ("xs", x)]):
            ds = f.create_dataset(n, shape=(s,), dtype=">f4")
            ds.make_scale(NOT_A_VARIABLE + b"%10d" % s)
            scales.append(ds)
        data = box_like((m, z, y, x), 30)
        ds = f.create_dataset("data", data=data)
        for i, s in enumerate(scales):
            ds.dims[i].attach_scale(s)
    np.savez(OUT / "plain_hdf5_old_style.npz", data=data)
    print(f"{path.name}: {path.stat().st_size} bytes")


def main():
    OUT.mkdir(exist_ok=True)
    generations = [("earliest", "earliest"),         # superblock 0: what netcdf-c >= 4.6.2 writes (bounds EARLIEST..V18)
                   ("v18", ("v108", "v108")),        # superblock 2 (netcdf-c 4.4 - 4.6.1 on HDF5 1.8 with LATEST bounds)
                   ("latest", ("v110", "v110"))]     # superblock 3, version-4 layout messages, new chunk indexes
    for name, libver in generations:
        write(name, libver, generator_layout)
        write(name, libver, chunked_deflate)
        write(name, libver, time_axis_many_variables)
    write("earliest", "earliest", box_ensemble)
    plain_hdf5_old_style(OUT / "plain_hdf5_old_style_earliest.nc")
    return 0


if __name__ == "__main__":
    sys.exit(main())
