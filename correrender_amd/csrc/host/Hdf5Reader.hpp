// Hdf5Reader.hpp -- a small read-only decoder of the HDF5 container, exactly as far as NetCDF-4 files need it, with no
// dependency on libhdf5 / netcdf-c.
//
// Why it exists: the reference's generator writes `format='NETCDF4_CLASSIC'`
// (scripts/generate_synth_box_ensembles.py:151-158) and its loader opens whatever nc_open opens
// (src/Loaders/NetCdfLoader.cpp:282-339, 493-499) -- so the input files of the correlation path are HDF5 containers.  A
// machine that runs the reference has netcdf-c (NetCdfLoader binds it with dlopen when it is there); this decoder makes
// the path file -> VolumeData -> calculateCpu work WITHOUT it, e.g. on a GPU node that has nothing but this library.
//
// Written from the published format ("HDF5 File Format Specification Version 3.0"), pinned against files the real
// libhdf5 1.10.6 wrote (tests/golden/netcdf4/, tests/golden/make_netcdf4_fixtures.py).  Decoded:
//   superblock versions 0-3; object headers version 1 and 2 (continuation chunks); groups stored as symbol tables
//   (v1 B-tree + local heap), as compact link messages and as dense links (fractal heap + v2 B-tree);
//   dataspace v1/v2; datatypes: IEEE floats and integers of either byte order, fixed and variable-length strings,
//   object references, vlen sequences (DIMENSION_LIST); attributes v1-v3, compact or dense; global heap collections;
//   data layout v3 (compact, contiguous, chunked through the v1 B-tree) and v4 (single chunk, implicit, fixed array,
//   extensible array and v2 B-tree chunk indexes are reported as unsupported unless trivial); filters deflate, shuffle,
//   fletcher32.  Everything else raises Hdf5Error with the feature's name.
#pragma once
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace crfhost {

struct Hdf5Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct Hdf5Datatype {
    enum Class { FIXED = 0, FLOAT = 1, STRING = 3, COMPOUND = 6, REFERENCE = 7, VLEN = 9, OTHER = -1 };
    int cls = OTHER;
    uint32_t size = 0;        // bytes per element
    bool bigEndian = false;   // FIXED / FLOAT
    bool isSigned = false;    // FIXED
    bool vlenString = false;  // VLEN: a variable-length string (else a sequence of `base`)
    std::shared_ptr<Hdf5Datatype> base;  // VLEN
};

struct Hdf5Attribute {
    std::string name;
    Hdf5Datatype type;
    std::vector<uint64_t> shape;        // empty: scalar
    std::vector<unsigned char> raw;     // element data as stored
    // decoded conveniences
    bool isString = false;
    std::string text;                   // first string element
    bool isNumeric = false;
    std::vector<double> numbers;        // every element, converted
    std::vector<uint64_t> references;   // DIMENSION_LIST: the object header address each element's first reference names
};

struct Hdf5Filter {
    int id = 0;
    std::vector<uint32_t> clientData;
};

struct Hdf5Dataset {
    std::string name;              // link name in the root group
    uint64_t headerAddress = 0;
    std::vector<uint64_t> shape;
    Hdf5Datatype type;
    std::map<std::string, Hdf5Attribute> attributes;
    // storage
    int layoutClass = -1;          // 0 compact, 1 contiguous, 2 chunked
    uint64_t dataAddress = ~uint64_t(0), dataSize = 0;   // contiguous (address may be undefined: never written)
    std::vector<unsigned char> compactData;
    std::vector<uint64_t> chunkShape;                    // chunked: elements per chunk along every dimension
    uint64_t chunkIndexAddress = ~uint64_t(0);
    int chunkIndexType = 0;        // 0: v1 B-tree (layout v3); layout v4: 1 single chunk, 2 implicit, 3 fixed array, ...
    uint64_t singleChunkSize = 0;  // layout v4 single chunk with filters
    uint32_t singleChunkMask = 0;
    std::vector<Hdf5Filter> filters;
    bool isDimensionScale() const;
};

class Hdf5File {
public:
    explicit Hdf5File(const std::string& path);
    ~Hdf5File();
    Hdf5File(const Hdf5File&) = delete;
    Hdf5File& operator=(const Hdf5File&) = delete;

    /// The datasets linked from the root group, in link order (sub-groups are skipped: NETCDF4_CLASSIC has none).
    const std::vector<Hdf5Dataset>& datasets() const { return datasets_; }
    const std::map<std::string, Hdf5Attribute>& rootAttributes() const { return rootAttributes_; }
    int superblockVersion() const { return superblockVersion_; }

    /// Reads the hyperslab [start, start + count) of a float / double dataset, converted to float, row-major.
    void readFloats(const Hdf5Dataset& ds, const std::vector<uint64_t>& start, const std::vector<uint64_t>& count,
                    float* out) const;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
    std::vector<Hdf5Dataset> datasets_;
    std::map<std::string, Hdf5Attribute> rootAttributes_;
    int superblockVersion_ = -1;
};

}  // namespace crfhost
