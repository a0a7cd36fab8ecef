// NetCdfLoader.cpp -- see NetCdfLoader.hpp.  File format: "The NetCDF Classic Format Specification" (header grammar:
// magic numrecs dim_list gatt_list var_list; all integers big-endian, names and values padded to 4 bytes).
#include "NetCdfLoader.hpp"

#include <algorithm>
#include <map>

#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "Calculator.hpp"
#include "VolumeData.hpp"

namespace crfhost {

namespace {
enum { NC_BYTE = 1, NC_CHAR = 2, NC_SHORT = 3, NC_INT = 4, NC_FLOAT = 5, NC_DOUBLE = 6 };
enum { TAG_DIMENSION = 0x0A, TAG_VARIABLE = 0x0B, TAG_ATTRIBUTE = 0x0C };

size_t typeSize(int type) {
    switch (type) {
        case NC_BYTE:
        case NC_CHAR: return 1;
        case NC_SHORT: return 2;
        case NC_INT:
        case NC_FLOAT: return 4;
        case NC_DOUBLE: return 8;
        default: return 0;
    }
}

// sequential big-endian reader over the header
struct Cursor {
    FILE* f;
    const std::string& path;
    void bytes(void* dst, size_t n) {
        if (n && std::fread(dst, 1, n, f) != n) throw CalculatorError("Error in NetCdfLoader: truncated header in \"" + path + "\".");
    }
    uint32_t u32() {
        unsigned char b[4];
        bytes(b, 4);
        return (uint32_t(b[0]) << 24) | (uint32_t(b[1]) << 16) | (uint32_t(b[2]) << 8) | uint32_t(b[3]);
    }
    uint64_t u64() {
        const uint64_t hi = u32();
        return (hi << 32) | u32();
    }
    void skipPadding(uint64_t n) {
        const uint64_t pad = (4 - (n & 3)) & 3;
        unsigned char b[4];
        bytes(b, size_t(pad));
    }
    std::string name() {
        const uint32_t n = u32();
        if (n > (1u << 20)) throw CalculatorError("Error in NetCdfLoader: implausible name length in \"" + path + "\".");
        std::string s(n, '\0');
        bytes(s.data(), n);
        skipPadding(n);
        return s;
    }
};

float beFloat(const unsigned char* p) {
    const uint32_t u = (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]);
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
double beDouble(const unsigned char* p) {
    uint64_t u = 0;
    for (int i = 0; i < 8; i++) u = (u << 8) | p[i];
    double d;
    std::memcpy(&d, &u, 8);
    return d;
}

bool isOneOf(const std::string& s, std::initializer_list<const char*> names) {
    for (const char* n : names)
        if (s == n) return true;
    return false;
}
}  // namespace

// ---- netcdf-c through dlopen (NetCDF-4 / CDF-5 files) ------------------------------------------------------------
// The handful of entry points of the library's stable C API that the reference's loader uses as well
// (NetCdfLoader.cpp:282-339 nc_open / nc_inq / nc_inq_dim / nc_inq_var, :93-99 and :826-935 nc_get_vara_float,
// :753-765 nc_get_att_text / nc_get_att_float).  nc_type values are those of netcdf.h and equal the classic tags above.
struct NetCdfLoader::Library {
    void* handle = nullptr;
    int (*open)(const char*, int, int*) = nullptr;
    int (*close)(int) = nullptr;
    int (*inq)(int, int*, int*, int*, int*) = nullptr;
    int (*inq_dim)(int, int, char*, size_t*) = nullptr;
    int (*inq_var)(int, int, char*, int*, int*, int*, int*) = nullptr;
    int (*inq_att)(int, int, const char*, int*, size_t*) = nullptr;
    int (*get_att_text)(int, int, const char*, char*) = nullptr;
    int (*get_att_float)(int, int, const char*, float*) = nullptr;
    int (*get_vara_float)(int, int, const size_t*, const size_t*, float*) = nullptr;
    const char* (*strerror)(int) = nullptr;
    std::string tried;
    bool load() {
        std::vector<std::string> names;
        if (const char* env = std::getenv("CRF_LIBNETCDF"); env && *env) names.push_back(env);
        for (const char* n : {"libnetcdf.so", "libnetcdf.so.19", "libnetcdf.so.18", "libnetcdf.so.15", "libnetcdf.so.13",
                              "libnetcdf.so.11", "libnetcdf.so.7"})
            names.push_back(n);
        for (const std::string& n : names) {
            handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
            tried += (tried.empty() ? "" : ", ") + n;
        }
        if (!handle) return false;
        auto sym = [&](const char* name) { return dlsym(handle, name); };
        open = reinterpret_cast<decltype(open)>(sym("nc_open"));
        close = reinterpret_cast<decltype(close)>(sym("nc_close"));
        inq = reinterpret_cast<decltype(inq)>(sym("nc_inq"));
        inq_dim = reinterpret_cast<decltype(inq_dim)>(sym("nc_inq_dim"));
        inq_var = reinterpret_cast<decltype(inq_var)>(sym("nc_inq_var"));
        inq_att = reinterpret_cast<decltype(inq_att)>(sym("nc_inq_att"));
        get_att_text = reinterpret_cast<decltype(get_att_text)>(sym("nc_get_att_text"));
        get_att_float = reinterpret_cast<decltype(get_att_float)>(sym("nc_get_att_float"));
        get_vara_float = reinterpret_cast<decltype(get_vara_float)>(sym("nc_get_vara_float"));
        strerror = reinterpret_cast<decltype(strerror)>(sym("nc_strerror"));
        return open && close && inq && inq_dim && inq_var && inq_att && get_att_text && get_att_float && get_vara_float;
    }
    ~Library() {
        if (handle) dlclose(handle);
    }
};

void NetCdfLoader::openWithLibrary(const char* why) {
    if (file) std::fclose(file);
    file = nullptr;
    library = std::make_unique<Library>();
    if (!library->load()) {
        const std::string tried = library->tried;
        const bool found = library->handle != nullptr;
        library.reset();
        error(std::string(why) + " is read through the netcdf-c library, which " +
              (found ? "lacks the expected nc_* entry points" : "was not found (tried " + tried + ")") +
              "; install libnetcdf or set CRF_LIBNETCDF, or convert the file with `nccopy -k classic`");
    }
    auto check = [&](int status, const char* what) {
        if (status != 0)
            error(std::string(what) + " failed: " + (library->strerror ? library->strerror(status) : "netcdf error " + std::to_string(status)));
    };
    check(library->open(path.c_str(), 0 /* NC_NOWRITE */, &ncid), "nc_open");
    int ndims = 0, nvars = 0, ngatts = 0, unlimited = -1;
    check(library->inq(ncid, &ndims, &nvars, &ngatts, &unlimited), "nc_inq");
    char name[257];
    for (int d = 0; d < ndims; d++) {
        size_t len = 0;
        check(library->inq_dim(ncid, d, name, &len), "nc_inq_dim");
        dims.push_back(Dim{name, uint64_t(len)});  // the CURRENT length, also for an unlimited dimension
    }
    for (int i = 0; i < nvars; i++) {
        Var v;
        int type = 0, rank = 0, natts = 0;
        int dimids[1024];
        check(library->inq_var(ncid, i, name, &type, &rank, dimids, &natts), "nc_inq_var");
        v.name = name;
        v.type = type;
        v.varid = i;
        v.dimids.assign(dimids, dimids + rank);
        int atype = 0;
        size_t alen = 0;
        if (library->inq_att(ncid, i, "standard_name", &atype, &alen) == 0 && atype == NC_CHAR && alen < (1u << 16)) {
            std::string text(alen, '\0');
            check(library->get_att_text(ncid, i, "standard_name", text.data()), "nc_get_att_text");
            while (!text.empty() && text.back() == '\0') text.pop_back();
            v.standardName = text;
        }
        for (const char* fill : {"missing_value", "_FillValue"}) {  // the later one wins, as in the classic path
            if (library->inq_att(ncid, i, fill, &atype, &alen) == 0 && alen >= 1 && (atype == NC_FLOAT || atype == NC_DOUBLE)) {
                std::vector<float> values(alen);
                check(library->get_att_float(ncid, i, fill, values.data()), "nc_get_att_float");
                v.hasFill = true;
                v.fillValue = values[0];
            }
        }
        vars.push_back(std::move(v));
    }
}

// NetCDF-4 through the built-in HDF5 decoder: the netCDF-4 data model on top of HDF5 objects ("NetCDF-4/HDF5 file
// format", netcdf-c docs/file_format_specifications.md):
//   dimension  = a dimension-scale dataset (attribute CLASS = "DIMENSION_SCALE") named like the dimension; its NAME
//                attribute starts with "This is a netCDF dimension but not a netCDF variable." when no coordinate
//                variable of that name exists; _Netcdf4Dimid orders the dimensions;
//   variable   = every other dataset (and the coordinate variables); its dimensions are the scales its DIMENSION_LIST
//                attribute refers to (object references); datasets without one get anonymous dimensions, as in nc_open.
void NetCdfLoader::openWithHdf5Reader() {
    if (file) std::fclose(file);
    file = nullptr;
    hdf5 = std::make_unique<Hdf5File>(path);
    const std::string notAVariable = "This is a netCDF dimension but not a netCDF variable.";
    const auto& sets = hdf5->datasets();
    // 1. dimensions, ordered by _Netcdf4Dimid where present (else by appearance)
    struct Scale {
        int dimid;
        size_t order;
        const Hdf5Dataset* ds;
    };
    std::vector<Scale> scales;
    for (size_t i = 0; i < sets.size(); i++) {
        if (!sets[i].isDimensionScale() || sets[i].shape.size() != 1) continue;
        int dimid = 1 << 30;
        auto it = sets[i].attributes.find("_Netcdf4Dimid");
        if (it != sets[i].attributes.end() && it->second.isNumeric && !it->second.numbers.empty()) dimid = int(it->second.numbers[0]);
        scales.push_back(Scale{dimid, i, &sets[i]});
    }
    std::stable_sort(scales.begin(), scales.end(), [](const Scale& a, const Scale& b) { return a.dimid < b.dimid; });
    std::map<uint64_t, int> dimOfHeader;
    for (const Scale& s : scales) {
        dimOfHeader[s.ds->headerAddress] = int(dims.size());
        dims.push_back(Dim{s.ds->name, s.ds->shape[0]});
    }
    // 2. variables
    for (size_t i = 0; i < sets.size(); i++) {
        const Hdf5Dataset& ds = sets[i];
        if (ds.isDimensionScale()) {
            auto it = ds.attributes.find("NAME");
            if (it != ds.attributes.end() && it->second.isString && it->second.text.compare(0, notAVariable.size(), notAVariable) == 0)
                continue;  // a dimension without a coordinate variable
        }
        Var v;
        v.name = ds.name;
        v.varid = int(i);
        v.type = ds.type.cls == Hdf5Datatype::FLOAT ? (ds.type.size == 4 ? NC_FLOAT : ds.type.size == 8 ? NC_DOUBLE : 0) : 0;
        auto list = ds.attributes.find("DIMENSION_LIST");
        for (size_t d = 0; d < ds.shape.size(); d++) {
            int dimid = -1;
            if (ds.isDimensionScale() && ds.shape.size() == 1) {
                dimid = dimOfHeader.at(ds.headerAddress);  // a coordinate variable is its own dimension
            } else if (list != ds.attributes.end() && d < list->second.references.size()) {
                auto hit = dimOfHeader.find(list->second.references[d]);
                if (hit != dimOfHeader.end() && dims[size_t(hit->second)].length == ds.shape[d]) dimid = hit->second;
            }
            if (dimid < 0) {  // no scale attached: an anonymous dimension of this length (nc_open names them phony_dim_N)
                dimid = int(dims.size());
                dims.push_back(Dim{"phony_dim_" + std::to_string(dimid), ds.shape[d]});
            }
            v.dimids.push_back(dimid);
        }
        auto sn = ds.attributes.find("standard_name");
        if (sn != ds.attributes.end() && sn->second.isString) v.standardName = sn->second.text;
        for (const char* fill : {"missing_value", "_FillValue"}) {  // the later one wins, as in the classic path
            auto a = ds.attributes.find(fill);
            if (a != ds.attributes.end() && a->second.isNumeric && a->second.type.cls == Hdf5Datatype::FLOAT && !a->second.numbers.empty()) {
                v.hasFill = true;
                v.fillValue = float(a->second.numbers[0]);
            }
        }
        vars.push_back(std::move(v));
    }
}

void NetCdfLoader::error(const std::string& msg) const {
    throw CalculatorError("Error in NetCdfLoader: " + msg + " (file \"" + path + "\").");
}

NetCdfLoader::~NetCdfLoader() {
    if (file) std::fclose(file);
    if (library && ncid >= 0) library->close(ncid);
}

uint64_t NetCdfLoader::dimLength(int dimid) const {
    const Dim& d = dims.at(size_t(dimid));
    return (d.length == 0 && !library && !hdf5) ? numRecs : d.length;
}

NetCdfLoader::NetCdfLoader(const std::string& filePath) : path(filePath) {
    file = std::fopen(path.c_str(), "rb");
    if (!file) error("file could not be opened");
    unsigned char magic[4] = {0, 0, 0, 0};
    if (std::fread(magic, 1, 4, file) != 4) error("truncated header");
    if (magic[0] == 0x89 && magic[1] == 'H' && magic[2] == 'D' && magic[3] == 'F') {
        // NetCDF-4 = an HDF5 container.  Default: the built-in decoder (Hdf5Reader, no library needed).  netcdf-c takes
        // over when CRF_LIBNETCDF names it or CRF_NETCDF_BACKEND=library asks for it, and as the fallback for whatever the
        // decoder does not read (it says which feature).
        const char* backend = std::getenv("CRF_NETCDF_BACKEND");
        const bool wantLibrary = std::getenv("CRF_LIBNETCDF") != nullptr || (backend && std::string(backend) == "library");
        if (wantLibrary) {
            openWithLibrary("NetCDF-4 (an HDF5 container)");
        } else {
            try {
                openWithHdf5Reader();
            } catch (const Hdf5Error& e) {
                const std::string why = e.what();
                hdf5.reset();
                dims.clear();
                vars.clear();
                try {
                    openWithLibrary("NetCDF-4 (an HDF5 container)");
                } catch (const CalculatorError&) {
                    error("the built-in HDF5 decoder could not read this NetCDF-4 file (" + why +
                          ") and the netcdf-c library is not available (install libnetcdf or set CRF_LIBNETCDF, or convert the "
                          "file with `nccopy -k classic`)");
                }
                warnings.push_back("built-in HDF5 decoder: " + why + "; read through netcdf-c instead");
            }
        }
    } else if (magic[0] == 'C' && magic[1] == 'D' && magic[2] == 'F' && magic[3] == 5) {
        openWithLibrary("CDF-5 (64-bit data)");
    } else {
        if (magic[0] != 'C' || magic[1] != 'D' || magic[2] != 'F') error("not a NetCDF file");
        version = magic[3];
        if (version != 1 && version != 2) error("unknown NetCDF classic version byte");
        parseClassicHeader();
    }
    deriveGridAndFields();
}

void NetCdfLoader::parseClassicHeader() {
    Cursor c{file, path};
    const uint32_t recs = c.u32();
    numRecs = recs == 0xFFFFFFFFu ? 0 : recs;  // STREAMING: record count not recorded; fixed by the file size below

    auto listHeader = [&](uint32_t expectedTag) -> uint32_t {
        const uint32_t tag = c.u32(), n = c.u32();
        if (tag == 0 && n == 0) return 0;  // ABSENT
        if (tag != expectedTag) error("malformed header (unexpected list tag)");
        return n;
    };
    struct Attr {
        std::string name;
        int type;
        std::vector<unsigned char> raw;  // big-endian values
        uint64_t nelems;
    };
    auto readAttrs = [&](std::vector<Attr>& out) {
        const uint32_t n = listHeader(TAG_ATTRIBUTE);
        for (uint32_t i = 0; i < n; i++) {
            Attr a;
            a.name = c.name();
            a.type = int(c.u32());
            a.nelems = c.u32();
            const size_t ts_ = typeSize(a.type);
            if (!ts_) error("attribute \"" + a.name + "\" has an unknown type");
            const uint64_t nbytes = a.nelems * ts_;
            if (nbytes > (uint64_t(1) << 28)) error("implausible attribute size");
            a.raw.resize(size_t(nbytes));
            c.bytes(a.raw.data(), size_t(nbytes));
            c.skipPadding(nbytes);
            out.push_back(std::move(a));
        }
    };

    const uint32_t ndims = listHeader(TAG_DIMENSION);
    for (uint32_t i = 0; i < ndims; i++) {
        Dim d;
        d.name = c.name();
        d.length = c.u32();
        dims.push_back(d);
    }
    std::vector<Attr> globalAttrs;
    readAttrs(globalAttrs);
    const uint32_t nvars = listHeader(TAG_VARIABLE);
    for (uint32_t i = 0; i < nvars; i++) {
        Var v;
        v.name = c.name();
        const uint32_t rank = c.u32();
        if (rank > 1024) error("implausible variable rank");
        for (uint32_t r = 0; r < rank; r++) {
            const uint32_t id = c.u32();
            if (id >= dims.size()) error("variable \"" + v.name + "\" refers to an unknown dimension");
            v.dimids.push_back(int(id));
        }
        std::vector<Attr> attrs;
        readAttrs(attrs);
        for (const Attr& a : attrs) {
            if (a.name == "standard_name" && a.type == NC_CHAR) {
                v.standardName.assign(reinterpret_cast<const char*>(a.raw.data()), a.raw.size());
                while (!v.standardName.empty() && v.standardName.back() == '\0') v.standardName.pop_back();
            } else if ((a.name == "missing_value" || a.name == "_FillValue") && a.nelems >= 1) {
                if (a.type == NC_FLOAT) {
                    v.hasFill = true;
                    v.fillValue = beFloat(a.raw.data());
                } else if (a.type == NC_DOUBLE) {
                    v.hasFill = true;
                    v.fillValue = float(beDouble(a.raw.data()));
                }
            }
        }
        v.type = int(c.u32());
        v.vsize = c.u32();
        v.begin = version == 1 ? uint64_t(c.u32()) : c.u64();
        v.isRecord = !v.dimids.empty() && dims[size_t(v.dimids[0])].length == 0;
        vars.push_back(std::move(v));
    }

    // record layout: records of all record variables are interleaved; the padding of vsize is omitted when there is
    // exactly one record variable (classic format specification, "Note on padding")
    int numRecordVars = 0;
    for (const Var& v : vars) numRecordVars += v.isRecord ? 1 : 0;
    for (const Var& v : vars) {
        if (!v.isRecord) continue;
        uint64_t slab = typeSize(v.type);
        for (size_t d = 1; d < v.dimids.size(); d++) slab *= dims[size_t(v.dimids[d])].length;
        recordSize += numRecordVars == 1 ? slab : ((slab + 3) & ~uint64_t(3));
    }
    if (recs == 0xFFFFFFFFu && recordSize > 0) {
        uint64_t firstBegin = std::numeric_limits<uint64_t>::max();
        for (const Var& v : vars)
            if (v.isRecord) firstBegin = std::min(firstBegin, v.begin);
        std::fseek(file, 0, SEEK_END);
        const uint64_t size = uint64_t(std::ftell(file));
        numRecs = size > firstBegin ? (size - firstBegin) / recordSize : 0;
    }
}

// Common to both back ends: the grid from the first float/double variable with (z, y, x) trailing dimensions, the
// member / time axis, the field list.
void NetCdfLoader::deriveGridAndFields() {
    int representative = -1;
    for (size_t i = 0; i < vars.size() && representative < 0; i++) {
        const Var& v = vars[i];
        if ((v.type != NC_FLOAT && v.type != NC_DOUBLE) || (v.dimids.size() != 3 && v.dimids.size() != 4)) continue;
        representative = int(i);
    }
    if (representative < 0) error("no 3-D or 4-D floating-point variable found");
    {
        const Var& v = vars[size_t(representative)];
        const size_t r = v.dimids.size();
        zs = int(dimLength(v.dimids[r - 3]));
        ys = int(dimLength(v.dimids[r - 2]));
        xs = int(dimLength(v.dimids[r - 1]));
        const std::string &nz = dims[size_t(v.dimids[r - 3])].name, &ny = dims[size_t(v.dimids[r - 2])].name,
                          &nx = dims[size_t(v.dimids[r - 1])].name;
        if (!isOneOf(nz, {"z", "zs", "lev"}) || !isOneOf(ny, {"y", "ys", "lat"}) || !isOneOf(nx, {"x", "xs", "lon"}))
            warnings.push_back("dimensions (" + nz + ", " + ny + ", " + nx + ") taken positionally as (z, y, x)");
        if (r == 4) {
            const std::string& lead = dims[size_t(v.dimids[0])].name;
            const int n = int(dimLength(v.dimids[0]));
            if (lead == "time") {
                ts = n;
            } else if (isOneOf(lead, {"ensemble", "member", "members"})) {
                es = n;
            } else {
                warnings.push_back("Warning in NetCdfLoader::setInputFiles: Unknown dimension name. Assuming time.");
                ts = n;
            }
        }
    }
    if (xs <= 0 || ys <= 0 || zs <= 0) error("empty grid");

    for (size_t i = 0; i < vars.size(); i++) {
        const Var& v = vars[i];
        if ((v.type != NC_FLOAT && v.type != NC_DOUBLE) || (v.dimids.size() != 3 && v.dimids.size() != 4)) continue;
        const size_t r = v.dimids.size();
        if (int(dimLength(v.dimids[r - 3])) != zs || int(dimLength(v.dimids[r - 2])) != ys ||
            int(dimLength(v.dimids[r - 1])) != xs)
            continue;
        if (r == 4 && int(dimLength(v.dimids[0])) != (ts > 1 ? ts : es) && !(ts == 1 && es == 1)) continue;
        Field f{v.standardName.empty() ? v.name : v.standardName, int(i), v.hasFill,
                v.hasFill ? v.fillValue : std::numeric_limits<float>::quiet_NaN()};
        fields.push_back(f);
        fieldNames.push_back(f.name);
    }
}

// one xs*ys*zs slab: the whole variable (rank 3) or index `leadingIndex` of its first dimension (rank 4)
void NetCdfLoader::readSlab(const Var& v, uint64_t leadingIndex, float* out) const {
    const uint64_t n = uint64_t(xs) * uint64_t(ys) * uint64_t(zs);
    if (hdf5) {
        const Hdf5Dataset& ds = hdf5->datasets().at(size_t(v.varid));
        const bool lead = v.dimids.size() == 4;
        if (lead && leadingIndex >= dimLength(v.dimids[0])) error("index outside the leading dimension of \"" + v.name + "\"");
        std::vector<uint64_t> start(ds.shape.size(), 0), count(ds.shape);
        if (lead) {
            start[0] = leadingIndex;
            count[0] = 1;
        }
        try {
            hdf5->readFloats(ds, start, count, out);
        } catch (const Hdf5Error& e) {
            error(std::string("reading \"") + v.name + "\" failed: " + e.what());
        }
        return;
    }
    if (library) {  // netcdf-c converts NC_DOUBLE to float itself (the reference's loadFloatArray3D/4D, :93-120)
        const bool lead = v.dimids.size() == 4;
        if (lead && leadingIndex >= dimLength(v.dimids[0])) error("index outside the leading dimension of \"" + v.name + "\"");
        const size_t start4[4] = {size_t(leadingIndex), 0, 0, 0}, count4[4] = {1, size_t(zs), size_t(ys), size_t(xs)};
        const int status = library->get_vara_float(ncid, v.varid, lead ? start4 : start4 + 1, lead ? count4 : count4 + 1, out);
        if (status != 0)
            error("nc_get_vara_float failed on \"" + v.name + "\": " +
                  (library->strerror ? library->strerror(status) : "netcdf error " + std::to_string(status)));
        return;
    }
    const size_t es_ = typeSize(v.type);
    uint64_t offset = v.begin;
    if (v.dimids.size() == 4) {
        if (leadingIndex >= dimLength(v.dimids[0])) error("index outside the leading dimension of \"" + v.name + "\"");
        offset += v.isRecord ? leadingIndex * recordSize : leadingIndex * n * es_;
    } else if (v.isRecord) {
        error("3-D record variables (an UNLIMITED z axis) are not supported");
    }
    std::vector<unsigned char> raw(size_t(n * es_));
#if defined(_WIN32)
    if (_fseeki64(file, (long long)offset, SEEK_SET) != 0) error("seek failed");
#else
    if (fseeko(file, off_t(offset), SEEK_SET) != 0) error("seek failed");
#endif
    if (std::fread(raw.data(), 1, raw.size(), file) != raw.size()) error("variable \"" + v.name + "\" is truncated");
    if (v.type == NC_FLOAT) {
        for (uint64_t i = 0; i < n; i++) out[i] = beFloat(raw.data() + 4 * i);
    } else {
        for (uint64_t i = 0; i < n; i++) out[i] = float(beDouble(raw.data() + 8 * i));
    }
}

void NetCdfLoader::getFieldEntry(const std::string& fieldName, int timeStepIdx, int memberIdx, float* out) const {
    const Field* field = nullptr;
    for (const Field& f : fields)
        if (f.name == fieldName) field = &f;
    if (!field) error("Error in NetCdfLoader::getFieldEntry: Unknown field name \"" + fieldName + "\"");
    const Var& v = vars[size_t(field->var)];
    uint64_t lead = 0;
    if (v.dimids.size() == 4) lead = uint64_t(ts > 1 ? timeStepIdx : (es > 1 ? memberIdx : 0));  // :869-875
    readSlab(v, lead, out);
    if (field->hasFill) {
        const size_t n = size_t(xs) * size_t(ys) * size_t(zs);
        for (size_t i = 0; i < n; i++)
            if (out[i] == field->fillValue) out[i] = std::numeric_limits<float>::quiet_NaN();
    }
}

std::shared_ptr<VolumeData> NetCdfLoader::createVolumeData() const {
    auto vol = std::make_shared<VolumeData>(xs, ys, zs, ts, es);
    std::vector<float> buffer(size_t(xs) * size_t(ys) * size_t(zs));
    for (const std::string& name : fieldNames)
        for (int t = 0; t < ts; t++)
            for (int e = 0; e < es; e++) {
                getFieldEntry(name, t, e, buffer.data());
                vol->setFieldData(name, t, e, buffer.data());
            }
    return vol;
}

}  // namespace crfhost
