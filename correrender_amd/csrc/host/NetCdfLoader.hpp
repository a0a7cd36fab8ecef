// NetCdfLoader.hpp -- reader for the on-disk input format of the correlation path: NetCDF
// files holding `data(member, lev, lat, lon)` float32 volumes, as written by the reference's generator
// (scripts/generate_synth_box_ensembles.py:151-158) and read by its NetCdfLoader
// (src/Loaders/NetCdfLoader.cpp:286-560 setInputFiles, :826-935 getFieldEntry).
//
// Three back ends behind one metadata model:
//   * the CLASSIC file formats are parsed directly, without any library: CDF-1 (32-bit offsets) and CDF-2 (64-bit
//     offsets), fixed-size and record (UNLIMITED first dimension) variables, NC_FLOAT / NC_DOUBLE data;
//   * NetCDF-4 files (HDF5 containers -- what the generator's `format='NETCDF4_CLASSIC'` writes and what the paper's data
//     sets are) are decoded by Hdf5Reader (this directory: superblocks 0-3, old- and new-style groups, contiguous /
//     chunked / compressed float and double variables, dimension scales), with no library at all; pinned against files
//     written by the real libhdf5 (tests/golden/netcdf4/);
//   * the netcdf-c library the reference itself links (nc_open ... nc_get_vara_float, NetCdfLoader.cpp:282-339,
//     826-935), loaded at RUN TIME with dlopen, reads CDF-5 and is the fallback for whatever HDF5 feature the decoder
//     does not cover; it is used FIRST when CRF_LIBNETCDF names it (or CRF_NETCDF_BACKEND=library).  Library name:
//     $CRF_LIBNETCDF, else libnetcdf.so[.19|.18|.15|.13|.11|.7].  This image has no libnetcdf, so that back end is
//     exercised against a test double of the library's C API (tests/fake_libnetcdf.c) -- see INTEGRATION.md.
//
// Conventions kept from the reference loader:
//   * the grid is taken from the first floating-point variable with 3 or 4 dimensions whose trailing dimensions are
//     (z, y, x); accepted names z|zs|lev, y|ys|lat, x|xs|lon (NetCdfLoader.cpp:323-328) -- any other name is accepted
//     positionally, like the reference's (time, level, rlat, rlon) examples (:405-409);
//   * a leading 4th dimension named "time" is the time axis, "ensemble" / "member" / "members" the ensemble axis,
//     anything else is assumed to be time with a warning (:493-507);
//   * fields = every float/double variable whose trailing three dimension LENGTHS match the grid (:741-749); the field
//     name is the `standard_name` attribute when present, else the variable name (:753-757);
//   * values equal to `missing_value` / `_FillValue` become NaN (:758-765, :886-893);
//   * NC_DOUBLE data is converted to float (:93-99).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "Hdf5Reader.hpp"

namespace crfhost {

class VolumeData;

class NetCdfLoader {
public:
    /// Parses the header.  Throws CalculatorError (Calculator.hpp) on a malformed or unsupported file.
    explicit NetCdfLoader(const std::string& filePath);
    ~NetCdfLoader();

    int getGridSizeX() const { return xs; }
    int getGridSizeY() const { return ys; }
    int getGridSizeZ() const { return zs; }
    int getTimeStepCount() const { return ts; }
    int getEnsembleMemberCount() const { return es; }
    const std::vector<std::string>& getFieldNames() const { return fieldNames; }
    const std::vector<std::string>& getWarnings() const { return warnings; }

    /// xs*ys*zs floats of one (field, time step, member) in IDXS order; fill values -> NaN.  NetCdfLoader::getFieldEntry.
    void getFieldEntry(const std::string& fieldName, int timeStepIdx, int memberIdx, float* out) const;

    /// Builds a VolumeData holding every field of the file (all time steps and members).
    std::shared_ptr<VolumeData> createVolumeData() const;

private:
    struct Dim {
        std::string name;
        uint64_t length;  // 0 = the record (UNLIMITED) dimension
    };
    struct Var {
        std::string name;
        std::vector<int> dimids;
        int type = 0;
        // the attributes the loader acts on, extracted by whichever back end read the header
        std::string standardName;
        bool hasFill = false;
        float fillValue = 0.0f;
        // classic files: where the data lies; library back end: the variable id
        uint64_t vsize = 0, begin = 0;
        bool isRecord = false;
        int varid = -1;
    };
    struct Field {
        std::string name;
        int var;
        bool hasFill;
        float fillValue;
    };

    std::unique_ptr<Hdf5File> hdf5;  // non-null: the file is read by the built-in HDF5 decoder (Var::varid = dataset index)
    struct Library;  // the netcdf-c entry points resolved with dlsym (NetCdfLoader.cpp)
    std::string path;
    mutable FILE* file = nullptr;
    std::unique_ptr<Library> library;  // non-null: the file is open through netcdf-c (ncid below)
    int ncid = -1;
    int version = 1;
    uint64_t numRecs = 0, recordSize = 0;
    std::vector<Dim> dims;
    std::vector<Var> vars;
    std::vector<Field> fields;
    std::vector<std::string> fieldNames;
    std::vector<std::string> warnings;
    int xs = 0, ys = 0, zs = 0, ts = 1, es = 1;

    void parseClassicHeader();
    void openWithLibrary(const char* why);
    void openWithHdf5Reader();
    void deriveGridAndFields();
    uint64_t dimLength(int dimid) const;
    void readSlab(const Var& v, uint64_t leadingIndex, float* out) const;
    [[noreturn]] void error(const std::string& msg) const;
};

}  // namespace crfhost
