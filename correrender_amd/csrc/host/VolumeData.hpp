// VolumeData.hpp -- minimal host-side stand-in for the reference's field registry
// (/root/reference/src/Volume/VolumeData.hpp:95-483, VolumeData.cpp), reproducing only the contract the correlation
// calculators depend on:
//   grid / ensemble getters                       VolumeData.hpp:167-171
//   getFieldEntryCpu(type, name, t, e)            VolumeData.cpp:1202-1226: input fields come from storage; a field
//       whose name belongs to a host calculator is produced by allocating `new float[xs*ys*zs]`, calling
//       calc->calculateCpu(t, e, buffer) and wrapping the buffer in a HostCacheEntry that later delete[]s it
//       (Cache/HostCacheEntry.hpp:39-50) -- the callee must neither retain nor free the buffer;
//   getMinMaxScalarFieldValue(name, t, e)         VolumeData.cpp:1632-1670 (per (field,t,e) extrema, cached);
//       -- symmetrised to +-max|.| for a field flagged divergent (getIsScalarFieldDivergent, :616-621, :1661-1666);
//   getBoundingBoxRendering()                     VolumeData.hpp:205, computed by setGridExtent (VolumeData.cpp:322-330):
//       the world-space box picking positions are expressed in (ICorrelationCalculator::setReferencePointFromWorld);
//   addCalculator                                  VolumeData.cpp:1046-1086 (initialize, id, setVolumeData(this,true),
//       registered under getOutputFieldName()).
// No loaders, no device caches, no rendering: inputs are handed in as arrays.
#pragma once
#include <array>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "Calculator.hpp"

namespace crfhost {

class HostCacheEntryType {
public:
    HostCacheEntryType(size_t numEntries, float* dataOwned) : numEntries(numEntries), dataFloat(dataOwned) {}
    ~HostCacheEntryType() { delete[] dataFloat; }
    HostCacheEntryType(const HostCacheEntryType&) = delete;
    HostCacheEntryType& operator=(const HostCacheEntryType&) = delete;
    template <class T>
    const T* data() const { return reinterpret_cast<const T*>(dataFloat); }
    template <class T>
    T dataAt(size_t idx) const { return T(dataFloat[idx]); }
    size_t getNumEntries() const { return numEntries; }

private:
    size_t numEntries;
    float* dataFloat;
};
typedef std::shared_ptr<HostCacheEntryType> HostCacheEntry;

/// Axis-aligned box with float corners (the two members of sgl::AABB3 the path reads).
struct AABB3 {
    std::array<float, 3> min{0.f, 0.f, 0.f}, max{0.f, 0.f, 0.f};
};

class VolumeData {
public:
    VolumeData(int xs, int ys, int zs, int ts, int es) : xs(xs), ys(ys), zs(zs), ts(ts), es(es) { setGridExtent(1.f, 1.f, 1.f); }

    /// Cell spacing -> box = [0, (n-1) d] and the rendering box: the box's dimensions divided by the largest one,
    /// times -+0.25 (VolumeData::setGridExtent, VolumeData.cpp:322-330).
    void setGridExtent(float dx, float dy, float dz);
    const AABB3& getBoundingBoxRendering() const { return boxRendering; }
    /// VolumeData.cpp:616-621: only the field named "Helicity" is centred at zero.
    bool getIsScalarFieldDivergent(const std::string& fieldName) const { return fieldName == "Helicity"; }

    int getGridSizeX() const { return xs; }
    int getGridSizeY() const { return ys; }
    int getGridSizeZ() const { return zs; }
    int getTimeStepCount() const { return ts; }
    int getEnsembleMemberCount() const { return es; }
    size_t getSlice3dEntryCount() const { return size_t(xs) * size_t(ys) * size_t(zs); }
    int getStandardScalarFieldIdx() const { return 0; }

    /// Registers the data of one (field, time step, ensemble member): xs*ys*zs floats, copied.
    void setFieldData(const std::string& fieldName, int timeStepIdx, int ensembleIdx, const float* values);
    std::vector<std::string> getFieldNames(FieldType) const;

    HostCacheEntry getFieldEntryCpu(FieldType fieldType, const std::string& fieldName, int timeStepIdx = -1,
                                    int ensembleIdx = -1);
    std::pair<float, float> getMinMaxScalarFieldValue(const std::string& fieldName, int timeStepIdx = -1,
                                                      int ensembleIdx = -1);
    void addCalculator(const CalculatorPtr& calculator);
    /// Evicts the cached outputs of dirty calculators (the role of VolumeData::renderGuiCalculators, :1852-1936).
    void updateCalculators();
    size_t getNewCalculatorUseCount(CalculatorType t) { return ++calculatorTypeUseCounts[t]; }  // VolumeData.cpp:2283-2285
    /// Monotonic id of the input data (changes whenever setFieldData is called): lets a calculator keep a device copy.
    uint64_t getDataGeneration() const { return dataGeneration; }

private:
    typedef std::tuple<std::string, int, int> Access;
    int xs, ys, zs, ts, es;
    AABB3 box, boxRendering;
    std::vector<std::string> fieldNames;
    std::map<Access, HostCacheEntry> storage;         // input fields
    std::map<Access, HostCacheEntry> hostFieldCache;  // calculator outputs
    std::map<Access, std::pair<float, float>> fieldMinMaxCache;
    std::map<std::string, CalculatorPtr> calculatorsHost;
    std::vector<CalculatorPtr> calculators;
    std::map<CalculatorType, size_t> calculatorTypeUseCounts;
    uint64_t dataGeneration = 0;
};

}  // namespace crfhost
