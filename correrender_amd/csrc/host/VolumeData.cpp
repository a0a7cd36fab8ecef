#include "VolumeData.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace crfhost {

void VolumeData::setGridExtent(float dx, float dy, float dz) {
    box.min = {0.0f, 0.0f, 0.0f};
    box.max = {float(xs - 1) * dx, float(ys - 1) * dy, float(zs - 1) * dz};
    const float maxDimension = std::max(box.max[0], std::max(box.max[1], box.max[2]));
    for (int i = 0; i < 3; i++) {
        const float normalizedDimension = box.max[i] / maxDimension;
        boxRendering.min[i] = -normalizedDimension * 0.25f;
        boxRendering.max[i] = normalizedDimension * 0.25f;
    }
}

void VolumeData::setFieldData(const std::string& fieldName, int timeStepIdx, int ensembleIdx, const float* values) {
    const size_t n = getSlice3dEntryCount();
    auto* copy = new float[n];
    std::memcpy(copy, values, n * sizeof(float));
    storage[Access(fieldName, timeStepIdx, ensembleIdx)] = std::make_shared<HostCacheEntryType>(n, copy);
    if (std::find(fieldNames.begin(), fieldNames.end(), fieldName) == fieldNames.end()) fieldNames.push_back(fieldName);
    fieldMinMaxCache.clear();
    hostFieldCache.clear();
    dataGeneration++;
}

std::vector<std::string> VolumeData::getFieldNames(FieldType) const {
    std::vector<std::string> names = fieldNames;
    for (const auto& c : calculators) names.push_back(c->getOutputFieldName());
    return names;
}

HostCacheEntry VolumeData::getFieldEntryCpu(FieldType, const std::string& fieldName, int timeStepIdx, int ensembleIdx) {
    if (timeStepIdx < 0) timeStepIdx = 0;
    if (ensembleIdx < 0) ensembleIdx = 0;
    const Access access(fieldName, timeStepIdx, ensembleIdx);
    auto itCalc = calculatorsHost.find(fieldName);
    if (itCalc == calculatorsHost.end()) {
        auto it = storage.find(access);
        if (it == storage.end())
            throw CalculatorError("Error in VolumeData::getFieldEntryCpu: Trying to access field '" + fieldName +
                                  "' that is not available.");
        return it->second;
    }
    auto itCache = hostFieldCache.find(access);
    if (itCache != hostFieldCache.end()) return itCache->second;
    const size_t numEntries = getSlice3dEntryCount();
    auto* buffer = new float[numEntries];  // VolumeData.cpp:1222
    try {
        itCalc->second->calculateCpu(timeStepIdx, ensembleIdx, buffer);
    } catch (...) {
        delete[] buffer;
        throw;
    }
    HostCacheEntry entry = std::make_shared<HostCacheEntryType>(numEntries, buffer);  // takes ownership, :1226
    hostFieldCache[access] = entry;
    return entry;
}

std::pair<float, float> VolumeData::getMinMaxScalarFieldValue(const std::string& fieldName, int timeStepIdx,
                                                              int ensembleIdx) {
    if (timeStepIdx < 0) timeStepIdx = 0;
    if (ensembleIdx < 0) ensembleIdx = 0;
    auto itCalc = calculatorsHost.find(fieldName);
    if (itCalc != calculatorsHost.end() && itCalc->second->getHasFixedRange()) return itCalc->second->getFixedRange();
    const Access access(fieldName, timeStepIdx, ensembleIdx);
    auto it = fieldMinMaxCache.find(access);
    if (it != fieldMinMaxCache.end()) return it->second;
    HostCacheEntry entry = getFieldEntryCpu(FieldType::SCALAR, fieldName, timeStepIdx, ensembleIdx);
    const float* v = entry->data<float>();
    float mn = std::numeric_limits<float>::max(), mx = std::numeric_limits<float>::lowest();
    for (size_t i = 0; i < entry->getNumEntries(); i++) {
        if (v[i] < mn) mn = v[i];
        if (v[i] > mx) mx = v[i];
    }
    // Is this a divergent scalar field? If yes, the range is centred at zero (VolumeData.cpp:1661-1666).
    if (getIsScalarFieldDivergent(fieldName)) {
        const float maxAbs = std::max(std::abs(mn), std::abs(mx));
        mn = -maxAbs;
        mx = maxAbs;
    }
    fieldMinMaxCache[access] = {mn, mx};
    return {mn, mx};
}

void VolumeData::addCalculator(const CalculatorPtr& calculator) {
    calculator->initialize();
    calculator->setCalculatorId(calculators.size());
    calculator->setVolumeData(this, true);
    calculators.push_back(calculator);
    // Every calculator of this stand-in fulfils calculateCpu (a HIP backend has no Vulkan image to fill, so it is
    // dispatched like FilterDevice::CPU calculators are: VolumeData.cpp:1055-1059,1214-1226).
    calculatorsHost[calculator->getOutputFieldName()] = calculator;
}

void VolumeData::updateCalculators() {
    for (auto& c : calculators) {
        const bool nameChanged = c->getHasNameChanged();
        if (c->getIsDirty() || nameChanged) {
            for (auto it = calculatorsHost.begin(); it != calculatorsHost.end();)
                it = (it->second == c) ? calculatorsHost.erase(it) : std::next(it);
            calculatorsHost[c->getOutputFieldName()] = c;
            hostFieldCache.clear();
        }
    }
}

}  // namespace crfhost
