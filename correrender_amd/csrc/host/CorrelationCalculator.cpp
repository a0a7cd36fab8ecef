#include "CorrelationCalculator.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <sstream>

namespace crfhost {

const char* const CORRELATION_MEASURE_TYPE_NAMES[7] = {
        "Pearson", "Spearman", "Kendall", "Mutual Information (Binned)", "Mutual Information (Kraskov)",
        "Binned MI Correlation Coefficient", "KMI Correlation Coefficient"};
const char* const CORRELATION_MEASURE_TYPE_IDS[7] = {
        "pearson", "spearman", "kendall", "mi_binned", "mi_kraskov",
        "binned_mi_correlation_coefficient", "kmi_correlation_coefficient"};
const char* const CORRELATION_MODE_NAMES[2] = {"Ensemble", "Time"};
const char* const CORRELATION_FIELD_MODE_NAMES[3] = {"Single", "Separate", "Separate Symmetric"};

static inline int iceil(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------------------
// ICorrelationCalculator (CorrelationCalculator.cpp:78-217, 397-565)
// ---------------------------------------------------------------------------------------------------------
void ICorrelationCalculator::setVolumeData(VolumeData* _volumeData, bool isNewData) {
    Calculator::setVolumeData(_volumeData, isNewData);
    const int es = _volumeData->getEnsembleMemberCount();
    const int ts = _volumeData->getTimeStepCount();
    if (isEnsembleMode && es <= 1 && ts > 1) {
        isEnsembleMode = false;
    } else if (!isEnsembleMode && ts <= 1 && es > 1) {
        isEnsembleMode = true;
    }
    scalarFieldNames.clear();
    for (const std::string& name : volumeData->getFieldNames(FieldType::SCALAR))
        if (name != getOutputFieldName()) scalarFieldNames.push_back(name);
    if (isNewData) {
        referencePointIndex = {volumeData->getGridSizeX() / 2, volumeData->getGridSizeY() / 2,
                               volumeData->getGridSizeZ() / 2};
        fieldIndex = fieldIndex2 = volumeData->getStandardScalarFieldIdx();
        fieldIndexGui = fieldIndex2Gui = volumeData->getStandardScalarFieldIdx();
    }
}

int ICorrelationCalculator::getCorrelationMemberCount() const {
    return isEnsembleMode ? volumeData->getEnsembleMemberCount() : volumeData->getTimeStepCount();
}

HostCacheEntry ICorrelationCalculator::getFieldEntryCpu(const std::string& fieldName, int fieldIdx, int timeStepIdx,
                                                        int ensembleIdx) {
    return volumeData->getFieldEntryCpu(FieldType::SCALAR, fieldName, isEnsembleMode ? timeStepIdx : fieldIdx,
                                        isEnsembleMode ? fieldIdx : ensembleIdx);
}

std::pair<float, float> ICorrelationCalculator::getMinMaxScalarFieldValue(const std::string& fieldName, int fieldIdx,
                                                                          int timeStepIdx, int ensembleIdx) {
    return volumeData->getMinMaxScalarFieldValue(fieldName, isEnsembleMode ? timeStepIdx : fieldIdx,
                                                 isEnsembleMode ? fieldIdx : ensembleIdx);
}

void ICorrelationCalculator::setReferencePoint(const std::array<int, 3>& referencePoint) {
    if (referencePointIndex != referencePoint) {
        const std::array<int, 3> maxCoord{volumeData->getGridSizeX() - 1, volumeData->getGridSizeY() - 1,
                                          volumeData->getGridSizeZ() - 1};
        for (int i = 0; i < 3; i++) referencePointIndex[i] = std::clamp(referencePoint[i], 0, maxCoord[i]);
        dirty = true;
    }
}

void ICorrelationCalculator::setReferencePointFromWorld(const std::array<float, 3>& worldPosition) {
    const std::array<int, 3> maxCoord{volumeData->getGridSizeX() - 1, volumeData->getGridSizeY() - 1,
                                      volumeData->getGridSizeZ() - 1};
    const AABB3& gridAabb = volumeData->getBoundingBoxRendering();
    std::array<int, 3> referencePointNew{};
    for (int i = 0; i < 3; i++) {
        // float arithmetic in the reference's order: (p - min) / (max - min), times maxCoord, glm::round, int cast, clamp
        float position = (worldPosition[i] - gridAabb.min[i]) / (gridAabb.max[i] - gridAabb.min[i]);
        position *= float(maxCoord[i]);
        const float rounded = std::round(position);  // glm::round: half away from zero
        // a one-cell axis gives 0/0: the reference converts that NaN to int (undefined behaviour, INT_MIN on x86-64)
        // and clamps it to 0
        referencePointNew[i] = std::isnan(rounded) ? 0 : std::clamp(int(std::clamp(rounded, -2.0e9f, 2.0e9f)), 0, maxCoord[i]);
    }
    if (referencePointIndex != referencePointNew) {
        referencePointIndex = referencePointNew;
        dirty = true;
    }
}

// Settings keys of ICorrelationCalculator (the reference's keys and value spellings, CorrelationCalculator.cpp:397-565):
// the field mode comes first because it decides which field-index keys apply.
const SettingBinding<ICorrelationCalculator> ICorrelationCalculator::kSettings[] = {
    {"correlation_field_mode",
     [](ICorrelationCalculator& c, const SettingsMap& m) {
         std::string name;
         if (!m.getValueOpt("correlation_field_mode", name)) return false;
         const auto* end = CORRELATION_FIELD_MODE_NAMES + 3;
         const auto* hit = std::find(CORRELATION_FIELD_MODE_NAMES, end, name);
         if (hit != end) c.correlationFieldMode = CorrelationFieldMode(hit - CORRELATION_FIELD_MODE_NAMES);
         return true;
     },
     [](const ICorrelationCalculator& c, SettingsMap& m) {
         m.addKeyValue("correlation_field_mode", CORRELATION_FIELD_MODE_NAMES[int(c.correlationFieldMode)]);
     }},
    {"scalar_field_idx_ref",
     [](ICorrelationCalculator& c, const SettingsMap& m) {
         if (c.correlationFieldMode == CorrelationFieldMode::SINGLE || !m.getValueOpt("scalar_field_idx_ref", c.fieldIndex2Gui))
             return false;
         c.fieldIndex2 = c.fieldIndex2Gui;
         return true;
     },
     [](const ICorrelationCalculator& c, SettingsMap& m) {
         if (c.correlationFieldMode != CorrelationFieldMode::SINGLE) m.addKeyValue("scalar_field_idx_ref", c.fieldIndex2Gui);
     }},
    {"scalar_field_idx_query",
     [](ICorrelationCalculator& c, const SettingsMap& m) {
         if (c.correlationFieldMode == CorrelationFieldMode::SINGLE || !m.getValueOpt("scalar_field_idx_query", c.fieldIndexGui))
             return false;
         c.fieldIndex = c.fieldIndexGui;
         return true;
     },
     [](const ICorrelationCalculator& c, SettingsMap& m) {
         if (c.correlationFieldMode != CorrelationFieldMode::SINGLE) m.addKeyValue("scalar_field_idx_query", c.fieldIndexGui);
     }},
    {"scalar_field_idx",
     [](ICorrelationCalculator& c, const SettingsMap& m) {
         if (c.correlationFieldMode != CorrelationFieldMode::SINGLE || !m.getValueOpt("scalar_field_idx", c.fieldIndexGui))
             return false;
         c.fieldIndex = c.fieldIndexGui;
         return true;
     },
     [](const ICorrelationCalculator& c, SettingsMap& m) {
         if (c.correlationFieldMode == CorrelationFieldMode::SINGLE) m.addKeyValue("scalar_field_idx", c.fieldIndexGui);
     }},
    {"correlation_mode",
     [](ICorrelationCalculator& c, const SettingsMap& m) {
         std::string name;
         if (!m.getValueOpt("correlation_mode", name)) return false;
         c.isEnsembleMode = name == CORRELATION_MODE_NAMES[0];
         c.onCorrelationMemberCountChanged();
         return true;
     },
     [](const ICorrelationCalculator& c, SettingsMap& m) {
         m.addKeyValue("correlation_mode", CORRELATION_MODE_NAMES[c.isEnsembleMode ? 0 : 1]);
     }},
    {"reference_point_x", [](ICorrelationCalculator& c, const SettingsMap& m) { return m.getValueOpt("reference_point_x", c.referencePointIndex[0]); },
     [](const ICorrelationCalculator& c, SettingsMap& m) { m.addKeyValue("reference_point_x", c.referencePointIndex[0]); }},
    {"reference_point_y", [](ICorrelationCalculator& c, const SettingsMap& m) { return m.getValueOpt("reference_point_y", c.referencePointIndex[1]); },
     [](const ICorrelationCalculator& c, SettingsMap& m) { m.addKeyValue("reference_point_y", c.referencePointIndex[1]); }},
    {"reference_point_z", [](ICorrelationCalculator& c, const SettingsMap& m) { return m.getValueOpt("reference_point_z", c.referencePointIndex[2]); },
     [](const ICorrelationCalculator& c, SettingsMap& m) { m.addKeyValue("reference_point_z", c.referencePointIndex[2]); }},
    // the only data mode of this backend; written for state files, never read
    {"data_mode", [](ICorrelationCalculator&, const SettingsMap&) { return false; },
     [](const ICorrelationCalculator&, SettingsMap& m) { m.addKeyValue("data_mode", "Buffer Array"); }},
    {"use_buffer_tiling", [](ICorrelationCalculator& c, const SettingsMap& m) { return m.getValueOpt("use_buffer_tiling", c.useBufferTiling); },
     [](const ICorrelationCalculator& c, SettingsMap& m) { m.addKeyValue("use_buffer_tiling", c.useBufferTiling); }},
    {"use_time_lag_correlations",
     [](ICorrelationCalculator& c, const SettingsMap& m) { return m.getValueOpt("use_time_lag_correlations", c.useTimeLagCorrelations); },
     [](const ICorrelationCalculator& c, SettingsMap& m) { m.addKeyValue("use_time_lag_correlations", c.useTimeLagCorrelations); }},
    {"time_lag_time_step_idx",
     [](ICorrelationCalculator& c, const SettingsMap& m) { return m.getValueOpt("time_lag_time_step_idx", c.timeLagTimeStepIdx); },
     [](const ICorrelationCalculator& c, SettingsMap& m) { m.addKeyValue("time_lag_time_step_idx", c.timeLagTimeStepIdx); }},
};

void ICorrelationCalculator::setSettings(const SettingsMap& settings) {
    if (loadSettings(*this, kSettings, settings)) dirty = true;
}

void ICorrelationCalculator::getSettings(SettingsMap& settings) { storeSettings(*this, kSettings, settings); }

// ---------------------------------------------------------------------------------------------------------
// CorrelationCalculator (CorrelationCalculator.cpp:569-779)
// ---------------------------------------------------------------------------------------------------------
CorrelationCalculator::CorrelationCalculator(int device) : device(device), devices{device} {}

CorrelationCalculator::~CorrelationCalculator() { releaseBackend(); }

void CorrelationCalculator::throwBackendError(const char* where) {
    throw CalculatorError(std::string("Error in CorrelationCalculator::") + where + ": " +
                          (group ? crf_group_last_error(group) : crf_last_error(ctx)));
}

std::string CorrelationCalculator::getOutputFieldName() {
    std::string outputFieldName = CORRELATION_MEASURE_TYPE_NAMES[int(correlationMeasureType)];
    if (int(correlationMeasureType) <= int(CorrelationMeasureType::KENDALL)) outputFieldName += " Correlation";
    if (calculatorConstructorUseCount > 1) outputFieldName += " (" + std::to_string(calculatorConstructorUseCount) + ")";
    return outputFieldName;
}

void CorrelationCalculator::setVolumeData(VolumeData* _volumeData, bool isNewData) {
    ICorrelationCalculator::setVolumeData(_volumeData, isNewData);
    if (isNewData) calculatorConstructorUseCount = volumeData->getNewCalculatorUseCount(CalculatorType::CORRELATION);
    if (isNewData || cachedMemberCount != getCorrelationMemberCount()) onCorrelationMemberCountChanged();
}

void CorrelationCalculator::onCorrelationMemberCountChanged() {
    const int cs = getCorrelationMemberCount();
    k = std::max(iceil(3 * cs, 100), 1);
    kMax = std::max(iceil(7 * cs, 100), 20);
    cachedMemberCount = cs;
}

bool CorrelationCalculator::getIsRealtime() const {
    if (!useGpu) return false;
    return correlationMeasureType == CorrelationMeasureType::PEARSON || getCorrelationMemberCount() < 200;
}

bool CorrelationCalculator::getHasFixedRange() const {
    return correlationMeasureType != CorrelationMeasureType::MUTUAL_INFORMATION_BINNED &&
           correlationMeasureType != CorrelationMeasureType::MUTUAL_INFORMATION_KRASKOV;
}

std::pair<float, float> CorrelationCalculator::getFixedRange() const {
    if (getHasFixedRange()) {
        if (calculateAbsoluteValue || isMeasureCorrelationCoefficientMI(correlationMeasureType)) return {0.0f, 1.0f};
        return {-1.0f, 1.0f};
    }
    return {0.0f, 1.0f};
}

// Settings keys of CorrelationCalculator (CorrelationCalculator.cpp:706-779), plus "devices": this backend's own key,
// a comma-separated list of HIP device ordinals -- more than one spreads the grid over a crf_group (z-slabs, one per
// device); written back only when it is not the single default device, so reference state files round-trip unchanged.
const SettingBinding<CorrelationCalculator> CorrelationCalculator::kSettings[] = {
    {"correlation_measure_type",
     [](CorrelationCalculator& c, const SettingsMap& m) {
         std::string id;
         if (!m.getValueOpt("correlation_measure_type", id)) return false;
         const auto* end = CORRELATION_MEASURE_TYPE_IDS + 7;
         const auto* hit = std::find_if(CORRELATION_MEASURE_TYPE_IDS, end, [&](const char* s) { return id == s; });
         if (hit != end) c.correlationMeasureType = CorrelationMeasureType(hit - CORRELATION_MEASURE_TYPE_IDS);
         c.hasNameChanged = true;
         return true;
     },
     [](const CorrelationCalculator& c, SettingsMap& m) {
         m.addKeyValue("correlation_measure_type", CORRELATION_MEASURE_TYPE_IDS[int(c.correlationMeasureType)]);
     }},
    // "CPU" | "Vulkan" | "CUDA"; anything else means the accelerator (CorrelationCalculator.cpp:721-747)
    {"device",
     [](CorrelationCalculator& c, const SettingsMap& m) {
         std::string name;
         if (!m.getValueOpt("device", name)) return false;
         const bool before = c.useGpu;
         c.useGpu = name != "CPU";
         c.useCuda = name == "CUDA";
         c.hasFilterDeviceChanged = before != c.useGpu;
         return true;
     },
     [](const CorrelationCalculator& c, SettingsMap& m) {
         m.addKeyValue("device", !c.useGpu ? "CPU" : (!c.useCuda ? "Vulkan" : "CUDA"));
     }},
    {"calculate_absolute_value",
     [](CorrelationCalculator& c, const SettingsMap& m) { return m.getValueOpt("calculate_absolute_value", c.calculateAbsoluteValue); },
     [](const CorrelationCalculator& c, SettingsMap& m) { m.addKeyValue("calculate_absolute_value", c.calculateAbsoluteValue); }},
    {"mi_bins", [](CorrelationCalculator& c, const SettingsMap& m) { return m.getValueOpt("mi_bins", c.numBins); },
     [](const CorrelationCalculator& c, SettingsMap& m) { m.addKeyValue("mi_bins", c.numBins); }},
    {"kmi_neighbors", [](CorrelationCalculator& c, const SettingsMap& m) { return m.getValueOpt("kmi_neighbors", c.k); },
     [](const CorrelationCalculator& c, SettingsMap& m) { m.addKeyValue("kmi_neighbors", c.k); }},
    {"kraskov_estimator_index",
     [](CorrelationCalculator& c, const SettingsMap& m) {
         if (!m.getValueOpt("kraskov_estimator_index", c.kraskovEstimatorIndex)) return false;
         c.kraskovEstimatorIndex = std::clamp(c.kraskovEstimatorIndex, 1, 2);
         return true;
     },
     [](const CorrelationCalculator& c, SettingsMap& m) { m.addKeyValue("kraskov_estimator_index", c.kraskovEstimatorIndex); }},
    {"devices",
     [](CorrelationCalculator& c, const SettingsMap& m) {
         std::string list;
         if (!m.getValueOpt("devices", list)) return false;
         std::vector<int> ordinals;
         std::istringstream is(list);
         for (std::string item; std::getline(is, item, ',');)
             if (!item.empty()) ordinals.push_back(std::stoi(item));
         if (ordinals.empty()) ordinals.push_back(c.device);
         if (ordinals != c.devices) {
             c.devices = ordinals;
             c.releaseBackend();  // the members have to be distributed again
         }
         return true;
     },
     [](const CorrelationCalculator& c, SettingsMap& m) {
         if (c.devices.size() == 1 && c.devices[0] == c.device) return;
         std::string list;
         for (size_t i = 0; i < c.devices.size(); i++) list += (i ? "," : "") + std::to_string(c.devices[i]);
         m.addKeyValue("devices", list);
     }},
};

void CorrelationCalculator::setSettings(const SettingsMap& settings) {
    ICorrelationCalculator::setSettings(settings);
    if (loadSettings(*this, kSettings, settings)) dirty = true;
}

void CorrelationCalculator::getSettings(SettingsMap& settings) {
    ICorrelationCalculator::getSettings(settings);
    storeSettings(*this, kSettings, settings);
}

void CorrelationCalculator::releaseBackend() {
    if (ctx) crf_destroy(ctx);
    if (group) crf_group_destroy(group);
    ctx = nullptr;
    group = nullptr;
    residentGeneration = ~uint64_t(0);
}

// Keeps the cs member volumes resident in HBM across evaluations; re-uploads only when the member set changes
// (what the reference's LRU field cache does for its per-call getFieldEntryCpu gathers, CorrelationCalculator.cpp:791-800).
void CorrelationCalculator::ensureMembersResident(int timeStepIdx, int ensembleIdx, int cs) {
    const std::string& fieldName = scalarFieldNames.at(size_t(fieldIndexGui));
    const int fixedIdx = isEnsembleMode ? timeStepIdx : ensembleIdx;  // the index that is NOT the member axis
    if ((ctx || group) && residentGeneration == volumeData->getDataGeneration() && residentField == fieldName &&
        residentCs == cs && residentEnsembleMode == isEnsembleMode &&
        (isEnsembleMode ? residentT == fixedIdx : residentE == fixedIdx))
        return;
    // one device: a plain context; several ("devices" setting): a device group -- z-slabs, one worker per device
    if (devices.size() > 1) {
        if (!group && crf_group_create(devices.data(), int(devices.size()), &group) != CRF_OK)
            throw CalculatorError(std::string("Error in CorrelationCalculator::calculateCpu: ") + crf_group_last_error(nullptr));
        if (crf_group_set_grid(group, volumeData->getGridSizeX(), volumeData->getGridSizeY(), volumeData->getGridSizeZ(), cs))
            throwBackendError("calculateCpu");
    } else {
        if (!ctx && crf_create(devices[0], &ctx) != CRF_OK)
            throw CalculatorError(std::string("Error in CorrelationCalculator::calculateCpu: ") + crf_last_error(nullptr));
        if (crf_set_grid(ctx, volumeData->getGridSizeX(), volumeData->getGridSizeY(), volumeData->getGridSizeZ(), cs))
            throwBackendError("calculateCpu");
    }
    std::vector<HostCacheEntry> fieldEntries;
    std::vector<const float*> fields;
    fieldEntries.reserve(size_t(cs));
    fields.reserve(size_t(cs));
    for (int fieldIdx = 0; fieldIdx < cs; fieldIdx++) {
        HostCacheEntry fieldEntry = getFieldEntryCpu(fieldName, fieldIdx, timeStepIdx, ensembleIdx);
        fieldEntries.push_back(fieldEntry);
        fields.push_back(fieldEntry->data<float>());
    }
    if (group ? crf_group_upload_members(group, fields.data()) : crf_upload_members(ctx, fields.data()))
        throwBackendError("calculateCpu");
    residentGeneration = volumeData->getDataGeneration();
    residentField = fieldName;
    residentCs = cs;
    residentEnsembleMode = isEnsembleMode;
    residentT = isEnsembleMode ? fixedIdx : -1;
    residentE = isEnsembleMode ? -1 : fixedIdx;
}

// SEPARATE_SYMMETRIC: the second field's members (fieldEntriesSecondary, CorrelationCalculator.cpp:1182-1216).  Uploaded
// per evaluation: the mode is rare and a second residency cache would have to follow the primary one's invalidation.
void CorrelationCalculator::uploadSecondaryMembers(int timeStepIdx, int ensembleIdx, int cs) {
    const std::string& fieldName = scalarFieldNames.at(size_t(fieldIndex2Gui));
    std::vector<HostCacheEntry> fieldEntries;
    std::vector<const float*> fields;
    for (int fieldIdx = 0; fieldIdx < cs; fieldIdx++) {
        fieldEntries.push_back(getFieldEntryCpu(fieldName, fieldIdx, timeStepIdx, ensembleIdx));
        fields.push_back(fieldEntries.back()->data<float>());
    }
    if (group ? crf_group_upload_secondary_members(group, fields.data()) : crf_upload_secondary_members(ctx, fields.data()))
        throwBackendError("calculateCpu");
}

void CorrelationCalculator::calculateCpu(int timeStepIdx, int ensembleIdx, float* buffer) {
    const int xs = volumeData->getGridSizeX();
    const int ys = volumeData->getGridSizeY();
    const int cs = getCorrelationMemberCount();
    ensureMembersResident(timeStepIdx, ensembleIdx, cs);

    crf_params params{};
    params.measure = int(correlationMeasureType);
    params.ref_x = referencePointIndex[0];
    params.ref_y = referencePointIndex[1];
    params.ref_z = referencePointIndex[2];
    params.k = k;
    params.kraskov_estimator_index = kraskovEstimatorIndex;
    params.num_bins = numBins;

    // SEPARATE mode: reference vector from the second field, optionally at a lagged time step (:804-813)
    std::vector<float> referenceValues;
    if (correlationFieldMode == CorrelationFieldMode::SEPARATE) {
        const size_t referencePointIdx = (size_t(referencePointIndex[2]) * size_t(ys) + size_t(referencePointIndex[1])) *
                                                 size_t(xs) + size_t(referencePointIndex[0]);  // IDXS
        const int timeStepIdxReference = useTimeLagCorrelations ? timeLagTimeStepIdx : timeStepIdx;
        referenceValues.resize(size_t(cs));
        for (int c = 0; c < cs; c++) {
            HostCacheEntry fieldEntry = getFieldEntryCpu(scalarFieldNames.at(size_t(fieldIndex2Gui)), c,
                                                         timeStepIdxReference, ensembleIdx);
            referenceValues[size_t(c)] = fieldEntry->dataAt<float>(referencePointIdx);
        }
        params.reference_values = referenceValues.data();
    }

    if (isMeasureBinnedMI(correlationMeasureType)) {  // :820-846
        float minFieldValRef = std::numeric_limits<float>::max();
        float maxFieldValRef = std::numeric_limits<float>::lowest();
        const std::string& refField = scalarFieldNames.at(
                size_t(correlationFieldMode != CorrelationFieldMode::SINGLE ? fieldIndex2Gui : fieldIndexGui));
        for (int fieldIdx = 0; fieldIdx < cs; fieldIdx++) {
            auto [minVal, maxVal] = getMinMaxScalarFieldValue(refField, fieldIdx, timeStepIdx, ensembleIdx);
            minFieldValRef = std::min(minFieldValRef, minVal);
            maxFieldValRef = std::max(maxFieldValRef, maxVal);
        }
        float minFieldValQuery = minFieldValRef, maxFieldValQuery = maxFieldValRef;
        if (correlationFieldMode != CorrelationFieldMode::SINGLE) {
            minFieldValQuery = std::numeric_limits<float>::max();
            maxFieldValQuery = std::numeric_limits<float>::lowest();
            for (int fieldIdx = 0; fieldIdx < cs; fieldIdx++) {
                auto [minVal, maxVal] = getMinMaxScalarFieldValue(scalarFieldNames.at(size_t(fieldIndexGui)), fieldIdx,
                                                                  timeStepIdx, ensembleIdx);
                minFieldValQuery = std::min(minFieldValQuery, minVal);
                maxFieldValQuery = std::max(maxFieldValQuery, maxVal);
            }
        }
        params.min_ref = minFieldValRef;
        params.max_ref = maxFieldValRef;
        params.min_query = minFieldValQuery;
        params.max_query = maxFieldValQuery;
    }

    // SEPARATE_SYMMETRIC: the reference's calculateCpu has no branch for it (it evaluates the SINGLE mode); its
    // accelerator path correlates the two fields voxel by voxel (:1182-1229, CorrelationMain.glsl:10-15) and that is
    // what this backend computes: X = scalarFields (field 1), Y = scalarFieldsSecondary (field 2).
    if (correlationFieldMode == CorrelationFieldMode::SEPARATE_SYMMETRIC) {
        uploadSecondaryMembers(timeStepIdx, ensembleIdx, cs);
        params.flags |= CRF_FLAG_SYMMETRIC;
        std::swap(params.min_ref, params.min_query);  // the ranges above are (field 2, field 1): X is field 1 here
        std::swap(params.max_ref, params.max_query);
    }

    // kernel time of this evaluation: the sum over the voxel ranges of the host-output path (slowest device of a group)
    double ms = 0.0;
    int launches = 0;
    if (group) {
        crf_group_set_profiling(group, 1);
        if (crf_group_compute(group, &params, buffer)) throwBackendError("calculateCpu");
        lastKernelMs = (crf_group_take_kernel_time(group, &ms, &launches) == CRF_OK && launches > 0) ? ms : -1.0;
    } else {
        crf_set_profiling(ctx, 1);
        if (crf_compute(ctx, &params, buffer)) throwBackendError("calculateCpu");
        lastKernelMs = (crf_take_kernel_time(ctx, &ms, &launches) == CRF_OK && launches > 0) ? ms : -1.0;
    }
}

}  // namespace crfhost
