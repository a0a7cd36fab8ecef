#include "CorrelationCalculator.hpp"

#include <algorithm>
#include <cmath>
#include <limits>

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

void ICorrelationCalculator::setSettings(const SettingsMap& settings) {
    std::string correlationFieldModeString;
    if (settings.getValueOpt("correlation_field_mode", correlationFieldModeString)) {
        for (int i = 0; i < 3; i++) {
            if (correlationFieldModeString == CORRELATION_FIELD_MODE_NAMES[i]) {
                correlationFieldMode = CorrelationFieldMode(i);
                break;
            }
        }
        dirty = true;
    }
    if (correlationFieldMode != CorrelationFieldMode::SINGLE) {
        if (settings.getValueOpt("scalar_field_idx_ref", fieldIndex2Gui)) {
            fieldIndex2 = fieldIndex2Gui;
            dirty = true;
        }
        if (settings.getValueOpt("scalar_field_idx_query", fieldIndexGui)) {
            fieldIndex = fieldIndexGui;
            dirty = true;
        }
    } else if (settings.getValueOpt("scalar_field_idx", fieldIndexGui)) {
        fieldIndex = fieldIndexGui;
        dirty = true;
    }
    std::string ensembleModeName;
    if (settings.getValueOpt("correlation_mode", ensembleModeName)) {
        isEnsembleMode = ensembleModeName == CORRELATION_MODE_NAMES[0];
        onCorrelationMemberCountChanged();
        dirty = true;
    }
    bool referencePointChanged = false;
    referencePointChanged |= settings.getValueOpt("reference_point_x", referencePointIndex[0]);
    referencePointChanged |= settings.getValueOpt("reference_point_y", referencePointIndex[1]);
    referencePointChanged |= settings.getValueOpt("reference_point_z", referencePointIndex[2]);
    if (referencePointChanged) dirty = true;
    if (settings.getValueOpt("use_buffer_tiling", useBufferTiling)) dirty = true;
    if (settings.getValueOpt("use_time_lag_correlations", useTimeLagCorrelations)) dirty = true;
    if (settings.getValueOpt("time_lag_time_step_idx", timeLagTimeStepIdx)) dirty = true;
}

void ICorrelationCalculator::getSettings(SettingsMap& settings) {
    settings.addKeyValue("correlation_field_mode", CORRELATION_FIELD_MODE_NAMES[int(correlationFieldMode)]);
    if (correlationFieldMode != CorrelationFieldMode::SINGLE) {
        settings.addKeyValue("scalar_field_idx_ref", fieldIndex2Gui);
        settings.addKeyValue("scalar_field_idx_query", fieldIndexGui);
    } else {
        settings.addKeyValue("scalar_field_idx", fieldIndexGui);
    }
    settings.addKeyValue("correlation_mode", CORRELATION_MODE_NAMES[isEnsembleMode ? 0 : 1]);
    settings.addKeyValue("reference_point_x", referencePointIndex[0]);
    settings.addKeyValue("reference_point_y", referencePointIndex[1]);
    settings.addKeyValue("reference_point_z", referencePointIndex[2]);
    settings.addKeyValue("data_mode", "Buffer Array");
    settings.addKeyValue("use_buffer_tiling", useBufferTiling);
    settings.addKeyValue("use_time_lag_correlations", useTimeLagCorrelations);
    settings.addKeyValue("time_lag_time_step_idx", timeLagTimeStepIdx);
}

// ---------------------------------------------------------------------------------------------------------
// CorrelationCalculator (CorrelationCalculator.cpp:569-779)
// ---------------------------------------------------------------------------------------------------------
CorrelationCalculator::CorrelationCalculator(int device) : device(device) {}

CorrelationCalculator::~CorrelationCalculator() {
    if (ctx) crf_destroy(ctx);
}

void CorrelationCalculator::throwBackendError(const char* where) {
    throw CalculatorError(std::string("Error in CorrelationCalculator::") + where + ": " + crf_last_error(ctx));
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

void CorrelationCalculator::setSettings(const SettingsMap& settings) {
    ICorrelationCalculator::setSettings(settings);
    std::string correlationMeasureTypeName;
    if (settings.getValueOpt("correlation_measure_type", correlationMeasureTypeName)) {
        for (int i = 0; i < 7; i++) {
            if (correlationMeasureTypeName == CORRELATION_MEASURE_TYPE_IDS[i]) {
                correlationMeasureType = CorrelationMeasureType(i);
                break;
            }
        }
        hasNameChanged = true;
        dirty = true;
    }
    std::string deviceName;
    if (settings.getValueOpt("device", deviceName)) {
        // "CPU" | "Vulkan" | "CUDA"; anything else falls back to the accelerator (CorrelationCalculator.cpp:721-747).
        const bool useGpuOld = useGpu;
        useGpu = deviceName != "CPU";
        useCuda = deviceName == "CUDA";
        hasFilterDeviceChanged = useGpuOld != useGpu;
        dirty = true;
    }
    if (settings.getValueOpt("calculate_absolute_value", calculateAbsoluteValue)) dirty = true;
    if (settings.getValueOpt("mi_bins", numBins)) dirty = true;
    if (settings.getValueOpt("kmi_neighbors", k)) dirty = true;
    if (settings.getValueOpt("kraskov_estimator_index", kraskovEstimatorIndex)) {
        kraskovEstimatorIndex = std::clamp(kraskovEstimatorIndex, 1, 2);
        dirty = true;
    }
}

void CorrelationCalculator::getSettings(SettingsMap& settings) {
    ICorrelationCalculator::getSettings(settings);
    settings.addKeyValue("correlation_measure_type", CORRELATION_MEASURE_TYPE_IDS[int(correlationMeasureType)]);
    settings.addKeyValue("device", !useGpu ? "CPU" : (!useCuda ? "Vulkan" : "CUDA"));
    settings.addKeyValue("calculate_absolute_value", calculateAbsoluteValue);
    settings.addKeyValue("mi_bins", numBins);
    settings.addKeyValue("kmi_neighbors", k);
    settings.addKeyValue("kraskov_estimator_index", kraskovEstimatorIndex);
}

// Keeps the cs member volumes resident in HBM across evaluations; re-uploads only when the member set changes
// (what the reference's LRU field cache does for its per-call getFieldEntryCpu gathers, CorrelationCalculator.cpp:791-800).
void CorrelationCalculator::ensureMembersResident(int timeStepIdx, int ensembleIdx, int cs) {
    const std::string& fieldName = scalarFieldNames.at(size_t(fieldIndexGui));
    const int fixedIdx = isEnsembleMode ? timeStepIdx : ensembleIdx;  // the index that is NOT the member axis
    if (ctx && residentGeneration == volumeData->getDataGeneration() && residentField == fieldName &&
        residentCs == cs && residentEnsembleMode == isEnsembleMode &&
        (isEnsembleMode ? residentT == fixedIdx : residentE == fixedIdx))
        return;
    if (!ctx && crf_create(device, &ctx) != CRF_OK)
        throw CalculatorError(std::string("Error in CorrelationCalculator::calculateCpu: ") + crf_last_error(nullptr));
    if (crf_set_grid(ctx, volumeData->getGridSizeX(), volumeData->getGridSizeY(), volumeData->getGridSizeZ(), cs))
        throwBackendError("calculateCpu");
    std::vector<HostCacheEntry> fieldEntries;
    std::vector<const float*> fields;
    fieldEntries.reserve(size_t(cs));
    fields.reserve(size_t(cs));
    for (int fieldIdx = 0; fieldIdx < cs; fieldIdx++) {
        HostCacheEntry fieldEntry = getFieldEntryCpu(fieldName, fieldIdx, timeStepIdx, ensembleIdx);
        fieldEntries.push_back(fieldEntry);
        fields.push_back(fieldEntry->data<float>());
    }
    if (crf_upload_members(ctx, fields.data())) throwBackendError("calculateCpu");
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
    if (crf_upload_secondary_members(ctx, fields.data())) throwBackendError("calculateCpu");
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

    crf_set_profiling(ctx, 1);
    if (crf_compute(ctx, &params, buffer)) throwBackendError("calculateCpu");
    double ms = 0.0;
    int launches = 0;
    lastKernelMs = (crf_take_kernel_time(ctx, &ms, &launches) == CRF_OK && launches > 0) ? ms / launches : -1.0;
}

}  // namespace crfhost
