// EnsembleCalculators.cpp -- see EnsembleCalculators.hpp.
#include "EnsembleCalculators.hpp"

#include <algorithm>

namespace crfhost {

const char* const COMPARISON_OPERATOR_NAMES[6] = {">", ">=", "<", "<=", "==", "!="};
const char* const DKL_ESTIMATOR_TYPE_NAMES[2] = {"Binning", "Entropy k-NN"};

static inline int iceil(int a, int b) { return (a + b - 1) / b; }

EnsembleReduceCalculator::~EnsembleReduceCalculator() {
    if (ctx) crf_destroy(ctx);
}

void EnsembleReduceCalculator::throwBackendError(const char* where) const {
    throw CalculatorError(std::string("Error in ") + baseName + " calculator (" + where + "): " + crf_last_error(ctx));
}

std::string EnsembleReduceCalculator::getOutputFieldName() {
    std::string outputFieldName = baseName;
    if (calculatorConstructorUseCount > 1) outputFieldName += " (" + std::to_string(calculatorConstructorUseCount) + ")";
    return outputFieldName;
}

void EnsembleReduceCalculator::setVolumeData(VolumeData* _volumeData, bool isNewData) {
    Calculator::setVolumeData(_volumeData, isNewData);
    if (isNewData) calculatorConstructorUseCount = volumeData->getNewCalculatorUseCount(type);
    scalarFieldNames.clear();
    for (const std::string& name : volumeData->getFieldNames(FieldType::SCALAR))
        if (name != getOutputFieldName()) scalarFieldNames.push_back(name);  // EnsembleMeanCalculator.cpp:64-71
    if (isNewData) scalarFieldIndex = scalarFieldIndexGui = volumeData->getStandardScalarFieldIdx();
}

void EnsembleReduceCalculator::setSettings(const SettingsMap& settings) {
    if (settings.getValueOpt("scalar_field_idx", scalarFieldIndexGui)) {  // EnsembleMeanCalculator.cpp:183-190
        scalarFieldIndex = scalarFieldIndexGui;
        dirty = true;
    }
}

void EnsembleReduceCalculator::getSettings(SettingsMap& settings) {
    settings.addKeyValue("scalar_field_idx", scalarFieldIndexGui);
}

int EnsembleReduceCalculator::getMemberCount() const {
    return membersAreEnsemble() ? volumeData->getEnsembleMemberCount() : volumeData->getTimeStepCount();
}

crf_context* EnsembleReduceCalculator::residentContext(int timeStepIdx, int ensembleIdx) {
    const int cs = getMemberCount();
    const bool ensembleAxis = membersAreEnsemble();
    const std::string& fieldName = scalarFieldNames.at(size_t(scalarFieldIndex));
    const int fixedIdx = ensembleAxis ? timeStepIdx : ensembleIdx;
    if (ctx && residentGeneration == volumeData->getDataGeneration() && residentField == fieldName && residentCs == cs &&
        residentEnsembleAxis == ensembleAxis && residentFixedIdx == fixedIdx)
        return ctx;
    if (!ctx && crf_create(device, &ctx) != CRF_OK)
        throw CalculatorError(std::string("Error in ") + baseName + " calculator: " + crf_last_error(nullptr));
    if (crf_set_grid(ctx, volumeData->getGridSizeX(), volumeData->getGridSizeY(), volumeData->getGridSizeZ(), cs))
        throwBackendError("crf_set_grid");
    std::vector<HostCacheEntry> entries;
    std::vector<const float*> fields;
    for (int c = 0; c < cs; c++) {
        entries.push_back(volumeData->getFieldEntryCpu(FieldType::SCALAR, fieldName, ensembleAxis ? timeStepIdx : c,
                                                       ensembleAxis ? c : ensembleIdx));
        fields.push_back(entries.back()->data<float>());
    }
    if (crf_upload_members(ctx, fields.data())) throwBackendError("crf_upload_members");
    residentGeneration = volumeData->getDataGeneration();
    residentField = fieldName;
    residentCs = cs;
    residentEnsembleAxis = ensembleAxis;
    residentFixedIdx = fixedIdx;
    return ctx;
}

void EnsembleMeanCalculator::calculateCpu(int timeStepIdx, int ensembleIdx, float* buffer) {
    if (crf_compute_ensemble_stat(residentContext(timeStepIdx, ensembleIdx), CRF_ENSEMBLE_MEAN, buffer))
        throwBackendError("calculateCpu");
}

void EnsembleSpreadCalculator::calculateCpu(int timeStepIdx, int ensembleIdx, float* buffer) {
    if (crf_compute_ensemble_stat(residentContext(timeStepIdx, ensembleIdx), CRF_ENSEMBLE_SPREAD, buffer))
        throwBackendError("calculateCpu");
}

// ---- SetPredicateCalculator ------------------------------------------------------------------------------------
void SetPredicateCalculator::setVolumeData(VolumeData* _volumeData, bool isNewData) {
    EnsembleReduceCalculator::setVolumeData(_volumeData, isNewData);
    const int es = volumeData->getEnsembleMemberCount(), ts = volumeData->getTimeStepCount();
    if (isEnsembleMode && es <= 1 && ts > 1) {  // SetPredicateCalculator.cpp:62-69
        isEnsembleMode = false;
    } else if (!isEnsembleMode && ts <= 1 && es > 1) {
        isEnsembleMode = true;
    }
    if (isNewData) countLower = countUpper = getMemberCount() / 2;  // :143-144
}

void SetPredicateCalculator::setSettings(const SettingsMap& settings) {
    EnsembleReduceCalculator::setSettings(settings);
    std::string modeName;
    if (settings.getValueOpt("correlation_mode", modeName)) {  // :357-366
        if (modeName == "Ensemble") isEnsembleMode = true;
        if (modeName == "Time") isEnsembleMode = false;
        dirty = true;
    }
    if (settings.getValueOpt("count_lower", countLower)) dirty = true;
    if (settings.getValueOpt("count_upper", countUpper)) dirty = true;
    std::string opName;
    if (settings.getValueOpt("comparison_operator_type", opName)) {  // :388-397
        for (int i = 0; i < 6; i++)
            if (opName == COMPARISON_OPERATOR_NAMES[i]) comparisonOperatorType = ComparisonOperatorType(i);
        dirty = true;
    }
    if (settings.getValueOpt("comparison_value", comparisonValue)) dirty = true;
}

void SetPredicateCalculator::getSettings(SettingsMap& settings) {
    settings.addKeyValue("correlation_mode", isEnsembleMode ? "Ensemble" : "Time");
    settings.addKeyValue("count_lower", countLower);
    settings.addKeyValue("count_upper", countUpper);
    EnsembleReduceCalculator::getSettings(settings);
    settings.addKeyValue("comparison_operator_type", COMPARISON_OPERATOR_NAMES[int(comparisonOperatorType)]);
    settings.addKeyValue("comparison_value", comparisonValue);
}

void SetPredicateCalculator::calculateCpu(int timeStepIdx, int ensembleIdx, float* buffer) {
    if (crf_compute_set_predicate(residentContext(timeStepIdx, ensembleIdx), int(comparisonOperatorType), comparisonValue,
                                  countLower, countUpper, buffer))
        throwBackendError("calculateCpu");
}

// ---- DKLCalculator -----------------------------------------------------------------------------------------------
void DKLCalculator::setVolumeData(VolumeData* _volumeData, bool isNewData) {
    EnsembleReduceCalculator::setVolumeData(_volumeData, isNewData);
    if (isNewData) {  // onMemberCountChanged, DKLCalculator.cpp:94-101
        const int cs = volumeData->getEnsembleMemberCount();
        k = std::max(iceil(3 * cs, 100), 1);
        kMax = std::max(iceil(7 * cs, 100), 20);
    }
}

void DKLCalculator::setSettings(const SettingsMap& settings) {
    EnsembleReduceCalculator::setSettings(settings);
    std::string name;
    if (settings.getValueOpt("estimator_type", name)) {  // :394-404
        for (int i = 0; i < 2; i++)
            if (name == DKL_ESTIMATOR_TYPE_NAMES[i]) estimatorType = DKLEstimatorType(i);
        dirty = true;
    }
    if (settings.getValueOpt("mi_bins", numBins)) dirty = true;
    if (settings.getValueOpt("knn_neighbors", k)) dirty = true;
}

void DKLCalculator::getSettings(SettingsMap& settings) {
    EnsembleReduceCalculator::getSettings(settings);
    settings.addKeyValue("estimator_type", DKL_ESTIMATOR_TYPE_NAMES[int(estimatorType)]);
    settings.addKeyValue("mi_bins", numBins);
    settings.addKeyValue("knn_neighbors", k);
}

void DKLCalculator::calculateCpu(int timeStepIdx, int ensembleIdx, float* buffer) {
    if (crf_compute_dkl(residentContext(timeStepIdx, ensembleIdx), int(estimatorType), numBins, k, buffer))
        throwBackendError("calculateCpu");
}

}  // namespace crfhost
