// EnsembleCalculators.hpp -- host-side mirrors of the reference's sibling per-voxel ensemble calculators (SURVEY
// section 8(f) rank 3), each forwarding calculateCpu to the C ABI of include/corrfield.h:
//   EnsembleMeanCalculator     src/Calculators/EnsembleMeanCalculator.{hpp,cpp}    "Ensemble Mean"
//   EnsembleSpreadCalculator   src/Calculators/EnsembleSpreadCalculator.{hpp,cpp}  "Ensemble Variance" (sic: it writes the
//                              sample standard deviation, EnsembleSpreadCalculator.cpp:110-145)
//   SetPredicateCalculator     src/Calculators/SetPredicateCalculator.{hpp,cpp}    "Set Predicate"
//   DKLCalculator              src/Calculators/DKLCalculator.{hpp,cpp}             "KL-Divergence"
// Same output names (with " (n)" for duplicates), settings keys, defaults and member-axis rules as the reference; no
// GUI, no Vulkan passes.  The member volumes are kept resident in HBM across evaluations like CorrelationCalculator's.
#pragma once
#include <string>
#include <vector>

#include "../../../include/corrfield.h"
#include "VolumeData.hpp"

namespace crfhost {

enum class ComparisonOperatorType { GREATER, GREATER_EQUAL, LESS, LESS_EQUAL, EQUAL, NOT_EQUAL };  // SetPredicateCalculator.hpp:41-43
extern const char* const COMPARISON_OPERATOR_NAMES[6];                                              // :44-46
enum class DKLEstimatorType { BINNED, ENTROPY_KNN };                                                // DKLCalculator.hpp:39-41
extern const char* const DKL_ESTIMATOR_TYPE_NAMES[2];                                               // :42-44

// Common part: field selection, member axis, residency of the members on the device.
class EnsembleReduceCalculator : public Calculator {
public:
    explicit EnsembleReduceCalculator(int deviceOrdinal, const char* baseName, CalculatorType type)
        : device(deviceOrdinal), baseName(baseName), type(type) {}
    ~EnsembleReduceCalculator() override;
    CalculatorType getCalculatorType() const override { return type; }
    void setVolumeData(VolumeData* _volumeData, bool isNewData) override;
    std::string getOutputFieldName() override;
    FilterDevice getFilterDevice() override { return FilterDevice::CPU; }
    void setSettings(const SettingsMap& settings) override;
    void getSettings(SettingsMap& settings) override;

protected:
    /// Member axis: the ensemble members (EnsembleMean/Spread/DKL always, :98 / DKLCalculator.cpp:138); SetPredicate may
    /// switch to time steps (SetPredicateCalculator.cpp:107-118).
    virtual bool membersAreEnsemble() const { return true; }
    int getMemberCount() const;
    crf_context* residentContext(int timeStepIdx, int ensembleIdx);
    [[noreturn]] void throwBackendError(const char* where) const;

    int device;
    std::string baseName;
    CalculatorType type;
    std::vector<std::string> scalarFieldNames;
    int scalarFieldIndex = 0, scalarFieldIndexGui = 0;
    crf_context* ctx = nullptr;
    uint64_t residentGeneration = ~uint64_t(0);
    std::string residentField;
    int residentFixedIdx = -1, residentCs = -1;
    bool residentEnsembleAxis = true;
};

class EnsembleMeanCalculator : public EnsembleReduceCalculator {
public:
    explicit EnsembleMeanCalculator(int deviceOrdinal = 0)
        : EnsembleReduceCalculator(deviceOrdinal, "Ensemble Mean", CalculatorType::ENSEMBLE_MEAN) {}
    void calculateCpu(int timeStepIdx, int ensembleIdx, float* buffer) override;
};

class EnsembleSpreadCalculator : public EnsembleReduceCalculator {
public:
    explicit EnsembleSpreadCalculator(int deviceOrdinal = 0)
        : EnsembleReduceCalculator(deviceOrdinal, "Ensemble Variance", CalculatorType::ENSEMBLE_SPREAD) {}
    void calculateCpu(int timeStepIdx, int ensembleIdx, float* buffer) override;
};

class SetPredicateCalculator : public EnsembleReduceCalculator {
public:
    explicit SetPredicateCalculator(int deviceOrdinal = 0)
        : EnsembleReduceCalculator(deviceOrdinal, "Set Predicate", CalculatorType::SET_PREDICATE) {}
    void setVolumeData(VolumeData* _volumeData, bool isNewData) override;
    void setSettings(const SettingsMap& settings) override;
    void getSettings(SettingsMap& settings) override;
    void calculateCpu(int timeStepIdx, int ensembleIdx, float* buffer) override;
    bool getHasFixedRange() const override { return true; }
    std::pair<float, float> getFixedRange() const override { return {0.0f, 1.0f}; }
    int getCountLower() const { return countLower; }
    int getCountUpper() const { return countUpper; }

protected:
    bool membersAreEnsemble() const override { return isEnsembleMode; }

private:
    bool isEnsembleMode = true;
    int countLower = 1, countUpper = 1;  // SetPredicateCalculator.hpp:107; reset to cs/2 with new data (:143-144)
    float comparisonValue = 0.0f;
    ComparisonOperatorType comparisonOperatorType = ComparisonOperatorType::GREATER;
};

class DKLCalculator : public EnsembleReduceCalculator {
public:
    explicit DKLCalculator(int deviceOrdinal = 0)
        : EnsembleReduceCalculator(deviceOrdinal, "KL-Divergence", CalculatorType::DKL_CALCULATOR) {}
    void setVolumeData(VolumeData* _volumeData, bool isNewData) override;
    void setSettings(const SettingsMap& settings) override;
    void getSettings(SettingsMap& settings) override;
    void calculateCpu(int timeStepIdx, int ensembleIdx, float* buffer) override;
    int getNumNeighbors() const { return k; }

private:
    DKLEstimatorType estimatorType = DKLEstimatorType::ENTROPY_KNN;  // DKLCalculator.hpp:126
    int numBins = 80, k = 3, kMax = 20;                               // :129-131
};

}  // namespace crfhost
