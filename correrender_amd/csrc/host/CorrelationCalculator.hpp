// CorrelationCalculator.hpp -- host-side mirror of the reference's correlation calculator, backed by libcorrfield.
//   class ICorrelationCalculator   /root/reference/src/Calculators/CorrelationCalculator.hpp:66-139
//   class CorrelationCalculator    CorrelationCalculator.hpp:150-214, CorrelationCalculator.cpp:576-1154
//   enums / string ids             /root/reference/src/Calculators/CorrelationDefines.hpp:41-75
// Same entry points (calculateCpu(t, e, buffer), setSettings/getSettings with the reference's keys, setReferencePoint,
// getOutputFieldName, getFixedRange ...) and the same state defaults; the hot loop of calculateCpu
// (CorrelationCalculator.cpp:868-1142) is replaced by one crf_compute call.
#pragma once
#include <array>
#include <string>
#include <vector>

#include "../../../include/corrfield.h"
#include "Calculator.hpp"
#include "VolumeData.hpp"

namespace crfhost {

enum class CorrelationMeasureType {
    PEARSON, SPEARMAN, KENDALL, MUTUAL_INFORMATION_BINNED, MUTUAL_INFORMATION_KRASKOV,
    BINNED_MI_CORRELATION_COEFFICIENT, KMI_CORRELATION_COEFFICIENT
};
extern const char* const CORRELATION_MEASURE_TYPE_NAMES[7];
extern const char* const CORRELATION_MEASURE_TYPE_IDS[7];
extern const char* const CORRELATION_MODE_NAMES[2];
extern const char* const CORRELATION_FIELD_MODE_NAMES[3];
inline bool isMeasureBinnedMI(CorrelationMeasureType m) {
    return m == CorrelationMeasureType::MUTUAL_INFORMATION_BINNED ||
           m == CorrelationMeasureType::BINNED_MI_CORRELATION_COEFFICIENT;
}
inline bool isMeasureKraskovMI(CorrelationMeasureType m) {
    return m == CorrelationMeasureType::MUTUAL_INFORMATION_KRASKOV ||
           m == CorrelationMeasureType::KMI_CORRELATION_COEFFICIENT;
}
inline bool isMeasureCorrelationCoefficientMI(CorrelationMeasureType m) {
    return m == CorrelationMeasureType::BINNED_MI_CORRELATION_COEFFICIENT ||
           m == CorrelationMeasureType::KMI_CORRELATION_COEFFICIENT;
}

enum class CorrelationFieldMode { SINGLE, SEPARATE, SEPARATE_SYMMETRIC };

class ICorrelationCalculator : public Calculator {
public:
    void setVolumeData(VolumeData* _volumeData, bool isNewData) override;
    bool getComputesCorrelation() const override { return true; }
    FieldType getOutputFieldType() override { return FieldType::SCALAR; }
    int getInputFieldIndex() const { return fieldIndex; }
    void setReferencePoint(const std::array<int, 3>& referencePoint);
    /// World position (a picking hit on the volume's rendering box) -> nearest grid point, clamped
    /// (CorrelationCalculator.cpp:204-217).
    void setReferencePointFromWorld(const std::array<float, 3>& worldPosition);
    const std::array<int, 3>& getReferencePoint() const { return referencePointIndex; }
    bool getIsEnsembleMode() const { return isEnsembleMode; }
    int getCorrelationMemberCount() const;
    HostCacheEntry getFieldEntryCpu(const std::string& fieldName, int fieldIdx, int timeStepIdx, int ensembleIdx);
    std::pair<float, float> getMinMaxScalarFieldValue(const std::string& fieldName, int fieldIdx, int timeStepIdx,
                                                      int ensembleIdx);
    void setSettings(const SettingsMap& settings) override;
    void getSettings(SettingsMap& settings) override;

protected:
    virtual void onCorrelationMemberCountChanged() {}
    static const SettingBinding<ICorrelationCalculator> kSettings[12];
    std::vector<std::string> scalarFieldNames;
    int fieldIndex = 0, fieldIndexGui = 0;
    int fieldIndex2 = 0, fieldIndex2Gui = 0;
    std::array<int, 3> referencePointIndex{0, 0, 0};
    bool useBufferTiling = true;
    CorrelationFieldMode correlationFieldMode = CorrelationFieldMode::SINGLE;
    bool isEnsembleMode = true;
    bool useTimeLagCorrelations = false;
    int timeLagTimeStepIdx = 0;
};

class CorrelationCalculator : public ICorrelationCalculator {
public:
    /// device: HIP device ordinal the members are kept on (the "devices" setting may name several).
    explicit CorrelationCalculator(int device = 0);
    ~CorrelationCalculator() override;
    CalculatorType getCalculatorType() const override { return CalculatorType::CORRELATION; }
    std::string getOutputFieldName() override;
    void setVolumeData(VolumeData* _volumeData, bool isNewData) override;
    /// The HIP backend fills host buffers through calculateCpu, so it registers like a CPU filter
    /// (CorrelationCalculator.cpp:635-644 returns VULKAN/CUDA only for the Vulkan/CUDA image path).
    FilterDevice getFilterDevice() override { return FilterDevice::CPU; }
    bool getIsRealtime() const;
    bool getHasFixedRange() const override;
    std::pair<float, float> getFixedRange() const override;
    void calculateCpu(int timeStepIdx, int ensembleIdx, float* buffer) override;
    void setSettings(const SettingsMap& settings) override;
    void getSettings(SettingsMap& settings) override;

    CorrelationMeasureType getCorrelationMeasureType() const { return correlationMeasureType; }
    int getKraskovNumNeighbors() const { return k; }
    int getKraskovNumNeighborsMax() const { return kMax; }
    /// Kernel time of the last calculateCpu in ms (HIP events around the per-voxel kernels; slowest device of a group),
    /// < 0 if unavailable.
    double getLastKernelTimeMs() const { return lastKernelMs; }

protected:
    void onCorrelationMemberCountChanged() override;

private:
    static const SettingBinding<CorrelationCalculator> kSettings[7];
    void ensureMembersResident(int timeStepIdx, int ensembleIdx, int cs);
    void releaseBackend();
    void uploadSecondaryMembers(int timeStepIdx, int ensembleIdx, int cs);
    [[noreturn]] void throwBackendError(const char* where);

    int device;                 ///< default device ordinal (constructor)
    std::vector<int> devices;   ///< "devices" setting; more than one entry = crf_group
    crf_context* ctx = nullptr;
    crf_group* group = nullptr;
    // identity of the member set currently resident in HBM
    uint64_t residentGeneration = ~uint64_t(0);
    std::string residentField;
    int residentT = -1, residentE = -1, residentCs = -1;
    bool residentEnsembleMode = true;
    int cachedMemberCount = 0;
    CorrelationMeasureType correlationMeasureType = CorrelationMeasureType::MUTUAL_INFORMATION_KRASKOV;
    bool useGpu = true;   ///< "device" setting: "CPU" is accepted and remembered, evaluation always runs on the GPU.
    bool useCuda = false;
    bool calculateAbsoluteValue = false;  ///< kept for settings round trips; NOT applied on the calculateCpu path
                                          ///  (reference: shader define only, CorrelationCalculator.cpp:1662-1664).
    int numBins = 80;
    int k = 3;
    int kMax = 20;
    int kraskovEstimatorIndex = 1;
    double lastKernelMs = -1.0;
};

}  // namespace crfhost
