// Calculator.hpp -- host-side mirror of the reference's calculator plugin surface, restricted to what the
// correlation-field path touches.  Same names, argument meaning and behaviour as
//   class Calculator              /root/reference/src/Calculators/Calculator.hpp:86-138
//   enum class FilterDevice       Calculator.hpp:79-81
//   enum class CalculatorType     Calculator.hpp:57-63  (the ensemble calculators implemented here)
//   class SettingsMap             /root/reference/src/Utils/InternalState.hpp:41-113
// so that tests against this mirror read like tests against the reference classes, and so that the reference-side
// subclass shown in INTEGRATION.md is a line-for-line transplant.  No rendering, GUI or Vulkan types.
#pragma once
#include <cstdint>
#include <initializer_list>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>

namespace crfhost {

enum class FieldType : uint32_t { SCALAR = 0 };
enum class CalculatorType : uint32_t {  // values of Calculator.hpp:57-63
    ENSEMBLE_MEAN = 6, ENSEMBLE_SPREAD = 7, SET_PREDICATE = 8, CORRELATION = 10, DKL_CALCULATOR = 15, INVALID = 16
};
enum class FilterDevice { CPU, VULKAN, CUDA };

// All values are strings; booleans accept "1"/"true" (InternalState.hpp:56-62).
class SettingsMap {
public:
    SettingsMap() = default;
    explicit SettingsMap(std::map<std::string, std::string> m) : settings(std::move(m)) {}
    SettingsMap(std::initializer_list<std::pair<const std::string, std::string>> il) : settings(il) {}
    void addKeyValue(const std::string& key, const std::string& value) { settings[key] = value; }
    void addKeyValue(const std::string& key, const char* value) { settings[key] = value; }
    void addKeyValue(const std::string& key, bool value) { settings[key] = value ? "1" : "0"; }
    template <class T>
    void addKeyValue(const std::string& key, const T& value) {
        std::ostringstream os;
        os << value;
        settings[key] = os.str();
    }
    bool getValueOpt(const char* key, std::string& toset) const {
        auto it = settings.find(key);
        if (it == settings.end()) return false;
        toset = it->second;
        return true;
    }
    bool getValueOpt(const char* key, bool& toset) const {
        auto it = settings.find(key);
        if (it == settings.end()) return false;
        toset = (it->second == "true") || (it->second == "1");
        return true;
    }
    template <class T>
    bool getValueOpt(const char* key, T& toset) const {
        auto it = settings.find(key);
        if (it == settings.end()) return false;
        std::istringstream is(it->second);
        is >> toset;
        return true;
    }
    const std::map<std::string, std::string>& getMap() const { return settings; }

private:
    std::map<std::string, std::string> settings;
};

// One persisted settings key of a calculator: how it is read from a SettingsMap into the object (returns true when the
// key was present and applied -- the calculator then marks itself dirty) and how it is written back.  The calculators
// keep one static table of these per class; setSettings / getSettings walk the table in order (keys that decide how
// later ones are interpreted -- e.g. the field mode before the field indices -- simply come first).
template <class Owner>
struct SettingBinding {
    const char* key;
    bool (*load)(Owner&, const SettingsMap&);
    void (*store)(const Owner&, SettingsMap&);  // null: not written back
};
template <class Owner, size_t N>
bool loadSettings(Owner& owner, const SettingBinding<Owner> (&table)[N], const SettingsMap& settings) {
    bool any = false;
    for (const auto& b : table) any |= b.load(owner, settings);
    return any;
}
template <class Owner, size_t N>
void storeSettings(const Owner& owner, const SettingBinding<Owner> (&table)[N], SettingsMap& settings) {
    for (const auto& b : table)
        if (b.store) b.store(owner, settings);
}

class VolumeData;

// The reference reports errors through sgl::Logfile::get()->throwError (logs, then throws); here: an exception.
struct CalculatorError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

class Calculator {
public:
    virtual ~Calculator() = default;
    virtual void initialize() {}
    void setCalculatorId(size_t id) { calculatorId = id; }
    size_t getCalculatorId() const { return calculatorId; }
    virtual bool getComputesCorrelation() const { return false; }
    virtual CalculatorType getCalculatorType() const = 0;
    virtual void setVolumeData(VolumeData* _volumeData, bool isNewData) { volumeData = _volumeData; (void)isNewData; }
    void setIsDirty() { dirty = true; }
    bool getIsDirty() { bool d = dirty; dirty = false; return d; }
    bool getIsDirtyDontReset() const { return dirty; }
    bool getHasNameChanged() { bool d = hasNameChanged; hasNameChanged = false; return d; }
    bool getHasFilterDeviceChanged() { bool d = hasFilterDeviceChanged; hasFilterDeviceChanged = false; return d; }

    virtual FieldType getOutputFieldType() { return FieldType::SCALAR; }
    virtual std::string getOutputFieldName() = 0;
    virtual FilterDevice getFilterDevice() = 0;
    virtual bool getHasFixedRange() const { return false; }
    virtual std::pair<float, float> getFixedRange() const { return {-1.0f, 1.0f}; }
    virtual void setSettings(const SettingsMap&) {}
    virtual void getSettings(SettingsMap&) {}

    /// Writes the derived data to the output data of size VolumeData::xs*ys*zs (Calculator.hpp:123-124).
    virtual void calculateCpu(int timeStepIdx, int ensembleIdx, float* buffer) {
        (void)timeStepIdx; (void)ensembleIdx; (void)buffer;
    }

protected:
    VolumeData* volumeData = nullptr;
    bool dirty = false;
    bool hasNameChanged = false;
    bool hasFilterDeviceChanged = false;
    size_t calculatorId = 0;
    size_t calculatorConstructorUseCount = 0;
};

typedef std::shared_ptr<Calculator> CalculatorPtr;

}  // namespace crfhost
