// host_adapter_test.cpp -- driver used by tests/test_host_adapter.py.
//   host_adapter_test settings                 CPU only: settings keys, defaults, naming, ranges, clamping
//   host_adapter_test compute <in.bin> <dir> [devices]
//                                              GPU: runs scripted scenarios through VolumeData::getFieldEntryCpu ->
//                                              CorrelationCalculator::calculateCpu and dumps the fields for the
//                                              Python side to compare with the oracle.
// in.bin: int32 xs, ys, zs, ts, es, nfields; then for field f, time t, member e: xs*ys*zs float32.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "CorrelationCalculator.hpp"
#include "EnsembleCalculators.hpp"
#include "NetCdfLoader.hpp"

using namespace crfhost;

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) {                                                               \
            std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            std::exit(1);                                                            \
        }                                                                            \
    } while (0)

static std::shared_ptr<VolumeData> makeVolume(int xs, int ys, int zs, int ts, int es, int nfields,
                                              const std::vector<float>* data) {
    auto vol = std::make_shared<VolumeData>(xs, ys, zs, ts, es);
    const size_t n = size_t(xs) * ys * zs;
    std::vector<float> zeros(n, 0.0f);
    size_t off = 0;
    for (int f = 0; f < nfields; f++)
        for (int t = 0; t < ts; t++)
            for (int e = 0; e < es; e++) {
                // the third field carries the one name the reference treats as divergent (VolumeData.cpp:616-621)
                vol->setFieldData(f == 0 ? "data" : f == 2 ? "Helicity" : "data" + std::to_string(f + 1), t, e,
                                  data ? data->data() + off : zeros.data());
                off += n;
            }
    return vol;
}

static int testSettings() {
    auto vol = makeVolume(32, 24, 16, 1, 64, 1, nullptr);
    auto calc = std::make_shared<CorrelationCalculator>(0);
    vol->addCalculator(calc);
    // defaults (CorrelationCalculator.hpp:206-212, CorrelationCalculator.cpp:104-110,592-598)
    CHECK(calc->getCorrelationMeasureType() == CorrelationMeasureType::MUTUAL_INFORMATION_KRASKOV);
    CHECK(calc->getOutputFieldName() == "Mutual Information (Kraskov)");
    CHECK((calc->getReferencePoint() == std::array<int, 3>{16, 12, 8}));
    CHECK(calc->getCorrelationMemberCount() == 64 && calc->getIsEnsembleMode());
    CHECK(calc->getKraskovNumNeighbors() == 2 && calc->getKraskovNumNeighborsMax() == 20);  // ceil(192/100), max(ceil(448/100),20)
    CHECK(calc->getFilterDevice() == FilterDevice::CPU && calc->getComputesCorrelation());
    CHECK(!calc->getHasFixedRange());
    // the canned state of src/Replicability/ReplicabilityState.hpp:40-60
    SettingsMap state{{"calculate_absolute_value", "0"}, {"correlation_measure_type", "pearson"},
                       {"correlation_mode", "Ensemble"}, {"data_mode", "Buffer Array"}, {"device", "CUDA"},
                       {"fix_picking_z", "1"}, {"kmi_neighbors", "30"}, {"kraskov_estimator_index", "1"},
                       {"mi_bins", "80"}, {"reference_point_x", "16"}, {"reference_point_y", "16"},
                       {"reference_point_z", "16"}, {"scalar_field_idx", "0"}, {"use_buffer_tiling", "1"},
                       {"use_separate_fields", "0"}};
    calc->setSettings(state);
    CHECK(calc->getIsDirtyDontReset());
    CHECK(calc->getOutputFieldName() == "Pearson Correlation");
    CHECK(calc->getHasFixedRange() && calc->getFixedRange() == std::make_pair(-1.0f, 1.0f));
    SettingsMap out;
    calc->getSettings(out);
    const auto& m = out.getMap();
    CHECK(m.at("correlation_measure_type") == "pearson" && m.at("device") == "CUDA");
    CHECK(m.at("kmi_neighbors") == "30" && m.at("mi_bins") == "80" && m.at("kraskov_estimator_index") == "1");
    CHECK(m.at("reference_point_x") == "16" && m.at("reference_point_y") == "16" && m.at("reference_point_z") == "16");
    CHECK(m.at("correlation_mode") == "Ensemble" && m.at("calculate_absolute_value") == "0");
    CHECK(m.at("scalar_field_idx") == "0" && m.at("correlation_field_mode") == "Single");
    // every measure id round-trips and names follow CorrelationCalculator.hpp:156-165
    const char* names[7] = {"Pearson Correlation", "Spearman Correlation", "Kendall Correlation",
                            "Mutual Information (Binned)", "Mutual Information (Kraskov)",
                            "Binned MI Correlation Coefficient", "KMI Correlation Coefficient"};
    for (int i = 0; i < 7; i++) {
        calc->setSettings(SettingsMap{{"correlation_measure_type", CORRELATION_MEASURE_TYPE_IDS[i]}});
        CHECK(calc->getOutputFieldName() == names[i]);
        SettingsMap o2;
        calc->getSettings(o2);
        CHECK(o2.getMap().at("correlation_measure_type") == CORRELATION_MEASURE_TYPE_IDS[i]);
        const bool fixed = i != 3 && i != 4;
        CHECK(calc->getHasFixedRange() == fixed);
        if (i >= 5) CHECK(calc->getFixedRange() == std::make_pair(0.0f, 1.0f));
    }
    calc->setSettings(SettingsMap{{"kraskov_estimator_index", "7"}, {"device", "CPU"}, {"calculate_absolute_value", "true"},
                                   {"correlation_measure_type", "kendall"}});
    SettingsMap o3;
    calc->getSettings(o3);
    CHECK(o3.getMap().at("kraskov_estimator_index") == "2");  // clamped to [1,2], CorrelationCalculator.cpp:765
    CHECK(o3.getMap().at("device") == "CPU" && o3.getMap().at("calculate_absolute_value") == "1");
    CHECK(calc->getFixedRange() == std::make_pair(0.0f, 1.0f));  // abs flag reports [0,1] (CorrelationCalculator.hpp:173-184)
    CHECK(!calc->getIsRealtime());
    calc->setSettings(SettingsMap{{"device", "SomethingElse"}});  // unknown -> accelerator
    SettingsMap o4;
    calc->getSettings(o4);
    CHECK(o4.getMap().at("device") == "Vulkan");
    // "devices" (this backend's key): absent from the state while it is the single default device, round-trips otherwise
    CHECK(o4.getMap().count("devices") == 0);
    calc->setSettings(SettingsMap{{"devices", "0,1,2,3"}});
    SettingsMap o5;
    calc->getSettings(o5);
    CHECK(o5.getMap().at("devices") == "0,1,2,3");
    calc->setSettings(SettingsMap{{"devices", "0"}});
    SettingsMap o6;
    calc->getSettings(o6);
    CHECK(o6.getMap().count("devices") == 0);
    // reference point clamps to the grid and marks dirty (CorrelationCalculator.cpp:190-202)
    (void)calc->getIsDirty();
    calc->setReferencePoint({100, -5, 3});
    CHECK((calc->getReferencePoint() == std::array<int, 3>{31, 0, 3}) && calc->getIsDirtyDontReset());
    // setReferencePointFromWorld (CorrelationCalculator.cpp:204-217): the rendering box of a 32 x 24 x 16 grid with unit
    // spacing is +-0.25 * (31, 23, 15) / 31 (VolumeData.cpp:322-330); its centre is the middle of the index range
    {
        const AABB3& b = vol->getBoundingBoxRendering();
        CHECK(b.min[0] == -0.25f && b.max[0] == 0.25f);
        CHECK(b.max[1] == (23.0f / 31.0f) * 0.25f && b.min[2] == -(15.0f / 31.0f) * 0.25f);
        (void)calc->getIsDirty();
        calc->setReferencePointFromWorld({0.0f, 0.0f, 0.0f});
        CHECK((calc->getReferencePoint() == std::array<int, 3>{16, 12, 8}));  // round(15.5), round(11.5), round(7.5): half away from zero
        CHECK(calc->getIsDirtyDontReset());
        (void)calc->getIsDirty();
        calc->setReferencePointFromWorld({0.0f, 0.0f, 0.0f});
        CHECK(!calc->getIsDirtyDontReset());                                    // unchanged point: not dirty
        calc->setReferencePointFromWorld({b.min[0], b.max[1], b.min[2]});
        CHECK((calc->getReferencePoint() == std::array<int, 3>{0, 23, 0}));
        calc->setReferencePointFromWorld({10.0f, -10.0f, b.max[2] * 0.5f});    // outside the box: clamped
        CHECK((calc->getReferencePoint() == std::array<int, 3>{31, 0, 11}));   // z: 0.75 * 15 = 11.25
        // anisotropic spacing changes the box, not the mapping of its corners
        vol->setGridExtent(1.0f, 2.0f, 0.5f);
        const AABB3& b2 = vol->getBoundingBoxRendering();
        CHECK(b2.max[1] == 0.25f && b2.max[0] == (31.0f / 46.0f) * 0.25f);
        calc->setReferencePointFromWorld({b2.max[0], b2.max[1], b2.max[2]});
        CHECK((calc->getReferencePoint() == std::array<int, 3>{31, 23, 15}));
        vol->setGridExtent(1.0f, 1.0f, 1.0f);
    }
    // divergent fields: getMinMaxScalarFieldValue centres the range at zero (VolumeData.cpp:616-621, 1661-1666)
    {
        VolumeData small(2, 2, 1, 1, 2);
        const float a[4] = {-1.0f, 0.5f, 2.0f, 0.0f}, b4[4] = {-3.5f, 0.0f, 1.0f, 1.5f};
        small.setFieldData("Helicity", 0, 0, a);
        small.setFieldData("Helicity", 0, 1, b4);
        small.setFieldData("data", 0, 0, a);
        CHECK(small.getIsScalarFieldDivergent("Helicity") && !small.getIsScalarFieldDivergent("data"));
        CHECK(small.getMinMaxScalarFieldValue("Helicity", 0, 0) == std::make_pair(-2.0f, 2.0f));
        CHECK(small.getMinMaxScalarFieldValue("Helicity", 0, 1) == std::make_pair(-3.5f, 3.5f));
        CHECK(small.getMinMaxScalarFieldValue("data", 0, 0) == std::make_pair(-1.0f, 2.0f));
    }
    // second calculator of the same type gets a numbered name
    auto calc2 = std::make_shared<CorrelationCalculator>(0);
    vol->addCalculator(calc2);
    CHECK(calc2->getOutputFieldName() == "Mutual Information (Kraskov) (2)");
    // time mode is selected automatically when there is one member and several time steps (:85-91)
    auto volT = makeVolume(8, 8, 4, 10, 1, 1, nullptr);
    auto calcT = std::make_shared<CorrelationCalculator>(0);
    volT->addCalculator(calcT);
    CHECK(!calcT->getIsEnsembleMode() && calcT->getCorrelationMemberCount() == 10);
    CHECK(calcT->getKraskovNumNeighbors() == 1);
    // ---- the sibling ensemble calculators: names, per-type numbering, settings keys, defaults
    auto volS = makeVolume(8, 8, 4, 1, 50, 1, nullptr);
    auto mean = std::make_shared<EnsembleMeanCalculator>(0);
    auto spread = std::make_shared<EnsembleSpreadCalculator>(0);
    auto pred = std::make_shared<SetPredicateCalculator>(0);
    auto dkl = std::make_shared<DKLCalculator>(0);
    auto dkl2 = std::make_shared<DKLCalculator>(0);
    for (CalculatorPtr c : {CalculatorPtr(mean), CalculatorPtr(spread), CalculatorPtr(pred), CalculatorPtr(dkl), CalculatorPtr(dkl2)})
        volS->addCalculator(c);
    CHECK(mean->getOutputFieldName() == "Ensemble Mean" && spread->getOutputFieldName() == "Ensemble Variance");
    CHECK(pred->getOutputFieldName() == "Set Predicate" && dkl->getOutputFieldName() == "KL-Divergence");
    CHECK(dkl2->getOutputFieldName() == "KL-Divergence (2)");  // numbered per calculator type (VolumeData.cpp:2283-2285)
    CHECK(pred->getCountLower() == 25 && pred->getCountUpper() == 25);  // cs / 2 (SetPredicateCalculator.cpp:143-144)
    CHECK(dkl->getNumNeighbors() == 2);                                  // max(ceil(3 * 50 / 100), 1)
    pred->setSettings(SettingsMap{{"comparison_operator_type", "<="}, {"comparison_value", "0.5"}, {"count_lower", "10"},
                                   {"count_upper", "40"}});
    SettingsMap ps;
    pred->getSettings(ps);
    CHECK(ps.getMap().at("comparison_operator_type") == "<=" && ps.getMap().at("count_lower") == "10" &&
          ps.getMap().at("count_upper") == "40" && ps.getMap().at("comparison_value") == "0.5" &&
          ps.getMap().at("correlation_mode") == "Ensemble" && ps.getMap().at("scalar_field_idx") == "0");
    dkl->setSettings(SettingsMap{{"estimator_type", "Binning"}, {"mi_bins", "32"}, {"knn_neighbors", "5"}});
    SettingsMap ds;
    dkl->getSettings(ds);
    CHECK(ds.getMap().at("estimator_type") == "Binning" && ds.getMap().at("mi_bins") == "32" &&
          ds.getMap().at("knn_neighbors") == "5");
    CHECK(pred->getHasFixedRange() && pred->getFixedRange() == std::make_pair(0.0f, 1.0f));
    std::puts("SETTINGS-OK");
    return 0;
}

static void dump(const std::string& path, const float* v, size_t n) {
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(v), std::streamsize(n * sizeof(float)));
}

static int testCompute(const char* inPath, const std::string& outDir, const char* devices) {
    std::ifstream f(inPath, std::ios::binary);
    int32_t h[6];
    f.read(reinterpret_cast<char*>(h), sizeof h);
    const int xs = h[0], ys = h[1], zs = h[2], ts = h[3], es = h[4], nf = h[5];
    const size_t n = size_t(xs) * ys * zs;
    std::vector<float> data(n * size_t(ts) * size_t(es) * size_t(nf));
    f.read(reinterpret_cast<char*>(data.data()), std::streamsize(data.size() * sizeof(float)));
    CHECK(bool(f));
    auto vol = makeVolume(xs, ys, zs, ts, es, nf, &data);
    auto calc = std::make_shared<CorrelationCalculator>(0);
    vol->addCalculator(calc);
    // optional: spread the grid over a device group ("0,0" = two z-slabs rehearsed on one GPU)
    if (devices) calc->setSettings(SettingsMap{{"devices", devices}});
    auto eval = [&](const std::string& tag, int t, int e) {
        vol->updateCalculators();
        HostCacheEntry entry = vol->getFieldEntryCpu(FieldType::SCALAR, calc->getOutputFieldName(), t, e);
        CHECK(entry->getNumEntries() == n);
        dump(outDir + "/" + tag + ".bin", entry->data<float>(), n);
        // a second request is served from the host cache: same buffer, no recomputation
        CHECK(vol->getFieldEntryCpu(FieldType::SCALAR, calc->getOutputFieldName(), t, e).get() == entry.get());
    };
    const int tFixed = ts > 1 && es > 1 ? 1 : 0;
    for (int i = 0; i < 7; i++) {  // every measure at the default reference point (grid centre), default k
        calc->setSettings(SettingsMap{{"correlation_measure_type", CORRELATION_MEASURE_TYPE_IDS[i]}});
        eval(std::string("m_") + CORRELATION_MEASURE_TYPE_IDS[i], tFixed, 0);
    }
    calc->setSettings(SettingsMap{{"correlation_measure_type", "pearson"}});
    calc->setReferencePoint({1, 2, 3});
    eval("pearson_ref123", tFixed, 0);
    calc->setSettings(SettingsMap{{"correlation_measure_type", "mi_kraskov"}, {"kmi_neighbors", "3"},
                                   {"kraskov_estimator_index", "2"}});
    eval("kraskov2_k3_ref123", tFixed, 0);
    if (nf > 1) {  // SEPARATE mode: reference vector from the second field (+ time lag when there are time steps)
        calc->setSettings(SettingsMap{{"correlation_measure_type", "spearman"}, {"correlation_field_mode", "Separate"},
                                       {"scalar_field_idx_ref", "1"}, {"scalar_field_idx_query", "0"}});
        eval("spearman_separate", tFixed, 0);
        calc->setSettings(SettingsMap{{"correlation_measure_type", "mi_binned"}});
        eval("binned_separate", tFixed, 0);
        if (ts > 1) {
            calc->setSettings(SettingsMap{{"correlation_measure_type", "pearson"}, {"use_time_lag_correlations", "1"},
                                           {"time_lag_time_step_idx", "0"}});
            eval("pearson_separate_lag0", tFixed, 0);
        }
    }
    if (nf > 1) {  // SEPARATE_SYMMETRIC: field 1 vs field 2 voxel by voxel
        calc->setSettings(SettingsMap{{"correlation_measure_type", "kendall"}, {"use_time_lag_correlations", "0"},
                                       {"correlation_field_mode", "Separate Symmetric"}});
        eval("kendall_symmetric", tFixed, 0);
        calc->setSettings(SettingsMap{{"correlation_measure_type", "mi_binned"}});
        eval("binned_symmetric", tFixed, 0);
    }
    if (nf > 2) {  // a divergent field: the binned-MI range is centred at zero (VolumeData.cpp:1661-1666)
        calc->setSettings(SettingsMap{{"correlation_measure_type", "mi_binned"}, {"correlation_field_mode", "Single"}});
        calc->setSettings(SettingsMap{{"scalar_field_idx", "2"}});
        calc->setReferencePoint({1, 2, 3});
        eval("binned_helicity", tFixed, 0);
        calc->setSettings(SettingsMap{{"scalar_field_idx", "0"}});
    }
    {  // the sibling ensemble calculators on the first field
        auto mean = std::make_shared<EnsembleMeanCalculator>(0);
        auto spread = std::make_shared<EnsembleSpreadCalculator>(0);
        auto pred = std::make_shared<SetPredicateCalculator>(0);
        auto dklK = std::make_shared<DKLCalculator>(0);
        auto dklB = std::make_shared<DKLCalculator>(0);
        for (CalculatorPtr c : {CalculatorPtr(mean), CalculatorPtr(spread), CalculatorPtr(pred), CalculatorPtr(dklK), CalculatorPtr(dklB)})
            vol->addCalculator(c);
        pred->setSettings(SettingsMap{{"comparison_operator_type", ">"}, {"comparison_value", "0.25"}});
        dklB->setSettings(SettingsMap{{"estimator_type", "Binning"}, {"mi_bins", "16"}});
        vol->updateCalculators();
        const char* tags[5] = {"ensemble_mean", "ensemble_spread", "set_predicate", "dkl_knn", "dkl_binned"};
        CalculatorPtr calcs[5] = {mean, spread, pred, dklK, dklB};
        for (int i = 0; i < 5; i++) {
            HostCacheEntry entry = vol->getFieldEntryCpu(FieldType::SCALAR, calcs[i]->getOutputFieldName(), tFixed, 0);
            dump(outDir + "/" + tags[i] + ".bin", entry->data<float>(), n);
        }
    }
    std::printf("COMPUTE-OK kernel_ms=%.4f\n", calc->getLastKernelTimeMs());
    return 0;
}

// netcdf <file> <outdir>: CPU only -- parses the file, prints the metadata, dumps every (field, t, e) volume.
// netcdf_compute <file> <outdir>: GPU -- the file's first field through CorrelationCalculator (Pearson, grid centre).
static int testNetCdf(const char* file, const std::string& outDir, bool compute) {
    NetCdfLoader loader(file);
    std::printf("grid %d %d %d ts %d es %d\n", loader.getGridSizeX(), loader.getGridSizeY(), loader.getGridSizeZ(),
                loader.getTimeStepCount(), loader.getEnsembleMemberCount());
    for (const auto& name : loader.getFieldNames()) std::printf("field %s\n", name.c_str());
    for (const auto& w : loader.getWarnings()) std::printf("warning %s\n", w.c_str());
    const size_t n = size_t(loader.getGridSizeX()) * loader.getGridSizeY() * loader.getGridSizeZ();
    if (!compute) {
        std::vector<float> buffer(n);
        for (const auto& name : loader.getFieldNames())
            for (int t = 0; t < loader.getTimeStepCount(); t++)
                for (int e = 0; e < loader.getEnsembleMemberCount(); e++) {
                    loader.getFieldEntry(name, t, e, buffer.data());
                    dump(outDir + "/" + name + "_t" + std::to_string(t) + "_e" + std::to_string(e) + ".bin", buffer.data(), n);
                }
        std::puts("NETCDF-OK");
        return 0;
    }
    auto vol = loader.createVolumeData();
    auto calc = std::make_shared<CorrelationCalculator>(0);
    vol->addCalculator(calc);
    calc->setSettings(SettingsMap{{"correlation_measure_type", "pearson"}});
    vol->updateCalculators();
    HostCacheEntry entry = vol->getFieldEntryCpu(FieldType::SCALAR, calc->getOutputFieldName(), 0, 0);
    dump(outDir + "/pearson.bin", entry->data<float>(), n);
    std::puts("NETCDF-COMPUTE-OK");
    return 0;
}

int main(int argc, char** argv) {
    try {
        if (argc >= 2 && std::string(argv[1]) == "settings") return testSettings();
        if (argc >= 4 && std::string(argv[1]) == "compute") return testCompute(argv[2], argv[3], argc >= 5 ? argv[4] : nullptr);
        if (argc >= 4 && std::string(argv[1]) == "netcdf") return testNetCdf(argv[2], argv[3], false);
        if (argc >= 4 && std::string(argv[1]) == "netcdf_compute") return testNetCdf(argv[2], argv[3], true);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 2;
    }
    std::fprintf(stderr, "usage: host_adapter_test settings | compute <in.bin> <outdir>\n");
    return 64;
}
