/*
 * corr_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C++17 CPU restatement of the reference's per-voxel ensemble correlation path
 * (chrismile/Correrender, CorrelationCalculator::calculateCpu and the estimator primitives it calls).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker.  The product (correrender_amd/, libcorrfield.so) never links or calls it.
 *
 * Reference locations restated here (paths relative to /root/reference):
 *   src/Calculators/Correlation.cpp:100-133   computePearson2<float> (field-gather form)
 *   src/Calculators/Correlation.cpp:141-174   computePearson2<float> (two dense arrays)
 *   src/Calculators/Correlation.cpp:277-303   computeRanks (fractional ranks)
 *   src/Calculators/Correlation.cpp:305-329   computeTiesB
 *   src/Calculators/Correlation.cpp:368-455   computeKendall<int32_t> (tau-b, joint ties ignored)
 *   src/Calculators/MutualInformation.cpp:45-143   computeMutualInformationBinned<double>
 *   src/Calculators/MutualInformation.cpp:159-259  averageDigamma<double, includeCenter>
 *   src/Calculators/MutualInformation.cpp:399-444  computeMutualInformationKraskov<double>  (KSG-1)
 *   src/Calculators/MutualInformation.cpp:449-509  computeMutualInformationKraskov2<double> (KSG-2)
 *   src/Calculators/CorrelationCalculator.cpp:781-1154  calculateCpu (driver loop, NaN rule, cs==1 rule,
 *                                                       binned normalisation, MI-CC post-map)
 *   src/Loaders/DataSet.hpp:37                 IDXS(x,y,z) = z*xs*ys + y*xs + x
 *   src/Calculators/EnsembleMeanCalculator.cpp:110-134, EnsembleSpreadCalculator.cpp:110-145   oracle_ensemble_stat
 *   src/Renderers/Diagram/HEBChartCorrelation.cpp:493-600  pair evaluation of HEBChart::computeCorrelations (the CPU twin
 *                                                       of the request mode): oracle_pair_requests
 *
 * PINNING STATUS
 *   Pearson / Spearman / Kendall: PINNED.  oracle/Makefile compiles the reference's own Correlation.cpp into
 *     oracle/_ref/libref_corr.so and tests/test_oracle_vs_ref.py + oracle/make_golden.py check this file
 *     bit-for-bit against it; the reference outputs are committed under tests/golden/.
 *   binned MI / Kraskov MI: PARITY UNPINNED against reference object code.  MutualInformation.cpp needs
 *     boost::math::digamma, sgl::KdTreed, sgl::XorshiftRandomGenerator and glm, none of which are in
 *     /root/reference or this image, and the reference holds no tests/golden vectors for the path.  These two
 *     estimators are a restatement of the published algorithm anchored on the reference's call sites:
 *       - boost::math::digamma is only ever evaluated at positive integers (MutualInformation.cpp:235,237,
 *         438,439,503,504): psi(n) = -gamma + H_{n-1}, tabulated here in long double.
 *       - sgl::KdTreed<double,2,CHEBYSHEV>::findKNearestNeighbors is an exact k-NN search
 *         (MutualInformation.cpp:426-434): any exact search yields the same distances; brute force here.
 *       - sgl::XorshiftRandomGenerator (github.com/chrismile/sgl, cloned unpinned at HEAD by build.sh:1039-1043)
 *         is NOT available: the 1e-10 tie-breaking noise stream is this repo's own documented xorshift32
 *         stream (see noise01()).  On tie-free data the noise cannot change any neighbour count, so results
 *         are independent of the stream; on exact ties they are stream-dependent (documented in DESIGN.md).
 *     They are cross-checked in tests/ against independent numpy/scipy formulations and analytic values.
 *
 * Build: see oracle/Makefile (g++ -std=c++17 -O2 -ffp-contract=off -fopenmp; no -march, no fast-math, mirroring
 * the reference's CMakeLists.txt:13,34-36 so fp32 semantics are those of the reference build).
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <utility>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---------------------------------------------------------------------------------------------------------
// Pearson, "formula 2", fp32, three strictly sequential passes (Correlation.cpp:100-133 and :141-174).
// The reference has two overloads with identical arithmetic; they differ only in how y is fetched.
// ---------------------------------------------------------------------------------------------------------
template <class FetchY>
inline float pearson2_f32(const float* x, int n, FetchY fetchY) {
    const float nf = float(n);
    const float invN = 1.0f / nf;
    float meanX = 0.0f, meanY = 0.0f;
    for (int e = 0; e < n; e++) {
        meanX += invN * x[e];
        meanY += invN * fetchY(e);
    }
    const float invNm1 = 1.0f / (nf - 1.0f);
    float varX = 0.0f, varY = 0.0f;
    for (int e = 0; e < n; e++) {
        const float dx = x[e] - meanX;
        const float dy = fetchY(e) - meanY;
        varX += invNm1 * dx * dx;  // parsed (invNm1*dx)*dx, as in the reference
        varY += invNm1 * dy * dy;
    }
    const float sdX = std::sqrt(varX);
    const float sdY = std::sqrt(varY);
    float r = 0.0f;
    for (int e = 0; e < n; e++) {
        r += invNm1 * ((x[e] - meanX) / sdX) * ((fetchY(e) - meanY) / sdY);
    }
    return r;
}

// ---------------------------------------------------------------------------------------------------------
// Fractional ("mid") ranks, 1-based (Correlation.cpp:277-303).  Result is independent of the sort algorithm.
// ---------------------------------------------------------------------------------------------------------
struct RankScratch {
    std::vector<std::pair<float, int>> order;
};

void fractional_ranks(const float* v, float* ranks, int n, RankScratch& s) {
    s.order.resize(size_t(n));
    for (int i = 0; i < n; i++) s.order[size_t(i)] = {v[i], i};
    std::sort(s.order.begin(), s.order.end());
    float firstRankOfRun = 1.0f;
    int i = 0;
    while (i < n) {
        int j = i + 1;
        while (j < n && s.order[size_t(j)].first == s.order[size_t(i)].first) j++;
        const int m = j - i;
        const float r = firstRankOfRun + float(m - 1) * 0.5f;
        for (int t = i; t < j; t++) ranks[s.order[size_t(t)].second] = r;
        firstRankOfRun += float(m);
        i = j;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Kendall tau-b with int32 counters (Correlation.cpp:423-455).  S_y = number of strict inversions of the y
// sequence after sorting the (x, y) pairs lexicographically; the reference counts them with an iterative
// top-down merge (Correlation.cpp:368-421), here a plain bottom-up merge count -- the integer is the same.
// n1/n2 = sum over runs of equal x / equal y of t(t-1)/2 (Correlation.cpp:305-329); joint ties n3 := 0.
// ---------------------------------------------------------------------------------------------------------
struct KendallScratch {
    std::vector<std::pair<float, float>> joint;
    std::vector<float> a, b, t;
};

int32_t tie_pairs(const float* v, int n, std::vector<float>& tmp) {
    tmp.assign(v, v + n);
    std::sort(tmp.begin(), tmp.end());
    int32_t ties = 0;
    int i = 0;
    while (i < n) {
        int j = i + 1;
        while (j < n && tmp[size_t(j)] == tmp[size_t(i)]) j++;
        const int32_t m = j - i;
        ties += m * (m - 1) / 2;
        i = j;
    }
    return ties;
}

int32_t strict_inversions(std::vector<float>& a, std::vector<float>& b, int n) {
    // bottom-up merge sort of a[0..n), counting pairs i<j with a[j] < a[i] (equal values are not inversions).
    int32_t inv = 0;
    b.resize(size_t(n));
    for (int w = 1; w < n; w *= 2) {
        for (int lo = 0; lo < n; lo += 2 * w) {
            const int mid = std::min(lo + w, n), hi = std::min(lo + 2 * w, n);
            int i = lo, j = mid, o = lo;
            while (i < mid && j < hi) {
                if (a[size_t(j)] < a[size_t(i)]) {
                    inv += int32_t(mid - i);
                    b[size_t(o++)] = a[size_t(j++)];
                } else {
                    b[size_t(o++)] = a[size_t(i++)];
                }
            }
            while (i < mid) b[size_t(o++)] = a[size_t(i++)];
            while (j < hi) b[size_t(o++)] = a[size_t(j++)];
        }
        a.swap(b);
    }
    return inv;
}

float kendall_tau_b_i32(const float* x, const float* y, int n, KendallScratch& s) {
    s.joint.resize(size_t(n));
    for (int i = 0; i < n; i++) s.joint[size_t(i)] = {x[i], y[i]};
    std::sort(s.joint.begin(), s.joint.end());
    s.a.resize(size_t(n));
    for (int i = 0; i < n; i++) s.a[size_t(i)] = s.joint[size_t(i)].second;
    const int32_t Sy = strict_inversions(s.a, s.b, n);
    const int32_t nn = int32_t(n);
    const int32_t n0 = (nn * (nn - 1)) / 2;
    const int32_t n1 = tie_pairs(x, n, s.t);
    const int32_t n2 = tie_pairs(y, n, s.t);
    const int32_t n3 = 0;
    const int32_t numerator = n0 - n1 - n2 + n3 - 2 * Sy;
    const float denominator = std::sqrt(float(n0 - n1)) * std::sqrt(float(n0 - n2));
    return float(numerator) / denominator;
}

// ---------------------------------------------------------------------------------------------------------
// Binned mutual information, fp64 histograms (MutualInformation.cpp:45-143).  Inputs already normalised.
// ---------------------------------------------------------------------------------------------------------
struct BinnedScratch {
    std::vector<double> hx, hy, hxy;
};

float mi_binned_f64(const float* x01, const float* y01, int numBins, int n, BinnedScratch& s) {
    const size_t nb = size_t(numBins);
    s.hx.assign(nb, 0.0);
    s.hy.assign(nb, 0.0);
    s.hxy.assign(nb * nb, 0.0);
    for (int e = 0; e < n; e++) {
        const double vx = x01[e], vy = y01[e];
        if (!std::isnan(vx) && !std::isnan(vy)) {
            // int(t) as the reference's x86-64 build converts it (cvttsd2si: INT_MIN outside the int range) -- spelled
            // out instead of relying on undefined behaviour; only reachable with extrema far narrower than the data
            auto to_int = [](double t) { return (t > -2147483649.0 && t < 2147483648.0) ? int(t) : std::numeric_limits<int>::min(); };
            const int bx = std::clamp(to_int(vx * double(numBins)), 0, numBins - 1);
            const int by = std::clamp(to_int(vy * double(numBins)), 0, numBins - 1);
            s.hxy[size_t(bx) * nb + size_t(by)] += 1.0;
        }
    }
    double total = 0.0;
    for (size_t i = 0; i < nb * nb; i++) total += s.hxy[i];
    for (size_t i = 0; i < nb * nb; i++) s.hxy[i] /= total;
    for (size_t bx = 0; bx < nb; bx++) {
        for (size_t by = 0; by < nb; by++) {
            s.hx[bx] += s.hxy[bx * nb + by];
            s.hy[by] += s.hxy[bx * nb + by];
        }
    }
    const double eps1 = 0.5 / double(n);
    const double eps2 = 0.5 / double(n * n);  // int product, as in the reference
    double mi = 0.0;
    for (size_t b = 0; b < nb; b++) {
        const double px = s.hx[b], py = s.hy[b];
        if (px > eps1) mi -= px * std::log(px);
        if (py > eps1) mi -= py * std::log(py);
    }
    for (size_t i = 0; i < nb * nb; i++) {
        const double pxy = s.hxy[i];
        if (pxy > eps2) mi += pxy * std::log(pxy);
    }
    return float(mi);
}

// ---------------------------------------------------------------------------------------------------------
// Kraskov-Stoegbauer-Grassberger kNN estimators in fp64.
// ---------------------------------------------------------------------------------------------------------

// psi at non-negative integers.  boost::math::digamma(int) at the reference's call sites is the digamma
// function at a positive integer, i.e. -gamma + H_{n-1}; psi(0) is a pole (boost raises), NaN here.
struct DigammaTable {
    std::vector<double> v;
    void ensure(int nmax) {
        if (int(v.size()) > nmax) return;
        v.resize(size_t(nmax) + 1);
        const long double gamma = 0.577215664901532860606512090082402431L;
        long double h = 0.0L;
        v[0] = std::numeric_limits<double>::quiet_NaN();
        for (int n = 1; n <= nmax; n++) {
            v[size_t(n)] = double(h - gamma);
            h += 1.0L / (long double)n;
        }
    }
};

// This repo's documented tie-breaking noise stream (stands in for sgl::XorshiftRandomGenerator, which is not
// available): Marsaglia xorshift32 (13,17,5) seeded with the low 32 bits of the reference's seed constants
// (617406168 for the reference vector, 864730169 for the query vector, MutualInformation.cpp:410-411), one
// draw per member, u = (state >> 8) * 2^-24 in [0,1) as a float; noise = double(u) * 1e-10.
struct Xorshift32 {
    uint32_t s;
    explicit Xorshift32(uint32_t seed) : s(seed ? seed : 0x9E3779B9u) {}
    float next01() {
        s ^= s << 13;
        s ^= s >> 17;
        s ^= s << 5;
        return float(s >> 8) * (1.0f / 16777216.0f);
    }
};
constexpr uint32_t SEED_REF = 617406168u;
constexpr uint32_t SEED_QUERY = 864730169u;
constexpr double NOISE_SCALE = 1e-10;   // default_epsilon<double>::noise, MutualInformation.cpp:165
constexpr double COUNT_SLACK = 1e-15;   // default_epsilon<double>::value, MutualInformation.cpp:163

struct KraskovScratch {
    std::vector<double> px, py, d, dx, dy, sorted, tmp;
    DigammaTable psi;
};

// Optional replacement of the noise tables (oracle_set_kraskov_noise): noise values in [0, 1e-10) per member, e.g. the
// stream of the reference's own generator produced by a build that has sgl.  Read-only while fields are evaluated.
std::vector<double> g_noise_override[2];

void noisy_coords(const float* v, int n, uint32_t seed, std::vector<double>& out) {
    out.resize(size_t(n));
    const std::vector<double>& over = g_noise_override[seed == SEED_REF ? 0 : 1];
    if (int(over.size()) >= n) {
        for (int e = 0; e < n; e++) out[size_t(e)] = double(v[e]) + over[size_t(e)];
        return;
    }
    Xorshift32 g(seed);
    for (int e = 0; e < n; e++) out[size_t(e)] = double(v[e]) + double(g.next01()) * NOISE_SCALE;
}

// mean over e of psi(count_e) (includeCenter) or psi(count_e - 1); count_e = #{j : c-r <= v_j < c+r},
// r = dist_e -/+ 1e-15, at least 1 (MutualInformation.cpp:167-259, binary-search variant).
template <bool includeCenter>
double average_digamma(const std::vector<double>& coord, const std::vector<double>& dist, int n, KraskovScratch& s) {
    s.sorted = coord;
    std::sort(s.sorted.begin(), s.sorted.end());
    const double factor = 1.0 / double(n);
    double mean = 0.0;
    for (int e = 0; e < n; e++) {
        const double c = coord[size_t(e)];
        const double r = includeCenter ? dist[size_t(e)] - COUNT_SLACK : dist[size_t(e)] + COUNT_SLACK;
        const double lo = c - r, hi = c + r;
        const auto itLo = std::lower_bound(s.sorted.begin(), s.sorted.end(), lo);
        const auto itHi = std::lower_bound(itLo, s.sorted.end(), hi);
        const int count = std::max(int(itHi - itLo), 1);
        mean += factor * s.psi.v[size_t(includeCenter ? count : count - 1)];
    }
    return mean;
}

float mi_kraskov_f64(const float* x, const float* y, int k, int n, int estimator, KraskovScratch& s) {
    s.psi.ensure(std::max(n, k) + 1);
    noisy_coords(x, n, SEED_REF, s.px);
    noisy_coords(y, n, SEED_QUERY, s.py);
    const int kk = std::min(k, n - 1);  // (k+1)-th smallest incl. self; a kd-tree returns at most n points
    s.d.resize(size_t(n));
    s.dx.resize(size_t(n));
    s.dy.resize(size_t(n));
    s.tmp.resize(size_t(n));
    std::vector<std::pair<double, int>> cand;
    for (int e = 0; e < n; e++) {
        if (estimator == 1) {
            for (int j = 0; j < n; j++) {
                s.tmp[size_t(j)] = std::max(std::abs(s.px[size_t(e)] - s.px[size_t(j)]),
                                            std::abs(s.py[size_t(e)] - s.py[size_t(j)]));
            }
            std::nth_element(s.tmp.begin(), s.tmp.begin() + kk, s.tmp.end());
            s.d[size_t(e)] = s.tmp[size_t(kk)];
        } else {
            cand.resize(size_t(n));
            for (int j = 0; j < n; j++) {
                cand[size_t(j)] = {std::max(std::abs(s.px[size_t(e)] - s.px[size_t(j)]),
                                            std::abs(s.py[size_t(e)] - s.py[size_t(j)])), j};
            }
            std::partial_sort(cand.begin(), cand.begin() + kk + 1, cand.end());
            double ex = std::numeric_limits<double>::lowest(), ey = ex;
            for (int t = 0; t <= kk; t++) {
                const int j = cand[size_t(t)].second;
                ex = std::max(ex, std::abs(s.px[size_t(e)] - s.px[size_t(j)]));
                ey = std::max(ey, std::abs(s.py[size_t(e)] - s.py[size_t(j)]));
            }
            s.dx[size_t(e)] = ex;
            s.dy[size_t(e)] = ey;
        }
    }
    double a, b, c;
    if (estimator == 1) {
        a = average_digamma<true>(s.px, s.d, n, s);
        b = average_digamma<true>(s.py, s.d, n, s);
        c = s.psi.v[size_t(k)];
    } else {
        a = average_digamma<false>(s.px, s.dx, n, s);
        b = average_digamma<false>(s.py, s.dy, n, s);
        c = s.psi.v[size_t(k)] - 1.0 / double(k);
    }
    const double d = s.psi.v[size_t(n)];
    const double mi = -a - b + c + d;
    return std::max(float(mi), 0.0f);
}

inline float mi_to_cc(float mi) {  // CorrelationCalculator.cpp:1071-1073,1130-1132 (fp32)
    return std::sqrt(1.0f - std::exp(-2.0f * mi));
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// DKLCalculator: Kullback-Leibler divergence between a voxel's normalised ensemble distribution and N(0,1)
// (src/Calculators/DKL.cpp:38-165, driver src/Calculators/DKLCalculator.cpp:134-262, Real = double).  DKL.cpp needs
// boost::math::digamma and sgl's Math.hpp (PI, TWO_PI, sqr, iceil) and cannot be compiled here: restated.  PARITY
// UNPINNED for the two sgl constants: sgl declares PI / TWO_PI as `float` (so `std::log(sgl::TWO_PI)` is the float
// overload); should a build use double constants the results move by < 1e-7 absolute, inside the stated tolerance.
// ---------------------------------------------------------------------------------------------------------------
static const float SGL_PI = 3.1415926535897932f;
static const float SGL_TWO_PI = SGL_PI * 2.0f;

// computeDKLBinned<double> (DKL.cpp:38-84); overwrites v with the normalised values like the reference
static float dkl_binned_f64(float* v, int numBins, int es, std::vector<double>& hist) {
    const double factor = 1.0 / double(es);
    double mean = 0, variance = 0;
    for (int c = 0; c < es; c++) mean += factor * v[c];
    for (int c = 0; c < es; c++) {
        const double diff = mean - v[c];
        variance += factor * diff * diff;
    }
    const double stdev = std::sqrt(variance);
    double minVal = std::numeric_limits<double>::max(), maxVal = std::numeric_limits<double>::lowest();
    for (int c = 0; c < es; c++) {
        const double val = (v[c] - mean) / stdev;
        v[c] = float(val);
        minVal = std::min(minVal, val);
        maxVal = std::max(maxVal, val);
    }
    minVal -= 0.01;
    maxVal += 0.01;
    const double binFactor = double(numBins) / (maxVal - minVal);
    const double binFactorInv = (maxVal - minVal) / double(numBins);
    hist.assign(size_t(numBins), 0.0);
    for (int c = 0; c < es; c++) {
        const double t = (v[c] - minVal) * binFactor;
        // int(NaN) is INT_MIN on x86-64 (cvttsd2si) and clamps to bin 0 -- spelled out instead of relying on UB
        const int raw = std::isnan(t) ? std::numeric_limits<int>::min()
                                      : (t >= 2147483648.0 ? std::numeric_limits<int>::min()
                                                           : (t <= -2147483649.0 ? std::numeric_limits<int>::min() : int(t)));
        hist[size_t(std::clamp(raw, 0, numBins - 1))] += 1.0;
    }
    double dkl = 0;
    for (int b = 0; b < numBins; b++) {
        if (hist[size_t(b)] > 0) {
            const double px = hist[size_t(b)] / double(es);
            const double center = (double(b) + 0.5) * binFactorInv + minVal;
            dkl += std::log(px * binFactor / (std::sqrt(0.5 / double(SGL_PI)) * std::exp(-0.5 * (center * center)))) * px;
        }
    }
    if (std::isinf(dkl)) return std::numeric_limits<float>::quiet_NaN();
    return float(dkl);
}

// findKNearestNeighbors<float> (DKL.cpp:98-130): distance from data[i] to the farther end of a window of k+1
// consecutive sorted values containing i, the window placed by a step-halving descent (NOT an exhaustive search).
// `data` is the float array, so the template deduces Real = float even inside computeDKLKNNEstimate<double>.
static float dkl_window_distance(const float* data, int N, int k, int i) {
    const float big = std::numeric_limits<float>::max();
    int step = (k + 1) / 2;  // sgl::iceil(k, 2)
    int l = std::max(i - step, 0);
    int r = l + k;
    if (r >= N) {
        r = N - 1;
        l = N - k - 1;
    }
    const float vc = data[i];
    while (true) {
        const float vl = l >= 0 ? vc - data[l] : big;
        const float vr = r < N ? data[r] - vc : big;
        const float diff = std::max(vl, vr);
        const float vl0 = (l - step <= i && l - step >= 0) ? vc - data[l - step] : big;
        const float vr0 = (r - step >= i && r - step < N) ? data[r - step] - vc : big;
        const float diff0 = std::max(vl0, vr0);
        const float vl1 = (l + step <= i && l + step >= 0) ? vc - data[l + step] : big;
        const float vr1 = (r + step >= i && r + step < N) ? data[r + step] - vc : big;
        const float diff1 = std::max(vl1, vr1);
        const float d01 = diff0 - diff1;
        int dir = d01 > 0.0f ? 1 : (d01 < 0.0f ? -1 : 0);
        if (diff < diff0 && diff < diff1) dir = 0;
        l += dir * step;
        r += dir * step;
        if (step == 1) break;
        step = (step + 1) / 2;
    }
    return std::max(vc - data[l], data[r] - vc);
}

// computeDKLKNNEstimate<double> (DKL.cpp:132-165): Kozachenko-Leonenko entropy estimate of the normalised sample
static float dkl_knn_f64(float* v, int k, int es, const DigammaTable& psi) {
    const double factor = 1.0 / double(es);
    double mean = 0, variance = 0;
    for (int c = 0; c < es; c++) mean += factor * v[c];
    for (int c = 0; c < es; c++) {
        const double diff = mean - v[c];
        variance += factor * diff * diff;
    }
    const double stdev = std::sqrt(variance);
    bool anyNan = false;
    for (int c = 0; c < es; c++) {
        v[c] = float((v[c] - mean) / stdev);
        anyNan = anyNan || std::isnan(v[c]);
    }
    if (anyNan) return std::numeric_limits<float>::quiet_NaN();  // stdev == 0: every later term is NaN (no sort of NaNs)
    std::sort(v, v + es);
    double entropy = 0, secondMoment = 0;
    for (int c = 0; c < es; c++) {
        const double nnDist = dkl_window_distance(v, es, k, c);
        entropy += factor * std::log(nnDist);
        const double value = v[c];
        secondMoment += factor * value * value;
    }
    entropy += psi.v[size_t(es)] - psi.v[size_t(k)] + std::log(2.0);
    float dkl = float(-entropy + 0.5 * double(std::log(SGL_TWO_PI)) + 0.5 * secondMoment);
    if (std::isinf(dkl)) return std::numeric_limits<float>::quiet_NaN();
    return std::max(dkl, 0.0f);
}

extern "C" {

enum {
    ORACLE_PEARSON = 0, ORACLE_SPEARMAN = 1, ORACLE_KENDALL = 2, ORACLE_MI_BINNED = 3, ORACLE_MI_KRASKOV = 4,
    ORACLE_BINNED_MI_CC = 5, ORACLE_KMI_CC = 6
};

// --- primitives on two dense arrays -----------------------------------------------------------------------
float oracle_pearson2(const float* x, const float* y, int n) {
    return pearson2_f32(x, n, [y](int e) { return y[e]; });
}
void oracle_ranks(const float* v, float* ranks, int n) {
    RankScratch s;
    fractional_ranks(v, ranks, n, s);
}
float oracle_spearman(const float* x, const float* y, int n) {
    RankScratch s;
    std::vector<float> rx((size_t)n), ry((size_t)n);
    fractional_ranks(x, rx.data(), n, s);
    fractional_ranks(y, ry.data(), n, s);
    return oracle_pearson2(rx.data(), ry.data(), n);
}
float oracle_kendall(const float* x, const float* y, int n) {
    KendallScratch s;
    return kendall_tau_b_i32(x, y, n, s);
}
float oracle_mi_binned(const float* x01, const float* y01, int numBins, int n) {
    BinnedScratch s;
    return mi_binned_f64(x01, y01, numBins, n, s);
}
float oracle_mi_kraskov(const float* x, const float* y, int k, int n, int estimator) {
    KraskovScratch s;
    return mi_kraskov_f64(x, y, k, n, estimator, s);
}
double oracle_digamma_int(int n) {
    DigammaTable t;
    t.ensure(std::max(n, 1));
    return t.v[size_t(n)];
}
// Replaces the per-member tie-breaking noise (values already scaled, i.e. u * 1e-10) of the reference / query vector
// for every later Kraskov evaluation; n == 0 restores the documented default stream.  Mirrors crf_set_kraskov_noise.
void oracle_set_kraskov_noise(const double* ref_noise, const double* query_noise, int n) {
    g_noise_override[0].assign(ref_noise, ref_noise + (n > 0 ? n : 0));
    g_noise_override[1].assign(query_noise, query_noise + (n > 0 ? n : 0));
}
// u_e of the documented noise stream (which = 0: reference vector, 1: query vector)
void oracle_noise01(int which, int n, float* out) {
    Xorshift32 g(which == 0 ? SEED_REF : SEED_QUERY);
    for (int e = 0; e < n; e++) out[e] = g.next01();
}

// per-member extrema, the raw material of VolumeData::getMinMaxScalarFieldValue (VolumeData.cpp:1632-1670);
// the calculator then takes min of mins / max of maxes over members (CorrelationCalculator.cpp:822-829).
void oracle_minmax(const float* const* fields, int cs, size_t numVoxels, float* outMin, float* outMax) {
    float mn = std::numeric_limits<float>::max(), mx = std::numeric_limits<float>::lowest();
    for (int c = 0; c < cs; c++) {
        float cmn = std::numeric_limits<float>::max(), cmx = std::numeric_limits<float>::lowest();
        for (size_t i = 0; i < numVoxels; i++) {
            const float v = fields[c][i];
            if (v < cmn) cmn = v;
            if (v > cmx) cmx = v;
        }
        mn = std::min(mn, cmn);
        mx = std::max(mx, cmx);
    }
    *outMin = mn;
    *outMax = mx;
}

/*
 * The calculateCpu driver (CorrelationCalculator.cpp:781-1154).
 *   fields        cs pointers to xs*ys*zs fp32 volumes (IDXS order)
 *   refValues     cs reference values (SINGLE mode: fields[c][IDXS(ref)]; SEPARATE mode: from the other field)
 *   voxelBegin/End  half-open voxel range to evaluate (the whole grid is [0, xs*ys*zs)); out is indexed by the
 *                 absolute voxel index minus voxelBegin.  Sub-ranges exist for sampled parity at full size.
 *   minRef..maxQuery  only read for the binned measures
 *   numThreads    <=0: OpenMP default
 */
int oracle_correlation_field(
        int measure, const float* const* fields, int cs, size_t voxelBegin, size_t voxelEnd,
        const float* refValues, int k, int kraskovEstimator, int numBins,
        float minRef, float maxRef, float minQuery, float maxQuery, float* out, int numThreads) {
    if (cs < 1 || voxelEnd < voxelBegin) return 1;
#ifdef _OPENMP
    const int nt = numThreads > 0 ? numThreads : omp_get_max_threads();
#else
    const int nt = 1;
    (void)numThreads;
#endif
    const float qnan = std::numeric_limits<float>::quiet_NaN();
    std::vector<float> ref(refValues, refValues + cs);
    const bool binned = measure == ORACLE_MI_BINNED || measure == ORACLE_BINNED_MI_CC;
    const bool kraskov = measure == ORACLE_MI_KRASKOV || measure == ORACLE_KMI_CC;
    if (binned) {
        for (int c = 0; c < cs; c++) ref[size_t(c)] = (ref[size_t(c)] - minRef) / (maxRef - minRef);
    }
    std::vector<float> refRanks;
    if (measure == ORACLE_SPEARMAN) {
        RankScratch s;
        refRanks.resize(size_t(cs));
        fractional_ranks(ref.data(), refRanks.data(), cs, s);
    }
    const long long nvox = (long long)(voxelEnd - voxelBegin);
#pragma omp parallel num_threads(nt)
    {
        std::vector<float> q((size_t)cs), qr((size_t)cs);
        RankScratch rs;
        KendallScratch ks;
        BinnedScratch bs;
        KraskovScratch ms;
#pragma omp for schedule(static)
        for (long long i = 0; i < nvox; i++) {
            const size_t idx = voxelBegin + size_t(i);
            if (cs == 1) {
                out[i] = 1.0f;
                continue;
            }
            if (measure == ORACLE_PEARSON) {
                out[i] = pearson2_f32(ref.data(), cs, [&](int e) { return fields[e][idx]; });
                continue;
            }
            bool isNan = false;
            for (int c = 0; c < cs; c++) {
                q[size_t(c)] = fields[c][idx];
                if (std::isnan(q[size_t(c)])) {
                    isNan = true;
                    break;
                }
                if (binned) q[size_t(c)] = (q[size_t(c)] - minQuery) / (maxQuery - minQuery);
            }
            if (isNan) {
                out[i] = qnan;
                continue;
            }
            float v;
            if (measure == ORACLE_SPEARMAN) {
                fractional_ranks(q.data(), qr.data(), cs, rs);
                const float* qrp = qr.data();
                v = pearson2_f32(refRanks.data(), cs, [qrp](int e) { return qrp[e]; });
            } else if (measure == ORACLE_KENDALL) {
                v = kendall_tau_b_i32(ref.data(), q.data(), cs, ks);
            } else if (binned) {
                v = mi_binned_f64(ref.data(), q.data(), numBins, cs, bs);
                if (measure == ORACLE_BINNED_MI_CC) v = mi_to_cc(v);
            } else if (kraskov) {
                v = mi_kraskov_f64(ref.data(), q.data(), k, cs, kraskovEstimator == 2 ? 2 : 1, ms);
                if (measure == ORACLE_KMI_CC) v = mi_to_cc(v);
            } else {
                v = qnan;
            }
            out[i] = v;
        }
    }
    return 0;
}

/*
 * Pair requests (HEBChartCorrelation.cpp:493-600, the per-pair body): X[c] = fields[c][idxI], Y[c] = fields[c][idxJ];
 * a NaN in X or Y -> no entry in the reference (:506-514,:545-551), NaN here; Spearman ranks both vectors (:515-517,
 * :572-573); binned MI normalises X and Y with the extrema over BOTH vectors of the pair (:518-527,:553-566); Kraskov is
 * KSG-1 (:579-581); MI-CC variants map sqrt(1-exp(-2 MI)) in fp32 (:585-592); useAbsoluteCorrelationMeasure takes |.|
 * (:594-596).
 */
int oracle_pair_requests(int measure, const float* const* fields, int cs, const size_t* idxI, const size_t* idxJ,
                         size_t numRequests, int k, int numBins, int useAbs, float* out) {
    const float qnan = std::numeric_limits<float>::quiet_NaN();
#pragma omp parallel
    {
        std::vector<float> X((size_t)cs), Y((size_t)cs), rx((size_t)cs), ry((size_t)cs);
        RankScratch rs;
        KendallScratch ks;
        BinnedScratch bs;
        KraskovScratch ms;
#pragma omp for schedule(static)
        for (long long r = 0; r < (long long)numRequests; r++) {
            bool isNan = false;
            for (int c = 0; c < cs; c++) {
                X[size_t(c)] = fields[c][idxI[r]];
                Y[size_t(c)] = fields[c][idxJ[r]];
                isNan = isNan || std::isnan(X[size_t(c)]) || std::isnan(Y[size_t(c)]);
            }
            if (isNan) {
                out[r] = qnan;
                continue;
            }
            if (cs == 1) {
                out[r] = 1.0f;
                continue;
            }
            float v = qnan;
            const float* xp = X.data();
            const float* yp = Y.data();
            if (measure == ORACLE_PEARSON) {
                v = pearson2_f32(xp, cs, [yp](int e) { return yp[e]; });
            } else if (measure == ORACLE_SPEARMAN) {
                fractional_ranks(xp, rx.data(), cs, rs);
                fractional_ranks(yp, ry.data(), cs, rs);
                const float* ryp = ry.data();
                v = pearson2_f32(rx.data(), cs, [ryp](int e) { return ryp[e]; });
            } else if (measure == ORACLE_KENDALL) {
                v = kendall_tau_b_i32(xp, yp, cs, ks);
            } else if (measure == ORACLE_MI_BINNED || measure == ORACLE_BINNED_MI_CC) {
                float mn = std::numeric_limits<float>::max(), mx = std::numeric_limits<float>::lowest();
                for (int c = 0; c < cs; c++) {
                    mn = std::min(mn, std::min(X[size_t(c)], Y[size_t(c)]));
                    mx = std::max(mx, std::max(X[size_t(c)], Y[size_t(c)]));
                }
                for (int c = 0; c < cs; c++) {
                    X[size_t(c)] = (X[size_t(c)] - mn) / (mx - mn);
                    Y[size_t(c)] = (Y[size_t(c)] - mn) / (mx - mn);
                }
                v = mi_binned_f64(xp, yp, numBins, cs, bs);
                if (measure == ORACLE_BINNED_MI_CC) v = mi_to_cc(v);
            } else if (measure == ORACLE_MI_KRASKOV || measure == ORACLE_KMI_CC) {
                v = mi_kraskov_f64(xp, yp, k, cs, 1, ms);
                if (measure == ORACLE_KMI_CC) v = mi_to_cc(v);
            }
            if (useAbs) v = std::abs(v);
            out[r] = v;
        }
    }
    return 0;
}

/*
 * CorrelationFieldMode::SEPARATE_SYMMETRIC.  PARITY UNPINNED BY THE REFERENCE'S CPU CODE: calculateCpu has no branch for
 * this mode (it falls into SINGLE, CorrelationCalculator.cpp:802-818); only the Vulkan path implements it
 * (CorrelationCalculator.cpp:1182-1229, `#define referencePointIdx currentPointIdx` CorrelationMain.glsl:10-15), with
 * the fp32 shader arithmetic that matches no CPU result (SURVEY 8a "GPU != CPU notes").  The engine therefore DEFINES
 * the mode as calculateCpu's own per-voxel computation with the reference vector taken from the reference field at
 * the same voxel -- which is what this function evaluates, literally, by calling oracle_correlation_field on the
 * one-voxel range [v, v+1) with refValues[c] = fieldsRef[c][v].  A NaN in the reference vector gives NaN (the query
 * side already does, :929-940); Kraskov is KSG-1 (the shaders ignore the estimator index).
 */
int oracle_symmetric_field(int measure, const float* const* fieldsRef, const float* const* fieldsQuery, int cs,
                           size_t voxelBegin, size_t voxelEnd, int k, int numBins, float minRef, float maxRef,
                           float minQuery, float maxQuery, float* out) {
    if (cs < 1 || voxelEnd < voxelBegin) return 1;
    const float qnan = std::numeric_limits<float>::quiet_NaN();
    int status = 0;
#pragma omp parallel
    {
        std::vector<float> ref((size_t)cs);
#pragma omp for schedule(static)
        for (long long i = 0; i < (long long)(voxelEnd - voxelBegin); i++) {
            const size_t v = voxelBegin + size_t(i);
            bool isNan = false;
            for (int c = 0; c < cs; c++) {
                ref[size_t(c)] = fieldsRef[c][v];
                isNan = isNan || std::isnan(ref[size_t(c)]);
            }
            if (isNan && cs > 1) {
                out[i] = qnan;
                continue;
            }
            if (oracle_correlation_field(measure, fieldsQuery, cs, v, v + 1, ref.data(), k, 1, numBins, minRef, maxRef,
                                         minQuery, maxQuery, out + i, 1) != 0) {
#pragma omp atomic write
                status = 1;
            }
        }
    }
    return status;
}

/*
 * Ensemble mean (kind 0) / spread (kind 1): EnsembleMeanCalculator.cpp:110-134, EnsembleSpreadCalculator.cpp:110-145.
 * (Those translation units need sgl/VolumeData and cannot be compiled here; plain fp32 loops, restated.)
 */
int oracle_ensemble_stat(int kind, const float* const* fields, int es, size_t numPoints, float* out) {
#pragma omp parallel for schedule(static)
    for (long long p = 0; p < (long long)numPoints; p++) {
        int numValid = 0;
        float mean = 0.0f;
        for (int e = 0; e < es; e++) {
            const float v = fields[e][p];
            if (!std::isnan(v)) {
                mean += v;
                numValid++;
            }
        }
        if (kind == 0) {
            out[p] = numValid >= 1 ? mean / float(numValid) : std::numeric_limits<float>::quiet_NaN();
        } else if (numValid > 1) {
            mean = mean / float(numValid);
            float varSum = 0.0f;
            for (int e = 0; e < es; e++) {
                const float v = fields[e][p];
                if (!std::isnan(v)) {
                    const float diff = mean - v;
                    varSum += diff * diff;
                }
            }
            out[p] = std::sqrt(varSum / float(numValid - 1));
        } else {
            out[p] = std::numeric_limits<float>::quiet_NaN();
        }
    }
    return 0;
}

/*
 * Benchmark helper (bench.py cpu_baseline, "bound" variant): copies every member volume with the SAME static OpenMP
 * partition over voxels the field loops use, so that with OMP_PROC_BIND / OMP_PLACES set each thread first-touches -- and
 * thereby places on its own NUMA node -- the pages it will read.  dst volumes must be untouched allocations.
 */
int oracle_first_touch_copy(const float* const* src_members, float* const* dst_members, int cs, int64_t n) {
    if (!src_members || !dst_members || cs < 1 || n < 0) return 1;
    for (int c = 0; c < cs; c++) {
        const float* s = src_members[c];
        float* d = dst_members[c];
#pragma omp parallel for schedule(static)
        for (int64_t v = 0; v < n; v++) d[v] = s[v];
    }
    return 0;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/*
 * SetPredicateCalculator::calculateCpu (SetPredicateCalculator.cpp:171-206): op follows ComparisonOperatorType
 * (SetPredicateCalculator.hpp:41-43: > >= < <= == !=).
 */
int oracle_set_predicate(int op, float comparisonValue, int countLower, int countUpper, const float* const* fields,
                         int cs, size_t numPoints, float* out) {
    if (op < 0 || op > 5) return 1;
#pragma omp parallel for schedule(static)
    for (long long p = 0; p < (long long)numPoints; p++) {
        int count = 0;
        for (int c = 0; c < cs; c++) {
            const float v = fields[c][p];
            bool hit = false;
            switch (op) {
                case 0: hit = v > comparisonValue; break;
                case 1: hit = v >= comparisonValue; break;
                case 2: hit = v < comparisonValue; break;
                case 3: hit = v <= comparisonValue; break;
                case 4: hit = v == comparisonValue; break;
                default: hit = v != comparisonValue; break;
            }
            if (hit) count++;
        }
        if (countLower == countUpper) {
            out[p] = std::clamp(float(count) - float(countLower), 0.0f, 1.0f);
        } else {
            out[p] = std::clamp((float(count) - float(countLower)) / (float(countUpper) - float(countLower)), 0.0f, 1.0f);
        }
    }
    return 0;
}

/*
 * DKLCalculator::calculateCpu (DKLCalculator.cpp:134-262): estimator 0 = binned (numBins), 1 = entropy k-NN (k);
 * cs == 1 -> 1.0; a NaN member value -> NaN.  Requires 1 <= k < cs for the k-NN estimator.
 */
int oracle_dkl_field(int estimator, const float* const* fields, int cs, size_t numPoints, int numBins, int k, float* out) {
    if (cs < 1 || (estimator != 0 && estimator != 1)) return 1;
    if (estimator == 1 && cs > 1 && (k < 1 || k >= cs)) return 1;
    if (estimator == 0 && numBins < 1) return 1;
    DigammaTable psi;
    psi.ensure(cs);
#pragma omp parallel
    {
        std::vector<float> vals((size_t)cs);
        std::vector<double> hist;
#pragma omp for schedule(static)
        for (long long p = 0; p < (long long)numPoints; p++) {
            if (cs == 1) {
                out[p] = 1.0f;
                continue;
            }
            bool isNan = false;
            for (int c = 0; c < cs; c++) {
                vals[size_t(c)] = fields[c][p];
                isNan = isNan || std::isnan(vals[size_t(c)]);
            }
            if (isNan) {
                out[p] = std::numeric_limits<float>::quiet_NaN();
                continue;
            }
            out[p] = estimator == 0 ? dkl_binned_f64(vals.data(), numBins, cs, hist) : dkl_knn_f64(vals.data(), k, cs, psi);
        }
    }
    return 0;
}

/*
 * Linear (IDXS) -> 8x8x4-tiled buffer layout (VolumeData.cpp:1581-1621): tiled must hold
 * ceil(xs/8)*ceil(ys/8)*ceil(zs/4)*256 floats.
 */
void oracle_tile_field(const float* linear, int xs, int ys, int zs, float* tiled) {
    const uint32_t tx = 8, ty = 8, tz = 4, tileNumVoxels = tx * ty * tz;
    const uint32_t xst = (uint32_t(xs) + tx - 1) / tx, yst = (uint32_t(ys) + ty - 1) / ty, zst = (uint32_t(zs) + tz - 1) / tz;
    for (uint32_t tileIdx = 0; tileIdx < xst * yst * zst; tileIdx++) {
        const uint32_t xt = tileIdx % xst, yt = (tileIdx / xst) % yst, zt = tileIdx / (xst * yst);
        for (uint32_t voxelIdx = 0; voxelIdx < tileNumVoxels; voxelIdx++) {
            const uint32_t x = voxelIdx % tx + xt * tx, y = (voxelIdx / tx) % ty + yt * ty, z = voxelIdx / (tx * ty) + zt * tz;
            float value = 0.0f;
            if (x < uint32_t(xs) && y < uint32_t(ys) && z < uint32_t(zs))
                value = linear[(size_t(z) * size_t(ys) + y) * size_t(xs) + x];
            tiled[size_t(tileIdx) * tileNumVoxels + voxelIdx] = value;
        }
    }
}

}  // extern "C"
