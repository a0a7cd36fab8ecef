// kernels_generic.hip -- Spearman / Kendall / binned MI / Kraskov MI for ANY member count (cs up to kMaxGenericMembers).
//
// The register-resident kernels (kernels_rank.hip, kernels_binned.hip, kernels_kraskov.hip) are instantiated for cs <= 128; the reference has
// no such limit (its own synthetic data set has 1000 members, scripts/generate_synth_box_ensembles.py:47).  These
// kernels keep the same mapping -- one lane = one voxel -- but hold the voxel's cs values in a per-lane column
// [member][lane] of a tile that lives in LDS when it fits and otherwise in a global workspace slice owned by the block
// (persistent blocks, grid-stride over 64-voxel tiles), and use O(cs^2) counting formulations with runtime loops:
//   Spearman  2*rank_e = 1 + sum_j (2 [v_j < v_e] + [v_j == v_e])           (mid-ranks, Correlation.cpp:277-303)
//   Kendall   S_y = #{ a < b in x order, not in the same x-tie group : y_a > y_b },  n2 = #{ a < b : y_a == y_b }
//   binned    first-occurrence scan over the voxel's cell codes (the skipped-sample path of mi_binned_kernel)
//   Kraskov   the same brute-force k-select as mi_kraskov_kernel, tile pointer instead of LDS
// Integer cores are exact; fp32 tails use the reference's operation order; fp64 sums of the MI estimators differ from
// the reference's order at the 1e-16 level (see kernels_binned.hip).  Throughput is secondary here: at cs = 1000 the
// pair loops are ~10^6 compares per voxel, ~20 ms for the reference's 128x128x32 data set.
#include <cstdlib>

#include "crf_device.h"
#include "crf_internal.h"

namespace crf {

namespace {
constexpr int kGenericBlocks = 1024;        // persistent blocks (4 per CU)
constexpr size_t kLdsTileLimit = 60 * 1024;  // use LDS for the tile when it fits under the default 64 KB limit

__host__ __device__ inline size_t tile_bytes(int cs) { return size_t(cs) * 64 * (sizeof(float) + sizeof(uint16_t)); }
}  // namespace

size_t direct_rank_workspace_bytes(int cs, size_t num_voxels, int measure);

size_t generic_workspace_bytes(int cs, size_t num_voxels) {
    const size_t tiles = (num_voxels + 63) / 64;
    const size_t generic = tile_bytes(cs) <= kLdsTileLimit
                               ? 0
                               : tile_bytes(cs) * (tiles < size_t(kGenericBlocks) ? tiles : size_t(kGenericBlocks));
    const size_t direct = direct_rank_workspace_bytes(cs, num_voxels, 1);  // Spearman's doubled ranks
    return generic > direct ? generic : direct;
}

__device__ __forceinline__ float mi_to_cc_generic(float mi);  // defined below (same map as crf_mi_device.h)

// ---- per-voxel evaluators: `vals` / `aux` point at this lane's column (stride 64 elements) ------------------

// Spearman: ranks by counting, then computePearson2<float>(referenceRanks, ranks, cs) in member order.
__device__ float spearman_voxel(const float* vals, uint16_t* aux, const float* __restrict__ prep_a, int cs) {
    // register blocking: 8 members are ranked per sweep over the column, so every value read from the tile (LDS or the
    // L2-resident workspace) serves 8 comparisons
    constexpr int TE = 8;
#pragma unroll 1
    for (int e0 = 0; e0 < cs; e0 += TE) {
        float ve[TE];
        uint32_t sc[TE];
#pragma unroll
        for (int t = 0; t < TE; t++) {
            ve[t] = vals[(e0 + t < cs ? e0 + t : cs - 1) * 64];
            sc[t] = 0u;
        }
#pragma unroll 2
        for (int j = 0; j < cs; j++) {
            const float vj = vals[j * 64];
#pragma unroll
            for (int t = 0; t < TE; t++) sc[t] += (vj < ve[t]) ? 2u : ((vj == ve[t]) ? 1u : 0u);
        }
#pragma unroll
        for (int t = 0; t < TE; t++)
            if (e0 + t < cs) aux[(e0 + t) * 64] = uint16_t(sc[t] + 1u);  // 2 * rank (self contributes the +1 of [v_e == v_e])
    }
    const float n = float(cs);
    const float invN = 1.0f / n;
    const float invNm1 = 1.0f / (n - 1.0f);
    float meanY = 0.0f;
#pragma unroll 1
    for (int e = 0; e < cs; e++) meanY += invN * (0.5f * float(aux[e * 64]));
    float varY = 0.0f;
#pragma unroll 1
    for (int e = 0; e < cs; e++) {
        const float d = 0.5f * float(aux[e * 64]) - meanY;
        varY += invNm1 * d * d;
    }
    const float sdY = sqrtf(varY);
    float r = 0.0f;
#pragma unroll 1
    for (int e = 0; e < cs; e++) r += prep_a[e] * ((0.5f * float(aux[e * 64]) - meanY) / sdY);
    return r;
}

// Kendall tau-b; vals hold the voxel's values in reference-sorted order (slot = position in the x order).
__device__ float kendall_voxel(const float* vals, const int* __restrict__ prep, int cs) {
    const int* gend = prep + cs;  // slot -> last slot of its x-tie group
    int32_t discordant = 0, n2 = 0;
    // register blocking over i (8 rows per sweep).  For row i the sweep covers j > i: ties in y count for every such j,
    // discordance only for j beyond the row's x-tie group (j > gend[i]).
    constexpr int TI = 8;
#pragma unroll 1
    for (int i0 = 0; i0 < cs; i0 += TI) {
        float yi[TI];
        int gi[TI];
#pragma unroll
        for (int t = 0; t < TI; t++) {
            const int i = i0 + t < cs ? i0 + t : cs - 1;
            yi[t] = vals[i * 64];
            gi[t] = i0 + t < cs ? gend[i] : cs;  // rows past the end: nothing counts (j > cs never holds)
        }
#pragma unroll 2
        for (int j = i0 + 1; j < cs; j++) {
            const float yj = vals[j * 64];
#pragma unroll
            for (int t = 0; t < TI; t++) {
                const bool after = j > i0 + t && i0 + t < cs;
                n2 += (after && yj == yi[t]) ? 1 : 0;
                discordant += (after && j > gi[t] && yi[t] > yj) ? 1 : 0;
            }
        }
    }
    const int32_t n = cs;
    const int32_t n0 = (n * (n - 1)) / 2;
    const int32_t n1 = prep[2 * cs];
    const int32_t numerator = n0 - n1 - n2 - 2 * discordant;
    const float denominator = sqrtf(float(n0 - n1)) * sqrtf(float(n0 - n2));
    return float(numerator) / denominator;
}

// Binned MI from the voxel's cell codes (b1 << 8 | b0, 0xFFFF = skipped sample) in aux.
__device__ float binned_voxel(const uint16_t* aux, int total, bool table_ok, const double* __restrict__ tableT, int cs) {
    double mi = 0.0;
    if (total > 0) {
        const double tot = double(total);
        const double eps1 = 0.5 / double(cs);
        const double eps2 = 0.5 / double(cs * cs);
#pragma unroll 1
        for (int i = 0; i < cs; i++) {
            const uint32_t ci = aux[i * 64];
            if (ci == 0xFFFFu) continue;
            int cx = 0, cy = 0, cxy = 0;
            bool fx = true, fy = true, fxy = true;
#pragma unroll 2
            for (int j = 0; j < cs; j++) {
                const uint32_t cj = aux[j * 64];
                const bool ok = cj != 0xFFFFu;
                const bool ex = ok && (cj & 0xFFu) == (ci & 0xFFu);
                const bool ey = ok && (cj >> 8) == (ci >> 8);
                cx += ex;
                cy += ey;
                cxy += (ex && ey);
                if (j < i) {
                    fx = fx && !ex;
                    fy = fy && !ey;
                    fxy = fxy && !(ex && ey);
                }
            }
            if (table_ok) {  // every sample valid: p = c/cs, p ln p from the host-built table
                if (fx) mi -= tableT[cx];
                if (fy) mi -= tableT[cy];
                if (fxy) mi += tableT[cxy];
            } else {
                if (fx) {
                    const double p = double(cx) / tot;
                    if (p > eps1) mi -= p * log(p);
                }
                if (fy) {
                    const double p = double(cy) / tot;
                    if (p > eps1) mi -= p * log(p);
                }
                if (fxy) {
                    const double p = double(cxy) / tot;
                    if (p > eps2) mi += p * log(p);
                }
            }
        }
    }
    return float(mi);
}

__device__ __forceinline__ int count_less_generic(const double* tab, int n, int top, double v) {
    int pos = 0;
    for (int step = top; step >= 1; step >>= 1) {
        const int idx = pos + step;
        const int probe = idx <= n ? idx : n;
        pos = (idx <= n && tab[probe - 1] < v) ? idx : pos;
    }
    return pos;
}

// Kraskov KSG-1 / KSG-2, any k: selection of the k-th neighbour distance by repeated minimum passes.
__device__ float kraskov_voxel(const float* vals, const double* __restrict__ px, const double* __restrict__ spx,
                               const double* __restrict__ nq, const double* __restrict__ psi, int cs, int k,
                               int estimator, double c_term) {
    const int kk = k < cs - 1 ? k : cs - 1;
    int top = 1;
    while (top * 2 <= cs) top *= 2;
    const double factor = 1.0 / double(cs);
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    const double slack = 1e-15;
    double sum_x = 0.0, sum_y = 0.0;
#pragma unroll 1
    for (int i = 0; i < cs; i++) {
        const double pxi = px[i];
        const double pyi = double(vals[i * 64]) + nq[i];
        double cur = -1.0, m = 0.0;
        int cnt = 0;
#pragma unroll 1
        for (int pass = 0; pass < kk; pass++) {
            m = inf;
            int c = 0;
#pragma unroll 2
            for (int j = 0; j < cs; j++) {
                const double d = fmax(fabs(pxi - px[j]), fabs(pyi - (double(vals[j * 64]) + nq[j])));
                if (d > cur && j != i) {
                    c = (d < m) ? 1 : (d == m ? c + 1 : c);
                    m = fmin(m, d);
                }
            }
            cnt += c;
            if (cnt >= kk) break;
            cur = m;
        }
        double rx, ry;
        if (estimator == 1) {
            rx = ry = m - slack;
        } else {
            double ex = 0.0, ey = 0.0;
#pragma unroll 2
            for (int j = 0; j < cs; j++) {
                const double ax = fabs(pxi - px[j]), ay = fabs(pyi - (double(vals[j * 64]) + nq[j]));
                const bool in = fmax(ax, ay) <= m;
                ex = in ? fmax(ex, ax) : ex;
                ey = in ? fmax(ey, ay) : ey;
            }
            rx = ex + slack;
            ry = ey + slack;
        }
        int cx = count_less_generic(spx, cs, top, pxi + rx) - count_less_generic(spx, cs, top, pxi - rx);
        const double loy = pyi - ry, hiy = pyi + ry;
        int cy = 0;
#pragma unroll 4
        for (int j = 0; j < cs; j++) {
            const double pyj = double(vals[j * 64]) + nq[j];
            cy += (pyj >= loy && pyj < hiy) ? 1 : 0;
        }
        cx = cx > 1 ? cx : 1;
        cy = cy > 1 ? cy : 1;
        if (estimator != 1) {
            cx -= 1;
            cy -= 1;
        }
        sum_x += factor * psi[cx];
        sum_y += factor * psi[cy];
    }
    const double mi = -sum_x - sum_y + c_term + psi[cs];
    const float res = float(mi);
    return (res < 0.0f) ? 0.0f : res;
}

// ---------------------------------------------------------------------------------------------------------------
// Spearman / Kendall beyond the register kernels (cs > 128; the LDS-tile kernel above measured 3.8x slower at 130
// members because a 50 KB tile per wave leaves one wave per SIMD): the O(cs^2) sweeps read the
// values straight from the member volumes (a wave's read of one member is the same coalesced 256 B as a read of a
// workspace copy would be, and with 16 rows per sweep there are 16 comparisons per value read), so no per-block copy of
// the values exists, the grid is not limited by workspace size (the old scheme ran one wave per SIMD) and only
// Spearman keeps a per-voxel column: the 16-bit doubled ranks, in a global workspace slice per block.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kDirectBlocks = 4096;
constexpr int kDirectRows = 16;

size_t direct_rank_workspace_bytes(int cs, size_t num_voxels, int measure) {
    if (measure != 1) return 0;
    const size_t tiles = (num_voxels + 63) / 64;
    return size_t(cs) * 64 * sizeof(uint16_t) * (tiles < size_t(kDirectBlocks) ? tiles : size_t(kDirectBlocks));
}

// MEASURE 1: Spearman (prep = float a_e), 2: Kendall (prep = int perm / gend / n1 with stride cs)
// LIST: the voxels are those of a todo list {count, voxel indices...} (the ones a sort-based kernel deferred: ties)
template <int MEASURE, bool LIST = false>
__global__ __launch_bounds__(64) void direct_rank_kernel(const float* const* __restrict__ members,
                                                         const void* __restrict__ prep, float* __restrict__ out,
                                                         size_t num_voxels, int cs, uint16_t* __restrict__ workspace,
                                                         const uint32_t* __restrict__ todo) {
    constexpr int measure = MEASURE;
    constexpr int T = kDirectRows;
    const int lane = threadIdx.x;
    const uint32_t bytes = uint32_t(num_voxels) * 4u;
    const int* prep_i = static_cast<const int*>(prep);
    const float* prep_a = static_cast<const float*>(prep);
    uint16_t* aux = workspace ? workspace + size_t(blockIdx.x) * size_t(cs) * 64 + lane : nullptr;
    const size_t listed = LIST ? size_t(todo[0]) : 0;
    const size_t tiles = ((LIST ? listed : num_voxels) + 63) / 64;
#pragma unroll 1
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        size_t v = t * 64 + lane;
        if constexpr (LIST) v = v < listed ? size_t(todo[1 + v]) : num_voxels;  // a lane past the end of the list: idle
        const uint32_t off = v < num_voxels ? uint32_t(v) * 4u : kOutOfRangeOffset;
        float res;
        bool is_nan = false;
        if constexpr (measure == 1) {
#pragma unroll 1
            for (int e0 = 0; e0 < cs; e0 += T) {
                float ve[T];
                uint32_t sc[T];
#pragma unroll
                for (int r = 0; r < T; r++) {
                    ve[r] = load_member_cached(members[e0 + r < cs ? e0 + r : cs - 1], bytes, off);
                    is_nan |= ve[r] != ve[r];
                    sc[r] = 0u;
                }
#pragma unroll 4
                for (int j = 0; j < cs; j++) {
                    const float vj = load_member_cached(members[j], bytes, off);
#pragma unroll
                    for (int r = 0; r < T; r++) sc[r] += (vj < ve[r]) ? 2u : ((vj == ve[r]) ? 1u : 0u);
                }
#pragma unroll
                for (int r = 0; r < T; r++)
                    if (e0 + r < cs) aux[size_t(e0 + r) * 64] = uint16_t(sc[r] + 1u);  // 2 * rank
            }
            // computePearson2<float>(referenceRanks, ranks, cs) in member order
            const float n = float(cs);
            const float invN = 1.0f / n;
            const float invNm1 = 1.0f / (n - 1.0f);
            float meanY = 0.0f;
#pragma unroll 4
            for (int e = 0; e < cs; e++) meanY += invN * (0.5f * float(aux[size_t(e) * 64]));
            float varY = 0.0f;
#pragma unroll 4
            for (int e = 0; e < cs; e++) {
                const float d = 0.5f * float(aux[size_t(e) * 64]) - meanY;
                varY += invNm1 * d * d;
            }
            const float sdY = sqrtf(varY);
            float r = 0.0f;
#pragma unroll 4
            for (int e = 0; e < cs; e++) r += prep_a[e] * ((0.5f * float(aux[size_t(e) * 64]) - meanY) / sdY);
            res = r;
        } else {
            const int* gend = prep_i + cs;  // slot -> last slot of its x-tie group (slots = reference-sorted order)
            int32_t discordant = 0, n2 = 0;
#pragma unroll 1
            for (int i0 = 0; i0 < cs; i0 += T) {
                float yi[T];
                int gi[T];
#pragma unroll
                for (int r = 0; r < T; r++) {
                    const int i = i0 + r < cs ? i0 + r : cs - 1;
                    yi[r] = load_member_cached(members[prep_i[i]], bytes, off);
                    is_nan |= yi[r] != yi[r];
                    gi[r] = i0 + r < cs ? gend[i] : cs;
                    if (i0 + r >= cs) yi[r] = __uint_as_float(0x7FC00000u);  // a row past the end matches nothing
                }
                // the block's own columns: row r only counts j > i0 + r
#pragma unroll 1
                for (int j = i0 + 1; j < i0 + T && j < cs; j++) {
                    const float yj = load_member_cached(members[prep_i[j]], bytes, off);
#pragma unroll
                    for (int r = 0; r < T; r++) {
                        const bool after = j > i0 + r;
                        n2 += (after && yj == yi[r]) ? 1 : 0;
                        discordant += (after && j > gi[r] && yi[r] > yj) ? 1 : 0;
                    }
                }
                // every later column counts for all rows of the block
#pragma unroll 4
                for (int j = i0 + T; j < cs; j++) {
                    const float yj = load_member_cached(members[prep_i[j]], bytes, off);
#pragma unroll
                    for (int r = 0; r < T; r++) {
                        n2 += (yj == yi[r]) ? 1 : 0;
                        discordant += (j > gi[r] && yi[r] > yj) ? 1 : 0;
                    }
                }
            }
            const int32_t n = cs;
            const int32_t n0 = (n * (n - 1)) / 2;
            const int32_t n1 = prep_i[2 * cs];
            const int32_t numerator = n0 - n1 - n2 - 2 * discordant;
            const float denominator = sqrtf(float(n0 - n1)) * sqrtf(float(n0 - n2));
            res = float(numerator) / denominator;
        }
        if (is_nan) res = __uint_as_float(0x7FC00000u);
        if (v < num_voxels) store_result_nt(out + v, res);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Symmetric field mode (CRF_FLAG_SYMMETRIC), Spearman and Kendall, any member count: both vectors of a voxel are voxel
// dependent, so nothing is prepared; same direct-read scheme as direct_rank_kernel with two member tables.
//   Spearman: doubled ranks of X and of Y by counting (two 16-bit columns per voxel in the block's workspace slice),
//             then computePearson2<float> on the two rank vectors in member order.
//   Kendall : tau-b in pair form -- S_y = #{x_a < x_b, y_a > y_b}, n1 = #{x_a == x_b}, n2 = #{y_a == y_b} over
//             unordered pairs (the sort of computeKendall orders equal x by y, so x-tied pairs add no inversion;
//             Correlation.cpp:423-455).
// ---------------------------------------------------------------------------------------------------------------
size_t direct_symmetric_workspace_bytes(int cs, size_t num_voxels, int measure) {
    if (measure == 2) return 0;  // Kendall keeps nothing per voxel
    const size_t tiles = (num_voxels + 63) / 64;
    return size_t(cs) * 64 * 2 * sizeof(uint16_t) * (tiles < size_t(kDirectBlocks) ? tiles : size_t(kDirectBlocks));
}

struct SymmetricBinnedArgs {
    int num_bins;  // <= 255: the code 0xFFFF of a skipped sample can then never equal a valid bin pair
    float min_x, max_x, min_y, max_y;
    int to_cc;
};

// measure 1: Spearman, 2: Kendall, 3 / 5: binned MI / its correlation coefficient (cell codes b1 << 8 | b0 in the
// voxel's 16-bit workspace column, then counts of equal bins / cells by 16-row sweeps; the first occurrence of a bin
// or cell contributes its term, exactly like the skipped-sample path of mi_binned_kernel)
// (one instantiation per measure: sharing one kernel cost Kendall 60 % through register allocation)
template <int MEASURE>
__global__ __launch_bounds__(64) void direct_symmetric_kernel(const float* const* __restrict__ members_x,
                                                              const float* const* __restrict__ members_y,
                                                              float* __restrict__ out, size_t num_voxels, int cs,
                                                              uint16_t* __restrict__ workspace, SymmetricBinnedArgs ba,
                                                              const double* __restrict__ tableT) {
    constexpr int measure = MEASURE;
    constexpr int T = kDirectRows;
    const int lane = threadIdx.x;
    const uint32_t bytes = uint32_t(num_voxels) * 4u;
    uint16_t* rx = workspace ? workspace + size_t(blockIdx.x) * size_t(cs) * 128 + lane : nullptr;
    uint16_t* ry = rx ? rx + size_t(cs) * 64 : nullptr;
    const size_t tiles = (num_voxels + 63) / 64;
#pragma unroll 1
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t v = t * 64 + lane;
        const uint32_t off = v < num_voxels ? uint32_t(v) * 4u : kOutOfRangeOffset;
        float res;
        bool is_nan = false;
        if constexpr (measure == 1) {
#pragma unroll 1
            for (int side = 0; side < 2; side++) {
                const float* const* __restrict__ m = side == 0 ? members_x : members_y;
                uint16_t* r2 = side == 0 ? rx : ry;
#pragma unroll 1
                for (int e0 = 0; e0 < cs; e0 += T) {
                    float ve[T];
                    uint32_t sc[T];
#pragma unroll
                    for (int r = 0; r < T; r++) {
                        ve[r] = load_member_cached(m[e0 + r < cs ? e0 + r : cs - 1], bytes, off);
                        is_nan |= ve[r] != ve[r];
                        sc[r] = 0u;
                    }
#pragma unroll 4
                    for (int j = 0; j < cs; j++) {
                        const float vj = load_member_cached(m[j], bytes, off);
#pragma unroll
                        for (int r = 0; r < T; r++) sc[r] += (vj < ve[r]) ? 2u : ((vj == ve[r]) ? 1u : 0u);
                    }
#pragma unroll
                    for (int r = 0; r < T; r++)
                        if (e0 + r < cs) r2[size_t(e0 + r) * 64] = uint16_t(sc[r] + 1u);
                }
            }
            const float n = float(cs);
            const float invN = 1.0f / n;
            const float invNm1 = 1.0f / (n - 1.0f);
            float meanX = 0.0f, meanY = 0.0f;
#pragma unroll 4
            for (int e = 0; e < cs; e++) {
                meanX += invN * (0.5f * float(rx[size_t(e) * 64]));
                meanY += invN * (0.5f * float(ry[size_t(e) * 64]));
            }
            float varX = 0.0f, varY = 0.0f;
#pragma unroll 4
            for (int e = 0; e < cs; e++) {
                const float dx = 0.5f * float(rx[size_t(e) * 64]) - meanX, dy = 0.5f * float(ry[size_t(e) * 64]) - meanY;
                varX += invNm1 * dx * dx;
                varY += invNm1 * dy * dy;
            }
            const float sdX = sqrtf(varX), sdY = sqrtf(varY);
            float r = 0.0f;
#pragma unroll 4
            for (int e = 0; e < cs; e++)
                r += invNm1 * ((0.5f * float(rx[size_t(e) * 64]) - meanX) / sdX) *
                     ((0.5f * float(ry[size_t(e) * 64]) - meanY) / sdY);
            res = r;
        } else if constexpr (measure == 3 || measure == 5) {
            uint16_t* codes = rx;
            const float range_x = ba.max_x - ba.min_x, range_y = ba.max_y - ba.min_y;
            const double nbd = double(ba.num_bins);
            int total = 0;
#pragma unroll 4
            for (int e = 0; e < cs; e++) {
                const float xv = load_member_cached(members_x[e], bytes, off);
                const float yv = load_member_cached(members_y[e], bytes, off);
                is_nan |= xv != xv || yv != yv;
                const float x01 = (xv - ba.min_x) / range_x, y01 = (yv - ba.min_y) / range_y;
                const bool valid = (x01 == x01) && (y01 == y01);
                int b0 = bin_index_x86(double(x01) * nbd), b1 = bin_index_x86(double(y01) * nbd);
                b0 = b0 < 0 ? 0 : (b0 > ba.num_bins - 1 ? ba.num_bins - 1 : b0);
                b1 = b1 < 0 ? 0 : (b1 > ba.num_bins - 1 ? ba.num_bins - 1 : b1);
                codes[size_t(e) * 64] = valid ? uint16_t((b1 << 8) | b0) : uint16_t(0xFFFF);
                total += valid ? 1 : 0;
            }
            double mi = 0.0;
            const bool table_ok = total == cs;
            const double tot = double(total);
            const double eps1 = 0.5 / double(cs), eps2 = 0.5 / double(cs * cs);
#pragma unroll 1
            for (int i0 = 0; i0 < cs && total > 0; i0 += T) {
                uint32_t ci[T];
                int cx[T], cy[T], cxy[T], bx[T], by[T], bxy[T];  // counts over all j / over j < i
#pragma unroll
                for (int r = 0; r < T; r++) {
                    ci[r] = i0 + r < cs ? uint32_t(codes[size_t(i0 + r) * 64]) : 0xFFFFu;
                    cx[r] = cy[r] = cxy[r] = bx[r] = by[r] = bxy[r] = 0;
                }
#pragma unroll 2
                for (int j = 0; j < cs; j++) {
                    const uint32_t cj = codes[size_t(j) * 64];
#pragma unroll
                    for (int r = 0; r < T; r++) {
                        const int ex = ((cj ^ ci[r]) & 0xFFu) == 0u ? 1 : 0;
                        const int ey = ((cj ^ ci[r]) >> 8) == 0u ? 1 : 0;
                        const int exy = cj == ci[r] ? 1 : 0;
                        const int before = j < i0 + r ? 1 : 0;
                        cx[r] += ex;
                        cy[r] += ey;
                        cxy[r] += exy;
                        bx[r] += ex & before;
                        by[r] += ey & before;
                        bxy[r] += exy & before;
                    }
                }
#pragma unroll
                for (int r = 0; r < T; r++) {
                    if (ci[r] == 0xFFFFu) continue;  // skipped sample (or a row past the end)
                    if (table_ok) {
                        if (bx[r] == 0) mi -= tableT[cx[r]];
                        if (by[r] == 0) mi -= tableT[cy[r]];
                        if (bxy[r] == 0) mi += tableT[cxy[r]];
                    } else {
                        if (bx[r] == 0) {
                            const double pr = double(cx[r]) / tot;
                            if (pr > eps1) mi -= pr * log(pr);
                        }
                        if (by[r] == 0) {
                            const double pr = double(cy[r]) / tot;
                            if (pr > eps1) mi -= pr * log(pr);
                        }
                        if (bxy[r] == 0) {
                            const double pr = double(cxy[r]) / tot;
                            if (pr > eps2) mi += pr * log(pr);
                        }
                    }
                }
            }
            res = float(mi);
            if (measure == 5) res = mi_to_cc_generic(res);
        } else {
            int32_t discordant = 0, n1 = 0, n2 = 0;
#pragma unroll 1
            for (int i0 = 0; i0 < cs; i0 += T) {
                float xi[T], yi[T];
#pragma unroll
                for (int r = 0; r < T; r++) {
                    const int i = i0 + r < cs ? i0 + r : cs - 1;
                    xi[r] = load_member_cached(members_x[i], bytes, off);
                    yi[r] = load_member_cached(members_y[i], bytes, off);
                    is_nan |= xi[r] != xi[r] || yi[r] != yi[r];
                }
#pragma unroll 2
                for (int j = i0 + 1; j < cs; j++) {
                    const float xj = load_member_cached(members_x[j], bytes, off);
                    const float yj = load_member_cached(members_y[j], bytes, off);
#pragma unroll
                    for (int r = 0; r < T; r++) {
                        const bool after = j > i0 + r && i0 + r < cs;
                        n1 += (after && xi[r] == xj) ? 1 : 0;
                        n2 += (after && yi[r] == yj) ? 1 : 0;
                        discordant += (after && ((xi[r] < xj && yi[r] > yj) || (xj < xi[r] && yj > yi[r]))) ? 1 : 0;
                    }
                }
            }
            const int32_t n = cs;
            const int32_t n0 = (n * (n - 1)) / 2;
            const int32_t numerator = n0 - n1 - n2 - 2 * discordant;
            res = float(numerator) / (sqrtf(float(n0 - n1)) * sqrtf(float(n0 - n2)));
        }
        if (is_nan) res = __uint_as_float(0x7FC00000u);
        if (cs == 1) res = 1.0f;
        if (v < num_voxels) store_result_nt(out + v, res);
    }
}

hipError_t launch_direct_symmetric(const float* const* d_members_x, const float* const* d_members_y, int cs,
                                   size_t num_voxels, int measure, int num_bins, float min_x, float max_x, float min_y,
                                   float max_y, const double* d_tables, unsigned char* d_workspace, float* d_out,
                                   hipStream_t s) {
    if (measure != 1 && measure != 2 && measure != 3 && measure != 5) return hipErrorNotSupported;
    if (measure != 2 && !d_workspace) return hipErrorInvalidValue;
    const size_t tiles = (num_voxels + 63) / 64;
    const unsigned blocks = unsigned(tiles < size_t(kDirectBlocks) ? tiles : size_t(kDirectBlocks));
    const SymmetricBinnedArgs ba{num_bins, min_x, max_x, min_y, max_y, measure == 5};
    uint16_t* ws = reinterpret_cast<uint16_t*>(d_workspace);
    const double* tableT = d_tables + (cs + 1);
#define CRF_LAUNCH_SYMMETRIC(M)                                                                                       \
    hipLaunchKernelGGL((direct_symmetric_kernel<M>), dim3(blocks), dim3(64), 0, s, d_members_x, d_members_y, d_out,   \
                       num_voxels, cs, ws, ba, tableT)
    switch (measure) {
        case 1: CRF_LAUNCH_SYMMETRIC(1); break;
        case 2: CRF_LAUNCH_SYMMETRIC(2); break;
        case 3: CRF_LAUNCH_SYMMETRIC(3); break;
        default: CRF_LAUNCH_SYMMETRIC(5); break;
    }
#undef CRF_LAUNCH_SYMMETRIC
    return hipGetLastError();
}

__global__ __launch_bounds__(64) void generic_kernel(const float* const* __restrict__ members,
                                                     const void* __restrict__ prep, const double* __restrict__ tables,
                                                     float* __restrict__ out, size_t num_voxels, int cs, GenericArgs a,
                                                     unsigned char* __restrict__ workspace) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* tile = workspace ? workspace + size_t(blockIdx.x) * tile_bytes(cs) : smem;
    const int lane = threadIdx.x;
    float* vals = reinterpret_cast<float*>(tile) + lane;                                        // [cs][64]
    uint16_t* aux = reinterpret_cast<uint16_t*>(tile + size_t(cs) * 64 * sizeof(float)) + lane;  // [cs][64]
    const uint32_t bytes = uint32_t(num_voxels) * 4u;
    const int* prep_i = static_cast<const int*>(prep);
    const size_t tiles = (num_voxels + 63) / 64;
    const bool binned = a.measure == 3 || a.measure == 5;
    const bool kendall = a.measure == 2;
#pragma unroll 1
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t v = t * 64 + lane;
        const uint32_t byte_offset = uint32_t(v) * 4u;  // lanes past the end read 0
        bool is_nan = false;
        int total = 0;
        const float range_q = a.max_query - a.min_query;
#pragma unroll 4
        for (int e = 0; e < cs; e++) {
            const float y = load_member_nt(members[kendall ? prep_i[e] : e], bytes, byte_offset);
            is_nan |= (y != y);
            vals[e * 64] = y;
            if (binned) {  // cell code, as in mi_binned_kernel
                const float q01 = (y - a.min_query) / range_q;
                const int b0 = prep_i[e];
                const bool valid = (q01 == q01) && b0 != 0xFFFF;
                int b1 = bin_index_x86(double(q01) * double(a.num_bins));
                b1 = b1 < 0 ? 0 : (b1 > a.num_bins - 1 ? a.num_bins - 1 : b1);
                aux[e * 64] = valid ? uint16_t((b1 << 8) | b0) : uint16_t(0xFFFF);
                total += valid ? 1 : 0;
            }
        }
        float res;
        switch (a.measure) {
            case 1: res = spearman_voxel(vals, aux, static_cast<const float*>(prep), cs); break;
            case 2: res = kendall_voxel(vals, prep_i, cs); break;
            case 3:
            case 5: {
                const bool table_ok = total == cs && prep_i[cs] != 0;
                res = binned_voxel(aux, total, table_ok, tables + (cs + 1), cs);
                if (a.measure == 5) res = mi_to_cc_generic(res);
                break;
            }
            default: {
                const double* prep_d = static_cast<const double*>(prep);
                res = kraskov_voxel(vals, prep_d, prep_d + cs, tables + 3 * cs + 2, tables, cs, a.k, a.estimator,
                                    a.kraskov_c);
                if (a.measure == 6) res = mi_to_cc_generic(res);
                break;
            }
        }
        if (is_nan) res = __uint_as_float(0x7FC00000u);
        if (v < num_voxels) store_result_nt(out + v, res);
    }
}

// glibc-compatible expf for the MI-CC map: see crf_mi_device.h (same algorithm and table).
__device__ const uint64_t kExp2Tab32G[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

__device__ __forceinline__ float mi_to_cc_generic(float mi) {
    const float x = -2.0f * mi;
    float e;
    if (!(x > -80.0f && x < 80.0f)) {
        e = expf(x);
    } else {
        const double shift = 0x1.8p+52;
        double z = (0x1.71547652b82fep+0 * 32.0) * double(x);
        double kd = z + shift;
        const uint64_t ki = uint64_t(__double_as_longlong(kd));
        kd -= shift;
        const double r = z - kd;
        const double s = __longlong_as_double((long long)(kExp2Tab32G[ki & 31u] + (ki << 47)));
        z = (0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0) * r + (0x1.ebfce50fac4f3p-3 / 32.0 / 32.0);
        const double r2 = r * r;
        double y = (0x1.62e42ff0c52d6p-1 / 32.0) * r + 1.0;
        y = z * r2 + y;
        e = float(y * s);
    }
    return sqrtf(1.0f - e);
}

// =========================================================================================================
// Pair-request mode: the estimator between the ensemble vectors of two arbitrary voxels per request -- the reference's
// CorrelationComputePass request mode (src/Calculators/CorrelationCalculator.hpp:250-258, request layout
// {xi,yi,zi,i,xj,yj,zj,j} src/Renderers/Diagram/HEBChart.hpp:166-168, Data/Shaders/Correlation/RequestsBuffer.glsl:22-36)
// with the semantics of its CPU twin HEBChart::computeCorrelations (src/Renderers/Diagram/HEBChartCorrelation.cpp:493-600):
// both vectors are voxel dependent (no shared reference side), binned MI normalises with the extrema of the two
// vectors of the pair (:556-566), Kraskov is KSG-1, NaN in either vector yields NaN (the CPU twin emits no entry).
// One lane = one request; 2*cs gathered loads per request; same tile scheme as above (x, y, two u16 columns).
// =========================================================================================================
namespace {
__host__ __device__ inline size_t pair_tile_bytes(int cs) {
    return size_t(cs) * 64 * (2 * sizeof(float) + 2 * sizeof(uint16_t));
}
}  // namespace

size_t pair_workspace_bytes(int cs, size_t num_requests) {
    if (pair_tile_bytes(cs) <= kLdsTileLimit) return 0;
    const size_t tiles = (num_requests + 63) / 64;
    return pair_tile_bytes(cs) * (tiles < size_t(kGenericBlocks) ? tiles : size_t(kGenericBlocks));
}

__device__ float pearson_pair(const float* x, const float* y, int cs, float half_scale_x, float half_scale_y,
                              const uint16_t* rx, const uint16_t* ry) {
    // computePearson2<float>(X, Y, cs) (Correlation.cpp:141-174); with rx/ry != null the inputs are the half-integer
    // ranks 0.5 * r2 (Spearman)
    auto X = [&](int e) { return rx ? half_scale_x * float(rx[e * 64]) : x[e * 64]; };
    auto Y = [&](int e) { return ry ? half_scale_y * float(ry[e * 64]) : y[e * 64]; };
    const float n = float(cs);
    const float invN = 1.0f / n;
    const float invNm1 = 1.0f / (n - 1.0f);
    float meanX = 0.0f, meanY = 0.0f;
#pragma unroll 1
    for (int e = 0; e < cs; e++) {
        meanX += invN * X(e);
        meanY += invN * Y(e);
    }
    float varX = 0.0f, varY = 0.0f;
#pragma unroll 1
    for (int e = 0; e < cs; e++) {
        const float dx = X(e) - meanX, dy = Y(e) - meanY;
        varX += invNm1 * dx * dx;
        varY += invNm1 * dy * dy;
    }
    const float sdX = sqrtf(varX), sdY = sqrtf(varY);
    float r = 0.0f;
#pragma unroll 1
    for (int e = 0; e < cs; e++) r += invNm1 * ((X(e) - meanX) / sdX) * ((Y(e) - meanY) / sdY);
    return r;
}

__device__ void ranks2_column(const float* v, uint16_t* r2, int cs) {
#pragma unroll 1
    for (int e = 0; e < cs; e++) {
        const float ve = v[e * 64];
        uint32_t s = 0;
#pragma unroll 4
        for (int j = 0; j < cs; j++) {
            const float vj = v[j * 64];
            s += (vj < ve) ? 2u : ((vj == ve) ? 1u : 0u);
        }
        r2[e * 64] = uint16_t(s + 1u);
    }
}

__device__ float kendall_pair(const float* x, const float* y, int cs) {
    int32_t discordant = 0, n1 = 0, n2 = 0;
#pragma unroll 1
    for (int a = 0; a < cs; a++) {
        const float xa = x[a * 64], ya = y[a * 64];
#pragma unroll 4
        for (int b = a + 1; b < cs; b++) {
            const float xb = x[b * 64], yb = y[b * 64];
            n1 += (xa == xb) ? 1 : 0;
            n2 += (ya == yb) ? 1 : 0;
            discordant += ((xa < xb && ya > yb) || (xb < xa && yb > ya)) ? 1 : 0;
        }
    }
    const int32_t n = cs;
    const int32_t n0 = (n * (n - 1)) / 2;
    const int32_t numerator = n0 - n1 - n2 - 2 * discordant;
    return float(numerator) / (sqrtf(float(n0 - n1)) * sqrtf(float(n0 - n2)));
}

__device__ float kraskov_pair(const float* x, const float* y, const double* __restrict__ nr,
                              const double* __restrict__ nq, const double* __restrict__ psi, int cs, int k,
                              double c_term) {
    const int kk = k < cs - 1 ? k : cs - 1;
    const double factor = 1.0 / double(cs);
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    double sum_x = 0.0, sum_y = 0.0;
#pragma unroll 1
    for (int i = 0; i < cs; i++) {
        const double pxi = double(x[i * 64]) + nr[i], pyi = double(y[i * 64]) + nq[i];
        double cur = -1.0, m = 0.0;
        int cnt = 0;
#pragma unroll 1
        for (int pass = 0; pass < kk; pass++) {
            m = inf;
            int c = 0;
#pragma unroll 2
            for (int j = 0; j < cs; j++) {
                const double d = fmax(fabs(pxi - (double(x[j * 64]) + nr[j])), fabs(pyi - (double(y[j * 64]) + nq[j])));
                if (d > cur && j != i) {
                    c = (d < m) ? 1 : (d == m ? c + 1 : c);
                    m = fmin(m, d);
                }
            }
            cnt += c;
            if (cnt >= kk) break;
            cur = m;
        }
        const double r = m - 1e-15;
        const double lox = pxi - r, hix = pxi + r, loy = pyi - r, hiy = pyi + r;
        int cx = 0, cy = 0;
#pragma unroll 2
        for (int j = 0; j < cs; j++) {
            const double pxj = double(x[j * 64]) + nr[j], pyj = double(y[j * 64]) + nq[j];
            cx += (pxj >= lox && pxj < hix) ? 1 : 0;
            cy += (pyj >= loy && pyj < hiy) ? 1 : 0;
        }
        sum_x += factor * psi[cx > 1 ? cx : 1];
        sum_y += factor * psi[cy > 1 ? cy : 1];
    }
    const double mi = -sum_x - sum_y + c_term + psi[cs];
    const float res = float(mi);
    return (res < 0.0f) ? 0.0f : res;
}

// members_i / members_j: the member sets the first / second voxel of a request is read from (the same table for the
// request mode; primary / secondary field for the symmetric field mode).  requests == nullptr: request r is the
// voxel pair (r, r) -- SEPARATE_SYMMETRIC, CorrelationMain.glsl:10-15.  a.fixed_ranges: binned MI normalises with
// the given global ranges (CorrelationCalculator.cpp:820-846) instead of the pair's own extrema (HEBChart).
__global__ __launch_bounds__(64) void pair_request_kernel(const float* const* __restrict__ members_i,
                                                          const float* const* __restrict__ members_j,
                                                          const uint32_t* __restrict__ requests,
                                                          const double* __restrict__ tables, float* __restrict__ out,
                                                          size_t num_requests, size_t num_voxels, int xs, int ys, int cs,
                                                          PairArgs a, unsigned char* __restrict__ workspace) {
    const int measure = a.measure, num_bins = a.num_bins, k = a.k, use_abs = a.use_abs;
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* tile = workspace ? workspace + size_t(blockIdx.x) * pair_tile_bytes(cs) : smem;
    const int lane = threadIdx.x;
    float* x = reinterpret_cast<float*>(tile) + lane;
    float* y = x + size_t(cs) * 64;
    uint16_t* ax = reinterpret_cast<uint16_t*>(tile + size_t(cs) * 64 * 2 * sizeof(float)) + lane;
    uint16_t* ay = ax + size_t(cs) * 64;
    const uint32_t bytes = uint32_t(num_voxels) * 4u;
    const size_t tiles = (num_requests + 63) / 64;
#pragma unroll 1
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t r = t * 64 + lane;
        const bool active = r < num_requests;
        uint32_t vi = 0, vj = 0;
        if (active) {
            if (requests) {
                const uint32_t* q = requests + r * 8;
                vi = (q[2] * uint32_t(ys) + q[1]) * uint32_t(xs) + q[0];  // IDXS
                vj = (q[6] * uint32_t(ys) + q[5]) * uint32_t(xs) + q[4];
            } else {
                vi = vj = uint32_t(r);
            }
        }
        bool is_nan = false;
        float mn = __uint_as_float(0x7F7FFFFFu), mx = __uint_as_float(0xFF7FFFFFu);
#pragma unroll 4
        for (int e = 0; e < cs; e++) {
            const float xv = load_member(members_i[e], vi * 4u);
            const float yv = load_member(members_j[e], vj * 4u);
            is_nan |= (xv != xv) || (yv != yv);
            x[e * 64] = xv;
            y[e * 64] = yv;
            mn = fminf(mn, fminf(xv, yv));
            mx = fmaxf(mx, fmaxf(xv, yv));
        }
        (void)bytes;
        float res;
        switch (measure) {
            case 0: res = pearson_pair(x, y, cs, 0.f, 0.f, nullptr, nullptr); break;
            case 1:
                ranks2_column(x, ax, cs);
                ranks2_column(y, ay, cs);
                res = pearson_pair(x, y, cs, 0.5f, 0.5f, ax, ay);
                break;
            case 2: res = kendall_pair(x, y, cs); break;
            case 3:
            case 5: {
                int total = 0;
                const float mnx = a.fixed_ranges ? a.min_ref : mn, mny = a.fixed_ranges ? a.min_query : mn;
                const float range_x = a.fixed_ranges ? a.max_ref - a.min_ref : mx - mn;
                const float range_y = a.fixed_ranges ? a.max_query - a.min_query : mx - mn;
#pragma unroll 2
                for (int e = 0; e < cs; e++) {
                    const float x01 = (x[e * 64] - mnx) / range_x, y01 = (y[e * 64] - mny) / range_y;
                    const bool valid = (x01 == x01) && (y01 == y01);
                    int b0 = bin_index_x86(double(x01) * double(num_bins)), b1 = bin_index_x86(double(y01) * double(num_bins));
                    b0 = b0 < 0 ? 0 : (b0 > num_bins - 1 ? num_bins - 1 : b0);
                    b1 = b1 < 0 ? 0 : (b1 > num_bins - 1 ? num_bins - 1 : b1);
                    ax[e * 64] = valid ? uint16_t((b1 << 8) | b0) : uint16_t(0xFFFF);
                    total += valid ? 1 : 0;
                }
                res = binned_voxel(ax, total, total == cs, tables + (cs + 1), cs);
                if (measure == 5) res = mi_to_cc_generic(res);
                break;
            }
            default:
                res = kraskov_pair(x, y, tables + 2 * (cs + 1), tables + 3 * cs + 2, tables, cs, k, a.kraskov_c);
                if (measure == 6) res = mi_to_cc_generic(res);
                break;
        }
        if (use_abs) res = fabsf(res);
        if (is_nan) res = __uint_as_float(0x7FC00000u);
        if (cs == 1) res = 1.0f;
        if (active) out[r] = res;
    }
}

hipError_t launch_pair_requests(const float* const* d_members_i, const float* const* d_members_j, int cs, int xs, int ys,
                                size_t num_voxels, const uint32_t* d_requests, size_t num_requests, const PairArgs& a,
                                const double* d_tables, unsigned char* d_workspace, float* d_out, hipStream_t s) {
    if (num_requests == 0) return hipSuccess;
    const size_t tiles = (num_requests + 63) / 64;
    const unsigned blocks = unsigned(tiles < size_t(kGenericBlocks) ? tiles : size_t(kGenericBlocks));
    const bool use_lds = pair_tile_bytes(cs) <= kLdsTileLimit;
    if (!use_lds && !d_workspace) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pair_request_kernel, dim3(blocks), dim3(64), use_lds ? pair_tile_bytes(cs) : 0, s, d_members_i,
                       d_members_j, d_requests, d_tables, d_out, num_requests, num_voxels, xs, ys, cs, a,
                       use_lds ? nullptr : d_workspace);
    return hipGetLastError();
}

hipError_t launch_generic(const float* const* d_members, int cs, size_t num_voxels, const RefSource& ref,
                          const GenericArgs& a, const double* d_tables, float* d_prep, unsigned char* d_workspace,
                          float* d_out, hipStream_t s, hipEvent_t ev_begin, hipEvent_t ev_end, LaunchInfo* info,
                          uint32_t* d_todo) {
    switch (ref.prepare() ? a.measure : -1) {
        case -1: break;
        case 1: launch_spearman_prep(ref, d_members, cs, d_prep, s); break;
        case 2: launch_kendall_prep(ref, d_members, cs, cs, reinterpret_cast<int*>(d_prep), s); break;
        case 3:
        case 5: {
            const BinnedArgs b{a.num_bins, a.min_ref, a.max_ref, a.min_query, a.max_query, a.measure == 5};
            launch_binned_prep(ref, d_members, cs, cs, b, d_tables + (cs + 1), reinterpret_cast<int*>(d_prep), s);
            break;
        }
        case 4:
        case 6:
            launch_kraskov_prep(ref, d_members, cs, d_tables + 2 * (cs + 1), reinterpret_cast<double*>(d_prep), s);
            break;
        default: return hipErrorInvalidValue;
    }
    if (!ref.run()) return hipGetLastError();
    const size_t tiles = (num_voxels + 63) / 64;
    const char* force_tile = getenv("CRF_RANK_TILE");  // tuning: keep the LDS-tile kernel wherever the tile fits
    if ((a.measure == 1 || a.measure == 2) &&
        (tile_bytes(cs) > kLdsTileLimit || !(force_tile && *force_tile == '1'))) {
        const unsigned dblocks = unsigned(tiles < size_t(kDirectBlocks) ? tiles : size_t(kDirectBlocks));
        if (a.measure == 1 && !d_workspace) return hipErrorInvalidValue;
        if (ev_begin) (void)hipEventRecord(ev_begin, s);
        // Spearman at 129..256 members: two sorted chunks merged through LDS (kernels_rank.hip: spearman_pair_kernel),
        // then this file's counting kernel over the voxels it deferred (ties); Kendall the same (kendall_pair_kernel).
        // CRF_RANK_PAIR=0: counting kernel for all.
        const char* pair_env = getenv("CRF_RANK_PAIR");
        if (a.measure == 1 && !(pair_env && *pair_env == '0') &&
            launch_spearman_pair(d_members, d_prep, d_out, num_voxels, cs, d_todo, s)) {
            hipLaunchKernelGGL((direct_rank_kernel<1, true>), dim3(dblocks < 1024u ? dblocks : 1024u), dim3(64), 0, s,
                               d_members, static_cast<const void*>(d_prep), d_out, num_voxels, cs,
                               reinterpret_cast<uint16_t*>(d_workspace), static_cast<const uint32_t*>(d_todo));
            if (ev_end) (void)hipEventRecord(ev_end, s);
            if (info) info->kernel_name = "spearman_pair_kernel";
            return hipGetLastError();
        }
        if (a.measure == 2 && !(pair_env && *pair_env == '0') &&
            launch_kendall_pair(d_members, reinterpret_cast<const int*>(d_prep), d_out, num_voxels, cs, d_todo, s)) {
            hipLaunchKernelGGL((direct_rank_kernel<2, true>), dim3(dblocks < 1024u ? dblocks : 1024u), dim3(64), 0, s,
                               d_members, static_cast<const void*>(d_prep), d_out, num_voxels, cs,
                               reinterpret_cast<uint16_t*>(d_workspace), static_cast<const uint32_t*>(d_todo));
            if (ev_end) (void)hipEventRecord(ev_end, s);
            if (info) info->kernel_name = "kendall_pair_kernel";
            return hipGetLastError();
        }
        if (a.measure == 1)
            hipLaunchKernelGGL((direct_rank_kernel<1, false>), dim3(dblocks), dim3(64), 0, s, d_members,
                               static_cast<const void*>(d_prep), d_out, num_voxels, cs,
                               reinterpret_cast<uint16_t*>(d_workspace), static_cast<const uint32_t*>(nullptr));
        else
            hipLaunchKernelGGL((direct_rank_kernel<2, false>), dim3(dblocks), dim3(64), 0, s, d_members,
                               static_cast<const void*>(d_prep), d_out, num_voxels, cs,
                               reinterpret_cast<uint16_t*>(d_workspace), static_cast<const uint32_t*>(nullptr));
        if (ev_end) (void)hipEventRecord(ev_end, s);
        if (info) info->kernel_name = "direct_rank_kernel";
        return hipGetLastError();
    }
    const unsigned blocks = unsigned(tiles < size_t(kGenericBlocks) ? tiles : size_t(kGenericBlocks));
    const bool use_lds = tile_bytes(cs) <= kLdsTileLimit;
    if (!use_lds && !d_workspace) return hipErrorInvalidValue;
    if (ev_begin) (void)hipEventRecord(ev_begin, s);
    hipLaunchKernelGGL(generic_kernel, dim3(blocks), dim3(64), use_lds ? tile_bytes(cs) : 0, s, d_members,
                       static_cast<const void*>(d_prep), d_tables, d_out, num_voxels, cs, a,
                       use_lds ? nullptr : d_workspace);
    if (ev_end) (void)hipEventRecord(ev_end, s);
    if (info) info->kernel_name = "generic_kernel";
    return hipGetLastError();
}

}  // namespace crf
