// kernels_kraskov.hip -- Kraskov (KSG) k-nearest-neighbour mutual information on gfx950 (binned MI: kernels_binned.hip).
//
// Binned MI (computeMutualInformationBinned<double>, MutualInformation.cpp:45-143; driver
// CorrelationCalculator.cpp:820-846,1026-1085).  The reference fills an 80x80 fp64 histogram per voxel although at
// most cs cells are occupied.  Here one lane owns one voxel and works on the <= cs occupied cells only: each sample
// becomes a 16-bit cell code (b1 << 8 | b0), the codes are sorted by a register min/max network, and one scan over
// the sorted codes yields the run lengths of equal cells (joint histogram) and of equal b1 (query marginal).  With
// every sample valid, a cell/marginal probability is c/cs for an integer count c, so p*ln(p) comes from a cs+1 entry
// fp64 table built on the host with the same libm log as the reference; the reference-vector marginal is voxel
// independent and is summed once per evaluation.  MI = -sum_x - sum_y + sum_xy, accumulated in fp64, returned as
// float.  fp64 sums are taken in sorted-cell order, which is not the reference's bin-index order: a difference of
// a few 1e-16 relative before the final cast to float (tolerance 1e-5 relative per the north star; in practice the
// float results are bit-identical except at rounding boundaries).  Voxels with skipped samples (normalised value
// NaN, MutualInformation.cpp:64 -- needs infinities in the data or max == min) take a compact O(cs^2) path.
//
// Kraskov kNN MI (KSG-1 computeMutualInformationKraskov<double>, MutualInformation.cpp:399-444; KSG-2 :449-509;
// averageDigamma :167-259).  fp64 throughout, as the reference.  One lane owns one voxel; the voxel's cs values are
// parked in that lane's LDS column; for each point a brute-force Chebyshev k-select over all points keeps the k+1
// smallest distances in registers (sorted insertion, min/max only), then the marginal counts are brute-force
// compares against [c-r, c+r); psi(n) comes from a table.  The tie-breaking noise (1e-10 * u) uses this repo's
// documented xorshift32 stream, identical to oracle/corr_oracle.cpp (sgl's generator is not available).
#include <cstdlib>
#include <type_traits>

#include "crf_device.h"
#include "crf_internal.h"

#include <cstdlib>

#include "crf_mi_device.h"

namespace crf {

// ---------------------------------------------------------------------------------------------------------
// Kraskov
// ---------------------------------------------------------------------------------------------------------
// prep (fp64 view): [0, cs) px_e = double(ref_e) + noise_ref_e (MutualInformation.cpp:417-420), member order;
//                   [cs, 2cs) the same values sorted ascending (the reference sorts them for its 1-D range counts,
//                   MutualInformation.cpp:187; voxel independent, so sorted once per evaluation)
// the distance table lies in the preparation buffer behind the two coordinate vectors (up to 128 members: 133 KB)
constexpr int kDxtMaxMembers = 128;
__host__ __device__ inline int dxt_offset(int cs) { return (2 * cs + 7) / 8 * 8; }  // in doubles
static_assert(size_t(((2 * kDxtMaxMembers + 7) / 8 * 8) + 128 * 128) * sizeof(double) <= kPrepBytes - 16, "prep buffer");

__global__ __launch_bounds__(256) void kraskov_prep_kernel(RefSource src, const float* const* __restrict__ members,
                                                           int cs, const double* __restrict__ noise_ref,
                                                           double* __restrict__ prep) {
    extern __shared__ double px[];  // cs doubles
    for (int e = threadIdx.x; e < cs; e += blockDim.x) px[e] = double(load_ref(src, members, e)) + noise_ref[e];
    __syncthreads();
    for (int e = threadIdx.x; e < cs; e += blockDim.x) {
        const double v = px[e];
        int rank = 0;
        for (int j = 0; j < cs; j++) rank += (px[j] < v || (px[j] == v && j < e)) ? 1 : 0;
        prep[e] = v;
        prep[cs + rank] = v;
    }
    // x-distance table for the tile-free kernel (cs <= kDxtMaxMembers): T[j][i] = |px_i - px_j| for candidate j and
    // point i, laid out [round_up(cs, 16)][round_up(cs, 8)] behind the two vectors (64-byte aligned).  The diagonal holds
    // the largest finite double (a point is not its own neighbour) and the rows of candidates past the end +inf, so the
    // kernel needs neither a self test nor an end test per pair; it reads a row segment of 8 with one scalar load.
    if (cs <= kDxtMaxMembers) {
        double* table = prep + dxt_offset(cs);
        const int rows = (cs + 15) / 16 * 16, cols = (cs + 7) / 8 * 8;
        for (int idx = threadIdx.x; idx < rows * cols; idx += blockDim.x) {
            const int j = idx / cols, i = idx - j * cols;
            double d = __longlong_as_double(0x7FF0000000000000ll);
            if (j < cs && i < cs) d = (i == j) ? __longlong_as_double(0x7FEFFFFFFFFFFFFFll) : fabs(px[i] - px[j]);
            table[idx] = d;
        }
    }
}

constexpr double kCountSlack = 1e-15;  // default_epsilon<double>::value, MutualInformation.cpp:163

// fp64 min / max / Chebyshev distance as single instructions.  fmin()/fmax() make hipcc canonicalise each operand first
// (an extra v_max_f64 x, x); the operands here are never signalling NaNs and the k-select only needs "one of the two".
__device__ __forceinline__ double min_f64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double max_f64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double chebyshev_f64(double dx, double dy) {  // max(|dx|, |dy|)
    double r;
    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(dx), "v"(dy));
    return r;
}

__device__ __forceinline__ double chebyshev_f64_sx(double dy, double abs_dx_uniform) {  // max(|dy|, s): s in SGPRs
    double r;
    asm("v_max_f64 %0, |%1|, %2" : "=v"(r) : "v"(dy), "s"(abs_dx_uniform));
    return r;
}

// #{ t : tab[t] < v } for an ascending table of n entries; top = largest power of two <= n.  Branch-free binary search;
// the table is shared by the wave (LDS), the probe index is per lane.
__device__ __forceinline__ int count_less(const double* tab, int n, int top, double v) {
    int pos = 0;
    for (int step = top; step >= 1; step >>= 1) {
        const int idx = pos + step;
        const int probe = idx <= n ? idx : n;  // keep the read in range; the result is discarded when idx > n
        pos = (idx <= n && tab[probe - 1] < v) ? idx : pos;
    }
    return pos;
}

// The same search for NCH bounds at once: the steps are the outer loop, so a step issues NCH independent LDS reads before
// it waits (one search after the other is a chain of log2(n) dependent LDS round trips each -- and the && above
// compiles to exec-masked branches; at 8 points x 2 bounds per sweep that latency was as long as the k-select itself).
template <int NCH>
__device__ __forceinline__ void count_less_batch(const double* tab, int n, int top, const double (&v)[NCH],
                                                 int (&pos)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; c++) pos[c] = 0;
#pragma unroll 1
    for (int step = top; step >= 1; step >>= 1) {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int idx = pos[c] + step;
            const bool ok = idx <= n;
            const double val = tab[(ok ? idx : n) - 1];
            pos[c] = (ok & (val < v[c])) ? idx : pos[c];
        }
    }
}

// K > 0: the K = k nearest OTHER points are kept in registers (sorted insertion: min/max only).  K == 0: any k,
// selection by repeated minimum passes.  TI points are processed concurrently per lane.
template <int K, int TI, bool DXT = false>
__global__ __launch_bounds__(64) void mi_kraskov_kernel(const float* const* __restrict__ members,
                                                        const double* __restrict__ prep_px,
                                                        const double* __restrict__ table_psi,
                                                        const double* __restrict__ noise_query,
                                                        float* __restrict__ out, size_t num_voxels, int cs, int k,
                                                        int estimator, int to_cc, double c_term) {
    extern __shared__ __align__(16) unsigned char smem[];
    double* s_px = reinterpret_cast<double*>(smem);  // member order
    double* s_spx = s_px + cs;                       // ascending
    double* s_nq = s_spx + cs;
    double* s_psi = s_nq + cs;                                                  // cs + 1 entries: psi(0..cs)
    float* s_y = reinterpret_cast<float*>(s_psi + (cs + 1) + ((cs + 1) & 1));  // keep 16-byte alignment
    const int lane = threadIdx.x;
    for (int i = lane; i < cs; i += 64) {
        s_px[i] = prep_px[i];
        s_spx[i] = prep_px[cs + i];
        s_nq[i] = noise_query[i];
    }
    for (int i = lane; i <= cs; i += 64) s_psi[i] = table_psi[i];
    const size_t v = size_t(blockIdx.x) * 64 + lane;
    const uint32_t byte_offset = uint32_t(v) * 4u, bytes = uint32_t(num_voxels) * 4u;  // lanes past the end read 0
    bool is_nan = false;
#pragma unroll 8
    for (int e = 0; e < cs; e++) {
        const float y = load_member_nt(members[e], bytes, byte_offset);
        is_nan |= (y != y);
        s_y[e * 64 + lane] = y;
    }
    __syncthreads();

    const int kk = k < cs - 1 ? k : cs - 1;  // neighbours besides the point itself (a kd-tree returns at most cs points)
    int top = 1;
    while (top * 2 <= cs) top *= 2;
    const double factor = 1.0 / double(cs);
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    double sum_x = 0.0, sum_y = 0.0;

#pragma unroll 1
    for (int i0 = 0; i0 < cs; i0 += TI) {
        double pxi[TI], pyi[TI], rx[TI], ry[TI];
#pragma unroll
        for (int t = 0; t < TI; t++) {
            const int ii = (i0 + t < cs) ? i0 + t : cs - 1;
            pxi[t] = s_px[ii];
            pyi[t] = double(s_y[ii * 64 + lane]) + s_nq[ii];
        }
        const int i1 = (i0 + TI < cs) ? i0 + TI : cs;
        // ---- pass A: Chebyshev distance to the k-th neighbour (the (k+1)-th smallest distance including the zero
        //      distance to the point itself, MutualInformation.cpp:430-434)
        double dk[TI];
        if constexpr (K > 0) {
            double best[TI][K];
#pragma unroll
            for (int t = 0; t < TI; t++)
#pragma unroll
                for (int q = 0; q < K; q++) best[t][q] = inf;
            auto visit = [&](int j, bool may_be_self) {
                const double pxj = s_px[j];
                const double pyj = double(s_y[j * 64 + lane]) + s_nq[j];
#pragma unroll
                for (int t = 0; t < TI; t++) {
                    double d = chebyshev_f64(pxi[t] - pxj, pyi[t] - pyj);
                    if (may_be_self) d = (j == i0 + t) ? inf : d;
#pragma unroll
                    for (int q = 0; q < K; q++) {
                        const double lo = min_f64(best[t][q], d);
                        if (q + 1 < K) d = max_f64(best[t][q], d);
                        best[t][q] = lo;
                    }
                }
            };
            if constexpr (DXT) {
                // x distances from the prepared table (scalar loads; its diagonal excludes the point itself): see
                // kraskov_direct_kernel
                const double* dx_col = prep_px + dxt_offset(cs) + i0;
                const int dxt_cols = (cs + 7) / 8 * 8;
#pragma unroll 2
                for (int j = 0; j < cs; j++) {
                    const double pyj = double(s_y[j * 64 + lane]) + s_nq[j];
                    const double* dx_row = dx_col + size_t(j) * size_t(dxt_cols);
#pragma unroll
                    for (int t = 0; t < TI; t++) {
                        double d = chebyshev_f64_sx(pyi[t] - pyj, dx_row[t]);
#pragma unroll
                        for (int q = 0; q < K; q++) {
                            const double lo = min_f64(best[t][q], d);
                            if (q + 1 < K) d = max_f64(best[t][q], d);
                            best[t][q] = lo;
                        }
                    }
                }
            } else {
#pragma unroll 2
            for (int j = 0; j < i0; j++) visit(j, false);
#pragma unroll 1
            for (int j = i0; j < i1; j++) visit(j, true);
#pragma unroll 2
            for (int j = i1; j < cs; j++) visit(j, false);
            }
#pragma unroll
            for (int t = 0; t < TI; t++) dk[t] = best[t][K - 1];
        } else {
#pragma unroll
            for (int t = 0; t < TI; t++) {
                double cur = -1.0, m = 0.0;
                int cnt = 0;
#pragma unroll 1
                for (int pass = 0; pass < kk; pass++) {
                    m = inf;
                    int c = 0;
#pragma unroll 2
                    for (int j = 0; j < cs; j++) {
                        const double pxj = s_px[j];
                        const double pyj = double(s_y[j * 64 + lane]) + s_nq[j];
                        const double d = fmax(fabs(pxi[t] - pxj), fabs(pyi[t] - pyj));
                        if (d > cur && j != i0 + t) {
                            c = (d < m) ? 1 : (d == m ? c + 1 : c);
                            m = fmin(m, d);
                        }
                    }
                    cnt += c;
                    if (cnt >= kk) break;
                    cur = m;
                }
                dk[t] = m;
            }
        }
        // ---- search radii
        if (estimator == 1) {
#pragma unroll
            for (int t = 0; t < TI; t++) rx[t] = ry[t] = dk[t] - kCountSlack;  // includeCenter, :196-197
        } else {
            // KSG-2: extents of the k+1 nearest points (incl. itself) per dimension (:485-499), then +slack (:198-199)
            double ex[TI], ey[TI];
#pragma unroll
            for (int t = 0; t < TI; t++) ex[t] = ey[t] = 0.0;
#pragma unroll 2
            for (int j = 0; j < cs; j++) {
                const double pxj = s_px[j];
                const double pyj = double(s_y[j * 64 + lane]) + s_nq[j];
#pragma unroll
                for (int t = 0; t < TI; t++) {
                    const double ax = fabs(pxi[t] - pxj), ay = fabs(pyi[t] - pyj);
                    const bool in = fmax(ax, ay) <= dk[t];
                    ex[t] = in ? fmax(ex[t], ax) : ex[t];
                    ey[t] = in ? fmax(ey[t], ay) : ey[t];
                }
            }
#pragma unroll
            for (int t = 0; t < TI; t++) {
                rx[t] = ex[t] + kCountSlack;
                ry[t] = ey[t] + kCountSlack;
            }
        }
        // ---- pass C: marginal counts  #{ j : c - r <= v_j < c + r }  (:201-233).  x: two binary searches in the sorted
        //      reference coordinates (like the reference); y: compares against every point.
        double loy[TI], hiy[TI];
        int cx[TI], cy[TI];
        {
            double bound[TI];
            int less[TI];
#pragma unroll
            for (int t = 0; t < TI; t++) bound[t] = pxi[t] + rx[t];
            count_less_batch<TI>(s_spx, cs, top, bound, cx);
#pragma unroll
            for (int t = 0; t < TI; t++) bound[t] = pxi[t] - rx[t];
            count_less_batch<TI>(s_spx, cs, top, bound, less);
#pragma unroll
            for (int t = 0; t < TI; t++) cx[t] -= less[t];
        }
#pragma unroll
        for (int t = 0; t < TI; t++) {
            loy[t] = pyi[t] - ry[t];
            hiy[t] = pyi[t] + ry[t];
            cy[t] = 0;
        }
#pragma unroll 2
        for (int j = 0; j < cs; j++) {
            const double pyj = double(s_y[j * 64 + lane]) + s_nq[j];
#pragma unroll
            for (int t = 0; t < TI; t++) cy[t] += (pyj >= loy[t] && pyj < hiy[t]) ? 1 : 0;
        }
#pragma unroll
        for (int t = 0; t < TI; t++) {
            if (i0 + t < cs) {
                int nx = cx[t] > 1 ? cx[t] : 1;
                int ny = cy[t] > 1 ? cy[t] : 1;
                if (estimator != 1) {
                    nx -= 1;  // psi(n - 1), psi(0) = NaN (pole)
                    ny -= 1;
                }
                sum_x += factor * s_psi[nx];
                sum_y += factor * s_psi[ny];
            }
        }
    }
    const double d = s_psi[cs];
    const double mi = -sum_x - sum_y + c_term + d;
    float res = float(mi);
    res = (res < 0.0f) ? 0.0f : res;  // std::max(float(mi), 0.0f), :443
    if (to_cc) res = mi_to_cc(res);
    if (is_nan) res = __uint_as_float(0x7FC00000u);
    if (v < num_voxels) store_result_nt(out + v, res);
}

// ---------------------------------------------------------------------------------------------------------------
// Kraskov for any member count and any k <= 128: no per-voxel tile at all.  A lane keeps, for TI points at a time, the
// K >= k smallest distances in registers (sorted insertion, 2K min/max per candidate) while it sweeps the members
// straight from the member volumes (coalesced 256 B per wave and member, as a tile read would be); the marginal counts
// take a second sweep.  One sweep per TI points instead of the k sweeps per point of the repeated-minimum selection:
// at 1000 members and k = 30 that is ~60x fewer distance evaluations.  LDS holds the voxel-independent tables only.
// ---------------------------------------------------------------------------------------------------------------
// Blocks of 4 waves share one copy of the tables in LDS (24 KB at 1000 members: with one wave per block LDS would cap
// the occupancy at 1.5 waves per SIMD).
// SYM (symmetric field mode): the X side is voxel dependent too -- x_e = members_x[e][v] + noise_ref[e] (prep_px then
// points at the noise_ref table), X counts by comparison like the Y counts instead of the binary search.
// Occupancy the register allocator is held to (waves per SIMD): what the instantiations reached before the work split
// below was added; left to itself the compiler now settles one step lower for K <= 2 and <8, 4>.
constexpr int direct_min_waves(int K, int TI, bool SYM) {
    if (SYM) return K <= 8 ? 2 : (K <= 32 ? 3 : 2);
    return K <= 2 ? 4 : (K == 3 ? 3 : (K == 4 ? 2 : (K <= 32 ? 3 : 1)));
}

// STAGE (not SYM, share mode): the block copies a tile's cs x 64 values into LDS once (each wave fetches every fourth
// member, non-temporal) and every sweep of every wave reads them from there -- one HBM / L2 read per value instead of
// 2 cs / TI + 1 reads served by L1 / L2 (measured at the fabric counters: 1.18x the algorithmic bytes at 4 points per sweep).
template <int K, int TI, bool SYM = false, bool DXT = false, bool STAGE = false>
__global__ __launch_bounds__(256, direct_min_waves(K, TI, SYM)) void kraskov_direct_kernel(const float* const* __restrict__ members,
                                                            const float* const* __restrict__ members_x,
                                                            const double* __restrict__ prep_px,
                                                            const double* __restrict__ table_psi,
                                                            const double* __restrict__ noise_query,
                                                            float* __restrict__ out, size_t num_voxels, int cs, int k,
                                                            int estimator, int to_cc, double c_term, int share) {
    extern __shared__ __align__(16) unsigned char smem[];
    double* s_px = reinterpret_cast<double*>(smem);  // member order
    double* s_spx = s_px + cs;                       // ascending
    double* s_nq = s_spx + cs;
    float* s_tile = reinterpret_cast<float*>(s_nq + cs);  // STAGE: [cs][64], the current tile's values
    // partial sums of the block's 4 tiles, one slot per (tile, wave): [16][64] x (sum_x, sum_y, NaN flag)
    __shared__ double s_sum_x[16 * 64];
    __shared__ double s_sum_y[16 * 64];
    __shared__ int s_nan[16 * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));  // uniform: i0 below stays in SGPRs
    const double* __restrict__ dxt = prep_px + dxt_offset(cs);  // DXT only
    const int dxt_cols = (cs + 7) / 8 * 8;
    for (int i = threadIdx.x; i < cs; i += 256) {
        s_px[i] = prep_px[i];                         // SYM: noise_ref[i]
        s_spx[i] = SYM ? 0.0 : prep_px[cs + i];
        s_nq[i] = noise_query[i];
    }
    __syncthreads();
    constexpr int JB = 16;  // member values fetched per batch of the sweeps
    const uint32_t bytes = uint32_t(num_voxels) * 4u;
    const int kk = k < cs - 1 ? k : cs - 1;
    int top = 1;
    while (top * 2 <= cs) top *= 2;
    const double factor = 1.0 / double(cs);
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    const size_t tiles = (num_voxels + 63) / 64;
    const size_t groups = (tiles + 3) / 4;
    const int passes = (cs + TI - 1) / TI;
    // A block takes 4 voxel tiles at a time = 4 * passes work items (tile, TI points).  share: the items are dealt to
    // the waves round-robin, so the four waves work on the SAME tile at the same time (for passes % 4 == 0; on two
    // neighbouring tiles otherwise) and the cs / TI re-reads of a tile's member values come from the CU's L1 / the
    // XCD's L2 -- with a tile per wave (share = 0) a CU's 12 waves cycle through 12 x cs x 256 B, ~6 MB per XCD, and
    // the re-reads miss the 4 MB L2 (measured: 48 GB fetched per 256^3 x 64 evaluation, 11x the volume).  The waves'
    // partial sums meet in LDS; the order of the additions depends on cs and TI only, not on the grid.
#pragma unroll 1
    for (size_t group = blockIdx.x; group < groups; group += gridDim.x) {
#pragma unroll
      for (int t4 = 0; t4 < 4; t4++) {
          s_sum_x[(t4 * 4 + wave) * 64 + lane] = 0.0;
          s_sum_y[(t4 * 4 + wave) * 64 + lane] = 0.0;
          s_nan[(t4 * 4 + wave) * 64 + lane] = 0;
      }
#pragma unroll 1
      for (int t4 = 0; t4 < 4; t4++) {
        const size_t tile = group * 4 + size_t(share ? t4 : wave);
        if (tile >= tiles || (!share && t4 > 0)) continue;
        const size_t v = tile * 64 + lane;
        const uint32_t off = v < num_voxels ? uint32_t(v) * 4u : kOutOfRangeOffset;
        if constexpr (STAGE) {
            __syncthreads();  // every wave has finished reading the previous tile
#pragma unroll 1
            for (int m0 = wave; m0 < cs; m0 += 4 * 8) {
                float stage[8];
#pragma unroll
                for (int q = 0; q < 8; q++) stage[q] = load_member_nt(members[m0 + 4 * q < cs ? m0 + 4 * q : cs - 1], bytes, off);
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (m0 + 4 * q < cs) s_tile[(m0 + 4 * q) * 64 + lane] = stage[q];
            }
            __syncthreads();
        }
        // one member value of this lane's voxel: from the staged tile or from memory (L1 / L2 after the first touch)
        auto value_of = [&](int m) -> float {
            if constexpr (STAGE) return s_tile[m * 64 + lane];
            else return load_member_cached(members[m], bytes, off);
        };
        bool is_nan = false;
        double sum_x = 0.0, sum_y = 0.0;
        // share: the 4 * passes items (tile, TI points) of the group are dealt to the waves round-robin
        const int first = share ? (wave - t4 * passes) & 3 : 0;
#pragma unroll 1
        for (int i0 = first * TI; i0 < cs; i0 += (share ? 4 : 1) * TI) {
            double pxi[TI], pyi[TI], dk[TI], rx[TI], ry[TI];
            double best[TI][K];
#pragma unroll
            for (int t = 0; t < TI; t++) {
                const int ii = (i0 + t < cs) ? i0 + t : cs - 1;
                const float y = value_of(ii);
                is_nan |= y != y;
                if constexpr (SYM) {
                    const float x = load_member_cached(members_x[ii], bytes, off);
                    is_nan |= x != x;
                    pxi[t] = double(x) + s_px[ii];
                } else {
                    pxi[t] = s_px[ii];
                }
                pyi[t] = double(y) + s_nq[ii];
#pragma unroll
                for (int q = 0; q < K; q++) best[t][q] = inf;
            }
            // ---- sweep A: the K smallest Chebyshev distances to OTHER points (MutualInformation.cpp:430-434)
            //      (the member values come in batches of JB loads issued back to back: one memory latency per batch)
#pragma unroll 1
            for (int j0 = 0; j0 < cs; j0 += JB) {
                float yb[JB], xb[SYM ? JB : 1];
#pragma unroll
                for (int u = 0; u < JB; u++) {
                    yb[u] = value_of(j0 + u < cs ? j0 + u : cs - 1);
                    if constexpr (SYM) xb[u] = load_member_cached(members_x[j0 + u < cs ? j0 + u : cs - 1], bytes, off);
                }
                // A candidate past the end gets +inf coordinates (one uniform select per candidate, not one per pair;
                // both coordinates: with only y = inf a point whose own y is +inf would see |inf - inf| = NaN and
                // v_max_f64 would fall back to the finite x distance).  The point itself is masked by overwriting the
                // HIGH dword of its distance with 0x7FEFFFFF -- a finite value beyond any real distance, whatever the
                // low dword holds: one v_cndmask per pair instead of the two of a 64-bit select.
                // the second half of a batch is skipped when the batch ends within its first half (a wave-uniform branch:
                // 56 members sweep 56 candidate slots, not 64)
                auto candidate_a = [&](int u) {
                    const int j = j0 + u;
                    const int jc = j < cs ? j : cs - 1;
                    const double pxj = j < cs ? (SYM ? double(xb[SYM ? u : 0]) + s_px[jc] : s_px[jc]) : inf;
                    const double pyj = (DXT || j < cs) ? double(yb[u]) + s_nq[jc] : inf;
                    // DXT: |px_i - px_j| comes from the prepared table through scalar loads -- uniform, so it rides in
                    // SGPRs as the third operand of the max; the table's diagonal and its rows past the end replace the
                    // self test and the end test (7 instead of 9 vector instructions per pair for K = 3).  (Requesting
                    // the row of candidate u + 1 before the pairs of candidate u changed nothing: 26.9 vs 27.0 ms.)
                    const double* dx_row = DXT ? dxt + size_t(j) * size_t(dxt_cols) + i0 : nullptr;
#pragma unroll
                    for (int t = 0; t < TI; t++) {
                        double d;
                        if constexpr (DXT) {
                            d = chebyshev_f64_sx(pyi[t] - pyj, dx_row[t]);
                        } else {
                            d = chebyshev_f64(pxi[t] - pxj, pyi[t] - pyj);
                            const uint64_t bits = uint64_t(__double_as_longlong(d));
                            const uint32_t hi = (j == i0 + t) ? 0x7FEFFFFFu : uint32_t(bits >> 32);
                            d = __longlong_as_double((long long)((uint64_t(hi) << 32) | uint32_t(bits)));
                        }
#pragma unroll
                        for (int q = 0; q < K; q++) {
                            const double lo = min_f64(best[t][q], d);
                            if (q + 1 < K) d = max_f64(best[t][q], d);
                            best[t][q] = lo;
                        }
                    }
                };
#pragma unroll
                for (int u = 0; u < JB / 2; u++) candidate_a(u);
                if (j0 + JB / 2 < cs) {
#pragma unroll
                    for (int u = JB / 2; u < JB; u++) candidate_a(u);
                }
            }
#pragma unroll
            for (int t = 0; t < TI; t++) {
                double sel = best[t][0];
#pragma unroll
                for (int q = 1; q < K; q++) sel = (q == kk - 1) ? best[t][q] : sel;
                dk[t] = sel;
            }
            if (estimator == 1) {
#pragma unroll
                for (int t = 0; t < TI; t++) rx[t] = ry[t] = dk[t] - kCountSlack;  // includeCenter, :196-197
            } else {
                double ex[TI], ey[TI];
#pragma unroll
                for (int t = 0; t < TI; t++) ex[t] = ey[t] = 0.0;
#pragma unroll 2
                for (int j = 0; j < cs; j++) {
                    const double pxj = SYM ? double(load_member_cached(members_x[j], bytes, off)) + s_px[j] : s_px[j];
                    const double pyj = double(value_of(j)) + s_nq[j];
#pragma unroll
                    for (int t = 0; t < TI; t++) {
                        const double ax = fabs(pxi[t] - pxj), ay = fabs(pyi[t] - pyj);
                        const bool in = fmax(ax, ay) <= dk[t];
                        ex[t] = in ? fmax(ex[t], ax) : ex[t];
                        ey[t] = in ? fmax(ey[t], ay) : ey[t];
                    }
                }
#pragma unroll
                for (int t = 0; t < TI; t++) {
                    rx[t] = ex[t] + kCountSlack;
                    ry[t] = ey[t] + kCountSlack;
                }
            }
            // ---- sweep C: marginal counts (:201-233)
            double loy[TI], hiy[TI], lox[TI], hix[TI];
            int cx[TI], cy[TI];
#pragma unroll
            for (int t = 0; t < TI; t++) {
                lox[t] = pxi[t] - rx[t];
                hix[t] = pxi[t] + rx[t];
                loy[t] = pyi[t] - ry[t];
                hiy[t] = pyi[t] + ry[t];
                cx[t] = cy[t] = 0;
            }
            if constexpr (!SYM) {
                int less[TI];
                count_less_batch<TI>(s_spx, cs, top, hix, cx);
                count_less_batch<TI>(s_spx, cs, top, lox, less);
#pragma unroll
                for (int t = 0; t < TI; t++) cx[t] -= less[t];
            }
#pragma unroll 1
            for (int j0 = 0; j0 < cs; j0 += JB) {
                float yb[JB], xb[SYM ? JB : 1];
#pragma unroll
                for (int u = 0; u < JB; u++) {
                    yb[u] = value_of(j0 + u < cs ? j0 + u : cs - 1);
                    if constexpr (SYM) xb[u] = load_member_cached(members_x[j0 + u < cs ? j0 + u : cs - 1], bytes, off);
                }
                auto candidate_c = [&](int u) {
                    const int jc = j0 + u < cs ? j0 + u : cs - 1;
                    // a candidate past the end is +inf: inside no [lo, hi) (hi is finite unless the voxel holds an
                    // infinity, and then the result is NaN or decided by the NaN flag anyway -- see below)
                    const double pyj = j0 + u < cs ? double(yb[u]) + s_nq[jc] : inf;
#pragma unroll
                    for (int t = 0; t < TI; t++) cy[t] += (pyj >= loy[t] && pyj < hiy[t]) ? 1 : 0;
                    if constexpr (SYM) {
                        const double pxj = j0 + u < cs ? double(xb[u]) + s_px[jc] : inf;
#pragma unroll
                        for (int t = 0; t < TI; t++) cx[t] += (pxj >= lox[t] && pxj < hix[t]) ? 1 : 0;
                    }
                };
#pragma unroll
                for (int u = 0; u < JB / 2; u++) candidate_c(u);
                if (j0 + JB / 2 < cs) {
#pragma unroll
                    for (int u = JB / 2; u < JB; u++) candidate_c(u);
                }
            }
#pragma unroll
            for (int t = 0; t < TI; t++) {
                if (i0 + t < cs) {
                    int nx = cx[t] > 1 ? cx[t] : 1;
                    int ny = cy[t] > 1 ? cy[t] : 1;
                    if (estimator != 1) {
                        nx -= 1;  // psi(n - 1), psi(0) = NaN (pole)
                        ny -= 1;
                    }
                    sum_x += factor * table_psi[nx];
                    sum_y += factor * table_psi[ny];
                }
            }
        }
        const int slot = ((share ? t4 : wave) * 4 + wave) * 64 + lane;
        s_sum_x[slot] = sum_x;
        s_sum_y[slot] = sum_y;
        s_nan[slot] = is_nan ? 1 : 0;
      }
      __syncthreads();
      {   // wave w finishes tile w of the group
        double sum_x = 0.0, sum_y = 0.0;
        int is_nan = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            sum_x += s_sum_x[(wave * 4 + w) * 64 + lane];
            sum_y += s_sum_y[(wave * 4 + w) * 64 + lane];
            is_nan |= s_nan[(wave * 4 + w) * 64 + lane];
        }
        const size_t v = (group * 4 + size_t(wave)) * 64 + lane;
        const double mi = -sum_x - sum_y + c_term + table_psi[cs];
        float res = float(mi);
        res = (res < 0.0f) ? 0.0f : res;  // std::max(float(mi), 0.0f), :443
        if (to_cc) res = mi_to_cc(res);
        if (is_nan) res = __uint_as_float(0x7FC00000u);
        if (v < num_voxels) store_result_nt(out + v, res);
      }
      __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Up to 64 members, k <= 4: the tile-free kernel with the y counts taken by binary search (r02).  A block works on ONE
// 64-voxel tile at a time: one of its four waves (rotating with the tile, so the extra work spreads over the SIMDs)
// loads the voxel's cs query values, sorts py_e = y_e + noise_e with a register min/max network and parks the sorted
// column in LDS [rank][lane]; after the barrier the four waves split the tile's points (8 per sweep) and take
// #{ j : lo <= py_j < hi } as lower_bound(hi) - lower_bound(lo) in the lane's column -- 2 x 7 probes instead of cs
// compare pairs (3.4 of the ~13 vector instructions per point pair).  Same counts as the compare sweep: the column
// holds the same py values, pads are +inf (below no finite bound; a NaN voxel is flagged, its result discarded).
// ---------------------------------------------------------------------------------------------------------------
// lower_bound in the lane's sorted column (base + lane, stride 64) for NCH bounds at once, steps outermost as above
template <int NS, int NCH>
__device__ __forceinline__ void count_less_column_batch(const double* col, const double (&v)[NCH], int (&pos)[NCH]) {
    constexpr bool kPow2 = (NS & (NS - 1)) == 0;
    // NS a power of two: the last entry is tested on its own, the search runs over the first NS - 1 (probe index < NS - 1)
    constexpr int kTop = kPow2 ? NS / 2 : (NS >= 32 ? 32 : (NS >= 16 ? 16 : 8));
#pragma unroll
    for (int c = 0; c < NCH; c++) pos[c] = 0;
#pragma unroll
    for (int step = kTop; step >= 1; step >>= 1) {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int idx = pos[c] + step;
            if constexpr (kPow2) {
                pos[c] = col[(idx - 1) * 64] < v[c] ? idx : pos[c];
            } else {
                const bool ok = idx <= NS;
                const double val = col[((ok ? idx : NS) - 1) * 64];
                pos[c] = (ok & (val < v[c])) ? idx : pos[c];
            }
        }
    }
    if constexpr (kPow2) {
#pragma unroll
        for (int c = 0; c < NCH; c++) pos[c] = col[(NS - 1) * 64] < v[c] ? NS : pos[c];
    }
}

// waves per SIMD the allocator is asked for; the sort holds NS doubles in registers
constexpr int sorted_min_waves(int K, int NS) {
    const int main_phase = K <= 2 ? 4 : (K == 3 ? 3 : 2);
    const int sort_phase = NS <= 48 ? 3 : 2;
    return main_phase < sort_phase ? main_phase : sort_phase;
}

template <int K, int NS, int NW>
__global__ __launch_bounds__(64 * NW, sorted_min_waves(K, NS)) void kraskov_sorted_kernel(
    const float* const* __restrict__ members, const double* __restrict__ prep_px, const double* __restrict__ table_psi,
    const double* __restrict__ noise_query, float* __restrict__ out, size_t num_voxels, int cs, int k, int estimator,
    int to_cc, double c_term) {
    constexpr int TI = 8;
    constexpr int JB = 16;
    extern __shared__ __align__(16) unsigned char smem[];
    double* s_px = reinterpret_cast<double*>(smem);  // member order
    double* s_spx = s_px + cs;                       // ascending
    double* s_nq = s_spx + cs;
    __shared__ double s_col[NS * 64];  // the tile's py values, ascending per lane: [rank][lane]
    __shared__ double s_sum_x[NW * 64];
    __shared__ double s_sum_y[NW * 64];
    __shared__ int s_nan[NW * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    for (int i = threadIdx.x; i < cs; i += 64 * NW) {
        s_px[i] = prep_px[i];
        s_spx[i] = prep_px[cs + i];
        s_nq[i] = noise_query[i];
    }
    __syncthreads();
    const uint32_t bytes = uint32_t(num_voxels) * 4u;
    const int kk = k < cs - 1 ? k : cs - 1;
    int top = 1;
    while (top * 2 <= cs) top *= 2;
    const double factor = 1.0 / double(cs);
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    const size_t tiles = (num_voxels + 63) / 64;
    const double* col = s_col + lane;
#pragma unroll 1
    for (size_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const size_t v = tile * 64 + lane;
        const uint32_t off = v < num_voxels ? uint32_t(v) * 4u : kOutOfRangeOffset;
        {   // every wave fetches a quarter of the members (one batch of loads) and parks py in the column, member order
            static_assert(NS % NW == 0, "members per wave");
            constexpr int Q = NS / NW;
            float yq[Q];
            bool is_nan = false;
#pragma unroll
            for (int u = 0; u < Q; u++) {
                const int j = wave * Q + u;
                yq[u] = load_member_cached(members[j < cs ? j : cs - 1], bytes, off);
            }
#pragma unroll
            for (int u = 0; u < Q; u++) {
                const int j = wave * Q + u;
                is_nan |= (j < cs) && (yq[u] != yq[u]);
                s_col[j * 64 + lane] = j < cs ? double(yq[u]) + s_nq[j < cs ? j : cs - 1] : inf;
            }
            s_nan[wave * 64 + lane] = is_nan ? 1 : 0;
        }
        __syncthreads();
        if (wave == int(tile % NW)) {  // one wave sorts the column in registers
            composite_t a[NS];
#pragma unroll
            for (int j = 0; j < NS; j++) a[j] = s_col[j * 64 + lane];
            SortNet<NS>::sort(a);
#pragma unroll
            for (int j = 0; j < NS; j++) s_col[j * 64 + lane] = a[j];
        }
        __syncthreads();
        double sum_x = 0.0, sum_y = 0.0;
#pragma unroll 1
        for (int i0 = wave * TI; i0 < cs; i0 += NW * TI) {
            double pxi[TI], pyi[TI], dk[TI], rx[TI], ry[TI];
            double best[TI][K];
            float yi[TI];
#pragma unroll
            for (int t = 0; t < TI; t++) yi[t] = load_member_cached(members[(i0 + t < cs) ? i0 + t : cs - 1], bytes, off);
#pragma unroll
            for (int t = 0; t < TI; t++) {
                const int ii = (i0 + t < cs) ? i0 + t : cs - 1;
                pxi[t] = s_px[ii];
                pyi[t] = double(yi[t]) + s_nq[ii];
#pragma unroll
                for (int q = 0; q < K; q++) best[t][q] = inf;
            }
            // ---- sweep A: as kraskov_direct_kernel
#pragma unroll 1
            for (int j0 = 0; j0 < cs; j0 += JB) {
                float yb[JB];
#pragma unroll
                for (int u = 0; u < JB; u++)
                    yb[u] = load_member_cached(members[j0 + u < cs ? j0 + u : cs - 1], bytes, off);
#pragma unroll
                for (int u = 0; u < JB; u++) {
                    const int j = j0 + u;
                    const int jc = j < cs ? j : cs - 1;
                    const double pxj = j < cs ? s_px[jc] : inf;
                    const double pyj = j < cs ? double(yb[u]) + s_nq[jc] : inf;
                    // (the scalar-loaded distance table of kraskov_direct_kernel made this kernel slower: 64 members
                    // k = 3 34.6 vs 31.2 ms at its two waves per SIMD)
#pragma unroll
                    for (int t = 0; t < TI; t++) {
                        double d = chebyshev_f64(pxi[t] - pxj, pyi[t] - pyj);
                        const uint64_t bits = uint64_t(__double_as_longlong(d));
                        const uint32_t hi = (j == i0 + t) ? 0x7FEFFFFFu : uint32_t(bits >> 32);
                        d = __longlong_as_double((long long)((uint64_t(hi) << 32) | uint32_t(bits)));
#pragma unroll
                        for (int q = 0; q < K; q++) {
                            const double lo = min_f64(best[t][q], d);
                            if (q + 1 < K) d = max_f64(best[t][q], d);
                            best[t][q] = lo;
                        }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < TI; t++) {
                double sel = best[t][0];
#pragma unroll
                for (int q = 1; q < K; q++) sel = (q == kk - 1) ? best[t][q] : sel;
                dk[t] = sel;
            }
            if (estimator == 1) {
#pragma unroll
                for (int t = 0; t < TI; t++) rx[t] = ry[t] = dk[t] - kCountSlack;  // includeCenter, :196-197
            } else {
                double ex[TI], ey[TI];
#pragma unroll
                for (int t = 0; t < TI; t++) ex[t] = ey[t] = 0.0;
#pragma unroll 2
                for (int j = 0; j < cs; j++) {
                    const double pxj = s_px[j];
                    const double pyj = double(load_member_cached(members[j], bytes, off)) + s_nq[j];
#pragma unroll
                    for (int t = 0; t < TI; t++) {
                        const double ax = fabs(pxi[t] - pxj), ay = fabs(pyi[t] - pyj);
                        const bool in = fmax(ax, ay) <= dk[t];
                        ex[t] = in ? fmax(ex[t], ax) : ex[t];
                        ey[t] = in ? fmax(ey[t], ay) : ey[t];
                    }
                }
#pragma unroll
                for (int t = 0; t < TI; t++) {
                    rx[t] = ex[t] + kCountSlack;
                    ry[t] = ey[t] + kCountSlack;
                }
            }
            // ---- marginal counts (:201-233): both by binary search (all of them first: 32 independent probe chains)
            int cx[TI], cy[TI];
            {
                double bound[TI];
                int less[TI];
#pragma unroll
                for (int t = 0; t < TI; t++) bound[t] = pxi[t] + rx[t];
                count_less_batch<TI>(s_spx, cs, top, bound, cx);
#pragma unroll
                for (int t = 0; t < TI; t++) bound[t] = pxi[t] - rx[t];
                count_less_batch<TI>(s_spx, cs, top, bound, less);
#pragma unroll
                for (int t = 0; t < TI; t++) cx[t] -= less[t];
#pragma unroll
                for (int t = 0; t < TI; t++) bound[t] = pyi[t] + ry[t];
                count_less_column_batch<NS, TI>(col, bound, cy);
#pragma unroll
                for (int t = 0; t < TI; t++) bound[t] = pyi[t] - ry[t];
                count_less_column_batch<NS, TI>(col, bound, less);
#pragma unroll
                for (int t = 0; t < TI; t++) {
                    cy[t] -= less[t];
                    cy[t] = cy[t] > 0 ? cy[t] : 0;  // an empty [lo, hi) with lo > hi
                }
            }
#pragma unroll
            for (int t = 0; t < TI; t++) {
                if (i0 + t < cs) {
                    int nx = cx[t] > 1 ? cx[t] : 1;
                    int ny = cy[t] > 1 ? cy[t] : 1;
                    if (estimator != 1) {
                        nx -= 1;  // psi(n - 1), psi(0) = NaN (pole)
                        ny -= 1;
                    }
                    sum_x += factor * table_psi[nx];
                    sum_y += factor * table_psi[ny];
                }
            }
        }
        s_sum_x[wave * 64 + lane] = sum_x;
        s_sum_y[wave * 64 + lane] = sum_y;
        __syncthreads();
        if (wave == int((tile + 1) % NW)) {
            double tx = 0.0, ty = 0.0;
            int any_nan = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                tx += s_sum_x[w * 64 + lane];
                ty += s_sum_y[w * 64 + lane];
                any_nan |= s_nan[w * 64 + lane];
            }
            const double mi = -tx - ty + c_term + table_psi[cs];
            float res = float(mi);
            res = (res < 0.0f) ? 0.0f : res;  // std::max(float(mi), 0.0f), :443
            if (to_cc) res = mi_to_cc(res);
            if (any_nan) res = __uint_as_float(0x7FC00000u);
            if (v < num_voxels) store_result_nt(out + v, res);
        }
    }
}

// LDS of kraskov_direct_kernel's partial sums: 16 (tile, wave) slots x 64 lanes x (2 doubles + 1 int)
constexpr size_t kDirectSumBytes = 16 * 64 * (2 * sizeof(double) + sizeof(int));

static int direct_share_tiles() {
    const char* e = getenv("CRF_KRASKOV_SHARE");  // tuning: 0 = a voxel tile per wave (the round-1 work split)
    return !(e && e[0] == '0');
}

void launch_kraskov_prep(const RefSource& ref, const float* const* d_members, int cs, const double* noise_ref,
                         double* d_prep, hipStream_t s) {
    hipLaunchKernelGGL(kraskov_prep_kernel, dim3(1), dim3(256), size_t(cs) * sizeof(double), s, ref, d_members, cs,
                       noise_ref, d_prep);
}

// O(cs) histogram kernel for any member count; hipErrorNotSupported when num_bins is too large for its LDS rows

hipError_t launch_mi_kraskov_direct(const float* const* d_members, int cs, size_t num_voxels, const RefSource& ref,
                                    const KraskovArgs& a, const double* d_tables, float* d_prep, float* d_out,
                                    hipStream_t s, hipEvent_t ev_begin, hipEvent_t ev_end, LaunchInfo* info) {
    const int kk = a.k < cs - 1 ? a.k : cs - 1;
    const size_t lds = size_t(3 * cs) * sizeof(double);
    if (kk > 128 || lds + kDirectSumBytes > 60 * 1024) return hipErrorNotSupported;
    const double* psi = d_tables;
    const double* noise_ref = d_tables + 2 * (cs + 1);
    const double* noise_query = noise_ref + cs;
    double* prep = reinterpret_cast<double*>(d_prep);
    if (ref.prepare()) launch_kraskov_prep(ref, d_members, cs, noise_ref, prep, s);
    if (!ref.run()) return hipGetLastError();
    const size_t tiles = (num_voxels + 63) / 64;
    const size_t groups = (tiles + 3) / 4;
    // one block per group of four voxel tiles up to 65536 blocks (r01 capped the grid at 4096: with ~768 blocks resident
    // that is 5.3 rounds of equally long blocks, the last round a third full)
    const char* cap_env = getenv("CRF_KRASKOV_GRID");  // tuning
    const size_t cap = cap_env && atoi(cap_env) > 0 ? size_t(atoi(cap_env)) : 65536;
    const unsigned blocks = unsigned(groups < cap ? groups : cap);
    const int share = direct_share_tiles();
    // The tile staged in LDS (table kernels, 4 points per sweep): every member value is fetched ONCE.  Up to 64 members the
    // 16 KB tile costs no occupancy -- 256^3: 64 members k = 3 / 4 26.3 / 29.7 ms against 26.2 / 31.6 ms, 32 members 7.0
    // against 7.3 -- beyond that it does (80 / 100 members k = 3: 46.4 / 80.0 against 40.2 / 64.6 ms); k = 1, 2 keep their
    // 8-point sweeps (64 members k = 2: 23.2 ms staged with 4 points against 21.6).  CRF_KRASKOV_STAGE=0/1 overrides.
    const char* stage_env = getenv("CRF_KRASKOV_STAGE");
    const bool stage_fits = lds + size_t(cs) * 256 + kDirectSumBytes <= 64 * 1024;
    const bool stage = share && stage_fits && (stage_env ? *stage_env == '1' : (cs <= 64 && kk >= 3));
    const size_t lds_stage = lds + size_t(cs) * 256;
    if (ev_begin) (void)hipEventRecord(ev_begin, s);
#define CRF_LAUNCH_DIRECT(K, TI)                                                                                    \
    hipLaunchKernelGGL((kraskov_direct_kernel<K, TI>), dim3(blocks), dim3(256), lds, s, d_members, nullptr, prep, psi, \
                       noise_query, d_out, num_voxels, cs, a.k, a.estimator, int(a.to_cc), a.c_term, share)
    // K = k exactly for the small k (the sorted insertion costs 2K - 1 min/max per candidate: K = 4 for k = 3 is 7
    // instead of 5; measured at 256^3 x 64, k = 3: <3, 8> 36.2 ms vs <4, 8> 48.6 ms)
    // up to 128 members the x distances can come from the prepared table (scalar loads)
    const char* dxt_env = getenv("CRF_KRASKOV_DXT");  // tuning: 0 = compute them per pair
    // (the table outgrows the 16 KB scalar cache early but keeps paying up to 112 members -- 256^3: 100 members k = 3 / 4
    // 67.8 / 79.5 vs 76.2 / 87.7 ms -- and for k = 2, 4 up to 128: 81.6 / 122.5 vs 85.6 / 129.9 ms; k = 1, 3 at 128: 74.5 /
    // 109.9 vs 73.3 / 107.6 ms)
    const bool table_pays = cs <= 112 || kk == 2 || kk == 4;
    const bool use_dxt = cs <= kDxtMaxMembers && ((dxt_env && *dxt_env == '1') || (table_pays && !(dxt_env && *dxt_env == '0')));
#define CRF_LAUNCH_DIRECT_DXT_TI(K, TI)                                                                                  \
    hipLaunchKernelGGL((kraskov_direct_kernel<K, TI, false, true>), dim3(blocks), dim3(256), lds, s, d_members, nullptr, \
                       prep, psi, noise_query, d_out, num_voxels, cs, a.k, a.estimator, int(a.to_cc), a.c_term, share)
#define CRF_LAUNCH_DIRECT_DXT(K) CRF_LAUNCH_DIRECT_DXT_TI(K, 8)
#define CRF_LAUNCH_STAGED(K)                                                                                              \
    hipLaunchKernelGGL((kraskov_direct_kernel<K, 4, false, true, true>), dim3(blocks), dim3(256), lds_stage, s, d_members, \
                       nullptr, prep, psi, noise_query, d_out, num_voxels, cs, a.k, a.estimator, int(a.to_cc), a.c_term, share)
    // 4 points per sweep instead of 8 for K = 3 and 4: half the registers (94 / 119 instead of 156 / 188: 5 / 4 waves per
    // SIMD instead of 3 / 2) and one s_load_dwordx8 per candidate row -- 256^3 with the table: k = 4 at 32 / 48 / 64 / 80
    // members 8.7 / 17.5 / 30.5 / 46.4 ms against 10.6 / 21.6 / 37.9 / 57.7 ms, k = 3 at 48 / 64 / 80: 14.8 / 25.8 / 39.3
    // against 15.4 / 27.0 / 41.7 ms; K = 1, 2 lose (64 members: 20.6 / 23.0 vs 18.0 / 21.1 ms).  Without the table
    // (beyond 80 members) K = 4 gains (128 members 129 vs 149 ms), K = 3 loses (117 vs 108 ms).  2 points per sweep
    // (77-79 registers, 6 waves) lose again: 64 members k = 3 / 4 31.5 / 34.8 ms.
    const char* ti4_env = getenv("CRF_KRASKOV_TI4");  // tuning: 1 = 4 points per sweep for every K <= 4, 0 = 8
    const bool ti4_table = ti4_env ? *ti4_env == '1' : kk >= 3;
    const bool ti4_k4 = ti4_env ? *ti4_env == '1' : true;
    const bool ti4_k3_plain = ti4_env && *ti4_env == '1';
    const bool ti4 = ti4_table;
    if (use_dxt && kk <= 4 && stage) {
        switch (kk) {
            case 1: CRF_LAUNCH_STAGED(1); break;
            case 2: CRF_LAUNCH_STAGED(2); break;
            case 3: CRF_LAUNCH_STAGED(3); break;
            default: CRF_LAUNCH_STAGED(4); break;
        }
    } else if (use_dxt && kk <= 4) {
        switch (kk) {
            case 1:
                if (ti4) {
                    CRF_LAUNCH_DIRECT_DXT_TI(1, 4);
                } else {
                    CRF_LAUNCH_DIRECT_DXT(1);
                }
                break;
            case 2:
                if (ti4) {
                    CRF_LAUNCH_DIRECT_DXT_TI(2, 4);
                } else {
                    CRF_LAUNCH_DIRECT_DXT(2);
                }
                break;
            case 3:
                if (ti4) {
                    CRF_LAUNCH_DIRECT_DXT_TI(3, 4);
                } else {
                    CRF_LAUNCH_DIRECT_DXT(3);
                }
                break;
            default:
                if (ti4) {
                    CRF_LAUNCH_DIRECT_DXT_TI(4, 4);
                } else {
                    CRF_LAUNCH_DIRECT_DXT(4);
                }
                break;
        }
    } else if (kk == 1) {
        CRF_LAUNCH_DIRECT(1, 8);
    } else if (kk == 2) {
        CRF_LAUNCH_DIRECT(2, 8);
    } else if (kk == 3) {
        if (ti4_k3_plain) {
            CRF_LAUNCH_DIRECT(3, 4);
        } else {
            CRF_LAUNCH_DIRECT(3, 8);
        }
    } else if (kk <= 4) {
        if (ti4_k4) {
            CRF_LAUNCH_DIRECT(4, 4);
        } else {
            CRF_LAUNCH_DIRECT(4, 8);
        }
    } else if (kk <= 8) {
        CRF_LAUNCH_DIRECT(8, 4);
    } else if (kk <= 16) {
        CRF_LAUNCH_DIRECT(16, 2);
    } else if (kk <= 32) {
        CRF_LAUNCH_DIRECT(32, 1);
    } else if (kk <= 64) {
        CRF_LAUNCH_DIRECT(64, 1);
    } else {
        CRF_LAUNCH_DIRECT(128, 1);
    }
#undef CRF_LAUNCH_DIRECT
#undef CRF_LAUNCH_DIRECT_DXT
#undef CRF_LAUNCH_DIRECT_DXT_TI
    if (ev_end) (void)hipEventRecord(ev_end, s);
    if (info) info->kernel_name = "kraskov_direct_kernel";
    return hipGetLastError();
}

// sorted-column kernel: cs <= 64, k <= 4
hipError_t launch_mi_kraskov_sorted(const float* const* d_members, int cs, size_t num_voxels, const RefSource& ref,
                                    const KraskovArgs& a, const double* d_tables, float* d_prep, float* d_out,
                                    hipStream_t s, hipEvent_t ev_begin, hipEvent_t ev_end, LaunchInfo* info) {
    const int kk = a.k < cs - 1 ? a.k : cs - 1;
    if (kk > 4 || kk < 1 || cs > 64) return hipErrorNotSupported;
    const size_t lds = size_t(3 * cs) * sizeof(double);
    const double* psi = d_tables;
    const double* noise_ref = d_tables + 2 * (cs + 1);
    const double* noise_query = noise_ref + cs;
    double* prep = reinterpret_cast<double*>(d_prep);
    if (ref.prepare()) launch_kraskov_prep(ref, d_members, cs, noise_ref, prep, s);
    if (!ref.run()) return hipGetLastError();
    const size_t tiles = (num_voxels + 63) / 64;
    const unsigned blocks = unsigned(tiles < 65536 ? tiles : 65536);
    if (ev_begin) (void)hipEventRecord(ev_begin, s);
#define CRF_LAUNCH_SORTED(K, NS)                                                                                     \
    hipLaunchKernelGGL((kraskov_sorted_kernel<K, NS, 2>), dim3(blocks), dim3(128), lds, s, d_members, prep, psi,      \
                       noise_query, d_out, num_voxels, cs, a.k, a.estimator, int(a.to_cc), a.c_term)
#define CRF_LAUNCH_SORTED_K(NS)                   \
    switch (kk) {                                 \
        case 1: CRF_LAUNCH_SORTED(1, NS); break;  \
        case 2: CRF_LAUNCH_SORTED(2, NS); break;  \
        case 3: CRF_LAUNCH_SORTED(3, NS); break;  \
        default: CRF_LAUNCH_SORTED(4, NS); break; \
    }
    if (cs <= 32) {
        CRF_LAUNCH_SORTED_K(32)
    } else if (cs <= 48) {
        CRF_LAUNCH_SORTED_K(48)
    } else {
        CRF_LAUNCH_SORTED_K(64)
    }
#undef CRF_LAUNCH_SORTED_K
#undef CRF_LAUNCH_SORTED
    if (ev_end) (void)hipEventRecord(ev_end, s);
    if (info) info->kernel_name = "kraskov_sorted_kernel";
    return hipGetLastError();
}

// symmetric field mode: X = d_members_x (reference field), Y = d_members_y (query field); KSG-1
hipError_t launch_mi_kraskov_symmetric(const float* const* d_members_x, const float* const* d_members_y, int cs,
                                       size_t num_voxels, int k, double c_term, bool to_cc, const double* d_tables,
                                       float* d_out, hipStream_t s) {
    if (cs == 1) return launch_fill(d_out, num_voxels, 1.0f, s);
    const int kk = k < cs - 1 ? k : cs - 1;
    const size_t lds = size_t(3 * cs) * sizeof(double);
    if (kk > 64 || lds + kDirectSumBytes > 60 * 1024) return hipErrorNotSupported;
    const double* psi = d_tables;
    const double* noise_ref = d_tables + 2 * (cs + 1);
    const double* noise_query = noise_ref + cs;
    const size_t tiles = (num_voxels + 63) / 64;
    const size_t groups = (tiles + 3) / 4;
    const unsigned blocks = unsigned(groups < 65536 ? groups : 65536);
    const int share = direct_share_tiles();
#define CRF_LAUNCH_SYM(K, TI)                                                                                        \
    hipLaunchKernelGGL((kraskov_direct_kernel<K, TI, true>), dim3(blocks), dim3(256), lds, s, d_members_y, d_members_x, \
                       noise_ref, psi, noise_query, d_out, num_voxels, cs, k, 1, int(to_cc), c_term, share)
    if (kk == 1) {
        CRF_LAUNCH_SYM(1, 8);
    } else if (kk == 2) {
        CRF_LAUNCH_SYM(2, 8);
    } else if (kk == 3) {
        CRF_LAUNCH_SYM(3, 8);
    } else if (kk <= 4) {
        CRF_LAUNCH_SYM(4, 8);
    } else if (kk <= 8) {
        CRF_LAUNCH_SYM(8, 4);
    } else if (kk <= 16) {
        CRF_LAUNCH_SYM(16, 2);
    } else if (kk <= 32) {
        CRF_LAUNCH_SYM(32, 1);
    } else {
        CRF_LAUNCH_SYM(64, 1);
    }
#undef CRF_LAUNCH_SYM
    return hipGetLastError();
}

hipError_t launch_mi_kraskov(const float* const* d_members, int cs, size_t num_voxels, const RefSource& ref,
                             const KraskovArgs& a, const double* d_tables, float* d_prep, float* d_out, hipStream_t s,
                             hipEvent_t ev_begin, hipEvent_t ev_end, LaunchInfo* info) {
    if (cs == 1) {
        if (!ref.run()) return hipSuccess;
        if (ev_begin) (void)hipEventRecord(ev_begin, s);
        hipError_t e = launch_fill(d_out, num_voxels, 1.0f, s);
        if (ev_end) (void)hipEventRecord(ev_end, s);
        if (info) info->kernel_name = "fill_kernel";
        return e;
    }
    const int kk = a.k < cs - 1 ? a.k : cs - 1;
    const char* force_direct = getenv("CRF_KRASKOV_DIRECT");  // tuning: the tile-free kernel for every k
    // The LDS-tile kernels below are instantiated for k <= 4 and hold a 256*cs-byte column per wave, which caps the
    // occupancy beyond ~80 members (measured at 256^3, k = 3: 80 members 71 vs 72 ms, 96: 112 vs 100 ms, 128: 226 vs
    // 171 ms, tile vs tile-free): the tile-free kernel takes over there and for every larger k.
    // r02 dispatch (256^3, profiles/r02_kraskov_tile_vs_direct.txt, profiles/tuning_r02.md).  Small member counts: the
    // LDS-column kernel with 8 points per sweep (three to four waves per SIMD, no batch padding): k = 3 at 32 / 40
    // members 7.2 / 11.5 ms vs 7.2 / 12.4 ms tile-free.  Beyond that the tile-free kernel with the scalar-loaded
    // x-distance table, 8 points per sweep for k = 1, 2 and 4 for k = 3, 4 (k = 3: 48 / 56 / 64 / 80 members 14.8 / 22.2 /
    // 25.8 / 39.3 ms; k = 4: 32 / 48 / 64 / 80 members 8.5 / 17.5 / 30.5 / 46.4 ms, LDS column 9.5 / 21.7 / - / -) -- its
    // four waves share a voxel tile, so its re-reads stay in L1 / L2 (see the kernel).  The sorted-column kernel (k = 4
    // at 64 members: 34.9 ms) is no longer the fastest anywhere and runs only when asked for (CRF_KRASKOV_SORTED=1).
    const char* sorted = getenv("CRF_KRASKOV_SORTED");  // tuning: 1 = wherever it exists, 0 = never
    if (kk <= 4 && cs <= 64 && !(sorted && *sorted == '0') && !(force_direct && *force_direct == '1') &&
        (sorted && *sorted == '1')) {
        hipError_t e = launch_mi_kraskov_sorted(d_members, cs, num_voxels, ref, a, d_tables, d_prep, d_out, s, ev_begin,
                                                ev_end, info);
        if (e != hipErrorNotSupported) return e;
    }
    const char* force_tile = getenv("CRF_KRASKOV_TILE");  // tuning: the LDS-column kernel wherever it exists
    const bool prefer_direct = kk <= 2 ? cs > 44 : (kk == 3 ? cs > 40 : (cs > 36 || (cs > 28 && cs <= 32)));
    if (kk > 4 || cs > 80 || (force_direct && *force_direct == '1') ||
        (prefer_direct && !(force_tile && *force_tile == '1'))) {
        hipError_t e = launch_mi_kraskov_direct(d_members, cs, num_voxels, ref, a, d_tables, d_prep, d_out, s, ev_begin,
                                                ev_end, info);
        if (e != hipErrorNotSupported) return e;
    }
    const double* psi = d_tables;
    const double* noise_ref = d_tables + 2 * (cs + 1);
    const double* noise_query = noise_ref + cs;
    double* prep = reinterpret_cast<double*>(d_prep);
    if (ref.prepare()) launch_kraskov_prep(ref, d_members, cs, noise_ref, prep, s);
    if (!ref.run()) return hipGetLastError();
    const unsigned blocks = unsigned((num_voxels + 63) / 64);
    size_t lds = size_t(4 * cs + 1 + ((cs + 1) & 1)) * sizeof(double) + size_t(cs) * 64 * sizeof(float);
    if (const char* pad = getenv("CRF_KRASKOV_LDS_PAD")) lds += size_t(atoi(pad));  // tuning: occupancy experiments
    if (ev_begin) (void)hipEventRecord(ev_begin, s);
    // 16 points per sweep only where the column caps the occupancy at two waves per SIMD anyway (more than 56 members)
    const char* narrow = getenv("CRF_KRASKOV_TI8");  // tuning: 8 points per sweep for every member count
    const bool wide = cs > 56 && (cs % 16 == 0 || cs % 16 > 8) && !(narrow && *narrow == '1');
    // the distance table pays here while it fits the 16 KB scalar cache (256^3, k = 3: 32 members 7.2 vs 8.1 ms; 40
    // members 12.0 vs 11.6 ms; 48 members 17.3 vs 16.4 ms)
    const char* dxt_env = getenv("CRF_KRASKOV_DXT");  // tuning: 0 = x distances computed per pair, 1 = table up to 80
    const bool use_dxt = (dxt_env && *dxt_env == '1') ? cs <= kDxtMaxMembers : (cs <= 32 && !(dxt_env && *dxt_env == '0'));
#define CRF_LAUNCH_KRASKOV(K, TI)                                                                                        \
    if (use_dxt && K > 0 && TI <= 8)                                                                                     \
        hipLaunchKernelGGL((mi_kraskov_kernel<K, TI, (K > 0 && TI <= 8)>), dim3(blocks), dim3(64), lds, s, d_members,    \
                           prep, psi, noise_query, d_out, num_voxels, cs, a.k, a.estimator, int(a.to_cc), a.c_term);     \
    else                                                                                                                 \
        hipLaunchKernelGGL((mi_kraskov_kernel<K, TI>), dim3(blocks), dim3(64), lds, s, d_members, prep, psi,            \
                           noise_query, d_out, num_voxels, cs, a.k, a.estimator, int(a.to_cc), a.c_term)
    switch (kk) {
        case 1: CRF_LAUNCH_KRASKOV(1, 8); break;
        // 16 points per sweep where the member count fills the last tile well: 253 VGPRs still give the 2 waves per SIMD
        // that the LDS column allows anyway (256^3 x 64, k = 3: 37.4 ms vs 39.7 ms at 8 points, 45.3 ms at 4)
        case 2:
            if (wide) {
                CRF_LAUNCH_KRASKOV(2, 16);
            } else {
                CRF_LAUNCH_KRASKOV(2, 8);
            }
            break;
        case 3:
            if (wide) {
                CRF_LAUNCH_KRASKOV(3, 16);
            } else {
                CRF_LAUNCH_KRASKOV(3, 8);
            }
            break;
        case 4: CRF_LAUNCH_KRASKOV(4, 4); break;
        default: CRF_LAUNCH_KRASKOV(0, 1); break;
    }
#undef CRF_LAUNCH_KRASKOV
    if (ev_end) (void)hipEventRecord(ev_end, s);
    if (info) info->kernel_name = "mi_kraskov_kernel";
    return hipGetLastError();
}

}  // namespace crf
