// kernels_rank.hip -- the two rank-based estimators on gfx950: Spearman and Kendall tau-b.
//
// Both are integer/ordering problems per voxel followed by a short fp32 tail, and both are reproduced bit-exactly:
//   Spearman  (CorrelationCalculator.cpp:902-962): fractional ranks of the voxel's cs values (computeRanks,
//             Correlation.cpp:277-303: a run of m equal values starting at 1-based rank R gets R + (m-1)/2), then
//             computePearson2<float>(referenceRanks, ranks, cs) (Correlation.cpp:141-174) in member order.
//   Kendall   (CorrelationCalculator.cpp:963-1025, computeKendall<int32_t>, Correlation.cpp:423-455):
//             num = n0 - n1 - n2 - 2*S_y (joint ties n3 := 0), tau = float(num) / (sqrtf(n0-n1) * sqrtf(n0-n2)),
//             S_y = strict inversions of y after sorting the (x, y) pairs lexicographically = number of pairs with
//             x_a < x_b and y_a > y_b.
//
// Mapping: ONE LANE PER VOXEL, the voxel's cs values live in that lane's registers as 64-bit composites
// (order-preserving key of the value << 32 | slot) and are sorted by a fully unrolled Batcher merge-exchange
// network (tools/gen_sortnet.py): static register indices only, no divergence, no LDS traffic in the sort.  Loads
// stay coalesced (a wave load = 64 consecutive voxels of one member).
//   Spearman: a forward and a backward scan over the sorted registers find each tie run [start, end]; 2*rank =
//     start + end + 2 is scattered as a 16-bit value to LDS row `slot` (column = lane, so the scatter is
//     conflict-free and private to the lane: LDS is used as per-lane indexed scratch, no barrier), read back in
//     member order and fed to the same fp32 three-pass Pearson tail as the Pearson kernel.
//   Kendall: members are LOADED in reference-sorted order (slot = position in the x order, permutation prepared once
//     per evaluation), so after sorting by (y, slot) the slot sequence is a permutation whose inversions are the
//     discordant pairs; they are counted with a per-lane bitset of seen slots (popcount of the bits above the
//     current slot's x-tie group), ties in y by run lengths, ties in x once per evaluation.
// NaN in the voxel's values -> quiet NaN (CorrelationCalculator.cpp:929-940, 1002-1013).
#include <cstdlib>

#include "crf_device.h"
#include "crf_internal.h"

namespace crf {

// Ends a sorting network for the compiler: an empty asm "modifies" every element, so nothing that follows is mixed into the
// network's last stages.  Left alone the compiler starts the scans / searches that consume the sorted array while the last
// exchanges are still pending, the live ranges of both overlap, and a network over N values needs ~2N registers
// (spearman_u32_kernel: 116 B of scratch per lane at 128 members, 25.1 -> 21.5 ms at 512^3 x 128 once they were gone;
// mi_binned_kernel: 40 B of scratch at 64 / 128 members gone, three waves per SIMD instead of two at 96 members: 2.33 ->
// 2.15 ms at 256^3).  Measured neutral for the other rank kernels, slightly negative (+2 %) for the two-field kernels of
// kernels_symmetric*.hip, which therefore do without it.
template <class T, int N>
__device__ __forceinline__ void pin_array(T (&a)[N]) {
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("" : "+v"(a[i]));
}

constexpr uint32_t kPadKey = 0xFFFFFFFFu;  // sorts after every real value (orderable_key(+inf) = 0xFF800000)

// NaNs sort to the ends: a positive NaN has an orderable key above key(+inf) = 0xFF800000, a negative NaN one below
// key(-inf) = 0x007FFFFF.  Looking at the smallest and the largest REAL key after the sort replaces a per-value
// `y != y` test (which the compiler sinks to the end of the kernel, keeping every raw value alive in a register).
__device__ __forceinline__ bool keys_hold_nan(uint32_t smallest_key, uint32_t largest_key) {
    return smallest_key < 0x007FFFFFu || largest_key > 0xFF800000u;
}

// ---------------------------------------------------------------------------------------------------------
// Reference-side preparation
// ---------------------------------------------------------------------------------------------------------
// Spearman: prep[e] = a_e = invNm1 * ((rx_e - mean) / sd) over the reference RANKS rx (CorrelationCalculator.cpp:859-865).
__global__ __launch_bounds__(256) void spearman_prep_kernel(RefSource src, const float* const* __restrict__ members,
                                                            int cs, float* __restrict__ prep) {
    extern __shared__ float prep_lds[];  // 2 * cs floats
    float* ref = prep_lds;
    float* rx = prep_lds + cs;
    __shared__ float sh[2];
    for (int i = threadIdx.x; i < cs; i += blockDim.x) ref[i] = load_ref(src, members, i);
    __syncthreads();
    for (int i = threadIdx.x; i < cs; i += blockDim.x) {
        const float v = ref[i];
        int s = 0;  // sum over j of sign(v_i - v_j); 2*rank_i = cs + 1 + s
        for (int j = 0; j < cs; j++) {
            const float w = ref[j];
            s += (w < v) ? 1 : ((v < w) ? -1 : 0);
        }
        rx[i] = 0.5f * float(cs + 1 + s);
    }
    __syncthreads();
    const float n = float(cs);
    const float invN = 1.0f / n;
    const float invNm1 = 1.0f / (n - 1.0f);
    if (threadIdx.x == 0) {
        float mean = 0.0f;
        for (int e = 0; e < cs; e++) mean += invN * rx[e];
        float var = 0.0f;
        for (int e = 0; e < cs; e++) {
            const float d = rx[e] - mean;
            var += invNm1 * d * d;
        }
        sh[0] = mean;
        sh[1] = sqrtf(var);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < cs; e += blockDim.x) prep[e] = invNm1 * ((rx[e] - sh[0]) / sh[1]);
    for (int e = cs + threadIdx.x; e < kMaxSortMembers; e += blockDim.x) prep[e] = 0.0f;  // pads: pearson_tail
}

// Kendall: prep as int32: [0, N) perm (slot -> member), [N, 2N) gend (slot -> last slot of its x-tie group),
// [2N] n1 = sum over x-tie groups t(t-1)/2 (computeTiesB, Correlation.cpp:305-329), [2N+1] 1 if x has ties.
__global__ __launch_bounds__(256) void kendall_prep_kernel(RefSource src, const float* const* __restrict__ members,
                                                           int cs, int n_pad, int* __restrict__ prep) {
    extern __shared__ float prep_lds[];  // cs floats
    float* ref = prep_lds;
    __shared__ int n1;
    if (threadIdx.x == 0) n1 = 0;
    for (int i = threadIdx.x; i < cs; i += blockDim.x) ref[i] = load_ref(src, members, i);
    __syncthreads();
    for (int i = threadIdx.x; i < n_pad; i += blockDim.x) {
        if (i >= cs) {
            prep[i] = 0;
            prep[n_pad + i] = i;
        }
    }
    for (int i = threadIdx.x; i < cs; i += blockDim.x) {
        const float v = ref[i];
        int less = 0, eq_before = 0, eq_total = 0;
        for (int j = 0; j < cs; j++) {
            const float w = ref[j];
            less += (w < v) ? 1 : 0;
            const int eq = (w == v) ? 1 : 0;
            eq_total += eq;
            eq_before += (j < i) ? eq : 0;
        }
        const int pos = less + eq_before;
        prep[pos] = i;
        prep[n_pad + pos] = less + eq_total - 1;
        if (eq_before) atomicAdd(&n1, eq_before);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        prep[2 * n_pad] = n1;
        prep[2 * n_pad + 1] = n1 != 0;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Spearman
// ---------------------------------------------------------------------------------------------------------
// LIST == false: lane = voxel blockIdx*64+lane, straight-line code.  LIST == true: the kernel walks the list
// todo[1 .. 1+todo[0]) of voxel indices (voxels a fast kernel deferred because they contain ties) with a grid-stride loop.
// Two instantiations rather than one kernel with an optional loop: inside a loop the compiler hoists the N member
// buffer descriptors (4 SGPRs each) out of it as loop invariants, overflows the 102 SGPRs and parks them in VGPR lanes
// -- 694 v_writelane / v_readlane around the 64 loads of the 64-member kernel, 15 % of its vector instructions, for a
// loop that ran exactly once.
template <int N, bool EXACT, int MIN_WAVES, bool LIST>
__global__ __launch_bounds__(64, MIN_WAVES) void spearman_kernel(const float* const* __restrict__ members,
                                                                 const float* __restrict__ prep,
                                                                 float* __restrict__ out, size_t num_voxels, int cs,
                                                                 const uint32_t* __restrict__ todo) {
    __shared__ uint16_t rank2[N * 64];
    // N = 16 serves 9..15 members only (<= 8 members: the N = 8 instantiation): its first 8 slots need no guard
    constexpr int SURE = (N == 16) ? 8 : 0;
    const int lane = threadIdx.x;
    const uint32_t bytes = uint32_t(num_voxels) * 4u;
    const uint32_t count = LIST ? todo[0] : 0u;
    for (uint32_t base = blockIdx.x * 64u; !LIST || base < count; base += gridDim.x * 64u) {
    const uint32_t item = base + lane;
    const bool active = LIST ? item < count : item < num_voxels;
    const size_t v = LIST ? (active ? todo[1 + item] : num_voxels) : item;
    const uint32_t byte_offset = uint32_t(v) * 4u;  // inactive lanes are out of range: they read 0 and store nothing

    composite_t a[N];
    {
        // all loads are issued before the first use; the guarded instantiation is branch free: a slot past cs loads at
        // kOutOfRangeOffset (value 0, no memory request) and gets the pad key, which sorts last
        float y[N];
#pragma unroll
        for (int e = 0; e < N; e++)
            y[e] = load_member_nt(members[(EXACT || e < SURE) ? e : (e < cs ? e : cs - 1)], bytes,
                                  (EXACT || e < SURE || e < cs) ? byte_offset : kOutOfRangeOffset);
#pragma unroll
        for (int e = 0; e < N; e++) {
            const float yc = y[e] + 0.0f;  // -0.0 -> +0.0 so that key equality is float equality
            a[e] = make_composite((EXACT || e < SURE || e < cs) ? orderable_key(yc) : kPadKey, uint32_t(e));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    SortNet<N>::sort(a);
    pin_array(a);  // the network ends here (crf_device.h)
    __builtin_amdgcn_sched_barrier(0);
    bool is_nan = composite_key(a[0]) < 0x007FFFFFu;
    if constexpr (EXACT) {
        is_nan |= composite_key(a[N - 1]) > 0xFF800000u;
    } else {
#pragma unroll
        for (int p = (SURE > 0 ? SURE : 0); p < N; p++)  // position cs - 1 holds the largest real key
            is_nan |= composite_key(a[p]) > ((p == cs - 1) ? 0xFF800000u : 0xFFFFFFFFu);
    }

    // Tie-free fast path: when no lane of the wave has two equal values among its members -- the rule for
    // continuous data; tied voxels cluster in space, so whole waves are free of them -- rank = position + 1 and the two
    // tie-run scans (~8 vector instructions per element) are skipped.  Same values as the scans would produce.
    bool scanned = true;
    {
        // (guarded instantiations: the pads tie with each other and are left out of the test; their rank slots are
        // masked when the ranks are read back)
        uint32_t tie_min = 0xFFFFFFFFu;  // min over neighbours of (key ^ previous key): 0 iff the voxel has a tie
#pragma unroll
        for (int p = 1; p < N; p++) {
            const uint32_t x = composite_key(a[p]) ^ composite_key(a[p - 1]);
            tie_min = min(tie_min, (EXACT || p < SURE || p < cs) ? x : 0xFFFFFFFFu);
        }
        if (!__any(tie_min == 0u)) {
            scanned = false;
#pragma unroll
            for (int p = 0; p < N; p++)
                if (EXACT || p < SURE || p < cs) rank2[(composite_low(a[p]) & 0xFFu) * 64 + lane] = uint16_t(2 * p + 2);
        }
    }
    if (scanned) {
    // (the pads form a tie run of their own behind the cs real elements: scanning them too is harmless and keeps the
    // guarded instantiation free of branches)
    // forward scan: first position of the tie run each sorted position belongs to, parked in bits 8..15 of the low word
    uint32_t run_start = 0;
#pragma unroll
    for (int p = 0; p < N; p++) {
        if (p > 0) {
            const bool same = composite_key(a[p]) == composite_key(a[p - 1]);
            run_start = same ? run_start : uint32_t(p);
        }
        a[p] = composite_or_low(a[p], run_start << 8);
        if ((p & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // bound the scheduler's window: register pressure
    }
    // backward scan: last position of the run; 2*rank = start + end + 2; scatter to the member's LDS row
    uint32_t run_end = 0;
#pragma unroll
    for (int p = N - 1; p >= 0; p--) {
        bool same = false;
        if (p < N - 1) same = composite_key(a[p]) == composite_key(a[p + 1]);  // real vs pad: equal only for a NaN
        run_end = same ? run_end : uint32_t(p);
        const uint32_t low = composite_low(a[p]);
        const uint32_t slot = low & 0xFFu;
        const uint32_t start = (low >> 8) & 0xFFu;
        rank2[slot * 64 + lane] = uint16_t(start + run_end + 2u);
        if ((p & 3) == 0) __builtin_amdgcn_sched_barrier(0);
    }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ranks back in member order (same lane wrote them: program order suffices, no barrier)
    // (all reads first, then the conversions: this file is compiled without the machine scheduler, and read / convert
    // per element was one LDS round trip per member -- 64 dependent waits, tools/isa_lds_chains.py)
    float r[N];
    uint32_t raw[N];
#pragma unroll
    for (int e = 0; e < N; e++) raw[e] = rank2[e * 64 + lane];
#pragma unroll
    for (int e = 0; e < N; e++) r[e] = (EXACT || e < SURE || e < cs) ? 0.5f * float(raw[e]) : 0.0f;
    float res = pearson_tail<N, EXACT, SURE>(r, prep, cs);
    if (is_nan) res = __uint_as_float(0x7FC00000u);
    if (active) store_result_nt(out + v, res);
    if constexpr (!LIST) break;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Kendall
// ---------------------------------------------------------------------------------------------------------
template <int N, bool EXACT, int MIN_WAVES, bool LIST>  // LIST: see spearman_kernel
__global__ __launch_bounds__(64, MIN_WAVES) void kendall_kernel(const float* const* __restrict__ members,
                                                                const int* __restrict__ prep, float* __restrict__ out,
                                                                size_t num_voxels, int cs,
                                                                const uint32_t* __restrict__ todo) {
    __shared__ uint8_t gend_lds[N];
    constexpr int SURE = (N == 16) ? 8 : 0;  // see spearman_kernel
    const int lane = threadIdx.x;
    const uint32_t bytes = uint32_t(num_voxels) * 4u;
    const bool x_ties = prep[2 * N + 1] != 0;  // wave-uniform
    if (x_ties) {
        for (int i = lane; i < N; i += 64) gend_lds[i] = uint8_t(prep[N + i]);
        __syncthreads();
    }
    const uint32_t count = LIST ? todo[0] : 0u;
    for (uint32_t base = blockIdx.x * 64u; !LIST || base < count; base += gridDim.x * 64u) {
    const uint32_t item = base + lane;
    const bool active = LIST ? item < count : item < num_voxels;
    const size_t v = LIST ? (active ? todo[1 + item] : num_voxels) : item;
    const uint32_t byte_offset = uint32_t(v) * 4u;

    composite_t a[N];
    {
        // slot e = e-th smallest reference value (prep[e] = member index; pads point at member 0 and load at
        // kOutOfRangeOffset: no memory request)
        float y[N];
#pragma unroll
        for (int e = 0; e < N; e++) y[e] = load_member_nt(members[prep[e]], bytes, (EXACT || e < SURE || e < cs) ? byte_offset : kOutOfRangeOffset);
#pragma unroll
        for (int e = 0; e < N; e++) {
            const float yc = y[e] + 0.0f;
            a[e] = make_composite((EXACT || e < SURE || e < cs) ? orderable_key(yc) : kPadKey, uint32_t(e));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    SortNet<N>::sort(a);
    pin_array(a);  // the network ends here (crf_device.h)
    __builtin_amdgcn_sched_barrier(0);
    bool is_nan = composite_key(a[0]) < 0x007FFFFFu;
    if constexpr (EXACT) {
        is_nan |= composite_key(a[N - 1]) > 0xFF800000u;
    } else {
#pragma unroll
        for (int p = (SURE > 0 ? SURE : 0); p < N; p++)
            is_nan |= composite_key(a[p]) > ((p == cs - 1) ? 0xFF800000u : 0xFFFFFFFFu);
    }

    constexpr int W = (N + 63) / 64;
    uint64_t seen[W];
#pragma unroll
    for (int w = 0; w < W; w++) seen[w] = 0ull;
    int32_t discordant = 0, n2 = 0, run = 0;
    uint32_t prev_key = 0;
#pragma unroll
    for (int p = 0; p < N; p++) {
        // Guarded instantiation, branch free: the pads (slots >= cs, pad key) sort behind the cs real elements in slot
        // order, so when one is visited every seen slot is below it -- it adds no discordant pair; its tie run is
        // masked out of n2.
        // ties in y: a run of t equal values contributes 0+1+...+(t-1) = t(t-1)/2
        const uint32_t key = composite_key(a[p]);
        if (p > 0) {
            run = ((EXACT || p < SURE || p < cs) && key == prev_key) ? run + 1 : 0;
            n2 += run;
        }
        prev_key = key;
        uint32_t slot = composite_low(a[p]) & 0xFFu;
        if ((p & 3) == 0) order_after(slot, seen[0]);  // keep the N mask computations from being hoisted en bloc
        const uint32_t g = x_ties ? uint32_t(gend_lds[slot]) : slot;  // last slot with the same x (pads: themselves)
        // already-seen slots (smaller y, or equal y and smaller slot) with strictly larger x: slot' > g
        if constexpr (W == 1) {
            discordant += __popcll(seen[0] & (0xFFFFFFFFFFFFFFFEull << g));
            seen[0] |= 1ull << slot;
        } else {
            const uint64_t gm = 0xFFFFFFFFFFFFFFFEull << (g & 63u);
            const uint64_t sbit = 1ull << (slot & 63u);
#pragma unroll
            for (int w = 0; w < W; w++) {
                const uint64_t mask = (uint32_t(w) > (g >> 6)) ? ~0ull : ((uint32_t(w) == (g >> 6)) ? gm : 0ull);
                discordant += __popcll(seen[w] & mask);
                seen[w] |= (uint32_t(w) == (slot >> 6)) ? sbit : 0ull;
            }
        }
        if ((p & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    const int32_t n = cs;
    const int32_t n0 = (n * (n - 1)) / 2;
    const int32_t n1 = prep[2 * N];
    const int32_t numerator = n0 - n1 - n2 - 2 * discordant;
    const float denominator = sqrtf(float(n0 - n1)) * sqrtf(float(n0 - n2));
    float res = float(numerator) / denominator;
    if (is_nan) res = __uint_as_float(0x7FC00000u);
    if (active) store_result_nt(out + v, res);
    if constexpr (!LIST) break;
    }
}

// ---------------------------------------------------------------------------------------------------------
// 64 < cs <= 128: split-sort kernels.
//
// 128 (key, slot) composites are 256 VGPRs: the monolithic kernels above then run one wave per SIMD with the overflow
// in AGPRs/scratch, and a lone wave issues fp32/integer VALU at half rate.  Here the voxel's values are handled as two
// chunks of 64 (A = slots 0..63, B = slots 64..cs-1), each sorted on its own in 128 VGPRs; A's sorted keys are parked
// in the lane's LDS column and every element of B finds its position among them by a 7-probe binary search, which
// gives all cross-chunk order information (rank of b in the union = own position + #{A < b}; the A side follows from
// a per-lane histogram of those positions and a prefix sum).  Peak ~175 VGPRs and 16 KB LDS per wave: two waves per
// SIMD.  The kernels handle TIE-FREE voxels only (the overwhelmingly common case for continuous data); a voxel with
// two equal values is appended to a todo list that the monolithic kernel then walks.
// ---------------------------------------------------------------------------------------------------------
// number of keys in the lane's sorted LDS column (CH keys, stride 64 dwords) that are < key.  `tie_min` keeps the
// running minimum of (k ^ key) over the probed lower-bound elements: it reaches 0 iff some key of the column equals
// `key` (two VALU ops, no lane-mask bookkeeping).
template <int CH>
__device__ __forceinline__ uint32_t lower_bound_col(const uint32_t* col, uint32_t key, uint32_t& tie_min) {
    uint32_t pos = 0;
#pragma unroll
    for (int s = CH / 2; s >= 1; s >>= 1) pos += (col[(pos + uint32_t(s) - 1u) * 64u] < key) ? uint32_t(s) : 0u;
    const uint32_t k = col[pos * 64u];
    tie_min = min(tie_min, k ^ key);
    return pos + ((k < key) ? 1u : 0u);
}

// The same search for G keys at once, the steps as the outer loop: a step issues G independent LDS reads before it
// waits.  One search after the other is a chain of log2(CH) + 1 dependent LDS round trips each; the compiler paired
// them at best (ISA of r01: ds_read, ds_read, s_waitcnt lgkmcnt(1), lgkmcnt(0), ...), 64 x 7 / 2 round trips per voxel.
template <int CH, int G>
__device__ __forceinline__ void lower_bound_col_batch(const uint32_t* col, const uint32_t (&key)[G], uint32_t& tie_min,
                                                      uint32_t (&less)[G]) {
    uint32_t pos[G];
#pragma unroll
    for (int g = 0; g < G; g++) pos[g] = 0;
#pragma unroll
    for (int s = CH / 2; s >= 1; s >>= 1) {
        uint32_t probe[G];
#pragma unroll
        for (int g = 0; g < G; g++) probe[g] = col[(pos[g] + uint32_t(s) - 1u) * 64u];
        __builtin_amdgcn_sched_barrier(0);  // (without the fences the instruction selector re-pairs the chains)
#pragma unroll
        for (int g = 0; g < G; g++) pos[g] += (probe[g] < key[g]) ? uint32_t(s) : 0u;
        __builtin_amdgcn_sched_barrier(0);
    }
    uint32_t k[G];
#pragma unroll
    for (int g = 0; g < G; g++) k[g] = col[pos[g] * 64u];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < G; g++) {
        tie_min = min(tie_min, k[g] ^ key[g]);
        less[g] = pos[g] + ((k[g] < key[g]) ? 1u : 0u);
    }
}

// loads slots base..base+CH-1 of the lane's voxel as composites (low word = slot - base), pads beyond cs
// SURE: the first SURE slots of the chunk are members whatever cs is (caller's contract): no guard on them
template <int CH, bool EXACT, bool PERMUTED, int SURE = 0>
__device__ __forceinline__ void load_chunk(composite_t (&a)[CH], const float* const* __restrict__ members,
                                           const int* __restrict__ perm, int base, int cs, uint32_t bytes,
                                           uint32_t byte_offset) {
    float y[CH];  // all loads first (slots past cs: out-of-range offset, no memory request), then the conversion
#pragma unroll
    for (int e = 0; e < CH; e++) {
        const bool real = EXACT || e < SURE || base + e < cs;
        const int slot = real ? base + e : cs - 1;
        y[e] = load_member_nt(PERMUTED ? members[perm[slot]] : members[slot], bytes,
                              real ? byte_offset : kOutOfRangeOffset);
    }
#pragma unroll
    for (int e = 0; e < CH; e++) {
        const float yc = y[e] + 0.0f;
        a[e] = make_composite((EXACT || e < SURE || base + e < cs) ? orderable_key(yc) : kPadKey, uint32_t(e));
    }
}

// CH < cs <= 2*CH (CH = 64: the 65..128 member kernel; CH = 32: 33..64 members)
// CHB: size of chunk B's sorting network (a multiple of 8, <= CH): CH < cs <= CH + CHB.  A cs of 72 thus sorts 64 + 8
// values instead of 64 + 64.  NPAD = member-count padding of the preparation tables (pearson_tail works on CH + CHB).
// The Spearman kernel requires the TIGHT network, CH + CHB - 8 < cs: only the last 8 slots of chunk B can be pads, so
// the per-element guards (wave-uniform branches that keep the binary searches of different elements from overlapping:
// +35-45 % at cs = 88 / 100) are compile-time true for all the others.
template <int CH, bool EXACT, int MIN_WAVES, int CHB = CH>
__global__ __launch_bounds__(64, MIN_WAVES) void spearman_split_kernel(const float* const* __restrict__ members,
                                                                       const float* __restrict__ prep,
                                                                       float* __restrict__ out, size_t num_voxels,
                                                                       int cs, uint32_t* __restrict__ todo) {
    __shared__ uint32_t lds[CH * 64];   // phase 1-3: sorted keys of chunk A [q][lane]; afterwards positions + histogram
    __shared__ uint8_t slotA[CH * 64];  // slot (0..CH-1) of the q-th smallest element of chunk A, [q][lane]
    static_assert((CH + CHB) * 64 + (CH + 1) * 64 <= CH * 64 * 4 && CHB <= CH, "positions + histogram must fit in the key array");
    const int lane = threadIdx.x;
    const size_t v = size_t(blockIdx.x) * 64 + lane;
    const bool active = v < num_voxels;
    const uint32_t byte_offset = uint32_t(v) * 4u, bytes = uint32_t(num_voxels) * 4u;
    const int nB = cs - CH;
    constexpr int SURE_B = CHB - 8;  // chunk B slots that are members for every cs of this instantiation
    bool is_nan = false;
    uint32_t tie_min = 0xFFFFFFFFu;  // min over compared key pairs of (k1 ^ k2): 0 iff the voxel has a tie
    uint32_t tie_or = 0u;            // OR over the neighbours of both sorted chunks: 0 = every member of a chunk equal
    uint32_t first_a = 0u, last_b = 0u;
    uint32_t infoB[CHB];  // for the p-th smallest element of chunk B: #{A < b_p} | slot << 8
    {
        composite_t a[CH];
        load_chunk<CH, true, false>(a, members, nullptr, 0, cs, bytes, byte_offset);
        __builtin_amdgcn_sched_barrier(0);
        SortNet<CH>::sort(a);
        pin_array(a);  // the network ends here (crf_device.h)
        __builtin_amdgcn_sched_barrier(0);
        uint32_t prev = 0;
#pragma unroll
        for (int q = 0; q < CH; q++) {
            const uint32_t key = composite_key(a[q]);
            if (q > 0) tie_min = min(tie_min, key ^ prev);
            if (q > 0) tie_or |= key ^ prev;
            if (q == 0) first_a = key;
            if (q == 0) is_nan |= key < 0x007FFFFFu;
            if (q == CH - 1) is_nan |= key > 0xFF800000u;
            prev = key;
            lds[q * 64 + lane] = key;
            slotA[q * 64 + lane] = uint8_t(composite_low(a[q]) & 0xFFu);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    uint32_t byte_offset_b = byte_offset;
    order_after(tie_min, byte_offset_b);  // chunk B is not touched before chunk A is fully consumed
    {
        composite_t b[CHB];
        load_chunk<CHB, EXACT, false, SURE_B>(b, members, nullptr, CH, cs, bytes, byte_offset_b);
        __builtin_amdgcn_sched_barrier(0);
        SortNet<CHB>::sort(b);
        pin_array(b);  // the network ends here (crf_device.h)
        __builtin_amdgcn_sched_barrier(0);
        uint32_t prev = 0;
        constexpr int G = 4;  // searches in flight (8: scratch in the 64 + 64 and 32 + 32 kernels)
        // The last 8 slots may be pads (wave-uniform, p >= nB).  When they are all members (cs = CH + CHB) the last batch
        // runs like the others; otherwise element by element behind wave-uniform branches (8 x 7 serial LDS round
        // trips; the select-masked batch form of it costs 270 B of scratch here).
#pragma unroll
        for (int p0 = 0; p0 < CHB; p0 += G) {
            const bool guarded = !EXACT && p0 >= SURE_B;  // compile time per batch
            if (!guarded || p0 + G <= nB) {
                uint32_t key[G], less[G];
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const int p = p0 + g;
                    key[g] = composite_key(b[p]);
                    if (p > 0) tie_min = min(tie_min, key[g] ^ prev);
                    if (p > 0) tie_or |= key[g] ^ prev;
                    if (p == 0) is_nan |= key[g] < 0x007FFFFFu;
                    // the largest key of the voxel sits at position nB - 1 (CHB - 1 in the exact instantiation)
                    if (EXACT ? p == CHB - 1 : guarded) is_nan |= (EXACT || p == nB - 1) && key[g] > 0xFF800000u;
                    prev = key[g];
                }
                lower_bound_col_batch<CH, G>(&lds[lane], key, tie_min, less);  // #{A < b_p}, 0..CH
#pragma unroll
                for (int g = 0; g < G; g++) infoB[p0 + g] = less[g] | ((composite_low(b[p0 + g]) & 0xFFu) << 8);
            } else {
#pragma unroll
                for (int p = p0; p < p0 + G; p++) {
                    infoB[p] = 0u;
                    if (p < nB) {
                        const uint32_t key = composite_key(b[p]);
                        if (p > 0) tie_min = min(tie_min, key ^ prev);
                        if (p > 0) tie_or |= key ^ prev;
                        if (p == 0) is_nan |= key < 0x007FFFFFu;
                        if (p == nB - 1) is_nan |= key > 0xFF800000u;
                        prev = key;
                        infoB[p] = lower_bound_col<CH>(&lds[lane], key, tie_min) | ((composite_low(b[p]) & 0xFFu) << 8);
                    }
                }
            }
        }
        last_b = prev;
    }
    __builtin_amdgcn_sched_barrier(0);
    // LDS is re-used from here on (this lane's binary searches are complete; LDS operations of a wave stay in order)
    uint8_t* pos_of = reinterpret_cast<uint8_t*>(lds);  // [slot 0..2CH-1][lane]: 0-based position in the union
    uint8_t* hist = pos_of + (CH + CHB) * 64;               // [0..CH][lane]: #{b : #{A < b} == t}
    // hist[t] = #{b : #{A < b} <= t} for the t that occur, 0 elsewhere: B is sorted, so `less` is non-decreasing in p
    // and the LAST p of a run of equal `less` leaves p + 1 = the cumulative count (LDS operations of a wave stay in
    // order).  Stores only: the r01 form incremented hist[less] (read, wait, add, write), 64 dependent LDS round trips.
#pragma unroll
    for (int t = 0; t <= CH; t++) hist[t * 64 + lane] = 0;
#pragma unroll
    for (int p = 0; p < CHB; p++) {
        if (EXACT || p < SURE_B || p < nB) {
            const uint32_t less = infoB[p] & 0xFFu;
            const uint32_t slot = infoB[p] >> 8;
            hist[less * 64 + lane] = uint8_t(p + 1);
            pos_of[(CH + slot) * 64 + lane] = uint8_t(uint32_t(p) + less);
        }
    }
    uint32_t below = 0;  // #{B < a_q} = #{b : #{A < b} <= q} (no ties): running maximum of the cumulative counts
#pragma unroll
    for (int q0 = 0; q0 < CH; q0 += 4) {  // 8 independent reads per step, then the dependent stores
        uint32_t cum[4], slot[4];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            cum[g] = uint32_t(hist[(q0 + g) * 64 + lane]);
            slot[g] = uint32_t(slotA[(q0 + g) * 64 + lane]);
        }
#pragma unroll
        for (int g = 0; g < 4; g++) {
            below = max(below, cum[g]);
            pos_of[slot[g] * 64 + lane] = uint8_t(uint32_t(q0 + g) + below);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    float r[CH + CHB];
    {
        uint32_t raw[CH + CHB];  // all reads first (see spearman_kernel)
#pragma unroll
        for (int e = 0; e < CH + CHB; e++) raw[e] = pos_of[e * 64 + lane];
#pragma unroll
        for (int e = 0; e < CH + CHB; e++) r[e] = (EXACT || e < CH + SURE_B || e < cs) ? float(raw[e] + 1u) : 0.0f;
    }
    // Every member equal (a mask; whole regions of such voxels in real ensembles): every fractional rank is (cs + 1) / 2
    // (Correlation.cpp:277-303) and the tail gives what the exact kernel would -- no deferral.
    const bool all_equal = tie_or == 0u && first_a == last_b && !is_nan;
    if (__builtin_amdgcn_ballot_w64(all_equal) != 0) {  // wave-uniform
        asm volatile("" ::: "memory");
        const float tied_rank = 0.5f * float(cs + 1);
#pragma unroll
        for (int e = 0; e < CH + CHB; e++)
            r[e] = (all_equal && (EXACT || e < CH + SURE_B || e < cs)) ? tied_rank : r[e];
    }
    float res = pearson_tail<CH + CHB, EXACT, CH + SURE_B>(r, prep, cs);
    if (is_nan) res = __uint_as_float(0x7FC00000u);
    if (active) {
        if (tie_min == 0u && !is_nan && !all_equal) {
            todo[1 + atomicAdd(&todo[0], 1u)] = uint32_t(v);
        } else {
            store_result_nt(out + v, res);
        }
    }
}

// u32 networks for the member counts between the ones crf_device.h instantiates (sortnet.inc has every multiple of 8)
#define CRF_CE(i, j)                           \
    {                                          \
        const uint32_t lo_ = a[i], hi_ = a[j]; \
        a[i] = lo_ < hi_ ? lo_ : hi_;          \
        a[j] = lo_ < hi_ ? hi_ : lo_;          \
    }
#define CRF_SORTNET32(NN)                                                        \
    template <>                                                                  \
    struct SortNet32<NN> {                                                       \
        static __device__ __forceinline__ void sort(uint32_t (&a)[NN]);          \
    };
CRF_SORTNET32(40)
CRF_SORTNET32(56)
CRF_SORTNET32(72)
CRF_SORTNET32(88)
CRF_SORTNET32(104)
CRF_SORTNET32(120)
#undef CRF_SORTNET32
__device__ __forceinline__ void SortNet32<40>::sort(uint32_t (&a)[40]) {
#define CRF_SORTNET_N 40
#include "sortnet.inc"
}
__device__ __forceinline__ void SortNet32<56>::sort(uint32_t (&a)[56]) {
#define CRF_SORTNET_N 56
#include "sortnet.inc"
}
__device__ __forceinline__ void SortNet32<72>::sort(uint32_t (&a)[72]) {
#define CRF_SORTNET_N 72
#include "sortnet.inc"
}
__device__ __forceinline__ void SortNet32<88>::sort(uint32_t (&a)[88]) {
#define CRF_SORTNET_N 88
#include "sortnet.inc"
}
__device__ __forceinline__ void SortNet32<104>::sort(uint32_t (&a)[104]) {
#define CRF_SORTNET_N 104
#include "sortnet.inc"
}
__device__ __forceinline__ void SortNet32<120>::sort(uint32_t (&a)[120]) {
#define CRF_SORTNET_N 120
#include "sortnet.inc"
}
#undef CRF_CE

// ---------------------------------------------------------------------------------------------------------------
// 65..128 members, ONE sorting network over 32-bit composites (r03).  The split kernel above sorts two chunks of 64-bit
// composites and pays for the cross-chunk order with 64 binary searches, a histogram pass and two LDS key columns
// (~7.9 k vector instructions per voxel at 128 members).  Here a composite is (key & ~127) | slot: the order-preserving
// key with its 7 low bits dropped, the member's slot in their place -- 128 composites fit in 128 registers and ONE
// Batcher network over u32 min / max sorts them.  Two members whose keys agree in the upper 25 bits (values within
// 2^-16 relative of each other: ~4 % of the voxels of N(0,1) data have such a pair) would be ordered by slot, which is
// wrong: the dropped bits are parked in the lane's LDS column, and after the sort every adjacent pair with equal upper
// bits is put right from them (a wave-uniform branch per position, taken for a few positions per wave).  A run of three
// such keys, or two keys that are equal in all 32 bits (a tie: fractional ranks), defers the voxel to the exact kernel
// through the todo list, like the split kernels do.  Ranks of a tie-free voxel are positions + 1.
// ---------------------------------------------------------------------------------------------------------------
template <int N, int MIN_WAVES, bool EXACT>  // EXACT: cs == N, no pads
__global__ __launch_bounds__(64, MIN_WAVES) void spearman_u32_kernel(const float* const* __restrict__ members,
                                                                     const float* __restrict__ prep,
                                                                     float* __restrict__ out, size_t num_voxels, int cs,
                                                                     uint32_t* __restrict__ todo) {
    static_assert(N <= 128 && N % 8 == 0, "slots are 7 bits");
    __shared__ uint8_t pos_of[N * 64];   // [slot][lane]: 0-based position of the member in the sorted order
    __shared__ uint8_t low_of[N * 64];   // [slot][lane]: the 7 key bits the composite dropped
    constexpr int SURE = EXACT ? N : N - 8;  // slots that are members for every cs of this instantiation (N - 8 < cs <= N)
    const int lane = threadIdx.x;
    const size_t v = size_t(blockIdx.x) * 64 + lane;
    const bool active = v < num_voxels;
    const uint32_t byte_offset = uint32_t(v) * 4u, bytes = uint32_t(num_voxels) * 4u;
    bool is_nan = false, defer = false;
    uint32_t key_min = 0xFFFFFFFFu, key_max = 0u;
    uint32_t a[N];  // the loaded values, then their composites, in place (one register per member)
#pragma unroll
    for (int e = 0; e < N; e++) {  // all loads first (slots past cs: out-of-range offset, no memory request)
        const bool real = e < SURE || e < cs;
        a[e] = __float_as_uint(load_member_nt(members[real ? e : cs - 1], bytes, real ? byte_offset : kOutOfRangeOffset));
    }
#pragma unroll
    for (int e = 0; e < N; e++) {
        const float yc = __uint_as_float(a[e]) + 0.0f;  // -0.0 -> +0.0: key equality is float equality
        const bool real = e < SURE || e < cs;
        const uint32_t okey = orderable_key(yc);
        // NaNs map beyond the infinities at either end of the key range; tracked on the keys, here and now (a float
        // compare `yc != yc` gets sunk to the end of the kernel by the compiler, which then keeps all N values alive)
        key_min = min(key_min, okey);
        key_max = max(key_max, real ? okey : 0u);
        const uint32_t key = real ? okey : 0xFFFFFFFFu;  // pads sort last
        low_of[e * 64 + lane] = uint8_t(key & 0x7Fu);
        a[e] = (key & ~0x7Fu) | uint32_t(e);
        if ((e & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    {   // pinned here (an empty asm that "modifies" the flag): otherwise the min / max chain is sunk to the kernel's end
        uint32_t nan_flag = (key_min < 0x007FFFFFu || key_max > 0xFF800000u) ? 1u : 0u;
        asm volatile("" : "+v"(nan_flag));
        is_nan = nan_flag != 0u;
    }
    __builtin_amdgcn_sched_barrier(0);
    SortNet32<N>::sort(a);
    pin_array(a);  // the network ends here (crf_device.h)
    __builtin_amdgcn_sched_barrier(0);
    // adjacent composites with equal upper bits: order them by the dropped bits (positions q, q + 1 both real members)
    bool prev_close = false;
#pragma unroll
    for (int q = 0; q + 1 < N; q++) {
        const bool both_real = q + 1 < SURE || q + 1 < cs;
        const bool close = both_real && ((a[q] ^ a[q + 1]) < 128u);
        if (__builtin_amdgcn_ballot_w64(close) != 0) {  // wave-uniform, rare
            asm volatile("" ::: "memory");  // a real branch: left alone the compiler if-converts the block (30 instructions per position)
            if (close) {
                const uint32_t sa = a[q] & 0x7Fu, sb = a[q + 1] & 0x7Fu;
                const uint32_t la = low_of[sa * 64 + lane], lb = low_of[sb * 64 + lane];
                defer |= (la == lb) | prev_close;  // a true tie, or a run of three: the exact kernel's business
                if (la > lb) {
                    const uint32_t t = a[q];
                    a[q] = a[q + 1];
                    a[q + 1] = t;
                }
            }
        }
        prev_close = close;
    }
    __builtin_amdgcn_sched_barrier(0);
    // the sorted composites are pinned here: otherwise the address arithmetic of the scatter below is hoisted into the
    // end of the network, one more live register per member (116 B of scratch per lane at 128 members)
#pragma unroll
    for (int q = 0; q < N; q++) asm volatile("" : "+v"(a[q]));
#pragma unroll
    for (int q = 0; q < N; q++) {
        pos_of[(a[q] & 0x7Fu) * 64 + lane] = uint8_t(q);
        if ((q & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    float r[N];
#pragma unroll
    for (int e = 0; e < N; e++) a[e] = pos_of[e * 64 + lane];  // all reads first, then the conversions (see spearman_kernel)
#pragma unroll
    for (int e = 0; e < N; e++) r[e] = (e < SURE || e < cs) ? float(a[e] + 1u) : 0.0f;
    // Every member equal (a mask; such voxels come in whole regions): all cs values tie, every fractional rank is
    // (cs + 1) / 2 (Correlation.cpp:277-303) and the tail below gives what the exact kernel would -- no deferral.
    // (key_min also sees the pads' 0.0: it can only make the test fail, never pass wrongly.)
    const bool all_equal = key_min == key_max && !is_nan;
    if (__builtin_amdgcn_ballot_w64(all_equal) != 0) {  // wave-uniform
        asm volatile("" ::: "memory");
        const float tied_rank = 0.5f * float(cs + 1);
#pragma unroll
        for (int e = 0; e < N; e++) r[e] = (all_equal && (e < SURE || e < cs)) ? tied_rank : r[e];
        defer = defer && !all_equal;
    }
    float res = pearson_tail<N, EXACT, SURE>(r, prep, cs);
    if (is_nan) res = __uint_as_float(0x7FC00000u);
    if (active) {
        if (defer && !is_nan) {
            todo[1 + atomicAdd(&todo[0], 1u)] = uint32_t(v);
        } else {
            store_result_nt(out + v, res);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 129..256 members (r03): TWO chunks of N <= 128 members, each sorted like spearman_u32_kernel sorts its one (u32
// composites, one register network, close pairs put right from the dropped bits), merged through LDS.  Before this the
// step from 128 to 129 members was a step to the O(cs^2) counting kernel (256^3: 3.1 -> 46 ms; 164 ms at 256 members).
//   chunk A = members [0, N): sorted, composites to the lane's LDS column comp_a;
//   chunk B = members [N, cs): sorted in registers; every element looks up how many of A lie below it (branch-free binary
//     search in comp_a on the upper 25 key bits, 8 elements per step so that 8 LDS reads are in flight; an A element
//     with the same upper bits is ordered from the dropped bits, a second one or a true tie defers the voxel);
//     rank - 1 of B's q-th element = q + that count, and a histogram over the counts gives the A side by a prefix sum:
//     #{B below A[p]} = #{q : count_q <= p};
//   ranks go to LDS by member (over the dropped bits, which are dead by then) and the fp32 tail -- computePearson2<float>
//   over the ranks in member order -- streams them from there in rolled loops.
// LDS per wave: 384 N + 512 bytes (N = 128: 49 KB, three waves per CU; N = 72: 28 KB, five).
// Voxels with ties (fractional ranks) or three keys within 2^-16 relative go through the todo list to the counting
// kernel (direct_rank_kernel, LIST form), as in the narrower kernels.
// ---------------------------------------------------------------------------------------------------------------
// perm: null = slot e of the chunk is member first + e; else member perm[first + e] (Kendall: reference-sorted order)
template <int N, int SURE>
__device__ __forceinline__ void sort_chunk_u32(const float* const* __restrict__ members, int first, int cs, uint32_t bytes,
                                               uint32_t byte_offset, uint8_t* __restrict__ low_col, uint32_t (&a)[N],
                                               bool& is_nan, bool& defer, uint32_t& key_lo, uint32_t& key_hi,
                                               const int* __restrict__ perm = nullptr) {
    // key_lo / key_hi: smallest key of the chunk (a pad's 0.0 included) and largest key of its members.  key_lo == key_hi
    // means every member of the chunk is equal (masks: whole regions of such voxels); the close-pair walk below is
    // skipped for such a lane (every position would take its branch) and the caller decides what the ties mean.
    uint32_t key_min = 0xFFFFFFFFu, key_max = 0u;
#pragma unroll
    for (int e = 0; e < N; e++) {  // all loads first (slots past cs: out-of-range offset, no memory request)
        const bool real = e < SURE || first + e < cs;
        const int idx = real ? first + e : cs - 1;
        a[e] = __float_as_uint(
            load_member_nt(members[perm ? perm[idx] : idx], bytes, real ? byte_offset : kOutOfRangeOffset));
    }
#pragma unroll
    for (int e = 0; e < N; e++) {
        const float yc = __uint_as_float(a[e]) + 0.0f;  // -0.0 -> +0.0: key equality is float equality
        const bool real = e < SURE || first + e < cs;
        const uint32_t okey = orderable_key(yc);
        key_min = min(key_min, okey);
        key_max = max(key_max, real ? okey : 0u);
        const uint32_t key = real ? okey : 0xFFFFFFFFu;  // pads sort last
        low_col[e * 64] = uint8_t(key & 0x7Fu);
        a[e] = (key & ~0x7Fu) | uint32_t(e);
        if ((e & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    {   // pinned here (see spearman_u32_kernel)
        uint32_t nan_flag = keys_hold_nan(key_min, key_max) ? 1u : 0u;
        asm volatile("" : "+v"(nan_flag));
        is_nan |= nan_flag != 0u;
        asm volatile("" : "+v"(key_min), "+v"(key_max));
    }
    key_lo = key_min;
    key_hi = key_max;
    const bool chunk_equal = key_min == key_max;
    __builtin_amdgcn_sched_barrier(0);
    SortNet32<N>::sort(a);
    pin_array(a);
    __builtin_amdgcn_sched_barrier(0);
    bool prev_close = false;
#pragma unroll
    for (int q = 0; q + 1 < N; q++) {
        const bool both_real = q + 1 < SURE || first + q + 1 < cs;
        const bool close = both_real && !chunk_equal && ((a[q] ^ a[q + 1]) < 128u);
        if (__builtin_amdgcn_ballot_w64(close) != 0) {  // wave-uniform, rare
            asm volatile("" ::: "memory");              // a real branch (see spearman_u32_kernel)
            if (close) {
                const uint32_t la = low_col[(a[q] & 0x7Fu) * 64], lb = low_col[(a[q + 1] & 0x7Fu) * 64];
                defer |= (la == lb) | prev_close;
                if (la > lb) {
                    const uint32_t t = a[q];
                    a[q] = a[q + 1];
                    a[q + 1] = t;
                }
            }
        }
        prev_close = close;
    }
    __builtin_amdgcn_sched_barrier(0);
    pin_array(a);
}

template <int N, int MIN_WAVES>
__global__ __launch_bounds__(64, MIN_WAVES) void spearman_pair_kernel(const float* const* __restrict__ members,
                                                                      const float* __restrict__ prep,
                                                                      float* __restrict__ out, size_t num_voxels, int cs,
                                                                      uint32_t* __restrict__ todo) {
    static_assert(N <= 128 && N % 8 == 0 && N >= 72, "two chunks of N members, slots are 7 bits");
    __shared__ uint32_t comp_a[N * 64];          // [position][lane]: chunk A's sorted composites
    __shared__ uint8_t low_of[(2 * N + 8) * 64];  // [member][lane]: dropped key bits; after the searches rows [N, 2N] hold
                                                  // the marks (see below), and in the end all rows hold rank - 1 by member
    uint8_t* const hist = low_of + N * 64;        // [count][lane], count = 0..N
    constexpr int TOP = N == 128 ? 128 : 64;  // largest power of two <= N
    constexpr int G = N % 16 == 0 ? 16 : 8;    // B elements whose searches run together (their LDS reads are in flight together)
    const int lane = threadIdx.x;
    const size_t v = size_t(blockIdx.x) * 64 + lane;
    const bool active = v < num_voxels;
    const uint32_t byte_offset = uint32_t(v) * 4u, bytes = uint32_t(num_voxels) * 4u;
    bool is_nan = false, defer = false;
    uint32_t a[N];
    // ---- chunk A = members [0, N) (all of them members: cs > 2 N - 16 >= N), then chunk B = members [N, 2 N), whose last
    //      16 slots may be padding.  ONE copy of the loads / keys / network / close-pair code runs twice (a rolled loop):
    //      the kernel is straight-line code far beyond the instruction cache and is paced by instruction fetch.
    uint32_t lo_a = 0u, hi_a = 0u, lo_b = 0u, hi_b = 0u;
#pragma unroll 1
    for (int chunk = 0; chunk < 2; chunk++) {
        uint32_t lo, hi;
        sort_chunk_u32<N, N - 16>(members, chunk * N, cs, bytes, byte_offset, low_of + chunk * N * 64 + lane, a, is_nan,
                                  defer, lo, hi);
        if (chunk == 0) {
            lo_a = lo;
            hi_a = hi;
#pragma unroll
            for (int q = 0; q < N; q++) comp_a[q * 64 + lane] = a[q];
        } else {
            lo_b = lo;
            hi_b = hi;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // every member of the voxel equal (a mask): all ranks (cs + 1) / 2, answered by the tail below; a single equal chunk
    // is a run of ties like any other: the counting kernel's business
    const bool all_equal = lo_a == hi_a && lo_b == hi_b && lo_a == lo_b && !is_nan;
    defer |= (lo_a == hi_a || lo_b == hi_b) && !all_equal;
    const int nb = cs - N;  // chunk B's members sit at the sorted positions [0, nb): pads sort last
    // (This file is compiled in source order, -enable-misched=0: every group of LDS reads below is written out before
    // the first use of any of them, so that the reads of a group are in flight together.)
    // ---- every B element: how many of A lie below it
#pragma unroll
    for (int q0 = 0; q0 < N; q0 += G) {
        if (q0 < N - 16 || q0 < nb) {  // uniform: a group of pads only
            uint32_t up[G], pos[G], val[G];
#pragma unroll
            for (int u = 0; u < G; u++) {
                up[u] = a[q0 + u] & ~0x7Fu;
                pos[u] = 0u;
            }
            // (up's slot bits are 0, so composite < up  <=>  upper bits < up: no masking of the probed value)
            if constexpr (N == 128) {  // a power of two: lower bound by halving, no index ever leaves the column
#pragma unroll
                for (int half = 64; half >= 1; half >>= 1) {
#pragma unroll
                    for (int u = 0; u < G; u++) val[u] = comp_a[(pos[u] + uint32_t(half - 1)) * 64 + lane];
#pragma unroll
                    for (int u = 0; u < G; u++) pos[u] = val[u] < up[u] ? pos[u] + uint32_t(half) : pos[u];
                }
#pragma unroll
                for (int u = 0; u < G; u++) val[u] = comp_a[pos[u] * 64 + lane];  // pos <= 127 here
#pragma unroll
                for (int u = 0; u < G; u++) pos[u] = val[u] < up[u] ? pos[u] + 1u : pos[u];
            } else {
#pragma unroll
                for (int step = TOP; step >= 1; step >>= 1) {
#pragma unroll
                    for (int u = 0; u < G; u++) {
                        const uint32_t idx = pos[u] + uint32_t(step);
                        val[u] = comp_a[((idx <= uint32_t(N) ? idx : uint32_t(N)) - 1u) * 64 + lane];
                    }
#pragma unroll
                    for (int u = 0; u < G; u++) {
                        const uint32_t idx = pos[u] + uint32_t(step);
                        pos[u] = (idx <= uint32_t(N) && val[u] < up[u]) ? idx : pos[u];
                    }
                }
            }
            // the A element at that position, if its upper bits are the same: order by the dropped bits
#pragma unroll
            for (int u = 0; u < G; u++) val[u] = comp_a[(pos[u] < uint32_t(N) ? pos[u] : uint32_t(N - 1)) * 64 + lane];
#pragma unroll
            for (int u = 0; u < G; u++) {
                const int q = q0 + u;
                const bool real = q < N - 16 || q < nb;
                const bool close = real && !all_equal && pos[u] < uint32_t(N) && ((val[u] ^ a[q]) < 128u);
                if (__builtin_amdgcn_ballot_w64(close) != 0) {  // wave-uniform, rare
                    asm volatile("" ::: "memory");
                    if (close) {
                        const uint32_t la = low_of[(val[u] & 0x7Fu) * 64 + lane];
                        const uint32_t lb = low_of[(N + (a[q] & 0x7Fu)) * 64 + lane];
                        bool more = false;  // a second A element with these upper bits
                        if (pos[u] + 1u < uint32_t(N)) more = ((comp_a[(pos[u] + 1u) * 64 + lane] ^ a[q]) < 128u);
                        defer |= (la == lb) | more;
                        if (la < lb) pos[u] += 1u;
                    }
                }
                if (real) a[q] = ((uint32_t(q) + pos[u]) << 7) | (a[q] & 0x7Fu);  // rank - 1, slot
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- the dropped bits are dead now: their B half becomes the marks.  The counts ascend with q, so the last q that
    //      leaves its mark at a count is the number of B elements with at most that count, minus one (writes only: no
    //      read-modify-write chain through LDS), and #{B below A[p]} = #{q : count_q <= p} is a running maximum.
    for (int i = lane; i < (N + 8) * 16; i += 64) reinterpret_cast<uint32_t*>(hist)[i] = 0u;
    __syncthreads();  // one wave per block: orders the cooperative clearing before the marks
#pragma unroll
    for (int q = 0; q < N; q++) {
        if (q < N - 16 || q < nb) hist[((a[q] >> 7) - uint32_t(q)) * 64 + lane] = uint8_t(q + 1);
        if ((q & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    // ---- ranks by member: A from the running maximum of the marks, B from the registers
    {
        uint32_t below = 0u;  // B elements below A[p] = #{q : count_q <= p}
#pragma unroll
        for (int p0 = 0; p0 < N; p0 += 8) {
            uint32_t h[8], ca[8];
#pragma unroll
            for (int u = 0; u < 8; u++) h[u] = hist[(p0 + u) * 64 + lane];
#pragma unroll
            for (int u = 0; u < 8; u++) ca[u] = comp_a[(p0 + u) * 64 + lane];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                below = max(below, h[u]);
                low_of[(ca[u] & 0x7Fu) * 64 + lane] = uint8_t(uint32_t(p0 + u) + below);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < N; q++) {
            if (q < N - 16 || q < nb) low_of[(N + (a[q] & 0x7Fu)) * 64 + lane] = uint8_t(a[q] >> 7);
            if ((q & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- computePearson2<float>(referenceRanks, ranks, cs) in member order (Correlation.cpp:141-174), ranks from LDS,
    //      16 at a time (a rank past cs reads as whatever the column holds and is not used)
    const uint8_t* rank_col = low_of + lane;
    const float tied_rank = 0.5f * float(cs + 1);  // every fractional rank of an all-equal voxel (Correlation.cpp:277-303)
    const float n = float(cs);
    const float invN = 1.0f / n;
    const float invNm1 = 1.0f / (n - 1.0f);
    float meanY = 0.0f;
#pragma unroll 1
    for (int e0 = 0; e0 < cs; e0 += 16) {
        uint32_t rk[16];
#pragma unroll
        for (int u = 0; u < 16; u++) rk[u] = rank_col[(e0 + u) * 64];
#pragma unroll
        for (int u = 0; u < 16; u++)
            if (e0 + u < cs) meanY += invN * (all_equal ? tied_rank : float(rk[u] + 1u));
    }
    float varY = 0.0f;
#pragma unroll 1
    for (int e0 = 0; e0 < cs; e0 += 16) {
        uint32_t rk[16];
#pragma unroll
        for (int u = 0; u < 16; u++) rk[u] = rank_col[(e0 + u) * 64];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const float d = (all_equal ? tied_rank : float(rk[u] + 1u)) - meanY;
            if (e0 + u < cs) varY += invNm1 * d * d;
        }
    }
    const float sdY = sqrtf(varY);
    float res = 0.0f;
    if (__all(exact_div_guard(meanY, sdY))) {
        const float rcp = 1.0f / sdY;
#pragma unroll 1
        for (int e0 = 0; e0 < cs; e0 += 16) {
            uint32_t rk[16];
#pragma unroll
            for (int u = 0; u < 16; u++) rk[u] = rank_col[(e0 + u) * 64];
#pragma unroll
            for (int u = 0; u < 16; u++)
                if (e0 + u < cs) res += prep[e0 + u] * exact_div((all_equal ? tied_rank : float(rk[u] + 1u)) - meanY, sdY, rcp);
        }
    } else {
#pragma unroll 4
        for (int e = 0; e < cs; e++) res += prep[e] * (((all_equal ? tied_rank : float(uint32_t(rank_col[e * 64]) + 1u)) - meanY) / sdY);
    }
    if (is_nan) res = __uint_as_float(0x7FC00000u);
    if (active) {
        if (defer && !is_nan) {
            todo[1 + atomicAdd(&todo[0], 1u)] = uint32_t(v);
        } else {
            store_result_nt(out + v, res);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Kendall at 129..256 members (r03), on the same two sorted chunks as spearman_pair_kernel.  Members are taken in
// reference-sorted order (slot = position in the x order, kendall_prep_kernel), chunk A = the first N positions, chunk B the
// rest, each sorted by y.  Discordant pairs (x_i < x_j beyond i's x-tie group, y_i > y_j) =
//   inversions of the slot sequence inside A + inside B (a 128-bit "seen" set per lane: walking a chunk in ascending y,
//   an element is discordant with every seen slot above its x-tie group's end)
//   + for every B element the A elements above it in y: N - #{A below it}, the count the merge search yields.
// Voxels with y ties or three close keys are deferred to the counting kernel (n2 = 0 for the others).  An x-tie group
// that straddles the chunk boundary (its pairs would have to be taken out of the cross term) defers every voxel: the
// reference vector decides, once per evaluation, and it is rare.
// ---------------------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ int32_t chunk_inversions_u32(const uint32_t (&a)[N], const uint8_t* __restrict__ gend_tab,
                                                        int first, int cs) {
    uint32_t w0 = 0u, w1 = 0u, w2 = 0u, w3 = 0u;  // seen slots 0..127
    int32_t inv = 0;
#pragma unroll
    for (int q0 = 0; q0 < N; q0 += 8) {
        uint32_t ge[8];
#pragma unroll
        for (int u = 0; u < 8; u++) ge[u] = gend_tab[a[q0 + u] & 0x7Fu];  // reads first (source order is the schedule)
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int q = q0 + u;
            if (q < N - 16 || first + q < cs) {  // pads sort last: positions past the chunk's members
                const uint32_t s = a[q] & 0x7Fu;
                const uint32_t wi = ge[u] >> 5, sh = ge[u] & 31u;
                const uint32_t sel = wi == 0u ? w0 : wi == 1u ? w1 : wi == 2u ? w2 : w3;
                uint32_t c = __builtin_popcount((sel >> sh) >> 1);  // seen slots above the group's end, same word
                c += wi < 1u ? __builtin_popcount(w1) : 0u;
                c += wi < 2u ? __builtin_popcount(w2) : 0u;
                c += wi < 3u ? __builtin_popcount(w3) : 0u;
                inv += int32_t(c);
                const uint32_t ws = s >> 5, bit = 1u << (s & 31u);
                w0 |= ws == 0u ? bit : 0u;
                w1 |= ws == 1u ? bit : 0u;
                w2 |= ws == 2u ? bit : 0u;
                w3 |= ws == 3u ? bit : 0u;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return inv;
}

template <int N, int MIN_WAVES>
__global__ __launch_bounds__(64, MIN_WAVES) void kendall_pair_kernel(const float* const* __restrict__ members,
                                                                     const int* __restrict__ prep,
                                                                     float* __restrict__ out, size_t num_voxels, int cs,
                                                                     uint32_t* __restrict__ todo) {
    static_assert(N <= 128 && N % 8 == 0 && N >= 72, "two chunks of N slots, 7 slot bits");
    __shared__ uint32_t comp_a[N * 64];     // [position][lane]: chunk A's sorted composites
    __shared__ uint8_t low_of[2 * N * 64];  // [slot][lane]: dropped key bits
    __shared__ uint8_t gend_tab[2 * N];     // per chunk: slot -> end of its x-tie group, as a slot of the chunk
    constexpr int TOP = N == 128 ? 128 : 64;
    constexpr int G = N % 16 == 0 ? 16 : 8;
    const int lane = threadIdx.x;
    const size_t v = size_t(blockIdx.x) * 64 + lane;
    const bool active = v < num_voxels;
    const uint32_t byte_offset = uint32_t(v) * 4u, bytes = uint32_t(num_voxels) * 4u;
    const int* gend = prep + cs;
    for (int i = lane; i < 2 * N; i += 64) {
        const int first = i < N ? 0 : N;
        const int g = i < cs ? gend[i] - first : N - 1;
        gend_tab[i] = uint8_t(g < N - 1 ? g : N - 1);
    }
    bool is_nan = false;
    bool defer = gend[N - 1] >= N;  // an x-tie group across the chunk boundary: the counting kernel for every voxel
    __syncthreads();
    uint32_t a[N];
    int32_t discordant = 0;
    uint32_t lo_a = 0u, hi_a = 0u, lo_b = 0u, hi_b = 0u;
#pragma unroll 1
    for (int chunk = 0; chunk < 2; chunk++) {  // one copy of the sort and of the inversion walk, run twice
        uint32_t lo, hi;
        sort_chunk_u32<N, N - 16>(members, chunk * N, cs, bytes, byte_offset, low_of + chunk * N * 64 + lane, a, is_nan,
                                  defer, lo, hi, prep);
        discordant += chunk_inversions_u32<N>(a, gend_tab + chunk * N, chunk * N, cs);
        if (chunk == 0) {
            lo_a = lo;
            hi_a = hi;
#pragma unroll
            for (int q = 0; q < N; q++) comp_a[q * 64 + lane] = a[q];
        } else {
            lo_b = lo;
            hi_b = hi;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // every member of the voxel equal (a mask): n2 = n0 and no discordant pair, answered below; a single equal chunk is a
    // run of y ties like any other: the counting kernel's business
    const bool all_equal = lo_a == hi_a && lo_b == hi_b && lo_a == lo_b && !is_nan;
    defer |= (lo_a == hi_a || lo_b == hi_b) && !all_equal;
    const int nb = cs - N;
    // ---- every B element: the A elements above it in y (the merge search of spearman_pair_kernel)
#pragma unroll
    for (int q0 = 0; q0 < N; q0 += G) {
        if (q0 < N - 16 || q0 < nb) {
            uint32_t up[G], pos[G], val[G];
#pragma unroll
            for (int u = 0; u < G; u++) {
                up[u] = a[q0 + u] & ~0x7Fu;
                pos[u] = 0u;
            }
            if constexpr (N == 128) {
#pragma unroll
                for (int half = 64; half >= 1; half >>= 1) {
#pragma unroll
                    for (int u = 0; u < G; u++) val[u] = comp_a[(pos[u] + uint32_t(half - 1)) * 64 + lane];
#pragma unroll
                    for (int u = 0; u < G; u++) pos[u] = val[u] < up[u] ? pos[u] + uint32_t(half) : pos[u];
                }
#pragma unroll
                for (int u = 0; u < G; u++) val[u] = comp_a[pos[u] * 64 + lane];
#pragma unroll
                for (int u = 0; u < G; u++) pos[u] = val[u] < up[u] ? pos[u] + 1u : pos[u];
            } else {
#pragma unroll
                for (int step = TOP; step >= 1; step >>= 1) {
#pragma unroll
                    for (int u = 0; u < G; u++) {
                        const uint32_t idx = pos[u] + uint32_t(step);
                        val[u] = comp_a[((idx <= uint32_t(N) ? idx : uint32_t(N)) - 1u) * 64 + lane];
                    }
#pragma unroll
                    for (int u = 0; u < G; u++) {
                        const uint32_t idx = pos[u] + uint32_t(step);
                        pos[u] = (idx <= uint32_t(N) && val[u] < up[u]) ? idx : pos[u];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < G; u++) val[u] = comp_a[(pos[u] < uint32_t(N) ? pos[u] : uint32_t(N - 1)) * 64 + lane];
#pragma unroll
            for (int u = 0; u < G; u++) {
                const int q = q0 + u;
                const bool real = q < N - 16 || q < nb;
                const bool close = real && !all_equal && pos[u] < uint32_t(N) && ((val[u] ^ a[q]) < 128u);
                if (__builtin_amdgcn_ballot_w64(close) != 0) {  // wave-uniform, rare
                    asm volatile("" ::: "memory");
                    if (close) {
                        const uint32_t la = low_of[(val[u] & 0x7Fu) * 64 + lane];
                        const uint32_t lb = low_of[(N + (a[q] & 0x7Fu)) * 64 + lane];
                        bool more = false;
                        if (pos[u] + 1u < uint32_t(N)) more = ((comp_a[(pos[u] + 1u) * 64 + lane] ^ a[q]) < 128u);
                        defer |= (la == lb) | more;
                        if (la < lb) pos[u] += 1u;
                    }
                }
                if (real) discordant += int32_t(uint32_t(N) - pos[u]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // tau-b (Correlation.cpp:423-455): n2 = 0 for a voxel without y ties
    const int32_t n = cs;
    const int32_t n0 = (n * (n - 1)) / 2;
    const int32_t n1 = prep[2 * cs];
    const int32_t n2 = all_equal ? n0 : 0;
    const int32_t numerator = n0 - n1 - n2 - (all_equal ? 0 : 2 * discordant);
    const float denominator = sqrtf(float(n0 - n1)) * sqrtf(float(n0 - n2));
    float res = float(numerator) / denominator;
    if (is_nan) res = __uint_as_float(0x7FC00000u);
    if (active) {
        if (defer && !is_nan && !all_equal) {
            todo[1 + atomicAdd(&todo[0], 1u)] = uint32_t(v);
        } else {
            store_result_nt(out + v, res);
        }
    }
}

// discordant pairs inside one sorted chunk: inversions of the slot sequence (slots 0..CH-1), one 64-bit "seen" set
template <int CH, int SURE = 0>
__device__ __forceinline__ int32_t chunk_inversions(const composite_t (&a)[CH], int count, bool exact) {
    int32_t inv = 0;
    if constexpr (CH <= 32) {
        // slots 0..31: a 32-bit set (one shift, and, popcount, or each instead of the two-dword forms: ~4 vector
        // instructions per element less)
        uint32_t seen = 0u;
#pragma unroll
        for (int p = 0; p < CH; p++) {
            if (exact || p < SURE || p < count) {
                uint32_t slot = composite_low(a[p]) & 0xFFu;
                if ((p & 3) == 0) order_after(slot, seen);  // keep the mask computations from being hoisted en bloc
                inv += __popc(seen & (0xFFFFFFFEu << slot));  // already-seen slots above this one
                seen |= 1u << slot;
            }
        }
    } else {
        uint64_t seen = 0ull;
#pragma unroll
        for (int p = 0; p < CH; p++) {
            if (exact || p < SURE || p < count) {
                uint32_t slot = composite_low(a[p]) & 0xFFu;
                if ((p & 3) == 0) order_after(slot, seen);  // keep the mask computations from being hoisted en bloc
                inv += __popcll(seen & (0xFFFFFFFFFFFFFFFEull << slot));  // already-seen slots above this one
                seen |= 1ull << slot;
            }
        }
    }
    return inv;
}

// CHB, NPAD: see spearman_split_kernel; the preparation tables use the stride NPAD = pad_pow2(cs)
template <int CH, bool EXACT, int MIN_WAVES, int CHB = CH, int NPAD = 2 * CH>
__global__ __launch_bounds__(64, MIN_WAVES) void kendall_split_kernel(const float* const* __restrict__ members,
                                                                      const int* __restrict__ prep,
                                                                      float* __restrict__ out, size_t num_voxels,
                                                                      int cs, uint32_t* __restrict__ todo) {
    __shared__ uint32_t keysA[CH * 64];
    const int lane = threadIdx.x;
    const size_t v = size_t(blockIdx.x) * 64 + lane;
    const bool active = v < num_voxels;
    const uint32_t byte_offset = uint32_t(v) * 4u, bytes = uint32_t(num_voxels) * 4u;
    const int nB = cs - CH;
    constexpr int SURE_B = CHB - 8;  // CH + CHB - 8 < cs (see spearman_split_kernel)
    // prep layout of launch_kendall_prep with n_pad = 2*CH; x-tie groups may straddle the chunks: monolithic kernel
    const bool x_ties = prep[2 * NPAD + 1] != 0;
    if (x_ties) {
        if (active) todo[1 + atomicAdd(&todo[0], 1u)] = uint32_t(v);
        return;
    }
    bool is_nan = false, all_equal = false;
    uint32_t tie_min = 0xFFFFFFFFu;
    uint32_t tie_or = 0u;  // OR of the differences of neighbours in the sorted chunks: 0 = every member of a chunk equal
    int32_t discordant = 0;
    {
        composite_t a[CH];
        load_chunk<CH, true, true>(a, members, prep, 0, cs, bytes, byte_offset);
        __builtin_amdgcn_sched_barrier(0);
        SortNet<CH>::sort(a);
        pin_array(a);  // the network ends here (crf_device.h)
        __builtin_amdgcn_sched_barrier(0);
        uint32_t prev = 0;
#pragma unroll
        for (int q = 0; q < CH; q++) {
            const uint32_t key = composite_key(a[q]);
            if (q > 0) tie_min = min(tie_min, key ^ prev);
            if (q > 0) tie_or |= key ^ prev;
            if (q == 0) is_nan |= key < 0x007FFFFFu;
            if (q == CH - 1) is_nan |= key > 0xFF800000u;
            prev = key;
            keysA[q * 64 + lane] = key;
        }
        discordant += chunk_inversions<CH>(a, CH, true);
    }
    __builtin_amdgcn_sched_barrier(0);
    uint32_t byte_offset_b = byte_offset;
    order_after(discordant, byte_offset_b);  // chunk B is not touched before chunk A is fully consumed
    order_after(tie_min, byte_offset_b);
    {
        composite_t b[CHB];
        load_chunk<CHB, EXACT, true, SURE_B>(b, members, prep, CH, cs, bytes, byte_offset_b);
        __builtin_amdgcn_sched_barrier(0);
        SortNet<CHB>::sort(b);
        pin_array(b);  // the network ends here (crf_device.h)
        __builtin_amdgcn_sched_barrier(0);
        uint32_t prev = 0;
        constexpr int G = 4;  // see spearman_split_kernel
#pragma unroll
        for (int p0 = 0; p0 < CHB; p0 += G) {
            const bool guarded = !EXACT && p0 >= SURE_B;  // compile time per batch
            if (!guarded || p0 + G <= nB) {
                uint32_t key[G], less[G];
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const int p = p0 + g;
                    key[g] = composite_key(b[p]);
                    if (p > 0) tie_min = min(tie_min, key[g] ^ prev);
                    if (p > 0) tie_or |= key[g] ^ prev;
                    if (p == 0) is_nan |= key[g] < 0x007FFFFFu;
                    if (EXACT ? p == CHB - 1 : guarded) is_nan |= (EXACT || p == nB - 1) && key[g] > 0xFF800000u;
                    prev = key[g];
                }
                lower_bound_col_batch<CH, G>(&keysA[lane], key, tie_min, less);
                // every a in A has a smaller x than b: the pair is discordant iff y_a > y_b
#pragma unroll
                for (int g = 0; g < G; g++) discordant += CH - int32_t(less[g]);
            } else {
#pragma unroll
                for (int p = p0; p < p0 + G; p++) {
                    if (p < nB) {
                        const uint32_t key = composite_key(b[p]);
                        if (p > 0) tie_min = min(tie_min, key ^ prev);
                        if (p > 0) tie_or |= key ^ prev;
                        if (p == 0) is_nan |= key < 0x007FFFFFu;
                        if (p == nB - 1) is_nan |= key > 0xFF800000u;
                        prev = key;
                        discordant += CH - int32_t(lower_bound_col<CH>(&keysA[lane], key, tie_min));
                    }
                }
            }
        }
        discordant += chunk_inversions<CHB, SURE_B>(b, nB, EXACT);
        // every member equal (a mask: whole regions of such voxels in real ensembles): n2 = n0, no discordant pair, and
        // with n1 = 0 (x ties went the other way above) tau = 0 / 0 -- no need to send the voxel to the exact kernel
        all_equal = tie_or == 0u && keysA[lane] == prev;
    }
    const int32_t n = cs;
    const int32_t n0 = (n * (n - 1)) / 2;
    const int32_t n1 = prep[2 * NPAD];  // 0 here
    const int32_t numerator = n0 - n1 - 2 * discordant;  // n2 = 0: no ties in y
    const float denominator = sqrtf(float(n0 - n1)) * sqrtf(float(n0));
    float res = float(numerator) / denominator;
    if (is_nan || all_equal) res = __uint_as_float(0x7FC00000u);
    if (active) {
        if (tie_min == 0u && !is_nan && !all_equal) {
            todo[1 + atomicAdd(&todo[0], 1u)] = uint32_t(v);
        } else {
            store_result_nt(out + v, res);
        }
    }
}

namespace {

// tuning switches (tools/tune_pearson.py): CRF_RANK_SPLIT64=1 uses the split-sort kernels for 33..64 members too
bool env_split64() {
    const char* v = getenv("CRF_RANK_SPLIT64");
    return v && *v == '1';
}

bool env_split64_default_on() {
    const char* v = getenv("CRF_RANK_SPLIT64");
    return !(v && *v == '0');
}

// CRF_RANK_EXACT=0 forces the guarded instantiations
bool env_exact() {
    const char* v = getenv("CRF_RANK_EXACT");
    return !(v && *v == '0');
}

constexpr unsigned kTodoBlocks = 2048;  // grid of the list-walking pass (grid-stride over the deferred voxels)

template <int N, int MIN_WAVES>
void launch_spearman_n(const float* const* d_members, const float* d_prep, float* d_out, size_t num_voxels, int cs,
                       hipStream_t s, const uint32_t* todo = nullptr) {
    const unsigned blocks = todo ? kTodoBlocks : unsigned((num_voxels + 63) / 64);
    const bool exact = cs == N && env_exact();
#define CRF_LAUNCH_SPEARMAN(EX, LIST)                                                                                  \
    hipLaunchKernelGGL((spearman_kernel<N, EX, MIN_WAVES, LIST>), dim3(blocks), dim3(64), 0, s, d_members, d_prep, d_out, \
                       num_voxels, cs, todo)
    if (todo) {
        if (exact) CRF_LAUNCH_SPEARMAN(true, true); else CRF_LAUNCH_SPEARMAN(false, true);
    } else {
        if (exact) CRF_LAUNCH_SPEARMAN(true, false); else CRF_LAUNCH_SPEARMAN(false, false);
    }
#undef CRF_LAUNCH_SPEARMAN
}

template <int N, int MIN_WAVES>
void launch_kendall_n(const float* const* d_members, const int* d_prep, float* d_out, size_t num_voxels, int cs,
                      hipStream_t s, const uint32_t* todo = nullptr) {
    const unsigned blocks = todo ? kTodoBlocks : unsigned((num_voxels + 63) / 64);
    const bool exact = cs == N && env_exact();
#define CRF_LAUNCH_KENDALL(EX, LIST)                                                                                  \
    hipLaunchKernelGGL((kendall_kernel<N, EX, MIN_WAVES, LIST>), dim3(blocks), dim3(64), 0, s, d_members, d_prep, d_out, \
                       num_voxels, cs, todo)
    if (todo) {
        if (exact) CRF_LAUNCH_KENDALL(true, true); else CRF_LAUNCH_KENDALL(false, true);
    } else {
        if (exact) CRF_LAUNCH_KENDALL(true, false); else CRF_LAUNCH_KENDALL(false, false);
    }
#undef CRF_LAUNCH_KENDALL
}

// split-sort launchers: chunk A = CH members, chunk B sorted by a CHB-network (CH < cs <= CH + CHB)
template <int CH, int CHB, int WAVES>
void launch_spearman_split(bool exact, const float* const* d_members, const float* d_prep, float* d_out,
                           size_t num_voxels, int cs, hipStream_t s, uint32_t* d_todo) {
    const unsigned blocks = unsigned((num_voxels + 63) / 64);
    if (exact && cs == CH + CHB)
        hipLaunchKernelGGL((spearman_split_kernel<CH, true, WAVES, CHB>), dim3(blocks), dim3(64), 0, s, d_members, d_prep,
                           d_out, num_voxels, cs, d_todo);
    else
        hipLaunchKernelGGL((spearman_split_kernel<CH, false, WAVES, CHB>), dim3(blocks), dim3(64), 0, s, d_members,
                           d_prep, d_out, num_voxels, cs, d_todo);
}
template <int CH, int CHB, int WAVES>
void launch_kendall_split(bool exact, const float* const* d_members, const int* d_prep, float* d_out, size_t num_voxels,
                          int cs, hipStream_t s, uint32_t* d_todo) {
    const unsigned blocks = unsigned((num_voxels + 63) / 64);
    if (exact && cs == CH + CHB)
        hipLaunchKernelGGL((kendall_split_kernel<CH, true, WAVES, CHB, 2 * CH>), dim3(blocks), dim3(64), 0, s, d_members,
                           d_prep, d_out, num_voxels, cs, d_todo);
    else
        hipLaunchKernelGGL((kendall_split_kernel<CH, false, WAVES, CHB, 2 * CH>), dim3(blocks), dim3(64), 0, s, d_members,
                           d_prep, d_out, num_voxels, cs, d_todo);
}

int pad_pow2(int cs) { return cs <= 8 ? 8 : cs <= 16 ? 16 : cs <= 32 ? 32 : cs <= 64 ? 64 : 128; }

int env_int(const char* name, int fallback) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : fallback;
}
bool env_flag(const char* name) { return env_int(name, 0) == 1; }

// waves/SIMD the 64-member kernels are compiled for (register cap 512/256/168); CRF_RANK_WAVES overrides for tuning.
int env_waves(int fallback) {
    const char* v = getenv("CRF_RANK_WAVES");
    return (v && *v) ? atoi(v) : fallback;
}

}  // namespace

// 129..256 members: spearman_pair_kernel; the caller runs the counting kernel over d_todo afterwards (voxels with ties).
bool launch_spearman_pair(const float* const* d_members, const float* d_prep, float* d_out, size_t num_voxels, int cs,
                          uint32_t* d_todo, hipStream_t s) {
    if (cs <= 128 || cs > 256 || !d_todo) return false;
    (void)hipMemsetAsync(d_todo, 0, sizeof(uint32_t), s);
    const unsigned blocks = unsigned((num_voxels + 63) / 64);
    const int n = ((cs + 1) / 2 + 7) / 8 * 8;  // chunk size: 2 n - 16 < cs <= 2 n
#define CRF_LAUNCH_PAIR(NN)                                                                                             \
    case NN:                                                                                                            \
        hipLaunchKernelGGL((spearman_pair_kernel<NN, (NN > 112 ? 1 : 2)>), dim3(blocks), dim3(64), 0, s, d_members,     \
                           d_prep, d_out,                                                                               \
                           num_voxels, cs, d_todo);                                                                     \
        break
    switch (n) {
        CRF_LAUNCH_PAIR(72);
        CRF_LAUNCH_PAIR(80);
        CRF_LAUNCH_PAIR(88);
        CRF_LAUNCH_PAIR(96);
        CRF_LAUNCH_PAIR(104);
        CRF_LAUNCH_PAIR(112);
        CRF_LAUNCH_PAIR(120);
        CRF_LAUNCH_PAIR(128);
        default: return false;
    }
#undef CRF_LAUNCH_PAIR
    return true;
}

// 129..256 members: kendall_pair_kernel (prep: kendall_prep_kernel's tables with stride cs); the caller runs the counting
// kernel over d_todo afterwards.  One wave per SIMD for the register budget: at two the kernel needs > 256 registers and
// spills 0.3 KB to scratch, at one the overflow goes to AGPRs (LDS allows 3-5 waves per CU anyway).
bool launch_kendall_pair(const float* const* d_members, const int* d_prep, float* d_out, size_t num_voxels, int cs,
                         uint32_t* d_todo, hipStream_t s) {
    if (cs <= 128 || cs > 256 || !d_todo) return false;
    (void)hipMemsetAsync(d_todo, 0, sizeof(uint32_t), s);
    const unsigned blocks = unsigned((num_voxels + 63) / 64);
    const int n = ((cs + 1) / 2 + 7) / 8 * 8;
#define CRF_LAUNCH_PAIR(NN)                                                                                             \
    case NN:                                                                                                            \
        hipLaunchKernelGGL((kendall_pair_kernel<NN, 1>), dim3(blocks), dim3(64), 0, s, d_members,                       \
                           d_prep, d_out, num_voxels, cs, d_todo);                                                      \
        break
    switch (n) {
        CRF_LAUNCH_PAIR(72);
        CRF_LAUNCH_PAIR(80);
        CRF_LAUNCH_PAIR(88);
        CRF_LAUNCH_PAIR(96);
        CRF_LAUNCH_PAIR(104);
        CRF_LAUNCH_PAIR(112);
        CRF_LAUNCH_PAIR(120);
        CRF_LAUNCH_PAIR(128);
        default: return false;
    }
#undef CRF_LAUNCH_PAIR
    return true;
}

void launch_spearman_prep(const RefSource& ref, const float* const* d_members, int cs, float* d_prep, hipStream_t s) {
    hipLaunchKernelGGL(spearman_prep_kernel, dim3(1), dim3(256), size_t(2 * cs) * sizeof(float), s, ref, d_members, cs,
                       d_prep);
}

void launch_kendall_prep(const RefSource& ref, const float* const* d_members, int cs, int n_pad, int* d_prep,
                         hipStream_t s) {
    hipLaunchKernelGGL(kendall_prep_kernel, dim3(1), dim3(256), size_t(cs) * sizeof(float), s, ref, d_members, cs, n_pad,
                       d_prep);
}

hipError_t launch_spearman(const float* const* d_members, int cs, size_t num_voxels, const RefSource& ref, float* d_prep,
                           uint32_t* d_todo, float* d_out, hipStream_t s, hipEvent_t ev_begin, hipEvent_t ev_end,
                           LaunchInfo* info) {
    bool split = false, u32 = false;
    if (cs == 1) {
        if (!ref.run()) return hipSuccess;
        if (ev_begin) (void)hipEventRecord(ev_begin, s);
        hipError_t e = launch_fill(d_out, num_voxels, 1.0f, s);
        if (ev_end) (void)hipEventRecord(ev_end, s);
        if (info) info->kernel_name = "fill_kernel";
        return e;
    }
    if (ref.prepare()) launch_spearman_prep(ref, d_members, cs, d_prep, s);
    if (!ref.run()) return hipGetLastError();
    if (ev_begin) (void)hipEventRecord(ev_begin, s);
    switch (pad_pow2(cs)) {
        case 8: launch_spearman_n<8, 4>(d_members, d_prep, d_out, num_voxels, cs, s); break;
        case 16: launch_spearman_n<16, 4>(d_members, d_prep, d_out, num_voxels, cs, s); break;
        case 32:
            // measured at 256^3: cs = 32: monolithic unguarded 0.62-0.66 ms, split 0.65 ms; cs = 24: monolithic guarded
            // 1.2-2.0 ms, split 0.56 ms; cs = 20: 1.2-2.1 ms vs 0.49 ms
            if (d_todo && cs > 16 && (cs < 32 || env_flag("CRF_RANK_SPLIT32"))) {
                (void)hipMemsetAsync(d_todo, 0, sizeof(uint32_t), s);
                const unsigned blocks = unsigned((num_voxels + 63) / 64);
                if (cs <= 24)
                    hipLaunchKernelGGL((spearman_split_kernel<16, false, 4, 8>), dim3(blocks), dim3(64), 0, s, d_members,
                                       d_prep, d_out, num_voxels, cs, d_todo);
                else
                    hipLaunchKernelGGL((spearman_split_kernel<16, false, 4, 16>), dim3(blocks), dim3(64), 0, s, d_members,
                                       d_prep, d_out, num_voxels, cs, d_todo);
                launch_spearman_n<32, 4>(d_members, d_prep, d_out, num_voxels, cs, s, d_todo);
                split = true;
                break;
            }
            switch (env_int("CRF_RANK_WAVES32", 4)) {
                case 2: launch_spearman_n<32, 2>(d_members, d_prep, d_out, num_voxels, cs, s); break;
                case 3: launch_spearman_n<32, 3>(d_members, d_prep, d_out, num_voxels, cs, s); break;
                default: launch_spearman_n<32, 4>(d_members, d_prep, d_out, num_voxels, cs, s); break;
            }
            break;
        case 64:
            // 33..64 members: the u32 network too (256^3: 40 members 0.77 -> 0.70 ms, 48: 0.95 -> 0.85, 56: 1.16 -> 1.02,
            // 64: 1.22 -> 1.17, 33: unchanged); CRF_RANK_U32=0 keeps the 64-bit-composite kernels
            if (d_todo && cs > 32 && env_int("CRF_RANK_U32", 1) != 0) {
                (void)hipMemsetAsync(d_todo, 0, sizeof(uint32_t), s);
                const unsigned blocks = unsigned((num_voxels + 63) / 64);
#define CRF_LAUNCH_U32_SMALL(NN)                                                                                          \
    if (cs == NN)                                                                                                        \
        hipLaunchKernelGGL((spearman_u32_kernel<NN, 2, true>), dim3(blocks), dim3(64), 0, s, d_members, d_prep, d_out,   \
                           num_voxels, cs, d_todo);                                                                     \
    else                                                                                                                 \
        hipLaunchKernelGGL((spearman_u32_kernel<NN, 2, false>), dim3(blocks), dim3(64), 0, s, d_members, d_prep, d_out,  \
                           num_voxels, cs, d_todo)
                if (cs <= 40) { CRF_LAUNCH_U32_SMALL(40); }
                else if (cs <= 48) { CRF_LAUNCH_U32_SMALL(48); }
                else if (cs <= 56) { CRF_LAUNCH_U32_SMALL(56); }
                else { CRF_LAUNCH_U32_SMALL(64); }
#undef CRF_LAUNCH_U32_SMALL
                launch_spearman_n<64, 2>(d_members, d_prep, d_out, num_voxels, cs, s, d_todo);  // voxels with ties
                u32 = true;
                break;
            }
            // measured at 256^3 (profiles/tuning_r01.md): cs = 64: monolithic unguarded 1.75 ms vs split 1.87-1.96 ms;
            // cs = 48: monolithic guarded 5.26 ms vs split 1.65 ms; cs = 40: 5.45 ms vs 1.43 ms
            if (d_todo && cs > 32 && (cs < 64 || env_split64())) {
                (void)hipMemsetAsync(d_todo, 0, sizeof(uint32_t), s);
                const bool exact = env_exact() && getenv("CRF_RANK_EXACT");
                if (cs <= 40)  // chunk B sorted by the smallest network of a multiple of 8 elements that holds it
                    launch_spearman_split<32, 8, 4>(exact, d_members, d_prep, d_out, num_voxels, cs, s, d_todo);
                else if (cs <= 48)
                    launch_spearman_split<32, 16, 4>(exact, d_members, d_prep, d_out, num_voxels, cs, s, d_todo);
                else if (cs <= 56)
                    launch_spearman_split<32, 24, 4>(exact, d_members, d_prep, d_out, num_voxels, cs, s, d_todo);
                else
                    launch_spearman_split<32, 32, 4>(exact, d_members, d_prep, d_out, num_voxels, cs, s, d_todo);
                launch_spearman_n<64, 2>(d_members, d_prep, d_out, num_voxels, cs, s, d_todo);
                split = true;
                break;
            }
            switch (env_waves(2)) {
                case 1: launch_spearman_n<64, 1>(d_members, d_prep, d_out, num_voxels, cs, s); break;
                case 3: launch_spearman_n<64, 3>(d_members, d_prep, d_out, num_voxels, cs, s); break;
                default: launch_spearman_n<64, 2>(d_members, d_prep, d_out, num_voxels, cs, s); break;
            }
            break;
        default:
            // measured at 256^3 x 128 (profiles/tuning_r01.md): split 5.1 ms (unguarded) / 7.4 ms (guarded) vs
            // monolithic 7.5 ms / 19 ms
            if (d_todo && env_waves(2) != 0) {
                (void)hipMemsetAsync(d_todo, 0, sizeof(uint32_t), s);
                const bool wide_exact = env_exact() && getenv("CRF_RANK_EXACT");  // only on request
                const unsigned blocks = unsigned((num_voxels + 63) / 64);
#define CRF_LAUNCH_U32(NN)                                                                                               \
    if (cs == NN)                                                                                                        \
        hipLaunchKernelGGL((spearman_u32_kernel<NN, 2, true>), dim3(blocks), dim3(64), 0, s, d_members, d_prep, d_out,   \
                           num_voxels, cs, d_todo);                                                                     \
    else                                                                                                                 \
        hipLaunchKernelGGL((spearman_u32_kernel<NN, 2, false>), dim3(blocks), dim3(64), 0, s, d_members, d_prep, d_out,  \
                           num_voxels, cs, d_todo)
                // One network over 32-bit composites (spearman_u32_kernel, padded to a multiple of 8): 256^3, split-sort ->
                // u32 network: 72 members 1.73 -> 1.35 ms, 80: 1.89 -> 1.52, 96: 2.60 -> 2.02, 100: 2.77 -> 2.24, 112: 3.08
                // -> 2.39, 128: 3.50 -> 2.72; 512^3 x 128 (BASELINE configs[3]) 27.6 -> 21.5 ms, bit-identical fields.
                // CRF_RANK_U32=0 keeps the split-sort kernels.
                const int u32_env = env_int("CRF_RANK_U32", -1);
                if (cs > 64 && u32_env != 0) {
                    if (cs <= 72) { CRF_LAUNCH_U32(72); }
                    else if (cs <= 80) { CRF_LAUNCH_U32(80); }
                    else if (cs <= 88) { CRF_LAUNCH_U32(88); }
                    else if (cs <= 96) { CRF_LAUNCH_U32(96); }
                    else if (cs <= 104) { CRF_LAUNCH_U32(104); }
                    else if (cs <= 112) { CRF_LAUNCH_U32(112); }
                    else if (cs <= 120) { CRF_LAUNCH_U32(120); }
                    else { CRF_LAUNCH_U32(128); }
                    u32 = true;
                } else
#undef CRF_LAUNCH_U32
                if (cs <= 72)
                    launch_spearman_split<64, 8, 2>(env_exact(), d_members, d_prep, d_out, num_voxels, cs, s, d_todo);
                else if (cs <= 80)
                    launch_spearman_split<64, 16, 2>(env_exact(), d_members, d_prep, d_out, num_voxels, cs, s, d_todo);
                else if (cs <= 88)  // merge-exchange networks exist for any size
                    launch_spearman_split<64, 24, 2>(env_exact(), d_members, d_prep, d_out, num_voxels, cs, s, d_todo);
                else if (cs <= 96)
                    launch_spearman_split<64, 32, 2>(env_exact(), d_members, d_prep, d_out, num_voxels, cs, s, d_todo);
                else if (cs <= 104)
                    launch_spearman_split<64, 40, 2>(env_exact(), d_members, d_prep, d_out, num_voxels, cs, s, d_todo);
                else if (cs <= 112)
                    launch_spearman_split<64, 48, 2>(env_exact(), d_members, d_prep, d_out, num_voxels, cs, s, d_todo);
                else if (cs <= 120)  // 56 / 64: the unguarded instantiations spill (120: 3.96 vs 3.78 ms, 128: 4.83 vs 4.10 ms)
                    launch_spearman_split<64, 56, 2>(wide_exact, d_members, d_prep, d_out, num_voxels, cs, s, d_todo);
                else
                    launch_spearman_split<64, 64, 2>(wide_exact, d_members, d_prep, d_out, num_voxels, cs, s, d_todo);
                launch_spearman_n<128, 1>(d_members, d_prep, d_out, num_voxels, cs, s, d_todo);  // voxels with ties
                split = true;
            } else {
                launch_spearman_n<128, 1>(d_members, d_prep, d_out, num_voxels, cs, s);
            }
            break;
    }
    if (ev_end) (void)hipEventRecord(ev_end, s);
    if (info) info->kernel_name = u32 ? "spearman_u32_kernel" : split ? "spearman_split_kernel" : "spearman_kernel";
    return hipGetLastError();
}

hipError_t launch_kendall(const float* const* d_members, int cs, size_t num_voxels, const RefSource& ref, float* d_prep,
                          uint32_t* d_todo, float* d_out, hipStream_t s, hipEvent_t ev_begin, hipEvent_t ev_end,
                          LaunchInfo* info) {
    bool split = false;
    if (cs == 1) {
        if (!ref.run()) return hipSuccess;
        if (ev_begin) (void)hipEventRecord(ev_begin, s);
        hipError_t e = launch_fill(d_out, num_voxels, 1.0f, s);
        if (ev_end) (void)hipEventRecord(ev_end, s);
        if (info) info->kernel_name = "fill_kernel";
        return e;
    }
    const int n_pad = pad_pow2(cs);
    int* prep = reinterpret_cast<int*>(d_prep);
    if (ref.prepare()) launch_kendall_prep(ref, d_members, cs, n_pad, prep, s);
    if (!ref.run()) return hipGetLastError();
    if (ev_begin) (void)hipEventRecord(ev_begin, s);
    switch (n_pad) {
        case 8: launch_kendall_n<8, 4>(d_members, prep, d_out, num_voxels, cs, s); break;
        case 16: launch_kendall_n<16, 4>(d_members, prep, d_out, num_voxels, cs, s); break;
        case 32:
            // measured at 256^3: cs = 32: split 0.55 ms vs monolithic 0.63-1.23 ms; cs = 24: 0.46 vs 1.0-2.4 ms;
            // cs = 20: 0.41 vs 1.0-2.6 ms.  CRF_RANK_SPLIT32=0 selects the monolithic kernel.
            if (d_todo && cs > 16 && env_int("CRF_RANK_SPLIT32", 1) != 0) {
                (void)hipMemsetAsync(d_todo, 0, sizeof(uint32_t), s);
                const unsigned blocks = unsigned((num_voxels + 63) / 64);
                if (cs <= 24)
                    hipLaunchKernelGGL((kendall_split_kernel<16, false, 4, 8, 32>), dim3(blocks), dim3(64), 0, s, d_members,
                                       prep, d_out, num_voxels, cs, d_todo);
                else
                    hipLaunchKernelGGL((kendall_split_kernel<16, false, 4, 16, 32>), dim3(blocks), dim3(64), 0, s,
                                       d_members, prep, d_out, num_voxels, cs, d_todo);
                launch_kendall_n<32, 4>(d_members, prep, d_out, num_voxels, cs, s, d_todo);
                split = true;
                break;
            }
            switch (env_int("CRF_RANK_WAVES32", 4)) {
                case 2: launch_kendall_n<32, 2>(d_members, prep, d_out, num_voxels, cs, s); break;
                case 3: launch_kendall_n<32, 3>(d_members, prep, d_out, num_voxels, cs, s); break;
                default: launch_kendall_n<32, 4>(d_members, prep, d_out, num_voxels, cs, s); break;
            }
            break;
        case 64:
            // measured at 256^3: cs = 64: split (guarded) 1.49 ms vs monolithic 2.64 ms; cs = 48: 1.24 vs 4.22 ms;
            // cs = 40: 1.05 vs 4.12 ms.  CRF_RANK_SPLIT64=0 selects the monolithic kernel.
            if (d_todo && cs > 32 && env_split64_default_on()) {
                (void)hipMemsetAsync(d_todo, 0, sizeof(uint32_t), s);
                const bool exact = env_exact() && getenv("CRF_RANK_EXACT");
                if (cs <= 40)
                    launch_kendall_split<32, 8, 4>(exact, d_members, prep, d_out, num_voxels, cs, s, d_todo);
                else if (cs <= 48)
                    launch_kendall_split<32, 16, 4>(exact, d_members, prep, d_out, num_voxels, cs, s, d_todo);
                else if (cs <= 56)
                    launch_kendall_split<32, 24, 4>(exact, d_members, prep, d_out, num_voxels, cs, s, d_todo);
                else
                    launch_kendall_split<32, 32, 4>(exact, d_members, prep, d_out, num_voxels, cs, s, d_todo);
                launch_kendall_n<64, 1>(d_members, prep, d_out, num_voxels, cs, s, d_todo);
                split = true;
                break;
            }
            switch (env_waves(1)) {
                case 3: launch_kendall_n<64, 3>(d_members, prep, d_out, num_voxels, cs, s); break;
                case 2: launch_kendall_n<64, 2>(d_members, prep, d_out, num_voxels, cs, s); break;
                default: launch_kendall_n<64, 1>(d_members, prep, d_out, num_voxels, cs, s); break;
            }
            break;
        default:
            if (d_todo && env_waves(2) != 0) {
                (void)hipMemsetAsync(d_todo, 0, sizeof(uint32_t), s);
                // the guarded instantiation compiles to 226 VGPRs without scratch and is the fastest for every cs
                // (4.8 ms at 256^3 x 128 vs 12.8 ms unguarded, 13.9-27.7 ms monolithic)
                const bool exact = env_exact() && getenv("CRF_RANK_EXACT");
                if (cs <= 72)
                    launch_kendall_split<64, 8, 2>(exact, d_members, prep, d_out, num_voxels, cs, s, d_todo);
                else if (cs <= 80)
                    launch_kendall_split<64, 16, 2>(exact, d_members, prep, d_out, num_voxels, cs, s, d_todo);
                else if (cs <= 88)
                    launch_kendall_split<64, 24, 2>(exact, d_members, prep, d_out, num_voxels, cs, s, d_todo);
                else if (cs <= 96)
                    launch_kendall_split<64, 32, 2>(exact, d_members, prep, d_out, num_voxels, cs, s, d_todo);
                else if (cs <= 104)
                    launch_kendall_split<64, 40, 2>(exact, d_members, prep, d_out, num_voxels, cs, s, d_todo);
                else if (cs <= 112)
                    launch_kendall_split<64, 48, 2>(exact, d_members, prep, d_out, num_voxels, cs, s, d_todo);
                else if (cs <= 120)
                    launch_kendall_split<64, 56, 2>(exact, d_members, prep, d_out, num_voxels, cs, s, d_todo);
                else
                    launch_kendall_split<64, 64, 2>(exact, d_members, prep, d_out, num_voxels, cs, s, d_todo);
                launch_kendall_n<128, 1>(d_members, prep, d_out, num_voxels, cs, s, d_todo);  // voxels with ties
                split = true;
            } else {
                launch_kendall_n<128, 1>(d_members, prep, d_out, num_voxels, cs, s);
            }
            break;
    }
    if (ev_end) (void)hipEventRecord(ev_end, s);
    if (info) info->kernel_name = split ? "kendall_split_kernel" : "kendall_kernel";
    return hipGetLastError();
}

}  // namespace crf
