// rdst_kernels.hip — gfx950 (MI355X, CDNA4) radix-sort hot path + its C ABI (include/rdst_hip.h).
//
// What is here, and which reference function each piece is the device twin of
// (paths relative to the reference tree):
//
//   K1 hist_kernel        one coalesced read of the key slice -> 256-bin histograms of EVERY
//                         level, split by position range and by group of the previous digit
//                         (what the look-back chains of K3 start from).  get_counts_with_ends
//                         (src/sort_utils.rs:109-180), get_tile_counts (:193-244)
//   K2 scan_kernel        256-bin exclusive scan per level, the level-skip plan, chain tables.
//                         get_prefix_sums (:10-20), level skipping of lsb_sort_adapter
//                         (src/sorts/lsb_sort.rs:62-83)
//   K3 onesweep_kernel    one stable counting-sort pass: wave-ballot ranking, tile prefix by
//                         chained scan with decoupled look-back, one chain per source segment
//                         and XCD.                               out_of_place_sort
//                         (src/sorts/out_of_place_sort.rs:52-108), mt_lsb_sort (:40-133)
//   K4 key map            fused into K1/K3 digit extraction.     src/radix_key_impl.rs:3-185
//   K6 level_counts_kernel single-level histogram + already_sorted flag (parity hook)
//
// Data layout in HBM: keys are a dense array of K (unsigned 1-, 2-, 4-, 8- or 16-byte bit patterns;
// signed and float keys stay in their raw encoding in memory, the order-preserving map is applied
// in registers for digit extraction only).  `keys` and `tmp` ping-pong per executed pass.  The workspace
// holds, per sort: an error word, per-level and per-chain tile tickets, the plan, K1's count tables
// u64[L][8][256] (per position range, and per group of the previous digit), look-back status words
// (u32 if n < 2^30, else u64) [L][rows][256] in two copies, digit totals and bucket starts
// u64[L][256], chain starts u64[L][8][256] and the chain tables.
//
// Written for wave64 / 256 CUs in 8 XCDs only; no other target is supported.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <stddef.h>
#include <mutex>
#include <type_traits>
#include <string>
#include <vector>
#include <map>
#include <utility>

#include "rdst_hip.h"

namespace {

constexpr int RADIX = 256;
constexpr int MAX_LEVELS = 16;

// look-back status word: top two bits state, the rest value (u32 for n < 2^30, u64 above)
constexpr uint32_t ST_EMPTY = 0, ST_AGG = 1, ST_INCL = 2;
template <typename S> struct StatusWord {
    static constexpr int SHIFT = sizeof(S) * 8 - 2;
    static constexpr S MASK = (S(1) << SHIFT) - 1;
};
// bounded spins: s_sleep(2) is ~128 clocks; 1<<22 polls is seconds, never reached in a healthy run
constexpr uint32_t SPIN_LIMIT = 1u << 22;

#ifndef RDST_PRIO_LOAD
#define RDST_PRIO_LOAD 2
#endif
#ifndef RDST_PRIO_LB
#define RDST_PRIO_LB 3
#endif
#ifndef RDST_PRIO_SCAN
#define RDST_PRIO_SCAN 3  // digit sums, tile scan, look-back: the four waves the block's other eight wait for
#endif
#ifndef RDST_PRIO_SCATTER
#define RDST_PRIO_SCATTER 2
#endif
#ifndef RDST_LB_WINDOW
#define RDST_LB_WINDOW 8  // predecessor status words fetched per look-back round trip
#endif
constexpr uint32_t RDST_FAST_RANK = 1u << 16;  // bit of the pass kernel's flag word (its low bits: ablation switches of tools/ builds)
constexpr uint32_t RDST_FAST_RANK_SELFTEST = 1u << 17;  // treat every round of the fast ranking as failed: exercises its fallback
constexpr uint32_t ERR_LOOKBACK_TIMEOUT = 1;
constexpr uint32_t ERR_SCATTER_RANGE = 2;  // a computed destination fell outside [0, n): never stored
constexpr uint32_t ERR_LOCAL_OVERFLOW = 4;  // a bucket larger than the K4 kernel it reached can take (the routes' own tests rule it out): never sorted

// Three routes through the kernels, chosen ON THE DEVICE from a sample and the counts of the key multiset — the device
// form of Tuner::pick_algorithm(params, counts) (src/tuner.rs:33-35, src/sorter.rs:67-76).  Tried in this order; the
// host launches every kernel of every route, each looks at the plan first:
//   ROUTE_ATOMIC  (4- and 8-byte keys, atomic_min_len() <= n < 2^30) the hybrid route without its counting read.  An MSD pass needs no
//                 stable order and no exact global offsets up front, only ROOM: pass A scatters by the top byte into
//                 256 x 8 over-provisioned areas (XCD slice x digit) of the workspace, a tile claiming its space per digit
//                 with ONE returning global atomic where K3 walks back over its predecessors; pass B scatters every area by
//                 the second byte into 65 536 slots (4-byte keys: low halves only); K4 sorts each slot's bucket to its
//                 exact place (exclusive scan of the 65 536 claim counters).  u32: 8 + 6 + 6 = 20 bytes per key, u64: 48.
//                 Uniform keys never overflow an area (capacity = mean + max(1 / 8, 8 sigma)); a claim that does not fit
//                 gives the route up.  Keys that share their top bits (the sample sees it, pass A checks it) are
//                 bucketed by the 16 bits below those (Plan::win_shift).
//   ROUTE_HYBRID  the shape of rdst's own 10^9-key route (two MSD levels, then Lsb on
//                 ~15 k-key chunks, src/sorts/lsb_sort.rs:39-127) with exact counts: K1h counts the top 16 bits, two K3
//                 passes order the slice by them (levels L-2, L-1), and K4 sorts every one of the
//                 65 536 buckets by the remaining levels inside LDS — one read and one coalesced write
//                 instead of L-2 scatter passes (u32: 24 instead of 36 bytes per key; u64: 56 instead
//                 of 136).  8-byte keys: when every bucket fits K4's tile.  4-byte keys: buckets of any size — up to
//                 4 096 of them may hold 65 536 keys and more (the giant kernels) — unless more than a third of the keys
//                 sit in buckets of one to four tiles.
//   ROUTE_LSD     K1, K2, one K3 pass per level (k*(2L+1) bytes per key): everything else.
constexpr uint32_t ROUTE_LSD = 0, ROUTE_HYBRID = 1, ROUTE_ATOMIC = 2;
constexpr int MSD_SLICES = 8;  // areas per top digit in pass A: blocks b and b + 8 share an XCD, so a digit's 8 frontiers stay with one L2 each
constexpr int H16_BINS = 65536;

struct Plan {
    uint32_t skip[MAX_LEVELS];        // pass would move nothing (one bin holds every key)
    uint32_t src_is_tmp[MAX_LEVELS];  // which buffer the pass reads
    uint32_t chain_mode[MAX_LEVELS];  // how the pass's source splits into look-back chains (CHAIN_*)
    uint32_t result_in_tmp;           // where the data sits after the last executed pass
    uint32_t executed;                // number of passes executed
    uint32_t first_level;             // lowest executed level: below it nothing has ordered the keys (order test of K3's fast ranking)
    uint32_t route;                   // ROUTE_ATOMIC (msd_finish_kernel), else ROUTE_HYBRID or ROUTE_LSD (route_kernel)
    uint32_t local_sort;              // hybrid route and the slice is not already sorted: K4 runs
    uint32_t gross_skew;              // a sample of the keys already rules the hybrid route out (presample_kernel): K1h returns at once
    uint32_t top_skew;                // the sample's top bytes are far from uniform: the atomic route's areas would overflow, its passes return at once
    uint32_t giants, giant_count_items, giant_expand_items;  // hybrid route, 4-byte keys: buckets of 65 536 keys and more and the work items of their kernels (route_kernel)
    // ROUTE_ATOMIC's bucket window: the sample's keys share their top win_shift bits (value win_top) — keys below 2^30, one
    // rank's share of a sharded sort — so the 16 bits that make the 65 536 buckets start win_shift bits lower (pass A checks
    // every key against win_top; one that differs gives the route up)
    uint32_t win_shift, win_top;
    // the sample flagged the keys (top_skew / gross_skew), so K1h counted them BEFORE the MSD passes and route_kernel has
    // already decided: HYBRID — the two MSD passes then run in their exact form (claims on cursors that start at the exact
    // offsets K1h's counts give: no room to run out of, no look-back, any distribution) in place of the two K3 passes — or LSD
    uint32_t pre;
    uint32_t pre_skip_a;  // ... and every key holds the same top byte (4-byte keys): the exact pass A would be a copy, pass B reads the slice itself
    uint32_t low_dups;                // (4-byte keys) the sample's low halves repeat: K4's first kernel (4-bit counters) would refuse most buckets, it hands them all on
    uint32_t sorted_known;            // K1h swept the whole slice and met no inversion: K1 need not read it again (K2 turns every pass off)
    // the sample says the counts would send the sort down the LSD route anyway (most keys in buckets of one to four tiles, more
    // giants than tables; 8-byte keys: a bucket over the tile): neither the atomic route nor K1h is tried — the slice is not
    // read twice for counting, which is what made such inputs slower than the LSD-only setting
    uint32_t predict_lsd;
};

// top 16 bits (of the mapped key) of the keys of bucket b: b itself, or — atomic route with a lowered window — the shared top
// bits followed by the bucket's upper ones (its lowest win_shift bits lie below bit W - 16: they are part of what K4 sorts)
__device__ __forceinline__ uint32_t bucket_prefix16(const Plan* plan, uint32_t bucket) {
    const uint32_t sh = plan->route == 2u /* ROUTE_ATOMIC */ ? plan->win_shift : 0u;
    return (((plan->win_top << 16) | bucket) >> sh) & 0xFFFFu;
}

// Look-back chains.  One chain over all tiles makes every tile walk back over ~(status latency /
// tile start interval) predecessor rows, and those re-reads are fabric traffic on the scale of
// the keys themselves.  A pass therefore splits its source into CHAINS contiguous
// segments whose digit counts are known BEFORE the pass, each with its own chain: the start rate
// per chain, and with it the walk depth, drops CHAINS-fold.  Segment counts that one read of
// the unsorted keys can give (K1):
//   CHAIN_POS   first executed pass: segment = position range of the input (K1's pieces)
//   CHAIN_PAIR  pass p right after pass p-1: segment = the buckets of a group of p-1 digits, its
//               counts = the joint histogram (digit_{p-1} group, digit_p) — a property of the key
//               multiset, unlike range counts of a later pass's (permuted) source
//   CHAIN_ONE   anything else (a skipped level in between): one chain
constexpr int CHAINS = 8;
constexpr uint32_t CHAIN_ONE = 0, CHAIN_POS = 1, CHAIN_PAIR = 2;
constexpr uint32_t TILE_ALIGN = 32;  // keys; every tile but a chain's first starts on a multiple
constexpr int TICKET_STRIDE = 32;    // words: every chain's ticket counter, and the mask word, on a 128-byte line of its own
constexpr int TICKET_ROW = (CHAINS + 1) * TICKET_STRIDE;  // words per level

struct LevelChains {  // written by K2 for every executed level, read by K3's ticket holder
    uint64_t seg_lo[CHAINS + 1];  // chain c covers source indices [seg_lo[c], seg_lo[c+1])
    uint32_t row0[CHAINS];        // first status row of chain c
    uint32_t ntiles[CHAINS];      // tiles of chain c
    uint32_t shared[CHAINS];      // chain much longer than the others: blocks of every XCD will end up on it
    uint32_t total_tiles;
    uint32_t pad[3];
};

typedef unsigned __int128 u128;  // u128 / i128 keys (src/radix_key_impl.rs:39-46, :123-130)

struct KeyMap {  // order-preserving map as two xor masks (src/radix_key_impl.rs)
    u128 neg;  // xor applied when the sign bit is set
    u128 pos;  // xor applied when it is clear
};

template <typename K>
__device__ __forceinline__ K map_key(K k, K neg, K pos) {
    constexpr int W = sizeof(K) * 8;
    return (K)(k ^ ((K)(k >> (W - 1)) ? neg : pos));
}
template <typename K>
__device__ __forceinline__ K unmap_key(K m, K neg, K pos) {
    constexpr int W = sizeof(K) * 8;
    return (K)(m ^ ((K)(m >> (W - 1)) ? pos : neg));
}
template <typename K>
__device__ __forceinline__ uint32_t digit_of(K mapped, int shift) {
    return (uint32_t)(mapped >> shift) & 0xFFu;
}

template <typename S>
__device__ __forceinline__ S ld_relaxed(const S* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename S>
__device__ __forceinline__ void st_relaxed(S* p, S v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Four status words in one write-through store.  A scalar (4-byte) sc1 store is a fabric write of its
// own and costs about six times a 16-byte one per byte (MI355X_MICROARCH.md); the protocol needs every
// WORD to arrive whole, which a naturally aligned dword inside a 16-byte store does.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_far_x4(uint32_t* p, u32x4 v) {
    // s_nop: a VMEM store of more than 64 bits must not be followed at once by a VALU write to its data
    // registers (a manually inserted wait state of the ISA; the compiler adds it for its own stores but
    // cannot see into the asm — without it the key-value kernel overwrote the first word in flight)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
#ifndef RDST_FAR_X4
#define RDST_FAR_X4 1
#endif
template <typename S>
__device__ __forceinline__ void st_near(S* p, S v) {  // stays (dirty) in this XCD's L2
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// value held by lane - 1 (lane 0: unspecified), as a DPP wave shift: no LDS traffic
template <typename K>
__device__ __forceinline__ K lane_below(K x) {
    constexpr int WAVE_SHR1 = 0x138;
    if constexpr (sizeof(K) <= 4) {
        return (K)__builtin_amdgcn_mov_dpp((int)x, WAVE_SHR1, 0xf, 0xf, false);
    } else {
        K r = 0;
#pragma unroll
        for (int w = 0; w < (int)sizeof(K) / 4; ++w)
            r |= (K)(uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(x >> (32 * w)), WAVE_SHR1, 0xf, 0xf, false) << (32 * w);
        return r;
    }
}

// value held by lane 63 (wave-uniform result)
template <typename K>
__device__ __forceinline__ K lane63_of(K x) {
    if constexpr (sizeof(K) <= 4) {
        return (K)(uint32_t)__builtin_amdgcn_readlane((int)x, 63);
    } else {
        K r = 0;
#pragma unroll
        for (int w = 0; w < (int)sizeof(K) / 4; ++w) r |= (K)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> (32 * w)), 63) << (32 * w);
        return r;
    }
}

// lanes of this wave holding the same 8-bit digit (all 64 lanes must be active)
__device__ __forceinline__ uint64_t match_any8(uint32_t d) {
    uint64_t m = ~0ull;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}
__device__ __forceinline__ uint32_t lanes_below(uint64_t mask) {  // popcount(mask & lanes < me)
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// ------------------------------------------------------------------------------------------
// K1: every level's histogram from one read.  One 1024-thread block per CU (grid = #CUs), each
// sweeping a contiguous piece with 16-byte loads, four in flight per lane.
//
// LDS atomics on a plain [level][digit] table are bank-bound on uniform keys: the 32 lanes of a
// half-wave pick 32 random banks, ~3.5 of them collide on the busiest bank, and the first version
// of this kernel sat at 72 % conflict cycles (profiles/r01a_pmc_summary.json).  So the table is
// replicated COPIES times along the bank axis — word (level*256 + digit)*COPIES + (lane % COPIES)
// — which puts lane l of a half-wave on bank l: no two lanes of one LDS cycle share a bank
// (COPIES = 32; with 16 copies for 8-byte keys two lanes can, at most 2-way).  128 KiB of LDS.
// The copies are folded (rotated reads, conflict-free) and added to the global table at the end.
// VEC = keys per 16-byte load (1 = unaligned fallback).
// ------------------------------------------------------------------------------------------
constexpr int HIST_THREADS = 1024;

// K1's grid is cut into CHAINS position ranges of whole pieces: block b counts for range b*CHAINS/grid
__host__ __device__ inline uint32_t hist_range_of(uint32_t block, uint32_t grid) { return (uint32_t)((uint64_t)block * CHAINS / grid); }
__host__ __device__ inline uint32_t hist_first_block(uint32_t range, uint32_t grid) {  // smallest b with hist_range_of(b) >= range
    return (uint32_t)(((uint64_t)range * grid + CHAINS - 1) / CHAINS);
}
__host__ __device__ inline uint64_t hist_piece(uint64_t n, uint32_t grid, uint64_t gran) {
    uint64_t piece = (n + grid - 1) / grid;
    return (piece + gran - 1) / gran * gran;
}

// LDS plan of K1.  PAIR == false: one [256] table per level, COPIES bank columns each.
// PAIR == true (the chain split needs it): level 0 as before, and for every level l >= 1 a joint
// table over (digit_l, group of digit_{l-1}) — entry digit_l * CHAINS + (digit_{l-1} >> 5), one
// 11-bit field of the key — with PCOPIES bank columns; its sum over the groups is level l's plain
// histogram, so a key still costs one LDS atomic per level.
template <int LEVELS, bool PAIR>
struct HistPlan {
    static constexpr int GROUP_BITS = 3;
    static_assert((1 << GROUP_BITS) == CHAINS, "digit groups == chains");
    static constexpr int COPIES = !PAIR ? (LEVELS <= 4 ? 32 : (LEVELS <= 8 ? 16 : 8))
                                        : (LEVELS <= 4 ? 32 : (LEVELS <= 8 ? 16 : 8));  // level 0 (all levels if !PAIR)
    static constexpr int PCOPIES = LEVELS <= 2 ? 8 : (LEVELS <= 4 ? 4 : (LEVELS <= 8 ? 2 : 1));
    static constexpr int PLAIN_LEVELS = PAIR ? 1 : LEVELS;
    static constexpr int PAIR_LEVELS = PAIR ? LEVELS - 1 : 0;
    static constexpr int PAIR_BASE = PLAIN_LEVELS * RADIX * COPIES;          // first word of the joint tables
    static constexpr int PAIR_WORDS = RADIX * CHAINS * PCOPIES;              // words per joint table
    static constexpr int WORDS = PAIR_BASE + PAIR_LEVELS * PAIR_WORDS;       // <= 32768 words = 128 KiB
    static_assert(WORDS <= 32768, "K1 LDS budget");
};

template <typename K, int LEVELS, int VEC, bool PAIR>
__global__ __launch_bounds__(HIST_THREADS) void hist_kernel(const K* __restrict__ keys, uint64_t n, K neg, K pos,
                                                            unsigned long long* __restrict__ hpos /* [LEVELS][CHAINS][256]: counts per position range */,
                                                            unsigned long long* __restrict__ hpair /* [LEVELS][CHAINS][256]: counts per group of the previous digit (PAIR) */,
                                                            uint32_t* __restrict__ inversion /* set if keys[i-1] > keys[i] anywhere */,
                                                            const Plan* __restrict__ plan /* nullable: the hybrid route has its own counts (K1h) */,
                                                            int base_level /* table row and digit of this kernel's level 0 (one-level counts of a single pass: LEVELS == 1) */) {
    using P = HistPlan<LEVELS, PAIR>;
    if (plan && (plan->route != ROUTE_LSD || plan->sorted_known)) return;  // (sorted: K2 turns every pass off whatever the counts)
    constexpr int COPIES = P::COPIES, PCOPIES = P::PCOPIES, WORDS = P::WORDS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* s_h = reinterpret_cast<uint32_t*>(smem);
    const int tid = threadIdx.x;
    for (int i = tid; i < WORDS; i += HIST_THREADS) s_h[i] = 0;
    __syncthreads();

    constexpr uint64_t GRAN = (uint64_t)HIST_THREADS * VEC * 4;  // one unrolled sweep of the block
    const uint64_t piece = hist_piece(n, gridDim.x, GRAN);
    const uint64_t p_begin = (uint64_t)blockIdx.x * piece;
    uint64_t p_end = p_begin + piece;
    if (p_end > n) p_end = n;
    uint32_t* mine = s_h + (tid & (COPIES - 1));                    // my bank column, plain tables
    uint32_t* mine_p = s_h + P::PAIR_BASE + (tid & (PCOPIES - 1));  // and of the joint tables

    // Besides counting, the sweep looks for an inversion in mapped-key order: a slice that is
    // already sorted needs no pass at all (the whole-slice form of rdst's already_sorted exits,
    // src/sorter.rs:59-65, src/sort_utils.rs:125-136).  `before` = mapped key at the previous index.
#ifndef RDST_NO_SORTED_CHECK
    bool inv = false;
#else
    bool inv = true;  // A/B build without the test: never claim "sorted"
#endif
    auto count = [&](K raw, K before, bool careful) -> K {
        const K m = map_key<K>(raw, neg, pos);
#ifndef RDST_NO_SORTED_CHECK
        inv |= before > m;
#endif
#pragma unroll
        for (int l = 0; l < P::PLAIN_LEVELS; ++l) {
            const uint32_t dg = digit_of(m, (l + base_level) * 8);
            uint32_t* w = &mine[(l * RADIX + dg) * COPIES];
            if (PAIR || !careful) atomicAdd(w, 1u);
            else if (COPIES >= 32 || __all((int)(dg == (uint32_t)__builtin_amdgcn_readfirstlane((int)dg))) == 0) atomicAdd(w, 1u);
            else if ((tid & 63) == 0) atomicAdd(w, 64u);
        }
#pragma unroll
        for (int l = 1; l <= P::PAIR_LEVELS; ++l) {
            // bits [8l-3, 8l+8): digit_l above the top GROUP_BITS bits of digit_{l-1}
            const uint32_t e = (uint32_t)(m >> (8 * l - P::GROUP_BITS)) & (uint32_t)(RADIX * CHAINS - 1);
            uint32_t* w = &mine_p[((l - 1) * RADIX * CHAINS + e) * PCOPIES];
            // sorted or low-entropy input puts a whole wave on one entry, which the few bank columns of
            // the joint tables would serialise 16 ways and more: one lane adds for the wave then
            if (careful && __all((int)(e == (uint32_t)__builtin_amdgcn_readfirstlane((int)e))) != 0) {
                if ((tid & 63) == 0) atomicAdd(w, 64u);
            } else {
                atomicAdd(w, 1u);
            }
        }
        return m;
    };
    auto mapped_at = [&](uint64_t idx) -> K { return idx == 0 ? (K)0 : map_key<K>(keys[idx - 1], neg, pos); };  // key before idx

    struct alignas(sizeof(K) * VEC) V { K e[VEC]; };
    uint64_t i = p_begin + (uint64_t)tid * VEC;
    constexpr uint64_t STRIDE = (uint64_t)HIST_THREADS * VEC;
    // the batched loop runs while the LAST lane of the wave still has a full batch: whole waves
    // enter and leave it together (the wave-wide shortcuts below count on all 64 lanes)
    const uint64_t lane_rest = (uint64_t)(63 - (tid & 63)) * VEC;
    for (; i + lane_rest + 3 * STRIDE + VEC <= p_end; i += 4 * STRIDE) {
        V v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const V*>(keys + i + u * STRIDE);
#ifndef RDST_NO_SORTED_CHECK
        // key before my vector = last key of the lane below (its vector ends where mine begins);
        // only lane 0 of a wave has to fetch it, in the same batch as the vectors
        K edge[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) edge[u] = ((tid & 63) == 0 && i + u * STRIDE > 0) ? keys[i + u * STRIDE - 1] : (K)0;
#endif
        // looking for wave-uniform entries costs a few instructions per key and level, so it is done
        // only for a batch whose first key has one in some level (random keys: never)
        bool careful = false;
        if constexpr (LEVELS >= 2) {
            const K m0 = map_key<K>(v[0].e[0], neg, pos);
#pragma unroll
            for (int l = 1; l < LEVELS; ++l) {
                const uint32_t e = (uint32_t)(m0 >> (8 * l - 3)) & (uint32_t)(RADIX * CHAINS - 1);
                careful |= __all((int)(e == (uint32_t)__builtin_amdgcn_readfirstlane((int)e))) != 0;
            }
        }
        auto batch = [&](bool c) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#ifndef RDST_NO_SORTED_CHECK
                K before = lane_below<K>(map_key<K>(v[u].e[VEC - 1], neg, pos));
                if ((tid & 63) == 0) before = (i + u * STRIDE > 0) ? map_key<K>(edge[u], neg, pos) : (K)0;
#else
                K before = 0;
#endif
#pragma unroll
                for (int e = 0; e < VEC; ++e) before = count(v[u].e[e], before, c);
            }
        };
        if (careful) batch(true);
        else batch(false);
    }
    for (; i < p_end; i += STRIDE) {  // remainder of the piece, element-wise
        K before = mapped_at(i);
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            if (i + e < p_end) before = count(keys[i + e], before, false);
    }
    if (inv) atomicOr(inversion, 1u);
    __syncthreads();
    // fold the copies: thread j owns (level, digit) pair j (+1024, ...); reading copy (c + j) % COPIES
    // in step c keeps the lanes of a half-wave on distinct banks.  The block's piece lies inside
    // one position range (hist_range_of), whose table receives the counts.
    const uint32_t range = hist_range_of(blockIdx.x, gridDim.x);
    for (int j = tid; j < LEVELS * RADIX; j += HIST_THREADS) {
        const int l = j / RADIX, d = j % RADIX;
        uint32_t c = 0;
        if (l < P::PLAIN_LEVELS) {
#pragma unroll 8
            for (int k = 0; k < COPIES; ++k) c += s_h[j * COPIES + ((k + j) & (COPIES - 1))];
        } else {
            const uint32_t* tab = s_h + P::PAIR_BASE + ((l - 1) * RADIX * CHAINS + d * CHAINS) * PCOPIES;
#pragma unroll
            for (int g = 0; g < CHAINS; ++g) {
                uint32_t cg = 0;
#pragma unroll
                for (int k = 0; k < PCOPIES; ++k) cg += tab[g * PCOPIES + k];
                if (cg) atomicAdd(&hpair[((size_t)l * CHAINS + g) * RADIX + d], (unsigned long long)cg);
                c += cg;
            }
        }
        if (c) atomicAdd(&hpos[((size_t)(l + base_level) * CHAINS + range) * RADIX + d], (unsigned long long)c);
    }
}

// ------------------------------------------------------------------------------------------
// A look before K1h.  An input the hybrid route cannot take (some 16-bit prefix holding more than a tile) would
// otherwise pay K1h's full read (0.8 ms per 10^9 u32 keys) before falling back to the LSD route.  One block counts
// the prefixes of 65 536 keys at evenly spaced positions (8-bit counters, 64 KiB of LDS): the largest bucket the
// route accepts expects about one hit, so PRESAMPLE_LIMIT hits on one prefix cannot happen by chance
// (Poisson(1.1) >= 12: 1e-9 per bucket) and mean a bucket thousands of times over the bound; K1h then returns at
// once and the sort goes the LSD way.  Milder skew is not seen here and is caught by K1h's exact counts as before.
// Cost: ~10 us per hybrid-eligible sort (one batch of eight loads per thread).
// ------------------------------------------------------------------------------------------
constexpr uint32_t PRESAMPLE_KEYS = 8192;  // 65 536 spread-out keys cost 0.17 ms (a TLB miss each); 8 192 in one batch of loads: ~0.01 ms
constexpr uint64_t PRESAMPLE_MIN_LEN = 1ull << 26;  // the shortest slice a byte-saving route is tried on by default (atomic_min_len): a full tile expects 2 hits there, the limit is 20
constexpr size_t presample_lds_bytes() { return 2 * H16_BINS + 32 + 4 * RADIX; }
// low halves of the sample that were seen before: S - D (1 - exp(-S / D)) for S = 8 192 samples over D equally likely values:
// ~490 on uniform keys (D = 65 536), 3 000 for D = 8 192, 5 000 for D ~ 3 500 — where a bucket of 15 000 keys holds each value
// 4-5 times on average and one of them 16 times often enough that K4's first kernel refuses a good part of the buckets
constexpr uint32_t PRESAMPLE_DUP_LIMIT = 5000;
constexpr uint32_t PRESAMPLE_TOP_LIMIT = 2 * PRESAMPLE_KEYS / RADIX;  // twice a top byte's share of the sample (Poisson(32) >= 64: 2e-7)

template <typename K, bool MAPPED>
__global__ __launch_bounds__(1024) void presample_kernel(const K* __restrict__ keys, uint64_t n, K neg, K pos, Plan* __restrict__ plan, uint32_t limit,
                                                         uint32_t giant_min /* 4-byte keys: buckets of this many keys and more are the giant kernels' (0: no such kernels here) */,
                                                         uint32_t giant_max /* 4-byte keys: count tables for that many such buckets */, uint32_t bucket_cap /* 8-byte keys: the largest bucket K4 takes */) {
    constexpr int W = sizeof(K) * 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* s_c = reinterpret_cast<uint32_t*>(smem);               // 65 536 8-bit counters, four per word
    uint32_t* s_skew = reinterpret_cast<uint32_t*>(smem + H16_BINS);   // [0] a prefix over the limit, [1] a top byte over its limit, [2] repeated low halves, [4] / [5] AND / OR of the top 16 bits
    uint32_t* s_top = reinterpret_cast<uint32_t*>(smem + H16_BINS + 32); // [256] the sample's top bytes
    uint32_t* s_low = reinterpret_cast<uint32_t*>(smem + H16_BINS + 32 + 4 * RADIX);  // 65 536 8-bit counters of the low halves (4-byte keys)
    const int tid = threadIdx.x;
    for (int i = tid; i < H16_BINS / 4; i += 1024) { s_c[i] = 0; s_low[i] = 0; }
    if (tid < RADIX) s_top[tid] = 0;
    if (tid < 8) s_skew[tid] = 0;  // ([4] / [5] are set below, before their first use)
    const uint64_t step = n / PRESAMPLE_KEYS;
    bool skew = false;
    static_assert(PRESAMPLE_KEYS == 8 * 1024, "eight keys per thread, one batch");
    K v[8];
    {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = keys[((uint64_t)u * 1024 + tid) * step];
        // the window: how many of their top 8 bits do the samples share?
        if (tid == 0) { s_skew[4] = 0xFFFFu; s_skew[5] = 0; }
        __syncthreads();
        uint32_t band = 0xFFFFu, bor = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            v[u] = MAPPED ? map_key<K>(v[u], neg, pos) : v[u];
            band &= (uint32_t)(v[u] >> (W - 16));
            bor |= (uint32_t)(v[u] >> (W - 16));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            band &= __shfl_xor(band, o);
            bor |= __shfl_xor(bor, o);
        }
        if ((tid & 63) == 0) { atomicAnd(&s_skew[4], band); atomicOr(&s_skew[5], bor); }
        __syncthreads();
        const uint32_t diff = (s_skew[4] ^ s_skew[5]) & 0xFFFFu;
        const uint32_t lead = diff ? (uint32_t)__builtin_clz(diff) - 16u : 16u;
        const uint32_t ws = lead < 8u ? lead : 8u;  // at most a byte: the buckets stay 16 bits of the key
        if (tid == 0) { plan->win_shift = ws; plan->win_top = ws ? s_skew[5] >> (16u - ws) : 0u; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const K m = v[u];
            const uint32_t b = (uint32_t)(m >> (W - 16 - (int)ws)) & 0xFFFFu;  // the bucket the atomic route would put the key in
            const uint32_t sh = (b & 3u) * 8u;
            const uint32_t old = atomicAdd(&s_c[b >> 2], 1u << sh);
            skew |= ((old >> sh) & 0xFFu) + 1u >= limit;  // limit <= 64: the flag fires long before a counter could carry
            atomicAdd(&s_top[b >> 8], 1u);
            if constexpr (sizeof(K) == 4) {
                const uint32_t lo = (uint32_t)m & 0xFFFFu, lsh = (lo & 3u) * 8u;
                const uint32_t seen = (atomicAdd(&s_low[lo >> 2], 1u << lsh) >> lsh) & 0xFFu;  // (a value 256 times carries: such a sample is far over the limit already)
                if (seen) atomicAdd(&s_skew[2], 1u);
            }
        }
    }
    if (skew) s_skew[0] = 1;
    __syncthreads();
    // the atomic route's areas hold a top byte's share of the keys plus a percent: a byte with twice its share of the sample
    // would overflow its areas a third of the way through pass A — that route is not tried (the hybrid one is)
    if (tid < RADIX && s_top[tid] >= PRESAMPLE_TOP_LIMIT) s_skew[1] = 1;
    // Where would exact counts send this sort?  A top byte that drew s samples holds about n * s / 8 192 keys; if they spread
    // evenly over its 256 buckets, each holds est = that / 256.  (They need not — a bimodal input puts half the slice into ONE
    // bucket — so only classes that a spread-out byte cannot leave by being lumpier are predicted: lumpier means bigger.)
    if (tid < RADIX) {
        const uint32_t hits = s_top[tid];
        const uint64_t est = n * hits / ((uint64_t)PRESAMPLE_KEYS * RADIX);
        if constexpr (sizeof(K) == 4) {
            // [7] bytes whose buckets look giant-sized (256 giants each)
            if (giant_min && hits >= 8 && est >= (uint64_t)giant_min + giant_min / 4) atomicAdd(&s_skew[7], 1u);
        } else {
            if (bucket_cap && hits >= 16 && est > (uint64_t)bucket_cap * 2) s_skew[3] = 1;  // some bucket of that byte is over the tile for sure
        }
    }
    __syncthreads();
    if (tid == 0) {
        bool lsd = false;
        if constexpr (sizeof(K) == 4) lsd = giant_min && (uint64_t)s_skew[7] * RADIX > (uint64_t)giant_max + giant_max / 2;
        else lsd = s_skew[3] != 0;
        if (lsd) plan->predict_lsd = 1;
    }
    if (tid == 0 && s_skew[0]) plan->gross_skew = 1;
    if (tid == 0 && s_skew[1]) plan->top_skew = 1;
    if (tid == 0 && s_skew[2] >= PRESAMPLE_DUP_LIMIT) plan->low_dups = 1;
}

// ------------------------------------------------------------------------------------------
// K1h: histogram of the TOP 16 BITS of the mapped key (65 536 buckets) from one read — what the
// hybrid route needs: bucket sizes (does every bucket fit K4's tile?), bucket starts, and, summed
// the right way, everything K2 wants for the two K3 passes on levels L-2 and L-1 (digit totals,
// the per-position-range counts of level L-2, the (digit_{L-1}, group of digit_{L-2}) joint counts).
// One LDS atomic per key.  Same sweep as K1 (same pieces, same inversion test).
//
// LDS: 65 536 16-bit counters, two per word (128 KiB).  A block counts up to n/grid keys, so a
// counter can overflow on skewed input; a low half then carries into the high half, a high half
// wraps.  Either way the DECODED counters sum to less than the keys the block has counted (a carry
// loses 65 535, a wrap 65 536, nothing ever gains), so one block-wide sum at the end detects any
// overflow exactly, at no cost per key — and a block with such a bucket could never have been on the
// hybrid route anyway.  The flag sends the sort down the LSD route.
// ------------------------------------------------------------------------------------------
constexpr int H16_WORDS = H16_BINS / 2;
#ifndef RDST_H16_BATCH
#define RDST_H16_BATCH 4
#endif

// GIANT (4-byte keys): buckets of any size are counted exactly.  A counter is then 15 bits and a guard bit: the add that sets
// the guard (it sees 32 767 or less before, 32 768 or more after) owns the wrap — it takes the 32 768 back out of the
// counter and adds them to the global tables at once.  The guard keeps a wrap from carrying into the neighbouring counter
// (for that, 32 768 more adds would have to hit the counter between the owner's two instructions).
template <typename K, int VEC, bool MAPPED, bool GIANT = false>
__global__ __launch_bounds__(HIST_THREADS) void hist16_kernel(const K* __restrict__ keys, uint64_t n, K neg, K pos,
                                                              uint32_t* __restrict__ h16 /* [65536] bucket counts */,
                                                              unsigned long long* __restrict__ hpos16 /* [2][CHAINS][256]: digits L-2 and L-1 per position range */,
                                                              uint32_t* __restrict__ inversion, uint32_t* __restrict__ overflow,
                                                              const Plan* __restrict__ plan, uint32_t pre_launch) {
    constexpr int W = sizeof(K) * 8;
    if (pre_launch) {  // before the MSD passes: only if the sample flagged the keys (the atomic route will not be tried)
        if (!(plan->top_skew || (GIANT && plan->gross_skew))) return;
    } else if (plan->pre) {
        return;  // counted already
    }
    if (plan->gross_skew && !GIANT) return;  // the sample ruled the hybrid route out: K1 will count (and look for inversions) instead
    if (plan->predict_lsd) return;            // ... or predicts that the counts would (route_kernel then leaves the route at LSD)
    if (plan->route == ROUTE_ATOMIC) return;  // the atomic route was tried first and took the sort
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* s_h = reinterpret_cast<uint32_t*>(smem);
    __shared__ unsigned long long s_sum;
    const int tid = threadIdx.x;
    for (int i = tid; i < H16_WORDS; i += HIST_THREADS) s_h[i] = 0;
    if (tid == 0) s_sum = 0;
    __syncthreads();

    constexpr uint64_t GRAN = (uint64_t)HIST_THREADS * VEC * 4;
    const uint64_t piece = hist_piece(n, gridDim.x, GRAN);
    const uint64_t p_begin = (uint64_t)blockIdx.x * piece;
    uint64_t p_end = p_begin + piece;
    if (p_end > n) p_end = n;

    bool inv = false;
    auto mapped = [&](K raw) -> K { return MAPPED ? map_key<K>(raw, neg, pos) : raw; };
    auto count = [&](K raw, K before, bool careful) -> K {
        const K m = mapped(raw);
        inv |= before > m;
        const uint32_t b = (uint32_t)(m >> (W - 16));
        uint32_t* w = &s_h[b >> 1];
        const uint32_t sh = (b & 1u) * 16;
        const uint32_t inc = 1u << sh;
        // sorted or low-entropy input puts a whole wave on one counter (64 serialised atomics): one lane adds then
        const bool uniform = careful && __all((int)(b == (uint32_t)__builtin_amdgcn_readfirstlane((int)b))) != 0;
        if constexpr (GIANT) {
            const uint32_t k = uniform ? 64u : 1u;
            if (!uniform || (tid & 63) == 0) {
                const uint32_t before_add = (atomicAdd(w, k << sh) >> sh) & 0xFFFFu;
                if (before_add < 0x8000u && before_add + k >= 0x8000u) {  // this add set the guard bit: the wrap is mine
                    atomicSub(w, 0x8000u << sh);
                    atomicAdd(&h16[b], 0x8000u);
                    atomicAdd(&hpos16[(size_t)hist_range_of(blockIdx.x, gridDim.x) * RADIX + (b & 255u)], 0x8000ull);
                    atomicAdd(&hpos16[(size_t)(CHAINS + hist_range_of(blockIdx.x, gridDim.x)) * RADIX + (b >> 8)], 0x8000ull);
                }
            }
        } else if (uniform) {
            if ((tid & 63) == 0) atomicAdd(w, inc * 64u);
        } else {
            atomicAdd(w, inc);
        }
        return m;
    };
    auto mapped_at = [&](uint64_t idx) -> K { return idx == 0 ? (K)0 : mapped(keys[idx - 1]); };

    struct alignas(sizeof(K) * VEC) V { K e[VEC]; };
    uint64_t i = p_begin + (uint64_t)tid * VEC;
    constexpr uint64_t STRIDE = (uint64_t)HIST_THREADS * VEC;
    constexpr int NB = RDST_H16_BATCH;       // vectors per lane and batch
    constexpr uint64_t S4 = NB * STRIDE;  // one batch
    const uint64_t lane_rest = (uint64_t)(63 - (tid & 63)) * VEC;
    // full batches of this wave (while its LAST lane still has one: whole waves enter and leave together)
    const uint64_t need = i + lane_rest + (NB - 1) * STRIDE + VEC;
    const uint64_t nb = need <= p_end ? (p_end - need) / S4 + 1 : 0;
    // Loads run one batch ahead of the counting (two register sets, ping-pong): with one block per CU a
    // wave that waits for its own loads before counting leaves the memory pipeline idle meanwhile.
    auto load = [&](V (&v)[NB], K (&edge)[NB], uint64_t at) {
#pragma unroll
        for (int u = 0; u < NB; ++u) v[u] = *reinterpret_cast<const V*>(keys + at + u * STRIDE);  // (non-temporal loads: 0.89 instead of 0.80 ms)
        // key before my vector = last key of the lane below; only lane 0 of a wave has to fetch it
#pragma unroll
        for (int u = 0; u < NB; ++u) edge[u] = ((tid & 63) == 0 && at + u * STRIDE > 0) ? keys[at + u * STRIDE - 1] : (K)0;
    };
    auto proc = [&](V (&v)[NB], K (&edge)[NB], uint64_t at) {
        const uint32_t b0 = (uint32_t)(mapped(v[0].e[0]) >> (W - 16));
        const bool careful = __all((int)(b0 == (uint32_t)__builtin_amdgcn_readfirstlane((int)b0))) != 0;
        auto batch = [&](bool c) {
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                K before = lane_below<K>(mapped(v[u].e[VEC - 1]));
                if ((tid & 63) == 0) before = (at + u * STRIDE > 0) ? mapped(edge[u]) : (K)0;
#pragma unroll
                for (int e = 0; e < VEC; ++e) before = count(v[u].e[e], before, c);
            }
        };
        if (careful) batch(true);
        else batch(false);
    };
    {
        V va[NB], vb[NB];
        K ea[NB], eb[NB];
        uint64_t b = 0;
        if (nb) load(va, ea, i);
        while (b + 2 <= nb) {  // va holds batch b
            load(vb, eb, i + S4);
            proc(va, ea, i);
            if (b + 2 < nb) load(va, ea, i + 2 * S4);
            proc(vb, eb, i + S4);
            i += 2 * S4;
            b += 2;
        }
        if (b < nb) {
            proc(va, ea, i);
            i += S4;
        }
    }
    for (; i < p_end; i += STRIDE) {
        K before = mapped_at(i);
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            if (i + e < p_end) before = count(keys[i + e], before, false);
    }
    if (inv) atomicOr(inversion, 1u);
    __syncthreads();

    // fold into the global table; the decoded counters must add up to the keys of the piece
    unsigned long long local = 0;
    for (int w = tid; w < H16_WORDS; w += HIST_THREADS) {
        const uint32_t v = s_h[w];
        const uint32_t lo = v & 0xFFFFu, hi = v >> 16;
        local += lo + hi;
        if (lo) atomicAdd(&h16[2 * w], lo);
        if (hi) atomicAdd(&h16[2 * w + 1], hi);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((tid & 63) == 0) atomicAdd(&s_sum, local);
    // digit L-2 (the low byte of the bucket index) of this piece, for its position range's table
    {
        const int d = tid & 255, q = tid >> 8;  // four threads per digit, 64 values of the high byte each
        uint32_t c = 0;
        for (int h = q * 64; h < q * 64 + 64; ++h) {
            const uint32_t v = s_h[h * 128 + (d >> 1)];
            c += (d & 1) ? (v >> 16) : (v & 0xFFFFu);
        }
        const uint32_t range = hist_range_of(blockIdx.x, gridDim.x);
        if (c) atomicAdd(&hpos16[(size_t)range * RADIX + d], (unsigned long long)c);
        // and digit L-1 (the high byte): if level L-2 turns out trivial, pass L-1 is the first to run and splits its source
        // by position.  (Same-bank reads, 64 per thread, once per block.)
        uint32_t ch = 0;
        for (int lo2 = q * 32; lo2 < q * 32 + 32; ++lo2) {
            const uint32_t v = s_h[d * 128 + lo2];
            ch += (v >> 16) + (v & 0xFFFFu);
        }
        if (ch) atomicAdd(&hpos16[(size_t)(CHAINS + range) * RADIX + d], (unsigned long long)ch);
    }
    __syncthreads();
    if (tid == 0 && !GIANT) {  // (GIANT: wraps were moved to the global tables as they happened, nothing is lost)
        const uint64_t counted = p_end > p_begin ? p_end - p_begin : 0;
        if (s_sum != counted) atomicOr(overflow, 1u);
    }
}

// ------------------------------------------------------------------------------------------
// Route decision + the hybrid route's tables: one block.  Thread t owns buckets [64t, 64t + 64), i.e.
// high byte t / 4 and two digit groups of the low byte.  If no counter overflowed and the largest
// bucket fits K4's tile: route = HYBRID, bucket starts (exclusive scan in bucket order = key order),
// and the count tables K2 reads for levels L-2 and L-1 in the places K1 would have put them.
// ------------------------------------------------------------------------------------------
struct RouteArgs {
    const uint32_t* h16;              // [65536]
    const unsigned long long* hpos16; // [2][CHAINS][256]
    const uint32_t* overflow;
    const uint32_t* inversion;        // K1h's "some key is smaller than its predecessor"
    uint32_t* bstart;                 // [65537] out
    unsigned long long* hpos;         // [levels][CHAINS][256] (zeroed)
    unsigned long long* hpair;        // [levels][CHAINS][256] (zeroed)
    Plan* plan;
    uint64_t n;
    uint32_t levels, cap, allow_skip;
    // 4-byte keys: buckets of 65 536 keys and more ("giants") are sorted by the giant kernels of K4 — at most giant_max of
    // them (a 256-KiB table each); 0: such a bucket sends the sort down the LSD route
    uint32_t giant_max;
    uint32_t* glist;        // [giant_max] out: the giants, in bucket order
    uint32_t* gcount_item;  // [giant_max + 1] out: first work item of each in the counting kernel (two per chunk of GIANT_CHUNK keys)
    uint32_t* gexp_item;    // [giant_max + 1] out: first work item of each in the expanding kernel (one per GIANT_OUT positions)
    // the launch before the MSD passes (Plan::pre): on HYBRID it also sets the exact form of those passes up
    uint32_t pre_launch;
    uint32_t msd_tile;      // keys per tile of the MSD passes
    uint32_t* cursor_a;     // [8][256] out: where the keys of top digit d from position range r go in tmp
    uint32_t* cursor_b;     // [65536] out: where bucket b's keys go (= bstart[b])
    uint32_t* xtile0;       // [257] out: first tile of top digit d's region in pass B's grid
    uint32_t skip_a_ok;     // pass B can read the caller's slice in place of pass A's output (4-byte keys: it writes halves elsewhere)
};
constexpr uint32_t GIANT_MIN = 65536;      // keys: below it the 16-bit counters of the LDS kernels do
constexpr uint32_t GIANT_CHUNK = 1u << 19; // keys a block of the giant counting kernel streams per item (and value half), at least
// chunk of a giant of cnt keys when the sort has `giants` of them: about 256 chunks in all and never more than 128 per giant —
// every item ends with up to 32 768 adds to the giant's table (a 10^9-key giant cut into 2^19-key chunks spent a third of its
// time on 125 M of them; 1 800 giants of 10^6 keys likewise), and a giant that is ONE chunk needs no adds at all: its two
// items (the halves of the value range) store their counters
__host__ __device__ inline uint32_t giant_chunk_of(uint32_t cnt, uint32_t giants) {
    const uint32_t parts = giants >= 256u ? 1u : (256u / giants > 128u ? 128u : 256u / giants);
    const uint32_t c = ((cnt + parts - 1u) / parts + 8191u) & ~8191u;
    return c > GIANT_CHUNK ? c : GIANT_CHUNK;
}
constexpr uint32_t GIANT_OUT = 1u << 12;   // positions a block of the giant expanding kernel writes per item
constexpr uint32_t GIANT_TABLE = H16_BINS + 16;  // words per giant: 65 536 counts / prefixes, then the total

__global__ __launch_bounds__(1024) void route_kernel(RouteArgs a) {
    __shared__ uint32_t s_wsum[16], s_wmax[16], s_wg[16], s_wci[16], s_wei[16], s_tiles[RADIX], s_tw[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.plan->route == ROUTE_ATOMIC) return;  // tried first, and it took the sort (msd_finish_kernel): nothing to decide
    if (a.plan->predict_lsd) return;            // K1h did not count: the route stays LSD (the cleared plan / msd_finish_kernel)
    if (a.pre_launch) {
        if (!(a.plan->top_skew || (a.giant_max && a.plan->gross_skew))) return;  // K1h did not run either: the atomic route goes first
    } else if (a.plan->pre) {
        return;  // decided before the MSD passes
    }
    uint32_t c[64];
    const uint4* src = reinterpret_cast<const uint4*>(a.h16) + (size_t)tid * 16;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint4 v = src[k];
        c[4 * k] = v.x; c[4 * k + 1] = v.y; c[4 * k + 2] = v.z; c[4 * k + 3] = v.w;
    }
    const bool giants_ok = a.giant_max != 0;
    uint32_t sum0 = 0, sum1 = 0, mx = 0;  // mx: the largest bucket the LDS kernels of K4 would have to take
    uint32_t ng = 0;                      // my giants
#pragma unroll
    for (int k = 0; k < 64; ++k) {
        if (k < 32) sum0 += c[k];
        else sum1 += c[k];
        if (giants_ok && c[k] >= GIANT_MIN) ++ng;
        else mx = c[k] > mx ? c[k] : mx;
    }
    const uint32_t mine = sum0 + sum1;
    uint32_t incl = mine, wmax = mx, ig = ng;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(incl, o), yg = __shfl_up(ig, o);
        if (lane >= o) { incl += y; ig += yg; }
        const uint32_t m = __shfl_xor(wmax, o);
        wmax = m > wmax ? m : wmax;
    }
    if (lane == 63) { s_wsum[wave] = incl; s_wg[wave] = ig; }
    if (lane == 0) s_wmax[wave] = wmax;
    __syncthreads();
    uint32_t excl = incl - mine, bmax = 0, eg = ig - ng, tg = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) { excl += s_wsum[w]; eg += s_wg[w]; }
        bmax = s_wmax[w] > bmax ? s_wmax[w] : bmax;
        tg += s_wg[w];
    }
    // the giants' work items (the chunking depends on how many giants there are: a second scan)
    uint32_t nci = 0, nei = 0;
    if (ng) {
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            if (c[k] >= GIANT_MIN) {
                const uint32_t ck = giant_chunk_of(c[k], tg);
                nci += 2u * ((c[k] + ck - 1) / ck);
                nei += (c[k] + GIANT_OUT - 1) / GIANT_OUT;
            }
        }
    }
    uint32_t ici = nci, iei = nei;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t yc = __shfl_up(ici, o), ye = __shfl_up(iei, o);
        if (lane >= o) { ici += yc; iei += ye; }
    }
    if (lane == 63) { s_wci[wave] = ici; s_wei[wave] = iei; }
    __syncthreads();
    uint32_t eci = ici - nci, eei = iei - nei, tci = 0, tei = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) { eci += s_wci[w]; eei += s_wei[w]; }
        tci += s_wci[w]; tei += s_wei[w];
    }
    // (buckets over the counting K4's tile are no reason to leave: the expanding kernel takes 10^9 keys in such buckets in 2.3-2.8 ms,
    // which keeps the hybrid route 0.7 ms and more ahead of the LSD one — round 2's form of it, at 7 ns per 1 000 keys, did not)
    const bool hybrid = *a.overflow == 0 && bmax <= a.cap && (giants_ok ? tg <= a.giant_max : a.plan->gross_skew == 0);  // (gross skew without the giant kernels: K1h returned at once, its counts are all zero)
    if (tid == 0) {
        a.plan->route = hybrid ? ROUTE_HYBRID : ROUTE_LSD;
        if (a.pre_launch) a.plan->pre = 1;
        a.plan->sorted_known = ((giants_ok || a.plan->gross_skew == 0) && *a.inversion == 0 && a.allow_skip) ? 1u : 0u;
        a.plan->giants = hybrid ? tg : 0u;
        a.plan->giant_count_items = hybrid ? tci : 0u;
        a.plan->giant_expand_items = hybrid ? tei : 0u;
    }
    if (!hybrid) return;
    if (giants_ok && tg) {
        const uint32_t bucket0 = (uint32_t)tid * 64u;
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            if (c[k] >= GIANT_MIN) {
                a.glist[eg] = bucket0 + (uint32_t)k;
                a.gcount_item[eg] = eci;
                a.gexp_item[eg] = eei;
                ++eg;
                eci += 2u * ((c[k] + giant_chunk_of(c[k], tg) - 1) / giant_chunk_of(c[k], tg));
                eei += (c[k] + GIANT_OUT - 1) / GIANT_OUT;
            }
        }
        if (tid == 1023) { a.gcount_item[tg] = tci; a.gexp_item[tg] = tei; }
    }
    uint32_t run = excl;
    uint4* dst = reinterpret_cast<uint4*>(a.bstart) + (size_t)tid * 16;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        uint4 v;
        v.x = run; run += c[4 * k];
        v.y = run; run += c[4 * k + 1];
        v.z = run; run += c[4 * k + 2];
        v.w = run; run += c[4 * k + 3];
        dst[k] = v;
    }
    if (tid == 1023) a.bstart[H16_BINS] = run;  // == n
    const uint32_t top = a.levels - 1, dh = (uint32_t)tid >> 2, q = (uint32_t)tid & 3u;
    // level L-1, pair counts: (digit dh, group of the level L-2 digit)
    a.hpair[((size_t)top * CHAINS + 2 * q) * RADIX + dh] = sum0;
    a.hpair[((size_t)top * CHAINS + 2 * q + 1) * RADIX + dh] = sum1;
    if (a.pre_launch) {
        // the exact form of the MSD passes: every claim counter starts where its digit's / bucket's keys belong ...
        uint32_t r2 = excl;
        uint32_t* cb = a.cursor_b + (size_t)tid * 64;
#pragma unroll
        for (int k = 0; k < 64; ++k) { cb[k] = r2; r2 += c[k]; }
        uint32_t dtot = mine;  // my digit's keys: four threads per top digit
        dtot += __shfl_xor(dtot, 1);
        dtot += __shfl_xor(dtot, 2);
        if ((tid & 3) == 0) {
            // pass A claims per (position range, top digit): the digit's start (thread 4 d owns its first 64 buckets: its
            // exclusive sum) + the digit's keys in the earlier ranges (K1h counted them per range)
            uint32_t at = excl;
            const uint32_t d = (uint32_t)tid >> 2;
            for (int r = 0; r < CHAINS; ++r) {
                a.cursor_a[r * RADIX + d] = at;
                at += (uint32_t)a.hpos16[(size_t)(CHAINS + r) * RADIX + d];
            }
            s_tiles[d] = (dtot + a.msd_tile - 1) / a.msd_tile;
            if (a.skip_a_ok && (uint64_t)dtot == a.n) a.plan->pre_skip_a = 1;
        }
        __syncthreads();
        // ... and pass B's grid: the tiles of digit d's region start at xtile0[d]
        if (tid < RADIX) {
            const uint32_t t = s_tiles[tid];
            uint32_t it = t;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t y = __shfl_up(it, o);
                if (lane >= o) it += y;
            }
            if (lane == 63) s_tw[wave] = it;
            s_tiles[tid] = it - t;  // exclusive inside the wave
        }
        __syncthreads();
        if (tid < RADIX) {
            uint32_t off = 0;
            for (int w = 0; w < wave; ++w) off += s_tw[w];
            a.xtile0[tid] = s_tiles[tid] + off;
            if (tid == RADIX - 1) a.xtile0[RADIX] = s_tw[0] + s_tw[1] + s_tw[2] + s_tw[3];
        }
    }
    // levels L-2 and L-1, per position range: as K1h counted them (pass L-1 normally runs second and takes its chains from the
    // pair counts; with a trivial level L-2 it runs first and splits by position)
    for (int j = tid; j < CHAINS * RADIX; j += 1024) {
        a.hpos[(size_t)(top - 1) * CHAINS * RADIX + j] = a.hpos16[j];
        a.hpos[(size_t)top * CHAINS * RADIX + j] = a.hpos16[CHAINS * RADIX + j];
    }
}

// ROUTE_ATOMIC, after its two scatter passes: did every claim fit?  Then the 65 536 claim counters are the bucket lengths:
// their exclusive scan gives every bucket's place in the sorted slice (thread t owns 64 buckets, as in route_kernel).
struct MsdFinishArgs {
    const uint32_t* cursor_b;   // [65536] keys claimed per slot
    const uint32_t* overflow;
    const uint32_t* inversion;
    uint32_t* bstart;           // [65537] out
    Plan* plan;
    uint64_t n;
    uint32_t allow_skip;
};

__global__ __launch_bounds__(1024) void msd_finish_kernel(MsdFinishArgs a) {
    __shared__ uint32_t s_wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.plan->pre) return;  // the passes ran in their exact form for the hybrid route (or not at all): decided already
    const bool ok = a.plan->gross_skew == 0 && a.plan->top_skew == 0 && a.plan->predict_lsd == 0 && *a.overflow == 0;
    const bool sorted = ok && a.allow_skip && *a.inversion == 0;
    if (tid == 0) {
        a.plan->route = ok ? ROUTE_ATOMIC : ROUTE_LSD;
        a.plan->local_sort = ok && !sorted ? 1u : 0u;
        a.plan->sorted_known = sorted ? 1u : 0u;
    }
    if (!ok || sorted) return;
    uint32_t c[64];
    const uint4* src = reinterpret_cast<const uint4*>(a.cursor_b) + (size_t)tid * 16;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint4 v = src[k];
        c[4 * k] = v.x; c[4 * k + 1] = v.y; c[4 * k + 2] = v.z; c[4 * k + 3] = v.w;
    }
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < 64; ++k) mine += c[k];
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    uint32_t run = incl - mine;
    for (int w = 0; w < wave; ++w) run += s_wsum[w];
    uint4* dst = reinterpret_cast<uint4*>(a.bstart) + (size_t)tid * 16;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        uint4 v;
        v.x = run; run += c[4 * k];
        v.y = run; run += c[4 * k + 1];
        v.z = run; run += c[4 * k + 2];
        v.w = run; run += c[4 * k + 3];
        dst[k] = v;
    }
    if (tid == 1023) a.bstart[H16_BINS] = run;  // == n
}

// ------------------------------------------------------------------------------------------
// K2: one block of 256 threads.  Digit totals of every level (sum of the range tables), their
// exclusive scan -> bucket start table, the skip plan, and for every executed level the chain
// tables: segment bounds, tile counts, ticket order, and each chain's per-digit start
// (bucket start + the digit's count in the earlier segments).
// ------------------------------------------------------------------------------------------
struct ScanArgs {
    const unsigned long long* hpos;   // [levels][CHAINS][256] from K1
    const unsigned long long* hpair;  // [levels][CHAINS][256] from K1 (level 0 unused)
    unsigned long long* hist;         // [levels][256] out: digit totals
    uint64_t* base;                   // [levels][256] out
    uint64_t* cbase;                  // [levels][CHAINS][256] out
    LevelChains* chains;              // [levels] out
    Plan* plan;
    uint32_t* tickets;                // [levels][TICKET_ROW]: K2 marks the empty chains in the mask word
    const uint32_t* inversion;
    uint64_t n, hist_piece;
    uint32_t levels, allow_skip, level_lo, level_hi, hist_grid, tile, use_chains;
    uint32_t halves;  // hybrid route hands K4 the 16-bit halves pass L-1 writes: that pass must run even if its level is trivial
    uint32_t deliver_tmp;  // the caller wants the result in tmp (run_split_sort's parts): a K4 that writes whole keys from the workspace writes them there
};

constexpr int SCAN_GROUPS = 4;  // levels handled side by side, 256 threads (one per digit) each

__global__ __launch_bounds__(256 * SCAN_GROUPS) void scan_kernel(ScanArgs a) {
    __shared__ uint64_t s_wave_sum[SCAN_GROUPS][4];
    __shared__ uint32_t s_trivial[MAX_LEVELS];
    __shared__ uint32_t s_mode[MAX_LEVELS], s_skip[MAX_LEVELS];
    __shared__ uint64_t s_seg[SCAN_GROUPS][CHAINS + 1];
    __shared__ uint64_t s_group_start[MAX_LEVELS][CHAINS];  // bucket start of the first digit of each digit group
    const int g = threadIdx.x / RADIX, d = threadIdx.x % RADIX, lane = d & 63, wv = d >> 6;
    if (threadIdx.x < MAX_LEVELS) s_trivial[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t l0 = 0; l0 < a.levels; l0 += SCAN_GROUPS) {
        const uint32_t l = l0 + (uint32_t)g;
        const bool on = l < a.levels;
        uint64_t total = 0;
        if (on) {
            for (int r = 0; r < CHAINS; ++r) total += a.hpos[((size_t)l * CHAINS + r) * RADIX + d];
            a.hist[(size_t)l * RADIX + d] = total;
            if (total == a.n) s_trivial[l] = 1;
        }
        // inclusive scan over the group's 256 totals: shuffles inside a wave, then the waves' sums
        uint64_t incl = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63) s_wave_sum[g][wv] = incl;
        __syncthreads();
        uint64_t excl = incl - total;
        for (int w = 0; w < wv; ++w) excl += s_wave_sum[g][w];
        if (on) {
            a.base[(size_t)l * RADIX + d] = excl;
            if (d % (RADIX / CHAINS) == 0) s_group_start[l][d / (RADIX / CHAINS)] = excl;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        uint32_t in_tmp = 0, executed = 0;
        int prev = -1;  // last executed level
        const bool already_sorted = a.allow_skip && *a.inversion == 0;  // nothing to do at all
        // hybrid route: only the two top levels are scatter passes, K4 does the rest
        const bool hybrid = a.plan->route == ROUTE_HYBRID;
        const bool atomic_done = a.plan->route == ROUTE_ATOMIC;  // its own kernels did (or do) everything: no scatter pass of K3 runs
        const bool by_msd = hybrid && a.plan->pre != 0;  // the MSD passes ordered the slice by the top 16 bits (exact form): no K3 pass runs
        const uint32_t level_lo = by_msd ? a.levels : (hybrid ? a.levels - 2 : (atomic_done ? a.levels : a.level_lo));
        if (!atomic_done) a.plan->local_sort = hybrid && !already_sorted ? 1u : 0u;
        for (uint32_t l = 0; l < MAX_LEVELS; ++l) {
            const bool active = l >= level_lo && l < a.level_hi && l < a.levels;
            const bool needed = hybrid && a.halves && l + 1 == a.levels;
            const bool skip = !active || already_sorted || (a.allow_skip && s_trivial[l] && !needed);
            a.plan->skip[l] = s_skip[l] = skip ? 1u : 0u;
            a.plan->src_is_tmp[l] = in_tmp;
            uint32_t mode = CHAIN_ONE;
            if (!skip && a.use_chains) {
                if (prev < 0) mode = CHAIN_POS;  // reads the array K1 counted, in K1's order
                else if (prev == (int)l - 1 && a.hpair) mode = CHAIN_PAIR;
            }
            a.plan->chain_mode[l] = s_mode[l] = mode;
            if (!skip) {
                if (prev < 0) a.plan->first_level = l;
                in_tmp ^= 1u; ++executed; prev = (int)l;
            }
        }
        // (hybrid route with the 16-bit hand-off: K4 reads the halves from the workspace and writes whole keys — to the caller's
        // array, whichever buffer the last pass read: no copy-back even after an odd number of passes)
        // (the atomic route's K4 reads its slots: the same, and either buffer is as good a destination — deliver_tmp picks)
        const bool from_workspace = (hybrid && a.halves) || atomic_done;
        a.plan->result_in_tmp = from_workspace ? ((a.deliver_tmp && a.plan->local_sort) ? 1u : 0u) : in_tmp;
        a.plan->executed = executed;
    }
    __syncthreads();
    for (uint32_t l0 = 0; l0 < a.levels; l0 += SCAN_GROUPS) {
        const uint32_t l = l0 + (uint32_t)g;
        const bool on = l < a.levels && !s_skip[l];
        const uint32_t mode = on ? s_mode[l] : CHAIN_ONE;
        LevelChains* lc = a.chains + l;
        if (on && d <= CHAINS) {
            uint64_t lo;
            if (d == CHAINS) lo = a.n;
            else if (mode == CHAIN_POS) lo = (uint64_t)hist_first_block((uint32_t)d, a.hist_grid) * a.hist_piece;
            else if (mode == CHAIN_PAIR) lo = s_group_start[l - 1][d];
            else lo = d == 0 ? 0 : a.n;
            s_seg[g][d] = lo < a.n ? lo : a.n;
        }
        __syncthreads();
        if (on && d == 0) {
            uint32_t row = 0;
            for (int c = 0; c < CHAINS; ++c) {
                const uint64_t lo = s_seg[g][c], hi = s_seg[g][c + 1];
                lc->seg_lo[c] = lo;
                const uint32_t nt = hi > lo ? (uint32_t)((hi - (lo & ~(uint64_t)(TILE_ALIGN - 1)) + a.tile - 1) / a.tile) : 0u;
                lc->ntiles[c] = nt;
                lc->row0[c] = row;
                row += nt;
            }
            lc->seg_lo[CHAINS] = s_seg[g][CHAINS];
            lc->total_tiles = row;
            // walkers of such a chain skip the L2-resident copy of the status rows: most of its
            // tiles are written on other XCDs
            uint32_t empty = 0;
            for (int c = 0; c < CHAINS; ++c) {
                lc->shared[c] = (uint64_t)lc->ntiles[c] * CHAINS * 2 > (uint64_t)row * 3 ? 1u : 0u;
                if (lc->ntiles[c] == 0) empty |= 1u << c;
            }
            a.tickets[(size_t)l * TICKET_ROW + CHAINS * TICKET_STRIDE] = empty;
        }
        if (on) {  // chain c's first destination of digit d
            uint64_t acc = a.base[(size_t)l * RADIX + d];
            for (int c = 0; c < CHAINS; ++c) {
                a.cbase[((size_t)l * CHAINS + c) * RADIX + d] = acc;
                uint64_t cnt;
                if (mode == CHAIN_POS) cnt = a.hpos[((size_t)l * CHAINS + c) * RADIX + d];
                else if (mode == CHAIN_PAIR) cnt = a.hpair[((size_t)l * CHAINS + c) * RADIX + d];
                else cnt = c == 0 ? a.hist[(size_t)l * RADIX + d] : 0;
                acc += cnt;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// K3: one stable scatter pass.  grid >= number of tiles, block = NWAVES*64, one tile of
// NWAVES*64*KPT keys per block.  The source is split into CHAINS segments with a look-back chain
// each (LevelChains); a block takes the next tile of "its" chain from that chain's ticket counter,
// so that a tile's predecessors have always started: forward progress of the look-back does not
// depend on the order in which the hardware dispatches workgroups, nor on where (which XCD) they land.
// Every cross-workgroup word is an agent-scope relaxed atomic whose value carries its own state
// bits (no separate flag, hence no fence).
//   S       status word: u32 while every prefix fits 30 bits (n < 2^30), u64 above
//   MAPPED  key kind needs the order-preserving map (signed / float); unsigned keys skip it
//   NARROW  n * sizeof(K) < 2^32: destinations are 32-bit byte offsets from a uniform base
// ------------------------------------------------------------------------------------------

#ifdef RDST_EXPERIMENTS
__device__ uint32_t* g_exp_stats = nullptr;  // [tiles][4] look-back records of pass 0 (tools/ only)
__device__ uint32_t* g_exp_timeline = nullptr;  // [tiles][12] shader-clock stamps of thread 0 along a tile of pass 0
__device__ uint32_t g_exp_timeline_kernel = 0;  // whose stamps: 0 K3 (level 0), 1 local_wide2_sort_kernel, 2 / 3 msd_scatter_kernel pass A / B, 5 local_expand_sort_kernel
#define RDST_STAMP(k) do { if (tl_on) tl[k] = (uint32_t)__builtin_amdgcn_s_memtime(); } while (0)
// the same for the other kernels: RDST_TL_BEGIN(which) declares the record, RDST_TL_END(row) stores it (slots 10, 11: XCC id, block)
#define RDST_TL_BEGIN(which) uint32_t tl[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const bool tl_on = g_exp_timeline != nullptr && g_exp_timeline_kernel == (which) && threadIdx.x == 0; RDST_STAMP(0)
#define RDST_TL_END(row) do { if (tl_on) { uint32_t* rec_ = g_exp_timeline + (size_t)(row) * 12; for (int k_ = 0; k_ < 10; ++k_) rec_[k_] = tl[k_]; \
    rec_[10] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); rec_[11] = blockIdx.x; } } while (0)
#else
#define RDST_STAMP(k) do {} while (0)
#define RDST_TL_BEGIN(which) do {} while (0)
#define RDST_TL_END(row) do {} while (0)
#endif

// lanes below me holding my digit, 4 VALU per bit: my bit as a 0 / -1 mask (v_bfe_i32), the
// wave's ballot of that bit (v_cmp), then per 32-lane half ONE v_bitop3_b32 that keeps in `same`
// only the lanes whose bit equals mine:  same &= ~(ballot ^ my_bit)   (truth table 0x90).
__device__ __forceinline__ uint32_t peers_below(uint32_t word, int bit0) {
    // bit by bit (ballot, then the two mask updates that read it): computing the eight ballots first
    // removes the wait states after each ballot but measured 4-6 % slower (64-bit encodings, 16 more SGPRs live)
    uint32_t same_lo = ~0u, same_hi = ~0u;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const int m = __builtin_amdgcn_sbfe((int)word, (unsigned)(bit0 + b), 1u);  // 0 or -1
        const uint64_t bal = __builtin_amdgcn_ballot_w64(m != 0);
        same_lo = __builtin_amdgcn_bitop3_b32(same_lo, (uint32_t)bal, (uint32_t)m, 0x90);
        same_hi = __builtin_amdgcn_bitop3_b32(same_hi, (uint32_t)(bal >> 32), (uint32_t)m, 0x90);
    }
    return __builtin_amdgcn_mbcnt_hi(same_hi, __builtin_amdgcn_mbcnt_lo(same_lo, 0u));
}

// the same, plus the size of my digit's group in this wave (heavy-digit path)
__device__ __forceinline__ uint32_t peers_below_total(uint32_t word, int bit0, uint32_t& total) {
    uint32_t same_lo = ~0u, same_hi = ~0u;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const int m = __builtin_amdgcn_sbfe((int)word, (unsigned)(bit0 + b), 1u);
        const uint64_t bal = __builtin_amdgcn_ballot_w64(m != 0);
        same_lo = __builtin_amdgcn_bitop3_b32(same_lo, (uint32_t)bal, (uint32_t)m, 0x90);
        same_hi = __builtin_amdgcn_bitop3_b32(same_hi, (uint32_t)(bal >> 32), (uint32_t)m, 0x90);
    }
    total = (uint32_t)__builtin_popcount(same_lo) + (uint32_t)__builtin_popcount(same_hi);
    return __builtin_amdgcn_mbcnt_hi(same_hi, __builtin_amdgcn_mbcnt_lo(same_lo, 0u));
}

template <typename K>
__device__ __forceinline__ uint32_t digit_word(K mapped, int shift) {  // 32-bit half that holds the digit
    if constexpr (sizeof(K) > 4) return (uint32_t)(mapped >> (shift & ~31));
    else return (uint32_t)mapped;
}

// Decoupled look-back: thread d walks digit d's status words of rows t-1, t-2, ... and returns in
// `excl` the sum up to and including the first INCLUSIVE one.  The walk is the latency chain of
// the whole pass (each round trip crosses the fabric), so RDST_LB_WINDOW predecessor words are
// fetched at once and then consumed in order.  Bounded: returns false if a word stays EMPTY.
template <typename S>
__device__ __forceinline__ bool lookback_walk(const S* __restrict__ status, const S* __restrict__ near_rows, uint32_t t, int tid,
                                              const uint32_t* err, uint64_t& excl
#ifdef RDST_EXPERIMENTS
                                              , int level, uint32_t stat_row
#endif
) {
    constexpr int SSHIFT = StatusWord<S>::SHIFT;
    constexpr S SMASK = StatusWord<S>::MASK;
    constexpr int LB_WINDOW = RDST_LB_WINDOW;
    int64_t prev = (int64_t)t - 1;  // next row whose word is still to be consumed
    uint32_t spins = 0;
    bool done = false;
    // `near_rows` is the copy of the rows kept in the writer's L2 (plain stores): a walker on the
    // writer's XCD — the normal case, a chain stays with one XCD — is served from that L2.  A word
    // seen EMPTY there is either not written yet or was written on another XCD (whose L2 this
    // one never sees): from then on the walk reads the write-through copy, which every XCD sees.
    const S* rows = near_rows;
#ifdef RDST_EXPERIMENTS
    const uint64_t lb_t0 = __builtin_amdgcn_s_memtime();
    uint32_t lb_iters = 0;
#endif
    while (!done) {
#ifdef RDST_EXPERIMENTS
        ++lb_iters;
#endif
        S v[LB_WINDOW];
#pragma unroll
        for (int k = 0; k < LB_WINDOW; ++k) {
            const int64_t idx = prev - k;
            // row 0 is always INCLUSIVE, so an index below 0 is never consumed
            v[k] = idx >= 0 ? ld_relaxed<S>(rows + (size_t)idx * RADIX + tid) : ((S)ST_INCL << SSHIFT);
        }
        bool blocked = false;
        int consumed = 0;
#pragma unroll
        for (int k = 0; k < LB_WINDOW; ++k) {
            const uint32_t st = (uint32_t)(v[k] >> SSHIFT);
            if (!done && !blocked) {
                if (st == ST_EMPTY) {
                    blocked = true;
                } else {
                    excl += (uint64_t)(v[k] & SMASK);
                    ++consumed;
                    done = st == ST_INCL;
                }
            }
        }
        prev -= consumed;
        if (blocked && !done) {
            if (rows != status) { rows = status; continue; }  // no wait before the first look at the far copy
            __builtin_amdgcn_s_sleep(2);
            ++spins;
            if (spins > SPIN_LIMIT || ((spins & 255u) == 0 && ld_relaxed<uint32_t>(err) != 0)) return false;
        }
    }
#ifdef RDST_EXPERIMENTS
    if (tid == 0 && g_exp_stats && level == 0) {  // digit 0's walker, one record per row, no atomics
        uint32_t* rec = g_exp_stats + (size_t)stat_row * 4;
        rec[0] = lb_iters;
        rec[1] = spins;
        rec[2] = (uint32_t)((int64_t)t - 1 - prev);
        rec[3] = (uint32_t)(__builtin_amdgcn_s_memtime() - lb_t0);
    }
#endif
    return true;
}

// registers are capped so that the LDS-limited number of blocks per CU (3 at 32 KiB of staging,
// more below) is not cut further by VGPRs: second launch-bound argument = waves per SIMD
constexpr int pass_lds_bytes(int nwaves, int delta_bytes, int stage_bytes) {  // wave tables, per-digit deltas, misc, staging
    return nwaves * 1024 + RADIX * delta_bytes + 80 + stage_bytes;
}
constexpr int blocks_per_cu(int nwaves, int delta_bytes, int stage_bytes) {
    const int lds = pass_lds_bytes(nwaves, delta_bytes, stage_bytes);
    int b = 160 * 1024 / lds;
    if (b * nwaves > 32) b = 32 / nwaves;
    if (b > 3) b = 3;  // beyond three the register budget (<= 64) costs more than it buys
    return b < 1 ? 1 : b;
}
struct NoVal {};  // keys only
template <typename V> struct ValBytes { static constexpr int value = (int)sizeof(V); };
template <> struct ValBytes<NoVal> { static constexpr int value = 0; };

// V: payload carried with every key (key-value sort: 4- or 8-byte values, whole tile staged), or NoVal
// OUT16 (4-byte keys, NARROW, the hybrid route's pass on level L-1 only): after this pass a key's position tells
// its top 16 bits (it lies in bucket b = [bstart[b], bstart[b+1])), so the pass stores only the low halves of the
// MAPPED keys, as a 16-bit array `out16` in the workspace, and K4 reads those: 2 bytes per key less written, 2 less
// read.  Decided at run time (the route is the device's choice): on the LSD route the same kernel stores whole keys.
// PERSIST: the grid is two blocks per CU and a block takes tile after tile until the chains are dry, instead of one block
// per tile.  Built for ONE use: the LSD fallback behind the atomic route (4-byte keys, shape 4) — on the good path those four
// launches have nothing to do, and returning from ~59 000 workgroups costs 0.041 ms each where 512 cost 0.003; on the
// fallback path the loop costs a few per cent of a pass (the compiler hoists lane-invariant addresses out of it and spills
// 14 dwords per lane; an opaque thread id per iteration keeps it to that).
template <typename K, typename S, int KPT, int NWAVES, int STAGES, bool MAPPED, bool NARROW, typename V = NoVal, bool OUT16 = false, bool PERSIST = false>
__global__ __launch_bounds__(NWAVES * 64, (blocks_per_cu(NWAVES, NARROW ? 4 : 8, NWAVES * 64 * KPT * ((int)sizeof(K) + ValBytes<V>::value) / STAGES) * NWAVES + 3) / 4) void onesweep_kernel(
    K* __restrict__ buf_keys, K* __restrict__ buf_tmp, V* __restrict__ buf_vals, V* __restrict__ buf_vtmp, uint16_t* __restrict__ out16, uint64_t n, int level,
    const uint64_t* __restrict__ cbase /* [CHAINS][256] of this level */, S* __restrict__ status /* [rows][256] of this level */,
    S* __restrict__ status_near /* same shape: the copy that stays in the writer's L2 */,
    const LevelChains* __restrict__ chains /* of this level */, uint32_t* __restrict__ ticket /* of this level: [CHAINS + 1][TICKET_STRIDE], per chain, then the mask of chains handed out */,
    const Plan* __restrict__ plan, uint32_t* __restrict__ err, K neg, K pos, uint32_t ablate) {
    // `ablate` is always 0 in the product build; tools/ builds with -DRDST_EXPERIMENTS can switch
    // stages off to price them (results are then wrong by design, stores stay in range).
#ifdef RDST_EXPERIMENTS
#define RDST_ABL(bit) (((ablate) >> (bit)) & 1u)
#else
#define RDST_ABL(bit) false
#endif
#ifdef RDST_EXPERIMENTS
    uint32_t tl[12];
    const bool tl_on = g_exp_timeline != nullptr && g_exp_timeline_kernel == 0 && level == 0 && threadIdx.x == 0;
    RDST_STAMP(0);
#endif
    constexpr int BLOCK = NWAVES * 64;
    constexpr int TILE = BLOCK * KPT;
    constexpr int STAGE_KEYS = TILE / STAGES;  // keys staged through LDS at a time
    constexpr int SPT = KPT / STAGES;          // slots each thread scatters per stage
    constexpr int SSHIFT = StatusWord<S>::SHIFT;
    constexpr S SMASK = StatusWord<S>::MASK;
    using D = typename std::conditional<NARROW, uint32_t, uint64_t>::type;  // destination offset (bytes if NARROW, else elements)
    static_assert(BLOCK >= RADIX, "need one thread per digit");
    static_assert(KPT % STAGES == 0 && (STAGES == 1 || TILE <= 65536), "stage split / 16-bit slot packing");
    constexpr bool HAS_V = ValBytes<V>::value != 0;
    constexpr bool FAR_X4 = RDST_FAR_X4 && sizeof(S) == 4;  // 64-bit status words (n >= 2^30) keep the per-word stores
    static_assert(!HAS_V || STAGES == 1, "payloads are staged with the whole tile");
    static_assert(!OUT16 || (NARROW && !HAS_V && sizeof(K) == 4), "16-bit hand-off: 4-byte keys, 32-bit offsets");
    // whole tile staged: the running slots of step 5 count in bytes of the staging buffer (one
    // shift-add per key to the LDS address); two stages: in keys (16-bit packing)
    constexpr uint32_t SLOT_UNIT = STAGES == 1 ? (uint32_t)sizeof(K) : 1u;

    if (plan->skip[level]) return;
    const bool halves = OUT16 && plan->route == ROUTE_HYBRID;  // block-uniform
    const bool from_tmp = plan->src_is_tmp[level] != 0;
    const K* __restrict__ src = from_tmp ? buf_tmp : buf_keys;
    K* __restrict__ dst = from_tmp ? buf_keys : buf_tmp;
    const V* __restrict__ vsrc = from_tmp ? buf_vtmp : buf_vals;
    V* __restrict__ vdst = from_tmp ? buf_vals : buf_vtmp;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* wave_hist = reinterpret_cast<uint32_t*>(smem);                          // [NWAVES][256]
    constexpr int DELTA_BYTES = RADIX * (int)sizeof(D);
    D* s_delta = reinterpret_cast<D*>(smem + NWAVES * 1024);                                   // [256]
    uint32_t* s_misc = reinterpret_cast<uint32_t*>(smem + NWAVES * 1024 + DELTA_BYTES);        // [16]
    uint64_t* s_begin = reinterpret_cast<uint64_t*>(smem + NWAVES * 1024 + DELTA_BYTES + 64);  // [2]
    K* s_keys = reinterpret_cast<K*>(smem + NWAVES * 1024 + DELTA_BYTES + 80);                 // [STAGE_KEYS]
    V* s_vals = reinterpret_cast<V*>(smem + NWAVES * 1024 + DELTA_BYTES + 80 + sizeof(K) * STAGE_KEYS);  // [STAGE_KEYS] (HAS_V)

    const int shift = level * 8;
    const int bit0 = shift & 31;

    // Block b takes a tile of chain b % CHAINS — blocks b and b + 8 share an XCD, so a chain's status
    // rows and its 256 scatter frontiers stay with one XCD's CUs — in ticket order, so a tile's
    // predecessors have always started; when that chain is handed out it tries the next ones
    // (segments of a skewed pass differ in length).  The grid has at least one block per tile.
    int tid_opaque = threadIdx.x;
#pragma unroll 1
    for (;;) {  // one tile (PERSIST: tile after tile; the ticket that finds every chain dry ends the block)
    if constexpr (PERSIST) asm volatile("" : "+v"(tid_opaque));  // nothing derived from it is loop-invariant for the compiler
    const int tid = tid_opaque, lane = tid & 63, wave = tid >> 6;
    // Wave priority: the phases that issue memory traffic (the tile's loads, the look-back, the
    // scatter) go ahead of other waves' arithmetic (counting, ranking), so the memory pipeline is fed
    // while the vector ALU works through the ranking of the CU's other block (A/B: pass 1.84 -> 1.75 ms);
    // so do the short serial stretches of the first four waves (digit sums, tile scan, look-back).
    __builtin_amdgcn_s_setprio(RDST_PRIO_LOAD);
    if (tid == 0) {
        // The chain's table entries and the mask of chains already handed out are requested before
        // the ticket, so that only arithmetic follows the atomic's round trip.  A block that finds
        // a chain dry says so in the mask (word CHAINS of the ticket row): later blocks go straight
        // to a chain that still has tiles instead of paying an atomic per dry chain.
        uint32_t c = blockIdx.x % CHAINS, t = ~0u, valid = 0, row0 = 0;
        uint64_t begin = 0;
        const uint32_t dry = ld_relaxed<uint32_t>(ticket + CHAINS * TICKET_STRIDE);
#pragma unroll 1
        for (int tries = 0; tries < CHAINS; ++tries) {
            if (!((dry >> c) & 1u)) {
                const uint32_t nt = chains->ntiles[c];
                const uint64_t lo = chains->seg_lo[c], hi = chains->seg_lo[c + 1];
                row0 = chains->row0[c];
                s_misc[8] = chains->shared[c];
                const uint32_t k = atomicAdd(ticket + c * TICKET_STRIDE, 1u);
                if (k < nt) {
                    t = k;
                    // every tile but a chain's first starts on a multiple of TILE_ALIGN keys
                    const uint64_t a0 = lo & ~(uint64_t)(TILE_ALIGN - 1);
                    begin = t == 0 ? lo : a0 + (uint64_t)t * TILE;
                    const uint64_t end = a0 + (uint64_t)(t + 1) * TILE < hi ? a0 + (uint64_t)(t + 1) * TILE : hi;
                    valid = (uint32_t)(end - begin);
                    break;
                }
                atomicOr(ticket + CHAINS * TICKET_STRIDE, 1u << c);
            }
            c = (c + 1) % CHAINS;
        }
        s_misc[0] = t;
        s_misc[1] = 0;  // block-wide failure flag
        s_misc[2] = c;
        s_misc[3] = row0;
        s_begin[0] = begin;
        s_begin[1] = valid;
    }
    __syncthreads();
    RDST_STAMP(1);
    const uint32_t t = s_misc[0];  // tile index inside its chain
    if (t == ~0u) return;          // every chain is handed out
    const uint32_t chain = s_misc[2];
    const uint32_t valid = (uint32_t)s_begin[1];
    const uint64_t tile_begin = s_begin[0];
    const bool full = valid == (uint32_t)TILE;
    const uint32_t chain_row0 = s_misc[3];
    S* const cstatus = status + (size_t)chain_row0 * RADIX;  // the chain's rows
    S* const cstatus_near = (s_misc[8] ? status : status_near) + (size_t)chain_row0 * RADIX;  // what walkers read first
    S* row = cstatus + (size_t)t * RADIX;
    S* row_near = status_near + ((size_t)chain_row0 + t) * RADIX;
    const uint64_t* __restrict__ base = cbase + (size_t)chain * RADIX;

    // 1. load, wave-striped: lane l of wave w takes keys w*64*KPT + i*64 + l (256 contiguous
    //    bytes per wave-instruction for 4-byte keys), so index order == (wave, i, lane) order.
    K mk[KPT];
    V mv[HAS_V ? KPT : 1];
    {
        const K* tsrc = src + tile_begin;
        const uint32_t wbase = (uint32_t)wave * 64u * KPT + (uint32_t)lane;
        if constexpr (HAS_V) {
            const V* tv = vsrc + tile_begin;
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t idx = wbase + i * 64;
                if (full || idx < valid) mv[i] = tv[idx];
            }
        }
        if (full) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) mk[i] = tsrc[wbase + i * 64];
            if constexpr (MAPPED) {
#pragma unroll
                for (int i = 0; i < KPT; ++i) mk[i] = map_key<K>(mk[i], neg, pos);
            }
        } else {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t idx = wbase + i * 64;
                // out-of-range slots: mapped key all ones -> digit 255, ranked after every
                // real key of the tile, never stored
                K v = (K)~(K)0;
                if (idx < valid) {
                    v = tsrc[idx];
                    if constexpr (MAPPED) v = map_key<K>(v, neg, pos);
                }
                mk[i] = v;
            }
        }
    }

    RDST_STAMP(2);
    __builtin_amdgcn_s_setprio(0);
    // 2. per-wave 256-bin histogram in LDS ("early counts")
    uint32_t* wh = wave_hist + wave * RADIX;
#pragma unroll
    for (int j = 0; j < 4; ++j) wh[lane + 64 * j] = 0;
    // Sorted or low-entropy input puts many lanes of a round on one bin, which would serialise
    // the LDS atomics up to 64 ways.  A wave whose first round looks like that — one digit in all
    // lanes, or many lanes holding their neighbour's digit — takes the careful path: a round with a
    // single digit is counted (and later ranked) by one lane; in any other round the lanes of a
    // digit are matched by ballots and its first lane adds the group's size.  Random keys: never.
    uint32_t uniform_rounds = 0;  // bit i: round i holds a single digit (wave-uniform value)
    bool careful = false;         // wave-uniform
    constexpr bool CAN_FAST = STAGES == 1 && !HAS_V && TILE <= 65536;
    uint32_t run_index[CAN_FAST ? (KPT + 1) / 2 : 1];  // fast ranking: index in the (wave, digit) run, 16-bit pairs
    bool fast = false;                                // wave-uniform
    {
        const uint32_t d0 = digit_of(mk[0], shift);
        const uint32_t dn = (uint32_t)__builtin_amdgcn_mov_dpp((int)d0, 0x138, 0xf, 0xf, false);  // lane - 1's digit
        careful = __builtin_popcountll(__builtin_amdgcn_ballot_w64(d0 == dn) & ~1ull) >= 8;
        // ONE heavy digit (a float column's sign-and-exponent byte: three keys in four; the bimodal input's zero bytes) does
        // not need the ballots: the lanes that hold it are one ballot away from their ranks (lane order), one of them adds the
        // group's size, and the other lanes — few, spread over the other digits — use the returning add of the fast form.
        // Chosen when a quarter of the first round holds the digit of one of three probed lanes and the rest of the round
        // does not look repetitive itself; everything else that looks repetitive stays on the careful path.
        bool heavy = false;   // wave-uniform
        uint32_t hd = 0;      // the heavy digit
        // (Not in the persistent form of the kernel: the loop already spills there, and this path tripled it — every pass of the
        // fallback behind the atomic route got 0.1-0.2 ms slower for the one level of a float column it makes 0.3-0.8 ms faster.)
        if constexpr (CAN_FAST && !PERSIST) {
            if (careful && full && (ablate & RDST_FAST_RANK) && !RDST_ABL(3)) {
                uint32_t best = 0;
#pragma unroll
                for (int probe = 0; probe < 3; ++probe) {
                    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)d0, probe * 21);
                    const uint32_t k = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(d0 == c));
                    if (k > best) { best = k; hd = c; }
                }
                const uint64_t rest_rep = __builtin_amdgcn_ballot_w64(d0 == dn && d0 != hd) & ~1ull;
                if (best >= 16 && __builtin_popcountll(rest_rep) < 8) { heavy = true; careful = false; }
            }
        }
        if constexpr (CAN_FAST) fast = !careful && full && (ablate & RDST_FAST_RANK) && !RDST_ABL(3);
        if (careful) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t d = digit_of(mk[i], shift);
                if (__all((int)(d == (uint32_t)__builtin_amdgcn_readfirstlane((int)d))) != 0) {
                    if (lane == 0) wh[d] += 64u;
                    uniform_rounds |= 1u << i;
                } else {
                    uint32_t total;
                    const uint32_t below = peers_below_total(digit_word<K>(mk[i], shift), bit0, total);
                    if (below == 0) atomicAdd(&wh[d], total);
                }
            }
        } else if (fast && heavy) {
            if constexpr (CAN_FAST && !PERSIST) {
                // (the table is this wave's own: the heavy digit's lanes are ranked from a running count in a scalar register —
                // no LDS round trip — and the count is stored once at the end; the other lanes' digits are other words)
                uint32_t run_h = 0;
#pragma unroll
                for (int i = 0; i < KPT; ++i) {
                    const uint32_t d = digit_of(mk[i], shift);
                    const uint64_t m = __builtin_amdgcn_ballot_w64(d == hd);
                    uint32_t r;
                    if (d == hd) r = run_h + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    else r = atomicAdd(&wh[d], 1u);
                    run_h += (uint32_t)__builtin_popcountll(m);
                    if (i & 1) run_index[i >> 1] |= r << 16;
                    else run_index[i >> 1] = r;
                }
                if (lane == 0) wh[hd] = run_h;
            }
        } else if (fast) {
            // the count doubles as the ranking: the returning add hands every key its index inside the
            // wave's run of its digit (step 5)
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t r = atomicAdd(&wh[digit_of(mk[i], shift)], 1u);
                if (i & 1) run_index[i >> 1] |= r << 16;
                else run_index[i >> 1] = r;
            }
        } else {
#pragma unroll
            for (int i = 0; i < KPT; ++i) atomicAdd(&wh[digit_of(mk[i], shift)], 1u);
        }
    }
    RDST_STAMP(3);
    __syncthreads();
    RDST_STAMP(4);

    if (tid < RADIX) __builtin_amdgcn_s_setprio(RDST_PRIO_SCAN);  // the four waves every other wave of the block waits for
    // 3. thread d: digit count over the block's waves; publish the tile aggregate at once
    uint32_t cw[NWAVES];
    uint32_t count_d = 0, pub = 0;
    if (tid < RADIX) {
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) {
            cw[w] = wave_hist[w * RADIX + tid];
            count_d += cw[w];
        }
        pub = count_d;
        if (tid == RADIX - 1) pub -= (uint32_t)TILE - valid;  // sentinels are not keys
        const S word = ((S)(t == 0 ? ST_INCL : ST_AGG) << SSHIFT) | (S)pub;
        st_near<S>(&row_near[tid], word);
        // the write-through copy: 32-bit words go out four at a time from one wave after the next
        // barrier (the staging buffer is still free and lends 1 KiB for the exchange)
        if constexpr (FAR_X4) reinterpret_cast<S*>(s_keys)[tid] = word;
        else st_relaxed<S>(&row[tid], word);
    }

    // 4. exclusive scan of the 256 digit counts -> start of each digit's run inside the tile
    uint32_t incl = count_d;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
    }
    if (tid < RADIX && lane == 63) s_misc[4 + wave] = incl;
    __syncthreads();
    if constexpr (FAR_X4) {
        if (wave == 0) st_far_x4(reinterpret_cast<uint32_t*>(row) + 4 * lane, reinterpret_cast<const u32x4*>(s_keys)[lane]);
    }
    uint32_t local_off = 0;
    if (tid < RADIX) {
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += s_misc[4 + w];
        local_off = woff + incl - count_d;
        uint32_t run = local_off;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) {
            wave_hist[w * RADIX + tid] = run * SLOT_UNIT;  // first slot of (wave w, digit d) inside the tile
            run += cw[w];
        }
    }
    __syncthreads();
    RDST_STAMP(5);

    __builtin_amdgcn_s_setprio(0);
    // 5. stable rank inside the wave: lanes with my digit and a lower lane id go first; rounds
    //    go in order.  The running slot of (wave, digit) lives in LDS: every lane reads it, then
    //    every lane adds one (LDS operations of one wave execute in program order), so after the
    //    round it has advanced by the size of the digit's group.
    //    STAGES == 1: the key goes straight to its slot of the LDS staging buffer (step 6).
    //    STAGES  > 1: the tile is larger than the staging buffer; slots are kept (16-bit pairs).
    uint32_t slots[STAGES > 1 ? KPT / 2 : 1];
    auto place = [&](int i, uint32_t sl) {  // key i of this lane goes to slot sl (in SLOT_UNITs) of the tile
        if constexpr (STAGES == 1) {
            *reinterpret_cast<K*>(reinterpret_cast<unsigned char*>(s_keys) + sl) = mk[i];
            if constexpr (HAS_V) {
                const uint32_t sv = sizeof(V) == sizeof(K) ? sl : (sizeof(V) > sizeof(K) ? sl * (uint32_t)(sizeof(V) / sizeof(K)) : sl / (uint32_t)(sizeof(K) / sizeof(V)));
                *reinterpret_cast<V*>(reinterpret_cast<unsigned char*>(s_vals) + sv) = mv[i];
            }
        } else {
            // pin the slot here: its inputs are eight ballots (SGPR pairs); left alone the
            // compiler sinks the arithmetic to the first use, after the look-back, and keeps
            // every round's ballots alive (KPT * 8 SGPR pairs -> spills)
            asm volatile("" : "+v"(sl));
            if (i & 1) slots[i >> 1] |= sl << 16;
            else slots[i >> 1] = sl;
        }
    };
    // Separate loops: sharing one loop body lets the compiler hoist half of the bit tests above the
    // (wave-uniform) choice and pay for them twice on the plain path.
    //
    // Fast form (keys only, whole tile staged, full tiles): the counting pass of step 2 used a RETURNING
    // LDS add, which handed every key its index inside the wave's run of its digit; the slot is the
    // run's start plus that index, no second atomic.  The lanes of a digit's group received their
    // indices in whatever order the LDS served them — on gfx950 that is ascending lane order
    // (tools/probe/lds_order_probe.hip: 10^10 lane-rounds, no exception), i.e. the stable rank, but it
    // is not a documented property, so it is not trusted: the source of pass p is sorted by the p
    // digits below the current one, hence a run is in stable-equivalent order iff every key is >= the
    // key in the slot before it on those low bits (any order among keys equal there is as good: later
    // passes only look at higher digits).  Each lane compares against the slot before its own; if
    // any lane of the wave fails, the wave's share of the tile is ranked again with the ballots.
    // Pass 0 has no lower digits: any order will do.
    if constexpr (CAN_FAST) {
        if (fast) {
            // key bits the earlier passes of THIS sort have ordered: from the first executed level up to the
            // current digit (the hybrid route starts at level L-2; the bits below stay unordered until K4)
            const int lo = (int)plan->first_level * 8;
            const int low_bits = shift > lo ? shift - lo : 0;
            const K order_mask = low_bits ? (K)((((K)1 << low_bits) - 1) << lo) : (K)0;
            bool out_of_order = false;
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t idx = (i & 1) ? (run_index[i >> 1] >> 16) : (run_index[i >> 1] & 0xFFFFu);
                const uint32_t sl = wh[digit_of(mk[i], shift)] + idx * SLOT_UNIT;  // start of the (wave, digit) run + index
                place(i, sl);
                if (low_bits) {
                    const K prev = *reinterpret_cast<const K*>(reinterpret_cast<const unsigned char*>(s_keys) + sl - SLOT_UNIT);
                    out_of_order |= idx != 0 && (K)(prev & order_mask) > (K)(mk[i] & order_mask);
                }
            }
            // never seen to fail; the ballot loop below redoes the wave's tile share if it ever does
            if (__builtin_amdgcn_ballot_w64(out_of_order) == 0 && !(ablate & RDST_FAST_RANK_SELFTEST)) goto ranked;
        }
    }
    if (!careful) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            uint32_t* slot = &wh[digit_of(mk[i], shift)];
            const uint32_t b = *slot;
            const uint32_t below = RDST_ABL(3) ? 0u : peers_below(digit_word<K>(mk[i], shift), bit0);
            __builtin_amdgcn_wave_barrier();
            atomicAdd(slot, SLOT_UNIT);
            place(i, b + below * SLOT_UNIT);
        }
    } else {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            uint32_t* slot = &wh[digit_of(mk[i], shift)];
            const uint32_t b = *slot;
            uint32_t below;
            if ((uniform_rounds >> i) & 1u) {  // one digit in all 64 lanes
                below = (uint32_t)lane;
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) *slot = b + 64u * SLOT_UNIT;
            } else {                           // heavy digits: the group's first lane advances the slot
                uint32_t total;
                below = peers_below_total(digit_word<K>(mk[i], shift), bit0, total);
                __builtin_amdgcn_wave_barrier();
                if (below == 0) *slot = b + total * SLOT_UNIT;
            }
            place(i, b + below * SLOT_UNIT);
        }
    }

ranked:
    RDST_STAMP(6);
    // 7. decoupled look-back over the earlier tiles (thread d walks digit d)
    __builtin_amdgcn_s_setprio(RDST_PRIO_LB);
    if (tid < RADIX) {
        uint64_t excl = 0;
        bool fail = false;
        if (t > 0 && !RDST_ABL(0)) {
            fail = !lookback_walk<S>(cstatus, cstatus_near, t, tid, err, excl
#ifdef RDST_EXPERIMENTS
                                     , level, chain_row0 + t
#endif
            );
            if (!fail) {
                const S word = ((S)ST_INCL << SSHIFT) | ((S)(excl + pub) & SMASK);
                st_near<S>(&row_near[tid], word);
                // exchange through the walker's OWN wave table (its ranking is over; a slower wave may still be
                // reading its own)
                if constexpr (FAR_X4) reinterpret_cast<S*>(wave_hist)[wave * RADIX + lane] = word;
                else st_relaxed<S>(&row[tid], word);
            }
        }
        if (fail) {
            atomicOr(err, ERR_LOOKBACK_TIMEOUT);
            s_misc[1] = 1;
        }
        const uint64_t first = base[tid] + excl - (uint64_t)local_off;  // dst index of tile slot 0 of digit d (mod 2^64)
        if constexpr (NARROW) s_delta[tid] = (uint32_t)first * (uint32_t)sizeof(K);
        else s_delta[tid] = first;
    }
    RDST_STAMP(7);
    __syncthreads();
    RDST_STAMP(8);
    if (s_misc[1]) return;  // never store with an unknown prefix
    __builtin_amdgcn_s_setprio(RDST_PRIO_SCATTER);
    if constexpr (FAR_X4) {
        if (wave == 0 && t > 0)  // words 4l..4l+3 sit in the table of wave l / 16
            st_far_x4(reinterpret_cast<uint32_t*>(row) + 4 * lane, *reinterpret_cast<const u32x4*>(wave_hist + (lane >> 4) * RADIX + ((4 * lane) & 63)));
    }

    // 6 + 8. per stage: (STAGES > 1) keys whose slot falls into the stage go to the LDS buffer;
    //    then consecutive threads take consecutive slots of a digit's run and store them.  LDS
    //    reads are batched (keys, then per-digit destinations) so their latencies overlap.
    bool bad = false;
#pragma unroll
    for (int stage = 0; stage < STAGES; ++stage) {
        if constexpr (STAGES > 1) {
            if (stage > 0) __syncthreads();  // previous stage's slots have all been read
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t sl = (i & 1) ? (slots[i >> 1] >> 16) : (slots[i >> 1] & 0xFFFFu);
                const uint32_t rel = sl - (uint32_t)(stage * STAGE_KEYS);
                if (rel < (uint32_t)STAGE_KEYS) s_keys[rel] = mk[i];
            }
            __syncthreads();
        }
        // slots of this stage in sub-batches (fewer live registers than one batch of SPT)
        constexpr int SUB = SPT % 6 == 0 ? 6 : (SPT % 4 == 0 ? 4 : (SPT < 6 ? SPT : 6));  // the last batch may be shorter
#pragma unroll
        for (int i0 = 0; i0 < SPT; i0 += SUB) {
            K kk[SUB];
            D dd[SUB];
            V vv[HAS_V ? SUB : 1];
#pragma unroll
            for (int i = 0; i < SUB; ++i)
                if (i0 + i < SPT) kk[i] = s_keys[tid + (i0 + i) * BLOCK];
            if constexpr (HAS_V) {
#pragma unroll
                for (int i = 0; i < SUB; ++i)
                    if (i0 + i < SPT) vv[i] = s_vals[tid + (i0 + i) * BLOCK];
            }
#pragma unroll
            for (int i = 0; i < SUB; ++i)
                if (i0 + i < SPT) dd[i] = s_delta[digit_of(kk[i], shift)];
#pragma unroll
            for (int i = 0; i < SUB; ++i) {
                if (i0 + i >= SPT) continue;
                const uint32_t p = (uint32_t)(stage * STAGE_KEYS) + (uint32_t)tid + (uint32_t)(i0 + i) * BLOCK;  // slot in the tile
                K out = kk[i];
                if constexpr (MAPPED) out = unmap_key<K>(out, neg, pos);
                if (RDST_ABL(1) && out != (K)0x12345) continue;  // no stores
                if (RDST_ABL(2)) {                                // sequential stores instead of the scatter
                    if (full || p < valid) dst[tile_begin + p] = out;
                    continue;
                }
                if constexpr (NARROW) {
                    const uint32_t g = dd[i] + p * (uint32_t)sizeof(K);  // byte offset, < 2^32
                    const bool ok = g < (uint32_t)n * (uint32_t)sizeof(K);
                    bad |= !ok && (full || p < valid);
                    if constexpr (OUT16) {
                        if (halves) {  // low half of the mapped key, at the same element index of the 16-bit array
                            if (ok && (full || p < valid)) *reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(out16) + (g >> 1)) = (uint16_t)kk[i];
                            continue;
                        }
                    }
                    if (ok && (full || p < valid)) {
                        *reinterpret_cast<K*>(reinterpret_cast<unsigned char*>(dst) + g) = out;
                        if constexpr (HAS_V) {  // same element index, in the value array's stride
                            const uint32_t gv = sizeof(V) == sizeof(K) ? g : (sizeof(V) > sizeof(K) ? g * (uint32_t)(sizeof(V) / sizeof(K)) : g / (uint32_t)(sizeof(K) / sizeof(V)));
                            *reinterpret_cast<V*>(reinterpret_cast<unsigned char*>(vdst) + gv) = vv[i];
                        }
                    }
                } else {
                    const uint64_t g = dd[i] + p;
                    const bool ok = g < n;
                    bad |= !ok && (full || p < valid);
                    if (ok && (full || p < valid)) {
                        dst[g] = out;
                        if constexpr (HAS_V) vdst[g] = vv[i];
                    }
                }
            }
        }
    }
    // cannot happen with consistent counts; the range test keeps a logic error from faulting
    if (bad) atomicOr(err, ERR_SCATTER_RANGE | ((uint32_t)(level + 1) << 8));  // bits 8..: which level(s) (diagnostics)
#ifdef RDST_EXPERIMENTS
    RDST_STAMP(9);
    if (tl_on) {
        uint32_t* rec = g_exp_timeline + (size_t)(chain_row0 + t) * 12;
        for (int k = 0; k < 10; ++k) rec[k] = tl[k];
        rec[10] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));  // XCC_ID
        rec[11] = blockIdx.x;
    }
#endif
    if constexpr (!PERSIST) break;
    __syncthreads();  // the next tile reuses the LDS
    }
#undef RDST_ABL
}

// ------------------------------------------------------------------------------------------
// MSD scatter with claimed space (ROUTE_ATOMIC).  The tile part is K3's — wave-striped load, per-wave LDS histograms,
// tile scan, ranking by returning LDS add (no order test: an MSD pass has nothing to preserve), whole tile staged, runs
// stored by consecutive threads — without tickets, chains, status rows or look-back: thread d claims the tile's run of
// digit d with `atomicAdd(&cursor[area of d], count)`.  Sources are `areas` of `area_cap` slots each, area a holding
// area_count[a] keys (pass A: ONE area, the input slice itself; pass B: the 2 048 areas pass A filled); tile j of area a
// is block a * tiles_per_area + j.  Destinations: pass A: area (digit, blockIdx % 8) of dst_cap keys; pass B: slot
// (top digit of the source area, digit) of dst_cap keys, low halves only (HALVES).  A claim that would pass the capacity
// raises *overflow and stores nothing.  The key map is applied for the digit and, for whole keys, undone at the store.
// ------------------------------------------------------------------------------------------
struct MsdRanges {
    uint64_t start[CHAINS + 1];  // K1h's position ranges: [start[r], start[r + 1])
};

// SECOND: pass B (sources are pass A's areas, destinations the bucket slots); HALVES: 4-byte keys' pass B stores low halves.
template <typename K, int KPT, int NWAVES, bool MAPPED, bool SECOND, bool HALVES>
#ifndef RDST_MSD_MINWAVES
#define RDST_MSD_MINWAVES ((2 * NWAVES + 3) / 4)  // two blocks per CU
#endif
__global__ __launch_bounds__(NWAVES * 64, RDST_MSD_MINWAVES) void msd_scatter_kernel(
    const K* __restrict__ src, const uint32_t* __restrict__ area_count /* nullable: one area of n keys */, uint64_t n, uint32_t area_cap,
    uint32_t tiles_per_area, K* __restrict__ dst, uint16_t* __restrict__ dst16, uint32_t* __restrict__ cursor, uint32_t dst_cap, int shift,
    uint32_t slices /* areas per top digit (pass A: of the destination, pass B: of the source) */,
    const Plan* __restrict__ plan, uint32_t* __restrict__ overflow, uint32_t* __restrict__ inversion /* pass A sets it; pass B reads it */,
    K neg, K pos,
    // the exact form (Plan::pre, hybrid route): pass A keys -> xbuf by top digit, pass B xbuf -> buckets (4-byte keys: halves in
    // dst16; 8-byte keys: whole keys in xout); the cursors start at the exact offsets (route_kernel), bstart / xtile0 give pass
    // B's regions and grid
    K* __restrict__ xbuf, K* __restrict__ xout, const uint32_t* __restrict__ bstart, const uint32_t* __restrict__ xtile0,
    // exact pass A: the slice is cut at K1h's eight position ranges (their per-digit counts are exact: hpos16's second table), each
    // range is tiled on its own and claims from its own 256 counters — block b works on range b % 8, so a range's tiles (and
    // its frontiers) stay with one XCD, and a heavy digit's claims queue eight times shorter than on one counter
    MsdRanges xr) {
    static_assert(!HALVES || (SECOND && sizeof(K) == 4), "halves: the second pass of 4-byte keys");
    constexpr int BLOCK = NWAVES * 64, TILE = BLOCK * KPT;
    constexpr uint32_t SLOT_UNIT = (uint32_t)sizeof(K);
    const bool exact = plan->pre != 0;  // block-uniform
    if (exact) {
        if (plan->route != ROUTE_HYBRID || plan->sorted_known) return;
        if (!SECOND && plan->pre_skip_a) return;
    } else {
        // the sample already gave the route up (plan words and pass A's inversion flag are written by EARLIER launches: block-uniform)
        if (plan->gross_skew || plan->top_skew || plan->predict_lsd) return;
        if (SECOND && *inversion == 0) return;  // pass A met no inversion: the slice is sorted, nothing to do
    }
    const int win = exact ? 0 : (int)plan->win_shift;  // the buckets' 16 bits start this far below the key's top (presample_kernel)
    shift -= win;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* wave_hist = reinterpret_cast<uint32_t*>(smem);                           // [NWAVES][256]
    uint32_t* s_delta = reinterpret_cast<uint32_t*>(smem + NWAVES * 1024);             // [256] destination of tile slot 0 of a digit's run (elements, mod 2^32)
    uint32_t* s_misc = reinterpret_cast<uint32_t*>(smem + NWAVES * 1024 + 1024);       // [32]; [16 + w]: what wave w saw in the overflow flag
    K* s_keys = reinterpret_cast<K*>(smem + NWAVES * 1024 + 1024 + 128);               // [TILE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ... or an earlier tile did.  The flag is raised by blocks on other XCDs WHILE this kernel runs (hence a coherent load), so two
    // waves of one block can see different values — and the exit must be the BLOCK's: a wave that left alone would leave its
    // stale wave_hist table to be summed into the claims of the waves that stayed (round 2 did that; with the knob
    // set_hybrid(..., min_len) lowered the garbage claims could store outside the workspace).  Block-uniform without a barrier of
    // its own (__syncthreads_or is a software reduction, two barriers and LDS traffic: +0.1 ms per pass when it sat here):
    // every wave leaves what it saw in its own word of s_misc before the barrier the kernel has anyway, behind the tile's loads
    // and counts, and every thread reads all the words behind it.  Pass B does not even wait for the flag before its loads (the
    // latency hides behind them: -2 %); a wave of pass A does, and one that finds the flag up goes straight to that barrier
    // without loading anything — behind a failed claim the rest of the pass costs microseconds, not a read of the slice.
    static_assert(NWAVES <= 16, "one word of s_misc per wave");
    uint32_t gave_up = 0;
    if (!exact) {
        gave_up = ld_relaxed<uint32_t>(overflow);
        if constexpr (!SECOND) {
            if (gave_up) {
                if (lane == 0) s_misc[16 + wave] = 1u;
                __syncthreads();
                return;
            }
        }
    }
    uint32_t area = blockIdx.x / tiles_per_area, j = blockIdx.x % tiles_per_area;
    if constexpr (SECOND) {
        // pass B of the atomic route: a slot (top digit d, second digit) is appended to by the tiles of d's eight areas.  Blocks b
        // and b + 8 share an XCD: hand top digit d to XCD d mod 8, so that a slot's frontier line is completed in ONE L2 instead of
        // going out to HBM in pieces from eight
        if (!exact && slices == (uint32_t)CHAINS) {
            const uint32_t x = blockIdx.x % CHAINS, k = blockIdx.x / CHAINS;
            const uint32_t q = k / tiles_per_area;           // (digit of this XCD, slice)
            j = k % tiles_per_area;
            area = (q % CHAINS) * RADIX + (x + CHAINS * (q / CHAINS));
        }
        // (the grid also covers the exact form's tiles: blocks past the last area have nothing to do here)
        if (!exact && (blockIdx.x / tiles_per_area >= RADIX * slices || area >= RADIX * slices)) return;
    }
    uint64_t acount = area_count ? (uint64_t)area_count[area] : n;
    const K* asrc = src + (uint64_t)area * area_cap;
    if (exact) {
        if constexpr (!SECOND) {
            area = blockIdx.x % CHAINS; j = blockIdx.x / CHAINS;
            acount = xr.start[area + 1] - xr.start[area];
            asrc = src + xr.start[area];
        } else {  // the region of top digit d in xbuf: tiles xtile0[d] .. xtile0[d + 1]
            if (blockIdx.x >= xtile0[RADIX]) return;
            uint32_t d = 0;
#pragma unroll
            for (int b = 7; b >= 0; --b) {
                const uint32_t c = d | (1u << b);
                if (xtile0[c] <= blockIdx.x) d = c;
            }
            area = d; j = blockIdx.x - xtile0[d];
            const uint32_t r0 = bstart[d * RADIX];
            acount = bstart[(d + 1) * RADIX] - r0;
            asrc = (plan->pre_skip_a ? static_cast<const K*>(xout) : static_cast<const K*>(xbuf)) + r0;  // (one top byte: pass A was skipped, the region is the slice itself)
        }
    }
    const uint64_t tile_off = (uint64_t)j * TILE;
    if (tile_off >= acount) return;
    const uint32_t valid = acount - tile_off < (uint64_t)TILE ? (uint32_t)(acount - tile_off) : (uint32_t)TILE;
    const bool full = valid == (uint32_t)TILE;
    const K* tsrc = asrc + tile_off;
    RDST_TL_BEGIN(SECOND ? 3u : 2u);
    __builtin_amdgcn_s_setprio(RDST_PRIO_LOAD);
    K mk[KPT];
    const uint32_t wbase = (uint32_t)wave * 64u * KPT + (uint32_t)lane;
    // pass A reads the caller's slice in its order: the already-sorted exit (src/sorter.rs:59-65) looks here.  Index order is
    // (wave, round, lane): the key before a wave's first is fetched with the tile (one lane), the others are in registers
    K edge = 0;
    if constexpr (!SECOND) {
        if (lane == 0 && tile_off + (uint64_t)wave * 64u * KPT > 0 && (uint32_t)wave * 64u * KPT < valid) edge = tsrc[(int64_t)wbase - 1];
    }
    if (full) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) mk[i] = tsrc[wbase + i * 64];
        if constexpr (MAPPED) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) mk[i] = map_key<K>(mk[i], neg, pos);
        }
    } else {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t idx = wbase + i * 64;
            K v = (K)~(K)0;  // digit 255 and the largest key: ranked (stably) after the real keys of the tile, never stored
            if (idx < valid) {
                v = tsrc[idx];
                if constexpr (MAPPED) v = map_key<K>(v, neg, pos);
            }
            mk[i] = v;
        }
    }
    if constexpr (!SECOND) if (!exact) {  // (exact form: K1h has looked for inversions already)
        if constexpr (MAPPED) edge = map_key<K>(edge, neg, pos);
        if (tile_off + (uint64_t)wave * 64u * KPT == 0) edge = 0;  // the slice's first key has no predecessor
        bool inv = false;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            K before = lane_below<K>(mk[i]);
            if (lane == 0) before = i == 0 ? edge : lane63_of<K>(mk[i > 0 ? i - 1 : 0]);
            inv |= before > mk[i];  // (padding is the largest key and sits at the end: it never counts as an inversion)
        }
        // (one word for the whole grid: look before setting — 700 000 waves OR-ing the same word took 8 ms, ~88 atomics per us)
        if (__builtin_amdgcn_ballot_w64(inv) != 0 && lane == 0 && ld_relaxed<uint32_t>(inversion) == 0) atomicOr(inversion, 1u);
        if (win) {  // every key must carry the top bits the sample's keys share: one that does not gives the route up
            constexpr int KW = (int)sizeof(K) * 8;
            const K want = (K)plan->win_top;
            bool stray = false;
#pragma unroll
            for (int i = 0; i < KPT; ++i) stray |= (full || wbase + i * 64 < valid) && (K)(mk[i] >> (KW - win)) != want;
            if (__builtin_amdgcn_ballot_w64(stray) != 0 && lane == 0 && ld_relaxed<uint32_t>(overflow) == 0) atomicOr(overflow, 1u);
        }
    }
    RDST_STAMP(1);
    __builtin_amdgcn_s_setprio(0);
    uint32_t* wh = wave_hist + wave * RADIX;
#pragma unroll
    for (int q = 0; q < 4; ++q) wh[lane + 64 * q] = 0;
    const int bit0 = shift & 31;
    uint32_t uniform_rounds = 0;
    uint32_t run_index[(KPT + 1) / 2];
    bool careful, fast, heavy = false;
    constexpr int NHEAVY = 5;
    uint32_t hd3[NHEAVY] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};  // wave-uniform
    {
        const uint32_t d0 = digit_of(mk[0], shift);
        const uint32_t dn = (uint32_t)__builtin_amdgcn_mov_dpp((int)d0, 0x138, 0xf, 0xf, false);
        careful = __builtin_popcountll(__builtin_amdgcn_ballot_w64(d0 == dn) & ~1ull) >= 8;
        // up to five heavy digits (a float column's sign-and-exponent byte: uniform [0, 1) puts 50 % / 37 % / 9 % of the keys on
        // three values, normal(0, 1) 24 / 24 / 19 / 19 % on four; K3's step 3 has the one-digit form): their lanes rank by one
        // ballot each, the others by the returning add.  The candidates are the digits of five probed lanes; one that less
        // than an eighth of the round holds, or that an earlier probe found, is dropped.  (Three candidates, round 2, left a
        // normal column's fourth digit to the test below half of the time: one wave in four took the bit-by-bit path — 37
        // instructions per key — and the other eleven of its block waited at two barriers: 16.8 us per tile against 11.3.)
        if (careful && full) {
            bool any = false;
#pragma unroll
            for (int probe = 0; probe < NHEAVY; ++probe) {
                const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)d0, probe * 13);
                const uint32_t k = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(d0 == c));
                bool dup = false;
#pragma unroll
                for (int e = 0; e < probe; ++e) dup |= hd3[e] == c;
                hd3[probe] = (k >= 8 && !dup) ? c : 0xFFFFFFFFu;  // (no digit is 0xFFFFFFFF: a dropped candidate matches nothing)
                any |= hd3[probe] != 0xFFFFFFFFu;
            }
            bool listed = false;
#pragma unroll
            for (int e = 0; e < NHEAVY; ++e) listed |= d0 == hd3[e];
            const uint64_t rest_rep = __builtin_amdgcn_ballot_w64(d0 == dn && !listed) & ~1ull;
            if (any && __builtin_popcountll(rest_rep) < 8) { heavy = true; careful = false; }
        }
        fast = !careful && full;  // any order inside a run will do (no order test) — but a partial tile's padding must stay BEHIND
                                  // the real keys of digit 255, which only the stable forms below guarantee
        if (careful) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t d = digit_of(mk[i], shift);
                if (__all((int)(d == (uint32_t)__builtin_amdgcn_readfirstlane((int)d))) != 0) {
                    if (lane == 0) wh[d] += 64u;
                    uniform_rounds |= 1u << i;
                } else {
                    uint32_t total;
                    const uint32_t below = peers_below_total(digit_word<K>(mk[i], shift), bit0, total);
                    if (below == 0) atomicAdd(&wh[d], total);
                }
            }
        } else if (fast && heavy) {
            // (the table is this wave's own: a heavy digit's lanes are ranked from a running count in a scalar register — no LDS
            // round trip between the rounds — and the count is stored once at the end; the other lanes' digits are other words)
            uint32_t run3[NHEAVY] = {0, 0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t d = digit_of(mk[i], shift);
                uint32_t r = 0;
                bool ranked = false;
#pragma unroll
                for (int h = 0; h < NHEAVY; ++h) {
                    const uint64_t m = __builtin_amdgcn_ballot_w64(d == hd3[h]);
                    if (d == hd3[h]) {
                        r = run3[h] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        ranked = true;
                    }
                    run3[h] += (uint32_t)__builtin_popcountll(m);
                }
                if (!ranked) r = atomicAdd(&wh[d], 1u);
                if (i & 1) run_index[i >> 1] |= r << 16;
                else run_index[i >> 1] = r;
            }
#pragma unroll
            for (int h = 0; h < NHEAVY; ++h)
                if (lane == 0 && hd3[h] != 0xFFFFFFFFu) wh[hd3[h]] = run3[h];
        } else if (fast) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t r = atomicAdd(&wh[digit_of(mk[i], shift)], 1u);
                if (i & 1) run_index[i >> 1] |= r << 16;
                else run_index[i >> 1] = r;
            }
        } else {
#pragma unroll
            for (int i = 0; i < KPT; ++i) atomicAdd(&wh[digit_of(mk[i], shift)], 1u);
        }
    }
    RDST_STAMP(2);
    if (lane == 0) s_misc[16 + wave] = gave_up;
    __syncthreads();
    {
        uint32_t any = 0;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) any |= s_misc[16 + w];
        if (any) return;  // block-uniform (see the top of the kernel)
    }
    RDST_STAMP(3);
    if (tid < RADIX) __builtin_amdgcn_s_setprio(RDST_PRIO_SCAN);
    uint32_t cw[NWAVES];
    uint32_t count_d = 0, pub = 0;
    if (tid < RADIX) {
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) {
            cw[w] = wave_hist[w * RADIX + tid];
            count_d += cw[w];
        }
        pub = count_d;
        if (tid == RADIX - 1) pub -= (uint32_t)TILE - valid;  // the padding of a partial tile is not keys
    }
    uint32_t incl = count_d;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
    }
    if (tid < RADIX && lane == 63) s_misc[4 + wave] = incl;
    if (tid == 0) s_misc[1] = 0;
    __syncthreads();
    if (tid < RADIX) {
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += s_misc[4 + w];
        const uint32_t local_off = woff + incl - count_d;
        uint32_t run = local_off;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) {
            wave_hist[w * RADIX + tid] = run * SLOT_UNIT;
            run += cw[w];
        }
        // claim the run's space: one returning atomic per digit and tile
        // pass A: area (slice, digit) — blocks b and b + 8 share an XCD, so a digit's 8 frontiers stay with one L2 each, and the
        // tiles are dealt to the slices in turn: every slice gets its share of the keys give or take a tile;
        // pass B: slot (top digit of the source area, digit)
        const uint32_t where = SECOND ? (area % RADIX) * RADIX + (uint32_t)tid : (exact ? area : blockIdx.x % slices) * RADIX + (uint32_t)tid;
        uint32_t got = 0;
        if (pub) {
            // (Pass A has only 256 x 8 counters for ~59 000 tiles x 256 claims.  Measured: device-scope claims cost the pass nothing —
            // 1.555 ms against 1.554 with XCD-local workgroup-scope claims on per-XCD counters.)
            got = atomicAdd(&cursor[where], pub);
            if (!exact && got + pub > dst_cap) {  // no room: give the route up, store nothing of this tile
                if (ld_relaxed<uint32_t>(overflow) == 0) atomicOr(overflow, 1u);  // (look first: thousands of tiles get here, see the inversion flag)
                s_misc[1] = 1;
            }
        }
        // (exact form: the counter started at the digit's / bucket's place in the destination — `got` is an element index)
        s_delta[tid] = (exact ? 0u : where * dst_cap) + got - local_off;  // (mod 2^32; the destination arrays hold fewer than 2^32 elements)
    }
    RDST_STAMP(4);
    __syncthreads();
    RDST_STAMP(5);
    __builtin_amdgcn_s_setprio(0);
    if (fast) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t idx = (i & 1) ? (run_index[i >> 1] >> 16) : (run_index[i >> 1] & 0xFFFFu);
            const uint32_t sl = wh[digit_of(mk[i], shift)] + idx * SLOT_UNIT;
            *reinterpret_cast<K*>(reinterpret_cast<unsigned char*>(s_keys) + sl) = mk[i];
        }
    } else if (!careful) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            uint32_t* slot = &wh[digit_of(mk[i], shift)];
            const uint32_t b = *slot;
            const uint32_t below = peers_below(digit_word<K>(mk[i], shift), bit0);
            __builtin_amdgcn_wave_barrier();
            atomicAdd(slot, SLOT_UNIT);
            *reinterpret_cast<K*>(reinterpret_cast<unsigned char*>(s_keys) + b + below * SLOT_UNIT) = mk[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            uint32_t* slot = &wh[digit_of(mk[i], shift)];
            const uint32_t b = *slot;
            uint32_t below;
            if ((uniform_rounds >> i) & 1u) {
                below = (uint32_t)lane;
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) *slot = b + 64u * SLOT_UNIT;
            } else {
                uint32_t total;
                below = peers_below_total(digit_word<K>(mk[i], shift), bit0, total);
                __builtin_amdgcn_wave_barrier();
                if (below == 0) *slot = b + total * SLOT_UNIT;
            }
            *reinterpret_cast<K*>(reinterpret_cast<unsigned char*>(s_keys) + b + below * SLOT_UNIT) = mk[i];
        }
    }
    RDST_STAMP(6);
    __syncthreads();
    RDST_STAMP(7);
    if (s_misc[1]) return;
    __builtin_amdgcn_s_setprio(RDST_PRIO_SCATTER);
    constexpr int SUB = KPT % 6 == 0 ? 6 : (KPT % 4 == 0 ? 4 : (KPT < 6 ? KPT : 6));
#pragma unroll
    for (int i0 = 0; i0 < KPT; i0 += SUB) {
        K kk[SUB];
        uint32_t dd[SUB];
#pragma unroll
        for (int i = 0; i < SUB; ++i)
            if (i0 + i < KPT) kk[i] = s_keys[tid + (i0 + i) * BLOCK];
#pragma unroll
        for (int i = 0; i < SUB; ++i)
            if (i0 + i < KPT) dd[i] = s_delta[digit_of(kk[i], shift)];
#pragma unroll
        for (int i = 0; i < SUB; ++i) {
            if (i0 + i >= KPT) continue;
            const uint32_t p = (uint32_t)tid + (uint32_t)(i0 + i) * BLOCK;  // slot in the tile; the padding sits at the end
            if (full || p < valid) {
                const uint32_t g = dd[i] + p;
                if constexpr (HALVES) dst16[g] = (uint16_t)kk[i];
                else (exact ? (SECOND ? xout : xbuf) : dst)[g] = MAPPED ? unmap_key<K>(kk[i], neg, pos) : kk[i];
            }
        }
    }
    RDST_STAMP(8);
    RDST_TL_END(blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// Small slices: the whole LSD sort in one workgroup.  Up to 64 KiB of keys live in registers
// (wave-striped like a K3 tile) and are re-ordered through LDS once per level with K3's counting,
// scan and ballot ranking; no workspace, no look-back, one launch (~5 us per level instead of the
// ~80 us the general pipeline costs below a tile of input).  Slots past n hold the largest mapped
// key: stable passes keep them behind every real key, and they are never stored.
// (src/sorts/lsb_sort.rs:39-127 is the reference's own small-chunk sorter.)
// ------------------------------------------------------------------------------------------
constexpr int SMALL_WAVES = 16;
constexpr int SMALL_THREADS = SMALL_WAVES * 64;
constexpr int small_kpt(size_t key_bytes) { return key_bytes <= 4 ? 16 : (key_bytes == 8 ? 8 : 4); }

template <typename K, int LEVELS, bool MAPPED>
__global__ __launch_bounds__(SMALL_THREADS) void small_sort_kernel(K* __restrict__ keys, uint32_t n, K neg, K pos) {
    constexpr int KPT = small_kpt(sizeof(K));
    constexpr int TILE = SMALL_THREADS * KPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* wave_hist = reinterpret_cast<uint32_t*>(smem);                        // [SMALL_WAVES][256]
    uint32_t* s_sum = reinterpret_cast<uint32_t*>(smem + SMALL_WAVES * 1024);       // [4]
    K* stage = reinterpret_cast<K*>(smem + SMALL_WAVES * 1024 + 16);                // [TILE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // only as many rounds as the slice needs (a short slice would otherwise be mostly padding, all of it
    // on one bin): key index = wave * 64 * rounds + round * 64 + lane
    const int rounds = (int)((n + SMALL_THREADS - 1) / SMALL_THREADS);
    const uint32_t live = (uint32_t)rounds * SMALL_THREADS;
    const uint32_t wbase = (uint32_t)wave * 64u * (uint32_t)rounds + (uint32_t)lane;
    K mk[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t idx = wbase + i * 64;
        K v = (K) ~(K)0;
        if (i < rounds && idx < n) {
            v = keys[idx];
            if constexpr (MAPPED) v = map_key<K>(v, neg, pos);
        }
        mk[i] = v;
    }
    uint32_t* wh = wave_hist + wave * RADIX;
    for (int level = 0; level < LEVELS; ++level) {
        const int shift = level * 8, bit0 = shift & 31;
#pragma unroll
        for (int j = 0; j < 4; ++j) wh[lane + 64 * j] = 0;
#pragma unroll
        for (int i = 0; i < KPT; ++i)
            if (i < rounds) atomicAdd(&wh[digit_of(mk[i], shift)], 1u);
        __syncthreads();
        uint32_t cw[SMALL_WAVES], count_d = 0;
        if (tid < RADIX) {
#pragma unroll
            for (int w = 0; w < SMALL_WAVES; ++w) { cw[w] = wave_hist[w * RADIX + tid]; count_d += cw[w]; }
        }
        const bool trivial = __syncthreads_or(tid < RADIX && count_d == live) != 0;  // one digit holds everything
        if (trivial) continue;  // block-uniform; the tables are re-zeroed at the top
        uint32_t incl = count_d;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (tid < RADIX && lane == 63) s_sum[wave] = incl;
        __syncthreads();
        if (tid < RADIX) {
            uint32_t run = incl - count_d;
            for (int w = 0; w < wave; ++w) run += s_sum[w];
#pragma unroll
            for (int w = 0; w < SMALL_WAVES; ++w) { wave_hist[w * RADIX + tid] = run; run += cw[w]; }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            if (i < rounds) {  // block-uniform
                uint32_t* slot = &wh[digit_of(mk[i], shift)];
                const uint32_t b = *slot;
                const uint32_t below = peers_below(digit_word<K>(mk[i], shift), bit0);
                __builtin_amdgcn_wave_barrier();
                atomicAdd(slot, 1u);
                stage[b + below] = mk[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; ++i)
            if (i < rounds) mk[i] = stage[wbase + i * 64];
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t idx = wbase + i * 64;
        if (i < rounds && idx < n) keys[idx] = MAPPED ? unmap_key<K>(mk[i], neg, pos) : mk[i];
    }
}

// ------------------------------------------------------------------------------------------
// K4: the hybrid route's last step.  After the two K3 passes the slice is ordered by the top 16
// bits of the mapped key; bucket b = [bstart[b], bstart[b+1]) holds the keys with prefix b and is
// at most one tile long.  One workgroup per bucket: the keys are read once (wave-striped, like a K3
// tile), LSD-sorted by the remaining LEVELS-2 digits through LDS — per level K3's counting, tile
// scan and ranking (returning LDS add + order test, ballots as fallback, group leaders for heavy
// digits) — and written back in place with coalesced stores.  This is the device twin of what rdst
// itself does below its two MSD levels at this size: Sorter::lsb_sort_adapter on ~15 k-key chunks
// (src/sorts/lsb_sort.rs:39-127, picked by src/tuners/standard_tuner.rs:46-48).
// Slots past the bucket's end hold the bucket's largest possible key (prefix | ones): a real key
// that ties with it on every remaining digit has the same bits, so whichever of them lands in the
// first `cnt` slots, the stored values are the same.
// ------------------------------------------------------------------------------------------
constexpr int local_waves(size_t key_bytes) { return key_bytes <= 4 ? 12 : 16; }
constexpr int local_kpt(size_t key_bytes) { return key_bytes <= 4 ? 22 : 16; }
constexpr int local_tile(size_t key_bytes) { return local_waves(key_bytes) * 64 * local_kpt(key_bytes); }
constexpr size_t local_lds_bytes(size_t key_bytes) { return (size_t)local_waves(key_bytes) * 1024 + 64 + key_bytes * local_tile(key_bytes); }

template <typename K, int NWAVES, int KPT, bool MAPPED>
__device__ __forceinline__ void local_sort_bucket(K* __restrict__ buf, const uint16_t* __restrict__ src16 /* nullable: low halves of the mapped keys */,
                                                  const K* __restrict__ alt_src /* nullable: whole (raw) keys lie here, not at their final place */,
                                                  const uint32_t soff /* where the bucket lies in src16 / alt_src */, const uint32_t bucket,
                                                  const uint32_t top16 /* top 16 bits of the bucket's (mapped) keys: bucket_prefix16 */, const uint32_t start, const uint32_t cnt,
                                                  uint32_t* __restrict__ err, K neg, K pos, uint32_t flags) {
    constexpr int BLOCK = NWAVES * 64, TILE = BLOCK * KPT, W = sizeof(K) * 8, LOCAL = (int)sizeof(K) - 2;
    constexpr uint32_t SLOT_UNIT = (uint32_t)sizeof(K);  // running slots count in bytes of the staging buffer
    static_assert(BLOCK >= RADIX && TILE <= 65536, "one thread per digit / 16-bit run indices");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (cnt <= 1) {
        if (sizeof(K) == 4 && src16 && cnt == 1 && tid == 0) {  // the key exists only as its low half
            const K m = (K)((K)top16 << (W - 16)) | (K)src16[soff];
            buf[start] = MAPPED ? unmap_key<K>(m, neg, pos) : m;
        }
        if (alt_src && cnt == 1 && tid == 0) buf[start] = alt_src[soff];
        return;
    }
    if (cnt > (uint32_t)TILE) {  // the route test rules it out; never sort a truncated bucket
        if (tid == 0) atomicOr(err, ERR_LOCAL_OVERFLOW);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* wave_hist = reinterpret_cast<uint32_t*>(smem);                       // [NWAVES][256]
    uint32_t* s_misc = reinterpret_cast<uint32_t*>(smem + NWAVES * 1024);          // [16]
    unsigned char* stage = smem + NWAVES * 1024 + 64;                              // K[TILE]
    // only as many rounds as the bucket needs: key index = wave * 64 * rounds + round * 64 + lane
    const int rounds = (int)((cnt + BLOCK - 1) / BLOCK);
    const uint32_t live = (uint32_t)rounds * BLOCK;
    const uint32_t wbase = (uint32_t)wave * 64u * (uint32_t)rounds + (uint32_t)lane;
    const K sentinel = (K)((K)top16 << (W - 16)) | (K)(((K)1 << (W - 16)) - 1);

    __builtin_amdgcn_s_setprio(RDST_PRIO_LOAD);
    K mk[KPT];
    {
        const K* tsrc = alt_src ? alt_src + soff : buf + start;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            K v = sentinel;
            if (i < rounds) {  // block-uniform
                const uint32_t idx = wbase + i * 64;
                const uint32_t at = idx < cnt ? idx : cnt - 1;
                if (sizeof(K) == 4 && src16) {  // block-uniform
                    if (idx < cnt) v = (K)((K)top16 << (W - 16)) | (K)src16[soff + at];
                } else {
                    const K raw = tsrc[at];
                    if (idx < cnt) v = MAPPED ? map_key<K>(raw, neg, pos) : raw;
                }
            }
            mk[i] = v;
        }
    }
    uint32_t* wh = wave_hist + wave * RADIX;
#pragma unroll 1
    for (int level = 0; level < LOCAL; ++level) {
        const int shift = level * 8, bit0 = shift & 31;
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) wh[lane + 64 * j] = 0;
        if (tid == 0) s_misc[0] = 0;  // "one digit holds every slot" flag of this level
        // counting, as in K3 step 2
        uint32_t uniform_rounds = 0;
        uint32_t run_index[(KPT + 1) / 2];
        bool careful, fast;
        {
            const uint32_t d0 = digit_of(mk[0], shift);
            const uint32_t dn = (uint32_t)__builtin_amdgcn_mov_dpp((int)d0, 0x138, 0xf, 0xf, false);
            careful = __builtin_popcountll(__builtin_amdgcn_ballot_w64(d0 == dn) & ~1ull) >= 8;
            fast = !careful && (flags & RDST_FAST_RANK);
            if (careful) {
#pragma unroll
                for (int i = 0; i < KPT; ++i) {
                    if (i < rounds) {
                        const uint32_t d = digit_of(mk[i], shift);
                        if (__all((int)(d == (uint32_t)__builtin_amdgcn_readfirstlane((int)d))) != 0) {
                            if (lane == 0) wh[d] += 64u;
                            uniform_rounds |= 1u << i;
                        } else {
                            uint32_t total;
                            const uint32_t below = peers_below_total(digit_word<K>(mk[i], shift), bit0, total);
                            if (below == 0) atomicAdd(&wh[d], total);
                        }
                    }
                }
            } else if (fast) {
#pragma unroll
                for (int i = 0; i < KPT; ++i) {
                    if (i < rounds) {
                        const uint32_t r = atomicAdd(&wh[digit_of(mk[i], shift)], 1u);
                        if (i & 1) run_index[i >> 1] |= r << 16;
                        else run_index[i >> 1] = r;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < KPT; ++i)
                    if (i < rounds) atomicAdd(&wh[digit_of(mk[i], shift)], 1u);
            }
        }
        __syncthreads();
        if (tid < RADIX) __builtin_amdgcn_s_setprio(RDST_PRIO_SCAN);
        uint32_t cw[NWAVES];
        uint32_t count_d = 0;
        if (tid < RADIX) {
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) {
                cw[w] = wave_hist[w * RADIX + tid];
                count_d += cw[w];
            }
            if (count_d == live) s_misc[0] = 1;  // nothing to reorder on this digit (block-wide)
        }
        uint32_t incl = count_d;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (tid < RADIX && lane == 63) s_misc[4 + wave] = incl;
        __syncthreads();
        if (s_misc[0]) {  // block-uniform
            __syncthreads();  // every thread has read the flag before thread 0 clears it again
            continue;
        }
        if (tid < RADIX) {
            uint32_t woff = 0;
            for (int w = 0; w < wave; ++w) woff += s_misc[4 + w];
            uint32_t run = woff + incl - count_d;
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) {
                wave_hist[w * RADIX + tid] = run * SLOT_UNIT;  // first slot of (wave w, digit d)
                run += cw[w];
            }
        }
        __syncthreads();
        __builtin_amdgcn_s_setprio(0);
        // ranking, as in K3 step 5; the key goes straight to its slot of the staging buffer
        bool ranked = false;
        if (fast) {
            bool out_of_order = false;
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                if (i < rounds) {
                    const uint32_t idx = (i & 1) ? (run_index[i >> 1] >> 16) : (run_index[i >> 1] & 0xFFFFu);
                    const uint32_t sl = wh[digit_of(mk[i], shift)] + idx * SLOT_UNIT;
                    *reinterpret_cast<K*>(stage + sl) = mk[i];
                    if (shift) {  // level 0 has no lower digits: any order will do
                        const K prev = *reinterpret_cast<const K*>(stage + sl - SLOT_UNIT);
                        const int up = W - shift;
                        out_of_order |= idx != 0 && (K)(prev << up) > (K)(mk[i] << up);
                    }
                }
            }
            ranked = __builtin_amdgcn_ballot_w64(out_of_order) == 0 && !(flags & RDST_FAST_RANK_SELFTEST);
        }
        if (!ranked) {
            if (!careful) {
#pragma unroll
                for (int i = 0; i < KPT; ++i) {
                    if (i < rounds) {
                        uint32_t* slot = &wh[digit_of(mk[i], shift)];
                        const uint32_t b = *slot;
                        const uint32_t below = peers_below(digit_word<K>(mk[i], shift), bit0);
                        __builtin_amdgcn_wave_barrier();
                        atomicAdd(slot, SLOT_UNIT);
                        *reinterpret_cast<K*>(stage + b + below * SLOT_UNIT) = mk[i];
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < KPT; ++i) {
                    if (i < rounds) {
                        uint32_t* slot = &wh[digit_of(mk[i], shift)];
                        const uint32_t b = *slot;
                        uint32_t below;
                        if ((uniform_rounds >> i) & 1u) {
                            below = (uint32_t)lane;
                            __builtin_amdgcn_wave_barrier();
                            if (lane == 0) *slot = b + 64u * SLOT_UNIT;
                        } else {
                            uint32_t total;
                            below = peers_below_total(digit_word<K>(mk[i], shift), bit0, total);
                            __builtin_amdgcn_wave_barrier();
                            if (below == 0) *slot = b + total * SLOT_UNIT;
                        }
                        *reinterpret_cast<K*>(stage + b + below * SLOT_UNIT) = mk[i];
                    }
                }
            }
        }
        __syncthreads();
        // the next level reads the slice back in its new order (wave-striped again)
#pragma unroll
        for (int i = 0; i < KPT; ++i)
            if (i < rounds) mk[i] = reinterpret_cast<const K*>(stage)[wbase + i * 64];
    }
    // `mk` holds the sorted bucket in (wave, round, lane) order; out through LDS for coalesced stores would
    // cost another round trip, and the wave-striped order already stores 256 contiguous bytes per instruction
    __builtin_amdgcn_s_setprio(RDST_PRIO_SCATTER);
    {
        K* tdst = buf + start;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            if (i < rounds) {
                const uint32_t idx = wbase + i * 64;
                if (idx < cnt) tdst[idx] = MAPPED ? unmap_key<K>(mk[i], neg, pos) : mk[i];
            }
        }
    }
}

// one workgroup per bucket (8-byte keys), or — `list` given — a few workgroups working off the list of
// buckets the counting kernel below had to leave alone (4-byte keys)
template <typename K, int NWAVES, int KPT, bool MAPPED>
__global__ __launch_bounds__(NWAVES * 64, (sizeof(K) <= 4 ? 2 : 1) * NWAVES / 4) void local_sort_kernel(
    K* __restrict__ buf_keys, K* __restrict__ buf_tmp, const uint32_t* __restrict__ bstart, const Plan* __restrict__ plan,
    uint32_t* __restrict__ err, K neg, K pos, uint32_t flags, const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_count,
    const uint16_t* __restrict__ src16, const uint32_t* __restrict__ slot_count, uint32_t slot_cap, const K* __restrict__ alt_src) {
    if (!plan->local_sort) return;
    if (plan->route != ROUTE_ATOMIC) { slot_count = nullptr; alt_src = nullptr; }  // the hybrid route's buckets lie at their final place
    K* __restrict__ buf = plan->result_in_tmp ? buf_tmp : buf_keys;
    if (list == nullptr) {
        const uint32_t bucket = blockIdx.x;
        const uint32_t start = bstart[bucket];
        local_sort_bucket<K, NWAVES, KPT, MAPPED>(buf, src16, nullptr, start, bucket, bucket_prefix16(plan, bucket), start, bstart[bucket + 1] - start, err, neg, pos, flags);
        return;
    }
    const uint32_t todo = *list_count;
#pragma unroll 1
    for (uint32_t e = blockIdx.x; e < todo; e += gridDim.x) {
        const uint32_t bucket = list[e];
        const uint32_t start = bstart[bucket];
        const uint32_t cnt = slot_count ? slot_count[bucket] : bstart[bucket + 1] - start;
        local_sort_bucket<K, NWAVES, KPT, MAPPED>(buf, src16, alt_src, slot_count ? bucket * slot_cap : start, bucket, bucket_prefix16(plan, bucket), start, cnt, err, neg, pos, flags);
        __syncthreads();  // the next bucket reuses the LDS
    }
}

// K4 for 4-byte keys: counting sort BY VALUE.  Inside a bucket the top 16 bits are the bucket index, so a key
// is its low 16 bits: nothing has to be ranked wave by wave or moved twice.
//   count     one RETURNING LDS add per key on a table of 65 536 four-bit counters (32 KiB): the nibble it
//             returns is the key's index among the keys of the same value (any order among equal keys is THE
//             order: they are the same bits)
//   scan      thread t owns values [VPT*t, VPT*t + VPT) as words k * BLOCK + t (conflict-free); it writes a
//             16-bit exclusive prefix per word (8 values)
//   place     every key: prefix of its word + the nibbles below its own in that word + its index -> its slot
//   store     the sorted low halves go through LDS (the tables are dead by then and lend the space) and out
//             with coalesced stores, prefix and inverse key map applied on the way
// ~5 LDS operations and ~30 vector instructions per key, no divergence; the ranked two-pass form above costs
// twice the LDS work, which is what bounds it.  A value seen 16 times in one bucket (skewed low
// bits) would carry into the neighbouring counter: the add that would do it sees the nibble at 15, the block
// then leaves the bucket untouched and queues it for the generic kernel above.
constexpr int COUNT_TILE = local_tile(4);  // the route's bucket bound for 4-byte keys
constexpr size_t count_lds_bytes() { return 32768 + 16384 + 128; }
static_assert(2 * ((size_t)COUNT_TILE + 4) <= 32768 + 16384, "the output staging (shifted by up to three keys) aliases the counter and prefix tables");

__device__ __forceinline__ uint32_t nibble_sum(uint32_t x, uint32_t acc) {
    const uint32_t t = (x & 0x0F0F0F0Fu) + ((x >> 4) & 0x0F0F0F0Fu);
    return __builtin_amdgcn_sad_u8(t, 0u, acc);  // acc + the four byte sums
}

// The kernel is bound by vector-instruction issue, not by memory or LDS (round 3: 46.7 VALU instructions per key = 1.19 of its
// 1.37 ms at one wave-instruction per clock and CU), so its body exists in four forms chosen per bucket, block-uniformly:
//   FAST   the bucket holds at least COUNT_SAFE keys per thread (uniform 10^9-key sorts: always — the mean is 7.5 sigma
//          above): the first COUNT_SAFE rounds of every phase run without the `idx < cnt` predicate (-8 instructions per key)
//   VEC    (atomic route: the halves lie in a slot, 128-byte aligned) the halves arrive as four 16-byte vectors and one
//          16-bit load per thread instead of 33 16-bit loads with their address arithmetic and clamps
// and the sorted bucket leaves in 16-byte stores of four keys (the staging is shifted by the destination's misalignment, so that
// an aligned LDS read of four halves is an aligned global store of four keys): 9 store rounds instead of 33.
constexpr int COUNT_SAFE = 24;  // rounds (keys per thread) that need no predicate in a FAST bucket

// FROM16: the bucket arrives as the low halves of the mapped keys (pass L-1 stored only those, see OUT16 of K3)
template <int BLOCK, bool MAPPED, bool FROM16, bool FAST, bool VEC>
__device__ __forceinline__ void count_sort_bucket(uint32_t* __restrict__ buf, const uint16_t* __restrict__ src16, uint32_t soff, uint32_t start, uint32_t cnt,
                                                  uint32_t bucket, const Plan* __restrict__ plan, uint32_t neg, uint32_t pos, uint32_t* __restrict__ list,
                                                  uint32_t* __restrict__ list_count, unsigned char* smem) {
    constexpr int MAXR = (COUNT_TILE + BLOCK - 1) / BLOCK;
    constexpr int VPT = H16_BINS / BLOCK, WPT = VPT / 8;  // values / counter words per thread
    constexpr int LOG_VPT = BLOCK == 1024 ? 6 : (BLOCK == 512 ? 7 : 8);
    static_assert((1 << LOG_VPT) == VPT, "block size");
    static_assert(!VEC || (FROM16 && MAXR == 33 && BLOCK == 512), "vector loads: four vectors of eight halves and one half per thread");
    static_assert(COUNT_SAFE % 8 == 0 && COUNT_SAFE < MAXR, "whole vectors");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* cnt4 = reinterpret_cast<uint32_t*>(smem);                    // [WPT][BLOCK] words of eight 4-bit counters
    uint16_t* prefix = reinterpret_cast<uint16_t*>(smem + 32768);          // [WPT][BLOCK] keys below the word's first value
    uint16_t* out16 = reinterpret_cast<uint16_t*>(smem);                   // [cnt + 3] sorted low halves (aliases both, later)
    uint32_t* s_wsum = reinterpret_cast<uint32_t*>(smem + 32768 + 16384);  // [16] wave sums, [16] overflow flag
    // key i of this thread is element idx_of(i) of the bucket (any assignment will do: the order inside a bucket means nothing
    // yet).  Either way idx_of grows with i, so a thread's valid keys are its first `nvalid`: ONE register holds every predicate
    // of every phase.  (Each phase compares against its own opaque copy: left alone, the compiler computes the 33 lane masks once
    // and keeps them in 66 scalar registers across all phases — which, with everything else that is live, it then spills.)
    auto idx_of = [&](int i) -> uint32_t {
        if constexpr (VEC) return i < 32 ? 8u * ((uint32_t)tid + (uint32_t)(i >> 3) * BLOCK) + (uint32_t)(i & 7) : 32u * BLOCK + (uint32_t)tid;
        else return (uint32_t)tid + (uint32_t)i * BLOCK;
    };
    uint32_t nvalid = 0;
    if constexpr (VEC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t v0 = 8u * ((uint32_t)tid + (uint32_t)j * BLOCK);
            nvalid += cnt > v0 ? (cnt - v0 < 8u ? cnt - v0 : 8u) : 0u;
        }
        nvalid += 32u * BLOCK + (uint32_t)tid < cnt ? 1u : 0u;
    } else {
        nvalid = cnt > (uint32_t)tid ? (cnt - (uint32_t)tid + BLOCK - 1u) / BLOCK : 0u;
    }
#define RDST_K4_PHASE() uint32_t nv_ = nvalid; asm volatile("" : "+v"(nv_))
#define RDST_K4_VALID(i) ((FAST && (i) < COUNT_SAFE) || (uint32_t)(i) < nv_)
    // Without predicates the unrolled rounds of a phase are one basic block each, and the compiler, left alone, carries every
    // round's table address and shift from the count phase over to the place phase (66 registers it does not have: 230 dwords of
    // spills) instead of recomputing three instructions: the keys are made opaque between the phases.
#define RDST_K4_FENCE(i) do { if (((i) & 3) == 3 || (i) == MAXR - 1) { _Pragma("unroll") for (int f_ = (i) - ((i) & 3); f_ <= (i); ++f_) asm volatile("" : "+v"(kv[f_])); } } while (0)  // (a group of rounds is pinned where it is computed: left alone, every round's LDS reads are issued first and their results spilled)
#define RDST_K4_OPAQUE() do { _Pragma("unroll") for (int i_ = 0; i_ < MAXR; ++i_) asm volatile("" : "+v"(kv[i_])); } while (0)
    __builtin_amdgcn_s_setprio(RDST_PRIO_LOAD);
    uint32_t kv[MAXR];
    if constexpr (VEC) {
        const uint4* vsrc = reinterpret_cast<const uint4*>(src16 + soff);  // (slot_cap is a multiple of 64 halves, the halves array 256-byte aligned)
        uint4 q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t vi = (uint32_t)tid + j * BLOCK;
            if ((FAST && j < COUNT_SAFE / 8) || 8u * vi < cnt) q[j] = vsrc[vi];  // (a vector that starts inside the bucket lies inside the slot: its capacity is a multiple of 64)
            else q[j] = make_uint4(0, 0, 0, 0);
        }
        kv[32] = 32u * BLOCK + (uint32_t)tid < cnt ? (uint32_t)src16[soff + 32u * BLOCK + (uint32_t)tid] : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t d[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                kv[8 * j + 2 * e] = d[e] & 0xFFFFu;
                kv[8 * j + 2 * e + 1] = d[e] >> 16;
            }
        }
    } else {
        RDST_K4_PHASE();
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const uint32_t idx = idx_of(i);
            const uint32_t at = RDST_K4_VALID(i) ? idx : cnt - 1;
            if constexpr (FROM16) kv[i] = src16[soff + at];
            else kv[i] = buf[start + at];
        }
    }
#pragma unroll
    for (int k = 0; k < WPT; ++k) cnt4[k * BLOCK + tid] = 0;
    if (tid == 0) s_wsum[16] = 0;
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);
    auto word_of = [](uint32_t v) -> uint32_t { return v >> 3; };  // (linear: consecutive values lie on consecutive banks — dense ids put 128 of them on ONE bank in the thread-major table —, and the address is two instructions instead of five; the scan pays with a rotated read order)
    bool over = false;
    {
    RDST_K4_PHASE();
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        if (RDST_K4_VALID(i)) {
            const uint32_t v = FROM16 ? kv[i] : ((MAPPED ? map_key<uint32_t>(kv[i], neg, pos) : kv[i]) & 0xFFFFu);
            const uint32_t sh = (v & 7u) * 4u;
            const uint32_t old = atomicAdd(&cnt4[word_of(v)], 1u << sh);
            const uint32_t mine = (old >> sh) & 15u;
            over |= mine == 15u;
            kv[i] = v | (mine << 16);
        }
        RDST_K4_FENCE(i);
    }
    }
    if (over) s_wsum[16] = 1;
    __syncthreads();
    if (s_wsum[16]) {  // block-uniform: a value 16 times
        // one value only (the bimodal bench input: every bucket of the shifted half)?  Then the bucket is written at once ...
        bool differ = false;
        uint32_t mine_first = 0;
        bool have = false;
        RDST_K4_PHASE();
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            if (RDST_K4_VALID(i)) {
                if (!have) { mine_first = kv[i] & 0xFFFFu; have = true; }
                differ |= (kv[i] & 0xFFFFu) != mine_first;
            }
        }
        // (kv: value | index << 16 by now; element 0 of the bucket is key 0 of thread 0 in both assignments)
        if (tid == 0) s_wsum[17] = mine_first;
        const int any = __syncthreads_or((int)differ);       // every thread of one value ...
        const bool agree = !have || mine_first == s_wsum[17];
        if (!any && !__syncthreads_or((int)!agree)) {       // ... and the same one
            const uint32_t m = (bucket_prefix16(plan, bucket) << 16) | s_wsum[17];
            const uint32_t out = MAPPED ? unmap_key<uint32_t>(m, neg, pos) : m;
            for (uint32_t idx = (uint32_t)tid; idx < cnt; idx += BLOCK) buf[start + idx] = out;
            return;
        }
        // ... else it stays as it is, for the next kernel
        if (tid == 0) list[atomicAdd(list_count, 1u)] = bucket;
        return;
    }
    {
        // thread t owns words [WPT t, WPT t + WPT) and reads them in a rotated order — word (k + r) mod WPT in step k, r = t / (64 / WPT)
        // — so that the 64 lanes of a step sit on 64 different banks; the prefixes are put right afterwards: the words from r up
        // arrive first (before them lie the words below r: total - the sum through the last word), the words below r after those
        constexpr int ROT_SHIFT = WPT == 16 ? 2 : 3;
        static_assert(WPT == 16 || WPT == 8, "64 banks / WPT lanes per rotation");
        const uint32_t r = ((uint32_t)tid >> ROT_SHIFT) & (uint32_t)(WPT - 1);
        uint32_t arr[WPT];  // keys in the words that arrived before step k
        uint32_t run = 0, through_last = 0;
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            const uint32_t j = ((uint32_t)k + r) & (uint32_t)(WPT - 1);
            arr[k] = run;
            run = nibble_sum(cnt4[WPT * tid + j], run);
            through_last = j == (uint32_t)(WPT - 1) ? run : through_last;
        }
        const uint32_t below_r = run - through_last;  // keys in my words 0 .. r - 1
        uint32_t incl = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t base = incl - run;
#pragma unroll
        for (int x = 0; x < BLOCK / 64; ++x)
            if (x < wave) base += s_wsum[x];
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            const uint32_t j = ((uint32_t)k + r) & (uint32_t)(WPT - 1);
            const uint32_t before = (uint32_t)k + r < (uint32_t)WPT ? below_r + arr[k] : arr[k] - through_last;
            prefix[WPT * tid + j] = (uint16_t)(base + before);
        }
    }
    __syncthreads();
    // the staging is shifted by the destination's misalignment (in keys, 0..3): LDS quad q then is the 16-byte aligned global quad q
    uint32_t* tdst = buf + start;
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(tdst) >> 2) & 3u;
    {
    RDST_K4_PHASE();
    RDST_K4_OPAQUE();
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        if (RDST_K4_VALID(i)) {
            const uint32_t v = kv[i] & 0xFFFFu, mine = kv[i] >> 16;
            const uint32_t wd = word_of(v);
            const uint32_t below = cnt4[wd] & ((1u << ((v & 7u) * 4u)) - 1u);
            const uint32_t slot = nibble_sum(below, (uint32_t)prefix[wd] + mine + mis);
            kv[i] = v | (slot << 16);
        }
        RDST_K4_FENCE(i);
    }
    }
    __syncthreads();  // every read of the tables is done: their space becomes the output staging
    {
    RDST_K4_PHASE();
    RDST_K4_OPAQUE();
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        if (RDST_K4_VALID(i)) out16[kv[i] >> 16] = (uint16_t)kv[i];
    }
    }
    __syncthreads();
    __builtin_amdgcn_s_setprio(RDST_PRIO_SCATTER);
    const uint32_t top = bucket_prefix16(plan, bucket) << 16;
    const uint32_t quads = (cnt + mis + 3u) >> 2;  // staged positions [mis, mis + cnt) in quads of four
    constexpr int QR = (COUNT_TILE + 3 + 3) / 4 / BLOCK + 1;
    const uint2* o2 = reinterpret_cast<const uint2*>(out16);
    uint4* gq = reinterpret_cast<uint4*>(tdst - mis);  // 16-byte aligned (quad 0 may begin before the bucket: its first `mis` keys are not stored)
#pragma unroll
    for (int r = 0; r < QR; ++r) {
        const uint32_t qi = (uint32_t)tid + (uint32_t)r * BLOCK;
        if (qi < quads) {
            const uint2 h = o2[qi];
            uint4 k4;
            k4.x = top | (h.x & 0xFFFFu); k4.y = top | (h.x >> 16); k4.z = top | (h.y & 0xFFFFu); k4.w = top | (h.y >> 16);
            if constexpr (MAPPED) {
                k4.x = unmap_key<uint32_t>(k4.x, neg, pos); k4.y = unmap_key<uint32_t>(k4.y, neg, pos);
                k4.z = unmap_key<uint32_t>(k4.z, neg, pos); k4.w = unmap_key<uint32_t>(k4.w, neg, pos);
            }
            const uint32_t p0 = 4u * qi;  // staged position of the quad's first key; the bucket's keys are [mis, mis + cnt)
            if (p0 >= mis && p0 + 3u < mis + cnt) {
                gq[qi] = k4;
            } else {  // the bucket's first and last quad
                uint32_t* g = reinterpret_cast<uint32_t*>(gq + qi);
                if (p0 + 0u >= mis && p0 + 0u < mis + cnt) g[0] = k4.x;
                if (p0 + 1u >= mis && p0 + 1u < mis + cnt) g[1] = k4.y;
                if (p0 + 2u >= mis && p0 + 2u < mis + cnt) g[2] = k4.z;
                if (p0 + 3u >= mis && p0 + 3u < mis + cnt) g[3] = k4.w;
            }
        }
    }
}
#undef RDST_K4_PHASE
#undef RDST_K4_FENCE
#undef RDST_K4_OPAQUE
#undef RDST_K4_VALID

template <int BLOCK, bool MAPPED, bool FROM16>
__global__ __launch_bounds__(BLOCK, BLOCK == 1024 ? 8 : 6) void local_count_sort_kernel(
    uint32_t* __restrict__ buf_keys, uint32_t* __restrict__ buf_tmp, const uint16_t* __restrict__ src16, const uint32_t* __restrict__ bstart,
    const Plan* __restrict__ plan, uint32_t* __restrict__ err, uint32_t neg, uint32_t pos, uint32_t* __restrict__ list,
    uint32_t* __restrict__ list_count, const uint32_t* __restrict__ slot_count /* ROUTE_ATOMIC: bucket b's halves lie in slot b (slot_cap
    entries) of src16 and number slot_count[b]; NULL: they lie at their final place, bstart */, uint32_t slot_cap) {
    if (!plan->local_sort) return;
    if (plan->route != ROUTE_ATOMIC) slot_count = nullptr;  // the hybrid route's buckets lie at their final place (src16: position for position)
    uint32_t* __restrict__ buf = plan->result_in_tmp ? buf_tmp : buf_keys;
    const uint32_t bucket = blockIdx.x;
    const uint32_t start = bstart[bucket], cnt = slot_count ? slot_count[bucket] : bstart[bucket + 1] - start;
    const uint32_t soff = slot_count ? bucket * slot_cap : start;  // where the bucket's keys lie now
    const int tid = threadIdx.x;
    if (cnt <= 1) {
        if constexpr (FROM16) {  // the key exists only as its low half: put it back together
            if (cnt == 1 && tid == 0) {
                const uint32_t m = (bucket_prefix16(plan, bucket) << 16) | (uint32_t)src16[soff];
                buf[start] = MAPPED ? unmap_key<uint32_t>(m, neg, pos) : m;
            }
        }
        return;
    }
    if (cnt >= GIANT_MIN && plan->giants) return;  // the giant kernels' (route_kernel listed it)
    if (cnt > (uint32_t)COUNT_TILE || plan->low_dups) {  // more than this kernel stages, or (the sample says) low halves its 4-bit counters cannot count: handed on
        if (tid == 0) list[atomicAdd(list_count, 1u)] = bucket;
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const bool fast = cnt >= (uint32_t)COUNT_SAFE * BLOCK;  // block-uniform
    if constexpr (FROM16 && BLOCK == 512) {
        if (slot_count) {  // the atomic route: aligned slots
            if (fast) count_sort_bucket<BLOCK, MAPPED, FROM16, true, true>(buf, src16, soff, start, cnt, bucket, plan, neg, pos, list, list_count, smem);
            else count_sort_bucket<BLOCK, MAPPED, FROM16, false, true>(buf, src16, soff, start, cnt, bucket, plan, neg, pos, list, list_count, smem);
            return;
        }
    }
    if (fast) count_sort_bucket<BLOCK, MAPPED, FROM16, true, false>(buf, src16, soff, start, cnt, bucket, plan, neg, pos, list, list_count, smem);
    else count_sort_bucket<BLOCK, MAPPED, FROM16, false, false>(buf, src16, soff, start, cnt, bucket, plan, neg, pos, list, list_count, smem);
}

// K4 for the buckets the kernel above hands on (4-byte keys): more keys than its tile, or a value seen 16 times.
// Counting sort by value again, but the keys are only COUNTED — a bucket's keys are its index and their low 16 bits,
// so the sorted bucket is the count table written out: value v, count[v] times.  Nothing is staged, ranked or
// moved, the bucket may hold anything below 65 536 keys (16-bit counters cannot carry then, whatever the keys are)
// and duplicates cost nothing extra.
//   count     one LDS add per key on 65 536 16-bit counters (128 KiB; a wave whose 64 keys are one value adds once)
//   scan      thread t owns 64 consecutive values (32 words, stored word-major so the sweep is conflict-free) and
//             replaces the counts by exclusive prefixes
//   expand    output position i holds the largest v with prefix[v] <= i: run starts marked in the (cleared) table, a
//             max-scan over the positions, then 16-byte stores of (bucket << 16 | v) with the key map undone
// One block per CU walks the list.  It runs on what the kernel above refuses: skewed low halves (a bucket of 15 000
// keys over 256 distinct values) and buckets of up to four tiles.
constexpr int EXPAND_THREADS = 1024;
constexpr uint32_t EXPAND_MAX = 65535;  // keys per bucket
constexpr int EXPAND_WORDS = H16_BINS / 2 + H16_BINS / 64;  // two counters per word, one pad word after every 32 (a multiple of 4)
constexpr size_t expand_lds_bytes() { return 4 * (size_t)EXPAND_WORDS + 128; }

template <bool MAPPED, bool FROM16>
__global__ __launch_bounds__(EXPAND_THREADS) void local_expand_sort_kernel(
    uint32_t* __restrict__ buf_keys, uint32_t* __restrict__ buf_tmp, const uint16_t* __restrict__ src16, const uint32_t* __restrict__ bstart,
    const Plan* __restrict__ plan, uint32_t* __restrict__ err, uint32_t neg, uint32_t pos, const uint32_t* __restrict__ list,
    const uint32_t* __restrict__ list_count, const uint32_t* __restrict__ slot_count, uint32_t slot_cap) {
    constexpr int BLOCK = EXPAND_THREADS, WPT = H16_BINS / 2 / BLOCK;  // 32 words of two counters per thread
    if (!plan->local_sort) return;
    if (plan->route != ROUTE_ATOMIC) slot_count = nullptr;
    uint32_t* __restrict__ buf = plan->result_in_tmp ? buf_tmp : buf_keys;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // word w = v / 2 lives at w + w / 32: thread t's 32 consecutive words start at 33 t (the scan's sweep is conflict-free) and
    // neighbouring values stay neighbours (a bucket whose low halves differ in a few bits only — what this kernel is for —
    // would hit two or four banks with a thread-major table)
    uint32_t* tab = reinterpret_cast<uint32_t*>(smem);
    uint32_t* s_wsum = reinterpret_cast<uint32_t*>(smem + 4 * (size_t)EXPAND_WORDS);  // [16]
    uint32_t* s_wmax = s_wsum + 16;                                                    // [16]
    uint16_t* tab16 = reinterpret_cast<uint16_t*>(smem);  // the table again, as what the expansion makes of it: one half per output position
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto word_of = [](uint32_t v) -> uint32_t { return (v >> 1) + (v >> 6); };
    auto count_key = [&](uint32_t x, bool valid) {
        const uint32_t x0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
        if (__all((int)(valid && x == x0)) != 0) {  // 64 keys of one value: one add, not 64 on one address
            if (lane == 0) atomicAdd(&tab[word_of(x0)], 64u << ((x0 & 1u) * 16u));
        } else if (valid) {
            atomicAdd(&tab[word_of(x)], 1u << ((x & 1u) * 16u));
        }
    };
    auto clear_table = [&]() {
        uint4* t4 = reinterpret_cast<uint4*>(tab);
        for (int i = tid; i < EXPAND_WORDS / 4; i += BLOCK) t4[i] = make_uint4(0, 0, 0, 0);
    };
    const uint32_t total = *list_count;
    // FROM16: eight halves per load, four loads per thread in flight — a bucket of up to 32 768 keys is one round trip to memory
    // (3.5 us of a 30 000-key bucket's 18, with the table cleared meanwhile.  Starting the next bucket's trip early — into
    // registers, or one touch per 128-byte line for L2 — moved the wait to the next vmcnt the compiler placed and gained nothing.)
    constexpr int U2 = 4;
    uint4 q[U2];
    auto load = [&](uint64_t base, uint64_t g1) {  // (src16's base is 16-byte aligned and its end padded)
#pragma unroll
        for (int j = 0; j < U2; ++j) {
            const uint64_t p = base + ((uint64_t)j * BLOCK + (uint64_t)tid) * 8;
            q[j] = p < g1 ? *reinterpret_cast<const uint4*>(src16 + p) : uint4{0, 0, 0, 0};
        }
    };
    for (uint32_t e = blockIdx.x; e < total; e += gridDim.x) {
        RDST_TL_BEGIN(5u);
        const uint32_t bucket = list[e];
        const uint32_t start = bstart[bucket], cnt = slot_count ? slot_count[bucket] : bstart[bucket + 1] - start;
        const uint32_t soff = slot_count ? bucket * slot_cap : start;
        if (cnt > EXPAND_MAX) {  // the route checks every bucket against this bound before it sends any here
            if (tid == 0) atomicOr(err, ERR_LOCAL_OVERFLOW);
            continue;
        }
        // count: the table is cleared while the first loads are in flight
        if constexpr (FROM16) {
            const uint64_t g0 = soff, g1 = (uint64_t)soff + cnt;  // element range in src16
            uint64_t base = g0 & ~7ull;
            load(base, g1);
            clear_table();
            __syncthreads();
            RDST_STAMP(1);
            for (;;) {  // block-uniform trip count
#pragma unroll
                for (int j = 0; j < U2; ++j) {
                    const uint64_t p = base + ((uint64_t)j * BLOCK + (uint64_t)tid) * 8;
                    const uint32_t w[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
                    const bool whole = p >= g0 && p + 8 <= g1;  // all eight halves are the bucket's
                    // the lanes hold the SAME eight halves (a constant, a pattern whose period divides eight): 64 keys per value and add
                    // would be 64 adds on one address — one lane adds 64 per half instead
                    const uint32_t f[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)w[0]), (uint32_t)__builtin_amdgcn_readfirstlane((int)w[1]),
                                           (uint32_t)__builtin_amdgcn_readfirstlane((int)w[2]), (uint32_t)__builtin_amdgcn_readfirstlane((int)w[3])};
                    if (__all((int)(whole && w[0] == f[0] && w[1] == f[1] && w[2] == f[2] && w[3] == f[3])) != 0) {
                        if (lane == 0) {
#pragma unroll
                            for (int h = 0; h < 8; ++h) {
                                const uint32_t x = (f[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu;
                                atomicAdd(&tab[word_of(x)], 64u << ((x & 1u) * 16u));
                            }
                        }
                    } else if (whole) {
#pragma unroll
                        for (int h = 0; h < 8; ++h) {
                            const uint32_t x = (w[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu;
                            atomicAdd(&tab[word_of(x)], 1u + (x & 1u) * 0xFFFFu);
                        }
                    } else {
#pragma unroll
                        for (int h = 0; h < 8; ++h) {
                            const uint32_t x = (w[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu;
                            if (p + h >= g0 && p + h < g1) atomicAdd(&tab[word_of(x)], 1u + (x & 1u) * 0xFFFFu);
                        }
                    }
                }
                base += (uint64_t)BLOCK * 8 * U2;
                if (base >= g1) break;
                load(base, g1);
            }
        } else {
            clear_table();
            __syncthreads();
            RDST_STAMP(1);
            constexpr int U = 8;  // keys in flight per thread
            for (uint32_t base = 0; base < cnt; base += BLOCK * U) {  // wave-uniform trip count
                uint32_t v[U];
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const uint32_t idx = base + (uint32_t)j * BLOCK + (uint32_t)tid;
                    v[j] = buf[start + (idx < cnt ? idx : cnt - 1)];
                }
#pragma unroll
                for (int j = 0; j < U; ++j)
                    count_key((MAPPED ? map_key<uint32_t>(v[j], neg, pos) : v[j]) & 0xFFFFu, base + (uint32_t)j * BLOCK + (uint32_t)tid < cnt);
            }
        }
        RDST_STAMP(2);
        __syncthreads();
        RDST_STAMP(3);
        // scan: my 64 values' counts leave the table for registers (their words are cleared for the expansion below), `below` =
        // the keys under my first value
        uint32_t cw[WPT];
        uint32_t below;
        {
            uint32_t run = 0;
#pragma unroll
            for (int k = 0; k < WPT; ++k) {
                cw[k] = tab[(WPT + 1) * tid + k];
                run += (cw[k] & 0xFFFFu) + (cw[k] >> 16);
            }
#pragma unroll
            for (int k = 0; k <= WPT; ++k) tab[(WPT + 1) * tid + k] = 0;  // (with the pad word: the expansion's 33rd)
            uint32_t incl = run;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t y = __shfl_up(incl, o);
                if (lane >= o) incl += y;
            }
            if (lane == 63) s_wsum[wave] = incl;
            __syncthreads();
            below = incl - run;
#pragma unroll
            for (int x = 0; x < BLOCK / 64; ++x)
                if (x < wave) below += s_wsum[x];
        }
        RDST_STAMP(4);
        // expand (round 3 form): the value at output position p is the largest v with prefix[v] <= p, and values grow with the
        // position, so it is a running maximum over marked run starts.  Every value that has keys writes itself at the position
        // its run starts at (16 bits per position in the cleared table: 0 = no mark, and value 0 can only start at the first
        // position, where nothing lies below it); a max-scan — thread t owns 66 positions, its 33 words, conflict-free as
        // above — spreads the marks, in place; then the sorted halves are read back four at a time and leave as whole keys
        // in 16-byte stores (positions are counted from the 16-byte boundary below the bucket's first key: q = p + mis).
        // ~2 LDS operations and ~20 vector instructions per key, all of them with every lane busy (round 2 searched the table
        // per position, 16 dependent LDS reads: 37 us of a 30 000-key bucket's 50).
        uint32_t* tdst = buf + start;
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(tdst) >> 2) & 3u;
        const uint32_t q_end = cnt + mis;  // <= 65 538 <= 66 positions x 1 024 threads
        {
            uint32_t at = below + mis;
#pragma unroll
            for (int k = 0; k < WPT; ++k) {
                const uint32_t c0 = cw[k] & 0xFFFFu, c1 = cw[k] >> 16;
                const uint32_t v0 = 64u * (uint32_t)tid + 2u * (uint32_t)k;
                if (c0) tab16[at] = (uint16_t)v0;
                at += c0;
                if (c1) tab16[at] = (uint16_t)(v0 + 1u);
                at += c1;
            }
        }
        __syncthreads();
        constexpr uint32_t WAVE_POSITIONS = 64u * 2u * (WPT + 1);
        const bool scans = (uint32_t)wave * WAVE_POSITIONS < q_end;  // wave-uniform: my wave's positions hold keys
        {
            uint32_t m = 0;
            if (scans) {
#pragma unroll
                for (int k = 0; k <= WPT; ++k) {
                    const uint32_t w = tab[(WPT + 1) * tid + k];
                    const uint32_t lo = w & 0xFFFFu, hi = w >> 16;
                    m = lo > m ? lo : m;
                    const uint32_t first = m;
                    m = hi > m ? hi : m;
                    if (k < WPT) cw[k] = first | (m << 16);
                    else below = first | (m << 16);
                }
            }
            uint32_t incl = m;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t y = __shfl_up(incl, o);
                if (lane >= o && y > incl) incl = y;
            }
            if (lane == 63) s_wmax[wave] = incl;
            uint32_t carry = __shfl_up(incl, 1);
            if (lane == 0) carry = 0;
            __syncthreads();
            if (scans) {
#pragma unroll
                for (int x = 0; x < BLOCK / 64; ++x)
                    if (x < wave && s_wmax[x] > carry) carry = s_wmax[x];
                // (marks grow with the position: my positions before my first mark take the carry, the others are above it already)
#pragma unroll
                for (int k = 0; k <= WPT; ++k) {
                    const uint32_t w = k < WPT ? cw[k] : below;
                    const uint32_t lo = w & 0xFFFFu, hi = w >> 16;
                    tab[(WPT + 1) * tid + k] = (lo > carry ? lo : carry) | ((hi > carry ? hi : carry) << 16);
                }
            }
        }
        __syncthreads();
        {
            const uint32_t top = bucket_prefix16(plan, bucket) << 16;
            uint4* gq = reinterpret_cast<uint4*>(tdst - mis);  // quad j = shifted positions 4 j .. 4 j + 3, 16-byte aligned
            const uint2* lq = reinterpret_cast<const uint2*>(tab16);
            const uint32_t quads = (q_end + 3u) >> 2;
            for (uint32_t j = tid; j < quads; j += BLOCK) {
                const uint2 m = lq[j];
                uint32_t k4[4] = {top | (m.x & 0xFFFFu), top | (m.x >> 16), top | (m.y & 0xFFFFu), top | (m.y >> 16)};
                if constexpr (MAPPED) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) k4[c] = unmap_key<uint32_t>(k4[c], neg, pos);
                }
                const uint32_t q0 = 4u * j;
                if (q0 >= mis && q0 + 3u < q_end) {
                    gq[j] = make_uint4(k4[0], k4[1], k4[2], k4[3]);
                } else {  // the bucket's first and last quad
                    uint32_t* g = reinterpret_cast<uint32_t*>(gq + j);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (q0 + (uint32_t)c >= mis && q0 + (uint32_t)c < q_end) g[c] = k4[c];
                }
            }
        }
        RDST_STAMP(5);
        __syncthreads();  // the next bucket clears the table
        RDST_STAMP(6); RDST_STAMP(7); RDST_STAMP(8);
        RDST_TL_END(e);
    }
}

// Between the two: buckets of up to 17 408 keys whose low halves repeat (what the 4-bit counters of the first kernel
// refuse).  The first kernel's scheme — returning add, scan, place, stage, store; keys in registers throughout — on the
// 16-bit counters and the padded table of the expanding kernel: a counter holds any count a bucket of this size can
// produce, so every bucket that fits the registers is sorted here whatever its keys are, at ~5 LDS operations per key
// instead of the expanding kernel's ~20.  One block per CU walks the list (the table fills the LDS) and hands buckets
// over its tile to a second list, for the expanding kernel.
constexpr int COUNT16_THREADS = 1024, COUNT16_KPT = 17;
constexpr int COUNT16_TILE = COUNT16_THREADS * COUNT16_KPT;  // 17 408 >= the route's tile (16 896)
static_assert(COUNT16_TILE >= COUNT_TILE && 2 * (size_t)COUNT16_TILE <= 4 * (size_t)EXPAND_WORDS, "staging aliases the table");

template <bool MAPPED, bool FROM16>
__global__ __launch_bounds__(COUNT16_THREADS) void local_count16_sort_kernel(
    uint32_t* __restrict__ buf_keys, uint32_t* __restrict__ buf_tmp, const uint16_t* __restrict__ src16, const uint32_t* __restrict__ bstart,
    const Plan* __restrict__ plan, uint32_t neg, uint32_t pos, const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_count,
    uint32_t* __restrict__ list2, uint32_t* __restrict__ list2_count, const uint32_t* __restrict__ slot_count, uint32_t slot_cap) {
    constexpr int BLOCK = COUNT16_THREADS, KPT = COUNT16_KPT, WPT = H16_BINS / 2 / BLOCK;
    if (!plan->local_sort) return;
    if (plan->route != ROUTE_ATOMIC) slot_count = nullptr;
    uint32_t* __restrict__ buf = plan->result_in_tmp ? buf_tmp : buf_keys;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* tab = reinterpret_cast<uint32_t*>(smem);   // 65 536 16-bit counters, then prefixes (layout: local_expand_sort_kernel)
    uint16_t* out16 = reinterpret_cast<uint16_t*>(smem); // [cnt] sorted low halves (aliases the table, later)
    uint32_t* s_wsum = reinterpret_cast<uint32_t*>(smem + 4 * (size_t)EXPAND_WORDS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto word_of = [](uint32_t v) -> uint32_t { return (v >> 1) + (v >> 6); };
    const uint32_t total = *list_count;
    // the keys of the next bucket are fetched while this one is sorted (one block per CU: nothing else hides the latency)
    uint32_t nxt[KPT], nxt_first = 0;
    auto fetch = [&](uint32_t e) {
        if (e >= total) return;
        const uint32_t bucket = list[e];
        const uint32_t start = bstart[bucket], cnt = slot_count ? slot_count[bucket] : bstart[bucket + 1] - start;
        const uint32_t soff = slot_count ? bucket * slot_cap : start;
        if (cnt > (uint32_t)COUNT16_TILE || cnt == 0) return;
        if constexpr (FROM16) nxt_first = src16[soff];
        else nxt_first = buf[start];
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t idx = (uint32_t)tid + i * BLOCK;
            const uint32_t at = idx < cnt ? idx : cnt - 1;
            if constexpr (FROM16) nxt[i] = src16[soff + at];
            else nxt[i] = buf[start + at];
        }
    };
    fetch(blockIdx.x);
    for (uint32_t e = blockIdx.x; e < total; e += gridDim.x) {
        const uint32_t bucket = list[e];
        const uint32_t start = bstart[bucket], cnt = slot_count ? slot_count[bucket] : bstart[bucket + 1] - start;
        if (cnt > (uint32_t)COUNT16_TILE) {
            if (tid == 0) list2[atomicAdd(list2_count, 1u)] = bucket;
            fetch(e + gridDim.x);
            continue;
        }
        uint32_t kv[KPT];
#pragma unroll
        for (int i = 0; i < KPT; ++i) kv[i] = nxt[i];
        const uint32_t first = nxt_first;
        // (in place — not FROM16, the hybrid route fed with whole keys — the next bucket is another range of buf: no hazard)
        fetch(e + gridDim.x);
        {   // a bucket of one value (the bimodal bench input: every bucket of the shifted half) is written at once
            bool differ = false;
#pragma unroll
            for (int i = 0; i < KPT; ++i) differ |= kv[i] != first;  // (slots past cnt repeat the last key)
            if (!__syncthreads_or((int)differ)) {
                if constexpr (FROM16) {
                    const uint32_t m = (bucket_prefix16(plan, bucket) << 16) | first;
                    const uint32_t out = MAPPED ? unmap_key<uint32_t>(m, neg, pos) : m;
#pragma unroll
                    for (int i = 0; i < KPT; ++i) {
                        const uint32_t idx = (uint32_t)tid + i * BLOCK;
                        if (idx < cnt) buf[start + idx] = out;
                    }
                }  // (whole keys in place are where they belong already)
                continue;
            }
        }
        {
            uint4* t4 = reinterpret_cast<uint4*>(tab);
            const uint4 z = {0, 0, 0, 0};
            for (int i = tid; i < EXPAND_WORDS / 4; i += BLOCK) t4[i] = z;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t idx = (uint32_t)tid + i * BLOCK;
            const bool valid = idx < cnt;  // wave-uniform except in the bucket's last wave
            const uint32_t v = FROM16 ? kv[i] : ((MAPPED ? map_key<uint32_t>(kv[i], neg, pos) : kv[i]) & 0xFFFFu);
            const uint32_t sh = (v & 1u) * 16u;
            const uint32_t v0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
            uint32_t mine = 0;
            if (__all((int)(valid && v == v0)) != 0) {  // 64 keys of one value: one add for the wave, the lanes take consecutive indices
                uint32_t old = 0;
                if (lane == 0) old = atomicAdd(&tab[word_of(v0)], 64u << sh);
                mine = (((uint32_t)__builtin_amdgcn_readfirstlane((int)old) >> sh) & 0xFFFFu) + (uint32_t)lane;
            } else if (valid) {
                mine = (atomicAdd(&tab[word_of(v)], 1u << sh) >> sh) & 0xFFFFu;
            }
            kv[i] = v | (mine << 16);
        }
        __syncthreads();
        {
            uint32_t run = 0;
#pragma unroll
            for (int k = 0; k < WPT; ++k) {
                const uint32_t w = tab[(WPT + 1) * tid + k];
                run += (w & 0xFFFFu) + (w >> 16);
            }
            uint32_t incl = run;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t y = __shfl_up(incl, o);
                if (lane >= o) incl += y;
            }
            if (lane == 63) s_wsum[wave] = incl;
            __syncthreads();
            uint32_t below = incl - run;
#pragma unroll
            for (int x = 0; x < BLOCK / 64; ++x)
                if (x < wave) below += s_wsum[x];
#pragma unroll
            for (int k = 0; k < WPT; ++k) {
                const uint32_t w = tab[(WPT + 1) * tid + k];
                const uint32_t lo = below, hi = below + (w & 0xFFFFu);
                below = hi + (w >> 16);
                tab[(WPT + 1) * tid + k] = lo | (hi << 16);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t v = kv[i] & 0xFFFFu;
            const uint32_t slot = ((tab[word_of(v)] >> ((v & 1u) * 16u)) & 0xFFFFu) + (kv[i] >> 16);
            kv[i] = v | (slot << 16);
        }
        __syncthreads();  // the table is dead: its space becomes the output staging
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t idx = (uint32_t)tid + i * BLOCK;
            if (idx < cnt) out16[kv[i] >> 16] = (uint16_t)kv[i];
        }
        __syncthreads();
        uint32_t* tdst = buf + start;
        const uint32_t top = bucket_prefix16(plan, bucket) << 16;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t idx = (uint32_t)tid + i * BLOCK;
            if (idx < cnt) {
                const uint32_t m = top | (uint32_t)out16[idx];
                tdst[idx] = MAPPED ? unmap_key<uint32_t>(m, neg, pos) : m;
            }
        }
        __syncthreads();  // the next bucket clears the table
    }
}

// K4 for giants (4-byte keys): buckets of 65 536 keys and more — the reference's bimodal bench input puts half the slice in
// one, a column of small integers or of floats of one magnitude all of it in a few.  The expanding idea across workgroups:
// a bucket's sorted form is its count table written out, so the keys are read once (as 16-bit halves) and never moved.
//   zero     the giants' tables (65 536 u32 counts each, in the workspace)
//   count    work item = (giant, chunk of 2^19 keys, half of the value range): 32-bit counters for 32 768 values in LDS
//            (128 KiB), one add per key of that half, then the non-zero counters are added to the giant's table
//            (<= 32 768 global adds per 2^19 keys read; reading each chunk twice keeps the counters 32 bits wide)
//   scan     one block per giant: exclusive prefixes in place, the total behind them
//   expand   work item = (giant, 2^14 output positions): two probes of the table find the values the range spans, that
//            slice of the prefixes goes to LDS, every position searches it (as many steps as the slice needs: a dense
//            bucket's range spans a handful of values) and is stored coalesced, key map undone
// 2 (or 4) + 4 bytes per key.  route_kernel lists the giants and their work items; at most GIANT_MAX tables exist.
constexpr int GIANT_THREADS = 1024;
constexpr int GIANT_ITEMS_LDS = 4096 + 16;  // first work item of every giant (GIANT_MAX + 1 entries)
constexpr size_t giant_count_lds_bytes() { return 4 * 32768 + 4 * ((size_t)GIANT_ITEMS_LDS) + 64; }

__global__ __launch_bounds__(256) void giant_zero_kernel(const Plan* __restrict__ plan, uint32_t* __restrict__ tables) {
    if (!plan->local_sort || plan->route != ROUTE_HYBRID || plan->giants == 0) return;
    const uint64_t vecs = (uint64_t)plan->giants * GIANT_TABLE / 4;
    uint4* t4 = reinterpret_cast<uint4*>(tables);
    const uint4 z = {0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < vecs; i += (uint64_t)gridDim.x * 256) t4[i] = z;
}

// the giant a work item belongs to: largest g with item0[g] <= it (item0 in LDS, G + 1 entries, item0[G] = all items)
__device__ __forceinline__ uint32_t giant_of_item(const uint32_t* item0, uint32_t G, uint32_t it) {
    uint32_t g = 0;
#pragma unroll
    for (int b = 12; b >= 0; --b) {
        const uint32_t c = g | (1u << b);
        if (c < G && item0[c] <= it) g = c;
    }
    return g;
}

template <bool MAPPED, bool FROM16>
__global__ __launch_bounds__(GIANT_THREADS) void giant_count_kernel(
    const uint32_t* __restrict__ buf_keys, const uint32_t* __restrict__ buf_tmp, const uint16_t* __restrict__ src16, const uint32_t* __restrict__ bstart,
    const Plan* __restrict__ plan, uint32_t neg, uint32_t pos, const uint32_t* __restrict__ glist, const uint32_t* __restrict__ gcount_item,
    uint32_t* __restrict__ tables) {
    constexpr int BLOCK = GIANT_THREADS, U = 8;
    if (!plan->local_sort || plan->route != ROUTE_HYBRID || plan->giants == 0) return;
    const uint32_t* __restrict__ buf = plan->result_in_tmp ? buf_tmp : buf_keys;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* tab = reinterpret_cast<uint32_t*>(smem);                  // [32768]
    uint32_t* item0 = reinterpret_cast<uint32_t*>(smem + 4 * 32768);   // [G + 1]
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t G = plan->giants, items = plan->giant_count_items;
    for (uint32_t i = tid; i <= G; i += BLOCK) item0[i] = gcount_item[i];
    __syncthreads();
    for (uint32_t it = blockIdx.x; it < items; it += gridDim.x) {
        const uint32_t g = giant_of_item(item0, G, it);
        const uint32_t local = it - item0[g], chunk = local >> 1, half = local & 1u;
        const uint32_t bucket = glist[g], start = bstart[bucket], cnt = bstart[bucket + 1] - start;
        const uint32_t ck = giant_chunk_of(cnt, G);
        const bool alone = cnt <= ck;  // the giant is one chunk: its two items store their halves of the table (no zeroing needed, no adds)
        const uint32_t c0 = chunk * ck, c1 = cnt - c0 < ck ? cnt : c0 + ck;
        {
            uint4* t4 = reinterpret_cast<uint4*>(tab);
            const uint4 z = {0, 0, 0, 0};
            for (int i = tid; i < 32768 / 4; i += BLOCK) t4[i] = z;
        }
        __syncthreads();
        if constexpr (FROM16) {
            // eight halves per load (16 bytes per lane, as every streaming kernel here): with 2-byte loads the kernel ran at 2 TB/s
            constexpr int U2 = 4;
            const uint64_t g0 = (uint64_t)start + c0, g1 = (uint64_t)start + c1;  // element range in src16 (16-byte aligned base)
            for (uint64_t base = g0 & ~7ull; base < g1; base += (uint64_t)BLOCK * 8 * U2) {  // wave-uniform trip count
                uint4 q[U2];
#pragma unroll
                for (int j = 0; j < U2; ++j) {
                    const uint64_t p = base + ((uint64_t)j * BLOCK + (uint64_t)tid) * 8;
                    q[j] = p < g1 ? *reinterpret_cast<const uint4*>(src16 + p) : uint4{0, 0, 0, 0};
                }
#pragma unroll
                for (int j = 0; j < U2; ++j) {
                    const uint64_t p = base + ((uint64_t)j * BLOCK + (uint64_t)tid) * 8;
                    const uint32_t w[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
                    // (all but a chunk's first and last wave: every lane's eight halves are the chunk's — no 64-bit bounds per key)
                    auto add = [&](uint32_t x, bool mine) {
                        const uint32_t x0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
                        if (__all((int)(mine && x == x0)) != 0) {
                            if (lane == 0) atomicAdd(&tab[x0 & 0x7FFFu], 64u);
                        } else if (mine) {
                            atomicAdd(&tab[x & 0x7FFFu], 1u);
                        }
                    };
                    if (__all((int)(p >= g0 && p + 8 <= g1)) != 0) {
                        // 64 keys of one value in a wave's add are 64 adds on one address: that happens when the lanes hold the SAME
                        // eight halves (a constant, or a pattern whose period divides eight) — tested once per vector, not per key
                        const uint32_t f[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)w[0]), (uint32_t)__builtin_amdgcn_readfirstlane((int)w[1]),
                                               (uint32_t)__builtin_amdgcn_readfirstlane((int)w[2]), (uint32_t)__builtin_amdgcn_readfirstlane((int)w[3])};
                        if (__all((int)(w[0] == f[0] && w[1] == f[1] && w[2] == f[2] && w[3] == f[3])) != 0) {
                            if (lane == 0) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) {
                                    const uint32_t x = (f[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu;
                                    if ((x >> 15) == half) atomicAdd(&tab[x & 0x7FFFu], 64u);
                                }
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const uint32_t x = (w[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu;
                                if ((x >> 15) == half) atomicAdd(&tab[x & 0x7FFFu], 1u);
                            }
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const uint32_t x = (w[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu;
                            add(x, p + e >= g0 && p + e < g1 && (x >> 15) == half);
                        }
                    }
                }
            }
        } else
        for (uint32_t base = c0; base < c1; base += BLOCK * U) {  // wave-uniform trip count
            uint32_t v[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const uint32_t idx = base + (uint32_t)j * BLOCK + (uint32_t)tid;
                const uint32_t at = idx < c1 ? idx : c1 - 1;
                v[j] = buf[start + at];
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const uint32_t idx = base + (uint32_t)j * BLOCK + (uint32_t)tid;
                const uint32_t x = FROM16 ? v[j] : ((MAPPED ? map_key<uint32_t>(v[j], neg, pos) : v[j]) & 0xFFFFu);
                const bool mine = idx < c1 && (x >> 15) == half;
                const uint32_t x0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
                if (__all((int)(mine && x == x0)) != 0) {  // 64 keys of one value: one add
                    if (lane == 0) atomicAdd(&tab[x0 & 0x7FFFu], 64u);
                } else if (mine) {
                    atomicAdd(&tab[x & 0x7FFFu], 1u);
                }
            }
        }
        __syncthreads();
        uint32_t* T = tables + (size_t)g * GIANT_TABLE + half * 32768u;
        if (alone) {
            for (int i = tid; i < 32768; i += BLOCK) T[i] = tab[i];
        } else {
            for (int i = tid; i < 32768; i += BLOCK) {
                const uint32_t c = tab[i];
                if (c) atomicAdd(&T[i], c);
            }
        }
        __syncthreads();  // the next item clears the table
    }
}

__global__ __launch_bounds__(GIANT_THREADS) void giant_scan_kernel(const Plan* __restrict__ plan, uint32_t* __restrict__ tables) {
    if (!plan->local_sort || plan->route != ROUTE_HYBRID || plan->giants == 0) return;
    __shared__ uint32_t s_wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t g = blockIdx.x; g < plan->giants; g += gridDim.x) {
        uint4* T4 = reinterpret_cast<uint4*>(tables + (size_t)g * GIANT_TABLE) + (size_t)tid * 16;  // my 64 values
        uint32_t c[64];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint4 v = T4[k];
            c[4 * k] = v.x; c[4 * k + 1] = v.y; c[4 * k + 2] = v.z; c[4 * k + 3] = v.w;
        }
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < 64; ++k) run += c[k];
        uint32_t incl = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t below = incl - run;
#pragma unroll
        for (int x = 0; x < GIANT_THREADS / 64; ++x)
            if (x < wave) below += s_wsum[x];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            uint4 v;
            v.x = below; below += c[4 * k];
            v.y = below; below += c[4 * k + 1];
            v.z = below; below += c[4 * k + 2];
            v.w = below; below += c[4 * k + 3];
            T4[k] = v;
        }
        if (tid == GIANT_THREADS - 1) tables[(size_t)g * GIANT_TABLE + H16_BINS] = below;  // == the bucket's length
        __syncthreads();
    }
}

// An expanding work item, written by giant_split_kernel (one thread per item): where its output goes, which table it
// reads, and the values its positions span — the largest v with P[v] <= its first position and the same for its last (two
// 16-step searches of the giant's prefixes, in L2).  A kernel of its own, after the scan: with the searches (or two probes
// by the whole block) at the head of every expanding item the expansion ran at 1.6 TB/s, most of an item's 20 us waiting.
struct GiantItem {
    uint32_t dst;    // element index of the item's first output in the sorted slice
    uint32_t o0;     // its position in the giant
    uint32_t n_out;  // outputs (GIANT_OUT but for a giant's last item)
    uint32_t g;      // the giant (its table)
    uint32_t vlo, vhi;
    uint32_t top;    // the giant's bucket index << 16
    uint32_t pad;
};
static_assert(sizeof(GiantItem) == 32, "two 16-byte loads");

__global__ __launch_bounds__(256) void giant_split_kernel(const Plan* __restrict__ plan, const uint32_t* __restrict__ bstart,
                                                          const uint32_t* __restrict__ glist, const uint32_t* __restrict__ gexp_item,
                                                          const uint32_t* __restrict__ tables, GiantItem* __restrict__ recs) {
    if (!plan->local_sort || plan->route != ROUTE_HYBRID || plan->giants == 0) return;
    const uint32_t G = plan->giants, items = plan->giant_expand_items;
    for (uint32_t it = blockIdx.x * 256u + threadIdx.x; it < items; it += gridDim.x * 256u) {
        const uint32_t g = giant_of_item(gexp_item, G, it);
        const uint32_t j = it - gexp_item[g];
        const uint32_t bucket = glist[g], start = bstart[bucket], cnt = bstart[bucket + 1] - start;
        const uint32_t o0 = j * GIANT_OUT, o1 = cnt - o0 < GIANT_OUT ? cnt : o0 + GIANT_OUT;
        const uint32_t* __restrict__ P = tables + (size_t)g * GIANT_TABLE;
        uint32_t a = 0, b = 0;  // P[0] = 0 <= every position
#pragma unroll
        for (int s = 15; s >= 0; --s) {
            const uint32_t ca = a | (1u << s), cb = b | (1u << s);
            if (P[ca] <= o0) a = ca;
            if (P[cb] <= o1 - 1) b = cb;
        }
        GiantItem r;
        r.dst = start + o0; r.o0 = o0; r.n_out = o1 - o0; r.g = g; r.vlo = a; r.vhi = b; r.top = bucket << 16; r.pad = 0;
        recs[it] = r;
    }
}

// 256 threads and 16 KiB of LDS: eight blocks per CU, so that one item's waits (its record, its slice of the table) hide
// behind seven others' stores.
constexpr int GIANT_XTHREADS = 256;
// Expanding a giant's count table (round 3 form).  An item is GIANT_OUT consecutive output positions of one giant; the value
// at position i is the largest v with P[v] <= i.  Round 2 searched for it per position (a 9-12-step binary search of the item's
// slice of the prefixes in LDS: ~10 LDS reads and ~25 vector instructions per key, 2.0 ms per 10^9 keys of a float column
// against 1.0 for its stores alone).  Values grow with the position, so it is a running maximum instead: every value that has
// keys marks the position its run starts at (clamped to the item's first position: the run the item begins in) with its
// index, and an inclusive max-scan over the item's positions — 16 per thread, then across the wave and the block's four
// waves — spreads the marks.  ~1.3 LDS operations and ~9 vector instructions per key.
template <bool MAPPED>
__global__ __launch_bounds__(GIANT_XTHREADS, 8) void giant_expand_kernel(uint32_t* __restrict__ buf_keys, uint32_t* __restrict__ buf_tmp,
                                                                        const Plan* __restrict__ plan, uint32_t neg, uint32_t pos,
                                                                        const uint32_t* __restrict__ tables, const GiantItem* __restrict__ recs) {
    constexpr int BLOCK = GIANT_XTHREADS, PER = (int)GIANT_OUT / BLOCK;
    static_assert(PER == 16 && BLOCK == 256, "sixteen positions per thread, four waves");
    if (!plan->local_sort || plan->route != ROUTE_HYBRID || plan->giants == 0) return;
    uint32_t* __restrict__ buf = plan->result_in_tmp ? buf_tmp : buf_keys;
    __shared__ __attribute__((aligned(16))) uint32_t s_v[GIANT_OUT];  // per position: 1 + index (from the item's first value) of the value whose run starts there, 0: none
    __shared__ uint32_t s_wave[BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t items = plan->giant_expand_items;
    for (uint32_t it = blockIdx.x; it < items; it += gridDim.x) {
        const GiantItem r = recs[it];  // (uniform: scalar loads)
        const uint32_t* __restrict__ P = tables + (size_t)r.g * GIANT_TABLE;  // P[v] = keys below value v, P[65536] = the giant's length
        const uint32_t o0 = r.o0, o1 = r.o0 + r.n_out;
        uint4* mine = reinterpret_cast<uint4*>(s_v + PER * tid);
#pragma unroll
        for (int q = 0; q < PER / 4; ++q) mine[q] = make_uint4(0, 0, 0, 0);
        __syncthreads();
        // the runs that begin (or are under way) inside the item.  vlo is the value position o0 lies in, so position 0 gets a mark.
        const uint32_t len = r.vhi - r.vlo + 1;
        for (uint32_t k = (uint32_t)tid; k < len; k += BLOCK) {
            const uint32_t p = P[r.vlo + k], q = P[r.vlo + k + 1];
            const uint32_t start = p > o0 ? p : o0;
            if (q > start && start < o1) s_v[start - o0] = k + 1u;
        }
        __syncthreads();
        uint32_t x[PER];
#pragma unroll
        for (int q = 0; q < PER / 4; ++q) {
            const uint4 v = mine[q];
            x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int i = 1; i < PER; ++i) x[i] = x[i] > x[i - 1] ? x[i] : x[i - 1];
        uint32_t incl = x[PER - 1];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o);
            if (lane >= o && y > incl) incl = y;
        }
        if (lane == 63) s_wave[wave] = incl;
        uint32_t carry = __shfl_up(incl, 1);
        if (lane == 0) carry = 0;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w)
            if (w < wave && s_wave[w] > carry) carry = s_wave[w];
#pragma unroll
        for (int q = 0; q < PER / 4; ++q) {
            uint4 v;
            v.x = x[4 * q] > carry ? x[4 * q] : carry; v.y = x[4 * q + 1] > carry ? x[4 * q + 1] : carry;
            v.z = x[4 * q + 2] > carry ? x[4 * q + 2] : carry; v.w = x[4 * q + 3] > carry ? x[4 * q + 3] : carry;
            mine[q] = v;
        }
        __syncthreads();
        uint32_t* __restrict__ out = buf + r.dst;  // indexed by position in the item
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t i = (uint32_t)tid + (uint32_t)j * BLOCK;
            const uint32_t m = r.top | ((r.vlo + s_v[i] - 1u) & 0xFFFFu);
            if (i < r.n_out) out[i] = MAPPED ? unmap_key<uint32_t>(m, neg, pos) : m;
        }
        __syncthreads();  // the next item overwrites s_v
    }
}

// K4 for 8-byte keys: the counting idea with a payload.  Inside a bucket 48 bits remain; the kernel orders the
// bucket by bits [32, 48) with the counting sort of local_count_sort_kernel (same tables, same returning add) and
// then puts the keys that tie on those 16 bits (uniform keys: one in nine, in groups of two or three; a group is
// at most 15 by the counters' width) in order of their low 32 bits by counting, for each, the members of its
// group that precede it — instead of six ranked passes.  Keys stay in registers throughout; LDS holds the
// counters (32 KiB), the per-word prefixes (16 KiB) and the low halves at their slots (64 KiB), and that last
// region then serves as the staging buffer for coalesced stores, half a tile at a time.  A counter that would
// pass 15 sends the bucket to the generic kernel, untouched.
constexpr int WIDE_THREADS = 1024;
constexpr int WIDE_TILE = local_tile(8);   // 16 384
constexpr size_t wide_lds_bytes() { return 32768 + 16384 + 4 * (size_t)WIDE_TILE + 128; }
static_assert(WIDE_TILE % (2 * WIDE_THREADS) == 0, "two output halves of whole rounds");

template <bool MAPPED>
__global__ __launch_bounds__(WIDE_THREADS, 4) void local_wide_sort_kernel(
    uint64_t* __restrict__ buf_keys, uint64_t* __restrict__ buf_tmp, const uint32_t* __restrict__ bstart, const Plan* __restrict__ plan,
    uint32_t* __restrict__ err, uint64_t neg, uint64_t pos, uint32_t* __restrict__ list, uint32_t* __restrict__ list_count) {
    constexpr int BLOCK = WIDE_THREADS, MAXR = WIDE_TILE / BLOCK, WPT = H16_BINS / BLOCK / 8, LOG_VPT = 6;
    constexpr int HALF = WIDE_TILE / 2;
    if (!plan->local_sort) return;
    uint64_t* __restrict__ buf = plan->result_in_tmp ? buf_tmp : buf_keys;
    const uint32_t bucket = blockIdx.x;
    const uint32_t start = bstart[bucket], cnt = bstart[bucket + 1] - start;
    if (cnt <= 1) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (cnt > (uint32_t)WIDE_TILE) {
        if (tid == 0) atomicOr(err, ERR_LOCAL_OVERFLOW);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* cnt4 = reinterpret_cast<uint32_t*>(smem);                              // [WPT][BLOCK] eight 4-bit counters per word
    uint16_t* prefix = reinterpret_cast<uint16_t*>(smem + 32768);                    // [WPT][BLOCK]
    uint32_t* low32 = reinterpret_cast<uint32_t*>(smem + 32768 + 16384);             // [WIDE_TILE] low halves at their slots
    uint64_t* out64 = reinterpret_cast<uint64_t*>(smem + 32768 + 16384);             // [HALF] output staging (same space, later)
    uint32_t* s_wsum = reinterpret_cast<uint32_t*>(smem + 32768 + 16384 + 4 * WIDE_TILE);  // [16] wave sums, [16] overflow flag
    __builtin_amdgcn_s_setprio(RDST_PRIO_LOAD);
    const uint64_t* tsrc = buf + start;
    uint64_t mk[MAXR];
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const uint32_t idx = (uint32_t)tid + i * BLOCK;
        mk[i] = tsrc[idx < cnt ? idx : cnt - 1];
    }
#pragma unroll
    for (int k = 0; k < WPT; ++k) cnt4[k * BLOCK + tid] = 0;
    if (tid == 0) s_wsum[16] = 0;
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);
    auto word_of = [](uint32_t v) -> uint32_t { return ((v >> 3) & (uint32_t)(WPT - 1)) * BLOCK + (v >> LOG_VPT); };
    // Per-key state lives in the key's own top 16 bits (inside a bucket they are the bucket index, put back at the
    // end): first the index among equal values, then the slot.  (A second register array for it made the compiler
    // spill the keys themselves: 146 dwords of scratch per lane.)
    constexpr uint64_t LOW48 = (1ull << 48) - 1;
    bool over = false;
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const uint32_t idx = (uint32_t)tid + i * BLOCK;
        if (idx < cnt) {
            if constexpr (MAPPED) mk[i] = map_key<uint64_t>(mk[i], neg, pos);
            const uint32_t v = (uint32_t)(mk[i] >> 32) & 0xFFFFu;
            const uint32_t sh = (v & 7u) * 4u;
            const uint32_t old = atomicAdd(&cnt4[word_of(v)], 1u << sh);
            const uint32_t mine = (old >> sh) & 15u;
            over |= mine == 15u;
            mk[i] = (mk[i] & LOW48) | ((uint64_t)mine << 48);
        }
    }
    if (over) s_wsum[16] = 1;
    __syncthreads();
    if (s_wsum[16]) {  // block-uniform: the bucket stays as it is, for the generic kernel
        if (tid == 0) list[atomicAdd(list_count, 1u)] = bucket;
        return;
    }
    {
        uint32_t pre[WPT];
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            pre[k] = run;
            run = nibble_sum(cnt4[k * BLOCK + tid], run);
        }
        uint32_t incl = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t base = incl - run;
#pragma unroll
        for (int x = 0; x < BLOCK / 64; ++x)
            if (x < wave) base += s_wsum[x];
#pragma unroll
        for (int k = 0; k < WPT; ++k) prefix[k * BLOCK + tid] = (uint16_t)(base + pre[k]);
    }
    __syncthreads();
    // slot by (value, index): the low half goes there, so that the members of a group can look at each other
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const uint32_t idx = (uint32_t)tid + i * BLOCK;
        if (idx < cnt) {
            const uint32_t v = (uint32_t)(mk[i] >> 32) & 0xFFFFu, mine = (uint32_t)(mk[i] >> 48);
            const uint32_t wd = word_of(v), sh = (v & 7u) * 4u;
            const uint32_t slot = nibble_sum(cnt4[wd] & ((1u << sh) - 1u), (uint32_t)prefix[wd] + mine);
            low32[slot] = (uint32_t)mk[i];
            mk[i] = (mk[i] & LOW48) | ((uint64_t)slot << 48);
        }
    }
    __syncthreads();
    // ties: my place inside my group = members with a smaller low half (equal ones: by slot, so places stay distinct)
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const uint32_t idx = (uint32_t)tid + i * BLOCK;
        if (idx < cnt) {
            const uint32_t v = (uint32_t)(mk[i] >> 32) & 0xFFFFu, slot = (uint32_t)(mk[i] >> 48);
            const uint32_t wd = word_of(v), sh = (v & 7u) * 4u;
            const uint32_t w = cnt4[wd];
            const uint32_t group = (w >> sh) & 15u;
            if (group >= 2u) {
                const uint32_t first = nibble_sum(w & ((1u << sh) - 1u), (uint32_t)prefix[wd]);
                const uint32_t lo = (uint32_t)mk[i];
                uint32_t rank = 0;
                for (uint32_t j = 0; j < group; ++j) {
                    const uint32_t other = low32[first + j];
                    rank += (other < lo || (other == lo && first + j < slot)) ? 1u : 0u;
                }
                mk[i] = (mk[i] & LOW48) | ((uint64_t)(first + rank) << 48);
            }
        }
    }
    __syncthreads();  // every look at the low halves is done: their space becomes the output staging
    uint64_t* tdst = buf + start;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h == 1 && cnt <= (uint32_t)HALF) break;  // block-uniform
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const uint32_t idx = (uint32_t)tid + i * BLOCK;
            const uint32_t rel = (uint32_t)(mk[i] >> 48) - (uint32_t)(h * HALF);
            if (idx < cnt && rel < (uint32_t)HALF) out64[rel] = (mk[i] & LOW48) | ((uint64_t)bucket_prefix16(plan, bucket) << 48);
        }
        __syncthreads();
        __builtin_amdgcn_s_setprio(RDST_PRIO_SCATTER);
#pragma unroll
        for (int i = 0; i < HALF / BLOCK; ++i) {
            const uint32_t rel = (uint32_t)tid + i * BLOCK, at = rel + (uint32_t)(h * HALF);
            if (at < cnt) tdst[at] = MAPPED ? unmap_key<uint64_t>(out64[rel], neg, pos) : out64[rel];
        }
        if (h == 0) __syncthreads();  // the second half reuses the staging
    }
}



// K4 for 8-byte keys, two workgroups per CU.  As local_wide_sort_kernel above, made to fit twice: the prefix table per PAIR
// of counter words (8 KiB; a key in the odd word adds the even word's nibbles),
// and for the tie order only bits [16, 32) of the low half in LDS (32 KiB instead of 64): 72 KiB per block.  Two members of
// a group that agree on those 16 bits as well cannot be ordered from what is staged — about one bucket in seventy on uniform
// keys — and such a bucket is left untouched for the generic kernel, like one whose counters overflow.
constexpr int WIDE2_THREADS = 1024;  // 16 keys per thread, 64 VGPRs: two blocks = 32 waves per CU (512 threads x 32 keys: 128 VGPRs, 16 waves: 5.29 ms)
constexpr size_t wide2_lds_bytes() { return 32768 + 8192 + 2 * (size_t)local_tile(8) + 128; }

template <bool MAPPED>
__global__ __launch_bounds__(WIDE2_THREADS, 8) void local_wide2_sort_kernel(
    uint64_t* __restrict__ buf_keys, uint64_t* __restrict__ buf_tmp, const uint32_t* __restrict__ bstart, const Plan* __restrict__ plan,
    uint32_t* __restrict__ err, uint64_t neg, uint64_t pos, uint32_t* __restrict__ list, uint32_t* __restrict__ list_count,
    const uint64_t* __restrict__ src_slots /* ROUTE_ATOMIC: bucket b's keys lie in slot b (slot_cap keys) and number slot_count[b]; NULL: in place */,
    const uint32_t* __restrict__ slot_count, uint32_t slot_cap) {
    constexpr int TILE = local_tile(8);
    constexpr int BLOCK = WIDE2_THREADS, MAXR = TILE / BLOCK, WPT = H16_BINS / BLOCK / 8, LOG_VPT = 6;
    constexpr int HALF = TILE / 2;
    static_assert((size_t)HALF * 8 <= 32768 + 8192 + 2 * (size_t)TILE, "output staging fits the dead tables");
    if (!plan->local_sort) return;
    if (plan->route != ROUTE_ATOMIC) src_slots = nullptr;  // the hybrid route's buckets lie at their final place
    uint64_t* __restrict__ buf = plan->result_in_tmp ? buf_tmp : buf_keys;
    const uint32_t bucket = blockIdx.x;
    const uint32_t start = bstart[bucket], cnt = src_slots ? slot_count[bucket] : bstart[bucket + 1] - start;
    if (cnt == 0 || (cnt == 1 && !src_slots)) return;  // (a single key in a slot still has to be moved to its place)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (cnt > (uint32_t)TILE) {
        if (tid == 0) atomicOr(err, ERR_LOCAL_OVERFLOW);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* cnt4 = reinterpret_cast<uint32_t*>(smem);                              // [WPT][BLOCK] eight 4-bit counters per word
    uint16_t* prefix2 = reinterpret_cast<uint16_t*>(smem + 32768);                   // [WPT / 2][BLOCK] keys below the word pair
    uint16_t* mid16 = reinterpret_cast<uint16_t*>(smem + 32768 + 8192);              // [TILE] bits [16, 32) of the keys at their slots
    uint64_t* out64 = reinterpret_cast<uint64_t*>(smem);                             // [HALF] output staging (everything above is dead by then)
    uint32_t* s_wsum = reinterpret_cast<uint32_t*>(smem + 32768 + 8192 + 2 * TILE);  // [8] wave sums, [16] overflow / ambiguity flag
    RDST_TL_BEGIN(1);
    __builtin_amdgcn_s_setprio(RDST_PRIO_LOAD);
    const uint64_t* tsrc = src_slots ? src_slots + (uint64_t)bucket * slot_cap : buf + start;
    uint64_t mk[MAXR];
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const uint32_t idx = (uint32_t)tid + i * BLOCK;
        mk[i] = tsrc[idx < cnt ? idx : cnt - 1];
    }
#pragma unroll
    for (int k = 0; k < WPT; ++k) cnt4[k * BLOCK + tid] = 0;
    if (tid == 0) s_wsum[16] = 0;
    RDST_STAMP(1);
    __syncthreads();
    RDST_STAMP(2);
    __builtin_amdgcn_s_setprio(0);
    auto word_of = [](uint32_t v) -> uint32_t { return ((v >> 3) & (uint32_t)(WPT - 1)) * BLOCK + (v >> LOG_VPT); };
    constexpr uint64_t LOW48 = (1ull << 48) - 1;  // per-key state rides in the key's top 16 bits (the bucket index, restored at the end)
    bool flag = false;
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const uint32_t idx = (uint32_t)tid + i * BLOCK;
        if (idx < cnt) {
            if constexpr (MAPPED) mk[i] = map_key<uint64_t>(mk[i], neg, pos);
            const uint32_t v = (uint32_t)(mk[i] >> 32) & 0xFFFFu;
            const uint32_t sh = (v & 7u) * 4u;
            const uint32_t old = atomicAdd(&cnt4[word_of(v)], 1u << sh);
            const uint32_t mine = (old >> sh) & 15u;
            flag |= mine == 15u;
            mk[i] = (mk[i] & LOW48) | ((uint64_t)mine << 48);
        }
    }
    if (flag) s_wsum[16] = 1;
    RDST_STAMP(3);
    __syncthreads();
    if (s_wsum[16]) {  // block-uniform: the bucket stays as it is, for the generic kernel
        if (tid == 0) list[atomicAdd(list_count, 1u)] = bucket;
        return;
    }
    {
        uint32_t pre[WPT / 2];
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            if ((k & 1) == 0) pre[k >> 1] = run;
            run = nibble_sum(cnt4[k * BLOCK + tid], run);
        }
        uint32_t incl = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t base = incl - run;
#pragma unroll
        for (int x = 0; x < BLOCK / 64; ++x)
            if (x < wave) base += s_wsum[x];
#pragma unroll
        for (int k = 0; k < WPT / 2; ++k) prefix2[k * BLOCK + tid] = (uint16_t)(base + pre[k]);
    }
    __syncthreads();
    RDST_STAMP(4);
    // first slot of a value: the pair's prefix, the even word's nibbles if the value sits in the odd word, the nibbles below it
    auto first_slot = [&](uint32_t v, uint32_t w, uint32_t wd) -> uint32_t {
        const uint32_t k = (v >> 3) & (uint32_t)(WPT - 1);
        uint32_t at = prefix2[(k >> 1) * BLOCK + (v >> LOG_VPT)];
        if (k & 1u) at = nibble_sum(cnt4[wd - BLOCK], at);
        return nibble_sum(w & ((1u << ((v & 7u) * 4u)) - 1u), at);
    };
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const uint32_t idx = (uint32_t)tid + i * BLOCK;
        if (idx < cnt) {
            const uint32_t v = (uint32_t)(mk[i] >> 32) & 0xFFFFu, mine = (uint32_t)(mk[i] >> 48);
            const uint32_t wd = word_of(v);
            const uint32_t slot = first_slot(v, cnt4[wd], wd) + mine;
            mid16[slot] = (uint16_t)((uint32_t)mk[i] >> 16);
            mk[i] = (mk[i] & LOW48) | ((uint64_t)slot << 48);
        }
    }
    RDST_STAMP(5);
    __syncthreads();
    // ties: my place inside my group = members with smaller bits [16, 32) (equal ones would need the low 16 bits: give up)
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const uint32_t idx = (uint32_t)tid + i * BLOCK;
        if (idx < cnt) {
            const uint32_t v = (uint32_t)(mk[i] >> 32) & 0xFFFFu, slot = (uint32_t)(mk[i] >> 48);
            const uint32_t wd = word_of(v);
            const uint32_t w = cnt4[wd];
            const uint32_t group = (w >> ((v & 7u) * 4u)) & 15u;
            if (group >= 2u) {
                const uint32_t first = first_slot(v, w, wd);
                const uint32_t mid = (uint32_t)mk[i] >> 16;
                uint32_t rank = 0;
                for (uint32_t j = 0; j < group; ++j) {
                    const uint32_t other = mid16[first + j];
                    rank += other < mid ? 1u : 0u;
                    flag |= other == mid && first + j != slot;
                }
                mk[i] = (mk[i] & LOW48) | ((uint64_t)(first + rank) << 48);
            }
        }
    }
    if (flag) s_wsum[16] = 1;
    RDST_STAMP(6);
    __syncthreads();  // every look at the tables and the staged bits is done: their space becomes the output staging
    if (s_wsum[16]) {
        if (tid == 0) list[atomicAdd(list_count, 1u)] = bucket;
        return;
    }
    uint64_t* tdst = buf + start;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h == 1 && cnt <= (uint32_t)HALF) break;  // block-uniform
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const uint32_t idx = (uint32_t)tid + i * BLOCK;
            const uint32_t rel = (uint32_t)(mk[i] >> 48) - (uint32_t)(h * HALF);
            if (idx < cnt && rel < (uint32_t)HALF) out64[rel] = (mk[i] & LOW48) | ((uint64_t)bucket_prefix16(plan, bucket) << 48);
        }
        __syncthreads();
        if (h == 0) RDST_STAMP(7);
        __builtin_amdgcn_s_setprio(RDST_PRIO_SCATTER);
#pragma unroll
        for (int i = 0; i < HALF / BLOCK; ++i) {
            const uint32_t rel = (uint32_t)tid + i * BLOCK, at = rel + (uint32_t)(h * HALF);
            if (at < cnt) tdst[at] = MAPPED ? unmap_key<uint64_t>(out64[rel], neg, pos) : out64[rel];
        }
        if (h == 0) __syncthreads();  // the second half reuses the staging
        if (h == 0) RDST_STAMP(8);
    }
    RDST_STAMP(9);
    RDST_TL_END(bucket);
}

// K4 for 8-byte keys, third form: local_wide2_sort_kernel with a third fewer LDS instructions.  By the kernel's timeline
// (tools/timeline2.py, round 3) a bucket's 31.8 us were 25 us of LDS-bound phases — ties 8.8, staging 6, scan 3.8, place 3.0,
// count 2.9 — against 4 us of loads and 3 of stores: two blocks per CU kept the ONE LDS busy, not the HBM.  Changes:
//  * a prefix per counter word again (16 KiB instead of 8: the block takes exactly half of the CU's 160 KiB, the wave sums and
//    the flag live in corners of the staged-bits array that are free when they are needed) — a key in an odd word no longer
//    reads the even word as well, in the place phase and in the tie phase;
//  * a key's index among equals and the size of its group are kept in two packed register pairs (4 bits x 16 keys each)
//    instead of being re-read: the tie phase reads nothing but the staged bits of the group's members, and only for keys
//    that have company (one in five on uniform keys);
//  * two members of a group that agree on bits [16, 32) as well are settled by a small list instead of sending the bucket to the
//    generic kernel (below).
// (A fourth form — predicate-free rounds for full buckets as in count_sort_bucket, one 8-byte LDS read per tie group — was
// correct and, for unsigned keys, 40 % SLOWER: at the 64 registers two blocks per CU allow it spilled a dozen values per thread
// per phase, 100 KB of scratch traffic per 244-KB bucket.  Signed and float keys, whose map keeps fewer values live, ran as
// fast as this form, not faster.)
constexpr int WIDE3_THREADS = 1024;
constexpr uint32_t WIDE3_AMB_MAX = 256;  // entries of 8 bytes in the (16-KiB) prefix table
constexpr size_t wide3_lds_bytes() { return 32768 + 16384 + 2 * (size_t)local_tile(8); }  // 81 920 = 160 KiB / 2
static_assert(wide3_lds_bytes() * 2 <= 160 * 1024, "two blocks per CU");

template <bool MAPPED>
__global__ __launch_bounds__(WIDE3_THREADS, 8) void local_wide3_sort_kernel(
    uint64_t* __restrict__ buf_keys, uint64_t* __restrict__ buf_tmp, const uint32_t* __restrict__ bstart, const Plan* __restrict__ plan,
    uint32_t* __restrict__ err, uint64_t neg, uint64_t pos, uint32_t* __restrict__ list, uint32_t* __restrict__ list_count,
    const uint64_t* __restrict__ src_slots /* ROUTE_ATOMIC: bucket b's keys lie in slot b (slot_cap keys) and number slot_count[b]; NULL: in place */,
    const uint32_t* __restrict__ slot_count, uint32_t slot_cap) {
    constexpr int TILE = local_tile(8);
    constexpr int BLOCK = WIDE3_THREADS, MAXR = TILE / BLOCK, WPT = H16_BINS / BLOCK / 8, LOG_VPT = 6;
    constexpr int HALF = TILE / 2;
    static_assert(MAXR == 16, "two 64-bit registers of 4-bit fields, one field per key");
    static_assert((size_t)HALF * 8 <= 32768 + 16384 + 2 * (size_t)TILE - 4, "output staging fits below the flag word");
    if (!plan->local_sort) return;
    if (plan->route != ROUTE_ATOMIC) src_slots = nullptr;  // the hybrid route's buckets lie at their final place
    uint64_t* __restrict__ buf = plan->result_in_tmp ? buf_tmp : buf_keys;
    const uint32_t bucket = blockIdx.x;
    const uint32_t start = bstart[bucket], cnt = src_slots ? slot_count[bucket] : bstart[bucket + 1] - start;
    if (cnt == 0 || (cnt == 1 && !src_slots)) return;  // (a single key in a slot still has to be moved to its place)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (cnt > (uint32_t)TILE) {
        if (tid == 0) atomicOr(err, ERR_LOCAL_OVERFLOW);
        return;
    }
    if (cnt > (uint32_t)TILE - 2u) {  // the last two entries of the staged bits are the flag word: such a bucket goes to the generic kernel
        if (tid == 0) list[atomicAdd(list_count, 1u)] = bucket;
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* cnt4 = reinterpret_cast<uint32_t*>(smem);                              // [WPT][BLOCK] eight 4-bit counters per word
    uint16_t* prefix = reinterpret_cast<uint16_t*>(smem + 32768);                    // [WPT][BLOCK] keys below the word
    uint16_t* mid16 = reinterpret_cast<uint16_t*>(smem + 32768 + 16384);             // [TILE] bits [16, 32) of the keys at their slots
    uint32_t* s_wsum = reinterpret_cast<uint32_t*>(smem + 32768 + 16384);            // [16] wave sums (scan phase: the staged bits are not there yet)
    uint32_t* s_flag = reinterpret_cast<uint32_t*>(smem + 32768 + 16384 + 2 * TILE - 4);  // count phase: overflow; tie phase: length of the list (never a slot: cnt <= TILE - 2)
    uint2* amb = reinterpret_cast<uint2*>(smem + 32768);                             // [WIDE3_AMB_MAX] keys with an equal in bits [16, 48) (tie phase: the prefixes are dead)
    uint64_t* out64 = reinterpret_cast<uint64_t*>(smem);                             // [HALF] output staging (the tables are dead by then)
    RDST_TL_BEGIN(1);
    __builtin_amdgcn_s_setprio(RDST_PRIO_LOAD);
    const uint64_t* tsrc = src_slots ? src_slots + (uint64_t)bucket * slot_cap : buf + start;
    uint64_t mk[MAXR];
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const uint32_t idx = (uint32_t)tid + i * BLOCK;
        mk[i] = tsrc[idx < cnt ? idx : cnt - 1];
    }
#pragma unroll
    for (int k = 0; k < WPT; ++k) cnt4[k * BLOCK + tid] = 0;
    if (tid == 0) *s_flag = 0;
    RDST_STAMP(1);
    __syncthreads();
    RDST_STAMP(2);
    __builtin_amdgcn_s_setprio(0);
    auto word_of = [](uint32_t v) -> uint32_t { return v >> 3; };  // (linear: consecutive values lie on consecutive banks — dense ids put 128 of them on ONE bank in the thread-major table —, and the address is two instructions instead of five; the scan pays with a rotated read order)
    constexpr uint64_t LOW48 = (1ull << 48) - 1;  // a key's slot rides in its top 16 bits (the bucket index, restored at the end)
    uint64_t mine_pack = 0, group_pack = 0;       // per key: index among the keys of its value, size of that group
    bool flag = false;
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const uint32_t idx = (uint32_t)tid + i * BLOCK;
        if (idx < cnt) {
            if constexpr (MAPPED) mk[i] = map_key<uint64_t>(mk[i], neg, pos);
            const uint32_t v = (uint32_t)(mk[i] >> 32) & 0xFFFFu;
            const uint32_t sh = (v & 7u) * 4u;
            const uint32_t old = atomicAdd(&cnt4[word_of(v)], 1u << sh);
            const uint32_t mine = (old >> sh) & 15u;
            flag |= mine == 15u;
            mine_pack |= (uint64_t)mine << (4 * i);
        }
    }
    if (flag) *s_flag = 1;
    RDST_STAMP(3);
    __syncthreads();
    if (*s_flag) {  // block-uniform: the bucket stays as it is, for the generic kernel
        if (tid == 0) list[atomicAdd(list_count, 1u)] = bucket;
        return;
    }
    {
        // thread t owns words [WPT t, WPT t + WPT) and reads them in a rotated order — word (k + r) mod WPT in step k, r = t / (64 / WPT)
        // — so that the 64 lanes of a step sit on 64 different banks; the prefixes are put right afterwards: the words from r up
        // arrive first (before them lie the words below r: total - the sum through the last word), the words below r after those
        constexpr int ROT_SHIFT = WPT == 16 ? 2 : 3;
        static_assert(WPT == 16 || WPT == 8, "64 banks / WPT lanes per rotation");
        const uint32_t r = ((uint32_t)tid >> ROT_SHIFT) & (uint32_t)(WPT - 1);
        uint32_t arr[WPT];  // keys in the words that arrived before step k
        uint32_t run = 0, through_last = 0;
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            const uint32_t j = ((uint32_t)k + r) & (uint32_t)(WPT - 1);
            arr[k] = run;
            run = nibble_sum(cnt4[WPT * tid + j], run);
            through_last = j == (uint32_t)(WPT - 1) ? run : through_last;
        }
        const uint32_t below_r = run - through_last;  // keys in my words 0 .. r - 1
        uint32_t incl = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t base = incl - run;
#pragma unroll
        for (int x = 0; x < BLOCK / 64; ++x)
            if (x < wave) base += s_wsum[x];
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            const uint32_t j = ((uint32_t)k + r) & (uint32_t)(WPT - 1);
            const uint32_t before = (uint32_t)k + r < (uint32_t)WPT ? below_r + arr[k] : arr[k] - through_last;
            prefix[WPT * tid + j] = (uint16_t)(base + before);
        }
    }
    __syncthreads();  // (also: every read of the wave sums is done before the staged bits overwrite them)
    RDST_STAMP(4);
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const uint32_t idx = (uint32_t)tid + i * BLOCK;
        if (idx < cnt) {
            const uint32_t v = (uint32_t)(mk[i] >> 32) & 0xFFFFu;
            const uint32_t wd = word_of(v), sh = (v & 7u) * 4u;
            const uint32_t w = cnt4[wd];
            const uint32_t first = nibble_sum(w & ((1u << sh) - 1u), (uint32_t)prefix[wd]);
            const uint32_t slot = first + ((uint32_t)(mine_pack >> (4 * i)) & 15u);
            group_pack |= (uint64_t)((w >> sh) & 15u) << (4 * i);
            mid16[slot] = (uint16_t)((uint32_t)mk[i] >> 16);
            mk[i] = (mk[i] & LOW48) | ((uint64_t)slot << 48);
        }
    }
    RDST_STAMP(5);
    __syncthreads();
    // ties: my place inside my group = first slot + members with smaller bits [16, 32).  A key with a member that agrees on those
    // bits as well (one bucket in seventy holds such a pair on uniform keys: 0.28 of K4's 4.6 ms went to the generic kernel for
    // them) keeps its slot for now, clears its group field — valid keys have at least 1 there — and enters a list in the prefix
    // table, dead by now: (group's first slot, staged bits | low 16 bits, slot).  The list's length is counted in the flag word,
    // which the count phase left at 0.
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const uint32_t group = (uint32_t)(group_pack >> (4 * i)) & 15u;  // (0 for the slots past cnt)
        if (group >= 2u) {
            const uint32_t slot = (uint32_t)(mk[i] >> 48);
            const uint32_t first = slot - ((uint32_t)(mine_pack >> (4 * i)) & 15u);
            const uint32_t mid = ((uint32_t)mk[i] >> 16) & 0xFFFFu;
            uint32_t rank = 0;
            bool twin = false;
            for (uint32_t j = 0; j < group; ++j) {
                const uint32_t other = mid16[first + j];
                rank += other < mid ? 1u : 0u;
                twin |= other == mid && first + j != slot;
            }
            if (twin) {
                group_pack &= ~(15ull << (4 * i));
                const uint32_t e = atomicAdd(s_flag, 1u);
                if (e < WIDE3_AMB_MAX) amb[e] = make_uint2((first << 16) | mid, (((uint32_t)mk[i] & 0xFFFFu) << 16) | slot);
            } else {
                mk[i] = (mk[i] & LOW48) | ((uint64_t)(first + rank) << 48);
            }
        }
    }
    RDST_STAMP(6);
    __syncthreads();  // every look at the staged bits is done, the list is complete
    const uint32_t namb = *s_flag;
    if (namb > WIDE3_AMB_MAX) {  // block-uniform: heavily repeated keys go to the generic kernel after all
        if (tid == 0) list[atomicAdd(list_count, 1u)] = bucket;
        return;
    }
    if (namb) {  // block-uniform and rare (a pair of entries in one bucket of seventy): nothing in here has to be fast
#pragma unroll 1
        for (int i = 0; i < MAXR; ++i) {
            // a loop, not sixteen copies; the keys of the list are found by their cleared group field, and register i of the key
            // array is reached through selects
            if ((uint32_t)tid + (uint32_t)i * BLOCK >= cnt || ((uint32_t)(group_pack >> (4 * i)) & 15u) != 0u) continue;
            uint64_t key = 0;
#pragma unroll
            for (int q = 0; q < MAXR; ++q) key = q == i ? mk[q] : key;
            const uint32_t slot = (uint32_t)(key >> 48);
            const uint32_t v = (uint32_t)(key >> 32) & 0xFFFFu;
            const uint32_t group = (cnt4[word_of(v)] >> ((v & 7u) * 4u)) & 15u;  // (the counters are intact until the output is staged)
            const uint32_t first = slot - ((uint32_t)(mine_pack >> (4 * i)) & 15u);
            const uint32_t mid = ((uint32_t)key >> 16) & 0xFFFFu;
            uint32_t rank = 0;
            for (uint32_t j = 0; j < group; ++j) rank += mid16[first + j] < mid ? 1u : 0u;
            const uint32_t me_x = (first << 16) | mid, me_y = (((uint32_t)key & 0xFFFFu) << 16) | slot;
            uint32_t before = 0;  // entries of my group and staged bits that go before me: smaller low bits, or the same and an earlier slot
            for (uint32_t e = 0; e < namb; ++e) {
                const uint2 x = amb[e];
                before += (x.x == me_x && x.y < me_y) ? 1u : 0u;
            }
            key = (key & LOW48) | ((uint64_t)(first + rank + before) << 48);
#pragma unroll
            for (int q = 0; q < MAXR; ++q) mk[q] = q == i ? key : mk[q];
        }
        __syncthreads();  // the list lies where the output staging begins
    }
    uint64_t* tdst = buf + start;
    const uint64_t top = (uint64_t)bucket_prefix16(plan, bucket) << 48;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h == 1 && cnt <= (uint32_t)HALF) break;  // block-uniform
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const uint32_t idx = (uint32_t)tid + i * BLOCK;
            const uint32_t rel = (uint32_t)(mk[i] >> 48) - (uint32_t)(h * HALF);
            if (idx < cnt && rel < (uint32_t)HALF) out64[rel] = (mk[i] & LOW48) | top;
        }
        __syncthreads();
        if (h == 0) RDST_STAMP(7);
        __builtin_amdgcn_s_setprio(RDST_PRIO_SCATTER);
#pragma unroll
        for (int i = 0; i < HALF / BLOCK; ++i) {
            const uint32_t rel = (uint32_t)tid + i * BLOCK, at = rel + (uint32_t)(h * HALF);
            if (at < cnt) tdst[at] = MAPPED ? unmap_key<uint64_t>(out64[rel], neg, pos) : out64[rel];
        }
        if (h == 0) __syncthreads();  // the second half reuses the staging
        if (h == 0) RDST_STAMP(8);
    }
    RDST_STAMP(9);
    RDST_TL_END(bucket);
}

// result sits in tmp after an odd number of executed passes: copy back
// (src/sorts/lsb_sort.rs:117-126)
// (want_tmp: the other way round — `keys` is the caller's tmp, `tmp` its keys, and the copy runs when the result is NOT in tmp)
template <typename K, int VEC>
__global__ __launch_bounds__(256) void copyback_kernel(K* __restrict__ keys, const K* __restrict__ tmp,
                                                       uint64_t n, const Plan* __restrict__ plan, uint32_t want_tmp) {
    if ((plan->result_in_tmp != 0u) == (want_tmp != 0u)) return;
    struct alignas(sizeof(K) * VEC) V { K e[VEC]; };
    const uint64_t nvec = n / VEC;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride)
        reinterpret_cast<V*>(keys)[i] = reinterpret_cast<const V*>(tmp)[i];
    if (blockIdx.x == 0) {
        const uint64_t i = nvec * VEC + threadIdx.x;
        if (i < n) keys[i] = tmp[i];
    }
}

// [u8; N] keys (src/radix_key_impl.rs:78-85: level l reads byte N-1-l, i.e. lexicographic order): the
// N bytes become the low N bytes of an unsigned W-byte integer, first byte most significant; the
// integer sort then skips the W-N constant top levels by itself.  And back.
template <typename K>
__global__ __launch_bounds__(256) void bytes_expand_kernel(const unsigned char* __restrict__ raw, K* __restrict__ out, uint64_t n, uint32_t nb) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const unsigned char* p = raw + i * nb;
        K v = 0;
        for (uint32_t j = 0; j < nb; ++j) v = (K)(v << 8) | (K)p[j];
        out[i] = v;
    }
}
template <typename K>
__global__ __launch_bounds__(256) void bytes_compact_kernel(const K* __restrict__ in, unsigned char* __restrict__ raw, uint64_t n, uint32_t nb) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        unsigned char* p = raw + i * nb;
        K v = in[i];
        for (uint32_t j = nb; j-- > 0;) { p[j] = (unsigned char)v; v >>= 8; }
    }
}

// Records with a built-in key field: (key, row index) pairs out of the rows, and
// the rows back in the order of the sorted indices.  UNIT = widest word the row size allows.
template <typename K, typename I>
__global__ __launch_bounds__(256) void extract_key_kernel(const unsigned char* __restrict__ rec, uint64_t n, uint32_t rbytes,
                                                          uint32_t key_off, K* __restrict__ keys, I* __restrict__ idx) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        keys[i] = *reinterpret_cast<const K*>(rec + i * rbytes + key_off);
        idx[i] = (I)i;
    }
}
template <typename UNIT, typename I>
__global__ __launch_bounds__(256) void gather_records_kernel(const UNIT* __restrict__ rec, UNIT* __restrict__ out,
                                                             const I* __restrict__ idx, uint64_t n, uint32_t units) {
    const uint64_t total = n * units, stride = (uint64_t)gridDim.x * 256;
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
        const uint64_t i = g / units;
        const uint32_t w = (uint32_t)(g - i * units);
        out[g] = rec[(uint64_t)idx[i] * units + w];
    }
}

// Streaming yardsticks for the bench (rdst_hip_stream_copy / _read / _fill): what this HBM delivers to the simplest kernels
// there are, beside the 8 TB/s spec.  The shape is the fastest of tools/probe/copy_sweep.hip's sweep (profiles/r03_copy_sweep.json):
// 16 bytes per lane, ONE CONTIGUOUS PIECE PER BLOCK (a grid-stride sweep loses 10-15 % to first-level TLB misses: every 4 KiB step
// of a block lands in another 2 MiB fragment), eight vectors in flight, non-temporal loads (reads: 7.0 instead of 6.1 TB/s).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
template <int MODE /* 0 copy, 1 read, 2 fill */>
__global__ __launch_bounds__(256) void stream_kernel(u32x4_t* __restrict__ dst, const u32x4_t* __restrict__ src, uint64_t nvec, uint32_t* __restrict__ sink) {
    constexpr int U = 8;
    const uint64_t piece = ((nvec + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    uint64_t i = (uint64_t)blockIdx.x * piece + threadIdx.x;
    const uint64_t end = (uint64_t)(blockIdx.x + 1) * piece < nvec ? (uint64_t)(blockIdx.x + 1) * piece : nvec;
    u32x4_t acc = {0, 0, 0, 0};
    for (; i + (U - 1) * 256 < end; i += U * 256) {
        u32x4_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = MODE == 2 ? (u32x4_t){(uint32_t)i, 1u, 2u, 3u} : __builtin_nontemporal_load(src + i + u * 256);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE == 0) __builtin_nontemporal_store(v[u], dst + i + u * 256);
            else if (MODE == 2) dst[i + u * 256] = v[u];
            else acc ^= v[u];
        }
    }
    for (; i < end; i += 256) {
        const u32x4_t v = MODE == 2 ? (u32x4_t){(uint32_t)i, 1u, 2u, 3u} : __builtin_nontemporal_load(src + i);
        if (MODE == 1) acc ^= v;
        else dst[i] = v;
    }
    if (MODE == 1 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) *sink = acc.x;  // keeps the loads alive; practically never taken
}

// Status rows of the levels only the LSD route uses: cleared after the route decision, and only if it fell that way
// (`unless_atomic`: behind a failed atomic route the hybrid route's two passes need their rows too: clear unless the atomic route took the sort)
__global__ __launch_bounds__(256) void clear_unless_hybrid_kernel(const Plan* __restrict__ plan, uint4* __restrict__ a, uint64_t na, uint4* __restrict__ b,
                                                                  uint64_t nb, uint32_t unless_atomic) {
    if (unless_atomic ? plan->route == ROUTE_ATOMIC : plan->route != ROUTE_LSD) return;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    const uint4 z = {0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < na; i += stride) a[i] = z;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += stride) b[i] = z;
}

// Sharded route, skewed top byte: the shard is ordered by the top 16 bits of the mapped key (two stable passes);
// bucket b's length = (index after its last key) - (index of its first).  Every boundary between neighbours of
// different buckets adds its index to the bucket that ends there and subtracts it from the one that begins; the
// slice's end adds n to the last bucket.  `counts` [65536] must be zero.
template <typename K>
__global__ __launch_bounds__(256) void top16_counts_kernel(const K* __restrict__ keys, uint64_t n, K neg, K pos, unsigned long long* __restrict__ counts) {
    constexpr int W = sizeof(K) * 8;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const uint32_t p = (uint32_t)(map_key<K>(keys[i], neg, pos) >> (W - 16));
        if (i + 1 == n) {
            atomicAdd(&counts[p], (unsigned long long)n);
        } else {
            const uint32_t q = (uint32_t)(map_key<K>(keys[i + 1], neg, pos) >> (W - 16));
            if (p != q) {
                atomicAdd(&counts[p], (unsigned long long)(i + 1));
                atomicAdd(&counts[q], (unsigned long long)0 - (unsigned long long)(i + 1));
            }
        }
    }
}

// Low-memory route: one round of the Regions level's block swaps (rdst_regions.cpp plans them; the ranges of one round
// are pairwise disjoint).  One workgroup per chunk of at most SWAP_CHUNK elements: exchange `len` elements at `a` and `b`.
// Device twin of the parallel `swap_with_slice` round of src/sorts/regions_sort.rs:247-251.
constexpr uint32_t SWAP_CHUNK = 8192;
struct SwapChunk { uint64_t a, b; uint32_t len, pad; };

template <typename K>
__global__ __launch_bounds__(256) void swap_ranges_kernel(K* __restrict__ keys, const SwapChunk* __restrict__ chunks) {
    const SwapChunk c = chunks[blockIdx.x];
    K* pa = keys + c.a;
    K* pb = keys + c.b;
    for (uint32_t i = threadIdx.x; i < c.len; i += 256 * 4) {
        K va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * 256 < c.len) { va[u] = pa[i + u * 256]; vb[u] = pb[i + u * 256]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * 256 < c.len) { pa[i + u * 256] = vb[u]; pb[i + u * 256] = va[u]; }
    }
}

__global__ void raise_error_kernel(uint32_t* err, uint32_t bits) { atomicOr(err, bits); }  // rdst_hip_debug_raise_device_error

// K6: one level's histogram + "digit sequence has an inversion" flag
template <typename K>
__global__ __launch_bounds__(256) void level_counts_kernel(const K* __restrict__ keys, uint64_t n, int shift,
                                                           K neg, K pos, unsigned long long* __restrict__ counts,
                                                           uint32_t* __restrict__ unsorted) {
    __shared__ uint32_t s_h[RADIX];
    __shared__ uint32_t s_uns;
    s_h[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_uns = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    bool uns = false;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const uint32_t d = digit_of(map_key<K>(keys[i], neg, pos), shift);
        if (i > 0) {
            const uint32_t dp = digit_of(map_key<K>(keys[i - 1], neg, pos), shift);
            uns |= d < dp;
        }
        atomicAdd(&s_h[d], 1u);
    }
    if (uns) s_uns = 1;
    __syncthreads();
    if (s_h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)s_h[threadIdx.x]);
    if (threadIdx.x == 0 && s_uns) atomicOr(unsorted, 1u);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
thread_local std::string g_last_error;

int fail(int code, const char* what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    g_last_error = buf;
    return code;
}
#define HIP_TRY(expr)                                                   \
    do {                                                                \
        hipError_t e__ = (expr);                                        \
        if (e__ != hipSuccess) return fail(RDST_ERR_HIP, #expr, e__);   \
    } while (0)

struct PassCfg { int nwaves, kpt4, kpt8, stages; };
// Scatter-kernel shapes.  Larger tiles mean longer runs per digit and fewer look-backs;
// staging the whole tile at once saves a second pass over the slots.  The defaults are the
// largest tiles that neither spill registers nor cost a block per CU.
constexpr PassCfg kPassCfgs[] = {
    {8, 16, 8, 1},    // 0: 512 threads,  8192 / 4096 keys per tile, whole tile staged in LDS (32 KiB)
    {8, 24, 12, 2},   // 1: 512 threads, 12288 / 6144 keys per tile, staged in two halves (24 KiB)
    {12, 24, 12, 2},  // 2: 768 threads, 18432 / 9216 keys per tile, two halves (36 KiB)   <- default, 4-byte keys from 4 GiB up, 16-byte keys
    {12, 28, 14, 2},  // 3: 768 threads, 21504 / 10752 keys per tile, two halves (42 KiB)  <- default, 8-byte keys
    {12, 22, 11, 1},  // 4: 768 threads, 16896 / 8448 keys per tile, whole tile staged (66 KiB; two blocks per CU only with 32-bit deltas)  <- default, keys up to 4 bytes below 4 GiB
    {12, 20, 10, 1},  // 5: 768 threads, 15360 / 7680 keys per tile, whole tile staged (60 KiB: two blocks per CU with 64-bit deltas too)  <- default, signed / float 8-byte keys
};
constexpr int kNumPassCfgs = sizeof(kPassCfgs) / sizeof(kPassCfgs[0]);
// 4-byte and narrower keys: the whole tile staged once (config 4) while two blocks still fit a CU,
// i.e. while destinations are 32-bit offsets (n * size < 4 GiB); otherwise the two-stage shapes
constexpr int default_cfg(uint32_t elem_bytes, uint64_t n, bool mapped = false) {
    // 8-byte keys: unsigned ones run at copy speed either way (3.5 ms per 10^9-key pass against 3.4 for a copy of the
    // same bytes); signed and float ones pay five more vector instructions per key use for the map, which the
    // ballot ranking of the two-stage shape cannot hide (4.1-4.2 ms) and the returning-add ranking of shape 5 can (3.6)
    if (elem_bytes == 8) return mapped ? 5 : 3;
    if (elem_bytes <= 4 && n * elem_bytes < (1ull << 32)) return 4;
    return 2;
}
// keys per thread for a key width, from the table's 8-byte figure: same bytes per thread
constexpr int kpt_for(int kpt8, size_t elem_bytes) { return elem_bytes <= 4 ? kpt8 * 2 : (elem_bytes == 8 ? kpt8 : (kpt8 / 2) & ~1); }

struct Tuning {
    int pass_cfg = -1;  // < 0: default_cfg()
    int hist_bpc = 0;
    bool profiling = false;
    bool chains = true;
    int fast_rank = 1;
    bool small_sort = true;
    bool hybrid = true;                 // consider the hybrid route at all
    bool count_sort = true;             // 4-byte keys: K4 as a counting sort by value (false: the generic ranked passes)
    bool halves = true;                 // 4-byte keys: pass L-1 hands K4 the low halves only (16-bit array in the workspace)
    bool presample = true;              // a 65 536-key sample before K1h: gross skew goes straight to the LSD route
    bool wide2 = true;                  // 8-byte keys: K4 as two 512-thread blocks per CU (false: one 1024-thread block)
    bool wide3 = true;                  // ... in its third form (local_wide3_sort_kernel); false: local_wide2_sort_kernel
    bool atomic_route = true;           // 4-byte keys: try ROUTE_ATOMIC (no counting read) before anything else
    bool exact_msd = true;              // behind a sample that flags the keys, K1h runs before the MSD passes and they take their exact form for the hybrid route
    bool giants = true;                 // 4-byte keys, hybrid route: buckets of 65 536 keys and more are sorted by the giant kernels (else: LSD route)
    bool chain_routes = true;           // behind a failed atomic route try the hybrid route before the LSD one
    bool expand = true;                 // 4-byte keys: buckets the counting K4 refuses go to the expanding one (any bucket below 65 536 keys)
    bool atomic_wide = true;            // ROUTE_ATOMIC for 8-byte keys too (whole keys in the slots)
    bool persist_fallback = true;       // behind the atomic route the LSD passes run as persistent blocks (cheap to skip)
    bool split_always = false;          // ... at every length, in eight parts (tests)
    bool split = true;                  // 8-byte keys beyond the atomic route's window: one exact pass on the top byte, then its groups as slices of their own (run_split_sort)
    bool predict = true;                // the sample may predict the LSD route (Plan::predict_lsd): neither MSD passes nor K1h are tried
    uint64_t hybrid_min_len = 0;        // rdst_hip_set_hybrid's min_len; 0: the measured defaults below (atomic_min_len, hybrid_min_len)
};
uint32_t g_ablate = 0;  // only ever set by the RDST_EXPERIMENTS build
#ifdef RDST_EXPERIMENTS
size_t g_exp_lds_total = 0;
#endif
Tuning g_tuning;
std::mutex g_mutex;

struct DeviceState {
    bool init = false;
    int cus = 256;
    void* ws = nullptr;
    size_t ws_bytes = 0;
    uint32_t* host_err = nullptr;  // pinned
    // Device error word: a small allocation of its own, outside the per-sort workspace, so that it survives
    // the clear at the start of every pipeline and a re-allocation of the workspace.  Kernels only OR
    // into it; rdst_hip_device_status (and the blocking entry points) read AND clear it.
    uint32_t* err_dev = nullptr;
    hipStream_t host_stream = nullptr;  // the host entry points' own stream: created once (creating and destroying one per call cost ~0.1 ms)
    void* host_buf = nullptr;           // their device buffer (keys + tmp), kept and grown on demand
    size_t host_buf_bytes = 0;
    std::mutex host_mutex;              // one host-slice sort per device at a time (they share stream and buffer)
    hipEvent_t host_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // around H2D, sort, D2H of the most recent host-slice sort
    bool host_timed = false;
    hipEvent_t last_done = nullptr;  // recorded after every enqueue that uses the workspace
    hipStream_t last_stream = nullptr;
    bool have_last = false;
    bool last_plan_valid = false;  // the most recent call ran a pipeline whose Plan sits at last_plan_off of the workspace
    size_t last_plan_off = 0;
    // per-kernel timing (rdst_hip_set_profiling): events recorded between the launches of the
    // most recent pipeline, on the stream the kernels run on
    std::vector<hipEvent_t> prof_events;
    std::vector<uint32_t> prof_kinds;  // prof_kinds[i]: what ran between event i and event i + 1 (RDST_STAGE_* | level << 8)
    uint32_t prof_used = 0;
    struct ProfRun { uint32_t begin, count; };
    std::vector<ProfRun> prof_runs;  // one per pipeline since profiling was (re-)enabled
};
DeviceState g_dev[16];

#ifndef RDST_MSD_WAVES
#define RDST_MSD_WAVES 12
#endif
#ifndef RDST_MSD_KPT4
#define RDST_MSD_KPT4 22
#endif
#ifndef RDST_MSD_KPT8
#define RDST_MSD_KPT8 11
#endif
constexpr int MSD_WAVES = RDST_MSD_WAVES;
constexpr int msd_kpt(size_t key_bytes) { return key_bytes == 4 ? RDST_MSD_KPT4 : RDST_MSD_KPT8; }  // ROUTE_ATOMIC tiles: 66 KiB of keys, two blocks per CU

struct Layout {
    uint32_t levels, tile, status_bytes;  // status_bytes: 4 or 8 per word
    uint64_t tiles;
    size_t off_err, off_tickets, off_plan, off_hpos, off_hpair, off_h16, off_hpos16, off_status, off_status_near, zero_bytes, off_hist, off_base,
        off_cbase, off_chains, off_bstart, off_fblist, off_fblist2, off_xtile0, off_glist, off_gsplit, off_gtables, off_halves, off_cursor_a, off_cursor_b, off_msd_a, total;
    uint32_t msd_cap_a, msd_slices;  // ROUTE_ATOMIC: keys an area of pass A holds; areas per top digit
    uint32_t slot_cap;               // and keys a bucket's slot (pass B's destination) holds
};

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int tile_keys(int cfg, uint32_t elem_bytes) {
    const PassCfg& p = kPassCfgs[cfg];
    return p.nwaves * 64 * kpt_for(p.kpt8, elem_bytes);
}

// count tables of the hybrid route's giant buckets (256 KiB each: 1 GiB, in the atomic route's areas when it has them) —
// 10^9 normally distributed f32 keys make ~1 800 giants
constexpr uint32_t GIANT_MAX = 4096;
static_assert(GIANT_MAX + 1 <= GIANT_ITEMS_LDS, "the giant kernels keep every giant's first work item in LDS");

Layout make_layout(uint64_t n, uint32_t elem_bytes, uint32_t levels, int cfg, uint32_t tile_override = 0, bool want_halves = false,
                   bool want_msd = false, bool want_giants = false) {
    Layout L{};
    L.levels = levels;
    L.tile = tile_override ? tile_override : (uint32_t)tile_keys(cfg, elem_bytes);
    L.tiles = n / L.tile + CHAINS + 2;  // status rows per level: every chain may end and begin on partial tiles
    L.status_bytes = n < (1ull << 30) ? 4 : 8;  // an inclusive prefix can reach n
    size_t o = 0;
    L.off_err = o; o += 64;  // cleared flags of one sort: [1] inversion seen, [2] a K1h counter overflowed, [3] length of K4's first hand-on list, [4] an area or a slot of the atomic route overflowed, [5] length of K4's second list (the error word itself lives in DeviceState::err_dev)
    o = align_up(o, 128);
    L.off_tickets = o; o += sizeof(uint32_t) * MAX_LEVELS * TICKET_ROW;  // per chain + mask of chains handed out, a line each
    L.off_plan = o; o += align_up(sizeof(Plan), 16);
    L.off_hpos = o; o += sizeof(uint64_t) * (size_t)levels * CHAINS * RADIX;
    L.off_hpair = o; o += sizeof(uint64_t) * (size_t)levels * CHAINS * RADIX;
    o = align_up(o, 128);
    L.off_cursor_a = o; o += sizeof(uint32_t) * RADIX * MSD_SLICES;         // ROUTE_ATOMIC: claim counters of pass A [slice][digit]: 8 lines per slice ...
    L.off_cursor_b = o; o += sizeof(uint32_t) * (size_t)H16_BINS;           // ... and of pass B (one per bucket)
    L.off_h16 = o; o += sizeof(uint32_t) * (size_t)H16_BINS;              // hybrid route: bucket counts (K1h)
    L.off_hpos16 = o; o += sizeof(uint64_t) * 2 * (size_t)CHAINS * RADIX; // and its level L-2 / level L-1 counts per position range
    L.off_status = o; o += (size_t)L.status_bytes * levels * L.tiles * RADIX;
    L.off_status_near = o; o += (size_t)L.status_bytes * levels * L.tiles * RADIX;
    L.zero_bytes = align_up(o, 16); o = L.zero_bytes;  // everything up to here is cleared per sort
    L.off_hist = o; o += sizeof(uint64_t) * (size_t)levels * RADIX;
    L.off_base = o; o += sizeof(uint64_t) * (size_t)levels * RADIX;
    L.off_cbase = o; o += sizeof(uint64_t) * (size_t)levels * CHAINS * RADIX;
    L.off_chains = o; o += sizeof(LevelChains) * (size_t)levels;
    o = align_up(o, 16);
    L.off_bstart = o; o += sizeof(uint32_t) * ((size_t)H16_BINS + 4);     // hybrid route: bucket starts
    L.off_fblist = o; o += sizeof(uint32_t) * (size_t)H16_BINS;           // buckets the first K4 kernel hands on (count: header word 3)
    L.off_fblist2 = o; o += sizeof(uint32_t) * (size_t)H16_BINS;          // and those the second one hands on (count: header word 5)
    L.off_xtile0 = o; o += sizeof(uint32_t) * (RADIX + 8);                  // exact form of the MSD passes: first tile of every top digit's region in pass B's grid
    L.off_glist = o; o += sizeof(uint32_t) * 3 * ((size_t)GIANT_MAX + 16); // giants of the hybrid route: buckets, first counting item, first expanding item
    o = align_up(o, 32);
    L.off_gsplit = o;
    if (want_giants) o += sizeof(GiantItem) * (size_t)(n / GIANT_OUT + GIANT_MAX + 16);
    o = align_up(o, 256);
    L.off_halves = o;                                                       // hybrid route, 4-byte keys: low halves between pass L-1 and K4
    if (want_halves && !want_msd) o += align_up(sizeof(uint16_t) * n, 256);
    // ROUTE_ATOMIC: 65 536 slots of one K4 tile each — low halves of 4-byte keys, whole 8-byte keys
    L.slot_cap = 0;
    if (want_msd) {  // a slot holds a uniform bucket with ten sigma to spare (8 sigma fit a K4 tile: atomic_eligible), at most one K4 tile
        const double mean = (double)n / H16_BINS;
        const uint64_t want = ((uint64_t)(mean + 10.0 * __builtin_sqrt(mean)) + 64 + 63) / 64 * 64;
        L.slot_cap = (uint32_t)(want < (uint64_t)local_tile(elem_bytes) ? want : (uint64_t)local_tile(elem_bytes));
        o += align_up((elem_bytes == 4 ? sizeof(uint16_t) : (size_t)elem_bytes) * H16_BINS * L.slot_cap, 256);
        if (o - L.off_halves < sizeof(uint16_t) * n) o = L.off_halves + align_up(sizeof(uint16_t) * n, 256);  // (the hybrid route's halves use the same space)
    }
    L.off_msd_a = o;
    L.msd_cap_a = 0;
    L.msd_slices = 1;
    if (want_msd) {
        // areas of pass A: tiles are dealt to the slices in turn, so a slice gets its share of the keys give or take a tile;
        // capacity = mean + max(1 / 8, 8 sigma) + two tiles' worth of one digit, whole 64s.  Few tiles: one slice.
        // (An eighth of slack, 0.5 GB per 10^9 u32 keys: dense ids below a bound that is no power of two — 10^9 of 2^30: the
        // populated top digits hold 7 % over the mean — stay on this route; with 1 % they overflowed an area, were counted by
        // K1h, made more giants than tables and ended on the LSD route, slower than the LSD-only setting.)
        const uint64_t TILE = (uint64_t)MSD_WAVES * 64 * msd_kpt(elem_bytes);
        const uint64_t tiles = (n + TILE - 1) / TILE;
        L.msd_slices = tiles >= 64 * MSD_SLICES ? MSD_SLICES : 1;
        const double mean = (double)n / (RADIX * L.msd_slices);
#ifndef RDST_AREA_SLACK
#define RDST_AREA_SLACK 0.125
#endif
        const double slack = mean * RDST_AREA_SLACK > 8.0 * __builtin_sqrt(mean) ? mean * RDST_AREA_SLACK : 8.0 * __builtin_sqrt(mean);
        L.msd_cap_a = (uint32_t)(((uint64_t)(mean + slack) + 2 * TILE / RADIX + 64) / 64 * 64);
        o += align_up((size_t)elem_bytes * L.msd_cap_a * RADIX * L.msd_slices, 256);
    }
    // the giants' count tables: they may use the areas of pass A (dead once the atomic route has failed, and the giant kernels
    // only run behind the hybrid route's own passes); without areas they get their own space
    L.off_gtables = L.off_msd_a;
    if (want_giants) {
        const size_t need = sizeof(uint32_t) * (size_t)GIANT_TABLE * GIANT_MAX;
        if (o - L.off_msd_a < need) o = L.off_msd_a + need;
    }
    L.total = align_up(o, 256);
    return L;
}

int current_device_state(DeviceState** out, int* dev_out = nullptr) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(RDST_ERR_NO_DEVICE, "hipGetDevice", e);
    if (dev < 0 || dev >= 16) return fail(RDST_ERR_NO_DEVICE, "device ordinal out of range");
    DeviceState& D = g_dev[dev];
    if (!D.init) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            char b[160];
            snprintf(b, sizeof b, "device %d is %s; this library is built for gfx950 only", dev, prop.gcnArchName);
            return fail(RDST_ERR_NO_DEVICE, b);
        }
        D.cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        HIP_TRY(hipHostMalloc((void**)&D.host_err, 64, hipHostMallocDefault));
        HIP_TRY(hipMalloc((void**)&D.err_dev, 256));
        HIP_TRY(hipMemset(D.err_dev, 0, 256));
        HIP_TRY(hipEventCreateWithFlags(&D.last_done, hipEventDisableTiming));
        D.init = true;
    }
    *out = &D;
    if (dev_out) *dev_out = dev;
    return RDST_OK;
}

int ensure_workspace(DeviceState& D, size_t bytes) {
    if (D.ws_bytes >= bytes) return RDST_OK;
    if (D.ws) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(D.ws));
        D.ws = nullptr;
        D.ws_bytes = 0;
    }
    // head room so that slightly longer slices do not re-allocate; none for the big layouts (the atomic route's workspace is
    // 1.7 x the slice: an eighth of that on top is gigabytes)
    size_t want = align_up(bytes + (bytes < ((size_t)1 << 30) ? bytes / 8 : 0), 1 << 20);
    hipError_t e = hipMalloc(&D.ws, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // (not sticky: the caller may retry with a leaner layout)
        D.ws = nullptr;
        return fail(RDST_ERR_HIP, "hipMalloc(workspace)", e);
    }
    D.ws_bytes = want;
    D.last_plan_valid = false;
    return RDST_OK;
}

// The workspace is shared by every call on the device: work queued on another stream must
// finish before this stream reuses it.
int workspace_acquire(DeviceState& D, hipStream_t s) {
    if (D.have_last && D.last_stream != s) HIP_TRY(hipStreamWaitEvent(s, D.last_done, 0));
    return RDST_OK;
}
int workspace_release(DeviceState& D, hipStream_t s) {
    HIP_TRY(hipEventRecord(D.last_done, s));
    D.last_stream = s;
    D.have_last = true;
    return RDST_OK;
}

// `kind`: the stage that ENDS at this mark (ignored for a run's first mark)
int prof_mark(DeviceState& D, hipStream_t s, uint32_t kind = 0) {
    if (!g_tuning.profiling || D.prof_runs.empty() || D.prof_used >= 8192) return RDST_OK;
    if (D.prof_used == D.prof_events.size()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        D.prof_events.push_back(e);
        D.prof_kinds.push_back(0);
    }
    if (D.prof_used > D.prof_runs.back().begin) D.prof_kinds[D.prof_used - 1] = kind;
    HIP_TRY(hipEventRecord(D.prof_events[D.prof_used++], s));
    D.prof_runs.back().count = D.prof_used - D.prof_runs.back().begin;
    return RDST_OK;
}

// Below these lengths the buckets are too small for one workgroup each to pay off: K4 has a floor of 65 536 workgroups' fixed
// costs (0.61 ms for 4-byte keys, 1.17 ms for 8-byte keys, whatever n: tools/window_probe.py).  The atomic route passes the
// LSD one at ~1.6 x 10^8 u32 keys (2 x 10^8: 1.44 against 1.58 ms) and at ~5 x 10^7 u64 keys (2^26: 1.87 against 2.19 ms; 2^27:
// 2.36 against 4.10); the K1h hybrid route (one more read, exact passes) keeps round 2's 2^28.
uint64_t atomic_min_len(size_t key_bytes) { return g_tuning.hybrid_min_len ? g_tuning.hybrid_min_len : (key_bytes == 8 ? 1ull << 26 : 3ull << 26); }
uint64_t hybrid_min_len() { return g_tuning.hybrid_min_len ? g_tuning.hybrid_min_len : 1ull << 28; }

// ROUTE_ATOMIC: 4- and 8-byte keys, and a length at which a uniform bucket (n / 65 536 keys) stays 8 sigma below the K4 tile
bool atomic_eligible(uint64_t n, size_t key_bytes, int cfg) {
    if (!g_tuning.hybrid || !g_tuning.atomic_route || n < atomic_min_len(key_bytes) || n >= (1ull << 30)) return false;
    if (key_bytes == 4 ? cfg != 4 : (key_bytes != 8 || !g_tuning.atomic_wide || !g_tuning.count_sort || !g_tuning.wide2)) return false;
    const double mean = (double)n / H16_BINS;
    return mean + 8.0 * __builtin_sqrt(mean) <= (double)local_tile(key_bytes);
}

bool hybrid_eligible(uint64_t n, size_t key_bytes) {
    const uint64_t cap = key_bytes == 4 && g_tuning.count_sort && g_tuning.expand ? EXPAND_MAX : (uint64_t)local_tile(key_bytes);
    return g_tuning.hybrid && (key_bytes == 4 || key_bytes == 8) && n >= hybrid_min_len() && n <= (uint64_t)H16_BINS * cap &&
           n < (1ull << 32);
}

// hipFuncSetAttribute acts on the CURRENT device's copy of the function: remember (device, kernel) -> bytes
int ensure_lds_attr(const void* fn, size_t lds) {
    static std::map<std::pair<int, const void*>, size_t> done;  // callers hold g_mutex
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    auto it = done.find({dev, fn});
    if (it != done.end() && it->second == lds) return RDST_OK;
    HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    done[{dev, fn}] = lds;
    return RDST_OK;
}

KeyMap key_map_for(rdst_key_kind kind, uint32_t elem_bytes) {
    const u128 msb = (u128)1 << (elem_bytes * 8 - 1);
    const u128 ones = elem_bytes == 16 ? ~(u128)0 : (((u128)1 << (elem_bytes * 8)) - 1);
    switch (kind) {
        case RDST_KEY_SIGNED: return {msb, msb};
        case RDST_KEY_FLOAT: return {ones, msb};
        default: return {0, 0};
    }
}

template <typename K, int LEVELS, int VEC, bool PAIR>
int launch_hist_v(const K* keys, uint64_t n, uint32_t blocks, KeyMap km, unsigned long long* hpos, unsigned long long* hpair,
                  uint32_t* inversion, const Plan* plan, hipStream_t s, uint64_t* piece_out, int base_level = 0) {
    *piece_out = hist_piece(n, blocks, (uint64_t)HIST_THREADS * VEC * 4);
    constexpr size_t lds = (size_t)HistPlan<LEVELS, PAIR>::WORDS * sizeof(uint32_t);
    if (int rc = ensure_lds_attr(reinterpret_cast<const void*>(&hist_kernel<K, LEVELS, VEC, PAIR>), lds)) return rc;
    hipLaunchKernelGGL((hist_kernel<K, LEVELS, VEC, PAIR>), dim3(blocks), dim3(HIST_THREADS), lds, s, keys, n, (K)km.neg,
                       (K)km.pos, hpos, hpair, inversion, plan, base_level);
    HIP_TRY(hipGetLastError());
    return RDST_OK;
}

// pair == true also fills the joint tables the chain split of the later passes needs
template <typename K, int LEVELS>
int launch_hist(const K* keys, uint64_t n, uint32_t blocks, KeyMap km, unsigned long long* hpos, unsigned long long* hpair,
                bool pair, uint32_t* inversion, const Plan* plan, hipStream_t s, uint64_t* piece_out, int base_level = 0) {
    const bool aligned = (reinterpret_cast<uintptr_t>(keys) & 15u) == 0;
    constexpr int V = 16 / sizeof(K);
    if constexpr (LEVELS >= 2) {
        if (pair) {
            if (aligned) return launch_hist_v<K, LEVELS, V, true>(keys, n, blocks, km, hpos, hpair, inversion, plan, s, piece_out);
            return launch_hist_v<K, LEVELS, 1, true>(keys, n, blocks, km, hpos, hpair, inversion, plan, s, piece_out);
        }
    }
    if (aligned) return launch_hist_v<K, LEVELS, V, false>(keys, n, blocks, km, hpos, hpair, inversion, plan, s, piece_out, base_level);
    return launch_hist_v<K, LEVELS, 1, false>(keys, n, blocks, km, hpos, hpair, inversion, plan, s, piece_out, base_level);
}

// the look before K1h / before the atomic route's first pass
template <typename K>
int launch_presample(const K* keys, uint64_t n, KeyMap km, Plan* plan, hipStream_t s) {
    if (!g_tuning.presample || n < PRESAMPLE_MIN_LEN) return RDST_OK;
    const bool mapped = km.neg != 0 || km.pos != 0;
    const uint32_t limit = 12u + (uint32_t)(4ull * (uint64_t)local_tile(sizeof(K)) * PRESAMPLE_KEYS / n);
    constexpr size_t plds = presample_lds_bytes();
    // what route_kernel will hold against the exact counts (RouteArgs::giant_max, cap), for the sample's prediction
    const bool big4 = sizeof(K) == 4 && g_tuning.count_sort && g_tuning.expand && g_tuning.predict;
    const uint32_t giant_min = big4 ? GIANT_MIN : 0u;
    const uint32_t giant_max = big4 && g_tuning.giants && n < (1ull << 30) ? GIANT_MAX : 0u;
    const uint32_t bucket_cap = sizeof(K) == 8 && g_tuning.predict ? (uint32_t)local_tile(8) : 0u;
    if (mapped) {
        if (int rc = ensure_lds_attr(reinterpret_cast<const void*>(&presample_kernel<K, true>), plds)) return rc;
        hipLaunchKernelGGL((presample_kernel<K, true>), dim3(1), dim3(1024), plds, s, keys, n, (K)km.neg, (K)km.pos, plan, limit, giant_min, giant_max, bucket_cap);
    } else {
        if (int rc = ensure_lds_attr(reinterpret_cast<const void*>(&presample_kernel<K, false>), plds)) return rc;
        hipLaunchKernelGGL((presample_kernel<K, false>), dim3(1), dim3(1024), plds, s, keys, n, (K)km.neg, (K)km.pos, plan, limit, giant_min, giant_max, bucket_cap);
    }
    HIP_TRY(hipGetLastError());
    return RDST_OK;
}

// K1h: the hybrid route's 65 536-bin count (same grid and pieces as K1)
template <typename K>
int launch_hist16(const K* keys, uint64_t n, uint32_t blocks, KeyMap km, uint32_t* h16, unsigned long long* hpos16, uint32_t* inversion,
                  uint32_t* overflow, Plan* plan, hipStream_t s, bool sample_first = true, bool giant = false, uint32_t pre_launch = 0) {
    const bool aligned = (reinterpret_cast<uintptr_t>(keys) & 15u) == 0;
    const bool mapped = km.neg != 0 || km.pos != 0;
    if (sample_first)
        if (int rc = launch_presample<K>(keys, n, km, plan, s)) return rc;
    constexpr int V = 16 / sizeof(K);
    constexpr size_t lds = (size_t)H16_WORDS * sizeof(uint32_t);
#define RDST_H16(VEC, MAPPED)                                                                                              \
    do {                                                                                                                   \
        if (int rc = ensure_lds_attr(reinterpret_cast<const void*>(&hist16_kernel<K, VEC, MAPPED>), lds)) return rc;       \
        hipLaunchKernelGGL((hist16_kernel<K, VEC, MAPPED>), dim3(blocks), dim3(HIST_THREADS), lds, s, keys, n, (K)km.neg,  \
                           (K)km.pos, h16, hpos16, inversion, overflow, plan, pre_launch);                                 \
    } while (0)
#define RDST_H16G(VEC, MAPPED)                                                                                             \
    do {                                                                                                                   \
        if (int rc = ensure_lds_attr(reinterpret_cast<const void*>(&hist16_kernel<K, VEC, MAPPED, true>), lds)) return rc; \
        hipLaunchKernelGGL((hist16_kernel<K, VEC, MAPPED, true>), dim3(blocks), dim3(HIST_THREADS), lds, s, keys, n,       \
                           (K)km.neg, (K)km.pos, h16, hpos16, inversion, overflow, plan, pre_launch);                      \
    } while (0)
    if constexpr (sizeof(K) == 4) {
        if (giant) {
            if (aligned) { if (mapped) RDST_H16G(V, true); else RDST_H16G(V, false); }
            else { if (mapped) RDST_H16G(1, true); else RDST_H16G(1, false); }
            HIP_TRY(hipGetLastError());
            return RDST_OK;
        }
    }
#undef RDST_H16G
    if (aligned) { if (mapped) RDST_H16(V, true); else RDST_H16(V, false); }
    else { if (mapped) RDST_H16(1, true); else RDST_H16(1, false); }
#undef RDST_H16
    HIP_TRY(hipGetLastError());
    return RDST_OK;
}

#ifndef RDST_COUNT_THREADS
#define RDST_COUNT_THREADS 512  // three blocks per CU (48 KiB of LDS each, 80 VGPRs): 1.41 ms per 10^9 keys against 1.65 for two blocks of 1024
#endif
constexpr int COUNT_THREADS = RDST_COUNT_THREADS;
// K4: one workgroup per bucket of the hybrid route.  4-byte keys: the counting kernel, then the generic one
// over the (normally empty) list of buckets it could not take; 8-byte keys: the generic one over all buckets.
struct GiantArgs {
    uint32_t *glist, *gcount_item, *gexp_item, *tables;
    GiantItem* recs;
};

template <typename K>
int launch_local_sort(K* keys, K* tmp, const uint32_t* bstart, const Plan* plan, uint32_t* err, KeyMap km, uint32_t* list,
                      uint32_t* list_count, const uint16_t* src16, int cus, hipStream_t s, const uint32_t* slot_count = nullptr,
                      uint32_t slot_cap = 0, const K* src_slots = nullptr, uint32_t* list2 = nullptr, uint32_t* list2_count = nullptr,
                      const GiantArgs* ga = nullptr) {
    constexpr int NW = local_waves(sizeof(K)), KPT = local_kpt(sizeof(K));
    constexpr size_t lds = local_lds_bytes(sizeof(K));
    const bool mapped = km.neg != 0 || km.pos != 0;
    const uint32_t flags = (g_tuning.fast_rank ? RDST_FAST_RANK : 0u) | (g_tuning.fast_rank == 2 ? RDST_FAST_RANK_SELFTEST : 0u);
    const bool counting = sizeof(K) == 4 && g_tuning.count_sort;
    if constexpr (sizeof(K) == 4) {
        if (counting) {
            constexpr size_t clds = count_lds_bytes();
#define RDST_COUNT(MAPPED, FROM16)                                                                                                   \
    do {                                                                                                                             \
        if (int rc = ensure_lds_attr(reinterpret_cast<const void*>(&local_count_sort_kernel<COUNT_THREADS, MAPPED, FROM16>), clds)) return rc; \
        hipLaunchKernelGGL((local_count_sort_kernel<COUNT_THREADS, MAPPED, FROM16>), dim3(H16_BINS), dim3(COUNT_THREADS), clds, s, keys, tmp, \
                           src16, bstart, plan, err, (uint32_t)km.neg, (uint32_t)km.pos, list, list_count, slot_count, slot_cap); \
    } while (0)
            if (src16) { if (mapped) RDST_COUNT(true, true); else RDST_COUNT(false, true); }
            else { if (mapped) RDST_COUNT(true, false); else RDST_COUNT(false, false); }
#undef RDST_COUNT
            HIP_TRY(hipGetLastError());
            if (g_tuning.expand && list2) {  // what the counting kernel listed (a value 16 times, a bucket over its tile): one block per CU
                constexpr size_t elds = expand_lds_bytes();
#define RDST_COUNT16(MAPPED, FROM16)                                                                                                 \
    do {                                                                                                                             \
        if (int rc = ensure_lds_attr(reinterpret_cast<const void*>(&local_count16_sort_kernel<MAPPED, FROM16>), elds)) return rc;    \
        hipLaunchKernelGGL((local_count16_sort_kernel<MAPPED, FROM16>), dim3((uint32_t)cus), dim3(COUNT16_THREADS), elds, s, keys, tmp, \
                           src16, bstart, plan, (uint32_t)km.neg, (uint32_t)km.pos, list, list_count, list2, list2_count, slot_count, slot_cap); \
    } while (0)
                if (src16) { if (mapped) RDST_COUNT16(true, true); else RDST_COUNT16(false, true); }
                else { if (mapped) RDST_COUNT16(true, false); else RDST_COUNT16(false, false); }
#undef RDST_COUNT16
                HIP_TRY(hipGetLastError());
#define RDST_EXPAND(MAPPED, FROM16)                                                                                                  \
    do {                                                                                                                             \
        if (int rc = ensure_lds_attr(reinterpret_cast<const void*>(&local_expand_sort_kernel<MAPPED, FROM16>), elds)) return rc;     \
        hipLaunchKernelGGL((local_expand_sort_kernel<MAPPED, FROM16>), dim3((uint32_t)cus), dim3(EXPAND_THREADS), elds, s, keys, tmp, \
                           src16, bstart, plan, err, (uint32_t)km.neg, (uint32_t)km.pos, list2, list2_count, slot_count, slot_cap); \
    } while (0)
                if (src16) { if (mapped) RDST_EXPAND(true, true); else RDST_EXPAND(false, true); }
                else { if (mapped) RDST_EXPAND(true, false); else RDST_EXPAND(false, false); }
#undef RDST_EXPAND
                HIP_TRY(hipGetLastError());
                if (ga) {  // the giants of the hybrid route (route_kernel listed them; none: four launches that return at once)
                    constexpr size_t clds2 = giant_count_lds_bytes();
                    hipLaunchKernelGGL(giant_zero_kernel, dim3((uint32_t)cus * 4), dim3(256), 0, s, plan, ga->tables);
#define RDST_GCOUNT(MAPPED, FROM16)                                                                                                  \
    do {                                                                                                                             \
        if (int rc = ensure_lds_attr(reinterpret_cast<const void*>(&giant_count_kernel<MAPPED, FROM16>), clds2)) return rc;          \
        hipLaunchKernelGGL((giant_count_kernel<MAPPED, FROM16>), dim3((uint32_t)cus), dim3(GIANT_THREADS), clds2, s, keys, tmp, src16, \
                           bstart, plan, (uint32_t)km.neg, (uint32_t)km.pos, ga->glist, ga->gcount_item, ga->tables);               \
    } while (0)
                    if (src16) { if (mapped) RDST_GCOUNT(true, true); else RDST_GCOUNT(false, true); }
                    else { if (mapped) RDST_GCOUNT(true, false); else RDST_GCOUNT(false, false); }
#undef RDST_GCOUNT
                    hipLaunchKernelGGL(giant_scan_kernel, dim3((uint32_t)cus), dim3(GIANT_THREADS), 0, s, plan, ga->tables);
                    hipLaunchKernelGGL(giant_split_kernel, dim3((uint32_t)cus * 2), dim3(256), 0, s, plan, bstart, ga->glist, ga->gexp_item, ga->tables, ga->recs);
                    if (mapped) hipLaunchKernelGGL((giant_expand_kernel<true>), dim3((uint32_t)cus * 8), dim3(GIANT_XTHREADS), 0, s, keys, tmp, plan, (uint32_t)km.neg, (uint32_t)km.pos, ga->tables, ga->recs);
                    else hipLaunchKernelGGL((giant_expand_kernel<false>), dim3((uint32_t)cus * 8), dim3(GIANT_XTHREADS), 0, s, keys, tmp, plan, (uint32_t)km.neg, (uint32_t)km.pos, ga->tables, ga->recs);
                    HIP_TRY(hipGetLastError());
                }
                return RDST_OK;  // (the list is spent: nothing is left for the ranked kernel)
            }
        }
    }
    bool listed = counting;
    if constexpr (sizeof(K) == 8) {
        if (g_tuning.count_sort) {
            constexpr size_t wlds = wide_lds_bytes();
            if (g_tuning.wide2 && g_tuning.wide3) {  // two 1024-thread blocks per CU, a third fewer LDS instructions than wide2
                constexpr size_t w3 = wide3_lds_bytes();
                if (mapped) {
                    if (int rc = ensure_lds_attr(reinterpret_cast<const void*>(&local_wide3_sort_kernel<true>), w3)) return rc;
                    hipLaunchKernelGGL((local_wide3_sort_kernel<true>), dim3(H16_BINS), dim3(WIDE3_THREADS), w3, s, keys, tmp, bstart, plan, err, (uint64_t)km.neg, (uint64_t)km.pos, list, list_count, src_slots, slot_count, slot_cap);
                } else {
                    if (int rc = ensure_lds_attr(reinterpret_cast<const void*>(&local_wide3_sort_kernel<false>), w3)) return rc;
                    hipLaunchKernelGGL((local_wide3_sort_kernel<false>), dim3(H16_BINS), dim3(WIDE3_THREADS), w3, s, keys, tmp, bstart, plan, err, (uint64_t)km.neg, (uint64_t)km.pos, list, list_count, src_slots, slot_count, slot_cap);
                }
            } else if (g_tuning.wide2) {  // the same with half the prefix table and every per-key state re-read (A/B, tests)
                constexpr size_t w2 = wide2_lds_bytes();
                if (mapped) {
                    if (int rc = ensure_lds_attr(reinterpret_cast<const void*>(&local_wide2_sort_kernel<true>), w2)) return rc;
                    hipLaunchKernelGGL((local_wide2_sort_kernel<true>), dim3(H16_BINS), dim3(WIDE2_THREADS), w2, s, keys, tmp, bstart, plan, err, (uint64_t)km.neg, (uint64_t)km.pos, list, list_count, src_slots, slot_count, slot_cap);
                } else {
                    if (int rc = ensure_lds_attr(reinterpret_cast<const void*>(&local_wide2_sort_kernel<false>), w2)) return rc;
                    hipLaunchKernelGGL((local_wide2_sort_kernel<false>), dim3(H16_BINS), dim3(WIDE2_THREADS), w2, s, keys, tmp, bstart, plan, err, (uint64_t)km.neg, (uint64_t)km.pos, list, list_count, src_slots, slot_count, slot_cap);
                }
            } else if (mapped) {
                if (int rc = ensure_lds_attr(reinterpret_cast<const void*>(&local_wide_sort_kernel<true>), wlds)) return rc;
                hipLaunchKernelGGL((local_wide_sort_kernel<true>), dim3(H16_BINS), dim3(WIDE_THREADS), wlds, s, keys, tmp, bstart, plan, err, (uint64_t)km.neg, (uint64_t)km.pos, list, list_count);
            } else {
                if (int rc = ensure_lds_attr(reinterpret_cast<const void*>(&local_wide_sort_kernel<false>), wlds)) return rc;
                hipLaunchKernelGGL((local_wide_sort_kernel<false>), dim3(H16_BINS), dim3(WIDE_THREADS), wlds, s, keys, tmp, bstart, plan, err, (uint64_t)km.neg, (uint64_t)km.pos, list, list_count);
            }
            HIP_TRY(hipGetLastError());
            listed = true;
        }
    }
    const dim3 grid(listed ? (uint32_t)((sizeof(K) == 8 ? 1 : 2) * cus) : (uint32_t)H16_BINS);
    const uint32_t* wl = listed ? list : nullptr;
    if (mapped) {
        if (int rc = ensure_lds_attr(reinterpret_cast<const void*>(&local_sort_kernel<K, NW, KPT, true>), lds)) return rc;
        hipLaunchKernelGGL((local_sort_kernel<K, NW, KPT, true>), grid, dim3(NW * 64), lds, s, keys, tmp, bstart, plan, err, (K)km.neg, (K)km.pos, flags, wl, list_count, src16, slot_count, slot_cap, src_slots);
    } else {
        if (int rc = ensure_lds_attr(reinterpret_cast<const void*>(&local_sort_kernel<K, NW, KPT, false>), lds)) return rc;
        hipLaunchKernelGGL((local_sort_kernel<K, NW, KPT, false>), grid, dim3(NW * 64), lds, s, keys, tmp, bstart, plan, err, (K)km.neg, (K)km.pos, flags, wl, list_count, src16, slot_count, slot_cap, src_slots);
    }
    HIP_TRY(hipGetLastError());
    return RDST_OK;
}

template <typename K, typename S, int KPT, int NWAVES, int STAGES, bool MAPPED, bool NARROW, typename V = NoVal, bool OUT16 = false, bool PERSIST = false>
int launch_pass_t(K* keys, K* tmp, uint64_t n, int level, const Layout& L, char* ws, KeyMap km, int cus, hipStream_t s,
                  V* vals = nullptr, V* vtmp = nullptr, uint16_t* out16 = nullptr) {
    constexpr int TILE = NWAVES * 64 * KPT;
    size_t lds = (size_t)pass_lds_bytes(NWAVES, NARROW ? 4 : 8, ((int)sizeof(K) + ValBytes<V>::value) * (TILE / STAGES));
    auto kernel = &onesweep_kernel<K, S, KPT, NWAVES, STAGES, MAPPED, NARROW, V, OUT16, PERSIST>;
#ifdef RDST_EXPERIMENTS
    if (g_exp_lds_total > lds) lds = g_exp_lds_total;  // fewer blocks per CU
#endif
    if (int rc = ensure_lds_attr(reinterpret_cast<const void*>(kernel), lds)) return rc;
    const uint64_t* cbase = reinterpret_cast<const uint64_t*>(ws + L.off_cbase) + (size_t)level * CHAINS * RADIX;
    S* status = reinterpret_cast<S*>(ws + L.off_status) + (size_t)level * L.tiles * RADIX;
    S* status_near = reinterpret_cast<S*>(ws + L.off_status_near) + (size_t)level * L.tiles * RADIX;
    const LevelChains* chains = reinterpret_cast<const LevelChains*>(ws + L.off_chains) + level;
    uint32_t* ticket = reinterpret_cast<uint32_t*>(ws + L.off_tickets) + (size_t)level * TICKET_ROW;
    const Plan* plan = reinterpret_cast<const Plan*>(ws + L.off_plan);
    DeviceState* D = nullptr;
    if (int rc = current_device_state(&D)) return rc;
    uint32_t* err = D->err_dev;
    // >= one block per tile of any chain split; PERSIST: as many blocks as stay resident (two per CU for the shape it is built for)
    const uint64_t resident = (uint64_t)cus * 2;
    const dim3 grid((uint32_t)(PERSIST && resident < L.tiles ? resident : L.tiles)), block(NWAVES * 64);
    hipLaunchKernelGGL((onesweep_kernel<K, S, KPT, NWAVES, STAGES, MAPPED, NARROW, V, OUT16, PERSIST>), grid, block, lds, s, keys, tmp, vals, vtmp, out16, n,
                       level, cbase, status, status_near, chains, ticket, plan, err, (K)km.neg, (K)km.pos, g_ablate | (g_tuning.fast_rank ? RDST_FAST_RANK : 0u) | (g_tuning.fast_rank == 2 ? RDST_FAST_RANK_SELFTEST : 0u));
    HIP_TRY(hipGetLastError());
    return RDST_OK;
}

template <typename K, typename S, bool MAPPED, bool NARROW>
int launch_pass_s(int cfg, K* keys, K* tmp, uint64_t n, int level, const Layout& L, char* ws, KeyMap km, int cus, hipStream_t s) {
    switch (cfg) {
#ifdef RDST_EXPERIMENTS  // the two small shapes are measured in DESIGN.md and only built into the tools library
        case 0: return launch_pass_t<K, S, kpt_for(8, sizeof(K)), 8, 1, MAPPED, NARROW>(keys, tmp, n, level, L, ws, km, cus, s);
        case 1: return launch_pass_t<K, S, kpt_for(12, sizeof(K)), 8, 2, MAPPED, NARROW>(keys, tmp, n, level, L, ws, km, cus, s);
#endif
        case 2: return launch_pass_t<K, S, kpt_for(12, sizeof(K)), 12, 2, MAPPED, NARROW>(keys, tmp, n, level, L, ws, km, cus, s);
        case 3: return launch_pass_t<K, S, kpt_for(14, sizeof(K)), 12, 2, MAPPED, NARROW>(keys, tmp, n, level, L, ws, km, cus, s);
        case 4: return launch_pass_t<K, S, kpt_for(11, sizeof(K)), 12, 1, MAPPED, NARROW>(keys, tmp, n, level, L, ws, km, cus, s);
        case 5: return launch_pass_t<K, S, kpt_for(10, sizeof(K)), 12, 1, MAPPED, NARROW>(keys, tmp, n, level, L, ws, km, cus, s);
    }
    return fail(RDST_ERR_ARG, "pass config not built into this library (0 and 1 exist in the tools build only)");
}

// the shapes the 16-bit hand-off of the hybrid route is built for: the default pass shape of 4-byte keys below 2^30
template <typename K>
bool halves_possible(int cfg, uint64_t n) { return sizeof(K) == 4 && cfg == 4 && n < (1ull << 30); }

template <typename K>
int launch_pass(int cfg, K* keys, K* tmp, uint64_t n, int level, const Layout& L, char* ws, KeyMap km, int cus, hipStream_t s,
                uint16_t* out16 = nullptr, bool persist = false) {
    const bool mapped = km.neg != 0 || km.pos != 0;
    const bool narrow = n * sizeof(K) < (1ull << 32);
    if constexpr (sizeof(K) == 4) {
        if (persist && out16 && cfg == 4 && L.status_bytes == 4 && narrow) {  // behind the atomic route: level L-1, whose pass feeds K4 on the hybrid route
            constexpr int KPT = kpt_for(11, sizeof(K));
            return mapped ? launch_pass_t<K, uint32_t, KPT, 12, 1, true, true, NoVal, true, true>(keys, tmp, n, level, L, ws, km, cus, s, nullptr, nullptr, out16)
                          : launch_pass_t<K, uint32_t, KPT, 12, 1, false, true, NoVal, true, true>(keys, tmp, n, level, L, ws, km, cus, s, nullptr, nullptr, out16);
        }
        if (persist && cfg == 4 && L.status_bytes == 4 && narrow) {  // the LSD fallback behind the atomic route
            constexpr int KPT = kpt_for(11, sizeof(K));
            return mapped ? launch_pass_t<K, uint32_t, KPT, 12, 1, true, true, NoVal, false, true>(keys, tmp, n, level, L, ws, km, cus, s)
                          : launch_pass_t<K, uint32_t, KPT, 12, 1, false, true, NoVal, false, true>(keys, tmp, n, level, L, ws, km, cus, s);
        }
    }
    if constexpr (sizeof(K) == 8) {
        if (persist && cfg == 5 && L.status_bytes == 4) {  // the same for 8-byte keys (64-bit deltas at every length: one shape)
            return mapped ? launch_pass_t<K, uint32_t, 10, 12, 1, true, false, NoVal, false, true>(keys, tmp, n, level, L, ws, km, cus, s)
                          : launch_pass_t<K, uint32_t, 10, 12, 1, false, false, NoVal, false, true>(keys, tmp, n, level, L, ws, km, cus, s);
        }
    }
    if constexpr (sizeof(K) == 4) {
        if (out16 && halves_possible<K>(cfg, n) && L.status_bytes == 4 && narrow) {
            constexpr int KPT = kpt_for(11, sizeof(K));
            return mapped ? launch_pass_t<K, uint32_t, KPT, 12, 1, true, true, NoVal, true>(keys, tmp, n, level, L, ws, km, cus, s, nullptr, nullptr, out16)
                          : launch_pass_t<K, uint32_t, KPT, 12, 1, false, true, NoVal, true>(keys, tmp, n, level, L, ws, km, cus, s, nullptr, nullptr, out16);
        }
    }
    if (L.status_bytes == 4) {
        if (narrow) {
            return mapped ? launch_pass_s<K, uint32_t, true, true>(cfg, keys, tmp, n, level, L, ws, km, cus, s)
                          : launch_pass_s<K, uint32_t, false, true>(cfg, keys, tmp, n, level, L, ws, km, cus, s);
        }
        return mapped ? launch_pass_s<K, uint32_t, true, false>(cfg, keys, tmp, n, level, L, ws, km, cus, s)
                      : launch_pass_s<K, uint32_t, false, false>(cfg, keys, tmp, n, level, L, ws, km, cus, s);
    }
    return mapped ? launch_pass_s<K, unsigned long long, true, false>(cfg, keys, tmp, n, level, L, ws, km, cus, s)
                  : launch_pass_s<K, unsigned long long, false, false>(cfg, keys, tmp, n, level, L, ws, km, cus, s);
}

// Key-value passes: one shape, 768 threads, the whole tile of (key, value) pairs staged in LDS.
constexpr int pair_kpt(size_t key_bytes, size_t val_bytes) { return key_bytes + val_bytes <= 8 ? 11 : (key_bytes + val_bytes <= 12 ? 7 : 5); }
constexpr int PAIR_WAVES = 12;

template <typename K, typename V>
int launch_pass_pairs(K* keys, K* tmp, V* vals, V* vtmp, uint64_t n, int level, const Layout& L, char* ws, KeyMap km, int cus,
                      hipStream_t s) {
    constexpr int KPT = pair_kpt(sizeof(K), sizeof(V));
    const bool mapped = km.neg != 0 || km.pos != 0;
    const bool narrow = n * (sizeof(K) > sizeof(V) ? sizeof(K) : sizeof(V)) < (1ull << 32);
    if (L.status_bytes == 4) {
        if (narrow) {
            return mapped ? launch_pass_t<K, uint32_t, KPT, PAIR_WAVES, 1, true, true, V>(keys, tmp, n, level, L, ws, km, cus, s, vals, vtmp)
                          : launch_pass_t<K, uint32_t, KPT, PAIR_WAVES, 1, false, true, V>(keys, tmp, n, level, L, ws, km, cus, s, vals, vtmp);
        }
        return mapped ? launch_pass_t<K, uint32_t, KPT, PAIR_WAVES, 1, true, false, V>(keys, tmp, n, level, L, ws, km, cus, s, vals, vtmp)
                      : launch_pass_t<K, uint32_t, KPT, PAIR_WAVES, 1, false, false, V>(keys, tmp, n, level, L, ws, km, cus, s, vals, vtmp);
    }
    return mapped ? launch_pass_t<K, unsigned long long, KPT, PAIR_WAVES, 1, true, false, V>(keys, tmp, n, level, L, ws, km, cus, s, vals, vtmp)
                  : launch_pass_t<K, unsigned long long, KPT, PAIR_WAVES, 1, false, false, V>(keys, tmp, n, level, L, ws, km, cus, s, vals, vtmp);
}

// The whole device-side pipeline for levels [level_lo, level_hi): memset, K1, K2, passes,
// optional copy-back.  `allow_skip` turns on level skipping.  With copy_back == false the
// result stays where the last executed pass put it (scatter hook: exactly one pass keys->tmp).
template <typename K, int LEVELS, typename V = NoVal>
int run_pipeline(K* keys, K* tmp, uint64_t n, rdst_key_kind kind, uint32_t level_lo, uint32_t level_hi,
                 bool allow_skip, bool copy_back, hipStream_t s, Layout* layout_out, char** ws_out, V* vals = nullptr,
                 V* vtmp = nullptr, bool deliver_tmp = false /* whole key-only sorts: leave the result in tmp, not in keys */) {
    constexpr bool HAS_V = ValBytes<V>::value != 0;
    if (deliver_tmp && (HAS_V || !copy_back)) return fail(RDST_ERR_ARG, "deliver_tmp: whole key-only sorts");
    DeviceState* D;
    int rc = current_device_state(&D);
    if (rc) return rc;
    if constexpr (!HAS_V) {
        // a slice of at most one small tile: the one-workgroup sort, in place, no workspace
        if (g_tuning.small_sort && level_lo == 0 && level_hi == (uint32_t)LEVELS && allow_skip && copy_back && !layout_out && !deliver_tmp &&
            n <= (uint64_t)SMALL_THREADS * small_kpt(sizeof(K))) {
            const KeyMap km = key_map_for(kind, sizeof(K));
            const size_t lds = (size_t)SMALL_WAVES * 1024 + 16 + sizeof(K) * SMALL_THREADS * small_kpt(sizeof(K));
            const bool mapped = km.neg != 0 || km.pos != 0;
            const void* fn = mapped ? reinterpret_cast<const void*>(&small_sort_kernel<K, LEVELS, true>)
                                    : reinterpret_cast<const void*>(&small_sort_kernel<K, LEVELS, false>);
            if ((rc = ensure_lds_attr(fn, lds))) return rc;
            if (mapped) hipLaunchKernelGGL((small_sort_kernel<K, LEVELS, true>), dim3(1), dim3(SMALL_THREADS), lds, s, keys, (uint32_t)n, (K)km.neg, (K)km.pos);
            else hipLaunchKernelGGL((small_sort_kernel<K, LEVELS, false>), dim3(1), dim3(SMALL_THREADS), lds, s, keys, (uint32_t)n, (K)km.neg, (K)km.pos);
            HIP_TRY(hipGetLastError());
            D->last_plan_valid = false;
            return RDST_OK;
        }
    }
    int cfg = g_tuning.pass_cfg;
    if (cfg < 0 || cfg >= kNumPassCfgs) cfg = default_cfg(sizeof(K), n, kind != RDST_KEY_UNSIGNED);
    // Hybrid route (whole sorts of 4- and 8-byte keys, long enough that a bucket is worth a workgroup, short
    // enough that 65 536 tiles can hold it): K1h counts the buckets, route_kernel decides.  If it says LSD,
    // K1 runs as ever (the slice is then read twice for counting); if it says hybrid, K1 returns at once.
    const bool whole_sort = !HAS_V && level_lo == 0 && level_hi == (uint32_t)LEVELS && allow_skip && copy_back;
    // which routes this sort may take, and the workspace that needs.  `lean`: the LSD route only (0.13 x the slice instead of
    // 1.7 x) — what is left when the device cannot spare the space of the byte-saving routes.
    struct RoutePick { bool try_atomic, try_hybrid, halves, giants; int cfg; Layout L; };
    const int cfg0 = cfg;
    auto pick_routes = [&](bool lean) -> RoutePick {
        RoutePick r{};
        r.cfg = cfg0;
        r.try_atomic = !lean && whole_sort && atomic_eligible(n, sizeof(K), r.cfg);
        // 8-byte keys: the fallback's passes run as persistent blocks of shape 5 (eight skipped passes of one block per tile cost 0.5 ms per 10^9 keys)
        if (r.try_atomic && sizeof(K) == 8 && g_tuning.persist_fallback && g_tuning.pass_cfg < 0) r.cfg = 5;
        // Behind a failed atomic route (an area or a slot overflowed) the hybrid route is tried next — exact counts, any bucket
        // the local sort takes — and the LSD route last.  One launch sequence serves all three: every kernel looks at the plan.
        const bool halves_cfg = g_tuning.halves && g_tuning.count_sort && halves_possible<K>(r.cfg, n);
        r.try_hybrid = !lean && whole_sort && hybrid_eligible(n, sizeof(K)) && (!r.try_atomic || (g_tuning.chain_routes && (sizeof(K) == 8 || halves_cfg)));
        r.halves = r.try_hybrid && halves_cfg;
        // 4-byte keys: the hybrid route takes buckets of any size (K1h counts them exactly, the giant kernels of K4 sort them)
        r.giants = r.try_hybrid && sizeof(K) == 4 && g_tuning.giants && g_tuning.count_sort && g_tuning.expand && n < (1ull << 30);
        r.L = make_layout(n, sizeof(K), LEVELS, r.cfg, HAS_V ? PAIR_WAVES * 64 * pair_kpt(sizeof(K), ValBytes<V>::value) : 0, r.halves, r.try_atomic, r.giants);
        return r;
    };
    RoutePick rp = pick_routes(false);
    if (rp.L.tiles >= (1ull << 31)) return fail(RDST_ERR_ARG, "len too large for one launch");
    rc = ensure_workspace(*D, rp.L.total);
    if (rc && (rp.try_atomic || rp.try_hybrid)) {  // no room for the areas and slots: the LSD route needs an eighth of the slice
        rp = pick_routes(true);
        rc = ensure_workspace(*D, rp.L.total);
    }
    if (rc) return rc;
    const bool try_atomic = rp.try_atomic, try_hybrid = rp.try_hybrid, halves = rp.halves, giants = rp.giants;
    cfg = rp.cfg;
    const Layout L = rp.L;
    char* ws = static_cast<char*>(D->ws);
    const KeyMap km = key_map_for(kind, sizeof(K));
    rc = workspace_acquire(*D, s);
    if (rc) return rc;

    // only the status rows of the passes that can run need clearing
    const size_t status_lo = L.off_status + (size_t)L.status_bytes * level_lo * L.tiles * RADIX;
    const size_t status_hi = L.off_status + (size_t)L.status_bytes * level_hi * L.tiles * RADIX;
    if (g_tuning.profiling) D->prof_runs.push_back({D->prof_used, 0});
    if ((rc = prof_mark(*D, s))) return rc;
    // hybrid-eligible sorts clear the rows of the two levels that route uses now and leave the others to a conditional
    // kernel behind the route decision (1 B u32: 0.07 ms of clearing -> half)
    const size_t level_rows = (size_t)L.status_bytes * L.tiles * RADIX;  // one level, one copy
    const bool split_clear = try_hybrid && !try_atomic && LEVELS > 2;
    if (try_atomic) {
        HIP_TRY(hipMemsetAsync(ws, 0, L.off_status, s));  // the status rows are the LSD route's: cleared behind the route decision, if it fell that way
    } else if (split_clear) {
        HIP_TRY(hipMemsetAsync(ws, 0, L.off_status, s));
        HIP_TRY(hipMemsetAsync(ws + L.off_status + level_rows * (LEVELS - 2), 0, level_rows * 2, s));
        HIP_TRY(hipMemsetAsync(ws + L.off_status_near + level_rows * (LEVELS - 2), 0, level_rows * 2, s));
    } else if (level_lo == 0 && level_hi == (uint32_t)LEVELS) {
        HIP_TRY(hipMemsetAsync(ws, 0, L.zero_bytes, s));  // header, count tables and both copies of the status rows are contiguous
    } else {
        HIP_TRY(hipMemsetAsync(ws, 0, L.off_status, s));
        if (status_hi > status_lo) {
            HIP_TRY(hipMemsetAsync(ws + status_lo, 0, status_hi - status_lo, s));
            HIP_TRY(hipMemsetAsync(ws + status_lo + (L.off_status_near - L.off_status), 0, status_hi - status_lo, s));
        }
    }
    if ((rc = prof_mark(*D, s, RDST_STAGE_CLEAR))) return rc;

    // K1: one 128-KiB-LDS block per CU (more only on request), but no more than the data needs
    uint64_t blocks = (uint64_t)(g_tuning.hist_bpc > 0 ? g_tuning.hist_bpc : 1) * D->cus;
    const uint64_t per_block_min = (uint64_t)HIST_THREADS * (16 / sizeof(K)) * 4;
    const uint64_t max_useful = (n + per_block_min - 1) / per_block_min;
    if (blocks > max_useful) blocks = max_useful;
    // Every block pays for zeroing and folding 128 KiB of LDS and for ~7 000 global adds on shared
    // addresses (measured: ~8 us + 0.2 us per block), against ~20 GB/s of sweep per block: below a few
    // hundred MB the best block count is about sqrt(bytes / 4 KiB), not one per CU.
    uint64_t balanced = 1;
    while (balanced * balanced * 4096 < n * sizeof(K)) ++balanced;
    if (blocks > balanced) blocks = balanced;
    if (blocks < 1) blocks = 1;
    unsigned long long* hpos = reinterpret_cast<unsigned long long*>(ws + L.off_hpos);
    uint32_t* inversion = reinterpret_cast<uint32_t*>(ws + L.off_err) + 1;  // second word of the (cleared) header
    uint64_t piece = 0;
    // the joint tables pay off only when a second pass can follow a first
    const bool pair = g_tuning.chains && LEVELS >= 2 && level_hi > level_lo + 1;
    unsigned long long* hpair = reinterpret_cast<unsigned long long*>(ws + L.off_hpair);
    Plan* plan = reinterpret_cast<Plan*>(ws + L.off_plan);
    // K1h + the route decision, launched twice behind a tried atomic route: before the MSD passes (they run only if the sample
    // flagged the keys; the MSD passes then take their exact form for the hybrid route) and after them (what is left)
    auto count_and_route = [&](uint32_t pre_launch, bool sample_first, hipStream_t cs) -> int {
      if constexpr (!HAS_V && (sizeof(K) == 4 || sizeof(K) == 8)) {
        uint32_t* overflow16 = reinterpret_cast<uint32_t*>(ws + L.off_err) + 2;
        uint32_t* h16 = reinterpret_cast<uint32_t*>(ws + L.off_h16);
        unsigned long long* hpos16 = reinterpret_cast<unsigned long long*>(ws + L.off_hpos16);
        if (int r = launch_hist16<K>(keys, n, (uint32_t)blocks, km, h16, hpos16, inversion, overflow16, plan, cs, sample_first, giants, pre_launch)) return r;
        if (!pre_launch)
            if (int r = prof_mark(*D, cs, RDST_STAGE_HIST16)) return r;
        RouteArgs ra{};
        ra.h16 = h16;
        ra.hpos16 = hpos16;
        ra.overflow = overflow16;
        ra.inversion = inversion;
        ra.allow_skip = allow_skip ? 1u : 0u;
        ra.bstart = reinterpret_cast<uint32_t*>(ws + L.off_bstart);
        ra.hpos = hpos;
        ra.hpair = hpair;
        ra.plan = plan;
        ra.n = n;
        ra.levels = (uint32_t)LEVELS;
        ra.cap = sizeof(K) == 4 && g_tuning.count_sort && g_tuning.expand ? EXPAND_MAX : (uint32_t)local_tile(sizeof(K));
        ra.giant_max = giants ? GIANT_MAX : 0u;
        ra.glist = reinterpret_cast<uint32_t*>(ws + L.off_glist);
        ra.gcount_item = ra.glist + GIANT_MAX + 16;
        ra.gexp_item = ra.gcount_item + GIANT_MAX + 16;
        ra.pre_launch = pre_launch;
        ra.msd_tile = (uint32_t)(MSD_WAVES * 64 * msd_kpt(sizeof(K)));
        ra.cursor_a = reinterpret_cast<uint32_t*>(ws + L.off_cursor_a);
        ra.cursor_b = reinterpret_cast<uint32_t*>(ws + L.off_cursor_b);
        ra.xtile0 = reinterpret_cast<uint32_t*>(ws + L.off_xtile0);
        ra.skip_a_ok = sizeof(K) == 4 ? 1u : 0u;
        hipLaunchKernelGGL(route_kernel, dim3(1), dim3(1024), 0, cs, ra);
        HIP_TRY(hipGetLastError());
      }
        return RDST_OK;
    };
    if constexpr (!HAS_V && (sizeof(K) == 4 || sizeof(K) == 8)) {
        if (try_atomic) {
            constexpr int KPT = msd_kpt(sizeof(K)), NW = MSD_WAVES, TILE = NW * 64 * KPT, W = (int)sizeof(K) * 8;
            constexpr bool HALF = sizeof(K) == 4;  // 4-byte keys leave pass B as their low halves
            constexpr size_t mlds = (size_t)NW * 1024 + 1024 + 128 + sizeof(K) * TILE;
            const uint32_t slot_cap = L.slot_cap;
            uint32_t* overflow = reinterpret_cast<uint32_t*>(ws + L.off_err) + 4;
            uint32_t* cursor_a = reinterpret_cast<uint32_t*>(ws + L.off_cursor_a);
            uint32_t* cursor_b = reinterpret_cast<uint32_t*>(ws + L.off_cursor_b);
            K* area_a = reinterpret_cast<K*>(ws + L.off_msd_a);
            uint16_t* slots16 = HALF ? reinterpret_cast<uint16_t*>(ws + L.off_halves) : nullptr;
            K* slots = HALF ? nullptr : reinterpret_cast<K*>(ws + L.off_halves);
            const bool mapped = km.neg != 0 || km.pos != 0;
            if ((rc = launch_presample<K>(keys, n, km, plan, s))) return rc;
            if constexpr (sizeof(K) == 4 || sizeof(K) == 8) {
                if (try_hybrid && g_tuning.exact_msd)
                    if ((rc = count_and_route(1u, false, s))) return rc;
            }
            if ((rc = prof_mark(*D, s, RDST_STAGE_SAMPLE))) return rc;
            MsdRanges xr{};
            uint32_t tiles_x = 0;  // exact pass A: eight ranges tiled on their own, block b -> range b % 8
            {
                const uint64_t xvec = (reinterpret_cast<uintptr_t>(keys) & 15u) == 0 ? 16 / sizeof(K) : 1;  // as launch_hist16 picks K1h's loads
                const uint64_t xpiece = hist_piece(n, (uint32_t)blocks, (uint64_t)HIST_THREADS * xvec * 4);
                for (int r = 0; r <= CHAINS; ++r) {
                    const uint64_t at = (uint64_t)hist_first_block((uint32_t)r, (uint32_t)blocks) * xpiece;
                    xr.start[r] = at < n ? at : n;
                }
                for (int r = 0; r < CHAINS; ++r) {
                    const uint32_t t = (uint32_t)((xr.start[r + 1] - xr.start[r] + TILE - 1) / TILE);
                    if (t > tiles_x) tiles_x = t;
                }
                tiles_x *= CHAINS;
            }
            const uint32_t* bstart_x = reinterpret_cast<const uint32_t*>(ws + L.off_bstart);
            const uint32_t* xtile0 = reinterpret_cast<const uint32_t*>(ws + L.off_xtile0);
            const uint32_t tiles_a = (uint32_t)((n + TILE - 1) / TILE);
            const uint32_t tpa = (L.msd_cap_a + TILE - 1) / TILE;
#define RDST_MSD(MAPPED, SECOND, GRID, ...)                                                                                              \
    do {                                                                                                                                 \
        if ((rc = ensure_lds_attr(reinterpret_cast<const void*>(&msd_scatter_kernel<K, KPT, NW, MAPPED, SECOND, (SECOND && HALF)>), mlds))) return rc; \
        hipLaunchKernelGGL((msd_scatter_kernel<K, KPT, NW, MAPPED, SECOND, (SECOND && HALF)>), dim3(GRID), dim3(NW * 64), mlds, s, __VA_ARGS__); \
    } while (0)
            // pass A: the slice, by its top byte, into 256 x 8 areas
            const uint32_t grid_a = tiles_a > tiles_x ? tiles_a : tiles_x;
            if (mapped) RDST_MSD(true, false, grid_a, keys, nullptr, n, 0u, grid_a, area_a, nullptr, cursor_a, L.msd_cap_a, W - 8, L.msd_slices, plan, overflow, inversion, (K)km.neg, (K)km.pos, tmp, keys, bstart_x, xtile0, xr);
            else RDST_MSD(false, false, grid_a, keys, nullptr, n, 0u, grid_a, area_a, nullptr, cursor_a, L.msd_cap_a, W - 8, L.msd_slices, plan, overflow, inversion, (K)km.neg, (K)km.pos, tmp, keys, bstart_x, xtile0, xr);
            HIP_TRY(hipGetLastError());
            if ((rc = prof_mark(*D, s, RDST_STAGE_MSD_A))) return rc;
            // pass B: every area, by the second byte, into the slot of its bucket
            uint32_t grid_b = (uint32_t)RADIX * L.msd_slices * tpa;
            if (grid_b < tiles_a + RADIX) grid_b = tiles_a + RADIX;  // (the exact form: every top digit's region ends on a partial tile)
            if (mapped) RDST_MSD(true, true, grid_b, area_a, cursor_a, 0ull, L.msd_cap_a, tpa, slots, slots16, cursor_b, slot_cap, W - 16, L.msd_slices, plan, overflow, inversion, (K)km.neg, (K)km.pos, tmp, keys, bstart_x, xtile0, xr);
            else RDST_MSD(false, true, grid_b, area_a, cursor_a, 0ull, L.msd_cap_a, tpa, slots, slots16, cursor_b, slot_cap, W - 16, L.msd_slices, plan, overflow, inversion, (K)km.neg, (K)km.pos, tmp, keys, bstart_x, xtile0, xr);
#undef RDST_MSD
            HIP_TRY(hipGetLastError());
            if ((rc = prof_mark(*D, s, RDST_STAGE_MSD_B))) return rc;
            MsdFinishArgs fa{};
            fa.cursor_b = cursor_b;
            fa.overflow = overflow;
            fa.inversion = inversion;
            fa.bstart = reinterpret_cast<uint32_t*>(ws + L.off_bstart);
            fa.plan = plan;
            fa.n = n;
            fa.allow_skip = allow_skip ? 1u : 0u;
            hipLaunchKernelGGL(msd_finish_kernel, dim3(1), dim3(1024), 0, s, fa);
            HIP_TRY(hipGetLastError());
            if ((rc = prof_mark(*D, s, RDST_STAGE_ROUTE))) return rc;
            if (!try_hybrid) {  // the LSD route's status rows, if the route fell that way
                const uint64_t vecs = level_rows * LEVELS / 16;
                hipLaunchKernelGGL(clear_unless_hybrid_kernel, dim3((uint32_t)D->cus * 4), dim3(256), 0, s, plan, reinterpret_cast<uint4*>(ws + L.off_status),
                                   vecs, reinterpret_cast<uint4*>(ws + L.off_status_near), vecs, 0u);
                HIP_TRY(hipGetLastError());
            }
        }
    }
    if constexpr (!HAS_V && (sizeof(K) == 4 || sizeof(K) == 8)) {
        if (try_hybrid) {
            if ((rc = count_and_route(0u, !try_atomic, s))) return rc;
            if (split_clear) {
                const uint64_t vecs = level_rows * (LEVELS - 2) / 16;  // rows are multiples of 1 KiB
                hipLaunchKernelGGL(clear_unless_hybrid_kernel, dim3((uint32_t)D->cus * 4), dim3(256), 0, s, plan,
                                   reinterpret_cast<uint4*>(ws + L.off_status), vecs, reinterpret_cast<uint4*>(ws + L.off_status_near), vecs, 0u);
                HIP_TRY(hipGetLastError());
            } else if (try_atomic) {  // nothing was cleared up front: every level's rows, unless the atomic route took the sort
                const uint64_t vecs = level_rows * LEVELS / 16;
                hipLaunchKernelGGL(clear_unless_hybrid_kernel, dim3((uint32_t)D->cus * 4), dim3(256), 0, s, plan,
                                   reinterpret_cast<uint4*>(ws + L.off_status), vecs, reinterpret_cast<uint4*>(ws + L.off_status_near), vecs, 1u);
                HIP_TRY(hipGetLastError());
            }
            if ((rc = prof_mark(*D, s, RDST_STAGE_ROUTE))) return rc;
        }
    }
    // a single pass (the parity hook, the sharded route's split) counts its own level only: one LDS atomic per key
    if (LEVELS > 1 && level_hi == level_lo + 1) rc = launch_hist<K, 1>(keys, n, (uint32_t)blocks, km, hpos, hpair, false, inversion, nullptr, s, &piece, (int)level_lo);
    else rc = launch_hist<K, LEVELS>(keys, n, (uint32_t)blocks, km, hpos, hpair, pair, inversion, (try_hybrid || try_atomic) ? plan : nullptr, s, &piece);
    if (rc) return rc;
    if ((rc = prof_mark(*D, s, RDST_STAGE_HIST))) return rc;
    ScanArgs sa{};
    sa.hpos = hpos;
    sa.hpair = pair ? hpair : nullptr;
    sa.hist = reinterpret_cast<unsigned long long*>(ws + L.off_hist);
    sa.base = reinterpret_cast<uint64_t*>(ws + L.off_base);
    sa.cbase = reinterpret_cast<uint64_t*>(ws + L.off_cbase);
    sa.chains = reinterpret_cast<LevelChains*>(ws + L.off_chains);
    sa.plan = plan;
    sa.tickets = reinterpret_cast<uint32_t*>(ws + L.off_tickets);
    sa.inversion = inversion;
    sa.n = n;
    sa.hist_piece = piece;
    sa.levels = (uint32_t)LEVELS;
    sa.allow_skip = allow_skip ? 1u : 0u;
    sa.level_lo = level_lo;
    sa.level_hi = level_hi;
    sa.hist_grid = (uint32_t)blocks;
    sa.tile = L.tile;
    sa.use_chains = g_tuning.chains ? 1u : 0u;
    sa.halves = halves ? 1u : 0u;
    sa.deliver_tmp = deliver_tmp ? 1u : 0u;
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(256 * SCAN_GROUPS), 0, s, sa);
    HIP_TRY(hipGetLastError());
    if ((rc = prof_mark(*D, s, RDST_STAGE_SCAN))) return rc;
    for (uint32_t level = level_lo; level < level_hi; ++level) {
        if constexpr (HAS_V) rc = launch_pass_pairs<K, V>(keys, tmp, vals, vtmp, n, (int)level, L, ws, km, D->cus, s);
        else rc = launch_pass<K>(cfg, keys, tmp, n, (int)level, L, ws, km, D->cus, s,
                                 halves && level + 1 == (uint32_t)LEVELS ? reinterpret_cast<uint16_t*>(ws + L.off_halves) : nullptr,
                                 try_atomic && g_tuning.persist_fallback);
        if (rc) return rc;
        if ((rc = prof_mark(*D, s, RDST_STAGE_PASS | (level << 8)))) return rc;
    }
    if constexpr (!HAS_V && (sizeof(K) == 4 || sizeof(K) == 8)) {
        if (try_hybrid || try_atomic) {
            // one K4 for both routes: the atomic route's buckets lie in the slots, the hybrid route's at their final place
            // (4-byte keys: as low halves, in the same region of the workspace either way)
            const bool from16 = sizeof(K) == 4 && (try_atomic || halves);
            GiantArgs ga{};
            ga.glist = reinterpret_cast<uint32_t*>(ws + L.off_glist);
            ga.gcount_item = ga.glist + GIANT_MAX + 16;
            ga.gexp_item = ga.gcount_item + GIANT_MAX + 16;
            ga.tables = reinterpret_cast<uint32_t*>(ws + L.off_gtables);
            ga.recs = reinterpret_cast<GiantItem*>(ws + L.off_gsplit);
            rc = launch_local_sort<K>(keys, tmp, reinterpret_cast<const uint32_t*>(ws + L.off_bstart), plan, D->err_dev, km,
                                      reinterpret_cast<uint32_t*>(ws + L.off_fblist), reinterpret_cast<uint32_t*>(ws + L.off_err) + 3,
                                      from16 ? reinterpret_cast<const uint16_t*>(ws + L.off_halves) : nullptr, D->cus, s,
                                      try_atomic ? reinterpret_cast<const uint32_t*>(ws + L.off_cursor_b) : nullptr, L.slot_cap,
                                      try_atomic && sizeof(K) == 8 ? reinterpret_cast<const K*>(ws + L.off_halves) : nullptr,
                                      reinterpret_cast<uint32_t*>(ws + L.off_fblist2), reinterpret_cast<uint32_t*>(ws + L.off_err) + 5,
                                      giants ? &ga : nullptr);
            if (rc) return rc;
            if ((rc = prof_mark(*D, s, RDST_STAGE_LOCAL))) return rc;
        }
    }
    if (copy_back) {
        const bool aligned = ((reinterpret_cast<uintptr_t>(keys) | reinterpret_cast<uintptr_t>(tmp)) & 15u) == 0;
        uint64_t cblocks = (n * sizeof(K) / 16 + 255) / 256;
        const uint64_t cap = (uint64_t)D->cus * 16;
        if (cblocks > cap) cblocks = cap;
        if (cblocks < 1) cblocks = 1;
        const Plan* plan = reinterpret_cast<const Plan*>(ws + L.off_plan);
        constexpr int VEC = 16 / sizeof(K);
        K* cdst = deliver_tmp ? tmp : keys;
        const K* csrc = deliver_tmp ? keys : tmp;
        if (aligned)
            hipLaunchKernelGGL((copyback_kernel<K, VEC>), dim3((uint32_t)cblocks), dim3(256), 0, s, cdst, csrc, n, plan, deliver_tmp ? 1u : 0u);
        else
            hipLaunchKernelGGL((copyback_kernel<K, 1>), dim3((uint32_t)cblocks), dim3(256), 0, s, cdst, csrc, n, plan, deliver_tmp ? 1u : 0u);
        HIP_TRY(hipGetLastError());
        if constexpr (HAS_V) {
            const bool valigned = ((reinterpret_cast<uintptr_t>(vals) | reinterpret_cast<uintptr_t>(vtmp)) & 15u) == 0;
            uint64_t vblocks = (n * sizeof(V) / 16 + 255) / 256;
            if (vblocks > cap) vblocks = cap;
            if (vblocks < 1) vblocks = 1;
            constexpr int VVEC = 16 / sizeof(V);
            if (valigned)
                hipLaunchKernelGGL((copyback_kernel<V, VVEC>), dim3((uint32_t)vblocks), dim3(256), 0, s, vals, vtmp, n, plan, 0u);
            else
                hipLaunchKernelGGL((copyback_kernel<V, 1>), dim3((uint32_t)vblocks), dim3(256), 0, s, vals, vtmp, n, plan, 0u);
            HIP_TRY(hipGetLastError());
        }
        if ((rc = prof_mark(*D, s, RDST_STAGE_COPYBACK))) return rc;
    }
    if (layout_out) *layout_out = L;
    if (ws_out) *ws_out = ws;
    D->last_plan_valid = true;
    D->last_plan_off = L.off_plan;
    return workspace_release(*D, s);
}

// Keys beyond the atomic route's window (a uniform bucket of n / 65 536 keys no longer fits K4's tile: n > 1.04 x 10^9):
// the reference recurses where a bucket is too big for the sort at hand (src/sorter.rs:131-138,
// src/sorts/recombinating_sort.rs:68-88).  One exact scatter pass on the top byte (K1 counts that level, K3 moves the keys:
// keys -> tmp, 24 bytes per key), its 256 counts come to the host, and aligned groups of top bytes — halved until the group
// fits the window — are sorted as slices of their own: their keys share the group's top bits, so the sample lowers the window
// past them and the atomic route takes the part as it takes a 10^9-key slice (48 bytes per key, result written straight into
// the caller's array: deliver_tmp).  8-byte keys: 72 bytes per key where the LSD route moves 136; 4-byte keys: 32 against 36.  A single top byte over the window is
// sorted by whatever run_pipeline picks for it (skewed keys: the LSD route).
constexpr uint64_t SPLIT4_MIN_LEN = 1'300'000'000ull;
bool split_eligible(uint64_t n, size_t key_bytes, int levels) {
    if ((key_bytes != 4 && key_bytes != 8) || levels != (int)key_bytes || !g_tuning.split || !g_tuning.hybrid || !g_tuning.atomic_route || !g_tuning.count_sort) return false;
    if (key_bytes == 8 && (!g_tuning.atomic_wide || !g_tuning.wide2)) return false;
    if (g_tuning.split_always) return n >= 2 && n < (1ull << 36);
    if (n >= (1ull << 36) || n < atomic_min_len(key_bytes) || atomic_eligible(n, key_bytes, default_cfg(key_bytes, n, true))) return false;
    // 4-byte keys have the K1h hybrid route beyond the window (24 bytes per key, every bucket the expanding K4's): 7.0 ms at 2^30
    // keys where the split's 12 + 20 bytes per key take 8.1; from SPLIT4_MIN_LEN keys up the split wins (1.6 x 10^9: 14.1 -> 11 ms)
    return key_bytes == 8 || n >= SPLIT4_MIN_LEN;
}

template <typename K, int LV>
int run_split_sort(K* keys, K* tmp, uint64_t n, rdst_key_kind kind, hipStream_t s) {
    Layout L{};
    char* ws = nullptr;
    int rc = run_pipeline<K, LV>(keys, tmp, n, kind, LV - 1, LV, false, false, s, &L, &ws);
    if (rc) return rc;
    uint64_t counts[RADIX];
    HIP_TRY(hipMemcpyAsync(counts, ws + L.off_hist + sizeof(uint64_t) * (size_t)(LV - 1) * RADIX, sizeof counts, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    uint64_t start[RADIX + 1];
    start[0] = 0;
    for (int d = 0; d < RADIX; ++d) start[d + 1] = start[d] + counts[d];
    if (start[RADIX] != n) return fail(RDST_ERR_DEVICE, "split pass: the top level's counts do not add up");
    struct Part { uint64_t off, cnt; };
    std::vector<Part> parts;
    // [lo, hi): an aligned group of top bytes.  A leaf: fits the window, or is too short to gain from a finer split, or one byte.
    auto leaf = [&](uint64_t cnt, int width) {
        if (g_tuning.split_always) return width <= 32;  // (tests: eight parts whatever the length)
        return width == 1 || cnt < 2 * atomic_min_len(sizeof(K)) || atomic_eligible(cnt, sizeof(K), default_cfg(sizeof(K), cnt, kind != RDST_KEY_UNSIGNED));
    };
    struct Range { int lo, hi; };
    std::vector<Range> todo{{0, RADIX}};
    while (!todo.empty()) {
        const Range r = todo.back();
        todo.pop_back();
        const uint64_t cnt = start[r.hi] - start[r.lo];
        if (cnt == 0) continue;
        if (leaf(cnt, r.hi - r.lo)) {
            parts.push_back({start[r.lo], cnt});
        } else {
            const int mid = (r.lo + r.hi) / 2;
            todo.push_back({mid, r.hi});
            todo.push_back({r.lo, mid});
        }
    }
    for (const Part& p : parts) {
        if (p.cnt == 1) {
            HIP_TRY(hipMemcpyAsync(keys + p.off, tmp + p.off, sizeof(K), hipMemcpyDeviceToDevice, s));
            continue;
        }
        // the part lies in tmp (the split pass put it there); the caller's array is its scratch and its destination
        rc = run_pipeline<K, LV, NoVal>(tmp + p.off, keys + p.off, p.cnt, kind, 0, LV, true, true, s, nullptr, nullptr, static_cast<NoVal*>(nullptr), static_cast<NoVal*>(nullptr), true);
        if (rc) return rc;
    }
    return RDST_OK;
}

template <typename K, int LV>
int sort_whole(K* keys, K* tmp, uint64_t n, rdst_key_kind kind, hipStream_t s) {
    if constexpr (sizeof(K) == 8 || sizeof(K) == 4) {
        if (split_eligible(n, sizeof(K), LV)) return run_split_sort<K, LV>(keys, tmp, n, kind, s);
    }
    return run_pipeline<K, LV>(keys, tmp, n, kind, 0, LV, true, true, s, nullptr, nullptr);
}

template <typename K, int LV>
int split_top16_t(void* dev_keys, void* dev_tmp, uint64_t len, rdst_key_kind kind, KeyMap km, uint32_t blocks, uint64_t* dev_counts16, hipStream_t s) {
    if constexpr (LV >= 2) {
        int rc = run_pipeline<K, LV>(static_cast<K*>(dev_keys), static_cast<K*>(dev_tmp), len, kind, LV - 2, LV, false, false, s, nullptr, nullptr);
        if (rc) return rc;
        hipLaunchKernelGGL((top16_counts_kernel<K>), dim3(blocks), dim3(256), 0, s, static_cast<const K*>(dev_keys), len, (K)km.neg, (K)km.pos,
                           reinterpret_cast<unsigned long long*>(dev_counts16));
        return RDST_OK;
    } else {
        return fail(RDST_ERR_UNSUPPORTED, "a 16-bit split needs keys of at least two bytes");
    }
}

// run CALL with K = the unsigned integer type of `elem_bytes` bytes and LV = its RadixKey::LEVELS
#define RDST_BY_WIDTH(elem_bytes, CALL)                                            \
    switch (elem_bytes) {                                                          \
        case 1: { using K = uint8_t; constexpr int LV = 1; CALL; } break;          \
        case 2: { using K = uint16_t; constexpr int LV = 2; CALL; } break;         \
        case 4: { using K = uint32_t; constexpr int LV = 4; CALL; } break;         \
        case 8: { using K = uint64_t; constexpr int LV = 8; CALL; } break;         \
        default: { using K = u128; constexpr int LV = 16; CALL; } break;           \
    }

int check_common(const void* p, uint64_t len, uint32_t elem_bytes, rdst_key_kind kind, uint32_t levels) {
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4 && elem_bytes != 8 && elem_bytes != 16)
        return fail(RDST_ERR_UNSUPPORTED, "device path is built for 1-, 2-, 4-, 8- and 16-byte keys");
    if (kind == RDST_KEY_FLOAT && elem_bytes != 4 && elem_bytes != 8) return fail(RDST_ERR_UNSUPPORTED, "float keys are f32 / f64");
    if (levels == 0) return fail(RDST_ERR_ARG, "RadixKey must have at least 1 level");
    if (levels != elem_bytes) return fail(RDST_ERR_ARG, "levels must equal the element width for built-in key types");
    if (kind == RDST_KEY_BYTES_BE) return fail(RDST_ERR_UNSUPPORTED, "[u8; N] keys go through the host entry point rdst_hip_sort");
    if ((int)kind < 0 || (int)kind > 3) return fail(RDST_ERR_ARG, "unknown key kind");
    if (len > 0 && p == nullptr) return fail(RDST_ERR_ARG, "null key pointer");
    if (reinterpret_cast<uintptr_t>(p) % elem_bytes) return fail(RDST_ERR_ALIGN, "key pointer not aligned to the element size");
    if (len >= (1ull << 36)) return fail(RDST_ERR_ARG, "len too large");
    return RDST_OK;
}

// ------------------------------------------------------------------------------------------
// Low-memory route (rdst_hip_sort_device_lowmem, rdst_hip_partition_device): host drivers.  Blocking by design: the swap
// plan of a Regions level is made on the host from the tile x digit counts, as the reference makes it serially
// (src/sorts/regions_sort.rs:235-239).
// ------------------------------------------------------------------------------------------
struct DeviceBuf {  // a hipMalloc that frees itself
    void* p = nullptr;
    ~DeviceBuf() { if (p) (void)hipFree(p); }
};

// Step (1) of a Regions level + its counts: every tile of `tile_len` keys grouped by digit `level` through `tmp`
// (one K3 pass tile -> tmp, copied back), the 256 digit counts of tile t in counts_host[t * 256 ..].
template <typename K, int LV>
int regions_group_tiles(K* keys, uint64_t n, rdst_key_kind kind, uint32_t level, K* tmp, uint64_t tile_len, std::vector<uint64_t>& counts_host,
                        hipStream_t s) {
    const uint64_t tiles = (n + tile_len - 1) / tile_len;
    DeviceBuf dcounts;
    HIP_TRY(hipMalloc(&dcounts.p, sizeof(uint64_t) * RADIX * tiles));
    for (uint64_t t = 0; t < tiles; ++t) {
        K* tile = keys + t * tile_len;
        const uint64_t tn = t + 1 < tiles ? tile_len : n - t * tile_len;
        Layout L;
        char* ws = nullptr;
        int rc = run_pipeline<K, LV>(tile, tmp, tn, kind, level, level + 1, false, false, s, &L, &ws);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(static_cast<uint64_t*>(dcounts.p) + t * RADIX, ws + L.off_hist + sizeof(uint64_t) * (size_t)level * RADIX,
                               sizeof(uint64_t) * RADIX, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(tile, tmp, sizeof(K) * tn, hipMemcpyDeviceToDevice, s));
        DeviceState* D;
        if ((rc = current_device_state(&D))) return rc;
        if ((rc = workspace_release(*D, s))) return rc;  // the count copy reads the workspace
    }
    counts_host.resize(RADIX * tiles);
    HIP_TRY(hipMemcpyAsync(counts_host.data(), dcounts.p, sizeof(uint64_t) * RADIX * tiles, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return RDST_OK;
}

// Steps (2)+(3): plan the swaps for the given columns and run them, round by round.  starts[buckets + 1]: region borders.
template <typename K>
int regions_swap(K* keys, uint64_t n, uint64_t tile_len, const std::vector<uint64_t>& col_counts, uint32_t columns, const uint32_t* col_bucket,
                 uint32_t buckets, std::vector<uint64_t>& starts, hipStream_t s) {
    const uint64_t tiles = (n + tile_len - 1) / tile_len;
    std::vector<rdst_swap_op> ops(4 * tiles * columns + 4 * (uint64_t)buckets * buckets + 16);
    constexpr uint32_t MAX_ROUNDS = 4096;
    std::vector<uint64_t> rounds(MAX_ROUNDS + 1);
    uint64_t nops = 0;
    uint32_t nrounds = 0;
    starts.assign(buckets + 1, 0);
    if (rdst_regions_plan(col_counts.data(), tiles, tile_len, n, columns, col_bucket, buckets, ops.data(), ops.size(), &nops, rounds.data(),
                          MAX_ROUNDS, &nrounds, starts.data()) != RDST_OK)
        return fail(RDST_ERR_DEVICE, "regions plan: the tile counts are inconsistent");
    if (nops == 0) return RDST_OK;
    std::vector<SwapChunk> chunks;
    std::vector<uint64_t> round_chunk(nrounds + 1, 0);
    for (uint32_t r = 0; r < nrounds; ++r) {
        for (uint64_t i = rounds[r]; i < rounds[r + 1]; ++i)
            for (uint64_t off = 0; off < ops[i].len; off += SWAP_CHUNK) {
                const uint64_t m = ops[i].len - off < SWAP_CHUNK ? ops[i].len - off : SWAP_CHUNK;
                chunks.push_back({ops[i].a + off, ops[i].b + off, (uint32_t)m, 0});
            }
        round_chunk[r + 1] = chunks.size();
    }
    DeviceBuf dchunks;
    HIP_TRY(hipMalloc(&dchunks.p, sizeof(SwapChunk) * chunks.size()));
    HIP_TRY(hipMemcpyAsync(dchunks.p, chunks.data(), sizeof(SwapChunk) * chunks.size(), hipMemcpyHostToDevice, s));
    for (uint32_t r = 0; r < nrounds; ++r) {
        const uint64_t c0 = round_chunk[r], c1 = round_chunk[r + 1];
        for (uint64_t c = c0; c < c1; c += (1u << 30)) {  // (a grid dimension holds 2^31 - 1 blocks)
            const uint64_t m = c1 - c < (1u << 30) ? c1 - c : (1u << 30);
            hipLaunchKernelGGL((swap_ranges_kernel<K>), dim3((uint32_t)m), dim3(256), 0, s, keys, static_cast<const SwapChunk*>(dchunks.p) + c);
        }
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(s));  // `chunks` and the device list go out of scope
    return RDST_OK;
}

// Sort keys[0, n) by levels [0, level] in place, given that all its keys agree on the levels above `level`.
template <typename K, int LV>
int lowmem_sort(K* keys, uint64_t n, rdst_key_kind kind, int level, K* tmp, uint64_t tmp_len, hipStream_t s) {
    if (n <= 1 || level < 0) return RDST_OK;
    if (n <= tmp_len)  // fits the scratch: the ordinary route on the levels that are left (a constant level is skipped by the plan)
        return run_pipeline<K, LV>(keys, tmp, n, kind, 0, (uint32_t)level + 1, true, true, s, nullptr, nullptr);
    std::vector<uint64_t> counts;
    int rc = regions_group_tiles<K, LV>(keys, n, kind, (uint32_t)level, tmp, tmp_len, counts, s);
    if (rc) return rc;
    std::vector<uint64_t> starts;
    if ((rc = regions_swap<K>(keys, n, tmp_len, counts, RADIX, nullptr, RADIX, starts, s))) return rc;
    // every digit now lies in its region.  Neighbouring regions that fit the scratch together are sorted in one call (all
    // levels up to `level`: the call spans several digits); a region that does not fit takes another Regions level.
    uint32_t d = 0;
    while (d < (uint32_t)RADIX) {
        const uint64_t lo = starts[d];
        uint64_t size = starts[d + 1] - lo;
        if (size > tmp_len) {
            if ((rc = lowmem_sort<K, LV>(keys + lo, size, kind, level - 1, tmp, tmp_len, s))) return rc;
            ++d;
            continue;
        }
        uint32_t e = d + 1;
        while (e < (uint32_t)RADIX && starts[e + 1] - lo <= tmp_len) ++e;
        size = starts[e] - lo;
        if (size > 1 && (rc = run_pipeline<K, LV>(keys + lo, tmp, size, kind, 0, (uint32_t)level + 1, true, true, s, nullptr, nullptr))) return rc;
        d = e;
    }
    return RDST_OK;
}

template <typename K, int LV>
int lowmem_entry(void* dev_keys, uint64_t len, rdst_key_kind kind, void* dev_tmp, uint64_t tmp_len, hipStream_t s) {
    return lowmem_sort<K, LV>(static_cast<K*>(dev_keys), len, kind, LV - 1, static_cast<K*>(dev_tmp), tmp_len, s);
}
template <typename K, int LV>
int partition_entry(void* dev_keys, uint64_t len, rdst_key_kind kind, uint32_t level, uint32_t digit, void* dev_tmp, uint64_t tmp_len,
                    uint64_t* split_out, hipStream_t s) {
    std::vector<uint64_t> counts;
    int rc = regions_group_tiles<K, LV>(static_cast<K*>(dev_keys), len, kind, level, static_cast<K*>(dev_tmp), tmp_len, counts, s);
    if (rc) return rc;
    // a tile lies as [digits below][the digit][digits above]: three columns, two buckets (the digit first)
    const uint64_t tiles = (len + tmp_len - 1) / tmp_len;
    std::vector<uint64_t> cols(3 * tiles, 0);
    for (uint64_t t = 0; t < tiles; ++t)
        for (uint32_t d = 0; d < (uint32_t)RADIX; ++d) cols[3 * t + (d < digit ? 0 : (d == digit ? 1 : 2))] += counts[t * RADIX + d];
    const uint32_t col_bucket[3] = {1, 0, 1};
    std::vector<uint64_t> starts;
    if ((rc = regions_swap<K>(static_cast<K*>(dev_keys), len, tmp_len, cols, 3, col_bucket, 2, starts, s))) return rc;
    *split_out = starts[1];
    return RDST_OK;
}
uint64_t usable_scratch(uint64_t tmp_len) { return tmp_len / 4096 * 4096; }  // tiles start on 16-byte boundaries for every key width

int read_device_error(DeviceState& D, hipStream_t s) {
    if (!D.err_dev) return RDST_OK;
    HIP_TRY(hipMemcpyAsync(D.host_err, D.err_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (*D.host_err != 0) {
        const uint32_t word = *D.host_err;
        HIP_TRY(hipMemsetAsync(D.err_dev, 0, sizeof(uint32_t), s));  // reported once: the next check starts clean
        HIP_TRY(hipStreamSynchronize(s));
        char b[192];
        snprintf(b, sizeof b, "device error word = 0x%x (1 = look-back spin bound expired, 2 = scatter destination out of range, 4 = hybrid-route bucket larger than a tile)", word);
        return fail(RDST_ERR_DEVICE, b);
    }
    return RDST_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

const char* rdst_hip_last_error(void) { return g_last_error.c_str(); }
int rdst_hip_abi_version(void) { return RDST_HIP_ABI_VERSION; }

int rdst_hip_set_tuning(int pass_config, int hist_blocks_per_cu) {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (pass_config >= kNumPassCfgs) return fail(RDST_ERR_ARG, "tuning value out of range");
    g_tuning.pass_cfg = pass_config >= 0 ? pass_config : -1;
    g_tuning.hist_bpc = hist_blocks_per_cu > 0 ? hist_blocks_per_cu : 0;
    return RDST_OK;
}

int rdst_hip_set_fast_rank(int enabled) {
    std::lock_guard<std::mutex> lock(g_mutex);
    g_tuning.fast_rank = enabled == 2 ? 2 : (enabled != 0 ? 1 : 0);
    return RDST_OK;
}

int rdst_hip_set_small_sort(int enabled) {
    std::lock_guard<std::mutex> lock(g_mutex);
    g_tuning.small_sort = enabled != 0;
    return RDST_OK;
}

int rdst_hip_set_hybrid(int enabled, uint64_t min_len) {
    std::lock_guard<std::mutex> lock(g_mutex);
    g_tuning.hybrid = enabled != 0;
    g_tuning.count_sort = enabled != 2;  // 2: hybrid route with the generic local sort for every key width (A/B, tests)
    g_tuning.halves = enabled != 3;      // 3: counting K4 reading whole keys (no 16-bit hand-off) (A/B, tests)
    g_tuning.presample = enabled != 5;   // 5: no sample before K1h: every hybrid-eligible sort counts all its keys' prefixes first (tests)
    g_tuning.wide2 = enabled != 6;       // 6: 8-byte keys with the one-block-per-CU form of K4 (A/B, tests)
    g_tuning.wide3 = enabled != 15;      // 15: the default with the second form of the 8-byte K4 (local_wide2_sort_kernel) (A/B, tests)
    g_tuning.atomic_route = enabled == 1 || enabled == 8;  // 1: the default (4- and 8-byte keys try the atomic route first); 2..7: the K1h hybrid route for every key width (7: with the default forms of K4)
    g_tuning.exact_msd = enabled != 12;    // 12: the default without the exact form of the MSD passes (the hybrid route's passes are K3's) (A/B, tests)
    g_tuning.atomic_route = g_tuning.atomic_route || enabled == 12;
    g_tuning.giants = enabled != 11;       // 11: the default without the giant kernels (a bucket of 65 536 keys sends the sort down the LSD route) (A/B, tests)
    g_tuning.atomic_route = g_tuning.atomic_route || enabled == 11;
    g_tuning.chain_routes = enabled != 10; // 10: the default, but a failed atomic route falls straight to the LSD route (A/B, tests)
    g_tuning.atomic_route = g_tuning.atomic_route || enabled == 10;
    g_tuning.expand = enabled != 9;        // 9: the K1h hybrid route without the expanding K4 (buckets up to one tile; refused buckets to the ranked kernel) (A/B, tests)
    g_tuning.atomic_wide = enabled != 8;   // 8: the atomic route for 4-byte keys only, 8-byte keys on the K1h hybrid route (A/B, tests)
    g_tuning.predict = enabled != 14;      // 14: the default without the sample's prediction of the LSD route (A/B, tests)
    g_tuning.split = enabled != 16;        // 16: the default without the split of 8-byte slices beyond the atomic route's window (they take the LSD route) (A/B, tests)
    g_tuning.split_always = enabled == 17; // 17: the default with that split at every length, in eight parts (tests)
    g_tuning.atomic_route = g_tuning.atomic_route || enabled == 14 || enabled == 15 || enabled == 16 || enabled == 17;
    g_tuning.hybrid_min_len = min_len;
    return RDST_OK;
}

int rdst_hip_last_route(void* stream, uint32_t* route_out) {
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceState* D;
    int rc = current_device_state(&D);
    if (rc) return rc;
    if (!route_out) return fail(RDST_ERR_ARG, "null output");
    *route_out = RDST_ROUTE_LSD;
    if (!D->ws || !D->last_plan_valid) return RDST_OK;  // no pipeline yet (or the one-workgroup sort)
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((rc = workspace_acquire(*D, s))) return rc;  // the sort may have run on another stream
    HIP_TRY(hipMemcpyAsync(D->host_err + 4, static_cast<char*>(D->ws) + D->last_plan_off + offsetof(Plan, route), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *route_out = D->host_err[4];
    return RDST_OK;
}

int rdst_hip_release_workspace(void) {
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceState* D;
    int rc = current_device_state(&D);
    if (rc) return rc;
    if (!D->ws) return RDST_OK;
    HIP_TRY(hipDeviceSynchronize());  // sorts still queued on any stream use it
    HIP_TRY(hipFree(D->ws));
    D->ws = nullptr;
    D->ws_bytes = 0;
    D->last_plan_valid = false;
    D->have_last = false;
    return RDST_OK;
}

int rdst_hip_stream_copy(void* dev_dst, const void* dev_src, uint64_t bytes, void* stream) {
    if (!dev_dst || !dev_src) return fail(RDST_ERR_ARG, "null pointer");
    if ((reinterpret_cast<uintptr_t>(dev_dst) | reinterpret_cast<uintptr_t>(dev_src) | bytes) & 15u) return fail(RDST_ERR_ALIGN, "16-byte alignment");
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceState* D;
    int rc = current_device_state(&D);
    if (rc) return rc;
    hipLaunchKernelGGL((stream_kernel<0>), dim3((uint32_t)D->cus * 8), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<u32x4_t*>(dev_dst),
                       static_cast<const u32x4_t*>(dev_src), bytes / 16, D->err_dev + 8);
    HIP_TRY(hipGetLastError());
    return RDST_OK;
}

int rdst_hip_stream_read(const void* dev_src, uint64_t bytes, void* stream) {
    if (!dev_src) return fail(RDST_ERR_ARG, "null pointer");
    if ((reinterpret_cast<uintptr_t>(dev_src) | bytes) & 15u) return fail(RDST_ERR_ALIGN, "16-byte alignment");
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceState* D;
    int rc = current_device_state(&D);
    if (rc) return rc;
    hipLaunchKernelGGL((stream_kernel<1>), dim3((uint32_t)D->cus * 8), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<u32x4_t*>(nullptr),
                       static_cast<const u32x4_t*>(dev_src), bytes / 16, D->err_dev + 8);
    HIP_TRY(hipGetLastError());
    return RDST_OK;
}

int rdst_hip_stream_fill(void* dev_dst, uint64_t bytes, void* stream) {
    if (!dev_dst) return fail(RDST_ERR_ARG, "null pointer");
    if ((reinterpret_cast<uintptr_t>(dev_dst) | bytes) & 15u) return fail(RDST_ERR_ALIGN, "16-byte alignment");
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceState* D;
    int rc = current_device_state(&D);
    if (rc) return rc;
    hipLaunchKernelGGL((stream_kernel<2>), dim3((uint32_t)D->cus * 8), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<u32x4_t*>(dev_dst),
                       static_cast<const u32x4_t*>(nullptr), bytes / 16, D->err_dev + 8);
    HIP_TRY(hipGetLastError());
    return RDST_OK;
}

int rdst_hip_debug_raise_device_error(uint32_t bits, void* stream) {
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceState* D;
    int rc = current_device_state(&D);
    if (rc) return rc;
    hipLaunchKernelGGL(raise_error_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), D->err_dev, bits);
    HIP_TRY(hipGetLastError());
    return RDST_OK;
}

int rdst_hip_set_chain_split(int enabled) {
    std::lock_guard<std::mutex> lock(g_mutex);
    g_tuning.chains = enabled != 0;
    return RDST_OK;
}

#ifdef RDST_EXPERIMENTS
int rdst_hip_exp_set_ablation(uint32_t mask) { g_ablate = mask; return 0; }
int rdst_hip_exp_set_lds(uint32_t bytes) { g_exp_lds_total = bytes; return 0; }
// per-tile look-back records of level 0: windows fetched, blocked re-polls, tiles consumed, shader clocks
int rdst_hip_exp_timeline(uint32_t* host_out, uint64_t tiles) {
    static uint32_t* dev = nullptr;
    static uint64_t cap = 0;
    if (host_out == nullptr) {
        if (cap < tiles) {
            if (dev) (void)hipFree(dev);
            if (hipMalloc((void**)&dev, tiles * 48) != hipSuccess) return -1;
            cap = tiles;
        }
        if (hipMemset(dev, 0, tiles * 48) != hipSuccess) return -1;
        return hipMemcpyToSymbol(HIP_SYMBOL(g_exp_timeline), &dev, sizeof dev) == hipSuccess ? 0 : -1;
    }
    if (hipDeviceSynchronize() != hipSuccess || !dev) return -1;
    uint32_t* null_dev = nullptr;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_exp_timeline), &null_dev, sizeof null_dev);
    return hipMemcpy(host_out, dev, tiles * 48, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
int rdst_hip_exp_timeline_select(uint32_t which) {  // 0 K3 level 0, 1 local_wide2_sort_kernel, 2 / 3 msd_scatter_kernel pass A / B
    return hipMemcpyToSymbol(HIP_SYMBOL(g_exp_timeline_kernel), &which, sizeof which) == hipSuccess ? 0 : -1;
}
int rdst_hip_exp_stats(uint32_t* host_out, uint64_t tiles) {
    static uint32_t* dev = nullptr;
    static uint64_t cap = 0;
    if (host_out == nullptr) {  // arm: (re)allocate and clear
        if (cap < tiles) {
            if (dev) (void)hipFree(dev);
            if (hipMalloc((void**)&dev, tiles * 16) != hipSuccess) return -1;
            cap = tiles;
        }
        if (hipMemset(dev, 0, tiles * 16) != hipSuccess) return -1;
        return hipMemcpyToSymbol(HIP_SYMBOL(g_exp_stats), &dev, sizeof dev) == hipSuccess ? 0 : -1;
    }
    if (hipDeviceSynchronize() != hipSuccess || !dev) return -1;
    return hipMemcpy(host_out, dev, tiles * 16, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

int rdst_hip_set_profiling(int enabled) {
    std::lock_guard<std::mutex> lock(g_mutex);
    g_tuning.profiling = enabled != 0;
    for (auto& D : g_dev) {  // (re-)enabling starts a fresh record list
        D.prof_used = 0;
        D.prof_runs.clear();
    }
    return RDST_OK;
}

int rdst_hip_profile_runs(void) {
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceState* D;
    if (current_device_state(&D)) return 0;
    return (int)D->prof_runs.size();
}

int rdst_hip_profile_run(int run, float* out_ms, uint32_t capacity, uint32_t* n_out) {
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceState* D;
    int rc = current_device_state(&D);
    if (rc) return rc;
    if (!out_ms || !n_out) return fail(RDST_ERR_ARG, "null output");
    *n_out = 0;
    if (run < 0) run += (int)D->prof_runs.size();  // -1 = most recent
    if (run < 0 || run >= (int)D->prof_runs.size()) return fail(RDST_ERR_ARG, "no such profiled run");
    const auto r = D->prof_runs[run];
    if (r.count < 2) return RDST_OK;
    HIP_TRY(hipEventSynchronize(D->prof_events[r.begin + r.count - 1]));
    const uint32_t n = r.count - 1;
    for (uint32_t i = 0; i < n && i < capacity; ++i)
        HIP_TRY(hipEventElapsedTime(&out_ms[i], D->prof_events[r.begin + i], D->prof_events[r.begin + i + 1]));
    *n_out = n < capacity ? n : capacity;
    return RDST_OK;
}

int rdst_hip_profile_run_stages(int run, uint32_t* stages_out, uint32_t capacity, uint32_t* n_out) {
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceState* D;
    int rc = current_device_state(&D);
    if (rc) return rc;
    if (!stages_out || !n_out) return fail(RDST_ERR_ARG, "null output");
    *n_out = 0;
    if (run < 0) run += (int)D->prof_runs.size();
    if (run < 0 || run >= (int)D->prof_runs.size()) return fail(RDST_ERR_ARG, "no such profiled run");
    const auto r = D->prof_runs[run];
    if (r.count < 2) return RDST_OK;
    const uint32_t n = r.count - 1;
    for (uint32_t i = 0; i < n && i < capacity; ++i) stages_out[i] = D->prof_kinds[r.begin + i];
    *n_out = n < capacity ? n : capacity;
    return RDST_OK;
}

uint64_t rdst_hip_workspace_bytes(uint64_t len, uint32_t elem_bytes) {
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4 && elem_bytes != 8 && elem_bytes != 16) return 0;
    int cfg = g_tuning.pass_cfg;
    if (cfg < 0 || cfg >= kNumPassCfgs) cfg = default_cfg(elem_bytes, len, true);  // the shape with the smaller tiles: an upper bound for every key kind
    const bool msd = atomic_eligible(len, elem_bytes, cfg);
    const bool halves = !msd && elem_bytes == 4 && hybrid_eligible(len, 4) && g_tuning.halves && g_tuning.count_sort && cfg == 4 && len < (1ull << 30);
    const bool giants = elem_bytes == 4 && hybrid_eligible(len, 4) && g_tuning.giants && g_tuning.count_sort && g_tuning.expand && len < (1ull << 30);
    const uint64_t whole = make_layout(len, elem_bytes, elem_bytes, cfg, 0, halves, msd, giants).total;
    if (split_eligible(len, elem_bytes, (int)elem_bytes)) {  // the split pass's tables for the whole slice, then a part's areas and slots (a part is at most the window)
        uint64_t part = len < (1ull << 30) ? len : (1ull << 30) - 1;
        while (part > (1ull << 26) && !atomic_eligible(part, elem_bytes, default_cfg(elem_bytes, part, true))) part -= part / 64;
        const uint64_t of_part = make_layout(part, elem_bytes, elem_bytes, default_cfg(elem_bytes, part, true), 0, false, true, false).total;
        return whole > of_part ? whole : of_part;
    }
    return whole;
}

int rdst_hip_sort_device(void* dev_keys, void* dev_tmp, uint64_t len, uint32_t elem_bytes, rdst_key_kind kind,
                         uint32_t levels, void* stream) {
    int rc = check_common(dev_keys, len, elem_bytes, kind, levels);
    if (rc) return rc;
    if (len <= 1) return RDST_OK;  // radix_sort_builder.rs:151
    if (dev_tmp == nullptr) return fail(RDST_ERR_ARG, "null tmp pointer");
    if (reinterpret_cast<uintptr_t>(dev_tmp) % elem_bytes) return fail(RDST_ERR_ALIGN, "tmp pointer not aligned to the element size");
    std::lock_guard<std::mutex> lock(g_mutex);
    hipStream_t s = static_cast<hipStream_t>(stream);
    RDST_BY_WIDTH(elem_bytes, rc = (sort_whole<K, LV>(static_cast<K*>(dev_keys), static_cast<K*>(dev_tmp), len, kind, s)));
    return rc;
}

int rdst_hip_sort_pairs_device(void* dev_keys, void* dev_vals, void* dev_tmp_keys, void* dev_tmp_vals, uint64_t len,
                               uint32_t key_bytes, rdst_key_kind kind, uint32_t levels, uint32_t val_bytes, void* stream) {
    int rc = check_common(dev_keys, len, key_bytes, kind, levels);
    if (rc) return rc;
    if (key_bytes != 4 && key_bytes != 8) return fail(RDST_ERR_UNSUPPORTED, "key-value sorts take 4- or 8-byte keys");
    if (val_bytes != 4 && val_bytes != 8) return fail(RDST_ERR_UNSUPPORTED, "key-value sorts carry 4- or 8-byte values");
    if (len <= 1) return RDST_OK;
    if (!dev_vals || !dev_tmp_keys || !dev_tmp_vals) return fail(RDST_ERR_ARG, "null value / tmp pointer");
    if (reinterpret_cast<uintptr_t>(dev_tmp_keys) % key_bytes) return fail(RDST_ERR_ALIGN, "tmp key pointer not aligned to the key size");
    if ((reinterpret_cast<uintptr_t>(dev_vals) | reinterpret_cast<uintptr_t>(dev_tmp_vals)) % val_bytes)
        return fail(RDST_ERR_ALIGN, "value pointer not aligned to the value size");
    std::lock_guard<std::mutex> lock(g_mutex);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define RDST_PAIRS(KT, LV, VT)                                                                                          \
    rc = (run_pipeline<KT, LV, VT>(static_cast<KT*>(dev_keys), static_cast<KT*>(dev_tmp_keys), len, kind, 0, LV, true, true, s, \
                                   nullptr, nullptr, static_cast<VT*>(dev_vals), static_cast<VT*>(dev_tmp_vals)))
    if (key_bytes == 4) {
        if (val_bytes == 4) RDST_PAIRS(uint32_t, 4, uint32_t);
        else RDST_PAIRS(uint32_t, 4, uint64_t);
    } else {
        if (val_bytes == 4) RDST_PAIRS(uint64_t, 8, uint32_t);
        else RDST_PAIRS(uint64_t, 8, uint64_t);
    }
#undef RDST_PAIRS
    return rc;
}

int rdst_hip_device_status(void* stream) {
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceState* D;
    int rc = current_device_state(&D);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipStreamSynchronize(s));
    return read_device_error(*D, s);
}

namespace {
// [u8; N] host slice: H2D of the raw bytes, expand to W-byte integers, sort those, compact, D2H
int sort_byte_keys_host(void* host_data, uint64_t len, uint32_t nb, const rdst_hip_opts* opts) {
    int prev_dev = -1;
    if (opts && opts->device >= 0) {
        HIP_TRY(hipGetDevice(&prev_dev));
        HIP_TRY(hipSetDevice(opts->device));
    }
    const uint32_t w = nb <= 4 ? 4 : (nb <= 8 ? 8 : 16);
    void *d_raw = nullptr, *d_keys = nullptr, *d_tmp = nullptr;
    hipStream_t s = nullptr;
    auto cleanup = [&]() {
        for (void* p : {d_raw, d_keys, d_tmp})
            if (p) (void)hipFree(p);
        if (s) (void)hipStreamDestroy(s);
        if (prev_dev >= 0) (void)hipSetDevice(prev_dev);
    };
    hipError_t e;
#define RDST_B_TRY(expr) if ((e = (expr)) != hipSuccess) { if (s) (void)hipStreamSynchronize(s); cleanup(); return fail(RDST_ERR_HIP, #expr, e); }
    RDST_B_TRY(hipStreamCreate(&s));
    RDST_B_TRY(hipMalloc(&d_raw, (size_t)len * nb));
    RDST_B_TRY(hipMalloc(&d_keys, (size_t)len * w));
    RDST_B_TRY(hipMalloc(&d_tmp, (size_t)len * w));
    RDST_B_TRY(hipMemcpyAsync(d_raw, host_data, (size_t)len * nb, hipMemcpyHostToDevice, s));
    uint64_t blocks = (len + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const unsigned char* raw = static_cast<const unsigned char*>(d_raw);
    if (w == 4) hipLaunchKernelGGL((bytes_expand_kernel<uint32_t>), dim3((uint32_t)blocks), dim3(256), 0, s, raw, static_cast<uint32_t*>(d_keys), len, nb);
    else if (w == 8) hipLaunchKernelGGL((bytes_expand_kernel<uint64_t>), dim3((uint32_t)blocks), dim3(256), 0, s, raw, static_cast<uint64_t*>(d_keys), len, nb);
    else hipLaunchKernelGGL((bytes_expand_kernel<u128>), dim3((uint32_t)blocks), dim3(256), 0, s, raw, static_cast<u128*>(d_keys), len, nb);
    RDST_B_TRY(hipGetLastError());
    int rc = rdst_hip_sort_device(d_keys, d_tmp, len, w, RDST_KEY_UNSIGNED, w, s);
    if (rc == RDST_OK) rc = rdst_hip_device_status(s);
    if (rc != RDST_OK) { (void)hipStreamSynchronize(s); cleanup(); return rc; }
    unsigned char* rawo = static_cast<unsigned char*>(d_raw);
    if (w == 4) hipLaunchKernelGGL((bytes_compact_kernel<uint32_t>), dim3((uint32_t)blocks), dim3(256), 0, s, static_cast<const uint32_t*>(d_keys), rawo, len, nb);
    else if (w == 8) hipLaunchKernelGGL((bytes_compact_kernel<uint64_t>), dim3((uint32_t)blocks), dim3(256), 0, s, static_cast<const uint64_t*>(d_keys), rawo, len, nb);
    else hipLaunchKernelGGL((bytes_compact_kernel<u128>), dim3((uint32_t)blocks), dim3(256), 0, s, static_cast<const u128*>(d_keys), rawo, len, nb);
    RDST_B_TRY(hipGetLastError());
    // the host buffer is written only now, after the device reported success
    RDST_B_TRY(hipMemcpyAsync(host_data, d_raw, (size_t)len * nb, hipMemcpyDeviceToHost, s));
    RDST_B_TRY(hipStreamSynchronize(s));
#undef RDST_B_TRY
    cleanup();
    return RDST_OK;
}
}  // namespace

int rdst_hip_sort(void* host_data, uint64_t len, uint32_t elem_bytes, rdst_key_kind kind, uint32_t levels,
                  const rdst_hip_opts* opts) {
    if (kind == RDST_KEY_BYTES_BE) {  // [u8; N], src/radix_key_impl.rs:78-85
        if (elem_bytes == 0 || elem_bytes > 16) return fail(RDST_ERR_UNSUPPORTED, "[u8; N] keys are built for N in 1..16");
        if (levels != elem_bytes) return fail(RDST_ERR_ARG, "levels must equal N for [u8; N]");
        if (len > 0 && host_data == nullptr) return fail(RDST_ERR_ARG, "null key pointer");
        if (len >= (1ull << 36)) return fail(RDST_ERR_ARG, "len too large");
        if (len <= 1) return RDST_OK;
        return sort_byte_keys_host(host_data, len, elem_bytes, opts);
    }
    int rc = check_common(host_data, len, elem_bytes, kind, levels);
    if (rc) return rc;
    if (len <= 1) return RDST_OK;
    int prev_dev = -1;
    if (opts && opts->device >= 0) {
        HIP_TRY(hipGetDevice(&prev_dev));
        HIP_TRY(hipSetDevice(opts->device));
    }
    const size_t bytes = (size_t)len * elem_bytes;
    const bool low_memory = opts && opts->low_memory != 0;
    // low-memory route: the scratch is len / 64 elements (at least 65 536) instead of a second array
    const uint64_t scratch_len = low_memory ? (len / 64 > 65536 ? len / 64 : 65536) : len;
    const size_t half = align_up(bytes, 256);
    const size_t second = low_memory ? align_up((size_t)scratch_len * elem_bytes, 256) : half;
    // Stream and device buffer (keys + tmp) are kept per device: hipMalloc / hipFree pairs and a stream
    // per call were most of the 0.55 ms this entry point cost on small slices.  (A stream-ordered pool
    // allocation per call was tried first: with the system HIP runtime a sort whose buffer the pool
    // had just recycled read stale keys — test_cpp_mirror — so the buffer is plain hipMalloc memory.)
    DeviceState* D = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        rc = current_device_state(&D);
    }
    if (rc != RDST_OK) { if (prev_dev >= 0) (void)hipSetDevice(prev_dev); return rc; }
    std::lock_guard<std::mutex> host_lock(D->host_mutex);
    auto done = [&](int code) { if (prev_dev >= 0) (void)hipSetDevice(prev_dev); return code; };
    hipError_t e;
    if (!D->host_stream && (e = hipStreamCreateWithFlags(&D->host_stream, hipStreamNonBlocking)) != hipSuccess)
        return done(fail(RDST_ERR_HIP, "hipStreamCreate", e));
    hipStream_t s = D->host_stream;
    // (The allocation scheme of commit 74cafa2 — one stream-ordered pool allocation per call — produced the round-1 abort
    // examined in DESIGN.md §5; it exists only in the tools build, behind RDST_HOST_ALLOC=pool*, never in the product.)
#ifdef RDST_EXPERIMENTS
    static const char* alloc_mode = getenv("RDST_HOST_ALLOC");
    static const bool use_pool = alloc_mode && strncmp(alloc_mode, "pool", 4) == 0;
    static const bool pool_sync = alloc_mode && strcmp(alloc_mode, "pool_sync") == 0;    // + stream sync between H2D and the sort
    static const bool host_debug = getenv("RDST_HOST_DEBUG") != nullptr;                // print the buffers of every call
    static const bool pool_nofree = alloc_mode && strcmp(alloc_mode, "pool_nofree") == 0;  // blocks are never returned: no recycling
#else
    constexpr bool use_pool = false, pool_sync = false, host_debug = false, pool_nofree = false;
#endif
    void* pool_buf = nullptr;
    if (use_pool) {
        if ((e = hipMallocAsync(&pool_buf, half + second, s)) != hipSuccess) return done(fail(RDST_ERR_HIP, "hipMallocAsync(keys + tmp)", e));
    } else if (D->host_buf_bytes < half + second) {
        if (D->host_buf) { (void)hipStreamSynchronize(s); (void)hipFree(D->host_buf); D->host_buf = nullptr; D->host_buf_bytes = 0; }
        const size_t want = half + second < (size_t)(64u << 20) ? (half + second) + (half + second) / 2 : half + second;  // head room for small slices only
        if ((e = hipMalloc(&D->host_buf, want)) != hipSuccess) return done(fail(RDST_ERR_HIP, "hipMalloc(keys + tmp)", e));
        D->host_buf_bytes = want;
    }
    void* d_keys = use_pool ? pool_buf : D->host_buf;
    void* d_tmp = static_cast<char*>(d_keys) + half;
    struct PoolGuard { void* p; hipStream_t s; ~PoolGuard() { if (p) (void)hipFreeAsync(p, s); } } pool_guard{pool_nofree ? nullptr : pool_buf, s};
    D->host_timed = false;
    for (auto& ev : D->host_ev)
        if (!ev && (e = hipEventCreate(&ev)) != hipSuccess) return done(fail(RDST_ERR_HIP, "hipEventCreate", e));
    (void)hipEventRecord(D->host_ev[0], s);
    if ((e = hipMemcpyAsync(d_keys, host_data, bytes, hipMemcpyHostToDevice, s)) != hipSuccess) { (void)hipStreamSynchronize(s); return done(fail(RDST_ERR_HIP, "H2D", e)); }
    (void)hipEventRecord(D->host_ev[1], s);
    if (pool_sync) (void)hipStreamSynchronize(s);
    if (host_debug) {
        fprintf(stderr, "[host] len=%llu elem=%u keys=[%p,%p) tmp=[%p,%p) ws=[%p,%p)\n", (unsigned long long)len, elem_bytes, d_keys,
                (void*)(static_cast<char*>(d_keys) + bytes), d_tmp, (void*)(static_cast<char*>(d_tmp) + bytes), D->ws,
                (void*)(static_cast<char*>(D->ws) + D->ws_bytes));
        fflush(stderr);
    }
    if (low_memory) rc = rdst_hip_sort_device_lowmem(d_keys, len, elem_bytes, kind, levels, d_tmp, scratch_len, s);
    else rc = rdst_hip_sort_device(d_keys, d_tmp, len, elem_bytes, kind, levels, s);
    if (host_debug) {
        fprintf(stderr, "[host]   after enqueue: ws=[%p,%p) rc=%d\n", D->ws, (void*)(static_cast<char*>(D->ws) + D->ws_bytes), rc);
        fflush(stderr);
    }
    if (rc == RDST_OK) rc = rdst_hip_device_status(s);
    if (rc != RDST_OK) { (void)hipStreamSynchronize(s); return done(rc); }
    // the host buffer is written only now, after the device reported success
    (void)hipEventRecord(D->host_ev[2], s);
    if ((e = hipMemcpyAsync(host_data, d_keys, bytes, hipMemcpyDeviceToHost, s)) != hipSuccess) { (void)hipStreamSynchronize(s); return done(fail(RDST_ERR_HIP, "D2H", e)); }
    (void)hipEventRecord(D->host_ev[3], s);
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return done(fail(RDST_ERR_HIP, "sync", e));
    D->host_timed = true;
    if (D->host_buf_bytes > ((size_t)1 << 30)) {  // do not sit on gigabytes between calls
        (void)hipFree(D->host_buf);
        D->host_buf = nullptr;
        D->host_buf_bytes = 0;
    }
    return done(RDST_OK);
}

int rdst_hip_host_timing(float* h2d_ms, float* sort_ms, float* d2h_ms) {
    DeviceState* D = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        int rc = current_device_state(&D);
        if (rc) return rc;
    }
    std::lock_guard<std::mutex> host_lock(D->host_mutex);
    if (!D->host_timed) return fail(RDST_ERR_ARG, "no host-slice sort has completed on this device yet");
    float t[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i) HIP_TRY(hipEventElapsedTime(&t[i], D->host_ev[i], D->host_ev[i + 1]));
    if (h2d_ms) *h2d_ms = t[0];
    if (sort_ms) *sort_ms = t[1];
    if (d2h_ms) *d2h_ms = t[2];
    return RDST_OK;
}

int rdst_hip_sort_records(void* host_records, uint64_t len, uint32_t record_bytes, uint32_t key_offset, uint32_t key_bytes,
                          rdst_key_kind kind, const rdst_hip_opts* opts) {
    if (key_bytes != 4 && key_bytes != 8) return fail(RDST_ERR_UNSUPPORTED, "record sorts take a 4- or 8-byte key field");
    if (kind == RDST_KEY_FLOAT || kind == RDST_KEY_SIGNED || kind == RDST_KEY_UNSIGNED) {} else return fail(RDST_ERR_ARG, "unknown key kind");
    if (record_bytes == 0 || (uint64_t)key_offset + key_bytes > record_bytes) return fail(RDST_ERR_ARG, "key field outside the record");
    if (key_offset % key_bytes || record_bytes % key_bytes) return fail(RDST_ERR_ALIGN, "key field not naturally aligned inside the record");
    if (len > 0 && host_records == nullptr) return fail(RDST_ERR_ARG, "null record pointer");
    if (len >= (1ull << 36)) return fail(RDST_ERR_ARG, "len too large");
    if (len <= 1) return RDST_OK;
    int prev_dev = -1;
    if (opts && opts->device >= 0) {
        HIP_TRY(hipGetDevice(&prev_dev));
        HIP_TRY(hipSetDevice(opts->device));
    }
    const size_t bytes = (size_t)len * record_bytes;
    const uint32_t idx_bytes = len < (1ull << 32) ? 4 : 8;
    void *d_rec = nullptr, *d_out = nullptr, *d_keys = nullptr, *d_tk = nullptr, *d_idx = nullptr, *d_ti = nullptr;
    hipStream_t s = nullptr;
    auto cleanup = [&]() {
        for (void* p : {d_rec, d_out, d_keys, d_tk, d_idx, d_ti})
            if (p) (void)hipFree(p);
        if (s) (void)hipStreamDestroy(s);
        if (prev_dev >= 0) (void)hipSetDevice(prev_dev);
    };
    hipError_t e;
#define RDST_REC_TRY(expr) if ((e = (expr)) != hipSuccess) { if (s) (void)hipStreamSynchronize(s); cleanup(); return fail(RDST_ERR_HIP, #expr, e); }
    RDST_REC_TRY(hipStreamCreate(&s));
    RDST_REC_TRY(hipMalloc(&d_rec, bytes));
    RDST_REC_TRY(hipMalloc(&d_out, bytes));
    RDST_REC_TRY(hipMalloc(&d_keys, (size_t)len * key_bytes));
    RDST_REC_TRY(hipMalloc(&d_tk, (size_t)len * key_bytes));
    RDST_REC_TRY(hipMalloc(&d_idx, (size_t)len * idx_bytes));
    RDST_REC_TRY(hipMalloc(&d_ti, (size_t)len * idx_bytes));
    RDST_REC_TRY(hipMemcpyAsync(d_rec, host_records, bytes, hipMemcpyHostToDevice, s));
    uint64_t blocks = (len + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const unsigned char* rec = static_cast<const unsigned char*>(d_rec);
    if (key_bytes == 4) {
        if (idx_bytes == 4) hipLaunchKernelGGL((extract_key_kernel<uint32_t, uint32_t>), dim3((uint32_t)blocks), dim3(256), 0, s, rec, len, record_bytes, key_offset, static_cast<uint32_t*>(d_keys), static_cast<uint32_t*>(d_idx));
        else hipLaunchKernelGGL((extract_key_kernel<uint32_t, uint64_t>), dim3((uint32_t)blocks), dim3(256), 0, s, rec, len, record_bytes, key_offset, static_cast<uint32_t*>(d_keys), static_cast<uint64_t*>(d_idx));
    } else {
        if (idx_bytes == 4) hipLaunchKernelGGL((extract_key_kernel<uint64_t, uint32_t>), dim3((uint32_t)blocks), dim3(256), 0, s, rec, len, record_bytes, key_offset, static_cast<uint64_t*>(d_keys), static_cast<uint32_t*>(d_idx));
        else hipLaunchKernelGGL((extract_key_kernel<uint64_t, uint64_t>), dim3((uint32_t)blocks), dim3(256), 0, s, rec, len, record_bytes, key_offset, static_cast<uint64_t*>(d_keys), static_cast<uint64_t*>(d_idx));
    }
    RDST_REC_TRY(hipGetLastError());
    int rc = rdst_hip_sort_pairs_device(d_keys, d_idx, d_tk, d_ti, len, key_bytes, kind, key_bytes, idx_bytes, s);
    if (rc == RDST_OK) rc = rdst_hip_device_status(s);
    if (rc != RDST_OK) { (void)hipStreamSynchronize(s); cleanup(); return rc; }
    // rows in the order of the sorted indices, in the widest units the row size allows
    const uint32_t unit = record_bytes % 16 == 0 ? 16 : (record_bytes % 8 == 0 ? 8 : 4);
    const uint64_t total_units = len * (record_bytes / unit);
    uint64_t gblocks = (total_units + 255) / 256;
    if (gblocks > 256 * 32) gblocks = 256 * 32;
    struct alignas(16) U16 { uint64_t a, b; };
#define RDST_GATHER(UT, IT) hipLaunchKernelGGL((gather_records_kernel<UT, IT>), dim3((uint32_t)gblocks), dim3(256), 0, s, static_cast<const UT*>(d_rec), static_cast<UT*>(d_out), static_cast<const IT*>(d_idx), len, record_bytes / unit)
    if (idx_bytes == 4) {
        if (unit == 16) RDST_GATHER(U16, uint32_t); else if (unit == 8) RDST_GATHER(uint64_t, uint32_t); else RDST_GATHER(uint32_t, uint32_t);
    } else {
        if (unit == 16) RDST_GATHER(U16, uint64_t); else if (unit == 8) RDST_GATHER(uint64_t, uint64_t); else RDST_GATHER(uint32_t, uint64_t);
    }
#undef RDST_GATHER
    RDST_REC_TRY(hipGetLastError());
    // the host buffer is written only now, after the device reported success
    RDST_REC_TRY(hipMemcpyAsync(host_records, d_out, bytes, hipMemcpyDeviceToHost, s));
    RDST_REC_TRY(hipStreamSynchronize(s));
#undef RDST_REC_TRY
    cleanup();
    return RDST_OK;
}

int rdst_hip_all_level_counts(const void* dev_keys, uint64_t len, uint32_t elem_bytes, rdst_key_kind kind,
                              uint32_t levels, uint64_t* counts_out, void* stream) {
    int rc = check_common(dev_keys, len, elem_bytes, kind, levels);
    if (rc) return rc;
    if (!counts_out) return fail(RDST_ERR_ARG, "null counts_out");
    memset(counts_out, 0, sizeof(uint64_t) * levels * RADIX);
    if (len == 0) return RDST_OK;
    std::lock_guard<std::mutex> lock(g_mutex);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Layout L;
    char* ws = nullptr;
    // levels [0,0): histogram + scan only, no pass runs
    RDST_BY_WIDTH(elem_bytes, rc = (run_pipeline<K, LV>(const_cast<K*>(static_cast<const K*>(dev_keys)), nullptr, len, kind, 0, 0, false, false, s, &L, &ws)));
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(counts_out, ws + L.off_hist, sizeof(uint64_t) * levels * RADIX, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return RDST_OK;
}

int rdst_hip_scatter_level(const void* dev_src, void* dev_dst, uint64_t len, uint32_t elem_bytes, rdst_key_kind kind,
                           uint32_t level, uint64_t* counts_out, void* stream) {
    int rc = check_common(dev_src, len, elem_bytes, kind, elem_bytes);
    if (rc) return rc;
    if (level >= elem_bytes) return fail(RDST_ERR_ARG, "level out of range");
    if (counts_out) memset(counts_out, 0, sizeof(uint64_t) * RADIX);
    if (len == 0) return RDST_OK;
    if (dev_dst == nullptr) return fail(RDST_ERR_ARG, "null dst pointer");
    if (reinterpret_cast<uintptr_t>(dev_dst) % elem_bytes) return fail(RDST_ERR_ALIGN, "dst pointer not aligned to the element size");
    std::lock_guard<std::mutex> lock(g_mutex);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Layout L;
    char* ws = nullptr;
    // one un-skippable pass keys -> tmp, no copy-back: src is only read
    RDST_BY_WIDTH(elem_bytes, rc = (run_pipeline<K, LV>(const_cast<K*>(static_cast<const K*>(dev_src)), static_cast<K*>(dev_dst), len, kind, level, level + 1, false, false, s, &L, &ws)));
    if (rc) return rc;
    if (counts_out)
        HIP_TRY(hipMemcpyAsync(counts_out, ws + L.off_hist + sizeof(uint64_t) * (size_t)level * RADIX, sizeof(uint64_t) * RADIX, hipMemcpyDeviceToHost, s));
    DeviceState* D;
    rc = current_device_state(&D);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    return read_device_error(*D, s);
}

int rdst_hip_sort_device_lowmem(void* dev_keys, uint64_t len, uint32_t elem_bytes, rdst_key_kind kind, uint32_t levels, void* dev_tmp,
                                uint64_t tmp_len, void* stream) {
    int rc = check_common(dev_keys, len, elem_bytes, kind, levels);
    if (rc) return rc;
    if (len <= 1) return RDST_OK;
    if (dev_tmp == nullptr) return fail(RDST_ERR_ARG, "null tmp pointer");
    if (reinterpret_cast<uintptr_t>(dev_tmp) % elem_bytes) return fail(RDST_ERR_ALIGN, "tmp pointer not aligned to the element size");
    tmp_len = usable_scratch(tmp_len);
    if (tmp_len < 65536) return fail(RDST_ERR_ARG, "the scratch must hold at least 65536 elements");
    std::lock_guard<std::mutex> lock(g_mutex);
    hipStream_t s = static_cast<hipStream_t>(stream);
    RDST_BY_WIDTH(elem_bytes, rc = (lowmem_entry<K, LV>(dev_keys, len, kind, dev_tmp, tmp_len, s)));
    if (rc) return rc;
    DeviceState* D;
    if ((rc = current_device_state(&D))) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    return read_device_error(*D, s);
}

int rdst_hip_partition_device(void* dev_keys, uint64_t len, uint32_t elem_bytes, rdst_key_kind kind, uint32_t level, uint32_t digit,
                              void* dev_tmp, uint64_t tmp_len, uint64_t* split_out, void* stream) {
    int rc = check_common(dev_keys, len, elem_bytes, kind, elem_bytes);
    if (rc) return rc;
    if (level >= elem_bytes || digit > 255) return fail(RDST_ERR_ARG, "level / digit out of range");
    if (!split_out) return fail(RDST_ERR_ARG, "null split_out");
    *split_out = 0;
    if (len == 0) return RDST_OK;
    if (dev_tmp == nullptr) return fail(RDST_ERR_ARG, "null tmp pointer");
    if (reinterpret_cast<uintptr_t>(dev_tmp) % elem_bytes) return fail(RDST_ERR_ALIGN, "tmp pointer not aligned to the element size");
    tmp_len = usable_scratch(tmp_len);
    if (tmp_len < 65536) return fail(RDST_ERR_ARG, "the scratch must hold at least 65536 elements");
    std::lock_guard<std::mutex> lock(g_mutex);
    hipStream_t s = static_cast<hipStream_t>(stream);
    RDST_BY_WIDTH(elem_bytes, rc = (partition_entry<K, LV>(dev_keys, len, kind, level, digit, dev_tmp, tmp_len, split_out, s)));
    if (rc) return rc;
    DeviceState* D;
    if ((rc = current_device_state(&D))) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    return read_device_error(*D, s);
}

int rdst_hip_split_top_level_device(const void* dev_src, void* dev_dst, uint64_t len, uint32_t elem_bytes, rdst_key_kind kind,
                                    uint64_t* dev_counts, void* stream) {
    int rc = check_common(dev_src, len, elem_bytes, kind, elem_bytes);
    if (rc) return rc;
    if (!dev_counts) return fail(RDST_ERR_ARG, "null counts pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (len == 0) {
        HIP_TRY(hipMemsetAsync(dev_counts, 0, sizeof(uint64_t) * RADIX, s));
        return RDST_OK;
    }
    if (dev_dst == nullptr) return fail(RDST_ERR_ARG, "null dst pointer");
    if (reinterpret_cast<uintptr_t>(dev_dst) % elem_bytes) return fail(RDST_ERR_ALIGN, "dst pointer not aligned to the element size");
    std::lock_guard<std::mutex> lock(g_mutex);
    Layout L;
    char* ws = nullptr;
    const uint32_t top = elem_bytes - 1;
    // one un-skippable pass src -> dst on the top level; its K1 counts that level only; nothing blocks
    RDST_BY_WIDTH(elem_bytes, rc = (run_pipeline<K, LV>(const_cast<K*>(static_cast<const K*>(dev_src)), static_cast<K*>(dev_dst), len, kind, top, top + 1, false, false, s, &L, &ws)));
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(dev_counts, ws + L.off_hist + sizeof(uint64_t) * (size_t)top * RADIX, sizeof(uint64_t) * RADIX, hipMemcpyDeviceToDevice, s));
    DeviceState* D;
    if ((rc = current_device_state(&D))) return rc;
    return workspace_release(*D, s);  // the copy reads the workspace: later calls on other streams wait for it too
}

int rdst_hip_split_top16_device(void* dev_keys, void* dev_tmp, uint64_t len, uint32_t elem_bytes, rdst_key_kind kind,
                                uint64_t* dev_counts16, void* stream) {
    int rc = check_common(dev_keys, len, elem_bytes, kind, elem_bytes);
    if (rc) return rc;
    if (elem_bytes < 2) return fail(RDST_ERR_UNSUPPORTED, "a 16-bit split needs keys of at least two bytes");
    if (!dev_counts16) return fail(RDST_ERR_ARG, "null counts pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(dev_counts16, 0, sizeof(uint64_t) * H16_BINS, s));
    if (len == 0) return RDST_OK;
    if (dev_tmp == nullptr) return fail(RDST_ERR_ARG, "null tmp pointer");
    if (reinterpret_cast<uintptr_t>(dev_tmp) % elem_bytes) return fail(RDST_ERR_ALIGN, "tmp pointer not aligned to the element size");
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceState* D;
    if ((rc = current_device_state(&D))) return rc;
    const KeyMap km = key_map_for(kind, elem_bytes);
    uint64_t blocks = (len + 255) / 256;
    if (blocks > (uint64_t)D->cus * 16) blocks = (uint64_t)D->cus * 16;
    // two stable passes on levels L-2, L-1 (keys -> tmp -> keys): the shard ends up ordered by its top 16 bits, in place
    RDST_BY_WIDTH(elem_bytes, rc = (split_top16_t<K, LV>(dev_keys, dev_tmp, len, kind, km, (uint32_t)blocks, dev_counts16, s)));
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return RDST_OK;
}

int rdst_hip_level_counts(const void* dev_keys, uint64_t len, uint32_t elem_bytes, rdst_key_kind kind, uint32_t level,
                          uint64_t counts[256], uint8_t* already_sorted, uint8_t* first_digit, uint8_t* last_digit,
                          void* stream) {
    int rc = check_common(dev_keys, len, elem_bytes, kind, elem_bytes);
    if (rc) return rc;
    if (level >= elem_bytes) return fail(RDST_ERR_ARG, "level out of range");
    if (!counts) return fail(RDST_ERR_ARG, "null counts");
    memset(counts, 0, sizeof(uint64_t) * RADIX);
    if (already_sorted) *already_sorted = 1;
    if (first_digit) *first_digit = 0;
    if (last_digit) *last_digit = 0;
    if (len == 0) return RDST_OK;  // sort_utils.rs:116-118
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceState* D;
    rc = current_device_state(&D);
    if (rc) return rc;
    rc = ensure_workspace(*D, 1 << 20);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(D->ws);
    // scratch inside the workspace head: 256 u64 counts + flag
    unsigned long long* d_counts = reinterpret_cast<unsigned long long*>(ws + 4096);
    uint32_t* d_flag = reinterpret_cast<uint32_t*>(ws + 4096 + sizeof(uint64_t) * RADIX);
    rc = workspace_acquire(*D, s);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(d_counts, 0, sizeof(uint64_t) * RADIX + 16, s));
    const KeyMap km = key_map_for(kind, elem_bytes);
    uint64_t blocks = (len + 255) / 256;
    const uint64_t cap = (uint64_t)D->cus * 8;
    if (blocks > cap) blocks = cap;
    const int shift = (int)level * 8;
    RDST_BY_WIDTH(elem_bytes, (void)LV; hipLaunchKernelGGL((level_counts_kernel<K>), dim3((uint32_t)blocks), dim3(256), 0, s, static_cast<const K*>(dev_keys), len, shift, (K)km.neg, (K)km.pos, d_counts, d_flag));
    HIP_TRY(hipGetLastError());
    uint32_t flag = 0;
    u128 first = 0, last = 0;
    HIP_TRY(hipMemcpyAsync(counts, d_counts, sizeof(uint64_t) * RADIX, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&flag, d_flag, sizeof flag, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&first, dev_keys, elem_bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&last, static_cast<const char*>(dev_keys) + (len - 1) * elem_bytes, elem_bytes, hipMemcpyDeviceToHost, s));
    rc = workspace_release(*D, s);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    auto host_digit = [&](u128 raw) -> uint8_t {
        const bool sign = ((raw >> (elem_bytes * 8 - 1)) & 1) != 0;
        const u128 m = raw ^ (sign ? km.neg : km.pos);
        return (uint8_t)(m >> shift);
    };
    if (already_sorted) *already_sorted = flag ? 0 : 1;
    if (first_digit) *first_digit = host_digit(first);
    if (last_digit) *last_digit = host_digit(last);
    return RDST_OK;
}

}  // extern "C"
