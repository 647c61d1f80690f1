/*
 * rdst_oracle.c — CPU restatement of the reference's radix-sort path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the device route (rdst_amd/) never does and has no
 * CPU fallback.
 *
 * What it restates (nessex/rdst, crate version 0.20.14 + the unreleased 0.21.0 tree; paths
 * below are relative to the reference tree): RadixKey for every built-in type
 * (src/radix_key_impl.rs), the histogram / prefix-sum / partition primitives
 * (src/sort_utils.rs), the four out-of-place scatters (src/sorts/out_of_place_sort.rs), the
 * nine algorithms and their adapters (src/sorts/ *.rs), the dispatcher (src/sorter.rs), the
 * three stock tuners (src/tuners/ *.rs) and the builder entry (src/radix_sort_builder.rs).
 * rayon's work-stealing pool becomes OpenMP tasks; std::sync::Mutex::try_lock becomes
 * omp_test_lock.  Each function cites the lines it follows.
 *
 * Parity pin: the reference is Rust and cannot be built or run in this environment (no
 * cargo/rustc, no network), so this restatement is pinned by the reference's own known
 * answers instead — see tests/golden/ and tests/test_oracle_golden.py — and, for every
 * built-in key type, by uniqueness of the sorted array (the RadixKey impl consumes every bit
 * of the value through a bijection, so any correct sort of the multiset is THE output of
 * radix_sort_unstable()), which tests/ check against an independent numpy sort.
 */
#define _GNU_SOURCE
#include <assert.h>
#include <omp.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rdst_oracle.h"

/* Algorithm — src/tuner.rs:12-22, declaration order */
enum {
    ALGO_MT_OOP = 0,
    ALGO_MT_LSB = 1,
    ALGO_SCANNING = 2,
    ALGO_RECOMBINATING = 3,
    ALGO_COMPARATIVE = 4,
    ALGO_LR_LSB = 5,
    ALGO_LSB = 6,
    ALGO_REGIONS = 7,
    ALGO_SKA = 8
};

struct sorter { /* src/sorter.rs:10-22 */
    bool multi_threaded;
    size_t threads;     /* rayon::current_num_threads() as seen by the algorithms */
    size_t omp_threads; /* size of the OpenMP team actually started */
    rdst_o_pick_fn pick; /* TunerRef */
    void* pick_ctx;
    rdst_o_trace_fn trace; /* `work_profiles` feature: sorter.rs:78-79 */
};

static void* xmalloc(size_t n) {
    void* p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "rdst_oracle: out of memory (%zu bytes)\n", n); abort(); }
    return p;
}
static void* xrealloc(void* q, size_t n) {
    void* p = realloc(q, n ? n : 1);
    if (!p) { fprintf(stderr, "rdst_oracle: out of memory (%zu bytes)\n", n); abort(); }
    return p;
}
static inline size_t div_ceil(size_t a, size_t b) { return (a + b - 1) / b; }

/* get_prefix_sums — src/sort_utils.rs:10-20 */
static void get_prefix_sums(const size_t counts[256], size_t sums[256]) {
    size_t running_total = 0;
    for (int i = 0; i < 256; ++i) {
        sums[i] = running_total;
        running_total += counts[i];
    }
}

/* get_end_offsets — src/sort_utils.rs:23-31 */
static void get_end_offsets(const size_t counts[256], const size_t prefix_sums[256], size_t end_offsets[256]) {
    for (int i = 0; i < 256; ++i) end_offsets[i] = prefix_sums[i == 255 ? 255 : i + 1]; /* saturating_add(1) */
    end_offsets[255] += counts[255];
}

/* aggregate_tile_counts — src/sort_utils.rs:247-249 */
static void aggregate_tile_counts(const size_t* tile_counts, size_t tiles, size_t out[256]) {
    for (int i = 0; i < 256; ++i) {
        size_t s = 0;
        for (size_t t = 0; t < tiles; ++t) s += tile_counts[t * 256 + i];
        out[i] = s;
    }
}

/* ---- tuners ------------------------------------------------------------------------------- */

/* the skew test shared by the three tuners: any count >= (len/256)*2 once len >= 5000 */
static bool any_count_over_threshold(size_t input_len, const size_t counts[256]) {
    if (input_len < 5000) return false;
    const size_t distribution_threshold = (input_len / 256) * 2;
    for (int i = 0; i < 256; ++i)
        if (counts[i] >= distribution_threshold) return true;
    return false;
}

/* StandardTuner::pick_algorithm — src/tuners/standard_tuner.rs:13-63 */
static int standard_tuner(void* ctx, const struct rdst_o_tuning_params* p, const size_t counts[256]) {
    (void)ctx;
    const size_t n = p->input_len;
    if (n <= 128) return ALGO_COMPARATIVE;
    const size_t depth = p->total_levels - p->level - 1;
    if (any_count_over_threshold(n, counts)) {
        if (depth == 0) { /* :26-32 */
            if (n <= 200000) return ALGO_LR_LSB;
            if (n <= 350000) return ALGO_SKA;
            if (n <= 4000000) return ALGO_MT_LSB;
            return ALGO_REGIONS;
        }
        if (n <= 200000) return ALGO_LR_LSB; /* :34-40 */
        if (n <= 800000) return ALGO_SKA;
        if (n <= 5000000) return ALGO_RECOMBINATING;
        return ALGO_REGIONS;
    }
    if (depth > 0) { /* :46-53 */
        if (n <= 200000) return ALGO_LSB;
        if (n <= 800000) return ALGO_SKA;
        if (n <= 50000000) return ALGO_RECOMBINATING;
        return ALGO_SCANNING;
    }
    if (n <= 150000) return ALGO_LSB; /* :55-61 */
    if (n <= 260000) return ALGO_SKA;
    if (n <= 50000000) return ALGO_RECOMBINATING;
    return ALGO_SCANNING;
}

/* LowMemoryTuner::pick_algorithm — src/tuners/low_memory_tuner.rs:16-42 */
static int low_memory_tuner(void* ctx, const struct rdst_o_tuning_params* p, const size_t counts[256]) {
    (void)ctx;
    const size_t n = p->input_len;
    if (n <= 128) return ALGO_COMPARATIVE;
    if (any_count_over_threshold(n, counts)) {
        if (n <= 50000) return ALGO_LR_LSB;
        if (n <= 1000000) return ALGO_SKA;
        return ALGO_REGIONS;
    }
    if (n <= 50000) return ALGO_LSB;
    if (n <= 1000000) return ALGO_SKA;
    return ALGO_REGIONS;
}

/* SingleThreadedTuner::pick_algorithm — src/tuners/single_threaded_tuner.rs:16-42 */
static int single_threaded_tuner(void* ctx, const struct rdst_o_tuning_params* p, const size_t counts[256]) {
    (void)ctx;
    const size_t n = p->input_len;
    if (n <= 128) return ALGO_COMPARATIVE;
    const size_t depth = p->total_levels - p->level - 1;
    if (any_count_over_threshold(n, counts)) return (n > 100000 && depth < 2) ? ALGO_SKA : ALGO_LR_LSB;
    return (n > 800000 && depth == 0) ? ALGO_SKA : ALGO_LSB;
}

/* SingleAlgoTuner — src/test_utils.rs:40-49 */
static int single_algo_tuner(void* ctx, const struct rdst_o_tuning_params* p, const size_t counts[256]) {
    (void)p;
    (void)counts;
    return *(const int*)ctx;
}

/* ---- element types: RadixKey impls — src/radix_key_impl.rs -------------------------------- */
typedef unsigned __int128 u128;
typedef struct { uint8_t b[3]; } bytes3;
typedef struct { uint8_t b[4]; } bytes4;

/* unsigned: (self >> (level * 8)) as u8 — :3-76 */
#define T uint8_t
#define SUF u8
#define LEVELS 1
#define GET_LEVEL(v, level) (v)
#include "rdst_oracle_impl.h"

#define T uint16_t
#define SUF u16
#define LEVELS 2
#define GET_LEVEL(v, level) ((v) >> ((level) * 8))
#include "rdst_oracle_impl.h"

#define T uint32_t
#define SUF u32
#define LEVELS 4
#define GET_LEVEL(v, level) ((v) >> ((level) * 8))
#include "rdst_oracle_impl.h"

#define T uint64_t
#define SUF u64
#define LEVELS 8
#define GET_LEVEL(v, level) ((v) >> ((level) * 8))
#include "rdst_oracle_impl.h"

#define T u128
#define SUF u128
#define LEVELS 16
#define GET_LEVEL(v, level) ((v) >> ((level) * 8))
#include "rdst_oracle_impl.h"

/* [u8; N]: self[N - 1 - level] — :78-85 (N = 3, the width the reference tests: radix_sort.rs:221-229) */
#define T bytes3
#define SUF b3
#define LEVELS 3
#define GET_LEVEL(v, level) ((v).b[3 - (level)-1])
#include "rdst_oracle_impl.h"

/* The three user-defined keys of examples/impl_radix_key.rs:5-56 (known-answer vectors only):
 * PackedU8 = lexicographic [u8; 4]; EvenSortedPackedU8 / OddSortedPackedU8 = 2-level partial keys. */
#define T bytes4
#define SUF b4
#define LEVELS 4
#define GET_LEVEL(v, level) ((v).b[3 - (level)])
#include "rdst_oracle_impl.h"

#define T bytes4
#define SUF pk_even
#define LEVELS 2
#define GET_LEVEL(v, level) ((v).b[3 - (level)*2])
#include "rdst_oracle_impl.h"

#define T bytes4
#define SUF pk_odd
#define LEVELS 2
#define GET_LEVEL(v, level) ((v).b[3 - ((level)*2 + 1)])
#include "rdst_oracle_impl.h"

/* signed: ((self ^ MIN) >> (level * 8)) as u8 — :87-160.  Elements are held as raw bit
 * patterns; xor with the sign bit then a logical shift yields the same byte as the
 * reference's arithmetic shift, because only the low 8 bits of the shifted value are kept
 * and level*8 + 8 <= width. */
#define T uint8_t
#define SUF i8
#define LEVELS 1
#define GET_LEVEL(v, level) ((uint8_t)((v) ^ 0x80u))
#include "rdst_oracle_impl.h"

#define T uint16_t
#define SUF i16
#define LEVELS 2
#define GET_LEVEL(v, level) ((uint16_t)((v) ^ 0x8000u) >> ((level) * 8))
#include "rdst_oracle_impl.h"

#define T uint32_t
#define SUF i32
#define LEVELS 4
#define GET_LEVEL(v, level) (((v) ^ 0x80000000u) >> ((level) * 8))
#include "rdst_oracle_impl.h"

#define T uint64_t
#define SUF i64
#define LEVELS 8
#define GET_LEVEL(v, level) (((v) ^ 0x8000000000000000ull) >> ((level) * 8))
#include "rdst_oracle_impl.h"

#define T u128
#define SUF i128
#define LEVELS 16
#define GET_LEVEL(v, level) (((v) ^ ((u128)1 << 127)) >> ((level) * 8))
#include "rdst_oracle_impl.h"

/* f32 — :162-173: s = bits as i32; s ^= (((s >> 31) as u32) >> 1) as i32; ((s ^ MIN) >> level*8) as u8 */
static inline uint32_t f32_key(uint32_t bits) {
    int32_t s = (int32_t)bits;
    s ^= (int32_t)(((uint32_t)(s >> 31)) >> 1);
    return (uint32_t)s ^ 0x80000000u;
}
#define T uint32_t
#define SUF f32
#define LEVELS 4
#define GET_LEVEL(v, level) (f32_key(v) >> ((level) * 8))
#include "rdst_oracle_impl.h"

/* f64 — :175-185 */
static inline uint64_t f64_key(uint64_t bits) {
    int64_t s = (int64_t)bits;
    s ^= (int64_t)(((uint64_t)(s >> 63)) >> 1);
    return (uint64_t)s ^ 0x8000000000000000ull;
}
#define T uint64_t
#define SUF f64
#define LEVELS 8
#define GET_LEVEL(v, level) (f64_key(v) >> ((level) * 8))
#include "rdst_oracle_impl.h"

/* ---- exported surface ---------------------------------------------------------------------- */

static const size_t k_elem_bytes[RDST_O_NUM_TYPES] = {1, 2, 4, 8, 16, 1, 2, 4, 8, 16, 4, 8, 3, 4, 4, 4};
static const size_t k_levels[RDST_O_NUM_TYPES] = {1, 2, 4, 8, 16, 1, 2, 4, 8, 16, 4, 8, 3, 4, 2, 2};

size_t rdst_oracle_elem_bytes(int type_id) { return (type_id < 0 || type_id >= RDST_O_NUM_TYPES) ? 0 : k_elem_bytes[type_id]; }
size_t rdst_oracle_levels(int type_id) { return (type_id < 0 || type_id >= RDST_O_NUM_TYPES) ? 0 : k_levels[type_id]; }

#define DISPATCH(type_id, CALL)                                         \
    switch (type_id) {                                                  \
        case RDST_O_U8: { typedef uint8_t E; CALL(u8); } break;         \
        case RDST_O_U16: { typedef uint16_t E; CALL(u16); } break;      \
        case RDST_O_U32: { typedef uint32_t E; CALL(u32); } break;      \
        case RDST_O_U64: { typedef uint64_t E; CALL(u64); } break;      \
        case RDST_O_U128: { typedef u128 E; CALL(u128); } break;        \
        case RDST_O_I8: { typedef uint8_t E; CALL(i8); } break;         \
        case RDST_O_I16: { typedef uint16_t E; CALL(i16); } break;      \
        case RDST_O_I32: { typedef uint32_t E; CALL(i32); } break;      \
        case RDST_O_I64: { typedef uint64_t E; CALL(i64); } break;      \
        case RDST_O_I128: { typedef u128 E; CALL(i128); } break;        \
        case RDST_O_F32: { typedef uint32_t E; CALL(f32); } break;      \
        case RDST_O_F64: { typedef uint64_t E; CALL(f64); } break;      \
        case RDST_O_B3: { typedef bytes3 E; CALL(b3); } break;          \
        case RDST_O_B4: { typedef bytes4 E; CALL(b4); } break;          \
        case RDST_O_PK_EVEN: { typedef bytes4 E; CALL(pk_even); } break; \
        case RDST_O_PK_ODD: { typedef bytes4 E; CALL(pk_odd); } break;  \
        default: return -1;                                             \
    }

int rdst_oracle_get_level(const void* elem, int type_id, size_t level) {
    if (level >= rdst_oracle_levels(type_id)) return -1;
#define CALL(S) return lvl_##S((const E*)elem, level)
    DISPATCH(type_id, CALL)
#undef CALL
    return -1;
}

int rdst_oracle_pick_algorithm(int tuner_id, const struct rdst_o_tuning_params* p, const size_t counts[256]) {
    switch (tuner_id) {
        case RDST_O_TUNER_STANDARD: return standard_tuner(NULL, p, counts);
        case RDST_O_TUNER_LOW_MEMORY: return low_memory_tuner(NULL, p, counts);
        case RDST_O_TUNER_SINGLE_THREADED: return single_threaded_tuner(NULL, p, counts);
        default: return -1;
    }
}

static int make_sorter(struct sorter* s, int tuner_id, int multi_threaded, int threads, rdst_o_pick_fn custom,
                       void* custom_ctx, rdst_o_trace_fn trace) {
    if (threads <= 0) threads = omp_get_max_threads();
    s->multi_threaded = multi_threaded != 0;
    s->threads = (size_t)threads;
    s->omp_threads = (size_t)threads;
    s->pick_ctx = custom_ctx;
    s->trace = trace;
    if (custom) { s->pick = custom; return 0; }
    switch (tuner_id) {
        case RDST_O_TUNER_STANDARD: s->pick = standard_tuner; break;
        case RDST_O_TUNER_LOW_MEMORY: s->pick = low_memory_tuner; break;
        case RDST_O_TUNER_SINGLE_THREADED: s->pick = single_threaded_tuner; break;
        default: return -1;
    }
    return 0;
}

/* radix_sort_builder().with_*().sort() — src/radix_sort_builder.rs:19-157 */
int rdst_oracle_sort(void* data, size_t len, int type_id, int tuner_id, int multi_threaded, int threads) {
    struct sorter s;
    if (make_sorter(&s, tuner_id, multi_threaded, threads, NULL, NULL, NULL)) return -1;
#define CALL(S) sort_top_##S(&s, (E*)data, len)
    DISPATCH(type_id, CALL)
#undef CALL
    return 0;
}

/* with_tuner(&custom) — src/radix_sort_builder.rs:128-132 */
int rdst_oracle_sort_with_tuner(void* data, size_t len, int type_id, rdst_o_pick_fn pick, void* ctx, int multi_threaded,
                                int threads, rdst_o_trace_fn trace) {
    struct sorter s;
    if (!pick) return -1;
    if (make_sorter(&s, 0, multi_threaded, threads, pick, ctx, trace)) return -1;
#define CALL(S) sort_top_##S(&s, (E*)data, len)
    DISPATCH(type_id, CALL)
#undef CALL
    return 0;
}

/* sort_single_algorithm — src/test_utils.rs:264-278 */
int rdst_oracle_sort_single_algorithm(void* data, size_t len, int type_id, int algorithm, int threads) {
    if (algorithm < 0 || algorithm > ALGO_SKA) return -1;
    int algo = algorithm;
    return rdst_oracle_sort_with_tuner(data, len, type_id, single_algo_tuner, &algo, 1, threads, NULL);
}

int rdst_oracle_get_counts_with_ends(const void* data, size_t len, int type_id, size_t level, size_t counts[256],
                                     uint8_t* already_sorted, uint8_t* first, uint8_t* last) {
    if (level >= rdst_oracle_levels(type_id)) return -1;
    bool s = true;
#define CALL(S) get_counts_with_ends_##S((const E*)data, len, level, counts, &s, first, last)
    DISPATCH(type_id, CALL)
#undef CALL
    *already_sorted = s;
    return 0;
}

int rdst_oracle_par_get_counts_with_ends(const void* data, size_t len, int type_id, size_t level, int threads,
                                         size_t counts[256], uint8_t* already_sorted, uint8_t* first, uint8_t* last) {
    if (level >= rdst_oracle_levels(type_id) || threads <= 0) return -1;
    bool s = true;
#define CALL(S)                                                                                             \
    _Pragma("omp parallel num_threads(threads)") _Pragma("omp single")                                      \
        par_get_counts_with_ends_##S((const E*)data, len, level, (size_t)threads, counts, &s, first, last)
    DISPATCH(type_id, CALL)
#undef CALL
    *already_sorted = s;
    return 0;
}

/* tile_counts_out: [max_tiles][256]; returns the number of tiles or -1 */
long rdst_oracle_get_tile_counts(const void* data, size_t len, int type_id, size_t tile_size, size_t level, int threads,
                                 size_t* tile_counts_out, size_t max_tiles, uint8_t* already_sorted) {
    if (level >= rdst_oracle_levels(type_id) || tile_size == 0 || threads <= 0) return -1;
    size_t tiles = 0;
    bool s = true;
    size_t* tc = NULL;
#define CALL(S)                                                                                   \
    _Pragma("omp parallel num_threads(threads)") _Pragma("omp single")                            \
        tc = get_tile_counts_##S((const E*)data, len, tile_size, level, (size_t)threads, &tiles, &s)
    DISPATCH(type_id, CALL)
#undef CALL
    if (tiles > max_tiles) { free(tc); return -1; }
    memcpy(tile_counts_out, tc, tiles * 256 * sizeof(size_t));
    free(tc);
    *already_sorted = s;
    return (long)tiles;
}

/* variant: 0 out_of_place_sort, 1 _with_counts, 2 lr_, 3 lr_with_counts (route_out_of_place_sort,
 * out_of_place_sort.rs:392-424).  counts = histogram of `level` over src; next_counts filled for 1/3. */
int rdst_oracle_out_of_place_sort(const void* src, void* dst, size_t len, int type_id, size_t level, int variant,
                                  const size_t counts[256], size_t next_counts[256]) {
    const size_t levels = rdst_oracle_levels(type_id);
    if (level >= levels || variant < 0 || variant > 3) return -1;
    const bool should_count = (variant & 1) != 0, lr = (variant & 2) != 0;
    if (should_count && level + 1 >= levels) return -1;
#define CALL(S) (void)route_out_of_place_sort_##S(should_count, lr, (const E*)src, (E*)dst, len, counts, level, next_counts)
    DISPATCH(type_id, CALL)
#undef CALL
    return 0;
}

/* Sorter::lsb_sort_adapter as the reference's unit tests call it (lsb_sort.rs:153-165):
 * end_counts = histogram of end_level over the input */
int rdst_oracle_lsb_sort_adapter(void* data, size_t len, int type_id, int lr, size_t start_level, size_t end_level) {
    const size_t levels = rdst_oracle_levels(type_id);
    if (end_level >= levels || start_level > end_level) return -1;
    size_t counts[256];
    bool s;
#define CALL(S)                                                             \
    get_counts_##S((const E*)data, len, end_level, counts, &s);             \
    lsb_sort_adapter_##S(lr != 0, (E*)data, len, counts, start_level, end_level)
    DISPATCH(type_id, CALL)
#undef CALL
    return 0;
}

/* mt_lsb_sort: one tile-parallel stable pass (mt_lsb_sort.rs:40-133) with the reference's own tile counts */
int rdst_oracle_mt_lsb_sort(const void* src, void* dst, size_t len, int type_id, size_t tile_size, size_t level,
                            int threads) {
    if (level >= rdst_oracle_levels(type_id) || tile_size == 0 || threads <= 0) return -1;
#define CALL(S)                                                                                                  \
    _Pragma("omp parallel num_threads(threads)") _Pragma("omp single") {                                         \
        size_t tiles;                                                                                            \
        bool srt;                                                                                                \
        size_t* tc = get_tile_counts_##S((const E*)src, len, tile_size, level, (size_t)threads, &tiles, &srt);   \
        mt_lsb_sort_##S((const E*)src, (E*)dst, len, tc, tiles, tile_size, level);                               \
        free(tc);                                                                                                \
    }
    DISPATCH(type_id, CALL)
#undef CALL
    return 0;
}

int rdst_oracle_num_procs(void) { return omp_get_num_procs(); }
