/*
 * rdst_hip.h — C ABI of the MI355X (gfx950) radix-sort hot path that sits behind
 * rdst's RadixSort / RadixKey / Tuner surface.
 *
 * The reference crate (nessex/rdst) has no FFI of its own; every entry point below
 * names the reference interface it stands in for (paths relative to the reference
 * tree, file:line).  A Rust shim binds these with a plain `extern "C"` block
 * (INTEGRATION.md shows it); this repo's own host mirrors (include/rdst.hpp for
 * C++, rdst_amd/ for Python) call exactly the same symbols.
 *
 * Conventions
 *   - every function returns 0 (RDST_OK) or a negative rdst_status; on any non-zero
 *     return the caller's key buffer is unmodified for the host entry point
 *     (rdst_hip_sort), so a shim can fall back to the CPU path and still honour
 *     rdst's infallible `fn sort(self)` contract (src/radix_sort_builder.rs:149-157).
 *   - plain pointers and sizes only; `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream).
 *   - device pointers must be aligned to the element size; 16-byte alignment gets
 *     the wide-load kernels.
 *   - the library owns its device workspace (histograms, look-back status words,
 *     tickets): one per device, grown on demand, process lifetime, mutex-guarded.
 *     Concurrent calls on one device serialise on that mutex.
 */
#ifndef RDST_HIP_H
#define RDST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDST_HIP_ABI_VERSION 2

/* Which built-in RadixKey mapping the element type uses (src/radix_key_impl.rs). */
typedef enum {
    RDST_KEY_UNSIGNED = 0, /* u8..u64: (self >> level*8) as u8          radix_key_impl.rs:3-76   */
    RDST_KEY_SIGNED   = 1, /* i8..i64: ((self ^ MIN) >> level*8) as u8  radix_key_impl.rs:87-160 */
    RDST_KEY_FLOAT    = 2, /* f32/f64: sign-magnitude flip, then ^ MIN  radix_key_impl.rs:162-185 */
    RDST_KEY_BYTES_BE = 3  /* [u8; N]: self[N - 1 - level], i.e. lexicographic  radix_key_impl.rs:78-85;
                              N = elem_bytes = levels in 1..16, host entry point only */
} rdst_key_kind;

typedef enum {
    RDST_OK              = 0,
    RDST_ERR_ARG         = -1, /* bad pointer / size / kind / levels (LEVELS == 0 panics in rdst: radix_sort_builder.rs:22) */
    RDST_ERR_UNSUPPORTED = -2, /* element width / kind not built for the device path (built: 1-, 2-, 4-, 8-, 16-byte integers, f32, f64, [u8; 1..16] through rdst_hip_sort) */
    RDST_ERR_HIP         = -3, /* a HIP runtime call failed; see rdst_hip_last_error() */
    RDST_ERR_NO_DEVICE   = -4, /* no usable gfx950 device */
    RDST_ERR_DEVICE      = -5, /* a kernel reported failure through the workspace error word (bounded spin expired) */
    RDST_ERR_ALIGN       = -6  /* pointer not aligned to the element size */
} rdst_status;

/* src/tuner.rs:2-8.  parent_len: -1 encodes None. */
typedef struct {
    uint64_t threads;
    uint64_t level;
    uint64_t total_levels;
    uint64_t input_len;
    int64_t  parent_len;
} rdst_tuning_params;

/* src/tuner.rs:12-22, in declaration order, plus the two device routes a GPU-aware
 * tuner may return (they need a new arm in the `match` at src/sorter.rs:81-102). */
typedef enum {
    RDST_ALGO_MT_OOP = 0,
    RDST_ALGO_MT_LSB = 1,
    RDST_ALGO_SCANNING = 2,
    RDST_ALGO_RECOMBINATING = 3,
    RDST_ALGO_COMPARATIVE = 4,
    RDST_ALGO_LR_LSB = 5,
    RDST_ALGO_LSB = 6,
    RDST_ALGO_REGIONS = 7,
    RDST_ALGO_SKA = 8,
    RDST_ALGO_GPU_LSD = 9,      /* whole slice on one device: rdst_hip_sort / rdst_hip_sort_device */
    RDST_ALGO_GPU_SHARDED = 10  /* slice spread over ranks: MSD split + exchange + local LSD */
} rdst_algorithm;

/* Stock tuners (src/tuners/, one file each) + the device-aware one. */
typedef enum {
    RDST_TUNER_STANDARD = 0,        /* src/tuners/standard_tuner.rs:10-64 */
    RDST_TUNER_LOW_MEMORY = 1,      /* src/tuners/low_memory_tuner.rs:13-43 */
    RDST_TUNER_SINGLE_THREADED = 2, /* src/tuners/single_threaded_tuner.rs:13-43 */
    RDST_TUNER_GPU = 3              /* StandardTuner, except depth-0 chunks >= gpu_min_len go to GPU_LSD */
} rdst_tuner_id;

/* What ran between two profiling marks (rdst_hip_profile_run_stages); a scatter pass carries its
 * level in bits 8..15. */
typedef enum {
    RDST_STAGE_CLEAR = 1,    /* workspace clear */
    RDST_STAGE_HIST = 2,     /* K1: every level's histogram (returns at once on the hybrid route) */
    RDST_STAGE_SCAN = 3,     /* K2 */
    RDST_STAGE_PASS = 4,     /* K3, one level (skipped levels return at once) */
    RDST_STAGE_COPYBACK = 5,
    RDST_STAGE_HIST16 = 6,   /* K1h: counts of the top 16 bits (hybrid route) */
    RDST_STAGE_ROUTE = 7,    /* route decision + bucket starts */
    RDST_STAGE_LOCAL = 8,    /* K4: per-bucket sort of the remaining levels inside LDS (hybrid and atomic routes) */
    RDST_STAGE_MSD_A = 10,   /* atomic route: scatter by the top byte into over-provisioned areas (claims instead of counts) */
    RDST_STAGE_MSD_B = 11,   /* atomic route: scatter of every area by the second byte into the bucket slots (low halves) */
    RDST_STAGE_SAMPLE = 12   /* the 8 192-key sample (and, if it flags the keys, K1h + the route decision) before the MSD passes */
} rdst_stage;

/* Device routes (rdst_hip_last_route). */
#define RDST_ROUTE_LSD 0u     /* one scatter pass per level */
#define RDST_ROUTE_HYBRID 1u  /* two scatter passes on the top 16 bits + one in-LDS sort per bucket */
#define RDST_ROUTE_ATOMIC 2u  /* the same without the counting read: MSD passes that claim space with atomics */

/* Options of the host entry point.  NULL = defaults. */
typedef struct {
    int32_t  device;       /* HIP device ordinal, -1 = current device */
    int32_t  low_memory;   /* != 0: the low-memory route (rdst_hip_sort_device_lowmem): the device holds the keys and a scratch of
                              len / 64 elements instead of two arrays — what `with_low_mem_tuner()` asks for
                              (src/radix_sort_builder.rs:74-77) */
    uint64_t reserved1;
} rdst_hip_opts;

/* ---- entry points ------------------------------------------------------------------ */

/* Replaces `RadixSort::radix_sort_unstable` for a host slice of a built-in key type
 * (src/radix_sort.rs:21-45 -> src/radix_sort_builder.rs:149-157 -> src/sorter.rs:106-119).
 * Sorts `len` elements of `elem_bytes` bytes in place, ascending in rdst's mapped-key
 * order.  Blocking.  len <= 1 is a no-op (radix_sort_builder.rs:151).  `levels` must
 * equal elem_bytes (RadixKey::LEVELS of every built-in type).  On failure the buffer
 * is untouched.  The library keeps one stream and one device buffer (keys + tmp, up to 1 GiB; larger ones
 * are released on return) per device for this entry point; concurrent calls on a device serialise. */
int rdst_hip_sort(void* host_data, uint64_t len, uint32_t elem_bytes, rdst_key_kind kind,
                  uint32_t levels, const rdst_hip_opts* opts);

/* Where the time of the most recent rdst_hip_sort on the current device went (HIP events on its stream): the copy
 * to the device, the device sort (incl. its status check), the copy back.  10^9 u32 keys: 2 x 4 GB at the link's
 * ~51 GB/s and 6 ms of sorting; a sort needs its whole input before its first output, so the two copies cannot
 * overlap each other. */
int rdst_hip_host_timing(float* h2d_ms, float* sort_ms, float* d2h_ms);

/* Device-resident form of the same call — the timed path.  `dev_keys` holds the keys and
 * receives the sorted result; `dev_tmp` is a caller-provided scratch of the same size
 * (the `tmp_bucket` of src/sorts/lsb_sort.rs:53).  Asynchronous on `stream`.  The LSD
 * pass loop is the device twin of Sorter::lsb_sort_adapter (src/sorts/lsb_sort.rs:39-127)
 * with mt_lsb_sort's (bucket, tile) offset table (src/sorts/mt_lsb_sort.rs:40-133)
 * replaced by an on-device chained scan.  One exception to "asynchronous": 4- and 8-byte keys beyond the byte-saving routes'
 * window (more than ~1.04e9 8-byte keys, 1.3e9 4-byte keys) are split on their top byte and sorted part by part, as
 * Sorter recurses into a bucket too big for the sort at hand (src/sorter.rs:131-138); the 256 counts of that split come to
 * the host, so the call waits once for `stream` and enqueues the rest. */
int rdst_hip_sort_device(void* dev_keys, void* dev_tmp, uint64_t len, uint32_t elem_bytes,
                         rdst_key_kind kind, uint32_t levels, void* stream);

/* Replaces `radix_sort_unstable` on a host slice of structs whose `RadixKey` reads one built-in
 * field (benches/struct_sort.rs:11-27: `LargeStruct { sort_key: f32, .. }` with
 * `get_level` = the field's; examples/impl_radix_key.rs:32-56).  `record_bytes` = size_of::<T>(),
 * the key is `key_bytes` (4 or 8) at `key_offset`, naturally aligned; rows travel to the device,
 * (key, row index) pairs are sorted there, the rows gathered and copied back.  Rows with equal
 * keys keep their input order (rdst promises none).  Blocking; on failure the slice is untouched. */
int rdst_hip_sort_records(void* host_records, uint64_t len, uint32_t record_bytes, uint32_t key_offset,
                          uint32_t key_bytes, rdst_key_kind kind, const rdst_hip_opts* opts);

/* Key-value sort, device-resident: sorts `dev_keys` (4- or 8-byte built-in keys) and carries
 * `dev_vals` (4- or 8-byte payloads, e.g. the index of the record a key came from) along.
 * the device route for slices of structs whose `RadixKey` is a built-in key
 * field (benches/struct_sort.rs:11-27, examples/impl_radix_key.rs:32-56) — extract
 * (key, index), sort the pairs here, gather the records.  The passes are stable, so pairs with
 * equal keys keep their input order; rdst itself promises no order among them
 * (src/radix_sort.rs:21-45 "unstable"), so this is one of the outputs rdst may produce.
 * Both tmp arrays have the size of their originals; asynchronous on `stream` like
 * rdst_hip_sort_device, failures are reported by rdst_hip_device_status. */
int rdst_hip_sort_pairs_device(void* dev_keys, void* dev_vals, void* dev_tmp_keys, void* dev_tmp_vals,
                               uint64_t len, uint32_t key_bytes, rdst_key_kind kind, uint32_t levels,
                               uint32_t val_bytes, void* stream);

/* Blocks until everything queued on `stream` by this library has finished and returns
 * RDST_ERR_DEVICE if any kernel raised the device error word since the last check.  The word is kept
 * outside the per-sort workspace: an error raised by an earlier asynchronous sort is still reported after
 * later sorts were enqueued, once; the check clears it. */
int rdst_hip_device_status(void* stream);

/* Bench yardsticks ("a device copy/triad ceiling on the box", beside the 8 TB/s spec): a
 * 16-bytes-per-lane streaming copy, a read-only sweep and a write-only fill of `bytes` (multiple of 16) on `stream`, in the
 * fastest shape the sweep of profiles/r03_copy_sweep.json found (one contiguous piece per block, non-temporal loads). */
int rdst_hip_stream_copy(void* dev_dst, const void* dev_src, uint64_t bytes, void* stream);
int rdst_hip_stream_read(const void* dev_src, uint64_t bytes, void* stream);
int rdst_hip_stream_fill(void* dev_dst, uint64_t bytes, void* stream);

/* Test hook: ORs `bits` into the device error word from a one-thread kernel on `stream`, exactly as a failing
 * sort kernel would.  The word is sticky: it survives later sorts and workspace growth until
 * rdst_hip_device_status (or a blocking entry point) reports and clears it. */
int rdst_hip_debug_raise_device_error(uint32_t bits, void* stream);

/* Parity hook for get_counts_with_ends (src/sort_utils.rs:109-180) /
 * par_get_counts_with_ends (:35-106): 256-bin histogram of digit `level` over a
 * device-resident slice, plus the `already_sorted` flag (digit sequence non-decreasing)
 * and the first and last digit.  Blocking.  Empty input: counts all zero, sorted = 1,
 * first = last = 0 (sort_utils.rs:116-118). */
int rdst_hip_level_counts(const void* dev_keys, uint64_t len, uint32_t elem_bytes,
                          rdst_key_kind kind, uint32_t level, uint64_t counts[256],
                          uint8_t* already_sorted, uint8_t* first_digit, uint8_t* last_digit,
                          void* stream);

/* Parity hook for the fused multi-level histogram kernel: counts_out[level*256 + digit]
 * for every level in [0, levels).  Same numbers get_counts (sort_utils.rs:183-190)
 * returns level by level.  Blocking. */
int rdst_hip_all_level_counts(const void* dev_keys, uint64_t len, uint32_t elem_bytes,
                              rdst_key_kind kind, uint32_t levels, uint64_t* counts_out,
                              void* stream);

/* Parity hook for one stable counting-sort pass: out_of_place_sort
 * (src/sorts/out_of_place_sort.rs:52-108) / mt_lsb_sort (src/sorts/mt_lsb_sort.rs:40-133)
 * on digit `level`: dev_dst[prefix[d]++] = dev_src[i] in index order.  This is also the
 * MSD split of the sharded route (top level, then contiguous digit ranges per rank).
 * counts_out (nullable) receives the 256 digit counts.  Blocking. */
int rdst_hip_scatter_level(const void* dev_src, void* dev_dst, uint64_t len, uint32_t elem_bytes,
                           rdst_key_kind kind, uint32_t level, uint64_t* counts_out, void* stream);

/* Sharded route (RDST_ALGO_GPU_SHARDED), device-side steps; the collectives between them are the caller's (RCCL through
 * torch.distributed in rdst_amd/sharded.py, or a shim's own ncclAllGather / ncclSend+ncclRecv: INTEGRATION.md).  The
 * reference's analogues are the MSD split of src/sorter.rs:131-138 and the tile -> bucket regrouping of
 * src/sorts/recombinating_sort.rs:68-88.
 *
 * rdst_hip_split_top_level_device: ONE stable pass on the most significant level, dev_src -> dev_dst (dev_src is only
 * read), i.e. the shard grouped by top digit, hence by owner rank (owners are contiguous digit ranges).  The 256 digit
 * counts are left in DEVICE memory (dev_counts, 256 x u64) for the all-gather.  Asynchronous: nothing blocks, the
 * histogram counts that one level only; failures surface in rdst_hip_device_status.
 *
 * rdst_hip_split_top16_device: the fallback when one top digit holds more than a rank's share: two
 * stable passes (levels L-2, L-1; dev_tmp is scratch) leave dev_keys ordered by the top 16 bits of the mapped key, IN
 * PLACE, and dev_counts16 (65 536 x u64, device) receives the bucket lengths — owners are then contiguous ranges of
 * 16-bit prefixes.  Asynchronous.  Keys of at least two bytes. */
int rdst_hip_split_top_level_device(const void* dev_src, void* dev_dst, uint64_t len, uint32_t elem_bytes,
                                    rdst_key_kind kind, uint64_t* dev_counts, void* stream);
int rdst_hip_split_top16_device(void* dev_keys, void* dev_tmp, uint64_t len, uint32_t elem_bytes,
                                rdst_key_kind kind, uint64_t* dev_counts16, void* stream);

/* ---- low-memory route ------------------------------------------------------------------------
 * Device twin of the route `with_low_mem_tuner()` selects (src/radix_sort_builder.rs:74-77; LowMemoryTuner picks Regions
 * above 10^6 elements, Ska below: src/tuners/low_memory_tuner.rs:36-41): sorts `dev_keys` IN PLACE with a scratch of
 * only `tmp_len` elements (`dev_tmp`; len / 64 is a good size, anything from 2^16 elements up works) instead of a second
 * array.  One Regions-sort level (src/sorts/regions_sort.rs:51-286) on the most significant unsorted digit: every tile of
 * tmp_len keys is grouped by digit through the scratch (the per-tile ska_sort of :216-223), the tile x digit counts
 * come to the host, which plans equal-length block swaps in rounds (the reference plans them serially too, :235-239) and
 * a swap kernel runs them; then groups of buckets that fit the scratch are sorted by the ordinary device route, and a
 * bucket that does not fit takes another Regions level on the next digit.  BLOCKING (the plan is made on the host).
 * Same result as rdst_hip_sort_device, bit for bit. */
int rdst_hip_sort_device_lowmem(void* dev_keys, uint64_t len, uint32_t elem_bytes, rdst_key_kind kind, uint32_t levels,
                                void* dev_tmp, uint64_t tmp_len, void* stream);

/* `partition_index` (src/sort_utils.rs:295-331: two-ended in-place partition by a predicate; Ska, Scanning and Regions
 * use it to move the largest bucket aside first) as a device operation: keys whose digit `level` equals `digit` first,
 * the others after, in place, with the same scratch and swap machinery (a two-country Regions level).  Returns the split
 * index in *split_out.  The order inside the two parts is unspecified, as in the reference.  BLOCKING. */
int rdst_hip_partition_device(void* dev_keys, uint64_t len, uint32_t elem_bytes, rdst_key_kind kind, uint32_t level,
                              uint32_t digit, void* dev_tmp, uint64_t tmp_len, uint64_t* split_out, void* stream);

/* The swap plan of one Regions level, host only (what rdst_hip_sort_device_lowmem runs; exposed so that it can be
 * checked without a device).  Every tile consists of `columns` runs, in order; tile_counts[t * columns + c] = length of
 * run c of tile t, col_bucket[c] = the bucket its keys belong to (NULL: column c is bucket c, columns == buckets);
 * tiles = ceil(len / tile_len).  ops_out[round_starts[r] .. round_starts[r + 1]) are the swaps of round r: exchange
 * `len` elements at `a` with those at `b`; the ranges of one round are pairwise disjoint.  starts_out (nullable,
 * buckets + 1 entries) receives the region borders. */
typedef struct { uint64_t a, b, len; } rdst_swap_op;
int rdst_regions_plan(const uint64_t* tile_counts, uint64_t tiles, uint64_t tile_len, uint64_t len, uint32_t columns,
                      const uint32_t* col_bucket, uint32_t buckets, rdst_swap_op* ops_out, uint64_t ops_capacity,
                      uint64_t* nops_out, uint64_t* round_starts, uint32_t max_rounds, uint32_t* nrounds_out,
                      uint64_t* starts_out);

/* Replaces `Tuner::pick_algorithm` (src/tuner.rs:33-35) for the stock tuners: pure
 * integer decision tables, host only.  counts has 256 entries (src/sorter.rs:67-76).
 * gpu_min_len is read only by RDST_TUNER_GPU.  For host slices the device route starts to win at
 * about 7·10^4 elements (pool allocation + PCIe both ways): RDST_GPU_MIN_LEN_HOST_SLICE. */
#define RDST_GPU_MIN_LEN_HOST_SLICE 65536u  /* measured break-even of rdst_hip_sort against the CPU route, see INTEGRATION.md */
int rdst_pick_algorithm(int tuner_id, const rdst_tuning_params* p, const uint64_t counts[256],
                        uint64_t gpu_min_len);

/* Bytes of device memory the workspace needs for a sort of `len` elements. */
uint64_t rdst_hip_workspace_bytes(uint64_t len, uint32_t elem_bytes);

/* The workspace is library-owned, one per device, grown on demand and kept between calls — the device twin of the
 * `tmp_bucket` the reference allocates and drops inside every sort (src/sorts/lsb_sort.rs:53, :126): for 10^9-key slices
 * it is 1.7 x the slice (the byte-saving routes' areas and slots).  This call gives it back (blocking: it waits for every
 * queued sort of the current device); the next sort allocates again.  If the allocation of a large workspace fails, the sort
 * takes the LSD route, whose workspace is an eighth of the slice, instead of failing. */
int rdst_hip_release_workspace(void);

/* Runtime knobs for experiments: pass_config selects the scatter-kernel shape (tile size and
 * LDS staging; negative = built-in default), hist_blocks_per_cu the histogram grid (<= 0 =
 * built-in).  Not part of the reference surface. */
int rdst_hip_set_tuning(int pass_config, int hist_blocks_per_cu);

/* Experiment knob: enabled == 0 makes every scatter pass use ONE look-back chain over all of its
 * tiles instead of splitting its source into segments with a chain each (default: split).
 * Results are identical either way.  Not part of the reference surface. */
int rdst_hip_set_chain_split(int enabled);

/* Experiment knob: enabled == 0 ranks every round of a scatter pass with wave ballots; the default
 * takes the slots a returning LDS add hands out and falls back to the ballots for any round whose
 * result fails the in-kernel order test (rdst_kernels.hip, step 5).  enabled == 2: self-test mode, every
 * round is treated as failed and redone (exercises the fallback).  Results are identical in all modes. */
int rdst_hip_set_fast_rank(int enabled);

/* Experiment knob: enabled == 0 sends slices of at most 64 KiB of keys through the general pipeline too,
 * instead of the one-workgroup LDS sort (the device twin of src/sorts/lsb_sort.rs:39-127).  Same results. */
int rdst_hip_set_small_sort(int enabled);

/* Per-kernel device timing for benchmarks.  While enabled, every pipeline (sort / hook call)
 * records HIP events on its own stream between its launches and appends one "run" to a
 * per-device list; enabling again clears the list.  rdst_hip_profile_run blocks until run
 * `run` (negative = counted from the most recent) has finished and writes the elapsed
 * milliseconds of its stages in launch order: [0] workspace clear, [1] multi-level histogram
 * (K1), [2] scan (K2), then one entry per scatter pass (K3) for the levels the call covered,
 * then the conditional copy-back if the call has one.  *n_out = entries written. */
int rdst_hip_set_profiling(int enabled);
int rdst_hip_profile_runs(void);
int rdst_hip_profile_run(int run, float* out_ms, uint32_t capacity, uint32_t* n_out);
/* The same run's stage codes, entry for entry (rdst_stage, a pass's level in bits 8..15): the hybrid route
 * inserts RDST_STAGE_HIST16 and RDST_STAGE_ROUTE after the clear and RDST_STAGE_LOCAL after the passes. */
int rdst_hip_profile_run_stages(int run, uint32_t* stages_out, uint32_t capacity, uint32_t* n_out);

/* Route choice.  Whole sorts of 4- and 8-byte keys with min_len <= len < 2^30 try routes that move fewer bytes than one
 * scatter pass per level — the device form of rdst's own MSD-then-Lsb route at this size (SURVEY.md §3.1;
 * src/tuners/standard_tuner.rs:46-62 picks by length and counts, too).  The device decides, in this order:
 *   ATOMIC — two MSD passes that claim space in over-provisioned areas with atomics (no counting read at all), then the
 *   in-LDS sort of every bucket; given up if an area or a slot overflows (keys far from uniform);
 *   HYBRID — K1h counts the top 16 bits exactly, two K3 passes, the in-LDS sort; 8-byte keys: if every bucket fits one tile,
 *   4-byte keys: buckets of any size (up to 4 096 of 65 536 keys and more);
 *   LSD — everything else.
 * enabled == 0: LSD route always; 1: the default above; 7: start at the hybrid route; 2 / 3 / 5 / 6 / 9: hybrid route for
 * every width with one detail changed, for A/B runs and tests — 2: ranked in-LDS sort for 4-byte keys as well, 3: pass L-1
 * hands K4 whole keys instead of 16-bit halves, 5: no key sample, 6: 8-byte keys with the one-block-per-CU form of K4, 9: no
 * expanding K4 (4-byte buckets up to one tile only); 8: the atomic route for 4-byte keys only; 10: a failed atomic route
 * falls straight to LSD; 11: no giant kernels; 12: no exact form of the MSD passes; 14: no prediction of the LSD route
 * from the sample; 15: the second form of the 8-byte K4; 16: no split of slices beyond the window; 17: that split at every
 * length, in eight parts.  min_len == 0 keeps the built-in thresholds (atomic route: 3 * 2^26 4-byte keys, 2^26 8-byte keys; K1h hybrid route: 2^28).  Results are identical
 * on every route.  Not part of the reference surface. */
int rdst_hip_set_hybrid(int enabled, uint64_t min_len);

/* Route the most recent sort enqueued by this library on the current device took (RDST_ROUTE_*).  Blocks on `stream`. */
int rdst_hip_last_route(void* stream, uint32_t* route_out);

/* Last error message of the calling thread ("" if none). */
const char* rdst_hip_last_error(void);

/* ABI version of the loaded library (RDST_HIP_ABI_VERSION at build time). */
int rdst_hip_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RDST_HIP_H */
