/*
 * rdst_oracle.h — exported surface of the CPU oracle (oracle/rdst_oracle.c).
 * TEST INFRASTRUCTURE: loaded only by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.
 */
#ifndef RDST_ORACLE_H
#define RDST_ORACLE_H
#include <stddef.h>
#include <stdint.h>

/* element types = the built-in RadixKey impls of src/radix_key_impl.rs (+ [u8; 3]) */
enum {
    RDST_O_U8 = 0, RDST_O_U16, RDST_O_U32, RDST_O_U64, RDST_O_U128,
    RDST_O_I8, RDST_O_I16, RDST_O_I32, RDST_O_I64, RDST_O_I128,
    RDST_O_F32, RDST_O_F64, RDST_O_B3,
    RDST_O_B4, RDST_O_PK_EVEN, RDST_O_PK_ODD, /* examples/impl_radix_key.rs */
    RDST_O_NUM_TYPES
};
enum { RDST_O_TUNER_STANDARD = 0, RDST_O_TUNER_LOW_MEMORY = 1, RDST_O_TUNER_SINGLE_THREADED = 2 };

/* TuningParams — src/tuner.rs:2-8; parent_len -1 = None */
struct rdst_o_tuning_params {
    size_t threads, level, total_levels, input_len;
    int64_t parent_len;
};
/* Tuner::pick_algorithm — src/tuner.rs:33-35; returns an Algorithm ordinal (tuner.rs:12-22) */
typedef int (*rdst_o_pick_fn)(void* ctx, const struct rdst_o_tuning_params* p, const size_t counts[256]);
/* the `work_profiles` println of src/sorter.rs:78-79 as a callback */
typedef void (*rdst_o_trace_fn)(void* ctx, size_t level, size_t len, int algorithm);

size_t rdst_oracle_elem_bytes(int type_id);
size_t rdst_oracle_levels(int type_id);
int rdst_oracle_get_level(const void* elem, int type_id, size_t level);
int rdst_oracle_pick_algorithm(int tuner_id, const struct rdst_o_tuning_params* p, const size_t counts[256]);
int rdst_oracle_sort(void* data, size_t len, int type_id, int tuner_id, int multi_threaded, int threads);
int rdst_oracle_sort_with_tuner(void* data, size_t len, int type_id, rdst_o_pick_fn pick, void* ctx, int multi_threaded,
                                int threads, rdst_o_trace_fn trace);
int rdst_oracle_sort_single_algorithm(void* data, size_t len, int type_id, int algorithm, int threads);
int rdst_oracle_get_counts_with_ends(const void* data, size_t len, int type_id, size_t level, size_t counts[256],
                                     uint8_t* already_sorted, uint8_t* first, uint8_t* last);
int rdst_oracle_par_get_counts_with_ends(const void* data, size_t len, int type_id, size_t level, int threads,
                                         size_t counts[256], uint8_t* already_sorted, uint8_t* first, uint8_t* last);
long rdst_oracle_get_tile_counts(const void* data, size_t len, int type_id, size_t tile_size, size_t level, int threads,
                                 size_t* tile_counts_out, size_t max_tiles, uint8_t* already_sorted);
int rdst_oracle_out_of_place_sort(const void* src, void* dst, size_t len, int type_id, size_t level, int variant,
                                  const size_t counts[256], size_t next_counts[256]);
int rdst_oracle_lsb_sort_adapter(void* data, size_t len, int type_id, int lr, size_t start_level, size_t end_level);
int rdst_oracle_mt_lsb_sort(const void* src, void* dst, size_t len, int type_id, size_t tile_size, size_t level,
                            int threads);
int rdst_oracle_num_procs(void);
#endif
