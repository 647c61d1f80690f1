// rdst_tuner.cpp — host-only decision tables behind rdst_pick_algorithm (include/rdst_hip.h).
// Restates Tuner::pick_algorithm of the three stock tuners as integer range tables and adds
// the device-aware variant.  No device code here.
//
//   StandardTuner        src/tuners/standard_tuner.rs:10-64
//   LowMemoryTuner       src/tuners/low_memory_tuner.rs:13-43
//   SingleThreadedTuner  src/tuners/single_threaded_tuner.rs:13-43
#include <stdint.h>
#include "rdst_hip.h"

namespace {

// `any count >= (len / 256) * 2` for len >= 5000 — the skew test all three tuners share
// (standard_tuner.rs:20-24, low_memory_tuner.rs:20-24, single_threaded_tuner.rs:22-26)
bool skewed(uint64_t len, const uint64_t counts[256]) {
    if (len < 5000) return false;
    const uint64_t threshold = (len / 256) * 2;
    for (int i = 0; i < 256; ++i)
        if (counts[i] >= threshold) return true;
    return false;
}

int standard(const rdst_tuning_params* p, const uint64_t counts[256]) {
    const uint64_t n = p->input_len;
    if (n <= 128) return RDST_ALGO_COMPARATIVE;
    const uint64_t depth = p->total_levels - p->level - 1;
    if (skewed(n, counts)) {
        if (depth == 0) {
            if (n <= 200000) return RDST_ALGO_LR_LSB;
            if (n <= 350000) return RDST_ALGO_SKA;
            if (n <= 4000000) return RDST_ALGO_MT_LSB;
            return RDST_ALGO_REGIONS;
        }
        if (n <= 200000) return RDST_ALGO_LR_LSB;
        if (n <= 800000) return RDST_ALGO_SKA;
        if (n <= 5000000) return RDST_ALGO_RECOMBINATING;
        return RDST_ALGO_REGIONS;
    }
    if (depth > 0) {
        if (n <= 200000) return RDST_ALGO_LSB;
        if (n <= 800000) return RDST_ALGO_SKA;
        if (n <= 50000000) return RDST_ALGO_RECOMBINATING;
        return RDST_ALGO_SCANNING;
    }
    if (n <= 150000) return RDST_ALGO_LSB;
    if (n <= 260000) return RDST_ALGO_SKA;
    if (n <= 50000000) return RDST_ALGO_RECOMBINATING;
    return RDST_ALGO_SCANNING;
}

int low_memory(const rdst_tuning_params* p, const uint64_t counts[256]) {
    const uint64_t n = p->input_len;
    if (n <= 128) return RDST_ALGO_COMPARATIVE;
    if (skewed(n, counts)) {
        if (n <= 50000) return RDST_ALGO_LR_LSB;
        if (n <= 1000000) return RDST_ALGO_SKA;
        return RDST_ALGO_REGIONS;
    }
    if (n <= 50000) return RDST_ALGO_LSB;
    if (n <= 1000000) return RDST_ALGO_SKA;
    return RDST_ALGO_REGIONS;
}

int single_threaded(const rdst_tuning_params* p, const uint64_t counts[256]) {
    const uint64_t n = p->input_len;
    if (n <= 128) return RDST_ALGO_COMPARATIVE;
    const uint64_t depth = p->total_levels - p->level - 1;
    if (skewed(n, counts)) return (n > 100000 && depth < 2) ? RDST_ALGO_SKA : RDST_ALGO_LR_LSB;
    return (n > 800000 && depth == 0) ? RDST_ALGO_SKA : RDST_ALGO_LSB;
}

}  // namespace

extern "C" int rdst_pick_algorithm(int tuner_id, const rdst_tuning_params* p, const uint64_t counts[256],
                                   uint64_t gpu_min_len) {
    if (!p || !counts || p->total_levels == 0 || p->level >= p->total_levels) return RDST_ERR_ARG;
    switch (tuner_id) {
        case RDST_TUNER_STANDARD: return standard(p, counts);
        case RDST_TUNER_LOW_MEMORY: return low_memory(p, counts);
        case RDST_TUNER_SINGLE_THREADED: return single_threaded(p, counts);
        case RDST_TUNER_GPU: {
            // Only a whole top-level slice can go to the device: deeper chunks are produced by
            // a CPU MSD pass and are already cache-sized.
            const uint64_t depth = p->total_levels - p->level - 1;
            if (depth == 0 && p->parent_len < 0 && p->input_len >= gpu_min_len) return RDST_ALGO_GPU_LSD;
            return standard(p, counts);
        }
        default: return RDST_ERR_ARG;
    }
}
