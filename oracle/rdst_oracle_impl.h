/*
 * rdst_oracle_impl.h — per-element-type body of the CPU oracle.  TEST INFRASTRUCTURE ONLY.
 *
 * Included once per element type by rdst_oracle.c with
 *     T          storage type of one element (bit pattern)
 *     SUF        suffix for the generated names
 *     LEVELS     RadixKey::LEVELS
 *     GET_LEVEL(v, level)   RadixKey::get_level for that type (src/radix_key_impl.rs)
 *
 * Every function cites the reference lines it restates (paths relative to the reference
 * tree).  rayon's par_chunks / par_bridge / into_par_iter become OpenMP tasks inside the
 * parallel region opened by the entry point; `threads` is what rayon::current_num_threads()
 * would return.
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define NAME(x) CAT(x, SUF)

/* RadixKeyChecked::get_level_checked — src/radix_key.rs:13-21 (the debug_assert is an assert here) */
static inline uint8_t NAME(lvl)(const T* v, size_t level) {
    assert(level < (size_t)LEVELS);
    return (uint8_t)(GET_LEVEL((*v), level));
}

/* get_counts_with_ends — src/sort_utils.rs:109-180 */
static void NAME(get_counts_with_ends)(const T* bucket, size_t len, size_t level, size_t counts_1[256],
                                       bool* already_sorted_out, uint8_t* first, uint8_t* last_out) {
    memset(counts_1, 0, 256 * sizeof(size_t));
    if (len == 0) { /* :116-118 */
        *already_sorted_out = true;
        *first = 0;
        *last_out = 0;
        return;
    }
    bool already_sorted = true;
    size_t continue_from = len;
    uint8_t last = 0;
    for (size_t i = 0; i < len; ++i) { /* :125-136 */
        const uint8_t b = NAME(lvl)(&bucket[i], level);
        counts_1[b] += 1;
        if (b < last) {
            continue_from = i + 1;
            already_sorted = false;
            break;
        }
        last = b;
    }
    if (continue_from == len) { /* :138-145 */
        *already_sorted_out = already_sorted;
        *first = NAME(lvl)(&bucket[0], level);
        *last_out = last;
        return;
    }
    size_t counts_2[256] = {0}, counts_3[256] = {0}, counts_4[256] = {0}; /* :147-149 */
    const T* rest = bucket + continue_from;
    const size_t rest_len = len - continue_from;
    const size_t n4 = rest_len / 4 * 4;
    for (size_t i = 0; i < n4; i += 4) { /* :153-163 */
        const uint8_t a = NAME(lvl)(&rest[i], level);
        const uint8_t b = NAME(lvl)(&rest[i + 1], level);
        const uint8_t c = NAME(lvl)(&rest[i + 2], level);
        const uint8_t d = NAME(lvl)(&rest[i + 3], level);
        counts_1[a] += 1;
        counts_2[b] += 1;
        counts_3[c] += 1;
        counts_4[d] += 1;
    }
    for (size_t i = n4; i < rest_len; ++i) counts_1[NAME(lvl)(&rest[i], level)] += 1; /* :165-168 */
    for (int i = 0; i < 256; ++i) counts_1[i] += counts_2[i] + counts_3[i] + counts_4[i]; /* :170-174 */
    *already_sorted_out = already_sorted;
    *first = NAME(lvl)(&bucket[0], level);
    *last_out = NAME(lvl)(&bucket[len - 1], level);
}

/* get_counts — src/sort_utils.rs:183-190 */
static void NAME(get_counts)(const T* bucket, size_t len, size_t level, size_t counts[256], bool* sorted) {
    uint8_t f, l;
    NAME(get_counts_with_ends)(bucket, len, level, counts, sorted, &f, &l);
}

/* par_get_counts_with_ends — src/sort_utils.rs:35-106 */
static void NAME(par_get_counts_with_ends)(const T* bucket, size_t len, size_t level, size_t threads,
                                           size_t msb_counts[256], bool* already_sorted_out, uint8_t* first,
                                           uint8_t* last) {
    if (len < 400000) { /* :42-44 */
        NAME(get_counts_with_ends)(bucket, len, level, msb_counts, already_sorted_out, first, last);
        return;
    }
    const size_t chunk_divisor = 8;
    const size_t chunk_size = div_ceil(div_ceil(len, threads), chunk_divisor); /* :46-48 */
    const size_t n_chunks = div_ceil(len, chunk_size);
    size_t* all_counts = (size_t*)xmalloc(n_chunks * 256 * sizeof(size_t));
    uint8_t* meta = (uint8_t*)xmalloc(n_chunks * 3); /* sorted, start, end per chunk */
#pragma omp taskloop default(shared) grainsize(1)
    for (size_t i = 0; i < n_chunks; ++i) { /* :51-57 */
        const size_t b = i * chunk_size;
        const size_t l = (b + chunk_size <= len) ? chunk_size : len - b;
        bool s;
        NAME(get_counts_with_ends)(bucket + b, l, level, all_counts + i * 256, &s, &meta[i * 3 + 1], &meta[i * 3 + 2]);
        meta[i * 3] = s;
    }
    memset(msb_counts, 0, 256 * sizeof(size_t));
    bool already_sorted = true;
    for (size_t i = 0; i < n_chunks; ++i) { /* :74-87 (receive order does not matter for sums) */
        if (!meta[i * 3]) already_sorted = false;
        for (int d = 0; d < 256; ++d) msb_counts[d] += all_counts[i * 256 + d];
    }
    if (already_sorted) { /* :91-98 */
        for (size_t i = 0; i + 1 < n_chunks; ++i) {
            if (meta[(i + 1) * 3 + 1] < meta[i * 3 + 2]) {
                already_sorted = false;
                break;
            }
        }
    }
    *already_sorted_out = already_sorted;
    *first = meta[1];
    *last = meta[(n_chunks - 1) * 3 + 2];
    free(all_counts);
    free(meta);
}

/* get_tile_counts — src/sort_utils.rs:193-244.  Returns malloc'd [tiles][256]; *tiles_out = tile count. */
static size_t* NAME(get_tile_counts)(const T* bucket, size_t len, size_t tile_size, size_t level, size_t threads,
                                     size_t* tiles_out, bool* already_sorted) {
    const size_t tiles = len == 0 ? 0 : div_ceil(len, tile_size);
    size_t* tile_counts = (size_t*)xmalloc((tiles ? tiles : 1) * 256 * sizeof(size_t));
    uint8_t* meta = (uint8_t*)xmalloc((tiles ? tiles : 1) * 3);
#pragma omp taskloop default(shared) grainsize(1)
    for (size_t i = 0; i < tiles; ++i) { /* :209-215: par_chunks(tile_size).map(par_get_counts_with_ends) */
        const size_t b = i * tile_size;
        const size_t l = (b + tile_size <= len) ? tile_size : len - b;
        bool s;
        NAME(par_get_counts_with_ends)(bucket + b, l, level, threads, tile_counts + i * 256, &s, &meta[i * 3 + 1],
                                       &meta[i * 3 + 2]);
        meta[i * 3] = s;
    }
    *tiles_out = tiles;
    if (tiles == 1) { /* :228-231 */
        *already_sorted = meta[0];
    } else {
        bool sorted = true; /* zero tiles: windows(2) is empty -> true (:233-243) */
        for (size_t i = 0; i + 1 < tiles; ++i) {
            if (!meta[i * 3] || !meta[(i + 1) * 3] || meta[(i + 1) * 3 + 1] < meta[i * 3 + 2]) {
                sorted = false;
                break;
            }
        }
        *already_sorted = sorted;
    }
    free(meta);
    return tile_counts;
}

/* out_of_place_sort — src/sorts/out_of_place_sort.rs:52-108 */
static void NAME(out_of_place_sort)(const T* src, T* dst, size_t len, const size_t counts[256], size_t level) {
    if (len < 2) { /* :65-68 */
        if (len) dst[0] = src[0];
        return;
    }
    size_t prefix_sums[256];
    get_prefix_sums(counts, prefix_sums);
    const size_t n8 = len / 8 * 8;
    for (size_t i = 0; i < n8; i += 8) { /* :75-101 */
        const uint8_t a = NAME(lvl)(&src[i], level), b = NAME(lvl)(&src[i + 1], level);
        const uint8_t c = NAME(lvl)(&src[i + 2], level), d = NAME(lvl)(&src[i + 3], level);
        const uint8_t e = NAME(lvl)(&src[i + 4], level), f = NAME(lvl)(&src[i + 5], level);
        const uint8_t g = NAME(lvl)(&src[i + 6], level), h = NAME(lvl)(&src[i + 7], level);
        dst[prefix_sums[a]++] = src[i];
        dst[prefix_sums[b]++] = src[i + 1];
        dst[prefix_sums[c]++] = src[i + 2];
        dst[prefix_sums[d]++] = src[i + 3];
        dst[prefix_sums[e]++] = src[i + 4];
        dst[prefix_sums[f]++] = src[i + 5];
        dst[prefix_sums[g]++] = src[i + 6];
        dst[prefix_sums[h]++] = src[i + 7];
    }
    for (size_t i = n8; i < len; ++i) dst[prefix_sums[NAME(lvl)(&src[i], level)]++] = src[i]; /* :103-107 */
}

/* out_of_place_sort_with_counts — src/sorts/out_of_place_sort.rs:111-199 */
static void NAME(out_of_place_sort_with_counts)(const T* src, T* dst, size_t len, const size_t counts[256], size_t level,
                                                size_t next_counts_0[256]) {
    memset(next_counts_0, 0, 256 * sizeof(size_t));
    if (len == 0) return; /* :124-125 */
    if (len == 1) {       /* :126-132: counts the CURRENT level for a single element */
        dst[0] = src[0];
        next_counts_0[NAME(lvl)(&src[0], level)] = 1;
        return;
    }
    const size_t next_level = level + 1;
    size_t prefix_sums[256], next_counts_1[256] = {0};
    get_prefix_sums(counts, prefix_sums);
    const size_t n8 = len / 8 * 8;
    for (size_t i = 0; i < n8; i += 8) { /* :141-183: scatter + two interleaved next-level arrays */
        for (int k = 0; k < 8; k += 2) {
            const uint8_t b0 = NAME(lvl)(&src[i + k], level), b1 = NAME(lvl)(&src[i + k + 1], level);
            dst[prefix_sums[b0]++] = src[i + k];
            dst[prefix_sums[b1]++] = src[i + k + 1];
            next_counts_0[NAME(lvl)(&src[i + k], next_level)] += 1;
            next_counts_1[NAME(lvl)(&src[i + k + 1], next_level)] += 1;
        }
    }
    for (size_t i = n8; i < len; ++i) { /* :185-191 */
        dst[prefix_sums[NAME(lvl)(&src[i], level)]++] = src[i];
        next_counts_0[NAME(lvl)(&src[i], next_level)] += 1;
    }
    for (int i = 0; i < 256; ++i) next_counts_0[i] += next_counts_1[i]; /* :193-196 */
}

/* lr_out_of_place_sort (+ _with_counts when next_counts != NULL) —
 * src/sorts/out_of_place_sort.rs:202-275 and :278-389 */
static void NAME(lr_out_of_place_sort)(const T* src, T* dst, size_t len, const size_t counts[256], size_t level,
                                       size_t* next_counts) {
    if (next_counts) memset(next_counts, 0, 256 * sizeof(size_t));
    if (next_counts && len == 0) return; /* :291-292 */
    if (next_counts && len == 1) {       /* :293-299 */
        dst[0] = src[0];
        next_counts[NAME(lvl)(&src[0], level)] = 1;
        return;
    }
    if (len < 2) { /* :215-218 */
        if (len) dst[0] = src[0];
        return;
    }
    const size_t next_level = level + 1;
    size_t offsets[256], ends[256], nc1[256] = {0};
    get_prefix_sums(counts, offsets);
    for (int i = 0; i < 256; ++i) ends[i] = offsets[i] + (counts[i] ? counts[i] - 1 : 0); /* :223-225 */
    size_t left = 0, right = len - 1;
    const size_t pre = len % 8;
    for (size_t k = 0; k < pre; ++k) { /* :231-237 */
        const uint8_t b = NAME(lvl)(&src[right], level);
        dst[ends[b]] = src[right];
        ends[b] = ends[b] ? ends[b] - 1 : 0; /* saturating_sub */
        if (next_counts) next_counts[NAME(lvl)(&src[right], next_level)] += 1;
        right = right ? right - 1 : 0;
    }
    if (pre == len) return; /* :239-241 */
    const size_t end = (len - pre) / 2;
    while (left < end) { /* :245-274 */
        for (int k = 0; k < 4; ++k) {
            const uint8_t bl = NAME(lvl)(&src[left + k], level);
            const uint8_t br = NAME(lvl)(&src[right - k], level);
            dst[offsets[bl]] = src[left + k];
            offsets[bl] += 1;
            dst[ends[br]] = src[right - k];
            ends[br] -= 1; /* wrapping_sub */
            if (next_counts) {
                next_counts[NAME(lvl)(&src[left + k], next_level)] += 1;
                nc1[NAME(lvl)(&src[right - k], next_level)] += 1;
            }
        }
        left += 4;
        right -= 4; /* may wrap on the last iteration exactly as the reference's usize does */
    }
    if (next_counts)
        for (int i = 0; i < 256; ++i) next_counts[i] += nc1[i];
}

/* route_out_of_place_sort — src/sorts/out_of_place_sort.rs:392-424.  Returns true if next_counts was filled. */
static bool NAME(route_out_of_place_sort)(bool should_count, bool lr, const T* src, T* dst, size_t len,
                                          const size_t counts[256], size_t level, size_t next_counts[256]) {
    if (lr && should_count) { NAME(lr_out_of_place_sort)(src, dst, len, counts, level, next_counts); return true; }
    if (lr) { NAME(lr_out_of_place_sort)(src, dst, len, counts, level, NULL); return false; }
    if (should_count) { NAME(out_of_place_sort_with_counts)(src, dst, len, counts, level, next_counts); return true; }
    NAME(out_of_place_sort)(src, dst, len, counts, level);
    return false;
}

/* Sorter::lsb_sort_adapter — src/sorts/lsb_sort.rs:39-127 */
static void NAME(lsb_sort_adapter)(bool lr, T* bucket, size_t len, const size_t end_counts[256], size_t start_level,
                                   size_t end_level) {
    if (len < 2) return;
    T* tmp_bucket = (T*)xmalloc(len * sizeof(T)); /* :53 */
    bool invert = false;
    size_t held[256], next[256];
    const size_t* level_counts;
    for (;;) { /* :62-83 */
        if (start_level == end_level) {
            level_counts = end_counts;
            break;
        }
        bool already_sorted;
        NAME(get_counts)(bucket, len, start_level, held, &already_sorted);
        if (!already_sorted) {
            level_counts = held;
            break;
        }
        start_level += 1;
    }
    for (size_t level = start_level; level <= end_level; ++level) { /* :85-115 */
        const T* src = invert ? tmp_bucket : bucket;
        T* dst = invert ? bucket : tmp_bucket;
        const bool should_count = end_level != 0 && level < (end_level - 1); /* :101 */
        const bool got = NAME(route_out_of_place_sort)(should_count, lr, src, dst, len, level_counts, level, next);
        if (got) {
            memcpy(held, next, sizeof held);
            level_counts = held;
        } else {
            level_counts = end_counts; /* :112 */
        }
        invert = !invert;
    }
    if (invert) memcpy(bucket, tmp_bucket, len * sizeof(T)); /* :117-126 */
    free(tmp_bucket);
}

/* mt_lsb_sort — src/sorts/mt_lsb_sort.rs:40-133: destination carved bucket-major, tile-minor;
 * every tile scatters with left/right cursors into its own 256 sub-slices */
static void NAME(mt_lsb_sort)(const T* src, T* dst, size_t len, const size_t* tile_counts, size_t tiles,
                              size_t tile_size, size_t level) {
    /* start[b*tiles + tile] = offset of that sub-slice inside dst (:51-54) */
    size_t* start = (size_t*)xmalloc(256 * tiles * sizeof(size_t));
    size_t run = 0;
    for (size_t b = 0; b < 256; ++b)
        for (size_t t = 0; t < tiles; ++t) {
            start[b * tiles + t] = run;
            run += tile_counts[t * 256 + b];
        }
#pragma omp taskloop default(shared) grainsize(1)
    for (size_t tile = 0; tile < tiles; ++tile) { /* :65-132 */
        const T* bucket = src + tile * tile_size;
        const size_t blen = (tile * tile_size + tile_size <= len) ? tile_size : len - tile * tile_size;
        if (blen == 0) continue;
        size_t offsets[256], ends[256];
        for (size_t b = 0; b < 256; ++b) {
            const size_t c = tile_counts[tile * 256 + b];
            offsets[b] = start[b * tiles + tile];
            ends[b] = offsets[b] + (c ? c - 1 : 0); /* :74-80 (absolute instead of slice-relative) */
        }
        size_t left = 0, right = blen - 1;
        const size_t pre = blen % 8;
        for (size_t k = 0; k < pre; ++k) { /* :86-92 */
            const uint8_t b = NAME(lvl)(&bucket[right], level);
            dst[ends[b]] = bucket[right];
            ends[b] -= 1;
            right = right ? right - 1 : 0;
        }
        if (pre == blen) continue; /* :94-96 */
        const size_t end = (blen - pre) / 2;
        while (left < end) { /* :100-130 */
            for (int k = 0; k < 4; ++k) {
                const uint8_t bl = NAME(lvl)(&bucket[left + k], level);
                const uint8_t br = NAME(lvl)(&bucket[right - k], level);
                dst[offsets[bl]] = bucket[left + k];
                offsets[bl] += 1;
                dst[ends[br]] = bucket[right - k];
                ends[br] -= 1;
            }
            left += 4;
            right -= 4;
        }
    }
    free(start);
}

static void NAME(par_copy_tiles)(T* dst, const T* src, size_t len, size_t tile_size) {
    const size_t tiles = div_ceil(len, tile_size);
#pragma omp taskloop default(shared) grainsize(1)
    for (size_t t = 0; t < tiles; ++t) {
        const size_t b = t * tile_size;
        const size_t l = (b + tile_size <= len) ? tile_size : len - b;
        memcpy(dst + b, src + b, l * sizeof(T));
    }
}

static void NAME(route)(const struct sorter* s, T* bucket, size_t len, const size_t counts[256], size_t level);

/* Sorter::mt_lsb_sort_adapter — src/sorts/mt_lsb_sort.rs:136-195 */
static void NAME(mt_lsb_sort_adapter)(const struct sorter* s, T* bucket, size_t len, size_t start_level,
                                      size_t end_level, size_t tile_size) {
    if (len < 2) return;
    T* tmp = (T*)xmalloc(len * sizeof(T));
    bool invert = false;
    for (size_t level = start_level; level <= end_level; ++level) {
        const T* src = invert ? tmp : bucket;
        T* dst = invert ? bucket : tmp;
        size_t tiles;
        bool already_sorted;
        size_t* tc = NAME(get_tile_counts)(src, len, tile_size, level, s->threads, &tiles, &already_sorted);
        if (already_sorted) { /* :170-172 */
            free(tc);
            continue;
        }
        NAME(mt_lsb_sort)(src, dst, len, tc, tiles, tile_size, level);
        free(tc);
        invert = !invert;
    }
    if (invert) NAME(par_copy_tiles)(bucket, tmp, len, tile_size); /* :179-194 */
    free(tmp);
}

/* Sorter::mt_oop_sort_adapter — src/sorts/mt_lsb_sort.rs:197-235 */
static void NAME(mt_oop_sort_adapter)(const struct sorter* s, T* bucket, size_t len, size_t level,
                                      const size_t counts[256], const size_t* tile_counts, size_t tiles,
                                      size_t tile_size) {
    if (len <= 1) return;
    T* tmp = (T*)xmalloc(len * sizeof(T));
    NAME(mt_lsb_sort)(bucket, tmp, len, tile_counts, tiles, tile_size, level);
    NAME(par_copy_tiles)(bucket, tmp, len, tile_size);
    free(tmp);
    if (level == 0) return;
    NAME(route)(s, bucket, len, counts, level - 1);
}

/* recombinating_sort — src/sorts/recombinating_sort.rs:32-89 */
static void NAME(recombinating_sort)(T* bucket, size_t len, const size_t counts[256], const size_t* tile_counts,
                                     size_t tiles, size_t tile_size, size_t level) {
    T* tmp = (T*)xmalloc(len * sizeof(T));
    size_t* sums = (size_t*)xmalloc(tiles * 256 * sizeof(size_t));
#pragma omp taskloop default(shared) grainsize(1)
    for (size_t t = 0; t < tiles; ++t) { /* :44-55: tile-local counting sort into the same tile of tmp */
        const size_t b = t * tile_size;
        const size_t l = (b + tile_size <= len) ? tile_size : len - b;
        NAME(out_of_place_sort)(bucket + b, tmp + b, l, tile_counts + t * 256, level);
        get_prefix_sums(tile_counts + t * 256, sums + t * 256);
    }
    size_t gstart[256];
    get_prefix_sums(counts, gstart);
#pragma omp taskloop default(shared) grainsize(8)
    for (size_t index = 0; index < 256; ++index) { /* :68-88: per global bucket, gather its run from every tile */
        T* global_chunk = bucket + gstart[index];
        size_t read_offset = 0, write_offset = 0;
        for (size_t t = 0; t < tiles; ++t) {
            const size_t read_start = read_offset + sums[t * 256 + index];
            const size_t n = tile_counts[t * 256 + index];
            memcpy(global_chunk + write_offset, tmp + read_start, n * sizeof(T));
            read_offset += tile_size;
            write_offset += n;
        }
    }
    free(sums);
    free(tmp);
}

/* partition_index — src/sort_utils.rs:295-331, specialised to "digit == want" */
static size_t NAME(partition_index_eq)(T* data, size_t len, size_t level, uint8_t want) {
    size_t front = 0, back = len, left_count = 0;
    for (;;) {
        T* left_item = NULL;
        while (front < back) {
            T* it = &data[front++];
            if (NAME(lvl)(it, level) == want) left_count += 1;
            else { left_item = it; break; }
        }
        if (!left_item) return left_count;
        T* right_item = NULL;
        while (front < back) {
            T* it = &data[--back];
            if (NAME(lvl)(it, level) == want) { right_item = it; break; }
        }
        if (!right_item) return left_count;
        const T tmp = *left_item;
        *left_item = *right_item;
        *right_item = tmp;
        left_count += 1;
    }
}

/* ska_sort — src/sorts/ska_sort.rs:28-88 */
static void NAME(ska_sort)(T* bucket, size_t len, size_t prefix_sums[256], const size_t end_offsets[256], size_t level) {
    size_t finished = 0, largest = 0;
    bool finished_map[256] = {false};
    uint8_t largest_index = 0;
    for (int i = 0; i < 256; ++i) { /* :41-50 */
        const size_t rem = end_offsets[i] - prefix_sums[i];
        if (rem == 0) {
            finished_map[i] = true;
            finished += 1;
        } else if (rem > largest) {
            largest = rem;
            largest_index = (uint8_t)i;
        }
    }
    if (largest == len) return; /* :52-54 */
    if (largest > len / 2) {    /* :55-65 */
        const size_t offs = NAME(partition_index_eq)(bucket + prefix_sums[largest_index],
                                                     end_offsets[largest_index] - prefix_sums[largest_index], level,
                                                     largest_index);
        prefix_sums[largest_index] += offs;
    }
    if (!finished_map[largest_index]) { /* :67-70 */
        finished_map[largest_index] = true;
        finished += 1;
    }
    while (finished != 256) { /* :72-87 */
        for (int b = 0; b < 256; ++b) {
            if (finished_map[b]) continue;
            if (prefix_sums[b] >= end_offsets[b]) {
                finished_map[b] = true;
                finished += 1;
            }
            for (size_t i = prefix_sums[b]; i < end_offsets[b]; ++i) {
                const uint8_t new_b = NAME(lvl)(&bucket[i], level);
                const T tmp = bucket[prefix_sums[new_b]];
                bucket[prefix_sums[new_b]] = bucket[i];
                bucket[i] = tmp;
                prefix_sums[new_b] += 1;
            }
        }
    }
}

/* Sorter::ska_sort_adapter — src/sorts/ska_sort.rs:91-113 */
static void NAME(ska_sort_adapter)(const struct sorter* s, T* bucket, size_t len, const size_t counts[256], size_t level) {
    if (len < 2) return;
    size_t prefix_sums[256], end_offsets[256];
    get_prefix_sums(counts, prefix_sums);
    get_end_offsets(counts, prefix_sums, end_offsets);
    NAME(ska_sort)(bucket, len, prefix_sums, end_offsets, level);
    if (level == 0) return;
    NAME(route)(s, bucket, len, counts, level - 1);
}

/* comparative_sort / cmp_packed — src/sorts/comparative_sort.rs:30-118: order by the key bytes
 * of levels start_level..=0, most significant first.  (The reference packs them into the
 * narrowest integer; comparing byte by byte from the top gives the same order.) */
static int NAME(cmp_levels)(const void* pa, const void* pb, void* ctx) {
    const size_t start_level = *(const size_t*)ctx;
    const T* a = (const T*)pa;
    const T* b = (const T*)pb;
    for (size_t level = start_level + 1; level-- > 0;) {
        const uint8_t al = NAME(lvl)(a, level), bl = NAME(lvl)(b, level);
        if (al != bl) return al < bl ? -1 : 1;
    }
    return 0;
}
static void NAME(comparative_sort)(T* bucket, size_t len, size_t start_level) {
    if (len < 2) return;
    qsort_r(bucket, len, sizeof(T), NAME(cmp_levels), &start_level);
}

/* ---- scanning sort — src/sorts/scanning_sort.rs:45-267 ------------------------------------ */
struct NAME(scanner_bucket) {
    uint8_t index;
    size_t len;
    omp_lock_t lock;
    size_t write_head, read_head;
    T* chunk;
    bool locally_partitioned;
};
struct NAME(stash) { T* v; size_t len, cap; };

static void NAME(stash_push)(struct NAME(stash) * s, T v) {
    if (s->len == s->cap) {
        s->cap = s->cap ? s->cap * 2 : 64;
        s->v = (T*)xrealloc(s->v, s->cap * sizeof(T));
    }
    s->v[s->len++] = v;
}

static int NAME(cmp_bucket_len_desc)(const void* a, const void* b) {
    const struct NAME(scanner_bucket)* x = (const struct NAME(scanner_bucket)*)a;
    const struct NAME(scanner_bucket)* y = (const struct NAME(scanner_bucket)*)b;
    if (x->len != y->len) return x->len > y->len ? -1 : 1;
    return (int)x->index - (int)y->index; /* sort_by_key is stable (:87) */
}

/* scanner_thread — :92-219 */
static void NAME(scanner_thread)(struct NAME(scanner_bucket) * sb, size_t n_buckets, size_t level,
                                 size_t scanner_read_size, size_t uniform_threshold) {
    struct NAME(stash) stash[256];
    memset(stash, 0, sizeof stash);
    size_t finished_count = 0;
    bool finished_map[256] = {false};
    for (size_t k = 0; k < n_buckets; ++k) { /* :108-126: pre-partition heavy buckets in place */
        struct NAME(scanner_bucket)* m = &sb[k];
        if (m->len < uniform_threshold) continue;
        if (!omp_test_lock(&m->lock)) continue;
        if (!m->locally_partitioned) {
            m->locally_partitioned = true;
            const size_t start = NAME(partition_index_eq)(m->chunk, m->len, level, m->index);
            m->read_head = start;
            m->write_head = start;
        }
        omp_unset_lock(&m->lock);
    }
    bool done = false;
    while (!done) { /* :128-218 */
        for (size_t k = 0; k < n_buckets && !done; ++k) {
            struct NAME(scanner_bucket)* m = &sb[k];
            if (finished_map[m->index]) continue;
            if (!omp_test_lock(&m->lock)) continue;
            if (m->write_head >= m->len) { /* :139-148 */
                finished_count += 1;
                finished_map[m->index] = true;
                omp_unset_lock(&m->lock);
                if (finished_count == n_buckets) done = true;
                continue;
            }
            const size_t remaining = m->len - m->read_head;
            const size_t to_read = remaining < scanner_read_size ? remaining : scanner_read_size; /* :151 */
            if (to_read > 0) { /* :153-188 */
                const T* rd = m->chunk + m->read_head;
                for (size_t i = 0; i < to_read; ++i) NAME(stash_push)(&stash[NAME(lvl)(&rd[i], level)], rd[i]);
                m->read_head += to_read;
            }
            struct NAME(stash)* mine = &stash[m->index];
            const size_t room = m->read_head - m->write_head;
            const size_t to_write = mine->len < room ? mine->len : room; /* :190-193 */
            if (to_write >= 1) {                                           /* :195-207 */
                const size_t split = mine->len - to_write;
                memcpy(m->chunk + m->write_head, mine->v + split, to_write * sizeof(T));
                mine->len = split;
                m->write_head += to_write;
                if (m->write_head >= m->len) { /* :209-216 */
                    finished_count += 1;
                    finished_map[m->index] = true;
                    if (finished_count == n_buckets) done = true;
                }
            }
            omp_unset_lock(&m->lock);
        }
    }
    for (int i = 0; i < 256; ++i) free(stash[i].v);
}

/* scanning_sort — :221-242 */
static void NAME(scanning_sort)(const struct sorter* s, T* bucket, size_t len, const size_t counts[256], size_t level) {
    size_t threads = s->threads;
    const size_t uniform_threshold = (size_t)((double)(len / threads) * 1.4);
    struct NAME(scanner_bucket)* sb = (struct NAME(scanner_bucket)*)xmalloc(256 * sizeof *sb);
    size_t off = 0;
    for (int i = 0; i < 256; ++i) { /* get_scanner_buckets :59-90; head = prefix - running = 0 */
        sb[i].index = (uint8_t)i;
        sb[i].len = counts[i];
        sb[i].chunk = bucket + off;
        sb[i].write_head = 0;
        sb[i].read_head = 0;
        sb[i].locally_partitioned = false;
        omp_init_lock(&sb[i].lock);
        off += counts[i];
    }
    qsort(sb, 256, sizeof *sb, NAME(cmp_bucket_len_desc));
    if (threads > 256) threads = 256;
    size_t lg = 0; /* ceil(log2(threads)) */
    while (((size_t)1 << lg) < threads) ++lg;
    const size_t scaling_factor = lg > 1 ? lg : 1;
    const size_t scanner_read_size = 32768 / scaling_factor; /* :231-232 */
#pragma omp taskloop default(shared) grainsize(1)
    for (size_t t = 0; t < threads; ++t) NAME(scanner_thread)(sb, 256, level, scanner_read_size, uniform_threshold);
    for (int i = 0; i < 256; ++i) omp_destroy_lock(&sb[i].lock);
    free(sb);
}

/* ---- regions sort — src/sorts/regions_sort.rs:51-286 --------------------------------------- */
struct NAME(edge) { uint8_t dst, init; T* slice; size_t len; };
struct NAME(edge_vec) { struct NAME(edge) * v; size_t len, cap; };
struct NAME(op) { struct NAME(edge) a, b; };

static void NAME(ev_push)(struct NAME(edge_vec) * ev, struct NAME(edge) e) {
    if (ev->len == ev->cap) {
        ev->cap = ev->cap ? ev->cap * 2 : 256;
        ev->v = (struct NAME(edge)*)xrealloc(ev->v, ev->cap * sizeof *ev->v);
    }
    ev->v[ev->len++] = e;
}

/* partition_index over edges with "field != country" (sort_utils.rs:295-331) */
static size_t NAME(partition_edges)(struct NAME(edge) * data, size_t len, uint8_t country, bool by_init) {
    size_t front = 0, back = len, left_count = 0;
    for (;;) {
        struct NAME(edge)* left_item = NULL;
        while (front < back) {
            struct NAME(edge)* it = &data[front++];
            if ((by_init ? it->init : it->dst) != country) left_count += 1;
            else { left_item = it; break; }
        }
        if (!left_item) return left_count;
        struct NAME(edge)* right_item = NULL;
        while (front < back) {
            struct NAME(edge)* it = &data[--back];
            if ((by_init ? it->init : it->dst) != country) { right_item = it; break; }
        }
        if (!right_item) return left_count;
        const struct NAME(edge) tmp = *left_item;
        *left_item = *right_item;
        *right_item = tmp;
        left_count += 1;
    }
}

/* generate_outbounds — :66-123 */
static void NAME(generate_outbounds)(T* bucket, const size_t* local_counts, size_t n_local, const size_t global_counts[256],
                                     struct NAME(edge_vec) * outbounds) {
    T* rem = bucket;
    size_t local_bucket = 0;
    unsigned local_country = 0, global_country = 0;
    size_t target_global_dist = global_counts[0];
    size_t target_local_dist = local_counts[0];
    while (!(global_country == 255 && local_country == 255 && local_bucket == n_local - 1)) {
        const size_t step = target_global_dist < target_local_dist ? target_global_dist : target_local_dist;
        if (step != 0) {
            T* slice = rem;
            rem += step;
            if (local_country != global_country) {
                struct NAME(edge) e = {(uint8_t)local_country, (uint8_t)global_country, slice, step};
                NAME(ev_push)(outbounds, e);
            }
        }
        if (step == target_global_dist && global_country < 255) {
            global_country += 1;
            target_global_dist = global_counts[global_country];
        } else {
            target_global_dist -= step;
        }
        if (step == target_local_dist && !(local_bucket == n_local - 1 && local_country == 255)) {
            if (local_country < 255) local_country += 1;
            else { local_bucket += 1; local_country = 0; }
            target_local_dist = local_counts[local_bucket * 256 + local_country];
        } else {
            target_local_dist -= step;
        }
    }
}

/* regions_sort — :206-262 (list_operations :126-204 inlined as the per-country block) */
static void NAME(regions_sort)(const struct sorter* s, T* bucket, size_t len, const size_t counts[256],
                               const size_t* tile_counts, size_t tiles, size_t tile_size, size_t level) {
#pragma omp taskloop default(shared) grainsize(1)
    for (size_t t = 0; t < tiles; ++t) { /* :216-223 */
        const size_t b = t * tile_size;
        const size_t l = (b + tile_size <= len) ? tile_size : len - b;
        size_t ps[256], eo[256];
        get_prefix_sums(tile_counts + t * 256, ps);
        get_end_offsets(tile_counts + t * 256, ps, eo);
        NAME(ska_sort)(bucket + b, l, ps, eo, level);
    }
    struct NAME(edge_vec) outbounds = {0}, current = {0}, inbounds = {0};
    struct NAME(op)* ops = NULL;
    size_t n_ops = 0, cap_ops = 0;
    NAME(generate_outbounds)(bucket, tile_counts, tiles, counts, &outbounds);
    for (;;) { /* :229-261 */
        if (outbounds.len == 0) break;
        for (unsigned country = 0; country < 256; ++country) { /* list_operations */
            const size_t ob = NAME(partition_edges)(outbounds.v, outbounds.len, (uint8_t)country, true);
            current.len = 0;
            for (size_t i = ob; i < outbounds.len; ++i) NAME(ev_push)(&current, outbounds.v[i]);
            outbounds.len = ob;
            const size_t p = NAME(partition_edges)(outbounds.v, outbounds.len, (uint8_t)country, false);
            inbounds.len = 0;
            for (size_t i = p; i < outbounds.len; ++i) NAME(ev_push)(&inbounds, outbounds.v[i]);
            outbounds.len = p;
            for (;;) {
                if (inbounds.len == 0) { /* :142-148 */
                    for (size_t i = 0; i < current.len; ++i) NAME(ev_push)(&outbounds, current.v[i]);
                    break;
                }
                struct NAME(edge) i_e = inbounds.v[--inbounds.len];
                if (current.len == 0) { /* :150-157 */
                    NAME(ev_push)(&outbounds, i_e);
                    for (size_t i = 0; i < inbounds.len; ++i) NAME(ev_push)(&outbounds, inbounds.v[i]);
                    break;
                }
                struct NAME(edge) o_e = current.v[--current.len];
                if (i_e.len < o_e.len) { /* :161-177 */
                    struct NAME(edge) rem_e = {o_e.dst, o_e.init, o_e.slice + i_e.len, o_e.len - i_e.len};
                    NAME(ev_push)(&current, rem_e);
                    o_e.len = i_e.len;
                } else if (i_e.len > o_e.len) { /* :178-194 */
                    struct NAME(edge) rem_e = {i_e.dst, i_e.init, i_e.slice + o_e.len, i_e.len - o_e.len};
                    NAME(ev_push)(&inbounds, rem_e);
                    i_e.len = o_e.len;
                }
                if (n_ops == cap_ops) {
                    cap_ops = cap_ops ? cap_ops * 2 : 256;
                    ops = (struct NAME(op)*)xrealloc(ops, cap_ops * sizeof *ops);
                }
                ops[n_ops].a = i_e;
                ops[n_ops].b = o_e;
                n_ops += 1;
            }
        }
        if (n_ops == 0) break; /* :242-244 */
        const size_t chunk = div_ceil(n_ops, s->threads);
        const size_t n_chunks = div_ceil(n_ops, chunk);
#pragma omp taskloop default(shared) grainsize(1)
        for (size_t c = 0; c < n_chunks; ++c) { /* :247-251 swap_with_slice */
            const size_t e = (c + 1) * chunk < n_ops ? (c + 1) * chunk : n_ops;
            for (size_t k = c * chunk; k < e; ++k) {
                T* x = ops[k].a.slice;
                T* y = ops[k].b.slice;
                for (size_t j = 0; j < ops[k].a.len; ++j) {
                    const T tmp = x[j];
                    x[j] = y[j];
                    y[j] = tmp;
                }
            }
        }
        for (size_t k = 0; k < n_ops; ++k) { /* :254-260 */
            struct NAME(edge) i_e = ops[k].a, o_e = ops[k].b;
            if (o_e.dst != i_e.init) {
                o_e.init = i_e.init;
                o_e.slice = i_e.slice;
                NAME(ev_push)(&outbounds, o_e);
            }
        }
        n_ops = 0;
    }
    free(outbounds.v);
    free(current.v);
    free(inbounds.v);
    free(ops);
}

/* ---- Sorter — src/sorter.rs:10-172 ---------------------------------------------------------- */

/* Sorter::handle_chunk — src/sorter.rs:24-103 */
static void NAME(handle_chunk)(const struct sorter* s, T* chunk, size_t len, size_t level, bool has_parent,
                               size_t parent_len, size_t threads) {
    if (len <= 1) return;
    if (len <= 128) { /* :33-38 */
        NAME(comparative_sort)(chunk, len, level);
        return;
    }
    size_t tile_size = len; /* :41-45 */
    if (s->multi_threaded && len >= 260000) {
        tile_size = div_ceil(len, threads);
        if (tile_size < 30000) tile_size = 30000;
    }
    size_t tiles;
    bool already_sorted;
    size_t* tile_counts = NAME(get_tile_counts)(chunk, len, tile_size, level, s->threads, &tiles, &already_sorted);
    size_t counts[256];
    aggregate_tile_counts(tile_counts, tiles, counts); /* :51-57 */
    if (already_sorted) {                              /* :59-65 */
        free(tile_counts);
        if (level != 0) NAME(route)(s, chunk, len, counts, level - 1);
        return;
    }
    struct rdst_o_tuning_params p = {threads, level, LEVELS, len, has_parent ? (int64_t)parent_len : -1};
    const int algorithm = s->pick(s->pick_ctx, &p, counts); /* :67-76 */
    if (s->trace) s->trace(s->pick_ctx, level, len, algorithm);
    switch (algorithm) { /* :81-102 */
        case ALGO_SCANNING:
            if (len >= 2) {
                NAME(scanning_sort)(s, chunk, len, counts, level);
                if (level != 0) NAME(route)(s, chunk, len, counts, level - 1);
            }
            break;
        case ALGO_RECOMBINATING: /* recombinating_sort_adapter — recombinating_sort.rs:92-113 */
            if (len >= 2) {
                NAME(recombinating_sort)(chunk, len, counts, tile_counts, tiles, tile_size, level);
                if (level != 0) NAME(route)(s, chunk, len, counts, level - 1);
            }
            break;
        case ALGO_LR_LSB: NAME(lsb_sort_adapter)(true, chunk, len, counts, 0, level); break;
        case ALGO_LSB: NAME(lsb_sort_adapter)(false, chunk, len, counts, 0, level); break;
        case ALGO_SKA: NAME(ska_sort_adapter)(s, chunk, len, counts, level); break;
        case ALGO_COMPARATIVE: NAME(comparative_sort)(chunk, len, level); break;
        case ALGO_REGIONS: /* regions_sort_adapter — regions_sort.rs:264-286 */
            if (len >= 2) {
                NAME(regions_sort)(s, chunk, len, counts, tile_counts, tiles, tile_size, level);
                if (level != 0) NAME(route)(s, chunk, len, counts, level - 1);
            }
            break;
        case ALGO_MT_OOP: NAME(mt_oop_sort_adapter)(s, chunk, len, level, counts, tile_counts, tiles, tile_size); break;
        case ALGO_MT_LSB: NAME(mt_lsb_sort_adapter)(s, chunk, len, 0, level, tile_size); break;
        default: abort();
    }
    free(tile_counts);
}

/* Sorter::route / route_multi_threaded / route_single_threaded — src/sorter.rs:121-171 */
static void NAME(route)(const struct sorter* s, T* bucket, size_t len, const size_t counts[256], size_t level) {
    size_t off[257];
    off[0] = 0;
    for (int i = 0; i < 256; ++i) off[i + 1] = off[i] + counts[i];
    (void)len;
    if (s->multi_threaded) { /* :131-138: par_bridge over the 256 sub-slices */
        const size_t threads = s->threads;
#pragma omp taskloop default(shared) grainsize(1)
        for (int i = 0; i < 256; ++i)
            NAME(handle_chunk)(s, bucket + off[i], counts[i], level, true, off[256], threads);
    } else { /* :151-157 */
        for (int i = 0; i < 256; ++i) NAME(handle_chunk)(s, bucket + off[i], counts[i], level, true, off[256], 1);
    }
}

/* RadixSortBuilder::sort + Sorter::route_top_level — radix_sort_builder.rs:149-157, sorter.rs:106-119 */
static void NAME(sort_top)(const struct sorter* s, T* data, size_t len) {
    if (len <= 1) return;
    const size_t threads = s->threads;
#pragma omp parallel num_threads((int)s->omp_threads)
#pragma omp single
    NAME(handle_chunk)(s, data, len, LEVELS - 1, false, 0, threads);
}

#undef T
#undef SUF
#undef LEVELS
#undef GET_LEVEL
#undef NAME
#undef CAT
#undef CAT_
