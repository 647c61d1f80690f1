// rdst_regions.cpp — host side of the low-memory device route: the swap plan of one Regions-sort level.
//
// Device twin of src/sorts/regions_sort.rs:51-286 (Obeya et al., "Theoretically-Efficient and Practical Parallel
// In-Place Radix Sorting", SPAA'19), the algorithm LowMemoryTuner picks above 10^6 elements
// (src/tuners/low_memory_tuner.rs:36-41).  The reference: (1) every tile is sorted by the level's digit in place
// (ska_sort per tile, :216-223); (2) from the tile x digit counts it lists, per "country" (the final region of a
// digit), the runs that sit there but belong elsewhere (outbound edges, :66-123), pairs them into swaps of equal
// length (:126-204) — serially (:235-239) — executes the swaps in parallel and loops until nothing is misplaced
// (:229-261).  Here step (1) is a K3 pass per tile through a scratch buffer of one tile (rdst_kernels.hip), the
// counts come back to the host, THIS file plans the swaps, and a swap kernel runs them round by round.
//
// The plan.  After step (1) tile t holds, in digit order, one run per digit d: tile_counts[t][d] keys (in general: one
// run per COLUMN, and col_bucket says which bucket a column's keys belong to — the partition uses three columns,
// below / equal / above the chosen digit, for two buckets).  Country d
// is [start[d], start[d + 1]) with start = exclusive scan of the column sums.  A run is cut at the country borders it
// crosses; a piece that lies in country c with digit d != c is FOREIGN in c and HOMELESS for d (the keys of the two
// kinds balance per country: foreign space in d == homeless keys of d).  One operation swaps m keys of a foreign piece
// X in country d (digit e) with m keys of a homeless piece Y of d (lying in country c): X's slots are then settled
// for good (settled keys never move again), Y's slots hold e-keys in c — settled if e == c, else a new piece foreign
// in c.  Operations of one round touch pairwise disjoint memory (a piece a swap has written is left alone until the
// next round), so a round is one kernel launch; every operation settles at least m keys, so the rounds end.
//
// Host only: no HIP here (bound by tests/test_regions_plan.py without a device).
#include <stdint.h>
#include <stddef.h>

#include <algorithm>
#include <string>
#include <vector>

#include "rdst_hip.h"

namespace {

struct Piece {
    uint64_t pos, len;
    uint32_t digit, country;
    uint32_t fresh_round;  // written by a swap of this round: usable from the next one on
};

}  // namespace

extern "C" int rdst_regions_plan(const uint64_t* tile_counts, uint64_t tiles, uint64_t tile_len, uint64_t len, uint32_t columns,
                                 const uint32_t* col_bucket, uint32_t buckets, rdst_swap_op* ops_out, uint64_t ops_capacity,
                                 uint64_t* nops_out, uint64_t* round_starts, uint32_t max_rounds, uint32_t* nrounds_out,
                                 uint64_t* starts_out) {
    if (!tile_counts || !nops_out || !nrounds_out || !round_starts || buckets == 0 || buckets > 256 || columns == 0 || columns > 256 ||
        tile_len == 0)
        return RDST_ERR_ARG;
    if (tiles != (len + tile_len - 1) / tile_len) return RDST_ERR_ARG;
    if (!col_bucket && columns != buckets) return RDST_ERR_ARG;
    auto bucket_of = [&](uint32_t col) -> uint32_t { return col_bucket ? col_bucket[col] : col; };
    for (uint32_t c = 0; c < columns; ++c)
        if (bucket_of(c) >= buckets) return RDST_ERR_ARG;
    *nops_out = 0;
    *nrounds_out = 0;
    round_starts[0] = 0;
    // country borders
    std::vector<uint64_t> start(buckets + 1, 0);
    for (uint64_t t = 0; t < tiles; ++t) {
        uint64_t in_tile = 0;
        for (uint32_t c = 0; c < columns; ++c) {
            start[bucket_of(c) + 1] += tile_counts[t * columns + c];
            in_tile += tile_counts[t * columns + c];
        }
        const uint64_t expect = t + 1 < tiles ? tile_len : len - t * tile_len;
        if (in_tile != expect) return RDST_ERR_ARG;  // the counts of a tile must add up to its length
    }
    for (uint32_t d = 0; d < buckets; ++d) start[d + 1] += start[d];
    if (start[buckets] != len) return RDST_ERR_ARG;
    if (starts_out)
        for (uint32_t d = 0; d <= buckets; ++d) starts_out[d] = start[d];
    auto country_of = [&](uint64_t pos) -> uint32_t {  // the country that contains position pos (empty countries skipped)
        return (uint32_t)(std::upper_bound(start.begin(), start.end(), pos) - start.begin() - 1);
    };
    // pieces: runs cut at country borders; only the misplaced ones are kept.  A piece is listed twice: under the
    // country it lies in (foreign there) and under its digit (homeless for it).  Lists only grow; consumed pieces have len 0.
    std::vector<Piece> pieces;
    std::vector<std::vector<uint32_t>> f_idx(buckets), h_idx(buckets);
    std::vector<size_t> f_head(buckets, 0), h_head(buckets, 0);
    auto add_piece = [&](uint64_t pos, uint64_t plen, uint32_t digit, uint32_t fresh) {
        while (plen > 0) {
            const uint32_t c = country_of(pos);
            const uint64_t room = start[c + 1] - pos;
            const uint64_t m = plen < room ? plen : room;
            if (c != digit) {
                pieces.push_back({pos, m, digit, c, fresh});
                f_idx[c].push_back((uint32_t)pieces.size() - 1);
                h_idx[digit].push_back((uint32_t)pieces.size() - 1);
            }
            pos += m;
            plen -= m;
        }
    };
    for (uint64_t t = 0; t < tiles; ++t) {
        uint64_t pos = t * tile_len;
        for (uint32_t col = 0; col < columns; ++col) {  // the runs of a tile, in the order they lie there
            const uint64_t c = tile_counts[t * columns + col];
            if (c) add_piece(pos, c, bucket_of(col), 0);
            pos += c;
        }
    }
    uint64_t nops = 0;
    uint32_t round = 0;
    for (;;) {
        ++round;  // pieces with fresh_round == round are written in this round: left alone until the next
        bool progressed = false;
        for (uint32_t k = 0; k < buckets; ++k) {
            const uint32_t d = (k + round) % buckets;  // rotate the first country: it always finds its pieces untouched
            std::vector<uint32_t>& F = f_idx[d];
            std::vector<uint32_t>& H = h_idx[d];
            while (f_head[d] < F.size() && pieces[F[f_head[d]]].len == 0) ++f_head[d];
            while (h_head[d] < H.size() && pieces[H[h_head[d]]].len == 0) ++h_head[d];
            size_t fi = f_head[d], hi = h_head[d];
            while (fi < F.size() && hi < H.size()) {
                const uint32_t xi = F[fi], yi = H[hi];
                if (pieces[xi].len == 0 || pieces[xi].fresh_round == round) { ++fi; continue; }
                if (pieces[yi].len == 0 || pieces[yi].fresh_round == round) { ++hi; continue; }
                const uint64_t m = pieces[xi].len < pieces[yi].len ? pieces[xi].len : pieces[yi].len;
                if (nops >= ops_capacity) return RDST_ERR_ARG;
                ops_out[nops++] = {pieces[xi].pos, pieces[yi].pos, m};
                progressed = true;
                const uint32_t e = pieces[xi].digit, c = pieces[yi].country;
                const uint64_t ypos = pieces[yi].pos;
                // X's first m slots are settled for good; Y's first m slots now hold e-keys in country c
                pieces[xi].pos += m; pieces[xi].len -= m;
                pieces[yi].pos += m; pieces[yi].len -= m;
                if (e != c) {
                    pieces.push_back({ypos, m, e, c, round});
                    f_idx[c].push_back((uint32_t)pieces.size() - 1);
                    h_idx[e].push_back((uint32_t)pieces.size() - 1);
                }
            }
        }
        bool any_left = false;
        for (uint32_t d = 0; d < buckets && !any_left; ++d)
            for (size_t i = f_head[d]; i < f_idx[d].size(); ++i)
                if (pieces[f_idx[d][i]].len != 0) { any_left = true; break; }
        if (progressed) {
            if (*nrounds_out >= max_rounds) return RDST_ERR_ARG;
            round_starts[++*nrounds_out] = nops;
        }
        if (!any_left) break;
        if (!progressed) return RDST_ERR_ARG;  // cannot happen: the first country of a round with work finds its pieces untouched
    }
    *nops_out = nops;
    return RDST_OK;
}
