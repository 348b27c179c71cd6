// topk_sym.hpp -- the tile schedule of the symmetric all-pairs sweep (topk_mfma.hip, topk_sweep_kernel<..., SYM>), usable on
// host and device: the kernel includes it, and so does the host library's test hook (gorse_amd/host/gorse_host_capi.cpp:
// gh_test_topk_sym_cover), so that "every (query, row) pair is covered exactly once" is checked without a GPU
// (tests/test_topk_sym_schedule_cpu.py).
//
// The queries are the stored rows q0 .. q0 + nq (q0 a multiple of the tile height); query block C = the queries
// [C * bq, (C + 1) * bq), bq a multiple of the tile height; its workgroup multiplies every row tile EXCEPT the whole tiles of later
// query blocks, and reads the tiles of the EARLIER query blocks along their rows too (what it finds there are candidates of those
// rows' queries among its own columns: the foreign lists).  A column emits only if its own tile holds query rows only -- a tile
// that also holds rows behind the query range is multiplied by every workgroup itself.
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GORSE_SYM_HD __host__ __device__ inline
#else
#define GORSE_SYM_HD inline
#endif

namespace gorse {

struct SymSchedule {
    int64_t skip_lo, skip_n;  // the tiles [skip_lo, skip_lo + skip_n) are left to later blocks' workgroups
    int64_t tr_lo, tr_hi;     // the tiles [tr_lo, tr_hi) are read along their rows as well
    int64_t q_full_end;       // tiles below this one (and from q0 / tile_rows on) hold query rows only
    // N rows in tiles of tile_rows; block = the query block of this workgroup
    GORSE_SYM_HD SymSchedule(int64_t q0, int64_t nq, int tile_rows, int bq, int64_t block) {
        const int64_t qt0 = q0 / tile_rows;
        const int64_t tpb = bq / tile_rows;
        q_full_end = (q0 + nq) / tile_rows;
        skip_lo = qt0 + (block + 1) * tpb;
        skip_n = q_full_end > skip_lo ? q_full_end - skip_lo : 0;
        tr_lo = qt0;
        tr_hi = qt0 + block * tpb;
    }
    GORSE_SYM_HD int64_t tiles(int64_t all_tiles) const { return all_tiles - skip_n; }
    GORSE_SYM_HD int64_t tile_index(int64_t tl) const { return tl < skip_lo ? tl : tl + skip_n; }  // the tl-th tile multiplied
    GORSE_SYM_HD bool transposed(int64_t tile) const { return tile >= tr_lo && tile < tr_hi; }
    // does the column of query q (row q0 + q) emit foreign candidates?
    GORSE_SYM_HD static bool column_emits(int64_t q0, int64_t nq, int tile_rows, int64_t q) {
        return q < nq && (q0 + q) / tile_rows < (q0 + nq) / tile_rows;
    }
};

}  // namespace gorse
