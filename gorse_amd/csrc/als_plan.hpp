// als_plan.hpp -- the thresholds of the ALS row plan (csrc/als.hip als_build_plan), shared with the CPU test of the host library
// (gh_test_als_long_row).  Pure host C++.  What is being planned: the rows of one half-sweep of eALS (model/cf/model.go:659-738), each
// solved independently -- short rows by one wave each, long rows cut into chunks whose partial Gram matrices are added up in a fixed order.
#pragma once
#include <cstdint>

namespace gorse {

// Rows with more entries than this are cut into chunks.  A short row is ONE wave's work from its first entry to its solve, so the
// longest short row is the row kernel's critical path: 4096 entries are 70-90 us of one wave, which a side of 50M entries never notices
// and a side of 1M entries (S-ml1m: the reference's own test shape) waits for with the chip empty -- its epoch at nFactors 8 takes
// 0.196 ms with the threshold at 4096, 0.125 at 1024, 0.095 at 256 (profiles/r05_zq_probe_als_plan.txt).  The threshold therefore follows
// the side's size: the even share of one of ~4096 wave slots, as a power of two between 256 (512 from nFactors 64 on: a chunk's partial
// Gram is d x d floats) and 4096.  It is taken from the WHOLE side, never from a rank's row range: every rank of a sharded sweep cuts the
// same rows the same way (the sharded epoch stays bit-equal to the unsharded one).
inline int64_t als_long_row_threshold(int64_t side_entries, int d) {
    const int64_t share = side_entries / 4096;
    int64_t long_row = d >= 64 ? 512 : 256;
    while (long_row < 4096 && long_row < share) long_row *= 2;
    return long_row;
}

}  // namespace gorse
