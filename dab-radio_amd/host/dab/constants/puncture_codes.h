// dab/constants/puncture_codes.h -- kept-count form of the puncturing vectors (reference: src/dab/constants/puncture_codes.h:42-72).
// Generated from the ETSI EN 300 401 table 13 rule instead of a literal table: PI_n keeps 8+n of every 32 mother bits, the
// e-th extra bit going to 4-bit group bitrev3(e mod 8).
#pragma once
#include <assert.h>
#include <stdint.h>
#include "utility/span.h"

struct DabPunctureTable {
    uint8_t pi[24][8];
    DabPunctureTable() {
        static const int order[8] = {0, 4, 2, 6, 1, 5, 3, 7};
        for (int n = 1; n <= 24; n++) {
            for (int g = 0; g < 8; g++) pi[n - 1][g] = 1;
            for (int e = 0; e < n; e++) pi[n - 1][order[e % 8]] += 1;
        }
    }
};
inline const DabPunctureTable& dab_puncture_table() { static const DabPunctureTable t; return t; }

static const uint8_t PI_X[6] = {2, 2, 2, 2, 2, 2};

static inline tcb::span<const uint8_t> GetPunctureCode(const int x) {
    assert(x >= 1 && x <= 24);
    return tcb::span<const uint8_t>(dab_puncture_table().pi[x - 1], 8);
}
