// ORACLE — test infrastructure only (see fo_common.hpp).
// Restates fiksi/src/rand.rs:9-40.
#pragma once
#include <cstdint>

namespace fo {

// 32-bit LCG, Numerical Recipes constants (rand.rs:24-30).
struct Rng {
    uint32_t state;

    // rand.rs:18-20
    static Rng from_seed(uint32_t seed) { return Rng{seed}; }

    // rand.rs:24-30: state = state * A + C (wrapping)
    uint32_t next_u32() {
        state = state * 1664525u + 1013904223u;
        return state;
    }

    // rand.rs:36-39: (1 / u32::MAX as f64) * val as f64
    double next_f64() {
        uint32_t val = next_u32();
        return (1.0 / 4294967295.0) * static_cast<double>(val);
    }
};

}  // namespace fo
