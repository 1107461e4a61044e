// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement (C++17, f64) of the reference's hot path. Nothing under fiksi_amd/ may include,
// link, load or call anything under oracle/. Only tests/, __graft_entry__.smoke() and bench.py's
// `cpu_baseline` leg use it, and only as the checker / reported baseline.
//
// Parity pinning: the oracle is pinned by the reference's own known-answer tests (see
// tests/test_oracle_golden.py): LCG sequence (fiksi/src/rand.rs:49-63), the three sparse-QR KATs
// (solvi/src/decomposition/sparse/qr.rs:376-652), the three symbolic KATs
// (solvi/src/decomposition/sparse/cholesky.rs:602-796), the upper-triangular solve
// (solvi/src/sparse_col_mat.rs:836-869), the colamd permutations (colamd_rs/src/lib.rs:253-321),
// the finite-difference property tests (fiksi/src/constraints/expressions.rs:1196-1509) and the
// end-to-end thresholds of fiksi/src/tests/*.rs. The reference is Rust; no Rust toolchain exists
// in the build image, so oracle/_ref cannot be built.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace fo {

// Rust's `usize::MAX`, used by solvi as "none".
constexpr size_t NONE = static_cast<size_t>(-1);

}  // namespace fo
