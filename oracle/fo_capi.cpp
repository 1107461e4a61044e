// ORACLE — TEST INFRASTRUCTURE ONLY (see fo_common.hpp). extern "C" surface over the CPU
// restatement so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can drive it
// through ctypes. Never linked into, loaded by, or called from the product library.
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "fo_analyze.hpp"
#include "fo_assemble.hpp"
#include "fo_colamd.hpp"
#include "fo_expressions.hpp"
#include "fo_lm.hpp"
#include "fo_qr.hpp"
#include "fo_recursive.hpp"
#include "fo_singlepass.hpp"
#include "fo_rand.hpp"
#include "fo_sparse.hpp"
#include "fo_symbolic.hpp"

using namespace fo;

namespace {

SparseColMatStructure make_structure(int64_t nrows, int64_t ncols, const int64_t* colptr, const int64_t* rowidx) {
    SparseColMatStructure s;
    s.nrows = static_cast<size_t>(nrows);
    s.ncols = static_cast<size_t>(ncols);
    s.column_pointers.assign(colptr, colptr + ncols + 1);
    s.row_indices.assign(rowidx, rowidx + colptr[ncols]);
    return s;
}

constexpr uint16_t NO_COMPONENT = 0xFFFF;

// Build the oracle's FlatSystem for system `s` of a flat batch (same array layout as
// include/fiksi_amd.h's fx_batch, restated here so the oracle stays self-contained).
FlatSystem make_system(uint32_t s, const uint32_t* var_off, const uint32_t* expr_off, const double* vars,
                       const uint8_t* var_fixed, const uint8_t* expr_tag, const uint32_t* expr_idx,
                       const double* expr_param, const uint16_t* var_comp, const uint16_t* expr_comp) {
    FlatSystem sys;
    uint32_t v0 = var_off[s], v1 = var_off[s + 1], e0 = expr_off[s], e1 = expr_off[s + 1];
    sys.variables.assign(vars + v0, vars + v1);
    sys.fixed.assign(var_fixed + v0, var_fixed + v1);
    sys.expressions.resize(e1 - e0);
    for (uint32_t e = e0; e < e1; ++e) {
        Expression& x = sys.expressions[e - e0];
        x.tag = expr_tag[e];
        for (int k = 0; k < 4; ++k) x.idx[k] = expr_idx[4 * static_cast<size_t>(e) + k];
        x.param = expr_param[e];
    }
    uint32_t ncomp = 0;
    for (uint32_t v = v0; v < v1; ++v) {
        uint16_t c = var_comp ? var_comp[v] : 0;
        if (c != NO_COMPONENT && c + 1u > ncomp) ncomp = c + 1u;
    }
    for (uint32_t e = e0; e < e1; ++e) {
        uint16_t c = expr_comp ? expr_comp[e] : 0;
        if (c != NO_COMPONENT && c + 1u > ncomp) ncomp = c + 1u;
    }
    sys.components.resize(ncomp);
    for (uint32_t v = v0; v < v1; ++v) {
        uint16_t c = var_comp ? var_comp[v] : 0;
        if (c != NO_COMPONENT) sys.components[c].variables.push_back(v - v0);
    }
    for (uint32_t e = e0; e < e1; ++e) {
        uint16_t c = expr_comp ? expr_comp[e] : 0;
        if (c != NO_COMPONENT) sys.components[c].expressions.push_back(e - e0);
    }
    return sys;
}

template <typename F>
void parallel_for(uint32_t n, uint32_t nthreads, F&& f) {
    if (nthreads <= 1 || n < 2) {
        for (uint32_t i = 0; i < n; ++i) f(i);
        return;
    }
    std::atomic<uint32_t> next{0};
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < nthreads; ++t) {
        pool.emplace_back([&] {
            for (;;) {
                uint32_t begin = next.fetch_add(64);
                if (begin >= n) return;
                uint32_t end = begin + 64 < n ? begin + 64 : n;
                for (uint32_t i = begin; i < end; ++i) f(i);
            }
        });
    }
    for (auto& th : pool) th.join();
}

}  // namespace

extern "C" {

struct fo_result {
    uint32_t accepted;  // accepted LM steps (Gauss-Newton iterations), summed over components
    uint32_t trials;    // factorizations, summed over components
    uint32_t exit;      // LmExit of the last component
    uint32_t ncomp;     // components solved
    double scale;       // system scale (1 when scaling is off)
    double sse0;        // initial SSE (scaled space), summed over components
    double sse;         // final SSE (scaled space), summed over components
};

// 0: the platform libm's atan2 (the reference on this platform); 1: the correctly rounded atan2 (fo_expressions.hpp)
void fo_set_atan2_mode(int mode) { fo::detail::atan2_mode() = mode ? 1 : 0; }
int fo_get_atan2_mode(void) { return fo::detail::atan2_mode(); }
void fo_atan2_batch(uint64_t n, const double* y, const double* x, double* out) {
    for (uint64_t i = 0; i < n; ++i) out[i] = fo::detail::vatan2(fo::detail::V2{x[i], y[i]});
}

// fiksi/src/rand.rs
void fo_rng_u32(uint32_t seed, uint32_t n, uint32_t* out) {
    Rng rng = Rng::from_seed(seed);
    for (uint32_t i = 0; i < n; ++i) out[i] = rng.next_u32();
}
void fo_rng_f64(uint32_t seed, uint32_t n, double* out) {
    Rng rng = Rng::from_seed(seed);
    for (uint32_t i = 0; i < n; ++i) out[i] = rng.next_f64();
}

// expressions.rs: one expression on gathered values. Returns the variable count.
int fo_expr_eval(uint8_t tag, const double* vars8, double param, double* residual, double* grad8) {
    Expression e{tag, {0, 2, 4, 6}, param};
    uint32_t idx[8];
    int k = variable_indices(e, idx);
    double g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = vars8[i];
    *residual = compute_residual_and_gradient(e, v, g);
    for (int i = 0; i < 8; ++i) grad8[i] = g[i];
    return k;
}

// colamd_rs::colamd. `colptr` (ncols+1) is overwritten with the permutation in [0, ncols).
int fo_colamd(int nrows, int ncols, const int* rowidx, int* colptr, int aggressive) {
    std::vector<int> p(colptr, colptr + ncols + 1);
    std::vector<int> rows(rowidx, rowidx + p[ncols]);
    ColamdOptions opt;
    opt.aggressive_row_absorption = aggressive != 0;
    if (!colamd(nrows, ncols, rows, p, opt)) return -1;
    for (int i = 0; i < ncols; ++i) colptr[i] = p[i];
    colptr[ncols] = -1;
    return 0;
}

// solvi symbolic phase on a CSC structure (natural ordering): etree, post-order, counts, R structure.
// l_rowidx must have room for sum(col_counts) entries (<= ncols*(ncols+1)/2).
int fo_symbolic(int64_t nrows, int64_t ncols, const int64_t* colptr, const int64_t* rowidx, int64_t* parents,
                int64_t* post, int64_t* row_counts, int64_t* col_counts, int64_t* l_colptr, int64_t* l_rowidx) {
    SparseColMatStructure a = make_structure(nrows, ncols, colptr, rowidx);
    auto par = elimination_tree<false>(a);
    auto po = post_order(par);
    auto counts = CholeskyCounts::build(a, par, po);
    auto cs = CholeskyStructure::build(a, par, po, counts);
    for (int64_t j = 0; j < ncols; ++j) {
        parents[j] = par[j] == NONE ? -1 : static_cast<int64_t>(par[j]);
        post[j] = static_cast<int64_t>(po[j]);
        row_counts[j] = static_cast<int64_t>(counts.row_counts[j]);
        col_counts[j] = static_cast<int64_t>(counts.col_counts[j]);
    }
    for (int64_t j = 0; j <= ncols; ++j) l_colptr[j] = static_cast<int64_t>(cs.l_structure.column_pointers[j]);
    for (size_t k = 0; k < cs.l_structure.row_indices.size(); ++k) l_rowidx[k] = static_cast<int64_t>(cs.l_structure.row_indices[k]);
    return 0;
}

// solvi sparse QR: factorize the CSC matrix and solve min |A x - b| in place (b has nrows entries,
// x is returned in b[0..ncols)). ordering: 0 natural, 1 colamd. R is returned in CSC form.
// SymbolicQr::build (qr.rs:118-206): permutations and the structures of H and R.
int fo_symbolic_qr(int64_t nrows, int64_t ncols, const int64_t* colptr, const int64_t* rowidx, int colamd_ordering,
                   int64_t* col_perm, int64_t* row_perm, int64_t* h_ptr, int64_t* h_rows, int64_t* r_ptr, int64_t* r_rows) {
    SparseColMatStructure a = make_structure(nrows, ncols, colptr, rowidx);
    SymbolicQr s = SymbolicQr::build(a, colamd_ordering ? QrOrdering::Colamd : QrOrdering::Natural);
    for (int64_t j = 0; j < ncols; ++j) col_perm[j] = static_cast<int64_t>(s.col_permutation[j]);
    for (int64_t i = 0; i < nrows; ++i) row_perm[i] = static_cast<int64_t>(s.row_permutation[i]);
    for (int64_t j = 0; j <= ncols; ++j) {
        h_ptr[j] = static_cast<int64_t>(s.h_structure.column_pointers[j]);
        r_ptr[j] = static_cast<int64_t>(s.r_structure.column_pointers[j]);
    }
    for (size_t k = 0; k < s.h_structure.row_indices.size(); ++k) h_rows[k] = static_cast<int64_t>(s.h_structure.row_indices[k]);
    for (size_t k = 0; k < s.r_structure.row_indices.size(); ++k) r_rows[k] = static_cast<int64_t>(s.r_structure.row_indices[k]);
    return 0;
}

int fo_qr_factor_solve(int64_t nrows, int64_t ncols, const int64_t* colptr, const int64_t* rowidx,
                       const double* values, int ordering, double* b, int64_t* r_colptr, int64_t* r_rowidx,
                       double* r_values, int64_t r_cap, int* solved) {
    SparseColMat a;
    a.structure = make_structure(nrows, ncols, colptr, rowidx);
    a.values.assign(values, values + colptr[ncols]);
    SymbolicQr sym = SymbolicQr::build(a.structure, ordering ? QrOrdering::Colamd : QrOrdering::Natural);
    Qr qr(sym);
    qr.factorize(a);
    if (b) *solved = qr.solve_mut(b) ? 1 : 0;
    int64_t nnz = static_cast<int64_t>(qr.r.values.size());
    if (r_colptr) {
        if (nnz > r_cap) return -1;
        for (int64_t j = 0; j <= ncols; ++j) r_colptr[j] = static_cast<int64_t>(qr.r.structure.column_pointers[j]);
        for (int64_t k = 0; k < nnz; ++k) {
            r_rowidx[k] = static_cast<int64_t>(qr.r.structure.row_indices[k]);
            r_values[k] = qr.r.values[k];
        }
    }
    return 0;
}

// TripletMat -> SparseColMat::from_triplet_mat (sorted, duplicates summed). Returns nnz.
int64_t fo_from_triplets(int64_t nrows, int64_t ncols, int64_t ntrip, const int64_t* rows, const int64_t* cols,
                         const double* vals, int64_t* colptr, int64_t* rowidx, double* values) {
    TripletMat t(static_cast<size_t>(nrows), static_cast<size_t>(ncols));
    for (int64_t k = 0; k < ntrip; ++k) t.push_triplet(static_cast<size_t>(rows[k]), static_cast<size_t>(cols[k]), vals[k]);
    SparseColMat m = SparseColMat::from_triplet_mat(t);
    for (size_t j = 0; j <= m.ncols(); ++j) colptr[j] = static_cast<int64_t>(m.structure.column_pointers[j]);
    for (size_t k = 0; k < m.values.size(); ++k) {
        rowidx[k] = static_cast<int64_t>(m.structure.row_indices[k]);
        values[k] = m.values[k];
    }
    return static_cast<int64_t>(m.values.size());
}

// SparseColMat::solve_upper_triangular_mut
int fo_solve_upper(int64_t n, const int64_t* colptr, const int64_t* rowidx, const double* values, double* b) {
    SparseColMat a;
    a.structure = make_structure(n, n, colptr, rowidx);
    a.values.assign(values, values + colptr[n]);
    return a.solve_upper_triangular_mut(b) ? 1 : 0;
}

// Subsystem::calculate_residuals_and_sparse_jacobian (subsystem.rs:126-166) for every system of
// a flat batch at the given variable values, followed by the reference's COO -> CSC conversion
// (duplicates summed) and a transposition to the row-major layout the GPU path emits:
//   r[expr_off[s] + row], jrow_ptr[total_exprs + 1] (global rows), jcol (system-local free
//   column), jval. Columns = ascending rank of non-fixed variables that belong to a component.
// Returns total nnz, or -1 if `cap` is too small.
int64_t fo_eval_batch(uint32_t n_systems, const uint32_t* var_off, const uint32_t* expr_off, const double* vars,
                      const uint8_t* var_fixed, const uint8_t* expr_tag, const uint32_t* expr_idx,
                      const double* expr_param, const uint16_t* var_comp, double* r, int64_t* jrow_ptr,
                      int32_t* jcol, double* jval, int64_t cap) {
    int64_t nnz_total = 0;
    jrow_ptr[0] = 0;
    for (uint32_t s = 0; s < n_systems; ++s) {
        FlatSystem sys = make_system(s, var_off, expr_off, vars, var_fixed, expr_tag, expr_idx, expr_param, var_comp, nullptr);
        Subsystem sub;
        sub.system_variables = sys.variables.data();
        sub.all_expressions = sys.expressions.data();
        sub.free_index.assign(sys.variables.size(), -1);
        for (uint32_t v = 0; v < sys.variables.size(); ++v) {
            uint16_t c = var_comp ? var_comp[var_off[s] + v] : 0;
            if (!sys.fixed[v] && c != NO_COMPONENT) {
                sub.free_index[v] = static_cast<int32_t>(sub.free_variables.size());
                sub.free_variables.push_back(v);
            }
        }
        for (uint32_t e = 0; e < sys.expressions.size(); ++e) sub.expressions.push_back(e);
        std::vector<double> x(sub.free_variables.size());
        for (size_t k = 0; k < x.size(); ++k) x[k] = sys.variables[sub.free_variables[k]];
        size_t m = sub.expressions.size(), nv = x.size();
        TripletMat coo(m, nv);
        sub.calculate_residuals_and_sparse_jacobian(x.data(), r + expr_off[s], coo);
        coo.nrows = m;
        coo.ncols = nv;
        SparseColMat csc = SparseColMat::from_triplet_mat(coo);
        // CSC -> CSR (entries of a row end up in ascending column order)
        std::vector<int64_t> count(m, 0);
        for (size_t row : csc.structure.row_indices) count[row] += 1;
        int64_t base = nnz_total;
        for (size_t row = 0; row < m; ++row) {
            jrow_ptr[expr_off[s] + row + 1] = jrow_ptr[expr_off[s] + row] + count[row];
        }
        nnz_total = jrow_ptr[expr_off[s] + m];
        if (nnz_total > cap) return -1;
        std::vector<int64_t> fill(m, 0);
        for (size_t col = 0; col < nv; ++col) {
            for (size_t p = csc.structure.column_pointers[col]; p < csc.structure.column_pointers[col + 1]; ++p) {
                size_t row = csc.structure.row_indices[p];
                int64_t dst = jrow_ptr[expr_off[s] + row] + fill[row]++;
                jcol[dst] = static_cast<int32_t>(col);
                jval[dst] = csc.values[p];
            }
        }
        (void)base;
    }
    return nnz_total;
}

// assemble::solve (Decomposer::None + LM) over a flat batch, `nthreads` worker threads over
// disjoint system ranges (the reference itself is single-threaded; systems are independent).
// mode bit0: scale by the system RMS (assemble/mod.rs:58-79); bit1: LCG perturbation (:113-124).
// mode 0 == levenberg_marquardt(Subsystem) on the values as given (L2 boundary).
// first_delta (optional, mode 0 only): per system, receives the first LM step of component 0 at
// var-offset positions of the free variables (others untouched).
int fo_solve_batch(uint32_t n_systems, const uint32_t* var_off, const uint32_t* expr_off, double* vars,
                   const uint8_t* var_fixed, const uint8_t* expr_tag, const uint32_t* expr_idx,
                   const double* expr_param, const uint16_t* var_comp, const uint16_t* expr_comp, uint32_t mode,
                   int ordering, uint32_t trial_cap, uint32_t nthreads, fo_result* results) {
    QrOrdering ord = ordering ? QrOrdering::Colamd : QrOrdering::Natural;
    parallel_for(n_systems, nthreads, [&](uint32_t s) {
        FlatSystem sys = make_system(s, var_off, expr_off, vars, var_fixed, expr_tag, expr_idx, expr_param, var_comp, expr_comp);
        fo_result res{};
        if (mode & 1u) {
            SolveStats st = solve(sys, (mode & 2u) != 0, ord, trial_cap, (mode & 4u) ? 1 : 0);  // bit 2: Optimizer::LBfgs
            res.scale = st.scale;
            for (const LmStats& c : st.components) {
                res.accepted += c.accepted;
                res.trials += c.trials;
                res.exit = c.exit;
                res.sse0 += c.sse_initial;
                res.sse += c.sse_final;
                res.ncomp += 1;
            }
        } else {
            // L2: no scaling; optional perturbation is not meaningful here. Each component is
            // solved against the snapshot of the incoming values (quirk Q2).
            res.scale = 1.;
            std::vector<double> snapshot = sys.variables;
            for (const Component& comp : sys.components) {
                if (comp.variables.empty()) continue;
                Subsystem sub;
                sub.system_variables = snapshot.data();
                sub.all_expressions = sys.expressions.data();
                sub.expressions = comp.expressions;
                sub.free_index.assign(sys.variables.size(), -1);
                for (uint32_t v : comp.variables) {
                    if (!sys.fixed[v]) {
                        sub.free_index[v] = static_cast<int32_t>(sub.free_variables.size());
                        sub.free_variables.push_back(v);
                    }
                }
                std::vector<double> x(sub.free_variables.size());
                for (size_t k = 0; k < x.size(); ++k) x[k] = snapshot[sub.free_variables[k]];
                LmStats c = levenberg_marquardt(sub, x.data(), ord, trial_cap);
                for (size_t k = 0; k < x.size(); ++k) sys.variables[sub.free_variables[k]] = x[k];
                res.accepted += c.accepted;
                res.trials += c.trials;
                res.exit = c.exit;
                res.sse0 += c.sse_initial;
                res.sse += c.sse_final;
                res.ncomp += 1;
            }
        }
        std::memcpy(vars + var_off[s], sys.variables.data(), sys.variables.size() * sizeof(double));
        if (results) results[s] = res;
    });
    return 0;
}

// Pose2D::transform_point and gradient_chain_rule_point with the inner gradients [1,0] and [0,1]
// (expressions.rs:1120-1157, as assemble/mod.rs:547-575 calls them). out: x, y, dx[3], dy[3].
void fo_pose_rows(const double* pose3, double u, double v, double* out8) {
    Pose2D pose = Pose2D::from_array(pose3);
    pose.transform_point(u, v, out8[0], out8[1]);
    pose.gradient_chain_rule_point(u, v, 1., 0., out8 + 2);
    pose.gradient_chain_rule_point(u, v, 0., 1., out8 + 5);
}

// assemble::solve with Decomposer::RecursiveAssembly (assemble/mod.rs:212-277) on ONE System given with its
// elements and constraints (fo_recursive.hpp). el_kind: 0 Length, 1 Point, 2 Line, 3 Circle; el_idx: variable
// index of a Length / a Point's x. con_inc holds 6 slots per constraint (con_ninc used). el_comp / con_comp:
// live component of each element / constraint in iteration order, 0xFFFF = none.
// plan_out receives the concatenated serialised plans (fo::serialise_plan) of the components; step_results one
// entry per solved cluster problem. flags: bit0 the reference would panic, bit1 search budget exhausted.
int fo_solve_recursive(uint32_t nvars, double* vars, const uint8_t* var_fixed, uint32_t nexprs, const uint8_t* expr_tag,
                       const uint32_t* expr_idx, const double* expr_param, uint32_t nel, const uint8_t* el_kind,
                       const uint32_t* el_idx, const uint16_t* el_comp, uint32_t ncon, const uint8_t* con_valency,
                       const uint32_t* con_expr, const uint8_t* con_ninc, const uint32_t* con_inc, const uint16_t* con_comp,
                       int perturb, int ordering, uint32_t trial_cap, uint64_t budget, uint32_t* plan_out, uint32_t plan_cap,
                       uint32_t* plan_len, fo_result* step_results, uint32_t step_cap, uint32_t* n_steps, uint32_t* flags,
                       uint32_t* step_sizes /* 2 per step: unknowns, rows; may be NULL */) {
    GeoSystem g;
    g.variables.assign(vars, vars + nvars);
    g.fixed.assign(var_fixed, var_fixed + nvars);
    g.expressions.resize(nexprs);
    for (uint32_t e = 0; e < nexprs; ++e) {
        g.expressions[e].tag = expr_tag[e];
        for (int k = 0; k < 4; ++k) g.expressions[e].idx[k] = expr_idx[4 * static_cast<size_t>(e) + k];
        g.expressions[e].param = expr_param[e];
    }
    uint32_t ncomp = 0;
    g.elements.resize(nel);
    for (uint32_t i = 0; i < nel; ++i) {
        g.elements[i] = GeoElementInfo{el_kind[i], el_idx[i]};
        if (el_comp[i] != NO_COMPONENT) ncomp = std::max<uint32_t>(ncomp, el_comp[i] + 1u);
    }
    g.constraints.resize(ncon);
    for (uint32_t c = 0; c < ncon; ++c) {
        g.constraints[c].valency = con_valency[c];
        g.constraints[c].expressions_idx = con_expr[c];
        g.constraints[c].incident_elements.assign(con_inc + 6 * static_cast<size_t>(c), con_inc + 6 * static_cast<size_t>(c) + con_ninc[c]);
        if (con_comp[c] != NO_COMPONENT) ncomp = std::max<uint32_t>(ncomp, con_comp[c] + 1u);
    }
    g.components.resize(ncomp);
    for (uint32_t i = 0; i < nel; ++i)
        if (el_comp[i] != NO_COMPONENT) g.components[el_comp[i]].elements.push_back(i);
    for (uint32_t c = 0; c < ncon; ++c)
        if (con_comp[c] != NO_COMPONENT) g.components[con_comp[c]].constraints.push_back(c);

    RecursiveStats st = solve_recursive_assembly(g, perturb != 0, ordering ? QrOrdering::Colamd : QrOrdering::Natural, trial_cap,
                                                 budget ? static_cast<size_t>(budget) : 200000);
    std::memcpy(vars, g.variables.data(), nvars * sizeof(double));
    uint32_t len = 0;
    for (const std::vector<uint32_t>& p : st.plans) {
        for (uint32_t w : p) {
            if (plan_out && len < plan_cap) plan_out[len] = w;
            ++len;
        }
    }
    if (plan_len) *plan_len = len;
    uint32_t ns = 0;
    for (const LmStats& c : st.steps) {
        if (step_results && ns < step_cap) {
            fo_result r{};
            r.accepted = c.accepted;
            r.trials = c.trials;
            r.exit = c.exit;
            r.ncomp = 1;
            r.scale = st.scale;
            r.sse0 = c.sse_initial;
            r.sse = c.sse_final;
            step_results[ns] = r;
            if (step_sizes) {
                step_sizes[2 * ns] = st.sizes[ns].first;
                step_sizes[2 * ns + 1] = st.sizes[ns].second;
            }
        }
        ++ns;
    }
    if (n_steps) *n_steps = ns;
    if (flags) *flags = (st.panicked ? 1u : 0u) | (st.exhausted ? 2u : 0u);
    return 0;
}

// assemble::solve with Decomposer::SinglePass (assemble/mod.rs:169-210): scale, perturb, then one LM
// per strongly connected block of expressions, in topological order, each seeing the previous results.
int fo_solve_single_pass_batch(uint32_t n_systems, const uint32_t* var_off, const uint32_t* expr_off, double* vars,
                               const uint8_t* var_fixed, const uint8_t* expr_tag, const uint32_t* expr_idx,
                               const double* expr_param, const uint16_t* var_comp, const uint16_t* expr_comp,
                               uint32_t perturb, int ordering, uint32_t trial_cap, uint32_t nthreads, fo_result* results) {
    QrOrdering ord = ordering ? QrOrdering::Colamd : QrOrdering::Natural;
    parallel_for(n_systems, nthreads, [&](uint32_t s) {
        FlatSystem sys = make_system(s, var_off, expr_off, vars, var_fixed, expr_tag, expr_idx, expr_param, var_comp, expr_comp);
        SolveStats st = solve_single_pass(sys, (perturb & 1u) != 0, ord, trial_cap, (perturb & 4u) ? 1 : 0);  // bit 2: LBfgs
        fo_result res{};
        res.scale = st.scale;
        for (const LmStats& c : st.components) {
            res.accepted += c.accepted;
            res.trials += c.trials;
            res.exit = c.exit;
            res.sse0 += c.sse_initial;
            res.sse += c.sse_final;
            res.ncomp += 1;
        }
        std::memcpy(vars + var_off[s], sys.variables.data(), sys.variables.size() * sizeof(double));
        if (results) results[s] = res;
    });
    return 0;
}

// The SinglePass decomposition of component `comp` of one system: unit_of_expr[e] = position of the
// block that solves expression e (or -1 when the expression is not matched and therefore skipped),
// and for every block its free variables (ascending) flattened into unit_vars with unit_var_off.
// Returns the number of blocks, or -1 if a capacity is too small.
int fo_single_pass_units(uint32_t nvars, uint32_t nexprs, const uint8_t* var_fixed, const uint8_t* expr_tag,
                         const uint32_t* expr_idx, const uint16_t* var_comp, uint16_t comp, int32_t* unit_of_expr,
                         uint32_t* unit_row_off, uint32_t* unit_rows, uint32_t* unit_var_off, uint32_t* unit_vars,
                         uint32_t cap) {
    std::vector<Expression> ex(nexprs);
    for (uint32_t e = 0; e < nexprs; ++e) {
        ex[e].tag = expr_tag[e];
        for (int k = 0; k < 4; ++k) ex[e].idx[k] = expr_idx[4 * static_cast<size_t>(e) + k];
        ex[e].param = 0.;
    }
    ExpressionGraph g = ExpressionGraph::build(nvars, ex);
    std::vector<uint32_t> free_sorted;
    for (uint32_t v = 0; v < nvars; ++v)
        if ((var_comp ? var_comp[v] : 0) == comp && !var_fixed[v]) free_sorted.push_back(v);
    auto sccs = find_strongly_connected_expressions(g, free_sorted);
    for (uint32_t e = 0; e < nexprs; ++e) unit_of_expr[e] = -1;
    uint32_t nr = 0, nv = 0;
    unit_row_off[0] = 0;
    unit_var_off[0] = 0;
    for (size_t u = 0; u < sccs.size(); ++u) {
        if (u + 1 > cap) return -1;
        for (uint32_t e : sccs[u].expressions) {
            if (nr >= cap) return -1;
            unit_of_expr[e] = static_cast<int32_t>(u);
            unit_rows[nr++] = e;
        }
        for (uint32_t v : sccs[u].free_variables) {
            if (nv >= cap) return -1;
            unit_vars[nv++] = v;
        }
        unit_row_off[u + 1] = nr;
        unit_var_off[u + 1] = nv;
    }
    return static_cast<int>(sccs.size());
}

// equations.rs:293-320 on a raw bipartite graph: expression e reads variables evars[eptr[e] .. eptr[e+1]).
// a_to_b[v] = matched expression or 0xFFFFFFFF. Returns the cardinality.
int fo_maximum_matching(uint32_t nvars, uint32_t nexprs, const uint32_t* eptr, const uint32_t* evars, uint32_t nfree,
                        const uint32_t* free_sorted, uint32_t* a_to_b) {
    ExpressionGraph g;
    g.variables.resize(nvars);
    g.expressions.resize(nexprs);
    for (uint32_t e = 0; e < nexprs; ++e)
        for (uint32_t p = eptr[e]; p < eptr[e + 1]; ++p) {
            g.expressions[e].push_back(evars[p]);
            g.variables[evars[p]].push_back(e);
        }
    std::vector<uint32_t> m = find_maximum_matching(g, std::vector<uint32_t>(free_sorted, free_sorted + nfree));
    int card = 0;
    for (uint32_t v = 0; v < nvars; ++v) {
        a_to_b[v] = m[v];
        card += m[v] != 0xFFFFFFFFu;
    }
    return card;
}

// Problem::calculate_residuals_and_jacobian (subsystem.rs:106-124) with every non-fixed variable of a
// component free (columns = ascending rank, as in the sparse structure) and every expression a row: the
// dense row-major Jacobian of System s at jac[jac_off[s]]; jac_off has n_systems + 1 entries (filled
// here); jac may be NULL to get the offsets only.
int fo_eval_dense_batch(uint32_t n_systems, const uint32_t* var_off, const uint32_t* expr_off, const double* vars,
                        const uint8_t* var_fixed, const uint8_t* expr_tag, const uint32_t* expr_idx,
                        const double* expr_param, const uint16_t* var_comp, double* residuals, double* jac,
                        uint64_t* jac_off) {
    jac_off[0] = 0;
    for (uint32_t s = 0; s < n_systems; ++s) {
        FlatSystem sys = make_system(s, var_off, expr_off, vars, var_fixed, expr_tag, expr_idx, expr_param, var_comp, nullptr);
        Subsystem sub;
        sub.system_variables = sys.variables.data();
        sub.all_expressions = sys.expressions.data();
        sub.free_index.assign(sys.variables.size(), -1);
        for (uint32_t v = 0; v < sys.variables.size(); ++v) {
            uint16_t c = var_comp ? var_comp[var_off[s] + v] : 0;
            if (c != 0xFFFF && !sys.fixed[v]) {
                sub.free_index[v] = static_cast<int32_t>(sub.free_variables.size());
                sub.free_variables.push_back(v);
            }
        }
        for (uint32_t e = 0; e < sys.expressions.size(); ++e) sub.expressions.push_back(e);
        const size_t nv = sub.free_variables.size(), ne = sub.expressions.size();
        jac_off[s + 1] = jac_off[s] + static_cast<uint64_t>(nv) * ne;
        if (!jac) continue;
        std::vector<double> x(nv), r(ne, 0.), j(nv * ne, 0.);
        for (size_t k = 0; k < nv; ++k) x[k] = sys.variables[sub.free_variables[k]];
        lbfgs_detail::residuals_and_dense_jacobian(sub, x.data(), r.data(), j.data());
        std::copy(j.begin(), j.end(), jac + jac_off[s]);
        if (residuals) std::copy(r.begin(), r.end(), residuals + expr_off[s]);
    }
    return 0;
}

// permutation.rs:41-80: applies the gather permutation to `values` in place through the swap
// sequence; returns the number of swaps.
int fo_permute(uint32_t n, const uint32_t* permutation, double* values) {
    std::vector<size_t> p(permutation, permutation + n);
    PermutationSequence seq = PermutationSequence::build_for_gather_permutation(p);
    seq.permute_slice(values);
    return static_cast<int>(seq.swap_sequence.size());
}

// First LM step only (lambda = 0.5) of component 0 for every system, on the values as given:
// delta for the free variables (ascending order) written to delta[var_off[s] + k], k < nfree.
int fo_first_step_batch(uint32_t n_systems, const uint32_t* var_off, const uint32_t* expr_off, const double* vars,
                        const uint8_t* var_fixed, const uint8_t* expr_tag, const uint32_t* expr_idx,
                        const double* expr_param, const uint16_t* var_comp, const uint16_t* expr_comp, int ordering,
                        double* delta) {
    QrOrdering ord = ordering ? QrOrdering::Colamd : QrOrdering::Natural;
    for (uint32_t s = 0; s < n_systems; ++s) {
        FlatSystem sys = make_system(s, var_off, expr_off, vars, var_fixed, expr_tag, expr_idx, expr_param, var_comp, expr_comp);
        if (sys.components.empty()) continue;
        const Component& comp = sys.components[0];
        Subsystem sub;
        sub.system_variables = sys.variables.data();
        sub.all_expressions = sys.expressions.data();
        sub.expressions = comp.expressions;
        sub.free_index.assign(sys.variables.size(), -1);
        for (uint32_t v : comp.variables) {
            if (!sys.fixed[v]) {
                sub.free_index[v] = static_cast<int32_t>(sub.free_variables.size());
                sub.free_variables.push_back(v);
            }
        }
        std::vector<double> x(sub.free_variables.size());
        for (size_t k = 0; k < x.size(); ++k) x[k] = sys.variables[sub.free_variables[k]];
        std::vector<double> d(x.size(), 0.);
        levenberg_marquardt(sub, x.data(), ord, 1, d.data());
        for (size_t k = 0; k < d.size(); ++k) delta[var_off[s] + k] = d[k];
    }
    return 0;
}

// calculate_system_scale (assemble/mod.rs:32-44) per system.
void fo_system_scale_batch(uint32_t n_systems, const uint32_t* var_off, const uint32_t* expr_off, const double* vars,
                           const uint8_t* expr_tag, const double* expr_param, double* scale) {
    for (uint32_t s = 0; s < n_systems; ++s) {
        FlatSystem sys;
        sys.variables.assign(vars + var_off[s], vars + var_off[s + 1]);
        for (uint32_t e = expr_off[s]; e < expr_off[s + 1]; ++e) {
            Expression x{expr_tag[e], {0, 0, 0, 0}, expr_param[e]};
            sys.expressions.push_back(x);
        }
        scale[s] = calculate_system_scale(sys);
    }
}

// Post-solve check: residual of every expression on the (unscaled) variables as given
// (constraints/mod.rs:144-168 for valency-1 constraints).
void fo_residuals_batch(uint32_t n_systems, const uint32_t* var_off, const uint32_t* expr_off, const double* vars,
                        const uint8_t* expr_tag, const uint32_t* expr_idx, const double* expr_param, double* r) {
    for (uint32_t s = 0; s < n_systems; ++s) {
        for (uint32_t e = expr_off[s]; e < expr_off[s + 1]; ++e) {
            Expression x;
            x.tag = expr_tag[e];
            for (int k = 0; k < 4; ++k) x.idx[k] = expr_idx[4 * static_cast<size_t>(e) + k];
            x.param = expr_param[e];
            r[e] = expression_residual(x, vars + var_off[s]);
        }
    }
}

// System::analyze (analyze/numerical/mod.rs:123-163) per system: dependent[e] = 1 for expressions that
// do not increase the rank of the dense Jacobian at the given (unscaled) variables.
void fo_analyze_batch(uint32_t n_systems, const uint32_t* var_off, const uint32_t* expr_off, const double* vars,
                      const uint8_t* expr_tag, const uint32_t* expr_idx, const double* expr_param, uint8_t* dependent) {
    for (uint32_t s = 0; s < n_systems; ++s) {
        uint32_t e0 = expr_off[s], ne = expr_off[s + 1] - e0;
        std::vector<Expression> ex(ne);
        for (uint32_t e = 0; e < ne; ++e) {
            ex[e].tag = expr_tag[e0 + e];
            for (int k = 0; k < 4; ++k) ex[e].idx[k] = expr_idx[4 * static_cast<size_t>(e0 + e) + k];
            ex[e].param = expr_param[e0 + e];
        }
        find_overconstraints(vars + var_off[s], var_off[s + 1] - var_off[s], ex.data(), ne, dependent + e0);
    }
}

}  // extern "C"
