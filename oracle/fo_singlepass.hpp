// ORACLE — test infrastructure only (see fo_common.hpp).
// Restates `Decomposer::SinglePass`:
//   fiksi/src/analyze/graph/equations.rs:147-221 (ExpressionGraph, find_strongly_connected_expressions)
//   :223-289 (MaskedExpressionGraph), :293-404 (Hopcroft-Karp maximum matching), :406-445
//   (MatchedBipartiteGraph), :447-550 (Tarjan-Pearce strongly connected components)
//   fiksi/src/assemble/mod.rs:169-210 (the SinglePass arm of assemble::solve).
// The reference keeps the matching in IndexMaps (insertion-ordered); the order in which expressions
// first enter the matching is the vertex order of the SCC search, which is reproduced here. The one
// unordered container is the HashSet of an SCC's free variables (equations.rs:198-216, randomly
// seeded hashbrown): any column order is a valid outcome of the reference; ascending is used.
#pragma once
#include <algorithm>
#include <cstdint>
#include <deque>
#include <vector>

#include "fo_assemble.hpp"
#include "fo_expressions.hpp"
#include "fo_lm.hpp"

namespace fo {

// equations.rs:147-181
struct ExpressionGraph {
    std::vector<std::vector<uint32_t>> variables;    // variable -> expressions (duplicates kept)
    std::vector<std::vector<uint32_t>> expressions;  // expression -> variables (variable_indices order)

    static ExpressionGraph build(size_t nvars, const std::vector<Expression>& exprs) {
        ExpressionGraph g;
        g.variables.resize(nvars);  // insert_variables::<N>() per element, lib.rs:386-389
        for (size_t e = 0; e < exprs.size(); ++e) {  // insert_expression, lib.rs:429-441
            uint32_t idx[8];
            int k = variable_indices(exprs[e], idx);
            g.expressions.emplace_back(idx, idx + k);
            for (int i = 0; i < k; ++i) g.variables[idx[i]].push_back(static_cast<uint32_t>(e));
        }
        return g;
    }
};

struct StronglyConnectedExpressions {  // equations.rs:159-162
    std::vector<uint32_t> free_variables;
    std::vector<uint32_t> expressions;
};

namespace sp_detail {

constexpr uint32_t UNMATCHED = 0xFFFFFFFFu;
constexpr uint32_t INF = 0xFFFFFFFFu;

struct Matching {                    // equations.rs:98-139
    std::vector<uint32_t> a_to_b;    // variable -> expression or UNMATCHED
    std::vector<uint32_t> b_to_a;    // expression -> variable or UNMATCHED
    std::vector<uint32_t> b_order;   // expressions in order of first insertion (IndexMap key order)
    void insert(uint32_t a, uint32_t b) {
        a_to_b[a] = b;
        if (b_to_a[b] == UNMATCHED) b_order.push_back(b);
        b_to_a[b] = a;
    }
};

struct Masked {  // equations.rs:223-289
    const ExpressionGraph& g;
    const std::vector<uint32_t>& free_sorted;  // ascending
    const std::vector<uint8_t>& is_free;
};

// equations.rs:326-370
inline bool bfs(const Masked& m, const Matching& mt, std::vector<uint32_t>& distance, uint32_t& dummy) {
    std::deque<uint32_t> queue;
    for (uint32_t a : m.free_sorted) {
        if (mt.a_to_b[a] != UNMATCHED) {
            distance[a] = INF;
        } else {
            distance[a] = 0;
            queue.push_back(a);
        }
    }
    dummy = INF;
    while (!queue.empty()) {
        uint32_t a = queue.front();
        queue.pop_front();
        uint32_t ad = distance[a];
        if (ad >= dummy) continue;
        uint32_t nd = ad == INF ? INF : ad + 1;  // saturating_add
        for (uint32_t b : m.g.variables[a]) {
            uint32_t ma = mt.b_to_a[b];
            if (ma == UNMATCHED) {
                if (dummy == INF) dummy = nd;
            } else if (distance[ma] == INF) {
                distance[ma] = nd;
                queue.push_back(ma);
            }
        }
    }
    return dummy != INF;
}

// equations.rs:372-403
inline bool dfs(const Masked& m, Matching& mt, std::vector<uint32_t>& distance, uint32_t dummy, uint32_t a) {
    uint32_t adp1 = distance[a] == INF ? INF : distance[a] + 1;
    for (uint32_t b : m.g.variables[a]) {
        uint32_t ma = mt.b_to_a[b];
        if (ma == UNMATCHED) {
            if (dummy == adp1) {
                mt.insert(a, b);
                return true;
            }
        } else if (distance[ma] == adp1 && dfs(m, mt, distance, dummy, ma)) {
            mt.insert(a, b);
            return true;
        }
    }
    distance[a] = INF;
    return false;
}

// equations.rs:423-445: neighbours of a matched expression in the directed interpretation
inline void neighbors(const Masked& m, const Matching& mt, uint32_t vertex, std::vector<uint32_t>& out) {
    out.clear();
    uint32_t matched_a = mt.b_to_a[vertex];
    for (uint32_t a : m.g.expressions[vertex]) {
        if (!m.is_free[a]) continue;  // neighbors_of_b filters by the free set (:279-287)
        if (!(a == matched_a || mt.a_to_b[a] == UNMATCHED)) continue;
        for (uint32_t b : m.g.variables[a]) {
            if (b != vertex && mt.b_to_a[b] != UNMATCHED) out.push_back(b);
        }
    }
}

struct Tarjan {  // equations.rs:465-549 (Pearce's Algorithm 3)
    const Masked& m;
    const Matching& mt;
    uint32_t index = 1, c = 0;
    std::vector<uint32_t> root_index;  // 0 = unvisited
    std::vector<uint32_t> stack;
    std::vector<std::vector<uint32_t>> sccs;

    void visit(uint32_t vertex) {
        bool root = true;
        uint32_t vertex_index = index;
        root_index[vertex] = vertex_index;
        index += 1;
        std::vector<uint32_t> nb;
        neighbors(m, mt, vertex, nb);
        for (uint32_t neighbor : nb) {
            if (root_index[neighbor] == 0) visit(neighbor);
            uint32_t ni = root_index[neighbor];
            if (ni < vertex_index) {
                vertex_index = ni;
                root_index[vertex] = vertex_index;
                root = false;
            }
        }
        if (root) {
            std::vector<uint32_t> scc{vertex};
            index -= 1;
            while (!stack.empty()) {
                uint32_t top = stack.back();
                if (vertex_index > root_index[top]) break;
                stack.pop_back();
                scc.push_back(top);
                root_index[top] = c;
                index -= 1;
            }
            root_index[vertex] = c;
            c -= 1;  // wrapping, as in the reference
            sccs.push_back(std::move(scc));
        } else {
            stack.push_back(vertex);
        }
    }
};

}  // namespace sp_detail

// equations.rs:293-320 (`find_maximum_matching`): variable -> expression (UNMATCHED = 0xFFFFFFFF).
inline std::vector<uint32_t> find_maximum_matching(const ExpressionGraph& g, const std::vector<uint32_t>& free_sorted) {
    using namespace sp_detail;
    std::vector<uint8_t> is_free(g.variables.size(), 0);
    for (uint32_t v : free_sorted) is_free[v] = 1;
    Masked m{g, free_sorted, is_free};
    Matching mt;
    mt.a_to_b.assign(g.variables.size(), UNMATCHED);
    mt.b_to_a.assign(g.expressions.size(), UNMATCHED);
    std::vector<uint32_t> distance(g.variables.size(), INF);
    uint32_t dummy = INF;
    while (bfs(m, mt, distance, dummy)) {
        for (uint32_t a : free_sorted) {
            if (mt.a_to_b[a] == UNMATCHED) dfs(m, mt, distance, dummy, a);
        }
    }
    return mt.a_to_b;
}

// equations.rs:186-221
inline std::vector<StronglyConnectedExpressions> find_strongly_connected_expressions(
    const ExpressionGraph& g, const std::vector<uint32_t>& free_sorted) {
    using namespace sp_detail;
    std::vector<uint8_t> is_free(g.variables.size(), 0);
    for (uint32_t v : free_sorted) is_free[v] = 1;
    Masked m{g, free_sorted, is_free};
    Matching mt;
    mt.a_to_b.assign(g.variables.size(), UNMATCHED);
    mt.b_to_a.assign(g.expressions.size(), UNMATCHED);
    std::vector<uint32_t> distance(g.variables.size(), INF);
    uint32_t dummy = INF;
    while (bfs(m, mt, distance, dummy)) {  // equations.rs:304-320
        for (uint32_t a : free_sorted) {
            if (mt.a_to_b[a] == UNMATCHED) dfs(m, mt, distance, dummy, a);
        }
    }
    Tarjan t{m, mt, 1, 0, {}, {}, {}};
    t.root_index.assign(g.expressions.size(), 0);
    // `c` starts at len_vertices - 1; every root_index value stored is either a visit index
    // (>= 1, < len + 1) or a component number counted down from there. To keep "0 = unvisited"
    // unambiguous the component numbers are offset by len + 1 (only comparisons matter).
    uint32_t nverts = static_cast<uint32_t>(mt.b_order.size());
    t.c = nverts + nverts + 1;  // larger than any visit index; decremented per SCC
    for (uint32_t vertex : mt.b_order) {
        if (t.root_index[vertex] == 0) t.visit(vertex);
    }
    std::vector<StronglyConnectedExpressions> out;
    for (size_t k = t.sccs.size(); k-- > 0;) {  // .rev()
        StronglyConnectedExpressions s;
        s.expressions = t.sccs[k];
        for (uint32_t e : s.expressions) {
            uint32_t mv = mt.b_to_a[e];
            for (uint32_t v : g.expressions[e]) {
                if (v == mv || (mt.a_to_b[v] == UNMATCHED && is_free[v])) s.free_variables.push_back(v);
            }
        }
        std::sort(s.free_variables.begin(), s.free_variables.end());
        s.free_variables.erase(std::unique(s.free_variables.begin(), s.free_variables.end()), s.free_variables.end());
        out.push_back(std::move(s));
    }
    return out;
}

// assemble/mod.rs:46-124 + the SinglePass arm :169-210.
inline SolveStats solve_single_pass(FlatSystem& s, bool perturb, QrOrdering ordering = QrOrdering::Colamd,
                                    uint32_t trial_cap = 0, int optimizer = 0) {
    SolveStats out;
    Rng rng = Rng::from_seed(42);
    double system_scale = calculate_system_scale(s);
    out.scale = system_scale;
    double system_scale_recip = 1. / system_scale;
    std::vector<double> variables_transformed(s.variables.size());
    for (size_t i = 0; i < s.variables.size(); ++i) variables_transformed[i] = s.variables[i] * system_scale_recip;
    std::vector<Expression> expressions_transformed;
    for (const Expression& e : s.expressions) expressions_transformed.push_back(transform(e, system_scale_recip));
    ExpressionGraph graph = ExpressionGraph::build(s.variables.size(), s.expressions);

    for (const Component& component : s.components) {
        if (component.variables.empty()) continue;
        std::vector<uint32_t> free_variables;
        for (uint32_t v : component.variables)
            if (!s.fixed[v]) free_variables.push_back(v);
        if (perturb) {
            for (uint32_t fv : free_variables) {
                double& variable = variables_transformed[fv];
                double a = rng.next_f64();
                double b = rng.next_f64();
                variable += variable * (1. / 8196.) * a + (1. / 65568.) * b;
            }
        }
        LmStats total;
        total.exit = LM_EXIT_SSE;
        for (const StronglyConnectedExpressions& scc : find_strongly_connected_expressions(graph, free_variables)) {
            std::vector<double> free_values(scc.free_variables.size());
            for (size_t k = 0; k < free_values.size(); ++k) free_values[k] = variables_transformed[scc.free_variables[k]];
            Subsystem subsystem;
            subsystem.system_variables = variables_transformed.data();
            subsystem.all_expressions = expressions_transformed.data();
            subsystem.expressions = scc.expressions;
            subsystem.free_variables = scc.free_variables;
            subsystem.free_index.assign(s.variables.size(), -1);
            for (size_t k = 0; k < scc.free_variables.size(); ++k) subsystem.free_index[scc.free_variables[k]] = static_cast<int32_t>(k);
            LmStats st = run_optimizer(optimizer, subsystem, free_values.data(), ordering, trial_cap);
            total.accepted += st.accepted;
            total.trials += st.trials;
            total.exit = st.exit;
            total.sse_initial += st.sse_initial;
            total.sse_final += st.sse_final;
            for (size_t k = 0; k < scc.free_variables.size(); ++k) {  // :201-207: both vectors are updated
                variables_transformed[scc.free_variables[k]] = free_values[k];
                s.variables[scc.free_variables[k]] = system_scale * free_values[k];
            }
        }
        out.components.push_back(total);
    }
    return out;
}

}  // namespace fo
