// ORACLE — test infrastructure only (see fo_common.hpp).
// Restates `Decomposer::RecursiveAssembly`:
//   fiksi/src/graph.rs:14-176,227-254          (Graph: degrees of freedom, valencies, incidence lists)
//   fiksi/src/analyze/graph/recursive_assembly.rs:164-480 (decompose::<3>, the Modified Frontier plan)
//   fiksi/src/analyze/graph/recursive_assembly.rs:499-645 (dense_bfs)
//   fiksi/src/assemble/mod.rs:212-277           (the RecursiveAssembly arm of assemble::solve)
//   fiksi/src/assemble/mod.rs:282-725           (ClusteredSystem and its `Problem` impl)
//   fiksi/src/constraints/expressions.rs:1094-1159 (Pose2D)
//
// The reference keeps vertices, edges, subgraphs and frontiers in randomly seeded hashbrown sets
// (recursive_assembly.rs:179-199, 499-645): the order in which it visits them — and with it which
// dense subgraph is found first, the order of a step's elements and therefore the column order of
// each cluster problem — differs from process to process. Every such visit is made in ASCENDING id
// order here; that is one of the orders the reference can take. Everything that is ordered in the
// reference (Vecs, the IndexMap of clusters, BTreeSets of a component) keeps its order.
//
// Pinned by the reference's Pose2D finite-difference test (expressions.rs:1470-1509) and by the threshold of its
// triangle test (tests/triangles.rs:10-37), both in tests/test_recursive_assembly.py. PLAN PARITY UNPINNED: the
// reference holds no known-answer vector for a plan (its plans are not reproducible between two of its own runs).
//
// Where the reference would panic (an `unwrap` on bookkeeping that is not there) `panicked` is set
// and the plan ends; where its exhaustive search would not finish, `exhausted` is set.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <deque>
#include <map>
#include <set>
#include <vector>

#include "fo_assemble.hpp"
#include "fo_expressions.hpp"
#include "fo_lm.hpp"

namespace fo {

// graph.rs:98-147 — only what the decomposition reads.
struct GeoGraph {
    struct Element {
        int16_t dof;
        std::vector<uint32_t> incident_constraints;  // one entry per occurrence (graph.rs:227-233)
    };
    struct Constraint {
        int16_t valency;
        std::vector<uint32_t> incident_elements;  // IncidentElements, 2..6 entries
    };
    std::vector<Element> elements;
    std::vector<Constraint> constraints;

    uint32_t add_element(int16_t dof) {  // graph.rs:160-176
        elements.push_back(Element{dof, {}});
        return static_cast<uint32_t>(elements.size() - 1);
    }
    uint32_t add_constraint(int16_t valency, const std::vector<uint32_t>& incident) {  // graph.rs:235-254
        uint32_t id = static_cast<uint32_t>(constraints.size());
        for (uint32_t e : incident) elements[e].incident_constraints.push_back(id);
        constraints.push_back(Constraint{valency, incident});
        return id;
    }
};

// IncidentElements::merge_elements, graph.rs:62-95
template <typename Pred>
inline std::vector<uint32_t> merge_elements(const std::vector<uint32_t>& incident, Pred merge, uint32_t into) {
    std::vector<uint32_t> out;
    bool merged = false;
    for (uint32_t e : incident) {
        if (merge(e)) {
            if (!merged) {
                out.push_back(into);
                merged = true;
            }
        } else {
            out.push_back(e);
        }
    }
    return out;
}

using IdSet = std::set<uint32_t>;
using IdListMap = std::map<uint32_t, std::vector<uint32_t>>;

// recursive_assembly.rs:76-117
struct RecombinationStep {
    std::vector<uint32_t> constraints;
    std::vector<uint32_t> elements;
    std::vector<uint32_t> free_elements;
    IdListMap on_frontiers;       // element -> clusters
    IdListMap owned_elements;     // cluster -> elements
    IdListMap frontier_elements;  // cluster -> elements

    const std::vector<uint32_t>* find(const IdListMap& m, uint32_t k) const {
        auto it = m.find(k);
        return it == m.end() ? nullptr : &it->second;
    }
};

struct RecombinationPlan {
    std::vector<RecombinationStep> steps;
    bool panicked = false;   // the reference would have panicked while planning
    bool exhausted = false;  // the exhaustive subgraph search went past the state budget
};

namespace ra_detail {

inline bool all_in(const std::vector<uint32_t>& els, const IdSet& set) {
    for (uint32_t e : els)
        if (!set.count(e)) return false;
    return true;
}

// recursive_assembly.rs:499-645. `k = -(D+1)`; the comparison `next_dof > k` is the reference's.
inline bool dense_bfs(const GeoGraph& graph, const std::vector<IdSet>& blocked, const IdSet& available_edges,
                      const IdSet& vertices, int16_t D, size_t budget, size_t& visited, IdSet& found, bool& exhausted) {
    const int k = -(D + 1);

    auto additional_valency = [&](const IdSet& next_subgraph, uint32_t new_vertex) {  // :508-531
        int add = 0;
        for (uint32_t edge_id : graph.elements[new_vertex].incident_constraints) {
            if (!available_edges.count(edge_id)) continue;
            const GeoGraph::Constraint& edge = graph.constraints[edge_id];
            if (all_in(edge.incident_elements, next_subgraph)) add += edge.valency;
        }
        return add;
    };
    auto extend_adjacent = [&](IdSet& adjacent, uint32_t from_vertex, const IdSet& subgraph) {  // :535-556
        for (uint32_t edge_id : graph.elements[from_vertex].incident_constraints) {
            if (!available_edges.count(edge_id)) continue;
            for (uint32_t inc : graph.constraints[edge_id].incident_elements) {
                if (vertices.count(inc) && !subgraph.count(inc)) adjacent.insert(inc);
            }
        }
    };

    struct State {  // :558-573
        IdSet subgraph;
        int dof;
        IdSet adjacent;
    };
    std::deque<State> queue;
    for (uint32_t vertex : vertices) {  // :577-597
        State s;
        s.subgraph.insert(vertex);
        extend_adjacent(s.adjacent, vertex, s.subgraph);
        s.dof = graph.elements[vertex].dof;
        queue.push_back(std::move(s));
    }
    while (!queue.empty()) {  // :599-641
        State cur = std::move(queue.front());
        queue.pop_front();
        for (uint32_t vertex : cur.adjacent) {
            if (++visited > budget) {
                exhausted = true;
                return false;
            }
            IdSet next = cur.subgraph;
            next.insert(vertex);
            int valency = additional_valency(next, vertex);
            int next_dof = cur.dof + graph.elements[vertex].dof - valency;
            bool is_blocked = std::find(blocked.begin(), blocked.end(), next) != blocked.end();
            if (!is_blocked && next_dof > k) {
                found = std::move(next);
                return true;
            }
            State ns;
            ns.adjacent = cur.adjacent;
            ns.adjacent.erase(vertex);
            extend_adjacent(ns.adjacent, vertex, next);
            ns.subgraph = std::move(next);
            ns.dof = next_dof;
            queue.push_back(std::move(ns));
        }
    }
    return false;
}

}  // namespace ra_detail

// recursive_assembly.rs:164-480 with D = 3.
inline RecombinationPlan decompose3(GeoGraph graph, const std::vector<uint32_t>& component_elements,
                                    const std::vector<uint32_t>& component_constraints, size_t budget = 200000) {
    using namespace ra_detail;
    const int16_t D = 3;
    const uint32_t num_real_constraints = static_cast<uint32_t>(graph.constraints.size());
    const uint32_t num_real_elements = static_cast<uint32_t>(graph.elements.size());

    IdSet vertices(component_elements.begin(), component_elements.end());
    IdSet available_edges(component_constraints.begin(), component_constraints.end());
    IdSet constraints_handled, vertices_handled;

    IdListMap on_frontiers, owned_elements, frontier_elements;
    std::map<uint32_t, uint32_t> owning_cluster;
    std::vector<IdSet> blocked_clusters;

    RecombinationPlan plan;
    std::vector<uint32_t> step_constraints, step_fixes_elements;
    size_t visited = 0;  // subgraphs grown so far, over all searches of this plan (`budget` bounds the total)

    for (uint32_t step = 0;; ++step) {
        const uint32_t cluster_key = step;

        IdSet subgraph;
        bool found = dense_bfs(graph, blocked_clusters, available_edges, vertices, D, budget, visited, subgraph, plan.exhausted);
        if (plan.exhausted) return plan;

        if (!found) {  // :211-252 — what is left is underconstrained: one step with all of it
            RecombinationStep rest;
            for (uint32_t edge : available_edges)
                if (edge < num_real_constraints && !constraints_handled.count(edge)) rest.constraints.push_back(edge);
            for (uint32_t v : vertices)
                if (v < num_real_elements && !vertices_handled.count(v)) rest.free_elements.push_back(v);
            if (!rest.constraints.empty()) {
                for (uint32_t v : vertices)
                    if (v < num_real_elements) rest.elements.push_back(v);
                rest.on_frontiers = on_frontiers;
                rest.owned_elements = owned_elements;
                rest.frontier_elements = frontier_elements;
                plan.steps.push_back(std::move(rest));
            }
            break;
        }

        // :260-309 — split the subgraph into core and frontier, collect this step's constraints
        std::vector<uint32_t> core, real_elements;
        IdSet frontier;
        for (uint32_t vertex : subgraph) {
            if (vertex < num_real_elements) real_elements.push_back(vertex);
            if (vertex < num_real_elements && !vertices_handled.count(vertex)) {
                step_fixes_elements.push_back(vertex);
                vertices_handled.insert(vertex);
                owning_cluster[vertex] = cluster_key;
            }
            bool frontier_vertex = false;
            for (uint32_t edge_id : graph.elements[vertex].incident_constraints) {
                if (!available_edges.count(edge_id)) continue;
                if (all_in(graph.constraints[edge_id].incident_elements, subgraph)) {
                    if (edge_id < num_real_constraints && !constraints_handled.count(edge_id)) {
                        step_constraints.push_back(edge_id);
                        constraints_handled.insert(edge_id);
                    }
                } else {
                    frontier_vertex = true;
                }
            }
            if (!frontier_vertex) core.push_back(vertex);
            else frontier.insert(vertex);
        }

        if (!step_constraints.empty()) {  // :311-321
            RecombinationStep s;
            s.constraints = std::move(step_constraints);
            step_constraints.clear();
            s.elements = real_elements;
            s.free_elements = step_fixes_elements;
            s.on_frontiers = on_frontiers;
            s.owned_elements = owned_elements;
            s.frontier_elements = frontier_elements;
            plan.steps.push_back(std::move(s));
        }

        if (!core.empty() || !step_fixes_elements.empty()) {  // :323-336
            owned_elements[cluster_key] = std::move(step_fixes_elements);
            step_fixes_elements.clear();
        }

        // :338-388 — bookkeeping of the core vertices: drop inner edges, merge owned clusters
        for (uint32_t vertex : core) {
            if (vertex < num_real_elements) {
                for (uint32_t edge_id : graph.elements[vertex].incident_constraints) {
                    bool inner = true;
                    for (uint32_t e : graph.constraints[edge_id].incident_elements)
                        inner = inner && std::find(core.begin(), core.end(), e) != core.end();
                    if (inner) available_edges.erase(edge_id);
                }
            }
            auto own = owning_cluster.find(vertex);
            if (own == owning_cluster.end()) {  // `.unwrap()` on None
                plan.panicked = true;
                return plan;
            }
            const uint32_t old_cluster_key = own->second;
            own->second = cluster_key;
            if (old_cluster_key != cluster_key) {
                auto oe = owned_elements.find(old_cluster_key);
                auto fe = frontier_elements.find(old_cluster_key);
                if (oe == owned_elements.end() || fe == frontier_elements.end() || !owned_elements.count(cluster_key)) {
                    plan.panicked = true;
                    return plan;
                }
                std::vector<uint32_t> old_owned = std::move(oe->second);
                owned_elements.erase(oe);
                for (uint32_t v : old_owned) owning_cluster[v] = cluster_key;
                std::vector<uint32_t>& mine = owned_elements[cluster_key];
                mine.insert(mine.end(), old_owned.begin(), old_owned.end());

                std::vector<uint32_t> old_frontier = std::move(fe->second);
                frontier_elements.erase(fe);
                for (uint32_t element : old_frontier) {
                    auto of = on_frontiers.find(element);
                    if (of == on_frontiers.end()) continue;
                    auto pos = std::find(of->second.begin(), of->second.end(), old_cluster_key);
                    if (pos == of->second.end()) {
                        plan.panicked = true;
                        return plan;
                    }
                    *pos = of->second.back();  // swap_remove
                    of->second.pop_back();
                }
            }
            on_frontiers.erase(vertex);
        }
        for (uint32_t vertex : frontier) {  // :389-398
            on_frontiers[vertex].push_back(cluster_key);
            if (vertex < num_real_elements) frontier_elements[cluster_key].push_back(vertex);
        }

        // :400-421 — fewer than two core vertices: nothing to contract, never look at this subgraph again
        if (subgraph.size() - frontier.size() <= 1) {
            blocked_clusters.push_back(subgraph);
            continue;
        }

        // :423-476 — contraction: the core becomes one vertex, its edges to the frontier are bundled
        for (uint32_t vertex : core) vertices.erase(vertex);
        const uint32_t core_vertex = graph.add_element(0);
        owning_cluster[core_vertex] = cluster_key;
        vertices.insert(core_vertex);

        int total_frontier_vertex_dof = 0, total_incoming_edge_valency = 0;
        for (uint32_t vertex : frontier) {
            total_frontier_vertex_dof += graph.elements[vertex].dof;
            int binary_edge_cluster_valency = 0;
            // (index loop: add_constraint below appends to incidence lists of *other* vertices only after this loop)
            const std::vector<uint32_t> incident = graph.elements[vertex].incident_constraints;
            for (uint32_t edge_id : incident) {
                if (!available_edges.count(edge_id)) continue;
                GeoGraph::Constraint& edge = graph.constraints[edge_id];
                if (all_in(edge.incident_elements, subgraph)) {
                    std::vector<uint32_t> merged =
                        merge_elements(edge.incident_elements, [&](uint32_t e) { return !frontier.count(e); }, core_vertex);
                    if (merged.size() == 2) {
                        binary_edge_cluster_valency += edge.valency;
                        available_edges.erase(edge_id);
                    } else {
                        edge.incident_elements = std::move(merged);
                    }
                }
            }
            if (binary_edge_cluster_valency > 0) {
                uint32_t cluster_edge =
                    graph.add_constraint(static_cast<int16_t>(binary_edge_cluster_valency), {vertex, core_vertex});
                available_edges.insert(cluster_edge);
                total_incoming_edge_valency += binary_edge_cluster_valency;
            }
        }
        if (total_incoming_edge_valency > 0) {
            graph.elements[core_vertex].dof =
                static_cast<int16_t>(total_frontier_vertex_dof - total_incoming_edge_valency - D);
        } else {
            vertices.erase(core_vertex);
        }
    }
    return plan;
}

// Flat serialisation of a plan, for comparing plans of two implementations:
// n_steps, then per step: |constraints| c.. |elements| e.. |free| f.. |on_frontiers| (el |cl| cl..)..
// |owned| (cl |el| el..).. |frontier_elements| (cl |el| el..)..   (maps in ascending key order)
inline std::vector<uint32_t> serialise_plan(const RecombinationPlan& plan) {
    std::vector<uint32_t> out;
    auto list = [&](const std::vector<uint32_t>& v) {
        out.push_back(static_cast<uint32_t>(v.size()));
        out.insert(out.end(), v.begin(), v.end());
    };
    auto map = [&](const IdListMap& m) {
        out.push_back(static_cast<uint32_t>(m.size()));
        for (const auto& kv : m) {
            out.push_back(kv.first);
            list(kv.second);
        }
    };
    out.push_back(static_cast<uint32_t>(plan.steps.size()));
    for (const RecombinationStep& s : plan.steps) {
        list(s.constraints);
        list(s.elements);
        list(s.free_elements);
        map(s.on_frontiers);
        map(s.owned_elements);
        map(s.frontier_elements);
    }
    return out;
}

// expressions.rs:1094-1159
struct Pose2D {
    double rotation, tx, ty;
    static Pose2D from_array(const double* p) { return Pose2D{p[0], p[1], p[2]}; }
    void transform_point(double u, double v, double& x, double& y) const {  // :1120-1134
        double s = std::sin(rotation), c = std::cos(rotation);
        double uc = u * c, us = u * s, vc = v * c, vs = v * s;
        x = tx + uc - vs;
        y = ty + us + vc;
    }
    void gradient_chain_rule_point(double u, double v, double gx, double gy, double out[3]) const {  // :1137-1157
        double s = std::sin(rotation), c = std::cos(rotation);
        double uc = u * c, us = u * s, vc = v * c, vs = v * s;
        out[0] = (-us - vc) * gx + (uc - vs) * gy;
        out[1] = gx;
        out[2] = gy;
    }
};

// What `System` holds beside the flat numeric state (lib.rs:123-137, 256-303).
struct GeoElementInfo {
    uint8_t kind;  // 0 Length, 1 Point, 2 Line, 3 Circle (EncodedElement)
    uint32_t idx;  // variable index of a Length / of a Point's x; unused otherwise
};
struct GeoConstraintInfo {
    uint8_t valency;            // ConstraintTag::valency
    uint32_t expressions_idx;   // first expression
    std::vector<uint32_t> incident_elements;
};
struct GeoComponent {
    std::vector<uint32_t> elements;     // ascending (BTreeSet)
    std::vector<uint32_t> constraints;  // ascending
};

// assemble/mod.rs:282-330 + build :333-476 + Problem impl :478-725
struct ClusteredSystem {
    uint32_t num_variables = 0;
    std::vector<uint32_t> step_plus_frontier_elements;
    std::vector<std::pair<uint32_t, std::vector<uint32_t>>> clusters;  // IndexMap: insertion order
    std::vector<uint32_t> expressions;
    uint32_t num_pose_expressions = 0;
    std::map<uint32_t, uint32_t> variable_mapping;  // global variable -> index in pose_and_element_variables
    bool panicked = false;

    // bound while solving
    const std::vector<GeoElementInfo>* elements = nullptr;
    const double* variables_transformed = nullptr;
    const Expression* expressions_transformed = nullptr;

    void build(const std::vector<GeoElementInfo>& els, const std::vector<GeoConstraintInfo>& cons,
               const double* vars_t, const RecombinationStep& step, std::vector<double>& pose_and_element_variables) {
        num_variables = 0;
        step_plus_frontier_elements.clear();
        clusters.clear();
        expressions.clear();
        num_pose_expressions = 0;
        variable_mapping.clear();
        pose_and_element_variables.clear();

        for (uint32_t c : step.constraints)  // :346-351
            for (uint32_t o = 0; o < cons[c].valency; ++o) expressions.push_back(cons[c].expressions_idx + o);

        step_plus_frontier_elements = step.elements;  // :353-354

        {  // :356-411 — transitive closure over shared frontier points
            std::vector<uint32_t> reachable;
            auto reach = [&](uint32_t element_id) {
                if (const std::vector<uint32_t>* cl = step.find(step.on_frontiers, element_id))
                    for (uint32_t c : *cl)
                        if (std::find(reachable.begin(), reachable.end(), c) == reachable.end()) reachable.push_back(c);
            };
            for (uint32_t element_id : step.elements) {
                if (els[element_id].kind != 1) continue;
                reach(element_id);
            }
            for (size_t i = 0; i < reachable.size(); ++i) {
                const std::vector<uint32_t>* fe = step.find(step.frontier_elements, reachable[i]);
                if (!fe) {  // `.unwrap()` on None
                    panicked = true;
                    return;
                }
                for (uint32_t element_id : *fe) {
                    if (els[element_id].kind != 1) continue;
                    reach(element_id);
                    const std::vector<uint32_t>* cl = step.find(step.on_frontiers, element_id);
                    size_t num_frontiers = cl ? cl->size() : 0;
                    bool have = std::find(step_plus_frontier_elements.begin(), step_plus_frontier_elements.end(),
                                          element_id) != step_plus_frontier_elements.end();
                    if (!have && num_frontiers > 1) step_plus_frontier_elements.push_back(element_id);
                }
            }
        }

        for (uint32_t element_id : step_plus_frontier_elements) {  // :413-429
            const std::vector<uint32_t>* cl = step.find(step.on_frontiers, element_id);
            if (!cl || els[element_id].kind != 1) continue;
            for (uint32_t cluster : *cl) {
                num_pose_expressions += 2;
                auto it = std::find_if(clusters.begin(), clusters.end(), [&](const auto& kv) { return kv.first == cluster; });
                if (it == clusters.end()) {
                    clusters.emplace_back(cluster, std::vector<uint32_t>{});
                    it = clusters.end() - 1;
                }
                it->second.push_back(element_id);
            }
        }

        pose_and_element_variables.assign(clusters.size() * 3, 0.);  // :431-432

        for (uint32_t element_id : step_plus_frontier_elements) {  // :434-474
            const GeoElementInfo& e = els[element_id];
            if (e.kind == 0) {
                variable_mapping[e.idx] = static_cast<uint32_t>(pose_and_element_variables.size());
                pose_and_element_variables.push_back(vars_t[e.idx]);
            } else if (e.kind == 1) {
                variable_mapping[e.idx] = static_cast<uint32_t>(pose_and_element_variables.size());
                variable_mapping[e.idx + 1] = static_cast<uint32_t>(pose_and_element_variables.size()) + 1;
                pose_and_element_variables.push_back(vars_t[e.idx]);
                pose_and_element_variables.push_back(vars_t[e.idx + 1]);
            }
        }
        num_variables = static_cast<uint32_t>(pose_and_element_variables.size());
    }

    // Problem, assemble/mod.rs:594-606
    uint32_t num_residuals() const { return static_cast<uint32_t>(expressions.size()) + num_pose_expressions; }

    // :608-664
    void calculate_residuals(const double* x, double* residuals) const {
        for (uint32_t i = 0; i < num_residuals(); ++i) residuals[i] = 0.;
        evaluate(x, residuals, nullptr);
    }
    // :478-590 (through :700-724)
    void calculate_residuals_and_sparse_jacobian(const double* x, double* residuals, TripletMat& jacobian) const {
        evaluate(x, residuals, &jacobian);
    }

    void evaluate(const double* x, double* residuals, TripletMat* jacobian) const {
        uint32_t idx[8];
        double vals[8] = {0, 0, 0, 0, 0, 0, 0, 0}, grad[8];
        const size_t offset = num_pose_expressions;
        for (size_t row = 0; row < expressions.size(); ++row) {
            const Expression& e = expressions_transformed[expressions[row]];
            int k = variable_indices(e, idx);
            for (int i = 0; i < k; ++i) vals[i] = x[variable_mapping.at(idx[i])];
            residuals[offset + row] = compute_residual_and_gradient(e, vals, grad);
            if (jacobian)
                for (int i = 0; i < k; ++i) jacobian->push_triplet(offset + row, variable_mapping.at(idx[i]), grad[i]);
        }
        size_t r = 0;
        for (size_t cluster_idx = 0; cluster_idx < clusters.size(); ++cluster_idx) {
            const size_t pose_start = 3 * cluster_idx;
            Pose2D pose = Pose2D::from_array(x + pose_start);
            for (uint32_t point_id : clusters[cluster_idx].second) {
                const uint32_t idx0 = (*elements)[point_id].idx;
                const double u = variables_transformed[idx0], v = variables_transformed[idx0 + 1];
                double tx, ty;
                pose.transform_point(u, v, tx, ty);
                const uint32_t updated = variable_mapping.at(idx0);
                residuals[r] = tx - x[updated];
                residuals[r + 1] = ty - x[updated + 1];
                if (jacobian) {
                    double gx[3], gy[3];
                    pose.gradient_chain_rule_point(u, v, 1., 0., gx);
                    pose.gradient_chain_rule_point(u, v, 0., 1., gy);
                    for (int q = 0; q < 3; ++q) jacobian->push_triplet(r, pose_start + q, gx[q]);
                    for (int q = 0; q < 3; ++q) jacobian->push_triplet(r + 1, pose_start + q, gy[q]);
                    jacobian->push_triplet(r, updated, -1.);
                    jacobian->push_triplet(r + 1, updated + 1, -1.);
                }
                r += 2;
            }
        }
    }
};

// Adapter with the member names `levenberg_marquardt` expects of a Problem.
struct ClusteredProblem {
    const ClusteredSystem& cs;
    uint32_t num_variables() const { return cs.num_variables; }
    uint32_t num_residuals() const { return cs.num_residuals(); }
    void calculate_residuals(const double* x, double* r) const { cs.calculate_residuals(x, r); }
    void calculate_residuals_and_sparse_jacobian(const double* x, double* r, TripletMat& j) const {
        cs.calculate_residuals_and_sparse_jacobian(x, r, j);
    }
};

struct GeoSystem {
    std::vector<double> variables;
    std::vector<uint8_t> fixed;
    std::vector<Expression> expressions;
    std::vector<GeoElementInfo> elements;
    std::vector<GeoConstraintInfo> constraints;
    std::vector<GeoComponent> components;  // live components in iteration order (assemble/mod.rs:81-89)
};

struct RecursiveStats {
    double scale = 0.;
    bool panicked = false, exhausted = false;
    std::vector<LmStats> steps;                 // one per solved step, components in order
    std::vector<std::pair<uint32_t, uint32_t>> sizes;  // (unknowns, rows) of each step's cluster problem
    std::vector<std::vector<uint32_t>> plans;   // serialised plan per component
};

inline GeoGraph geo_graph_of(const GeoSystem& s) {  // lib.rs:403 + constraints/mod.rs (graph.add_constraint calls)
    GeoGraph g;
    for (const GeoElementInfo& e : s.elements) g.add_element(e.kind == 0 ? 1 : e.kind == 1 ? 2 : 0);
    for (const GeoConstraintInfo& c : s.constraints) g.add_constraint(c.valency, c.incident_elements);
    return g;
}

// assemble/mod.rs:46-124 (scale, perturbation) + :212-277 (the arm). Mutates `s.variables`.
inline RecursiveStats solve_recursive_assembly(GeoSystem& s, bool perturb, QrOrdering ordering = QrOrdering::Colamd,
                                               uint32_t trial_cap = 0, size_t budget = 200000) {
    RecursiveStats out;
    Rng rng = Rng::from_seed(42);

    FlatSystem flat;
    flat.variables = s.variables;
    flat.expressions = s.expressions;
    const double system_scale = calculate_system_scale(flat);
    out.scale = system_scale;
    const double system_scale_recip = 1. / system_scale;

    std::vector<double> vt(s.variables.size());
    for (size_t i = 0; i < vt.size(); ++i) vt[i] = s.variables[i] * system_scale_recip;
    std::vector<Expression> et;
    et.reserve(s.expressions.size());
    for (const Expression& e : s.expressions) et.push_back(transform(e, system_scale_recip));

    const GeoGraph graph = geo_graph_of(s);

    for (const GeoComponent& comp : s.components) {
        if (comp.elements.empty()) continue;

        std::set<uint32_t> free_variables;  // :91-111
        for (uint32_t el : comp.elements) {
            const GeoElementInfo& e = s.elements[el];
            int n = e.kind == 0 ? 1 : e.kind == 1 ? 2 : 0;  // Lines / Circles never sit in a component
            for (int q = 0; q < n; ++q)
                if (!s.fixed[e.idx + q]) free_variables.insert(e.idx + q);
        }
        if (perturb) {  // :113-124
            for (uint32_t fv : free_variables) {
                double a = rng.next_f64();
                double b = rng.next_f64();
                vt[fv] += vt[fv] * (1. / 8196.) * a + (1. / 65568.) * b;
            }
        }

        RecombinationPlan plan = decompose3(graph, comp.elements, comp.constraints, budget);  // :213-217
        out.plans.push_back(serialise_plan(plan));
        if (plan.panicked || plan.exhausted) {
            out.panicked = out.panicked || plan.panicked;
            out.exhausted = out.exhausted || plan.exhausted;
            return out;
        }

        ClusteredSystem cs;
        cs.elements = &s.elements;
        cs.variables_transformed = vt.data();
        cs.expressions_transformed = et.data();
        std::vector<double> x;
        for (const RecombinationStep& step : plan.steps) {  // :220-276
            cs.build(s.elements, s.constraints, vt.data(), step, x);
            if (cs.panicked) {
                out.panicked = true;
                return out;
            }
            // every variable an expression of this step reads must be one of its unknowns (`.unwrap()`, :505-509)
            for (uint32_t eid : cs.expressions) {
                uint32_t idx[8];
                int k = variable_indices(et[eid], idx);
                for (int i = 0; i < k; ++i)
                    if (!cs.variable_mapping.count(idx[i])) {
                        out.panicked = true;
                        return out;
                    }
            }
            ClusteredProblem problem{cs};
            out.sizes.emplace_back(problem.num_variables(), problem.num_residuals());
            out.steps.push_back(levenberg_marquardt(problem, x.data(), ordering, trial_cap));

            for (const auto& kv : cs.variable_mapping) {  // :228-236
                vt[kv.first] = x[kv.second];
                s.variables[kv.first] = system_scale * x[kv.second];
            }
            for (size_t cluster_idx = 0; cluster_idx < cs.clusters.size(); ++cluster_idx) {  // :238-275
                Pose2D pose = Pose2D::from_array(x.data() + 3 * cluster_idx);
                const std::vector<uint32_t>* owned = step.find(step.owned_elements, cs.clusters[cluster_idx].first);
                if (!owned) continue;
                for (uint32_t element_id : *owned) {
                    bool in_step = std::find(cs.step_plus_frontier_elements.begin(), cs.step_plus_frontier_elements.end(),
                                             element_id) != cs.step_plus_frontier_elements.end();
                    if (in_step || s.elements[element_id].kind != 1) continue;
                    const uint32_t idx0 = s.elements[element_id].idx;
                    double tx, ty;
                    pose.transform_point(vt[idx0], vt[idx0 + 1], tx, ty);
                    vt[idx0] = tx;
                    vt[idx0 + 1] = ty;
                    s.variables[idx0] = system_scale * tx;
                    s.variables[idx0 + 1] = system_scale * ty;
                }
            }
        }
    }
    return out;
}

}  // namespace fo
