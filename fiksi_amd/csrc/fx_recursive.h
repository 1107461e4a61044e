// Host side of Decomposer::RecursiveAssembly: the recombination plan and the cluster problem of each step.
// Pure host bookkeeping (no arithmetic on variables): the solves run on the device (pose instantiation of the
// one-wavefront LM kernel, fx_kernels.hip), as do scaling / perturbation and the rigid moves (fx_cluster.hip).
//
// What it follows in the reference:
//   plan      fiksi/src/analyze/graph/recursive_assembly.rs:164-480 (`decompose::<3>`: find a dense subgraph, split it
//             into core and frontier, emit a step for its not-yet-solved constraints, contract the core) and :499-645
//             (`dense_bfs`: breadth-first growth of vertex sets until one passes `dof - valency > -(D+1)`)
//   problem   fiksi/src/assemble/mod.rs:333-476 (`ClusteredSystem::build`): unknowns = one pose per reachable cluster,
//             then the variables of the step's elements and of the points shared between several frontiers;
//             rows = two pose rows per (cluster, point on its frontier), then the step's expressions.
//
// The reference walks hashbrown sets seeded per process, so its plan differs from run to run; here every set is a
// sorted vector and is walked in ascending id order (one of the orders the reference may take). Lists the reference
// keeps ordered (Vec, IndexMap) keep their order. Where the reference would panic, `Plan::panicked` is set; its
// search is exhaustive and can take forever on sketches of more than a handful of elements: `budget` bounds the
// number of subgraphs all searches of one plan may grow.
#pragma once
#include <algorithm>
#include <cstdint>
#include <deque>
#include <utility>
#include <vector>

namespace fx {
namespace ra {

// a set of ids as a sorted vector
struct IdSet {
    std::vector<uint32_t> v;
    bool has(uint32_t x) const { return std::binary_search(v.begin(), v.end(), x); }
    void add(uint32_t x) {
        auto it = std::lower_bound(v.begin(), v.end(), x);
        if (it == v.end() || *it != x) v.insert(it, x);
    }
    void drop(uint32_t x) {
        auto it = std::lower_bound(v.begin(), v.end(), x);
        if (it != v.end() && *it == x) v.erase(it);
    }
    size_t size() const { return v.size(); }
    bool operator==(const IdSet& o) const { return v == o.v; }
};

struct Edge {
    int valency;
    std::vector<uint32_t> ends;  // incident vertices, 2..6, order of creation
};

// graph.rs:98-147 as far as the plan reads it
struct Graph {
    std::vector<int> dof;
    std::vector<std::vector<uint32_t>> edges_of;  // one entry per occurrence of the vertex in an edge
    std::vector<Edge> edges;
    uint32_t new_vertex(int d) {
        dof.push_back(d);
        edges_of.emplace_back();
        return (uint32_t)dof.size() - 1u;
    }
    uint32_t new_edge(int valency, const std::vector<uint32_t>& ends) {
        const uint32_t id = (uint32_t)edges.size();
        for (uint32_t e : ends) edges_of[e].push_back(id);
        edges.push_back(Edge{valency, ends});
        return id;
    }
};

// key -> list tables, ascending by key; an entry may exist with an empty list
using Table = std::vector<std::pair<uint32_t, std::vector<uint32_t>>>;
inline const std::vector<uint32_t>* lookup(const Table& t, uint32_t key) {
    auto it = std::lower_bound(t.begin(), t.end(), key, [](const auto& kv, uint32_t k) { return kv.first < k; });
    return (it != t.end() && it->first == key) ? &it->second : nullptr;
}

struct Step {                          // RecombinationStep, recursive_assembly.rs:76-117
    std::vector<uint32_t> constraints;
    std::vector<uint32_t> elements;
    std::vector<uint32_t> free_elements;
    Table on_frontiers;                // element -> clusters whose frontier it is on
    Table owned_elements;              // cluster -> elements it carries
    Table frontier_elements;           // cluster -> its frontier
};

struct Plan {
    std::vector<Step> steps;
    bool panicked = false;
    bool exhausted = false;
};

// dense tables with presence flags while planning; a snapshot turns them into a Table
struct Slots {
    std::vector<std::vector<uint32_t>> list;
    std::vector<uint8_t> present;
    void reach(uint32_t key) {
        if (key >= list.size()) {
            list.resize(key + 1);
            present.resize(key + 1, 0);
        }
    }
    bool has(uint32_t key) const { return key < present.size() && present[key]; }
    std::vector<uint32_t>& at(uint32_t key) {  // creates the entry
        reach(key);
        present[key] = 1;
        return list[key];
    }
    void erase(uint32_t key) {
        if (has(key)) {
            present[key] = 0;
            list[key].clear();
        }
    }
    Table snapshot() const {
        Table t;
        for (uint32_t k = 0; k < list.size(); ++k)
            if (present[k]) t.emplace_back(k, list[k]);
        return t;
    }
};

inline bool inside(const std::vector<uint32_t>& ends, const IdSet& set) {
    for (uint32_t e : ends)
        if (!set.has(e)) return false;
    return true;
}

// recursive_assembly.rs:499-645
inline bool find_dense(const Graph& g, const std::vector<IdSet>& blocked, const IdSet& live_edges, const IdSet& vertices, int D,
                       size_t budget, size_t& work, IdSet& out, bool& exhausted) {
    struct Grown {
        IdSet members;
        int dof;
        IdSet border;  // vertices one live edge away
    };
    auto widen = [&](IdSet& border, uint32_t from, const IdSet& members) {
        for (uint32_t e : g.edges_of[from]) {
            if (!live_edges.has(e)) continue;
            for (uint32_t w : g.edges[e].ends)
                if (vertices.has(w) && !members.has(w)) border.add(w);
        }
    };
    std::deque<Grown> todo;
    for (uint32_t v : vertices.v) {
        Grown s;
        s.members.add(v);
        s.dof = g.dof[v];
        widen(s.border, v, s.members);
        todo.push_back(std::move(s));
    }
    const int threshold = -(D + 1);
    while (!todo.empty()) {
        const Grown cur = std::move(todo.front());
        todo.pop_front();
        for (uint32_t v : cur.border.v) {
            if (++work > budget) {
                exhausted = true;
                return false;
            }
            Grown nxt;
            nxt.members = cur.members;
            nxt.members.add(v);
            int closed = 0;  // valency of the live edges of v that now lie inside
            for (uint32_t e : g.edges_of[v])
                if (live_edges.has(e) && inside(g.edges[e].ends, nxt.members)) closed += g.edges[e].valency;
            nxt.dof = cur.dof + g.dof[v] - closed;
            const bool is_blocked = std::find(blocked.begin(), blocked.end(), nxt.members) != blocked.end();
            if (!is_blocked && nxt.dof > threshold) {
                out = std::move(nxt.members);
                return true;
            }
            nxt.border = cur.border;
            nxt.border.drop(v);
            widen(nxt.border, v, nxt.members);
            todo.push_back(std::move(nxt));
        }
    }
    return false;
}

// recursive_assembly.rs:164-480, D = 3
inline Plan make_plan(Graph g, const std::vector<uint32_t>& component_elements, const std::vector<uint32_t>& component_constraints,
                      size_t budget) {
    const int D = 3;
    const uint32_t real_edges = (uint32_t)g.edges.size(), real_vertices = (uint32_t)g.dof.size();
    IdSet vertices, live_edges, edges_done, vertices_done;
    for (uint32_t e : component_elements) vertices.add(e);
    for (uint32_t c : component_constraints) live_edges.add(c);

    Slots on_frontiers, owned, frontier_of;
    std::vector<int64_t> owner(real_vertices, -1);
    std::vector<IdSet> blocked;
    Plan plan;
    std::vector<uint32_t> step_edges, step_new;
    size_t work = 0;  // subgraphs grown by all searches of this plan; `budget` bounds the total

    auto emit = [&](std::vector<uint32_t> constraints, std::vector<uint32_t> elements, const std::vector<uint32_t>& fresh) {
        Step st;
        st.constraints = std::move(constraints);
        st.elements = std::move(elements);
        st.free_elements = fresh;
        st.on_frontiers = on_frontiers.snapshot();
        st.owned_elements = owned.snapshot();
        st.frontier_elements = frontier_of.snapshot();
        plan.steps.push_back(std::move(st));
    };

    for (uint32_t key = 0;; ++key) {
        IdSet sub;
        const bool found = find_dense(g, blocked, live_edges, vertices, D, budget, work, sub, plan.exhausted);
        if (plan.exhausted) return plan;
        if (!found) {  // :211-252 the rest is under-constrained: solved in one go
            std::vector<uint32_t> rest_edges, rest_new, rest_all;
            for (uint32_t e : live_edges.v)
                if (e < real_edges && !edges_done.has(e)) rest_edges.push_back(e);
            for (uint32_t v : vertices.v) {
                if (v >= real_vertices) continue;
                rest_all.push_back(v);
                if (!vertices_done.has(v)) rest_new.push_back(v);
            }
            if (!rest_edges.empty()) emit(std::move(rest_edges), std::move(rest_all), rest_new);
            break;
        }

        // :260-309 core / frontier split, constraints that close inside the subgraph
        std::vector<uint32_t> core, real_members;
        IdSet frontier;
        for (uint32_t v : sub.v) {
            const bool real = v < real_vertices;
            if (real) real_members.push_back(v);
            if (real && !vertices_done.has(v)) {
                step_new.push_back(v);
                vertices_done.add(v);
                owner[v] = key;
            }
            bool reaches_out = false;
            for (uint32_t e : g.edges_of[v]) {
                if (!live_edges.has(e)) continue;
                if (inside(g.edges[e].ends, sub)) {
                    if (e < real_edges && !edges_done.has(e)) {
                        step_edges.push_back(e);
                        edges_done.add(e);
                    }
                } else {
                    reaches_out = true;
                }
            }
            if (reaches_out) frontier.add(v);
            else core.push_back(v);
        }
        if (!step_edges.empty()) {  // :311-321
            emit(std::move(step_edges), real_members, step_new);
            step_edges.clear();
        }
        if (!core.empty() || !step_new.empty()) {  // :323-336
            owned.at(key) = std::move(step_new);
            step_new.clear();
        }

        // :338-388
        for (uint32_t v : core) {
            if (v < real_vertices) {
                for (uint32_t e : g.edges_of[v]) {
                    bool within = true;
                    for (uint32_t w : g.edges[e].ends) within = within && std::find(core.begin(), core.end(), w) != core.end();
                    if (within) live_edges.drop(e);
                }
            }
            if (v >= owner.size() || owner[v] < 0) {
                plan.panicked = true;
                return plan;
            }
            const uint32_t before = (uint32_t)owner[v];
            owner[v] = key;
            if (before != key) {  // the cluster that carried v is absorbed
                if (!owned.has(before) || !frontier_of.has(before) || !owned.has(key)) {
                    plan.panicked = true;
                    return plan;
                }
                const std::vector<uint32_t> carried = owned.list[before];
                owned.erase(before);
                for (uint32_t w : carried) owner[w] = key;
                std::vector<uint32_t>& mine = owned.at(key);
                mine.insert(mine.end(), carried.begin(), carried.end());
                const std::vector<uint32_t> old_frontier = frontier_of.list[before];
                frontier_of.erase(before);
                for (uint32_t w : old_frontier) {
                    if (!on_frontiers.has(w)) continue;
                    std::vector<uint32_t>& cl = on_frontiers.list[w];
                    auto pos = std::find(cl.begin(), cl.end(), before);
                    if (pos == cl.end()) {
                        plan.panicked = true;
                        return plan;
                    }
                    *pos = cl.back();  // Vec::swap_remove
                    cl.pop_back();
                }
            }
            on_frontiers.erase(v);
        }
        for (uint32_t v : frontier.v) {  // :389-398
            on_frontiers.at(v).push_back(key);
            if (v < real_vertices) frontier_of.at(key).push_back(v);
        }

        if (sub.size() - frontier.size() <= 1) {  // :400-421
            blocked.push_back(sub);
            continue;
        }

        // :423-476 contraction
        for (uint32_t v : core) vertices.drop(v);
        const uint32_t hub = g.new_vertex(0);
        owner.resize(g.dof.size(), -1);
        owner[hub] = key;
        vertices.add(hub);
        int frontier_dof = 0, incoming = 0;
        for (uint32_t v : frontier.v) {
            frontier_dof += g.dof[v];
            int bundle = 0;
            const std::vector<uint32_t> touching = g.edges_of[v];
            for (uint32_t e : touching) {
                if (!live_edges.has(e)) continue;
                Edge& edge = g.edges[e];
                if (!inside(edge.ends, sub)) continue;
                std::vector<uint32_t> ends;
                bool placed = false;
                for (uint32_t w : edge.ends) {  // IncidentElements::merge_elements, graph.rs:62-95
                    if (frontier.has(w)) {
                        ends.push_back(w);
                    } else if (!placed) {
                        ends.push_back(hub);
                        placed = true;
                    }
                }
                if (ends.size() == 2) {
                    bundle += edge.valency;
                    live_edges.drop(e);
                } else {
                    edge.ends = std::move(ends);
                }
            }
            if (bundle > 0) {
                live_edges.add(g.new_edge(bundle, {v, hub}));
                incoming += bundle;
            }
        }
        if (incoming > 0) g.dof[hub] = frontier_dof - incoming - D;
        else vertices.drop(hub);
    }
    return plan;
}

// n_steps, then per step |constraints| .. |elements| .. |free| .. and the three tables as
// |entries| (key |list| ..).. — the form tests compare plans in
inline void serialise(const Plan& plan, std::vector<uint32_t>& out) {
    auto put = [&](const std::vector<uint32_t>& v) {
        out.push_back((uint32_t)v.size());
        out.insert(out.end(), v.begin(), v.end());
    };
    auto put_table = [&](const Table& t) {
        out.push_back((uint32_t)t.size());
        for (const auto& kv : t) {
            out.push_back(kv.first);
            put(kv.second);
        }
    };
    out.push_back((uint32_t)plan.steps.size());
    for (const Step& s : plan.steps) {
        put(s.constraints);
        put(s.elements);
        put(s.free_elements);
        put_table(s.on_frontiers);
        put_table(s.owned_elements);
        put_table(s.frontier_elements);
    }
}

// What a step's problem is made of (ClusteredSystem::build, assemble/mod.rs:333-476).
struct ClusterProblem {
    std::vector<uint32_t> members;  // step_plus_frontier_elements
    std::vector<std::pair<uint32_t, std::vector<uint32_t>>> clusters;  // insertion order: cluster -> points with a pose row pair
    bool panicked = false;
};

// `is_point(element)`: only points take part in rigid moves (:368-371, :383-386)
template <typename IsPoint>
inline ClusterProblem make_cluster_problem(const Step& step, IsPoint is_point) {
    ClusterProblem cp;
    cp.members = step.elements;
    std::vector<uint32_t> reach;
    auto note = [&](uint32_t element) {
        if (const std::vector<uint32_t>* cl = lookup(step.on_frontiers, element))
            for (uint32_t c : *cl)
                if (std::find(reach.begin(), reach.end(), c) == reach.end()) reach.push_back(c);
    };
    for (uint32_t e : step.elements)
        if (is_point(e)) note(e);
    for (size_t i = 0; i < reach.size(); ++i) {
        const std::vector<uint32_t>* fr = lookup(step.frontier_elements, reach[i]);
        if (!fr) {
            cp.panicked = true;
            return cp;
        }
        for (uint32_t e : *fr) {
            if (!is_point(e)) continue;
            note(e);
            const std::vector<uint32_t>* cl = lookup(step.on_frontiers, e);
            const bool shared = cl && cl->size() > 1;
            if (shared && std::find(cp.members.begin(), cp.members.end(), e) == cp.members.end()) cp.members.push_back(e);
        }
    }
    for (uint32_t e : cp.members) {
        if (!is_point(e)) continue;
        const std::vector<uint32_t>* cl = lookup(step.on_frontiers, e);
        if (!cl) continue;
        for (uint32_t c : *cl) {
            auto it = std::find_if(cp.clusters.begin(), cp.clusters.end(), [&](const auto& kv) { return kv.first == c; });
            if (it == cp.clusters.end()) {
                cp.clusters.emplace_back(c, std::vector<uint32_t>{});
                it = cp.clusters.end() - 1;
            }
            it->second.push_back(e);
        }
    }
    return cp;
}

}  // namespace ra
}  // namespace fx
