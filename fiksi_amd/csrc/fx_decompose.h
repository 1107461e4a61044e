// Host-side structural decomposition for `Decomposer::SinglePass`.
//
// Reference behaviour (fiksi/src/analyze/graph/equations.rs:186-221, assemble/mod.rs:169-210): the
// bipartite graph variables <-> expressions of the whole System, masked to the free variables of one
// connected component, gets a maximum matching (Hopcroft-Karp, :293-404); the matched expressions
// form a directed graph (:406-445) whose strongly connected components (Tarjan-Pearce, :447-550),
// taken in reverse order of discovery, are solved one after the other, each with its own
// Levenberg-Marquardt run that sees the values the earlier blocks produced.
//
// Maximum matchings are not unique, so the traversal orders below (free variables ascending,
// incidence lists in expression order, vertices in order of first match) are part of the contract:
// they make the blocks identical to the reference's. Everything is iterative: a sketch with tens of
// thousands of expressions must not depend on the host stack depth.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "fx_expr.h"

namespace fx {

// Whole-System incidence lists (lib.rs:386-389 and :429-441): every expression lists its variables
// in gradient order, every variable lists the expressions that read it, repeats included.
struct Incidence {
    std::vector<uint32_t> e_ptr, e_var;  // expression -> variables
    std::vector<uint32_t> v_ptr, v_expr; // variable -> expressions

    template <typename I>
    void build(uint32_t nvars, uint32_t nexprs, const uint8_t* tags, const I* idx4) {
        e_ptr.assign((size_t)nexprs + 1, 0);
        v_ptr.assign((size_t)nvars + 1, 0);
        e_var.clear();
        for (uint32_t e = 0; e < nexprs; ++e) {
            uint32_t v8[8];
            int k = expand_vars((int)(tags[e] & 0x7F), idx4 + 4 * (size_t)e, v8);
            for (int q = 0; q < k; ++q) {
                e_var.push_back(v8[q]);
                v_ptr[v8[q] + 1]++;
            }
            e_ptr[e + 1] = (uint32_t)e_var.size();
        }
        for (uint32_t v = 0; v < nvars; ++v) v_ptr[v + 1] += v_ptr[v];
        v_expr.assign(e_var.size(), 0);
        std::vector<uint32_t> fill(v_ptr.begin(), v_ptr.end() - 1);
        for (uint32_t e = 0; e < nexprs; ++e)
            for (uint32_t p = e_ptr[e]; p < e_ptr[e + 1]; ++p) v_expr[fill[e_var[p]]++] = e;
    }
};

// The blocks of one component, in solve order.
struct UnitList {
    std::vector<uint32_t> row_off{0}, rows;  // expressions of block u: rows[row_off[u] .. row_off[u+1])
    std::vector<uint32_t> var_off{0}, vars;  // its free variables, ascending
    uint32_t count() const { return (uint32_t)row_off.size() - 1; }
};

class SinglePassDecomposer {
  public:
    explicit SinglePassDecomposer(const Incidence& inc)
        : g_(inc),
          nv_((uint32_t)inc.v_ptr.size() - 1),
          ne_((uint32_t)inc.e_ptr.size() - 1),
          is_free_(nv_, 0),
          var_match_(nv_, NONE),
          expr_match_(ne_, NONE),
          dist_(nv_, INF),
          mark_(ne_, 0) {}

    // `free_sorted`: the component's free variables, ascending (BTreeSet order in the reference).
    void run(const std::vector<uint32_t>& free_sorted, UnitList& out) {
        for (uint32_t v : free_sorted) is_free_[v] = 1;
        order_.clear();
        match(free_sorted);
        components(out);
        // leave the scratch clean for the next component of the same System
        for (uint32_t v : free_sorted) {
            is_free_[v] = 0;
            var_match_[v] = NONE;
            dist_[v] = INF;
        }
        for (uint32_t e : order_) {
            expr_match_[e] = NONE;
            mark_[e] = 0;
        }
    }

  private:
    static constexpr uint32_t NONE = 0xFFFFFFFFu, INF = 0xFFFFFFFFu;
    const Incidence& g_;
    uint32_t nv_, ne_;
    std::vector<uint8_t> is_free_;
    std::vector<uint32_t> var_match_, expr_match_, dist_, mark_;
    std::vector<uint32_t> order_;  // expressions in order of their first match

    void pair(uint32_t v, uint32_t e) {
        var_match_[v] = e;
        if (expr_match_[e] == NONE) order_.push_back(e);
        expr_match_[e] = v;
    }

    // layers of the alternating-path search; returns the length of the shortest augmenting path
    uint32_t layers(const std::vector<uint32_t>& free_sorted) {
        std::vector<uint32_t> queue;
        for (uint32_t v : free_sorted) {
            if (var_match_[v] != NONE) {
                dist_[v] = INF;
            } else {
                dist_[v] = 0;
                queue.push_back(v);
            }
        }
        uint32_t shortest = INF;
        for (size_t head = 0; head < queue.size(); ++head) {
            uint32_t v = queue[head], d = dist_[v];
            if (d >= shortest) continue;
            uint32_t next = d == INF ? INF : d + 1;
            for (uint32_t p = g_.v_ptr[v]; p < g_.v_ptr[v + 1]; ++p) {
                uint32_t owner = expr_match_[g_.v_expr[p]];
                if (owner == NONE) {
                    if (shortest == INF) shortest = next;
                } else if (dist_[owner] == INF) {
                    dist_[owner] = next;
                    queue.push_back(owner);
                }
            }
        }
        return shortest;
    }

    // one depth-first augmentation from `start` along the layers (explicit stack)
    void augment(uint32_t start, uint32_t shortest) {
        struct Frame { uint32_t v, p; };
        std::vector<Frame> st{{start, g_.v_ptr[start]}};
        while (!st.empty()) {
            Frame& f = st.back();
            uint32_t v = f.v;
            uint32_t want = dist_[v] == INF ? INF : dist_[v] + 1;
            bool descended = false, found = false;
            while (f.p < g_.v_ptr[v + 1]) {
                uint32_t e = g_.v_expr[f.p];
                uint32_t owner = expr_match_[e];
                if (owner == NONE) {
                    if (shortest == want) {
                        found = true;
                        break;
                    }
                    ++f.p;
                } else if (dist_[owner] == want) {
                    st.push_back({owner, g_.v_ptr[owner]});  // f.p stays on e: resumed below
                    descended = true;
                    break;
                } else {
                    ++f.p;
                }
            }
            if (descended) continue;
            if (found) {
                // the path is complete: re-pair every frame with the expression it stopped on, deepest first
                for (size_t k = st.size(); k-- > 0;) pair(st[k].v, g_.v_expr[st[k].p]);
                return;
            }
            dist_[v] = INF;  // dead end
            st.pop_back();
            if (!st.empty()) ++st.back().p;  // the parent continues after the failed branch
        }
    }

    void match(const std::vector<uint32_t>& free_sorted) {
        for (;;) {
            uint32_t shortest = layers(free_sorted);
            if (shortest == INF) break;
            for (uint32_t v : free_sorted)
                if (var_match_[v] == NONE) augment(v, shortest);
        }
    }

    // successors of a matched expression: through its own variable and through unmatched free
    // variables, to every other matched expression that reads them
    void successors(uint32_t e, std::vector<uint32_t>& out) const {
        out.clear();
        uint32_t own = expr_match_[e];
        for (uint32_t p = g_.e_ptr[e]; p < g_.e_ptr[e + 1]; ++p) {
            uint32_t v = g_.e_var[p];
            if (!is_free_[v]) continue;
            if (v != own && var_match_[v] != NONE) continue;
            for (uint32_t q = g_.v_ptr[v]; q < g_.v_ptr[v + 1]; ++q) {
                uint32_t o = g_.v_expr[q];
                if (o != e && expr_match_[o] != NONE) out.push_back(o);
            }
        }
    }

    // Pearce's one-array SCC search. mark_[e]: 0 = unseen, otherwise a visit index (counted up from
    // 1) while e is open, or a component label (counted down from above every index) once closed.
    void components(UnitList& out) {
        struct Frame { uint32_t e, low; bool root; std::vector<uint32_t> next; size_t p; };
        std::vector<std::vector<uint32_t>> found;
        std::vector<uint32_t> open;
        uint32_t index = 1, label = 2 * (uint32_t)order_.size() + 1;
        std::vector<Frame> st;
        auto enter = [&](uint32_t e) {
            Frame f{e, index, true, {}, 0};
            mark_[e] = index++;
            successors(e, f.next);
            st.push_back(std::move(f));
        };
        for (uint32_t first : order_) {
            if (mark_[first] != 0) continue;
            enter(first);
            while (!st.empty()) {
                Frame& f = st.back();
                if (f.p < f.next.size()) {
                    uint32_t w = f.next[f.p];
                    if (mark_[w] == 0) {
                        enter(w);  // f.p is advanced when the child returns
                        continue;
                    }
                    if (mark_[w] < f.low) {
                        f.low = mark_[w];
                        mark_[f.e] = f.low;
                        f.root = false;
                    }
                    ++f.p;
                    continue;
                }
                // all successors done
                uint32_t e = f.e, low = f.low;
                bool root = f.root;
                st.pop_back();
                if (root) {
                    std::vector<uint32_t> scc{e};
                    index -= 1;
                    while (!open.empty() && !(low > mark_[open.back()])) {
                        uint32_t t = open.back();
                        open.pop_back();
                        scc.push_back(t);
                        mark_[t] = label;
                        index -= 1;
                    }
                    mark_[e] = label;
                    label -= 1;
                    found.push_back(std::move(scc));
                } else {
                    open.push_back(e);
                }
                // the parent now looks at this child again: mark_ is set, so it takes the compare branch
            }
        }
        out = UnitList{};
        for (size_t k = found.size(); k-- > 0;) {  // reverse discovery order = dependency order
            const auto& scc = found[k];
            size_t v_begin = out.vars.size();
            for (uint32_t e : scc) {
                out.rows.push_back(e);
                uint32_t own = expr_match_[e];
                for (uint32_t p = g_.e_ptr[e]; p < g_.e_ptr[e + 1]; ++p) {
                    uint32_t v = g_.e_var[p];
                    if (v == own || (is_free_[v] && var_match_[v] == NONE)) out.vars.push_back(v);
                }
            }
            std::sort(out.vars.begin() + v_begin, out.vars.end());
            out.vars.erase(std::unique(out.vars.begin() + v_begin, out.vars.end()), out.vars.end());
            out.row_off.push_back((uint32_t)out.rows.size());
            out.var_off.push_back((uint32_t)out.vars.size());
        }
    }
};

}  // namespace fx
