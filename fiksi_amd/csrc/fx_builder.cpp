// Host-side System builder behind include/fiksi_amd_builder.h. Bookkeeping only: it records
// elements, constraints, fixed variables and the incremental connected components exactly as the
// reference builder does, and flattens Systems into fx_batch. No numerics live here.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/fiksi_amd_builder.h"
#include "fx_recursive.h"

#include "fx_guard.h"  // fx::set_last_error (fx_entry.cpp): the text of fx_last_error(); the guards of the entry points below

namespace {

int bfail(int code, const char* msg) {
    fx::set_last_error(msg);
    return code;
}

// EncodedElement, fiksi/src/lib.rs:123-128
struct Element {
    int tag;       // fxs_element_tag
    uint32_t a, b; // Length{idx=a}; Point{idx=a}; Line{point1_idx=a, point2_idx=b}; Circle{center_idx=a, radius_idx=b}
};

// EncodedConstraint, fiksi/src/lib.rs:134-137
struct Constraint {
    int tag;
    uint32_t expressions_idx;
    uint8_t n_incident;     // IncidentElements handed to Graph::add_constraint (graph.rs:14-57)
    uint32_t incident[6];
};

struct Expr {
    uint8_t tag;
    uint32_t idx[4];
    double param;
};

// ConnectedComponent, fiksi/src/graph.rs:121-125 (BTreeSets -> std::set, ascending)
struct Component {
    std::set<uint32_t> elements;
    std::set<uint32_t> constraints;
};

}  // namespace

struct fxs_system {
    uint32_t id = 0;
    std::vector<Element> elements;
    std::vector<double> variables;
    std::vector<uint32_t> variable_to_primitive;  // lib.rs:279
    std::set<uint32_t> fixed_variables;           // lib.rs:283
    std::vector<Constraint> constraints;
    std::vector<Expr> expressions;
    // graph.rs:135-147
    std::vector<int32_t> element_component;  // 1-based index into `components`, 0 = None
    std::vector<Component> components;
    uint32_t listed = 0;  // scratch of fxs_systems_solve: how often the current call has listed this System so far
};

struct fxs_flat {
    std::vector<uint32_t> var_off, expr_off, expr_idx;
    std::vector<double> vars, expr_param;
    std::vector<uint8_t> var_fixed, expr_tag;
    std::vector<uint16_t> var_comp, expr_comp;
    fx_batch batch{};
};

namespace {

struct FlatGuard {  // a flat batch made for the length of one call
    fxs_flat* f;
    ~FlatGuard() { delete f; }
};

std::atomic<uint32_t> g_system_counter{0};

int valency_of(int tag) { return tag == FXS_POINT_POINT_COINCIDENCE ? 2 : 1; }

// Room for k more entries, growing geometrically (reserve(size + k) alone would reallocate at every call): what follows a
// successful make_room is push_backs that cannot throw, so a builder call either happens entirely or not at all.
template <typename V>
void make_room(V& v, size_t k) {
    if (v.capacity() - v.size() < k) v.reserve(std::max(v.size() + k, 2 * v.capacity()));
}

// System::add_element, lib.rs:363-407 (+ Graph::add_element, graph.rs:160-176)
int64_t add_element(fxs_system* s, int tag, const double* vars, int nvars, uint32_t a, uint32_t b) {
    uint32_t id = (uint32_t)s->elements.size();
    uint32_t variables_idx = (uint32_t)s->variables.size();
    make_room(s->variables, (size_t)nvars);
    make_room(s->variable_to_primitive, (size_t)nvars);
    make_room(s->element_component, 1);
    make_room(s->elements, 1);
    for (int i = 0; i < nvars; ++i) {
        s->variables.push_back(vars[i]);
        s->variable_to_primitive.push_back(id);
    }
    s->element_component.push_back(0);
    Element e{tag, a, b};
    if (tag == FXS_LENGTH || tag == FXS_POINT) e.a = variables_idx;
    s->elements.push_back(e);
    return id;
}

// Graph::merge_connected_components, graph.rs:178-225 — including its behaviour of re-labelling
// only the *incident* elements of an absorbed component. The absorbed components' members are added to
// the target in place; should that run out of memory half-way, what was added is taken out again and the graph is as before.
void merge_components(fxs_system* s, uint32_t constraint, const uint32_t* els, int n) {
    int32_t target = 0;
    size_t size_largest = 0;
    for (int i = 0; i < n; ++i) {
        int32_t ci = s->element_component[els[i]];
        if (ci != 0) {
            const Component& c = s->components[(size_t)ci - 1];
            if (c.elements.size() > size_largest) {
                target = ci;
                size_largest = c.elements.size();
            }
        }
    }
    size_t incoming = (size_t)n + 1;
    for (int i = 0; i < n; ++i) {
        const int32_t ci = s->element_component[els[i]];
        if (ci != 0 && ci != target) incoming += s->components[(size_t)ci - 1].elements.size() + s->components[(size_t)ci - 1].constraints.size();
    }
    std::vector<uint32_t> added_el, added_con;
    added_el.reserve(incoming);
    added_con.reserve(incoming);
    const bool fresh = target == 0;
    if (fresh) {
        s->components.emplace_back();
        target = (int32_t)s->components.size();
    }
    Component& tc = s->components[(size_t)target - 1];
    try {
        for (int i = 0; i < n; ++i) {
            const int32_t ci = s->element_component[els[i]];
            if (ci != 0 && ci != target) {
                const Component& c = s->components[(size_t)ci - 1];
                for (uint32_t e : c.elements)
                    if (tc.elements.insert(e).second) added_el.push_back(e);
                for (uint32_t k : c.constraints)
                    if (tc.constraints.insert(k).second) added_con.push_back(k);
            } else if (ci == 0) {
                if (tc.elements.insert(els[i]).second) added_el.push_back(els[i]);
            }
        }
        if (tc.constraints.insert(constraint).second) added_con.push_back(constraint);
    } catch (...) {
        for (uint32_t e : added_el) tc.elements.erase(e);
        for (uint32_t k : added_con) tc.constraints.erase(k);
        if (fresh) s->components.pop_back();
        throw;
    }
    // nothing below can throw
    for (int i = 0; i < n; ++i) {
        const int32_t ci = s->element_component[els[i]];
        if (ci != 0 && ci != target) {
            s->components[(size_t)ci - 1].elements.clear();
            s->components[(size_t)ci - 1].constraints.clear();
        }
        s->element_component[els[i]] = target;
    }
}

bool is_tag(const fxs_system* s, uint32_t el, int tag) { return el < s->elements.size() && s->elements[el].tag == tag; }

// variables of an element, `T::variable_indices` (elements/mod.rs:296-301,334-339,406-418,470-482)
int element_variables(const fxs_system* s, uint32_t el, uint32_t out[4]) {
    const Element& e = s->elements[el];
    switch (e.tag) {
        case FXS_LENGTH: out[0] = e.a; return 1;
        case FXS_POINT: out[0] = e.a; out[1] = e.a + 1; return 2;
        case FXS_LINE: out[0] = e.a; out[1] = e.a + 1; out[2] = e.b; out[3] = e.b + 1; return 4;
        case FXS_CIRCLE: out[0] = e.a; out[1] = e.a + 1; out[2] = e.b; return 3;
    }
    return 0;
}

// assemble/mod.rs:81-111: the live components in iteration order, empties skipped
void live_components(const fxs_system* s, std::vector<const Component*>& out) {
    out.clear();
    for (const Component& c : s->components) {
        if (c.elements.empty()) continue;
        out.push_back(&c);
    }
}

// Graph::add_element / add_constraint as lib.rs:403 and constraints/mod.rs:331-880 call them:
// a Length has 1 degree of freedom, a Point 2, Lines and Circles 0 (they own no variables).
fx::ra::Graph plan_graph(const fxs_system* s) {
    fx::ra::Graph g;
    for (const Element& e : s->elements) g.new_vertex(e.tag == FXS_LENGTH ? 1 : e.tag == FXS_POINT ? 2 : 0);
    for (const Constraint& c : s->constraints) g.new_edge(valency_of(c.tag), std::vector<uint32_t>(c.incident, c.incident + c.n_incident));
    return g;
}

constexpr uint64_t kDefaultPlanBudget = 200000;  // subgraphs the plan search may grow (the reference's search is unbounded)

// element fields an expression names (fx_batch::expr_idx)
int expr_fields(uint8_t tag) {
    static const int k[11] = {2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4};
    return tag < 11 ? k[tag] : 0;
}

}  // namespace

extern "C" {

int fxs_system_new(fxs_system** out) try {
    if (!out) return FX_ERR_INVALID;
    fxs_system* s = new (std::nothrow) fxs_system();
    if (!s) return FX_ERR_NOMEM;
    s->id = g_system_counter.fetch_add(1, std::memory_order_relaxed);
    *out = s;
    return FX_OK;
}
FX_CATCH_CODE

void fxs_system_free(fxs_system* s) try { delete s; } FX_CATCH_VOID
uint32_t fxs_system_id(const fxs_system* s) { return s ? s->id : 0; }
uint32_t fxs_num_elements(const fxs_system* s) { return s ? (uint32_t)s->elements.size() : 0; }
uint32_t fxs_num_constraints(const fxs_system* s) { return s ? (uint32_t)s->constraints.size() : 0; }
uint32_t fxs_num_variables(const fxs_system* s) { return s ? (uint32_t)s->variables.size() : 0; }
uint32_t fxs_num_expressions(const fxs_system* s) { return s ? (uint32_t)s->expressions.size() : 0; }

int64_t fxs_length_create(fxs_system* s, double length) try {  // elements/mod.rs:280-284
    if (!s) return FX_ERR_INVALID;
    return add_element(s, FXS_LENGTH, &length, 1, 0, 0);
}
FX_CATCH_CODE

int64_t fxs_point_create(fxs_system* s, double x, double y) try {  // elements/mod.rs:321-325
    if (!s) return FX_ERR_INVALID;
    double v[2] = {x, y};
    return add_element(s, FXS_POINT, v, 2, 0, 0);
}
FX_CATCH_CODE

int64_t fxs_line_create(fxs_system* s, uint32_t point1, uint32_t point2) try {  // elements/mod.rs:365-382
    if (!s || !is_tag(s, point1, FXS_POINT) || !is_tag(s, point2, FXS_POINT)) return bfail(FX_ERR_INVALID, "Line::create needs two points");
    return add_element(s, FXS_LINE, nullptr, 0, s->elements[point1].a, s->elements[point2].a);
}
FX_CATCH_CODE

int64_t fxs_circle_create(fxs_system* s, uint32_t center, uint32_t radius) try {  // elements/mod.rs:437-454
    if (!s || !is_tag(s, center, FXS_POINT) || !is_tag(s, radius, FXS_LENGTH)) return bfail(FX_ERR_INVALID, "Circle::create needs a point and a length");
    return add_element(s, FXS_CIRCLE, nullptr, 0, s->elements[center].a, s->elements[radius].a);
}
FX_CATCH_CODE

int fxs_element_tag_of(const fxs_system* s, uint32_t element) try {
    if (!s || element >= s->elements.size()) return FX_ERR_INVALID;
    return s->elements[element].tag;
}
FX_CATCH_CODE

int fxs_element_fix(fxs_system* s, uint32_t element) try {  // elements/mod.rs:60-65
    if (!s || element >= s->elements.size()) return FX_ERR_INVALID;
    uint32_t v[4];
    int n = element_variables(s, element, v);
    bool added[4] = {false, false, false, false};
    try {
        for (int i = 0; i < n; ++i) added[i] = s->fixed_variables.insert(v[i]).second;
    } catch (...) {  // (all of the element's variables or none)
        for (int i = 0; i < n; ++i)
            if (added[i]) s->fixed_variables.erase(v[i]);
        throw;
    }
    return FX_OK;
}
FX_CATCH_CODE

int fxs_element_unfix(fxs_system* s, uint32_t element) try {  // elements/mod.rs:80-85
    if (!s || element >= s->elements.size()) return FX_ERR_INVALID;
    uint32_t v[4];
    int n = element_variables(s, element, v);
    for (int i = 0; i < n; ++i) s->fixed_variables.erase(v[i]);
    return FX_OK;
}
FX_CATCH_CODE

int fxs_element_get_value(const fxs_system* s, uint32_t element, double out[4]) try {  // elements/mod.rs:88-100
    if (!s || !out || element >= s->elements.size()) return FX_ERR_INVALID;
    uint32_t v[4];
    int n = element_variables(s, element, v);
    for (int i = 0; i < n; ++i) out[i] = s->variables[v[i]];
    return n;
}
FX_CATCH_CODE

int fxs_point_update_value(fxs_system* s, uint32_t element, double x, double y) try {  // elements/mod.rs:560-568
    if (!s || !is_tag(s, element, FXS_POINT)) return FX_ERR_INVALID;
    s->variables[s->elements[element].a] = x;
    s->variables[s->elements[element].a + 1] = y;
    return FX_OK;
}
FX_CATCH_CODE

int fxs_length_update_value(fxs_system* s, uint32_t element, double length) try {  // elements/mod.rs:572-578
    if (!s || !is_tag(s, element, FXS_LENGTH)) return FX_ERR_INVALID;
    s->variables[s->elements[element].a] = length;
    return FX_OK;
}
FX_CATCH_CODE

int fxs_constraint_valency(int tag) { return (tag < 0 || tag > FXS_LINE_CIRCLE_TANGENCY) ? FX_ERR_INVALID : valency_of(tag); }

// constraints::*::create, constraints/mod.rs:317-891: graph.add_constraint(valency, incident
// primitive elements) then System::add_constraint(tag, expressions) (lib.rs:412-445).
int64_t fxs_constraint_create(fxs_system* s, int tag, const uint32_t* el, uint32_t n, double param) try {
    if (!s || !el) return FX_ERR_INVALID;
    static const int kArgs[11][4] = {
        {FXS_POINT, FXS_POINT, -1, -1},                 // PointPointCoincidence
        {FXS_POINT, FXS_POINT, -1, -1},                 // PointPointDistance
        {FXS_POINT, FXS_POINT, FXS_POINT, -1},          // PointPointPointAngle
        {FXS_POINT, FXS_LINE, -1, -1},                  // PointLineIncidence
        {FXS_POINT, FXS_LINE, -1, -1},                  // PointLineDistance
        {FXS_POINT, FXS_CIRCLE, -1, -1},                // PointCircleIncidence
        {FXS_POINT, FXS_POINT, FXS_POINT, FXS_POINT},   // SegmentSegmentLengthEquality
        {FXS_LINE, FXS_LINE, -1, -1},                   // LineLineAngle
        {FXS_LINE, FXS_LINE, -1, -1},                   // LineLineParallelism
        {FXS_LINE, FXS_LINE, -1, -1},                   // LineLinePerpendicularity
        {FXS_LINE, FXS_CIRCLE, -1, -1},                 // LineCircleTangency
    };
    if (tag < 0 || tag > FXS_LINE_CIRCLE_TANGENCY) return FX_ERR_INVALID;
    uint32_t want = 0;
    while (want < 4 && kArgs[tag][want] >= 0) ++want;
    if (n != want) return bfail(FX_ERR_INVALID, "wrong number of elements");
    for (uint32_t i = 0; i < n; ++i) {
        if (!is_tag(s, el[i], kArgs[tag][i])) return bfail(FX_ERR_INVALID, "wrong element type");
    }

    // flatten the handles to variable indices and list the incident primitive elements
    uint32_t vidx[8];   // element fields in expression order
    uint32_t inc[6];    // incident primitives
    int nf = 0, ni = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const Element& e = s->elements[el[i]];
        switch (e.tag) {
            case FXS_POINT:
                vidx[nf++] = e.a;
                // points are passed as their own element id; SegmentSegmentLengthEquality goes
                // through variable_to_primitive, which is the same id (constraints/mod.rs:650-657)
                inc[ni++] = s->variable_to_primitive[e.a];
                break;
            case FXS_LINE:
                vidx[nf++] = e.a;
                vidx[nf++] = e.b;
                inc[ni++] = s->variable_to_primitive[e.a];
                inc[ni++] = s->variable_to_primitive[e.b];
                break;
            case FXS_CIRCLE:
                vidx[nf++] = e.a;
                vidx[nf++] = e.b;
                inc[ni++] = s->variable_to_primitive[e.a];
                inc[ni++] = s->variable_to_primitive[e.b];
                break;
            default:
                return FX_ERR_INVALID;
        }
    }

    uint32_t cid = (uint32_t)s->constraints.size();
    make_room(s->constraints, 1);  // (everything that can fail comes first: the call happens entirely or not at all)
    make_room(s->expressions, 2);
    merge_components(s, cid, inc, ni);  // graph.rs:235-254
    uint32_t expressions_idx = (uint32_t)s->expressions.size();
    Constraint con{tag, expressions_idx, (uint8_t)ni, {0, 0, 0, 0, 0, 0}};
    for (int q = 0; q < ni; ++q) con.incident[q] = inc[q];
    s->constraints.push_back(con);

    auto push = [&](uint8_t etag, uint32_t i0, uint32_t i1, uint32_t i2, uint32_t i3, double p) {
        Expr x;
        x.tag = etag;
        x.idx[0] = i0; x.idx[1] = i1; x.idx[2] = i2; x.idx[3] = i3;
        x.param = p;
        s->expressions.push_back(x);
    };
    switch (tag) {
        case FXS_POINT_POINT_COINCIDENCE:  // constraints/mod.rs:335-351: x then y equality
            push(FX_VARIABLE_VARIABLE_EQUALITY, vidx[0], vidx[1], 0, 0, 0.);
            push(FX_VARIABLE_VARIABLE_EQUALITY, vidx[0] + 1, vidx[1] + 1, 0, 0, 0.);
            break;
        case FXS_POINT_POINT_DISTANCE: push(FX_POINT_POINT_DISTANCE, vidx[0], vidx[1], 0, 0, param); break;
        case FXS_POINT_POINT_POINT_ANGLE: push(FX_POINT_POINT_POINT_ANGLE, vidx[0], vidx[1], vidx[2], 0, param); break;
        case FXS_POINT_LINE_INCIDENCE: push(FX_POINT_LINE_INCIDENCE, vidx[0], vidx[1], vidx[2], 0, 0.); break;
        case FXS_POINT_LINE_DISTANCE: push(FX_POINT_LINE_DISTANCE, vidx[0], vidx[1], vidx[2], 0, param); break;
        case FXS_POINT_CIRCLE_INCIDENCE: push(FX_POINT_CIRCLE_INCIDENCE, vidx[0], vidx[1], vidx[2], 0, 0.); break;
        case FXS_SEGMENT_SEGMENT_LENGTH_EQUALITY:
            push(FX_SEGMENT_SEGMENT_LENGTH_EQUALITY, vidx[0], vidx[1], vidx[2], vidx[3], 0.);
            break;
        case FXS_LINE_LINE_ANGLE: push(FX_LINE_LINE_ANGLE, vidx[0], vidx[1], vidx[2], vidx[3], param); break;
        case FXS_LINE_LINE_PARALLELISM: push(FX_LINE_LINE_PARALLELISM, vidx[0], vidx[1], vidx[2], vidx[3], 0.); break;
        case FXS_LINE_LINE_PERPENDICULARITY: push(FX_LINE_LINE_PERPENDICULARITY, vidx[0], vidx[1], vidx[2], vidx[3], 0.); break;
        case FXS_LINE_CIRCLE_TANGENCY: push(FX_LINE_CIRCLE_TANGENCY, vidx[0], vidx[1], vidx[2], vidx[3], 0.); break;
    }
    return cid;
}
FX_CATCH_CODE

int fxs_constraint_tag_of(const fxs_system* s, uint32_t constraint) try {
    if (!s || constraint >= s->constraints.size()) return FX_ERR_INVALID;
    return s->constraints[constraint].tag;
}
FX_CATCH_CODE

// ConstraintHandle::update_parameter, constraints/mod.rs:992-1046 (the four parameterised kinds)
int fxs_constraint_update_parameter(fxs_system* s, uint32_t constraint, double value) try {
    if (!s || constraint >= s->constraints.size()) return FX_ERR_INVALID;
    const Constraint& c = s->constraints[constraint];
    switch (c.tag) {
        case FXS_POINT_POINT_DISTANCE:
        case FXS_POINT_POINT_POINT_ANGLE:
        case FXS_POINT_LINE_DISTANCE:
        case FXS_LINE_LINE_ANGLE:
            s->expressions[c.expressions_idx].param = value;
            return FX_OK;
        default:
            return FX_ERR_INVALID;
    }
}
FX_CATCH_CODE

int fxs_components(const fxs_system* s, uint32_t* n_components, uint16_t* element_comp, uint16_t* constraint_comp) try {
    if (!s) return FX_ERR_INVALID;
    std::vector<const Component*> live;
    live_components(s, live);
    if (n_components) *n_components = (uint32_t)live.size();
    if (element_comp) {
        for (size_t i = 0; i < s->elements.size(); ++i) element_comp[i] = (uint16_t)FX_NO_COMPONENT;
    }
    if (constraint_comp) {
        for (size_t i = 0; i < s->constraints.size(); ++i) constraint_comp[i] = (uint16_t)FX_NO_COMPONENT;
    }
    for (size_t c = 0; c < live.size(); ++c) {
        if (element_comp) for (uint32_t e : live[c]->elements) element_comp[e] = (uint16_t)c;
        if (constraint_comp) for (uint32_t k : live[c]->constraints) constraint_comp[k] = (uint16_t)c;
    }
    return FX_OK;
}
FX_CATCH_CODE

int fxs_export_graph(const fxs_system* s, uint8_t* element_kind, uint32_t* element_idx, uint8_t* constraint_valency,
                     uint32_t* constraint_expr, uint8_t* constraint_n_incident, uint32_t* constraint_incident) try {
    if (!s) return FX_ERR_INVALID;
    for (size_t i = 0; i < s->elements.size(); ++i) {
        if (element_kind) element_kind[i] = (uint8_t)s->elements[i].tag;
        if (element_idx) element_idx[i] = s->elements[i].a;
    }
    for (size_t c = 0; c < s->constraints.size(); ++c) {
        const Constraint& con = s->constraints[c];
        if (constraint_valency) constraint_valency[c] = (uint8_t)valency_of(con.tag);
        if (constraint_expr) constraint_expr[c] = con.expressions_idx;
        if (constraint_n_incident) constraint_n_incident[c] = con.n_incident;
        if (constraint_incident)
            for (int q = 0; q < 6; ++q) constraint_incident[6 * c + q] = con.incident[q];
    }
    return FX_OK;
}
FX_CATCH_CODE

int fxs_recursive_plan(const fxs_system* s, uint64_t budget, uint32_t* out, uint32_t capacity, uint32_t* length, uint32_t* flags) try {
    if (!s || !length) return FX_ERR_INVALID;
    std::vector<const Component*> live;
    live_components(s, live);
    const fx::ra::Graph graph = plan_graph(s);
    std::vector<uint32_t> words;
    uint32_t fl = 0;
    for (const Component* comp : live) {
        std::vector<uint32_t> els(comp->elements.begin(), comp->elements.end());
        std::vector<uint32_t> cons(comp->constraints.begin(), comp->constraints.end());
        const fx::ra::Plan plan = fx::ra::make_plan(graph, els, cons, budget ? budget : kDefaultPlanBudget);
        fx::ra::serialise(plan, words);
        fl |= (plan.panicked ? 1u : 0u) | (plan.exhausted ? 2u : 0u);
        if (fl) break;
    }
    *length = (uint32_t)words.size();
    if (flags) *flags = fl;
    if (out)
        for (size_t i = 0; i < words.size() && i < capacity; ++i) out[i] = words[i];
    return FX_OK;
}
FX_CATCH_CODE

int fxs_flatten(const fxs_system* const* systems, uint32_t n, fxs_flat** out) try {
    if (!out || (n && !systems)) return FX_ERR_INVALID;
    std::unique_ptr<fxs_flat> hold(new (std::nothrow) fxs_flat());  // (freed on every early way out, by code or by exception)
    fxs_flat* f = hold.get();
    if (!f) return FX_ERR_NOMEM;
    f->var_off.push_back(0);
    f->expr_off.push_back(0);
    std::vector<const Component*> live;
    for (uint32_t k = 0; k < n; ++k) {
        const fxs_system* s = systems[k];
        if (!s) return FX_ERR_INVALID;
        size_t v0 = f->vars.size(), e0 = f->expr_tag.size();
        f->vars.insert(f->vars.end(), s->variables.begin(), s->variables.end());
        f->var_fixed.resize(v0 + s->variables.size(), 0);
        for (uint32_t v : s->fixed_variables) f->var_fixed[v0 + v] = 1;
        f->var_comp.resize(v0 + s->variables.size(), (uint16_t)FX_NO_COMPONENT);
        f->expr_comp.resize(e0 + s->expressions.size(), (uint16_t)FX_NO_COMPONENT);
        for (const Expr& x : s->expressions) {
            f->expr_tag.push_back(x.tag);
            for (int q = 0; q < 4; ++q) f->expr_idx.push_back(x.idx[q]);
            f->expr_param.push_back(x.param);
        }
        // assemble/mod.rs:91-111: a component's variables are those of its elements;
        // :136-145: its rows are the expressions of its constraints.
        live_components(s, live);
        if (live.size() >= 0x7FFF) return FX_ERR_TOO_LARGE;
        for (size_t c = 0; c < live.size(); ++c) {
            for (uint32_t e : live[c]->elements) {
                uint32_t v[4];
                int nv = element_variables(s, e, v);
                for (int q = 0; q < nv; ++q) f->var_comp[v0 + v[q]] = (uint16_t)c;
            }
            for (uint32_t ci : live[c]->constraints) {
                const Constraint& con = s->constraints[ci];
                for (int q = 0; q < valency_of(con.tag); ++q) f->expr_comp[e0 + con.expressions_idx + q] = (uint16_t)c;
            }
        }
        f->var_off.push_back((uint32_t)f->vars.size());
        f->expr_off.push_back((uint32_t)f->expr_tag.size());
    }
    f->batch.n_systems = n;
    f->batch.var_off = f->var_off.data();
    f->batch.expr_off = f->expr_off.data();
    f->batch.vars = f->vars.data();
    f->batch.var_fixed = f->var_fixed.data();
    f->batch.expr_tag = f->expr_tag.data();
    f->batch.expr_idx = f->expr_idx.data();
    f->batch.expr_param = f->expr_param.data();
    f->batch.var_comp = f->var_comp.data();
    f->batch.expr_comp = f->expr_comp.data();
    *out = hold.release();
    return FX_OK;
}
FX_CATCH_CODE

const fx_batch* fxs_flat_batch(const fxs_flat* f) { return f ? &f->batch : nullptr; }
void fxs_flat_free(fxs_flat* f) try { delete f; } FX_CATCH_VOID

int fxs_flat_scatter(const fxs_flat* f, fxs_system* const* systems, uint32_t n) try {
    if (!f || (n && !systems) || n != f->batch.n_systems) return FX_ERR_INVALID;
    for (uint32_t k = 0; k < n; ++k) {
        fxs_system* s = systems[k];
        uint32_t v0 = f->var_off[k], nv = f->var_off[k + 1] - v0;
        if (!s || nv != s->variables.size()) return FX_ERR_INVALID;
        std::memcpy(s->variables.data(), f->vars.data() + v0, nv * sizeof(double));
    }
    return FX_OK;
}
FX_CATCH_CODE

}  // extern "C"

namespace {

// assemble::solve, Decomposer::RecursiveAssembly (assemble/mod.rs:212-277) for Systems that share one STRUCTURE (elements,
// constraints, components — values and parameters differ: one sketch, many parameter sets). Host: the plan — made once,
// it reads structure only — and the make-up of each cluster problem. Device: scale + perturbation of every System
// (fx_system_prepare_batch), step k of EVERY System in one cluster solve (fx_cluster_solve_batch on a batch of n cluster
// problems), the rigid moves (one fx_pose_transform_points), the un-scaling (fx_unscale_vars_strided) and the closing
// residual check. Round 2 did this one System at a time: three device round trips per step and System.
bool same_structure(const fxs_system* x, const fxs_system* y) {
    if (x->elements.size() != y->elements.size() || x->constraints.size() != y->constraints.size() ||
        x->expressions.size() != y->expressions.size() || x->variables.size() != y->variables.size() ||
        x->element_component != y->element_component || x->components.size() != y->components.size() ||
        x->fixed_variables != y->fixed_variables)
        return false;
    for (size_t i = 0; i < x->elements.size(); ++i)
        if (x->elements[i].tag != y->elements[i].tag || x->elements[i].a != y->elements[i].a || x->elements[i].b != y->elements[i].b) return false;
    for (size_t i = 0; i < x->constraints.size(); ++i) {
        const Constraint &c1 = x->constraints[i], &c2 = y->constraints[i];
        if (c1.tag != c2.tag || c1.expressions_idx != c2.expressions_idx || c1.n_incident != c2.n_incident ||
            std::memcmp(c1.incident, c2.incident, sizeof(uint32_t) * c1.n_incident) != 0)
            return false;
    }
    for (size_t i = 0; i < x->expressions.size(); ++i)
        if (x->expressions[i].tag != y->expressions[i].tag || std::memcmp(x->expressions[i].idx, y->expressions[i].idx, sizeof(x->expressions[i].idx)) != 0)
            return false;
    for (size_t i = 0; i < x->components.size(); ++i)
        if (x->components[i].elements != y->components[i].elements || x->components[i].constraints != y->components[i].constraints) return false;
    return true;
}

uint64_t structure_hash(const fxs_system* x) {
    uint64_t h = 0xcbf29ce484222325ull;
    auto mix = [&](uint64_t v) { h = (h ^ v) * 0x100000001b3ull; };
    mix(x->elements.size()); mix(x->constraints.size()); mix(x->expressions.size()); mix(x->variables.size());
    for (const Element& e : x->elements) mix(((uint64_t)e.tag << 48) ^ ((uint64_t)e.a << 24) ^ e.b);
    for (const Expr& e : x->expressions) mix(((uint64_t)e.tag << 56) ^ ((uint64_t)e.idx[0] << 42) ^ ((uint64_t)e.idx[1] << 28) ^ ((uint64_t)e.idx[2] << 14) ^ e.idx[3]);
    return h;
}

int solve_recursive_assembly(fxs_system* const* systems, uint32_t n_sys, fx_ctx* ctx, const fx_solving_opts* opts, fx_result* results) {
    fx_solving_opts o;
    if (opts) o = *opts; else fx_solving_opts_default(&o);
    // the arm always runs Levenberg-Marquardt (assemble/mod.rs:224-227 ignores opts.optimizer)
    fxs_system* s = systems[0];  // the structure
    std::vector<fx_result> total(n_sys);
    for (fx_result& t : total) {
        t = fx_result{};
        t.exit = FX_EXIT_SSE;
    }
    if (s->variables.empty()) {
        if (results) std::copy(total.begin(), total.end(), results);
        return FX_OK;
    }
    // FIKSI_AMD_TRACE=1: where the wall time of the arm goes, one line per phase on stderr
    const bool trace = std::getenv("FIKSI_AMD_TRACE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto stamp = [&](const char* what) {
        if (!trace) return;
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[fiksi_amd] recursive assembly, %u Systems: %-28s %8.3f ms\n", n_sys, what,
                     std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    fxs_flat* f = nullptr;
    int rc = fxs_flatten(systems, n_sys, &f);
    if (rc) return rc;
    struct Guard { fxs_flat* f; ~Guard() { fxs_flat_free(f); } } guard{f};
    stamp("flatten");

    const uint32_t nvars = (uint32_t)s->variables.size(), nexprs = (uint32_t)s->expressions.size();
    std::vector<double> vt((size_t)n_sys * nvars), params((size_t)n_sys * nexprs + 1), scales(n_sys, 1.);
    rc = fx_system_prepare_batch(ctx, &f->batch, o.perturb, vt.data(), params.data(), scales.data());
    if (rc) return rc;
    for (uint32_t k = 0; k < n_sys; ++k) total[k].scale = scales[k];
    std::vector<uint8_t> touched(nvars, 0);
    stamp("prepare (device)");

    const fx::ra::Graph graph = plan_graph(s);
    auto is_point = [&](uint32_t e) { return s->elements[e].tag == FXS_POINT; };

    std::vector<const Component*> live;
    live_components(s, live);
    std::vector<int32_t> local(nvars, -1);
    std::vector<fx_result> step_results(n_sys);
    for (const Component* comp : live) {
        std::vector<uint32_t> els(comp->elements.begin(), comp->elements.end());
        std::vector<uint32_t> cons(comp->constraints.begin(), comp->constraints.end());
        const fx::ra::Plan plan = fx::ra::make_plan(graph, els, cons, o.plan_budget ? (uint64_t)o.plan_budget * 1000u : kDefaultPlanBudget);
        if (plan.panicked)
            return bfail(FX_ERR_UNSUPPORTED, "RecursiveAssembly: the reference's planner would panic on this sketch (recursive_assembly.rs:357-375); System untouched");
        if (plan.exhausted) return bfail(FX_ERR_UNSUPPORTED, "RecursiveAssembly: the plan search ran out of its budget (fx_solving_opts.plan_budget); System untouched");

        for (const fx::ra::Step& step : plan.steps) {
            const fx::ra::ClusterProblem cp = fx::ra::make_cluster_problem(step, is_point);
            if (cp.panicked)
                return bfail(FX_ERR_UNSUPPORTED, "RecursiveAssembly: a step's expression names a variable hidden inside a contracted cluster (assemble/mod.rs:505-509); System untouched");

            // ---- the cluster problem's structure (one for all Systems). Unknowns: the poses, then the members' variables
            // (assemble/mod.rs:431-474); constants: the points as solved so far, one pair per pose row pair. `src[i]`: the
            // System variable the start value of entry i comes from (-1: a pose, starts at 0).
            std::vector<int32_t> src(3 * cp.clusters.size(), -1);
            std::fill(local.begin(), local.end(), -1);
            for (uint32_t e : cp.members) {
                const Element& el = s->elements[e];
                const int nn = el.tag == FXS_LENGTH ? 1 : el.tag == FXS_POINT ? 2 : 0;
                for (int q = 0; q < nn; ++q) {
                    local[el.a + q] = (int32_t)src.size();
                    src.push_back((int32_t)(el.a + q));
                }
            }
            const uint32_t n_unknown = (uint32_t)src.size();
            std::vector<uint8_t> tags;
            std::vector<uint32_t> idx;
            std::vector<int32_t> prm_src;  // expression the parameter comes from (-1: none, 0)
            for (size_t ci = 0; ci < cp.clusters.size(); ++ci) {  // pose rows come first (:547-588)
                for (uint32_t point : cp.clusters[ci].second) {
                    const uint32_t g = s->elements[point].a;
                    if (local[g] < 0) return bfail(FX_ERR_UNSUPPORTED, "RecursiveAssembly: a cluster row names a variable outside its cluster problem; System untouched");
                    const uint32_t was = (uint32_t)src.size();
                    src.push_back((int32_t)g);
                    src.push_back((int32_t)g + 1);
                    for (int q = 0; q < 2; ++q) {
                        tags.push_back((uint8_t)(FX_POSE_COINCIDENCE_X + q));
                        idx.push_back((uint32_t)(3 * ci));
                        idx.push_back(was);
                        idx.push_back((uint32_t)local[g] + (uint32_t)q);
                        idx.push_back(0);
                        prm_src.push_back(-1);
                    }
                }
            }
            for (uint32_t c : step.constraints) {  // :346-351
                const Constraint& con = s->constraints[c];
                for (int q = 0; q < valency_of(con.tag); ++q) {
                    const Expr& ex = s->expressions[con.expressions_idx + q];
                    tags.push_back(ex.tag);
                    for (int k = 0; k < 4; ++k) {
                        uint32_t m = 0;
                        if (k < expr_fields(ex.tag)) {
                            if (local[ex.idx[k]] < 0) return bfail(FX_ERR_UNSUPPORTED, "RecursiveAssembly: a step's expression names a variable hidden inside a contracted cluster; System untouched");  // the reference panics (`.unwrap()`, :505-509)
                            m = (uint32_t)local[ex.idx[k]];
                        }
                        idx.push_back(m);
                    }
                    prm_src.push_back((int32_t)(con.expressions_idx + q));
                }
            }
            // (a System's entries start at a multiple of three, so that its poses are whole entries of the pose table the
            // rigid moves index: up to two idle fixed entries at the end)
            while (src.size() % 3) src.push_back(-2);
            const uint32_t xs = (uint32_t)src.size(), nrows = (uint32_t)tags.size();

            // ---- the batch: n_sys cluster problems of this structure, each System's values
            std::vector<double> x((size_t)n_sys * xs), prm((size_t)n_sys * nrows);
            std::vector<uint8_t> fixed((size_t)n_sys * xs), tags_all((size_t)n_sys * nrows);
            std::vector<uint32_t> idx_all(4 * (size_t)n_sys * nrows), var_off(n_sys + 1), expr_off(n_sys + 1);
            for (uint32_t k = 0; k < n_sys; ++k) {
                var_off[k] = k * xs;
                expr_off[k] = k * nrows;
                const double* v = vt.data() + (size_t)k * nvars;
                for (uint32_t i = 0; i < xs; ++i) {
                    x[(size_t)k * xs + i] = src[i] >= 0 ? v[src[i]] : 0.;
                    fixed[(size_t)k * xs + i] = i >= n_unknown ? 1 : 0;
                }
                for (uint32_t r = 0; r < nrows; ++r) prm[(size_t)k * nrows + r] = prm_src[r] >= 0 ? params[(size_t)k * nexprs + (uint32_t)prm_src[r]] : 0.;
                std::copy(tags.begin(), tags.end(), tags_all.begin() + (size_t)k * nrows);
                std::copy(idx.begin(), idx.end(), idx_all.begin() + 4 * (size_t)k * nrows);
            }
            var_off[n_sys] = n_sys * xs;
            expr_off[n_sys] = n_sys * nrows;
            fx_batch b{};
            b.n_systems = n_sys;
            b.var_off = var_off.data();
            b.expr_off = expr_off.data();
            b.vars = x.data();
            b.var_fixed = fixed.data();
            b.expr_tag = tags_all.data();
            b.expr_idx = idx_all.data();
            b.expr_param = prm.data();
            stamp("step: host make-up");
            rc = fx_cluster_solve_batch(ctx, &b, &o.lm, step_results.data());
            if (rc) return rc;
            stamp("step: cluster solve (device)");
            for (uint32_t k = 0; k < n_sys; ++k) {
                const fx_result& r = step_results[k];
                total[k].accepted += r.accepted;
                total[k].trials += r.trials;
                total[k].exit = r.exit;
                total[k].ncomp += 1;
                total[k].sse0 += r.sse0;
                total[k].sse += r.sse;
                double* v = vt.data() + (size_t)k * nvars;
                for (uint32_t g = 0; g < nvars; ++g)  // :228-236
                    if (local[g] >= 0) v[g] = x[(size_t)k * xs + (size_t)local[g]];
            }
            for (uint32_t g = 0; g < nvars; ++g)
                if (local[g] >= 0) touched[g] = 1;
            // :238-275 what a moved cluster carries along — every System's points in one call: pose ci of System k is entry
            // k * xs / 3 + ci of the pose table (= x itself), its variables start at k * nvars
            std::vector<uint32_t> pose_of, point_var;
            for (size_t ci = 0; ci < cp.clusters.size(); ++ci) {
                const std::vector<uint32_t>* carried = fx::ra::lookup(step.owned_elements, cp.clusters[ci].first);
                if (!carried) continue;
                for (uint32_t e : *carried) {
                    if (!is_point(e) || std::find(cp.members.begin(), cp.members.end(), e) != cp.members.end()) continue;
                    pose_of.push_back((uint32_t)ci);
                    point_var.push_back(s->elements[e].a);
                    touched[s->elements[e].a] = touched[s->elements[e].a + 1] = 1;
                }
            }
            if (!pose_of.empty()) {
                const size_t np = pose_of.size();
                std::vector<uint32_t> pose_all(np * n_sys), var_all(np * n_sys);
                for (uint32_t k = 0; k < n_sys; ++k)
                    for (size_t i = 0; i < np; ++i) {
                        pose_all[(size_t)k * np + i] = k * (xs / 3u) + pose_of[i];
                        var_all[(size_t)k * np + i] = k * nvars + point_var[i];
                    }
                rc = fx_pose_transform_points(ctx, x.data(), n_sys * (xs / 3u), pose_all.data(), var_all.data(), (uint32_t)(np * n_sys), vt.data(),
                                              n_sys * nvars);
                if (rc) return rc;
            }
            stamp("step: carry results, moves");
        }
    }

    std::vector<double> solved((size_t)n_sys * nvars);
    for (uint32_t k = 0; k < n_sys; ++k) std::copy(systems[k]->variables.begin(), systems[k]->variables.end(), solved.begin() + (size_t)k * nvars);
    rc = fx_unscale_vars_strided(ctx, scales.data(), n_sys, nvars, vt.data(), touched.data(), solved.data());
    if (rc) return rc;
    stamp("unscale (device)");
    // the closing check: sum of squared expression residuals on the solved variables, from the device
    // (no LM step is taken: max_outer = 0)
    std::memcpy(f->vars.data(), solved.data(), solved.size() * sizeof(double));
    fx_lm_opts probe;
    fx_lm_opts_default(&probe);
    probe.max_outer = 0;
    std::vector<fx_result> pr(n_sys);
    rc = fx_lm_solve_batch(ctx, &f->batch, &probe, pr.data());
    if (rc) return rc;
    for (uint32_t k = 0; k < n_sys; ++k) {
        std::copy(solved.begin() + (size_t)k * nvars, solved.begin() + (size_t)(k + 1) * nvars, systems[k]->variables.begin());
        total[k].sse_unscaled = pr[k].sse_unscaled;
    }
    stamp("closing check (device)");
    if (results) std::copy(total.begin(), total.end(), results);
    return FX_OK;
}

}  // namespace

extern "C" {

int fxs_systems_solve(fxs_system* const* systems, uint32_t n, fx_ctx* ctx, const fx_solving_opts* opts,
                      fx_result* results) try {
    if (opts && opts->decomposer == 2) {
        // RecursiveAssembly plans from a System's elements; Systems of one structure (one sketch, many parameter sets)
        // share the plan and every device call — groups in order of their first System
        if (n && !systems) return FX_ERR_INVALID;
        for (uint32_t k = 0; k < n; ++k)
            if (!systems[k]) return FX_ERR_INVALID;
        // (a hash of the structure finds the candidates, same_structure() decides; a System listed several times is solved
        // once per listing, each starting from the result before: its j-th listing joins a group of j-th listings only, so
        // no group ever holds a System twice — and the groups run in the order they were opened)
        struct Group {
            uint64_t hash;
            uint32_t listing;  // the group holds Systems listed for the (listing + 1)-th time
            std::vector<fxs_system*> systems;
            std::vector<uint32_t> members;
        };
        std::vector<Group> groups;
        std::unordered_multimap<uint64_t, size_t> by_hash;
        for (uint32_t k = 0; k < n; ++k) systems[k]->listed = 0;
        for (uint32_t k = 0; k < n; ++k) {
            const uint64_t h = structure_hash(systems[k]);
            size_t at = groups.size();
            auto range = by_hash.equal_range(h);
            for (auto it = range.first; it != range.second; ++it) {
                if (groups[it->second].listing == systems[k]->listed && same_structure(groups[it->second].systems[0], systems[k])) {
                    at = it->second;
                    break;
                }
            }
            if (at == groups.size()) {
                groups.push_back(Group{h, systems[k]->listed, {}, {}});
                by_hash.emplace(h, at);
            }
            groups[at].systems.push_back(systems[k]);
            groups[at].members.push_back(k);
            systems[k]->listed += 1u;
        }
        for (Group& g : groups) {
            std::vector<fx_result> r(g.systems.size());
            int rc = solve_recursive_assembly(g.systems.data(), (uint32_t)g.systems.size(), ctx, opts, r.data());
            if (rc) return rc;
            if (results)
                for (size_t i = 0; i < g.members.size(); ++i) results[g.members[i]] = r[i];
        }
        return FX_OK;
    }
    fxs_flat* f = nullptr;
    int rc = fxs_flatten(systems, n, &f);
    if (rc) return rc;
    FlatGuard guard{f};
    rc = fx_system_solve_batch(ctx, &f->batch, opts, results);
    if (!rc) rc = fxs_flat_scatter(f, systems, n);
    return rc;
}
FX_CATCH_CODE

int fxs_system_solve(fxs_system* s, fx_ctx* ctx, const fx_solving_opts* opts, fx_result* result) try {
    fxs_system* one[1] = {s};
    return fxs_systems_solve(one, 1, ctx, opts, result);
}
FX_CATCH_CODE

int fxs_system_constraint_residuals(const fxs_system* s, fx_ctx* ctx, double* out) try {
    if (!s || !out) return FX_ERR_INVALID;
    const fxs_system* one[1] = {s};
    fxs_flat* f = nullptr;
    int rc = fxs_flatten(one, 1, &f);
    if (rc) return rc;
    FlatGuard guard{f};
    std::vector<double> r(s->expressions.size() + 1, 0.);
    rc = fx_constraint_residuals(ctx, &f->batch, r.data());
    if (rc) return rc;
    for (size_t c = 0; c < s->constraints.size(); ++c) {
        const Constraint& con = s->constraints[c];
        if (valency_of(con.tag) > 1) {  // constraints/mod.rs:99-105
            double sum = 0.;
            for (int q = 0; q < valency_of(con.tag); ++q) {
                double v = r[con.expressions_idx + q];
                sum += v * v;
            }
            out[c] = std::sqrt(sum);
        } else {
            out[c] = r[con.expressions_idx];
        }
    }
    return FX_OK;
}
FX_CATCH_CODE

int fxs_system_analyze(const fxs_system* s, fx_ctx* ctx, uint32_t* ids, uint32_t* n) try {
    if (!s || !ids || !n) return FX_ERR_INVALID;
    const fxs_system* one[1] = {s};
    fxs_flat* f = nullptr;
    int rc = fxs_flatten(one, 1, &f);
    if (rc) return rc;
    FlatGuard guard{f};
    std::vector<uint8_t> dep(s->expressions.size() + 1, 0);
    rc = fx_analyze_batch(ctx, &f->batch, dep.data());
    if (rc) return rc;
    // expression -> constraint (System::expression_to_constraint, lib.rs:301-302)
    uint32_t count = 0;
    for (size_t c = 0; c < s->constraints.size(); ++c) {
        const Constraint& con = s->constraints[c];
        for (int q = 0; q < valency_of(con.tag); ++q)
            if (dep[con.expressions_idx + q]) ids[count++] = (uint32_t)c;
    }
    *n = count;
    return FX_OK;
}
FX_CATCH_CODE

}  // extern "C"
