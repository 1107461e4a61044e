// fiksi.hpp — header-only C++ mirror of the reference crate's public builder API over the C ABI
// (fiksi_amd.h + fiksi_amd_builder.h). Names and argument order follow the Rust API:
//
//   fiksi::System                                  fiksi/src/lib.rs:256-466
//   fiksi::elements::{Length,Point,Line,Circle}    fiksi/src/elements/mod.rs:280,321,365,437
//   fiksi::constraints::*::create                  fiksi/src/constraints/mod.rs:317-891
//   fiksi::SolvingOptions, Decomposer, Optimizer   fiksi/src/lib.rs:154-237, solve/mod.rs:17-27
//
// Where the reference panics (handle of another System, lib.rs `assert_eq!`) this mirror throws
// std::logic_error; a failing device call throws fiksi::Error (the reference has no error channel
// on solve()). The handle types carry the element kind, so the misuse Rust rejects at compile time
// is rejected at compile time here too.
#pragma once
#include <array>
#include <stdexcept>
#include <string>
#include <vector>

#include "fiksi_amd.h"
#include "fiksi_amd_builder.h"

namespace fiksi {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& where)
        : std::runtime_error(where + " failed with fx_status " + std::to_string(c) + ": " + fx_last_error()), code(c) {}
};

enum class Optimizer { LevenbergMarquardt = 0, LBfgs = 1 };
enum class Decomposer { None = 0, SinglePass = 1, RecursiveAssembly = 2 };

struct SolvingOptions {  // lib.rs:205-237
    Optimizer optimizer = Optimizer::LevenbergMarquardt;
    Decomposer decomposer = Decomposer::None;
    bool perturb = true;
};

// One device context shared by System::solve when none is given (device 0, created on first use).
inline fx_ctx* default_context() {
    static fx_ctx* ctx = [] {
        fx_ctx* c = nullptr;
        int rc = fx_ctx_create(&c, 0);
        if (rc) throw Error(rc, "fx_ctx_create");
        return c;
    }();
    return ctx;
}

class System;

namespace elements {
struct Length {};
struct Point {};
struct Line {};
struct Circle {};
}  // namespace elements

template <class T>
struct ElementHandle {  // elements/mod.rs:26-33
    uint32_t system_id, id;
};
struct PointValue { double x, y; };
struct LineValue { PointValue p0, p1; };
struct CircleValue { PointValue center; double radius; };

template <class T>
struct ConstraintHandle {  // constraints/mod.rs:62-69
    uint32_t system_id, id;
};

// AnyElementHandle / AnyConstraintHandle (elements/mod.rs:114-245, constraints/mod.rs:170-315): a handle
// whose kind is only known at run time.
struct AnyElementHandle {
    uint32_t system_id, id;
    fxs_element_tag tag;
};
struct AnyConstraintHandle {
    uint32_t system_id, id;
    fxs_constraint_tag tag;
};
// System::analyze's result (analyze/mod.rs): the constraints found to over-constrain the System.
struct Analysis {
    std::vector<AnyConstraintHandle> overconstrained;
};

class System {
  public:
    System() {
        int rc = fxs_system_new(&h_);
        if (rc) throw Error(rc, "System::new");
    }
    ~System() { fxs_system_free(h_); }
    System(const System&) = delete;
    System& operator=(const System&) = delete;

    uint32_t id() const { return fxs_system_id(h_); }
    fxs_system* raw() { return h_; }
    const fxs_system* raw() const { return h_; }

    // lib.rs:464-466
    void solve(const SolvingOptions& opts = SolvingOptions(), fx_ctx* ctx = nullptr) {
        fx_solving_opts o;
        fx_solving_opts_default(&o);
        o.optimizer = static_cast<uint32_t>(opts.optimizer);
        o.decomposer = static_cast<uint32_t>(opts.decomposer);
        o.perturb = opts.perturb ? 1u : 0u;
        int rc = fxs_system_solve(h_, ctx ? ctx : default_context(), &o, &last_result);
        if (rc) throw Error(rc, "System::solve");
    }

    // lib.rs:331-345, :347-361
    std::vector<AnyElementHandle> get_element_handles() const {
        std::vector<AnyElementHandle> out;
        for (uint32_t i = 0, n = fxs_num_elements(h_); i < n; ++i)
            out.push_back({id(), i, static_cast<fxs_element_tag>(fxs_element_tag_of(h_, i))});
        return out;
    }
    std::vector<AnyConstraintHandle> get_constraint_handles() const {
        std::vector<AnyConstraintHandle> out;
        for (uint32_t i = 0, n = fxs_num_constraints(h_); i < n; ++i)
            out.push_back({id(), i, static_cast<fxs_constraint_tag>(fxs_constraint_tag_of(h_, i))});
        return out;
    }

    // lib.rs:454-459 (doc-hidden in the reference: all variables are treated as free)
    Analysis analyze(fx_ctx* ctx = nullptr) const {
        std::vector<uint32_t> ids(fxs_num_expressions(h_) + 1, 0u);
        uint32_t n = 0;
        int rc = fxs_system_analyze(h_, ctx ? ctx : default_context(), ids.data(), &n);
        if (rc) throw Error(rc, "System::analyze");
        Analysis a;
        for (uint32_t k = 0; k < n; ++k)
            a.overconstrained.push_back({id(), ids[k], static_cast<fxs_constraint_tag>(fxs_constraint_tag_of(h_, ids[k]))});
        return a;
    }

    std::vector<double> constraint_residuals(fx_ctx* ctx = nullptr) const {
        std::vector<double> r(fxs_num_constraints(h_) + 1, 0.0);
        int rc = fxs_system_constraint_residuals(h_, ctx ? ctx : default_context(), r.data());
        if (rc) throw Error(rc, "calculate_residual");
        r.pop_back();
        return r;
    }

    template <class T>
    void check(const ElementHandle<T>& e) const {
        if (e.system_id != id()) throw std::logic_error("Tried to get an element that is not part of this `System`");
    }
    template <class T>
    void check(const ConstraintHandle<T>& c) const {
        if (c.system_id != id()) throw std::logic_error("Tried to evaluate a constraint that is not part of this `System`");
    }

    fx_result last_result{};

  private:
    fxs_system* h_ = nullptr;
};

// ElementHandle methods (elements/mod.rs:60-112, 560-579)
template <class T> void fix(const ElementHandle<T>& e, System& s) { fxs_element_fix(s.raw(), e.id); }
template <class T> void unfix(const ElementHandle<T>& e, System& s) { fxs_element_unfix(s.raw(), e.id); }
inline double get_value(const ElementHandle<elements::Length>& e, const System& s) {
    s.check(e);
    double v[4];
    fxs_element_get_value(s.raw(), e.id, v);
    return v[0];
}
inline PointValue get_value(const ElementHandle<elements::Point>& e, const System& s) {
    s.check(e);
    double v[4];
    fxs_element_get_value(s.raw(), e.id, v);
    return {v[0], v[1]};
}
inline LineValue get_value(const ElementHandle<elements::Line>& e, const System& s) {
    s.check(e);
    double v[4];
    fxs_element_get_value(s.raw(), e.id, v);
    return {{v[0], v[1]}, {v[2], v[3]}};
}
inline CircleValue get_value(const ElementHandle<elements::Circle>& e, const System& s) {
    s.check(e);
    double v[4];
    fxs_element_get_value(s.raw(), e.id, v);
    return {{v[0], v[1]}, v[2]};
}
inline void update_value(const ElementHandle<elements::Point>& e, System& s, double x, double y) {
    fxs_point_update_value(s.raw(), e.id, x, y);
}
inline void update_value(const ElementHandle<elements::Length>& e, System& s, double length) {
    fxs_length_update_value(s.raw(), e.id, length);
}

namespace detail {
template <class T>
ElementHandle<T> made(System& s, int64_t rc, const char* what) {
    if (rc < 0) throw Error(static_cast<int>(rc), what);
    return ElementHandle<T>{s.id(), static_cast<uint32_t>(rc)};
}
template <class C, size_t N>
ConstraintHandle<C> constraint(System& s, int tag, const std::array<uint32_t, N>& el, double param, const char* what) {
    int64_t rc = fxs_constraint_create(s.raw(), tag, el.data(), static_cast<uint32_t>(N), param);
    if (rc < 0) throw Error(static_cast<int>(rc), what);
    return ConstraintHandle<C>{s.id(), static_cast<uint32_t>(rc)};
}
}  // namespace detail

namespace elements {
inline ElementHandle<Length> create_length(System& s, double length) { return detail::made<Length>(s, fxs_length_create(s.raw(), length), "Length::create"); }
inline ElementHandle<Point> create_point(System& s, double x, double y) { return detail::made<Point>(s, fxs_point_create(s.raw(), x, y), "Point::create"); }
inline ElementHandle<Line> create_line(System& s, ElementHandle<Point> p1, ElementHandle<Point> p2) {
    return detail::made<Line>(s, fxs_line_create(s.raw(), p1.id, p2.id), "Line::create");
}
inline ElementHandle<Circle> create_circle(System& s, ElementHandle<Point> center, ElementHandle<Length> radius) {
    return detail::made<Circle>(s, fxs_circle_create(s.raw(), center.id, radius.id), "Circle::create");
}
}  // namespace elements

namespace constraints {
using elements::Circle;
using elements::Length;
using elements::Line;
using elements::Point;
#define FIKSI_CONSTRAINT(Name) struct Name { static constexpr int VALENCY = 1; }
struct PointPointCoincidence { static constexpr int VALENCY = 2; };
FIKSI_CONSTRAINT(PointPointDistance);
FIKSI_CONSTRAINT(PointPointPointAngle);
FIKSI_CONSTRAINT(PointLineIncidence);
FIKSI_CONSTRAINT(PointLineDistance);
FIKSI_CONSTRAINT(PointCircleIncidence);
FIKSI_CONSTRAINT(SegmentSegmentLengthEquality);
FIKSI_CONSTRAINT(LineLineAngle);
FIKSI_CONSTRAINT(LineLineParallelism);
FIKSI_CONSTRAINT(LineLinePerpendicularity);
FIKSI_CONSTRAINT(LineCircleTangency);
#undef FIKSI_CONSTRAINT

inline auto create_point_point_coincidence(System& s, ElementHandle<Point> a, ElementHandle<Point> b) {
    return detail::constraint<PointPointCoincidence, 2>(s, FXS_POINT_POINT_COINCIDENCE, {a.id, b.id}, 0., "PointPointCoincidence::create");
}
inline auto create_point_point_distance(System& s, ElementHandle<Point> a, ElementHandle<Point> b, double distance) {
    return detail::constraint<PointPointDistance, 2>(s, FXS_POINT_POINT_DISTANCE, {a.id, b.id}, distance, "PointPointDistance::create");
}
inline auto create_point_point_point_angle(System& s, ElementHandle<Point> a, ElementHandle<Point> b, ElementHandle<Point> c, double angle) {
    return detail::constraint<PointPointPointAngle, 3>(s, FXS_POINT_POINT_POINT_ANGLE, {a.id, b.id, c.id}, angle, "PointPointPointAngle::create");
}
inline auto create_point_line_incidence(System& s, ElementHandle<Point> p, ElementHandle<Line> l) {
    return detail::constraint<PointLineIncidence, 2>(s, FXS_POINT_LINE_INCIDENCE, {p.id, l.id}, 0., "PointLineIncidence::create");
}
inline auto create_point_line_distance(System& s, ElementHandle<Point> p, ElementHandle<Line> l, double distance) {
    return detail::constraint<PointLineDistance, 2>(s, FXS_POINT_LINE_DISTANCE, {p.id, l.id}, distance, "PointLineDistance::create");
}
inline auto create_point_circle_incidence(System& s, ElementHandle<Point> p, ElementHandle<Circle> c) {
    return detail::constraint<PointCircleIncidence, 2>(s, FXS_POINT_CIRCLE_INCIDENCE, {p.id, c.id}, 0., "PointCircleIncidence::create");
}
inline auto create_segment_segment_length_equality(System& s, ElementHandle<Point> a1, ElementHandle<Point> a2,
                                                   ElementHandle<Point> b1, ElementHandle<Point> b2) {
    return detail::constraint<SegmentSegmentLengthEquality, 4>(s, FXS_SEGMENT_SEGMENT_LENGTH_EQUALITY, {a1.id, a2.id, b1.id, b2.id}, 0.,
                                                               "SegmentSegmentLengthEquality::create");
}
inline auto create_line_line_angle(System& s, ElementHandle<Line> a, ElementHandle<Line> b, double angle) {
    return detail::constraint<LineLineAngle, 2>(s, FXS_LINE_LINE_ANGLE, {a.id, b.id}, angle, "LineLineAngle::create");
}
inline auto create_line_line_parallelism(System& s, ElementHandle<Line> a, ElementHandle<Line> b) {
    return detail::constraint<LineLineParallelism, 2>(s, FXS_LINE_LINE_PARALLELISM, {a.id, b.id}, 0., "LineLineParallelism::create");
}
inline auto create_line_line_perpendicularity(System& s, ElementHandle<Line> a, ElementHandle<Line> b) {
    return detail::constraint<LineLinePerpendicularity, 2>(s, FXS_LINE_LINE_PERPENDICULARITY, {a.id, b.id}, 0., "LineLinePerpendicularity::create");
}
inline auto create_line_circle_tangency(System& s, ElementHandle<Line> l, ElementHandle<Circle> c) {
    return detail::constraint<LineCircleTangency, 2>(s, FXS_LINE_CIRCLE_TANGENCY, {l.id, c.id}, 0., "LineCircleTangency::create");
}
}  // namespace constraints

// ConstraintHandle methods (constraints/mod.rs:88-110, 992-1046)
template <class C>
double calculate_residual(const ConstraintHandle<C>& c, const System& s) {
    s.check(c);
    return s.constraint_residuals()[c.id];
}
inline void update_parameter(const ConstraintHandle<constraints::PointPointDistance>& c, System& s, double distance) { fxs_constraint_update_parameter(s.raw(), c.id, distance); }
inline void update_parameter(const ConstraintHandle<constraints::PointPointPointAngle>& c, System& s, double angle) { fxs_constraint_update_parameter(s.raw(), c.id, angle); }
inline void update_parameter(const ConstraintHandle<constraints::PointLineDistance>& c, System& s, double distance) { fxs_constraint_update_parameter(s.raw(), c.id, distance); }
inline void update_parameter(const ConstraintHandle<constraints::LineLineAngle>& c, System& s, double angle) { fxs_constraint_update_parameter(s.raw(), c.id, angle); }

}  // namespace fiksi
