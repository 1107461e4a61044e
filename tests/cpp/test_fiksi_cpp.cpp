// C++ host-side check of include/fiksi.hpp: three of the reference's end-to-end tests written against the
// C++ mirror (fiksi/src/tests/basic.rs:116-149, fixed.rs:10-43, fixed.rs:94-127). Built and run by
// tests/test_cpp_mirror.py. `--build-only` stops before the first solve (no GPU needed).
#include <cmath>
#include <cstdio>
#include <cstring>

#include "fiksi.hpp"

using namespace fiksi;

static int failures = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::printf("CHECK failed: %s (line %d)\n", #cond, __LINE__); \
            ++failures;                                                    \
        }                                                                  \
    } while (0)

static double rms(const std::vector<double>& v) {
    double s = 0;
    for (double x : v) s += x * x;
    return std::sqrt(s / (double)v.size());
}

int main(int argc, char** argv) {
    const bool build_only = argc > 1 && std::strcmp(argv[1], "--build-only") == 0;
    const double RESIDUAL_THRESHOLD = 1e-4;  // fiksi/src/tests/mod.rs:13

    // basic.rs:116-149 triangle_inscribed_circle
    System s;
    auto p0 = elements::create_point(s, 0., 0.);
    auto p1 = elements::create_point(s, 1., 0.5);
    auto p2 = elements::create_point(s, 1.5, 1.);
    auto p3 = elements::create_point(s, 2.8, 1.5);
    constraints::create_point_point_distance(s, p0, p1, 1.);
    constraints::create_point_point_distance(s, p0, p2, 1.);
    constraints::create_point_point_distance(s, p1, p2, 1.);
    auto line0 = elements::create_line(s, p0, p1);
    auto line1 = elements::create_line(s, p0, p2);
    auto line2 = elements::create_line(s, p1, p2);
    auto radius = elements::create_length(s, 1.);
    auto circle = elements::create_circle(s, p3, radius);
    constraints::create_line_circle_tangency(s, line0, circle);
    constraints::create_line_circle_tangency(s, line1, circle);
    constraints::create_line_circle_tangency(s, line2, circle);
    CHECK(fxs_num_elements(s.raw()) == 9 && fxs_num_constraints(s.raw()) == 6 && fxs_num_variables(s.raw()) == 9);

    // a handle of another System is rejected like the reference's assert_eq! (elements/mod.rs:90-93)
    System other;
    bool threw = false;
    try {
        get_value(p0, other);
    } catch (const std::logic_error&) {
        threw = true;
    }
    CHECK(threw);
    if (build_only) {
        std::printf("build-only ok\n");
        return failures ? 1 : 0;
    }

    s.solve(SolvingOptions());
    CHECK(rms(s.constraint_residuals()) < RESIDUAL_THRESHOLD);

    // fixed.rs:10-43 single_triangle_with_fixed_point
    System t;
    auto q0 = elements::create_point(t, 0., 0.);
    auto q1 = elements::create_point(t, 1., 0.5);
    auto q2 = elements::create_point(t, 2., 1.);
    fix(q1, t);
    constraints::create_point_point_distance(t, q0, q1, 1.);
    constraints::create_point_point_distance(t, q0, q2, 1.);
    auto d12 = constraints::create_point_point_distance(t, q1, q2, 1.);
    t.solve();
    CHECK(rms(t.constraint_residuals()) < RESIDUAL_THRESHOLD);
    CHECK(get_value(q1, t).x == 1. && get_value(q1, t).y == 0.5);  // bit-identical
    CHECK(std::fabs(calculate_residual(d12, t)) < RESIDUAL_THRESHOLD);

    // fixed.rs:94-127 fixed_with_coincidence
    System u;
    auto r0 = elements::create_point(u, 0., 0.);
    auto r1 = elements::create_point(u, 1., 0.5);
    auto r2 = elements::create_point(u, 2., 1.);
    auto r3 = elements::create_point(u, 5., 5.);
    fix(r3, u);
    constraints::create_point_point_distance(u, r0, r1, 1.);
    constraints::create_point_point_distance(u, r1, r2, 1.);
    constraints::create_point_point_coincidence(u, r2, r3);
    u.solve();
    CHECK(rms(u.constraint_residuals()) < RESIDUAL_THRESHOLD);
    CHECK(std::hypot(get_value(r2, u).x - 5., get_value(r2, u).y - 5.) < RESIDUAL_THRESHOLD);

    // fixed.rs also runs the sketch with Decomposer::SinglePass
    {
        System w;
        auto q0 = elements::create_point(w, 0., 0.);
        auto q1 = elements::create_point(w, 1., 0.5);
        auto q2 = elements::create_point(w, 2., 1.);
        auto q3 = elements::create_point(w, 5., 5.);
        fix(q3, w);
        constraints::create_point_point_distance(w, q0, q1, 1.);
        constraints::create_point_point_distance(w, q1, q2, 1.);
        constraints::create_point_point_coincidence(w, q2, q3);
        SolvingOptions o;
        o.decomposer = Decomposer::SinglePass;
        w.solve(o);
        CHECK(rms(w.constraint_residuals()) < RESIDUAL_THRESHOLD);
        CHECK(std::hypot(get_value(q2, w).x - 5., get_value(q2, w).y - 5.) < RESIDUAL_THRESHOLD);
        CHECK(get_value(q3, w).x == 5. && get_value(q3, w).y == 5.);
    }

    // System::analyze, the reference's `overconstrained` test (tests/basic.rs:88-112): four points, six
    // pairwise distances -> the constraint added last is designated
    {
        System w;
        auto q0 = elements::create_point(w, 0.123, 0.1);
        auto q1 = elements::create_point(w, 1.2, 0.);
        auto q2 = elements::create_point(w, -0.5, 1.1);
        auto q3 = elements::create_point(w, 1.599, 1.2);
        constraints::create_point_point_distance(w, q0, q1, 1.);
        constraints::create_point_point_distance(w, q0, q2, 1.5);
        constraints::create_point_point_distance(w, q1, q3, 1.7);
        constraints::create_point_point_distance(w, q2, q3, 1.2);
        constraints::create_point_point_distance(w, q1, q2, 2.);
        auto q0q3 = constraints::create_point_point_distance(w, q0, q3, 5.);
        CHECK(w.get_element_handles().size() == 4 && w.get_element_handles()[1].tag == FXS_POINT);
        CHECK(w.get_constraint_handles().size() == 6);
        Analysis an = w.analyze();
        CHECK(an.overconstrained.size() == 1 && an.overconstrained[0].id == q0q3.id);
        CHECK(an.overconstrained.size() == 1 && an.overconstrained[0].tag == FXS_POINT_POINT_DISTANCE);
    }

    // triangles.rs:10-37 runs its triangle under Decomposer::RecursiveAssembly too
    {
        System w;
        auto q0 = elements::create_point(w, 0., 0.);
        auto q1 = elements::create_point(w, 1., 0.5);
        auto q2 = elements::create_point(w, 2., 1.);
        constraints::create_point_point_distance(w, q0, q1, 1.);
        constraints::create_point_point_distance(w, q0, q2, 1.);
        constraints::create_point_point_distance(w, q1, q2, 1.);
        SolvingOptions o;
        o.decomposer = Decomposer::RecursiveAssembly;
        w.solve(o);
        CHECK(rms(w.constraint_residuals()) < RESIDUAL_THRESHOLD);
    }

    // unsupported options are errors, not silent fallbacks
    threw = false;
    try {
        SolvingOptions o;
        o.decomposer = static_cast<Decomposer>(7);
        u.solve(o);
    } catch (const Error& e) {
        threw = e.code == FX_ERR_UNSUPPORTED;
    }
    CHECK(threw);

    std::printf(failures ? "FAILED\n" : "all C++ mirror checks passed\n");
    return failures ? 1 : 0;
}
