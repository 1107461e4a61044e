// CPU check of the multifrontal plan (fiksi_amd/csrc/fx_front_plan.h) before any kernel sees it: builds the plan of a
// sketch the way the library does (fx_sparse_plan.h: plan_component), then walks the segment blobs exactly as the device
// code of fx_front.h does — staging tile, entries of A by record, children's contribution blocks through their byte maps,
// the partial Cholesky of a 16-lane row (lane = column, registers = rows), L / contribution / y storage, the backward
// sweep top-down — in plain scalar C++, and compares the step with a dense Cholesky solve of (Jt J + lambda I) x = -Jt r.
// Test infrastructure (tests/test_front_plan.py); nothing here is linked into the product.
//   front_harness <kind> <n_points> <seed>      kind: chain | hinged | grid
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../fiksi_amd/csrc/fx_sparse_plan.h"

using namespace fx::sparse_plan;

namespace {

struct Lcg {
    uint32_t s;
    double u() {
        s = s * 1664525u + 1013904223u;
        return (double)s / 4294967295.0;
    }
};

struct Sketch {
    std::vector<uint32_t> var_off{0}, expr_off{0}, expr_idx;
    std::vector<double> vars, expr_param;
    std::vector<uint8_t> var_fixed, expr_tag;
    fx_batch batch{};
    void point(double x, double y) {
        vars.push_back(x);
        vars.push_back(y);
        var_fixed.push_back(0);
        var_fixed.push_back(0);
    }
    void dist(uint32_t p, uint32_t q) {  // PointPointDistance on points p, q (fields: first variable of each)
        expr_tag.push_back(FX_POINT_POINT_DISTANCE);
        expr_idx.insert(expr_idx.end(), {2 * p, 2 * q, 0u, 0u});
        expr_param.push_back(1.0);
    }
    void angle(uint32_t p, uint32_t q, uint32_t r) {
        expr_tag.push_back(FX_POINT_POINT_POINT_ANGLE);
        expr_idx.insert(expr_idx.end(), {2 * p, 2 * q, 2 * r, 0u});
        expr_param.push_back(0.5);
    }
    void done() {
        var_off.push_back((uint32_t)vars.size());
        expr_off.push_back((uint32_t)expr_tag.size());
        batch.n_systems = 1;
        batch.var_off = var_off.data();
        batch.expr_off = expr_off.data();
        batch.vars = vars.data();
        batch.var_fixed = var_fixed.data();
        batch.expr_tag = expr_tag.data();
        batch.expr_idx = expr_idx.data();
        batch.expr_param = expr_param.data();
        batch.var_comp = nullptr;
        batch.expr_comp = nullptr;
    }
};

// one segment's sweep up: the fronts level by level (what is never written stays NaN: a read of padding that reaches a result shows)
struct Store {
    std::vector<double> l, u, x;  // segment-local: L blocks, contribution slots, x by local column
};

bool sweep_up(const uint32_t* w, const double* a_vals, const double* rhs, double lambda, Store& st, std::vector<double>& gu) {
    const uint32_t nlev = w[1], c0 = w[5];
    const uint32_t* lev = w + w[6];
    const uint32_t* fr = w + w[7];
    const uint32_t* recs = w + w[8];
    const uint32_t* cols = w + w[9];
    const uint32_t* kids = w + w[10];
    const double nan = std::nan("");
    st.l.assign(w[11], nan);
    st.u.assign((size_t)w[12] + 32, nan);
    st.x.assign(w[3], nan);
    for (uint32_t q = 0; q < nlev; ++q)
        for (uint32_t fi = lev[q]; fi < lev[q + 1]; ++fi) {
            const uint32_t* d = fr + fi * MF_FRONT_WORDS;
            const int npiv = (int)(d[0] & 0xFF), nbnd = (int)((d[0] >> 8) & 0xFF), F = npiv + nbnd;
            const uint32_t nch = (d[0] >> 16) & 0xFF, flags = d[0] >> 24;
            double tile[MF_TS][MF_TS];
            memset(tile, 0, sizeof(tile));
            for (uint32_t t = 0; t < d[2]; ++t) {
                const uint32_t r = recs[d[1] + t], li = r & 15u, lj = (r >> 4) & 15u;
                const double v = a_vals[r >> 8];
                tile[li][lj] = v;
                tile[lj][li] = v;
            }
            const uint32_t* fc = cols + d[3];
            for (int li = 0; li < npiv; ++li) {
                tile[li][li] += lambda;
                tile[li][F] = rhs[fc[li] - c0];
            }
            const uint32_t* kp = kids + d[4];
            for (uint32_t c = 0; c < nch; ++c, kp += MF_CHILD_WORDS) {
                const uint32_t uo = kp[0] & 0x7FFFFFFFu, nb = kp[1];
                const double* U = (kp[0] >> 31) ? gu.data() + uo : st.u.data() + uo;  // (the device adds 16 / W children per round: same sums up to their order)
                const uint8_t* map = reinterpret_cast<const uint8_t*>(kp + 2);
                for (uint32_t cc = 0; cc <= nb; ++cc) {      // lane cc: column cc of the child's block
                    const uint32_t mc = cc < nb ? map[cc] : (uint32_t)F;
                    for (uint32_t r = 0; r < MF_FMAX; ++r) tile[map[r]][mc] += U[MF_LS * cc + r];  // (r >= nb: padding into the spare row)
                }
            }
            // registers: lane = column, a[lane][i] = tile[i][lane]
            double a[MF_N][MF_N], invd[MF_N];
            for (uint32_t c = 0; c < MF_N; ++c) {
                invd[c] = 1.0;
                for (uint32_t i = 0; i < MF_N; ++i) a[c][i] = tile[i][c];
            }
            for (int K = 0; K < npiv; ++K) {
                const double piv = a[K][K];
                if (!(piv > 0.0)) return false;
                const double rs = 1.0 / std::sqrt(piv), ip = rs * rs;
                double mul[MF_N], col[MF_N];
                for (uint32_t i = 0; i < MF_N; ++i) col[i] = a[K][i];
                for (int c = 0; c < (int)MF_N; ++c) {
                    const double ljk = a[c][K] * rs;
                    mul[c] = c > K ? a[c][K] * ip : 0.0;
                    if (c >= K) a[c][K] = ljk;
                    if (c == K) invd[c] = rs;
                }
                for (int i = K + 1; i < (int)MF_N; ++i)
                    for (uint32_t c = 0; c < MF_N; ++c) a[c][i] -= col[i] * mul[c];
            }
            // every lane up to the right-hand side's stores its sixteen registers: pivots -> L, the right-hand side's lane too (y),
            // boundary and right-hand side -> the contribution block, rows shifted by npiv
            for (int c = 0; c < npiv; ++c) {
                for (uint32_t i = 0; i < MF_N; ++i) st.l[d[5] + MF_LS * c + i] = a[c][i];
                st.l[d[5] + MF_LS * c + c] = invd[c];
            }
            for (uint32_t i = 0; i < MF_N; ++i) st.l[d[5] + MF_LS * npiv + i] = a[F][i];
            if (nbnd) {
                double* U = (flags & MF_U_GLOBAL) ? gu.data() + d[6] : st.u.data() + d[6];
                for (int c = npiv; c <= F; ++c)
                    for (int i = 0; i < (int)MF_N; ++i) U[(int)MF_LS * (c - npiv) + (i - npiv)] = a[c][i];
            }
        }
    return true;
}

// ... and down: x of the segment's columns into st.x (local) and into x_all (the whole factor's numbering)
void sweep_down(const uint32_t* w, Store& st, std::vector<double>& x_all) {
    const uint32_t nlev = w[1], c0 = w[5], nc = w[3];
    const uint32_t* lev = w + w[6];
    const uint32_t* fr = w + w[7];
    const uint32_t* cols = w + w[9];
    for (uint32_t q = nlev; q-- > 0;)
        for (uint32_t fi = lev[q]; fi < lev[q + 1]; ++fi) {
            const uint32_t* d = fr + fi * MF_FRONT_WORDS;
            const uint32_t npiv = d[0] & 0xFF, nbnd = (d[0] >> 8) & 0xFF, F = npiv + nbnd;
            const uint32_t* fc = cols + d[3];
            double a[MF_N][MF_N], acc[MF_N], invd2[MF_N];
            for (uint32_t c = 0; c < MF_N; ++c) {
                acc[c] = 0.0;
                invd2[c] = 1.0;
                for (uint32_t i = 0; i < MF_N; ++i) a[c][i] = 0.0;
            }
            for (uint32_t c = 0; c < npiv; ++c) {
                for (uint32_t i = 0; i < MF_N; ++i) a[c][i] = st.l[d[5] + MF_LS * c + i];
                const double ad = a[c][c];  // 1 / d
                acc[c] = st.l[d[5] + MF_LS * npiv + c] / ad;
                invd2[c] = ad * ad;
            }
            for (uint32_t c = npiv; c < F; ++c) {
                const uint32_t col = fc[c];
                acc[c] = (col >= c0 && col < c0 + nc) ? st.x[col - c0] : x_all[col];
            }
            for (uint32_t K = MF_N; K-- > 0;) {
                const double t = acc[K] * invd2[K];
                for (uint32_t c = 0; c < K; ++c) acc[c] -= t * a[c][K];
            }
            for (uint32_t c = 0; c < npiv; ++c) {
                const double x = acc[c] * invd2[c];
                st.x[fc[c] - c0] = x;
                x_all[fc[c]] = x;
            }
        }
}

}  // namespace

int main(int argc, char** argv) {
    const char* kind = argc > 1 ? argv[1] : "chain";
    const uint32_t n = argc > 2 ? (uint32_t)atoi(argv[2]) : 400;
    Lcg g{argc > 3 ? (uint32_t)atoi(argv[3]) : 7u};
    Sketch s;
    if (!strcmp(kind, "chain")) {  // BASELINE cfg2's make-up: chain distances, skip distances, three-point angles
        for (uint32_t i = 0; i < n; ++i) s.point(i + 0.3 * g.u(), g.u());
        for (uint32_t i = 0; i + 1 < n; ++i) s.dist(i, i + 1);
        for (uint32_t i = 0; i + 2 < n && i < 2 * (n * 2 / 5); i += 2) s.dist(i, i + 2);
        for (uint32_t i = 1; i + 1 < n && i < 2 * (n * 3 / 5); i += 2) s.angle(i - 1, i, i + 1);
    } else if (!strcmp(kind, "hinged")) {  // fiksi_bench.rs:15-40: triangles around one hinge point
        s.point(0, 0);
        s.point(1, 0);
        for (uint32_t t = 0; t < n; ++t) {
            s.point(std::cos(0.3 * (t + 1)), std::sin(0.3 * (t + 1)));
            const uint32_t prev = t + 1, cur = t + 2;
            s.dist(0, prev);
            s.dist(prev, cur);
            s.dist(cur, 0);
        }
    } else if (!strcmp(kind, "file")) {  // System 0 of a batch dumped by tests/test_front_plan.py: <dir>/var_fixed.u8, expr_tag.u8, expr_idx.u32
        const std::string dir = argv[2];
        auto slurp = [&](const char* name, size_t elem) {
            std::vector<unsigned char> out;
            FILE* f = fopen((dir + "/" + name).c_str(), "rb");
            if (!f) return out;
            unsigned char buf[65536];
            size_t got;
            while ((got = fread(buf, 1, sizeof(buf), f)) > 0) out.insert(out.end(), buf, buf + got);
            fclose(f);
            (void)elem;
            return out;
        };
        const auto fx = slurp("var_fixed.u8", 1), tg = slurp("expr_tag.u8", 1), ix = slurp("expr_idx.u32", 4);
        if (fx.empty() || tg.empty() || ix.size() != tg.size() * 16) {
            printf("bad dump in %s\n", dir.c_str());
            return 2;
        }
        s.var_fixed.assign(fx.begin(), fx.end());
        s.vars.resize(fx.size());
        for (double& v : s.vars) v = g.u();
        s.expr_tag.assign(tg.begin(), tg.end());
        s.expr_idx.resize(ix.size() / 4);
        memcpy(s.expr_idx.data(), ix.data(), ix.size());
        s.expr_param.assign(tg.size(), 1.0);
    } else {  // a grid: separators grow with the side — beyond some size no front fits a row of lanes
        const uint32_t side = n;
        for (uint32_t i = 0; i < side * side; ++i) s.point(i % side + 0.1 * g.u(), i / side + 0.1 * g.u());
        for (uint32_t y = 0; y < side; ++y)
            for (uint32_t x = 0; x < side; ++x) {
                if (x + 1 < side) s.dist(y * side + x, y * side + x + 1);
                if (y + 1 < side) s.dist(y * side + x, (y + 1) * side + x);
            }
    }
    s.done();
    const uint32_t nvt = (uint32_t)s.vars.size(), net = (uint32_t)s.expr_tag.size();
    std::vector<uint32_t> rows(net), fvar;
    for (uint32_t i = 0; i < net; ++i) rows[i] = i;
    for (uint32_t i = 0; i < nvt; ++i)
        if (!s.var_fixed[i]) fvar.push_back(i);
    ComponentPlan P;
    plan_component(&s.batch, 0, rows, fvar, P);
    int failures = 0;
    for (int which = 0; which < 2; ++which) {
        const FrontPlan& fp = which ? P.fronts_parts : P.fronts_solo;
        if (which && P.parts.empty()) continue;
        if (!fp.ok) {
            printf("%s n=%u %s: no multifrontal build (a front beyond %u columns)\n", kind, n, which ? "parts" : "solo", MF_FMAX);
            continue;
        }
        // values: a random Jacobian on the sketch's pattern, A and the right-hand side by the library's own gather lists
        std::vector<double> jv(P.nnz_j), r(P.m);
        for (double& v : jv) v = 2.0 * g.u() - 1.0;
        for (double& v : r) v = 2.0 * g.u() - 1.0;
        std::vector<double> av(P.nnz_a, 0.0), rhs(P.nv, 0.0);
        for (uint32_t k = 0; k < P.nnz_a; ++k)
            for (uint32_t p = P.apair_ptr[k]; p < P.apair_ptr[k + 1]; ++p) av[k] += jv[P.apairs[2 * p]] * jv[P.apairs[2 * p + 1]];
        for (uint32_t c = 0; c < P.nv; ++c)
            for (uint32_t p = P.cptr[c]; p < P.cptr[c + 1]; ++p) rhs[c] += jv[P.cidx[p]] * -r[P.crow[p]];
        const double lambda = 0.5;
        // the multifrontal step: the parts' sweeps up, the top's, then down in the opposite order
        std::vector<Store> st(fp.nseg);
        std::vector<double> gu(fp.global_u_doubles + 32, std::nan("")), x(P.nv, 0.0);
        bool ok = true;
        for (uint32_t sg = 0; sg < fp.nseg && ok; ++sg)
            ok = sweep_up(fp.words.data() + fp.seg_off[sg], av.data() + fp.seg_a[sg], rhs.data() + fp.seg_col[sg], lambda, st[sg], gu);
        if (!ok) {
            printf("%s n=%u: a pivot was not positive\n", kind, n);
            return 1;
        }
        for (uint32_t sg = fp.nseg; sg-- > 0;) sweep_down(fp.words.data() + fp.seg_off[sg], st[sg], x);
        if (P.nv > 3000) {  // too large for the dense reference: the residual of the step in the sparse system itself
            std::vector<double> res(rhs), jx(P.m, 0.0);
            for (uint32_t row = 0; row < P.m; ++row)
                for (uint32_t p = P.jrow_ptr[row]; p < P.jrow_ptr[row + 1]; ++p) jx[row] += jv[p] * x[P.jcol[p]];
            for (uint32_t row = 0; row < P.m; ++row)
                for (uint32_t p = P.jrow_ptr[row]; p < P.jrow_ptr[row + 1]; ++p) res[P.jcol[p]] -= jv[p] * jx[row];
            double err = 0.0, nrm = 0.0;
            for (uint32_t c = 0; c < P.nv; ++c) {
                err = std::max(err, std::fabs(res[c] - lambda * x[c]));
                nrm = std::max(nrm, std::fabs(rhs[c]));
            }
            printf("%s n=%u %s: %u columns, %u fronts in %u segments, at most %u levels / %u fronts a level, nnz(L) %u; "
                   "max |b - (Jt J + lambda I) x| = %.3e (|b| up to %.3e)\n", kind, n, which ? "parts" : "solo", P.nv, fp.nfronts, fp.nseg,
                   fp.max_levels, fp.max_level_fronts, P.nnz_l, err, nrm);
            if (!(err <= 1e-10 * (1.0 + nrm))) ++failures;
            continue;
        }
        // dense reference: A = Jt J + lambda I in the permuted numbering, from the CSR of J
        std::vector<double> A((size_t)P.nv * P.nv, 0.0), b(rhs);
        for (uint32_t row = 0; row < P.m; ++row)
            for (uint32_t p = P.jrow_ptr[row]; p < P.jrow_ptr[row + 1]; ++p)
                for (uint32_t q = P.jrow_ptr[row]; q < P.jrow_ptr[row + 1]; ++q) A[(size_t)P.jcol[p] * P.nv + P.jcol[q]] += jv[p] * jv[q];
        for (uint32_t c = 0; c < P.nv; ++c) A[(size_t)c * P.nv + c] += lambda;
        for (uint32_t k = 0; k < P.nv; ++k) {  // Cholesky in place (lower), then the two solves
            for (uint32_t j = 0; j < k; ++j) A[(size_t)k * P.nv + k] -= A[(size_t)k * P.nv + j] * A[(size_t)k * P.nv + j];
            const double dk = std::sqrt(A[(size_t)k * P.nv + k]);
            A[(size_t)k * P.nv + k] = dk;
            for (uint32_t i = k + 1; i < P.nv; ++i) {
                double v = A[(size_t)i * P.nv + k];
                for (uint32_t j = 0; j < k; ++j) v -= A[(size_t)i * P.nv + j] * A[(size_t)k * P.nv + j];
                A[(size_t)i * P.nv + k] = v / dk;
            }
        }
        for (uint32_t i = 0; i < P.nv; ++i) {
            for (uint32_t j = 0; j < i; ++j) b[i] -= A[(size_t)i * P.nv + j] * b[j];
            b[i] /= A[(size_t)i * P.nv + i];
        }
        for (uint32_t i = P.nv; i-- > 0;) {
            for (uint32_t j = i + 1; j < P.nv; ++j) b[i] -= A[(size_t)j * P.nv + i] * b[j];
            b[i] /= A[(size_t)i * P.nv + i];
        }
        double err = 0.0, nrm = 0.0;
        for (uint32_t i = 0; i < P.nv; ++i) {
            err = std::max(err, std::fabs(x[i] - b[i]));
            nrm = std::max(nrm, std::fabs(b[i]));
        }
        uint32_t piv = 0;
        for (uint32_t sg = 0; sg < fp.nseg; ++sg) piv += fp.words[fp.seg_off[sg] + 3];
        printf("%s n=%u %s: %u columns, %u fronts in %u segments, at most %u levels / %u fronts a level, nnz(L) %u (dense rows: %u doubles); "
               "max |x - x_dense| = %.3e (|x| up to %.3e)\n", kind, n, which ? "parts" : "solo", P.nv, fp.nfronts, fp.nseg, fp.max_levels,
               fp.max_level_fronts, P.nnz_l, fp.max_l_doubles, err, nrm);
        if (piv != P.nv || !(err <= 1e-9 * (1.0 + nrm))) ++failures;
    }
    printf(failures ? "FAILED\n" : "front plan ok\n");
    return failures ? 1 : 0;
}
