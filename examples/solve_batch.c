/* Plain C host: two sketches through the C ABI (include/fiksi_amd.h), no builder, no C++.
 *
 *   gcc -std=c99 -Iinclude examples/solve_batch.c -Lfiksi_amd -lfiksi_amd -Wl,-rpath,$PWD/fiksi_amd -lm -o solve_batch
 *
 * System 0: the reference's quadrilateral (tests/basic.rs:95-98) with the six pairwise distances of a unit
 * square. System 1: a triangle with one fixed point (tests/fixed.rs:10-43). Prints the solved variables and
 * the per-System results; exit code 0 when both converged. */
#include <math.h>
#include <stdio.h>

#include "fiksi_amd.h"

int main(void) {
    /* variables: points are (x, y) pairs; System 0 has 4 points, System 1 has 3 */
    double vars[14] = {0.123, 0.1, 1.2, 0., -0.5, 1.1, 1.599, 1.2, /* | */ 0., 0., 1., 0.5, 2., 1.};
    uint8_t var_fixed[14] = {0, 0, 0, 0, 0, 0, 0, 0, /* | */ 0, 0, 1, 1, 0, 0};
    uint32_t var_off[3] = {0, 8, 14};
    /* expressions: all PointPointDistance (tag 1); element fields = system-local index of each point's x */
    uint8_t expr_tag[9] = {1, 1, 1, 1, 1, 1, 1, 1, 1};
    uint32_t expr_idx[9 * 4] = {0, 2, 0, 0, 2, 6, 0, 0, 6, 4, 0, 0, 4, 0, 0, 0, 0, 6, 0, 0, 2, 4, 0, 0,
                                /* | */ 0, 2, 0, 0, 0, 4, 0, 0, 2, 4, 0, 0};
    const double r2 = 1.4142135623730951;
    double expr_param[9] = {1., 1., 1., 1., r2, r2, /* | */ 1., 1., 1.};
    uint32_t expr_off[3] = {0, 6, 9};
    fx_batch batch;
    fx_solving_opts opts;
    fx_result res[2];
    fx_ctx* ctx = NULL;
    int rc, s, i;

    batch.n_systems = 2;
    batch.var_off = var_off;
    batch.expr_off = expr_off;
    batch.vars = vars;
    batch.var_fixed = var_fixed;
    batch.expr_tag = expr_tag;
    batch.expr_idx = expr_idx;
    batch.expr_param = expr_param;
    batch.var_comp = NULL; /* one connected component per System */
    batch.expr_comp = NULL;

    if (fx_batch_validate(&batch) != FX_OK) {
        fprintf(stderr, "invalid batch: %s\n", fx_last_error());
        return 2;
    }
    rc = fx_ctx_create(&ctx, 0);
    if (rc != FX_OK) { /* no gfx950 device: there is no CPU fallback */
        fprintf(stderr, "fx_ctx_create: %d (%s)\n", rc, fx_last_error());
        return 3;
    }
    fx_solving_opts_default(&opts); /* SolvingOptions::DEFAULT */
    rc = fx_system_solve_batch(ctx, &batch, &opts, res);
    if (rc != FX_OK) {
        fprintf(stderr, "fx_system_solve_batch: %d (%s)\n", rc, fx_last_error());
        fx_ctx_destroy(ctx);
        return 4;
    }
    for (s = 0; s < 2; ++s) {
        printf("system %d: %u accepted steps, %u trials, exit %u, unscaled SSE %.3e\n  vars:", s, res[s].accepted, res[s].trials,
               res[s].exit, res[s].sse_unscaled);
        for (i = (int)var_off[s]; i < (int)var_off[s + 1]; ++i) printf(" %.6f", vars[i]);
        printf("\n");
    }
    fx_ctx_destroy(ctx);
    /* converged as the reference's tests define it (RMS residual < 1e-4), the fixed point untouched */
    return (sqrt(res[0].sse_unscaled / 6.) < 1e-4 && sqrt(res[1].sse_unscaled / 3.) < 1e-4 && vars[10] == 1. && vars[11] == 0.5) ? 0 : 1;
}
