/* A plain-C caller of the C-ABI (include/qpn_hip.h), the way a foreign-language binding sees it: host arrays in,
 * host arrays out, no torch, no C++.  Solves the level-2 AVI of examples/simple_bilevel.jl (SURVEY.md section 8(c),
 * hand-derived known answers) through qpn_solve_mcp_csc -- PATHSolver.solve_mcp's own argument list, 1-based CSC --
 * and a two-node batch through qpn_solve_nodes.  Exit code 0 = all answers as expected. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "qpn_hip.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "abi_smoke: check failed at line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main(void)
{
    qpn_ctx *ctx = NULL;
    int rc = qpn_ctx_create(0, &ctx);
    if (rc != QPN_OK) { fprintf(stderr, "qpn_ctx_create: %s\n", qpn_strerror(rc)); return 2; }
    CHECK(qpn_abi_version() == QPN_ABI_VERSION);

    /* z = [y, xi, lambda, s], M = [[0,1,0,0],[2,0,-1,0],[1,0,0,-1],[0,0,1,0]], q = [0,-2x,0,0], l = [-inf,-inf,-inf,0], u = +inf */
    {
        const int32_t colptr[5] = {1, 3, 4, 6, 7};                 /* column-major, 1-based like Julia's SparseMatrixCSC */
        const int32_t rowval[6] = {2, 3, 1, 2, 4, 3};
        const double nzval[6] = {2.0, 1.0, 1.0, -1.0, 1.0, -1.0};
        const double xs[3] = {-2.0, 0.5, 0.0};
        const double want[3][4] = {{0, 0, 4, 0}, {0.5, 0, 0, 0.5}, {0, 0, 0, 0}};
        for (int t = 0; t < 3; ++t) {
            double q[4] = {0.0, -2.0 * xs[t], 0.0, 0.0};
            double l[4] = {-INFINITY, -INFINITY, -INFINITY, 0.0}, u[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
            double z[4] = {0, 0, 0, 0}, resid = -1.0;
            int32_t status = 0, pivots = -1;
            rc = qpn_solve_mcp_csc(ctx, 4, colptr, rowval, nzval, q, l, u, z, &status, &resid, &pivots, NULL);
            CHECK(rc == QPN_OK);
            CHECK(status == 1 && resid <= 1e-8);
            for (int i = 0; i < 4; ++i) CHECK(fabs(z[i] - want[t][i]) <= 1e-9);
        }
    }
    /* two nodes, n = 2, m = 1, no parameters: min 1/2 |x|^2 - c'x  s.t.  x1 + x2 <= 1   (column-major blocks per node) */
    {
        const double Qd[8] = {1, 0, 0, 1, 1, 0, 0, 1}, qd[4] = {-1.0, -1.0, -0.2, -0.3}, Ad[4] = {1, 1, 1, 1};
        const double lo[2] = {-INFINITY, -INFINITY}, hi[2] = {1.0, 1.0};
        double z[6] = {0}, resid[2];
        int32_t status[2] = {0, 0}, pivots[2];
        uint8_t active[6];
        rc = qpn_solve_nodes(ctx, 2, 2, 1, 0, Qd, NULL, qd, Ad, NULL, lo, hi, NULL, 0, z, status, resid, pivots, active, NULL, QPN_MEM_HOST);
        CHECK(rc == QPN_OK);
        CHECK(status[0] == 1 && status[1] == 1);
        CHECK(fabs(z[0] - 0.5) <= 1e-9 && fabs(z[1] - 0.5) <= 1e-9);       /* projected onto x1 + x2 = 1; multiplier -0.5 (upper bound) */
        CHECK(fabs(z[2] + 0.5) <= 1e-9);
        CHECK(fabs(z[3] - 0.2) <= 1e-9 && fabs(z[4] - 0.3) <= 1e-9 && fabs(z[5]) <= 1e-12);   /* inactive */
    }
    /* misuse comes back as a code and a message, not as a fault */
    {
        double z[4] = {0}; int32_t st = 0;
        rc = qpn_solve_mcp_csc(ctx, 4, NULL, NULL, NULL, NULL, NULL, NULL, z, &st, NULL, NULL, NULL);
        CHECK(rc != QPN_OK);
        CHECK(qpn_ctx_last_error(ctx) != NULL);
    }
    CHECK(qpn_ctx_destroy(ctx) == QPN_OK);
    printf("abi_smoke ok\n");
    return 0;
}
