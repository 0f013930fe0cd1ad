/* A plain-C caller of the C-ABI (include/qpn_hip.h), the way a foreign-language binding sees it: host arrays in,
 * host arrays out, no torch, no C++.  Solves the level-2 AVI of examples/simple_bilevel.jl (SURVEY.md section 8(c),
 * hand-derived known answers) through qpn_solve_mcp_csc -- PATHSolver.solve_mcp's own argument list, 1-based CSC --
 * a two-node batch through qpn_solve_nodes, the same nodes as resident records (qpn_nodes_upload / qpn_solve_nodes_h /
 * qpn_verify_nodes_h), and a two-player pool through qpn_assemble_pools.  Exit code 0 = all answers as expected. */
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
    /* the same two nodes as RESIDENT records: upload once, sweep twice with only the outputs travelling; then replace the
     * linear terms of the records and sweep again (the outer loop's "same nodes, new parameters") */
    {
        const double Qd[8] = {1, 0, 0, 1, 1, 0, 0, 1}, Ad[4] = {1, 1, 1, 1};
        double qd[4] = {-1.0, -1.0, -0.2, -0.3};
        const double lo[2] = {-INFINITY, -INFINITY}, hi[2] = {1.0, 1.0};
        qpn_nodes *nodes = NULL;
        rc = qpn_nodes_upload(ctx, 2, 2, 1, 0, Qd, NULL, qd, Ad, NULL, lo, hi, QPN_MEM_HOST, &nodes);
        CHECK(rc == QPN_OK && nodes != NULL);
        for (int sweep = 0; sweep < 2; ++sweep) {
            double x[4] = {9, 9, 9, 9};
            int32_t status[2] = {0, 0};
            /* only the statuses and the primal blocks come back (z == NULL) */
            rc = qpn_solve_nodes_h(ctx, nodes, NULL, 0, NULL, status, NULL, NULL, NULL, NULL, QPN_MEM_HOST, x, 2);
            CHECK(rc == QPN_OK && status[0] == 1 && status[1] == 1);
            CHECK(fabs(x[0] - 0.5) <= 1e-9 && fabs(x[1] - 0.5) <= 1e-9 && fabs(x[2] - 0.2) <= 1e-9 && fabs(x[3] - 0.3) <= 1e-9);
        }
        int32_t info[4] = {-1, -1, -1, -1};
        CHECK(qpn_nodes_info(ctx, nodes, info) == QPN_OK && info[0] == 2 && info[1] == 0);   /* no node needs the general kernel */
        qd[2] = -2.0; qd[3] = -2.0;                        /* node 2 now also hits x1 + x2 <= 1 */
        CHECK(qpn_nodes_update(ctx, nodes, QPN_NODE_Q, qd, QPN_MEM_HOST) == QPN_OK);
        {
            double z[6] = {0}; int32_t status[2] = {0, 0}, sol[2] = {0, 0}, path[2] = {0, 0}; double lam[2] = {0, 0};
            rc = qpn_solve_nodes_h(ctx, nodes, NULL, 0, z, status, NULL, NULL, NULL, NULL, QPN_MEM_HOST, NULL, 0);
            CHECK(rc == QPN_OK && status[1] == 1 && fabs(z[3] - 0.5) <= 1e-9 && fabs(z[4] - 0.5) <= 1e-9 && fabs(z[5] + 1.5) <= 1e-9);
            /* verify_solution (src/qp_processing.jl:57-149) on the resident records at that point: optimal, duals by least squares */
            const double xd[4] = {0.5, 0.5, 0.5, 0.5};
            rc = qpn_verify_nodes_h(ctx, nodes, xd, NULL, 0, 1e-4, sol, lam, path, QPN_MEM_HOST);
            CHECK(rc == QPN_OK && sol[0] == 1 && sol[1] == 1 && path[1] == 2 && fabs(lam[1] + 1.5) <= 1e-7);
        }
        CHECK(qpn_nodes_free(ctx, nodes) == QPN_OK);
    }
    /* a two-player pool assembled on the device (combine_gavis, src/avi.jl:305-377, reduced form): player 1 decides x1 with
     * the row x1 + x2 >= 1, player 2 decides x2 with no row; M = [[2 0 -1],[0 2 0],[1 1 0]] */
    {
        const int32_t n_i[2] = {1, 1}, m_i[2] = {1, 0}, dpos[2] = {0, 1};
        qpn_pool_shape shape = {2, 2, 0, n_i, m_i, dpos};
        const double Qd[4] = {2, 0, 0, 2}, qd[2] = {-1.0, -4.0}, Ad[2] = {1, 1}, lo[1] = {1.0}, hi[1] = {INFINITY};
        double M[9], q[3], l[3], u[3]; uint8_t kind[3]; int32_t N = 0;
        CHECK(qpn_pool_size(&shape, QPN_POOL_REDUCED, &N) == QPN_OK && N == 3);
        rc = qpn_assemble_pools(ctx, &shape, QPN_POOL_REDUCED, 1, Qd, 0, NULL, 0, qd, 0, Ad, 0, NULL, 0, lo, hi, 0, NULL, 0, M, 9, q, l, u,
                                kind, QPN_MEM_HOST);
        CHECK(rc == QPN_OK);
        const double wantM[9] = {2, 0, 1, 0, 2, 1, -1, 0, 0};                 /* column-major */
        for (int i = 0; i < 9; ++i) CHECK(M[i] == wantM[i]);
        CHECK(q[0] == -1.0 && q[1] == -4.0 && q[2] == 0.0 && kind[0] == QPN_ROW_STD && kind[2] == QPN_ROW_GAVI && l[2] == 1.0);
        double z[3] = {0, 0, 0}; int32_t status = 0;
        rc = qpn_solve_avi_batch(ctx, 1, 3, M, 9, q, l, u, kind, 3, z, &status, NULL, NULL, NULL, NULL, QPN_MEM_HOST);
        CHECK(rc == QPN_OK && status == 1 && fabs(z[0] - 0.5) <= 1e-9 && fabs(z[1] - 2.0) <= 1e-9 && fabs(z[2]) <= 1e-12);
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
