// Krylov solvers (placeholder until the preconditioned solvers land in this file).
#include "pl_internal.h"

extern "C" int pl_stokes_solve(pl_ctx* ctx, const double* rhs, double* x, int use_x0, double rtol, int maxit,
                               pl_solve_stats* stats) {
    (void)rhs; (void)x; (void)use_x0; (void)rtol; (void)maxit; (void)stats;
    return pl_fail(ctx, "pl_stokes_solve: not implemented yet");
}

extern "C" int pl_heat_solve(pl_ctx* ctx, const double* rhs, double* x, double rtol, int maxit,
                             pl_solve_stats* stats) {
    (void)rhs; (void)x; (void)rtol; (void)maxit; (void)stats;
    return pl_fail(ctx, "pl_heat_solve: not implemented yet");
}

void pl_solver_free(pl_ctx* ctx) { (void)ctx; }
