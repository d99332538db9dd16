// Device-resident time step (placeholder until the driver lands in this file).
#include "pl_internal.h"

extern "C" int pl_tracers_upload(pl_ctx* ctx, int64_t n, const double* tr_x, const double* tr_f) {
    (void)n; (void)tr_x; (void)tr_f; return pl_fail(ctx, "pl_tracers_upload: not implemented yet");
}
extern "C" int pl_tracers_download(pl_ctx* ctx, int64_t n, double* tr_x, double* tr_f) {
    (void)n; (void)tr_x; (void)tr_f; return pl_fail(ctx, "pl_tracers_download: not implemented yet");
}
extern "C" int pl_tracers_count(pl_ctx* ctx, int64_t* n) { (void)n; return pl_fail(ctx, "not implemented yet"); }
extern "C" int pl_step(pl_ctx* ctx, const pl_step_config* cfg, int it, pl_step_report* rep) {
    (void)cfg; (void)it; (void)rep; return pl_fail(ctx, "pl_step: not implemented yet");
}
extern "C" int pl_get_field(pl_ctx* ctx, const char* name, double* out) {
    (void)name; (void)out; return pl_fail(ctx, "pl_get_field: not implemented yet");
}
extern "C" int pl_get_tracer_velocity(pl_ctx* ctx, int64_t n, double* out) {
    (void)n; (void)out; return pl_fail(ctx, "pl_get_tracer_velocity: not implemented yet");
}
void pl_step_free(pl_ctx* ctx) { (void)ctx; }
