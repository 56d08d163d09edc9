// placeholder until the kernel-construction module lands (next commit)
#include "ps_common.h"
#define STUB(name, ...) extern "C" int name(__VA_ARGS__) { return ps_fail(PS_ERR_UNSUPPORTED, #name ": not built yet"); }
STUB(ps_model_create, ps_model**, int)
STUB(ps_model_destroy, ps_model*)
STUB(ps_model_set_wind, ps_model*, const double*, const int32_t*, int, int, int)
STUB(ps_model_prob_mass, ps_model*, int, const int32_t*, const double*, const double*, const double*, const double*, double, int, double, int, int32_t*, int64_t*, int32_t*, int32_t*)
STUB(ps_model_fetch_coo, ps_model*, int, int32_t*, int32_t*, double*, int64_t)
STUB(ps_model_fetch_debug, ps_model*, int, double*, int32_t*, double*, double*)
STUB(ps_model_mvn_cdf_values, ps_model*, double, double, double, double, double, double, int32_t*, double*, int64_t)
STUB(ps_chain_set_kernels_from_model, ps_solver*, ps_model*, int, int)
STUB(ps_solver_set_state_from_model, ps_solver*, ps_model*, int)
