// gh_hodlr.hip -- placeholder while the HODLR path is being written (replaced in the next commit)
#include "gh_common.h"
struct gh_hodlr { int dummy; };
static int ni() { gh_set_error("HODLR solver not implemented yet"); return GH_ERR_BAD_ARG; }
extern "C" int gh_hodlr_create(const gh_hodlr_opts*, gh_hodlr**) { return ni(); }
extern "C" void gh_hodlr_destroy(gh_hodlr*) {}
extern "C" int gh_hodlr_compute(gh_hodlr*, gh_kernel*, const double*, int64_t, int32_t, const double*, double*) { return ni(); }
extern "C" int gh_hodlr_solve(gh_hodlr*, const double*, int64_t, double*) { return ni(); }
extern "C" int gh_hodlr_dot_solve(gh_hodlr*, const double*, double*) { return ni(); }
extern "C" int gh_hodlr_get_inverse(gh_hodlr*, double*) { return ni(); }
extern "C" int gh_hodlr_ranks(const gh_hodlr*, int32_t*, int32_t, int32_t*) { return ni(); }
