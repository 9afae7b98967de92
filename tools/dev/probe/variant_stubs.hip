// Link stubs for tools/tune_*.py variant libraries: the register-resident regime-S kernels (posterior_shared_reg.hip, minutes
// to compile) are not part of a tuning variant; the symbols the other files reference resolve to "does not fit".
#include "bcbf_common.h"
namespace bcbf {
bool posterior_shared64_fits(int, int, int) { return false; }
bool posterior_shared_reg32_fits(int, int, int) { return false; }
template <typename T>
int launch_posterior_shared_reg(const T*, const T*, const T*, const T*, const T*, const T*, const T*, const T*, const T*, const T*, T*, T*, T*,
                                int, int, int, int, void*) { return BCBF_EINVAL; }
template int launch_posterior_shared_reg<float>(const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*, float*, float*, int, int, int, int, void*);
template int launch_posterior_shared_reg<double>(const double*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, double*, double*, double*, int, int, int, int, void*);
}
extern "C" int bcbf_posterior_shared_f64(const double*, const double*, const double*, const double*, const double*, const double*, const double*,
                                         const double*, const double*, const double*, double*, double*, double*, int, int, int, int, void*) { return BCBF_EINVAL; }
