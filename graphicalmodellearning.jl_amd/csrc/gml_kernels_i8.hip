// Exact fixed-point (int8-limb) device pass on v_mfma_i32_*_i8 -- placeholder until the
// kernels land; the FP64 path (gml_kernels_f64.hip) is the functional one.
#include "gml_dev.h"
#include "../../include/gml.h"
#include <string>

namespace gml {
int i8_pass(void **, const DevProblem &, const double *, const int *, const int *, const int *, int, int, int,
            bool, double *, double *, hipStream_t, hipEvent_t *, std::string *err) {
    if (err) *err = "GML_PREC_I8X is not implemented yet";
    return GML_EUNSUPPORTED;
}
void i8_free(void *) {}
} // namespace gml
