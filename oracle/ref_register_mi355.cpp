// oracle/ref_register_mi355.cpp — OUR 5-line shim linked next to the reference's UNMODIFIED tests/test-backend-ops.cpp:
// registers the MI355 plug-in with the reference's backend registry before main() runs, i.e. what the one-line
// `#ifdef GGML_USE_MI355 register_backend(ggml_backend_mi355_reg());` in ggml_backend_registry() would do
// (ggml/src/ggml-backend.cpp:549-563). Test infrastructure only.
#include "ggml-backend.h"
#include "ggml-backend-impl.h"
#define GGML_BACKEND_MI355_HAVE_GGML
#include "ggml_backend_mi355.h"

static struct mi355_registrar { mi355_registrar() { ggml_backend_register(ggml_backend_mi355_reg()); } } g_mi355_registrar;
