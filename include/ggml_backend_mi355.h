/* ggml_backend_mi355.h — the drop-in boundary: a ggml-backend plug-in for AMD Instinct MI355X (gfx950).
 *
 * Exported by libggml-mi355.so (prima_cpp_amd/csrc/ggml_backend_mi355.cpp, compiled against the host project's own
 * ggml headers). It implements the five vtables of the reference's ggml-backend plug-in interface
 *     ggml_backend_reg_i / ggml_backend_device_i / ggml_backend_buffer_type_i / ggml_backend_buffer_i / ggml_backend_i
 *     (reference: ggml/src/ggml-backend-impl.h:15-216)
 * exactly like the reference's CUDA plug-in does (ggml_backend_cuda_reg, ggml/src/ggml-cuda.cu:3302-3340), and routes
 * graph_compute to the C ABI of libprima_mi355.so (include/prima_mi355.h). Entry points mirror the CUDA plug-in's
 * public header (ggml/include/ggml-cuda.h) one for one:
 *
 *   this header                              reference interface it stands in for
 *   ---------------------------------------  ---------------------------------------------------------------
 *   ggml_backend_mi355_reg()                 ggml_backend_cuda_reg()                ggml-cuda.h / ggml-cuda.cu:3302
 *   ggml_backend_mi355_init(device)          ggml_backend_cuda_init(device)         ggml-cuda.cu:3342
 *   ggml_backend_is_mi355(backend)           ggml_backend_is_cuda(backend)          ggml-cuda.cu:2791
 *   ggml_backend_mi355_buffer_type(device)   ggml_backend_cuda_buffer_type(device)  ggml-cuda.cu:593
 *   ggml_backend_mi355_host_buffer_type()    ggml_backend_cuda_host_buffer_type()   ggml-cuda.cu:1001
 *   ggml_backend_mi355_get_device_count()    ggml_backend_cuda_get_device_count()   ggml-cuda.cu:2795
 *   ggml_backend_mi355_get_device_memory()   ggml_backend_cuda_get_device_memory()  ggml-cuda.cu:2805
 *
 * Registration: ggml_backend_register(ggml_backend_mi355_reg()) (ggml/src/ggml-backend.cpp:590) before the first
 * ggml_backend_dev_* call, or the one-line `#ifdef GGML_USE_MI355` next to the CUDA one in ggml_backend_registry()
 * (ggml-backend.cpp:549-551) - see INTEGRATION.md.
 */
#ifndef GGML_BACKEND_MI355_H
#define GGML_BACKEND_MI355_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* opaque ggml handles (typedefs of ggml/include/ggml-backend.h:10-16); a translation unit that already included
 * ggml-backend.h defines GGML_BACKEND_MI355_HAVE_GGML to skip them */
#ifndef GGML_BACKEND_MI355_HAVE_GGML
typedef struct ggml_backend_reg * ggml_backend_reg_t;
typedef struct ggml_backend * ggml_backend_t;
typedef struct ggml_backend_buffer_type * ggml_backend_buffer_type_t;
#endif

#define GGML_MI355_NAME "MI355"
#define GGML_MI355_MAX_DEVICES 16

__attribute__((visibility("default"))) ggml_backend_reg_t         ggml_backend_mi355_reg(void);
__attribute__((visibility("default"))) ggml_backend_t             ggml_backend_mi355_init(int device);
__attribute__((visibility("default"))) int                        ggml_backend_is_mi355(ggml_backend_t backend);
__attribute__((visibility("default"))) ggml_backend_buffer_type_t ggml_backend_mi355_buffer_type(int device);
__attribute__((visibility("default"))) ggml_backend_buffer_type_t ggml_backend_mi355_host_buffer_type(void);
__attribute__((visibility("default"))) int                        ggml_backend_mi355_get_device_count(void);
__attribute__((visibility("default"))) void                       ggml_backend_mi355_get_device_memory(int device, size_t * free, size_t * total);
/* Optional hint, also reachable as ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355_set_graph_reused") (the CUDA plug-in keeps its graph on its own
 * side of the boundary, ggml-cuda.cu:2513-2780; here a host that keeps its ggml_cgraph objects between tokens - oracle/ref_patches/graph_reuse.patch - says so
 * and the plug-in skips the per-token fingerprint walk over the nodes). reused == 0 restores the default. */
__attribute__((visibility("default"))) void                       ggml_backend_mi355_set_graph_reused(ggml_backend_t backend, int reused);

#ifdef __cplusplus
}
#endif
#endif
