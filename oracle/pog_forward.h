/*
 * pog_forward.h -- TEST INFRASTRUCTURE ONLY.  The reference-side binding (include/bsalign_poa_adapter.h) calls the bsa_pog_* entry points of
 * libbsalign_hip.so; the harness libraries under oracle/_ref must load without that library, so they define the entry points as hidden
 * forwarders and resolve them with dlsym from the handle of the REAL libbsalign_hip.so a test hands in (pog_forward_attach): every call lands in
 * the product's own code.
 */
#ifndef POG_FORWARD_H
#define POG_FORWARD_H
#include <dlfcn.h>
#include <stddef.h>
#include "bsalign_poa.h"

#define POG_HID __attribute__((visibility("hidden")))
#define POG_FWD(ret, name, decl, call, fail) \
	typedef ret (*pogfn_##name) decl; static pogfn_##name pogp_##name; \
	POG_HID ret name decl { if(pogp_##name == NULL) return fail; return pogp_##name call; }
#define POG_FWDV(name, decl, call) \
	typedef void (*pogfn_##name) decl; static pogfn_##name pogp_##name; \
	POG_HID void name decl { if(pogp_##name) pogp_##name call; }

POG_FWD(int, bsa_pog_create, (const bsa_pog_params_t *a, bsa_pog_t **b), (a, b), BSA_E_UNSUPPORTED)
POG_FWDV(bsa_pog_destroy, (bsa_pog_t *a), (a))
POG_FWDV(bsa_pog_clear, (bsa_pog_t *a), (a))
POG_FWD(int, bsa_pog_add_read, (bsa_pog_t *a, const uint8_t *b, uint32_t c, uint32_t *d), (a, b, c, d), BSA_E_UNSUPPORTED)
POG_FWD(int, bsa_pog_import, (bsa_pog_t *a, const bsa_pog_snapshot_t *b, const uint8_t *const *c), (a, b, c), BSA_E_UNSUPPORTED)
POG_FWD(int, bsa_pog_export, (const bsa_pog_t *g, uint32_t *a, uint32_t *b, uint32_t *c, uint32_t *d, uint32_t *e, bsa_pog_node_t *f, uint32_t *h, uint32_t *i, uint32_t *j, uint32_t *k, uint32_t *l, uint32_t *m, uint32_t *n),
	(g, a, b, c, d, e, f, h, i, j, k, l, m, n), BSA_E_UNSUPPORTED)
POG_FWD(int, bsa_pog_select, (bsa_pog_t *a, uint32_t b, uint32_t c, uint32_t d, bsa_pog_read_t *e, const uint32_t **f), (a, b, c, d, e, f), BSA_E_UNSUPPORTED)
POG_FWD(int, bsa_pog_needs_guide, (const bsa_pog_t *a, uint32_t b), (a, b), 0)
POG_FWD(int, bsa_pog_place, (bsa_pog_t *a, const bsa_pog_guide_t *b, const int32_t *c, bsa_pog_read_t *d), (a, b, c, d), BSA_E_UNSUPPORTED)
POG_FWD(int, bsa_pog_program, (bsa_pog_t *g, const bsa_poa_node_t **a, size_t *b, const bsa_poa_edge_t **c, size_t *d, const bsa_poa_cand_t **e, size_t *f, const uint8_t **h, bsa_sweep_params_t *i),
	(g, a, b, c, d, e, f, h, i), BSA_E_UNSUPPORTED)
POG_FWD(int, bsa_pog_run, (bsa_pog_t *a, bsa_pog_backend_fn b, void *c, bsa_poa_result_t *d, const bsa_poa_event_t **e), (a, b, c, d, e), BSA_E_UNSUPPORTED)
POG_FWD(int, bsa_pog_apply, (bsa_pog_t *a, bsa_result_t *b, uint32_t *c), (a, b, c), BSA_E_UNSUPPORTED)
POG_FWD(int, bsa_pog_aux_edges, (const bsa_pog_t *a, const uint64_t **b, size_t *c), (a, b, c), BSA_E_UNSUPPORTED)
POG_FWD(int, bsa_pog_abort, (bsa_pog_t *a), (a), BSA_E_UNSUPPORTED)
POG_FWD(int, bsa_pog_cut, (bsa_pog_t *a, uint32_t b, uint32_t c, uint32_t d), (a, b, c, d), BSA_E_UNSUPPORTED)
POG_FWD(int, bsa_pog_set_cpos, (bsa_pog_t *a, const uint32_t *b, const int32_t *c, size_t d), (a, b, c, d), BSA_E_UNSUPPORTED)
POG_FWD(int, bsa_pog_get_cpos, (const bsa_pog_t *a, const uint32_t *b, int32_t *c, size_t d), (a, b, c, d), BSA_E_UNSUPPORTED)
POG_FWDV(bsa_pog_seconds, (const bsa_pog_t *a, double *b), (a, b))

/* handle = dlopen handle of libbsalign_hip.so (ctypes: CDLL._handle).  Returns the number of entry points that were not found. */
static int pog_forward_attach(void *handle){
	int missing = 0;
#define POG_GET(name) do { pogp_##name = (pogfn_##name)dlsym(handle, #name); if(pogp_##name == NULL) missing ++; } while(0)
	POG_GET(bsa_pog_create); POG_GET(bsa_pog_destroy); POG_GET(bsa_pog_clear); POG_GET(bsa_pog_add_read); POG_GET(bsa_pog_import); POG_GET(bsa_pog_export);
	POG_GET(bsa_pog_select); POG_GET(bsa_pog_needs_guide); POG_GET(bsa_pog_place); POG_GET(bsa_pog_program); POG_GET(bsa_pog_run); POG_GET(bsa_pog_apply);
	POG_GET(bsa_pog_aux_edges); POG_GET(bsa_pog_abort); POG_GET(bsa_pog_cut); POG_GET(bsa_pog_set_cpos); POG_GET(bsa_pog_get_cpos); POG_GET(bsa_pog_seconds);
#undef POG_GET
	return missing;
}
#endif
