/*
 * misslap_diag.h -- diagnostics entry points of libmisslap_diag.so (the library built with -DMISSLAP_DIAG,
 * `python -m sslap_amd.build diag`).  NOT part of the product ABI: libmisslap.so does not export them and nothing
 * on the reference's path has a counterpart.  Used by tools/diag.py and tools/time_scan.py only.
 */
#ifndef MISSLAP_DIAG_H
#define MISSLAP_DIAG_H
#include "misslap.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Average duration (ms) of `reps` launches of an ablated full-scan bid kernel over the current unassigned list
 * (mode 0 complete, 1 no price gather, 2 no cross-lane reduction, 3 edge stream only, 4 gather from 4-byte prices; 10..16: the LDS-tiled kernel
 * and its ablations); results are discarded, the solver state is untouched. */
int misslap_debug_time_bid(misslap_solver *h, int32_t mode, int32_t reps, float *ms_avg);
#ifdef __cplusplus
}
#endif
#endif
