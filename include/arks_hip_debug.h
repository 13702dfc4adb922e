/*
 * arks_hip_debug.h -- diagnostics of libarks_hip.so.  NOT part of the drop-in boundary (include/arks_hip.h):
 * for tests, bench.py and profiling only; results never depend on these calls and a front end has no use for
 * them.
 */
#ifndef ARKS_HIP_DEBUG_H
#define ARKS_HIP_DEBUG_H

#include "arks_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Lengths of the work queues after the last map call on `idx` (waits for the device): out4[0] = reads
 * that took the slow kernel, out4[2] = reads that took the medium kernel ([1], [3]: their work
 * counters).  The reads the hot kernel of bestContig (Arcs/Arcs.cpp:939-1014) did not finish itself. */
int arks_debug_queue_counts(const arks_index* idx, unsigned* out4);

#ifdef __cplusplus
}
#endif
#endif
