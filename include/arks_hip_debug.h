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

/* n >= 1: the medium kernel (map_reads_b_kernel) of every later map call of this process on at most n waves, so that
 * even a test's short queue gives every wave several reads per grab (tiles of several gathered reads: the path a long
 * queue takes); 0: as many as the launch wants.  (Rounds 1-5: the environment variable ARKS_DEBUG_MEDIUM_BLOCKS, read
 * at every launch.) */
int arks_debug_set_medium_blocks(int n);

/* The RCCL entry points arks_exchange reaches (ncclSend / ncclRecv groups, ncclAllGather ...), with RCCL's own
 * signatures.  By default the table is filled from librccl.so.1 (dlopen).  A test installs stand-ins -- threads of
 * one process and device copies (tests/mock_rccl.cpp) -- and so runs the library's world > 1 code, which RCCL itself
 * refuses to do on a box with one GPU (two ranks may not share a device). */
typedef struct
{
	char internal[128];
} arks_rccl_unique_id; /* = ncclUniqueId */
typedef struct
{
	int (*GetVersion)(int*);                                               /* optional */
	int (*GetUniqueId)(arks_rccl_unique_id*);
	int (*CommInitRank)(void** comm, int world, arks_rccl_unique_id id, int rank);
	int (*CommDestroy)(void* comm);
	int (*CommAbort)(void* comm);                                          /* optional */
	int (*GroupStart)(void);
	int (*GroupEnd)(void);
	int (*Send)(const void* buf, size_t count, int dtype, int peer, void* comm, void* stream);
	int (*Recv)(void* buf, size_t count, int dtype, int peer, void* comm, void* stream);
	int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, void* stream);
	const char* (*GetErrorString)(int);
} arks_rccl_api;
/* api = NULL: librccl again.  The table must outlive every exchange made with it. */
int arks_exchange_debug_set_rccl(const arks_rccl_api* api);

#ifdef __cplusplus
}
#endif
#endif
