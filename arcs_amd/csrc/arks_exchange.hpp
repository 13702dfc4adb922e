// arks_exchange.hpp -- the sharded seed table as product code (BASELINE configs[3]; included by arks_capi.hip).
//
// One arks_exchange per rank: the rank's shard of the seed table (arks_index_build_seed_shard), its device buffers,
// and the transport that reaches the other ranks:
//   * RCCL (one process per GPU, xGMI): ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the caller's stream --
//     the library is opened on demand (dlopen: a process that never shards needs no RCCL, and a process that already
//     carries one -- PyTorch bundles its own -- keeps that single copy);
//   * local: the ranks are threads of ONE process (all shards on the visible device), buffers are handed over with
//     device copies behind a host barrier -- what a single-GPU box can run: tests, bench.py --sharded-index at N = 1.
// One call, arks_map_reads_exchanged_device, is the whole step for a batch of this rank's reads -- every rank calls it in
// step (the exchange is collective):
//   1. seeds listed and bucketed by owner on the device       launch_seed_buckets (count, scan, fill; arks_shard.hip)
//   2. counts to everybody (all-gather of world numbers)      one small device-to-host copy: the step's only host sync
//   3. seeds to their owners (8 B each)                       all-to-all #1
//   4. owners answer from their shard (16 B each)             seeds_probe_kernel
//   5. answers back, into send-buffer order                   all-to-all #2
//   6. the home finishes                                      map_reads_s_kernel<REMOTE> reads ans[2 slot[seed]]
// Replaces, for this path, what arcs_amd/dist.py did with torch sort / searchsorted / gather / index_put and two blocking
// all_to_all_single calls ("what's weak" 10).  Reference seam: the reads are independent
// (Arcs/Arcs.cpp:1169), the index is read-only while mapping (:969-971).
#pragma once
#include <chrono>
#include <condition_variable>
#include <dlfcn.h>

namespace {

// ---- RCCL, by name ------------------------------------------------------------------------------------------
struct Rccl
{
	typedef struct { char internal[128]; } UniqueId;
	typedef void* Comm;
	void* h = nullptr;
	int (*GetUniqueId)(UniqueId*) = nullptr;
	int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
	int (*CommDestroy)(Comm) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	int (*Send)(const void*, size_t, int, int, Comm, hipStream_t) = nullptr;
	int (*Recv)(void*, size_t, int, int, Comm, hipStream_t) = nullptr;
	int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
	const char* (*GetErrorString)(int) = nullptr;
	static constexpr int kUint64 = 5; // ncclUint64 (rccl.h: ncclInt8 0, Uint8 1, Int32 2, Uint32 3, Int64 4, Uint64 5)

	static Rccl*
	get()
	{
		static Rccl r;
		static std::once_flag once;
		std::call_once(once, [] {
			const char* names[] = { "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so" };
			for (const char* n : names) {
				r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); // the copy the process already carries
				if (r.h)
					break;
			}
			for (size_t i = 0; !r.h && i < sizeof names / sizeof names[0]; ++i)
				r.h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
			if (!r.h)
				return;
#define ARKS_RCCL_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.h, name))
			ARKS_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
			ARKS_RCCL_SYM(CommInitRank, "ncclCommInitRank");
			ARKS_RCCL_SYM(CommDestroy, "ncclCommDestroy");
			ARKS_RCCL_SYM(GroupStart, "ncclGroupStart");
			ARKS_RCCL_SYM(GroupEnd, "ncclGroupEnd");
			ARKS_RCCL_SYM(Send, "ncclSend");
			ARKS_RCCL_SYM(Recv, "ncclRecv");
			ARKS_RCCL_SYM(AllGather, "ncclAllGather");
			ARKS_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef ARKS_RCCL_SYM
			if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv ||
			    !r.AllGather || !r.GetErrorString)
				r.h = nullptr;
		});
		return r.h ? &r : nullptr;
	}
};
static_assert(sizeof(Rccl::UniqueId) == ARKS_EXCHANGE_ID_BYTES, "ncclUniqueId is 128 bytes");

int
fail_rccl(int e, const char* what)
{
	Rccl* r = Rccl::get();
	g_last_error = std::string(what) + ": " + (r ? r->GetErrorString(e) : "RCCL not loaded");
	return ARKS_ERR_HIP;
}
#define RCCL_TRY(expr)                                                                             \
	do {                                                                                           \
		int e_ = (expr);                                                                           \
		if (e_ != 0) {                                                                             \
			rc = fail_rccl(e_, #expr);                                                             \
			goto done;                                                                             \
		}                                                                                          \
	} while (0)

// ---- the ranks of one process -------------------------------------------------------------------------------
struct LocalGroup
{
	int world = 0;
	std::mutex m;
	std::condition_variable cv;
	int arrived = 0;
	u64 generation = 0;
	int members_alive = 0;
	// what every rank shows the others between two barriers
	std::vector<const u64*> counts_host; // 1 + world totals of the batch (pinned)
	std::vector<const u64*> send;        // bucketed seeds
	std::vector<const u64*> ans_out;     // answers to the seeds received
	std::vector<int> failed;

	bool aborted = false; // a rank left with an error (or never came): nobody waits for it again

	// false: the group is broken (a rank failed in an earlier call, or did not arrive within ten minutes)
	bool
	barrier()
	{
		std::unique_lock<std::mutex> lk(m);
		if (aborted)
			return false;
		const u64 gen = generation;
		if (++arrived == world) {
			arrived = 0;
			++generation;
			cv.notify_all();
			return true;
		}
		if (!cv.wait_for(lk, std::chrono::minutes(10), [&] { return generation != gen || aborted; }))
			aborted = true; // the ranks of a process call in step: one that is missing this long has died
		if (aborted) {
			cv.notify_all();
			return false;
		}
		return true;
	}
	void
	abort()
	{
		std::lock_guard<std::mutex> lk(m);
		aborted = true;
		cv.notify_all();
	}
};

struct GrowBuf
{
	void* p = nullptr;
	size_t cap = 0;
	hipError_t
	reserve(size_t bytes) // contents are not kept; the device is idle for this exchange when it grows
	{
		if (bytes <= cap)
			return hipSuccess;
		if (p)
			(void)hipFree(p);
		p = nullptr;
		cap = 0;
		const size_t want = bytes + bytes / 4 + 4096;
		hipError_t e = hipMalloc(&p, want);
		if (e == hipSuccess)
			cap = want;
		return e;
	}
	void
	release()
	{
		if (p)
			(void)hipFree(p);
		p = nullptr;
		cap = 0;
	}
	template <typename T>
	T*
	as() const
	{
		return static_cast<T*>(p);
	}
};

} // namespace

struct arks_exchange
{
	const arks_index* idx = nullptr;
	int rank = 0, world = 1, device = 0;
	Rccl::Comm comm = nullptr;   // RCCL transport
	LocalGroup* group = nullptr; // local transport (shared by the ranks of the process)
	GrowBuf cols, seed_off, slot, send, recv, ans_out, ans_back;
	u64* d_totals = nullptr; // 1 + world
	u64* d_all = nullptr;    // world x world: [p * world + o] = seeds rank p asks of owner o
	u64* h_totals = nullptr; // pinned: 1 + world, then world x world
	arks_exchange_stats last{};
	hipStream_t last_stream = nullptr; // the buffers are reused in the order of this stream
	bool used = false;
};

namespace {

void
exchange_release(arks_exchange* x)
{
	if (!x)
		return;
	DeviceGuard guard(x->device);
	(void)hipDeviceSynchronize();
	if (x->comm) {
		Rccl* r = Rccl::get();
		if (r)
			(void)r->CommDestroy(x->comm);
	}
	if (x->group) {
		bool last;
		{
			std::lock_guard<std::mutex> lk(x->group->m);
			last = --x->group->members_alive == 0;
		}
		if (last)
			delete x->group;
	}
	x->cols.release(), x->seed_off.release(), x->slot.release(), x->send.release(), x->recv.release();
	x->ans_out.release(), x->ans_back.release();
	if (x->d_totals)
		(void)hipFree(x->d_totals);
	if (x->d_all)
		(void)hipFree(x->d_all);
	if (x->h_totals)
		(void)hipHostFree(x->h_totals);
	delete x;
}

int
exchange_broken()
{
	g_last_error = "the local group is broken: a rank failed in this or an earlier call, or did not arrive";
	return ARKS_ERR_HIP;
}

int
exchange_new(arks_exchange** out, const arks_index* shard, int rank, int world)
{
	if (!out || !shard || shard->kind != 2 || world < 1 || world > 64 || rank < 0 || rank >= world ||
	    shard->seed_ranks != world || shard->seed_rank != rank)
		return ARKS_ERR_BAD_ARG;
	int rc = ARKS_OK;
	arks_exchange* x = new (std::nothrow) arks_exchange();
	if (!x)
		return ARKS_ERR_OOM;
	x->idx = shard, x->rank = rank, x->world = world, x->device = shard->device;
	DeviceGuard guard(x->device);
	void* p = nullptr;
	HIP_TRY(hipMalloc(&p, sizeof(u64) * (size_t)(1 + world)));
	x->d_totals = static_cast<u64*>(p);
	HIP_TRY(hipMalloc(&p, sizeof(u64) * (size_t)world * (size_t)world));
	x->d_all = static_cast<u64*>(p);
	HIP_TRY(hipHostMalloc(&p, sizeof(u64) * (size_t)(1 + world + world * world), hipHostMallocDefault));
	x->h_totals = static_cast<u64*>(p);
	*out = x;
	return ARKS_OK;
done:
	exchange_release(x);
	return rc;
}

} // namespace

extern "C" {

int
arks_exchange_unique_id(unsigned char* out_id)
{
	if (!out_id)
		return ARKS_ERR_BAD_ARG;
	Rccl* r = Rccl::get();
	if (!r) {
		g_last_error = "librccl.so.1 could not be opened";
		return ARKS_ERR_HIP;
	}
	Rccl::UniqueId id;
	int rc = ARKS_OK;
	RCCL_TRY(r->GetUniqueId(&id));
	std::memcpy(out_id, id.internal, sizeof id.internal);
done:
	return rc;
}

int
arks_exchange_create(arks_exchange** out, const arks_index* shard, const unsigned char* unique_id, int rank, int world)
{
	if (out)
		*out = nullptr;
	arks_exchange* x = nullptr;
	int rc = exchange_new(&x, shard, rank, world);
	if (rc != ARKS_OK)
		return rc;
	if (world > 1 || unique_id) {
		Rccl* r = Rccl::get();
		if (!r || !unique_id) {
			g_last_error = !unique_id ? "world > 1 needs the unique id of rank 0 (arks_exchange_unique_id)"
			                          : "librccl.so.1 could not be opened";
			exchange_release(x);
			return !unique_id ? ARKS_ERR_BAD_ARG : ARKS_ERR_HIP;
		}
		DeviceGuard guard(x->device);
		Rccl::UniqueId id;
		std::memcpy(id.internal, unique_id, sizeof id.internal);
		RCCL_TRY(r->CommInitRank(&x->comm, world, id, rank));
	}
	*out = x;
	return ARKS_OK;
done:
	exchange_release(x);
	return rc;
}

int
arks_exchange_create_local(arks_exchange** out, const arks_index* const* shards, int world)
{
	if (!out || !shards || world < 1 || world > 64)
		return ARKS_ERR_BAD_ARG;
	for (int r = 0; r < world; ++r)
		out[r] = nullptr;
	LocalGroup* g = new (std::nothrow) LocalGroup();
	if (!g)
		return ARKS_ERR_OOM;
	g->world = world;
	g->counts_host.assign((size_t)world, nullptr);
	g->send.assign((size_t)world, nullptr);
	g->ans_out.assign((size_t)world, nullptr);
	g->failed.assign((size_t)world, 0);
	int rc = ARKS_OK;
	for (int r = 0; r < world && rc == ARKS_OK; ++r) {
		rc = exchange_new(&out[r], shards[r], r, world);
		if (rc == ARKS_OK) {
			if (shards[r]->device != shards[0]->device)
				rc = ARKS_ERR_BAD_ARG; // the ranks of one process share one device
			out[r]->group = g;
			g->members_alive++;
		}
	}
	if (rc != ARKS_OK) {
		const bool any = g->members_alive > 0;
		for (int r = 0; r < world; ++r) {
			exchange_release(out[r]);
			out[r] = nullptr;
		}
		if (!any)
			delete g;
	}
	return rc;
}

int
arks_exchange_free(arks_exchange* x)
{
	exchange_release(x);
	return ARKS_OK;
}

int
arks_exchange_abort(arks_exchange* x)
{
	if (!x)
		return ARKS_ERR_BAD_ARG;
	if (x->group)
		x->group->abort();
	return ARKS_OK;
}

int
arks_exchange_last_stats(const arks_exchange* x, arks_exchange_stats* out)
{
	if (!x || !out)
		return ARKS_ERR_BAD_ARG;
	*out = x->last;
	return ARKS_OK;
}

int
arks_map_reads_exchanged_device(
    arks_exchange* x,
    const uint64_t* d_codes,
    const uint32_t* d_nmask,
    const uint64_t* d_word_off,
    const uint32_t* d_lens,
    const uint8_t* d_eval,
    int64_t n_reads,
    double j_index,
    int32_t* d_out_conreci,
    arks_map_stats* d_stats,
    void* stream)
{
	if (!x || n_reads < 0 || n_reads > 0xFFFFFFFFll)
		return ARKS_ERR_BAD_ARG;
	if (n_reads > 0 && (!d_codes || !d_nmask || !d_word_off || !d_lens || !d_out_conreci))
		return ARKS_ERR_BAD_ARG;
	const arks_index* idx = x->idx;
	const int W = x->world, me = x->rank;
	hipStream_t st = static_cast<hipStream_t>(stream);
	DeviceGuard guard(x->device);
	Rccl* rccl = x->comm ? Rccl::get() : nullptr;
	LocalGroup* g = x->group;
	int rc = ARKS_OK;
	// the buffers of the last call (its map kernel may still be reading them) are reused in stream order: a call on
	// another stream waits for the old one first
	if (x->used && x->last_stream != st)
		(void)hipStreamSynchronize(x->last_stream);
	x->last_stream = st;
	x->used = true;
	std::vector<u64> sc((size_t)W), rcv((size_t)W), soff((size_t)W + 1), roff((size_t)W + 1);
	u64 n_seeds = 0;
	const long nb = seed_bucket_blocks((long)n_reads);
	const u64* all = x->h_totals + 1 + W; // [p * W + o]
	// A local rank that fails must still meet the others at every barrier: errors are carried to the end.
#define EX_TRY(expr)                                                                               \
	do {                                                                                           \
		if (rc == ARKS_OK) {                                                                       \
			hipError_t e_ = (expr);                                                                \
			if (e_ != hipSuccess)                                                                  \
				rc = fail_hip(e_, #expr);                                                          \
		}                                                                                          \
	} while (0)
	// ---- 1. count + scan ------------------------------------------------------------------------------------
	EX_TRY(x->cols.reserve(sizeof(u32) * (size_t)(1 + W) * (size_t)(nb > 0 ? nb : 1)));
	EX_TRY(x->seed_off.reserve(sizeof(long) * ((size_t)n_reads + 1)));
	EX_TRY(launch_seed_buckets(
	    idx->bx.m, (const u64*)d_codes, d_nmask, (const u64*)d_word_off, d_lens, d_eval, (long)n_reads, idx->k, idx->bx.w,
	    (u32)W, x->cols.as<u32>(), x->d_totals, nullptr, nullptr, nullptr, 0, st));
	// ---- 2. everybody's counts ------------------------------------------------------------------------------
	EX_TRY(hipMemcpyAsync(x->h_totals, x->d_totals, sizeof(u64) * (size_t)(1 + W), hipMemcpyDeviceToHost, st));
	if (rccl && W > 1) {
		if (rc == ARKS_OK) {
			int e = rccl->AllGather(x->d_totals + 1, x->d_all, (size_t)W, Rccl::kUint64, x->comm, st);
			if (e != 0)
				rc = fail_rccl(e, "ncclAllGather(seed counts)");
		}
		EX_TRY(hipMemcpyAsync(x->h_totals + 1 + W, x->d_all, sizeof(u64) * (size_t)W * (size_t)W, hipMemcpyDeviceToHost, st));
	}
	EX_TRY(hipStreamSynchronize(st));
	if (g) {
		g->counts_host[(size_t)me] = x->h_totals;
		g->failed[(size_t)me] = rc != ARKS_OK;
		if (!g->barrier())
			return exchange_broken();
		for (int p = 0; p < W; ++p) {
			if (g->failed[(size_t)p] && rc == ARKS_OK) {
				g_last_error = "another rank of the local group failed";
				rc = ARKS_ERR_HIP;
			}
			for (int o = 0; o < W; ++o)
				x->h_totals[1 + W + p * W + o] = g->failed[(size_t)p] ? 0 : g->counts_host[(size_t)p][1 + o];
		}
		// (no second barrier: a rank overwrites its counts only in its NEXT call, which it enters behind the barriers
		// below -- and those every rank reaches after it has read the counts)
	} else if (W == 1)
		x->h_totals[2] = x->h_totals[1];
	if (rc != ARKS_OK && !g)
		return rc;
	n_seeds = x->h_totals[0];
	soff[0] = roff[0] = 0;
	for (int p = 0; p < W; ++p) {
		sc[(size_t)p] = all[(size_t)me * W + p];  // what I ask of p
		rcv[(size_t)p] = all[(size_t)p * W + me]; // what p asks of me
		soff[(size_t)p + 1] = soff[(size_t)p] + sc[(size_t)p];
		roff[(size_t)p + 1] = roff[(size_t)p] + rcv[(size_t)p];
	}
	{
		const u64 S = soff[(size_t)W], R = roff[(size_t)W];
		if (S > 0xFFFFFFFEull || n_seeds > 0xFFFFFFFEull)
			rc = rc == ARKS_OK ? ARKS_ERR_BAD_ARG : rc; // slots are 32-bit: split the batch
		EX_TRY(x->slot.reserve(sizeof(u32) * (size_t)(n_seeds + 1)));
		EX_TRY(x->send.reserve(sizeof(u64) * (size_t)(S + 1)));
		EX_TRY(x->ans_back.reserve(2 * sizeof(u64) * (size_t)(S + 1)));
		EX_TRY(x->recv.reserve(sizeof(u64) * (size_t)(R + 1)));
		EX_TRY(x->ans_out.reserve(2 * sizeof(u64) * (size_t)(R + 1)));
		x->last.seeds = n_seeds, x->last.sent = S - sc[(size_t)me], x->last.received = R - rcv[(size_t)me];
	}
	// ---- 3. fill, seeds to their owners ---------------------------------------------------------------------
	EX_TRY(launch_seed_buckets(
	    idx->bx.m, (const u64*)d_codes, d_nmask, (const u64*)d_word_off, d_lens, d_eval, (long)n_reads, idx->k, idx->bx.w,
	    (u32)W, x->cols.as<u32>(), x->d_totals, x->seed_off.as<long>(), x->slot.as<u32>(), x->send.as<u64>(), 1, st));
	// (my own seeds stay where they are: they are answered straight from the send buffer into the answers' place, below)
	if (rccl && W > 1 && rc == ARKS_OK) {
		RCCL_TRY(rccl->GroupStart());
		for (int p = 0; p < W; ++p) {
			if (p == me)
				continue;
			if (sc[(size_t)p])
				RCCL_TRY(rccl->Send(x->send.as<u64>() + soff[(size_t)p], sc[(size_t)p], Rccl::kUint64, p, x->comm, st));
			if (rcv[(size_t)p])
				RCCL_TRY(rccl->Recv(x->recv.as<u64>() + roff[(size_t)p], rcv[(size_t)p], Rccl::kUint64, p, x->comm, st));
		}
		RCCL_TRY(rccl->GroupEnd());
	} else if (g) {
		EX_TRY(hipStreamSynchronize(st)); // my send buffer is complete
		g->send[(size_t)me] = x->send.as<u64>();
		g->failed[(size_t)me] = rc != ARKS_OK;
		if (!g->barrier())
			return exchange_broken();
		for (int p = 0; p < W; ++p) {
			if (g->failed[(size_t)p] && rc == ARKS_OK) {
				g_last_error = "another rank of the local group failed";
				rc = ARKS_ERR_HIP;
			}
			if (p == me || !rcv[(size_t)p] || rc != ARKS_OK)
				continue;
			// p's seeds for me start behind what p asks of the owners in front of me
			u64 off = 0;
			for (int o = 0; o < me; ++o)
				off += all[(size_t)p * W + o];
			EX_TRY(hipMemcpyAsync(x->recv.as<u64>() + roff[(size_t)p], g->send[(size_t)p] + off, sizeof(u64) * rcv[(size_t)p],
			                      hipMemcpyDeviceToDevice, st));
		}
	}
	// ---- 4. the owner's answers -----------------------------------------------------------------------------
	// what the ranks in front of me asked, what the ranks behind me asked (my own part of the receive buffer lies between
	// them, unused), and my own seeds from where they are to where their answers belong
	EX_TRY(launch_seeds_probe(idx->bx.m, idx->bx, x->recv.as<u64>(), (long)roff[(size_t)me], x->ans_out.as<u64>(), st));
	EX_TRY(launch_seeds_probe(idx->bx.m, idx->bx, x->recv.as<u64>() + roff[(size_t)me + 1],
	                          (long)(roff[(size_t)W] - roff[(size_t)me + 1]), x->ans_out.as<u64>() + 2 * roff[(size_t)me + 1], st));
	EX_TRY(launch_seeds_probe(idx->bx.m, idx->bx, x->send.as<u64>() + soff[(size_t)me], (long)sc[(size_t)me],
	                          x->ans_back.as<u64>() + 2 * soff[(size_t)me], st));
	// ---- 5. answers back ------------------------------------------------------------------------------------
	if (rccl && W > 1 && rc == ARKS_OK) {
		RCCL_TRY(rccl->GroupStart());
		for (int p = 0; p < W; ++p) {
			if (p == me)
				continue;
			if (rcv[(size_t)p])
				RCCL_TRY(rccl->Send(x->ans_out.as<u64>() + 2 * roff[(size_t)p], 2 * rcv[(size_t)p], Rccl::kUint64, p, x->comm, st));
			if (sc[(size_t)p])
				RCCL_TRY(rccl->Recv(x->ans_back.as<u64>() + 2 * soff[(size_t)p], 2 * sc[(size_t)p], Rccl::kUint64, p, x->comm, st));
		}
		RCCL_TRY(rccl->GroupEnd());
	} else if (g) {
		EX_TRY(hipStreamSynchronize(st)); // my answers are complete (and I have read the others' seeds)
		g->ans_out[(size_t)me] = x->ans_out.as<u64>();
		g->failed[(size_t)me] = rc != ARKS_OK;
		if (!g->barrier())
			return exchange_broken();
		for (int p = 0; p < W; ++p) {
			if (g->failed[(size_t)p] && rc == ARKS_OK) {
				g_last_error = "another rank of the local group failed";
				rc = ARKS_ERR_HIP;
			}
			if (p == me || !sc[(size_t)p] || rc != ARKS_OK)
				continue;
			// my seeds lie in p's receive buffer behind those of the ranks in front of me
			u64 off = 0;
			for (int q = 0; q < me; ++q)
				off += all[(size_t)q * W + p];
			EX_TRY(hipMemcpyAsync(x->ans_back.as<u64>() + 2 * soff[(size_t)p], g->ans_out[(size_t)p] + 2 * off,
			                      2 * sizeof(u64) * sc[(size_t)p], hipMemcpyDeviceToDevice, st));
		}
		// (no wait here: the copies above are on this rank's stream; its next call -- on the same stream, see
		// arks_hip.h -- synchronises that stream before it lets any other rank write the buffers they read from,
		// so the map kernel below and the pair rule behind it overlap with the other ranks' next batch)
	}
	// ---- 6. the home finishes -------------------------------------------------------------------------------
	if (rc == ARKS_OK && n_reads > 0) {
		arks_index::QueueSet qs;
		rc = ensure_queue(idx, stream, n_reads, &qs);
		if (rc == ARKS_OK)
			EX_TRY(launch_map_reads_seeded(
			    idx->kw, (const u64*)d_codes, d_nmask, (const u64*)d_word_off, d_lens, d_eval, (long)n_reads, j_index, idx->geom,
			    idx->bx, idx->bxg, x->seed_off.as<long>(), x->ans_back.as<u64>(), d_out_conreci, reinterpret_cast<u64*>(d_stats),
			    qs.queue, qs.queue_count, idx->n_cu, st, x->slot.as<u32>()));
	}
done:
#undef EX_TRY
	if (rc != ARKS_OK && g)
		g->abort(); // this rank's caller will not call again: the others must not wait for it
	return rc;
}

} // extern "C"
