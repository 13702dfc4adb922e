// arks_exchange.hpp -- the sharded seed table as product code (BASELINE configs[3]; included by arks_capi.hip).
//
// One arks_exchange per rank: the rank's shard of the seed table (arks_index_build_seed_shard), two sets of device
// buffers (two batches in flight), and the transport that reaches the other ranks:
//   * RCCL (one process per GPU, xGMI): ncclAllGather of the counts, ncclGroupStart / ncclSend / ncclRecv /
//     ncclGroupEnd for the seeds and for the answers -- the library is opened on demand (dlopen: a process that never
//     shards needs no RCCL, and a process that already carries one -- PyTorch bundles its own -- keeps that copy);
//     every entry point is reached through one table of function pointers (arks_rccl_api), which a test replaces
//     with threads-and-device-copies stand-ins, so that the code below runs with world > 1 on a one-GPU box;
//   * direct: the ranks are threads of ONE process whose devices reach each other's memory (the same device, or
//     peers over xGMI): nothing is copied at all -- the owner's probe kernel reads the askers' send buffers where
//     they lie and writes its answers into the askers' answer buffers; host barriers and events order it.
// A batch takes two calls (round 4; one until round 3, with a host wait in the middle of it):
//   arks_exchange_submit    this rank alone: seeds listed and bucketed by owner in ONE launch (seed_bucket_kernel),
//                           counts on their way to the host
//   arks_exchange_complete  COLLECTIVE: counts to everybody, seeds to their owners (8 B each), owner-side probe
//                           (16 B back each), answers home, map_reads_s_kernel<REMOTE> reads ans[2 slot[seed]]
// With batch n + 1 submitted before batch n is completed (two streams), the host never waits for the device: the
// counts of batch n have long arrived when complete(n) asks for them, and batch n + 1's bucketing and batch n's
// probes, transfers and map kernel overlap.  arks_map_reads_exchanged_device = submit + complete (one batch at a time).
// Replaces, for this path, what arcs_amd/dist.py did with torch sort / searchsorted / gather / index_put and two blocking
// all_to_all_single calls ("what's weak" 10).  Reference seam: the reads are independent
// (Arcs/Arcs.cpp:1169), the index is read-only while mapping (:969-971).
#pragma once
#include "arks_hip_debug.h"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <dlfcn.h>

namespace {

// ---- RCCL, by name ------------------------------------------------------------------------------------------
constexpr int kNcclUint8 = 1, kNcclUint64 = 5; // rccl.h: ncclInt8 0, Uint8 1, Int32 2, Uint32 3, Int64 4, Uint64 5
static_assert(sizeof(arks_rccl_unique_id) == ARKS_EXCHANGE_ID_BYTES, "ncclUniqueId is 128 bytes");

std::atomic<const arks_rccl_api*> g_rccl_override{ nullptr };

const arks_rccl_api*
rccl_api()
{
	if (const arks_rccl_api* o = g_rccl_override.load(std::memory_order_acquire))
		return o;
	static arks_rccl_api r{};
	static void* h = nullptr;
	static std::once_flag once;
	std::call_once(once, [] {
		const char* names[] = { "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so" };
		for (const char* n : names) {
			h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); // the copy the process already carries
			if (h)
				break;
		}
		for (size_t i = 0; !h && i < sizeof names / sizeof names[0]; ++i)
			h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
		if (!h)
			return;
#define ARKS_RCCL_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name))
		ARKS_RCCL_SYM(GetVersion, "ncclGetVersion");
		ARKS_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
		ARKS_RCCL_SYM(CommInitRank, "ncclCommInitRank");
		ARKS_RCCL_SYM(CommDestroy, "ncclCommDestroy");
		ARKS_RCCL_SYM(CommAbort, "ncclCommAbort");
		ARKS_RCCL_SYM(GroupStart, "ncclGroupStart");
		ARKS_RCCL_SYM(GroupEnd, "ncclGroupEnd");
		ARKS_RCCL_SYM(Send, "ncclSend");
		ARKS_RCCL_SYM(Recv, "ncclRecv");
		ARKS_RCCL_SYM(AllGather, "ncclAllGather");
		ARKS_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef ARKS_RCCL_SYM
		if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv ||
		    !r.AllGather || !r.GetErrorString)
			h = nullptr; // (GetVersion and CommAbort are optional)
	});
	return h ? &r : nullptr;
}

int
fail_rccl(int e, const char* what)
{
	const arks_rccl_api* r = rccl_api();
	g_last_error = std::string(what) + ": " + (r ? r->GetErrorString(e) : "RCCL not loaded");
	return ARKS_ERR_HIP;
}

// ---- the ranks of one process -------------------------------------------------------------------------------
struct LocalGroup
{
	int world = 0;
	std::mutex m;
	std::condition_variable cv;
	int arrived = 0;
	u64 generation = 0;
	int members_alive = 0;
	bool aborted = false; // a rank left with an error (or never came): nobody waits for it again
	// what a rank shows the others for the batch in buffer set s (written before a barrier, read behind it)
	struct Shown
	{
		int status = 0, status2 = 0; // before the first / the second barrier of a batch
		const u64* counts = nullptr; // the batch's seeds per owner (ExSet::counts)
		u64 cap = 0;                 // region size of its send / answer buffers
		const u64* send = nullptr;
		u64* ans_back = nullptr;
		hipEvent_t probed = nullptr; // its answers to everybody are written when this event has happened
	};
	std::vector<Shown> shown[2]; // [buffer set][rank]

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
	// Contents are not kept.  Who may still touch the old block: the set's own kernels (its last map kernel reads slot,
	// chunk_off and the answers) and, in a local group, the peers' probe kernels (they read `send` and write `ans_back`)
	// -- and the set's stream has waited for every one of those (direct_finish: the owners' `probed` events) before its
	// map kernel.  So the block is free once `user` -- the stream the set was last used on -- has drained: waited for
	// HERE, explicitly (until round 4 this leaned on hipFree's device-wide wait, which also stalled the other
	// batch in flight).
	hipError_t
	reserve(size_t bytes, hipStream_t user, u64* drains)
	{
		if (bytes <= cap)
			return hipSuccess;
		if (p) {
			++*drains; // (arks_exchange_stats::stream_syncs)
			const hipError_t e = hipStreamSynchronize(user);
			if (e != hipSuccess)
				return e;
			(void)hipFree(p);
		}
		p = nullptr;
		cap = 0;
		const size_t want = bytes + bytes / 8 + 4096;
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

// one batch in flight: its buffers and what complete() needs to know of it
struct ExSet
{
	GrowBuf send, ans_back; // W regions of `cap` seeds (8 B) / answers (16 B), region o = what owner o is asked
	GrowBuf slot, chunk_off;
	GrowBuf recv, ans_out; // RCCL transport: what the others ask of me, and my answers, asker after asker
	arks::SeedBucketCtl* d_ctl = nullptr;
	arks::SeedBucketCtl* h_ctl = nullptr; // pinned
	u64* d_gather = nullptr;              // RCCL: 1 + W of mine, then world x (1 + W) of everybody
	u64* h_gather = nullptr;              // pinned, the same
	hipEvent_t counted = nullptr, probed = nullptr;
	bool gathered = false; // RCCL: the counts all-gather of this batch was enqueued at submit (behind `counted`'s copy)
	bool status_sent = false; // ... and carried this rank's failure (the peers know; nobody waits for it)
	hipStream_t st = nullptr;
	bool used = false, pending = false;
	int rc = ARKS_OK;
	u64 cap = 0, slot_cap = 0;
	u64 cap_floor = 0, slot_floor = 0; // set while a batch that did not fit is run again
	arks::SeedBucketBase base{};   // where d_ctl's counters stood before this batch's launch (they only count up)
	u64 launches = 0;              // bucket launches on d_ctl so far (SeedBucketBase::seq of the next one - 1)
	u64 counts[64] = { 0 };        // the batch's seeds per owner, and ...
	u64 n_seeds = 0;               // ... in all (differences of the counters; valid after ex_counts)
	arks_index::QueueSet qs{};     // the map kernels' queues and scratch on this batch's stream (scratch zeroed by the bucket launch)
	bool scratch_zeroed = false;
	const uint8_t* read_class = nullptr; // arks_exchange_submit_pairs: the gate is computed by the bucket kernel ...
	const uint8_t* pair_ok = nullptr;
	uint8_t* eval_out = nullptr;         // ... and written here (the map kernels read it)
	std::vector<u64> all, sc, rcv, soff, roff; // the batch's counts: everybody's, and this rank's tables
	// the batch
	const u64* codes = nullptr;
	const u32* nmask = nullptr;
	const u64* word_off = nullptr;
	const u32* lens = nullptr;
	const uint8_t* eval = nullptr;
	long n_reads = 0;
	double j_index = 0;
	int32_t* out = nullptr;
	arks_map_stats* stats = nullptr;
};

} // namespace

struct arks_exchange
{
	const arks_index* idx = nullptr;
	int rank = 0, world = 1, device = 0;
	void* comm = nullptr;        // RCCL transport
	LocalGroup* group = nullptr; // direct transport (shared by the ranks of the process)
	ExSet set[2];
	u64 submitted = 0, completed = 0;
	// seeds per read the regions are sized for: per owner, and in all (raised when a batch does not fit)
	// (first guess: a 10x read pair has 2 + 3 seeds at k = 60; a batch of another shape does not fit, says what it needs,
	// and is bucketed again -- once per shape; 3.3 until round 4: 30 % of the buffers of a 10x batch were never used)
	u64 stream_syncs = 0; // times the host drained a stream (arks_exchange_stats): a buffer grew, a set changed streams
	double per_read_owner = 0, per_read_all = 2.65;
	long largest_batch = 0;
	u64 reruns = 0;
	arks_exchange_stats last{};
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
		if (const arks_rccl_api* r = rccl_api())
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
	for (ExSet& s : x->set) {
		s.send.release(), s.ans_back.release(), s.slot.release(), s.chunk_off.release(), s.recv.release(), s.ans_out.release();
		if (s.d_ctl)
			(void)hipFree(s.d_ctl);
		if (s.h_ctl)
			(void)hipHostFree(s.h_ctl);
		if (s.d_gather)
			(void)hipFree(s.d_gather);
		if (s.h_gather)
			(void)hipHostFree(s.h_gather);
		if (s.counted)
			(void)hipEventDestroy(s.counted);
		if (s.probed)
			(void)hipEventDestroy(s.probed);
	}
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
	x->per_read_owner = 2.65 / world * 1.06;
	DeviceGuard guard(x->device);
	const size_t gw = (size_t)(1 + world) * (size_t)(1 + world);
	for (ExSet& s : x->set) {
		void* p = nullptr;
		HIP_TRY(hipMalloc(&p, sizeof(arks::SeedBucketCtl)));
		s.d_ctl = static_cast<arks::SeedBucketCtl*>(p);
		HIP_TRY(hipMemset(s.d_ctl, 0, sizeof(arks::SeedBucketCtl))); // once: the counters only count up from here
		HIP_TRY(hipHostMalloc(&p, sizeof(arks::SeedBucketCtl), hipHostMallocDefault));
		s.h_ctl = static_cast<arks::SeedBucketCtl*>(p);
		HIP_TRY(hipMalloc(&p, sizeof(u64) * gw));
		s.d_gather = static_cast<u64*>(p);
		HIP_TRY(hipHostMalloc(&p, sizeof(u64) * gw, hipHostMallocDefault));
		s.h_gather = static_cast<u64*>(p);
		HIP_TRY(hipEventCreateWithFlags(&s.counted, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&s.probed, hipEventDisableTiming));
	}
	*out = x;
	return ARKS_OK;
done:
	exchange_release(x);
	return rc;
}

// RCCL's ncclUint64 is what this file believes it to be (and the transport moves whole buffers): a send to
// oneself of a known pattern, checked byte for byte.  Runs once per communicator.
int
rccl_self_test(arks_exchange* x, const arks_rccl_api* r)
{
	int rc = ARKS_OK;
	u64* d = nullptr;
	u64 h[8] = { 0x0123456789ABCDEFull, 0xFEDCBA9876543210ull, 0x1111111111111111ull, ~0ull, 0, 0, 0, 0 };
	hipStream_t st = nullptr;
	int e = 0, e2 = 0;
	HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), sizeof h));
	HIP_TRY(hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice));
	e = r->GroupStart();
	if (e == 0) {
		e = r->Send(d, 4, kNcclUint64, x->rank, x->comm, st);
		if (e == 0)
			e = r->Recv(d + 4, 4, kNcclUint64, x->rank, x->comm, st);
		e2 = r->GroupEnd(); // (always: an open group would swallow every later call of this thread)
		if (e == 0)
			e = e2;
	}
	if (e != 0) {
		rc = fail_rccl(e, "RCCL self test (send to self)");
		goto done;
	}
	HIP_TRY(hipStreamSynchronize(st));
	HIP_TRY(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
	if (std::memcmp(h, h + 4, 4 * sizeof(u64)) != 0) {
		g_last_error = "RCCL self test: 4 x ncclUint64 did not move 32 bytes (data type numbering changed?)";
		rc = ARKS_ERR_HIP;
	}
done:
	if (d)
		(void)hipFree(d);
	if (st)
		(void)hipStreamDestroy(st);
	return rc;
}

// seeds (8 B, `unit` = 1) or answers (16 B, `unit` = 2) between all ranks: rank p gets cnt_to[p] items from
// mine + off_to[p] and I get cnt_from[p] items from it at theirs + off_from[p] (offsets and counts in items).
// The group is closed on every path; an error inside it aborts the communicator, so that the peers -- who may
// already wait in their own group for what this rank will never send -- get an error instead of a hang.
int
rccl_all_to_all(
    arks_exchange* x, const arks_rccl_api* r, const u64* mine, const u64* off_to, const u64* cnt_to, u64* theirs,
    const u64* off_from, const u64* cnt_from, u64 unit, hipStream_t st, const char* what)
{
	int e = r->GroupStart();
	if (e != 0)
		return fail_rccl(e, what);
	for (int p = 0; p < x->world && e == 0; ++p) {
		if (p == x->rank)
			continue;
		if (cnt_to[p])
			e = r->Send(mine + unit * off_to[p], unit * cnt_to[p], kNcclUint64, p, x->comm, st);
		if (e == 0 && cnt_from[p])
			e = r->Recv(theirs + unit * off_from[p], unit * cnt_from[p], kNcclUint64, p, x->comm, st);
	}
	const int e2 = r->GroupEnd();
	if (e == 0)
		e = e2;
	if (e != 0) {
		const int rc = fail_rccl(e, what);
		if (r->CommAbort && x->comm) {
			(void)r->CommAbort(x->comm);
			x->comm = nullptr; // (aborting frees it)
		}
		return rc;
	}
	return ARKS_OK;
}

// a kernel on device da reads a buffer that lives on device db and writes into another one there; db's stream waits for
// da's event and copies the result home: the direct transport's path, once, with a known answer
int
peer_path_check(int da, int db)
{
	constexpr int kN = 4096;
	int rc = ARKS_OK;
	u64 *src = nullptr, *dst = nullptr;
	hipStream_t sa = nullptr, sb = nullptr;
	hipEvent_t written = nullptr;
	std::vector<u64> h((size_t)kN), back((size_t)kN, 0);
	const u64 salt = 0x9E3779B97F4A7C15ull ^ ((u64)da << 32) ^ (u64)db;
	for (int i = 0; i < kN; ++i)
		h[(size_t)i] = 0x0123456789ABCDEFull * (u64)(i + 1);
	{
		DeviceGuard gb(db);
		HIP_TRY(hipMalloc(reinterpret_cast<void**>(&src), sizeof(u64) * kN));
		HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dst), sizeof(u64) * kN));
		HIP_TRY(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
		HIP_TRY(hipMemcpy(src, h.data(), sizeof(u64) * kN, hipMemcpyHostToDevice));
		HIP_TRY(hipMemset(dst, 0, sizeof(u64) * kN));
		HIP_TRY(hipDeviceSynchronize());
	}
	{
		DeviceGuard ga(da);
		HIP_TRY(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
		HIP_TRY(hipEventCreateWithFlags(&written, hipEventDisableTiming));
		HIP_TRY(arks::launch_peer_pattern(src, dst, kN, salt, sa)); // (reads and writes device db's memory)
		HIP_TRY(hipEventRecord(written, sa));
	}
	{
		DeviceGuard gb(db);
		HIP_TRY(hipStreamWaitEvent(sb, written, 0));
		HIP_TRY(hipMemcpyAsync(back.data(), dst, sizeof(u64) * kN, hipMemcpyDeviceToHost, sb));
		HIP_TRY(hipStreamSynchronize(sb));
	}
	for (int i = 0; i < kN && rc == ARKS_OK; ++i)
		if (back[(size_t)i] != (h[(size_t)i] ^ salt ^ (u64)i)) {
			g_last_error = "local group: a kernel on device " + std::to_string(da) + " does not reach the memory of device " +
			               std::to_string(db) + " (word " + std::to_string(i) + " of the check pattern came back wrong): peer access is "
			               "enabled but does not deliver";
			rc = ARKS_ERR_HIP;
		}
done:
	{
		DeviceGuard ga(da);
		if (sa)
			(void)hipStreamSynchronize(sa), (void)hipStreamDestroy(sa);
		if (written)
			(void)hipEventDestroy(written);
	}
	{
		DeviceGuard gb(db);
		if (sb)
			(void)hipStreamDestroy(sb);
		if (src)
			(void)hipFree(src);
		if (dst)
			(void)hipFree(dst);
	}
	return rc;
}

} // namespace

extern "C" {

int
arks_exchange_debug_set_rccl(const arks_rccl_api* api)
{
	g_rccl_override.store(api, std::memory_order_release);
	return ARKS_OK;
}

int
arks_exchange_unique_id(unsigned char* out_id)
{
	if (!out_id)
		return ARKS_ERR_BAD_ARG;
	const arks_rccl_api* r = rccl_api();
	if (!r) {
		g_last_error = "librccl.so.1 could not be opened";
		return ARKS_ERR_HIP;
	}
	arks_rccl_unique_id id;
	const int e = r->GetUniqueId(&id);
	if (e != 0)
		return fail_rccl(e, "ncclGetUniqueId");
	std::memcpy(out_id, id.internal, sizeof id.internal);
	return ARKS_OK;
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
		const arks_rccl_api* r = rccl_api();
		if (!r || !unique_id) {
			g_last_error = !unique_id ? "world > 1 needs the unique id of rank 0 (arks_exchange_unique_id)"
			                          : "librccl.so.1 could not be opened";
			exchange_release(x);
			return !unique_id ? ARKS_ERR_BAD_ARG : ARKS_ERR_HIP;
		}
		DeviceGuard guard(x->device);
		arks_rccl_unique_id id;
		std::memcpy(id.internal, unique_id, sizeof id.internal);
		const int e = r->CommInitRank(&x->comm, world, id, rank);
		if (e != 0) {
			x->comm = nullptr;
			rc = fail_rccl(e, "ncclCommInitRank");
		} else
			rc = rccl_self_test(x, r);
		if (rc != ARKS_OK) {
			exchange_release(x);
			return rc;
		}
	}
	*out = x;
	return ARKS_OK;
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
	g->shown[0].assign((size_t)world, LocalGroup::Shown());
	g->shown[1].assign((size_t)world, LocalGroup::Shown());
	int rc = ARKS_OK;
	for (int r = 0; r < world && rc == ARKS_OK; ++r) {
		rc = exchange_new(&out[r], shards[r], r, world);
		if (rc == ARKS_OK) {
			out[r]->group = g;
			g->members_alive++;
		}
	}
	// the ranks' devices must reach each other's memory: the same device, or peers (xGMI) -- the owner's probe kernel
	// reads the askers' buffers and writes into them
	for (int a = 0; a < world && rc == ARKS_OK; ++a)
		for (int b = 0; b < world && rc == ARKS_OK; ++b) {
			const int da = shards[a]->device, db = shards[b]->device;
			if (da == db)
				continue;
			int can = 0;
			if (hipDeviceCanAccessPeer(&can, da, db) != hipSuccess || !can) {
				g_last_error = "the devices of a local group must be able to access each other's memory";
				rc = ARKS_ERR_BAD_ARG;
				break;
			}
			DeviceGuard guard(da);
			const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
			if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
				rc = fail_hip(e, "hipDeviceEnablePeerAccess");
			(void)hipGetLastError();
		}
	// ... and do, before the first batch: the very path of a batch -- a kernel of device a reads a buffer of device b and
	// writes into another, b's stream waits for a's event and reads the result -- run once per ordered pair of devices
	// (once on itself where all ranks share one device) and checked word for word.  A peer mapping that is enabled but
	// does not deliver (IOMMU, a missing xGMI link, HSA_ENABLE_IPC_MODE_LEGACY left on) fails HERE, with a message,
	// not as wrong votes three stages later.
	if (rc == ARKS_OK) {
		std::vector<std::pair<int, int>> done_pairs;
		for (int a = 0; a < world && rc == ARKS_OK; ++a)
			for (int b = 0; b < world && rc == ARKS_OK; ++b) {
				const int da = shards[a]->device, db = shards[b]->device;
				bool seen = false;
				for (const auto& pr : done_pairs)
					seen = seen || (pr.first == da && pr.second == db);
				if (seen || (da == db && !done_pairs.empty()))
					continue;
				done_pairs.emplace_back(da, db);
				rc = peer_path_check(da, db);
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

} // extern "C"

namespace {

// the batch of set s bucketed with the current sizes (on its stream), counts on their way to the host
int
exchange_bucket(arks_exchange* x, ExSet& s)
{
	const arks_index* idx = x->idx;
	const int W = x->world;
	int rc = ARKS_OK;
	u64 cap = (u64)(x->per_read_owner * (double)s.n_reads) + 8192;
	u64 slot_cap = (u64)(x->per_read_all * (double)s.n_reads) + 8192;
	cap = cap < s.cap_floor ? s.cap_floor : cap;           // (a batch that is run again: what its counters asked for)
	slot_cap = slot_cap < s.slot_floor ? s.slot_floor : slot_cap;
	if ((long)s.n_reads > x->largest_batch)
		x->largest_batch = (long)s.n_reads;
	if (cap * (u64)W > 0xFFFFFFFEull || slot_cap > 0xFFFFFFFEull) {
		g_last_error = "a batch of the exchange holds more than 2^32 seeds: split it";
		return ARKS_ERR_BAD_ARG;
	}
	s.cap = cap, s.slot_cap = slot_cap;
	HIP_TRY(s.send.reserve(sizeof(u64) * (size_t)cap * (size_t)W, s.st, &x->stream_syncs));
	HIP_TRY(s.ans_back.reserve(2 * sizeof(u64) * (size_t)cap * (size_t)W, s.st, &x->stream_syncs));
	HIP_TRY(s.slot.reserve(sizeof(u32) * (size_t)slot_cap, s.st, &x->stream_syncs));
	HIP_TRY(s.chunk_off.reserve(sizeof(u32) * (size_t)(arks::seed_bucket_chunks(s.n_reads) + 1), s.st, &x->stream_syncs));
	// the map kernels' queues of this stream: made (or grown) here, so that the bucket launch can zero their scratch
	// block on the side -- one launch less per batch
	s.scratch_zeroed = false;
	if (s.n_reads > 0) {
		rc = ensure_queue(idx, s.st, s.n_reads, &s.qs);
		if (rc != ARKS_OK)
			goto done;
		s.scratch_zeroed = true;
	}
	{
		// the counters stand where the set's last launch left them: h_ctl holds that (the launch before this one on
		// this set is complete -- its counts were waited for in ex_counts, or this is the set's first)
		for (int o = 0; o < 64; ++o)
			s.base.fill[o] = s.launches ? s.h_ctl->fill[o * arks::kCtlStride] : 0;
		s.base.seeds = s.launches ? s.h_ctl->seeds[0] : 0;
		s.base.seq = ++s.launches;
		arks::SeedBucketGate gate;
		gate.read_class = s.read_class, gate.pair_ok = s.pair_ok, gate.eval_out = s.eval_out;
		HIP_TRY(arks::launch_seed_bucket(
		    idx->bx.m, s.codes, s.nmask, s.word_off, s.lens, s.eval, s.n_reads, idx->k, idx->bx.w, (u32)W, cap, slot_cap, s.d_ctl,
		    s.base, gate, s.scratch_zeroed ? s.qs.queue_count : nullptr, (int)(kMapScratchBytes / sizeof(u32)),
		    s.chunk_off.as<u32>(), s.slot.as<u32>(), s.send.as<u64>(), s.st));
	}
	HIP_TRY(hipMemcpyAsync(s.h_ctl, s.d_ctl, sizeof(arks::SeedBucketCtl), hipMemcpyDeviceToHost, s.st));
	HIP_TRY(hipEventRecord(s.counted, s.st));
done:
	return rc;
}

} // namespace

namespace {

int
exchange_submit(
    arks_exchange* x, const uint64_t* d_codes, const uint32_t* d_nmask, const uint64_t* d_word_off, const uint32_t* d_lens,
    const uint8_t* d_eval, const uint8_t* d_read_class, const uint8_t* d_pair_ok, uint8_t* d_eval_out, int64_t n_reads,
    double j_index, int32_t* d_out_conreci, arks_map_stats* d_stats, void* stream)
{
	if (!x || n_reads < 0 || n_reads > 0xFFFFFFFFll)
		return ARKS_ERR_BAD_ARG;
	if (n_reads > 0 && (!d_codes || !d_nmask || !d_word_off || !d_lens || !d_out_conreci))
		return ARKS_ERR_BAD_ARG;
	if (d_read_class && n_reads > 0 && (!d_eval_out || (n_reads & 1)))
		return ARKS_ERR_BAD_ARG; // (the gate is a rule over PAIRS: reads 2p, 2p + 1)
	if (x->submitted - x->completed >= 2) {
		g_last_error = "two batches are in flight already: arks_exchange_complete first";
		return ARKS_ERR_BAD_ARG;
	}
	ExSet& s = x->set[x->submitted & 1];
	hipStream_t st = static_cast<hipStream_t>(stream);
	DeviceGuard guard(x->device);
	// the buffers of the set's last batch (its map kernel may still be reading them) are reused in stream order: a
	// batch on another stream waits for the old one first
	if (s.used && s.st != st) {
		++x->stream_syncs;
		(void)hipStreamSynchronize(s.st);
	}
	s.st = st, s.used = true;
	s.codes = reinterpret_cast<const u64*>(d_codes), s.nmask = d_nmask, s.word_off = reinterpret_cast<const u64*>(d_word_off);
	s.lens = d_lens, s.eval = d_eval, s.n_reads = (long)n_reads, s.j_index = j_index, s.out = d_out_conreci, s.stats = d_stats;
	s.read_class = n_reads > 0 ? d_read_class : nullptr, s.pair_ok = d_pair_ok, s.eval_out = d_eval_out;
	s.rc = exchange_bucket(x, s);
	// (a batch that failed here is completed all the same: the other ranks learn of it there, nobody waits in vain)
	s.gathered = false;
	if (x->comm && x->world > 1) {
		// RCCL transport: the all-gather of (status, counts) rides on the batch's stream behind the bucket kernel -- from
		// the counters on the device, no host round trip -- and its result comes home with an event; complete() finds it
		// there.  (A batch that is bucketed again because a region was too small keeps its counts: the counters count on
		// when a block finds no room.)  A rank whose submit failed contributes its status: everybody decides alike.
		const arks_rccl_api* rccl = rccl_api();
		const int W = x->world;
		s.status_sent = s.rc != ARKS_OK;
		hipError_t he = arks::launch_gather_prep(s.d_ctl, s.base, (u64)(s.rc != ARKS_OK), W, s.d_gather, s.st);
		int e = 0;
		if (he == hipSuccess && rccl)
			e = rccl->AllGather(s.d_gather, s.d_gather + (1 + W), (size_t)(1 + W), kNcclUint64, x->comm, s.st);
		if (he == hipSuccess && e == 0)
			he = hipMemcpyAsync(s.h_gather + (1 + W), s.d_gather + (1 + W), sizeof(u64) * (size_t)W * (size_t)(1 + W), hipMemcpyDeviceToHost, s.st);
		if (he == hipSuccess && e == 0)
			he = hipEventRecord(s.counted, s.st); // (behind the ctl copy AND the gathered matrix: one wait in complete())
		if (!rccl || e != 0 || he != hipSuccess) {
			// nothing can be agreed any more: the communicator is of no use (the peers' calls fail or time out)
			const int rc2 = !rccl ? ARKS_ERR_HIP : (e != 0 ? fail_rccl(e, "ncclAllGather(seed counts)") : fail_hip(he, "seed counts"));
			if (rccl && rccl->CommAbort && x->comm) {
				(void)rccl->CommAbort(x->comm);
				x->comm = nullptr;
			}
			s.rc = s.rc != ARKS_OK ? s.rc : rc2;
		} else
			s.gathered = true;
	}
	s.pending = true;
	x->submitted++;
	return s.rc;
}

} // namespace

extern "C" {

int
arks_exchange_submit(
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
	return exchange_submit(x, d_codes, d_nmask, d_word_off, d_lens, d_eval, nullptr, nullptr, nullptr, n_reads, j_index, d_out_conreci,
	                       d_stats, stream);
}

int
arks_exchange_submit_pairs(
    arks_exchange* x,
    const uint64_t* d_codes,
    const uint32_t* d_nmask,
    const uint64_t* d_word_off,
    const uint32_t* d_lens,
    const uint8_t* d_read_class,
    const uint8_t* d_pair_ok,
    int64_t n_reads,
    double j_index,
    uint8_t* d_eval_out,
    int32_t* d_out_conreci,
    arks_map_stats* d_stats,
    void* stream)
{
	if (!d_read_class && n_reads > 0)
		return ARKS_ERR_BAD_ARG;
	return exchange_submit(x, d_codes, d_nmask, d_word_off, d_lens, nullptr, d_read_class, d_pair_ok, d_eval_out, n_reads, j_index,
	                       d_out_conreci, d_stats, stream);
}

} // extern "C"

namespace {

// errors are carried to the end of the collective part: a rank that fails must still meet the others
#define EX_TRY(expr)                                                                               \
	do {                                                                                           \
		if (rc == ARKS_OK) {                                                                       \
			hipError_t e_ = (expr);                                                                \
			if (e_ != hipSuccess)                                                                  \
				rc = fail_hip(e_, #expr);                                                          \
		}                                                                                          \
	} while (0)

// ---- 1. my counts (they left the device with the submit; a batch that did not fit its regions is run again) ----
int
ex_counts(arks_exchange* x, ExSet& s)
{
	const int W = x->world;
	int rc = s.rc;
	EX_TRY(hipEventSynchronize(s.counted));
	auto take = [&] { // the batch's counts: what the counters moved by
		for (int o = 0; o < 64; ++o)
			s.counts[o] = o < W ? s.h_ctl->fill[o * arks::kCtlStride] - s.base.fill[o] : 0;
		s.n_seeds = s.h_ctl->seeds[0] - s.base.seeds;
	};
	if (rc == ARKS_OK)
		take();
	for (int tries = 0; rc == ARKS_OK && s.h_ctl->overflow[0] == s.base.seq; ++tries) {
		// the regions are sized from what THIS batch needs (its absolute counts, + 10 %), and the sizes for later batches
		// follow the seeds per read of the batch that did not fit -- of a full-size batch only: a small batch of long
		// reads must not blow up the regions of every batch behind it
		u64 mx = 0;
		for (int o = 0; o < W; ++o)
			mx = s.counts[o] > mx ? s.counts[o] : mx;
		const double n = (double)(s.n_reads > 0 ? s.n_reads : 1);
		s.cap_floor = mx + mx / 10 + 8192, s.slot_floor = s.n_seeds + s.n_seeds / 20 + 8192;
		if (s.n_reads >= x->largest_batch / 2) {
			if ((double)mx / n * 1.1 > x->per_read_owner)
				x->per_read_owner = (double)mx / n * 1.1;
			if ((double)s.n_seeds / n * 1.05 > x->per_read_all)
				x->per_read_all = (double)s.n_seeds / n * 1.05;
		}
		x->reruns++;
		if (tries == 2) {
			g_last_error = "the exchange's regions overflowed three times in a row";
			rc = ARKS_ERR_HIP;
			break;
		}
		rc = exchange_bucket(x, s);
		EX_TRY(hipEventSynchronize(s.counted));
		if (rc == ARKS_OK)
			take();
	}
	s.cap_floor = s.slot_floor = 0;
	return rc;
}

// what I ask of whom and who asks what of me, from all[p * W + o] = seeds rank p asks of owner o
void
ex_tables(arks_exchange* x, ExSet& s)
{
	const int W = x->world, me = x->rank;
	s.sc.assign((size_t)W, 0), s.rcv.assign((size_t)W, 0), s.soff.assign((size_t)W, 0), s.roff.assign((size_t)W + 1, 0);
	u64 S = 0;
	for (int p = 0; p < W; ++p) {
		s.sc[(size_t)p] = s.all[(size_t)me * W + p];  // what I ask of p
		s.rcv[(size_t)p] = s.all[(size_t)p * W + me]; // what p asks of me
		s.soff[(size_t)p] = (u64)p * s.cap;           // ... lies in region p of my send buffer
		s.roff[(size_t)p + 1] = s.roff[(size_t)p] + s.rcv[(size_t)p];
		S += s.sc[(size_t)p];
	}
	x->last.seeds = s.n_seeds, x->last.sent = S - s.sc[(size_t)me], x->last.received = s.roff[(size_t)W] - s.rcv[(size_t)me];
	x->last.reruns = x->reruns;
	x->last.stream_syncs = x->stream_syncs;
}

// ---- 6. the home finishes: map_reads_s_kernel<REMOTE> + the general kernels -------------------------------------
int
ex_map(arks_exchange* x, ExSet& s)
{
	const arks_index* idx = x->idx;
	int rc = ARKS_OK;
	if (s.n_reads > 0) {
		// The queues were made at submit and their scratch block was zeroed by the bucket launch, in front of this one on
		// the stream -- unless somebody else has mapped on this (index, stream) since: the other batch in flight when the
		// caller gave both the same stream (order bucket A, bucket B, map A, map B: A dirties what B's bucket zeroed,
		// and a larger B has replaced A's queues), or a plain map call on the shard.  The set's generation tells: it
		// must stand exactly one past what the bucket step saw.  Otherwise: the fresh pointers, and the memset.
		arks_index::QueueSet now;
		rc = ensure_queue(idx, s.st, s.n_reads, &now);
		if (rc != ARKS_OK)
			return rc;
		const bool untouched = s.scratch_zeroed && now.gen == s.qs.gen + 1 && now.queue_count == s.qs.queue_count && now.queue == s.qs.queue;
		s.qs = now;
		s.scratch_zeroed = untouched;
		// (with the gate folded in, the map kernels read the eval the bucket kernel wrote)
		EX_TRY(launch_map_reads_seeded(
		    idx->kw, s.codes, s.nmask, s.word_off, s.lens, s.read_class ? s.eval_out : s.eval, s.n_reads, s.j_index, idx->geom, idx->bx,
		    idx->bxg, nullptr, s.ans_back.as<u64>(), s.out, reinterpret_cast<u64*>(s.stats), s.qs.queue, s.qs.queue_count, idx->n_cu,
		    s.st, s.slot.as<u32>(), s.chunk_off.as<u32>(), s.scratch_zeroed));
	}
	return rc;
}

// ---- the direct transport, stage by stage (between the stages every rank of the group must have done the one before:
//      a barrier when each rank has a thread of its own, a loop over the ranks when one thread drives them all) ------
void
direct_show(arks_exchange* x, int si, int rc) // after ex_counts
{
	ExSet& s = x->set[si];
	LocalGroup::Shown& mine = x->group->shown[si][(size_t)x->rank];
	mine.status = rc;
	mine.counts = s.counts;
	mine.cap = s.cap;
	mine.send = s.send.as<u64>();
	mine.ans_back = s.ans_back.as<u64>();
	mine.probed = s.probed;
}

// everybody's counts; I answer what everybody asks of me, reading the askers' send buffers (complete: every rank
// waited for its counts before it showed them) and writing into their answer buffers (free: their last map kernel on
// this set ran before their bucket kernel, in stream order)
int
direct_probe(arks_exchange* x, int si)
{
	ExSet& s = x->set[si];
	LocalGroup* g = x->group;
	const int W = x->world, me = x->rank;
	int rc = ARKS_OK;
	s.all.assign((size_t)W * (size_t)W, 0);
	for (int p = 0; p < W; ++p) {
		const LocalGroup::Shown& sh = g->shown[si][(size_t)p];
		if (sh.status != ARKS_OK) {
			// every rank reads the same flags: they all leave here
			if (p != me || g_last_error.empty())
				g_last_error = "another rank of the local group failed in this batch";
			return x->group->shown[si][(size_t)me].status != ARKS_OK ? x->group->shown[si][(size_t)me].status : ARKS_ERR_HIP;
		}
		for (int o = 0; o < W; ++o)
			s.all[(size_t)p * W + o] = sh.counts[o];
	}
	ex_tables(x, s);
	DeviceGuard guard(x->device);
	arks::ProbeSegs sg{};
	u64 run = 0;
	for (int p = 0; p < W; ++p) {
		if (!s.rcv[(size_t)p])
			continue;
		const LocalGroup::Shown& sh = g->shown[si][(size_t)p];
		const int i = sg.n_segs++;
		sg.src[i] = sh.send + (u64)me * sh.cap;
		sg.dst[i] = sh.ans_back + 2 * (u64)me * sh.cap;
		run += s.rcv[(size_t)p];
		sg.end[i] = run;
	}
	EX_TRY(arks::launch_seeds_probe_segs(x->idx->bx.m, x->idx->bx, sg, s.st, x->idx->n_cu));
	EX_TRY(hipEventRecord(s.probed, s.st));
	g->shown[si][(size_t)me].status2 = rc;
	return rc;
}

// every owner's event is recorded (a wait for an event that was never recorded is no wait): the answers are home when
// the stream has passed them; then the map kernels.  (The askers' buffers are read and written by other ranks' kernels
// until those events: a rank reuses a set only behind its own map kernel of that set, which waits for all of them.)
int
direct_finish(arks_exchange* x, int si, int rc)
{
	ExSet& s = x->set[si];
	LocalGroup* g = x->group;
	const int W = x->world, me = x->rank;
	DeviceGuard guard(x->device);
	for (int p = 0; p < W; ++p) {
		const LocalGroup::Shown& sh = g->shown[si][(size_t)p];
		if (sh.status2 != ARKS_OK && rc == ARKS_OK) {
			g_last_error = "another rank of the local group failed in this batch";
			rc = ARKS_ERR_HIP;
		}
		if (p != me && rc == ARKS_OK && s.sc[(size_t)p])
			EX_TRY(hipStreamWaitEvent(s.st, sh.probed, 0)); // p's answers to me
	}
	if (rc == ARKS_OK)
		rc = ex_map(x, s);
	return rc;
}

// the oldest submitted batch of x, taken out of the queue
int
ex_take(arks_exchange* x, int* si)
{
	if (!x)
		return ARKS_ERR_BAD_ARG;
	if (x->submitted == x->completed) {
		g_last_error = "arks_exchange_complete: nothing was submitted";
		return ARKS_ERR_BAD_ARG;
	}
	*si = (int)(x->completed & 1);
	x->completed++;
	x->set[*si].pending = false;
	return ARKS_OK;
}

} // namespace

extern "C" {

int
arks_exchange_complete(arks_exchange* x)
{
	int si = 0;
	{
		const int trc = ex_take(x, &si);
		if (trc != ARKS_OK)
			return trc;
	}
	ExSet& s = x->set[si];
	const arks_index* idx = x->idx;
	const int W = x->world, me = x->rank;
	hipStream_t st = s.st;
	DeviceGuard guard(x->device);
	const arks_rccl_api* rccl = x->comm ? rccl_api() : nullptr;
	LocalGroup* g = x->group;
	int rc = ex_counts(x, s);
	if (g) {
		// ---- direct transport, a thread per rank: the stages with barriers between them ------------------------------
		direct_show(x, si, rc);
		if (!g->barrier())
			return exchange_broken();
		rc = direct_probe(x, si);
		if (!g->barrier()) // (also keeps a fast rank's next batch from rewriting what a slow one still reads)
			return exchange_broken();
		bool agreed_failure = false;
		for (int p = 0; p < W; ++p)
			agreed_failure = agreed_failure || g->shown[si][(size_t)p].status != ARKS_OK;
		if (!agreed_failure)
			rc = direct_finish(x, si, rc);
		if (rc != ARKS_OK)
			g->abort(); // this rank's caller will not call again: the others must not wait for it
		return rc;
	}
	// ---- 2. everybody's counts and status: all[p * W + o] = seeds rank p asks of owner o -----------------------------
	s.all.assign((size_t)W * (size_t)W, 0);
	if (rccl && W > 1) {
		// (1 + W) numbers per rank: status, counts -- gathered on the batch's stream since the submit (round 5: complete()
		// used to enqueue the all-gather itself and wait for the stream, a host stall per batch); every rank holds the
		// same matrix and decides alike.  ex_counts has waited for the event behind the matrix.
		u64* hg = s.h_gather;
		if (!s.gathered) {
			if (rc == ARKS_OK) {
				g_last_error = "the counts of this batch were never gathered (its submit failed)";
				rc = ARKS_ERR_HIP;
			}
			return rc;
		}
		if (rc != ARKS_OK) {
			// a failure the all-gather carried (the submit's): every rank reads it and leaves, nobody waits.  One that came
			// after it -- a region that overflowed three times, a HIP error: the peers believe this rank fine and are on
			// their way into the exchange -- aborts the communicator, so that they get an error instead of a hang.
			if (!s.status_sent && rccl->CommAbort && x->comm) {
				(void)rccl->CommAbort(x->comm);
				x->comm = nullptr;
			}
			return rc;
		}
		int status = rc;
		for (int p = 0; p < W; ++p) {
			const u64* row = hg + (size_t)(1 + W) * (size_t)(1 + p);
			if (row[0] && status == ARKS_OK) {
				g_last_error = "another rank of the exchange failed in this batch";
				status = ARKS_ERR_HIP;
			}
			for (int o = 0; o < W; ++o)
				s.all[(size_t)p * W + o] = row[1 + o];
		}
		if (status != ARKS_OK)
			return status; // (every rank sees the same flags: all of them leave here, nobody waits)
	} else {
		if (rc != ARKS_OK)
			return rc;
		s.all[0] = s.counts[0];
	}
	// from here on rc == ARKS_OK on every rank
	ex_tables(x, s);
	const std::vector<u64>&sc = s.sc, &rcv = s.rcv, &soff = s.soff, &roff = s.roff;
	arks::ProbeSegs sg{};
	if (rccl && W > 1) {
		// ---- 3. seeds to their owners; 4. the owner's answers; 5. answers back -------------------------------------
		const u64 R = roff[(size_t)W];
		hipError_t he = s.recv.reserve(sizeof(u64) * (size_t)(R + 1), st, &x->stream_syncs);
		if (he == hipSuccess)
			he = s.ans_out.reserve(2 * sizeof(u64) * (size_t)(R + 1), st, &x->stream_syncs);
		if (he != hipSuccess) {
			// the others are on their way into the exchange: they must not wait for this rank
			rc = fail_hip(he, "hipMalloc(exchange receive buffers)");
			if (rccl->CommAbort && x->comm) {
				(void)rccl->CommAbort(x->comm);
				x->comm = nullptr;
			}
			return rc;
		}
		rc = rccl_all_to_all(x, rccl, s.send.as<u64>(), soff.data(), sc.data(), s.recv.as<u64>(), roff.data(), rcv.data(), 1, st,
		                     "seeds to their owners (ncclSend / ncclRecv)");
		if (rc != ARKS_OK)
			return rc;
		// what the ranks in front of me and behind me asked (from the receive buffer), and my own seeds from where they
		// are to where their answers belong
		u64 run = 0;
		for (int p = 0; p < W; ++p) {
			if (!rcv[(size_t)p])
				continue;
			const int i = sg.n_segs++;
			sg.src[i] = p == me ? s.send.as<u64>() + soff[(size_t)me] : s.recv.as<u64>() + roff[(size_t)p];
			sg.dst[i] = p == me ? s.ans_back.as<u64>() + 2 * soff[(size_t)me] : s.ans_out.as<u64>() + 2 * roff[(size_t)p];
			run += rcv[(size_t)p];
			sg.end[i] = run;
		}
		EX_TRY(arks::launch_seeds_probe_segs(idx->bx.m, idx->bx, sg, st, idx->n_cu));
		if (rc == ARKS_OK)
			rc = rccl_all_to_all(x, rccl, s.ans_out.as<u64>(), roff.data(), rcv.data(), s.ans_back.as<u64>(), soff.data(), sc.data(),
			                     2, st, "answers back (ncclSend / ncclRecv)");
		else if (rccl->CommAbort && x->comm) {
			(void)rccl->CommAbort(x->comm);
			x->comm = nullptr;
		}
		if (rc != ARKS_OK)
			return rc;
	} else {
		sg.n_segs = sc[0] ? 1 : 0;
		sg.src[0] = s.send.as<u64>();
		sg.dst[0] = s.ans_back.as<u64>();
		sg.end[0] = sc[0];
		EX_TRY(arks::launch_seeds_probe_segs(idx->bx.m, idx->bx, sg, st, idx->n_cu));
	}
	if (rc == ARKS_OK)
		rc = ex_map(x, s);
	return rc;
}

int
arks_exchange_complete_group(arks_exchange* const* xs, int world)
{
	if (!xs || world < 1 || world > 64 || !xs[0] || !xs[0]->group || xs[0]->group->world != world)
		return ARKS_ERR_BAD_ARG;
	LocalGroup* g = xs[0]->group;
	for (int r = 0; r < world; ++r)
		if (!xs[r] || xs[r]->group != g || xs[r]->rank != r || xs[r]->submitted == xs[r]->completed) {
			g_last_error = "arks_exchange_complete_group: the ranks of one local group, each with a submitted batch";
			return ARKS_ERR_BAD_ARG;
		}
	{
		std::lock_guard<std::mutex> lk(g->m);
		if (g->aborted)
			return exchange_broken();
	}
	std::vector<int> si((size_t)world, 0), rcs((size_t)world, ARKS_OK);
	int rc = ARKS_OK;
	for (int r = 0; r < world; ++r) {
		(void)ex_take(xs[r], &si[(size_t)r]);
		DeviceGuard guard(xs[r]->device);
		direct_show(xs[r], si[(size_t)r], ex_counts(xs[r], xs[r]->set[si[(size_t)r]]));
	}
	for (int r = 0; r < world; ++r) {
		rcs[(size_t)r] = direct_probe(xs[r], si[(size_t)r]);
		rc = rc == ARKS_OK ? rcs[(size_t)r] : rc;
	}
	for (int r = 0; r < world && rc == ARKS_OK; ++r) {
		rcs[(size_t)r] = direct_finish(xs[r], si[(size_t)r], rcs[(size_t)r]);
		rc = rc == ARKS_OK ? rcs[(size_t)r] : rc;
	}
	if (rc != ARKS_OK)
		g->abort();
	return rc;
}
#undef EX_TRY

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
	if (x && x->submitted != x->completed) {
		g_last_error = "a submitted batch is waiting for arks_exchange_complete";
		return ARKS_ERR_BAD_ARG;
	}
	const int rc = arks_exchange_submit(x, d_codes, d_nmask, d_word_off, d_lens, d_eval, n_reads, j_index, d_out_conreci, d_stats, stream);
	if (rc == ARKS_ERR_BAD_ARG && (!x || x->submitted == x->completed))
		return rc; // nothing was submitted (bad arguments): nothing to complete
	return arks_exchange_complete(x);
}

} // extern "C"
