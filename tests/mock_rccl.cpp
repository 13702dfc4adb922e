// tests/mock_rccl.cpp -- TEST INFRASTRUCTURE: stand-ins for the RCCL entry points arks_exchange reaches
// (include/arks_hip_debug.h: arks_rccl_api), so that the library's world > 1 code -- the ncclAllGather of the counts,
// the ncclSend / ncclRecv groups with their offset tables, the error paths -- runs on a box with ONE GPU, where RCCL
// itself refuses two ranks on a device.  The ranks are threads of one process; a "communicator" is a seat in a
// World; a group is executed at ncclGroupEnd: every rank waits for its stream, posts its sends, meets the others at a
// barrier, copies what it receives device to device (matching sends and receives of a pair of ranks in order, sizes
// checked), waits for its stream and meets the others again.  The semantics the exchange relies on are RCCL's: all
// ranks issue the same sequence of collectives; sends and receives of one group pair up by (rank, order).
//
// Fault injection (mock_rccl_fail): the n-th ncclSend / ncclRecv / ncclAllGather of the process returns an error --
// the group it sits in fails as a whole at ncclGroupEnd, as RCCL's does, and ncclCommAbort releases the peers.
// Counters (mock_rccl_counters) let a test assert that every opened group was closed.
//
// Built by tests/test_gpu_exchange.py with g++ against the HIP runtime; never part of the product library.
#include "arks_hip_debug.h"

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Op
{
	const void* buf;
	size_t bytes;
	int peer;
	bool send;
	void* comm;
	hipStream_t st;
};

struct World
{
	int world = 0;
	std::mutex m;
	std::condition_variable cv;
	int arrived = 0;
	unsigned long long gen = 0;
	bool aborted = false;
	int seats = 0, alive = 0;
	std::vector<std::vector<Op>> posted; // [rank]: the sends of the group being executed
	std::vector<const void*> ag;         // [rank]: all-gather sources

	bool
	barrier()
	{
		std::unique_lock<std::mutex> lk(m);
		if (aborted)
			return false;
		const unsigned long long g = gen;
		if (++arrived == world) {
			arrived = 0;
			++gen;
			cv.notify_all();
			return true;
		}
		if (!cv.wait_for(lk, std::chrono::seconds(120), [&] { return gen != g || aborted; }))
			aborted = true;
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

struct Comm
{
	World* w;
	int rank;
};

std::mutex g_m;
std::map<std::string, World*> g_worlds;
unsigned long long g_next_id = 1;

thread_local int t_depth = 0;
thread_local bool t_group_failed = false;
thread_local std::vector<Op> t_ops;

std::atomic<long> g_calls[3]; // send, recv, allgather
std::atomic<long> g_fail_at[3] = { { -1 }, { -1 }, { -1 } };
std::atomic<long> g_groups_opened{ 0 }, g_groups_closed{ 0 }, g_aborts{ 0 }, g_bytes_moved{ 0 };

size_t
dtype_size(int dtype)
{
	switch (dtype) {
	case 0: case 1: return 1; // ncclInt8, ncclUint8
	case 2: case 3: return 4; // ncclInt32, ncclUint32
	case 4: case 5: return 8; // ncclInt64, ncclUint64
	default: return 0;
	}
}

bool
inject(int kind)
{
	const long n = g_calls[kind].fetch_add(1);
	return g_fail_at[kind].load() == n;
}

int
execute(std::vector<Op>& ops)
{
	if (ops.empty())
		return 0;
	Comm* c = static_cast<Comm*>(ops[0].comm);
	World* w = c->w;
	const int me = c->rank;
	// what the stream says the buffers hold when it gets here
	std::vector<hipStream_t> streams;
	for (const Op& o : ops) {
		if (o.comm != ops[0].comm)
			return 5; // (one communicator per group is all the exchange uses)
		bool seen = false;
		for (hipStream_t s : streams)
			seen = seen || s == o.st;
		if (!seen)
			streams.push_back(o.st);
	}
	for (hipStream_t s : streams)
		if (hipStreamSynchronize(s) != hipSuccess)
			return 1;
	w->posted[(size_t)me].clear();
	for (const Op& o : ops)
		if (o.send)
			w->posted[(size_t)me].push_back(o);
	if (!w->barrier())
		return 6;
	int rc = 0;
	std::vector<size_t> cursor((size_t)w->world, 0);
	for (const Op& o : ops) {
		if (o.send)
			continue;
		const std::vector<Op>& theirs = w->posted[(size_t)o.peer];
		size_t& cur = cursor[(size_t)o.peer];
		while (cur < theirs.size() && theirs[cur].peer != me)
			++cur;
		if (cur == theirs.size() || theirs[cur].bytes != o.bytes) {
			rc = 5; // no matching send, or another size: invalid usage
			break;
		}
		if (hipMemcpyAsync(const_cast<void*>(o.buf), theirs[cur].buf, o.bytes, hipMemcpyDeviceToDevice, o.st) != hipSuccess) {
			rc = 1;
			break;
		}
		g_bytes_moved += (long)o.bytes;
		++cur;
	}
	for (hipStream_t s : streams)
		if (hipStreamSynchronize(s) != hipSuccess)
			rc = rc ? rc : 1;
	if (!w->barrier())
		return rc ? rc : 6;
	return rc;
}

int
m_version(int* v)
{
	*v = 22200; // "2.22.0"
	return 0;
}

int
m_unique_id(arks_rccl_unique_id* id)
{
	std::lock_guard<std::mutex> lk(g_m);
	std::memset(id->internal, 0, sizeof id->internal);
	std::snprintf(id->internal, sizeof id->internal, "mock-rccl-world-%llu", g_next_id++);
	return 0;
}

int
m_init(void** comm, int world, arks_rccl_unique_id id, int rank)
{
	std::lock_guard<std::mutex> lk(g_m);
	const std::string key(id.internal, strnlen(id.internal, sizeof id.internal));
	World*& w = g_worlds[key];
	if (!w) {
		w = new World();
		w->world = world;
		w->posted.resize((size_t)world);
		w->ag.assign((size_t)world, nullptr);
	}
	if (w->world != world || rank < 0 || rank >= world)
		return 4; // ncclInvalidArgument
	w->seats++, w->alive++;
	*comm = new Comm{ w, rank };
	return 0;
}

void
leave(Comm* c)
{
	std::lock_guard<std::mutex> lk(g_m);
	World* w = c->w;
	delete c;
	if (--w->alive == 0 && w->seats == w->world) {
		for (auto it = g_worlds.begin(); it != g_worlds.end(); ++it)
			if (it->second == w) {
				g_worlds.erase(it);
				break;
			}
		delete w;
	}
}

int
m_destroy(void* comm)
{
	leave(static_cast<Comm*>(comm));
	return 0;
}

int
m_abort(void* comm)
{
	g_aborts++;
	static_cast<Comm*>(comm)->w->abort();
	leave(static_cast<Comm*>(comm));
	return 0;
}

int
m_group_start()
{
	if (t_depth++ == 0) {
		g_groups_opened++;
		t_group_failed = false;
		t_ops.clear();
	}
	return 0;
}

int
m_group_end()
{
	if (t_depth == 0)
		return 5;
	if (--t_depth > 0)
		return 0;
	g_groups_closed++;
	std::vector<Op> ops;
	ops.swap(t_ops);
	if (t_group_failed)
		return 1; // a call of the group failed: nothing of it is launched
	return execute(ops);
}

int
m_p2p(const void* buf, size_t count, int dtype, int peer, void* comm, void* stream, bool send)
{
	const size_t sz = dtype_size(dtype);
	Comm* c = static_cast<Comm*>(comm);
	if (inject(send ? 0 : 1) || !sz || !c || peer < 0 || peer >= c->w->world) {
		if (t_depth)
			t_group_failed = true;
		return 1; // ncclUnhandledCudaError
	}
	Op o{ buf, count * sz, peer, send, comm, static_cast<hipStream_t>(stream) };
	if (t_depth) {
		t_ops.push_back(o);
		return 0;
	}
	std::vector<Op> one{ o };
	return execute(one);
}

int
m_send(const void* buf, size_t count, int dtype, int peer, void* comm, void* stream)
{
	return m_p2p(buf, count, dtype, peer, comm, stream, true);
}

int
m_recv(void* buf, size_t count, int dtype, int peer, void* comm, void* stream)
{
	return m_p2p(buf, count, dtype, peer, comm, stream, false);
}

int
m_allgather(const void* send, void* recv, size_t count, int dtype, void* comm, void* stream)
{
	const size_t bytes = count * dtype_size(dtype);
	Comm* c = static_cast<Comm*>(comm);
	if (inject(2) || !bytes || !c)
		return 1;
	World* w = c->w;
	hipStream_t st = static_cast<hipStream_t>(stream);
	if (hipStreamSynchronize(st) != hipSuccess)
		return 1;
	w->ag[(size_t)c->rank] = send;
	if (!w->barrier())
		return 6;
	int rc = 0;
	for (int p = 0; p < w->world && !rc; ++p)
		if (hipMemcpyAsync(static_cast<char*>(recv) + (size_t)p * bytes, w->ag[(size_t)p], bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
			rc = 1;
	if (hipStreamSynchronize(st) != hipSuccess)
		rc = rc ? rc : 1;
	if (!w->barrier())
		return rc ? rc : 6;
	return rc;
}

const char*
m_error_string(int e)
{
	switch (e) {
	case 0: return "no error";
	case 1: return "mock: unhandled error (injected, or a HIP call failed)";
	case 4: return "mock: invalid argument";
	case 5: return "mock: invalid usage (sends and receives of a group do not pair up)";
	case 6: return "mock: remote error (a rank aborted or never came)";
	default: return "mock: unknown error";
	}
}

const arks_rccl_api g_table = { m_version, m_unique_id, m_init, m_destroy, m_abort, m_group_start, m_group_end,
	                            m_send,    m_recv,      m_allgather, m_error_string };

} // namespace

extern "C" {

const arks_rccl_api*
mock_rccl_table()
{
	return &g_table;
}

// kind 0 = ncclSend, 1 = ncclRecv, 2 = ncclAllGather: the call number `nth` from now on fails (-1: none)
void
mock_rccl_fail(int kind, long nth)
{
	if (kind < 0 || kind > 2)
		return;
	g_fail_at[kind].store(nth < 0 ? -1 : g_calls[kind].load() + nth);
}

// out[0] sends, [1] receives, [2] all-gathers, [3] groups opened, [4] groups closed, [5] aborts, [6] bytes received
void
mock_rccl_counters(long* out)
{
	out[0] = g_calls[0].load(), out[1] = g_calls[1].load(), out[2] = g_calls[2].load();
	out[3] = g_groups_opened.load(), out[4] = g_groups_closed.load(), out[5] = g_aborts.load();
	out[6] = g_bytes_moved.load();
}

} // extern "C"
