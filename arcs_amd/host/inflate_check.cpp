// inflate_check.cpp -- test driver for fast_inflate.hpp: inflates a .gz file with GzInflater and with
// zlib's gzread, compares the two streams byte for byte and reports both rates.
//   inflate_check <file.gz> [chunk]      exit 0 = identical (or both fail), prints "same <bytes> ..."
//   inflate_check <file.gz> <chunk> pgz <threads> <compressed bytes per chunk> [chunks per stretch]
//       also the stream as pgzip.hpp delivers it (blocks decoded on several threads, then GzInflater::resume):
//       must equal GzInflater's, the same bytes and the same kind of end; prints "pgz same <bytes> parallel <bytes>"
#include "fast_inflate.hpp"
#include "pgzip.hpp"

#include <thread>

#include <chrono>
#include <cstdlib>
#include <string>

int
main(int argc, char** argv)
{
	if (argc < 2)
		return 2;
	const int chunk = argc > 2 ? std::atoi(argv[2]) : (1 << 18);
	std::vector<unsigned char> a, b, buf((size_t)chunk);
	auto now = [] { return std::chrono::steady_clock::now(); };
	int rc_a = 0, rc_b = 0;
	double rate_a = 0, rate_b = 0;
	{ // timing passes: inflate and discard
		auto t = now();
		size_t total = 0;
		{
			FILE* f = std::fopen(argv[1], "rb");
			if (!f)
				return 2;
			arks_host::GzInflater g(f);
			int n;
			while ((n = g.read(buf.data(), chunk)) > 0)
				total += (size_t)n;
		}
		rate_a = total / 1e6 / std::chrono::duration<double>(now() - t).count();
		t = now();
		total = 0;
		gzFile g = gzopen(argv[1], "r");
		gzbuffer(g, 1u << 20);
		int n;
		while ((n = gzread(g, buf.data(), (unsigned)chunk)) > 0)
			total += (size_t)n;
		gzclose(g);
		rate_b = total / 1e6 / std::chrono::duration<double>(now() - t).count();
	}
	auto t0 = now();
	{
		FILE* f = std::fopen(argv[1], "rb");
		if (!f)
			return 2;
		arks_host::GzInflater g(f);
		int n;
		while ((n = g.read(buf.data(), chunk)) > 0)
			a.insert(a.end(), buf.begin(), buf.begin() + n);
		rc_a = n;
	}
	auto t1 = now();
	{
		gzFile g = gzopen(argv[1], "r");
		gzbuffer(g, 1u << 20);
		int n;
		while ((n = gzread(g, buf.data(), (unsigned)chunk)) > 0)
			b.insert(b.end(), buf.begin(), buf.begin() + n);
		rc_b = n;
		gzclose(g);
	}
	auto t2 = now();
	const double sa = std::chrono::duration<double>(t1 - t0).count(), sb = std::chrono::duration<double>(t2 - t1).count();
	bool pgz_same = true;
	char pgz_msg[256] = "";
	if (argc > 5 && std::string(argv[3]) == "pgz") {
		const unsigned threads = (unsigned)std::atoi(argv[4]);
		const size_t pchunk = (size_t)std::atoll(argv[5]);
		const unsigned per = argc > 6 ? (unsigned)std::atoi(argv[6]) : 2 * threads;
		const arks_host::ParallelFor pf = [&](size_t n, const std::function<void(size_t)>& fn) {
			std::atomic<size_t> next{ 0 };
			auto work = [&] {
				for (size_t i; (i = next.fetch_add(1)) < n;)
					fn(i);
			};
			std::vector<std::thread> th;
			for (unsigned t = 1; t < threads && t < n; ++t)
				th.emplace_back(work);
			work();
			for (auto& t : th)
				t.join();
		};
		std::vector<unsigned char> c;
		int rc_c = 0;
		size_t parallel = 0;
		const auto t3 = now();
		{
			FILE* f = std::fopen(argv[1], "rb");
			if (!f)
				return 2;
			arks_host::GzInflater g(f);
			arks_host::GzStretches z(g.file(), pchunk, per);
			while (z.ok() && !z.at_end()) {
				const size_t o = c.size();
				const size_t got = z.next(pf, [&](size_t n) { // (room for what the stretch may hold; `got` is what it does)
					c.resize(o + n);
					return c.data() + o;
				});
				c.resize(o + got);
				parallel += got;
			}
			bool ok = true;
			if (z.ok() && z.started()) {
				const arks_host::GzResumePoint& r = z.resume();
				ok = r.member_start ? g.resume_member(r.bit >> 3) : g.resume(r.bit, r.window.data(), r.window.size(), r.crc, r.member_out);
			}
			int n = -1;
			while (ok && (n = g.read(buf.data(), chunk)) > 0)
				c.insert(c.end(), buf.begin(), buf.begin() + n);
			rc_c = n;
		}
		const double sc = std::chrono::duration<double>(now() - t3).count();
		pgz_same = c == a && (rc_c < 0) == (rc_a < 0);
		std::snprintf(pgz_msg, sizeof pgz_msg, "pgz %s %zu bytes parallel %zu rc %d  %.0f MB/s with %u threads\n", pgz_same ? "same" : "DIFFERENT",
		              c.size(), parallel, rc_c, c.size() / 1e6 / sc, threads);
	}
	const bool same = a == b && (rc_a < 0) == (rc_b < 0) && pgz_same;
	std::printf("%s %zu bytes (zlib %zu) rc %d/%d  fast %.0f MB/s  zlib %.0f MB/s\n", same ? "same" : "DIFFERENT", a.size(), b.size(),
	            rc_a, rc_b, rate_a, rate_b);
	std::fputs(pgz_msg, stdout);
	(void)sa, (void)sb;
	return same ? 0 : 1;
}
