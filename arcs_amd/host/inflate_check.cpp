// inflate_check.cpp -- test driver for fast_inflate.hpp: inflates a .gz file with GzInflater and with
// zlib's gzread, compares the two streams byte for byte and reports both rates.
//   inflate_check <file.gz> [chunk]      exit 0 = identical (or both fail), prints "same <bytes> ..."
#include "fast_inflate.hpp"

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
	const bool same = a == b && (rc_a < 0) == (rc_b < 0);
	std::printf("%s %zu bytes (zlib %zu) rc %d/%d  fast %.0f MB/s  zlib %.0f MB/s\n", same ? "same" : "DIFFERENT", a.size(), b.size(),
	            rc_a, rc_b, rate_a, rate_b);
	(void)sa, (void)sb;
	return same ? 0 : 1;
}
