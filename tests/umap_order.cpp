// umap_order.cpp -- test helper: the iteration order of std::unordered_map<std::string, int> after inserting
// the lines of stdin in order, with the libstdc++ this machine's product binaries are built against.  The
// reference writes the vertices of <base>.dist.gv in the iteration order of its ContigToLength map
// (Arcs/Arcs.cpp:1622, Arcs/Arcs.h:115), filled in FASTA order: tests/graph_ref.py takes that order from here,
// so that its .dist.gv can be compared with the product's as text.
#include <iostream>
#include <string>
#include <unordered_map>

int
main()
{
	std::unordered_map<std::string, int> m;
	std::string line;
	int i = 0;
	while (std::getline(std::cin, line))
		m[line] = i++;
	for (const auto& kv : m)
		std::cout << kv.first << "\n";
	return 0;
}
