// junction_check.cpp - the host junction stage (seeksv_amd/host/junction_stage.cpp) on its own, for tests/test_junction_stage.py:
//   junction_check join <clip.gz> <clip.bam> [search_length]   every junction of the join (and after the merge when search_length is given), one line each
//   junction_check slurp <file.gz>                             the decompressed bytes to stdout
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "../../seeksv_amd/host/junction_stage.h"

using namespace seeksv;

static std::string cig(const CigarVec &v)
{
	std::string s;
	for (auto &p : v) s += std::to_string(p.first) + p.second;
	return s.empty() ? "*" : s;
}

int main(int argc, char **argv)
{
	if (argc >= 3 && !strcmp(argv[1], "slurp")) {
		std::string text, err = slurp_gz(argv[2], text);
		if (!err.empty()) { std::cerr << err << std::endl; return 1; }
		fwrite(text.data(), 1, text.size(), stdout);
		return 0;
	}
	if (argc < 4 || strcmp(argv[1], "join")) { std::cerr << "usage: junction_check join clip.gz clip.bam [search_length] | slurp file.gz" << std::endl; return 2; }
	JunctionMap j2o;
	const std::string err = assemble_junctions(argv[2], argv[3], j2o);
	if (!err.empty()) { std::cerr << err << std::endl; return 1; }
	if (argc > 4) merge_junctions(j2o, atoi(argv[4]));
	for (auto &kv : j2o) {
		const Junction &j = kv.first;
		const OtherInfo &o = kv.second;
		std::cout << j.up_chr << '\t' << j.up_pos << '\t' << j.up_strand << '\t' << j.down_chr << '\t' << j.down_pos << '\t' << j.down_strand << '\t' << o.microhomology << '\t' << o.abnormal;
		for (const SeqInfo *s : {&o.up, &o.down})
			std::cout << '\t' << s->seq << '\t' << cig(s->cigar_vec) << '\t' << s->left_clipped << '\t' << s->right_clipped << '\t' << s->support << '\t' << s->uniq;
		std::cout << '\n';
	}
	return 0;
}
