// unmapped_pairs_check.cpp - seeksv_amd/host/unmapped_pairs.h against a direct restatement of StoreUnmapSeqAndQual (clip_reads.h:172-219) +
// GetSeqAndQual (clip_reads.cpp:375-388): random raw BAM records with names drawn from a small pool (mates next to each other, batches apart, the
// same end twice, three and four records of one name, names that never pair), odd / even / zero lengths, missing qualities, every batch cut.
#include <cstdio>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "unmapped_pairs.h"

static void put_record(std::vector<uint8_t> &out, const std::string &name, bool read1, int l_seq, int n_cigar, bool no_qual, std::mt19937 &rng)
{
	const size_t body = 32 + name.size() + 1 + 4 * (size_t)n_cigar + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
	const size_t at = out.size();
	out.resize(at + 4 + body, 0);
	uint8_t *r = out.data() + at + 4;
	const uint32_t bs = (uint32_t)body;
	memcpy(out.data() + at, &bs, 4);
	r[8] = (uint8_t)(name.size() + 1);
	const uint16_t nc = (uint16_t)n_cigar, flag = (uint16_t)(1 | (rng() & 1 ? 4 : 8) | (read1 ? 64 : 128));
	memcpy(r + 12, &nc, 2); memcpy(r + 14, &flag, 2);
	const int32_t ls = l_seq;
	memcpy(r + 16, &ls, 4);
	memcpy(r + 32, name.c_str(), name.size() + 1);
	uint8_t *s = r + 32 + name.size() + 1 + 4 * (size_t)n_cigar, *q = s + ((size_t)l_seq + 1) / 2;
	for (int k = 0; k < (l_seq + 1) / 2; ++k) s[k] = (uint8_t)rng();
	for (int k = 0; k < l_seq; ++k) q[k] = no_qual ? 0xff : (uint8_t)(rng() % 42);
}

struct Plain { std::string name, seq, qual; bool read1; };
static Plain decode(const uint8_t *raw, size_t off)
{
	static const char NT16[] = "=ACMGRSVTWYHKDBN";
	const uint8_t *r = raw + off + 4;
	uint16_t nc, flag; int32_t l; memcpy(&nc, r + 12, 2); memcpy(&flag, r + 14, 2); memcpy(&l, r + 16, 4);
	Plain p;
	p.name = (const char *)(r + 32);
	const uint8_t *s = r + 32 + r[8] + 4 * (size_t)nc, *q = s + ((size_t)l + 1) / 2;
	if (l) {
		for (int i = 0; i < l; ++i) p.seq += NT16[(s[i >> 1] >> ((~i & 1) << 2)) & 15];
		if (q[0] == 0xff) p.qual = "*"; else for (int i = 0; i < l; ++i) p.qual += (char)(q[i] + 33);
	}
	p.read1 = flag & 64;
	return p;
}

int main()
{
	int bad = 0, cases = 0;
	for (int seed = 0; seed < 60; ++seed) {
		std::mt19937 rng((unsigned)seed * 7919u + 1);
		const int n = 200 + (int)(rng() % 3000), pool = 20 + (int)(rng() % (unsigned)n);
		std::vector<uint8_t> raw;
		std::vector<size_t> off;
		for (int i = 0; i < n; ++i) {
			std::string name = "q" + std::to_string(rng() % (unsigned)pool);
			if (rng() % 50 == 0) name += std::string(rng() % 200, 'x'); // long names
			const int l = rng() % 20 == 0 ? 0 : (int)(rng() % 160);
			off.push_back(raw.size());
			put_record(raw, name, rng() & 1, l, (int)(rng() % 3), rng() % 9 == 0, rng);
			if (rng() % 3 == 0) { off.push_back(raw.size()); put_record(raw, name, rng() % 8 != 0 ? !(decode(raw.data(), off[off.size() - 2]).read1) : rng() & 1, (int)(rng() % 160), 1, false, rng); ++i; } // the mate right behind
		}
		// the reference's way
		std::string exp1, exp2;
		{
			std::map<std::string, std::pair<std::pair<std::string, std::string>, char>> m;
			for (size_t o : off) {
				const Plain p = decode(raw.data(), o);
				auto it = m.find(p.name);
				if (it != m.end()) {
					if (p.read1 && it->second.second == '2') {
						exp1 += "@" + it->first + "/1\n" + p.seq + "\n+\n" + p.qual + "\n";
						exp2 += "@" + it->first + "/2\n" + it->second.first.first + "\n+\n" + it->second.first.second + "\n";
						m.erase(it);
					} else if (!p.read1 && it->second.second == '1') {
						exp1 += "@" + it->first + "/1\n" + it->second.first.first + "\n+\n" + it->second.first.second + "\n";
						exp2 += "@" + it->first + "/2\n" + p.seq + "\n+\n" + p.qual + "\n";
						m.erase(it);
					}
				} else m.insert(std::make_pair(p.name, std::make_pair(std::make_pair(p.seq, p.qual), p.read1 ? '1' : '2')));
			}
		}
		// the pairer, under several batch cuts; a batch's bytes are scribbled over once the pairer has let go of them
		for (int cut = 0; cut < 4; ++cut) {
			std::string got1, got2;
			{
				seeksv::UnmappedPairs up([&](std::vector<std::string> &p) { for (auto &s : p) got1 += s; }, [&](std::vector<std::string> &p) { for (auto &s : p) got2 += s; }, 1 + cut);
				std::vector<std::vector<uint8_t>> live(2);
				size_t i = 0;
				int k = 0;
				while (i < off.size()) {
					size_t j = cut == 0 ? off.size() : std::min(off.size(), i + 1 + (cut == 1 ? 0 : rng() % (cut == 2 ? 7 : 900)));
					const size_t b = off[i], e = j < off.size() ? off[j] : raw.size();
					std::vector<uint8_t> &buf = live[(size_t)(k & 1)];
					// (submit(k) returns when batch k - 1 is done with: batch k - 2's buffer - this one - may be reused now)
					buf.assign(raw.begin() + (ptrdiff_t)b, raw.begin() + (ptrdiff_t)e);
					up.submit(buf.data(), buf.size());
					std::fill(live[(size_t)((k + 1) & 1)].begin(), live[(size_t)((k + 1) & 1)].end(), (uint8_t)0xEE);
					i = j; ++k;
				}
				up.finish();
			}
			++cases;
			if (got1 != exp1 || got2 != exp2) { ++bad; printf("seed %d cut %d: differs (%zu / %zu bytes against %zu / %zu)\n", seed, cut, got1.size(), got2.size(), exp1.size(), exp2.size()); }
		}
	}
	printf("%d cases, %d bad\n", cases, bad);
	return bad != 0;
}
