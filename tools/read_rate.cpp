// read_rate - how fast can this box move a file that sits in the page cache (/dev/shm) into the GPU?  The CLI's reader copies the file into
// page-locked staging buffers with pread on all CPUs the cgroup grants (profiles/r03_cli_at_scale_617M_records.json: 20 GB/s on 16 CPUs, the
// bound of `seeksv getclip -Z` at whole-genome size).  This tool measures the alternatives on the same box:
//   pread  T threads x slice size -> pinned          (what ssvh_bam_read_blocks does)
//   mmap + memcpy, T threads -> pinned
//   mmap + hipHostRegister(piece) + hipMemcpyAsync + hipHostUnregister      (no CPU copy: DMA out of the page cache)
//   hipMemcpy straight from the mapping (pageable: the runtime stages it)
// build: hipcc -O2 -std=c++17 tools/read_rate.cpp -o tools/read_rate -lpthread      usage: tools/read_rate [GB=8] [dir=/dev/shm]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F> static void par(int T, F f)
{
	std::vector<std::thread> th;
	for (int t = 1; t < T; ++t) th.emplace_back(f, t);
	f(0);
	for (auto &x : th) x.join();
}

int main(int argc, char **argv)
{
	const double gb = argc > 1 ? atof(argv[1]) : 8;
	const std::string dir = argc > 2 ? argv[2] : "/dev/shm";
	size_t bytes = (size_t)(gb * (1ull << 30)) & ~(size_t)((2 << 20) - 1);
	const bool existing = argc > 3; // an existing file (e.g. the BAM the CLI is timed on) instead of a fresh one
	const std::string path = existing ? argv[3] : dir + "/ssv_read_rate.bin";
	{
		FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
		char line[128] = "";
		if (f) { if (fgets(line, sizeof(line), f)) printf("cpu.max: %s", line); fclose(f); }
		printf("hardware threads: %u\n", std::thread::hardware_concurrency());
	}
	if (existing) {
		struct stat sb;
		if (stat(path.c_str(), &sb) != 0) { perror("stat"); return 1; }
		bytes = std::min(bytes, (size_t)sb.st_size & ~(size_t)((2 << 20) - 1));
		printf("existing file %s, first %.1f GB\n", path.c_str(), bytes / 1e9);
	} else
	// the file: written by 16 threads (its pages end up wherever those threads ran, like the BAM writer's)
	{
		const int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0600);
		if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { perror("create"); return 1; }
		const double t0 = now();
		par(16, [&](int t) {
			std::vector<uint8_t> buf((size_t)4 << 20);
			uint64_t x = 88172645463325252ull + (uint64_t)t;
			for (size_t off = (size_t)t * buf.size(); off < bytes; off += 16 * buf.size()) {
				for (size_t i = 0; i < buf.size(); i += 8) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; memcpy(&buf[i], &x, 8); }
				if (pwrite(fd, buf.data(), std::min(buf.size(), bytes - off), (off_t)off) < 0) perror("pwrite");
			}
		});
		printf("wrote %.1f GB in %.2f s\n", bytes / 1e9, now() - t0);
		close(fd);
	}
	const int fd = open(path.c_str(), O_RDONLY);
	void *pin = nullptr;
	const size_t STAGE = (size_t)2 << 30;
	double t0 = now();
	if (hipHostMalloc(&pin, STAGE, hipHostMallocDefault) != hipSuccess) { fprintf(stderr, "hipHostMalloc failed\n"); return 1; }
	printf("hipHostMalloc %.1f GB: %.3f s\n", STAGE / 1e9, now() - t0);
	void *dev = nullptr;
	if (hipMalloc(&dev, STAGE) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
	hipStream_t st;
	hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	uint8_t *P = static_cast<uint8_t *>(pin);

	// 1. pread into pinned memory
	for (int T : {8, 16, 32, 64}) for (size_t slice : {(size_t)1 << 20, (size_t)4 << 20, (size_t)32 << 20}) {
		const double t = now();
		for (size_t base = 0; base < bytes; base += STAGE) {
			const size_t n = std::min(STAGE, bytes - base);
			std::atomic<size_t> next{0};
			par(T, [&](int) {
				for (;;) {
					const size_t off = next.fetch_add(slice);
					if (off >= n) return;
					size_t done = 0; const size_t len = std::min(slice, n - off);
					while (done < len) { const ssize_t g = pread(fd, P + off + done, len - done, (off_t)(base + off + done)); if (g <= 0) { perror("pread"); return; } done += (size_t)g; }
				}
			});
		}
		printf("pread   T=%2d slice=%2zu MB: %6.2f GB/s\n", T, slice >> 20, bytes / 1e9 / (now() - t));
		fflush(stdout);
	}
	// 2. mmap + memcpy into pinned memory
	uint8_t *map = static_cast<uint8_t *>(mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0));
	if (map == MAP_FAILED) { perror("mmap"); return 1; }
	for (int pass = 0; pass < 2; ++pass) for (int T : {16, 32, 64}) {
		const size_t slice = (size_t)4 << 20;
		const double t = now();
		for (size_t base = 0; base < bytes; base += STAGE) {
			const size_t n = std::min(STAGE, bytes - base);
			std::atomic<size_t> next{0};
			par(T, [&](int) {
				for (;;) {
					const size_t off = next.fetch_add(slice);
					if (off >= n) return;
					memcpy(P + off, map + base + off, std::min(slice, n - off));
				}
			});
		}
		printf("mmap+memcpy (pass %d: %s) T=%2d: %6.2f GB/s\n", pass, pass ? "mapped" : "first touch of the mapping", T, bytes / 1e9 / (now() - t));
		fflush(stdout);
	}
	// 3. pinned -> device alone (the PCIe side), for scale
	{
		const double t = now();
		for (int k = 0; k < 4; ++k) hipMemcpyAsync(dev, pin, STAGE, hipMemcpyHostToDevice, st);
		hipStreamSynchronize(st);
		printf("pinned -> device: %6.2f GB/s\n", 4 * STAGE / 1e9 / (now() - t));
	}
	// 4. register pieces of the mapping and copy out of them
	size_t region = 0; // every variant gets 2 GB of the file that no variant before it has registered
	for (size_t piece : {(size_t)256 << 20, (size_t)1 << 30, (size_t)64 << 20}) {
		double t_reg = 0, t_cp = 0, t_unreg = 0;
		size_t done = 0;
		bool ok = true;
		const size_t r0 = region;
		region += (size_t)2 << 30;
		for (size_t base = r0; base + piece <= bytes && base < r0 + ((size_t)2 << 30); base += piece) {
			double t = now();
			hipError_t e = hipHostRegister(map + base, piece, hipHostRegisterDefault);
			if (e != hipSuccess) { printf("hipHostRegister(mapping of a tmpfs file): %s\n", hipGetErrorString(e)); ok = false; break; }
			t_reg += now() - t; t = now();
			e = hipMemcpyAsync(dev, map + base, piece, hipMemcpyHostToDevice, st);
			if (e == hipSuccess) e = hipStreamSynchronize(st);
			if (e != hipSuccess) { printf("copy out of a registered mapping: %s\n", hipGetErrorString(e)); ok = false; break; }
			t_cp += now() - t; t = now();
			hipHostUnregister(map + base);
			t_unreg += now() - t;
			done += piece;
		}
		if (ok && done) printf("register %4zu MB pieces: register %.3f s/GB, copy %6.2f GB/s, unregister %.3f s/GB  => %6.2f GB/s in all (serial)\n", piece >> 20, t_reg / (done / 1e9), done / 1e9 / t_cp,
		                       t_unreg / (done / 1e9), done / 1e9 / (t_reg + t_cp + t_unreg));
		fflush(stdout);
	}
	// 4b. registration on T threads at once (does the kernel's pinning scale?)
	for (int T : {4, 16}) {
		const size_t piece = (size_t)256 << 20;
		if (region + ((size_t)1 << 30) > bytes) break;
		uint8_t *const map0 = map;
		uint8_t *map = map0 + region; // (shadows: this variant's own GB)
		region += (size_t)1 << 30;
		const size_t total = (size_t)1 << 30;
		std::atomic<size_t> next{0};
		std::atomic<int> bad{0};
		const double t = now();
		par(T, [&](int) {
			for (;;) {
				const size_t off = next.fetch_add(piece);
				if (off >= total) return;
				if (hipHostRegister(map + off, piece, hipHostRegisterDefault) != hipSuccess) { bad++; return; }
			}
		});
		const double t_reg = now() - t;
		if (bad) { printf("parallel registration failed\n"); break; }
		const double t2 = now();
		for (size_t off = 0; off < total; off += STAGE) { hipError_t e = hipMemcpyAsync(dev, map + off, std::min(STAGE, total - off), hipMemcpyHostToDevice, st); if (e != hipSuccess) printf("copy: %s\n", hipGetErrorString(e)); }
		{ hipError_t e = hipStreamSynchronize(st); if (e != hipSuccess) printf("sync: %s\n", hipGetErrorString(e)); }
		const double t_cp = now() - t2;
		const double t3 = now();
		for (size_t off = 0; off < total; off += piece) hipHostUnregister(map + off);
		printf("register on %2d threads: %.3f s/GB (%.2f GB/s); copy %6.2f GB/s; unregister (1 thread) %.3f s/GB\n", T, t_reg / (total / 1e9), total / 1e9 / t_reg, total / 1e9 / t_cp, (now() - t3) / (total / 1e9));
		fflush(stdout);
	}
	// 5. hipMemcpy straight out of the mapping (pageable memory)
	{
		const size_t n = std::min(bytes, STAGE);
		const double t = now();
		hipError_t e = hipMemcpy(dev, map, n, hipMemcpyHostToDevice);
		printf("hipMemcpy from the mapping (pageable): %s, %6.2f GB/s\n", hipGetErrorString(e), n / 1e9 / (now() - t));
	}
	munmap(map, bytes);
	close(fd);
	if (!existing) unlink(path.c_str());
	return 0;
}
