// exit_cost - what does the END of a process cost?  `seeksv run` on a half-genome file: 0.38 s between its last statement and the caller's wait() returning (round 6,
// bench.py: exit_to_reaped_s).  A child process allocates D GB of device memory (4 GB pieces) and page-locks H GB of anonymous host memory (with or without
// transparent huge pages), touches both, and leaves through _exit; the parent times fork -> ready (pipe) and _exit -> reaped.
// build: hipcc -O2 tools/exit_cost.cpp -o tools/exit_cost      usage: tools/exit_cost
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(); }

static void child(int dev_gb, int host_gb, bool thp, bool free_first, int wfd, int n_streams = 0, bool destroy_streams = false)
{
	std::vector<hipStream_t> st((size_t)n_streams);
	void *tiny = nullptr;
	if (n_streams) { (void)hipMalloc(&tiny, 4096); for (auto &q : st) { (void)hipStreamCreateWithFlags(&q, hipStreamNonBlocking); (void)hipMemsetAsync(tiny, 0, 4096, q); (void)hipStreamSynchronize(q); } }
	std::vector<void *> d;
	for (int g = 0; g < dev_gb; g += 4) { void *p = nullptr; if (hipMalloc(&p, (size_t)4 << 30) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); _exit(2); } hipMemset(p, 1, (size_t)4 << 30); d.push_back(p); }
	hipDeviceSynchronize();
	std::vector<void *> h;
	double t_lock = 0;
	for (int g = 0; g < host_gb; ++g) {
		const size_t n = (size_t)1 << 30, slack = (size_t)2 << 20;
		uint8_t *m = static_cast<uint8_t *>(mmap(nullptr, n + slack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
		uint8_t *a = reinterpret_cast<uint8_t *>(((uintptr_t)m + slack - 1) & ~(uintptr_t)(slack - 1));
		if (thp) madvise(a, n, MADV_HUGEPAGE);
		std::vector<std::thread> th;
		for (int t = 0; t < 16; ++t) th.emplace_back([=] { for (size_t o = n / 16 * (size_t)t; o < n / 16 * (size_t)(t + 1); o += 4096) a[o] = 1; });
		for (auto &x : th) x.join();
		const double t0 = now();
		if (hipHostRegister(a, n, hipHostRegisterDefault) != hipSuccess) { fprintf(stderr, "hipHostRegister failed\n"); _exit(2); }
		t_lock += now() - t0;
		h.push_back(a);
	}
	double t_free = 0;
	if (free_first) {
		const double t0 = now();
		for (void *p : h) hipHostUnregister(p);
		for (void *p : d) hipFree(p);
		t_free = now() - t0;
	}
	if (destroy_streams) for (auto &q : st) (void)hipStreamDestroy(q);
	double msg[3] = {now(), t_lock, t_free};
	if (write(wfd, msg, sizeof(msg)) != (ssize_t)sizeof(msg)) _exit(3);
	_exit(0);
}

int main()
{
	struct Case { int dev_gb, host_gb; bool thp, free_first; } cases[] = {{0, 0, false, false}, {48, 0, false, false}, {0, 4, false, false}, {0, 4, true, false}, {48, 4, false, false}, {48, 4, true, false}, {48, 4, false, true}, {96, 4, true, false}};
	for (const Case &c : cases) for (int rep = 0; rep < 2; ++rep) {
		int fd[2];
		if (pipe(fd) != 0) return 1;
		const double t0 = now();
		const pid_t pid = fork();
		if (pid == 0) { close(fd[0]); child(c.dev_gb, c.host_gb, c.thp, c.free_first, fd[1]); }
		close(fd[1]);
		double msg[3] = {0, 0, 0};
		if (read(fd[0], msg, sizeof(msg)) != (ssize_t)sizeof(msg)) { printf("child failed\n"); }
		int st;
		waitpid(pid, &st, 0);
		const double t1 = now();
		close(fd[0]);
		printf("device %3d GB, page-locked host %d GB (%s)%s: set-up %.3f s (page-locking %.3f s), %s_exit -> reaped %.3f s\n", c.dev_gb, c.host_gb, c.thp ? "huge pages asked for" : "4 KB pages",
		       c.free_first ? ", freed by hand first" : "", msg[0] - t0, msg[1], c.free_first ? "hipFree + unregister by hand " : "", t1 - msg[0]);
		if (c.free_first) printf("    (by hand: %.3f s)\n", msg[2]);
		fflush(stdout);
	}
	// streams: every one is a hardware queue the kernel has to take down
	for (int ns : {0, 1, 3, 6, 12}) for (int destroy = 0; destroy < 2; ++destroy) for (int rep = 0; rep < 2; ++rep) {
		if (ns == 0 && destroy) continue;
		int fd[2];
		if (pipe(fd) != 0) return 1;
		const double t0 = now();
		const pid_t pid = fork();
		if (pid == 0) { close(fd[0]); child(0, 0, false, false, fd[1], ns, destroy != 0); }
		close(fd[1]);
		double msg[3] = {0, 0, 0};
		if (read(fd[0], msg, sizeof(msg)) != (ssize_t)sizeof(msg)) printf("child failed\n");
		int stt;
		waitpid(pid, &stt, 0);
		const double t1 = now();
		close(fd[0]);
		printf("%2d streams%s: set-up %.3f s, _exit -> reaped %.3f s\n", ns, destroy ? " (destroyed by hand first)" : "", msg[0] - t0, t1 - msg[0]);
		fflush(stdout);
	}
	return 0;
}
