// free_wipe_probe - do host <-> device copies slow down while the driver takes freed device memory back?  100 GB are allocated, written and freed; from that moment a 128 MB
// device-to-host copy runs again and again for four seconds: its rate and hipMemGetInfo's free bytes against the time since the free.  Then the same in a CHILD process that starts
// right after its parent-less predecessor (another child that held 100 GB) has exited - what a `seeksv` command sees that runs right behind another one.
// build: hipcc -O2 tools/free_wipe_probe.cpp -o tools/free_wipe_probe
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void hold(size_t gb)
{
	std::vector<void *> p;
	for (size_t g = 0; g < gb; g += 4) { void *q = nullptr; if (hipMalloc(&q, (size_t)4 << 30) != hipSuccess) break; hipMemset(q, 1, (size_t)4 << 30); p.push_back(q); }
	hipDeviceSynchronize();
	for (void *q : p) hipFree(q);
}

static void watch(const char *what, double seconds)
{
	const size_t sample = (size_t)128 << 20;
	hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	void *h = nullptr, *d = nullptr;
	hipHostMalloc(&h, sample, hipHostMallocDefault); hipMalloc(&d, sample); hipMemset(d, 3, sample); hipDeviceSynchronize();
	printf("%s\n   t [s]   D2H GB/s   free GB\n", what);
	const double t0 = now();
	double next = 0;
	while (now() - t0 < seconds) {
		const double a = now();
		hipMemcpyAsync(h, d, sample, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
		const double b = now();
		if (b - t0 >= next) { size_t fr = 0, tot = 0; hipMemGetInfo(&fr, &tot); printf("  %6.3f   %7.1f   %7.1f\n", b - t0, (double)sample / (b - a) / 1e9, (double)fr / 1e9); fflush(stdout); next += 0.1; }
	}
}

int main()
{
	if (fork() == 0) { // everything in children: the parent never touches the GPU
		hold(100);
		watch("the same process, right after it freed 100 GB (hipFree)", 2.5);
		_exit(0);
	}
	int st; wait(&st);
	if (fork() == 0) { watch("a new process, right after that one exited", 2.5); _exit(0); }
	wait(&st);
	if (fork() == 0) { hold(200); _exit(0); }   // (leaves with 200 GB just freed)
	wait(&st);
	if (fork() == 0) { watch("a new process, right after one that had held 200 GB exited", 4.0); _exit(0); }
	wait(&st);
	return 0;
}
