// What a command's start-up is made of on this box: tools/ctx_probe (built by `make tools/ctx_probe`) times the HIP runtime's first calls one by one,
// then libseeksv_hip's own first launches (the code object's load).  usage: tools/ctx_probe [n_streams]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
	const int ns = argc > 1 ? atoi(argv[1]) : 3;
	double t = now(), t0 = t;
	auto lap = [&](const char *what) { const double n = now(); printf("%-44s %8.1f ms\n", what, (n - t) * 1e3); t = n; };
	int nd = 0;
	hipGetDeviceCount(&nd); lap("hipGetDeviceCount (runtime start-up)");
	hipSetDevice(0); lap("hipSetDevice");
	hipStream_t st[8];
	for (int i = 0; i < ns && i < 8; ++i) { hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking); char b[64]; snprintf(b, sizeof b, "hipStreamCreate #%d", i + 1); lap(b); }
	void *p = nullptr;
	hipMalloc(&p, 1 << 20); lap("hipMalloc 1 MB (first)");
	hipMemsetAsync(p, 0, 1 << 20, st[0]); hipStreamSynchronize(st[0]); lap("first memset + sync (first dispatch on the stream)");
	void *big = nullptr;
	hipMalloc(&big, (size_t)4 << 30); lap("hipMalloc 4 GB");
	void *h = nullptr;
	hipHostMalloc(&h, (size_t)256 << 20); lap("hipHostMalloc 256 MB");
	void *lib = dlopen("libseeksv_hip.so", RTLD_NOW); lap("dlopen libseeksv_hip.so (registers its code object)");
	if (lib) {
		typedef int (*create_t)(int, void **);
		typedef int (*fn1_t)(void *);
		create_t create = (create_t)dlsym(lib, "ssv_ctx_create");
		void *ctx = nullptr;
		if (create) { create(0, &ctx); lap("ssv_ctx_create (three more streams, events)"); }
	}
	printf("%-44s %8.1f ms\n", "total", (now() - t0) * 1e3);
	return 0;
}
