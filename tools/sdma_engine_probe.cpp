// sdma_engine_probe - which SDMA engine copies over the host link at what rate?  hipMemcpyAsync's engine is the runtime's choice and on some boxes the same 0.55 GB copy takes 9.7 or
// 18.2 ms (tools/d2h_probe.cpp).  Here every engine that hsa_amd_memory_copy_engine_status reports for device <-> host is given the copy by name (hsa_amd_memory_async_copy_on_engine).
// build: hipcc -O2 tools/sdma_engine_probe.cpp -o tools/sdma_engine_probe -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <vector>
static hsa_agent_t g_gpu, g_cpu; static bool have_gpu = false, have_cpu = false;
static hsa_status_t on_agent(hsa_agent_t a, void *) {
	hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
	if (t == HSA_DEVICE_TYPE_GPU && !have_gpu) { g_gpu = a; have_gpu = true; }
	if (t == HSA_DEVICE_TYPE_CPU && !have_cpu) { g_cpu = a; have_cpu = true; }
	return HSA_STATUS_SUCCESS;
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	const size_t n = 553303952;
	void *d = nullptr, *h = nullptr;
	hipMalloc(&d, n); hipMemset(d, 3, n); hipHostMalloc(&h, n, hipHostMallocDefault); hipDeviceSynchronize();
	if (hsa_init() != HSA_STATUS_SUCCESS) { printf("hsa_init failed\n"); return 1; }
	hsa_iterate_agents(on_agent, nullptr);
	hsa_signal_t sig; hsa_signal_create(1, 0, nullptr, &sig);
	for (int dir = 0; dir < 2; ++dir) {
		hsa_agent_t dst_a = dir == 0 ? g_cpu : g_gpu, src_a = dir == 0 ? g_gpu : g_cpu;
		void *dst = dir == 0 ? h : d, *src = dir == 0 ? d : h;
		uint32_t avail = 0, pref = 0;
		const hsa_status_t s1 = hsa_amd_memory_copy_engine_status(dst_a, src_a, &avail);
		const hsa_status_t s2 = hsa_amd_memory_get_preferred_copy_engine(dst_a, src_a, &pref);
		printf("%s: engines available 0x%x (status %d), preferred 0x%x (status %d)\n", dir == 0 ? "device -> host" : "host -> device", avail, (int)s1, pref, (int)s2);
		for (int e = 0; e < 16; ++e) {
			if (!((avail >> e) & 1u)) continue;
			printf("  engine %2d:", e);
			for (int rep = 0; rep < 4; ++rep) {
				hsa_signal_store_relaxed(sig, 1);
				const double t0 = now();
				const hsa_status_t st = hsa_amd_memory_async_copy_on_engine(dst, dst_a, src, src_a, n, 0, nullptr, sig, (hsa_amd_sdma_engine_id_t)(1u << e), true);
				if (st != HSA_STATUS_SUCCESS) { printf(" (refused: %d)", (int)st); break; }
				hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
				printf(" %6.2f", now() - t0);
			}
			printf(" ms\n");
		}
		// the runtime's own choice, for comparison
		printf("  hsa_amd_memory_async_copy (the runtime picks):");
		for (int rep = 0; rep < 6; ++rep) {
			hsa_signal_store_relaxed(sig, 1);
			const double t0 = now();
			hsa_amd_memory_async_copy(dst, dst_a, src, src_a, n, 0, nullptr, sig);
			hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
			printf(" %6.2f", now() - t0);
		}
		printf(" ms\n");
	}
	return 0;
}
