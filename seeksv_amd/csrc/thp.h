// thp.h - transparent huge pages for the large host buffers (rows of text, inflated files, record arrays: hundreds of MB to GB each).
// A fresh 4 KB page costs a fault to get (~0.25 us/KB: writing 2.5 GB of rows is 600 K faults spread over the formatting threads) and, when the process ends, the kernel
// gives the pages back one by one on one thread before the caller's wait() returns (tools/exit_cost.cpp; `seeksv run` on half a genome: 0.38 s behind its last statement).
// With madvise(MADV_HUGEPAGE) the 2 MB aligned inside of a buffer comes in 2 MB pages where the kernel grants them (/sys/kernel/mm/transparent_hugepage/enabled: always or
// madvise; otherwise nothing changes).  Call it on a buffer BEFORE its first write.
#pragma once

#include <sys/mman.h>

#include <cstddef>
#include <cstdint>

namespace ssv {

inline void thp_advise(const void *p, size_t n)
{
	const uintptr_t huge = (uintptr_t)2 << 20;
	if (!p || n < 2 * huge) return;
	const uintptr_t lo = ((uintptr_t)p + huge - 1) & ~(huge - 1), hi = ((uintptr_t)p + n) & ~(huge - 1);
	if (hi > lo) (void)madvise(reinterpret_cast<void *>(lo), (size_t)(hi - lo), MADV_HUGEPAGE);
}

} // namespace ssv
