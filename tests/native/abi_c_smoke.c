/* A plain C99 translation unit over both public headers: proves that a C caller (cgo, JNI stubs, the reference's own C++ call sites ...)
 * can include and link the ABI without C++ or torch.  Without a GPU the HIP library must refuse to create a context; with one it must
 * hand out a context and take it back.  Compiled and run by tests/test_abi_load.py. */
#include <stdio.h>
#include <string.h>

#include "seeksv_hip.h"
#include "seeksv_host.h"

int main(void)
{
	ssv_ctx *ctx = NULL;
	ssv_batch_t b;
	ssv_cluster_table t;
	ssv_realign_hit h;
	int rc;
	memset(&b, 0, sizeof(b)); memset(&t, 0, sizeof(t)); memset(&h, 0, sizeof(h));
	if (ssv_abi_version() != SSV_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 2; }
	if (sizeof(h) != 32) { fprintf(stderr, "ssv_realign_hit layout\n"); return 2; }
	if (ssv_table_block_bytes(30, 121) != ((2 * 30 + 2 * 121 + 3) & ~3)) { fprintf(stderr, "block bytes\n"); return 2; }
	rc = ssv_ctx_create(0, &ctx);
	if (rc == SSV_OK) {
		if (ssv_clip_scan(ctx, NULL) != SSV_E_ARG) { fprintf(stderr, "NULL batch accepted\n"); return 3; }
		if (ssv_clip_scan(ctx, &b) != SSV_E_STATE) { fprintf(stderr, "scan before begin accepted\n"); return 3; }
		ssv_ctx_destroy(ctx);
		printf("gpu context ok\n");
	} else if (rc == SSV_E_NODEVICE) {
		printf("no device: %s\n", ssv_last_error(NULL));
	} else { fprintf(stderr, "unexpected status %d\n", rc); return 4; }
	if (ssvh_bam_open("/nonexistent/file.bam", (ssvh_bam **)&ctx) == 0) { fprintf(stderr, "opened a missing file\n"); return 5; }
	printf("host: %s\n", ssvh_last_error());
	return 0;
}
