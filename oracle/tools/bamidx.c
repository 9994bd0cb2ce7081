/* Test tooling (ours): build in.bam.bai through the reference's libbam (samtools 0.1.16). */
#include <stdio.h>
#include "bam.h"
int main(int argc, char **argv)
{
	if (argc != 2) { fprintf(stderr, "usage: bamidx in.bam\n"); return 2; }
	return bam_index_build(argv[1]);
}
