/* Test tooling (ours): SAM text -> BAM through the reference's libbam (samtools 0.1.16).
 * Stands in for `samtools view -Sb` of example/seeksv.sh:3 (no samtools in this image). */
#include <stdio.h>
#include "sam.h"
int main(int argc, char **argv)
{
	if (argc != 3) { fprintf(stderr, "usage: sam2bam in.sam out.bam\n"); return 2; }
	samfile_t *in = samopen(argv[1], "r", 0);
	if (!in || !in->header) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
	samfile_t *out = samopen(argv[2], "wb", in->header);
	if (!out) { fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
	bam1_t *b = bam_init1();
	long n = 0;
	while (samread(in, b) >= 0) { samwrite(out, b); ++n; }
	bam_destroy1(b);
	samclose(out);
	samclose(in);
	fprintf(stderr, "sam2bam: %ld records\n", n);
	return 0;
}
