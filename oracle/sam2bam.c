/* oracle/sam2bam.c -- TEST INFRASTRUCTURE ONLY.  Our own minimal SAM -> BAM
 * converter over the vendored samtools-0.1.19 library API (sam.h), so that
 * tools/make_goldens.py can feed synthetic alignments to the reference
 * binary without the samtools executable.  usage: sam2bam in.sam out.bam   */
#include <stdio.h>
#include "sam.h"

int main(int argc, char **argv)
{
   if (argc != 3) {
      fprintf(stderr, "usage: %s in.sam out.bam\n", argv[0]);
      return 2;
   }
   samfile_t *in = samopen(argv[1], "r", 0);
   if (!in) return 1;
   samfile_t *out = samopen(argv[2], "wb", in->header);
   if (!out) return 1;
   bam1_t *b = bam_init1();
   long n = 0;
   while (samread(in, b) >= 0) {
      samwrite(out, b);
      ++n;
   }
   bam_destroy1(b);
   samclose(out);
   samclose(in);
   fprintf(stderr, "sam2bam: %ld records\n", n);
   return 0;
}
