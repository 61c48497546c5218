/* A caller written against the reference's API (README.md:50-77 of the reference): only the include line differs.
 * usage: compat_main <mode> <bandwidth> <qseq ACGT> <tseq ACGT>     prints the result record and the CIGAR */
#include "bsalign_compat.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static u1i *encode(const char *s, u4i *len){
	u4i i, n = (u4i)strlen(s);
	u1i *r = (u1i*)malloc(n + 1);
	for(i = 0; i < n; i++){ switch(s[i]){ case 'A': r[i] = 0; break; case 'C': r[i] = 1; break; case 'G': r[i] = 2; break; default: r[i] = 3; } }
	*len = n;
	return r;
}

int main(int argc, char **argv){
	b1i mtx[16];
	b1v *mempool;
	u4v *cigars;
	u1i *q, *t;
	u4i ql, tl, i;
	seqalign_result_t rs;
	char *aln[3] = {NULL, NULL, NULL};
	int mode, bw;
	if(argc < 5) return 2;
	mode = atoi(argv[1]); bw = atoi(argv[2]);
	q = encode(argv[3], &ql); t = encode(argv[4], &tl);
	banded_striped_epi8_seqalign_set_score_matrix(mtx, 2, -6);
	mempool = adv_init_b1v(1024, 0, 16, 0);
	cigars = init_u4v(64);
	rs = banded_striped_epi8_seqalign_pairwise(q, ql, t, tl, mempool, cigars, mode, bw, mtx, -3, -2, 0, 0, 0);
	printf("ALIGN %d %d %d %d %d %d %d %d %d %d", rs.score, rs.qb, rs.qe, rs.tb, rs.te, rs.mat, rs.mis, rs.ins, rs.del, rs.aln);
	for(i = 0; i < cigars->size; i++) printf(" %u", cigars->buffer[i]);
	printf("\n");
	seqalign_cigar2alnstr(q, t, &rs, cigars, aln, 0);
	printf("%s\n%s\n%s\n", aln[0], aln[2], aln[1]);
	rs = striped_seqedit_pairwise(q, ql, t, tl, mode, 0, mempool, cigars, 0);
	printf("EDIT %d %d %d %d %d %d %d %d %d %d", rs.score, rs.qb, rs.qe, rs.tb, rs.te, rs.mat, rs.mis, rs.ins, rs.del, rs.aln);
	for(i = 0; i < cigars->size; i++) printf(" %u", cigars->buffer[i]);
	printf("\n");
	free_b1v(mempool); free_u4v(cigars); free(q); free(t);
	bsalign_compat_shutdown();
	return 0;
}
