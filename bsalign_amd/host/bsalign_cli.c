/*
 * bsalign_cli.c -- `bsalign-hip align|edit`: the reference's pairwise command lines on the MI355X library.
 *
 * Same options, same input handling and byte-identical output as `bsalign align` / `bsalign edit`
 * (/root/reference/main.c:258-385 and :120-256): sequences are read from FASTA/FASTQ (plain or gzip), every two
 * consecutive records form one (query, target) pair, and for a pair with rs.mat != 0 the tool prints
 *
 *     qtag qlen + qb qe ttag tlen + tb te score identity mat mis ins del
 *     <query alignment string>
 *     <match string>
 *     <target alignment string>
 *
 * (main.c:347-365; with -L, 100-column blocks annotated Q[..] / T[..], main.c:349-363).  The DP itself is the
 * reference-named single-pair API of include/bsalign_compat.h, i.e. a batch of one on the GPU; this tool is the
 * plumbing case C1 of BASELINE.json, not a throughput path (use bsa_align_batch for that).
 *
 * Sequence encoding follows seq2basebank (dna.h:653-669): A/a 0, C/c 1, G/g 2, T/t 3, anything else & 3 = 0.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <unistd.h>
#include <zlib.h>
#include "bsalign_compat.h"

typedef struct { char *s; size_t n, cap; } str_t;

static void str_push(str_t *x, char c){
	if(x->n + 1 >= x->cap){ x->cap = x->cap ? x->cap * 2 : 256; x->s = (char*)realloc(x->s, x->cap); }
	x->s[x->n ++] = c; x->s[x->n] = 0;
}
static void str_clear(str_t *x){ x->n = 0; if(x->s) x->s[0] = 0; }

/* minimal FASTA / FASTQ record reader: tag = first word of the header, sequence lines concatenated */
typedef struct { gzFile f; int peek; } reader_t;
static int rd_getc(reader_t *r){ int c; if(r->peek != -2){ c = r->peek; r->peek = -2; return c; } return gzgetc(r->f); }
static void rd_ungetc(reader_t *r, int c){ r->peek = c; }
static int rd_line(reader_t *r, str_t *out){
	int c, any = 0;
	str_clear(out);
	while((c = rd_getc(r)) != -1){
		any = 1;
		if(c == '\n') break;
		if(c != '\r') str_push(out, (char)c);
	}
	return any;
}
static int read_record(reader_t *r, str_t *tag, str_t *seq, str_t *tmp){
	int c;
	size_t i;
	str_clear(tag); str_clear(seq);
	do { c = rd_getc(r); } while(c == '\n' || c == '\r' || c == ' ');
	if(c == -1) return 0;
	if(c != '>' && c != '@') return 0;
	const int fastq = (c == '@');
	if(!rd_line(r, tmp)) return 0;
	for(i = 0; i < tmp->n && tmp->s[i] != ' ' && tmp->s[i] != '\t'; i++) str_push(tag, tmp->s[i]);
	if(tag->s == NULL) str_push(tag, 0), tag->n = 0;
	if(fastq){
		if(!rd_line(r, tmp)) return 1;
		for(i = 0; i < tmp->n; i++) str_push(seq, tmp->s[i]);
		rd_line(r, tmp);               /* '+' line */
		rd_line(r, tmp);               /* qualities */
		return 1;
	}
	for(;;){
		c = rd_getc(r);
		if(c == -1) break;
		if(c == '>'){ rd_ungetc(r, c); break; }
		rd_ungetc(r, c);
		if(!rd_line(r, tmp)) break;
		for(i = 0; i < tmp->n; i++) if(tmp->s[i] != ' ' && tmp->s[i] != '\t') str_push(seq, tmp->s[i]);
	}
	return 1;
}

static u1i base_code(char c){
	switch(c){
		case 'A': case 'a': return 0;
		case 'C': case 'c': return 1;
		case 'G': case 'g': return 2;
		case 'T': case 't': return 3;
		default: return 0;                 /* base_bit_table gives 4, & 0x03 -> 0 (dna.h:662) */
	}
}

static int usage(void){
	fprintf(stderr,
		"bsalign-hip: pairwise alignment of consecutive FASTA/FASTQ records on an AMD MI355X\n"
		"Usage: bsalign-hip align [-m global|extend|overlap] [-W bandwidth] [-M mat] [-X mis] [-O gapo1] [-E gape1]\n"
		"                         [-Q gapo2] [-P gape2] [-L 1] [-R repeats] [-v] <in.fa[.gz]> ...\n"
		"       bsalign-hip edit  [-m global|extend|overlap|kmer] [-k ksz] [-W bandwidth] [-R repeats] [-v] <in.fa[.gz]> ...\n"
		"Options and output are those of `bsalign align` / `bsalign edit` (penalties are given as positive numbers).\n");
	return 1;
}

int main(int argc, char **argv){
	if(argc < 2) return usage();
	const int is_edit = strcasecmp(argv[1], "edit") == 0;
	if(!is_edit && strcasecmp(argv[1], "align") != 0) return usage();
	argc --; argv ++;
	/* defaults: main.c:262-266 (align: overlap, M2 X-6 O-3 E-2 Q0 P0) and main.c:131-134 (edit: global) */
	int mode = is_edit ? SEQALIGN_MODE_GLOBAL : SEQALIGN_MODE_OVERLAP;
	int W_opt = 0, M = 2, X = -6, O = -3, E = -2, Q = 0, P = 0, line = 0, repm = 1, verbose = 0, ksz = 13, c;     /* ksz: main.c:141 */
	while((c = getopt(argc, argv, is_edit ? "hm:k:W:R:v" : "hm:W:M:X:O:E:Q:P:L:R:v")) != -1){
		switch(c){
			case 'm':
				if(strcasecmp(optarg, "GLOBAL") == 0) mode = SEQALIGN_MODE_GLOBAL;
				else if(strcasecmp(optarg, "EXTEND") == 0) mode = SEQALIGN_MODE_EXTEND;
				else if(strcasecmp(optarg, "OVERLAP") == 0) mode = SEQALIGN_MODE_OVERLAP;
				else if(is_edit && strcasecmp(optarg, "KMER") == 0) mode = SEQALIGN_MODE_KMER;      /* main.c:153 */
				else return usage();
				break;
			case 'k': ksz = atoi(optarg); break;
			case 'W': W_opt = atoi(optarg); break;
			case 'M': M = atoi(optarg); break;
			case 'X': X = - atoi(optarg); break;
			case 'O': O = - atoi(optarg); break;
			case 'E': E = - atoi(optarg); break;
			case 'Q': Q = - atoi(optarg); break;
			case 'P': P = - atoi(optarg); break;
			case 'L': line = atoi(optarg); break;
			case 'R': repm = atoi(optarg); break;
			case 'v': verbose ++; break;
			default: return usage();
		}
	}
	if(optind >= argc) return usage();
	if(is_edit && mode == SEQALIGN_MODE_OVERLAP && W_opt){
		fprintf(stderr, " ** disable band in bsalign-edit's overlap mode ** \n");      /* main.c:170-173 */
		W_opt = 0;
	}
	b1i mtx[16];
	banded_striped_epi8_seqalign_set_score_matrix(mtx, (b1i)M, (b1i)X);
	b1v *mempool = adv_init_b1v(1024, 0, 16, 0);
	u4v *cigars = init_u4v(64);
	str_t tag[2] = {{0, 0, 0}, {0, 0, 0}}, seq[2] = {{0, 0, 0}, {0, 0, 0}}, tmp = {0, 0, 0}, rtag = {0, 0, 0}, rseq = {0, 0, 0};
	char *alnstr[3] = {NULL, NULL, NULL};
	int strn = 0, have = 0, fi;
	for(fi = optind; fi < argc; fi++){
		reader_t rd;
		rd.f = strcmp(argv[fi], "-") ? gzopen(argv[fi], "rb") : gzdopen(0, "rb");
		rd.peek = -2;
		if(rd.f == NULL){ fprintf(stderr, " -- cannot open %s --\n", argv[fi]); return 1; }
		while(read_record(&rd, &rtag, &rseq, &tmp)){
			if(rseq.n == 0) continue;                                              /* main.c:310 */
			str_clear(&tag[have]); str_clear(&seq[have]);
			size_t i;
			for(i = 0; i < rtag.n; i++) str_push(&tag[have], rtag.s[i]);
			if(tag[have].s == NULL){ str_push(&tag[have], 0); tag[have].n = 0; }
			for(i = 0; i < rseq.n; i++) str_push(&seq[have], (char)base_code(rseq.s[i]));
			if(++have < 2) continue;
			have = 0;
			u1i *q = (u1i*)seq[0].s, *t = (u1i*)seq[1].s;
			const u4i qlen = (u4i)seq[0].n, tlen = (u4i)seq[1].n;
			seqalign_result_t rs;
			int rep;
			memset(&rs, 0, sizeof(rs));
			for(rep = 0; rep < (repm > 0 ? repm : 1); rep++){
				if(is_edit && mode == SEQALIGN_MODE_KMER) rs = kmer_striped_seqedit_pairwise((u1i)ksz, q, qlen, t, tlen, mempool, cigars, verbose);
				else if(is_edit) rs = striped_seqedit_pairwise(q, qlen, t, tlen, mode, (u4i)W_opt, mempool, cigars, verbose);
				else {
					const u4i W = (W_opt <= 0) ? (qlen + 15u) / 16u * 16u : (u4i)W_opt;      /* main.c:314-315 */
					rs = banded_striped_epi8_seqalign_pairwise(q, qlen, t, tlen, mempool, cigars, mode, W, mtx, (b1i)O, (b1i)E, (b1i)Q, (b1i)P, verbose);
				}
			}
			if(!rs.mat) continue;
			if(strn < rs.aln){
				strn = rs.aln;
				alnstr[0] = (char*)realloc(alnstr[0], strn + 1);
				alnstr[1] = (char*)realloc(alnstr[1], strn + 1);
				alnstr[2] = (char*)realloc(alnstr[2], strn + 1);
			}
			if(verbose){
				u8i ci;
				fflush(stdout);
				fprintf(stderr, "CIGAR: %d\t", rs.aln);
				for(ci = 0; ci < cigars->size; ci++){
					if((cigars->buffer[ci] >> 4) == 1) fprintf(stderr, "%c", "MIDNSHP=X*"[cigars->buffer[ci] & 0xf]);
					else fprintf(stderr, "%d%c", cigars->buffer[ci] >> 4, "MIDNSHP=X*"[cigars->buffer[ci] & 0xf]);
				}
				fprintf(stderr, "\n");
			}
			seqalign_cigar2alnstr(q, t, &rs, cigars, alnstr, (u4i)strn);
			fprintf(stdout, "%s\t%d\t+\t%d\t%d\t%s\t%d\t+\t%d\t%d\t", tag[0].s, (int)qlen, rs.qb, rs.qe, tag[1].s, (int)tlen, rs.tb, rs.te);
			fprintf(stdout, "%d\t%.3f\t%d\t%d\t%d\t%d\n", rs.score, 1.0 * rs.mat / rs.aln, rs.mat, rs.mis, rs.ins, rs.del);
			if(!is_edit && line > 0){
				int i2, b, e, qn = rs.qb, tn = rs.tb;
				char keep;
				for(b = 0; b < strn; b += 100){
					e = (b + 100 < strn) ? b + 100 : strn;
					for(i2 = b; i2 < e; i2++){
						if(alnstr[0][i2] != '-') qn ++;
						if(alnstr[1][i2] != '-') tn ++;
					}
					keep = alnstr[0][e]; alnstr[0][e] = 0; fprintf(stdout, "%s\tQ[%d]\n", alnstr[0] + b, qn); alnstr[0][e] = keep;
					keep = alnstr[2][e]; alnstr[2][e] = 0; fprintf(stdout, "%s\n", alnstr[2] + b); alnstr[2][e] = keep;
					keep = alnstr[1][e]; alnstr[1][e] = 0; fprintf(stdout, "%s\tT[%d]\n", alnstr[1] + b, tn); alnstr[1][e] = keep;
				}
			} else {
				fprintf(stdout, "%s\n%s\n%s\n", alnstr[0], alnstr[2], alnstr[1]);
			}
			fflush(stdout);
		}
		gzclose(rd.f);
	}
	free(alnstr[0]); free(alnstr[1]); free(alnstr[2]);
	free(tag[0].s); free(tag[1].s); free(seq[0].s); free(seq[1].s); free(tmp.s); free(rtag.s); free(rseq.s);
	free_b1v(mempool);
	free_u4v(cigars);
	bsalign_compat_shutdown();
	return 0;
}
