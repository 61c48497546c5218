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
 * (main.c:347-365; with -L, 100-column blocks annotated Q[..] / T[..], main.c:349-363).  Where the reference aligns one pair
 * at a time (main.c:311-326, :194-205), this tool collects pairs and sends them down as batches (bsa_align_batch /
 * bsa_edit_batch / bsa_kmer_edit_batch of include/bsalign_hip.h; -B pairs per batch, default 65536, at most 256 MB of bases)
 * and prints the records in input order: same bytes on stdout, the device busy.  `-B 1` and `-v` take the reference-named
 * single-pair API of include/bsalign_compat.h instead (a batch of one per pair: the plumbing case C1 of BASELINE.json).
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
#include "bsalign_hip.h"

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
		"                         [-Q gapo2] [-P gape2] [-L 1] [-R repeats] [-B pairs per batch] [-v] <in.fa[.gz]> ...\n"
		"       bsalign-hip edit  [-m global|extend|overlap|kmer] [-k ksz] [-W bandwidth] [-R repeats] [-B pairs per batch] [-v] <in.fa[.gz]> ...\n"
		"Options and output are those of `bsalign align` / `bsalign edit` (penalties are given as positive numbers).\n");
	return 1;
}

/* one alignment record on stdout (main.c:347-365 / :207-229) */
static void print_record(const char *qtag, const char *ttag, u1i *q, u4i qlen, u1i *t, u4i tlen, seqalign_result_t rs, u4v *cigars,
		char *alnstr[3], int *strn, int is_edit, int line, int verbose){
	if(!rs.mat) return;
	if(*strn < rs.aln){
		*strn = rs.aln;
		alnstr[0] = (char*)realloc(alnstr[0], *strn + 1);
		alnstr[1] = (char*)realloc(alnstr[1], *strn + 1);
		alnstr[2] = (char*)realloc(alnstr[2], *strn + 1);
	}
	if(verbose){
		u8i ci;
		fflush(stdout);
		fprintf(stderr, "CIGAR: %d\t", rs.aln);
		for(ci = 0; ci < cigars->size; ci++){
			const u4i len = cigars->buffer[ci] >> 4;
			const char op = "MIDNSHP=X*"[cigars->buffer[ci] & 0xf];
			if(len == 1) fprintf(stderr, "%c", op);
			else fprintf(stderr, "%d%c", len, op);
		}
		fprintf(stderr, "\n");
	}
	seqalign_cigar2alnstr(q, t, &rs, cigars, alnstr, (u4i)*strn);
	fprintf(stdout, "%s\t%d\t+\t%d\t%d\t%s\t%d\t+\t%d\t%d\t", qtag, (int)qlen, rs.qb, rs.qe, ttag, (int)tlen, rs.tb, rs.te);
	fprintf(stdout, "%d\t%.3f\t%d\t%d\t%d\t%d\n", rs.score, 1.0 * rs.mat / rs.aln, rs.mat, rs.mis, rs.ins, rs.del);
	if(!is_edit && line > 0){
		int i2, b, e, qn = rs.qb, tn = rs.tb;
		char keep;
		for(b = 0; b < *strn; b += 100){
			e = (b + 100 < *strn) ? b + 100 : *strn;
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

/* pairs waiting for the device: all bases in one blob, as the batch ABI takes them */
typedef struct {
	size_t n, cap;
	char **qtag, **ttag;
	uint64_t *qoff, *toff;
	uint32_t *qlen, *tlen;
	uint8_t *seqs; size_t sbytes, scap;
} queue_t;

static void queue_push(queue_t *Q, const str_t *tag, const str_t *seq){
	size_t k;
	if(Q->n == Q->cap){
		Q->cap = Q->cap ? Q->cap * 2 : 1024;
		Q->qtag = (char**)realloc(Q->qtag, Q->cap * sizeof(char*)); Q->ttag = (char**)realloc(Q->ttag, Q->cap * sizeof(char*));
		Q->qoff = (uint64_t*)realloc(Q->qoff, Q->cap * 8); Q->toff = (uint64_t*)realloc(Q->toff, Q->cap * 8);
		Q->qlen = (uint32_t*)realloc(Q->qlen, Q->cap * 4); Q->tlen = (uint32_t*)realloc(Q->tlen, Q->cap * 4);
	}
	for(k = 0; k < 2; k++){
		if(Q->sbytes + seq[k].n + 1 > Q->scap){
			while(Q->sbytes + seq[k].n + 1 > Q->scap) Q->scap = Q->scap ? Q->scap * 2 : (1u << 20);
			Q->seqs = (uint8_t*)realloc(Q->seqs, Q->scap);
		}
		memcpy(Q->seqs + Q->sbytes, seq[k].s, seq[k].n);
		if(k == 0){ Q->qoff[Q->n] = Q->sbytes; Q->qlen[Q->n] = (uint32_t)seq[k].n; Q->qtag[Q->n] = strdup(tag[k].s ? tag[k].s : ""); }
		else { Q->toff[Q->n] = Q->sbytes; Q->tlen[Q->n] = (uint32_t)seq[k].n; Q->ttag[Q->n] = strdup(tag[k].s ? tag[k].s : ""); }
		Q->sbytes += seq[k].n;
	}
	Q->n ++;
}

typedef struct {
	int is_edit, mode, W_opt, ksz, line, repm;
	b1i mtx[16]; int O, E, Qp, P;
} opts_t;

/* align everything that waits in ONE batch call, print the records in input order */
static void queue_flush(queue_t *Q, bsa_ctx_t *ctx, const opts_t *o, u4v *cigars, char *alnstr[3], int *strn){
	size_t k, cap = 16;
	int rc = BSA_OK, rep;
	if(Q->n == 0) return;
	for(k = 0; k < Q->n; k++) cap += (size_t)Q->qlen[k] + Q->tlen[k] + 8;
	bsa_result_t *out = (bsa_result_t*)calloc(Q->n, sizeof(bsa_result_t));
	uint32_t *status = (uint32_t*)calloc(Q->n, 4), *cig = (uint32_t*)malloc(cap * 4);
	uint64_t *coff = (uint64_t*)calloc(Q->n + 1, 8);
	for(rep = 0; rep < (o->repm > 0 ? o->repm : 1) && rc == BSA_OK; rep++){
		if(o->is_edit && o->mode == SEQALIGN_MODE_KMER){
			bsa_kmer_params_t kp; kp.ksz = (uint32_t)o->ksz; kp.threads = 0;
			rc = bsa_kmer_edit_batch(ctx, Q->seqs, Q->sbytes, Q->qoff, Q->qlen, Q->toff, Q->tlen, Q->n, &kp, out, cig, cap, coff, status);
		} else if(o->is_edit){
			bsa_edit_params_t ep; ep.mode = seqalign_mode_type(o->mode); ep.bandwidth = (uint32_t)o->W_opt;
			rc = bsa_edit_batch(ctx, Q->seqs, Q->sbytes, Q->qoff, Q->qlen, Q->toff, Q->tlen, Q->n, &ep, out, cig, cap, coff, status);
		} else {
			bsa_align_params_t ap;
			ap.mode = seqalign_mode_type(o->mode);
			ap.bandwidth = o->W_opt <= 0 ? 0u : (uint32_t)o->W_opt;              /* 0 = roundup(qlen, 16) per pair, main.c:314-315 */
			memcpy(ap.matrix, o->mtx, 16);
			ap.gapo1 = (int8_t)o->O; ap.gape1 = (int8_t)o->E; ap.gapo2 = (int8_t)o->Qp; ap.gape2 = (int8_t)o->P;
			rc = bsa_align_batch(ctx, Q->seqs, Q->sbytes, Q->qoff, Q->qlen, Q->toff, Q->tlen, Q->n, &ap, out, cig, cap, coff, status);
		}
	}
	if(rc != BSA_OK){
		fflush(stdout);
		fprintf(stderr, " -- device alignment failed (%d: %s) -- %s:%d --\n", rc, bsa_last_error(ctx), __FILE__, __LINE__);
		exit(1);
	}
	for(k = 0; k < Q->n; k++){
		seqalign_result_t rs;
		uint64_t w;
		if(status[k] & BSA_ST_TRACE){
			/* the reference's own traceback does not terminate (or crashes) on this pair: say so and go on with the next one */
			fflush(stdout);
			fprintf(stderr, " -- %s / %s: no alignment -- the traceback leaves the band (the reference does not terminate on this input) -- %s:%d --\n", Q->qtag[k], Q->ttag[k], __FILE__, __LINE__);
			free(Q->qtag[k]); free(Q->ttag[k]);
			continue;
		}
		memcpy(&rs, &out[k], sizeof(rs));
		clear_u4v(cigars);
		for(w = coff[k]; w < coff[k + 1]; w++) push_u4v(cigars, cig[w]);
		print_record(Q->qtag[k], Q->ttag[k], Q->seqs + Q->qoff[k], Q->qlen[k], Q->seqs + Q->toff[k], Q->tlen[k], rs, cigars, alnstr, strn, o->is_edit, o->line, 0);
		free(Q->qtag[k]); free(Q->ttag[k]);
	}
	free(out); free(status); free(cig); free(coff);
	Q->n = 0; Q->sbytes = 0;
}

int main(int argc, char **argv){
	if(argc < 2) return usage();
	const int is_edit = strcasecmp(argv[1], "edit") == 0;
	if(!is_edit && strcasecmp(argv[1], "align") != 0) return usage();
	argc --; argv ++;
	/* defaults: main.c:262-266 (align: overlap, M2 X-6 O-3 E-2 Q0 P0) and main.c:131-134 (edit: global) */
	int mode = is_edit ? SEQALIGN_MODE_GLOBAL : SEQALIGN_MODE_OVERLAP;
	int W_opt = 0, M = 2, X = -6, O = -3, E = -2, Q = 0, P = 0, line = 0, repm = 1, verbose = 0, ksz = 13, c;     /* ksz: main.c:141 */
	long batch = 65536;
	while((c = getopt(argc, argv, is_edit ? "hm:k:W:R:B:v" : "hm:W:M:X:O:E:Q:P:L:R:B:v")) != -1){
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
			case 'B': batch = atol(optarg); break;
			case 'v': verbose ++; break;
			default: return usage();
		}
	}
	if(optind >= argc) return usage();
	if(is_edit && mode == SEQALIGN_MODE_OVERLAP && W_opt){
		fprintf(stderr, " ** disable band in bsalign-edit's overlap mode ** \n");      /* main.c:170-173 */
		W_opt = 0;
	}
	const int single = verbose || batch <= 1;           /* one pair per call through the reference-named functions */
	opts_t opt;
	memset(&opt, 0, sizeof(opt));
	opt.is_edit = is_edit; opt.mode = mode; opt.W_opt = W_opt; opt.ksz = ksz; opt.line = line; opt.repm = repm;
	opt.O = O; opt.E = E; opt.Qp = Q; opt.P = P;
	banded_striped_epi8_seqalign_set_score_matrix(opt.mtx, (b1i)M, (b1i)X);
	b1v *mempool = adv_init_b1v(1024, 0, 16, 0);
	u4v *cigars = init_u4v(64);
	bsa_ctx_t *ctx = NULL;
	if(!single){
		const int rc = bsa_ctx_create(0, &ctx);
		if(rc != BSA_OK){ fprintf(stderr, " -- no usable HIP device (%d) -- %s:%d --\n", rc, __FILE__, __LINE__); return 1; }
		/* A process that runs once pays for every byte of device workspace it allocates (a fresh 100 GB allocation takes 2.5 - 4 s on this
		 * system, the whole-query alignment of 2000 x 10 kbp pairs in it 0.16 s): cap it, the batch then runs in chunks.  BSA_CLI_WS_GB overrides. */
		{
			const char *e = getenv("BSA_CLI_WS_GB");
			const double gb = e ? atof(e) : 16.0;
			if(gb > 0) bsa_ctx_set_workspace_limit(ctx, (size_t)(gb * 1073741824.0));
		}
	}
	queue_t Qu;
	memset(&Qu, 0, sizeof(Qu));
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
			if(!single){
				queue_push(&Qu, tag, seq);
				if(Qu.n >= (size_t)batch || Qu.sbytes >= ((size_t)256 << 20)) queue_flush(&Qu, ctx, &opt, cigars, alnstr, &strn);
				continue;
			}
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
					rs = banded_striped_epi8_seqalign_pairwise(q, qlen, t, tlen, mempool, cigars, mode, W, opt.mtx, (b1i)O, (b1i)E, (b1i)Q, (b1i)P, verbose);
				}
			}
			print_record(tag[0].s, tag[1].s, q, qlen, t, tlen, rs, cigars, alnstr, &strn, is_edit, line, verbose);
		}
		gzclose(rd.f);
	}
	if(!single){
		queue_flush(&Qu, ctx, &opt, cigars, alnstr, &strn);
		bsa_ctx_destroy(ctx);
		free(Qu.qtag); free(Qu.ttag); free(Qu.qoff); free(Qu.toff); free(Qu.qlen); free(Qu.tlen); free(Qu.seqs);
	}
	free(alnstr[0]); free(alnstr[1]); free(alnstr[2]);
	free(tag[0].s); free(tag[1].s); free(seq[0].s); free(seq[1].s); free(tmp.s); free(rtag.s); free(rseq.s);
	free_b1v(mempool);
	free_u4v(cigars);
	bsalign_compat_shutdown();
	return 0;
}
