/*
 * bsa_kmer.cpp -- k-mer anchored edit alignment (the reference's kmer_striped_seqedit_pairwise,
 * /root/reference/bsalign.h:1209-1536; `bsalign edit -m kmer -k <ksz>`, main.c:153,196-206) on top of the device
 * edit path.
 *
 * The reference finds k-mers that occur exactly once in each sequence and on the same strand, chains them (longest
 * increasing subsequence on the target offsets, then an iterated diagonal-outlier filter) and runs the plain edit DP
 * only between consecutive anchors: the head in front of the first anchor as a reversed EXTEND alignment, the gaps as
 * GLOBAL alignments, the tail as an EXTEND alignment.  Every one of those small alignments is independent of the
 * others, so here the host does the chaining (threads over pairs) and ALL segments of ALL pairs of a batch go to the
 * GPU as two bsa_edit_batch calls (EXTEND: reversed heads and tails; GLOBAL: gaps); the per-pair CIGAR is then stitched on the host.
 *
 * Host pieces are exported on their own (bsa_kmer_chain / bsa_kmer_segments / bsa_kmer_assemble) so that the chaining
 * and stitching can be checked without a GPU.  There is no CPU alignment in this file.
 */
#include "../../include/bsalign_hip.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <thread>
#include <vector>

namespace {

struct Hit  { uint32_t qoff, toff; uint8_t keep; };

const uint32_t NONE = 0xFFFFFFFFu;

/* one k-mer occurrence in 64 bits: canonical k-mer (30) | offset (32) | from the target (1) | reverse strand is canonical (1) */
inline uint32_t km_kmer(uint64_t x){ return (uint32_t)(x >> 34); }
inline uint32_t km_off(uint64_t x){ return (uint32_t)(x >> 2); }
inline uint32_t km_flg(uint64_t x){ return (uint32_t)(x >> 1) & 1u; }
inline uint32_t km_dir(uint64_t x){ return (uint32_t)x & 1u; }

/* all canonical k-mers of one sequence (bsalign.h:1236-1256) */
void kmers_of(std::vector<uint64_t> &dst, const uint8_t *seq, uint32_t len, uint32_t ksz, uint32_t flg){
	const uint32_t mask = 0xFFFFFFFFu >> ((16 - ksz) << 1), top = (ksz - 1) << 1;
	uint32_t fwd = 0, rev = 0, i;
	for(i = 0; i + 1 < ksz && i < len; i++){
		const uint32_t b = seq[i];
		fwd = (fwd << 2) | b;
		rev = (rev >> 2) | (((~b) & 3u) << top);
	}
	for(; i < len; i++){
		const uint32_t b = seq[i];
		fwd = ((fwd << 2) | b) & mask;
		rev = (rev >> 2) | (((~b) & 3u) << top);
		const uint32_t dir = rev < fwd;
		const uint64_t kmer = (dir ? rev : fwd) & 0x3FFFFFFFu;
		dst.push_back(kmer << 34 | (uint64_t)(i + 1 - ksz) << 2 | flg << 1 | dir);
	}
}

/* LSD radix sort on the k-mer bits only (2 * ksz of them); the order inside a run of equal k-mers is not used */
void sort_by_kmer(std::vector<uint64_t> &a, std::vector<uint64_t> &tmp, uint32_t ksz){
	const size_t n = a.size();
	if(n < 256){ std::sort(a.begin(), a.end()); return; }
	tmp.resize(n);
	uint64_t *src = a.data(), *dst = tmp.data();
	const uint32_t bits = 2 * ksz;
	const uint32_t passes = (bits + 10) / 11;
	const uint32_t per = (bits + passes - 1) / passes;
	for(uint32_t p = 0; p < passes; p++){
		const uint32_t sh = 34 + p * per, m = (1u << per) - 1;
		uint32_t cnt[2049];
		memset(cnt, 0, sizeof(uint32_t) * ((size_t)m + 2));
		for(size_t i = 0; i < n; i++) cnt[((src[i] >> sh) & m) + 1] ++;
		for(uint32_t v = 0; v < m; v++) cnt[v + 1] += cnt[v];
		for(size_t i = 0; i < n; i++) dst[cnt[(src[i] >> sh) & m] ++] = src[i];
		std::swap(src, dst);
	}
	if(src != a.data()) memcpy(a.data(), src, n * sizeof(uint64_t));
}

/* coverage threshold (bsalign.h:1220-1221) */
uint32_t min_cover(uint32_t qlen, uint32_t tlen, uint32_t ksz){
	uint32_t c = (uint32_t)(std::min(qlen, tlen) * 0.05 + 1);
	return std::min(c, 2 * ksz);
}

/* returns the anchors that survive, in query order (empty = align the whole pair globally) */
void chain(uint32_t ksz, const uint8_t *q, uint32_t qlen, const uint8_t *t, uint32_t tlen, std::vector<Hit> &hits){
	hits.clear();
	if(ksz > 15) ksz = 15;
	if(ksz == 0) return;
	const uint32_t cmin = min_cover(qlen, tlen, ksz);
	std::vector<uint64_t> km, tmp;
	km.reserve((size_t)qlen + tlen + 1);
	kmers_of(km, q, qlen, ksz, 0);
	kmers_of(km, t, tlen, ksz, 1);
	sort_by_kmer(km, tmp, ksz);
	const uint32_t cnt = (uint32_t)km.size();
	km.push_back(0);                                      /* the reference's zeroed sentinel: a run of k-mer 0 that reaches the end is never closed (bsalign.h:1259-1262) */
	for(uint32_t b = 0, i = 0; i <= cnt; i++){
		if(km_kmer(km[i]) == km_kmer(km[b])) continue;
		if(i - b == 2 && km_flg(km[b]) != km_flg(km[b + 1]) && km_dir(km[b]) == km_dir(km[b + 1])){
			const uint64_t kq = km_flg(km[b]) ? km[b + 1] : km[b], kt = km_flg(km[b]) ? km[b] : km[b + 1];
			Hit h; h.qoff = km_off(kq); h.toff = km_off(kt); h.keep = 0;
			hits.push_back(h);
		}
		b = i;
	}
	uint32_t n = (uint32_t)hits.size();
	if(n * ksz < cmin){ hits.clear(); return; }
	std::sort(hits.begin(), hits.end(), [](const Hit &a, const Hit &b){ return a.qoff < b.qoff; });     /* keys are unique */
	/* longest increasing subsequence over the target offsets, with the reference's predecessor rule (bsalign.h:1288-1315) */
	std::vector<uint32_t> tail(n), prev(n);
	tail[0] = 0; prev[0] = NONE;
	uint32_t len = 1;
	for(uint32_t i = 1; i < n; i++){
		const uint32_t tv = hits[i].toff;
		if(tv > hits[tail[len - 1]].toff){
			prev[i] = tail[len - 1];
			tail[len ++] = i;
		} else if(tv <= hits[tail[0]].toff){
			prev[i] = NONE;
			tail[0] = i;
		} else {
			uint32_t b = 0, e = len;
			while(b < e){
				const uint32_t m = b + ((e - b) >> 1);
				if(tv > hits[tail[m]].toff) b = m + 1;
				else if(tv < hits[tail[m]].toff) e = m;
				else { b = m; break; }
			}
			prev[i] = prev[tail[b - 1]];                   /* as written in the reference: the predecessor of that tail, not the tail */
			tail[b] = i;
		}
	}
	{
		uint32_t cov = 0, e = NONE, m = tail[len - 1];
		while(m != NONE){
			Hit &h = hits[m];
			h.keep = 1;
			cov += (h.toff + ksz <= e) ? ksz : e - h.toff;
			e = h.toff;
			m = prev[m];
		}
		if(cov < cmin){ hits.clear(); return; }
	}
	/* drop anchors whose diagonal is far from the mean; repeat until stable (bsalign.h:1347-1392) */
	std::vector<int> diag(n);
	for(;;){
		int tot = 0; uint32_t e = 0;
		for(uint32_t i = 0; i < n; i++){
			if(!hits[i].keep) continue;
			const int d = (int)hits[i].qoff - (int)hits[i].toff;
			tot += d;
			diag[e ++] = d;
		}
		if(e * ksz < cmin) break;
		const int mean = tot / (int)e;
		std::nth_element(diag.begin(), diag.begin() + e / 2, diag.begin() + e);        /* quick_median_array: the element of rank e/2 */
		const int median = diag[e / 2];
		const int var = std::max(std::abs(median - mean) * 3, 50);
		uint32_t dropped = 0;
		for(uint32_t i = 0; i < n; i++){
			if(!hits[i].keep) continue;
			const int d = (int)hits[i].qoff - (int)hits[i].toff;
			if(std::abs(d - mean) > var){ hits[i].keep = 0; dropped ++; }
		}
		if(dropped == 0) break;
	}
	uint32_t w = 0;
	for(uint32_t i = 0; i < n; i++) if(hits[i].keep) hits[w ++] = hits[i];
	hits.resize(w);
	uint32_t cov = 0, e = 0;
	for(uint32_t i = 0; i < w; i++){
		cov += (hits[i].toff >= e + ksz) ? ksz : hits[i].toff + ksz - e;
		e = hits[i].toff + ksz;
	}
	if(cov < cmin) hits.clear();
}

/* the segment list of one pair (bsalign.h:1451-1530): anchor i sits at (qoff + ksz/2, toff + ksz/2); the segment in
 * front of it is aligned, the anchor column itself is emitted as a match in front of the next non-empty segment */
uint32_t segments(uint32_t ksz, const uint64_t *maps, uint32_t kmap, uint32_t qlen, uint32_t tlen, bsa_kmer_seg_t *segs){
	if(ksz > 15) ksz = 15;
	uint32_t n = 0;
	if(kmap == 0){
		bsa_kmer_seg_t s = { 0, qlen, 0, tlen, BSA_MODE_GLOBAL, 0 };
		segs[n ++] = s;
		return n;
	}
	uint32_t qb = 0, tb = 0, ml = 0, mode = BSA_MODE_EXTEND | BSA_KMER_SEG_REVERSED;
	for(uint32_t i = 0; i <= kmap; i++){
		uint32_t qe, te;
		if(i == kmap){ qe = qlen; te = tlen; mode = BSA_MODE_EXTEND; }
		else { qe = (uint32_t)(maps[i] >> 32) + ksz / 2; te = (uint32_t)maps[i] + ksz / 2; ml ++; }
		if(!(qb == qe && tb == te)){
			bsa_kmer_seg_t s = { qb, qe, tb, te, mode, ml };
			segs[n ++] = s;
			ml = 0;
		}
		qb = qe + 1; tb = te + 1;
		mode = BSA_MODE_GLOBAL;
	}
	return n;                                               /* matches still pending in ml are dropped, as in the reference */
}

void push_op(uint32_t *cig, uint64_t &n, uint32_t op, uint32_t sz){      /* _push_cigar_u4v, bsalign.h:401-407 */
	if(n && (cig[n - 1] & 0xFu) == op) cig[n - 1] += sz << 4;
	else cig[n ++] = sz << 4 | op;
}

int assemble(const bsa_kmer_seg_t *segs, uint32_t nseg, const bsa_result_t *rs2, const uint32_t *const *seg_cig, const uint64_t *seg_ncig,
		bsa_result_t *out, uint32_t *cig, uint64_t cap, uint64_t *ncig){
	bsa_result_t R;
	memset(&R, 0, sizeof(R));
	uint64_t n = 0;
	for(uint32_t k = 0; k < nseg; k++){
		const bsa_kmer_seg_t &s = segs[k];
		const bsa_result_t &r = rs2[k];
		if(n + 1 + seg_ncig[k] > cap) return BSA_E_CIGAR_CAP;
		if(s.ml){ push_op(cig, n, 0, s.ml); R.mat += (int32_t)s.ml; R.aln += (int32_t)s.ml; }
		if(seg_ncig[k]) memcpy(cig + n, seg_cig[k], seg_ncig[k] * sizeof(uint32_t));       /* CIGRESV appends without merging (bsalign.h:974-975,1043) */
		n += seg_ncig[k];
		if(s.mode & BSA_KMER_SEG_REVERSED){
			R.qb = (int32_t)s.qe - r.qe; R.tb = (int32_t)s.te - r.te;
			R.qe = (int32_t)s.qe; R.te = (int32_t)s.te;
			std::reverse(cig, cig + n);
		} else {
			R.qe = (int32_t)s.qb + r.qe; R.te = (int32_t)s.tb + r.te;
		}
		R.mat += r.mat; R.mis += r.mis; R.ins += r.ins; R.del += r.del; R.aln += r.aln; R.score += r.score;
	}
	*out = R;
	*ncig = n;
	return BSA_OK;
}

template <typename F> void parallel_for(size_t n, unsigned threads, F body){
	if(threads == 0){
		const char *e = getenv("BSA_KMER_THREADS");
		threads = e ? (unsigned)atoi(e) : std::thread::hardware_concurrency();
		if(threads == 0) threads = 1;
	}
	if(threads > 64) threads = 64;
	if(threads > n) threads = (unsigned)(n ? n : 1);
	if(threads <= 1){ for(size_t k = 0; k < n; k++) body(k); return; }
	std::atomic<size_t> next(0);
	std::atomic<bool> failed(false);           // an exception in a worker (bad_alloc) must not reach std::terminate
	std::vector<std::thread> pool;
	for(unsigned w = 0; w < threads; w++) pool.emplace_back([&](){
		try {
			for(;;){
				const size_t b = next.fetch_add(64);
				if(b >= n || failed.load()) break;
				const size_t e = std::min(n, b + 64);
				for(size_t k = b; k < e; k++) body(k);
			}
		} catch(...){ failed.store(true); }
	});
	for(auto &th : pool) th.join();
	if(failed.load()) throw std::bad_alloc();  // re-raised on the calling thread, mapped to BSA_E_NOMEM by the entry points
}

} // namespace

extern "C" uint32_t bsa_kmer_chain(uint32_t ksz, const uint8_t *q, uint32_t qlen, const uint8_t *t, uint32_t tlen, uint64_t *maps, uint32_t cap){
	std::vector<Hit> hits;
	chain(ksz, q, qlen, t, tlen, hits);
	if(hits.size() > cap) return NONE;
	for(size_t i = 0; i < hits.size(); i++) maps[i] = ((uint64_t)hits[i].qoff << 32) | hits[i].toff;
	return (uint32_t)hits.size();
}

extern "C" uint32_t bsa_kmer_segments(uint32_t ksz, const uint64_t *maps, uint32_t kmap, uint32_t qlen, uint32_t tlen, bsa_kmer_seg_t *segs){
	return segments(ksz, maps, kmap, qlen, tlen, segs);
}

extern "C" int bsa_kmer_assemble(const bsa_kmer_seg_t *segs, uint32_t nseg, const bsa_result_t *seg_out, const uint32_t *seg_cigar,
		const uint64_t *seg_cigar_off, bsa_result_t *out, uint32_t *cigar, uint64_t cigar_cap_words, uint64_t *cigar_words){
	if(!segs || !seg_out || !out || !cigar || !cigar_words || (nseg && !seg_cigar_off)) return BSA_E_ARG;
	std::vector<const uint32_t*> ptr(nseg); std::vector<uint64_t> cnt(nseg);
	for(uint32_t k = 0; k < nseg; k++){ ptr[k] = seg_cigar + seg_cigar_off[k]; cnt[k] = seg_cigar_off[k + 1] - seg_cigar_off[k]; }
	return assemble(segs, nseg, seg_out, ptr.data(), cnt.data(), out, cigar, cigar_cap_words, cigar_words);
}

static int kmer_edit_batch_impl(bsa_ctx_t *ctx, const uint8_t *seqs, size_t seqs_bytes,
		const uint64_t *qoff, const uint32_t *qlen, const uint64_t *toff, const uint32_t *tlen, size_t n,
		const bsa_kmer_params_t *par, bsa_result_t *out, uint32_t *cigar, size_t cigar_cap_words, uint64_t *cigar_off, uint32_t *status);

extern "C" int bsa_kmer_edit_batch(bsa_ctx_t *ctx, const uint8_t *seqs, size_t seqs_bytes,
		const uint64_t *qoff, const uint32_t *qlen, const uint64_t *toff, const uint32_t *tlen, size_t n,
		const bsa_kmer_params_t *par, bsa_result_t *out, uint32_t *cigar, size_t cigar_cap_words, uint64_t *cigar_off, uint32_t *status){
	try {
		return kmer_edit_batch_impl(ctx, seqs, seqs_bytes, qoff, qlen, toff, tlen, n, par, out, cigar, cigar_cap_words, cigar_off, status);
	} catch(...){                               // host allocations (vectors, worker threads): no exception crosses the C ABI
		return BSA_E_NOMEM;
	}
}

static int kmer_edit_batch_impl(bsa_ctx_t *ctx, const uint8_t *seqs, size_t seqs_bytes,
		const uint64_t *qoff, const uint32_t *qlen, const uint64_t *toff, const uint32_t *tlen, size_t n,
		const bsa_kmer_params_t *par, bsa_result_t *out, uint32_t *cigar, size_t cigar_cap_words, uint64_t *cigar_off, uint32_t *status){
	if(!ctx || !par || !out || (n && (!seqs || !qoff || !qlen || !toff || !tlen))) return BSA_E_ARG;
	if(par->ksz == 0) return BSA_E_ARG;
	const uint32_t ksz = par->ksz > 15 ? 15 : par->ksz;
	for(size_t k = 0; k < n; k++)
		if(qoff[k] + qlen[k] > seqs_bytes || toff[k] + tlen[k] > seqs_bytes) return BSA_E_ARG;
	const bool timing = getenv("BSA_KMER_TIMING") != nullptr;          /* phase times on stderr */
	auto now = [](){ return std::chrono::steady_clock::now(); };
	auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b){ return std::chrono::duration<double, std::milli>(b - a).count(); };
	auto t0 = now();
	/* 1. chains and segment lists, threads over pairs */
	std::vector<std::vector<bsa_kmer_seg_t>> segs(n);
	std::vector<uint32_t> flag(n, 0);                      // a base code above 3 on an anchor column never reaches the device: look here
	parallel_for(n, par->threads, [&](size_t k){
		uint8_t any = 0;
		for(uint32_t i = 0; i < qlen[k]; i++) any |= seqs[qoff[k] + i];
		for(uint32_t i = 0; i < tlen[k]; i++) any |= seqs[toff[k] + i];
		if(any > 3) flag[k] = BSA_ST_BAD_BASE;
		std::vector<Hit> hits;
		chain(ksz, seqs + qoff[k], qlen[k], seqs + toff[k], tlen[k], hits);
		std::vector<uint64_t> maps(hits.size());
		for(size_t i = 0; i < hits.size(); i++) maps[i] = ((uint64_t)hits[i].qoff << 32) | hits[i].toff;
		segs[k].resize(hits.size() + 1);
		segs[k].resize(segments(ksz, maps.data(), (uint32_t)maps.size(), qlen[k], tlen[k], segs[k].data()));
	});
	auto t1 = now();
	/* 2. two device batches: EXTEND for the reversed heads and the tails (copied into one blob), GLOBAL for the gaps
	 *    (views into the caller's blob) */
	struct Job { std::vector<uint64_t> qo, to; std::vector<uint32_t> ql, tl; std::vector<bsa_result_t> rs; std::unique_ptr<uint32_t[]> cig; std::vector<uint64_t> coff; std::vector<uint32_t> st; size_t cap = 8; };
	Job job[2];
	std::vector<uint8_t> heads;
	std::vector<size_t> seg_base(n + 1, 0);
	for(size_t k = 0; k < n; k++) seg_base[k + 1] = seg_base[k] + segs[k].size();
	std::vector<uint8_t> seg_job(seg_base[n], 0xFF); std::vector<uint32_t> seg_idx(seg_base[n], 0);
	{
		/* where every pair's segments go: counts per pair, prefix sums, then all pairs fill their own ranges in parallel */
		std::vector<size_t> cnt0(n + 1, 0), cnt1(n + 1, 0), hb(n + 1, 0);
		parallel_for(n, par->threads, [&](size_t k){
			size_t c0 = 0, c1 = 0, h = 0;
			for(const bsa_kmer_seg_t &s : segs[k]){
				const uint32_t ql = s.qe - s.qb, tl = s.te - s.tb;
				if(ql == 0 || tl == 0) continue;              /* an empty side gives the all-zero result (bsalign.h:1051-1054) */
				if((s.mode & 3) == BSA_MODE_GLOBAL) c1 ++; else { c0 ++; h += (size_t)ql + tl; }
			}
			cnt0[k + 1] = c0; cnt1[k + 1] = c1; hb[k + 1] = h;
		});
		for(size_t k = 0; k < n; k++){ cnt0[k + 1] += cnt0[k]; cnt1[k + 1] += cnt1[k]; hb[k + 1] += hb[k]; }
		if(cnt0[n] > 0xFFFFFFF0ull || cnt1[n] > 0xFFFFFFF0ull) return BSA_E_ARG;
		heads.resize(hb[n]);
		for(int w = 0; w < 2; w++){
			const size_t m = w ? cnt1[n] : cnt0[n];
			job[w].qo.resize(m); job[w].to.resize(m); job[w].ql.resize(m); job[w].tl.resize(m);
		}
		std::vector<size_t> caps0(n, 0), caps1(n, 0);
		parallel_for(n, par->threads, [&](size_t k){
			size_t i0 = cnt0[k], i1 = cnt1[k], base = hb[k], cap0 = 0, cap1 = 0;
			for(size_t j = 0; j < segs[k].size(); j++){
				const bsa_kmer_seg_t &s = segs[k][j];
				const uint32_t ql = s.qe - s.qb, tl = s.te - s.tb;
				if(ql == 0 || tl == 0) continue;
				const int w = ((s.mode & 3) == BSA_MODE_GLOBAL) ? 1 : 0;
				Job &J = job[w];
				size_t &ix = w ? i1 : i0;
				if(w == 0){
					if(s.mode & BSA_KMER_SEG_REVERSED){
						std::reverse_copy(seqs + qoff[k], seqs + qoff[k] + s.qe, heads.begin() + base);
						std::reverse_copy(seqs + toff[k], seqs + toff[k] + s.te, heads.begin() + base + ql);
					} else {
						memcpy(heads.data() + base, seqs + qoff[k] + s.qb, ql);
						memcpy(heads.data() + base + ql, seqs + toff[k] + s.tb, tl);
					}
					J.qo[ix] = base; J.to[ix] = base + ql;
					base += (size_t)ql + tl;
					cap0 += (size_t)ql + tl + 2;
				} else {
					J.qo[ix] = qoff[k] + s.qb; J.to[ix] = toff[k] + s.tb;
					cap1 += (size_t)ql + tl + 2;
				}
				J.ql[ix] = ql; J.tl[ix] = tl;
				seg_job[seg_base[k] + j] = (uint8_t)w;
				seg_idx[seg_base[k] + j] = (uint32_t)ix;
				ix ++;
			}
			caps0[k] = cap0; caps1[k] = cap1;
		});
		for(size_t k = 0; k < n; k++){ job[0].cap += caps0[k]; job[1].cap += caps1[k]; }
	}
	auto t2 = now();
	for(int w = 0; w < 2; w++){
		Job &J = job[w];
		const size_t m = J.ql.size();
		if(m == 0) continue;
		J.rs.resize(m); J.cig.reset(new uint32_t[J.cap]); J.coff.resize(m + 1); J.st.assign(m, 0);       // the arena is not zero-filled
		bsa_edit_params_t ep;
		ep.mode = (w == 1) ? BSA_MODE_GLOBAL : BSA_MODE_EXTEND;
		ep.bandwidth = 0;
		const uint8_t *blob = (w == 0) ? heads.data() : seqs;
		const size_t bytes = (w == 0) ? heads.size() : seqs_bytes;
		const int rc = bsa_edit_batch(ctx, blob, bytes, J.qo.data(), J.ql.data(), J.to.data(), J.tl.data(), m, &ep,
			J.rs.data(), J.cig.get(), J.cap, J.coff.data(), J.st.data());
		if(rc != BSA_OK) return rc;
		if(timing){
			uint64_t sq = 0, st = 0; uint32_t mq = 0, mt = 0;
			for(size_t x = 0; x < m; x++){ sq += J.ql[x]; st += J.tl[x]; mq = std::max(mq, J.ql[x]); mt = std::max(mt, J.tl[x]); }
			fprintf(stderr, "[bsa_kmer]   batch %d: %zu segments, query %llu bases (max %u), target %llu (max %u), done at %.1f ms\n", w, m,
				(unsigned long long)sq, mq, (unsigned long long)st, mt, ms(t2, now()));
		}
	}
	auto t3 = now();
	/* 3. stitch per pair */
	std::vector<uint64_t> need(n + 1, 0);
	for(size_t k = 0; k < n; k++){
		uint64_t w = 0;
		for(size_t j = 0; j < segs[k].size(); j++){
			const size_t g = seg_base[k] + j;
			w += 1;
			if(seg_job[g] != 0xFF){ const Job &J = job[seg_job[g]]; w += J.coff[seg_idx[g] + 1] - J.coff[seg_idx[g]]; }
		}
		need[k + 1] = need[k] + w;
	}
	const bool want_cig = cigar != nullptr && cigar_off != nullptr;
	std::vector<uint32_t> scratch;
	uint32_t *work = cigar;
	if(!want_cig || need[n] > cigar_cap_words){
		if(want_cig){ cigar_off[n] = need[n]; return BSA_E_CIGAR_CAP; }      // the words a retry needs (bsalign_hip.h)
		scratch.resize(need[n] + 1); work = scratch.data();
	}
	std::vector<uint64_t> used(n, 0);
	std::atomic<int> bad(BSA_OK);
	parallel_for(n, par->threads, [&](size_t k){
		const size_t ns = segs[k].size();
		std::vector<bsa_result_t> rs(ns); std::vector<const uint32_t*> ptr(ns); std::vector<uint64_t> cnt(ns);
		uint32_t st = flag[k];
		for(size_t j = 0; j < ns; j++){
			const size_t g = seg_base[k] + j;
			if(seg_job[g] == 0xFF){ memset(&rs[j], 0, sizeof(bsa_result_t)); ptr[j] = nullptr; cnt[j] = 0; continue; }
			const Job &J = job[seg_job[g]];
			const uint32_t x = seg_idx[g];
			rs[j] = J.rs[x]; ptr[j] = J.cig.get() + J.coff[x]; cnt[j] = J.coff[x + 1] - J.coff[x];
			st |= J.st[x];
		}
		const int rc = assemble(segs[k].data(), (uint32_t)ns, rs.data(), ptr.data(), cnt.data(), &out[k], work + need[k], need[k + 1] - need[k], &used[k]);
		if(rc != BSA_OK) bad = rc;
		if(status) status[k] = st;
	});
	if(bad != BSA_OK) return bad;
	if(timing) fprintf(stderr, "[bsa_kmer] %zu pairs: chain %.1f ms, pack %.1f ms (%zu heads and tails, %zu gaps), device %.1f ms, stitch %.1f ms\n",
		n, ms(t0, t1), ms(t1, t2), job[0].ql.size(), job[1].ql.size(), ms(t2, t3), ms(t3, now()));
	if(want_cig){
		/* close the gaps left by merged match runs so that pair k owns cigar[cigar_off[k] .. cigar_off[k+1]) */
		uint64_t w = 0;
		for(size_t k = 0; k < n; k++){
			cigar_off[k] = w;
			if(w != need[k]) memmove(cigar + w, cigar + need[k], used[k] * sizeof(uint32_t));
			w += used[k];
		}
		cigar_off[n] = w;
	} else if(cigar_off){
		uint64_t w = 0;
		for(size_t k = 0; k < n; k++){ cigar_off[k] = w; w += used[k]; }
		cigar_off[n] = w;
	}
	return BSA_OK;
}
