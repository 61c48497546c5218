// bsa_msa.cpp -- host-only: the POA's binary and text MSA formats on plain arrays (include/bsalign_msa.h).
// Byte-identical to the reference's writers (dump_binary_msa_bspoa bspoa.h:1555-1586, print_msa_bspoa :1491-1553 with
// the row builders :1329-1483) and accepted by / accepting what its loader does (:1588-1685); checked against the real
// functions through oracle/_ref (tests/test_msa_formats_cpu.py) and against a committed fixture.
#include "../../include/bsalign_msa.h"
#include "../../include/bsalign_hip.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

// an append-only sink that keeps counting when the caller's buffer is full
struct Sink {
	uint8_t *out; size_t cap, n;
	void put(const void *p, size_t len){
		if(out && n + len <= cap) memcpy(out + n, p, len);
		n += len;
	}
	void u8(uint8_t v){ put(&v, 1); }
	void u32(uint32_t v){ put(&v, 4); }          // host order, as fwrite(&v, 4, 1, out) leaves it
};

inline const uint8_t *column(const uint8_t *cols, const uint32_t *idxs, uint32_t mrow, uint32_t i){
	return cols + (size_t)(idxs ? idxs[i] : i) * mrow;
}

const uint8_t TAG_META = 0x81, TAG_MSA = 0x22, TAG_END = 0xFF;

}

extern "C" int bsa_msa_binary_write(const uint8_t *cols, const uint32_t *idxs, uint32_t nseq, uint32_t mlen,
		const char *meta, uint32_t metalen, uint8_t *out, size_t cap, size_t *need){
	if((mlen && !cols) || !need) return BSA_E_ARG;
	const uint32_t mrow = nseq + 3;
	Sink s{out, cap, 0};
	if(meta && metalen){ s.u8(TAG_META); s.u32(metalen); s.put(meta, metalen); }
	s.u8(TAG_MSA); s.u32(mlen); s.u32(nseq);
	for(uint32_t i = 0; i < mlen; i++) s.put(column(cols, idxs, mrow, i), (size_t)nseq + 1);          // reads + consensus
	for(uint32_t plane = 1; plane <= 2; plane++)                                                        // then the two quality planes
		for(uint32_t i = 0; i < mlen; i++) s.u8(column(cols, idxs, mrow, i)[nseq + plane]);
	s.u8(TAG_END);
	*need = s.n;
	return (out && s.n <= cap) ? BSA_OK : BSA_E_NOMEM;
}

extern "C" int bsa_msa_binary_read(const uint8_t *in, size_t len, size_t *consumed, uint32_t *nseq_out, uint32_t *mlen_out,
		uint8_t *cols, size_t cols_cap, char *meta, size_t meta_cap, uint32_t *metalen_out){
	if(!in && len) return BSA_E_ARG;
	size_t p = 0;
	uint32_t nseq = 0, mlen = 0, metalen = 0;
	bool small = false, ended = false;
	auto rd32 = [&](uint32_t &v) -> bool { if(len - p < 4) return false; memcpy(&v, in + p, 4); p += 4; return true; };
	while(p < len){
		const uint8_t tag = in[p++];
		if(tag == TAG_END){ ended = true; break; }
		if(tag == TAG_META){
			uint32_t dl;
			if(!rd32(dl) || len - p < dl) return BSA_E_ARG;
			metalen = dl;
			if(meta){ if(dl <= meta_cap) memcpy(meta, in + p, dl); else small = true; }
			p += dl;
		} else if(tag == TAG_MSA){
			if(!rd32(mlen) || !rd32(nseq)) return BSA_E_ARG;
			const size_t mrow = (size_t)nseq + 3, body = (size_t)mlen * (nseq + 1), q = (size_t)mlen * 2;
			if(len - p < body || len - p - body < q) return BSA_E_ARG;
			if(cols){
				if((size_t)mlen * mrow > cols_cap) small = true;
				else {
					for(uint32_t i = 0; i < mlen; i++){
						uint8_t *c = cols + (size_t)i * mrow;
						memcpy(c, in + p + (size_t)i * (nseq + 1), (size_t)nseq + 1);
						c[nseq + 1] = in[p + body + i];
						c[nseq + 2] = in[p + body + mlen + i];
					}
				}
			}
			p += body + q;
		}
		// (the reference's loader steps over any other byte the same way)
	}
	if(!ended) return BSA_E_ARG;
	if(consumed) *consumed = p;
	if(nseq_out) *nseq_out = nseq;
	if(mlen_out) *mlen_out = mlen;
	if(metalen_out) *metalen_out = metalen;
	return small ? BSA_E_NOMEM : BSA_OK;
}

extern "C" int bsa_msa_consensus(const uint8_t *cols, const uint32_t *idxs, uint32_t nseq, uint32_t mlen,
		uint8_t *cns, uint8_t *qlt, uint8_t *alt, uint32_t *clen, uint8_t *rdseqs, uint64_t *rdoffs){
	if(mlen && !cols) return BSA_E_ARG;
	const uint32_t mrow = nseq + 3;
	uint32_t n = 0;
	for(uint32_t i = 0; i < mlen; i++){
		const uint8_t *c = column(cols, idxs, mrow, i);
		if(c[nseq] < 4){
			if(cns) cns[n] = c[nseq];
			if(qlt) qlt[n] = c[nseq + 1];
			if(alt) alt[n] = c[nseq + 2];
			n++;
		}
	}
	if(clen) *clen = n;
	if(rdoffs){
		uint64_t off = 0;
		for(uint32_t r = 0; r < nseq; r++){
			rdoffs[r] = off;
			for(uint32_t i = 0; i < mlen; i++){
				const uint8_t b = column(cols, idxs, mrow, i)[r];
				if(b < 4){ if(rdseqs) rdseqs[off] = b; off++; }
			}
		}
		rdoffs[nseq] = off;
	}
	return BSA_OK;
}

namespace {

// column ruler: "|%05u" at every tenth MSA column that still has six characters of room, '~' over variant columns
std::string msa_ruler(uint32_t beg, uint32_t end, const uint32_t *var_mpos, uint32_t nvar){
	std::string s;
	s.reserve(end - beg + 8);
	uint32_t filled = beg;                       // first column not yet covered by a label or a blank
	char lab[16];
	for(uint32_t i = beg; i < end; i++){
		if(i % 10 == 0 && filled + 6 <= end){ snprintf(lab, sizeof lab, "|%05u", i); s += lab; filled += 6; }
		else if(i >= filled){ s += ' '; filled++; }
	}
	for(uint32_t k = 0; k < nvar; k++){
		if(var_mpos[k] >= end) break;
		if(var_mpos[k] >= beg && var_mpos[k] - beg < s.size()) s[var_mpos[k] - beg] = '~';
	}
	return s;
}

// consensus ruler: the same labels, counted in consensus bases; NOTE the reference looks the consensus byte up in the
// column STORED at position i, not the i-th column of the MSA order (bspoa.h:1339: buffer[i * mrow + nseq]) -- kept
std::string cns_ruler(const uint8_t *cols, uint32_t nseq, uint32_t beg, uint32_t end, uint32_t cbeg){
	const uint32_t mrow = nseq + 3;
	std::string s;
	uint32_t j = cbeg, b = beg;
	char lab[16];
	for(uint32_t i = beg; i < end; i++){
		if(cols[(size_t)i * mrow + nseq] >= 4) continue;
		if(j % 10 == 0){
			while(b < i){ s += ' '; b++; }
			if(b + 6 < end){ snprintf(lab, sizeof lab, "|%05u", j); s += lab; b += 6; }
		}
		j++;
	}
	while(b < end){ s += ' '; b++; }
	return s;
}

}

extern "C" int bsa_msa_text(const uint8_t *cols, const uint32_t *idxs, uint32_t nseq, uint32_t mlen,
		const uint8_t *cns, const uint8_t *qlt, const uint8_t *alt, const uint32_t *var_mpos, uint32_t nvar,
		const char *label, uint32_t mbeg, uint32_t mend, uint32_t linewidth, char *out, size_t cap, size_t *need){
	if((mlen && !cols) || !need || !label) return BSA_E_ARG;
	const uint32_t mrow = nseq + 3;
	if(mend == 0 || mend > mlen) mend = mlen;
	if(mbeg > mend) return BSA_E_ARG;
	if(linewidth == 0 || linewidth > mend - mbeg) linewidth = mend - mbeg;
	Sink s{(uint8_t*)out, cap, 0};
	auto emit = [&](const std::string &t){ s.put(t.data(), t.size()); };
	const std::string lab(label);
	// bases already consumed by every row (reads 0 .. nseq-1, consensus = row nseq) in front of mbeg
	std::vector<uint32_t> roffs((size_t)nseq + 1, 0);
	for(uint32_t i = 0; i < mbeg; i++){
		const uint8_t *c = column(cols, idxs, mrow, i);
		for(uint32_t r = 0; r <= nseq; r++) roffs[r] += c[r] < 4;
	}
	char num[64];
	std::string line;
	for(uint32_t beg = mbeg, end; beg < mend; beg = end){
		end = (mend - beg < linewidth) ? mend : beg + linewidth;
		emit(lab + " MSA [POS] " + msa_ruler(beg, end, var_mpos, var_mpos ? nvar : 0) + "\n");
		const uint32_t cbeg = roffs[nseq];
		for(uint32_t r = 0; r < mrow; r++){
			line = lab + " MSA ";
			if(r <= nseq){
				if(r == nseq) line += "[CNS] ";
				else { snprintf(num, sizeof num, "[%03u] ", r); line += num; }
				uint32_t rend = roffs[r];
				for(uint32_t i = beg; i < end; i++){
					const uint8_t *c = column(cols, idxs, mrow, i);
					const uint8_t v = c[r];
					// a base or gap that disagrees with the consensus is printed in lower case; 5, 6 (outside the read) as '.', '*'
					line += (v <= 4 && v != c[nseq]) ? "acgt-.*"[v] : "ACGT-.*"[v < 7 ? v : 6];
					rend += v < 4;
				}
				snprintf(num, sizeof num, " %d\t%d\n", (int)roffs[r], (int)rend);
				line += num;
				roffs[r] = rend;
			} else {
				line += (r == nseq + 1) ? "[QLT] " : "[ALT] ";
				for(uint32_t i = beg; i < end; i++) line += (char)('!' + column(cols, idxs, mrow, i)[r]);
				line += "\n";
			}
			emit(line);
		}
		emit(lab + " MSA [POS] " + cns_ruler(cols, nseq, beg, end, cbeg) + "\n");
		// the window's consensus bases and their two quality strings
		const uint32_t cend = roffs[nseq], cn = cend - cbeg;
		static const char *names[3] = {" CNS\t", " QLT\t", " ALT\t"};
		const uint8_t *src[3] = {cns, qlt, alt};
		for(int k = 0; k < 3; k++){
			snprintf(num, sizeof num, "%d\t", (int)cn);
			line = lab + names[k] + num;
			for(uint32_t i = cbeg; i < cend; i++) line += src[k] ? (k == 0 ? "ACGTN-acgtn*"[src[k][i] < 12 ? src[k][i] : 4] : (char)('!' + src[k][i])) : '?';
			line += "\n";
			emit(line);
		}
	}
	*need = s.n;
	return (out && s.n <= cap) ? BSA_OK : BSA_E_NOMEM;
}
