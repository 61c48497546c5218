// bsa_cns.cpp -- consensus calling over the columns of a window's MSA (include/bsalign_msa.h: bsa_msa_call_consensus).
//
// The model is the reference's (cns_bspoa, bspoa.h:3457-3733, with the alignment events of bspoa.h:142-204 and the log-sum of
// :3413-3453): a left-to-right automaton over the MSA columns with five states (the consensus shows A, C, G, T or nothing in the
// column); entering state a from state e is worth the sum, over the reads present in the column, of the log probability of what
// each read does there (agrees / differs, starts or continues an insertion or a deletion, slips in a homopolymer).  What a read
// does depends on its symbol b, on the last base c the consensus emitted on the path into e, and on the read's own previous
// event d on that path -- so every state drags one event per read along.
//
// Organisation here (not the reference's): the model is tabulated once as cost[c][d][b][.] -- five doubles, one per target state,
// side by side -- so that a column is five sweeps over the present reads (one per source state) in each of which five running sums
// advance together; a source state that is dead costs nothing.  The automaton's memory is three flat arrays (mass, origin, anchor)
// of five entries a column.  Only the ORDER of the floating-point additions is the reference's, because the result has to be
// its bytes: every sum runs over the reads in row order starting from zero, the five arrivals of a state are merged in state
// order with the same two cut-offs (a term 40 below the running value is dropped), and the quality formulas keep their shape.
#include "../../include/bsalign_msa.h"
#include "../../include/bsalign_hip.h"
#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr double kDead = -1000000000.0;        // log probability of the impossible (bspoa.h:86)
constexpr double kPhredCap = 90.0;             // bspoa.h:87
constexpr int kGap = 4;                        // the fifth state / symbol: nothing in this column
constexpr uint8_t kAbsent = 4;                 // event code of a read that was not in the previous column
enum Event : uint8_t { kAligned = 0, kInserted = 1, kDeleted = 2 };

// log(exp(acc) + exp(v)), with the cut-offs of bspoa.h:3421-3447: the smaller term is ignored from 40 down
inline double merge_log(double acc, double v){
	if(v == kDead) return acc;
	const double top = (v > acc) ? v : acc, low = (v > acc) ? acc : v;
	if(top >= low + 40) return top;
	return top + std::log(1 + std::exp(low - top));
}
inline double merge_log5(const double *v){
	double acc = kDead;
	for(int i = 0; i < 5; i++) acc = merge_log(acc, v[i]);
	return acc;
}

struct StepModel {
	double cost[5][5][5][5];       // [anchor c][read's last event d][read symbol b][target state a]
	uint8_t next[5][5][5][5];      // the read's event after the step
	explicit StepModel(const bsa_cns_params_t &p){
		// natural logs of: agree, differ, open insertion, open deletion, extend insertion, extend deletion, homopolymer insertion / deletion
		const double lg[8] = { std::log((double)(1 - p.psub)), std::log((double)p.psub), std::log((double)p.pins), std::log((double)p.pdel),
		                       std::log((double)p.piex), std::log((double)p.pdex), std::log((double)p.hins), std::log((double)p.hdel) };
		for(int c = 0; c < 5; c++) for(int d = 0; d < 5; d++) for(int b = 0; b < 5; b++) for(int a = 0; a < 5; a++){
			int which; uint8_t ev;
			if(a < kGap && b < kGap){ which = (a == b) ? 0 : 1; ev = kAligned; }
			else if(a < kGap){                                   // the consensus has a base the read lacks
				which = (d == kDeleted) ? 5 : 3;
				if(a == c && lg[7] > lg[which]) which = 7;       // ... the same base again: a homopolymer slip if that is likelier
				ev = kDeleted;
			} else if(b < kGap){                                 // the read has a base the consensus lacks
				which = (d == kInserted) ? 4 : 2;
				if(b == c && lg[6] > lg[which]) which = 6;
				ev = kInserted;
			} else { which = 0; ev = (uint8_t)d; }               // neither has anything: the read's event stands, at the price of agreeing
			cost[c][d][b][a] = lg[which];
			next[c][d][b][a] = ev;
		}
	}
};

// log(k!) by running summation (the reference caches the same partial sums, bspoa.h:3391-3401)
struct LogFactorial {
	std::vector<double> v{0.0};
	double operator()(uint32_t k){
		while(v.size() <= k) v.push_back(v.back() + std::log((double)(uint32_t)v.size()));
		return v[k];
	}
};

inline uint8_t phred_byte(double q){ return (uint8_t)(int)(q < kPhredCap ? q : kPhredCap); }

// probability that fewer than k of n reads show an error of rate p: normal tail for large n, else the binomial sum
double fewer_errors(uint32_t n, uint32_t k, double p, LogFactorial &lf){
	if(n > 50 && n * p > 5 && n * (1 - p) > 5)
		return std::erfc(-((k - n * p) / std::sqrt(n * p * (1 - p))) / 1.4142135623731) / 2;
	double tail = 0;
	for(uint32_t e = 0; e < k; e++){
		const double ways = (n <= 1000) ? lf(n) - lf(e) - lf(n - e) : 1;          // (no cache beyond 1000 reads there: the term is taken as 1)
		tail += std::exp(std::log(p) * e + std::log(1 - p) * (n - e) + ways);
	}
	return tail;
}

}  // namespace

// The model's tables and constants for the device form (bsa_cns_dev.hip): the logarithms are taken HERE, by the host's libm, so that the
// sums the device forms from them are the host's bit for bit.  cost / next as in StepModel; lf[k] = log(k!) for k <= nlf - 1;
// consts = { ln 10, log(psub), log(1 - psub), (double)psub }.
extern "C" void bsa_cns_model_internal(const bsa_cns_params_t *par, double *cost, uint8_t *next, double *lf, uint32_t nlf, double *consts){
	const StepModel model(*par);
	memcpy(cost, model.cost, sizeof(model.cost));
	memcpy(next, model.next, sizeof(model.next));
	LogFactorial f;
	for(uint32_t k = 0; k < nlf; k++) lf[k] = f(k);
	const double psub = par->psub;
	consts[0] = std::log(10.0); consts[1] = std::log(psub); consts[2] = std::log(1 - psub); consts[3] = psub;
}

extern "C" int bsa_msa_call_consensus(uint8_t *cols, const uint32_t *idxs, uint32_t nall, uint32_t nseq, uint32_t nmax, uint32_t mlen,
		const bsa_cns_params_t *par, uint8_t *cns, uint8_t *qlt, uint8_t *alt, uint32_t *clen, double *score){
	if(!par || (mlen && !cols) || nseq > nall || nmax > nall) return BSA_E_ARG;
	if(clen) *clen = 0;
	if(score) *score = 0;
	if(mlen == 0) return BSA_OK;
	const size_t stride = (size_t)nall + 3;                     // a column: nall read symbols, then consensus, quality, alternative quality
	auto column = [&](uint32_t pos) -> uint8_t* { return cols + (size_t)(idxs ? idxs[pos] : pos) * stride; };
	const StepModel model(*par);

	// the automaton's memory, column-major: mass = log probability of all paths ending in the state, origin = the best source
	// state, anchor = last base on that best path.  Slot 0 is the start: only the empty state lives.
	std::vector<double> mass((size_t)(mlen + 1) * 5, kDead);
	std::vector<uint8_t> origin((size_t)(mlen + 1) * 5, (uint8_t)kGap), anchor((size_t)(mlen + 1) * 5, (uint8_t)kGap);
	mass[kGap] = 0;
	// one event per read and state, for the column behind (ev) and the one being built (ev2)
	std::vector<uint8_t> evbuf((size_t)nseq * 10, (uint8_t)kAligned);
	uint8_t *ev = evbuf.data(), *ev2 = evbuf.data() + (size_t)nseq * 5;
	std::vector<uint32_t> who; who.reserve(nseq);               // the reads present in the column, in row order
	std::vector<uint8_t> sym; sym.reserve(nseq);

	for(uint32_t pos = 0; pos < mlen; pos++){
		const uint8_t *colp = column(pos);
		const double *m0 = &mass[(size_t)pos * 5]; double *m1 = &mass[(size_t)(pos + 1) * 5];
		const uint8_t *a0 = &anchor[(size_t)pos * 5]; uint8_t *a1 = &anchor[(size_t)(pos + 1) * 5], *o1 = &origin[(size_t)(pos + 1) * 5];
		who.clear(); sym.clear();
		uint32_t votes[5] = {0, 0, 0, 0, 0};
		for(uint32_t r = 0; r < nseq; r++) if(colp[r] <= kGap){ who.push_back(r); sym.push_back(colp[r]); votes[colp[r]]++; }
		const uint32_t present = (uint32_t)who.size();
		const uint32_t quorum = (uint32_t)(0.1 * present);      // a state fewer than a tenth of the reads show is not considered
		// arrivals: arrive[e][a] = (what the reads say about e -> a) + mass of e
		double arrive[5][5];
		for(int e = 0; e < 5; e++){
			if(m0[e] == kDead){ for(int a = 0; a < 5; a++) arrive[e][a] = kDead; continue; }
			double run[5] = {0, 0, 0, 0, 0};
			const uint8_t *eve = ev + (size_t)e * nseq;
			const double (*tab)[5][5] = model.cost[a0[e]];
			for(uint32_t k = 0; k < present; k++){
				const double *row = tab[eve[who[k]]][sym[k]];
				run[0] += row[0]; run[1] += row[1]; run[2] += row[2]; run[3] += row[3]; run[4] += row[4];
			}
			for(int a = 0; a < 5; a++) arrive[e][a] = run[a] + m0[e];
		}
		for(int a = 0; a < 5; a++){
			uint8_t *eva = ev2 + (size_t)a * nseq;
			if(present && votes[a] < std::max(quorum, 1u)){       // unsupported
				m1[a] = kDead; o1[a] = (uint8_t)kGap; a1[a] = (uint8_t)kGap;
				memset(eva, kAligned, nseq);
				continue;
			}
			const double in[5] = {arrive[0][a], arrive[1][a], arrive[2][a], arrive[3][a], arrive[4][a]};
			m1[a] = merge_log5(in);
			int best = kGap;                                     // the empty state wins ties, then the lower base
			for(int e = 0; e < kGap; e++) if(in[e] > in[best]) best = e;
			o1[a] = (uint8_t)best;
			a1[a] = (a < kGap) ? (uint8_t)a : a0[best];
			memset(eva, kAbsent, nseq);
			const uint8_t *evb = ev + (size_t)best * nseq;
			const uint8_t (*nx)[5][5] = model.next[a0[best]];
			for(uint32_t k = 0; k < present; k++) eva[who[k]] = nx[evb[who[k]]][sym[k]][a];
		}
		std::swap(ev, ev2);
	}

	// the best final state and the chain of origins behind it
	int state = kGap;
	{
		const double *mf = &mass[(size_t)mlen * 5];
		for(int a = 0; a < kGap; a++) if(mf[a] > mf[state]) state = a;
		if(score) *score = mf[state];
	}
	for(uint32_t pos = mlen; pos-- > 0; ){
		column(pos)[nall] = (uint8_t)state;
		state = origin[(size_t)(pos + 1) * 5 + state];
	}

	// qualities: phred of the posterior of the called state among the five, and phred of seeing the runner-up allele that often by error
	const double ln10 = std::log(10.0), psub = par->psub;
	LogFactorial lf;
	uint32_t nc = 0;
	for(uint32_t pos = 0; pos < mlen; pos++){
		uint8_t *colp = column(pos);
		const uint32_t called = colp[nall];
		const double *m1 = &mass[(size_t)(pos + 1) * 5];
		const double all = merge_log5(m1);
		const double others = std::log(1 - std::exp(m1[called] - all));
		colp[nall + 1] = phred_byte(-(10 * (others) / ln10));
		uint32_t votes[5] = {0, 0, 0, 0, 0}, total = 0;
		for(uint32_t r = 0; r < nmax; r++) if(colp[r] <= kGap){ votes[colp[r]]++; total++; }
		uint32_t rival = (called + 1) % 5;
		for(uint32_t e = 0; e < 5; e++) if(e != called && votes[e] > votes[rival]) rival = e;
		const double by_error = fewer_errors(total, votes[rival], psub, lf);
		const double q2 = (by_error == 0) ? 0 : -(10 * std::log(1 - by_error) / ln10);
		colp[nall + 2] = phred_byte(q2);
		if(called < (uint32_t)kGap){
			if(cns) cns[nc] = colp[nall];
			if(qlt) qlt[nc] = colp[nall + 1];
			if(alt) alt[nc] = colp[nall + 2];
			nc++;
		}
	}
	if(clen) *clen = nc;
	return BSA_OK;
}
