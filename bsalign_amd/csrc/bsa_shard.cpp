// bsa_shard.cpp -- host side of the batch scatter: packing the pairs of one rank's contiguous
// range into the blob that travels to that rank.  Plain C ABI, no device code; the exchange itself is torch.distributed /
// RCCL in bsalign_amd/shard.py (or the caller's own transport).
#include "../../include/bsalign_hip.h"
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

static inline size_t pad16(size_t n){ return (n + 15) & ~(size_t)15; }

// bytes of the blob of pairs [first, first + count): target k, then query k, each padded to 16 bytes
extern "C" size_t bsa_shard_bytes(const uint32_t *qlen, const uint32_t *tlen, size_t first, size_t count){
	size_t acc = 0;
	for(size_t k = first; k < first + count; k++) acc += pad16(tlen[k]) + pad16(qlen[k]);
	return acc;
}

// fills out[0 .. bsa_shard_bytes) (padding zeroed) and the offsets of every pair inside it; threads = 0: hardware concurrency
extern "C" int bsa_shard_pack(const uint8_t *seqs, const uint64_t *qoff, const uint32_t *qlen, const uint64_t *toff, const uint32_t *tlen,
		size_t first, size_t count, uint8_t *out, size_t out_bytes, uint64_t *out_qoff, uint64_t *out_toff, unsigned threads){
	if(count && (!seqs || !qoff || !qlen || !toff || !tlen || !out || !out_qoff || !out_toff)) return BSA_E_ARG;
	size_t acc = 0;
	for(size_t i = 0; i < count; i++){
		const size_t k = first + i;
		out_toff[i] = acc; acc += pad16(tlen[k]);
		out_qoff[i] = acc; acc += pad16(qlen[k]);
	}
	if(acc > out_bytes) return BSA_E_ARG;
	auto body = [&](size_t i){
		const size_t k = first + i;
		uint8_t *t = out + out_toff[i], *q = out + out_qoff[i];
		memcpy(t, seqs + toff[k], tlen[k]); memset(t + tlen[k], 0, pad16(tlen[k]) - tlen[k]);
		memcpy(q, seqs + qoff[k], qlen[k]); memset(q + qlen[k], 0, pad16(qlen[k]) - qlen[k]);
	};
	if(threads == 0) threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
	if(threads <= 1 || count < 1024){ for(size_t i = 0; i < count; i++) body(i); return BSA_OK; }
	std::atomic<size_t> next(0);
	std::vector<std::thread> pool;
	try {
		for(unsigned w = 0; w < threads; w++) pool.emplace_back([&](){
			for(;;){
				const size_t b = next.fetch_add(256);
				if(b >= count) break;
				const size_t e = std::min(count, b + 256);
				for(size_t i = b; i < e; i++) body(i);
			}
		});
	} catch(...){ for(auto &th : pool) th.join(); for(size_t i = 0; i < count; i++) body(i); return BSA_OK; }
	for(auto &th : pool) th.join();
	return BSA_OK;
}
