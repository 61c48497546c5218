// bsa_shard_shm.cpp -- the shard exchange's transport between processes of ONE host through POSIX shared memory, buffers in host memory
// (bsa_shard_transport.h).  Not a product path: one process per GPU talks RCCL over xGMI.  It exists so that bsa_shard_scatter /
// bsa_shard_gather -- ranges, offsets, who sends what to whom, the agreement on errors -- execute at world size > 1 where there is neither a
// GPU nor RCCL (tests/test_shard_cpu.py starts two processes with BSA_SHARD_TRANSPORT=shm).
//
// One segment, named by the 128-byte id: a header (arrival counter) and, for every ordered pair of ranks, a single-producer single-consumer
// byte ring.  A message has no header: sender and receiver know its length (as with ncclSend / ncclRecv).  The operations of a group make
// progress together -- every posted send pushes what fits, every posted receive pops what is there -- until all are done.
#include "bsa_shard_transport.h"
#include "../../include/bsalign_hip.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

struct Ring {                                 // bytes flow src -> dst; head is the producer's, tail the consumer's
	std::atomic<uint64_t> head, tail;
	uint8_t pad[48];
};
struct Header { std::atomic<uint32_t> arrived, departed; uint32_t nranks, ring_bytes; uint8_t pad[48]; };

struct HostSpace : BsaShardSpace {
	void *alloc(size_t bytes) override { return malloc(bytes ? bytes : 1); }
	void release(void *p) override { free(p); }
	int to_space(void *d, const void *s, size_t n) override { if(n) memcpy(d, s, n); return BSA_OK; }
	int to_host(void *d, const void *s, size_t n) override { if(n) memcpy(d, s, n); return BSA_OK; }
	int within(void *d, const void *s, size_t n) override { if(n) memmove(d, s, n); return BSA_OK; }
	int sync() override { return BSA_OK; }
};

struct Op { bool is_send; uint8_t *buf; size_t bytes, done; int peer; };

struct ShmTransport : BsaShardTransport {
	int rank = 0, nranks = 1;
	uint8_t *base = nullptr; size_t seg_bytes = 0; uint32_t ring_bytes = 0;
	std::vector<Op> ops; bool grouping = false;
	Header *hdr() const { return (Header*)base; }
	Ring *ring(int src, int dst) const { return (Ring*)(base + sizeof(Header) + ((size_t)src * nranks + dst) * (sizeof(Ring) + ring_bytes)); }
	uint8_t *data(Ring *r) const { return (uint8_t*)(r + 1); }
	~ShmTransport() override {
		if(base){ hdr()->departed.fetch_add(1); munmap(base, seg_bytes); }
	}
	// one pass over the posted operations; true when all are complete
	bool progress(){
		bool all = true;
		// (a ring is one byte stream: an operation waits for the earlier ones on the same ring in the same direction, or a later one
		// would slip its bytes in between when room -- or data -- turns up while the pass is under way)
		std::vector<char> held(2 * (size_t)nranks, 0);
		for(Op &o : ops){
			if(o.done == o.bytes) continue;
			char &hold = held[(o.is_send ? 0 : (size_t)nranks) + (size_t)o.peer];
			if(hold){ all = false; continue; }
			Ring *r = o.is_send ? ring(rank, o.peer) : ring(o.peer, rank);
			uint64_t h = r->head.load(std::memory_order_acquire), t = r->tail.load(std::memory_order_acquire);
			if(o.is_send){
				size_t room = ring_bytes - (size_t)(h - t), n = std::min(room, o.bytes - o.done);
				for(size_t moved = 0; moved < n; ){
					const size_t at = (size_t)((h + moved) % ring_bytes), run = std::min(n - moved, (size_t)ring_bytes - at);
					memcpy(data(r) + at, o.buf + o.done + moved, run); moved += run;
				}
				if(n){ r->head.store(h + n, std::memory_order_release); o.done += n; }
			} else {
				size_t avail = (size_t)(h - t), n = std::min(avail, o.bytes - o.done);
				for(size_t moved = 0; moved < n; ){
					const size_t at = (size_t)((t + moved) % ring_bytes), run = std::min(n - moved, (size_t)ring_bytes - at);
					memcpy(o.buf + o.done + moved, data(r) + at, run); moved += run;
				}
				if(n){ r->tail.store(t + n, std::memory_order_release); o.done += n; }
			}
			if(o.done != o.bytes){ all = false; hold = 1; }
		}
		return all;
	}
	int run_ops(){
		const auto t0 = std::chrono::steady_clock::now();
		for(unsigned spins = 0; !progress(); spins++){
			if((spins & 63u) == 63u){
				std::this_thread::yield();
				if(std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)){ ops.clear(); return BSA_E_HIP; }      // a peer that never posts its side
			}
		}
		ops.clear();
		return BSA_OK;
	}
	int group_begin() override { grouping = true; return BSA_OK; }
	int send(const void *buf, size_t bytes, int peer) override {
		if(peer < 0 || peer >= nranks || peer == rank) return BSA_E_ARG;
		ops.push_back({true, (uint8_t*)buf, bytes, 0, peer});
		return grouping ? BSA_OK : run_ops();
	}
	int recv(void *buf, size_t bytes, int peer) override {
		if(peer < 0 || peer >= nranks || peer == rank) return BSA_E_ARG;
		ops.push_back({false, (uint8_t*)buf, bytes, 0, peer});
		return grouping ? BSA_OK : run_ops();
	}
	int group_end() override { grouping = false; return run_ops(); }
	int broadcast(void *buf, size_t bytes, int root) override {
		if(nranks == 1 || bytes == 0) return BSA_OK;
		group_begin();
		if(rank == root){ for(int k = 0; k < nranks; k++) if(k != root) send(buf, bytes, k); }
		else recv(buf, bytes, root);
		return group_end();
	}
	int allgather(const void *mine, void *all, size_t each) override {
		memcpy((uint8_t*)all + (size_t)rank * each, mine, each);
		if(nranks == 1 || each == 0) return BSA_OK;
		group_begin();
		for(int k = 0; k < nranks; k++) if(k != rank){ send(mine, each, k); recv((uint8_t*)all + (size_t)k * each, each, k); }
		return group_end();
	}
};

void seg_name(const uint8_t id[128], char out[64]){
	static const char hex[] = "0123456789abcdef";
	memcpy(out, "/bsa_shard_", 11);
	for(int i = 0; i < 16; i++){ out[11 + 2 * i] = hex[id[i] >> 4]; out[12 + 2 * i] = hex[id[i] & 15]; }
	out[43] = 0;
}

}  // namespace

int bsa_shm_unique_id(uint8_t id[128]){
	memset(id, 0, 128);
	FILE *f = fopen("/dev/urandom", "rb");
	if(!f) return BSA_E_UNSUPPORTED;
	const size_t got = fread(id, 1, 16, f);
	fclose(f);
	return got == 16 ? BSA_OK : BSA_E_UNSUPPORTED;
}

BsaShardSpace *bsa_host_space_create(){ return new HostSpace(); }

BsaShardTransport *bsa_shm_transport_create(int rank, int nranks, const uint8_t id[128]){
	char name[64]; seg_name(id, name);
	const uint32_t ring_bytes = nranks <= 4 ? (1u << 20) : (1u << 18);
	const size_t bytes = sizeof(Header) + (size_t)nranks * nranks * (sizeof(Ring) + ring_bytes);
	const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
	if(fd < 0) return nullptr;
	if(ftruncate(fd, (off_t)bytes) != 0){ close(fd); return nullptr; }          // (every rank sets the same size; new pages are zero: counters and rings start empty)
	void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	if(p == MAP_FAILED) return nullptr;
	ShmTransport *t = new ShmTransport();
	t->rank = rank; t->nranks = nranks; t->base = (uint8_t*)p; t->seg_bytes = bytes; t->ring_bytes = ring_bytes;
	t->hdr()->arrived.fetch_add(1);
	const auto t0 = std::chrono::steady_clock::now();
	while(t->hdr()->arrived.load() < (uint32_t)nranks){
		std::this_thread::sleep_for(std::chrono::milliseconds(1));
		if(std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)){ shm_unlink(name); delete t; return nullptr; }
	}
	if(rank == 0) shm_unlink(name);          // everybody has it mapped: the name can go, the memory stays until the last unmap
	return t;
}
