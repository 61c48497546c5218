// bsa_shard_transport.h -- what the shard exchange (bsa_shard_rccl.hip: bsa_shard_scatter / bsa_shard_gather) needs from the wire and from the
// memory its buffers live in.  Two implementations:
//   RCCL over xGMI, buffers in device memory            (bsa_shard_rccl.hip; the product path: one process per GPU)
//   POSIX shared memory between processes of one host,  (bsa_shard_shm.cpp; BSA_SHARD_TRANSPORT=shm: no GPU, no RCCL -- it exists so that the
//   buffers in host memory                               exchange's rank arithmetic runs at world size 2 in `pytest -m "not gpu"`)
// The exchange itself is written once against these two interfaces.
#pragma once
#include <stddef.h>
#include <stdint.h>

struct BsaShardSpace {                       // where message buffers live
	virtual ~BsaShardSpace(){}
	virtual void *alloc(size_t bytes) = 0;   // nullptr on failure
	virtual void release(void *p) = 0;
	virtual int to_space(void *dst, const void *host_src, size_t bytes) = 0;        // BSA_OK / BSA_E_HIP; ordered with the transport's operations
	virtual int to_host(void *host_dst, const void *src, size_t bytes) = 0;
	virtual int within(void *dst, const void *src, size_t bytes) = 0;
	virtual int sync() = 0;                  // everything issued so far has completed (copies, messages)
};

struct BsaShardTransport {
	virtual ~BsaShardTransport(){}
	virtual int broadcast(void *buf, size_t bytes, int root) = 0;                      // in place
	virtual int allgather(const void *mine, void *all, size_t bytes_each) = 0;         // all = nranks x bytes_each, rank order
	// point-to-point: the operations between begin and end are posted together and complete together (RCCL: one ncclGroup -- on MI355X one
	// message per xGMI link, all links busy at once; shm: progressed round-robin, so no order of posting can deadlock)
	virtual int group_begin() = 0;
	virtual int send(const void *buf, size_t bytes, int peer) = 0;
	virtual int recv(void *buf, size_t bytes, int peer) = 0;
	virtual int group_end() = 0;
};

// bsa_shard_shm.cpp
int bsa_shm_unique_id(uint8_t id[128]);
BsaShardTransport *bsa_shm_transport_create(int rank, int nranks, const uint8_t id[128]);      // nullptr on failure (rendezvous of all ranks, 120 s)
BsaShardSpace *bsa_host_space_create();
