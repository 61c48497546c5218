// bsa_batcher.hip -- many POA windows, one device sweep per read index.
//
// A single POA (end_bspoa, bspoa.h:4722-4776) aligns its reads one after the other: read r needs the graph that reads
// 0..r-1 built, so one window gives the device ONE sweep program at a time (3 us per row update against 0.28 us on a
// CPU core).  Windows are independent of each other, though, and a caller that has many of them
// (the reference's `bsalign poa` over a file of windows, a polisher over a genome) can advance all of them in lock-step.
// The batcher is the rendezvous for that: every window runs the reference's own host code on a host thread of its own
// (node selection, band placement, traceback into the graph stay the reference's C); where that code would call
// align_rd_bspoacore (bspoa.h:2515-2618) it calls bsa_sweep_batcher_submit() -- same arguments as the single-window
// backend of include/bsalign_poa_adapter.h -- and blocks until its program has run.
// Round 4: no lock-step any more.  A DISPATCHER thread owns the device side: whenever it is free and programs are waiting it takes
// all of them, packs them into one node / edge / candidate / query table, runs ONE launch per distinct parameter set (band width
// differs between the first read of a window and the later ones), copies results and traceback steps back and wakes exactly those
// submitters -- while the other windows' host code keeps the CPUs busy.  (Rounds 2-3 ran a batch only when EVERY live window was
// waiting, packed by the last arrival with everybody blocked: host and device took turns, and 4096 windows spent 1.3 s in one
// thread's memcpy.)  The batch size finds itself: what arrives during one launch is the next batch.  It never waits for more than
// there can be: with every live window waiting, or no window computing, it goes at once.  BSA_POA_BATCH_MIN=all restores lock-step.
// No window ever waits for a window that has finished: bsa_sweep_batcher_leave() takes a participant out.
#include "bsa_common.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

extern "C" int bsa_ctx_get_stream_internal(bsa_ctx_t *c, hipStream_t *st);

namespace {
struct Sub {
	const bsa_row_task_t *tasks; size_t ntasks;
	const uint8_t *query; uint32_t slen;
	bsa_sweep_params_t par;
	uint8_t *rows_out; size_t nblocks;
	bsa_sweep_result_t *res;
	int *rc;
	const uint8_t **rows_src;      // where the submitter finds its row blocks (pinned staging) after the batch ran
	size_t *rows_bytes;
	bool *done; std::condition_variable *cv;      // the submitter's own: nobody else is woken when its program has run
};
struct SubG {                      // a program of the graph form (bsa_poa_batcher_submit_graph)
	const bsa_poa_node_t *nodes; size_t nnodes;
	const bsa_poa_edge_t *edges; size_t nedges;
	const bsa_poa_cand_t *cands; size_t ncands;
	const uint8_t *query; uint32_t slen;
	bsa_sweep_params_t par;
	bsa_poa_result_t *res; size_t cap;
	int *rc;
	const uint32_t **ev_src;            // where the submitter finds its step words (pinned staging) after the batch ran
	int *outbuf;                        // ... and which download buffer that is (given back when the steps are expanded)
	bool *done; std::condition_variable *cv;
};
struct Pinned {
	void *p = nullptr; size_t cap = 0;
	~Pinned(){ if(p) (void)hipHostFree(p); }
	bool need(size_t n){
		if(n <= cap) return true;
		if(p) (void)hipHostFree(p);
		p = nullptr; cap = 0;
		const size_t want = n + n / 4 + 4096;
		if(hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess){ p = nullptr; (void)hipGetLastError(); return false; }
		cap = want; return true;
	}
};
struct Dev {
	void *p = nullptr; size_t cap = 0;
	~Dev(){ if(p) (void)hipFree(p); }
	bool need(size_t n){
		if(n <= cap) return true;
		if(p) (void)hipFree(p);
		p = nullptr; cap = 0;
		const size_t want = n + n / 4 + 4096;
		if(hipMalloc(&p, want) != hipSuccess){ p = nullptr; (void)hipGetLastError(); return false; }
		cap = want; return true;
	}
};
}

struct bsa_sweep_batcher;
#define BSA_BATCH_OUTBUFS 4
// the device side: buffers and statistics; only the dispatcher thread touches it (but for `outstanding`, under the batcher's lock)
struct Engine {
	bsa_sweep_batcher *parent = nullptr;
	bsa_ctx_t *ctx = nullptr;
	Pinned h_gin; Dev d_gin, d_gout;                // graph form: upload staging, device input / output
	struct Out { Pinned mem; int outstanding = 0; } gout[BSA_BATCH_OUTBUFS];       // download staging of the steps: in use until every submitter of its batch has expanded its own
	Pinned h_in, h_rows;                // rows form: upload staging (tasks | progs | qoff | qlen | queries), download staging (results | rows)
	int rows_outstanding = 0;
	Dev d_in, d_rows, d_res;
	hipEvent_t fin = nullptr;           // blocking-sync event: the dispatcher sleeps while the device works, the CPUs belong to the windows
	uint64_t batches = 0, launches = 0, programs = 0, tasks = 0, bytes_up = 0, bytes_down = 0;
	double device_ms = 0, wall_ms = 0, pack_ms = 0, devlock_ms = 0;
	uint64_t pub[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // what bsa_sweep_batcher_stats reports: copied from the counters above under the batcher's mutex after every batch (the counters themselves are the dispatcher's own)
	~Engine(){ if(fin) (void)hipEventDestroy(fin); }
};
struct bsa_sweep_batcher {
	Engine eng;
	std::mutex m;                         // pending programs, `active`, completion flags, download buffers
	std::condition_variable cv_work;                // the dispatcher's one condition: programs are pending OR a download buffer has been given back (two variables let new graph programs wait behind a rows reader)
	uint32_t active = 0;
	std::vector<Sub> pend;
	std::vector<SubG> pendg;
	bool stop = false, lockstep = false;
	std::thread disp;
	// host-side admission: one window thread per CPU the process may use is runnable at a time (bsa_sweep_batcher_enter); a thread
	// waiting in submit() gives its slot to another window.  Hundreds of runnable threads on a CPU-quota'd container otherwise
	// burn the quota in a fraction of each period and everybody is throttled for the rest of it.
	std::mutex tm; std::condition_variable tcv; int tokens = 0; bool gated = false;
	std::unordered_map<std::thread::id, bool> holds;
	void acquire(){ std::unique_lock<std::mutex> lk(tm); tcv.wait(lk, [&]{ return tokens > 0; }); tokens--; holds[std::this_thread::get_id()] = true; }
	void release(){ std::lock_guard<std::mutex> lk(tm); auto it = holds.find(std::this_thread::get_id()); if(it != holds.end() && it->second){ it->second = false; tokens++; tcv.notify_one(); } }
	bool holding(){ std::lock_guard<std::mutex> lk(tm); auto it = holds.find(std::this_thread::get_id()); return it != holds.end() && it->second; }
};
static int host_cpus(){
	if(const char *e = bsa_env("BSA_POA_HOST_THREADS")){ const int v = atoi(e); if(v >= 1) return v; }
	int n = (int)std::thread::hardware_concurrency();
	if(FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")){          // cgroup v2 quota: "<quota> <period>" or "max <period>"
		char q[32]; long per = 0;
		if(fscanf(f, "%31s %ld", q, &per) == 2 && per > 0 && strcmp(q, "max") != 0){ const long v = (atol(q) + per - 1) / per; if(v >= 1 && (n <= 0 || v < n)) n = (int)v; }
		fclose(f);
	}
	return n > 0 ? n : 16;
}
// field by field: the structs carry padding a caller need not have initialised
static bool same_params(const bsa_sweep_params_t &a, const bsa_sweep_params_t &b){
	return a.rows.mode == b.rows.mode && a.rows.bandwidth == b.rows.bandwidth && a.rows.M == b.rows.M && a.rows.X == b.rows.X && a.rows.refbonus == b.rows.refbonus &&
		a.rows.gapo1 == b.rows.gapo1 && a.rows.gape1 == b.rows.gape1 && a.rows.gapo2 == b.rows.gapo2 && a.rows.gape2 == b.rows.gape2 && a.T == b.T;
}

static size_t align16(size_t x){ return (x + 15) & ~(size_t)15; }

// wait for everything queued on the stream without spinning: the CPUs belong to the windows (a blocking-sync event wakes up milliseconds
// late in this runtime, hipStreamSynchronize spins: the event is polled between short sleeps)
static hipError_t wait_stream(Engine *b, hipStream_t st){
	if(!b->fin && hipEventCreateWithFlags(&b->fin, hipEventDisableTiming) != hipSuccess){ b->fin = nullptr; (void)hipGetLastError(); return hipStreamSynchronize(st); }
	hipError_t e = hipEventRecord(b->fin, st);
	if(e != hipSuccess) return e;
	for(unsigned spin = 0; ; spin++){
		e = hipEventQuery(b->fin);
		if(e != hipErrorNotReady) return e;
		if(spin < 20) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(40));
	}
}

// the graph-form programs of a batch: one bsa_poa_graph_run per distinct parameter set; what comes back is 32 bytes per program
// and its traceback steps -- no row block leaves the device.  `ob`: the download buffer of this batch.
static void run_batch_graph(Engine *b, std::vector<SubG> &P, int ob){
	const size_t n = P.size();
	if(n == 0) return;
	std::vector<int> grp(n, -1);
	std::vector<size_t> first;
	for(size_t k = 0; k < n; k++){
		for(size_t g = 0; g < first.size(); g++) if(same_params(P[first[g]].par, P[k].par)){ grp[k] = (int)g; break; }
		if(grp[k] < 0){ grp[k] = (int)first.size(); first.push_back(k); }
	}
	hipStream_t st = nullptr;
	const int rc0 = bsa_ctx_get_stream_internal(b->ctx, &st);
	Pinned &hout = b->gout[ob].mem;
	size_t ev_total = 0;
	std::vector<size_t> ev_off(n, 0);
	for(size_t k = 0; k < n; k++){ ev_off[k] = ev_total; ev_total += P[k].cap; }
	// device: results | packed-steps counter | packed steps | per-program walk scratch; host staging: packed steps of every parameter set, one after the other
	const size_t o_res = 0, o_cnt = align16(n * sizeof(bsa_poa_result_t)), o_pk = o_cnt + 16, o_ev = align16(o_pk + ev_total * 4);
	int rca = rc0;
	if(rca == BSA_OK && (!hout.need(ev_total * 4 + 64) || !b->d_gout.need(o_ev + ev_total * 4 + 64))) rca = BSA_E_NOMEM;
	size_t hused = 0;                   // words of the host staging in use
	for(size_t g = 0; g < first.size(); g++){
		int rc = rca;
		std::vector<size_t> mem;
		for(size_t k = 0; k < n; k++) if(grp[k] == (int)g) mem.push_back(k);
		const size_t np = mem.size();
		const bsa_sweep_params_t par = P[first[g]].par;
		size_t nn = 0, ne = 0, nc = 0, qb = 0; uint32_t max_slen = 0;
		for(size_t i = 0; i < np; i++){ const SubG &s = P[mem[i]]; nn += s.nnodes; ne += s.nedges; nc += s.ncands; qb += align16((size_t)s.slen + 16); max_slen = std::max(max_slen, s.slen); }
		const size_t o_n = 0, o_e = align16(o_n + nn * sizeof(bsa_poa_node_t)), o_c = align16(o_e + ne * sizeof(bsa_poa_edge_t)),
			o_p = align16(o_c + nc * sizeof(bsa_poa_cand_t)), o_q = align16(o_p + np * sizeof(bsa_poa_prog_t)), in_bytes = o_q + qb + 64;
		if(rc == BSA_OK && (!b->h_gin.need(in_bytes) || !b->d_gin.need(in_bytes))) rc = BSA_E_NOMEM;
		const auto tp0 = std::chrono::steady_clock::now();
		if(rc == BSA_OK){
			uint8_t *h = (uint8_t*)b->h_gin.p;
			bsa_poa_prog_t *hp = (bsa_poa_prog_t*)(h + o_p);
			size_t n0 = 0, e0 = 0, c0 = 0, q0 = 0;
			for(size_t i = 0; i < np; i++){
				const SubG &s = P[mem[i]];
				memcpy(h + o_n + n0 * sizeof(bsa_poa_node_t), s.nodes, s.nnodes * sizeof(bsa_poa_node_t));
				if(s.nedges) memcpy(h + o_e + e0 * sizeof(bsa_poa_edge_t), s.edges, s.nedges * sizeof(bsa_poa_edge_t));
				if(s.ncands) memcpy(h + o_c + c0 * sizeof(bsa_poa_cand_t), s.cands, s.ncands * sizeof(bsa_poa_cand_t));
				memcpy(h + o_q + q0, s.query, s.slen);
				bsa_poa_prog_t &pg = hp[i];
				memset(&pg, 0, sizeof(pg));
				pg.first_node = (uint32_t)n0; pg.nnodes = (uint32_t)s.nnodes; pg.first_edge = (uint32_t)e0; pg.nedges = (uint32_t)s.nedges;
				pg.first_cand = (uint32_t)c0; pg.ncands = (uint32_t)s.ncands; pg.slen = s.slen; pg.event_cap = (uint32_t)s.cap;
				pg.query_off = q0; pg.first_event = ev_off[mem[i]];
				n0 += s.nnodes; e0 += s.nedges; c0 += s.ncands; q0 += align16((size_t)s.slen + 16);
			}
		}
#define BCHK(x) do { if(rc == BSA_OK && (x) != hipSuccess){ rc = BSA_E_HIP; (void)hipGetLastError(); } } while(0)
		hipEvent_t e0 = nullptr, e1 = nullptr;
		std::vector<bsa_poa_result_t> hres(np);
		const auto tp1 = std::chrono::steady_clock::now();
		BCHK(hipEventCreate(&e0)); BCHK(hipEventCreate(&e1));
		BCHK(hipMemcpyAsync(b->d_gin.p, b->h_gin.p, in_bytes, hipMemcpyHostToDevice, st));
		BCHK(hipEventRecord(e0, st));
		uint8_t *dres = (uint8_t*)b->d_gout.p + o_res;
		if(rc == BSA_OK){
			const uint8_t *d = (const uint8_t*)b->d_gin.p;
			// results of this parameter set's programs at the start of the result area, copied out before the next set overwrites them
			rc = bsa_poa_graph_run(b->ctx, (const bsa_poa_node_t*)(d + o_n), nn, (const bsa_poa_edge_t*)(d + o_e), (const bsa_poa_cand_t*)(d + o_c),
				(const bsa_poa_prog_t*)(d + o_p), np, d + o_q, max_slen, &par, (bsa_poa_result_t*)dres, (uint32_t*)((uint8_t*)b->d_gout.p + o_ev),
				(uint32_t*)((uint8_t*)b->d_gout.p + o_pk), (uint64_t*)((uint8_t*)b->d_gout.p + o_cnt), nullptr, nullptr);
		}
		BCHK(hipEventRecord(e1, st));
		BCHK(hipMemcpyAsync(hres.data(), dres, np * sizeof(bsa_poa_result_t), hipMemcpyDeviceToHost, st));
		BCHK(wait_stream(b, st));
		// the steps of the set's programs: one copy of the packed words
		size_t total = 0, down = np * sizeof(bsa_poa_result_t);
		if(rc == BSA_OK) for(size_t i = 0; i < np; i++) if(hres[i].nevents > 0) total = std::max(total, (size_t)(uint32_t)hres[i].reserved + (size_t)hres[i].nevents);
		if(rc == BSA_OK && total > ev_total) rc = BSA_E_HIP;
		const size_t hbase = hused;
		if(rc == BSA_OK && total){
			BCHK(hipMemcpyAsync((uint32_t*)hout.p + hbase, (const uint8_t*)b->d_gout.p + o_pk, total * 4, hipMemcpyDeviceToHost, st));
			BCHK(wait_stream(b, st));
			down += total * 4; hused += total;
		}
		{
			const auto tp3 = std::chrono::steady_clock::now();
			b->pack_ms += std::chrono::duration<double, std::milli>(tp1 - tp0).count();
			b->devlock_ms += std::chrono::duration<double, std::milli>(tp3 - tp1).count();
		}
		if(rc == BSA_OK){ float ms = 0; if(hipEventElapsedTime(&ms, e0, e1) == hipSuccess) b->device_ms += ms; }
		if(e0) (void)hipEventDestroy(e0);
		if(e1) (void)hipEventDestroy(e1);
#undef BCHK
		for(size_t i = 0; i < np; i++){
			const SubG &s = P[mem[i]];
			*s.rc = rc; *s.outbuf = ob;
			if(rc == BSA_OK){ *s.res = hres[i]; *s.ev_src = (const uint32_t*)hout.p + hbase + (uint32_t)hres[i].reserved; }
		}
		b->launches++; b->programs += np; b->tasks += nn; b->bytes_up += in_bytes; b->bytes_down += down;
	}
}

// the rows-form programs of a batch (what the graph form declines: a window's first read, scores outside its guard)
static void run_batch_rows(Engine *b, std::vector<Sub> &P){
	const size_t n = P.size();
	if(n == 0) return;
	// groups of equal parameters, in order of first appearance
	std::vector<int> grp(n, -1);
	std::vector<size_t> first;
	for(size_t k = 0; k < n; k++){
		for(size_t g = 0; g < first.size(); g++) if(same_params(P[first[g]].par, P[k].par)){ grp[k] = (int)g; break; }
		if(grp[k] < 0){ grp[k] = (int)first.size(); first.push_back(k); }
	}
	hipStream_t st = nullptr;
	int rc0 = bsa_ctx_get_stream_internal(b->ctx, &st);
	// the download staging holds every program's row blocks until its submitter has copied them out
	size_t rows_total = 0;
	std::vector<size_t> rows_off(n, 0);
	for(size_t k = 0; k < n; k++){
		const bsa_rows_params_t *rp = &P[k].par.rows;
		const size_t blk = bsa_rows_block_bytes(rp->bandwidth, rp->gapo1, rp->gape1, rp->gapo2, rp->gape2);
		rows_off[k] = rows_total;
		if(P[k].rows_out) rows_total += align16(P[k].nblocks * blk);
	}
	if(rc0 == BSA_OK && !b->h_rows.need(rows_total + 64)) rc0 = BSA_E_NOMEM;
	for(size_t g = 0; g < first.size(); g++){
		int rc = rc0;
		std::vector<size_t> mem;
		for(size_t k = 0; k < n; k++) if(grp[k] == (int)g) mem.push_back(k);
		const size_t np = mem.size();
		const bsa_sweep_params_t par = P[first[g]].par;
		const bsa_rows_params_t *rp = &par.rows;
		const size_t blk = bsa_rows_block_bytes(rp->bandwidth, rp->gapo1, rp->gape1, rp->gapo2, rp->gape2);
		size_t ntasks = 0, qbytes = 0, nblocks = 0;
		for(size_t i = 0; i < np; i++){ const Sub &s = P[mem[i]]; ntasks += s.ntasks; qbytes += align16((size_t)s.slen + 64); nblocks += s.nblocks; }
		// upload staging: tasks | progs | qoff | qlen | queries
		const size_t o_tasks = 0, o_progs = align16(o_tasks + ntasks * sizeof(bsa_row_task_t)), o_qoff = align16(o_progs + np * sizeof(bsa_sweep_prog_t)),
			o_qlen = align16(o_qoff + np * 8), o_q = align16(o_qlen + np * 4), in_bytes = o_q + qbytes + 64;
		if(rc == BSA_OK && (!b->h_in.need(in_bytes) || !b->d_in.need(in_bytes) || !b->d_rows.need(nblocks * blk + 64) || !b->d_res.need(np * sizeof(bsa_sweep_result_t)))) rc = BSA_E_NOMEM;
		std::vector<size_t> blk0(np, 0);
		if(rc == BSA_OK){
			uint8_t *h = (uint8_t*)b->h_in.p;
			bsa_row_task_t *ht = (bsa_row_task_t*)(h + o_tasks);
			bsa_sweep_prog_t *hp = (bsa_sweep_prog_t*)(h + o_progs);
			uint64_t *hqo = (uint64_t*)(h + o_qoff);
			uint32_t *hql = (uint32_t*)(h + o_qlen);
			uint8_t *hq = h + o_q;
			size_t t0 = 0, q0 = 0, b0 = 0;
			for(size_t i = 0; i < np && rc == BSA_OK; i++){
				const Sub &s = P[mem[i]];
				if(s.ntasks == 0 || s.nblocks == 0 || t0 + s.ntasks > 0xFFFFFFF0ull || b0 + s.nblocks > 0xFFFFFFF0ull){ rc = BSA_E_ARG; break; }
				memcpy(ht + t0, s.tasks, s.ntasks * sizeof(bsa_row_task_t));
				for(size_t t = t0; t < t0 + s.ntasks; t++){
					if(ht[t].src >= s.nblocks || ht[t].dst >= s.nblocks){ rc = BSA_E_ARG; break; }
					ht[t].query = (uint32_t)i;
				}
				hp[i].first_task = (uint32_t)t0; hp[i].ntasks = (uint32_t)s.ntasks; hp[i].first_block = (uint32_t)b0; hp[i].reserved = 0;
				hqo[i] = q0; hql[i] = s.slen;
				memcpy(hq + q0, s.query, s.slen);
				blk0[i] = b0;
				t0 += s.ntasks; q0 += align16((size_t)s.slen + 64); b0 += s.nblocks;
			}
		}
#define BCHK(x) do { if(rc == BSA_OK && (x) != hipSuccess){ rc = BSA_E_HIP; (void)hipGetLastError(); } } while(0)
		hipEvent_t e0 = nullptr, e1 = nullptr;
		BCHK(hipEventCreate(&e0)); BCHK(hipEventCreate(&e1));
		BCHK(hipMemcpyAsync(b->d_in.p, b->h_in.p, in_bytes, hipMemcpyHostToDevice, st));
		BCHK(hipMemsetAsync(b->d_rows.p, 0, nblocks * blk, st));
		BCHK(hipEventRecord(e0, st));
		if(rc == BSA_OK){
			const uint8_t *d = (const uint8_t*)b->d_in.p;
			rc = bsa_sweep_run(b->ctx, (uint8_t*)b->d_rows.p, (const bsa_row_task_t*)(d + o_tasks), (const bsa_sweep_prog_t*)(d + o_progs), np,
				d + o_q, (const uint64_t*)(d + o_qoff), (const uint32_t*)(d + o_qlen), &par, (bsa_sweep_result_t*)b->d_res.p);
		}
		BCHK(hipEventRecord(e1, st));
		std::vector<bsa_sweep_result_t> hres(np);
		BCHK(hipMemcpyAsync(hres.data(), b->d_res.p, np * sizeof(bsa_sweep_result_t), hipMemcpyDeviceToHost, st));
		size_t down = 0;
		for(size_t i = 0; i < np; i++){
			const Sub &s = P[mem[i]];
			if(!s.rows_out) continue;
			BCHK(hipMemcpyAsync((uint8_t*)b->h_rows.p + rows_off[mem[i]], (const uint8_t*)b->d_rows.p + blk0[i] * blk, s.nblocks * blk, hipMemcpyDeviceToHost, st));
			down += s.nblocks * blk;
		}
		BCHK(wait_stream(b, st));
		if(rc == BSA_OK){ float ms = 0; if(hipEventElapsedTime(&ms, e0, e1) == hipSuccess) b->device_ms += ms; }
		if(e0) (void)hipEventDestroy(e0);
		if(e1) (void)hipEventDestroy(e1);
#undef BCHK
		for(size_t i = 0; i < np; i++){
			const Sub &s = P[mem[i]];
			*s.rc = rc;
			if(rc == BSA_OK){
				*s.res = hres[i];
				*s.rows_src = s.rows_out ? (const uint8_t*)b->h_rows.p + rows_off[mem[i]] : nullptr;
				*s.rows_bytes = s.rows_out ? s.nblocks * blk : 0;
			}
		}
		b->launches++; b->programs += np; b->tasks += ntasks; b->bytes_up += in_bytes; b->bytes_down += down + np * sizeof(bsa_sweep_result_t);
	}
}

// The dispatcher: takes whatever is pending whenever the device side is free.
static void dispatcher(bsa_sweep_batcher *bb){
	Engine *e = &bb->eng;
	std::unique_lock<std::mutex> lk(bb->m);
	for(;;){
		bb->cv_work.wait(lk, [&]{
			const size_t np = bb->pend.size() + bb->pendg.size();
			if(np == 0) return bb->stop;
			return !bb->lockstep || np >= bb->active;
		});
		if(bb->pend.empty() && bb->pendg.empty()){ if(bb->stop) break; continue; }
		std::vector<SubG> tg; std::vector<Sub> tr;
		tg.swap(bb->pendg);
		// rows-form programs wait for the previous rows batch's blocks to be collected (one download staging); graph batches need a free buffer
		if(e->rows_outstanding == 0) tr.swap(bb->pend);
		int ob = -1;
		if(!tg.empty()){
			for(;;){
				for(int k = 0; k < BSA_BATCH_OUTBUFS; k++) if(e->gout[k].outstanding == 0){ ob = k; break; }
				if(ob >= 0) break;
				bb->cv_work.wait(lk);
			}
			e->gout[ob].outstanding = (int)tg.size();
		}
		if(tg.empty() && tr.empty()){ bb->cv_work.wait(lk); continue; }       // (only rows programs, and the staging is still being read)
		e->rows_outstanding += (int)tr.size();
		lk.unlock();
		const auto w0 = std::chrono::steady_clock::now();
		run_batch_graph(e, tg, ob);
		run_batch_rows(e, tr);
		e->batches++;
		e->wall_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
		lk.lock();
		e->pub[0] = e->batches; e->pub[1] = e->launches; e->pub[2] = e->programs; e->pub[3] = e->tasks; e->pub[4] = e->bytes_up; e->pub[5] = e->bytes_down;
		e->pub[6] = (uint64_t)(e->device_ms * 1000.0); e->pub[7] = (uint64_t)(e->wall_ms * 1000.0);
		for(SubG &s : tg){ *s.done = true; s.cv->notify_one(); }
		for(Sub &s : tr){ *s.done = true; s.cv->notify_one(); }
	}
}

extern "C" int bsa_sweep_batcher_create(bsa_ctx_t *ctx, uint32_t participants, bsa_sweep_batcher_t **out){
	if(!ctx || !out || participants == 0) return BSA_E_ARG;
	bsa_sweep_batcher *b = new (std::nothrow) bsa_sweep_batcher();
	if(!b) return BSA_E_NOMEM;
	b->eng.parent = b; b->eng.ctx = ctx; b->active = participants;
	if(const char *e = bsa_env("BSA_POA_BATCH_MIN")) b->lockstep = strcmp(e, "all") == 0;
	b->disp = std::thread(dispatcher, b);
	*out = b;
	return BSA_OK;
}

extern "C" void bsa_sweep_batcher_destroy(bsa_sweep_batcher_t *b){
	if(!b) return;
	{ std::lock_guard<std::mutex> lk(b->m); b->stop = true; }
	b->cv_work.notify_all();
	if(b->disp.joinable()) b->disp.join();
	if(bsa_env("BSA_BATCH_TIMING")){
		const Engine *q = &b->eng;
		fprintf(stderr, "[bsa_sweep_batcher] %llu batches, %llu launches, %llu programs: packing %.1f ms, upload + kernels + download %.1f ms (kernels + upload %.1f ms), all inside batches %.1f ms\n",
			(unsigned long long)q->batches, (unsigned long long)q->launches, (unsigned long long)q->programs, q->pack_ms, q->devlock_ms, q->device_ms, q->wall_ms);
	}
	delete b;
}

// blocks until the dispatcher has run the caller's program; meanwhile the caller's host slot is somebody else's
template<class PushFn>
static int submit_and_wait(bsa_sweep_batcher *bb, bool *done, std::condition_variable *cv, PushFn push){
	std::unique_lock<std::mutex> lk(bb->m);
	if(bb->active == 0) return BSA_E_ARG;
	push();
	bb->cv_work.notify_one();
	const bool had = bb->gated && bb->holding();
	if(had) bb->release();
	cv->wait(lk, [&]{ return *done; });
	lk.unlock();
	if(had) bb->acquire();
	return BSA_OK;
}

extern "C" int bsa_sweep_batcher_submit(void *vb, const bsa_row_task_t *tasks, size_t ntasks, const uint8_t *query, uint32_t slen,
		const bsa_sweep_params_t *par, uint8_t *rows_out, size_t nblocks, bsa_sweep_result_t *res){
	bsa_sweep_batcher *bb = (bsa_sweep_batcher*)vb;
	if(!bb || !tasks || !query || !par || !res) return BSA_E_ARG;
	int rc = BSA_E_HIP;
	const uint8_t *src = nullptr; size_t nbytes = 0;
	bool done = false; std::condition_variable cv;
	const int sr = submit_and_wait(bb, &done, &cv, [&]{ bb->pend.push_back(Sub{tasks, ntasks, query, slen, *par, rows_out, nblocks, res, &rc, &src, &nbytes, &done, &cv}); });
	if(sr != BSA_OK) return sr;
	// the row blocks are copied out here, by every window's own thread; the staging is the next rows batch's once all have
	if(rc == BSA_OK && rows_out && src) memcpy(rows_out, src, nbytes);
	bool last;
	{ std::lock_guard<std::mutex> lk(bb->m); last = --bb->eng.rows_outstanding == 0; }
	if(last) bb->cv_work.notify_one();
	return rc;
}

// a window thread announces itself before it starts computing: it then runs only while it holds one of the host slots
extern "C" void bsa_sweep_batcher_enter(bsa_sweep_batcher_t *bb){
	if(!bb) return;
	// one and a half slots per CPU: a window thread is often blocked for a moment outside submit() (the allocator, a page fault), and with exactly
	// one slot per CPU the CPUs then idle -- 4096 windows: 6.27 s with 24 slots on the 16-CPU quota against 6.6-6.7 s with 16 (the reference: 6.65 s)
	{ std::lock_guard<std::mutex> lk(bb->tm); if(!bb->gated){ bb->gated = true; const int n = host_cpus(); bb->tokens = bsa_env("BSA_POA_HOST_THREADS") ? n : n + n / 2; } }
	bb->acquire();
}

extern "C" void bsa_sweep_batcher_leave(bsa_sweep_batcher_t *bb){
	if(!bb) return;
	if(bb->gated) bb->release();
	{ std::lock_guard<std::mutex> lk(bb->m); if(bb->active) bb->active--; }
	bb->cv_work.notify_one();            // (lock-step mode: the others may be complete now)
}

// graph form of submit(): the signature of the binding's graph backend (include/bsalign_poa_adapter.h).  A parameter set the
// wavefront kernel does not take is declined at once (BSA_E_UNSUPPORTED) -- the window then comes back through submit().
extern "C" int bsa_poa_batcher_submit_graph(void *vb, const bsa_poa_node_t *nodes, size_t nnodes, const bsa_poa_edge_t *edges, size_t nedges,
		const bsa_poa_cand_t *cands, size_t ncands, const uint8_t *query, uint32_t slen, const bsa_sweep_params_t *par,
		bsa_poa_result_t *res, bsa_poa_event_t *events, size_t events_cap){
	bsa_sweep_batcher *bb = (bsa_sweep_batcher*)vb;
	if(!bb || !nodes || !nnodes || !query || !par || !res || !events) return BSA_E_ARG;
	if(bsa_poa_graph_supported(par, slen) == 0 && (!bsa_poa_graph_gen_supported(par) || bsa_env("BSA_POA_NO_GEN"))) return BSA_E_UNSUPPORTED;        // (bands above 256 columns: the generic-width kernel behind the same call, bsa_poa_gen.hip)
	int rc = BSA_E_HIP, ob = -1;
	const uint32_t *src = nullptr;
	bool done = false; std::condition_variable cv;
	const int sr = submit_and_wait(bb, &done, &cv, [&]{ bb->pendg.push_back(SubG{nodes, nnodes, edges, nedges, cands, ncands, query, slen, *par, res, events_cap, &rc, &src, &ob, &done, &cv}); });
	if(sr != BSA_OK) return sr;
	if(rc == BSA_OK){
		if((size_t)res->nevents > events_cap) rc = BSA_E_ARG;
		else if(src) bsa_poa_expand_steps(src, res, events);       // every window's own thread expands its steps
		else res->reserved = 0;
	}
	if(ob >= 0){
		bool last;
		{ std::lock_guard<std::mutex> lk(bb->m); last = --bb->eng.gout[ob].outstanding == 0; }
		if(last) bb->cv_work.notify_one();            // (the dispatcher may be waiting for a free download buffer)
	}
	return rc;
}

// out[0..7] = batches, launches, programs, tasks, bytes uploaded, bytes downloaded, device microseconds, wall microseconds inside the batches
extern "C" void bsa_sweep_batcher_stats(bsa_sweep_batcher_t *bb, uint64_t out[8]){
	if(!bb || !out) return;
	std::lock_guard<std::mutex> lk(bb->m);
	for(int k = 0; k < 8; k++) out[k] = bb->eng.pub[k];
}
