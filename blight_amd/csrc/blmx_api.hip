// blmx_api.hip -- the C ABI of include/blight_mi355x.h: parameter validation (reference blight.h:75-92), the host FASTA
// reader (reference blight.cpp:212-229; I/O is host work), staging of host buffers, and dispatch to the HIP build and
// query paths.  No exception crosses the ABI; there is no CPU compute path.
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>

#include <execinfo.h>
#include <signal.h>

#include "blmx_internal.h"

using namespace blmx;

struct blmx_index {
	Index ix;
};

struct blmx_payload {
	blmx_index* owner;
	Payload pl;
};

namespace {

// Error texts are per calling thread (the reference's callers query from OpenMP threads): blmx_last_error returns the text of
// the calling thread's last failed call.
thread_local std::string g_create_err;
thread_local std::string g_last_err;

int fail(blmx_index* h, int code, const std::string& msg)
{
	g_last_err = msg;
	if (!h)  // calls without an index (communicator, primitives): blmx_last_error(NULL) reports them
		g_create_err = msg;
	return code;
}

int env_int(const char* name, int dflt)
{
	const char* e = getenv(name);
	return e ? atoi(e) : dflt;
}

template <typename F> int guarded(blmx_index* h, F&& f)
{
	try {
		f();
		if (h)
			h->ix.check_dropped();  // (a partition pass that lost k-mers: host-buffer calls have finished here, device calls tell the next call)
		return BLMX_OK;
	} catch (const StatusError& e) {
		return fail(h, e.code, e.msg);
	} catch (const std::bad_alloc&) {
		return fail(h, BLMX_ENOMEM, "host allocation failed");
	} catch (const std::exception& e) {
		return fail(h, BLMX_EHIP, e.what());
	}
}

void ensure(DevBuf& b, size_t bytes, hipStream_t s)
{
	if (bytes > b.bytes)
		BLMX_HIP(b.alloc(bytes + bytes / 4 + 256, false, s));
}

void build_host(blmx_index* h, const char* concat, const uint64_t* offsets, uint64_t nseq)
{
	Index& ix = h->ix;
	if (ix.built)
		throw StatusError{BLMX_ESTATE, "index already built"};
	BLMX_HIP(hipSetDevice(ix.prm.device));
	const uint64_t total = nseq ? offsets[nseq] : 0;
	if (nseq && offsets[0] != 0)
		throw StatusError{BLMX_EINVAL, "offsets[0] must be 0"};
	DevBuf d_ascii, d_offs;
	BLMX_HIP(d_ascii.alloc(total + 64, false, ix.stream));
	BLMX_HIP(d_offs.alloc((nseq + 1) * 8, true, ix.stream));
	if (total)
		BLMX_HIP(hipMemcpyAsync(d_ascii.p, concat, total, hipMemcpyHostToDevice, ix.stream));
	if (nseq)
		BLMX_HIP(hipMemcpyAsync(d_offs.p, offsets, (nseq + 1) * 8, hipMemcpyHostToDevice, ix.stream));
	build_from_device(ix, d_ascii.as<uint8_t>(), d_offs.as<uint64_t>(), nseq, total);
	ix.free_after_filter(d_ascii);  // (a hipFree here would wait for the filter's passes)
	ix.free_after_filter(d_offs);
	ix.built = true;
}

// host sequences -> device staging, then fn(d_ascii, d_offs, total)
template <typename F>
void with_staged_sequences(Index& ix, const char* concat, const uint64_t* offsets, uint64_t nseq, F&& fn)
{
	BLMX_HIP(hipSetDevice(ix.prm.device));
	const uint64_t total = nseq ? offsets[nseq] : 0;
	if (nseq && offsets[0] != 0)
		throw StatusError{BLMX_EINVAL, "offsets[0] must be 0"};
	DevBuf d_ascii, d_offs;
	BLMX_HIP(d_ascii.alloc(total + 64, false, ix.stream));
	BLMX_HIP(d_offs.alloc((nseq + 1) * 8, true, ix.stream));
	if (total)
		BLMX_HIP(hipMemcpyAsync(d_ascii.p, concat, total, hipMemcpyHostToDevice, ix.stream));
	if (nseq)
		BLMX_HIP(hipMemcpyAsync(d_offs.p, offsets, (nseq + 1) * 8, hipMemcpyHostToDevice, ix.stream));
	fn(d_ascii.as<uint8_t>(), d_offs.as<uint64_t>(), total);
}

}  // namespace

extern "C" {

const char* blmx_version(void) { return "blight-mi355x 0.1 (gfx950)"; }
// BLIGHT_SEGV_TRACE=1: print the native stack of the faulting thread on SIGSEGV (debugging aid: Python's faulthandler stops at the ctypes call)
namespace {
void segv_trace(int sig)
{
	void* frames[64];
	const int n = backtrace(frames, 64);
	backtrace_symbols_fd(frames, n, 2);
	signal(sig, SIG_DFL);
	raise(sig);
}
struct SegvTraceInit {
	SegvTraceInit()
	{
		const char* e = getenv("BLIGHT_SEGV_TRACE");
		if (e && atoi(e) != 0)
			signal(SIGSEGV, segv_trace);
	}
} segv_trace_init;
}  // namespace

int blmx_abi_version(void) { return BLMX_ABI_VERSION; }
const char* blmx_create_error(void) { return g_create_err.c_str(); }
const char* blmx_last_error(const blmx_index* ix) { return ix ? g_last_err.c_str() : g_create_err.c_str(); }

int blmx_create(const blmx_params* p, blmx_index** out)
{
	if (out)
		*out = nullptr;
	if (!p || !out) {
		g_create_err = "null argument";
		return BLMX_EINVAL;
	}
	blmx_params prm = *p;
	if (prm.kmer_bytes == 0)
		prm.kmer_bytes = prm.k <= 32 ? 8 : 16;
	if (prm.kmer_bytes != 8 && prm.kmer_bytes != 16) {
		g_create_err = "kmer_bytes must be 8 or 16";
		return BLMX_EINVAL;
	}
	// ctor checks and messages of the reference, blight.h:75-92
	const char* msg = nullptr;
	if (prm.k > prm.kmer_bytes * 4)
		msg = "kmer size too large";
	else if ((prm.m & 1) == 0)
		msg = "minimizer_length must be odd";
	else if (prm.m > 16)
		msg = "minimizer_length size too large";
	else if (prm.log2_mphf > 2 * prm.m - 1)
		msg = "log2_mphfs_number must not be larger than 2*minimizer_length - 1";
	else if (prm.log2_superbuckets > prm.log2_mphf)
		msg = "log2_superbuckets_number must not be larger than log2_mphfs_number";
	// limits of this implementation (outside the reference's own usable range: SURVEY D-3)
	else if (prm.k < prm.m || prm.k < 2)
		msg = "k must be at least max(2, minimizer_length)";
	else if (prm.kmer_bytes == 16 && prm.k > 63)
		msg = "kmer size too large";
	else if (prm.bits_saved > 20)
		msg = "bits_to_save must be at most 20";
	if (msg) {
		g_create_err = msg;
		return BLMX_EINVAL;
	}
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
		g_create_err = "no HIP device: the Blight MI355X path has no CPU fallback";
		return BLMX_EHIP;
	}
	if (prm.device < 0 || prm.device >= ndev) {
		g_create_err = "invalid device ordinal";
		return BLMX_EINVAL;
	}
	blmx_index* h = new (std::nothrow) blmx_index();
	if (!h) {
		g_create_err = "host allocation failed";
		return BLMX_ENOMEM;
	}
	h->ix.prm = prm;
	h->ix.KW = (int)prm.kmer_bytes / 8;
	h->ix.gshift = (int)(2 * prm.m - 1 - prm.log2_mphf);
	h->ix.nbuckets = 1u << (2 * prm.m - 1);
	h->ix.ngroups = 1u << prm.log2_mphf;
	// tuning knobs: seeded from the environment once per index (blmx_set_tuning changes them later)
	h->ix.bin_policy = env_int("BLIGHT_QUERY_BINNING", 0);
	h->ix.tune_bin_bits = env_int("BLIGHT_BIN_BITS", 0);
	h->ix.tune_filter_bits = env_int("BLIGHT_FILTER_BITS", -1);
	h->ix.tune_absent_filter = env_int("BLIGHT_ABSENT_FILTER", 1);
	h->ix.tune_filter_gate = env_int("BLIGHT_FILTER_GATE", 1);
	h->ix.tune_pos_hint = env_int("BLIGHT_POS_HINT", 1);
	h->ix.tune_mini_handoff = env_int("BLIGHT_MINI_HANDOFF", -1);
	h->ix.tune_query_heads = env_int("BLIGHT_QUERY_HEADS", 1);
	h->ix.tune_reads_mode = env_int("BLIGHT_READS_MODE", 0);
	h->ix.tune_partition = env_int("BLIGHT_PARTITION", 0);
	h->ix.tune_window_kb = env_int("BLIGHT_WINDOW_KB", 0);
	h->ix.tune_fat_lines = env_int("BLIGHT_FAT_LINES", -1);
	h->ix.tune_serve_burst = env_int("BLIGHT_SERVE_BURST", 1);
	h->ix.tune_build_passes = env_int("BLIGHT_BUILD_PASSES", 0);
	if (hipSetDevice(prm.device) != hipSuccess || hipStreamCreateWithFlags(&h->ix.stream, hipStreamNonBlocking) != hipSuccess) {
		g_create_err = "hipStreamCreate failed";
		delete h;
		return BLMX_EHIP;
	}
	*out = h;
	return BLMX_OK;
}

void blmx_destroy(blmx_index* h)
{
	if (!h)
		return;
	(void)hipSetDevice(h->ix.prm.device);
	(void)hipDeviceSynchronize();  // device-pointer calls may still be running on the caller's streams
	burst_destroy(h->ix);
	destroy_host_front(h->ix.hf);
	h->ix.hf = nullptr;
	shard_build_abort(h->ix);
	h->ix.filter_wait();
	if (h->ix.filter_ready)
		(void)hipEventDestroy(h->ix.filter_ready);
	if (h->ix.bg_stream)
		(void)hipStreamDestroy(h->ix.bg_stream);
	for (hipEvent_t e : h->ix.prof_ev)
		if (e)
			(void)hipEventDestroy(e);
	if (h->ix.bin_done)
		(void)hipEventDestroy(h->ix.bin_done);
	if (h->ix.h_dropped)
		(void)hipHostFree(h->ix.h_dropped);
	if (h->ix.stream) {
		(void)hipStreamSynchronize(h->ix.stream);
		(void)hipStreamDestroy(h->ix.stream);
	}
	delete h;
}

int blmx_build_from_fasta(blmx_index* h, const char* path)
{
	if (!h || !path)
		return BLMX_EINVAL;
	return guarded(h, [&] { build_from_fasta_streamed(h->ix, path); });
}

int blmx_build_from_sequences(blmx_index* h, const char* concat, const uint64_t* offsets, uint64_t nseq)
{
	if (!h || (nseq && (!concat || !offsets)))
		return BLMX_EINVAL;
	return guarded(h, [&] { build_host(h, concat, offsets, nseq); });
}

int blmx_build_from_device_sequences(blmx_index* h, const void* d_concat, const uint64_t* d_offsets, uint64_t nseq, uint64_t total_len)
{
	if (!h || (nseq && (!d_concat || !d_offsets)))
		return BLMX_EINVAL;
	return guarded(h, [&] {
		if (h->ix.built)
			throw StatusError{BLMX_ESTATE, "index already built"};
		BLMX_HIP(hipSetDevice(h->ix.prm.device));
		BLMX_HIP(hipDeviceSynchronize());  // the build runs on the library's own stream: wait for whoever produced the input
		build_from_device(h->ix, reinterpret_cast<const uint8_t*>(d_concat), d_offsets, nseq, total_len);
		h->ix.built = true;
	});
}

int blmx_query_kmers(blmx_index* h, const void* kmers, uint64_t nq, int64_t* ids_out)
{
	if (!h || (nq && (!kmers || !ids_out)))
		return BLMX_EINVAL;
	return guarded(h, [&] { query_kmers_host(h->ix, kmers, nullptr, nq, ids_out, kQueryIds); });
}

int blmx_query_kmers_with_minimizer(blmx_index* h, const void* kmers, const uint32_t* minis, uint64_t nq, int64_t* ids_out)
{
	if (!h || (nq && (!kmers || !ids_out || !minis)))
		return BLMX_EINVAL;
	return guarded(h, [&] { query_kmers_host(h->ix, kmers, minis, nq, ids_out, kQueryIds); });
}

int blmx_query_positions(blmx_index* h, const void* kmers, const uint32_t* minis, uint64_t nq, int32_t* pos_out)
{
	if (!h || (nq && (!kmers || !pos_out || !minis)))
		return BLMX_EINVAL;
	return guarded(h, [&] { query_kmers_host(h->ix, kmers, minis, nq, pos_out, kQueryPos); });
}

int blmx_query_membership(blmx_index* h, const void* kmers, uint64_t nq, uint8_t* present_out)
{
	if (!h || (nq && (!kmers || !present_out)))
		return BLMX_EINVAL;
	return guarded(h, [&] { query_kmers_host(h->ix, kmers, nullptr, nq, present_out, kQueryBool); });
}

uint64_t blmx_sequence_kmer_count(const blmx_index* h, const uint64_t* offsets, uint64_t nseq)
{
	if (!h || !offsets)
		return 0;
	const uint64_t k = h->ix.prm.k;
	uint64_t n = 0;
	for (uint64_t i = 0; i < nseq; i++) {
		const uint64_t len = offsets[i + 1] - offsets[i];
		if (len >= k)
			n += len - k + 1;
	}
	return n;
}

int blmx_query_sequences(blmx_index* h, const char* concat, const uint64_t* offsets, uint64_t nseq, int64_t* ids_out)
{
	if (!h || (nseq && (!concat || !offsets)))
		return BLMX_EINVAL;
	return guarded(h, [&] { query_sequences_host(h->ix, concat, offsets, nseq, ids_out); });
}

int blmx_save(blmx_index* h, const char* path)
{
	if (!h || !path)
		return BLMX_EINVAL;
	return guarded(h, [&] { save_index(h->ix, path); });
}

int blmx_load(blmx_index* h, const char* path)
{
	if (!h || !path)
		return BLMX_EINVAL;
	return guarded(h, [&] { load_index(h->ix, path); });
}

int blmx_peek_params(const char* path, blmx_params* out)
{
	if (!path || !out)
		return BLMX_EINVAL;
	return peek_index_params(path, *out);
}

int blmx_query_file(blmx_index* h, const char* path, uint64_t* good, uint64_t* erroneous)
{
	if (!h || !path || !good || !erroneous)
		return BLMX_EINVAL;
	return guarded(h, [&] {
		std::lock_guard<std::mutex> lock(h->ix.mu);
		query_file(h->ix, path, *good, *erroneous);
	});
}

int blmx_payload_create(blmx_index* h, int kind, uint32_t ncolors, blmx_payload** out)
{
	if (out)
		*out = nullptr;
	if (!h || !out)
		return BLMX_EINVAL;
	return guarded(h, [&] {
		std::unique_ptr<blmx_payload> p(new blmx_payload());
		p->owner = h;
		p->pl.ix = &h->ix;
		p->pl.kind = kind;
		p->pl.ncolors = kind == 0 ? 1 : ncolors;
		payload_init(p->pl);
		*out = p.release();
	});
}

void blmx_payload_destroy(blmx_payload* p)
{
	if (!p)
		return;
	(void)hipSetDevice(p->pl.ix->prm.device);
	(void)hipDeviceSynchronize();
	delete p;
}

int blmx_payload_add_file(blmx_payload* p, const char* path, uint32_t color, uint64_t* found_out, uint64_t* absent_out)
{
	if (!p || !path)
		return BLMX_EINVAL;
	return guarded(p->owner, [&] {
		std::lock_guard<std::mutex> lock(p->owner->ix.mu);
		uint64_t found = 0, absent = 0;
		payload_add_file(p->pl, path, color, found, absent);
		if (found_out)
			*found_out = found;
		if (absent_out)
			*absent_out = absent;
	});
}

int blmx_payload_add_ids_device(blmx_payload* p, const int64_t* d_ids, uint64_t n, uint32_t color, void* stream)
{
	if (!p || (n && !d_ids))
		return BLMX_EINVAL;
	return guarded(p->owner, [&] {
		BLMX_HIP(hipSetDevice(p->pl.ix->prm.device));
		payload_add_ids(p->pl, d_ids, n, color, reinterpret_cast<hipStream_t>(stream));
	});
}

int64_t blmx_payload_export(blmx_payload* p, void* out, uint64_t cap_bytes)
{
	if (!p)
		return BLMX_EINVAL;
	int64_t result = -1;
	const int rc = guarded(p->owner, [&] {
		BLMX_HIP(hipSetDevice(p->pl.ix->prm.device));
		BLMX_HIP(hipDeviceSynchronize());  // updates may have been enqueued on caller streams
		result = (int64_t)p->pl.bytes;
		const uint64_t n = std::min<uint64_t>(p->pl.bytes, cap_bytes);
		if (out && n)
			BLMX_HIP(hipMemcpy(out, p->pl.data.p, n, hipMemcpyDeviceToHost));
	});
	return rc == BLMX_OK ? result : rc;
}

int blmx_query_kmers_device(blmx_index* h, const void* d_kmers, uint64_t nq, int64_t* d_ids, void* stream)
{
	if (!h || (nq && (!d_kmers || !d_ids)))
		return BLMX_EINVAL;
	return guarded(h, [&] {
		if (!h->ix.built)
			throw StatusError{BLMX_ESTATE, "index not built"};
		BLMX_HIP(hipSetDevice(h->ix.prm.device));
		hipStream_t s = reinterpret_cast<hipStream_t>(stream);  // NULL = HIP's null stream, like any other stream
		launch_query_ids_auto(h->ix, d_kmers, nq, d_ids, s);
		h->ix.number_query.fetch_add(nq, std::memory_order_relaxed);
	});
}

int blmx_query_sequences_device(blmx_index* h, const void* d_concat, const uint64_t* d_offsets, uint64_t nseq, uint64_t total_len, uint64_t nk,
                                int64_t* d_ids, int* d_err_flag, void* stream)
{
	if (!h || (nseq && (!d_concat || !d_offsets)) || (nk && !d_ids))
		return BLMX_EINVAL;
	return guarded(h, [&] {
		query_sequences_device(h->ix, reinterpret_cast<const uint8_t*>(d_concat), d_offsets, nseq, total_len, nk, d_ids, d_err_flag,
		                       reinterpret_cast<hipStream_t>(stream));
	});
}

int blmx_set_profiling(blmx_index* h, int on)
{
	if (!h)
		return BLMX_EINVAL;
	h->ix.prof_on = on != 0;
	h->ix.prof_reset();
	return BLMX_OK;
}

int blmx_last_profile(blmx_index* h, float* ms_out, int cap)
{
	if (!h || !ms_out || cap < Index::kProfEvents - 1)
		return BLMX_EINVAL;
	Index& ix = h->ix;
	int last = -1;
	for (int i = 0; i < Index::kProfEvents; i++)
		if (ix.prof_set[i])
			last = i;
	if (last < 0)
		return 0;
	if (hipEventSynchronize(ix.prof_ev[last]) != hipSuccess)
		return BLMX_EHIP;
	int prev = -1;
	for (int i = 0; i < Index::kProfEvents; i++) {
		if (i > 0)
			ms_out[i - 1] = 0.0f;
		if (!ix.prof_set[i])
			continue;
		if (prev >= 0) {
			float ms = 0.0f;
			(void)hipEventElapsedTime(&ms, ix.prof_ev[prev], ix.prof_ev[i]);
			ms_out[i - 1] = ms;
		}
		prev = i;
	}
	return Index::kProfEvents - 1;
}

int blmx_kmer_groups_device(blmx_index* h, const void* d_kmers, uint64_t nq, uint32_t* d_groups, void* stream)
{
	if (!h || (nq && (!d_kmers || !d_groups)))
		return BLMX_EINVAL;
	return guarded(h, [&] {
		BLMX_HIP(hipSetDevice(h->ix.prm.device));
		hipStream_t s = reinterpret_cast<hipStream_t>(stream);  // NULL = HIP's null stream, like any other stream
		launch_kmer_groups(h->ix, d_kmers, nq, d_groups, s);
	});
}

int blmx_kmers_at_device(blmx_index* h, const void* d_ascii, const uint64_t* d_positions, uint64_t nq, void* d_out, int* d_err, void* stream)
{
	if (!h || (nq && (!d_ascii || !d_positions || !d_out)))
		return BLMX_EINVAL;
	return guarded(h, [&] {
		BLMX_HIP(hipSetDevice(h->ix.prm.device));
		hipStream_t s = reinterpret_cast<hipStream_t>(stream);  // NULL = HIP's null stream, like any other stream
		launch_kmers_at(h->ix, reinterpret_cast<const uint8_t*>(d_ascii), d_positions, nq, d_out, d_err, s);
	});
}

int blmx_canonicalize_device(blmx_index* h, void* d_kmers, uint64_t nq, void* stream)
{
	if (!h || (nq && !d_kmers))
		return BLMX_EINVAL;
	return guarded(h, [&] {
		BLMX_HIP(hipSetDevice(h->ix.prm.device));
		hipStream_t s = reinterpret_cast<hipStream_t>(stream);  // NULL = HIP's null stream, like any other stream
		launch_canonicalize(h->ix, d_kmers, nq, s);
	});
}

int blmx_set_shard(blmx_index* h, uint32_t rank, uint32_t world)
{
	if (!h)
		return BLMX_EINVAL;
	if (h->ix.built)
		return fail(h, BLMX_ESTATE, "blmx_set_shard must precede the build");
	if (world == 0 || world > 16 || rank >= world)  // the partitioned exchange splits its bins at up to 15 shard boundaries
		return fail(h, BLMX_EINVAL, "need rank < world <= 16");
	h->ix.shard_rank = rank;
	h->ix.shard_world = world;
	return BLMX_OK;
}

int blmx_shard_bounds(const blmx_index* h, uint32_t* bounds_out)
{
	if (!h || !bounds_out || !h->ix.built)
		return BLMX_EINVAL;
	for (size_t i = 0; i < h->ix.shard_bounds.size(); i++)
		bounds_out[i] = h->ix.shard_bounds[i];
	return BLMX_OK;
}

int blmx_route_count_device(blmx_index* h, const void* d_kmers, uint64_t nq, const uint32_t* d_bounds, uint32_t world, uint8_t* d_dest,
                            uint64_t* d_counts, void* stream)
{
	if (!h || !d_bounds || !d_counts || world == 0 || world > 255 || (nq && (!d_kmers || !d_dest)))
		return BLMX_EINVAL;
	return guarded(h, [&] {
		BLMX_HIP(hipSetDevice(h->ix.prm.device));
		hipStream_t s = reinterpret_cast<hipStream_t>(stream);  // NULL = HIP's null stream, like any other stream
		launch_route_count(h->ix, d_kmers, nq, d_bounds, world, d_dest, d_counts, s);
	});
}

int blmx_route_scatter_device(blmx_index* h, const void* d_kmers, const uint8_t* d_dest, uint64_t nq, const uint64_t* d_offsets, uint32_t world,
                              void* d_sorted, uint32_t* d_order, uint64_t* d_cursor, void* stream)
{
	if (!h || !d_offsets || !d_cursor || world == 0 || world > 255 || nq >= 0xffffffffull || (nq && (!d_kmers || !d_dest || !d_sorted || !d_order)))
		return BLMX_EINVAL;
	return guarded(h, [&] {
		BLMX_HIP(hipSetDevice(h->ix.prm.device));
		hipStream_t s = reinterpret_cast<hipStream_t>(stream);  // NULL = HIP's null stream, like any other stream
		launch_route_scatter(h->ix, d_kmers, d_dest, nq, d_offsets, world, d_sorted, d_order, d_cursor, s);
	});
}

int blmx_unpermute_device(blmx_index* h, const int64_t* d_vals, const uint32_t* d_order, uint64_t nq, int64_t* d_out, void* stream)
{
	if (!h || (nq && (!d_vals || !d_order || !d_out)))
		return BLMX_EINVAL;
	return guarded(h, [&] {
		BLMX_HIP(hipSetDevice(h->ix.prm.device));
		hipStream_t s = reinterpret_cast<hipStream_t>(stream);  // NULL = HIP's null stream, like any other stream
		launch_unpermute(d_vals, d_order, nq, d_out, s);
	});
}

int blmx_partition_device(blmx_index* h, const void* d_kmers, uint64_t nq, void* d_sorted_kmers, uint32_t* d_inv, uint64_t* d_rank_counts,
                          void* stream)
{
	if (!h || (nq && (!d_kmers || !d_sorted_kmers || !d_inv)))
		return BLMX_EINVAL;
	return guarded(h, [&] {
		if (!h->ix.built)
			throw StatusError{BLMX_ESTATE, "index not built"};
		BLMX_HIP(hipSetDevice(h->ix.prm.device));
		launch_partition(h->ix, d_kmers, nq, d_sorted_kmers, d_inv, d_rank_counts, reinterpret_cast<hipStream_t>(stream));
	});
}

int blmx_query_partitioned_device(blmx_index* h, const void* d_sorted_kmers, uint64_t nq, int64_t* d_ids, void* stream)
{
	if (!h || (nq && (!d_sorted_kmers || !d_ids)))
		return BLMX_EINVAL;
	return guarded(h, [&] {
		if (!h->ix.built)
			throw StatusError{BLMX_ESTATE, "index not built"};
		BLMX_HIP(hipSetDevice(h->ix.prm.device));
		launch_query_partitioned(h->ix, d_sorted_kmers, nq, d_ids, reinterpret_cast<hipStream_t>(stream));
		h->ix.number_query.fetch_add(nq, std::memory_order_relaxed);
	});
}

int blmx_gather_ids_device(blmx_index* h, const int64_t* d_ids_sorted, const uint32_t* d_inv, uint64_t nq, int64_t* d_ids_out, void* stream)
{
	if (!h || (nq && (!d_ids_sorted || !d_inv || !d_ids_out)))
		return BLMX_EINVAL;
	return guarded(h, [&] {
		BLMX_HIP(hipSetDevice(h->ix.prm.device));
		launch_gather_ids(h->ix, d_ids_sorted, d_inv, nq, d_ids_out, reinterpret_cast<hipStream_t>(stream));
	});
}

int blmx_comm_unique_id(void* id_out)
{
	if (!id_out)
		return BLMX_EINVAL;
	return guarded(nullptr, [&] { comm_unique_id(id_out); });
}

int blmx_comm_create(const void* id, uint32_t rank, uint32_t world, int device, blmx_comm** out)
{
	if (out)
		*out = nullptr;
	if (!id || !out)
		return BLMX_EINVAL;
	return guarded(nullptr, [&] { *out = reinterpret_cast<blmx_comm*>(comm_create(id, rank, world, device)); });
}

void blmx_comm_destroy(blmx_comm* c) { comm_destroy(reinterpret_cast<Comm*>(c)); }

int blmx_comm_loopback_id(void* id_out)
{
	if (!id_out)
		return BLMX_EINVAL;
	return guarded(nullptr, [&] { comm_loopback_id(id_out); });
}

int blmx_comm_set_tuning(blmx_comm* c, const char* key, int64_t value)
{
	if (!c || !key)
		return BLMX_EINVAL;
	return guarded(nullptr, [&] { comm_set_tuning(*reinterpret_cast<Comm*>(c), key, value); });
}

const char* blmx_comm_transport(const blmx_comm* c) { return c ? comm_transport(reinterpret_cast<const Comm*>(c)) : ""; }

int blmx_query_kmers_sharded_device(blmx_index* h, blmx_comm* c, const void* d_kmers, uint64_t nq, int64_t* d_ids, void* stream)
{
	if (!h || !c || (nq && (!d_kmers || !d_ids)))
		return BLMX_EINVAL;
	return guarded(h, [&] { query_kmers_sharded(h->ix, *reinterpret_cast<Comm*>(c), d_kmers, nullptr, nq, d_ids, reinterpret_cast<hipStream_t>(stream)); });
}

int blmx_build_sharded_from_sequences(blmx_index* h, blmx_comm* c, const char* concat, const uint64_t* offsets, uint64_t nseq)
{
	if (!h || !c || (nseq && (!concat || !offsets)))
		return BLMX_EINVAL;
	return guarded(h, [&] {
		// a rank whose staging fails (bad offsets, out of memory) still joins the collective, with a failure vote: nobody waits for it
		bool entered = false;
		try {
			with_staged_sequences(h->ix, concat, offsets, nseq, [&](const uint8_t* a, const uint64_t* o, uint64_t total) {
				entered = true;
				build_sharded(h->ix, *reinterpret_cast<Comm*>(c), a, o, nseq, total);
			});
		} catch (const StatusError& e) {
			if (entered)
				throw;
			build_sharded(h->ix, *reinterpret_cast<Comm*>(c), nullptr, nullptr, 0, 0, &e);
			throw;
		}
	});
}

int blmx_build_sharded_from_fasta(blmx_index* h, blmx_comm* c, const char* path)
{
	if (!h || !c || !path)
		return BLMX_EINVAL;
	return guarded(h, [&] { build_sharded_from_fasta(h->ix, *reinterpret_cast<Comm*>(c), path); });
}

int blmx_query_kmers_sharded(blmx_index* h, blmx_comm* c, const void* kmers, uint64_t nq, int64_t* ids_out)
{
	if (!h || !c || (nq && (!kmers || !ids_out)))
		return BLMX_EINVAL;
	return guarded(h, [&] { query_kmers_sharded_host(h->ix, *reinterpret_cast<Comm*>(c), kmers, nq, ids_out); });
}

int blmx_query_sequences_sharded(blmx_index* h, blmx_comm* c, const char* concat, const uint64_t* offsets, uint64_t nseq, int64_t* ids_out)
{
	if (!h || !c || (nseq && (!concat || !offsets)))
		return BLMX_EINVAL;
	return guarded(h, [&] { query_sequences_sharded_host(h->ix, *reinterpret_cast<Comm*>(c), concat, offsets, nseq, ids_out); });
}

int blmx_query_sequences_sharded_device(blmx_index* h, blmx_comm* c, const void* d_concat, const uint64_t* d_offsets, uint64_t nseq, uint64_t total_len,
                                        uint64_t nk, int64_t* d_ids, int* d_err_flag, void* stream)
{
	if (!h || !c || (nseq && (!d_concat || !d_offsets)) || (nk && !d_ids))
		return BLMX_EINVAL;
	return guarded(h, [&] {
		query_sequences_sharded_device(h->ix, *reinterpret_cast<Comm*>(c), reinterpret_cast<const uint8_t*>(d_concat), d_offsets, nseq, total_len, nk,
		                               d_ids, d_err_flag, reinterpret_cast<hipStream_t>(stream));
	});
}

int blmx_build_sharded_from_device_sequences(blmx_index* h, blmx_comm* c, const void* d_concat, const uint64_t* d_offsets, uint64_t nseq,
                                             uint64_t total_len)
{
	if (!h || !c || (nseq && (!d_concat || !d_offsets)))
		return BLMX_EINVAL;
	return guarded(h, [&] {
		BLMX_HIP(hipSetDevice(h->ix.prm.device));
		BLMX_HIP(hipDeviceSynchronize());
		build_sharded(h->ix, *reinterpret_cast<Comm*>(c), reinterpret_cast<const uint8_t*>(d_concat), d_offsets, nseq, total_len);
	});
}

int blmx_shard_build_begin(blmx_index* h, const char* concat, const uint64_t* offsets, uint64_t nseq, uint64_t* hist_out)
{
	if (!h || !hist_out || (nseq && (!concat || !offsets)))
		return BLMX_EINVAL;
	return guarded(h, [&] {
		with_staged_sequences(h->ix, concat, offsets, nseq,
		                      [&](const uint8_t* a, const uint64_t* o, uint64_t total) { shard_build_begin(h->ix, a, o, nseq, total, hist_out); });
	});
}

int blmx_shard_build_begin_device(blmx_index* h, const void* d_concat, const uint64_t* d_offsets, uint64_t nseq, uint64_t total_len, uint64_t* hist_out)
{
	if (!h || !hist_out || (nseq && (!d_concat || !d_offsets)))
		return BLMX_EINVAL;
	return guarded(h, [&] {
		BLMX_HIP(hipSetDevice(h->ix.prm.device));
		BLMX_HIP(hipDeviceSynchronize());
		shard_build_begin(h->ix, reinterpret_cast<const uint8_t*>(d_concat), d_offsets, nseq, total_len, hist_out);
	});
}

int blmx_shard_build_plan(blmx_index* h, const uint64_t* global_hist, uint32_t world, uint64_t* send_sk, uint64_t* send_words, const void** d_meta,
                          const void** d_words)
{
	if (!h || !global_hist || !send_sk || !send_words || !d_meta || !d_words)
		return BLMX_EINVAL;
	return guarded(h, [&] {
		BLMX_HIP(hipSetDevice(h->ix.prm.device));
		shard_build_plan(h->ix, global_hist, world, send_sk, send_words, d_meta, d_words);
	});
}

uint64_t blmx_shard_halo_words(const blmx_index* h) { return h ? shard_halo_words(h->ix) : 0; }

int blmx_shard_halo_get(blmx_index* h, uint64_t* record_out)
{
	if (!h || !record_out)
		return BLMX_EINVAL;
	return guarded(h, [&] {
		BLMX_HIP(hipSetDevice(h->ix.prm.device));
		shard_halo_get(h->ix, record_out);
	});
}

int blmx_shard_halo_set(blmx_index* h, const uint64_t* records, uint32_t world)
{
	if (!h || !records)
		return BLMX_EINVAL;
	return guarded(h, [&] {
		BLMX_HIP(hipSetDevice(h->ix.prm.device));
		shard_halo_set(h->ix, records, world);
	});
}

int blmx_shard_build_finish(blmx_index* h, const void* d_meta_in, const void* d_words_in, const uint64_t* recv_sk, const uint64_t* recv_words)
{
	if (!h || !recv_sk || !recv_words)
		return BLMX_EINVAL;
	return guarded(h, [&] {
		BLMX_HIP(hipSetDevice(h->ix.prm.device));
		BLMX_HIP(hipDeviceSynchronize());  // the caller's transport filled the incoming buffers
		shard_build_finish(h->ix, d_meta_in, d_words_in, recv_sk, recv_words);
		h->ix.built = true;
	});
}

int blmx_set_query_binning(blmx_index* h, int policy)
{
	if (!h)
		return BLMX_EINVAL;
	h->ix.bin_policy = policy < 0 ? -1 : (policy > 0 ? 1 : 0);
	return BLMX_OK;
}

int blmx_set_tuning(blmx_index* h, const char* key, int value)
{
	if (!h || !key)
		return BLMX_EINVAL;
	Index& ix = h->ix;
	const std::string k(key);
	if (k == "bin_bits")
		ix.tune_bin_bits = value;
	else if (k == "filter_bits")  // bits per key of the absent-key filter built at construction (0 = none); set before building
		ix.tune_filter_bits = value < 0 ? -1 : (value > 32 ? 32 : value);  // -1 = automatic
	else if (k == "pos_hint")  // before building: 0 = position fields as wide as the reference's (no sub-window hint), 1 = hint at b >= 7, 2 = at b = 6 too
		ix.tune_pos_hint = value < 0 ? 0 : (int)std::min<int64_t>(value, 2);
	else if (k == "filter_gate")  // 0: large batches never switch the filter off for their successors
		ix.tune_filter_gate = value != 0;
	else if (k == "filter_junctions")  // test hook, before building
		ix.tune_filter_junctions = value != 0;
	else if (k == "absent_filter") {  // 0 = queries ignore the filter the index has; 1 = use it (waits for its background pass)
		ix.tune_absent_filter = value != 0;
		if (value)
			ix.filter_wait();
	}
	else if (k == "mini_handoff")
		ix.tune_mini_handoff = value;
	else if (k == "query_heads")
		ix.tune_query_heads = value;
	else if (k == "reads_mode")
		ix.tune_reads_mode = value;
	else if (k == "partition")
		ix.tune_partition = value;
	else if (k == "slot_slack_pct")
		ix.tune_slot_slack_pct = value;
	else if (k == "narrow_ids")
		ix.tune_narrow_ids = value;
	else if (k == "window_kb")
		ix.tune_window_kb = value;
	else if (k == "slot_piece_min")
		ix.tune_slot_piece_min = value;
	else if (k == "serve_burst") {  // 0: small calls take one launch per round of the combiner (r03); 1: the bounded serving kernel
		if (!value)
			burst_stop(ix);
		ix.tune_serve_burst = value != 0;
	} else if (k == "fb_table")  // before building / loading: 0 = no hash tables over the fallback entries (lookups search the sorted arrays)
		ix.tune_fb_table = value != 0;
	else if (k == "bin_rounds")  // the one-pass partition kernel stages a tile in this many pieces: 2 (two workgroups per CU) or 1
		ix.tune_bin_rounds = value == 1 ? 1 : 2;
	else if (k == "max_batch")  // test hook: device-resident batches are answered in pieces of this many k-mers (0 = 2^30)
		ix.tune_max_batch = (int)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 30));
	else if (k == "test_ovf_cap")  // test hook: slots of the one-pass partition's overflow region (0 = a whole batch)
		ix.tune_test_ovf_cap = (int)value;
	else if (k == "fat_lines")  // before building: 0 = 16-B lines + filter array, 1 = fat level-0 lines, 2 = level-0 records, -1 = automatic
		ix.tune_fat_lines = value < 0 ? -1 : (int)std::min<int64_t>(value, 2);
	else if (k == "build_passes")  // before building: 0 = automatic
		ix.tune_build_passes = value < 0 ? 0 : (value > 64 ? 64 : value);
	else if (k == "query_binning")
		ix.bin_policy = value < 0 ? -1 : (value > 0 ? 1 : 0);
	else
		return fail(h, BLMX_EINVAL, "unknown tuning key");
	return BLMX_OK;
}

int blmx_serve_stats(blmx_index* h, uint64_t* out4)
{
	if (!h || !out4)
		return BLMX_EINVAL;
	burst_stats(h->ix, out4);
	return BLMX_OK;
}

int blmx_partition_stats(blmx_index* h, uint64_t* out4, int reset)
{
	if (!h || !out4)
		return BLMX_EINVAL;
	return guarded(h, [&] {
		BLMX_HIP(hipSetDevice(h->ix.prm.device));
		out4[0] = out4[1] = out4[2] = out4[3] = 0;
		if (!h->ix.b_pstats.p)
			return;
		BLMX_HIP(hipDeviceSynchronize());  // the counters are written on the callers' streams
		BLMX_HIP(hipMemcpy(out4, h->ix.b_pstats.p, 32, hipMemcpyDeviceToHost));
		if (reset)
			BLMX_HIP(hipMemset(h->ix.b_pstats.p, 0, 32));
	});
}

int blmx_release_scratch(uint64_t* device_bytes, uint64_t* pinned_bytes)
{
	(void)hipDeviceSynchronize();
	const uint64_t d = ScratchCache::get().release_all(), p = release_pinned_cache();
	if (device_bytes)
		*device_bytes = d;
	if (pinned_bytes)
		*pinned_bytes = p;
	return BLMX_OK;
}

int blmx_stats(const blmx_index* h, blmx_stats_t* out)
{
	if (!h || !out)
		return BLMX_EINVAL;
	*out = h->ix.st;
	out->number_query = h->ix.number_query.load(std::memory_order_relaxed) + burst_number_query(h->ix);
	out->nbuckets = h->ix.nbuckets;
	out->ngroups = h->ix.ngroups;
	out->k = h->ix.prm.k;
	out->m = h->ix.prm.m;
	out->log2_mphf = h->ix.prm.log2_mphf;
	out->bits_saved = h->ix.prm.bits_saved;
	out->kmer_bytes = h->ix.prm.kmer_bytes;
	out->shard_rank = h->ix.shard_rank;
	out->shard_world = h->ix.shard_world;
	return BLMX_OK;
}

int64_t blmx_export(blmx_index* h, int which, void* out, uint64_t cap)
{
	if (!h || !h->ix.built)
		return BLMX_ESTATE;
	Index& ix = h->ix;
	int64_t result = -1;
	const int rc = guarded(h, [&] {
		BLMX_HIP(hipSetDevice(ix.prm.device));
		hipStream_t s = ix.stream;
		auto copy_out = [&](const void* src, size_t elem, uint64_t avail) {
			result = (int64_t)avail;
			if (out && avail)
				memcpy(out, src, elem * (size_t)std::min<uint64_t>(avail, cap));
		};
		auto dev_out = [&](const void* dsrc, size_t elem, uint64_t avail) {
			result = (int64_t)avail;
			const uint64_t n = std::min<uint64_t>(avail, cap);
			if (out && n) {
				BLMX_HIP(hipMemcpyAsync(out, dsrc, elem * n, hipMemcpyDeviceToHost, s));
				BLMX_HIP(hipStreamSynchronize(s));
			}
		};
		const uint32_t ng = ix.ngroups;
		switch (which) {
		case 0:
		case 1: {
			std::vector<BucketDesc> hb(ix.nbuckets);
			BLMX_HIP(hipMemcpyAsync(hb.data(), ix.buckets.p, hb.size() * sizeof(BucketDesc), hipMemcpyDeviceToHost, s));
			BLMX_HIP(hipStreamSynchronize(s));
			if (which == 0) {
				std::vector<uint32_t> v(ix.nbuckets);
				for (uint32_t i = 0; i < ix.nbuckets; i++)
					v[i] = hb[i].nt;
				copy_out(v.data(), 4, v.size());
			} else {
				std::vector<uint64_t> v(ix.nbuckets);
				for (uint32_t i = 0; i < ix.nbuckets; i++)
					v[i] = hb[i].start;
				copy_out(v.data(), 8, v.size());
			}
			break;
		}
		case 2:
			dev_out(ix.seq.p, 8, (ix.st.total_nt + 31) / 32);
			break;
		case 3:
			if (ix.pos_hint == 0) {
				dev_out(ix.positions.p, 8, (ix.st.positions_total_size + 63) / 64);
			} else {  // the fields in HBM are wider than the reference's: repack to its layout
				DevBuf logical;
				positions_to_reference_layout(ix, logical, s);
				dev_out(logical.p, 8, (ix.st.positions_total_size + 63) / 64);
			}
			break;
		case 4:
		case 5:
		case 6:
		case 9: {
			std::vector<uint64_t> v(ng + (which == 9 ? 1 : 0));
			uint64_t ref_posstart = 0;  // the reference's offsets: its own field width (nbits without the hint bits) + 8 per group
			for (uint32_t g = 0; g < ng; g++) {
				v[g] = which == 4 ? ix.h_groups[g].nkeys : which == 5 ? ix.h_groups[g].prefix : which == 6 ? ref_posstart : ix.h_groups[g].wbase;
				ref_posstart += (uint64_t)(ix.h_groups[g].nbits - ix.pos_hint) * ix.h_groups[g].nkeys + 8;
			}
			if (which == 9)
				v[ng] = ix.total_logical_words;
			copy_out(v.data(), 8, v.size());
			break;
		}
		case 7: {
			std::vector<uint32_t> v(ng);
			for (uint32_t g = 0; g < ng; g++)
				v[g] = ix.h_groups[g].nbits - ix.pos_hint;
			copy_out(v.data(), 4, v.size());
			break;
		}
		case 8: {
			std::vector<uint64_t> v((size_t)ng * kLevels);
			for (uint32_t g = 0; g < ng; g++)
				for (int l = 0; l < kLevels; l++)
					v[(size_t)g * kLevels + l] = ix.h_groups[g].nkeys ? ix.h_groups[g].size[l] : 0;
			copy_out(v.data(), 8, v.size());
			break;
		}
		case 10: {  // logical level bit arrays: strip the rank word of every 16-B line
			std::vector<uint32_t> hl((size_t)ix.total_lines * kLineWords32);
			if (!hl.empty()) {
				BLMX_HIP(hipMemcpyAsync(hl.data(), ix.lines.p, hl.size() * 4, hipMemcpyDeviceToHost, s));
				BLMX_HIP(hipStreamSynchronize(s));
			}
			std::vector<uint64_t> v((size_t)ix.total_logical_words);
			for (uint32_t g = 0; g < ng; g++) {
				const GroupDesc& gd = ix.h_groups[g];
				// fat layout: level 0 in 64-B units of its own -- 96 bits each, or (records) 32
				const uint64_t w0 = gd.size[0] >> 5, per = ix.rec_lines ? 1 : kLineData32, F = (w0 + per - 1) / per;
				for (uint64_t w32 = 0; w32 < 2 * gd.nwords; w32++) {
					uint64_t unit = gd.line_base + w32 / kLineData32, slot = w32 % kLineData32;
					if (ix.fat_lines) {  // (cross-paired 128-B blocks of two units)
						const uint64_t u = w32 / per;
						unit = w32 < w0 ? gd.line_base + 8 * (u >> 1) + 4 * (u & 1) : gd.line_base + 8 * ((F + 1) / 2) + (w32 - w0) / kLineData32;
						slot = w32 < w0 ? w32 % per : (w32 - w0) % kLineData32;
					}
					const uint64_t x = hl[unit * kLineWords32 + 1 + slot];
					v[gd.wbase + (w32 >> 1)] |= x << (32 * (w32 & 1));
				}
			}
			copy_out(v.data(), 8, v.size());
			break;
		}
		case 11:
			dev_out(ix.fb_keys.p, 8 * (size_t)ix.KW, ix.st.fallback_keys);
			break;
		case 12:
			dev_out(ix.fb_ranks.p, 8, ix.st.fallback_keys);
			break;
		case 13:
			copy_out(ix.h_fb_group.data(), 4, ix.h_fb_group.size());
			break;
		default:
			throw StatusError{BLMX_EINVAL, "unknown export id"};
		}
	});
	return rc ? rc : result;
}

int blmx_primitive(int op, int kmer_bytes, const void* in, uint64_t n, int arg, int arg2, void* out)
{
	if (!in || !out || (kmer_bytes != 8 && kmer_bytes != 16) || op < 0 || op > 8 || (op == 7 && (arg2 < 0 || arg2 > kmer_bytes)) ||
	    (op == 8 && (arg < 1 || arg > 32 || n >= 0xffffffffull)))
		return BLMX_EINVAL;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
		return BLMX_EHIP;
	return guarded(nullptr, [&] {
		const size_t in_elem = (op <= 2 || op == 8) ? 4 : (op == 3 || op == 4 || op == 7) ? (size_t)kmer_bytes : (op == 5 ? 8 : 1);
		const size_t out_elem = (op == 0 || op == 1 || op == 2 || op == 4) ? 4
		                        : (op == 3 ? (size_t)kmer_bytes : (op == 5 ? 8 * kLevels : (op == 7 ? 16 : (op == 8 ? 8 : 1))));
		DevBuf d_in, d_out;
		BLMX_HIP(d_in.alloc(n * in_elem + 16, false, nullptr));
		BLMX_HIP(d_out.alloc(n * out_elem + 16, false, nullptr));
		BLMX_HIP(hipMemcpy(d_in.p, in, n * in_elem, hipMemcpyHostToDevice));
		if (op == 8)
			launch_sort_primitive(d_in.as<uint32_t>(), n, (unsigned)arg, d_out.as<uint32_t>(), nullptr);
		else
			launch_primitive(op, kmer_bytes / 8, d_in.p, n, arg, arg2, d_out.p, nullptr);
		BLMX_HIP(hipDeviceSynchronize());
		BLMX_HIP(hipMemcpy(out, d_out.p, n * out_elem, hipMemcpyDeviceToHost));
	});
}

}  // extern "C"
