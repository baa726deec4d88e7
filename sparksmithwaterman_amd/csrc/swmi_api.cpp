// swmi_api.cpp -- host runtime behind the C ABI of include/swmi.h.
//
// Owns device memory, the HIP stream, batching/chunking, the overflow re-runs and the
// host-side assembly of results (the part of the reference that builds Java Strings from
// the traceback stack, src/sw/SmithWaterman.java:418-431, and MapRef's aggregation,
// src/sw/Distribution.java:403-436).  All DP arithmetic and every traceback step run in
// the gfx950 kernels of swmi_kernels.hip: there is no CPU implementation of the
// algorithm in this library, and every entry point fails if no GPU is usable.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <unistd.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <unordered_map>
#include <deque>
#include <vector>

#include "../../include/swmi.h"
#include "swmi_device.h"
#include "swmi_io_internal.h"

extern "C" hipError_t swmi_launch_fill(const FillArgs *a, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop);
extern "C" hipError_t swmi_launch_traceback(const TraceArgs *a, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop);
extern "C" hipError_t swmi_launch_traceback_split(const TraceArgs *a, uint32_t n_windows, hipStream_t st);
extern "C" hipError_t swmi_launch_resident(const TraceArgs *a, const ResidentArgs *x, hipStream_t st);
extern "C" hipError_t swmi_launch_tfused(const TraceArgs *a, const TFusedArgs *x, hipStream_t st);
extern "C" hipError_t swmi_launch_encode(const uint8_t *raw, const uint64_t *raw_off, SeqDesc *desc, uint32_t *seqw,
                                         const uint8_t *lut, uint32_t n_seq, hipStream_t st);

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

int swmi_io_fail(int code, const std::string &msg) { return fail(code, "%s", msg.c_str()); }

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? SWMI_ERR_NOMEM : SWMI_ERR_HIP,            \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                       \
    } while (0)

// ------------------------------------------------------------------------------------------
// device buffer with capacity (grow-only)
// ------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    // Allocations leave headroom (x 1.5 for workspaces of 256 MiB and more, x 1.25 when growing): a stream's chunks differ by a
    // few per cent, and freeing and re-allocating a 20 GB workspace for every new largest chunk stalled every slot of a stream
    // for 1.5-3 s each time (hipFree / hipMalloc hold a device-wide lock; profiles/r03/config3_host_breakdown.txt).
    int reserve(size_t bytes) {
        if (bytes <= cap) return SWMI_OK;
        size_t want = bytes >= (256u << 20) ? bytes + bytes / 2 : bytes;      // (big workspaces: the first allocation already leaves room)
        if (p) { want = std::max(want, cap + cap / 4); (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess && want > bytes) { want = bytes; e = hipMalloc(&p, want); }
        if (e != hipSuccess) { p = nullptr; return fail(SWMI_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
        cap = want;
        return SWMI_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return (T *)p; }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }                 // (a buffer added to a struct later cannot be forgotten by its free function)
};

struct PinnedBuf {
    void *p = nullptr;      // host address
    void *dp = nullptr;     // the same memory as the GPU addresses it (zero-copy results)
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return SWMI_OK;
        if (p) { bytes = std::max(bytes, cap + cap / 4); (void)hipHostFree(p); p = nullptr; dp = nullptr; cap = 0; }      // (headroom: see DevBuf)
        hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocMapped);
        if (e != hipSuccess) { p = nullptr; return fail(SWMI_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
        e = hipHostGetDevicePointer(&dp, p, 0);
        if (e != hipSuccess) { (void)hipHostFree(p); p = nullptr; return fail(SWMI_ERR_HIP, "hipHostGetDevicePointer: %s", hipGetErrorString(e)); }
        cap = bytes;
        return SWMI_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; dp = nullptr; cap = 0; }
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf() { release(); }
};

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
struct swmi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    // options
    uint32_t cell_cap = 64;
    uint64_t max_workspace_bytes = 32ull << 30;
    int profiling = 0;
    uint32_t mode = 1;                      // requested pipeline (see swmi.h); mode 1 falls back to 2 for scores it cannot handle
    int zero_copy = 1;                      // kernels write results straight into pinned host memory (no D2H copy)
    uint64_t arena_words_per_pair = 160;    // first guess of the record arena (header + ops + two strings of a ~180-step alignment), grows on demand
    uint64_t arena_copy_wpp = 160;          // arena words per pair fetched with the first D2H (tracks the last run)
    uint64_t recs_per_pair_x16 = 32;        // first guess of the record table: entries per pair x 16, grows on demand
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    PinnedBuf h_err;                        // one host-mapped word the kernels raise on an internal failure (strip pipeline timeout)
    DevBuf d_hdr_ring;                      // arena headers of launches that run sw_tfused_kernel / sw_resident_pairs_kernel ONLY: a fresh zeroed
    uint32_t hdr_next = 0;                  // slot per launch, so that no kernel has to run first just to reset the bump pointer
    DevBuf d_lut;                           // 256-byte canonical-code table of the encode kernel
    int64_t spin_us = 2000;                 // how long a run polls its stream for completion before it blocks (a batch is sub-millisecond)
    uint32_t dbg_strip_spins = 0;           // test knob: spin budget of the strip pipeline (0 = default)
    uint32_t dbg_reverse_strips = 0;        // test knob: strip items dispatched consumer-first
    uint32_t auto_ties_x100 = 300;          // automatic traceback grain: split when a sampled pair has this many tied maxima (x 1/100) on average
    uint32_t col_chunks = 0;                // test knob: force this many column chunks per pair (0 = automatic)
    int tb_split = -1;                      // mode-1 traceback grain: -1 automatic, 0 one workgroup per pair, 1 one wavefront per window / alignment
    bool ext_events = false;                // SWMI_EXT_EVENTS=1: the plain two-kernel run is timed by the dispatches' own start/stop times (pure kernel
                                            // durations, as rocprofv3 shows them) -- measured 8-12 us per run DEARER than three hipEventRecord, so off
    int tfused = -1;                        // transposed sweep + traceback by one wavefront per pair (swmi_tfused.hip): -1 automatic, 0 never, 1 whenever a pair qualifies
    int resident = -1;                      // small pairs handled by one wavefront with the direction field in LDS: -1 automatic, 0 never, 1 whenever it fits
    int scores_only = 0;                    // 1: the sweep only -- every pair's score (and MapRef's totals), no tied-maximum lists, no alignments
    int stream_keep_records = 1;            // streams: 0 = a chunk's alignment records are dropped once its scores and counts are taken (a driver
                                            // that only needs totals and re-aligns its few winners, Distribution.java:341-353)
    int device_strings = 1;                 // the traceback kernels write both aligned strings behind every record (swmi_emit.h); 0: 2-bit ops only, strings built by the host
    bool cell_cap_set = false;              // cell_cap given by the caller (otherwise small launches get longer lists)
    // swmi_batch_run_async: one run in flight on the context's own host thread
    std::thread worker;
    std::mutex job_mu;
    std::condition_variable job_cv;
    std::atomic<int> job_state{0};          // 0 idle, 1 submitted, 2 finished, 3 quit
    swmi_batch *job_batch = nullptr;
    swmi_params job_params{};
    int job_rc = 0;
    std::string job_err;
};

// one alignment as parsed from the arena
struct HostAln {
    uint32_t rank;
    int32_t begin, end_i, end_j;
    uint32_t n_ops;
    const uint32_t *rec = nullptr;   // the alignment's payload in the arena: the two strings written by the kernels, or the packed ops
    int64_t str_id = -1;    // records without strings: >= 0 once the strings are built, offset of the reference-side string in swmi_batch::str_buf
};

struct PairRes {
    int32_t score = 0;
    uint32_t flags = 0;
    uint64_t n_cells = 0;
    uint64_t first = 0;     // index of the pair's first HostAln (ordered)
    uint64_t count = 0;     // alignment records present (0 when degenerate)
};

struct SiteRef { uint64_t pair; uint64_t k; int32_t begin; };

struct Work {            // one pair scheduled for a launch
    uint32_t pair;       // ref * n_reads + read
    uint64_t cells;      // m * n
    uint64_t dir_words;  // workspace dwords (direction field or checkpoints)
    uint64_t seam_words;
};

struct swmi_batch {
    uint32_t n_refs = 0, n_reads = 0;
    // original bytes (for string building: characters keep their case) and offsets
    std::vector<uint8_t> ref_bytes, read_bytes;
    // a streamed chunk keeps no copy of its references: their bytes are read again from the mapped file when an
    // alignment string is asked for (swmi_stream_push_file)
    const uint8_t *src_map = nullptr;
    std::vector<swmi_io_recpos> src_recs;
    std::unordered_map<uint32_t, std::vector<uint8_t>> src_cache;
    std::vector<uint64_t> ref_off, read_off;
    std::vector<SeqDesc> ref_desc, read_desc;
    // device
    DevBuf d_raw, d_raw_off;                // the caller's bytes as uploaded (input of the encode kernel) and their offsets
    DevBuf d_seqw, d_refs, d_reads, d_pairs, d_dir, d_seam, d_result, d_cells, d_cells_off, d_cells_cap, d_dbg, d_dbg2;
    DevBuf d_strip_items, d_progress;       // mode 1: strip-per-wavefront sweep of long reads
    DevBuf d_col_items;                     // mode 1: column chunks of single-strip pairs
    DevBuf d_win_off, d_queue;              // split traceback: per-pair window offsets, walk-item queue
    DevBuf d_tf_items;                      // pairs of the launch taken by sw_tfused_kernel
    DevBuf d_res_items;                     // resident pairs of the launch
    bool views_built = false;               // some MapRef view of the last run was built (they are reset by the next run)
    bool tb_split_used = false;             // the last run used the split traceback
    bool acgt_known = false;                // ref_desc/read_desc[].acgt fetched back from the device (set there by the encode kernel)
    PinnedBuf h_result;
    // per run
    swmi_params params{};
    bool has_run = false;
    int auto_choice = -1;                   // sampled pre-pass of the automatic traceback grain: 0 tie-heavy, 1 not, -1 not decided yet
    swmi_params auto_params{};
    std::vector<Work> work;                 // schedule (pairs sorted by work), valid for work_mode
    int work_mode = -1;
    bool work_tfused = false;               // the schedule's workspace sizes leave room for sw_tfused_kernel's column checkpoints
    uint32_t eff_mode = 1;                  // pipeline of the current run
    uint64_t work_cells = 0;
    std::vector<uint8_t> pairs_on_device;   // image of the PairDesc array currently in d_pairs
    const void *pairs_dev_ptr = nullptr;
    // what run_chunk derived for the chunk it prepared last: a repeated run of the same chunk with the same parameters
    // (bench.py's steps, a Spark job re-running a partition) skips the per-pair preparation altogether
    struct Prep {
        bool valid = false;
        size_t lo = 0, hi = 0;
        const void *work = nullptr;
        swmi_params params{};
        int mode = -1;
        uint64_t dir_words = 0, seam_words = 0;
        uint32_t max_path = 0, max_read = 0;
        size_t n_strip_items = 0, n_col_items = 0, n_strip_chunks = 0;
        uint64_t n_windows = 0, seam_priv_words = 0;
        size_t n_res = 0;
        uint32_t res_lds_words = 0, res_ops_words = 0;
        int resident_opt = -1;
        int tfused_opt = -1;
        size_t n_tf = 0;
        uint32_t tf_max_m = 0, tf_max_n = 0, tf_max_path = 0;
        bool exact = false, scores_only = false;
        uint32_t col_chunks_opt = 0;
        bool reverse_strips = false;
    } prep;
    std::vector<PairRes> pairs;             // by pair index
    // raw record streams of the last run (one per launch chunk), indexed lazily on the first alignment access
    // the record table entries (AlnRec, swmi_device.h) and the payload arenas of the launches, one RawChunk per launch
    struct RawChunk { size_t at, words; size_t tab_at, n_rec; size_t lo; std::vector<uint32_t> wpos; };   // wpos: re-run chunks only
    std::vector<uint32_t> raw;
    std::vector<AlnRec> rtab;
    // a run of ONE launch with results in pinned memory leaves its record stream where the kernels wrote it (the pinned block
    // is the batch's own and lives until the next run): copied into `raw` only when something needs it there
    const uint32_t *raw_ext = nullptr;
    const AlnRec *rtab_ext = nullptr;
    uint64_t raw_ext_records = 0, raw_ext_cap = 0;
    std::vector<RawChunk> raw_chunks;
    bool indexed = false;
    bool rec_strings = false;               // the records of the last run carry both aligned strings (option device_strings)
    bool scores_only = false;               // the last run computed scores only (option scores_only): no counts, no alignments
    bool records_dropped = false;           // a streamed chunk whose records were not kept (option stream_keep_records = 0)
    std::vector<HostAln> alns;              // grouped by pair, ordered as OptAlignments returns them
    std::vector<char> str_buf;              // every alignment's two NUL-terminated strings, at fixed offsets (str_at)
    std::vector<uint64_t> str_at;           // per alignment: offset of its reference-side string; the read side follows it
    // MapRef view cache
    std::vector<int8_t> ref_view_ready;
    std::vector<std::vector<SiteRef>> ref_sites;
    std::vector<uint64_t> ref_degenerate;   // leading (0,"","") sites per ref
    swmi_timing timing{};
};

static const uint8_t *code_table();

// ------------------------------------------------------------------------------------------
// library / context
// ------------------------------------------------------------------------------------------
extern "C" int swmi_abi_version(void) { return SWMI_ABI_VERSION; }

extern "C" const char *swmi_last_error(void) { return g_err.c_str(); }

extern "C" int swmi_device_count(int *count) {
    if (!count) return fail(SWMI_ERR_INVALID, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(SWMI_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return SWMI_OK;
}

extern "C" void swmi_default_params(swmi_params *p) {
    if (!p) return;
    p->match = 5; p->mismatch = -3; p->gap = -4;            // Distribution.java:36
    p->tie_mode = SWMI_TIE_SERIAL;
    p->types[0] = 'a'; p->types[1] = 'i'; p->types[2] = 'd'; p->types[3] = '-';   // Distribution.java:37
}

static void ctx_release(swmi_ctx *c) {
    for (auto &ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    c->h_err.release();
    c->d_lut.release();
    c->d_hdr_ring.release();
    delete c;
}

extern "C" int swmi_create(int device, swmi_ctx **out) {
    if (!out) return fail(SWMI_ERR_INVALID, "out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(SWMI_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= n) return fail(SWMI_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return fail(SWMI_ERR_NO_DEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SWMI_ERR_NO_DEVICE, "device %d is %s; the kernels are built for gfx950 (MI355X) only",
                    device, prop.gcnArchName);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(SWMI_ERR_NO_DEVICE, "hipSetDevice: %s", hipGetErrorString(e));
    // (no process-wide hipSetDeviceFlags: a run polls its own stream for `spin_us` before it blocks, see wait_stream)
    swmi_ctx *c = new swmi_ctx;
    c->device = device;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { c->stream = nullptr; ctx_release(c); return fail(SWMI_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    for (auto &ev : c->ev) {
        e = hipEventCreate(&ev);
        if (e != hipSuccess) { ev = nullptr; ctx_release(c); return fail(SWMI_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(e)); }
    }
    { int r = c->h_err.reserve(64); if (r) { ctx_release(c); return r; } }
    { static const char *ee = getenv("SWMI_EXT_EVENTS"); if (ee) c->ext_events = atoi(ee) != 0; }
    *(volatile uint32_t *)c->h_err.p = 0u;
    { int r = c->d_lut.reserve(256); if (r) { ctx_release(c); return r; } }
    e = hipMemcpy(c->d_lut.p, code_table(), 256, hipMemcpyHostToDevice);
    if (e != hipSuccess) { ctx_release(c); return fail(SWMI_ERR_HIP, "code table upload: %s", hipGetErrorString(e)); }
    *out = c;
    return SWMI_OK;
}

extern "C" void swmi_destroy(swmi_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->worker.joinable()) {
        {
            std::unique_lock<std::mutex> lk(ctx->job_mu);
            ctx->job_cv.wait(lk, [&] { return ctx->job_state.load() != 1; });      // a run still in flight
            ctx->job_state.store(3);
        }
        ctx->job_cv.notify_all();
        ctx->worker.join();
    }
    ctx_release(ctx);
}

extern "C" int swmi_set_option(swmi_ctx *ctx, const char *name, int64_t value) {
    if (!ctx || !name) return fail(SWMI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    if (!strcmp(name, "cell_cap")) {
        if (value < 1 || value > (1 << 20)) return fail(SWMI_ERR_INVALID, "cell_cap out of range");
        ctx->cell_cap = (uint32_t)value;
        ctx->cell_cap_set = true;
    } else if (!strcmp(name, "max_workspace_bytes")) {
        if (value < (1 << 20)) return fail(SWMI_ERR_INVALID, "max_workspace_bytes too small");
        ctx->max_workspace_bytes = (uint64_t)value;
    } else if (!strcmp(name, "mode")) {
        // -1 is the same as 1 (kept for callers that passed "automatic": mode 0 was measured and is never faster)
        if (value < -1 || value > 2) return fail(SWMI_ERR_INVALID, "mode must be -1 (automatic), 0, 1 or 2");
        ctx->mode = value < 0 ? 1u : (uint32_t)value;
    } else if (!strcmp(name, "auto_ties_x100")) {
        if (value < 100) return fail(SWMI_ERR_INVALID, "auto_ties_x100 out of range");
        ctx->auto_ties_x100 = (uint32_t)value;
    } else if (!strcmp(name, "spin_us")) {
        if (value < 0) return fail(SWMI_ERR_INVALID, "spin_us out of range");
        ctx->spin_us = value;
    } else if (!strcmp(name, "debug_strip_spins")) {
        ctx->dbg_strip_spins = (uint32_t)value;
    } else if (!strcmp(name, "debug_reverse_strips")) {
        ctx->dbg_reverse_strips = value != 0;
    } else if (!strcmp(name, "tfused")) {
        if (value < -1 || value > 1) return fail(SWMI_ERR_INVALID, "tfused must be -1 (automatic), 0 or 1");
        ctx->tfused = (int)value;
    } else if (!strcmp(name, "resident")) {
        if (value < -1 || value > 1) return fail(SWMI_ERR_INVALID, "resident must be -1 (automatic), 0 or 1");
        ctx->resident = (int)value;
    } else if (!strcmp(name, "scores_only")) {
        ctx->scores_only = value != 0;
    } else if (!strcmp(name, "stream_keep_records")) {
        ctx->stream_keep_records = value != 0;
    } else if (!strcmp(name, "device_strings")) {
        ctx->device_strings = value != 0;
    } else if (!strcmp(name, "tb_split")) {
        if (value < -1 || value > 1) return fail(SWMI_ERR_INVALID, "tb_split must be -1 (automatic), 0 or 1");
        ctx->tb_split = (int)value;
    } else if (!strcmp(name, "col_chunks")) {
        if (value < 0 || value > 4096) return fail(SWMI_ERR_INVALID, "col_chunks out of range");
        ctx->col_chunks = (uint32_t)value;
    } else if (!strcmp(name, "zero_copy")) {
        ctx->zero_copy = value != 0;
    } else if (!strcmp(name, "profiling")) {
        if (value < 0 || value > 2) return fail(SWMI_ERR_INVALID, "profiling must be 0, 1 (every stage) or 2 (the sweep only)");
        ctx->profiling = (int)value;
    } else if (!strcmp(name, "arena_words_per_pair")) {
        if (value < 1) return fail(SWMI_ERR_INVALID, "arena_words_per_pair out of range");
        ctx->arena_words_per_pair = (uint64_t)value;
    } else {
        return fail(SWMI_ERR_INVALID, "unknown option '%s'", name);
    }
    return SWMI_OK;
}

// ------------------------------------------------------------------------------------------
// sequence encoding
// ------------------------------------------------------------------------------------------
// Canonical base codes: Character.toUpperCase restricted to ISO-8859-1 input (SmithWaterman.java:311-312:
// a-z and 0xE0-0xFE except 0xF7 drop 0x20; 0xB5 and 0xFF map outside Latin-1 and only equal themselves) followed by a permutation of the byte values that puts the eight
// "fast" symbols A,C,G,T,N,U,R,Y on the codes 0,4,...,28 (their bit offsets in an 8 x int4 score profile), so that
// code(x) == code(y)  <=>  toUpperCase(x) == toUpperCase(y).  Sequences made of those symbols only (and scores within
// int4) run the v_dot8_i32_i4 cell stream -- a reference with N stretches stays on the fast path; any other byte alphabet
// runs the compare-and-select variant.
static const uint8_t *code_table() {
    static uint8_t T[256];
    static std::once_flag once;              // MapRef.call runs on every executor thread (Distribution.java:32,403)
    std::call_once(once, [] {
        uint8_t perm[256];
        for (int i = 0; i < 256; i++) perm[i] = (uint8_t)i;
        const uint8_t fast[8] = {'A', 'C', 'G', 'T', 'N', 'U', 'R', 'Y'};
        for (int k = 0; k < 8; k++) std::swap(perm[fast[k]], perm[4 * k]);   // -> 0,4,...,28
        for (int i = 0; i < 256; i++) {
            int u = i;
            if ((i >= 'a' && i <= 'z') || (i >= 0xE0 && i <= 0xFE && i != 0xF7)) u = i - 32;
            T[i] = perm[u];
        }
    });
    return T;
}

// Geometry of the byte images (swmi_device.h): every image 16-byte aligned and followed by SWMI_SEQ_PAD_WORDS zero
// dwords.  The bytes themselves are canonicalised on the GPU (sw_encode_kernel, swmi_prep.hip).
static uint64_t layout_sequences(const uint64_t *off, uint32_t n, uint64_t word0, std::vector<SeqDesc> &desc) {
    desc.resize(n);
    uint64_t at = word0;
    for (uint32_t s = 0; s < n; s++) {
        const uint64_t len = off[s + 1] - off[s];
        at = (at + 3) & ~(uint64_t)3;
        SeqDesc d{};
        d.len = (uint32_t)len;
        d.boff = (uint32_t)at;                 // (checked against 2^32 by the caller)
        d.acgt = 0;                            // set by the encode kernel
        desc[s] = d;
        at += (len + 3) / 4 + SWMI_SEQ_PAD_WORDS;
    }
    return at;
}

static int check_offsets(const uint64_t *off, uint32_t n, const char *what) {
    if (!off) return fail(SWMI_ERR_INVALID, "%s offsets are null", what);
    if (off[0] != 0) return fail(SWMI_ERR_INVALID, "%s offsets must start at 0", what);
    for (uint32_t k = 0; k < n; k++) {
        if (off[k + 1] < off[k]) return fail(SWMI_ERR_INVALID, "%s offsets decrease at %u", what, k);
        if (off[k + 1] - off[k] >= (1ull << 30))
            return fail(SWMI_ERR_UNSUPPORTED, "%s %u is longer than 2^30-1 bases", what, k);
    }
    return SWMI_OK;
}

// ------------------------------------------------------------------------------------------
// upload
// ------------------------------------------------------------------------------------------
extern "C" void swmi_batch_free(swmi_ctx *ctx, swmi_batch *b) {
    if (!b) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    b->d_raw.release(); b->d_raw_off.release();
    b->d_seqw.release(); b->d_refs.release(); b->d_reads.release(); b->d_pairs.release();
    b->d_dir.release(); b->d_seam.release(); b->d_result.release(); b->d_cells.release();
    b->d_cells_off.release(); b->d_cells_cap.release(); b->d_dbg.release(); b->d_dbg2.release();
    b->d_strip_items.release(); b->d_progress.release(); b->d_col_items.release(); b->d_win_off.release(); b->d_queue.release(); b->d_res_items.release();
    b->d_tf_items.release();
    b->h_result.release();
    delete b;
}

// Device side of an upload: geometry from the offsets already stored in the batch, the raw bytes H2D, and the
// canonical images written by sw_encode_kernel.  `ref_src` / `read_src` may be pinned (the streaming path) or pageable.
// Enqueued on `st` and synchronised before returning.
static int upload_device(swmi_ctx *ctx, swmi_batch *b, hipStream_t st, const uint8_t *ref_src, const uint8_t *read_src) {
    const uint32_t n_refs = b->n_refs, n_reads = b->n_reads;
    const uint64_t ref_total = b->ref_off[n_refs], read_total = b->read_off[n_reads];
    uint64_t words = layout_sequences(b->ref_off.data(), n_refs, 0, b->ref_desc);
    words = layout_sequences(b->read_off.data(), n_reads, words, b->read_desc);
    words = ((words + 3) & ~(uint64_t)3) + SWMI_SEQ_PAD_WORDS;
    if (words >= (1ull << 32)) return fail(SWMI_ERR_UNSUPPORTED, "sequence image exceeds 16 GiB");
    int rc;
    const uint64_t read_base = (ref_total + 15) & ~(uint64_t)15;
    if ((rc = b->d_seqw.reserve(words * 4))) return rc;
    if ((rc = b->d_raw.reserve(read_base + read_total + 16))) return rc;
    if ((rc = b->d_raw_off.reserve(((uint64_t)n_refs + n_reads + 2) * 8))) return rc;
    if ((rc = b->d_refs.reserve(std::max<size_t>(n_refs, 1) * sizeof(SeqDesc)))) return rc;
    if ((rc = b->d_reads.reserve(std::max<size_t>(n_reads, 1) * sizeof(SeqDesc)))) return rc;
    HIP_TRY(hipMemsetAsync(b->d_seqw.p, 0, words * 4, st));
    uint8_t *raw = b->d_raw.as<uint8_t>();
    uint64_t *roff = b->d_raw_off.as<uint64_t>();
    if (ref_total) HIP_TRY(hipMemcpyAsync(raw, ref_src, ref_total, hipMemcpyHostToDevice, st));
    if (read_total) HIP_TRY(hipMemcpyAsync(raw + read_base, read_src, read_total, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(roff, b->ref_off.data(), ((size_t)n_refs + 1) * 8, hipMemcpyHostToDevice, st));
    // (the reads' offsets are stored absolute -- from the start of `raw` -- so that one base pointer serves both sides)
    std::vector<uint64_t> read_abs(b->read_off);
    for (auto &o : read_abs) o += read_base;
    HIP_TRY(hipMemcpyAsync(roff + n_refs + 1, read_abs.data(), ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice, st));
    if (n_refs) HIP_TRY(hipMemcpyAsync(b->d_refs.p, b->ref_desc.data(), n_refs * sizeof(SeqDesc), hipMemcpyHostToDevice, st));
    if (n_reads) HIP_TRY(hipMemcpyAsync(b->d_reads.p, b->read_desc.data(), n_reads * sizeof(SeqDesc), hipMemcpyHostToDevice, st));
    HIP_TRY(swmi_launch_encode(raw, roff, b->d_refs.as<SeqDesc>(), b->d_seqw.as<uint32_t>(), ctx->d_lut.as<uint8_t>(), n_refs, st));
    HIP_TRY(swmi_launch_encode(raw, roff + n_refs + 1, b->d_reads.as<SeqDesc>(), b->d_seqw.as<uint32_t>(),
                               ctx->d_lut.as<uint8_t>(), n_reads, st));
    HIP_TRY(hipStreamSynchronize(st));
    // a new set of sequences invalidates everything derived from the old one
    b->work_mode = -1; b->prep.valid = false; b->pairs_dev_ptr = nullptr; b->pairs_on_device.clear();
    b->has_run = false; b->acgt_known = false; b->auto_choice = -1;
    return SWMI_OK;
}

extern "C" int swmi_batch_upload(swmi_ctx *ctx,
                                 const uint8_t *ref_bytes, const uint64_t *ref_off, uint32_t n_refs,
                                 const uint8_t *read_bytes, const uint64_t *read_off, uint32_t n_reads,
                                 swmi_batch **out) {
    if (!ctx || !out) return fail(SWMI_ERR_INVALID, "null argument");
    *out = nullptr;
    int rc;
    if ((rc = check_offsets(ref_off, n_refs, "reference"))) return rc;
    if ((rc = check_offsets(read_off, n_reads, "read"))) return rc;
    if ((n_refs && ref_off[n_refs] && !ref_bytes) || (n_reads && read_off[n_reads] && !read_bytes))
        return fail(SWMI_ERR_INVALID, "sequence bytes are null");
    if ((uint64_t)n_refs * n_reads >= (1ull << 32))
        return fail(SWMI_ERR_UNSUPPORTED, "more than 2^32-1 pairs in one batch");
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));

    std::unique_ptr<swmi_batch> b(new swmi_batch);
    b->n_refs = n_refs; b->n_reads = n_reads;
    b->ref_off.assign(ref_off, ref_off + n_refs + 1);
    b->read_off.assign(read_off, read_off + n_reads + 1);
    // the caller's buffers are only valid during this call: the original bytes are kept for the alignment strings
    // (characters keep their case, SmithWaterman.java:388-406)
    b->ref_bytes.assign(ref_bytes, ref_bytes + ref_off[n_refs]);
    b->read_bytes.assign(read_bytes, read_bytes + read_off[n_reads]);
    if ((rc = upload_device(ctx, b.get(), ctx->stream, ref_bytes, read_bytes))) { swmi_batch_free(ctx, b.release()); return rc; }
    *out = b.release();
    return SWMI_OK;
}

// ------------------------------------------------------------------------------------------
// run
// ------------------------------------------------------------------------------------------
namespace {

struct RunState {
    swmi_ctx *ctx;
    swmi_batch *b;
    FillArgs fa{};
    TraceArgs ta{};
    float fill_ms = 0, tb_ms = 0, d2h_ms = 0;
    uint32_t launches = 0;
    double enqueue_us = 0, wait_us = 0, copyout_us = 0, prep_us = 0, prep_upload_us = 0;
    bool one_wave_sweep = false;            // the strip pipeline gave up once in this run: long reads are swept by one wavefront
    bool tb_split = false;                  // mode 1: detect per window + walk per alignment instead of one workgroup per pair
    bool defer_copy = false;                // this launch is the whole run: its records may stay in the pinned block
    bool keep = true;                       // the launch's records belong to the batch's results (false: the sampled pre-pass)
};

// layout of the device result block: [ArenaHdr | PairOut x np | arena words ...]
inline size_t result_out_off() { return 64; }
// layout of the result block: [ArenaHdr | PairOut x np | record table, tab_cap entries | arena words ...]
inline size_t result_tab_off(size_t np) { return (64 + np * sizeof(PairOut) + 255) & ~(size_t)255; }
inline size_t result_arena_off(size_t np, uint64_t tab_cap) { return (result_tab_off(np) + tab_cap * sizeof(AlnRec) + 255) & ~(size_t)255; }
// dwords of one alignment's payload: the two strings, n_ops / 4 + 1 dwords each, or the ops packed 16 per dword (swmi_emit.h)
inline uint64_t rec_words(uint32_t n_ops, bool strings) {
    return strings ? 2 * ((uint64_t)n_ops / 4 + 1) : ((uint64_t)n_ops + 15) / 16;
}

}  // namespace

#define SWMI_RES_CELL_CAP 128u            // maximum cells a resident pair lists in LDS (sw_resident_pairs_kernel)

// longest possible traceback of an n x m pair: A + I <= m rows, A + D <= n columns, and -- with match > 0 > gap -- the
// score match*A + gap*(I + D) must stay positive (`while (score > 0)`), which caps the gap moves
static uint64_t path_bound(uint64_t n_, uint64_t m_, const swmi_params &p) {
    uint64_t path = n_ + m_;
    if (p.match > 0 && p.gap < 0 && p.mismatch <= p.match) {
        const uint64_t g = (uint64_t)(-(int64_t)p.gap), mt = (uint64_t)p.match;
        path = std::min(path, std::min(m_ + std::min(n_, mt * m_ / g), n_ + std::min(m_, mt * n_ / g)));
    }
    return path;
}

// bytes of LDS a traceback workgroup needs for pairs whose longest path / read are given
// (swmi_kernels.hip: SWMI_TB_BLOCKS = 16 blocks per mode-0 tile, SWMI_TB_WAVES = 8, SWMI_TB_REFWIN_WORDS = 96)
static uint64_t traceback_lds_bytes(uint32_t mode, uint64_t max_path, uint64_t max_read) {
    const uint64_t lds_words = (max_path + 3) / 4 + 1, lds_read_words = (max_read + 3) / 4 + 1;
    const uint64_t win = (uint64_t)SWMI_RMAX * 64;
    const uint64_t tile_words = mode == 0 ? 4ull * 16 * win : (mode == 1 ? 32 + 8ull * SWMI_CK_BLOCKS * win : 4ull * SWMI_CK_BLOCKS * win);
    return 4ull * (tile_words + 4ull * (lds_words + lds_read_words + 96));
}

// What a launch of work[lo, hi) needs besides the sequences: pair descriptors with their workspace offsets, the item lists of
// the kernels that take only some of the pairs (strips of long reads, column chunks, resident pairs, transposed pairs), the
// window offsets of the split traceback, and the sizes everything downstream is dimensioned by.  Derived on the host, uploaded,
// and cached in swmi_batch::Prep: a repeated run of the same chunk with the same parameters (bench.py's steps, a Spark job
// re-running a partition) skips all of it.
static int prepare_chunk(RunState &rs, const std::vector<Work> &work, size_t lo, size_t hi, const std::vector<uint64_t> *cells_exact) {
    swmi_ctx *ctx = rs.ctx;
    swmi_batch *b = rs.b;
    const size_t np = hi - lo;
    const uint32_t n_reads = b->n_reads;
    int rc;

    // pair descriptors, direction-field and seam offsets
    std::vector<PairDesc> pd;
    std::vector<StripItem> strip_items;      // mode 1: (pair, column chunk, strip) of every read longer than one strip, one wavefront each
    uint64_t seam_priv_words = 0;            // ... private seam rows of the column chunks among them (behind the shared rows in d_seam)
    size_t n_strip_chunks = 0;               // ... and how many chunk sweeps (0: every multi-strip pair in one)
    std::vector<ColItem> col_items;          // mode 1: column chunks of single-strip pairs when the launch has few pairs
    uint64_t dir_words = 0, seam_words = 0;
    uint32_t max_path = 0, max_read = 0;
    size_t n_strip_items = 0, n_col_items = 0;
    uint64_t n_windows = 0;
    std::vector<uint32_t> win_off;           // split traceback: first window of every pair
    std::vector<uint32_t> res_items;         // pairs handled whole by sw_resident_pairs_kernel
    size_t n_res = 0;
    uint32_t res_lds_words = 0, res_ops_words = 0;
    std::vector<uint32_t> tf_items;          // pairs handled whole by sw_tfused_kernel (transposed sweep + traceback)
    size_t n_tf = 0;
    uint32_t tf_max_m = 0, tf_max_n = 0, tf_max_path = 0;
    const uint32_t res_cell_cap = SWMI_RES_CELL_CAP;
    swmi_batch::Prep &pr = b->prep;
    const auto p0 = std::chrono::steady_clock::now();
    const bool prepared = pr.valid && pr.lo == lo && pr.hi == hi && pr.work == (const void *)work.data() && pr.mode == b->eff_mode &&
                          memcmp(&pr.params, &b->params, sizeof(swmi_params)) == 0 && b->pairs_dev_ptr == b->d_pairs.p &&
                          b->pairs_on_device.size() == np * sizeof(PairDesc) && pr.col_chunks_opt == ctx->col_chunks &&
                          pr.reverse_strips == (ctx->dbg_reverse_strips != 0) && pr.resident_opt == ctx->resident &&
                          pr.tfused_opt == ctx->tfused && pr.exact == (cells_exact != nullptr) && pr.scores_only == (ctx->scores_only != 0);
    if (prepared) {
        dir_words = pr.dir_words; seam_words = pr.seam_words; max_path = pr.max_path; max_read = pr.max_read;
        n_strip_items = pr.n_strip_items; n_col_items = pr.n_col_items; n_windows = pr.n_windows;
        seam_priv_words = pr.seam_priv_words; n_strip_chunks = pr.n_strip_chunks;
        n_res = pr.n_res; res_lds_words = pr.res_lds_words; res_ops_words = pr.res_ops_words;
        n_tf = pr.n_tf; tf_max_m = pr.tf_max_m; tf_max_n = pr.tf_max_n; tf_max_path = pr.tf_max_path;
    } else {
    // Column chunks (swmi_device.h: ColItem): a launch of few pairs leaves most of the 1024 SIMDs idle while every pair is
    // one dependent chain of n + 63 steps.  A positive-score path spans at most m + match*m/|gap| columns (A <= m
    // alignment moves, and match*A + gap*D > 0 bounds the deletions D), so a wavefront that starts that far to the left of
    // a window computes the window exactly: the reference is cut into chunks of windows, one wavefront each.
    const swmi_params &P = b->params;
    const bool cols_possible = b->eff_mode == 1 && P.match > 0 && P.gap < 0 && P.mismatch <= 0 &&
                               P.match <= 7 && P.mismatch >= -8 && (np <= 512 || ctx->col_chunks > 1) && ctx->col_chunks != 1;
    // Resident pairs (sw_resident_pairs_kernel): a pair whose whole direction field fits a wavefront's share of LDS -- and
    // whose reference is short next to its read, so that the alignments cover most of the matrix anyway -- skips checkpoints
    // and window re-sweeps altogether: one wavefront sweeps it twice inside LDS and walks ALL its alignments at once, one per
    // lane.  (The exact-size re-run of pairs with more tied maxima than an LDS list holds takes the ordinary path.)
    // (measured, profiles/r02/sweeps_*.md: below ~300 pairs the launch is latency-bound and the split traceback, which
    // spreads a pair's windows and alignments over many wavefronts, is faster)
    const bool res_possible = b->eff_mode == 1 && !cells_exact && !ctx->scores_only && ctx->resident != 0 && (ctx->resident == 1 || np >= 256) &&
                              P.match <= 7 && P.match >= -8 && P.mismatch >= -8 && P.mismatch <= 0 && P.gap <= 0;
    auto res_need_words = [&](uint32_t m_, uint32_t n_, uint32_t &opw) -> uint64_t {
        const uint32_t R_ = swmi_rows_per_lane(m_), lact = (m_ + R_ - 1) / R_;
        const uint64_t nblk = ((uint64_t)n_ + lact - 1 + 15) / 16, n_ck = (nblk + SWMI_CK_BLOCKS - 1) / SWMI_CK_BLOCKS;
        opw = (uint32_t)((path_bound(n_, m_, P) + 15) / 16 + 1);
        return nblk * R_ * 64 + ((n_ck + 1) & ~1ull) + 2ull * res_cell_cap + 64ull * opw + (n_ + 3) / 4 + 1 + (m_ + 3) / 4 + 1 + 8 +
               128;                                           // (SWMI_EMIT_SCRATCH_WORDS: the string scratch of swmi_emit.h)
    };
    bool res_maybe = false;
    if (res_possible)
        for (size_t k = 0; k < np && !res_maybe; k++) {
            const uint32_t m_ = b->read_desc[work[lo + k].pair % n_reads].len, n_ = b->ref_desc[work[lo + k].pair / n_reads].len;
            uint32_t opw;
            res_maybe = m_ <= 64u * SWMI_RMAX && (ctx->resident == 1 || (uint64_t)n_ <= 8ull * m_) &&
                        res_need_words(m_, n_, opw) * 4 <= (ctx->resident == 1 ? 40u : 20u) * 1024u;
        }
    // Transposed, fused pairs (swmi_tfused.hip): reference columns on the lanes, the read streaming through, sweep AND
    // traceback in one launch -- for the usual pair (fast symbols, int4 scores, gap < 0, read <= 256, reference <= 2560).
    // Its sweep needs 20 % fewer instructions, but its traceback (one wavefront per block of a pair) does not beat the
    // workgroup-per-pair kernel: measured 0.170 ms against 0.088 + 0.071 at the headline, and slower for big batches (LDS and
    // registers hold it to one or two wavefronts per SIMD).  Kept as an option, tested in every GPU parity test; not chosen.
    const bool tf_possible = b->eff_mode == 1 && !cells_exact && !ctx->scores_only && ctx->tfused == 1 &&      // (-1, automatic: not chosen -- DESIGN.md 4.4)
                             P.match > 0 && P.match <= 7 && P.mismatch >= -8 && P.mismatch <= P.match && P.gap < 0 && P.gap >= -64;
    if ((cols_possible || res_maybe || tf_possible) && !b->acgt_known) {
        // the fast-symbol flags are derived on the device by the encode kernel
        if (b->n_refs) HIP_TRY(hipMemcpy(b->ref_desc.data(), b->d_refs.p, b->n_refs * sizeof(SeqDesc), hipMemcpyDeviceToHost));
        if (b->n_reads) HIP_TRY(hipMemcpy(b->read_desc.data(), b->d_reads.p, b->n_reads * sizeof(SeqDesc), hipMemcpyDeviceToHost));
        b->acgt_known = true;
    }
    const uint64_t chunk_budget = ctx->col_chunks > 1 ? ctx->col_chunks : std::max<uint64_t>(1, 1024 / std::max<size_t>(np, 1));
    pd.resize(np);
    if (b->eff_mode == 1) win_off.resize(np + 1);
    uint32_t pb_m = 0xFFFFFFFFu, pb_n = 0xFFFFFFFFu, pb_val = 0;
    for (size_t k = 0; k < np; k++) {
        const Work &w = work[lo + k];
        PairDesc d{};
        d.ref_id = w.pair / n_reads;
        d.read_id = w.pair % n_reads;
        d.out_id = (uint32_t)k;
        const uint32_t m_ = b->read_desc[d.read_id].len, n_ = b->ref_desc[d.ref_id].len;
        if (b->eff_mode == 1) {
            const uint64_t rps = 64ull * swmi_rows_per_lane(m_);
            const uint64_t wb = ((uint64_t)n_ + 63u + 15u) / 16u;
            win_off[k] = (uint32_t)std::min<uint64_t>(n_windows, 0xFFFFFFFFu);
            n_windows += ((m_ + rps - 1) / rps) * ((wb + SWMI_CK_BLOCKS - 1u) / SWMI_CK_BLOCKS);
        }
        uint32_t res_opw = 0;
        const bool res_fit = res_maybe && m_ <= 64u * SWMI_RMAX && b->read_desc[d.read_id].acgt && b->ref_desc[d.ref_id].acgt &&
                             (ctx->resident == 1 || (uint64_t)n_ <= 8ull * m_) &&
                             res_need_words(m_, n_, res_opw) * 4 <= (ctx->resident == 1 ? 40u : 20u) * 1024u;
        const bool tf_fit = !res_fit && tf_possible && m_ <= SWMI_TF_MAX_M && n_ <= 64u * SWMI_TF_BMAX &&
                            b->read_desc[d.read_id].acgt && b->ref_desc[d.ref_id].acgt && w.dir_words >= swmi_tf_ck_words(m_);
        if (tf_fit) {
            d.pad = SWMI_PAD_RESIDENT;                  // (every other kernel skips the pair)
            tf_items.push_back((uint32_t)k);
            tf_max_m = std::max(tf_max_m, m_); tf_max_n = std::max(tf_max_n, n_);
            tf_max_path = std::max<uint32_t>(tf_max_path, (uint32_t)path_bound(n_, m_, P));
        } else if (res_fit) {
            d.pad = SWMI_PAD_RESIDENT;
            res_items.push_back((uint32_t)k);
            uint32_t opw;
            res_lds_words = std::max<uint32_t>(res_lds_words, (uint32_t)res_need_words(m_, n_, opw));
            res_ops_words = std::max(res_ops_words, opw);
        } else if (b->eff_mode == 1 && m_ > 64u * SWMI_RMAX && strip_items.size() < (1u << 30)) {
            const uint32_t strips = (m_ + 64u * SWMI_RMAX - 1u) / (64u * SWMI_RMAX);
            d.pad = (uint32_t)strip_items.size();
            auto push_strips = [&](StripItem it) {
                it.pair = (uint32_t)k;
                it.prog = (uint32_t)strip_items.size();
                it.pad = 0;
                if (ctx->dbg_reverse_strips) for (uint32_t st = strips; st-- > 0;) { it.strip = st; strip_items.push_back(it); }
                else                         for (uint32_t st = 0; st < strips; st++) { it.strip = st; strip_items.push_back(it); }
            };
            // Column chunks of a multi-strip pair (swmi_device.h: StripItem): the halo argument holds for the whole read, so
            // a chunk is a strip pipeline of its own over [col0, end of its windows], all strips starting from zero.  The
            // halo is long (2.25 m columns at the default scores): a chunk's body may be as short as a quarter of it -- the
            // launch has SIMDs to spare, and what counts is the length of the longest chain.
            uint64_t chunks = 1;
            const uint64_t wblocks = ((uint64_t)n_ + 63u + 15u) / 16u;
            const uint32_t n_ck = (uint32_t)((wblocks + SWMI_CK_BLOCKS - 1u) / SWMI_CK_BLOCKS);
            const uint32_t step_w = 16u * SWMI_CK_BLOCKS;
            const uint64_t span = (uint64_t)m_ + (uint64_t)(P.match > 0 ? P.match : 0) * m_ / (uint64_t)(P.gap < 0 ? -(int64_t)P.gap : 1) + 1;
            if (cols_possible) {                                   // (any symbols: the strip kernel has both cell streams)
                const uint64_t budget = ctx->col_chunks > 1 ? ctx->col_chunks : chunk_budget / strips;
                chunks = std::min<uint64_t>(budget, n_ / std::max<uint64_t>((span + 64) / 4, 256));
                chunks = std::min<uint64_t>(chunks, n_ck);
                if (chunks < 2) chunks = 1;
            }
            if (chunks == 1) {
                StripItem it{};
                it.col0 = 0; it.g_lo = 0; it.g_hi = 0xFFFFFFFFu; it.priv_stride = 0; it.priv_off = 0;
                push_strips(it);
            } else {
                const uint32_t wpc = (uint32_t)((n_ck + chunks - 1) / chunks);   // windows per chunk
                for (uint32_t g_lo = 0; g_lo < n_ck; g_lo += wpc) {
                    StripItem it{};
                    it.g_lo = g_lo;
                    it.g_hi = std::min(g_lo + wpc, n_ck);
                    if (it.g_hi < n_ck && (uint64_t)it.g_hi * step_w > n_) it.g_hi = n_ck;
                    const int64_t c0 = (int64_t)g_lo * step_w - 64 - (int64_t)span - 1;
                    it.col0 = g_lo == 0 || c0 <= 0 ? 0u : (uint32_t)(c0 / step_w * step_w);
                    // strip 0 runs furthest: 64 steps per strip below it past the chunk's last window, or to the reference's end
                    const uint64_t ext0 = (uint64_t)it.g_hi * step_w - it.col0 + 64ull * (strips - 1u);
                    const uint64_t n0 = it.g_hi < n_ck ? std::min<uint64_t>(n_ - it.col0, ext0) : n_ - it.col0;
                    it.priv_stride = (uint32_t)((n0 + 1 + 15) & ~15ull);
                    it.priv_off = seam_priv_words;                                  // (+ the shared rows of the launch: below)
                    seam_priv_words += (uint64_t)(strips - 1u) * it.priv_stride;
                    if (it.g_hi == n_ck) it.g_hi = 0xFFFFFFFFu;                     // (the pair's last chunk: every window from g_lo on)
                    push_strips(it);
                    n_strip_chunks++;
                    if (it.g_hi == 0xFFFFFFFFu) break;
                }
            }
        } else if (cols_possible && chunk_budget > 1 && m_ <= 64u * SWMI_RMAX && b->read_desc[d.read_id].acgt && b->ref_desc[d.ref_id].acgt) {
            const uint64_t span = (uint64_t)m_ + (uint64_t)P.match * m_ / (uint64_t)(-(int64_t)P.gap) + 1;   // columns a path can span
            const uint64_t wblocks = ((uint64_t)n_ + 63u + 15u) / 16u;
            const uint32_t n_ck = (uint32_t)((wblocks + SWMI_CK_BLOCKS - 1u) / SWMI_CK_BLOCKS);
            const uint32_t step_w = 16u * SWMI_CK_BLOCKS;                       // anti-diagonal steps (= columns of lane 0) per window
            uint64_t chunks = std::min<uint64_t>(chunk_budget, n_ / std::max<uint64_t>(span + 64, 256));
            chunks = std::min<uint64_t>(chunks, n_ck);
            static const bool dbg_one = getenv("SWMI_DEBUG_ONE_CHUNK") != nullptr;      // diagnostics: every pair through the chunk kernel, one chunk
            if (dbg_one) chunks = 1;
            if (chunks >= 2 || dbg_one) {
                const uint32_t wpc = (uint32_t)((n_ck + chunks - 1) / chunks);   // windows per chunk
                const size_t first = col_items.size();
                for (uint32_t g_lo = 0; g_lo < n_ck; g_lo += wpc) {
                    ColItem ci;
                    ci.pair = (uint32_t)k;
                    ci.g_lo = g_lo;
                    ci.g_hi = std::min(g_lo + wpc, n_ck);
                    // a chunk that is not the last must end where lane 0 is still inside the reference
                    if (ci.g_hi < n_ck && (uint64_t)ci.g_hi * step_w > n_) ci.g_hi = n_ck;
                    const int64_t c0 = (int64_t)g_lo * step_w - 64 - (int64_t)span - 1;
                    ci.col0 = g_lo == 0 || c0 <= 0 ? 0u : (uint32_t)(c0 / step_w * step_w);   // (on a window boundary: checkpoints stay aligned)
                    col_items.push_back(ci);
                    if (ci.g_hi == n_ck) break;
                }
                d.pad = SWMI_PAD_COLS | (uint32_t)(col_items.size() - first);
            }
        }
        pd[k] = d;
        if (m_ != pb_m || n_ != pb_n) { pb_m = m_; pb_n = n_; pb_val = (uint32_t)path_bound(n_, m_, b->params); }   // (runs of equal lengths)
        max_path = std::max<uint32_t>(max_path, pb_val);
        max_read = std::max(max_read, m_);
    }
    n_strip_items = strip_items.size();
    n_col_items = col_items.size();
    n_res = res_items.size();
    n_tf = tf_items.size();
    // the kernel lays every wavefront's LDS out with the launch-wide ops_words: size the share for that
    res_lds_words = 0;
    for (uint32_t k : res_items) {
        uint32_t opw;
        const uint32_t m_ = b->read_desc[pd[k].read_id].len, n_ = b->ref_desc[pd[k].ref_id].len;
        const uint64_t own = res_need_words(m_, n_, opw);
        res_lds_words = std::max<uint32_t>(res_lds_words, (uint32_t)(own + 64ull * (res_ops_words - opw)));
    }
    if (16ull * res_lds_words > 160ull * 1024) {          // four such shares do not fit a CU's LDS: the ordinary path for all of them
        for (uint32_t k : res_items) pd[k].pad = 0;
        res_items.clear();
        n_res = 0; res_lds_words = 0; res_ops_words = 0;
    }
    // workspace offsets, now that every pair's kernel is known: a resident pair keeps everything in LDS and gets none
    // (40,000 pairs of 80 x 400 were charged 0.6 GB of checkpoints nobody writes)
    {
        std::vector<uint8_t> is_res(np, 0);
        for (uint32_t k : res_items) is_res[k] = 1;
        for (size_t k = 0; k < np; k++) {
            pd[k].dir_off = dir_words;
            pd[k].seam_off = seam_words;
            if (!is_res[k]) { dir_words += work[lo + k].dir_words; seam_words += work[lo + k].seam_words; }
        }
    }
    if (b->eff_mode == 1) win_off[np] = (uint32_t)std::min<uint64_t>(n_windows, 0xFFFFFFFFu);
    for (StripItem &it : strip_items)
        if (it.priv_stride) it.priv_off += seam_words;
    }
    const auto p1 = std::chrono::steady_clock::now();
    if (!prepared && b->eff_mode == 1) {
        if ((rc = b->d_win_off.reserve((np + 1) * sizeof(uint32_t)))) return rc;
        HIP_TRY(hipMemcpyAsync(b->d_win_off.p, win_off.data(), (np + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));      // win_off is a local
    }
    if ((rc = b->d_pairs.reserve(np * sizeof(PairDesc)))) return rc;
    if (!prepared && !strip_items.empty()) {
        if ((rc = b->d_strip_items.reserve(strip_items.size() * sizeof(StripItem)))) return rc;
        if ((rc = b->d_progress.reserve(strip_items.size() * sizeof(uint32_t)))) return rc;
        HIP_TRY(hipMemcpyAsync(b->d_strip_items.p, strip_items.data(), strip_items.size() * sizeof(StripItem), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));      // strip_items is a local
    }
    if (!prepared && !res_items.empty()) {
        if ((rc = b->d_res_items.reserve(res_items.size() * sizeof(uint32_t)))) return rc;
        HIP_TRY(hipMemcpyAsync(b->d_res_items.p, res_items.data(), res_items.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));      // res_items is a local
    }
    if (!prepared && !tf_items.empty()) {
        if ((rc = b->d_tf_items.reserve(tf_items.size() * sizeof(uint32_t)))) return rc;
        HIP_TRY(hipMemcpyAsync(b->d_tf_items.p, tf_items.data(), tf_items.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));      // tf_items is a local
    }
    if (!prepared && !col_items.empty()) {
        if ((rc = b->d_col_items.reserve(col_items.size() * sizeof(ColItem)))) return rc;
        HIP_TRY(hipMemcpyAsync(b->d_col_items.p, col_items.data(), col_items.size() * sizeof(ColItem), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));      // col_items is a local
    }
    if ((rc = b->d_dir.reserve(std::max<uint64_t>(dir_words, 1) * 4))) return rc;
    if ((rc = b->d_seam.reserve(std::max<uint64_t>(seam_words + seam_priv_words, 1) * 4))) return rc;
    // repeated runs of one batch schedule the same pairs: skip the H2D copy when nothing changed
    if (!prepared && (b->pairs_dev_ptr != b->d_pairs.p || b->pairs_on_device.size() != np * sizeof(PairDesc) ||
        memcmp(b->pairs_on_device.data(), pd.data(), np * sizeof(PairDesc)) != 0)) {
        HIP_TRY(hipMemcpyAsync(b->d_pairs.p, pd.data(), np * sizeof(PairDesc), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));      // pd is a local: the copy must be done before it dies
        b->pairs_on_device.assign((const uint8_t *)pd.data(), (const uint8_t *)pd.data() + np * sizeof(PairDesc));
        b->pairs_dev_ptr = b->d_pairs.p;
    }
    if (!prepared) {
        pr.valid = b->pairs_dev_ptr == b->d_pairs.p;
        pr.lo = lo; pr.hi = hi; pr.work = (const void *)work.data(); pr.params = b->params; pr.mode = b->eff_mode;
        pr.dir_words = dir_words; pr.seam_words = seam_words; pr.max_path = max_path; pr.max_read = max_read;
        pr.n_strip_items = n_strip_items; pr.n_col_items = n_col_items; pr.n_windows = n_windows;
        pr.seam_priv_words = seam_priv_words; pr.n_strip_chunks = n_strip_chunks;
        pr.n_res = n_res; pr.res_lds_words = res_lds_words; pr.res_ops_words = res_ops_words;
        pr.resident_opt = ctx->resident; pr.exact = cells_exact != nullptr;
        pr.tfused_opt = ctx->tfused; pr.n_tf = n_tf; pr.tf_max_m = tf_max_m; pr.tf_max_n = tf_max_n; pr.tf_max_path = tf_max_path;
        pr.col_chunks_opt = ctx->col_chunks; pr.reverse_strips = ctx->dbg_reverse_strips != 0;
        pr.scores_only = ctx->scores_only != 0;
    }
    rs.prep_us += std::chrono::duration<double, std::micro>(p1 - p0).count();
    rs.prep_upload_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - p1).count();
    return SWMI_OK;
}

// SWMI_DEBUG_FILL=1: where the time of the traceback (per-pair ticks, walk / staging shares, the slowest pairs) and of the
// sweep (ticks, wave placement by HW_ID) went.  Diagnostics only.  (The traceback's per-pair counters are compiled into the
// kernels only with `make KFLAGS=-DSWMI_TB_DIAG`: without it this prints zeros for them.)
static int dump_traceback_diagnostics(swmi_batch *b, const TraceArgs &ta, size_t np, size_t n_tf) {
    if (!ta.dbg) return SWMI_OK;
    std::vector<unsigned long long> d(np * 4);
    HIP_TRY(hipMemcpy(d.data(), ta.dbg, np * 32, hipMemcpyDeviceToHost));
    if (n_tf) {          // sw_tfused_kernel's own fields: {ticks, sweep | prologue << 32, replay << 16 | steps, replays | walk << 32}
        double t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0; unsigned long long mx = 0;
        for (size_t k = 0; k < np; k++) {
            t0 += d[4*k]; t1 += d[4*k+1] & 0xFFFFFFFFull; t2 += d[4*k+1] >> 32; t3 += d[4*k+2] >> 16; t4 += d[4*k+3] >> 32; t5 += d[4*k+3] & 0xFFFFFFFFull;
            mx = std::max(mx, d[4*k]);
        }
        fprintf(stderr, "[swmi tfused dbg] ticks per sweeper wavefront: lifetime mean=%.0f max=%llu = prologue %.0f + sweep %.0f + first block task %.0f (%.2f block tasks per wavefront) + waiting for tasks %.0f + rest (more tasks, walk items) %.0f\n",
                t0 / np, mx, t2 / np, t1 / np, t3 / np, t5 / np, t4 / np, (t0 - t1 - t2 - t3 - t4) / np);
        {   // the distribution of the wavefront lifetimes and the slowest ones
            std::vector<size_t> idx(np);
            for (size_t k = 0; k < np; k++) idx[k] = k;
            std::sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return d[4 * x] < d[4 * y]; });
            auto at = [&](double q) { return d[4 * idx[std::min<size_t>(np - 1, (size_t)(q * np))]]; };
            fprintf(stderr, "[swmi tfused dbg]   lifetime p10=%llu p50=%llu p90=%llu p99=%llu max=%llu\n", at(0.10), at(0.50), at(0.90), at(0.99), d[4 * idx[np - 1]]);
            for (size_t t = 0; t < std::min<size_t>(np, 6); t++) {
                const size_t k = idx[np - 1 - t];
                fprintf(stderr, "[swmi tfused dbg]   slow wavefront (pair %zu): lifetime %llu, sweep %llu, first block task %llu, block tasks taken %llu, waiting %llu, alignments of its pair %llu\n",
                        k, d[4 * k], d[4 * k + 1] & 0xFFFFFFFFull, d[4 * k + 2] >> 16, d[4 * k + 3] & 0xFFFFFFFFull, d[4 * k + 3] >> 32,
                        (unsigned long long)((const PairOut *)((const uint8_t *)b->h_result.p + result_out_off()))[k].n_cells);
            }
        }
    }
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0; unsigned long long mx = 0;
    double a4 = 0;
    for (size_t k = 0; k < np; k++) { a0 += d[4*k]; a1 += d[4*k+1] & 0xFFFFFFFFull; a2 += d[4*k+2] & 0xFFFF; a4 += d[4*k+2] >> 16; a3 += d[4*k+3] & 0xFFFFFFFFull; mx = std::max(mx, d[4*k]); }
    fprintf(stderr, "[swmi tb dbg] wave ticks mean=%.0f max=%llu; walk ticks mean=%.0f; staging ticks mean=%.0f; steps mean=%.1f; iterations mean=%.2f\n",
            a0 / np, mx, a1 / np, a4 / np, a2 / np, a3 / np);
    const PairOut *po_dbg = (const PairOut *)((const uint8_t *)b->h_result.p + result_out_off());
    double cs[4] = {0, 0, 0, 0}, cw[4] = {0, 0, 0, 0}; unsigned long long cm[4] = {0, 0, 0, 0}; size_t cn[4] = {0, 0, 0, 0};
    for (size_t k = 0; k < np; k++) {
        const size_t c = std::min<uint64_t>(po_dbg[k].n_cells, 4) - (po_dbg[k].n_cells ? 1 : 0);
        cs[c] += d[4 * k]; cw[c] += d[4 * k + 1] & 0xFFFFFFFFull; cm[c] = std::max(cm[c], d[4 * k]); cn[c]++;
    }
    {   // the slowest pairs: what makes the launch's tail
        std::vector<size_t> idx(np);
        for (size_t k = 0; k < np; k++) idx[k] = k;
        const size_t top = std::min<size_t>(np, 6);
        std::partial_sort(idx.begin(), idx.begin() + top, idx.end(), [&](size_t x, size_t y) { return d[4 * x] > d[4 * y]; });
        for (size_t t = 0; t < top; t++)
            fprintf(stderr, "[swmi tb dbg]   slow pair %zu: ticks=%llu walk=%llu staging=%llu in %llu stagings, steps=%llu iterations=%llu alignments=%llu\n", idx[t],
                    d[4 * idx[t]], d[4 * idx[t] + 1] & 0xFFFFFFFFull, d[4 * idx[t] + 2] >> 16, d[4 * idx[t] + 3] >> 32, d[4 * idx[t] + 2] & 0xFFFF,
                    d[4 * idx[t] + 3] & 0xFFFFFFFFull, (unsigned long long)po_dbg[idx[t]].n_cells);
        for (size_t t = 0; t < top; t++)
            fprintf(stderr, "[swmi tb dbg]     ... of the staging time of pair %zu, %llu ticks waiting for helpers\n", idx[t], d[4 * idx[t] + 1] >> 32);
    }
    for (int c = 0; c < 4; c++)
        if (cn[c]) fprintf(stderr, "[swmi tb dbg]   %d%s alignment(s): %zu pairs, wave ticks mean=%.0f max=%llu, walk(slot 0) mean=%.0f\n",
                           c + 1, c == 3 ? "+" : "", cn[c], cs[c] / cn[c], cm[c], cw[c] / cn[c]);
    return SWMI_OK;
}

static int dump_fill_diagnostics(swmi_batch *b, const FillArgs &fa, size_t np) {
    if (!fa.dbg) return SWMI_OK;
    std::vector<unsigned long long> d(np * 2);
    HIP_TRY(hipMemcpy(d.data(), fa.dbg, np * 16, hipMemcpyDeviceToHost));
#ifdef SWMI_STRIP_DIAG
    // strip pipeline (reads of two strips or more): see fill_pair
    for (size_t k = 0; k < np && k < 8; k++)
        fprintf(stderr, "[swmi strip dbg] pair %zu: strip 0 lifetime %llu ticks, %llu waiting before publications; strip 1 lifetime %llu, "
                "%llu polls, %llu ticks in polls, %llu waiting for seam groups\n", k, d[2 * k] >> 32, d[2 * k] & 0xFFFFFFFFull,
                d[2 * k + 1] >> 40, (d[2 * k + 1] >> 32) & 0xFF, (d[2 * k + 1] >> 16) & 0xFFFF, d[2 * k + 1] & 0xFFFF);
#endif
    unsigned long long ev = 0, cyc = 0, evmax = 0, cmax = 0, cmin = ~0ull;
    for (size_t k = 0; k < np; k++) {
        ev += d[2 * k]; cyc += d[2 * k + 1];
        evmax = std::max(evmax, d[2 * k]); cmax = std::max(cmax, d[2 * k + 1]); cmin = std::min(cmin, d[2 * k + 1]);
    }
    fprintf(stderr, "[swmi fill dbg] pairs=%zu slow-path entries mean=%.1f max=%llu; wave ticks mean=%.0f min=%llu max=%llu\n",
            np, (double)ev / np, evmax, (double)cyc / np, cmin, cmax);
    if (b->eff_mode == 1) {
        // the fast sweep stores HW_ID | XCC_ID << 32 instead of an event count: placement of the waves
        std::unordered_map<unsigned long long, int> per_simd, per_cu;
        for (size_t k = 0; k < np; k++) {
            const unsigned long long hw = d[2 * k] & 0xFFFFFFFFull, xcc = (d[2 * k] >> 32) & 0xF;
            const unsigned long long simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            const unsigned long long cukey = (xcc << 16) | (se << 8) | (sh << 4) | cu;
            per_cu[cukey]++; per_simd[(cukey << 4) | simd]++;
        }
        int h_simd[9] = {0}, h_cu[17] = {0};
        for (auto &kv : per_simd) h_simd[std::min(kv.second, 8)]++;
        for (auto &kv : per_cu) h_cu[std::min(kv.second, 16)]++;
        fprintf(stderr, "[swmi fill dbg] placement: %zu CUs, %zu SIMDs used; SIMDs by waves held: 1:%d 2:%d 3:%d 4+:%d; CUs by waves held: 1-4:%d 5-8:%d 9+:%d\n",
                per_cu.size(), per_simd.size(), h_simd[1], h_simd[2], h_simd[3], h_simd[4] + h_simd[5] + h_simd[6] + h_simd[7] + h_simd[8],
                h_cu[1] + h_cu[2] + h_cu[3] + h_cu[4], h_cu[5] + h_cu[6] + h_cu[7] + h_cu[8], h_cu[9] + h_cu[10] + h_cu[11] + h_cu[12] + h_cu[13] + h_cu[14] + h_cu[15] + h_cu[16]);
    }
    return SWMI_OK;
}

// The records of a finished launch become part of the batch's results.  The only launch of a run leaves its table and
// payloads in the pinned block (the batch's own until the next run): indexed there when something asks for an alignment.
// With several launches in one run each one's records are copied out of the block, which the next launch writes again; the
// table is dense, so its sequential read also tells how much of the arena is in use.
static int keep_chunk_records(RunState &rs, const AlnRec *tab, uint64_t n_rec, const uint32_t *arena, uint64_t arena_cap, size_t lo) {
    swmi_batch *b = rs.b;
    if (rs.defer_copy && b->raw_chunks.empty()) {
        b->raw_ext = arena; b->rtab_ext = tab; b->raw_ext_records = n_rec; b->raw_ext_cap = arena_cap;
        b->raw_chunks.push_back(swmi_batch::RawChunk{0, 0, 0, (size_t)n_rec, lo, {}});
        return SWMI_OK;
    }
    uint64_t used = 0;
    for (uint64_t k = 0; k < n_rec; k++) {
        const uint64_t end = (((uint64_t)tab[k].off_hi << 32) | tab[k].off_lo) + rec_words(tab[k].n_ops, b->rec_strings);
        used = std::max(used, end);
    }
    if (used > arena_cap) return fail(SWMI_ERR_HIP, "record payloads overrun the arena");
    b->raw_chunks.push_back(swmi_batch::RawChunk{b->raw.size(), (size_t)used, b->rtab.size(), (size_t)n_rec, lo, {}});
    b->raw.insert(b->raw.end(), arena, arena + used);
    b->rtab.insert(b->rtab.end(), tab, tab + n_rec);
    return SWMI_OK;
}

// Runs the kernels for work[lo, hi) -- one launch of each kernel the chunk needs -- waits, and keeps the records; a record arena
// or table that proves too small is grown to the size the kernels asked for and the traceback repeated.
// Cell-list geometry: uniform (cell_cap per pair) when cells_exact is null, else exact per pair (the re-run of overflowed pairs).
static int run_chunk(RunState &rs, const std::vector<Work> &work, size_t lo, size_t hi,
                     const std::vector<uint64_t> *cells_exact, std::vector<PairOut> &outs) {
    swmi_ctx *ctx = rs.ctx;
    swmi_batch *b = rs.b;
    const size_t np = hi - lo;
    int rc;
    if ((rc = prepare_chunk(rs, work, lo, hi, cells_exact))) return rc;
    const swmi_batch::Prep &pr = b->prep;
    const uint64_t seam_words = pr.seam_words, n_windows = pr.n_windows;
    const uint32_t max_path = pr.max_path, max_read = pr.max_read;
    const size_t n_strip_items = pr.n_strip_items, n_col_items = pr.n_col_items, n_res = pr.n_res, n_tf = pr.n_tf;
    const uint32_t res_lds_words = pr.res_lds_words, res_ops_words = pr.res_ops_words, res_cell_cap = SWMI_RES_CELL_CAP;
    const uint32_t tf_max_m = pr.tf_max_m, tf_max_n = pr.tf_max_n, tf_max_path = pr.tf_max_path;
    if (seam_words) HIP_TRY(hipMemsetAsync(b->d_seam.p, 0, seam_words * 4, ctx->stream));

    // cell lists
    uint32_t cell_cap = ctx->cell_cap;
    std::vector<uint64_t> coff;
    std::vector<uint32_t> ccap;
    uint64_t cells_total = 0;
    if (cells_exact) {
        coff.resize(np); ccap.resize(np);
        for (size_t k = 0; k < np; k++) {
            coff[k] = cells_total;
            ccap[k] = (uint32_t)std::min<uint64_t>((*cells_exact)[lo + k], 0xFFFFFFFFu);
            cells_total += ccap[k];
        }
        if ((rc = b->d_cells_off.reserve(np * 8))) return rc;
        if ((rc = b->d_cells_cap.reserve(np * 4))) return rc;
        HIP_TRY(hipMemcpyAsync(b->d_cells_off.p, coff.data(), np * 8, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(b->d_cells_cap.p, ccap.data(), np * 4, hipMemcpyHostToDevice, ctx->stream));
    } else {
        // a launch of few pairs can afford long lists: a periodic reference against one read is ONE pair with a tied maximum
        // per period (EngineerData.java:118), and a list that overflows costs a second run of the pair
        if (!ctx->cell_cap_set) cell_cap = (uint32_t)std::min<uint64_t>(65536, std::max<uint64_t>(cell_cap, (4ull << 20) / np));
        cells_total = (uint64_t)np * cell_cap;
    }
    if ((rc = b->d_cells.reserve(std::max<uint64_t>(cells_total, 1) * sizeof(uint2)))) return rc;

    const uint32_t lds_words = (max_path + 3) / 4 + 1;        // one staged op per byte
    const uint32_t lds_read_words = (max_read + 3) / 4 + 1;
    {
        // the traceback stages one alignment's ops and the read per walker (4 per workgroup) next to its direction tiles:
        // 160 KB of LDS per workgroup bound the longest pair (m + n of about 16 k bases in mode 0, 24 k in modes 1/2)
        const uint64_t need = traceback_lds_bytes(b->eff_mode, max_path, max_read);
        if (need > 160ull * 1024)
            return fail(SWMI_ERR_UNSUPPORTED, "a pair of %u bases in total needs %llu bytes of LDS for the traceback (limit 163840)",
                        max_path, (unsigned long long)need);
    }

    const auto c0 = std::chrono::steady_clock::now();
    uint64_t arena_cap = std::max<uint64_t>(np * ctx->arena_words_per_pair, 1024);
    uint64_t tab_cap = std::max<uint64_t>(np * ctx->recs_per_pair_x16 / 16 + 64, 256);
    std::vector<uint8_t> saved_outs;       // PairOut block carried across an arena re-allocation
    for (int attempt = 0;; attempt++) {
        // (an overflow that growing cannot cure -- a path longer than the staging area -- must not retry for ever)
        if (attempt > 4) return fail(SWMI_ERR_HIP, "the record arena overflowed %d times in a row; results discarded", attempt);
        const size_t t_off = result_tab_off(np), a_off = result_arena_off(np, tab_cap);
        if ((rc = b->d_result.reserve(a_off + arena_cap * 4))) return rc;
        uint8_t *res = b->d_result.as<uint8_t>();
        if (attempt > 0) {         // (on the first attempt the fill kernel zeroes the arena header itself)
            HIP_TRY(hipMemsetAsync(res, 0, 64, ctx->stream));
            HIP_TRY(hipMemcpyAsync(res + result_out_off(), saved_outs.data(), saved_outs.size(),
                                   hipMemcpyHostToDevice, ctx->stream));
        }

        FillArgs &fa = rs.fa;
        fa.seqw = b->d_seqw.as<uint32_t>();
        fa.refs = b->d_refs.as<SeqDesc>();
        fa.reads = b->d_reads.as<SeqDesc>();
        fa.pairs = b->d_pairs.as<PairDesc>();
        fa.dir = b->d_dir.as<uint32_t>();
        fa.seam = b->d_seam.as<int32_t>();
        fa.out = (PairOut *)(res + result_out_off());
        fa.cells = b->d_cells.as<uint2>();
        fa.cells_off = cells_exact ? b->d_cells_off.as<uint64_t>() : nullptr;
        fa.cells_cap = cells_exact ? b->d_cells_cap.as<uint32_t>() : nullptr;
        fa.hdr = (ArenaHdr *)res;
        fa.dbg = nullptr;
        fa.dbg_pad = 0;
        static const bool dbg_fill = getenv("SWMI_DEBUG_FILL") != nullptr;      // (getenv walks the whole environment: once per process, not per run)
        if (dbg_fill) {                          // diagnostics: per-pair slow-path entries and wave cycles
            if ((rc = b->d_dbg.reserve(np * 16))) return rc;
            fa.dbg = b->d_dbg.as<unsigned long long>();
            fa.dbg_thr0 = getenv("SWMI_DEBUG_THR0") ? (uint32_t)atoi(getenv("SWMI_DEBUG_THR0")) : 1u;
            fa.dbg_pad = getenv("SWMI_DEBUG_SKIP") ? 1u : 0u;
        }
        fa.n_pairs = (uint32_t)np;
        fa.cell_cap = cell_cap;
        fa.match = b->params.match; fa.mismatch = b->params.mismatch; fa.gap = b->params.gap;
        fa.strict = b->params.tie_mode == SWMI_TIE_STRICT;
        fa.mode = b->eff_mode;
        const bool pipe = n_strip_items && !rs.one_wave_sweep;
        fa.skip_multi = pipe ? 1u : 0u;
        fa.strip_items = pipe ? b->d_strip_items.as<StripItem>() : nullptr;
        fa.progress = pipe ? b->d_progress.as<uint32_t>() : nullptr;
        fa.n_strip_items = pipe ? (uint32_t)n_strip_items : 0u;
        fa.err_host = (uint32_t *)ctx->h_err.dp;
        fa.pad3 = 0;
        if (attempt == 0) b->timing.col_chunks += (uint32_t)n_col_items + (pipe ? (uint32_t)pr.n_strip_chunks : 0u);
        fa.col_items = n_col_items ? b->d_col_items.as<ColItem>() : nullptr;
        fa.n_col_items = (uint32_t)n_col_items;
        fa.strip_spins = ctx->dbg_strip_spins;
        // split traceback: the walk-item queue; its counter is zeroed by the sweep kernel of the first attempt
        const bool split_q = rs.tb_split && !cells_exact && b->eff_mode == 1 && n_windows < 0xFFFFFFFFull;
        const uint64_t q_cap = std::min<uint64_t>(std::max<uint64_t>(cells_total, 1), 1ull << 24);
        if (split_q && (rc = b->d_queue.reserve(256 + q_cap * sizeof(uint4)))) return rc;
        fa.q_reset = split_q ? b->d_queue.as<uint32_t>() : nullptr;

        TraceArgs &ta = rs.ta;
        ta.seqw = fa.seqw; ta.refs = fa.refs; ta.reads = fa.reads; ta.pairs = fa.pairs;
        ta.dir = fa.dir; ta.out = fa.out; ta.cells = fa.cells;
        ta.cells_off = fa.cells_off; ta.cells_cap = fa.cells_cap;
        ta.hdr = (ArenaHdr *)res;
        ta.arena = (uint32_t *)(res + a_off);
        ta.arena_cap_words = arena_cap;
        ta.rec_tab = (AlnRec *)(res + t_off);
        ta.rec_tab_cap = (uint32_t)std::min<uint64_t>(tab_cap, 0xFFFFFFFFu);
        ta.n_pairs = fa.n_pairs; ta.cell_cap = fa.cell_cap;
        ta.match = fa.match; ta.mismatch = fa.mismatch; ta.gap = fa.gap; ta.strict = fa.strict;
        ta.lds_words = lds_words;
        ta.seam = fa.seam;
        ta.mode = b->eff_mode;
        ta.pad2 = 0;
        ta.lds_read_words = lds_read_words;
        ta.out_host = nullptr;
        // (without zero-copy results the kernels' give-up codes still need a host-visible word: the context's own)
        ta.ovf_host = (uint32_t *)ctx->h_err.dp + 4;
        ((volatile uint32_t *)ctx->h_err.p)[4] = 0u; ((volatile uint32_t *)ctx->h_err.p)[5] = 0u;
        ta.win_off = nullptr; ta.q_count = nullptr; ta.q_items = nullptr; ta.q_cap = 0; ta.pad4 = 0;
        const bool strings = ctx->device_strings != 0 && b->d_raw.p != nullptr;
        ta.raw = strings ? b->d_raw.as<uint8_t>() : nullptr;
        ta.raw_off = strings ? b->d_raw_off.as<uint64_t>() : nullptr;
        ta.raw_reads_at = b->n_refs + 1u;
        b->rec_strings = strings;
        ResidentArgs xa;
        xa.res_items = n_res ? b->d_res_items.as<uint32_t>() : nullptr;
        xa.n_res = (uint32_t)n_res; xa.res_lds_words = res_lds_words; xa.res_cell_cap = res_cell_cap; xa.res_ops_words = res_ops_words;
        TFusedArgs xt{};
        xt.items = n_tf ? b->d_tf_items.as<uint32_t>() : nullptr;
        xt.n_items = (uint32_t)n_tf;
        xt.cell_cap = 16;
        xt.tile_words = ((tf_max_m + 63u + 15u) / 16u) * 64u * SWMI_TF_BR;
        xt.ref_words = (tf_max_n + 3u) / 4u + 1u;
        xt.read_words = (tf_max_m + 3u) / 4u + 1u;
        xt.stage_words = (tf_max_path + 3u) / 4u + 1u + 128u;          // (+ SWMI_EMIT_SCRATCH_WORDS: swmi_emit.h)
        static const bool tf_marks = getenv("SWMI_DEBUG_MARKS") != nullptr;
        xt.debug_marks = tf_marks ? 1u : 0u;
        xt.lds_words = (xt.tile_words + 2u * xt.cell_cap + xt.stage_words + xt.ref_words + xt.read_words + 3u) & ~3u;
        {   // helper wavefronts while their LDS regions fit beside the four sweepers' (160 KB per workgroup)
            static const int tf_helpers = getenv("SWMI_TF_HELPERS") ? atoi(getenv("SWMI_TF_HELPERS")) : 2;      // (diagnostics: 0 .. SWMI_TF_HELPERS)
            const uint64_t region = 4ull * xt.lds_words, budget = 156ull * 1024;      // (160 KB less the workgroup's queues)
            const uint64_t left = budget > 4 * region ? budget - 4 * region : 0;
            xt.n_helpers = (uint32_t)std::min<uint64_t>((uint64_t)std::max(0, std::min(tf_helpers, (int)SWMI_TF_HELPERS)), region ? left / region : 0);
        }
        const bool zc = ctx->zero_copy != 0;
        if (zc) {
            // results land in pinned host memory while the kernel runs: [overflow word .. | PairOut x np | arena]
            if ((rc = b->h_result.reserve(a_off + arena_cap * 4))) return rc;
            uint8_t *hd = (uint8_t *)b->h_result.dp;
            *(volatile uint32_t *)b->h_result.p = 0u;
            ((volatile uint32_t *)b->h_result.p)[1] = 0u;
            ta.ovf_host = (uint32_t *)hd;
            ta.out_host = (PairOut *)(hd + result_out_off());
            ta.arena = (uint32_t *)(hd + a_off);
            ta.rec_tab = (AlnRec *)(hd + t_off);
        }
        ta.dbg = nullptr;
        if (dbg_fill) {
            if ((rc = b->d_dbg2.reserve(np * 32))) return rc;
            HIP_TRY(hipMemsetAsync(b->d_dbg2.p, 0, np * 32, ctx->stream));
            ta.dbg = b->d_dbg2.as<unsigned long long>();
        }

        // Every pair handled whole by sw_tfused_kernel / sw_resident_pairs_kernel: the sweep kernel would run only to zero the
        // arena header.  Such a launch takes its header from a ring of zeroed slots instead (results in pinned memory only: a
        // D2H copy fetches the header with the block it sits in).
        const bool whole_only = zc && n_tf + n_res == np;
        if (whole_only) {
            const uint32_t slots = 1024;
            if (!ctx->d_hdr_ring.p || ctx->hdr_next >= slots) {
                if ((rc = ctx->d_hdr_ring.reserve((size_t)slots * 64))) return rc;
                HIP_TRY(hipMemsetAsync(ctx->d_hdr_ring.p, 0, (size_t)slots * 64, ctx->stream));
                ctx->hdr_next = 0;
            }
            ta.hdr = (ArenaHdr *)(ctx->d_hdr_ring.as<uint8_t>() + (size_t)ctx->hdr_next++ * 64);
        }
        // diagnostics (SWMI_EXT_EVENTS=1): the plain case -- one sweep kernel, one traceback kernel -- timed by the kernels' own
        // dispatches instead of by events around them (the three events cost 3.5 us per run, tests/manual/prof_events_ab.py;
        // hipExtLaunchKernelGGL's start/stop events cost more: profiles/r02/ab_ext_events.txt)
        const bool split_now = rs.tb_split && !cells_exact && b->eff_mode == 1 && n_windows < 0xFFFFFFFFull;
        const bool ext_timing = ctx->profiling == 1 && attempt == 0 && !whole_only && !n_res && !n_tf && !split_now && b->eff_mode == 1 &&
                                !fa.n_strip_items && !fa.n_col_items && ctx->ext_events;
        if (ctx->profiling && !ext_timing) HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
        if (attempt == 0 && !whole_only) {       // the workspace survives an arena-overflow retry
            if (fa.n_strip_items) HIP_TRY(hipMemsetAsync(fa.progress, 0, (size_t)fa.n_strip_items * sizeof(uint32_t), ctx->stream));
            HIP_TRY(swmi_launch_fill(&fa, ctx->stream, ext_timing ? ctx->ev[0] : nullptr, ext_timing ? ctx->ev[1] : nullptr));
            rs.launches++;
        }
        if (n_tf) HIP_TRY(swmi_launch_tfused(&ta, &xt, ctx->stream));             // (sweep AND traceback of its pairs: timed with the sweep)
        if (whole_only && n_tf) rs.launches++;
        if (ctx->profiling && !ext_timing) HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));     // end of the sweep = start of the traceback
        if (n_res) HIP_TRY(swmi_launch_resident(&ta, &xa, ctx->stream));          // (timed with the traceback)
        if (attempt == 0) b->timing.resident_pairs += (uint32_t)n_res;
        if (attempt == 0) b->timing.tfused_pairs += (uint32_t)n_tf;
        // (the exact-size re-run of pairs whose lists overflowed takes one workgroup per pair: its lists have no per-window cap)
        const bool split = rs.tb_split && !cells_exact && b->eff_mode == 1 && n_windows < 0xFFFFFFFFull;
        const bool sweep_only = ctx->scores_only != 0 && !cells_exact;
        if (sweep_only) {
            // option scores_only: the pair outputs as the sweep kernels left them (the traceback kernels, which otherwise mirror
            // them into the host block, do not run)
            if ((rc = b->h_result.reserve(a_off + arena_cap * 4))) return rc;
            HIP_TRY(hipMemcpyAsync((uint8_t *)b->h_result.p + result_out_off(), res + result_out_off(), np * sizeof(PairOut),
                                   hipMemcpyDeviceToHost, ctx->stream));
        } else if (split) {
            ta.win_off = b->d_win_off.as<uint32_t>();
            ta.q_count = b->d_queue.as<uint32_t>();
            ta.q_items = (uint4 *)(b->d_queue.as<uint8_t>() + 256);
            ta.q_cap = (uint32_t)q_cap;
            if (attempt > 0 || whole_only) HIP_TRY(hipMemsetAsync(ta.q_count, 0, 4, ctx->stream));      // (no sweep kernel ran to zero it)
            HIP_TRY(swmi_launch_traceback_split(&ta, (uint32_t)n_windows, ctx->stream));
        } else if (n_res + n_tf < np) {
            HIP_TRY(swmi_launch_traceback(&ta, ctx->stream, ext_timing ? ctx->ev[2] : nullptr, ext_timing ? ctx->ev[3] : nullptr));
        }
        // (profiling = 2: only the sweep is bracketed -- two marker packets per run instead of three; each costs ~3.5 us of the step)
        const bool time_all = ctx->profiling == 1;
        if (time_all && !ext_timing) HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));

        // without zero-copy: one D2H of header + pair outputs + as much of the arena as the previous run used
        // (plus slack); the rare remainder is fetched after the header has been read
        // (header, pair outputs and as many table entries and arena words as the previous run used, plus slack)
        const uint64_t copy_words = std::min<uint64_t>(arena_cap, std::max<uint64_t>(256, np * ctx->arena_copy_wpp));
        const uint64_t copy_recs = std::min<uint64_t>(tab_cap, np * ctx->recs_per_pair_x16 / 16 + 64);
        if (!zc && !sweep_only) {
            if ((rc = b->h_result.reserve(a_off + arena_cap * 4))) return rc;
            HIP_TRY(hipMemcpyAsync(b->h_result.p, res, t_off + copy_recs * sizeof(AlnRec), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipMemcpyAsync((uint8_t *)b->h_result.p + a_off, res + a_off, copy_words * 4, hipMemcpyDeviceToHost, ctx->stream));
            if (time_all) HIP_TRY(hipEventRecord(ctx->ev[4], ctx->stream));
        }
        static const char *watchdog = getenv("SWMI_DEBUG_WATCHDOG");      // diagnostics: give up on a launch that does not end, show the kernel's marks
        if (watchdog) {
            const auto w0 = std::chrono::steady_clock::now();
            while (hipStreamQuery(ctx->stream) == hipErrorNotReady)
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() > atof(watchdog)) {
                    const volatile uint32_t *hm = (const volatile uint32_t *)b->h_result.p;
                    fprintf(stderr, "[swmi watchdog] launch still running after %s s; marks:", watchdog);
                    for (int k = 0; k < 12; k++) fprintf(stderr, " %08x", hm ? hm[k] : 0u);
                    fprintf(stderr, "\n");
                    fflush(stderr);
                    _exit(3);
                }
        }
        const auto c1 = std::chrono::steady_clock::now();
        {   // a batch is sub-millisecond: poll the stream for spin_us (this context only, no process-wide spin flag), then block
            hipError_t q = hipErrorNotReady;
            while (ctx->spin_us > 0 && (q = hipStreamQuery(ctx->stream)) == hipErrorNotReady &&
                   std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c1).count() < (double)ctx->spin_us)
                __builtin_ia32_pause();
            if (q != hipSuccess && q != hipErrorNotReady) return fail(SWMI_ERR_HIP, "hipStreamQuery: %s", hipGetErrorString(q));
            if (q != hipSuccess) HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
        const auto c2 = std::chrono::steady_clock::now();
        volatile uint32_t *giveup = zc ? (volatile uint32_t *)b->h_result.p + 1 : (volatile uint32_t *)ctx->h_err.p + 5;
        if (*giveup != 0u) {      // sw_tfused_kernel gave up a wait that cannot last (never seen)
            const uint32_t code = *giveup;
            *giveup = 0u;
            return fail(SWMI_ERR_HIP, "sw_tfused_kernel: internal wait abandoned (code %08x); results discarded", code);
        }
        if (*(volatile uint32_t *)ctx->h_err.p != 0u) {
            // a strip of the pipelined sweep gave up waiting for its producer wavefront (it was not dispatched, or did not
            // move for the whole spin budget): nothing of this launch is used.  The chunk is swept again with ONE wavefront
            // per pair, strip after strip -- no wavefront of that sweep waits for another workgroup.
            *(volatile uint32_t *)ctx->h_err.p = 0u;
            if (rs.one_wave_sweep)
                return fail(SWMI_ERR_HIP, "the sweep raised its error flag without the strip pipeline; results discarded");
            rs.one_wave_sweep = true;
            b->timing.strip_fallbacks++;
            return run_chunk(rs, work, lo, hi, cells_exact, outs);
        }
        rs.enqueue_us += std::chrono::duration<double, std::micro>(c1 - c0).count();
        rs.wait_us += std::chrono::duration<double, std::micro>(c2 - c1).count();
        if (ctx->profiling) {
            float ms = 0;
            if (attempt == 0 || whole_only) { HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1])); rs.fill_ms += ms; }
            if (time_all) { HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[ext_timing ? 2 : 1], ctx->ev[3])); rs.tb_ms += ms; }
            if (time_all && !zc) { HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4])); rs.d2h_ms += ms; }
        }
        if ((rc = dump_traceback_diagnostics(b, ta, np, n_tf))) return rc;
        if (attempt == 0 && (rc = dump_fill_diagnostics(b, fa, np))) return rc;
        const uint8_t *h = (const uint8_t *)b->h_result.p;
        if (sweep_only) {
            // scores only: a pair swept by several wavefronts (column chunks, strips) has only its maximum combined, and a
            // maximum of 0 is the degenerate case (what finish_pair does in the traceback kernels)
            outs.assign((const PairOut *)(h + result_out_off()), (const PairOut *)(h + result_out_off()) + np);
            for (size_t k = 0; k < np; k++) {
                PairOut &o = outs[k];
                if (o.score <= 0 && !(o.flags & SWMI_F_DEGENERATE)) {
                    const uint32_t pair = work[lo + k].pair;
                    o.score = 0; o.flags = SWMI_F_DEGENERATE;
                    o.n_cells = (uint64_t)b->read_desc[pair % b->n_reads].len * b->ref_desc[pair / b->n_reads].len;
                }
                o.flags &= SWMI_F_DEGENERATE;
            }
            return SWMI_OK;
        }
        ArenaHdr hdr_copy{};
        const ArenaHdr *hdr = (const ArenaHdr *)h;
        bool overflow;
        if (zc) {
            overflow = *(const volatile uint32_t *)h != 0u;
            if (overflow) {        // rare: how much was needed is in the device-side header
                HIP_TRY(hipMemcpy(&hdr_copy, ta.hdr, sizeof hdr_copy, hipMemcpyDeviceToHost));
                hdr = &hdr_copy;
            }
        }
        const uint64_t hdr_words = zc && !overflow ? 0 : hdr->reserved & SWMI_HDR_WORD_MASK;     // (zero-copy: only read after an overflow)
        const uint64_t hdr_recs = zc && !overflow ? 0 : hdr->reserved >> SWMI_HDR_WORD_BITS;
        if (!zc) overflow = hdr_words > arena_cap || hdr_recs > tab_cap;
        if (overflow) {            // records were dropped: grow to the exact need and redo the traceback
            if (zc) {
                saved_outs.resize(np * sizeof(PairOut));
                HIP_TRY(hipMemcpy(saved_outs.data(), res + result_out_off(), saved_outs.size(), hipMemcpyDeviceToHost));
            } else {
                saved_outs.assign(h + result_out_off(), h + result_out_off() + np * sizeof(PairOut));
            }
            for (size_t k = 0; k < np; k++) {
                PairOut &so = ((PairOut *)saved_outs.data())[k];
                so.flags &= ~SWMI_F_ARENA_OVF;
                // the split traceback counts a pair's cells by atomics and flags list overflows itself: start both over
                if (split && !(so.flags & SWMI_F_DEGENERATE)) { so.n_cells = 0; so.flags &= ~SWMI_F_CELL_OVF; }
            }
            arena_cap = std::max<uint64_t>(arena_cap, hdr_words + 1024);
            tab_cap = std::max<uint64_t>(tab_cap, hdr_recs + 64);
            // (the kernels reserve payload and table entry with one 64-bit counter: 36 bits of dwords, 28 bits of records)
            if (arena_cap >= SWMI_HDR_WORD_MASK || tab_cap >= (1ull << (64u - SWMI_HDR_WORD_BITS)) - 1ull)
                return fail(SWMI_ERR_UNSUPPORTED, "one launch would hold %llu alignment records in %llu arena dwords: lower max_workspace_bytes so that the batch runs in smaller launches",
                            (unsigned long long)tab_cap, (unsigned long long)arena_cap);
            ctx->arena_words_per_pair = std::max<uint64_t>(ctx->arena_words_per_pair, arena_cap / np + 1);
            ctx->recs_per_pair_x16 = std::max<uint64_t>(ctx->recs_per_pair_x16, tab_cap * 16 / np + 1);
            continue;
        }
        outs.assign((const PairOut *)(h + result_out_off()), (const PairOut *)(h + result_out_off()) + np);
        for (auto &o : outs) o.flags &= ~SWMI_F_ARENA_OVF;
        // how many records there are follows from the pair outputs (zero-copy: the header sits in device memory)
        uint64_t n_rec = 0;
        if (zc) {
            for (auto &o : outs)
                if (!(o.flags & (SWMI_F_DEGENERATE | SWMI_F_CELL_OVF))) n_rec += o.n_cells;
        } else {
            n_rec = hdr_recs;
        }
        if (n_rec > tab_cap) return fail(SWMI_ERR_HIP, "more records than the table holds");
        const AlnRec *tab = (const AlnRec *)(h + t_off);
        const uint32_t *aw = (const uint32_t *)(h + a_off);
        if (!zc) {
            if (n_rec > copy_recs)
                HIP_TRY(hipMemcpy((uint8_t *)b->h_result.p + t_off + copy_recs * sizeof(AlnRec), res + t_off + copy_recs * sizeof(AlnRec),
                                  (n_rec - copy_recs) * sizeof(AlnRec), hipMemcpyDeviceToHost));
            if (hdr_words > copy_words)
                HIP_TRY(hipMemcpy((uint8_t *)b->h_result.p + a_off + copy_words * 4, res + a_off + copy_words * 4,
                                  (hdr_words - copy_words) * 4, hipMemcpyDeviceToHost));
            ctx->arena_copy_wpp = hdr_words * 5 / (4 * np) + 2;
            ctx->recs_per_pair_x16 = std::max<uint64_t>(ctx->recs_per_pair_x16, n_rec * 20 / np + 1);
        }
        if (rs.keep) {
            if ((rc = keep_chunk_records(rs, tab, n_rec, aw, arena_cap, lo))) return rc;
            rs.copyout_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c2).count();
        }
        return SWMI_OK;
    }
}

// the records left in the pinned block move into the batch's own vectors (before the block is written again)
static int settle_raw(swmi_batch *b) {
    if (!b->raw_ext) return SWMI_OK;
    const uint32_t *aw = b->raw_ext;
    const AlnRec *tab = b->rtab_ext;
    b->raw_ext = nullptr; b->rtab_ext = nullptr;
    uint64_t used = 0;
    for (uint64_t k = 0; k < b->raw_ext_records; k++)
        used = std::max(used, (((uint64_t)tab[k].off_hi << 32) | tab[k].off_lo) + rec_words(tab[k].n_ops, b->rec_strings));
    if (used > b->raw_ext_cap) return fail(SWMI_ERR_HIP, "record payloads overrun the arena");
    b->raw.assign(aw, aw + used);
    b->rtab.assign(tab, tab + b->raw_ext_records);
    if (b->raw_chunks.size() == 1) { b->raw_chunks[0].at = 0; b->raw_chunks[0].words = (size_t)used; b->raw_chunks[0].tab_at = 0; }
    return SWMI_OK;
}

// Turns the record tables of the last run into per-pair alignment lists (first use of an alignment accessor).
static int ensure_indexed(swmi_batch *b) {
    if (b->indexed) return SWMI_OK;
    if (b->scores_only)
        return fail(SWMI_ERR_INVALID, "the batch was run with scores_only = 1: scores and totals only, no alignments");
    if (b->records_dropped)
        return fail(SWMI_ERR_INVALID, "this chunk's alignment records were not kept (option stream_keep_records = 0): scores, counts and totals only");
    const std::vector<Work> &work = b->work;
    // every launch's table (dense, read sequentially) with the arena its payload offsets refer to; the only launch of a
    // run may still sit in the pinned block the kernels wrote: indexed where it is, nothing copied
    struct Src { const AlnRec *tab; uint64_t n_rec; const uint32_t *arena; uint64_t words; const swmi_batch::RawChunk *c; };
    std::vector<Src> srcs;
    for (auto &c : b->raw_chunks) {
        if (b->raw_ext) srcs.push_back(Src{b->rtab_ext, b->raw_ext_records, b->raw_ext, b->raw_ext_cap, &c});
        else            srcs.push_back(Src{b->rtab.data() + c.tab_at, c.n_rec, b->raw.data() + c.at, c.words, &c});
    }
    auto wpos_of = [](const Src &sr, uint32_t out_id) -> uint64_t {
        return sr.c->wpos.empty() ? sr.c->lo + out_id : (out_id < sr.c->wpos.size() ? sr.c->wpos[out_id] : ~0ull);
    };
    for (auto &w : work) b->pairs[w.pair].count = 0;
    uint64_t total = 0;
    for (auto &sr : srcs) {
        for (uint64_t k = 0; k < sr.n_rec; k++) {
            const uint64_t wp = wpos_of(sr, sr.tab[k].out_id);
            if (wp >= work.size()) return fail(SWMI_ERR_HIP, "record of an unknown pair");
            b->pairs[work[wp].pair].count++;
        }
        total += sr.n_rec;
    }
    uint64_t run = 0;
    for (auto &w : work) { PairRes &pr = b->pairs[w.pair]; pr.first = run; run += pr.count; }
    if (run != total) return fail(SWMI_ERR_HIP, "record count mismatch");
    b->alns.assign(run, HostAln{});
    std::vector<uint32_t> cursor;
    const bool strict = b->params.tie_mode == SWMI_TIE_STRICT;
    for (auto &sr : srcs) {
        for (uint64_t k = 0; k < sr.n_rec; k++) {
            const AlnRec &e = sr.tab[k];
            const uint64_t wp = wpos_of(sr, e.out_id);
            PairRes &pr = b->pairs[work[wp].pair];
            const uint64_t off = ((uint64_t)e.off_hi << 32) | e.off_lo;
            if (off + rec_words(e.n_ops, b->rec_strings) > sr.words) return fail(SWMI_ERR_HIP, "record payload overruns the arena");
            HostAln a;
            a.rank = e.rank; a.begin = e.begin; a.end_i = e.end_i; a.end_j = e.end_j; a.n_ops = e.n_ops;
            a.rec = sr.arena + off;
            // records of the split traceback come in any order: placed as they come, ordered by their cell below
            if (e.rank == SWMI_RANK_BY_CELL) {
                if (cursor.empty()) cursor.assign(work.size(), 0u);
                uint32_t &c = cursor[wp];
                if (c >= pr.count) return fail(SWMI_ERR_HIP, "more records than counted for a pair");
                b->alns[pr.first + c++] = a;
            } else {
                if (e.rank >= pr.count) return fail(SWMI_ERR_HIP, "record rank %u out of range", e.rank);
                b->alns[pr.first + e.rank] = a;
            }
        }
    }
    // ordered as OptAlignments lists them: by the rank the traceback kernel computed, or by cell: row-major
    // (SmithWaterman.java:157-185), or per anti-diagonal with ascending j for the strict mode (DistributedSW.java:209-239)
    for (size_t wi = 0; wi < work.size(); wi++) {
        PairRes &pr = b->pairs[work[wi].pair];
        if (!(pr.flags & SWMI_PAIR_DEGENERATE) && pr.count != pr.n_cells)
            return fail(SWMI_ERR_HIP, "pair %u: %llu records for %llu max cells", work[wi].pair,
                        (unsigned long long)pr.count, (unsigned long long)pr.n_cells);
        if (!cursor.empty() && cursor[wi] > 1) {
            auto key = [strict](const HostAln &x) {
                return strict ? (((uint64_t)((uint32_t)x.end_i + (uint32_t)x.end_j)) << 32) | (uint32_t)x.end_j
                              : ((uint64_t)(uint32_t)x.end_i << 32) | (uint32_t)x.end_j;
            };
            std::sort(b->alns.begin() + pr.first, b->alns.begin() + pr.first + pr.count,
                      [&](const HostAln &x, const HostAln &y) { return key(x) < key(y); });
        }
        if (!cursor.empty() && cursor[wi] != 0)
            for (uint64_t k = 0; k < pr.count; k++) b->alns[pr.first + k].rank = (uint32_t)k;
        // DistributedSW.GetAlignments sorts the collected alignments by beginning (DistributedSW.java:480)
        if (strict && pr.count > 1)
            std::stable_sort(b->alns.begin() + pr.first, b->alns.begin() + pr.first + pr.count,
                             [](const HostAln &x, const HostAln &y) { return x.begin < y.begin; });
    }
    if (!b->rec_strings) {
        // records without strings (option device_strings = 0): one buffer for all strings the host builds (two mallocs per
        // alignment cost more than filling them)
        b->str_at.resize(run);
        uint64_t chars = 0;
        for (uint64_t k = 0; k < run; k++) { b->alns[k].str_id = -1; b->str_at[k] = chars; chars += 2ull * (b->alns[k].n_ops + 1); }
        b->str_buf.resize(chars);
    }
    b->indexed = true;
    return SWMI_OK;
}

extern "C" int swmi_batch_run(swmi_ctx *ctx, swmi_batch *b, const swmi_params *p) {
    if (!ctx || !b || !p) return fail(SWMI_ERR_INVALID, "null argument");
    if (p->tie_mode != SWMI_TIE_SERIAL && p->tie_mode != SWMI_TIE_STRICT)
        return fail(SWMI_ERR_INVALID, "unknown tie_mode %d", p->tie_mode);
    // GetAlignment tests `align == alignTypes[0]`, then `== alignTypes[1]`, else deletion
    // (SmithWaterman.java:388-401): with duplicate a/i/d characters the reference itself walks wrong
    // cells; that behaviour is not reproduced.
    if (p->types[0] == p->types[1] || p->types[0] == p->types[2] || p->types[1] == p->types[2])
        return fail(SWMI_ERR_UNSUPPORTED, "alignTypes a/i/d must be pairwise distinct");
    std::lock_guard<std::mutex> g(ctx->mu);
    static const bool host_dbg = getenv("SWMI_DEBUG_HOST") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) {
        return std::chrono::duration<double, std::micro>(c - a).count();
    };
    const auto h0 = now();
    {   // (hipSetDevice costs microseconds even when nothing changes; a sub-millisecond batch notices)
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != ctx->device) HIP_TRY(hipSetDevice(ctx->device));
    }
    b->params = *p;
    b->has_run = false;
    b->scores_only = ctx->scores_only != 0;

    const uint32_t n_refs = b->n_refs, n_reads = b->n_reads;
    const uint64_t n_pairs = (uint64_t)n_refs * n_reads;
    b->pairs.assign(n_pairs, PairRes{});
    b->alns.clear(); b->str_at.clear();
    b->raw.clear(); b->rtab.clear(); b->raw_chunks.clear(); b->indexed = false;
    b->raw_ext = nullptr; b->rtab_ext = nullptr;
    if (b->views_built || b->ref_view_ready.size() != n_refs) {     // (a run nobody read MapRef views of leaves them as they are)
        b->ref_view_ready.assign(n_refs, 0);
        b->ref_sites.assign(n_refs, {});
        b->ref_degenerate.assign(n_refs, 0);
        b->views_built = false;
    }
    b->timing = swmi_timing{};

    // pairs with an empty side never enter ScoreMatrix's loops (SmithWaterman.java:157-159): (0, [])
    // The schedule only depends on the sequence lengths and the pipeline mode: built once per batch.
    // mode 1 needs pad rows that cannot outgrow the real cells they derive from: mismatch <= 0 and gap <= 0
    b->eff_mode = (ctx->mode == 1 && (p->mismatch > 0 || p->gap > 0)) ? 2u : ctx->mode;
    // (measured, profiles/r02/sweeps_*.md: with ~5 alignments per pair the split traceback wins up to ~200 pairs; from a
    // few hundred pairs on one workgroup per pair keeps every SIMD busy anyway and its teams share the window re-sweeps)
    if (ctx->tb_split < 0 && !ctx->scores_only && b->eff_mode == 1 && n_pairs >= 64 && n_pairs <= 256 &&
        (b->auto_choice < 0 || memcmp(&b->auto_params, p, sizeof(swmi_params)) != 0)) {
        // Grain of the mode-1 traceback, chosen once per batch and parameter set: a sample of the pairs is aligned and the
        // tied maxima per pair are counted.  Periodic references (the reference's own EngineerData sets: every period ends
        // in a tied maximum, EngineerData.java:118) give every pair many alignments; one workgroup per pair then walks them
        // four at a time while most of the chip idles, so such batches take the split traceback (one wavefront per
        // window and per alignment, swmi_kernels.hip).
        std::vector<Work> sample;
        const uint64_t want = 48, stride = std::max<uint64_t>(1, n_pairs / want);
        for (uint64_t pi = stride / 2; pi < n_pairs && sample.size() < want; pi += stride) {
            const uint32_t m = b->read_desc[pi % n_reads].len, n = b->ref_desc[pi / n_reads].len;
            if (m == 0 || n == 0) continue;
            Work w;
            w.pair = (uint32_t)pi; w.cells = (uint64_t)m * n;
            w.dir_words = swmi_dir_words(m, n, 1); w.seam_words = swmi_seam_words(m, n);
            sample.push_back(w);
        }
        int choice = 1;
        if (!sample.empty()) {
            RunState rs0;
            rs0.ctx = ctx; rs0.b = b;
            rs0.tb_split = true;                      // (48 pairs: a small launch)
            rs0.keep = false;                         // (its records are not results)
            std::vector<PairOut> o0;
            int rc = run_chunk(rs0, sample, 0, sample.size(), nullptr, o0);
            if (rc) return rc;
            uint64_t cells = 0, live = 0;
            for (auto &o : o0)
                if (!(o.flags & SWMI_F_DEGENERATE)) { cells += o.n_cells; live++; }
            if (live && cells * 100 >= (uint64_t)ctx->auto_ties_x100 * live) choice = 0;
            b->prep.valid = false;                    // (the cached preparation was the sample's)
            b->timing = swmi_timing{};
        }
        b->auto_choice = choice;                      // 0: tie-heavy
        b->auto_params = *p;
    }
    if (b->eff_mode == 0) {
        // mode 0's 256-step direction tiles leave the least LDS for the staged alignment: batches with pairs too long for
        // it run as mode 1 (or 2) -- the results are the same
        uint32_t max_n = 0, max_m = 0;
        for (const auto &d : b->ref_desc) max_n = std::max(max_n, d.len);
        for (const auto &d : b->read_desc) max_m = std::max(max_m, d.len);
        if (traceback_lds_bytes(0, path_bound(max_n, max_m, *p), max_m) > 160ull * 1024)
            b->eff_mode = (p->mismatch > 0 || p->gap > 0) ? 2u : 1u;
    }
    if (b->work_mode != (int)b->eff_mode || b->work_tfused != (ctx->tfused == 1)) {
        b->work_tfused = ctx->tfused == 1;
        b->work.clear();
        b->work.reserve(n_pairs);
        b->work_cells = 0;
        // Longest first: the tail of a launch is made of short pairs.  The schedule depends on the LENGTHS only, so the two
        // sides are ordered by length once (n_refs log n_refs + n_reads log n_reads) and the pairs generated in that order --
        // by cells, exactly, whenever one side has a single length (one read, or a FASTA file of equal reads), else by
        // reference length first -- instead of sorting 10^6..10^8 pair records.
        std::vector<uint32_t> ro(n_refs), qo(n_reads);
        std::iota(ro.begin(), ro.end(), 0u);
        std::iota(qo.begin(), qo.end(), 0u);
        std::stable_sort(ro.begin(), ro.end(), [&](uint32_t a, uint32_t c) { return b->ref_desc[a].len > b->ref_desc[c].len; });
        std::stable_sort(qo.begin(), qo.end(), [&](uint32_t a, uint32_t c) { return b->read_desc[a].len > b->read_desc[c].len; });
        const bool refs_uniform = n_refs && b->ref_desc[ro.front()].len == b->ref_desc[ro.back()].len;
        auto add = [&](uint32_t r, uint32_t q, uint32_t n, uint32_t m, uint64_t dw, uint64_t sw) {
            Work w;
            w.pair = r * n_reads + q;
            w.cells = (uint64_t)m * n;
            w.dir_words = dw; w.seam_words = sw;
            b->work_cells += w.cells;
            b->work.push_back(w);
        };
        const bool tf = ctx->tfused == 1;
        if (refs_uniform) {                              // (reads outermost: descending m x the one n)
            for (uint32_t q : qo) {
                const uint32_t m = b->read_desc[q].len;
                if (m == 0) continue;
                const uint32_t n = n_refs ? b->ref_desc[ro[0]].len : 0;
                if (n == 0) break;
                const uint64_t dw = swmi_dir_words(m, n, b->eff_mode, tf), sw = swmi_seam_words(m, n);
                for (uint32_t r : ro) add(r, q, n, m, dw, sw);
            }
        } else {
            uint32_t last_m = 0xFFFFFFFFu;
            uint64_t dw = 0, sw = 0;
            for (uint32_t r : ro) {
                const uint32_t n = b->ref_desc[r].len;
                if (n == 0) continue;
                last_m = 0xFFFFFFFFu;
                for (uint32_t q : qo) {
                    const uint32_t m = b->read_desc[q].len;
                    if (m == 0) continue;
                    if (m != last_m) { dw = swmi_dir_words(m, n, b->eff_mode, tf); sw = swmi_seam_words(m, n); last_m = m; }
                    add(r, q, n, m, dw, sw);
                }
            }
        }
        b->work_mode = (int)b->eff_mode;
    }
    const std::vector<Work> &work = b->work;
    const uint64_t total_cells = b->work_cells;
    const auto h1 = now();
    RunState rs;
    rs.ctx = ctx; rs.b = b;
    rs.tb_split = b->eff_mode == 1 &&
                  (ctx->tb_split == 1 || (ctx->tb_split < 0 && (work.size() < 64 || (work.size() <= 256 && b->auto_choice == 0 &&
                                                                                      memcmp(&b->auto_params, p, sizeof(swmi_params)) == 0))));
    b->tb_split_used = rs.tb_split;

    std::vector<PairOut> outs;
    std::vector<uint32_t> arena;
    std::vector<size_t> ovf;                         // positions in `work` that overflowed their cell list
    uint64_t dir_bytes = 0;

    size_t lo = 0;
    while (lo < work.size()) {
        // chunk = as many pairs as fit the workspace cap (always at least one)
        uint64_t words = 0;
        size_t hi = lo;
        while (hi < work.size() && (hi == lo || (words + work[hi].dir_words) * 4 <= ctx->max_workspace_bytes)) {
            words += work[hi].dir_words;
            hi++;
        }
        dir_bytes += words * 4;
        rs.defer_copy = lo == 0 && hi == work.size();
        int rc = run_chunk(rs, work, lo, hi, nullptr, outs);
        rs.defer_copy = false;
        if (rc) return rc;
        for (size_t k = 0; k < hi - lo; k++) {
            PairRes &pr = b->pairs[work[lo + k].pair];
            pr.score = outs[k].score;
            pr.flags = (outs[k].flags & SWMI_F_DEGENERATE) ? SWMI_PAIR_DEGENERATE : 0u;
            pr.n_cells = outs[k].n_cells;
            if (outs[k].flags & SWMI_F_CELL_OVF) ovf.push_back(lo + k);
        }
        lo = hi;
    }

    const auto h2 = now();
    // pairs with more tied cells than cell_cap: run them again on the GPU with exact-size lists
    if (!ovf.empty()) {
        { int rc0 = settle_raw(b); if (rc0) return rc0; }       // (the re-run writes the pinned block again)
        std::vector<Work> w2;
        std::vector<uint64_t> exact;
        for (size_t pos : ovf) { w2.push_back(work[pos]); exact.push_back(b->pairs[work[pos].pair].n_cells); }
        size_t lo2 = 0;
        while (lo2 < w2.size()) {
            uint64_t words = 0;
            size_t hi2 = lo2;
            while (hi2 < w2.size() && (hi2 == lo2 || (words + w2[hi2].dir_words) * 4 <= ctx->max_workspace_bytes)) {
                words += w2[hi2].dir_words;
                hi2++;
            }
            int rc = run_chunk(rs, w2, lo2, hi2, &exact, outs);
            if (rc) return rc;
            swmi_batch::RawChunk &rc2 = b->raw_chunks.back();                              // (the launch's records, just appended)
            rc2.wpos.resize(hi2 - lo2);
            for (size_t k = 0; k < hi2 - lo2; k++) rc2.wpos[k] = (uint32_t)ovf[lo2 + k];   // chunk-local id -> position in `work`
            for (size_t k = 0; k < hi2 - lo2; k++) b->pairs[w2[lo2 + k].pair].n_cells = outs[k].n_cells;
            for (size_t k = 0; k < hi2 - lo2; k++)
                if (outs[k].flags & SWMI_F_CELL_OVF)
                    return fail(SWMI_ERR_HIP, "cell list overflowed again on the exact-size re-run");
            lo2 = hi2;
        }
        b->timing.rerun_pairs = (uint32_t)ovf.size();
    }

    b->timing.fill_ms = rs.fill_ms; b->timing.traceback_ms = rs.tb_ms; b->timing.d2h_ms = rs.d2h_ms;
    b->timing.total_ms = rs.fill_ms + rs.tb_ms + rs.d2h_ms;
    b->timing.fill_launches = rs.launches;
    b->timing.cells = total_cells;
    b->timing.dir_bytes = dir_bytes;
    b->has_run = true;
    if (host_dbg)
        fprintf(stderr, "[swmi host] setup %.1f us, chunks (launch+wait+parse) %.1f us [prepare %.1f, its uploads %.1f, enqueue %.1f, wait %.1f, copy-out %.1f], grouping %.1f us\n",
                us(h0, h1), us(h1, h2), rs.prep_us, rs.prep_upload_us, rs.enqueue_us, rs.wait_us, rs.copyout_us, us(h2, now()));
    return SWMI_OK;
}

// ---- asynchronous run: the same swmi_batch_run on the context's own host thread -------------------------------
// The caller gets its thread back while the GPU works (a Spark task can prepare its next partition, bench.py's rank
// can do the previous step's reduce).  The helper thread spins briefly between jobs, so back-to-back runs start
// without a wake-up latency, and sleeps when the context stays idle.
static void swmi_worker_loop(swmi_ctx *ctx) {
    for (;;) {
        int st = 0;
        for (int spin = 0; spin < 4000 && (st = ctx->job_state.load(std::memory_order_acquire)) != 1 && st != 3; ++spin)
            __builtin_ia32_pause();                  // back-to-back runs start without a wake-up; an idle context blocks
        if (st != 1 && st != 3) {
            std::unique_lock<std::mutex> lk(ctx->job_mu);
            ctx->job_cv.wait(lk, [&] { const int v = ctx->job_state.load(); return v == 1 || v == 3; });
            st = ctx->job_state.load();
        }
        if (st == 3) return;
        const int rc = swmi_batch_run(ctx, ctx->job_batch, &ctx->job_params);
        ctx->job_rc = rc;
        ctx->job_err = rc ? swmi_last_error() : "";
        { std::lock_guard<std::mutex> lk(ctx->job_mu); ctx->job_state.store(2, std::memory_order_release); }
        ctx->job_cv.notify_all();
    }
}

extern "C" int swmi_batch_run_async(swmi_ctx *ctx, swmi_batch *b, const swmi_params *p) {
    if (!ctx || !b || !p) return fail(SWMI_ERR_INVALID, "null argument");
    if (ctx->job_state.load(std::memory_order_acquire) != 0)
        return fail(SWMI_ERR_INVALID, "a run is already in flight on this context: call swmi_batch_wait first");
    if (!ctx->worker.joinable()) ctx->worker = std::thread(swmi_worker_loop, ctx);
    ctx->job_batch = b;
    ctx->job_params = *p;
    { std::lock_guard<std::mutex> lk(ctx->job_mu); ctx->job_state.store(1, std::memory_order_release); }
    ctx->job_cv.notify_all();
    return SWMI_OK;
}

extern "C" int swmi_batch_wait(swmi_ctx *ctx) {
    if (!ctx) return fail(SWMI_ERR_INVALID, "null argument");
    int st = ctx->job_state.load(std::memory_order_acquire);
    if (st == 0) return fail(SWMI_ERR_INVALID, "no run in flight on this context");
    for (int spin = 0; spin < 20000 && st != 2; ++spin) {           // a short bounded spin, then block
        __builtin_ia32_pause();
        st = ctx->job_state.load(std::memory_order_acquire);
    }
    if (st != 2) {
        std::unique_lock<std::mutex> lk(ctx->job_mu);
        ctx->job_cv.wait(lk, [&] { return ctx->job_state.load() == 2; });
    }
    const int rc = ctx->job_rc;
    const std::string err = ctx->job_err;
    ctx->job_state.store(0, std::memory_order_release);
    if (rc) return fail(rc, "%s", err.c_str());
    return SWMI_OK;
}

extern "C" int swmi_batch_mode(const swmi_batch *b, int *mode) {
    if (!b || !mode) return fail(SWMI_ERR_INVALID, "null argument");
    if (!b->has_run) return fail(SWMI_ERR_INVALID, "batch has no results (run it first)");
    *mode = (int)b->eff_mode;
    return SWMI_OK;
}

extern "C" int swmi_batch_timing(const swmi_batch *b, swmi_timing *t) {
    if (!b || !t) return fail(SWMI_ERR_INVALID, "null argument");
    *t = b->timing;
    return SWMI_OK;
}

extern "C" int swmi_align_batch(swmi_ctx *ctx, const swmi_params *p,
                                const uint8_t *ref_bytes, const uint64_t *ref_off, uint32_t n_refs,
                                const uint8_t *read_bytes, const uint64_t *read_off, uint32_t n_reads,
                                swmi_batch **out) {
    int rc = swmi_batch_upload(ctx, ref_bytes, ref_off, n_refs, read_bytes, read_off, n_reads, out);
    if (rc) return rc;
    rc = swmi_batch_run(ctx, *out, p);
    if (rc) { swmi_batch_free(ctx, *out); *out = nullptr; }
    return rc;
}

// ------------------------------------------------------------------------------------------
// result accessors
// ------------------------------------------------------------------------------------------
extern "C" uint64_t swmi_batch_n_pairs(const swmi_batch *b) { return b ? (uint64_t)b->n_refs * b->n_reads : 0; }

static int check_pair(const swmi_batch *b, uint64_t pair) {
    if (!b) return fail(SWMI_ERR_INVALID, "batch is null");
    if (!b->has_run) return fail(SWMI_ERR_INVALID, "batch has no results (run it first)");
    if (pair >= (uint64_t)b->n_refs * b->n_reads) return fail(SWMI_ERR_RANGE, "pair %llu out of range", (unsigned long long)pair);
    return SWMI_OK;
}

extern "C" int swmi_pair_score(const swmi_batch *b, uint64_t pair, int32_t *score) {
    int rc = check_pair(b, pair);
    if (rc) return rc;
    if (score) *score = b->pairs[pair].score;
    return SWMI_OK;
}

extern "C" int swmi_pair_n_alignments(const swmi_batch *b, uint64_t pair, uint64_t *n, uint32_t *flags) {
    int rc = check_pair(b, pair);
    if (rc) return rc;
    if (b->scores_only && !(b->pairs[pair].flags & SWMI_PAIR_DEGENERATE))
        return fail(SWMI_ERR_INVALID, "the batch was run with scores_only = 1: the number of alignments was not computed");
    if (n) *n = b->pairs[pair].n_cells;
    if (flags) *flags = b->pairs[pair].flags;
    return SWMI_OK;
}

// every pair's score and alignment count at once (bulk form of the two accessors above)
extern "C" int swmi_batch_pair_results(const swmi_batch *b, int32_t *scores, uint64_t *n_alignments, uint64_t n) {
    if (!b) return fail(SWMI_ERR_INVALID, "batch is null");
    if (!b->has_run) return fail(SWMI_ERR_INVALID, "batch has no results (run it first)");
    if (n != (uint64_t)b->n_refs * b->n_reads) return fail(SWMI_ERR_RANGE, "the batch has %llu pairs, not %llu",
                                                           (unsigned long long)b->n_refs * b->n_reads, (unsigned long long)n);
    if (b->scores_only && n_alignments)
        return fail(SWMI_ERR_INVALID, "the batch was run with scores_only = 1: pass n_alignments = NULL");
    for (uint64_t k = 0; k < n; k++) {
        if (scores) scores[k] = b->pairs[k].score;
        if (n_alignments) n_alignments[k] = b->pairs[k].n_cells;
    }
    return SWMI_OK;
}

// Pops the traceback "stack" into the two aligned strings (SmithWaterman.java:418-431): ops are stored
// from the max cell backwards, so the strings are built by walking them in reverse.
static void materialise(swmi_batch *b, uint64_t pair, HostAln &a, uint64_t slot) {
    const uint32_t r = (uint32_t)(pair / b->n_reads), q = (uint32_t)(pair % b->n_reads);
    const uint8_t *ref;
    if (b->src_map) {
        auto it = b->src_cache.find(r);
        if (it == b->src_cache.end()) {
            it = b->src_cache.emplace(r, std::vector<uint8_t>()).first;
            swmi_io_read_record(b->src_map, b->src_recs[r], it->second);
        }
        ref = it->second.data();
    } else {
        ref = b->ref_bytes.data() + b->ref_off[r];
    }
    const uint8_t *read = b->read_bytes.data() + b->read_off[q];
    char *sr = b->str_buf.data() + b->str_at[slot], *sq = sr + a.n_ops + 1;
    sr[a.n_ops] = 0; sq[a.n_ops] = 0;
    int64_t i = a.end_i, j = a.end_j;     // 1-based cell of the op being emitted (both >= 1 while ops remain)
    const uint32_t *ops = a.rec;
    // Four ops (one byte of the packed stream) at a time: a table gives, for each of the 256 byte values, how far behind the
    // current cell every op reads its reference / read base (or that it writes '_'), so the four characters of each string do
    // not wait for each other's i, j -- the per-op loop below is one dependent chain per character.
    struct Lut { uint8_t nref, nread, roff[4], qoff[4], rgap[4], qgap[4]; };
    static const Lut *lut = [] {
        static Lut t[256];
        for (int v = 0; v < 256; v++) {
            Lut &L = t[v];
            L.nref = L.nread = 0;
            for (int k = 0; k < 4; k++) {
                const uint32_t op = (v >> (2 * k)) & 3u;
                const bool use_ref = op != SWMI_DIR_I, use_read = op != SWMI_DIR_D;
                L.roff[k] = L.nref; L.qoff[k] = L.nread;
                L.rgap[k] = use_ref ? 0 : 0xFF; L.qgap[k] = use_read ? 0 : 0xFF;
                L.nref += use_ref; L.nread += use_read;
            }
        }
        return t;
    }();
    uint32_t t = 0;
    while (t + 4 <= a.n_ops && i >= 4 && j >= 4) {          // (t is a multiple of 4: the byte does not straddle a dword)
        const Lut &L = lut[(ops[t >> 4] >> (2 * (t & 15))) & 0xFFu];
        const uint32_t pos = a.n_ops - 1 - t;
        for (int k = 0; k < 4; k++) {
            const uint8_t rc = ref[j - 1 - L.roff[k]], qc = read[i - 1 - L.qoff[k]];       // (always inside: i, j >= 4)
            sr[pos - k] = (char)((rc & ~L.rgap[k]) | ('_' & L.rgap[k]));
            sq[pos - k] = (char)((qc & ~L.qgap[k]) | ('_' & L.qgap[k]));
        }
        j -= L.nref; i -= L.nread;
        t += 4;
    }
    for (; t < a.n_ops; t++) {
        const uint32_t op = (ops[t >> 4] >> (2 * (t & 15))) & 3u;
        const uint32_t pos = a.n_ops - 1 - t;
        // branch-free: gaps come at random places of a path (:388-406: alignment takes both, insertion the read's, deletion the reference's)
        const bool use_ref = op != SWMI_DIR_I, use_read = op != SWMI_DIR_D;
        sr[pos] = use_ref ? (char)ref[j - 1] : '_';
        sq[pos] = use_read ? (char)read[i - 1] : '_';
        j -= use_ref; i -= use_read;
    }
    a.str_id = (int64_t)b->str_at[slot];
}

static const char EMPTY_STR[1] = {0};

extern "C" int swmi_pair_alignment(swmi_batch *b, uint64_t pair, uint64_t k,
                                   int32_t *begin, int32_t *end_i, int32_t *end_j,
                                   const char **ref_aln, const char **read_aln, uint32_t *len) {
    int rc = check_pair(b, pair);
    if (rc) return rc;
    if ((rc = ensure_indexed(b))) return rc;
    PairRes &pr = b->pairs[pair];
    if (k >= pr.n_cells) return fail(SWMI_ERR_RANGE, "alignment %llu out of range", (unsigned long long)k);
    if (pr.flags & SWMI_PAIR_DEGENERATE) {
        // every cell, row-major, traces to (0, "", "")  (SmithWaterman.java:378-380)
        const uint32_t n = b->ref_desc[pair / b->n_reads].len;
        if (begin) *begin = 0;
        if (end_i) *end_i = (int32_t)(k / n) + 1;
        if (end_j) *end_j = (int32_t)(k % n) + 1;
        if (ref_aln) *ref_aln = EMPTY_STR;
        if (read_aln) *read_aln = EMPTY_STR;
        if (len) *len = 0;
        return SWMI_OK;
    }
    HostAln &a = b->alns[pr.first + k];
    if (begin) *begin = a.begin;
    if (end_i) *end_i = a.end_i;
    if (end_j) *end_j = a.end_j;
    if (len) *len = a.n_ops;
    if (b->rec_strings) {
        // both strings were written by the traceback kernel right behind the record (swmi_emit.h): pointers only
        const uint32_t *sr = a.rec;
        if (ref_aln) *ref_aln = (const char *)sr;
        if (read_aln) *read_aln = (const char *)(sr + a.n_ops / 4u + 1u);
        return SWMI_OK;
    }
    if (a.str_id < 0) materialise(b, pair, a, pr.first + k);
    if (ref_aln) *ref_aln = b->str_buf.data() + a.str_id;
    if (read_aln) *read_aln = b->str_buf.data() + a.str_id + a.n_ops + 1;
    return SWMI_OK;
}

// Everything OptAlignments returns for every pair of the batch, built in one call: record index + both strings of every
// alignment (SmithWaterman.java:418-431).  The accessors above then only hand out pointers.
extern "C" int swmi_batch_materialise_all(swmi_batch *b, uint64_t *n_alignments, uint64_t *n_chars) {
    if (!b) return fail(SWMI_ERR_INVALID, "batch is null");
    if (!b->has_run) return fail(SWMI_ERR_INVALID, "batch has no results (run it first)");
    int rc = ensure_indexed(b);
    if (rc) return rc;
    uint64_t na = 0, nc = 0;
    const uint64_t np = (uint64_t)b->n_refs * b->n_reads;
    for (uint64_t pair = 0; pair < np; pair++) {
        const PairRes &pr = b->pairs[pair];
        if (pr.flags & SWMI_PAIR_DEGENERATE) { na += pr.n_cells; continue; }     // (0, "", "") each: nothing to build
        for (uint64_t k = 0; k < pr.count; k++) nc += 2ull * b->alns[pr.first + k].n_ops;
        na += pr.count;
    }
    if (b->rec_strings) {          // the kernels wrote every string: the index is all there was to do
        if (n_alignments) *n_alignments = na;
        if (n_chars) *n_chars = nc;
        return SWMI_OK;
    }
    // every string has its own place in str_buf: pairs are built independently, by a few threads when there is enough to do
    // (streamed chunks re-read reference bytes through a cache that is not thread-safe: one thread there)
    auto build = [b](uint64_t lo, uint64_t hi) {
        for (uint64_t pair = lo; pair < hi; pair++) {
            PairRes &pr = b->pairs[pair];
            if (pr.flags & SWMI_PAIR_DEGENERATE) continue;
            for (uint64_t k = 0; k < pr.count; k++) {
                HostAln &a = b->alns[pr.first + k];
                if (a.str_id < 0) materialise(b, pair, a, pr.first + k);
            }
        }
    };
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    static const unsigned max_threads = getenv("SWMI_MAT_THREADS") ? (unsigned)atoi(getenv("SWMI_MAT_THREADS")) : 4u;
    // (measured at 412 k characters: 0.27 ms on one thread, no faster on 4 or 8 -- starting them costs what they save)
    const unsigned nt = (b->src_map || nc < 1000000) ? 1u : std::min<unsigned>({std::max(1u, max_threads), hw, (unsigned)(nc / 500000)});
    if (nt <= 1) {
        build(0, np);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(build, np * t / nt, np * (t + 1) / nt);
        build(0, np / nt);
        for (auto &x : th) x.join();
    }
    if (n_alignments) *n_alignments = na;
    if (n_chars) *n_chars = nc;
    return SWMI_OK;
}

// ------------------------------------------------------------------------------------------
// MapRef view (Distribution.java:403-436)
// ------------------------------------------------------------------------------------------
static int check_ref(const swmi_batch *b, uint32_t ref) {
    if (!b) return fail(SWMI_ERR_INVALID, "batch is null");
    if (!b->has_run) return fail(SWMI_ERR_INVALID, "batch has no results (run it first)");
    if (ref >= b->n_refs) return fail(SWMI_ERR_RANGE, "reference %u out of range", ref);
    return SWMI_OK;
}

extern "C" int swmi_ref_total(const swmi_batch *b, uint32_t ref, int32_t *total) {
    int rc = check_ref(b, ref);
    if (rc) return rc;
    uint32_t t = 0;                                  // Java int arithmetic wraps
    for (uint32_t q = 0; q < b->n_reads; q++) t += (uint32_t)b->pairs[(uint64_t)ref * b->n_reads + q].score;
    if (total) *total = (int32_t)t;
    return SWMI_OK;
}

extern "C" int swmi_ref_totals(const swmi_batch *b, int32_t *totals, uint32_t n) {
    if (!b || !totals) return fail(SWMI_ERR_INVALID, "null argument");
    if (!b->has_run) return fail(SWMI_ERR_INVALID, "batch has no results (run it first)");
    if (n != b->n_refs) return fail(SWMI_ERR_RANGE, "totals has %u entries, the batch has %u references", n, b->n_refs);
    for (uint32_t r = 0; r < b->n_refs; r++) {
        uint32_t t = 0;
        const PairRes *pr = b->pairs.data() + (uint64_t)r * b->n_reads;
        for (uint32_t q = 0; q < b->n_reads; q++) t += (uint32_t)pr[q].score;
        totals[r] = (int32_t)t;
    }
    return SWMI_OK;
}

static void build_ref_view(swmi_batch *b, uint32_t ref) {
    if (b->ref_view_ready[ref]) return;
    b->views_built = true;
    std::vector<SiteRef> &v = b->ref_sites[ref];
    uint64_t deg = 0;
    for (uint32_t q = 0; q < b->n_reads; q++) {
        const uint64_t pair = (uint64_t)ref * b->n_reads + q;
        const PairRes &pr = b->pairs[pair];
        if (pr.flags & SWMI_PAIR_DEGENERATE) { deg += pr.n_cells; continue; }   // begin 0: sorts before every real site
        for (uint64_t k = 0; k < pr.count; k++) v.push_back(SiteRef{pair, k, b->alns[pr.first + k].begin});
    }
    std::stable_sort(v.begin(), v.end(), [](const SiteRef &a, const SiteRef &c) { return a.begin < c.begin; });
    b->ref_degenerate[ref] = deg;
    b->ref_view_ready[ref] = 1;
}

extern "C" int swmi_ref_n_match_sites(swmi_batch *b, uint32_t ref, uint64_t *n) {
    int rc = check_ref(b, ref);
    if (rc) return rc;
    if ((rc = ensure_indexed(b))) return rc;
    build_ref_view(b, ref);
    if (n) *n = b->ref_degenerate[ref] + b->ref_sites[ref].size();
    return SWMI_OK;
}

extern "C" int swmi_ref_match_site(swmi_batch *b, uint32_t ref, uint64_t k, int32_t *begin,
                                   const char **ref_aln, const char **read_aln, uint32_t *len) {
    int rc = check_ref(b, ref);
    if (rc) return rc;
    if ((rc = ensure_indexed(b))) return rc;
    build_ref_view(b, ref);
    const uint64_t deg = b->ref_degenerate[ref];
    if (k < deg) {
        if (begin) *begin = 0;
        if (ref_aln) *ref_aln = EMPTY_STR;
        if (read_aln) *read_aln = EMPTY_STR;
        if (len) *len = 0;
        return SWMI_OK;
    }
    if (k - deg >= b->ref_sites[ref].size()) return fail(SWMI_ERR_RANGE, "match site %llu out of range", (unsigned long long)k);
    const SiteRef &s = b->ref_sites[ref][k - deg];
    return swmi_pair_alignment(b, s.pair, s.k, begin, nullptr, nullptr, ref_aln, read_aln, len);
}

// MapRef's output for a range of references in ONE call: what a per-partition binding (JNI, one call per Spark partition)
// hands back instead of three calls and two array allocations per match site (Distribution.java:419-433).
extern "C" int swmi_ref_sites_packed(swmi_batch *b, uint32_t ref_lo, uint32_t ref_hi,
                                     int32_t *totals, uint64_t *degenerate, uint64_t *site_first,
                                     int32_t *begins, uint32_t *lens, uint64_t *str_off, uint64_t sites_cap,
                                     uint8_t *blob, uint64_t blob_cap, uint64_t *n_sites, uint64_t *blob_bytes) {
    if (!b) return fail(SWMI_ERR_INVALID, "batch is null");
    if (!b->has_run) return fail(SWMI_ERR_INVALID, "batch has no results (run it first)");
    if (ref_lo > ref_hi || ref_hi > b->n_refs) return fail(SWMI_ERR_RANGE, "reference range [%u, %u) out of range", ref_lo, ref_hi);
    int rc = ensure_indexed(b);
    if (rc) return rc;
    // pass 1: counts (always), so that a caller may ask for the sizes first (begins == NULL or capacities too small)
    uint64_t ns = 0, nb = 0;
    for (uint32_t r = ref_lo; r < ref_hi; r++) {
        build_ref_view(b, r);
        for (const SiteRef &sr : b->ref_sites[r]) nb += 2ull * b->alns[b->pairs[sr.pair].first + sr.k].n_ops;
        ns += b->ref_sites[r].size();
    }
    if (n_sites) *n_sites = ns;
    if (blob_bytes) *blob_bytes = nb;
    const bool fill = begins && lens && str_off && (blob || nb == 0) && sites_cap >= ns && blob_cap >= nb;
    uint64_t s_at = 0, c_at = 0;
    for (uint32_t r = ref_lo; r < ref_hi; r++) {
        if (totals) { int32_t t = 0; (void)swmi_ref_total(b, r, &t); totals[r - ref_lo] = t; }
        if (degenerate) degenerate[r - ref_lo] = b->ref_degenerate[r];           // leading (0, "", "") sites, not listed one by one
        if (site_first) site_first[r - ref_lo] = s_at;
        if (fill)
            for (const SiteRef &sr : b->ref_sites[r]) {
                const PairRes &pr = b->pairs[sr.pair];
                HostAln &a = b->alns[pr.first + sr.k];
                const char *ra, *qa;
                if (b->rec_strings) {
                    const uint32_t *w = a.rec;
                    ra = (const char *)w; qa = (const char *)(w + a.n_ops / 4u + 1u);
                } else {
                    if (a.str_id < 0) materialise(b, sr.pair, a, pr.first + sr.k);
                    ra = b->str_buf.data() + a.str_id; qa = ra + a.n_ops + 1;
                }
                begins[s_at] = a.begin; lens[s_at] = a.n_ops; str_off[s_at] = c_at;
                memcpy(blob + c_at, ra, a.n_ops);
                memcpy(blob + c_at + a.n_ops, qa, a.n_ops);
                c_at += 2ull * a.n_ops;
                s_at++;
            }
        else s_at += b->ref_sites[r].size();
    }
    if (site_first) site_first[ref_hi - ref_lo] = s_at;
    if (!fill && begins) return fail(SWMI_ERR_RANGE, "%llu sites / %llu string bytes do not fit the buffers (%llu / %llu)",
                                     (unsigned long long)ns, (unsigned long long)nb, (unsigned long long)sites_cap, (unsigned long long)blob_cap);
    return SWMI_OK;
}


// ------------------------------------------------------------------------------------------
// streaming: a reference set too large for one batch, cut into chunks that flow through the GPU
// ------------------------------------------------------------------------------------------
// The reference loads a whole FASTA file (InOutOps.GetRefSeqs, src/sw/InOutOps.java:115-168), then maps it
// (src/sw/Distribution.java:329-338).  Here the file is cut into segments of records; host threads parse segments into
// pinned buffers while `slots` chunk workers -- each with its own context, HIP stream and device buffers -- upload
// (raw bytes, canonicalised on the GPU), run the full path and keep the results: chunk k+1's parse and H2D overlap chunk
// k's kernels and chunk k-1's result handling.  A chunk's results stay accessible as a results-only swmi_batch.
struct StreamChunk {
    uint32_t id = 0;
    PinnedBuf *buf = nullptr;                 // raw sequence bytes of the chunk (pinned: H2D at full PCIe rate)
    std::vector<uint64_t> off;                // n_refs + 1
    std::vector<swmi_io_recpos> recs;         // file sources only
    std::vector<uint8_t> keep;                // memory sources: the bytes kept for the alignment strings
};

struct swmi_stream {
    swmi_ctx *owner = nullptr;
    swmi_params params{};
    std::vector<uint8_t> read_bytes;
    std::vector<uint64_t> read_off;
    uint32_t n_reads = 0;
    uint64_t chunk_bytes = 32ull << 20;
    bool keep_records = true;
    // slots
    struct Slot { swmi_ctx *ctx = nullptr; swmi_batch *shell = nullptr; std::thread th; };
    std::vector<Slot> slots;
    // pinned buffer pool and the queue of parsed chunks
    std::vector<std::unique_ptr<PinnedBuf>> bufs;
    std::deque<PinnedBuf *> free_bufs;
    std::deque<StreamChunk *> ready;
    std::mutex mu;
    std::condition_variable cv_free, cv_ready;
    bool closing = false;
    std::atomic<int> err{0};                   // (read by the workers and parsers outside `mu`)
    std::string err_msg;
    uint32_t next_id = 0;                      // chunk ids in reference order
    uint32_t in_flight = 0;
    std::vector<swmi_batch *> results;         // by chunk id
    std::vector<uint64_t> first_ref;           // after finish: global index of a chunk's first reference
    bool finished = false;
    // file source
    const uint8_t *map_p = nullptr; size_t map_n = 0; int map_fd = -1;
    swmi_stream_stats stats{};
};

static void stream_fail(swmi_stream *s, int rc, const std::string &msg) {
    std::lock_guard<std::mutex> g(s->mu);
    if (!s->err.load()) { s->err_msg = msg; s->err.store(rc); }
    s->cv_free.notify_all(); s->cv_ready.notify_all();
}

// one chunk through one slot: upload, run, move the results out of the slot's shell
static int stream_process(swmi_stream *s, swmi_stream::Slot &sl, StreamChunk *c) {
    swmi_batch *b = sl.shell;
    const uint32_t n_refs = (uint32_t)(c->off.size() - 1);
    int rc;
    {
        std::lock_guard<std::mutex> g(sl.ctx->mu);
        HIP_TRY(hipSetDevice(sl.ctx->device));
        b->n_refs = n_refs; b->n_reads = s->n_reads;
        b->ref_off = c->off;
        b->read_off = s->read_off;
        if ((uint64_t)n_refs * s->n_reads >= (1ull << 32)) return fail(SWMI_ERR_UNSUPPORTED, "more than 2^32-1 pairs in one chunk");
        const auto t0 = std::chrono::steady_clock::now();
        if ((rc = upload_device(sl.ctx, b, sl.ctx->stream, (const uint8_t *)c->buf->p, s->read_bytes.data()))) return rc;
        const double up = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::lock_guard<std::mutex> g2(s->mu);
        s->stats.upload_ms += up;
    }
    const auto t1 = std::chrono::steady_clock::now();
    if ((rc = swmi_batch_run(sl.ctx, b, &s->params))) return rc;
    const double run = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    const auto t2 = std::chrono::steady_clock::now();
    // results-only batch of this chunk
    std::unique_ptr<swmi_batch> r(new swmi_batch);
    r->n_refs = n_refs; r->n_reads = s->n_reads;
    r->ref_off = std::move(b->ref_off);
    r->read_off = s->read_off;
    r->read_bytes = s->read_bytes;
    r->ref_desc = b->ref_desc; r->read_desc = b->read_desc;
    r->params = b->params; r->has_run = true; r->eff_mode = b->eff_mode;
    r->work = std::move(b->work); b->work.clear(); b->work_mode = -1;
    r->work_mode = (int)r->eff_mode;
    r->pairs = std::move(b->pairs);
    if (s->keep_records) {
        (void)settle_raw(b);
        r->raw = std::move(b->raw);
        r->rtab = std::move(b->rtab);
        r->raw_chunks = std::move(b->raw_chunks);
    } else {
        r->records_dropped = true;
        b->raw_ext = nullptr; b->rtab_ext = nullptr;
    }
    r->rec_strings = b->rec_strings;
    r->scores_only = b->scores_only;
    r->indexed = false;
    r->ref_view_ready.assign(n_refs, 0);
    r->ref_sites.assign(n_refs, {});
    r->ref_degenerate.assign(n_refs, 0);
    r->timing = b->timing;
    if (!c->recs.empty()) { r->src_map = s->map_p; r->src_recs = std::move(c->recs); }
    else r->ref_bytes = std::move(c->keep);
    b->pairs.clear(); b->raw.clear(); b->rtab.clear(); b->raw_chunks.clear(); b->has_run = false;
    std::lock_guard<std::mutex> g(s->mu);
    {
        static const bool host_dbg = getenv("SWMI_DEBUG_HOST") != nullptr;
        if (host_dbg) fprintf(stderr, "[swmi stream] chunk %u: %u refs, run %.1f ms, results moved out of the slot in %.1f ms\n", c->id, n_refs, run,
                              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count());
    }
    if (s->results.size() <= c->id) s->results.resize(c->id + 1, nullptr);
    s->results[c->id] = r.release();
    s->stats.run_ms += run;
    s->stats.gpu_sweep_ms += b->timing.fill_ms;
    s->stats.gpu_traceback_ms += b->timing.traceback_ms;
    s->stats.cells += b->timing.cells;
    s->stats.chunks++;
    return SWMI_OK;
}

static void stream_worker(swmi_stream *s, size_t slot) {
    swmi_stream::Slot &sl = s->slots[slot];
    for (;;) {
        StreamChunk *c = nullptr;
        {
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv_ready.wait(lk, [&] { return !s->ready.empty() || s->closing; });
            if (s->ready.empty()) return;
            c = s->ready.front();
            s->ready.pop_front();
        }
        int rc = s->err.load() ? s->err.load() : stream_process(s, sl, c);
        if (rc && !s->err.load()) stream_fail(s, rc, swmi_last_error());
        {
            std::lock_guard<std::mutex> g(s->mu);
            s->free_bufs.push_back(c->buf);
            s->in_flight--;
        }
        s->cv_free.notify_all();
        delete c;
    }
}

extern "C" int swmi_stream_open(swmi_ctx *ctx, const swmi_params *p, const uint8_t *read_bytes, const uint64_t *read_off,
                                uint32_t n_reads, uint32_t slots, uint64_t chunk_bytes, swmi_stream **out) {
    if (!ctx || !p || !out) return fail(SWMI_ERR_INVALID, "null argument");
    *out = nullptr;
    int rc;
    if ((rc = check_offsets(read_off, n_reads, "read"))) return rc;
    if (n_reads && read_off[n_reads] && !read_bytes) return fail(SWMI_ERR_INVALID, "sequence bytes are null");
    if (slots == 0) slots = 3;
    if (slots > 8) return fail(SWMI_ERR_INVALID, "at most 8 slots");
    std::unique_ptr<swmi_stream> s(new swmi_stream);
    s->owner = ctx;
    s->params = *p;
    s->n_reads = n_reads;
    s->read_off.assign(read_off, read_off + n_reads + 1);
    s->read_bytes.assign(read_bytes, read_bytes + read_off[n_reads]);
    if (chunk_bytes) s->chunk_bytes = std::max<uint64_t>(chunk_bytes, 1 << 16);
    s->keep_records = ctx->stream_keep_records != 0;
    s->slots.resize(slots);
    for (auto &sl : s->slots) {
        if ((rc = swmi_create(ctx->device, &sl.ctx))) { swmi_stream_close(s.release()); return rc; }
        // the slot contexts run what the caller's context would run
        sl.ctx->cell_cap = ctx->cell_cap; sl.ctx->cell_cap_set = ctx->cell_cap_set; sl.ctx->max_workspace_bytes = ctx->max_workspace_bytes;
        sl.ctx->profiling = ctx->profiling; sl.ctx->mode = ctx->mode; sl.ctx->zero_copy = ctx->zero_copy;
        sl.ctx->tb_split = ctx->tb_split; sl.ctx->col_chunks = ctx->col_chunks; sl.ctx->resident = ctx->resident; sl.ctx->tfused = ctx->tfused;
        sl.ctx->auto_ties_x100 = ctx->auto_ties_x100; sl.ctx->arena_words_per_pair = ctx->arena_words_per_pair;
        sl.ctx->device_strings = ctx->device_strings; sl.ctx->scores_only = ctx->scores_only;
        sl.ctx->spin_us = 50;                    // (a chunk takes milliseconds: the slot threads mostly block)
        sl.shell = new swmi_batch;
    }
    // pinned buffers: one being parsed into per parser thread (up to 6), one per slot in flight, two queued
    const size_t n_bufs = slots + 8;
    for (size_t k = 0; k < n_bufs; k++) {
        std::unique_ptr<PinnedBuf> pb(new PinnedBuf);
        if ((rc = pb->reserve(s->chunk_bytes + (1 << 20)))) { swmi_stream_close(s.release()); return rc; }
        s->free_bufs.push_back(pb.get());
        s->bufs.push_back(std::move(pb));
    }
    for (size_t k = 0; k < s->slots.size(); k++) s->slots[k].th = std::thread(stream_worker, s.get(), k);
    *out = s.release();
    return SWMI_OK;
}

// takes a free pinned buffer of at least `bytes` (blocks while all are in use)
static PinnedBuf *stream_take_buf(swmi_stream *s, uint64_t bytes) {
    PinnedBuf *pb = nullptr;
    {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv_free.wait(lk, [&] { return !s->free_bufs.empty() || s->err; });
        if (s->err) return nullptr;
        pb = s->free_bufs.front();
        s->free_bufs.pop_front();
    }
    if (pb->reserve(bytes)) {                  // (a record longer than a chunk: the buffer grows)
        stream_fail(s, SWMI_ERR_NOMEM, swmi_last_error());
        std::lock_guard<std::mutex> g(s->mu);
        s->free_bufs.push_back(pb);
        return nullptr;
    }
    return pb;
}

static void stream_submit(swmi_stream *s, StreamChunk *c) {
    {
        std::lock_guard<std::mutex> g(s->mu);
        // chunks are handed to the slots in id order: a later chunk parsed first waits in `ready` (sorted insert)
        auto it = s->ready.begin();
        while (it != s->ready.end() && (*it)->id < c->id) ++it;
        s->ready.insert(it, c);
        s->in_flight++;
    }
    s->cv_ready.notify_one();
}

extern "C" int swmi_stream_push(swmi_stream *s, const uint8_t *ref_bytes, const uint64_t *ref_off, uint32_t n_refs) {
    if (!s) return fail(SWMI_ERR_INVALID, "stream is null");
    if (s->finished) return fail(SWMI_ERR_INVALID, "the stream is finished");
    int rc;
    if ((rc = check_offsets(ref_off, n_refs, "reference"))) return rc;
    if (n_refs && ref_off[n_refs] && !ref_bytes) return fail(SWMI_ERR_INVALID, "sequence bytes are null");
    // cut the caller's references into chunks of about chunk_bytes
    uint32_t lo = 0;
    while (lo < n_refs) {
        uint32_t hi = lo;
        while (hi < n_refs && (hi == lo || ref_off[hi + 1] - ref_off[lo] <= s->chunk_bytes)) hi++;
        const uint64_t bytes = ref_off[hi] - ref_off[lo];
        PinnedBuf *pb = stream_take_buf(s, std::max<uint64_t>(bytes, 16));
        if (!pb) return fail(s->err.load() ? s->err.load() : SWMI_ERR_NOMEM, "%s", s->err_msg.c_str());
        StreamChunk *c = new StreamChunk;
        c->buf = pb;
        c->off.resize(hi - lo + 1);
        for (uint32_t k = lo; k <= hi; k++) c->off[k - lo] = ref_off[k] - ref_off[lo];
        memcpy(pb->p, ref_bytes + ref_off[lo], bytes);
        c->keep.assign(ref_bytes + ref_off[lo], ref_bytes + ref_off[hi]);        // for the alignment strings
        { std::lock_guard<std::mutex> g(s->mu); c->id = s->next_id++; }
        stream_submit(s, c);
        lo = hi;
    }
    return s->err ? fail(s->err, "%s", s->err_msg.c_str()) : SWMI_OK;
}

extern "C" int swmi_stream_push_file(swmi_stream *s, const char *path, const char *delimiter, uint32_t parse_threads) {
    if (!s || !path || !delimiter) return fail(SWMI_ERR_INVALID, "null argument");
    if (s->finished) return fail(SWMI_ERR_INVALID, "the stream is finished");
    if (s->map_p) return fail(SWMI_ERR_UNSUPPORTED, "one file per stream");
    const auto t0 = std::chrono::steady_clock::now();
    int rc = swmi_io_map(path, &s->map_p, &s->map_n, &s->map_fd);
    if (rc) return rc;
    const uint8_t *p = s->map_p;
    const size_t n = s->map_n;
    if (n == 0) return fail(SWMI_ERR_INVALID, "reference file has no record: %s", path);                // ref is null at InOutOps.java:153
    if (swmi_io_next_record(p, n, 0, delimiter) != 0)
        return fail(SWMI_ERR_INVALID, "reference file does not start with a metadata line: %s", path);  // seq is null at :148
    // segment boundaries: record starts about chunk_bytes apart
    std::vector<size_t> cut{0};
    while (cut.back() < n) {
        const size_t want = cut.back() + s->chunk_bytes;
        size_t nxt = want >= n ? n : swmi_io_next_record(p, n, want, delimiter);
        if (nxt <= cut.back()) nxt = n;
        cut.push_back(nxt);
    }
    const uint32_t n_seg = (uint32_t)(cut.size() - 1);
    uint32_t id0;
    { std::lock_guard<std::mutex> g(s->mu); id0 = s->next_id; s->next_id += n_seg; }
    if (parse_threads == 0) parse_threads = 6;
    parse_threads = std::min<uint32_t>(parse_threads, std::max<uint32_t>(1, n_seg));
    std::atomic<uint32_t> next{0};
    std::string delim(delimiter);
    auto parser = [&]() {
        for (;;) {
            const uint32_t k = next.fetch_add(1);
            if (k >= n_seg || s->err) return;
            // (segments are taken in id order; the workers pop the lowest ready id, so later segments never starve earlier ones)
            const auto p0 = std::chrono::steady_clock::now();
            PinnedBuf *pb = stream_take_buf(s, std::max<uint64_t>(cut[k + 1] - cut[k], 16));
            if (!pb) return;
            StreamChunk *c = new StreamChunk;
            c->id = id0 + k;
            c->buf = pb;
            int prc = swmi_io_parse_segment(p, cut[k], cut[k + 1], delim.c_str(), (uint8_t *)pb->p, c->off, c->recs);
            if (prc) {
                stream_fail(s, prc, swmi_last_error());
                std::lock_guard<std::mutex> g(s->mu);
                s->free_bufs.push_back(pb);
                delete c;
                return;
            }
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - p0).count();
            { std::lock_guard<std::mutex> g(s->mu); s->stats.parse_ms += ms; s->stats.bytes += c->off.back(); }
            stream_submit(s, c);
        }
    };
    std::vector<std::thread> th;
    for (uint32_t k = 0; k < parse_threads; k++) th.emplace_back(parser);
    for (auto &t : th) t.join();
    s->stats.push_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return s->err ? fail(s->err, "%s", s->err_msg.c_str()) : SWMI_OK;
}

extern "C" int swmi_stream_finish(swmi_stream *s) {
    if (!s) return fail(SWMI_ERR_INVALID, "stream is null");
    {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv_free.wait(lk, [&] { return s->in_flight == 0 || s->err; });
    }
    if (s->err) return fail(s->err, "%s", s->err_msg.c_str());
    if (!s->finished) {
        s->first_ref.assign(s->results.size() + 1, 0);
        for (size_t k = 0; k < s->results.size(); k++) {
            if (!s->results[k]) return fail(SWMI_ERR_HIP, "chunk %zu has no results", k);
            s->first_ref[k + 1] = s->first_ref[k] + s->results[k]->n_refs;
        }
        s->finished = true;
    }
    return SWMI_OK;
}

extern "C" uint64_t swmi_stream_n_refs(const swmi_stream *s) { return s && s->finished ? s->first_ref.back() : 0; }
extern "C" uint32_t swmi_stream_n_chunks(const swmi_stream *s) { return s && s->finished ? (uint32_t)s->results.size() : 0; }

extern "C" int swmi_stream_chunk(swmi_stream *s, uint32_t k, swmi_batch **batch, uint64_t *first_ref) {
    if (!s) return fail(SWMI_ERR_INVALID, "stream is null");
    if (!s->finished) return fail(SWMI_ERR_INVALID, "call swmi_stream_finish first");
    if (k >= s->results.size()) return fail(SWMI_ERR_RANGE, "chunk %u out of range", k);
    if (batch) *batch = s->results[k];
    if (first_ref) *first_ref = s->first_ref[k];
    return SWMI_OK;
}

extern "C" int swmi_stream_totals(const swmi_stream *s, int32_t *totals, uint64_t n) {
    if (!s || !totals) return fail(SWMI_ERR_INVALID, "null argument");
    if (!s->finished) return fail(SWMI_ERR_INVALID, "call swmi_stream_finish first");
    if (n != s->first_ref.back()) return fail(SWMI_ERR_RANGE, "the stream holds %llu references, not %llu",
                                               (unsigned long long)s->first_ref.back(), (unsigned long long)n);
    for (size_t k = 0; k < s->results.size(); k++) {
        int rc = swmi_ref_totals(s->results[k], totals + s->first_ref[k], s->results[k]->n_refs);
        if (rc) return rc;
    }
    return SWMI_OK;
}

// metadata line of a streamed reference (file sources), copied into buf (NUL-terminated, truncated to cap)
extern "C" int swmi_stream_metadata(const swmi_stream *s, uint64_t ref, char *buf, size_t cap) {
    if (!s || !buf || !cap) return fail(SWMI_ERR_INVALID, "null argument");
    if (!s->finished) return fail(SWMI_ERR_INVALID, "call swmi_stream_finish first");
    if (ref >= s->first_ref.back()) return fail(SWMI_ERR_RANGE, "reference %llu out of range", (unsigned long long)ref);
    const size_t k = (size_t)(std::upper_bound(s->first_ref.begin(), s->first_ref.end(), ref) - s->first_ref.begin()) - 1;
    const swmi_batch *b = s->results[k];
    buf[0] = 0;
    if (b->src_map) {
        const swmi_io_recpos &r = b->src_recs[ref - s->first_ref[k]];
        const size_t len = std::min<size_t>(cap - 1, r.meta_len);
        memcpy(buf, b->src_map + r.meta_pos, len);
        buf[len] = 0;
    }
    return SWMI_OK;
}

extern "C" int swmi_stream_get_stats(const swmi_stream *s, swmi_stream_stats *st) {
    if (!s || !st) return fail(SWMI_ERR_INVALID, "null argument");
    *st = s->stats;
    return SWMI_OK;
}

extern "C" void swmi_stream_close(swmi_stream *s) {
    if (!s) return;
    {
        std::lock_guard<std::mutex> g(s->mu);
        s->closing = true;
    }
    s->cv_ready.notify_all();
    for (auto &sl : s->slots) if (sl.th.joinable()) sl.th.join();
    for (auto *c : s->ready) { delete c; }
    for (auto &sl : s->slots) {
        if (sl.shell) swmi_batch_free(sl.ctx, sl.shell);
        if (sl.ctx) swmi_destroy(sl.ctx);
    }
    for (auto *r : s->results) if (r) swmi_batch_free(nullptr, r);
    for (auto &pb : s->bufs) pb->release();
    if (s->map_p) swmi_io_unmap(s->map_p, s->map_n, s->map_fd);
    delete s;
}
