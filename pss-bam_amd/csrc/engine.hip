// pss-bam_amd/csrc/engine.hip -- the C ABI declared in include/pssbam_hip.h.
//
// Host side of the MI355X tally engine: device-resident genome, BAM refID -> contig map
// with find_seq semantics, double-buffered H2D staging of record blocks, kernel choice
// and launch geometry, u64 counter block.  Written for gfx950 only; there is no CPU path.
#include "../../include/pssbam_hip.h"
#include "../../include/fasta-genome-io.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <string>
#include <thread>
#include <atomic>
#include <memory>
#include <mutex>
#include <vector>

#include "tally_kernels.h"

using namespace pssbam;

// --------------------------------------------------------------------------------------
// errors
// --------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static double feed_now() {   // host seconds, for the PSSBAM_STATS lines
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return fail(PSSBAM_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

extern "C" const char *pssbam_last_error(void) { return g_err; }

static bool device_is_gfx950(int dev) {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, dev) != hipSuccess) return false;
    return strncmp(pr.gcnArchName, "gfx950", 6) == 0;
}

extern "C" int pssbam_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int d = 0; d < n; d++) ok += device_is_gfx950(d) ? 1 : 0;
    return ok;
}

extern "C" int pssbam_warmup(int device) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(nullptr));  // forces the context
    return PSSBAM_OK;
}

// --------------------------------------------------------------------------------------
// engine
// --------------------------------------------------------------------------------------
struct Slot {  // one in-flight host-submitted block
    uint8_t *d_recs = nullptr;
    size_t recs_cap = 0;
    uint32_t *d_offs = nullptr;
    size_t offs_cap = 0;  // entries
    hipEvent_t copy_begin = nullptr, copied = nullptr, consumed = nullptr;
    bool busy = false;
    bool timed = false;       // copy_begin/copied hold an unread H2D duration
    uint64_t ticket = 0;      // the submit that last used the slot
};

// per super-batch: the output of ONE round of the inflate kernel's 65 536 lanes (a block per lane: 4.28 GB at htslib's
// 65 280-byte blocks) and room for its compressed bytes.  Super-batches live in a ring of slots that grows while the tally
// launches wait for the genome (pssbam_engine_feed_open) and is three deep otherwise.
static constexpr uint64_t FEED_OUT_TARGET = 4400ull << 20, FEED_COMP_CAP = 2ull << 30, FEED_OUT_SLACK = 160ull << 20,
                          FEED_OVERSHOOT = 64ull << 20,   // a submit may run this far past the target before the super-batch is cut
                          FEED_GAP = 16ull << 20;   // room in front of a super-batch's data for the record the previous one ended in
static constexpr int FEED_SLOTS_READY = 3, FEED_SLOTS_MAX = 40;

struct FeedAcc {  // one super-batch of compressed blocks: assembled chunk by chunk, then inflated in ONE launch
    uint8_t *d_comp = nullptr;      // compressed bytes of the chunks, back to back (each 16-byte aligned)
    size_t comp_cap = 0;
    uint64_t comp_used = 0;
    std::vector<pssbam_bgzf_block> blocks;   // in_off into d_comp, out_off into d_out (contiguous from FEED_GAP on)
    std::vector<uint32_t> sub_first;         // first block of every tally sub-batch (< 4 GiB of records each)
    uint64_t out_used = 0, sub_bytes = 0;
    void *d_blocks = nullptr;                // pssbam::BgzfBlock[]
    pssbam_bgzf_block *h_blocks = nullptr;   // page-locked copy of `blocks` for the upload (a pageable source makes hipMemcpyAsync wait for the stream)
    size_t h_blocks_cap = 0;
    // per block: chain pieces (first record start, records, end, last record start), suffix minimum, counts, bases
    uint64_t *d_a = nullptr, *d_e = nullptr, *d_last = nullptr, *d_nexta = nullptr;
    uint32_t *d_n = nullptr, *d_counts = nullptr, *d_base = nullptr;
    size_t blocks_cap = 0;
    uint8_t *d_out = nullptr;                // [0, FEED_GAP): the partial record carried over from the previous super-batch
    size_t out_cap = 0;
    uint32_t *d_offs = nullptr;
    size_t offs_cap = 0;
    uint32_t *d_nrecs = nullptr;
    size_t nrecs_cap = 0;
    uint64_t *d_chain = nullptr;             // [0] where this super-batch's chain starts, [1] where its tail starts, [2] links broken, [3] links repaired
    hipEvent_t consumed = nullptr, copies_done = nullptr, copies_done2 = nullptr, inflated = nullptr, inflate_done = nullptr;
    bool busy = false;                       // flushed; its buffers are in use until `consumed`
    bool held = false;                       // ... and its tally launches still wait for the genome (no `consumed` yet)
    uint64_t flush_seq = 0;                  // order of the flushes (the oldest busy slot frees first)
};

struct DeferredTally {   // a tally launch that waits for set_genome + set_references (pssbam_engine_feed_open)
    int slot;
    uint64_t sub_base, sub_len, offs_at;
    uint32_t n_bound, k;
    uint64_t sample_off;
    bool last_of_slot;
};

// Host ranges page-locked for genome uploads, shared by the engines of a process (one engine per GPU uploads the same
// contigs): registered once, released when the last upload that uses them has completed.
namespace {
struct PinEntry { size_t len; int refs; };
std::map<const void *, PinEntry> g_pins;
std::mutex g_pins_mu;
bool pin_acquire(const void *p, size_t len) {
    std::lock_guard<std::mutex> lk(g_pins_mu);
    auto it = g_pins.find(p);
    if (it != g_pins.end()) { it->second.refs++; return true; }
    if (hipHostRegister((void *)p, len, hipHostRegisterPortable) != hipSuccess) { (void)hipGetLastError(); return false; }
    g_pins[p] = PinEntry{len, 1};
    return true;
}
void pin_release(const void *p) {
    std::lock_guard<std::mutex> lk(g_pins_mu);
    auto it = g_pins.find(p);
    if (it == g_pins.end()) return;
    if (--it->second.refs == 0) { (void)hipHostUnregister((void *)p); g_pins.erase(it); }
}
}  // namespace

struct pssbam_engine {
    pssbam_config cfg{};
    std::string up_ctx, down_ctx, rg;
    bool has_rg = false;
    int device = 0;
    int n_cu = 0;
    hipStream_t stream = nullptr, copy_stream = nullptr, copy_stream2 = nullptr;
    hipStream_t inflate_stream[2] = {nullptr, nullptr};   // PSSBAM_FEED_INFLATE_STREAMS=2: the feed's inflate launches take turns on these (bgzf_api.h feed_flush; off by default)
    hipEvent_t feed_base_ev = nullptr;                    // time zero of the feed's kernel intervals (feed_status: their union)
    hipStream_t genome_stream = nullptr;   // genome upload + encode + pack: beside whatever the engine's stream runs
    // A HIP stream costs 20-30 ms to create (a hardware queue each): only the engine's own is made before create returns;
    // the copy streams and the genome's are made by a helper thread while the first blocks go out -- those are copied on
    // the engine's stream, in front of the kernel that reads them anyway -- and adopted when they are there (late_streams()).
    std::thread late_thread;
    std::atomic<int> late_ready{0};
    std::mutex late_mu;
    hipStream_t late_copy1 = nullptr, late_copy2 = nullptr, late_genome = nullptr;
    bool late_adopted = false;
    hipEvent_t genome_ready = nullptr;
    bool genome_wait_pending = false;      // the next tally launch makes the engine's stream wait for genome_ready
    std::vector<const void *> genome_pins; // host contigs page-locked for an upload still in flight
    std::vector<void *> retired;           // device buffers replaced while kernels might still read them: freed at destroy
    hipEvent_t copied2 = nullptr;   // second half of a split H2D copy
    bool own_stream = false;

    // genome
    uint8_t *d_genome = nullptr;
    uint32_t *d_genome4 = nullptr;  // 4-bit image for the tiled kernel's window gathers
    uint32_t acgt_ctx = 0;
    uint64_t genome_bytes = 0;
    std::vector<uint64_t> contig_start;  // per sorted contig
    std::vector<uint32_t> contig_len;
    std::vector<std::string> contig_ids;  // sorted by strcmp, like Genome.seqs
    int32_t star_contig = -1;
    // references
    uint4 *d_ref_info = nullptr;  // n_ref + 1 entries (last = RNAME "*")
    uint4 *h_ref_info = nullptr;  // the same in page-locked host memory: a kernel copies it over (below)
    size_t h_ref_cap = 0;
    int32_t n_ref = 0;
    bool have_refs = false;
    // -R
    uint8_t *d_rg = nullptr;
    // counters
    unsigned long long *d_counters = nullptr;      // block in use (own or caller-bound)
    unsigned long long *d_counters_own = nullptr;  // the engine's own allocation
    size_t n_counters = 0;
    uint32_t rows = 0, off_rev = 0, off_k5 = 0, off_k3 = 0, off_stats = 0;
    uint64_t n_bins = 0;
    // staging
    Slot slots[2];
    int next_slot = 0;
    uint64_t ticket_seq = 0;
    double h2d_ms = 0.0;      // summed H2D copy durations (events on the copy stream)
    uint64_t h2d_bytes = 0;
    // device-side inflate feed
    std::vector<FeedAcc *> feed;       // ring of super-batch slots
    int cur_feed = -1, spare_feed = -1; // slot being assembled; slot acquired ahead for a submit that will spill over
    uint64_t flush_seq = 0;
    bool feed_opened = false;          // pssbam_engine_feed_open: submit_bgzf may precede set_genome / set_references
    int32_t feed_n_ref = 0;            // ... with this many references in the BAM header
    uint64_t feed_mem_budget = 0;      // device bytes the ring may take while tallies are deferred (0: not worked out yet)
    std::vector<DeferredTally> deferred;
    uint8_t *d_carry = nullptr;        // the partial record a super-batch ended with, on its way into the next slot's gap
    uint8_t *h_handoff = nullptr;      // page-locked [8 + FEED_GAP]: the partial record ANOTHER engine's run ended with arrives here (feed_handoff)
    hipEvent_t handoff_ev = nullptr;   // recorded on this engine's stream behind the hand-off of its own tail
    uint64_t feed_out_target = FEED_OUT_TARGET, feed_comp_cap = FEED_COMP_CAP;   // per super-batch
    double feed_t_alloc = 0, feed_t_wait_busy = 0, feed_t_flush = 0;   // host seconds inside the feed (PSSBAM_STATS)
    double feed_t0 = 0, feed_t_first_flush = -1;                        // host clock of the first submit_bgzf; first flush, seconds after it
    uint64_t feed_slots_allocated = 0, feed_deferred_launches = 0, feed_early_flushes = 0;
    uint64_t feed_blocks_launched = 0, feed_lanes_launched = 0;   // blocks inflated / lanes their launches occupied (whole rounds of the kernel's grid)
    uint64_t feed_idle_min_blocks = 8192;   // a super-batch of at least so many blocks is flushed early when the device has run dry ($PSSBAM_FEED_IDLE_BLOCKS, 0 = never)
    uint64_t feed_block_target = 0;   // blocks per super-batch: a whole number of rounds of the inflate kernel's lanes
    std::vector<std::pair<uint64_t, hipEvent_t>> feed_copies;             // (ticket, copy-complete event) of submits
    std::vector<hipEvent_t> feed_event_pool;
    uint32_t *d_feed_flags = nullptr;
    uint64_t *d_feed_tail = nullptr;   // bytes of the partial record the last flushed super-batch ended with
    bool feed_fresh = true;            // nothing of the current stream has been flushed yet
    uint32_t feed_skip = 0;            // inflated bytes in front of the stream's first record (the BAM header)
    double inflate_ms = 0.0;  // summed inflate + CRC + index kernel durations
    uint64_t inflated_bytes = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> inflate_events;
    // timing
    hipEvent_t t_begin = nullptr, t_end = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> launch_events;
    std::vector<hipEvent_t> event_pool;
    double kernel_ms = 0.0;
    uint64_t kernel_launches = 0;
    // tuning overrides (environment, for experiments)
    int env_tile_reads = 0, env_grid_mult = 0, env_simple_blocks = 0, env_grid_wgs = 0, env_pieces = 0;
    bool warned_ablate = false;
    bool compact_plan_once = false;   // tally_compact: header decode + filters once per read, plan through LDS (PSSBAM_COMPACT_PLAN_ONCE)
    uint32_t prep_lds[32] = {0};    // prep_kernel's memo, by kernel variant
    int prep_occ[32] = {0};
    bool use_compact = true;        // -r N <= 16: tally_compact (PSSBAM_COMPACT=0 keeps tally_tiled, for A/B runs)
    uint32_t *d_scratch = nullptr;  // per-workgroup partial tables of the tiled kernel
    size_t scratch_slots = 0;
    uint32_t dev_pieces = 0;       // prefix pieces sampled from a device-resident block
    uint64_t dev_pieces_avg = 0;   // ... and the mean record size it was sampled at
};

static void ctx_mask(const char *set, uint32_t (&m)[8]) {
    // strchr(set, c) != NULL: every byte of the string, plus the terminator itself --
    // expressed over STORED genome bytes (enc_byte permutation, record_decode.h)
    memset(m, 0, sizeof m);
    for (const unsigned char *p = (const unsigned char *)set;; p++) {
        const uint32_t st = enc_byte(*p);
        m[st >> 5] |= 1u << (st & 31);
        if (!*p) break;
    }
}

static int env_int(const char *name) {
    const char *v = getenv(name);
    return v ? atoi(v) : 0;
}

// the streams made by the helper thread (engine struct): taken over when they are there, or -- wait -- waited for
static void late_streams(pssbam_engine *e, bool wait) {
    std::lock_guard<std::mutex> lk(e->late_mu);
    if (e->late_adopted) return;
    if (!wait && !e->late_ready.load(std::memory_order_acquire)) return;
    if (e->late_thread.joinable()) e->late_thread.join();
    e->copy_stream = e->late_copy1;
    e->copy_stream2 = e->late_copy2;
    e->genome_stream = e->late_genome;
    e->late_adopted = true;
}

extern "C" int pssbam_engine_create(const pssbam_config *cfg, pssbam_engine **out) {
    if (!cfg || !out) return fail(PSSBAM_EINVAL, "null argument");
    *out = nullptr;
    if (cfg->abi_version != PSSBAM_ABI_VERSION)
        return fail(PSSBAM_EINVAL, "abi_version %u != %u", cfg->abi_version, PSSBAM_ABI_VERSION);
    if (!(cfg->tally_mask & (PSSBAM_TALLY_PSS | PSSBAM_TALLY_KMER)) ||
        (cfg->tally_mask & ~(PSSBAM_TALLY_PSS | PSSBAM_TALLY_KMER)))
        return fail(PSSBAM_EINVAL, "tally_mask must be a non-empty subset of PSS|KMER");
    if ((cfg->tally_mask & PSSBAM_TALLY_PSS)) {
        if (cfg->pss.region_len < 0 || cfg->pss.region_len > 1000000)
            return fail(PSSBAM_EINVAL, "region_len %d out of range", cfg->pss.region_len);
        if (!cfg->pss.up_ctx || !cfg->pss.down_ctx) return fail(PSSBAM_EINVAL, "up_ctx/down_ctx must be set");
    }
    // 4^k 64-bit bins per table in device memory: k = 15 is 2 x 8.6 GB of the 288 GB (the reference's
    // tree grows without bound, kmer.c:67-98; beyond 15 the bin index no longer fits the kernels' u32)
    if ((cfg->tally_mask & PSSBAM_TALLY_KMER) && (cfg->kmer.klen < 1 || cfg->kmer.klen > PSSBAM_MAX_KLEN))
        return fail(PSSBAM_EINVAL, "klen %d outside the device range 1..%d", cfg->kmer.klen, PSSBAM_MAX_KLEN);

    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
        return fail(PSSBAM_ENODEV, "no HIP device available (this engine has no CPU path)");
    int dev = cfg->device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    if (dev >= n_dev) return fail(PSSBAM_ENODEV, "device %d does not exist (%d present)", dev, n_dev);
    if (!device_is_gfx950(dev) && !getenv("PSSBAM_ALLOW_OTHER_ARCH"))
        return fail(PSSBAM_ENODEV, "device %d is not gfx950", dev);
    HIP_TRY(hipSetDevice(dev));

    pssbam_engine *e = new pssbam_engine();
    // a failure below releases what was set up so far (destroy tolerates a half-built engine)
    std::unique_ptr<pssbam_engine, void (*)(pssbam_engine *)> guard(e, pssbam_engine_destroy);
    e->cfg = *cfg;
    e->device = dev;
    if (cfg->tally_mask & PSSBAM_TALLY_PSS) {
        e->up_ctx = cfg->pss.up_ctx;
        e->down_ctx = cfg->pss.down_ctx;
    }
    if (cfg->read_group) {
        e->rg = cfg->read_group;
        e->has_rg = true;
    }
    e->cfg.pss.up_ctx = e->cfg.pss.down_ctx = e->cfg.read_group = nullptr;
    const bool tstat = getenv("PSSBAM_STATS") != nullptr;
    const double tc0 = tstat ? feed_now() : 0.0;
    hipDeviceProp_t pr;
    HIP_TRY(hipGetDeviceProperties(&pr, dev));
    e->n_cu = pr.multiProcessorCount;
    const double tc1 = tstat ? feed_now() : 0.0;
    HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    if (getenv("PSSBAM_STREAMS_UP_FRONT")) {   // (A/B: all four streams before create returns, as in round 2)
        HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream2, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&e->genome_stream, hipStreamNonBlocking));
        e->late_adopted = true;
    } else {
        e->late_thread = std::thread([e, dev]() {
            if (hipSetDevice(dev) == hipSuccess) {
                if (hipStreamCreateWithFlags(&e->late_copy1, hipStreamNonBlocking) != hipSuccess) e->late_copy1 = nullptr;
                if (hipStreamCreateWithFlags(&e->late_copy2, hipStreamNonBlocking) != hipSuccess) e->late_copy2 = nullptr;
                if (hipStreamCreateWithFlags(&e->late_genome, hipStreamNonBlocking) != hipSuccess) e->late_genome = nullptr;
            }
            (void)hipGetLastError();
            e->late_ready.store(1, std::memory_order_release);
        });
    }
    HIP_TRY(hipEventCreateWithFlags(&e->genome_ready, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&e->copied2, hipEventDisableTiming));
    e->own_stream = true;
    HIP_TRY(hipEventCreate(&e->t_begin));
    HIP_TRY(hipEventCreate(&e->t_end));
    const double tc2 = tstat ? feed_now() : 0.0;

    e->rows = (cfg->tally_mask & PSSBAM_TALLY_PSS) ? (uint32_t)cfg->pss.region_len + 2u : 0u;
    e->n_bins = (cfg->tally_mask & PSSBAM_TALLY_KMER) ? (1ull << (2 * cfg->kmer.klen)) : 0ull;
    e->off_rev = e->rows * 16u;
    e->off_k5 = 2u * e->rows * 16u;
    e->off_k3 = (uint32_t)(e->off_k5 + e->n_bins);
    e->off_stats = (uint32_t)(e->off_k3 + e->n_bins);
    e->n_counters = (size_t)e->off_stats + PSSBAM_ST_N;
    HIP_TRY(hipMalloc(&e->d_counters_own, e->n_counters * sizeof(unsigned long long)));
    e->d_counters = e->d_counters_own;
    HIP_TRY(hipMemsetAsync(e->d_counters, 0, e->n_counters * sizeof(unsigned long long), e->stream));
    if (e->has_rg) {
        HIP_TRY(hipMalloc(&e->d_rg, e->rg.size() + 16));
        HIP_TRY(hipMemcpy(e->d_rg, e->rg.data(), e->rg.size(), hipMemcpyHostToDevice));
    }
    for (Slot &s : e->slots) {
        HIP_TRY(hipEventCreate(&s.copy_begin));
        HIP_TRY(hipEventCreate(&s.copied));
        HIP_TRY(hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming));
    }
    // the tiled kernels' per-workgroup scratch, for the largest grid the launcher picks on its own (8 workgroups per CU): a
    // first launch that had to allocate it would first wait for everything queued on the stream -- with the compressed feed
    // running ahead of the genome that is tens of ms of inflate kernels (launch_tally)
    e->scratch_slots = (size_t)e->n_cu * 8;
    HIP_TRY(hipMalloc(&e->d_scratch, e->scratch_slots * SCRATCH_WORDS * sizeof(uint32_t)));
    e->env_tile_reads = env_int("PSSBAM_TILE_READS");
    e->env_grid_mult = env_int("PSSBAM_GRID_MULT");
    e->env_simple_blocks = env_int("PSSBAM_SIMPLE_BLOCKS");
    e->env_grid_wgs = env_int("PSSBAM_GRID_WGS");
    e->env_pieces = env_int("PSSBAM_PIECES");
    if (getenv("PSSBAM_COMPACT")) e->use_compact = env_int("PSSBAM_COMPACT") != 0;
    if (getenv("PSSBAM_COMPACT_PLAN_ONCE")) e->compact_plan_once = env_int("PSSBAM_COMPACT_PLAN_ONCE") != 0;
    if (tstat)
        fprintf(stderr, "[pssbam] engine on device %d: device properties %.3f, its stream + events %.3f, counters + scratch (first allocations, first "
                        "enqueue) %.3f s\n", dev, tc1 - tc0, tc2 - tc1, feed_now() - tc2);
    *out = guard.release();
    return PSSBAM_OK;
}

extern "C" void pssbam_engine_destroy(pssbam_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    late_streams(e, true);
    for (hipStream_t is : e->inflate_stream)
        if (is) (void)hipStreamSynchronize(is);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->copy_stream) (void)hipStreamSynchronize(e->copy_stream);
    if (e->copy_stream2) (void)hipStreamSynchronize(e->copy_stream2);
    for (Slot &s : e->slots) {
        if (s.d_recs) (void)hipFree(s.d_recs);
        if (s.d_offs) (void)hipFree(s.d_offs);
        if (s.copied) (void)hipEventDestroy(s.copied);
        if (s.copy_begin) (void)hipEventDestroy(s.copy_begin);
        if (s.consumed) (void)hipEventDestroy(s.consumed);
    }
    if (e->genome_stream) (void)hipStreamSynchronize(e->genome_stream);
    for (const void *p : e->genome_pins) pin_release(p);
    e->genome_pins.clear();
    for (FeedAcc *sp : e->feed) {
        FeedAcc &s = *sp;
        void *ptrs[] = {s.d_comp, s.d_blocks, s.d_a, s.d_e, s.d_last, s.d_nexta, s.d_n, s.d_counts, s.d_base, s.d_out, s.d_offs, s.d_nrecs, s.d_chain};
        for (void *q : ptrs)
            if (q) (void)hipFree(q);
        if (s.h_blocks) (void)hipHostFree(s.h_blocks);
        if (s.consumed) (void)hipEventDestroy(s.consumed);
        if (s.copies_done) (void)hipEventDestroy(s.copies_done);
        if (s.copies_done2) (void)hipEventDestroy(s.copies_done2);
        if (s.inflated) (void)hipEventDestroy(s.inflated);
        if (s.inflate_done) (void)hipEventDestroy(s.inflate_done);
        delete sp;
    }
    e->feed.clear();
    if (e->d_carry) (void)hipFree(e->d_carry);
    if (e->h_handoff) (void)hipHostFree(e->h_handoff);
    if (e->handoff_ev) (void)hipEventDestroy(e->handoff_ev);
    for (void *q : e->retired) (void)hipFree(q);
    if (e->d_feed_tail) (void)hipFree(e->d_feed_tail);
    for (auto &p : e->feed_copies) (void)hipEventDestroy(p.second);
    for (hipEvent_t ev : e->feed_event_pool) (void)hipEventDestroy(ev);
    if (e->d_feed_flags) (void)hipFree(e->d_feed_flags);
    for (auto &p : e->inflate_events) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto &p : e->launch_events) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (hipEvent_t ev : e->event_pool) (void)hipEventDestroy(ev);
    if (e->t_begin) (void)hipEventDestroy(e->t_begin);
    if (e->t_end) (void)hipEventDestroy(e->t_end);
    if (e->d_scratch) (void)hipFree(e->d_scratch);
    if (e->d_genome) (void)hipFree(e->d_genome);
    if (e->d_genome4) (void)hipFree(e->d_genome4);
    if (e->d_ref_info) (void)hipFree(e->d_ref_info);
    if (e->h_ref_info) (void)hipHostFree(e->h_ref_info);
    if (e->d_rg) (void)hipFree(e->d_rg);
    if (e->d_counters_own) (void)hipFree(e->d_counters_own);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    if (e->copy_stream2) (void)hipStreamDestroy(e->copy_stream2);
    for (hipStream_t is : e->inflate_stream)
        if (is) (void)hipStreamDestroy(is);
    if (e->feed_base_ev) (void)hipEventDestroy(e->feed_base_ev);
    if (e->genome_stream) (void)hipStreamDestroy(e->genome_stream);
    if (e->genome_ready) (void)hipEventDestroy(e->genome_ready);
    if (e->copied2) (void)hipEventDestroy(e->copied2);
    delete e;
}

extern "C" int pssbam_engine_set_stream(pssbam_engine *e, void *hip_stream) {
    if (!e) return fail(PSSBAM_EINVAL, "null engine");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->own_stream) HIP_TRY(hipStreamDestroy(e->stream));
    e->stream = (hipStream_t)hip_stream;
    e->own_stream = false;
    return PSSBAM_OK;
}

// --------------------------------------------------------------------------------------
// genome + references
// --------------------------------------------------------------------------------------
static constexpr uint64_t CONTIG_ALIGN = 256, CONTIG_PAD = 256;

static hipEvent_t take_event(pssbam_engine *e);
static int feed_resume(pssbam_engine *e);

// The upload still in flight (if any) has completed: its page locks go back.
static int genome_settle(pssbam_engine *e) {
    if (e->genome_pins.empty()) return PSSBAM_OK;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipEventSynchronize(e->genome_ready));
    for (const void *p : e->genome_pins) pin_release(p);
    e->genome_pins.clear();
    return PSSBAM_OK;
}

// Lays the contigs out, allocates, and ENQUEUES copies + encode + 4-bit pack on the genome stream; tally launches wait
// for genome_ready on the device.  Nothing here waits for the engine's stream (which may be busy inflating).
static int genome_upload(pssbam_engine *e, size_t n, const char *const *ids, const uint8_t *const *seqs, const uint64_t *lens,
                         int seqs_on_device) {
    if (!e || (n && (!ids || !seqs || !lens))) return fail(PSSBAM_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    int rc = genome_settle(e);
    if (rc) return rc;
    // sorted view, as init_genome leaves Genome.seqs (fasta-genome-io.c:236)
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return strcmp(ids[a], ids[b]) < 0; });
    std::vector<uint64_t> start(n + 1, 0);
    std::vector<uint32_t> len(n + 1, 0);
    uint64_t total = CONTIG_PAD;
    for (size_t k = 0; k < n; k++) {
        if (lens[order[k]] > 0xFFFFFF00ull)
            return fail(PSSBAM_EINVAL, "contig %s has %llu bases; the engine addresses contigs below 4 Gi bases", ids[order[k]],
                        (unsigned long long)lens[order[k]]);
        start[k] = total;
        len[k] = (uint32_t)lens[order[k]];
        total += ((uint64_t)len[k] + CONTIG_PAD + CONTIG_ALIGN - 1) / CONTIG_ALIGN * CONTIG_ALIGN;
    }
    if (e->d_genome) {   // a genome is being replaced: kernels queued on the engine's stream may still read the old one
        HIP_TRY(hipStreamSynchronize(e->stream));
        HIP_TRY(hipFree(e->d_genome));
        e->d_genome = nullptr;
        if (e->d_genome4) { HIP_TRY(hipFree(e->d_genome4)); e->d_genome4 = nullptr; }
    }
    HIP_TRY(hipMalloc(&e->d_genome, total));
    late_streams(e, true);
    if (!e->genome_stream) HIP_TRY(hipStreamCreateWithFlags(&e->genome_stream, hipStreamNonBlocking));
    hipStream_t gs = e->genome_stream;
    if (seqs_on_device) {   // the caller's device arrays were written on ITS stream (the engine's, after set_stream)
        hipEvent_t ev = take_event(e);
        if (!ev) return fail(PSSBAM_EHIP, "hipEventCreate failed");
        HIP_TRY(hipEventRecord(ev, e->stream));
        HIP_TRY(hipStreamWaitEvent(gs, ev, 0));
        e->event_pool.push_back(ev);
    }
    // padding = raw NUL, like the terminator the reference finds at index len (fragkon.c odd-k
    // windows); the encode pass below turns it into the stored form of NUL ("not a base")
    HIP_TRY(hipMemsetAsync(e->d_genome, 0, total, gs));
    // Host contigs: large ones are page-locked for the copy (cheap when the loader put them on
    // transparent huge pages: 2 MiB per pin instead of 4 KiB), which turns a staged pageable copy into
    // one DMA at link speed; if the lock is refused the plain copy below does the job.  The locks are
    // shared by the engines of the process and go back when the last upload has completed.
    const bool try_lock = !seqs_on_device && !getenv("PSSBAM_NO_PIN");
    for (size_t k = 0; k < n; k++) {
        if (!len[k]) continue;
        const void *src = seqs[order[k]];
        if (try_lock && len[k] >= (32u << 20) && pin_acquire(src, len[k])) e->genome_pins.push_back(src);
        HIP_TRY(hipMemcpyAsync(e->d_genome + start[k], src, len[k],
                               seqs_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, gs));
    }
    // raw bytes (and NUL padding) are in place: one pass folds case and applies enc_byte to all
    hipLaunchKernelGGL(encode_genome_kernel, dim3(4096), dim3(256), 0, gs, e->d_genome, total / 16);
    HIP_TRY(hipGetLastError());
    {
        // 4 bits per base for the tiled kernel's windows: A C G T, or "other" with its -U / -D membership
        CtxSets sets;
        ctx_mask(e->up_ctx.c_str(), sets.up);
        ctx_mask(e->down_ctx.c_str(), sets.down);
        e->acgt_ctx = 0;
        for (uint32_t v = 0; v < 4; v++)
            e->acgt_ctx |= (((sets.up[0] >> v) & 1u) << (2 * v)) | (((sets.down[0] >> v) & 1u) << (2 * v + 1));
        const uint64_t n_out = total / 8, slack = 16;
        HIP_TRY(hipMalloc(&e->d_genome4, (n_out + slack) * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(e->d_genome4 + n_out, 0x44, slack * sizeof(uint32_t), gs));
        hipLaunchKernelGGL(pack_genome4_kernel, dim3(4096), dim3(256), 0, gs, e->d_genome, e->d_genome4, n_out, sets);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(e->genome_ready, gs));
    e->genome_wait_pending = true;
    e->contig_start.assign(start.begin(), start.begin() + n);
    e->contig_len.assign(len.begin(), len.begin() + n);
    e->genome_bytes = total;
    e->contig_ids.clear();
    for (size_t k = 0; k < n; k++) e->contig_ids.emplace_back(ids[order[k]]);
    e->star_contig = -1;
    if (e->have_refs) e->have_refs = false;   // (written only when it changes: another thread may be feeding, see pssbam_hip.h)
    {   // RNAME "*" (refID -1) goes through find_seq like any other name
        auto it = std::lower_bound(e->contig_ids.begin(), e->contig_ids.end(), std::string("*"),
                                   [](const std::string &a, const std::string &b) { return strcmp(a.c_str(), b.c_str()) < 0; });
        if (it != e->contig_ids.end() && *it == "*") e->star_contig = (int32_t)(it - e->contig_ids.begin());
    }
    return PSSBAM_OK;
}

extern "C" int pssbam_engine_set_genome_arrays(pssbam_engine *e, size_t n, const char *const *ids,
                                               const uint8_t *const *seqs, const uint64_t *lens,
                                               int seqs_on_device) {
    int rc = genome_upload(e, n, ids, seqs, lens, seqs_on_device);
    if (rc) return rc;
    if (e->genome_stream) HIP_TRY(hipStreamSynchronize(e->genome_stream));   // the caller may release its arrays when this returns
    return genome_settle(e);
}

static int genome_from_struct(pssbam_engine *e, const struct genome *g, bool wait) {
    if (!e || !g) return fail(PSSBAM_EINVAL, "null argument");
    std::vector<const char *> ids(g->n_seqs);
    std::vector<const uint8_t *> seqs(g->n_seqs);
    std::vector<uint64_t> lens(g->n_seqs);
    for (size_t i = 0; i < g->n_seqs; i++) {
        ids[i] = g->seqs[i]->id;
        seqs[i] = (const uint8_t *)g->seqs[i]->seq;
        lens[i] = g->seqs[i]->len;
    }
    return wait ? pssbam_engine_set_genome_arrays(e, g->n_seqs, ids.data(), seqs.data(), lens.data(), 0)
                : genome_upload(e, g->n_seqs, ids.data(), seqs.data(), lens.data(), 0);
}

extern "C" int pssbam_engine_set_genome(pssbam_engine *e, const struct genome *g) { return genome_from_struct(e, g, true); }

// Same, but returns as soon as the upload is enqueued: the Genome must stay untouched until pssbam_engine_genome_wait,
// _sync or _finish has returned.  Lets one host thread start the uploads of several GPUs at once.
extern "C" int pssbam_engine_set_genome_async(pssbam_engine *e, const struct genome *g) { return genome_from_struct(e, g, false); }

extern "C" int pssbam_engine_genome_wait(pssbam_engine *e) {
    if (!e) return fail(PSSBAM_EINVAL, "null engine");
    HIP_TRY(hipSetDevice(e->device));
    if (e->genome_stream) HIP_TRY(hipStreamSynchronize(e->genome_stream));
    return genome_settle(e);
}

extern "C" int pssbam_engine_set_references(pssbam_engine *e, int32_t n_ref, const char *const *names) {
    if (!e || n_ref < 0 || (n_ref && !names)) return fail(PSSBAM_EINVAL, "bad argument");
    if (!e->d_genome) return fail(PSSBAM_ESTATE, "set_genome must precede set_references");
    HIP_TRY(hipSetDevice(e->device));
    // ref_info[i] = {gbase lo, gbase hi, length, found}; entry n_ref answers RNAME "*" (refID -1)
    std::vector<uint4> info((size_t)n_ref + 1, make_uint4(0, 0, 0, 0));
    auto fill = [&](size_t slot, int32_t contig) {
        if (contig < 0) return;
        const uint64_t gb = e->contig_start[(size_t)contig];
        info[slot] = make_uint4((uint32_t)gb, (uint32_t)(gb >> 32), e->contig_len[(size_t)contig], 1u);
    };
    for (int32_t i = 0; i < n_ref; i++) {
        // bsearch with strcmp over the sorted ids == find_seq (fasta-genome-io.c:202-213)
        size_t lo = 0, hi = e->contig_ids.size();
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            const int c = strcmp(names[i], e->contig_ids[mid].c_str());
            if (c == 0) { fill((size_t)i, (int32_t)mid); break; }
            if (c < 0) hi = mid; else lo = mid + 1;
        }
    }
    fill((size_t)n_ref, e->star_contig);
    if (e->feed_opened && !e->deferred.empty() && n_ref != e->feed_n_ref)
        return fail(PSSBAM_ESTATE, "pssbam_engine_feed_open announced %d references, set_references brings %d", e->feed_n_ref, n_ref);
    // a table that is being REPLACED (SAM text: the list grows as new RNAMEs show up) may still be read by queued
    // kernels; the first one cannot be, so nothing waits for the engine's stream then
    if (e->d_ref_info) {
        if (e->have_refs) HIP_TRY(hipStreamSynchronize(e->stream));
        HIP_TRY(hipFree(e->d_ref_info));
        e->d_ref_info = nullptr;
    }
    HIP_TRY(hipMalloc(&e->d_ref_info, ((size_t)n_ref + 1) * sizeof(uint4)));
    // The table is a few hundred bytes, but a hipMemcpy of it queues behind whatever the copy engines are moving -- the 3 GB
    // genome, typically: 50 ms during which the calling thread feeds nothing.  It goes through page-locked host memory
    // instead and a one-workgroup kernel on the engine's stream reads it from there (ordered in front of every tally).
    if (e->h_ref_cap < (size_t)n_ref + 1) {
        if (e->h_ref_info) (void)hipHostFree(e->h_ref_info);
        e->h_ref_info = nullptr;
        e->h_ref_cap = (size_t)n_ref + 1 + 64;
        HIP_TRY(hipHostMalloc((void **)&e->h_ref_info, e->h_ref_cap * sizeof(uint4), hipHostMallocDefault));
    } else if (e->have_refs) HIP_TRY(hipStreamSynchronize(e->stream));   // (a kernel may still be reading the previous contents)
    memcpy(e->h_ref_info, info.data(), ((size_t)n_ref + 1) * sizeof(uint4));
    hipLaunchKernelGGL(copy_table_kernel, dim3(1), dim3(256), 0, e->stream, e->d_ref_info, (const uint4 *)e->h_ref_info, (uint32_t)n_ref + 1u);
    HIP_TRY(hipGetLastError());
    e->n_ref = n_ref;
    e->have_refs = true;
    return feed_resume(e);   // super-batches inflated ahead of the genome are tallied now
}

// --------------------------------------------------------------------------------------
// launch
// --------------------------------------------------------------------------------------
static hipEvent_t take_event(pssbam_engine *e) {
    if (!e->event_pool.empty()) {
        hipEvent_t ev = e->event_pool.back();
        e->event_pool.pop_back();
        return ev;
    }
    hipEvent_t ev = nullptr;
    (void)hipEventCreate(&ev);
    return ev;
}

static int resolve_launch_events(pssbam_engine *e) {
    for (auto &p : e->launch_events) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(p.second));
        HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
        e->kernel_ms += ms;
        e->kernel_launches++;
        e->event_pool.push_back(p.first);
        e->event_pool.push_back(p.second);
    }
    e->launch_events.clear();
    return PSSBAM_OK;
}

// dynamic-LDS limit + occupancy of one kernel variant, remembered per (variant, LDS size) so
// the steady state makes no runtime API calls per launch beyond the launches themselves
template <class K>
static int prep_kernel(pssbam_engine *e, int variant, K kernel, uint32_t lds_bytes, int *occ) {
    if (e->prep_lds[variant] == lds_bytes && e->prep_occ[variant] > 0) {
        *occ = e->prep_occ[variant];
        return PSSBAM_OK;
    }
    HIP_TRY(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, kernel, TILED_THREADS, lds_bytes));
    if (*occ < 1) return fail(PSSBAM_EHIP, "kernel does not fit a CU with %u bytes of LDS", lds_bytes);
    e->prep_lds[variant] = lds_bytes;
    e->prep_occ[variant] = *occ;
    return PSSBAM_OK;
}

// How many 16-byte pieces of a record the tiled kernel must stage so that everything the path
// reads (through QUAL[0]; the whole record when the -R filter walks the aux fields) is in LDS
// for typical records: sampled from the first records of a block.  Records that need more take
// the kernel's out-of-line global-memory path, so this is a performance choice only.
// (returns the largest prefix, in bytes, any sampled record needs)
static uint64_t sample_prefix_need(const uint8_t *bytes, uint64_t nbytes, bool whole_record, int max_records = 4096) {
    uint64_t o = 0, need_max = 64;
    for (int n = 0; n < max_records && o + 36 <= nbytes; n++) {
        uint32_t bs, l_seq;
        memcpy(&bs, bytes + o, 4);
        if (bs < 32 || o + 4 + (uint64_t)bs > nbytes) break;
        const uint32_t l_name = bytes[o + 12];
        uint16_t n_cig;
        memcpy(&n_cig, bytes + o + 16, 2);
        memcpy(&l_seq, bytes + o + 20, 4);
        const uint64_t qual_off = 36ull + l_name + 4ull * n_cig + ((uint64_t)l_seq + 1) / 2;
        const uint64_t need = whole_record ? 4ull + bs : std::min<uint64_t>(qual_off + 1, 4ull + bs);
        need_max = std::max(need_max, need);
        o += 4 + (uint64_t)bs;
    }
    return need_max;
}

static uint32_t pieces_for(uint64_t need_max) {
    uint64_t pieces = (need_max + 15 + 15) / 16;  // + worst-case misalignment of the record start
    // records sit pieces*16 bytes apart in LDS: an even piece count puts every record start of a
    // wave on few banks (pieces = 8 -> all on one); an odd count spreads them over 8
    pieces |= 1;
    return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(pieces, 5), 41);
}

// d_n_recs != NULL: the record count lives in device memory (blocks indexed on the device); n_records
// is then only an upper bound for the launch geometry, and the prefix sample is taken sample_off
// bytes into the block.
static int launch_tally(pssbam_engine *e, const uint8_t *d_recs, uint64_t nbytes, const uint32_t *d_offs,
                        uint32_t n_records, const uint8_t *host_sample, uint64_t host_sample_bytes,
                        const uint32_t *d_n_recs = nullptr, uint64_t sample_off = 0, const uint32_t *host_offsets = nullptr) {
    if (!n_records) return PSSBAM_OK;
    const pssbam_config &c = e->cfg;
    TallyParams P{};
    P.recs = d_recs;
    P.offs = d_offs;
    P.n_recs = n_records;
    P.n_recs_dev = d_n_recs;
    P.recs_bytes = nbytes;
    P.tally_mask = c.tally_mask;
    P.genome = e->d_genome;
    P.genome4 = e->d_genome4;
    P.acgt_ctx = e->acgt_ctx;
    P.ref_info = e->d_ref_info;
    P.n_ref = e->n_ref;
    const bool do_pss = (c.tally_mask & PSSBAM_TALLY_PSS) != 0, do_kmer = (c.tally_mask & PSSBAM_TALLY_KMER) != 0;
    if (do_pss) {
        P.N = c.pss.region_len;
        P.pss_min_mq = (uint32_t)c.pss.min_mq;
        P.pss_len_never = c.pss.min_read_len > 0xFFFFFFFFull ? 1u : 0u;
        P.pss_min_len = (uint32_t)std::min<uint64_t>(c.pss.min_read_len, 0xFFFFFFFFull);
        P.pss_max_len = (uint32_t)std::min<uint64_t>(c.pss.max_read_len, 0xFFFFFFFFull);
        P.pss_merged_only = c.pss.merged_only ? 1u : 0u;
        ctx_mask(e->up_ctx.c_str(), P.up_mask);
        ctx_mask(e->down_ctx.c_str(), P.down_mask);
    }
    if (do_kmer) {
        P.K = c.kmer.klen;
        P.fk_min_mq = (uint32_t)c.kmer.min_mq;
        P.fk_len_never = c.kmer.min_read_len > 0xFFFFFFFFull ? 1u : 0u;
        P.fk_min_len = (uint32_t)std::min<uint64_t>(c.kmer.min_read_len, 0xFFFFFFFFull);
        P.fk_max_len = (uint32_t)std::min<uint64_t>(c.kmer.max_read_len, 0xFFFFFFFFull);
        P.fk_merged_only = c.kmer.merged_only ? 1u : 0u;
    }
    P.rg = e->has_rg ? e->d_rg : nullptr;
    P.rg_len = (uint32_t)e->rg.size();
    P.counters = e->d_counters;
    P.off_rev = e->off_rev;
    P.off_k5 = e->off_k5;
    P.off_k3 = e->off_k3;
    P.off_stats = e->off_stats;

    int kernel = c.kernel;
    if (d_n_recs && kernel == PSSBAM_KERNEL_SIMPLE) return fail(PSSBAM_EINVAL, "device-indexed blocks need the tiled kernels");
    // the tiled kernel covers 32 table rows per pass over the block (measured on C3: N=30 18 G
    // reads/s, N=62 9.3 G, N=100 4.8 G; the generic kernel: 0.77 / 0.37 / 0.23 G and falling with
    // N), so it is the automatic choice for every N; the generic kernel is the cross-check
    const uint32_t n_passes = do_pss ? (e->rows + TILED_ROWS - 1) / TILED_ROWS : 1u;
    if (kernel == PSSBAM_KERNEL_AUTO) kernel = PSSBAM_KERNEL_TILED;

    if (e->genome_wait_pending) {   // the upload runs on its own stream
        HIP_TRY(hipStreamWaitEvent(e->stream, e->genome_ready, 0));
        e->genome_wait_pending = false;
    }
    hipEvent_t ev0 = take_event(e), ev1 = take_event(e);
    if (!ev0 || !ev1) return fail(PSSBAM_EHIP, "hipEventCreate failed");
    HIP_TRY(hipEventRecord(ev0, e->stream));

    if (kernel == PSSBAM_KERNEL_SIMPLE) {
        const uint32_t tab_bytes = do_pss ? 2u * e->rows * 16u * 4u : 0u;
        const bool lds_tab = do_pss && tab_bytes <= 60u * 1024u;
        uint32_t blocks = (uint32_t)std::min<uint64_t>(((uint64_t)n_records + 255) / 256, (uint64_t)e->n_cu * 8);
        if (e->env_simple_blocks > 0) blocks = (uint32_t)e->env_simple_blocks;
        if (lds_tab) hipLaunchKernelGGL(tally_simple<true>, dim3(blocks), dim3(256), tab_bytes, e->stream, P);
        else hipLaunchKernelGGL(tally_simple<false>, dim3(blocks), dim3(256), 0, e->stream, P);
    } else {
        // how much of each record goes through LDS: sampled from the block itself (host copy at
        // hand for submit(); for device-resident blocks a one-off 64 KiB read-back, remembered
        // while the mean record size stays put)
        const uint64_t avg = std::max<uint64_t>(40, nbytes / n_records);
        uint32_t pieces;
        if (host_sample) {
            // three regions of the block (start, middle, end: 1400 records each), found through the
            // caller's offset index -- a block whose later records are longer than its first ones must
            // not silently fall onto the one-lane path (stats.slow_path)
            uint64_t need = sample_prefix_need(host_sample, host_sample_bytes, e->has_rg, 1400);
            if (host_offsets && n_records > 4200u) {
                const uint32_t mid = host_offsets[n_records / 2], late = host_offsets[n_records - 1400u];
                need = std::max(need, sample_prefix_need(host_sample + mid, host_sample_bytes - mid, e->has_rg, 1400));
                need = std::max(need, sample_prefix_need(host_sample + late, host_sample_bytes - late, e->has_rg, 1400));
            }
            pieces = pieces_for(need);
        } else {
            if (!e->dev_pieces || (!d_n_recs && (avg * 8 < e->dev_pieces_avg * 7 || avg * 7 > e->dev_pieces_avg * 8))) {
                sample_off = std::min<uint64_t>(sample_off, nbytes);
                std::vector<uint8_t> head((size_t)std::min<uint64_t>(nbytes - sample_off, 1024 * 1024));
                HIP_TRY(hipMemcpyAsync(head.data(), d_recs + sample_off, head.size(), hipMemcpyDeviceToHost, e->stream));
                HIP_TRY(hipStreamSynchronize(e->stream));
                uint64_t need = sample_prefix_need(head.data(), head.size(), e->has_rg);
                if (!d_n_recs && n_records > 8192u) {   // device-resident block with a known count: its middle and end too
                    uint32_t at[2] = {0, 0};
                    HIP_TRY(hipMemcpyAsync(&at[0], d_offs + n_records / 2, 4, hipMemcpyDeviceToHost, e->stream));
                    HIP_TRY(hipMemcpyAsync(&at[1], d_offs + (n_records - 2048u), 4, hipMemcpyDeviceToHost, e->stream));
                    HIP_TRY(hipStreamSynchronize(e->stream));
                    for (int k = 0; k < 2; k++) {
                        if ((uint64_t)at[k] >= nbytes) continue;
                        head.resize((size_t)std::min<uint64_t>(nbytes - at[k], 512 * 1024));
                        HIP_TRY(hipMemcpyAsync(head.data(), d_recs + at[k], head.size(), hipMemcpyDeviceToHost, e->stream));
                        HIP_TRY(hipStreamSynchronize(e->stream));
                        need = std::max(need, sample_prefix_need(head.data(), head.size(), e->has_rg, 2048));
                    }
                }
                e->dev_pieces = pieces_for(need);
                e->dev_pieces_avg = avg;
            }
            pieces = e->dev_pieces;
        }
        if (e->env_pieces > 0) pieces = (uint32_t)std::min(std::max(e->env_pieces, 4), 64);
        uint32_t T = TILED_MAX_T;
        if (e->env_tile_reads > 0) T = std::min<uint32_t>(TILED_MAX_T, (uint32_t)(e->env_tile_reads + 15) / 16 * 16);
        P.reads_per_tile = T;
        P.prefix_pieces = pieces;
        P.xcd_map = (uint32_t)env_int("PSSBAM_XCD_MAP");
        P.ablate = (uint32_t)env_int("PSSBAM_ABLATE");  // profiling aid (tools/ablate.sh): switches kernel phases off
        if (P.ablate && !e->warned_ablate) {
            fprintf(stderr, "[pssbam] PSSBAM_ABLATE=%u: kernel phases are switched off, the tables are WRONG (profiling only)\n", P.ablate);
            e->warned_ablate = true;
        }
        const bool kmer_lds = do_kmer && c.kmer.klen <= KMER_LDS_MAX_K;
        const uint32_t n_tiles = (n_records + T - 1) / T;
        const uint32_t lds = tiled_lds_bytes(T, pieces);
        int occ = 0, rc = PSSBAM_OK;
        const int mult = e->env_grid_mult > 0 ? e->env_grid_mult : 1;
#define LAUNCH_TILED(PSS, KM, LK, LATER)                                                                  \
    do {                                                                                           \
        rc = prep_kernel(e, (LATER ? 8 : 0) | (PSS ? 4 : 0) | (KM ? 2 : 0) | (LK ? 1 : 0), tally_tiled<PSS, KM, LK, LATER>, lds, &occ); \
        if (rc == PSSBAM_OK) {                                                                     \
            uint32_t grid = (uint32_t)std::min<uint64_t>(n_tiles, (uint64_t)e->n_cu * occ * mult); \
            if (e->env_grid_wgs > 0) grid = (uint32_t)std::min<uint64_t>(n_tiles, (uint64_t)e->env_grid_wgs); \
            if (e->scratch_slots < grid) {                                                         \
                HIP_TRY(hipStreamSynchronize(e->stream));                                          \
                if (e->d_scratch) HIP_TRY(hipFree(e->d_scratch));                                  \
                e->d_scratch = nullptr;                                                            \
                e->scratch_slots = std::max<size_t>(grid, (size_t)e->n_cu * 8);                    \
                HIP_TRY(hipMalloc(&e->d_scratch, e->scratch_slots * SCRATCH_WORDS * sizeof(uint32_t))); \
            }                                                                                      \
            P.scratch = e->d_scratch;                                                              \
            hipLaunchKernelGGL((tally_tiled<PSS, KM, LK, LATER>), dim3(grid), dim3(TILED_THREADS), lds, e->stream, P); \
            hipLaunchKernelGGL(reduce_partials, dim3((SCRATCH_WORDS * REDUCE_GROUPS + 255) / 256), dim3(256), 0, e->stream, P, grid, \
                               (uint32_t)(LK ? 1 : 0));                                            \
        }                                                                                          \
    } while (0)
        P.row_base = 0;
        // -r N <= 16 (2 context rows + 16 positions): the short-window variant, one pass
#define LAUNCH_COMPACT(KM, LK) do { if (e->compact_plan_once) LAUNCH_COMPACT_(KM, LK, true, 24); else LAUNCH_COMPACT_(KM, LK, false, 16); } while (0)
#define LAUNCH_COMPACT_(KM, LK, ONCE, VAR)                                                           \
    do {                                                                                           \
        rc = prep_kernel(e, VAR | (KM ? 2 : 0) | (LK ? 1 : 0), tally_compact<KM, LK, ONCE>, lds, &occ); \
        if (rc == PSSBAM_OK) {                                                                     \
            uint32_t grid = (uint32_t)std::min<uint64_t>(n_tiles, (uint64_t)e->n_cu * occ * mult); \
            if (e->env_grid_wgs > 0) grid = (uint32_t)std::min<uint64_t>(n_tiles, (uint64_t)e->env_grid_wgs); \
            if (e->scratch_slots < grid) {                                                         \
                HIP_TRY(hipStreamSynchronize(e->stream));                                          \
                if (e->d_scratch) HIP_TRY(hipFree(e->d_scratch));                                  \
                e->d_scratch = nullptr;                                                            \
                e->scratch_slots = std::max<size_t>(grid, (size_t)e->n_cu * 8);                    \
                HIP_TRY(hipMalloc(&e->d_scratch, e->scratch_slots * SCRATCH_WORDS * sizeof(uint32_t))); \
            }                                                                                      \
            P.scratch = e->d_scratch;                                                              \
            hipLaunchKernelGGL((tally_compact<KM, LK, ONCE>), dim3(grid), dim3(TILED_THREADS), lds, e->stream, P); \
            hipLaunchKernelGGL(reduce_partials, dim3((SCRATCH_WORDS * REDUCE_GROUPS + 255) / 256), dim3(256), 0, e->stream, P, grid, \
                               (uint32_t)(LK ? 1 : 0));                                            \
        }                                                                                          \
    } while (0)
        if (do_pss && e->rows <= COMPACT_MAX_ROWS && e->use_compact && !e->has_rg) {
            if (!do_kmer && getenv("PSSBAM_COMPACT_DECODE_TWICE")) {   // diagnostics: what the shared header decode costs
                rc = prep_kernel(e, 20, tally_compact_decode_twice, lds, &occ);
                if (rc == PSSBAM_OK) {
                    const uint32_t grid = (uint32_t)std::min<uint64_t>(n_tiles, (uint64_t)e->n_cu * occ * mult);
                    if (e->scratch_slots >= grid) {
                        P.scratch = e->d_scratch;
                        hipLaunchKernelGGL(tally_compact_decode_twice, dim3(grid), dim3(TILED_THREADS), lds, e->stream, P);
                        hipLaunchKernelGGL(reduce_partials, dim3((SCRATCH_WORDS * REDUCE_GROUPS + 255) / 256), dim3(256), 0, e->stream, P, grid, 0u);
                    } else rc = fail(PSSBAM_ESTATE, "scratch too small for the diagnostic kernel");
                }
            } else if (!do_kmer) LAUNCH_COMPACT(false, false);
            else if (kmer_lds) LAUNCH_COMPACT(true, true);
            else LAUNCH_COMPACT(true, false);
        } else
        if (do_pss && do_kmer) { if (kmer_lds) LAUNCH_TILED(true, true, true, false); else LAUNCH_TILED(true, true, false, false); }
        else if (do_pss) LAUNCH_TILED(true, false, false, false);
        else { if (kmer_lds) LAUNCH_TILED(false, true, true, false); else LAUNCH_TILED(false, true, false, false); }
        // rows 32.. of a large -r: further passes over the same block, substitution rows only
        // (the status counters and the k-mer tally belong to pass 0)
        for (uint32_t pass = 1; pass < n_passes && rc == PSSBAM_OK; pass++) {
            P.row_base = pass * TILED_ROWS;
            LAUNCH_TILED(true, false, false, true);
        }
#undef LAUNCH_TILED
#undef LAUNCH_COMPACT
        if (rc != PSSBAM_OK) return rc;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev1, e->stream));
    e->launch_events.emplace_back(ev0, ev1);
    if (e->launch_events.size() > 4096) return resolve_launch_events(e);
    return PSSBAM_OK;
}

// A caller that will only hand over device-resident or compressed blocks can show the engine a few
// host-side records first: the tiled kernels' staged prefix is then sized from them instead of from a
// read-back of the first block (which has to wait for that block to be inflated).
extern "C" int pssbam_engine_hint_records(pssbam_engine *e, const void *records, uint64_t nbytes) {
    if (!e || (!records && nbytes)) return fail(PSSBAM_EINVAL, "null argument");
    if (nbytes < 36) return PSSBAM_OK;
    e->dev_pieces = pieces_for(sample_prefix_need((const uint8_t *)records, nbytes, e->has_rg, 1 << 16));
    e->dev_pieces_avg = 40;
    return PSSBAM_OK;
}

static int check_ready(pssbam_engine *e) {
    if (!e) return fail(PSSBAM_EINVAL, "null engine");
    if (!e->d_genome) return fail(PSSBAM_ESTATE, "set_genome has not been called");
    if (!e->have_refs) return fail(PSSBAM_ESTATE, "set_references has not been called");
    return PSSBAM_OK;
}

extern "C" int pssbam_engine_submit_device(pssbam_engine *e, const void *d_records, uint64_t nbytes,
                                           const uint32_t *d_offsets, uint32_t n_records) {
    int rc = check_ready(e);
    if (rc) return rc;
    if (n_records && (!d_records || !d_offsets)) return fail(PSSBAM_EINVAL, "null buffer");
    if (nbytes >= (1ull << 32)) return fail(PSSBAM_EINVAL, "record block must be < 4 GiB (got %llu)", (unsigned long long)nbytes);
    if (((uintptr_t)d_records & 15u) || ((uintptr_t)d_offsets & 3u))
        return fail(PSSBAM_EINVAL, "d_records must be 16-byte aligned, d_offsets 4-byte aligned");
    HIP_TRY(hipSetDevice(e->device));
    return launch_tally(e, (const uint8_t *)d_records, nbytes, d_offsets, n_records, nullptr, 0);
}

// books the H2D duration of a slot whose copy is known to be complete
static void book_copy_time(pssbam_engine *e, Slot &s) {
    if (!s.timed) return;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s.copy_begin, s.copied) == hipSuccess) e->h2d_ms += ms;
    s.timed = false;
}

extern "C" int pssbam_engine_submit_async(pssbam_engine *e, const void *records, uint64_t nbytes, const uint32_t *offsets,
                                          uint32_t n_records, uint64_t *ticket) {
    int rc = check_ready(e);
    if (rc) return rc;
    if (ticket) *ticket = 0;
    if (!n_records) return PSSBAM_OK;
    if (!records || !offsets) return fail(PSSBAM_EINVAL, "null buffer");
    if (nbytes >= (1ull << 32)) return fail(PSSBAM_EINVAL, "record block must be < 4 GiB (got %llu)", (unsigned long long)nbytes);
    if (offsets[n_records] != nbytes) return fail(PSSBAM_EFORMAT, "offsets[n_records] must equal nbytes");
    HIP_TRY(hipSetDevice(e->device));
    Slot &s = e->slots[e->next_slot];
    e->next_slot ^= 1;
    if (s.busy) HIP_TRY(hipEventSynchronize(s.consumed));  // the kernel that read this slot is done (so is its copy)
    s.busy = false;
    book_copy_time(e, s);
    if (s.recs_cap < nbytes + 64) {
        if (s.d_recs) HIP_TRY(hipFree(s.d_recs));
        s.d_recs = nullptr;
        s.recs_cap = (size_t)(nbytes + nbytes / 4 + 4096);
        HIP_TRY(hipMalloc(&s.d_recs, s.recs_cap));
    }
    if (s.offs_cap < (size_t)n_records + 1) {
        if (s.d_offs) HIP_TRY(hipFree(s.d_offs));
        s.d_offs = nullptr;
        s.offs_cap = (size_t)n_records + n_records / 4 + 1024;
        HIP_TRY(hipMalloc(&s.d_offs, s.offs_cap * sizeof(uint32_t)));
    }
    // H2D on the copy stream so it overlaps the previous block's kernel
    // large blocks go over two copy streams (two DMA engines): one stream alone does not fill
    // the PCIe link
    late_streams(e, true);   // (this path copies on a stream of its own from the first block on)
    if (!e->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    const uint64_t half = (nbytes >= (64ull << 20) && e->copy_stream2 && !getenv("PSSBAM_ONE_COPY_STREAM")) ? (nbytes / 2) & ~4095ull : 0;
    HIP_TRY(hipEventRecord(s.copy_begin, e->copy_stream));
    if (half) {
        HIP_TRY(hipStreamWaitEvent(e->copy_stream2, s.copy_begin, 0));
        HIP_TRY(hipMemcpyAsync(s.d_recs + half, (const uint8_t *)records + half, nbytes - half, hipMemcpyHostToDevice, e->copy_stream2));
        HIP_TRY(hipEventRecord(e->copied2, e->copy_stream2));
    }
    HIP_TRY(hipMemcpyAsync(s.d_recs, records, half ? half : nbytes, hipMemcpyHostToDevice, e->copy_stream));
    HIP_TRY(hipMemcpyAsync(s.d_offs, offsets, ((size_t)n_records + 1) * sizeof(uint32_t), hipMemcpyHostToDevice,
                           e->copy_stream));
    if (half) HIP_TRY(hipStreamWaitEvent(e->copy_stream, e->copied2, 0));  // `copied` then covers both halves
    HIP_TRY(hipEventRecord(s.copied, e->copy_stream));
    s.timed = true;
    e->h2d_bytes += nbytes + ((uint64_t)n_records + 1) * sizeof(uint32_t);
    HIP_TRY(hipStreamWaitEvent(e->stream, s.copied, 0));
    rc = launch_tally(e, s.d_recs, nbytes, s.d_offs, n_records, (const uint8_t *)records, nbytes, nullptr, 0, offsets);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(s.consumed, e->stream));
    s.busy = true;
    s.ticket = ++e->ticket_seq;
    if (ticket) *ticket = s.ticket;
    return PSSBAM_OK;
}

// 1 = the copy of that submit has completed (its host buffers are free), 0 = still in flight
extern "C" int pssbam_engine_copy_done(pssbam_engine *e, uint64_t ticket) {
    if (!e) return fail(PSSBAM_EINVAL, "null engine");
    if (!ticket) return 1;
    for (Slot &s : e->slots)
        if (s.ticket == ticket) {
            HIP_TRY(hipSetDevice(e->device));
            const hipError_t q = hipEventQuery(s.copied);
            if (q == hipSuccess) return 1;
            if (q == hipErrorNotReady) return 0;
            return fail(PSSBAM_EHIP, "hipEventQuery failed: %s", hipGetErrorString(q));
        }
    return 1;  // the slot has been reused since: that submit waited for the kernel behind this copy
}

extern "C" int pssbam_engine_wait_copied(pssbam_engine *e, uint64_t ticket) {
    if (!e) return fail(PSSBAM_EINVAL, "null engine");
    if (!ticket) return PSSBAM_OK;
    for (Slot &s : e->slots)
        if (s.ticket == ticket) {
            HIP_TRY(hipSetDevice(e->device));
            HIP_TRY(hipEventSynchronize(s.copied));
            return PSSBAM_OK;
        }
    return PSSBAM_OK;
}

extern "C" int pssbam_engine_submit(pssbam_engine *e, const void *records, uint64_t nbytes, const uint32_t *offsets,
                                    uint32_t n_records) {
    uint64_t ticket = 0;
    int rc = pssbam_engine_submit_async(e, records, nbytes, offsets, n_records, &ticket);
    if (rc) return rc;
    // contract: the caller's buffers are free for reuse when we return
    return pssbam_engine_wait_copied(e, ticket);
}

extern "C" int pssbam_engine_phase_times(pssbam_engine *e, double *h2d_ms, uint64_t *h2d_bytes, double *kernel_ms,
                                         uint64_t *n_launches) {
    if (!e) return fail(PSSBAM_EINVAL, "null engine");
    int rc = pssbam_engine_sync(e);
    if (rc) return rc;
    for (Slot &s : e->slots) book_copy_time(e, s);
    rc = resolve_launch_events(e);
    if (rc) return rc;
    if (h2d_ms) *h2d_ms = e->h2d_ms;
    if (h2d_bytes) *h2d_bytes = e->h2d_bytes;
    if (kernel_ms) *kernel_ms = e->kernel_ms;
    if (n_launches) *n_launches = e->kernel_launches;
    return PSSBAM_OK;
}

static int feed_flush(pssbam_engine *e);

extern "C" int pssbam_engine_sync(pssbam_engine *e) {
    if (!e) return fail(PSSBAM_EINVAL, "null engine");
    HIP_TRY(hipSetDevice(e->device));
    {
        const int rc = feed_flush(e);   // compressed blocks still being collected (pssbam_engine_submit_bgzf)
        if (rc) return rc;
    }
    if (e->copy_stream2) HIP_TRY(hipStreamSynchronize(e->copy_stream2));
    if (e->copy_stream) HIP_TRY(hipStreamSynchronize(e->copy_stream));
    if (e->genome_stream) HIP_TRY(hipStreamSynchronize(e->genome_stream));
    for (hipStream_t q : e->inflate_stream)
        if (q) HIP_TRY(hipStreamSynchronize(q));
    HIP_TRY(hipStreamSynchronize(e->stream));
    const int rc = genome_settle(e);
    if (rc) return rc;
    // (the streams are quiet either way: the caller's buffers are free)
    if (!e->deferred.empty())
        return fail(PSSBAM_ESTATE, "compressed blocks were fed (pssbam_engine_feed_open) but set_genome / set_references never followed");
    return PSSBAM_OK;
}

extern "C" int pssbam_engine_finish(pssbam_engine *e, unsigned long *fwd, unsigned long *rev, uint64_t *k5,
                                    uint64_t *k3, uint64_t stats[PSSBAM_ST_N]) {
    int rc = pssbam_engine_sync(e);
    if (rc) return rc;
    std::vector<unsigned long long> h(e->n_counters);
    HIP_TRY(hipMemcpy(h.data(), e->d_counters, e->n_counters * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    static_assert(sizeof(unsigned long) == 8, "LP64 expected");
    const size_t tab = (size_t)e->rows * 16;
    if (fwd) for (size_t i = 0; i < tab; i++) fwd[i] = (unsigned long)h[i];
    if (rev) for (size_t i = 0; i < tab; i++) rev[i] = (unsigned long)h[e->off_rev + i];
    if (k5) for (uint64_t i = 0; i < e->n_bins; i++) k5[i] = h[e->off_k5 + i];
    if (k3) for (uint64_t i = 0; i < e->n_bins; i++) k3[i] = h[e->off_k3 + i];
    if (stats) for (int i = 0; i < PSSBAM_ST_N; i++) stats[i] = h[e->off_stats + i];
    return PSSBAM_OK;
}

extern "C" int pssbam_engine_reset(pssbam_engine *e) {
    if (!e) return fail(PSSBAM_EINVAL, "null engine");
    HIP_TRY(hipSetDevice(e->device));
    e->feed_fresh = true;   // a compressed stream fed from here on starts a new record chain
    e->feed_skip = 0;
    if (!e->deferred.empty()) {   // inflated ahead of the genome and now given up: the slots go back once their kernels have run
        e->deferred.clear();
        for (FeedAcc *sp : e->feed)
            if (sp->held) {
                sp->held = false;
                if (!sp->consumed) HIP_TRY(hipEventCreateWithFlags(&sp->consumed, hipEventDisableTiming));
                HIP_TRY(hipEventRecord(sp->consumed, e->stream));
            }
    }
    if (e->d_feed_tail) HIP_TRY(hipMemsetAsync(e->d_feed_tail, 0, sizeof(uint64_t), e->stream));
    if (e->d_feed_flags) HIP_TRY(hipMemsetAsync(e->d_feed_flags, 0, sizeof(uint32_t), e->stream));
    HIP_TRY(hipMemsetAsync(e->d_counters, 0, e->n_counters * sizeof(unsigned long long), e->stream));
    return PSSBAM_OK;
}

extern "C" int pssbam_engine_counters_device(pssbam_engine *e, void **d_counters, size_t *n_u64) {
    if (!e || !d_counters || !n_u64) return fail(PSSBAM_EINVAL, "null argument");
    *d_counters = e->d_counters;
    *n_u64 = e->n_counters;
    return PSSBAM_OK;
}

extern "C" int pssbam_engine_genome_kmer_count(pssbam_engine *e, int klen, uint64_t *counts) {
    if (!e || !counts) return fail(PSSBAM_EINVAL, "null argument");
    if (klen < 1 || klen > PSSBAM_MAX_KLEN) return fail(PSSBAM_EINVAL, "klen %d outside the device range 1..%d", klen, PSSBAM_MAX_KLEN);
    if (!e->d_genome) return fail(PSSBAM_ESTATE, "set_genome has not been called");
    HIP_TRY(hipSetDevice(e->device));
    if (e->genome_wait_pending) {
        HIP_TRY(hipStreamWaitEvent(e->stream, e->genome_ready, 0));
        e->genome_wait_pending = false;
    }
    const size_t nb = (size_t)1 << (2 * klen);
    unsigned long long *d_bins = nullptr;
    HIP_TRY(hipMalloc(&d_bins, nb * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(d_bins, 0, nb * sizeof(unsigned long long), e->stream));
    hipError_t le = hipSuccess;
    if (klen <= 8 && e->d_genome4 && !getenv("PSSBAM_GKC_BYTES")) {
        // histogram in LDS from the packed genome; replication of the bins while they are few
        const uint32_t n_all = 1u << (2 * klen);
        const uint32_t n_bins = std::min<uint32_t>(n_all, 32768u);                 // per pass: <= 128 KiB of u32
        uint32_t rep_log2 = 0;
        while ((n_bins << (rep_log2 + 1)) * 4u <= 32768u && rep_log2 < 5) rep_log2++;  // up to 32 KiB of replicas
        const uint32_t lds = (n_bins << rep_log2) * 4u;
        le = hipFuncSetAttribute((const void *)genome_kmer_packed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const uint64_t spans = (e->genome_bytes + GKC4_SPAN - 1) / GKC4_SPAN;
        const uint32_t per_cu = lds > 65536u ? 1u : 2u;
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((spans + 511) / 512, (uint64_t)e->n_cu * per_cu);
        for (uint32_t lo = 0; lo < n_all && le == hipSuccess; lo += n_bins) {
            hipLaunchKernelGGL(genome_kmer_packed_kernel, dim3(blocks), dim3(512), lds, e->stream, e->d_genome4, e->genome_bytes, klen,
                               lo, n_bins, rep_log2, d_bins);
            le = hipGetLastError();
        }
    } else {
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((e->genome_bytes / GKC_SPAN + 255) / 256 + 1, (uint64_t)e->n_cu * 16);
        if (klen <= 6) hipLaunchKernelGGL(genome_kmer_kernel<true>, dim3(blocks), dim3(256), 0, e->stream, e->d_genome, e->genome_bytes, klen, d_bins);
        else hipLaunchKernelGGL(genome_kmer_kernel<false>, dim3(blocks), dim3(256), 0, e->stream, e->d_genome, e->genome_bytes, klen, d_bins);
        le = hipGetLastError();
    }
    if (le == hipSuccess) le = hipMemcpyAsync(counts, d_bins, nb * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream);
    if (le == hipSuccess) le = hipStreamSynchronize(e->stream);
    (void)hipFree(d_bins);
    if (le != hipSuccess) return fail(PSSBAM_EHIP, "genome k-mer count failed: %s", hipGetErrorString(le));
    return PSSBAM_OK;
}

extern "C" int pssbam_engine_bind_counters(pssbam_engine *e, void *d_counters, size_t n_u64) {
    if (!e) return fail(PSSBAM_EINVAL, "null engine");
    HIP_TRY(hipSetDevice(e->device));
    unsigned long long *target = d_counters ? (unsigned long long *)d_counters : e->d_counters_own;
    if (d_counters && (n_u64 != e->n_counters || ((uintptr_t)d_counters & 7u)))
        return fail(PSSBAM_EINVAL, "bound counter block must hold exactly %zu aligned u64 words", e->n_counters);
    if (target != e->d_counters) {
        HIP_TRY(hipMemcpyAsync(target, e->d_counters, e->n_counters * sizeof(unsigned long long),
                               hipMemcpyDeviceToDevice, e->stream));
        e->d_counters = target;
    }
    return PSSBAM_OK;
}

extern "C" int pssbam_engine_timer_begin(pssbam_engine *e) {
    if (!e) return fail(PSSBAM_EINVAL, "null engine");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipEventRecord(e->t_begin, e->stream));
    return PSSBAM_OK;
}

extern "C" int pssbam_engine_timer_end(pssbam_engine *e, float *ms) {
    if (!e || !ms) return fail(PSSBAM_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipEventRecord(e->t_end, e->stream));
    HIP_TRY(hipEventSynchronize(e->t_end));
    HIP_TRY(hipEventElapsedTime(ms, e->t_begin, e->t_end));
    return PSSBAM_OK;
}

extern "C" int pssbam_engine_kernel_time(pssbam_engine *e, double *total_ms, uint64_t *n_launches, int reset) {
    if (!e) return fail(PSSBAM_EINVAL, "null engine");
    HIP_TRY(hipSetDevice(e->device));
    int rc = resolve_launch_events(e);
    if (rc) return rc;
    if (total_ms) *total_ms = e->kernel_ms;
    if (n_launches) *n_launches = e->kernel_launches;
    if (reset) { e->kernel_ms = 0.0; e->kernel_launches = 0; }
    return PSSBAM_OK;
}

// --------------------------------------------------------------------------------------
// node-level reduce (single process, several devices) over RCCL
// --------------------------------------------------------------------------------------
#include <dlfcn.h>

namespace {
// the handful of RCCL entry points used, resolved from librccl.so on first use so that
// single-GPU users never pay for loading it
typedef struct ncclComm *ncclComm_t;
typedef int (*fn_CommInitAll)(ncclComm_t *, int, const int *);
typedef int (*fn_CommDestroy)(ncclComm_t);
typedef int (*fn_GroupStart)(void);
typedef int (*fn_GroupEnd)(void);
typedef int (*fn_Reduce)(const void *, void *, size_t, int, int, int, ncclComm_t, hipStream_t);
typedef const char *(*fn_GetErrorString)(int);
constexpr int NCCL_UINT64 = 5, NCCL_SUM = 0;  // rccl.h: ncclUint64, ncclSum
struct Rccl {
    void *h = nullptr;
    fn_CommInitAll CommInitAll = nullptr;
    fn_CommDestroy CommDestroy = nullptr;
    fn_GroupStart GroupStart = nullptr;
    fn_GroupEnd GroupEnd = nullptr;
    fn_Reduce Reduce = nullptr;
    fn_GetErrorString GetErrorString = nullptr;
};
Rccl g_rccl;
bool load_rccl() {
    if (g_rccl.h) return true;
    void *h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return false;
    g_rccl.CommInitAll = (fn_CommInitAll)dlsym(h, "ncclCommInitAll");
    g_rccl.CommDestroy = (fn_CommDestroy)dlsym(h, "ncclCommDestroy");
    g_rccl.GroupStart = (fn_GroupStart)dlsym(h, "ncclGroupStart");
    g_rccl.GroupEnd = (fn_GroupEnd)dlsym(h, "ncclGroupEnd");
    g_rccl.Reduce = (fn_Reduce)dlsym(h, "ncclReduce");
    g_rccl.GetErrorString = (fn_GetErrorString)dlsym(h, "ncclGetErrorString");
    if (!g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.GroupStart || !g_rccl.GroupEnd || !g_rccl.Reduce) {
        dlclose(h);
        return false;
    }
    g_rccl.h = h;
    return true;
}
}  // namespace

// One grouped ncclReduce over the engines' counter blocks (single process, one communicator per
// GPU).  PSSBAM_OK = done; 1 = RCCL is not usable here and nothing was touched (the caller sums on
// the host instead); negative = a collective failed half-way, the counters are not trustworthy.
static int reduce_rccl(pssbam_engine *const *engines, int n, int root) {
    if (!load_rccl()) return 1;
    std::vector<ncclComm_t> comms(n);
    std::vector<int> devs(n);
    for (int i = 0; i < n; i++) devs[i] = engines[i]->device;
    int rc = g_rccl.CommInitAll(comms.data(), n, devs.data());
    if (rc != 0) return 1;
    rc = g_rccl.GroupStart();
    for (int i = 0; i < n && rc == 0; i++) {
        (void)hipSetDevice(engines[i]->device);
        rc = g_rccl.Reduce(engines[i]->d_counters, engines[i]->d_counters, engines[i]->n_counters, NCCL_UINT64, NCCL_SUM,
                           root, comms[i], engines[i]->stream);
    }
    const int rc_end = g_rccl.GroupEnd();
    if (rc == 0) rc = rc_end;
    for (int i = 0; i < n; i++) {
        (void)hipSetDevice(engines[i]->device);
        (void)hipStreamSynchronize(engines[i]->stream);
    }
    for (int i = 0; i < n; i++) (void)g_rccl.CommDestroy(comms[i]);
    if (rc != 0) return fail(PSSBAM_EHIP, "ncclReduce failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return PSSBAM_OK;
}

extern "C" int pssbam_reduce_counters(pssbam_engine *const *engines, int n, int root) {
    if (!engines || n < 1 || root < 0 || root >= n) return fail(PSSBAM_EINVAL, "bad argument");
    for (int i = 0; i < n; i++) {
        if (!engines[i]) return fail(PSSBAM_EINVAL, "null engine %d", i);
        if (engines[i]->n_counters != engines[0]->n_counters)
            return fail(PSSBAM_EINVAL, "engines were created with different options");
        int rc = pssbam_engine_sync(engines[i]);
        if (rc) return rc;
    }
    if (n == 1) return PSSBAM_OK;
    // RCCL needs one distinct device per rank; anything else (two engines on one GPU, librccl
    // missing, a failing communicator) takes the host-side sum below, which is also the
    // cross-check path (PSSBAM_REDUCE=host)
    bool distinct = true;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++) distinct = distinct && engines[i]->device != engines[j]->device;
    // ... and it has to be worth a communicator: ncclCommInitAll over 8 GPUs takes seconds, the pss tables are 7 KB per
    // engine (8 small copies and a host add: microseconds) -- RCCL carries the block from 32 MiB up (k-mer bins at
    // k >= 11), or when asked to (PSSBAM_REDUCE=rccl)
    const char *force = getenv("PSSBAM_REDUCE");
    const bool big = engines[0]->n_counters * sizeof(unsigned long long) >= (32ull << 20);
    if (distinct && !(force && !strcmp(force, "host")) && (big || (force && !strcmp(force, "rccl")))) {
        const int rc = reduce_rccl(engines, n, root);
        if (rc <= 0) return rc;
    }

    const size_t nc = engines[0]->n_counters;
    std::vector<unsigned long long> sum(nc, 0ull), part(nc);
    for (int i = 0; i < n; i++) {
        HIP_TRY(hipSetDevice(engines[i]->device));
        HIP_TRY(hipMemcpy(part.data(), engines[i]->d_counters, nc * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (size_t k = 0; k < nc; k++) sum[k] += part[k];
    }
    HIP_TRY(hipSetDevice(engines[root]->device));
    HIP_TRY(hipMemcpy(engines[root]->d_counters, sum.data(), nc * sizeof(unsigned long long), hipMemcpyHostToDevice));
    return PSSBAM_OK;
}

extern "C" int pssbam_host_register(void *ptr, size_t bytes) {
    if (!ptr || !bytes) return fail(PSSBAM_EINVAL, "bad argument");
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return PSSBAM_OK;
}

extern "C" int pssbam_host_unregister(void *ptr) {
    if (!ptr) return fail(PSSBAM_EINVAL, "bad argument");
    HIP_TRY(hipHostUnregister(ptr));
    return PSSBAM_OK;
}

// --------------------------------------------------------------------------------------
// host helper: record index
// --------------------------------------------------------------------------------------
extern "C" int64_t pssbam_index_records(const void *bytes, uint64_t nbytes, uint32_t *offsets, uint64_t max_records,
                                        uint64_t *consumed) {
    const uint8_t *p = (const uint8_t *)bytes;
    uint64_t o = 0, n = 0;
    while (n < max_records && o + 4 <= nbytes && o < (1ull << 32) - 4) {
        uint32_t bs;
        memcpy(&bs, p + o, 4);
        if (bs < 32) { fail(PSSBAM_EFORMAT, "record %llu: block_size %u < 32", (unsigned long long)n, bs); return PSSBAM_EFORMAT; }
        const uint64_t next = o + 4 + (uint64_t)bs;
        if (next > nbytes || next >= (1ull << 32)) break;  // partial record: caller supplies more bytes
        if (offsets) offsets[n] = (uint32_t)o;
        n++;
        o = next;
    }
    if (offsets) offsets[n] = (uint32_t)o;
    if (consumed) *consumed = o;
    return (int64_t)n;
}

// --------------------------------------------------------------------------------------
// device-side BGZF inflate
// --------------------------------------------------------------------------------------
#include "bgzf_api.h"
