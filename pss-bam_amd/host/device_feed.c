/*
 * pss-bam_amd/host/device_feed.c -- the BAM feed with the inflate on the GPU.
 *
 * Replaces the same thing as bam_reader.c -- the `samtools view` child of the reference
 * (pss-bam.c:148-162, fragkon.c:84-93) -- but moves the decompression to where the records are
 * consumed: the host only walks BGZF block headers and hands COMPRESSED chunks to
 * pssbam_engine_submit_bgzf(); inflate, CRC-32, record index and tally all run on the device
 * (csrc/inflate_kernels.h).  PCIe carries the file (a tenth to a third of the record bytes) and the
 * host's cores are out of the data path:
 *
 *   loader threads   pread() fixed windows of the file into page-locked staging slots, in parallel
 *   this thread      walks the block headers of each window in file order (18 + 8 bytes per block),
 *                    cuts batches that inflate to < 1 GiB, and submits them -- asynchronously, in
 *                    runs of consecutive batches per GPU (contiguous record blocks)
 *
 * Records may cross BGZF blocks (htsjdk writers) and batches: the device stitches the record chain from
 * per-block pieces and carries the partial record at a super-batch's end into the next one.  Only
 * when the pieces do not link up (PSSBAM_FEED_RAGGED: a false start, a record above 16 MiB) does the
 * caller re-run the file through the host reader, whose indexer walks the chain serially.
 */
#include "device_feed.h"

#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

static double mono_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

#define MAX_STAGE 40
#define BLOCKS_PER_SCAN 65536u
#define OVER ((size_t)128 << 10) /* bytes read past a window: the block that starts inside it ends inside this */

typedef struct {
    uint8_t *buf;          /* window bytes [k*W, k*W + W + OVER) of the file */
    size_t len;
    long free_for;         /* chunk index that may load into this slot next */
    long loaded;           /* chunk index whose bytes are in buf, or -1 */
    int io_error;
    /* the loader's own walk of the window's block headers, from the first offset at which a chain of
     * BGZF headers starts (offset 0 in chunk 0): used by the submitting thread iff that offset is where
     * the previous chunk's chain really ends -- the walk then costs the submitting thread nothing */
    pssbam_bgzf_block *pre;
    size_t pre_start;      /* offset in buf the walk started from */
    int64_t pre_n;         /* blocks found (-1: none) */
} stage_t;

typedef struct {
    int fd;
    size_t file_size, W;
    long n_chunks;
    stage_t st[MAX_STAGE];
    int n_st;
    atomic_long next_chunk;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int stop;
} loader_t;

static void *loader_main(void *arg)
{
    loader_t *L = (loader_t *)arg;
    for (;;) {
        const long k = atomic_fetch_add(&L->next_chunk, 1);
        if (k >= L->n_chunks) break;
        stage_t *s = &L->st[k % L->n_st];
        pthread_mutex_lock(&L->mu);
        while (s->free_for != k && !L->stop) pthread_cond_wait(&L->cv, &L->mu);
        const int stop = L->stop;
        pthread_mutex_unlock(&L->mu);
        if (stop) break;
        const size_t off = (size_t)k * L->W;
        size_t want = L->file_size - off < L->W + OVER ? L->file_size - off : L->W + OVER, got = 0;
        int bad = 0;
        while (got < want) {
            const ssize_t n = pread(L->fd, s->buf + got, want - got, (off_t)(off + got));
            if (n < 0 && errno == EINTR) continue;
            if (n <= 0) { bad = 1; break; }
            got += (size_t)n;
        }
        memset(s->buf + got, 0, 16); /* the device decoder may read one dword past the payload */
        s->pre_n = -1;
        if (!bad && s->pre) {
            size_t st = 0;
            int found = k == 0;
            for (; !found && st + 64 < got && st < ((size_t)66 << 10); st++) { /* a block is at most 64 KiB: one starts in here */
                const uint8_t *h = s->buf + st;
                if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || h[3] != 4 || h[12] != 'B' || h[13] != 'C' || h[14] != 2 || h[15] != 0) continue;
                size_t o = st; /* three headers in a row make a false start improbable; the submitting thread checks anyway */
                int hops = 0;
                while (hops < 3 && o + 18 <= got && s->buf[o] == 0x1f && s->buf[o + 1] == 0x8b && s->buf[o + 12] == 'B' && s->buf[o + 13] == 'C') {
                    o += (size_t)(s->buf[o + 16] | (s->buf[o + 17] << 8)) + 1;
                    hops++;
                }
                if (hops == 3 || o >= got) found = 1;
                if (found) break;
            }
            if (found) {
                uint64_t consumed = 0;
                s->pre_start = st;
                s->pre_n = pssbam_bgzf_scan(s->buf + st, got - st, s->pre, BLOCKS_PER_SCAN, &consumed, NULL);
            }
        }
        pthread_mutex_lock(&L->mu);
        s->len = got;
        s->io_error = bad;
        s->loaded = k;
        pthread_cond_broadcast(&L->cv);
        pthread_mutex_unlock(&L->mu);
    }
    return NULL;
}

static size_t env_size(const char *name, size_t dflt)
{
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    const unsigned long long x = strtoull(v, NULL, 10);
    return x ? (size_t)x : dflt;
}

int device_feed_enabled(void)
{
    const char *v = getenv("PSSBAM_DEVICE_INFLATE");
    return v ? atoi(v) != 0 : 1;
}

/* a loader with its staging slots and threads: opened by run_device_feed, or ahead of it by
 * device_feed_prefetch (the first windows are then in the slots when the engines are ready) */
typedef struct feed_loader {
    loader_t L;
    pthread_t th[32];
    int n_th;
    uint8_t *stage_base;
    size_t stage_bytes;
    int registered, max_inflight;
    char path[4096];
} feed_loader;

static void loader_close(feed_loader *F)
{
    if (!F) return;
    pthread_mutex_lock(&F->L.mu);
    F->L.stop = 1;
    pthread_cond_broadcast(&F->L.cv);
    pthread_mutex_unlock(&F->L.mu);
    for (int t = 0; t < F->n_th; t++) pthread_join(F->th[t], NULL);
    if (F->registered) pssbam_host_unregister(F->stage_base);
    for (int i = 0; i < F->L.n_st; i++) free(F->L.st[i].pre);
    free(F->stage_base);
    if (F->L.fd >= 0) close(F->L.fd);
    pthread_mutex_destroy(&F->L.mu);
    pthread_cond_destroy(&F->L.cv);
    free(F);
}

/* NULL: not a regular file (*not_regular = 1: pipes and the like, the host reader copes) or out of memory */
static feed_loader *loader_open(const char *path, int n_gpus, int *not_regular)
{
    *not_regular = 0;
    if (strlen(path) >= sizeof ((feed_loader *)0)->path) return NULL;
    feed_loader *F = (feed_loader *)calloc(1, sizeof *F);
    if (!F) return NULL;
    loader_t *L = &F->L;
    strcpy(F->path, path);
    struct stat sb;
    L->fd = open(path, O_RDONLY);
    if (L->fd < 0 || fstat(L->fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size <= 0) {
        if (L->fd >= 0) close(L->fd);
        free(F);
        *not_regular = 1;
        return NULL;
    }
    L->file_size = (size_t)sb.st_size;
    L->W = env_size("PSSBAM_CHUNK_BYTES", (size_t)32 << 20);
    if (L->W < ((size_t)1 << 20)) L->W = (size_t)1 << 20;
    L->W &= ~(size_t)4095;
    L->n_chunks = (long)((L->file_size + L->W - 1) / L->W);
    F->max_inflight = n_gpus * 2 < 2 ? 2 : n_gpus * 2;
    L->n_st = F->max_inflight + 8 > MAX_STAGE ? MAX_STAGE : F->max_inflight + 8;   /* windows being read ahead + in flight */
    if ((long)L->n_st > L->n_chunks + 1) L->n_st = (int)L->n_chunks + 1;
    const size_t slot_bytes = (L->W + OVER + 4096 + 4095) & ~(size_t)4095;
    pthread_mutex_init(&L->mu, NULL);
    pthread_cond_init(&L->cv, NULL);
    F->stage_bytes = slot_bytes * (size_t)L->n_st;
    if (posix_memalign((void **)&F->stage_base, 4096, F->stage_bytes) != 0) {
        F->stage_base = NULL;
        loader_close(F);
        return NULL;
    }
    for (int i = 0; i < L->n_st; i++) {
        L->st[i].buf = F->stage_base + (size_t)i * slot_bytes;
        L->st[i].free_for = i;
        L->st[i].loaded = -1;
        L->st[i].pre = (pssbam_bgzf_block *)malloc(sizeof(pssbam_bgzf_block) * BLOCKS_PER_SCAN); /* NULL: no pre-scan, that is all */
    }
    long cpus = sysconf(_SC_NPROCESSORS_ONLN);
    int want = (int)env_size("PSSBAM_LOADER_THREADS", n_gpus > 2 ? 4 * (size_t)n_gpus : 8);
    if (want > 32) want = 32;
    if (cpus > 0 && want > cpus) want = (int)cpus;
    if ((long)want > L->n_chunks) want = (int)L->n_chunks;
    for (int t = 0; t < want; t++)
        if (pthread_create(&F->th[F->n_th], NULL, loader_main, L) == 0) F->n_th++;
    if (!F->n_th) {
        loader_close(F);
        return NULL;
    }
    return F;
}

static void loader_pin(feed_loader *F)
{
    if (F && !F->registered && !getenv("PSSBAM_NO_PIN")) F->registered = pssbam_host_register(F->stage_base, F->stage_bytes) == 0;
}

/* ---- start-up overlap: the front end opens the loader while the FASTA is still being parsed ---- */
static feed_loader *g_pre = NULL;
/* a loader whose feed has ended: its threads are done, but un-pinning and freeing 300+ MB of staging slots costs tens of
 * ms -- the caller's tables are ready without that (device_feed_prefetch_cancel does it, or process exit) */
static feed_loader *g_retired = NULL;

void device_feed_prefetch(const char *path)
{
    const char *ng = getenv("PSSBAM_NGPU");
    if (g_pre || !device_feed_enabled()) return;
    int not_regular, n = ng ? atoi(ng) : 1;   /* (the HIP runtime is not up yet: the count asked for sizes the slots) */
    g_pre = loader_open(path, n < 1 ? 1 : n > 64 ? 64 : n, &not_regular);
}

void device_feed_prefetch_pin(void) { loader_pin(g_pre); }   /* needs the HIP runtime: called by the warm-up thread */

void device_feed_prefetch_cancel(void)
{
    loader_close(g_pre);
    g_pre = NULL;
    loader_close(g_retired);
    g_retired = NULL;
}

int run_device_feed(pssbam_engine *const *eng, int n_gpus, const char *path, size_t header_bytes, int run, int verbose,
                    device_feed_stats *fs, const feed_gate *gate)
{
    int rc = -1;
    pssbam_bgzf_block *blocks = NULL, *grp = NULL;
    feed_loader *F = NULL;
    memset(fs, 0, sizeof *fs);
    int gate_open = gate == NULL;   /* genome + references are on the engines */
    if (g_pre && strcmp(g_pre->path, path) == 0) {
        F = g_pre;
        g_pre = NULL;
    } else {
        int not_regular = 0;
        device_feed_prefetch_cancel();
        F = loader_open(path, n_gpus, &not_regular);
        if (!F) {
            if (not_regular) { fs->fallback = 1; return 0; }
            fprintf(stderr, "Error: %s: cannot set up the device feed (memory / threads)\n", path);
            return -1;
        }
    }
    loader_t *const L = &F->L;
    const int n_th = F->n_th, max_inflight = F->max_inflight;
    size_t out_cap = env_size("PSSBAM_FEED_BATCH_BYTES", (size_t)768 << 20); /* inflated bytes per submit */
    if (out_cap > ((size_t)1 << 30)) out_cap = (size_t)1 << 30;
    /* several engines: a run per engine is as long as one of its super-batches (so the inflate kernel of every
     * GPU still gets full launches; a run switch flushes what the engine has collected), while the number of
     * copies in flight only has to cover the links' latency */
    if (n_gpus > 1 && !getenv("PSSBAM_RUN_BATCHES")) {
        const size_t per_super = ((size_t)10800 << 20) / out_cap;
        if ((size_t)run < per_super) run = (int)per_super;
    }
    blocks = (pssbam_bgzf_block *)malloc(sizeof *blocks * BLOCKS_PER_SCAN);
    grp = (pssbam_bgzf_block *)malloc(sizeof *grp * BLOCKS_PER_SCAN);
    if (!blocks || !grp) goto done;
    loader_pin(F);

    /* in-flight submits: (stage slot or -1, engine, ticket); a stage slot is handed back to the loaders
     * when the last submit that reads it has been copied */
    struct { long chunk; int g; uint64_t ticket; } fifo[128];
    int fifo_head = 0, fifo_len = 0;
    long pending_of_chunk[MAX_STAGE];
    memset(pending_of_chunk, 0, sizeof pending_of_chunk);
#define RETIRE()                                                                                         \
    do {                                                                                                 \
        const int q = fifo_head;                                                                         \
        const double tr_ = mono_s();                                                                     \
        if (pssbam_engine_wait_bgzf_copied(eng[fifo[q].g], fifo[q].ticket)) goto done;                    \
        t_wait_copy += mono_s() - tr_;                                                                   \
        stage_t *rs = &L->st[fifo[q].chunk % L->n_st];                                                      \
        if (--pending_of_chunk[fifo[q].chunk % L->n_st] == 0) {                                            \
            pthread_mutex_lock(&L->mu);                                                                    \
            rs->free_for = fifo[q].chunk + L->n_st;                                                        \
            rs->loaded = -1;                                                                              \
            pthread_cond_broadcast(&L->cv);                                                                \
            pthread_mutex_unlock(&L->mu);                                                                  \
        }                                                                                                 \
        fifo_head = (fifo_head + 1) % 128;                                                                \
        fifo_len--;                                                                                       \
    } while (0)

    double t_wait_load = 0, t_wait_copy = 0, t_scan = 0, t_submit = 0, t_gate = 0;
    const double t_entry = mono_s();
    double t_first_submit = -1, t_first_submit_dur = 0;   /* when the first chunk went to an engine, and how long that call took */
    uint64_t n_submits_ahead = 0;   /* submits that went in before the genome was set */
    long n_prescanned = 0;
    size_t pos = 0;                /* file offset of the next BGZF block */
    size_t skip = header_bytes;    /* inflated bytes still to skip in front of the first record */
    uint64_t n_submits = 0, submits_of[64] = {0};
    int last_g = -1;
    for (long k = 0; k < L->n_chunks; k++) {
        stage_t *s = &L->st[k % L->n_st];
        const double tw = mono_s();
        pthread_mutex_lock(&L->mu);
        while (s->loaded != k) pthread_cond_wait(&L->cv, &L->mu);
        pthread_mutex_unlock(&L->mu);
        t_wait_load += mono_s() - tw;
        if (s->io_error) { fprintf(stderr, "Error: %s: read failed\n", path); goto done; }
        const size_t win0 = (size_t)k * L->W, win_end = win0 + L->W; /* blocks STARTING in [win0, win_end) are this chunk's */
        int submitted_from_chunk = 0;
        while (pos < win_end && pos < L->file_size) {
            if (pos < win0) { fprintf(stderr, "Error: %s: BGZF block chain lost\n", path); goto done; }
            uint64_t consumed = 0, inflated = 0;
            const double ts = mono_s();
            int64_t n;
            const pssbam_bgzf_block *bl = blocks;
            if (s->pre && s->pre_n > 0 && s->pre_start == pos - win0) { /* the loader walked exactly this chain already */
                n = s->pre_n;
                bl = s->pre;
                s->pre_n = -1; /* (a second pass over this window, after 65536 blocks, walks for itself) */
                n_prescanned++;
            } else
                n = pssbam_bgzf_scan(s->buf + (pos - win0), s->len - (pos - win0), blocks, BLOCKS_PER_SCAN, &consumed, &inflated);
            t_scan += mono_s() - ts;
            if (n < 0) { fprintf(stderr, "Error: %s: %s\n", path, pssbam_last_error()); goto done; }
            if (n == 0) {
                if (win0 + s->len >= L->file_size) { fprintf(stderr, "Error: %s: truncated BGZF block at end of file\n", path); goto done; }
                fprintf(stderr, "Error: %s: BGZF block larger than the read-ahead\n", path);
                goto done;
            }
            /* keep the blocks that start inside this window; in_off is relative to buf + (pos - win0) */
            int64_t keep = 0;
            size_t start = pos;
            while (keep < n && start < win_end) {
                start = pos + (size_t)(bl[keep].in_off + bl[keep].in_len + 8);
                keep++;
            }
            const size_t chunk_rel = pos - win0;
            pos = start;
            /* cut into submits that inflate to <= out_cap; all-header blocks of the file's start are skipped */
            int64_t i = 0;
            while (i < keep && skip > 0 && skip >= bl[i].isize) { skip -= bl[i].isize; i++; }
            while (i < keep) {
                int64_t j = i;
                const uint64_t base_out = bl[i].out_off;
                while (j < keep && bl[j].out_off + bl[j].isize - base_out <= out_cap) j++;
                if (j == i) j = i + 1;
                const uint64_t base_in = bl[i].in_off & ~(uint64_t)3;
                for (int64_t b = i; b < j; b++) {
                    grp[b - i] = bl[b];
                    grp[b - i].in_off -= base_in;
                    grp[b - i].out_off -= base_out;
                }
                const uint64_t end_in = bl[j - 1].in_off + bl[j - 1].in_len;
                const int g = (int)((n_submits / (uint64_t)run) % (uint64_t)n_gpus);
                if (n_gpus > 1 && g != last_g) { /* the run of engine last_g ends here: the record it may end in is completed by engine g */
                    if (last_g >= 0 && pssbam_engine_feed_handoff(eng[last_g], eng[g])) { fprintf(stderr, "Error: GPU engine: %s\n", pssbam_last_error()); goto done; }
                    last_g = g;
                }
                uint64_t ticket = 0;
                if (!gate_open) {   /* has the genome arrived?  (sets it on the engines, which then tally what they inflated ahead) */
                    const int r = gate->poll(gate->ctx, 0);
                    if (r < 0) goto done;
                    gate_open = r > 0;
                }
                const double tsub = mono_s();
                int src = pssbam_engine_submit_bgzf(eng[g], s->buf + chunk_rel + base_in, end_in - base_in, grp, (uint32_t)(j - i), (uint32_t)skip,
                                                    &ticket);
                if (src == PSSBAM_EBUSY && !gate_open) {   /* every slot holds records that wait for the genome: so do we */
                    const double tg = mono_s();
                    if (gate->poll(gate->ctx, 1) <= 0) goto done;
                    t_gate += mono_s() - tg;
                    gate_open = 1;
                    src = pssbam_engine_submit_bgzf(eng[g], s->buf + chunk_rel + base_in, end_in - base_in, grp, (uint32_t)(j - i), (uint32_t)skip, &ticket);
                }
                if (src) {
                    fprintf(stderr, "Error: GPU engine: %s\n", pssbam_last_error());
                    goto done;
                }
                t_submit += mono_s() - tsub;
                if (t_first_submit < 0) { t_first_submit = tsub - t_entry; t_first_submit_dur = mono_s() - tsub; }
                if (!gate_open) n_submits_ahead++;
                skip = 0;
                n_submits++;
                submits_of[g & 63]++;
                fs->compressed_bytes += end_in - base_in;
                const int q = (fifo_head + fifo_len) % 128;
                fifo[q].chunk = k; fifo[q].g = g; fifo[q].ticket = ticket;
                fifo_len++;
                pending_of_chunk[k % L->n_st]++;
                submitted_from_chunk++;
                while (fifo_len > max_inflight || fifo_len >= 127) RETIRE();
                i = j;
            }
        }
        if (!submitted_from_chunk) { /* nothing read this slot: give it straight back */
            pthread_mutex_lock(&L->mu);
            s->free_for = k + L->n_st;
            s->loaded = -1;
            pthread_cond_broadcast(&L->cv);
            pthread_mutex_unlock(&L->mu);
        }
    }
    while (fifo_len > 0) RETIRE();
#undef RETIRE
    if (pos != L->file_size) { fprintf(stderr, "Error: %s: truncated BGZF block at end of file\n", path); goto done; }
    if (!gate_open) {   /* the whole file went in ahead of the genome */
        const double tg = mono_s();
        if (gate->poll(gate->ctx, 1) <= 0) goto done;
        t_gate += mono_s() - tg;
        gate_open = 1;
    }
    fs->n_submits = n_submits;
    /* how the blocks fared: one word per engine */
    for (int g = 0; g < n_gpus; g++) {
        uint32_t f = 0;
        double ms = 0;
        uint64_t ib = 0;
        if (pssbam_engine_feed_status(eng[g], &f, &ms, &ib)) { fprintf(stderr, "Error: GPU engine %d: %s\n", g, pssbam_last_error()); goto done; }
        fs->flags |= f;
        fs->inflate_ms += ms;
        fs->inflated_bytes += ib;
    }
    if (fs->flags & (PSSBAM_FEED_RAGGED | PSSBAM_FEED_BAD_RECORD)) fs->fallback = 1; /* the host reader follows records across blocks (and words the diagnosis) */
    else if (fs->flags & PSSBAM_FEED_BAD_BLOCK) { fprintf(stderr, "Error: %s: BGZF inflate / CRC check failed\n", path); goto done; }
    else if (fs->flags & PSSBAM_FEED_TRUNCATED) { fprintf(stderr, "Error: %s: truncated alignment record at end of file\n", path); goto done; }
    if (verbose)
        fprintf(stderr, "[pssbam] device feed: %llu submits, %.2f GB compressed over PCIe, %.2f GB inflated on the device in %.3f s "
                        "of kernel time (%.1f GB/s), %d loader threads, %d staging slots of %zu MiB%s\n",
                (unsigned long long)n_submits, fs->compressed_bytes * 1e-9, fs->inflated_bytes * 1e-9, fs->inflate_ms * 1e-3,
                fs->inflate_ms > 0 ? fs->inflated_bytes * 1e-6 / fs->inflate_ms : 0.0, n_th, L->n_st, L->W >> 20,
                fs->fallback ? "; records cross BGZF blocks -> host reader" : "");
    if (verbose && n_gpus > 1) {
        fprintf(stderr, "[pssbam] device feed, submits per engine:");
        for (int g = 0; g < n_gpus; g++) fprintf(stderr, " %llu", (unsigned long long)submits_of[g & 63]);
        fputc('\n', stderr);
    }
    if (verbose)
        fprintf(stderr, "[pssbam] device feed, this thread: waiting for loaders %.3f, block-header walk %.3f (%ld of %ld windows walked by "
                        "their loader), submit (incl. waiting for a free slot) %.3f, waiting for copies %.3f, waiting for the genome %.3f s "
                        "(%llu of the submits went in ahead of it); first submit %.3f s after the feed began (the call took %.3f s)\n", t_wait_load, t_scan,
                n_prescanned, L->n_chunks, t_submit, t_wait_copy, t_gate, (unsigned long long)n_submits_ahead, t_first_submit, t_first_submit_dur);
    rc = 0;
done:
    if (rc) {   /* nothing may still read the staging slots */
        pthread_mutex_lock(&L->mu);
        L->stop = 1;
        pthread_cond_broadcast(&L->cv);
        pthread_mutex_unlock(&L->mu);
        for (int g = 0; g < n_gpus; g++) (void)pssbam_engine_sync(eng[g]);
    }
    if (rc == 0 && !g_retired) g_retired = F;   /* (nothing reads the slots any more: every copy was waited for) */
    else loader_close(F);
    free(blocks);
    free(grp);
    return rc;
}
