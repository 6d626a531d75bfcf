/*
 * pss-bam_amd/host/frontend.c -- the loop that replaces
 *     popen("samtools view") ; while (fgets) { line2saml ; process_aln }
 * of the reference (pss-bam.c:760-783, fragkon.c:338-363): inflated BAM record batches go
 * to the GPU engines as they are, one engine per device, batches dealt round-robin
 * ("reads shard by record block"); the per-device counter blocks are summed with one RCCL
 * reduce at the end.
 */
#include "frontend.h"

#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "bam_reader.h"
#include "device_feed.h"
#include "sam_reader.h"

int frontend_fast_exit = 0;

/* Start-up work that overlaps the caller's FASTA load.  The reference is serial by construction -- load the
 * genome, then loop over the alignments (pss-bam.c:751-783) -- but only the TALLY needs reference bases: a helper
 * thread brings the HIP runtime up, creates the engines and runs the whole compressed feed (PCIe, inflate, CRC-32,
 * record index: device_feed.c) while init_genome is still parsing.  run_tally() then posts the Genome; the helper
 * uploads it to every engine at once (pssbam_engine_set_genome_async) and the super-batches inflated ahead of it
 * are tallied.  For inputs the device feed does not take (SAM text, PSSBAM_DEVICE_INFLATE=0) the helper only warms
 * the runtime up and page-locks the host reader's slots, as before. */
static bam_reader *early_rd = NULL;
static char early_path[4096];
static int early_registered = 0, early_light = 0;
static pthread_t warmup_thread;
static int warmup_running = 0;

static struct early_feed {
    int want;                       /* engines + feed on the helper thread */
    pssbam_config cfg;
    char *up, *down, *rg;           /* the strings cfg points at */
    uint64_t fasta_bytes;
    pssbam_engine *eng[64];
    int n_gpus, engines_ok, engines_done, fed, feed_rc, genome_set, failed;
    device_feed_stats dfs;
    char err[600];
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int posted, abandon;            /* posted: the genome upload is enqueued on every engine (by run_tally's thread) */
    double t0, t_hip, t_engines, t_posted, t_upload_dur, t_genome_set, t_genome_set_dur, t_feed_end;
} EF = {.mu = PTHREAD_MUTEX_INITIALIZER, .cv = PTHREAD_COND_INITIALIZER};

/* device_feed.h feed_gate: 1 = genome and references are on every engine.  The upload itself (page-locking the
 * contigs, 3 GB of copies per GPU) is enqueued by run_tally's thread -- 60 ms of host work the feeding thread does
 * not have: its super-batches keep going out meanwhile; here only the reference table follows (a few KB), after
 * which the engine tallies what it inflated ahead. */
static int early_gate(void *ctx, int block)
{
    (void)ctx;
    pthread_mutex_lock(&EF.mu);
    while (!EF.posted && !EF.abandon && block) pthread_cond_wait(&EF.cv, &EF.mu);
    const int abandon = EF.abandon, posted = EF.posted;
    pthread_mutex_unlock(&EF.mu);
    if (abandon) return -1;
    if (!posted) return 0;
    if (!EF.genome_set) {
        const double t = frontend_now_s();
        const bam_header *h = bam_reader_header(early_rd);
        for (int g = 0; g < EF.n_gpus; g++)
            if (pssbam_engine_set_references(EF.eng[g], h->n_ref, (const char *const *)h->ref_name)) {
                snprintf(EF.err, sizeof EF.err, "GPU engine: %s", pssbam_last_error());
                EF.failed = 1;
                return -1;
            }
        EF.genome_set = 1;
        EF.t_genome_set = frontend_now_s() - EF.t0;
        EF.t_genome_set_dur = frontend_now_s() - t;
    }
    return 1;
}

static void early_engines_done(void)
{
    pthread_mutex_lock(&EF.mu);
    EF.engines_done = 1;
    pthread_cond_broadcast(&EF.cv);
    pthread_mutex_unlock(&EF.mu);
}

static int feed_run(int n_gpus);

static void *pin_main(void *arg)
{
    (void)arg;
    device_feed_prefetch_pin();
    return NULL;
}

static void *reserve_main(void *arg)
{
    (void)pssbam_feed_reserve((int)(intptr_t)arg); /* best effort */
    return NULL;
}

typedef struct {
    pssbam_config cfg;
    int32_t n_ref;
    pssbam_engine **out;
    int rc, started;
    char err[400];
} engine_make_job;

static void *engine_make_main(void *arg)
{
    engine_make_job *j = (engine_make_job *)arg;
    if (pssbam_engine_create(&j->cfg, j->out) || pssbam_engine_feed_open(*j->out, j->n_ref, EF.fasta_bytes)) {
        j->rc = 1;
        snprintf(j->err, sizeof j->err, "%s", pssbam_last_error());   /* (the message is this thread's) */
    }
    return NULL;
}

static void early_feed_main(void)
{
    const int n = env_gpu_count(); /* the first HIP call: runtime start-up happens here */
    EF.t_hip = frontend_now_s() - EF.t0;
    const int have = pssbam_device_count();
    if (!getenv("PSSBAM_OVERSUBSCRIBE")) /* the feed's device buffers, beside the engine set-up below (detached: they only allocate; one per GPU) */
        for (int g = 0; g < n && g < (have > 0 ? have : 1); g++) {
            pthread_t th;
            if (pthread_create(&th, NULL, reserve_main, (void *)(intptr_t)g) == 0) pthread_detach(th);
        }
    /* the loader's staging slots (already being filled) are page-locked beside the engine set-up too: 320 MB take ~30 ms,
     * which used to sit between "engines up" and the first block going out */
    pthread_t pin_th;
    const int pin_started = pthread_create(&pin_th, NULL, pin_main, NULL) == 0;
    const bam_header *h = bam_reader_header(early_rd);
    /* one engine per GPU, all created at once: a device's first touch (context, queues, code objects) takes ~0.08 s, and
     * eight of them one after the other would cost more than the whole command does on one GPU */
    engine_make_job job[64];
    pthread_t th[64];
    for (int g = 0; g < n; g++) {
        job[g].cfg = EF.cfg;
        job[g].cfg.device = have > 0 ? g % have : g;
        job[g].n_ref = h->n_ref;
        job[g].out = &EF.eng[g];
        job[g].rc = 0;
        job[g].started = g > 0 && pthread_create(&th[g], NULL, engine_make_main, &job[g]) == 0;
    }
    for (int g = 0; g < n; g++)
        if (!job[g].started) (void)engine_make_main(&job[g]);   /* engine 0 here; the others too if a thread could not be had */
    for (int g = 0; g < n; g++)
        if (job[g].started) pthread_join(th[g], NULL);
    for (int g = 0; g < n; g++) {
        if (job[g].rc) {
            snprintf(EF.err, sizeof EF.err, "GPU engine %d: %s", g, job[g].err);
            EF.failed = 1;
            EF.n_gpus = n;   /* (whatever was created is destroyed with the rest: NULL entries are skipped) */
            early_engines_done();
            if (pin_started) pthread_join(pin_th, NULL);
            return;
        }
        EF.n_gpus = g + 1;
    }
    EF.engines_ok = 1;
    EF.t_engines = frontend_now_s() - EF.t0;
    early_engines_done();
    { /* the light reader has inflated the file's first records on the host: the engines size their staged
       * record prefix from them (no read-back from the device later) */
        const uint8_t *r0;
        const uint32_t *o0;
        size_t nb0;
        int slot0 = -1;
        if (bam_reader_next_hold(early_rd, &r0, &o0, &nb0, &slot0) > 0)
            for (int g = 0; g < n; g++) (void)pssbam_engine_hint_records(EF.eng[g], r0, nb0);
        if (slot0 >= 0) bam_reader_release(early_rd, slot0);
    }
    if (pin_started) pthread_join(pin_th, NULL);
    const feed_gate gate = {early_gate, NULL};
    EF.feed_rc = run_device_feed(EF.eng, n, early_path, bam_reader_header_bytes(early_rd), feed_run(n), getenv("PSSBAM_STATS") != NULL, &EF.dfs, &gate);
    EF.fed = 1;
    EF.t_feed_end = frontend_now_s() - EF.t0;
}

static void *warmup_main(void *arg)
{
    (void)arg;
    if (EF.want) {
        early_feed_main();
        return NULL;
    }
    const int n = env_gpu_count(); /* the first HIP call: runtime start-up happens here */
    for (int g = 0; g < n; g++) (void)pssbam_warmup(g); /* failures surface in pssbam_engine_create */
    if (early_rd && early_light && !getenv("PSSBAM_OVERSUBSCRIBE"))
        for (int g = 0; g < n; g++) (void)pssbam_feed_reserve(g); /* the device feed's buffers, while the FASTA loads (best effort) */
    if (early_rd && early_light) device_feed_prefetch_pin(); /* the loader's staging slots (already being filled) */
    if (early_rd && !early_light && !getenv("PSSBAM_NO_PIN")) {
        void *base;
        size_t bytes;
        bam_reader_buffer(early_rd, &base, &bytes);
        early_registered = pssbam_host_register(base, bytes) == 0; /* best effort: pageable works too */
    }
    return NULL;
}

/* Feed geometry for n GPUs: every GPU is dealt runs of `run` consecutive batches (a sorted BAM then
 * keeps each GPU's reference working set local) and up to n * run copies are in flight
 * at once, so the reader ring needs that many slots plus two for its own fill / index stages. */
static int feed_run(int n_gpus)
{
    const char *v = getenv("PSSBAM_RUN_BATCHES");
    int run = v ? atoi(v) : (n_gpus > 1 ? 2 : 1);
    return run < 1 ? 1 : run > 8 ? 8 : run;
}

static int feed_slots(int n_gpus)
{
    if (getenv("PSSBAM_SLOTS")) return 0; /* the reader reads the variable itself */
    return n_gpus > 1 ? n_gpus * feed_run(n_gpus) + 2 : 3;
}

static char *dup_or_null(const char *p) { return p ? strdup(p) : NULL; }

void frontend_warmup_start(const pssbam_config *cfg, const char *aln_path, const char *fasta_path)
{
    EF.t0 = frontend_now_s();
    if (aln_path && strlen(aln_path) < sizeof early_path && file_is_bam(aln_path) == 1) {
        char err[256];
        /* device feed: the reader is only asked for the BAM header (two threads, small batches) */
        early_light = device_feed_enabled() && !(cfg && cfg->kernel == PSSBAM_KERNEL_SIMPLE);
        early_rd = early_light ? bam_reader_open_slots(aln_path, 2, (size_t)8 << 20, 3, err, sizeof err)
                               : bam_reader_open_slots(aln_path, 0, 0, feed_slots(getenv("PSSBAM_NGPU") ? atoi(getenv("PSSBAM_NGPU")) : 1), err, sizeof err);
        /* (a failure is reported by run_tally's own open) */
        if (early_rd) strcpy(early_path, aln_path);
        if (early_rd && early_light) device_feed_prefetch(aln_path); /* its loader threads read the first windows meanwhile */
        if (early_rd && early_light && cfg && !getenv("PSSBAM_NO_EARLY_FEED")) {
            EF.want = 1;
            EF.cfg = *cfg;
            EF.cfg.pss.up_ctx = EF.up = dup_or_null(cfg->pss.up_ctx);
            EF.cfg.pss.down_ctx = EF.down = dup_or_null(cfg->pss.down_ctx);
            EF.cfg.read_group = EF.rg = dup_or_null(cfg->read_group);
            struct stat sb;
            EF.fasta_bytes = fasta_path && stat(fasta_path, &sb) == 0 ? (uint64_t)sb.st_size : 0;
        }
    }
    warmup_running = pthread_create(&warmup_thread, NULL, warmup_main, NULL) == 0;
    if (!warmup_running) EF.want = 0;
}

static int same_str(const char *a, const char *b) { return (!a && !b) || (a && b && strcmp(a, b) == 0); }

static int same_config(const pssbam_config *a, const pssbam_config *b)
{
    return a->tally_mask == b->tally_mask && a->kernel == b->kernel && same_str(a->read_group, b->read_group) &&
           (!(a->tally_mask & PSSBAM_TALLY_PSS) ||
            (a->pss.region_len == b->pss.region_len && a->pss.min_read_len == b->pss.min_read_len && a->pss.max_read_len == b->pss.max_read_len &&
             a->pss.min_mq == b->pss.min_mq && a->pss.merged_only == b->pss.merged_only && same_str(a->pss.up_ctx, b->pss.up_ctx) &&
             same_str(a->pss.down_ctx, b->pss.down_ctx))) &&
           (!(a->tally_mask & PSSBAM_TALLY_KMER) ||
            (a->kmer.klen == b->kmer.klen && a->kmer.min_mq == b->kmer.min_mq && a->kmer.min_read_len == b->kmer.min_read_len &&
             a->kmer.max_read_len == b->kmer.max_read_len && a->kmer.merged_only == b->kmer.merged_only));
}

/* ---- the exit that does not make the caller wait ---------------------------------------------------------------
 * Once the reports are on disk the command has nothing left to say, but the kernel still needs 0.2-0.3 s to take the
 * process apart (five hardware queues, ~30 GB of device buffers and their page tables, pinned staging slots -- measured
 * with tools/probe/exit_probe.hip and tools/feed_scan.py, profiles/r03_exit_teardown_probe.txt): a third of the whole
 * command on the 200 M-read shape.  So the work runs in a CHILD forked at the very top of main(), before any thread
 * or HIP call exists; the process the caller started only waits for one byte -- the exit status, sent when the
 * tables are written -- and returns it at once, while the child is dismantled in the background.  A child that
 * ends without sending it (a diagnosed exit(1), a signal) is waited for and its status / signal relayed.
 * PSSBAM_DETACH_EXIT=0 keeps everything in one process (the default when LD_PRELOAD names a ROCm profiler: it
 * brings the GPU runtime up before main, and a fork behind that is not safe). */
static int detach_fd = -1;

void frontend_detach_start(void)
{
    const char *v = getenv("PSSBAM_DETACH_EXIT"), *pre = getenv("LD_PRELOAD");
    /* (profilers live in LD_PRELOAD and bring the GPU runtime up before main: no fork behind them) */
    const int tooling = pre && (strstr(pre, "rocprof") || strstr(pre, "roctracer") || strstr(pre, "roctx") || strstr(pre, "omnitrace") ||
                                strstr(pre, "rocsys"));
    if (v ? atoi(v) == 0 : tooling) return;
    int fds[2];
    if (pipe(fds) != 0) return;
    fflush(NULL);
    const pid_t pid = fork();
    if (pid < 0) {
        close(fds[0]);
        close(fds[1]);
        return;
    }
    if (pid == 0) { /* the worker: carries on into main() */
        close(fds[0]);
        (void)fcntl(fds[1], F_SETFD, FD_CLOEXEC);
        detach_fd = fds[1];
        return;
    }
    close(fds[1]);
    unsigned char st = 0;
    ssize_t n;
    do n = read(fds[0], &st, 1);
    while (n < 0 && errno == EINTR);
    if (n == 1) _exit(st); /* the worker's tables are on disk and everything it had to print is printed */
    int ws = 0;
    while (waitpid(pid, &ws, 0) < 0 && errno == EINTR) {}
    if (WIFSIGNALED(ws)) {
        signal(WTERMSIG(ws), SIG_DFL);
        kill(getpid(), WTERMSIG(ws));
    }
    _exit(WIFEXITED(ws) ? WEXITSTATUS(ws) : 1);
}

int frontend_detached(void) { return detach_fd >= 0; }

void front_end_exit(int status)
{
    fflush(NULL);
    if (detach_fd >= 0) { /* releases the caller; what follows is only the teardown */
        const unsigned char st = (unsigned char)status;
        if (write(detach_fd, &st, 1) != 1) {}
        close(detach_fd);
        detach_fd = -1;
        /* a caller that reads our output through pipes waits for their END, not only for the process it started:
         * the address space (and with it the GPU context) goes before the kernel closes a dying process's files */
        close(0);
        close(1);
        close(2);
    }
    if (frontend_fast_exit) _exit(status);
    exit(status);
}

/* seconds since the process was created (exec + dynamic loading included): /proc/self/stat field 22 is the
 * start time in clock ticks after boot (10 ms resolution); -1 if it cannot be read */
double frontend_process_age_s(void)
{
    FILE *f = fopen("/proc/self/stat", "r");
    if (!f) return -1.0;
    char buf[2048];
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    const char *p = strrchr(buf, ')'); /* (the command name may hold spaces) */
    if (!p) return -1.0;
    unsigned long long start = 0;
    int field = 2;
    for (p++; *p && field < 22; p++)
        if (*p == ' ') field++;
    if (field != 22 || sscanf(p, "%llu", &start) != 1) return -1.0;
    struct timespec ts;
    if (clock_gettime(CLOCK_BOOTTIME, &ts) != 0) return -1.0;
    return ts.tv_sec + ts.tv_nsec * 1e-9 - (double)start / (double)sysconf(_SC_CLK_TCK);
}

double frontend_now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}
#define now_s frontend_now_s

int env_gpu_count(void)
{
    const char *v = getenv("PSSBAM_NGPU");
    int want = v ? atoi(v) : 1, have = pssbam_device_count();
    if (want < 1) want = 1;
    if (have < 1) have = 1; /* engine creation reports the real problem */
    /* PSSBAM_OVERSUBSCRIBE: several engines per device (engine g on device g % devices) -- lets
     * the multi-engine path (batch dealing, counter sum) be exercised on a one-GPU machine */
    if (getenv("PSSBAM_OVERSUBSCRIBE")) return want > 64 ? 64 : want;
    return want > have ? have : want;
}

void run_result_free(run_result *res)
{
    free(res->fwd);
    free(res->rev);
    free(res->k5);
    free(res->k3);
    memset(res, 0, sizeof *res);
}

int run_tally(const pssbam_config *cfg, Genome *genome, const char *aln_path, int n_gpus, run_result *res)
{
    pssbam_engine *eng[64] = {0};
    char err[512];
    int rc = -1, registered = 0;
    void *buf_base = NULL;
    size_t buf_bytes = 0;
    const double t0 = now_s();
    double t_mark = t0, t_open = 0, t_engine = 0, t_register = 0, t_read = 0, t_submit = 0, t_finish = 0;
    const int verbose = getenv("PSSBAM_STATS") != NULL;
    memset(res, 0, sizeof *res);
    if (n_gpus < 1) n_gpus = 1;
    if (n_gpus > 64) n_gpus = 64;

    /* BGZF BAM, or SAM text (plain / gzip): what `samtools view FILE` accepts */
    const int is_bam = file_is_bam(aln_path);
    bam_reader *rd = NULL;
    sam_reader *sd = NULL;
    int32_t refs_sent = -1;
    if (is_bam < 0) {
        fprintf(stderr, "Error: Unable to open %s.\n", aln_path);
        return -1;
    }
    device_feed_stats dfs;
    memset(&dfs, 0, sizeof dfs);
    int fed_on_device = 0, adopted = 0, light = 0;
    int device_feed = is_bam && device_feed_enabled() && cfg->kernel != PSSBAM_KERNEL_SIMPLE;
    if (warmup_running) { /* HIP is needed from here on; the helper thread may be feeding already */
        const int mine = EF.want && is_bam && strcmp(early_path, aln_path) == 0 && same_config(cfg, &EF.cfg) && device_feed;
        if (EF.want) {
            int upload_ok = 0;
            if (mine) { /* the engines exist a few dozen ms after the runtime is up: normally long before the FASTA is in */
                pthread_mutex_lock(&EF.mu);
                while (!EF.engines_done) pthread_cond_wait(&EF.cv, &EF.mu);
                pthread_mutex_unlock(&EF.mu);
                upload_ok = EF.engines_ok;
                const double tu = now_s();
                /* every GPU's upload is enqueued before any is waited for: the links run side by side, and the
                 * helper thread keeps feeding compressed blocks to the same engines meanwhile */
                for (int g = 0; g < EF.n_gpus && upload_ok; g++)
                    if (pssbam_engine_set_genome_async(EF.eng[g], genome)) {
                        snprintf(EF.err, sizeof EF.err, "GPU engine %d: %s", g, pssbam_last_error());
                        upload_ok = 0;
                    }
                EF.t_upload_dur = now_s() - tu;
            }
            pthread_mutex_lock(&EF.mu);
            if (mine && upload_ok) EF.posted = 1;
            else EF.abandon = 1;
            EF.t_posted = now_s() - EF.t0;
            pthread_cond_broadcast(&EF.cv);
            pthread_mutex_unlock(&EF.mu);
        }
        pthread_join(warmup_thread, NULL);
        warmup_running = 0;
        if (EF.want && mine && EF.engines_ok && !EF.failed && EF.posted) {
            adopted = 1;
            n_gpus = EF.n_gpus;
            for (int g = 0; g < n_gpus; g++) eng[g] = EF.eng[g];
        } else if (EF.want) { /* not this run's file / options, or the helper could not set up: start over below */
            for (int g = 0; g < EF.n_gpus; g++)
                if (EF.eng[g]) pssbam_engine_destroy(EF.eng[g]);
            if (EF.failed && verbose) fprintf(stderr, "[pssbam] early feed not usable (%s): regular start\n", EF.err);
        }
        EF.want = 0;
    }
    /* BAM: inflate on the GPU (device_feed.c) unless switched off or the cross-check kernel is forced;
     * the host reader then only parses the header -- and takes over if the file's records cross BGZF
     * blocks */
    if (is_bam && early_rd && strcmp(early_path, aln_path) == 0) {
        rd = early_rd;
        early_rd = NULL;
        light = early_light;
        if (early_registered) {
            bam_reader_buffer(rd, &buf_base, &buf_bytes);
            registered = 1;
        }
    } else if (is_bam && device_feed) {
        rd = bam_reader_open_slots(aln_path, 2, (size_t)8 << 20, 3, err, sizeof err);
        light = 1;
    } else if (is_bam) rd = bam_reader_open_slots(aln_path, 0, 0, feed_slots(n_gpus), err, sizeof err);
    else sd = sam_reader_open(aln_path, 0, err, sizeof err);
    if (is_bam && !rd && light) { /* e.g. a header larger than the light reader's batches: the full reader decides */
        light = 0;
        device_feed = 0;
        rd = bam_reader_open_slots(aln_path, 0, 0, feed_slots(n_gpus), err, sizeof err);
    }
    if (rd && light && !device_feed) { /* opened early for the device feed, which is not wanted after all */
        bam_reader_close(rd);
        light = 0;
        rd = bam_reader_open_slots(aln_path, 0, 0, feed_slots(n_gpus), err, sizeof err);
    }
    if (!rd && !sd) {
        fprintf(stderr, "Error: Unable to open %s: %s\n", aln_path, err);
        goto done;
    }
    t_open = now_s() - t_mark; t_mark = now_s();
    if (adopted) {
        /* the helper created the engines, fed the file and (usually) set the genome when it was posted */
        if (EF.feed_rc || EF.failed) {
            if (EF.err[0]) fprintf(stderr, "Error: %s\n", EF.err);
            goto done;
        }
        if (!EF.genome_set) { /* the feed ended before it looked at the gate (a file it does not take) */
            const bam_header *h = bam_reader_header(rd);
            for (int g = 0; g < n_gpus; g++)
                if (pssbam_engine_set_references(eng[g], h->n_ref, (const char *const *)h->ref_name)) {
                    fprintf(stderr, "Error: GPU engine %d: %s\n", g, pssbam_last_error());
                    goto done;
                }
        }
        refs_sent = bam_reader_header(rd)->n_ref;
        dfs = EF.dfs;
        if (verbose)
            fprintf(stderr, "[pssbam] early feed (helper thread, seconds after start-up began): HIP runtime up %.3f, engines %.3f, genome upload "
                            "enqueued %.3f (took this thread %.3f), references set + put-off tallies launched %.3f (took %.3f), feed drained %.3f\n",
                    EF.t_hip, EF.t_engines, EF.t_posted, EF.t_upload_dur, EF.t_genome_set, EF.t_genome_set_dur, EF.t_feed_end);
    } else {
        for (int g = 0; g < n_gpus; g++) {
            pssbam_config c = *cfg;
            const int have = pssbam_device_count();
            c.device = have > 0 ? g % have : g;
            if (pssbam_engine_create(&c, &eng[g])) {
                fprintf(stderr, "Error: GPU engine %d: %s\n", g, pssbam_last_error());
                goto done;
            }
        }
        /* every GPU's upload is enqueued before the first one is waited for */
        for (int g = 0; g < n_gpus; g++)
            if (pssbam_engine_set_genome_async(eng[g], genome)) {
                fprintf(stderr, "Error: GPU engine %d: %s\n", g, pssbam_last_error());
                goto done;
            }
    }
    t_engine = now_s() - t_mark; t_mark = now_s();
    if (device_feed && rd && !adopted) {
        const bam_header *h = bam_reader_header(rd);
        for (int g = 0; g < n_gpus; g++)
            if (pssbam_engine_set_references(eng[g], h->n_ref, (const char *const *)h->ref_name)) {
                fprintf(stderr, "Error: GPU engine %d: %s\n", g, pssbam_last_error());
                goto done;
            }
        refs_sent = h->n_ref;
        { /* the light reader has inflated the file's first records on the host: let the engines size their
           * staged record prefix from them (no read-back from the device later) */
            const uint8_t *r0;
            const uint32_t *o0;
            size_t nb0;
            int slot0 = -1;
            if (bam_reader_next_hold(rd, &r0, &o0, &nb0, &slot0) > 0)
                for (int g = 0; g < n_gpus; g++) (void)pssbam_engine_hint_records(eng[g], r0, nb0);
            if (slot0 >= 0) bam_reader_release(rd, slot0);
        }
        if (run_device_feed(eng, n_gpus, aln_path, bam_reader_header_bytes(rd), feed_run(n_gpus), verbose, &dfs, NULL)) goto done;
    }
    if (device_feed && rd) {
        if (dfs.fallback) {
            if (verbose) fprintf(stderr, "[pssbam] device feed not usable for this file: falling back to the host reader\n");
            for (int g = 0; g < n_gpus; g++)
                if (pssbam_engine_reset(eng[g])) { fprintf(stderr, "Error: GPU engine %d: %s\n", g, pssbam_last_error()); goto done; }
            bam_reader_close(rd);
            rd = bam_reader_open_slots(aln_path, 0, 0, feed_slots(n_gpus), err, sizeof err);
            light = 0;
            if (!rd) { fprintf(stderr, "Error: Unable to open %s: %s\n", aln_path, err); goto done; }
        } else fed_on_device = 1;
        t_submit = now_s() - t_mark; t_mark = now_s();
    }
    if (rd && !fed_on_device && !registered && !getenv("PSSBAM_NO_PIN")) {
        bam_reader_buffer(rd, &buf_base, &buf_bytes);
        registered = pssbam_host_register(buf_base, buf_bytes) == 0; /* best effort: pageable works too */
    }
    t_register = now_s() - t_mark;

    /* Batches in flight: (reader slot, engine, ticket).  A BAM batch is handed to its engine without
     * waiting for the copy (pssbam_engine_submit_async) and its slot goes back to the reader when
     * the copy has completed -- so the PCIe links of different GPUs carry copies at the same time
     * and one host thread is enough to deal them.  SAM text keeps the blocking submit. */
    struct { int slot, g; uint64_t ticket; double t_issue, t_done; } fifo[64];
    int fifo_head = 0, fifo_len = 0;
    const int run = feed_run(n_gpus);
    const int max_inflight = rd ? (bam_reader_slots(rd) - 2 < 1 ? 1 : bam_reader_slots(rd) - 2 > 62 ? 62 : bam_reader_slots(rd) - 2) : 0;
    double overlap_s = 0.0, win_lo[64], win_hi[64]; /* last copy window seen per engine */
    for (int g = 0; g < 64; g++) win_lo[g] = win_hi[g] = -1.0;
    uint64_t n_batches = 0;
#define RETIRE_OLDEST()                                                                                        \
    do {                                                                                                       \
        const int k = fifo_head;                                                                               \
        if (pssbam_engine_wait_copied(eng[fifo[k].g], fifo[k].ticket)) {                                       \
            fprintf(stderr, "Error: GPU engine: %s\n", pssbam_last_error());                                   \
            goto done;                                                                                         \
        }                                                                                                      \
        if (fifo[k].t_done < 0) fifo[k].t_done = now_s();                                                      \
        bam_reader_release(rd, fifo[k].slot);                                                                  \
        if (verbose && n_gpus > 1) {                                                                           \
            /* time this copy window shares with the latest window of every other engine */                    \
            for (int o = 0; o < n_gpus; o++) {                                                                 \
                if (o == fifo[k].g || win_hi[o] < 0) continue;                                                 \
                const double lo = fifo[k].t_issue > win_lo[o] ? fifo[k].t_issue : win_lo[o];                   \
                const double hi = fifo[k].t_done < win_hi[o] ? fifo[k].t_done : win_hi[o];                     \
                if (hi > lo) overlap_s += hi - lo;                                                             \
            }                                                                                                  \
            if (n_batches <= 24)                                                                               \
                fprintf(stderr, "[pssbam] copy window: engine %d  %.4f .. %.4f s\n", fifo[k].g, fifo[k].t_issue - t0, \
                        fifo[k].t_done - t0);                                                                  \
            win_lo[fifo[k].g] = fifo[k].t_issue;                                                               \
            win_hi[fifo[k].g] = fifo[k].t_done;                                                                \
        }                                                                                                      \
        fifo_head = (fifo_head + 1) % 64;                                                                      \
        fifo_len--;                                                                                            \
    } while (0)

    for (int turn = 0; !fed_on_device; turn++) {
        const uint8_t *recs;
        const uint32_t *offs;
        size_t nbytes;
        int slot = -1;
        t_mark = now_s();
        int64_t n = rd ? bam_reader_next_hold(rd, &recs, &offs, &nbytes, &slot) : sam_reader_next(sd, &recs, &offs, &nbytes);
        t_read += now_s() - t_mark; t_mark = now_s();
        if (n < 0) {
            fprintf(stderr, "Error: %s: %s\n", aln_path, rd ? bam_reader_error(rd) : sam_reader_error(sd));
            goto done;
        }
        if (n == 0) break;
        n_batches++;
        /* reference names: fixed by the BAM header; for SAM text the table grows as new RNAMEs
         * show up, so it is (re)sent whenever it changed */
        const int32_t n_ref = rd ? bam_reader_header(rd)->n_ref : sam_reader_n_ref(sd);
        if (n_ref != refs_sent) {
            const char *const *names = rd ? (const char *const *)bam_reader_header(rd)->ref_name : sam_reader_ref_names(sd);
            for (int g = 0; g < n_gpus; g++)
                if (pssbam_engine_set_references(eng[g], n_ref, names)) {
                    fprintf(stderr, "Error: GPU engine %d: %s\n", g, pssbam_last_error());
                    goto done;
                }
            refs_sent = n_ref;
        }
        const int g = (turn / run) % n_gpus;
        if (rd) {
            uint64_t ticket = 0;
            const double t_issue = now_s();
            if (pssbam_engine_submit_async(eng[g], recs, nbytes, offs, (uint32_t)n, &ticket)) {
                fprintf(stderr, "Error: GPU engine: %s\n", pssbam_last_error());
                goto done;
            }
            const int k = (fifo_head + fifo_len) % 64;
            fifo[k].slot = slot; fifo[k].g = g; fifo[k].ticket = ticket; fifo[k].t_issue = t_issue; fifo[k].t_done = -1.0;
            fifo_len++;
            if (verbose && n_gpus > 1) /* note completions as they are seen (the windows printed with PSSBAM_STATS) */
                for (int i = 0; i < fifo_len; i++) {
                    const int q = (fifo_head + i) % 64;
                    if (fifo[q].t_done < 0 && pssbam_engine_copy_done(eng[fifo[q].g], fifo[q].ticket) == 1) fifo[q].t_done = now_s();
                }
            while (fifo_len > max_inflight) RETIRE_OLDEST();
        } else if (pssbam_engine_submit(eng[g], recs, nbytes, offs, (uint32_t)n)) {
            fprintf(stderr, "Error: GPU engine: %s\n", pssbam_last_error());
            goto done;
        }
        t_submit += now_s() - t_mark;
    }
    while (fifo_len > 0) RETIRE_OLDEST();
#undef RETIRE_OLDEST
    if (verbose && n_gpus > 1 && !fed_on_device)
        fprintf(stderr, "[pssbam] feed: %llu batches in runs of %d over %d engines, up to %d copies in flight; copy windows of "
                        "different engines overlapped for %.3f s\n", (unsigned long long)n_batches, run, n_gpus, max_inflight, overlap_s);
    t_mark = now_s();
    if (refs_sent < 0) { /* no alignment at all: the engines still need a (possibly empty) table to finish */
        const int32_t n_ref = rd ? bam_reader_header(rd)->n_ref : sam_reader_n_ref(sd);
        const char *const *names = rd ? (const char *const *)bam_reader_header(rd)->ref_name : sam_reader_ref_names(sd);
        for (int g = 0; g < n_gpus; g++) (void)pssbam_engine_set_references(eng[g], n_ref, names);
    }
    if (pssbam_reduce_counters(eng, n_gpus, 0)) {
        fprintf(stderr, "Error: counter reduce: %s\n", pssbam_last_error());
        goto done;
    }
    if (cfg->tally_mask & PSSBAM_TALLY_PSS) {
        size_t cells = (size_t)(cfg->pss.region_len + 2) * 16;
        res->fwd = (unsigned long *)calloc(cells, sizeof(unsigned long));
        res->rev = (unsigned long *)calloc(cells, sizeof(unsigned long));
    }
    if (cfg->tally_mask & PSSBAM_TALLY_KMER) {
        size_t bins = (size_t)1 << (2 * cfg->kmer.klen);
        res->k5 = (uint64_t *)calloc(bins, sizeof(uint64_t));
        res->k3 = (uint64_t *)calloc(bins, sizeof(uint64_t));
    }
    if (pssbam_engine_finish(eng[0], res->fwd, res->rev, res->k5, res->k3, res->stats)) {
        fprintf(stderr, "Error: GPU engine: %s\n", pssbam_last_error());
        goto done;
    }
    t_finish = now_s() - t_mark;
    if (verbose) {
        double h2d = 0, ker = 0;
        uint64_t bytes = 0, launches = 0;
        for (int g = 0; g < n_gpus; g++) {
            double a = 0, c = 0;
            uint64_t b = 0, d = 0;
            if (pssbam_engine_phase_times(eng[g], &a, &b, &c, &d) == 0) { h2d += a; bytes += b; ker += c; launches += d; }
        }
        fprintf(stderr, "[pssbam] device: h2d %.3f kernel %.3f s\n", h2d * 1e-3, ker * 1e-3);
        /* what the GPU(s) spent in kernels, all engines summed (genome encode + 4-bit pack: ~6 ms per GPU, not timed) */
        fprintf(stderr, "[pssbam] gpu busy: inflate+crc+index %.3f tally %.3f s over %d engine(s)\n", fed_on_device ? dfs.inflate_ms * 1e-3 : 0.0, ker * 1e-3, n_gpus);
        fprintf(stderr, "[pssbam] device detail: %.2f GB copied (%.1f GB/s while copying), %llu tally launches\n", bytes * 1e-9,
                h2d > 0 ? bytes * 1e-6 / h2d : 0.0, (unsigned long long)launches);
    }
    if (verbose)
        fprintf(stderr, "[pssbam] phases: open %.3f engine+genome %.3f pin %.3f read(wait) %.3f submit %.3f reduce+finish %.3f s\n",
                t_open, t_engine, t_register, t_read, t_submit, t_finish);
    if (verbose && rd && !fed_on_device) {
        double ph[4];
        bam_reader_phase_seconds(rd, ph);
        fprintf(stderr, "[pssbam] reader thread: scan+carry %.3f inflate %.3f index %.3f wait-for-slot %.3f s\n", ph[0], ph[1], ph[2], ph[3]);
    }
    res->inflate_s = fed_on_device ? dfs.inflate_ms * 1e-3 : rd ? bam_reader_inflate_seconds(rd) : 0.0;
    if (sd) res->stats[PSSBAM_ST_PARSE_SKIP] += sam_reader_lines_skipped(sd), res->stats[PSSBAM_ST_RECORDS] += sam_reader_lines_skipped(sd);
    res->n_gpus = n_gpus;
    rc = 0;
done:
    t_mark = now_s();
    if (!(frontend_fast_exit && rc == 0)) {
        device_feed_prefetch_cancel(); /* a loader opened ahead of time that no feed took over; the staging slots of one that ran */
        for (int g = 0; g < n_gpus; g++)
            if (eng[g]) pssbam_engine_destroy(eng[g]);
        for (int g = 0; g < n_gpus; g++) (void)pssbam_feed_release(g); /* reserved feed buffers no engine took */
        if (registered) pssbam_host_unregister(buf_base);
        bam_reader_close(rd);
        sam_reader_close(sd);
    }
    if (verbose) fprintf(stderr, "[pssbam] teardown %.3f s\n", now_s() - t_mark);
    res->total_s = now_s() - t0;
    if (rc) run_result_free(res);
    return rc;
}
