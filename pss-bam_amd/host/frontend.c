/*
 * pss-bam_amd/host/frontend.c -- the loop that replaces
 *     popen("samtools view") ; while (fgets) { line2saml ; process_aln }
 * of the reference (pss-bam.c:760-783, fragkon.c:338-363): inflated BAM record batches go
 * to the GPU engines as they are, one engine per device, batches dealt round-robin
 * ("reads shard by record block"); the per-device counter blocks are summed with one RCCL
 * reduce at the end.
 */
#include "frontend.h"

#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "bam_reader.h"
#include "sam_reader.h"

int frontend_fast_exit = 0;

/* start-up work that overlaps the caller's FASTA load: the BAM reader opened early (it starts
 * inflating its first batches at once), HIP runtime + device contexts, page-locking of the
 * reader's slots */
static bam_reader *early_rd = NULL;
static char early_path[4096];
static int early_registered = 0;
static pthread_t warmup_thread;
static int warmup_running = 0;

static void *warmup_main(void *arg)
{
    (void)arg;
    const int n = env_gpu_count(); /* the first HIP call: runtime start-up happens here */
    for (int g = 0; g < n; g++) (void)pssbam_warmup(g); /* failures surface in pssbam_engine_create */
    if (early_rd && !getenv("PSSBAM_NO_PIN")) {
        void *base;
        size_t bytes;
        bam_reader_buffer(early_rd, &base, &bytes);
        early_registered = pssbam_host_register(base, bytes) == 0; /* best effort: pageable works too */
    }
    return NULL;
}

void frontend_warmup_start(const char *aln_path)
{
    if (aln_path && strlen(aln_path) < sizeof early_path && file_is_bam(aln_path) == 1) {
        char err[256];
        early_rd = bam_reader_open(aln_path, 0, 0, err, sizeof err); /* a failure is reported by run_tally's own open */
        if (early_rd) strcpy(early_path, aln_path);
    }
    warmup_running = pthread_create(&warmup_thread, NULL, warmup_main, NULL) == 0;
}

void front_end_exit(int status)
{
    fflush(NULL);
    if (frontend_fast_exit) _exit(status);
    exit(status);
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
    if (warmup_running) { /* HIP is needed from here on; the early reader's slots may be pinned by now */
        pthread_join(warmup_thread, NULL);
        warmup_running = 0;
    }
    if (is_bam && early_rd && strcmp(early_path, aln_path) == 0) {
        rd = early_rd;
        early_rd = NULL;
        if (early_registered) {
            bam_reader_buffer(rd, &buf_base, &buf_bytes);
            registered = 1;
        }
    } else if (is_bam) rd = bam_reader_open(aln_path, 0, 0, err, sizeof err);
    else sd = sam_reader_open(aln_path, 0, err, sizeof err);
    if (!rd && !sd) {
        fprintf(stderr, "Error: Unable to open %s: %s\n", aln_path, err);
        return -1;
    }
    t_open = now_s() - t_mark; t_mark = now_s();
    for (int g = 0; g < n_gpus; g++) {
        pssbam_config c = *cfg;
        const int have = pssbam_device_count();
        c.device = have > 0 ? g % have : g;
        if (pssbam_engine_create(&c, &eng[g]) || pssbam_engine_set_genome(eng[g], genome)) {
            fprintf(stderr, "Error: GPU engine %d: %s\n", g, pssbam_last_error());
            goto done;
        }
    }
    t_engine = now_s() - t_mark; t_mark = now_s();
    if (rd && !registered && !getenv("PSSBAM_NO_PIN")) {
        bam_reader_buffer(rd, &buf_base, &buf_bytes);
        registered = pssbam_host_register(buf_base, buf_bytes) == 0; /* best effort: pageable works too */
    }
    t_register = now_s() - t_mark;

    for (int turn = 0;; turn++) {
        const uint8_t *recs;
        const uint32_t *offs;
        size_t nbytes;
        t_mark = now_s();
        int64_t n = rd ? bam_reader_next(rd, &recs, &offs, &nbytes) : sam_reader_next(sd, &recs, &offs, &nbytes);
        t_read += now_s() - t_mark; t_mark = now_s();
        if (n < 0) {
            fprintf(stderr, "Error: %s: %s\n", aln_path, rd ? bam_reader_error(rd) : sam_reader_error(sd));
            goto done;
        }
        if (n == 0) break;
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
        if (pssbam_engine_submit(eng[turn % n_gpus], recs, nbytes, offs, (uint32_t)n)) {
            fprintf(stderr, "Error: GPU engine: %s\n", pssbam_last_error());
            goto done;
        }
        t_submit += now_s() - t_mark;
    }
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
    if (verbose)
        fprintf(stderr, "[pssbam] phases: open %.3f engine+genome %.3f pin %.3f read(wait) %.3f submit %.3f reduce+finish %.3f s\n",
                t_open, t_engine, t_register, t_read, t_submit, t_finish);
    if (verbose && rd) {
        double ph[4];
        bam_reader_phase_seconds(rd, ph);
        fprintf(stderr, "[pssbam] reader thread: scan+carry %.3f inflate %.3f index %.3f wait-for-slot %.3f s\n", ph[0], ph[1], ph[2], ph[3]);
    }
    res->inflate_s = rd ? bam_reader_inflate_seconds(rd) : 0.0;
    if (sd) res->stats[PSSBAM_ST_PARSE_SKIP] += sam_reader_lines_skipped(sd), res->stats[PSSBAM_ST_RECORDS] += sam_reader_lines_skipped(sd);
    res->n_gpus = n_gpus;
    rc = 0;
done:
    t_mark = now_s();
    if (!(frontend_fast_exit && rc == 0)) {
        for (int g = 0; g < n_gpus; g++)
            if (eng[g]) pssbam_engine_destroy(eng[g]);
        if (registered) pssbam_host_unregister(buf_base);
        bam_reader_close(rd);
        sam_reader_close(sd);
    }
    if (verbose) fprintf(stderr, "[pssbam] teardown %.3f s\n", now_s() - t_mark);
    res->total_s = now_s() - t0;
    if (rc) run_result_free(res);
    return rc;
}
