/* pss-bam_amd/host/frontend.h -- what the two command-line front ends share: open the
 * alignment input, drive one engine per GPU over its record batches, gather the tables. */
#ifndef PSSBAM_FRONTEND_H
#define PSSBAM_FRONTEND_H

#include <stdint.h>

#include "fasta-genome-io.h"
#include "pssbam_hip.h"

typedef struct run_result {
    unsigned long *fwd, *rev; /* (region_len+2)*16 each, or NULL */
    uint64_t *k5, *k3;        /* 4^klen each, or NULL            */
    uint64_t stats[PSSBAM_ST_N];
    double inflate_s, total_s;
    int n_gpus;
} run_result;

/* Streams every alignment of `aln_path` (BGZF BAM, or SAM text plain/gzip) through engines built from `cfg` on
 * n_gpus devices (batches dealt round-robin), sums the counter blocks onto device 0 with
 * RCCL when n_gpus > 1, and returns the tables in *res (caller frees with run_result_free).
 * Returns 0, or -1 after printing a diagnostic to stderr. */
int run_tally(const pssbam_config *cfg, Genome *genome, const char *aln_path, int n_gpus, run_result *res);

/* The command-line tools end right after their report is written: with this set (they set it
 * unless $PSSBAM_CLEAN_EXIT is), run_tally() leaves engines, pinned slots and the mapped input
 * to process exit instead of releasing ~4 GB piece by piece (0.12 s of a 0.7 s command), and
 * front_end_exit() ends the process without running destructors. */
/* Start-up work that overlaps the caller's FASTA load (returns at once).  If aln_path is a BGZF BAM the
 * device feed takes, a helper thread brings the HIP runtime up, creates the engines from `cfg` and feeds the
 * compressed file to them right away -- inflate, CRC-32 and record index need no genome; run_tally() (same
 * cfg, same path) posts the Genome when the caller has it and collects the result.  Otherwise (SAM text, host
 * inflate, cfg == NULL) the helper only warms the runtime up and page-locks the host reader's slots.
 * fasta_path (may be NULL) only sizes the device memory left alone for the genome.  aln_path may be NULL. */
void frontend_warmup_start(const pssbam_config *cfg, const char *aln_path, const char *fasta_path);

/* First statement of a front end's main(): forks the worker that runs the rest of main(); the process the caller
 * started returns the worker's exit status as soon as front_end_exit() sends it, without waiting for the worker's
 * teardown (frontend.c; PSSBAM_DETACH_EXIT=0 or a non-empty LD_PRELOAD: no fork). */
void frontend_detach_start(void);
int frontend_detached(void); /* 1 in a worker whose caller will be released by front_end_exit() */

extern int frontend_fast_exit;
void front_end_exit(int status);
void run_result_free(run_result *res);
double frontend_now_s(void); /* CLOCK_MONOTONIC seconds */
double frontend_process_age_s(void); /* seconds since the process was created (10 ms resolution), -1 if unknown */
int env_gpu_count(void); /* PSSBAM_NGPU, default 1, clamped to the devices present */
#endif
