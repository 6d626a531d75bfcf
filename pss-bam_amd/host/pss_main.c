/*
 * pss-bam_amd/host/pss_main.c -- the `pss-bam` command, MI355X edition.
 *
 * Same command line, same stderr banners, same two output files byte for byte as the
 * reference front end (/root/reference/pss-bam.c:650-805).  What changed underneath:
 *   - the BAM is read natively (BGZF inflate on host threads) instead of through a
 *     `samtools view` child and a text parser;
 *   - filtering and tallying of every alignment happen on the GPU(s) (include/pssbam_hip.h);
 *   - PSSBAM_NGPU=<n> in the environment spreads record batches over n GPUs of the node.
 * Differences on purpose: missing -F/-B/-o are detected reliably (the reference tests
 * uninitialised pointers), an unreadable FASTA/BAM is a diagnosed exit(1) instead of a
 * crash, and PSSBAM_STATS=1 prints the per-status record tallies to stderr.
 */
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "fasta-genome-io.h"
#include "frontend.h"
#include "report.h"

int main(int argc, char *argv[])
{
    const double age_main = frontend_process_age_s();
    frontend_detach_start();   /* the caller gets its prompt back when the reports are written, not when 30 GB of device buffers are gone */
    const double t_main = frontend_now_s();
    int region_len = 15, min_mq = 0, merged_only = 0, option;
    unsigned long min_read_len = 0, max_read_len = 250000000;
    const char *up_ctx = "ACGT", *down_ctx = "ACGT";
    char *fasta_fn = NULL, *bam_fn = NULL, *out_prefix = NULL, *read_group = NULL;

    while ((option = getopt(argc, argv, ":F:B:o:R:r:l:L:q:U:D:m")) != -1) {
        switch (option) {
        case 'F': fasta_fn = strdup(optarg); break;
        case 'B': bam_fn = strdup(optarg); break;
        case 'o': out_prefix = strdup(optarg); break;
        case 'r': region_len = atoi(optarg); break;
        case 'l': min_read_len = strtoul(optarg, NULL, 10); break;
        case 'L': max_read_len = strtoul(optarg, NULL, 10); break;
        case 'q': min_mq = atoi(optarg); break;
        case 'U': up_ctx = optarg; break;
        case 'D': down_ctx = optarg; break;
        case 'm': merged_only = 1; break;
        case 'R': read_group = strdup(optarg); break;
        case ':':
            fprintf(stderr, "Please enter required argument for option -%c.\n", optopt);
            exit(0);
        case '?':
            if (isprint(optopt)) fprintf(stderr, "Unknown option -%c.\n", optopt);
            else fprintf(stderr, "Unknown option character \\x%x.\n", optopt);
            break;
        default:
            fprintf(stderr, "Error parsing command-line options.\n");
            exit(0);
        }
    }
    for (int i = optind; i < argc; i++) fprintf(stderr, "Non-option argument %s\n", argv[i]);

    if (!fasta_fn || !bam_fn || !out_prefix) {
        fputs("pss-bam v1.2.1: Program for describing base context and counting\n"
              "the number of matches/mismatches in aligned reads to a genome.\n"
              "-F <reference FASTA (required)>\n"
              "-B <input BAM (required)>\n"
              "-o <output filename prefix (required)>\n"
              "-r <length in basepairs into the interior of alignments to report on (default: 15)>\n"
              "-l <minimum length of read to report (default: 0)>\n"
              "-L <maximum length of read to report (default: 250000000)>\n"
              "-q <map quality filter of read to report (default: 0)>\n"
              "-R <read group name to restrict analysis to (default: all reads)>\n"
              "-U <upstream context base filter; first base before alignment must be one of these (default: ACGT)>\n"
              "-D <downstream context base filter; first base before alignment must be one of these (default: ACGT)>\n"
              "-m <only consider merged reads>\n",
              stderr);
        exit(1);
    }
    if (region_len < 0) {
        fprintf(stderr, "-r must not be negative.\n");
        exit(1);
    }

    /* "Full command" banner: four shapes, as the reference prints them (pss-bam.c:728-749) */
    fprintf(stderr, "Full command: %s -F %s -B %s -o %s -r %d -l %lu -L %lu -q %d", argv[0], fasta_fn, bam_fn,
            out_prefix, region_len, min_read_len, max_read_len, min_mq);
    if (read_group) fprintf(stderr, " -R %s", read_group);
    fprintf(stderr, " -U %s -D %s%s\n", up_ctx, down_ctx, merged_only ? " -m" : "");

    pssbam_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = PSSBAM_ABI_VERSION;
    cfg.tally_mask = PSSBAM_TALLY_PSS;
    cfg.pss.region_len = region_len;
    cfg.pss.min_read_len = min_read_len;
    cfg.pss.max_read_len = max_read_len;
    cfg.pss.min_mq = min_mq;
    cfg.pss.up_ctx = up_ctx;
    cfg.pss.down_ctx = down_ctx;
    cfg.pss.merged_only = merged_only;
    cfg.read_group = read_group;
    cfg.device = 0;
    cfg.kernel = PSSBAM_KERNEL_AUTO;

    fprintf(stderr, "Reading genome sequence from:\n%s\n", fasta_fn);
    /* HIP start-up, engines and the compressed BAM feed (PCIe, inflate, CRC, record index) overlap the FASTA
     * load; only the tally launches wait for the genome (frontend.c) */
    frontend_warmup_start(&cfg, bam_fn, fasta_fn);
    const double t_fa = frontend_now_s();
    Genome *genome = init_genome(fasta_fn);
    if (getenv("PSSBAM_STATS")) fprintf(stderr, "[pssbam] genome: fasta load %.3f s\n", frontend_now_s() - t_fa);
    if (!genome) {
        fprintf(stderr, "Error: Unable to load genome from %s.\n", fasta_fn);
        exit(1);
    }
    fprintf(stderr, "Finished loading genome.\nCounting matches/mismatches from:\n%s\n", bam_fn);

    run_result res;
    frontend_fast_exit = getenv("PSSBAM_CLEAN_EXIT") == NULL;
    if (run_tally(&cfg, genome, bam_fn, env_gpu_count(), &res)) exit(1);

    double *fwd_rates = (double *)calloc((size_t)(region_len ? region_len : 1) * 12, sizeof(double));
    double *rev_rates = (double *)calloc((size_t)(region_len ? region_len : 1) * 12, sizeof(double));
    pss_sub_rates(region_len, res.fwd, fwd_rates);
    pss_sub_rates(region_len, res.rev, rev_rates);
    pss_write_counts(fasta_fn, bam_fn, out_prefix, region_len, res.fwd, res.rev);
    pss_write_rates(fasta_fn, bam_fn, out_prefix, region_len, fwd_rates, rev_rates);

    if (getenv("PSSBAM_STATS")) {
        static const char *nm[] = {"records", "rg_dropped", "parse_skip", "no_contig", "pss_ok", "pss_filtered"};
        for (int i = 0; i < 6; i++) fprintf(stderr, "[pssbam] %s=%llu\n", nm[i], (unsigned long long)res.stats[i]);
        fprintf(stderr, "[pssbam] slow_path=%llu%s\n", (unsigned long long)res.stats[PSSBAM_ST_SLOW_PATH],
                res.stats[PSSBAM_ST_SLOW_PATH] * 100 > res.stats[PSSBAM_ST_RECORDS]
                    ? "  (more than 1 % of the records were longer than the staged prefix and took the one-lane path: slower, same tables)" : "");
        fprintf(stderr, "[pssbam] gpus=%d inflate_s=%.3f total_s=%.3f\n", res.n_gpus, res.inflate_s, res.total_s);
        fprintf(stderr, "[pssbam] main() to reports written: %.3f s\n", frontend_now_s() - t_main);
        fprintf(stderr, "[pssbam] process creation to main(): %.2f s (exec + dynamic loading); main() to here %.3f s; exit: %s\n", age_main,
                frontend_now_s() - t_main, frontend_detached() ? "the caller is released now, this worker is torn down behind it" : "one process, teardown in the foreground");
    }
    if (frontend_fast_exit) { /* nothing left to do but to hand the memory back: let the OS */
        fprintf(stderr, "Done.\n");
        front_end_exit(0);
    }
    free(fwd_rates);
    free(rev_rates);
    run_result_free(&res);
    destroy_genome(genome);
    free(fasta_fn);
    free(bam_fn);
    free(out_prefix);
    free(read_group);
    fprintf(stderr, "Done.\n");
    return 0;
}
