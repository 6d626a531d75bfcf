/*
 * pss-bam_amd/host/fragkon_main.c -- the `fragkon` command, MI355X edition.
 *
 * Same options, stderr banners and stdout table as the reference front end
 * (/root/reference/fragkon.c:253-386); the k-mer tallies come from the GPU engine's flat
 * 4^k histograms (k <= 15 on the device).  See pss_main.c for what differs underneath.
 */
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "fasta-genome-io.h"
#include "frontend.h"
#include "report.h"
#include "sam-parse.h"

int main(int argc, char *argv[])
{
    frontend_detach_start();   /* (frontend.c: the caller does not wait for the teardown) */
    int klen = 8, min_mq = 0, merged_only = 0, option;
    unsigned long min_read_len = 0, max_read_len = 250000000;
    char *fasta_fn = NULL, *bam_fn = NULL;

    while ((option = getopt(argc, argv, ":F:B:k:l:L:q:m")) != -1) {
        switch (option) {
        case 'F': fasta_fn = strdup(optarg); break;
        case 'B': bam_fn = strdup(optarg); break;
        case 'k': klen = atoi(optarg); break;
        case 'l': min_read_len = strtoul(optarg, NULL, 10); break;
        case 'L': max_read_len = strtoul(optarg, NULL, 10); break;
        case 'q': min_mq = atoi(optarg); break;
        case 'm': merged_only = 1; break;
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

    if (!fasta_fn || !bam_fn) {
        fputs("fragkon: Program for describing kmer-based genomic sequence\n"
              "contexts around the fragmentation points of aligned reads.\n"
              "-F <reference FASTA (required)>\n"
              "-B <input BAM (required)>\n"
              "-k <kmer length (default: 8)>\n"
              "-l <minimum length of read to report (default: 0)>\n"
              "-L <maximum length of read to report (default: 250000000)>\n"
              "-q <map quality filter of read to report (default: 0)>\n"
              "-m <only consider merged reads>\n",
              stderr);
        exit(1);
    }
    if (klen < 1 || klen > PSSBAM_MAX_KLEN) {
        fprintf(stderr, "k-mer length %d is outside the range this build tallies on the GPU (1..%d).\n", klen, PSSBAM_MAX_KLEN);
        exit(1);
    }

    fputs("# Entered command: ", stderr);
    for (int i = 0; i < argc; i++) fprintf(stderr, "%s ", argv[i]);
    fputc('\n', stderr);
    fprintf(stderr, "Input kmer length = %d.\n", klen);
    if (klen & 1)
        fprintf(stderr, "    *** k is odd - counting %d bases outside %d bases inside of alignment.\n", klen / 2,
                klen / 2 + 1);
    fprintf(stderr, "Reading genome sequence from: %s\n", fasta_fn);
    pssbam_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = PSSBAM_ABI_VERSION;
    cfg.tally_mask = PSSBAM_TALLY_KMER;
    cfg.kmer.klen = klen;
    cfg.kmer.min_mq = min_mq;
    cfg.kmer.min_read_len = min_read_len;
    cfg.kmer.max_read_len = max_read_len;
    cfg.kmer.merged_only = merged_only;
    cfg.device = 0;
    cfg.kernel = PSSBAM_KERNEL_AUTO;

    /* HIP start-up, engines and the compressed BAM feed overlap the FASTA load (frontend.c) */
    frontend_warmup_start(&cfg, bam_fn, fasta_fn);
    Genome *genome = init_genome(fasta_fn);
    if (!genome) {
        fprintf(stderr, "Error: Unable to load genome from %s.\n", fasta_fn);
        exit(1);
    }
    fprintf(stderr, "Finished loading genome.\nCounting kmer contexts for: %s\n", bam_fn);

    run_result res;
    frontend_fast_exit = getenv("PSSBAM_CLEAN_EXIT") == NULL;
    if (run_tally(&cfg, genome, bam_fn, env_gpu_count(), &res)) exit(1);
    fragkon_write_table(stdout, fasta_fn, bam_fn, klen, res.k5, res.k3);
    fflush(stdout);
    if (getenv("PSSBAM_STATS")) {
        fprintf(stderr, "[pssbam] records=%llu kmer_ok=%llu kmer_filtered=%llu kmer_fail=%llu gpus=%d\n",
                (unsigned long long)res.stats[PSSBAM_ST_RECORDS], (unsigned long long)res.stats[PSSBAM_ST_KMER_OK],
                (unsigned long long)res.stats[PSSBAM_ST_KMER_FILTERED], (unsigned long long)res.stats[PSSBAM_ST_KMER_FAIL],
                res.n_gpus);
    }
    if (frontend_fast_exit) { /* nothing left to do but to hand the memory back: let the OS */
        fprintf(stderr, "Done.\n");
        front_end_exit(0);
    }
    run_result_free(&res);
    destroy_genome(genome);
    free(fasta_fn);
    free(bam_fn);
    fprintf(stderr, "Done.\n");
    return 0;
}
