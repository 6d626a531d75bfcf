/*
 * pss-bam_amd/host/gkc_main.c -- the `genome-kmer-count` command, MI355X edition: same options
 * and stdout as the reference (/root/reference/genome-kmer-count.c:23-66); the 4^k histogram of
 * all k-mer starts is computed on the GPU from the uploaded genome (k <= 15).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <unistd.h>

#include "fasta-genome-io.h"
#include "pssbam_hip.h"
#include "report.h"

#define K_DEF (4)

static void help(void)
{
    printf("genome-kmer-count -f <fasta genome file>\n");
    printf("                  -k <kmer size; default = %u>\n", K_DEF);
    printf("This program reports the number of observed number\n");
    printf("of all possible kmers of the given length in the\n");
    printf("input genome.\n");
    exit(0);
}

int main(int argc, char *argv[])
{
    int ich, k = K_DEF;
    char fa_in[MAX_FN_LEN + 1] = {'\0'};
    while ((ich = getopt(argc, argv, "f:k:")) != -1) {
        switch (ich) {
        case 'f': strncpy(fa_in, optarg, MAX_FN_LEN); break;
        case 'k': k = atoi(optarg); break;
        default: help();
        }
    }
    if (strlen(fa_in) == 0) help();
    if (k < 1 || k > PSSBAM_MAX_KLEN) {
        fprintf(stderr, "k-mer length %d is outside the range this build counts on the GPU (1..%d).\n", k, PSSBAM_MAX_KLEN);
        return 1;
    }
    Genome *genome = init_genome(fa_in);
    if (!genome) return 1;
    printf("Parsed input genome. Found %lu sequences.\n", genome->n_seqs);

    pssbam_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = PSSBAM_ABI_VERSION;
    cfg.tally_mask = PSSBAM_TALLY_KMER; /* an engine needs a tally; only its genome is used here */
    cfg.kmer.klen = k;
    cfg.kmer.max_read_len = 250000000;
    cfg.device = 0;
    pssbam_engine *eng = NULL;
    const size_t bins = (size_t)1 << (2 * k);
    uint64_t *counts = (uint64_t *)calloc(bins, sizeof(uint64_t));
    if (pssbam_engine_create(&cfg, &eng) || pssbam_engine_set_genome(eng, genome) ||
        pssbam_engine_genome_kmer_count(eng, k, counts)) {
        fprintf(stderr, "Error: GPU engine: %s\n", pssbam_last_error());
        return 1;
    }
    gkc_write_table(stdout, k, counts);
    pssbam_engine_destroy(eng);
    destroy_genome(genome);
    free(counts);
    return 0;
}
