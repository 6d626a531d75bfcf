/*
 * pss-bam_amd/host/report.c -- the text the two tools emit.  These files are the
 * byte-for-byte parity surface; formats follow
 * /root/reference/pss-bam.c:504-633 and fragkon.c:231-249,:367-368.
 */
#include "report.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

#define PSS_VERSION "1.2.1"
#define FN_BUF 2047 /* the reference formats the output name into a MAX_FN_LEN buffer */

/* substitution rates per interior position: count / (column total of that reference base).
 * Column totals are summed as integers first, then divided in double, in the reference's
 * order, so "%.5e" prints identically.  A position whose A, C, G or T column is empty keeps
 * twelve zeros (pss-bam.c:512-514). */
void pss_sub_rates(int region_len, const unsigned long *counts, double *rates)
{
    /* cells of the twelve off-diagonal (read base, reference base) pairs; cell & 3 = reference base */
    static const unsigned char off_diag[12] = {1, 2, 3, 4, 6, 7, 8, 9, 11, 12, 13, 14};
    for (int pos = 0; pos < region_len; pos++) {
        const unsigned long *row = counts + (size_t)(pos + 2) * 16;
        double *out = rates + (size_t)pos * 12;
        double col[4];
        int empty = 0;
        for (int ref = 0; ref < 4; ref++) {
            unsigned long sum = row[ref] + row[4 + ref] + row[8 + ref] + row[12 + ref];
            col[ref] = sum;
            empty |= (sum == 0);
        }
        for (int j = 0; j < 12; j++) out[j] = empty ? 0.0 : row[off_diag[j]] / col[off_diag[j] & 3];
    }
}

static void count_row(FILE *fp, int label, const unsigned long *row)
{
    fprintf(fp, "%d\t", label);
    for (int j = 0; j < 16; j++) fprintf(fp, "%lu\t", row[j]); /* every row ends in a TAB */
    fputc('\n', fp);
}

int pss_write_counts(const char *fasta_fn, const char *bam_fn, const char *out_prefix, int region_len,
                     const unsigned long *fwd, const unsigned long *rev)
{
    char fn[FN_BUF + 1];
    FILE *fp;
    snprintf(fn, sizeof fn, "%s.pss.counts.txt", out_prefix);
    fp = fopen(fn, "w");
    if (!fp) {
        fprintf(stderr, "ERROR: Cannot write to file %s\n.", fn);
        return 1;
    }
    /* the counts header carries a literal "v1.2.1:" (with the colon) */
    fprintf(fp, "### pss-bam.c v1.2.1:\n### FASTA: %s\n### BAM: %s\n### OUT: %s\n", fasta_fn, bam_fn, fn);
    fputs("### Format of table:\n", fp);
    fputs("### Counts of how often a read base and genome base were seen at\n", fp);
    fputs("### each position in the aligned reads.\n", fp);
    fputs("### First base is what was seen in the read.\n", fp);
    fputs("### Second base is what was in the genome at that position.\n", fp);
    fputs("### POS AA AC AG AT CA CC CG CT GA GC GG GT TA TC TG TT\n", fp);
    fputs("### Forward read substitution counts and base context\n", fp);
    for (int r = 0; r < region_len + 2; r++) count_row(fp, r - 2, fwd + (size_t)r * 16);
    /* two blank lines separate the gnuplot data sets */
    fputs("\n\n### Reverse read substitution counts and base context\n", fp);
    for (int pos = region_len - 1; pos >= 0; pos--) count_row(fp, pos, rev + (size_t)(pos + 2) * 16);
    count_row(fp, 1, rev + 16); /* first base past the alignment  */
    count_row(fp, 2, rev);      /* second base past the alignment */
    fclose(fp);
    return 0;
}

static void rate_row(FILE *fp, int label, const double *row)
{
    fprintf(fp, "%d\t", label);
    for (int j = 0; j < 12; j++) fprintf(fp, "%.5e\t", row[j]);
    fputc('\n', fp);
}

int pss_write_rates(const char *fasta_fn, const char *bam_fn, const char *out_prefix, int region_len,
                    const double *fwd_rates, const double *rev_rates)
{
    char fn[FN_BUF + 1];
    FILE *fp;
    snprintf(fn, sizeof fn, "%s.pss.rates.txt", out_prefix);
    fp = fopen(fn, "w");
    if (!fp) {
        fprintf(stderr, "ERROR: Cannot write to file %s\n.", fn);
        return 1;
    }
    fprintf(fp, "### pss-bam.c v%s\n### FASTA: %s\n### BAM: %s\n### OUT: %s\n", PSS_VERSION, fasta_fn, bam_fn, fn);
    fputs("### Format of table:\n", fp);
    fputs("### Substitution rates for all possible nucleotide substitutions at\n", fp);
    fputs("### each position in the aligned reads.\n", fp);
    fputs("### First base is what was seen in the read.\n", fp);
    fputs("### Second base is what was in the genome at that position.\n", fp);
    fputs("### POS AC AG AT CA CG CT GA GC GT TA TC TG\n", fp);
    fputs("### Forward read substitution rates\n", fp);
    for (int pos = 0; pos < region_len; pos++) rate_row(fp, pos, fwd_rates + (size_t)pos * 12);
    fputs("\n\n### Reverse read substitution rates\n", fp);
    for (int pos = region_len - 1; pos >= 0; pos--) rate_row(fp, pos, rev_rates + (size_t)pos * 12);
    fclose(fp);
    return 0;
}

int fragkon_write_table(FILE *out, const char *fasta_fn, const char *bam_fn, int klen, const uint64_t *k5,
                        const uint64_t *k3)
{
    const uint64_t bins = (uint64_t)1 << (2 * klen);
    char kmer[40];
    if (klen < 1 || klen > 31) return 1;
    fprintf(out, "### fragkon.c v0.3\n### %s\n### %s\n", fasta_fn, bam_fn);
    fprintf(out, "# KMER\t5' CONTEXT COUNTS\t3' CONTEXT COUNTS\n");
    kmer[klen] = '\0';
    /* ACGT-lexicographic enumeration == ascending bin index */
    for (uint64_t b = 0; b < bins; b++) {
        for (int i = 0; i < klen; i++) kmer[i] = "ACGT"[(b >> (2 * (klen - 1 - i))) & 3u];
        unsigned int c5 = k5[b] > UINT_MAX ? UINT_MAX : (unsigned int)k5[b]; /* counts stick at UINT_MAX */
        unsigned int c3 = k3[b] > UINT_MAX ? UINT_MAX : (unsigned int)k3[b];
        fprintf(out, "%s\t%u\t%u\n", kmer, c5, c3);
    }
    return 0;
}

/* genome-kmer-count's table (/root/reference/genome-kmer-count.c:56-66 prints kmer2count(): an
 * unsigned int that sticks at UINT_MAX, kmer.c:102-104) */
int gkc_write_table(FILE *out, int klen, const uint64_t *counts)
{
    const uint64_t bins = (uint64_t)1 << (2 * klen);
    char kmer[40];
    if (klen < 1 || klen > 31) return 1;
    kmer[klen] = '\0';
    for (uint64_t b = 0; b < bins; b++) {
        for (int i = 0; i < klen; i++) kmer[i] = "ACGT"[(b >> (2 * (klen - 1 - i))) & 3u];
        fprintf(out, "%s\t%u\n", kmer, counts[b] > UINT_MAX ? UINT_MAX : (unsigned int)counts[b]);
    }
    return 0;
}
