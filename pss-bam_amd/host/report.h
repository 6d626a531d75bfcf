/* pss-bam_amd/host/report.h -- report writers of the two front ends (byte-exact parity surface). */
#ifndef PSSBAM_REPORT_H
#define PSSBAM_REPORT_H
#include <stdint.h>
#include <stdio.h>

/* rates[i*12 + j], j = AC AG AT CA CG CT GA GC GT TA TC TG; counts = (region_len+2)*16 */
void pss_sub_rates(int region_len, const unsigned long *counts, double *rates);
int pss_write_counts(const char *fasta_fn, const char *bam_fn, const char *out_prefix, int region_len,
                     const unsigned long *fwd, const unsigned long *rev);
int pss_write_rates(const char *fasta_fn, const char *bam_fn, const char *out_prefix, int region_len,
                    const double *fwd_rates, const double *rev_rates);
/* k5 / k3: 4^klen 64-bit bins (clamped to UINT_MAX on output) */
int fragkon_write_table(FILE *out, const char *fasta_fn, const char *bam_fn, int klen, const uint64_t *k5,
                        const uint64_t *k3);
/* counts: 4^klen 64-bit bins (clamped to UINT_MAX on output) */
int gkc_write_table(FILE *out, int klen, const uint64_t *counts);
#endif
