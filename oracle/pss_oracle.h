/*
 * oracle/pss_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A CPU restatement (plain C, single thread) of the reference's per-read path, used
 * exclusively as the *checker* by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  Nothing under pss-bam_amd/ (the product) may include,
 * link or call anything declared here.
 *
 * Parity pin: this restatement is itself checked (tests/test_oracle_vs_ref.py)
 * against the UNMODIFIED reference compiled by oracle/Makefile into oracle/_ref/, on
 * randomised inputs, and against the golden vectors in tests/golden/ which were
 * produced by that same reference build (tests/golden/make_golden.py).
 *
 * Every function cites the reference file:line it follows (paths under
 * /root/reference).
 */
#ifndef PSS_ORACLE_H
#define PSS_ORACLE_H

#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- genome (fasta-genome-io.c:105-238) ---------------------------------------- */
typedef struct orc_contig {
    char *id;            /* chars after '>' up to first whitespace                 */
    unsigned char *seq;  /* upper-cased, whitespace stripped, NUL terminated       */
    size_t len;
} orc_contig;

typedef struct orc_genome {
    orc_contig *contigs; /* sorted by strcmp(id) like init_genome's qsort          */
    size_t n;
} orc_genome;

orc_genome *orc_genome_load(const char *fasta_path);   /* NULL on I/O / format error */
/* builds a genome from in-memory contigs (already in loaded form); copies nothing but
 * the pointers' contents -- used by bench.py to avoid writing a 3 GB FASTA.          */
orc_genome *orc_genome_from_arrays(size_t n, const char *const *ids,
                                   const unsigned char *const *seqs, const size_t *lens);
const orc_contig *orc_find_contig(const orc_genome *g, const char *id);
void orc_genome_free(orc_genome *g);

/* ---- one parsed SAM line (sam-parse.c:10-91) -------------------------------------- */
typedef struct orc_aln {
    char *rname, *cigar, *seq;  /* point into the caller's scratch                  */
    unsigned int flag, mapq;
    unsigned long pos;          /* 1-based POS                                      */
    int isize;                  /* TLEN, or strlen(SEQ) when the 0x1 bit is clear   */
    int seq_len;                /* strlen(SEQ)                                      */
} orc_aln;

/* ---- pss-bam (pss-bam.c:12-18 option globals) ------------------------------------ */
typedef struct orc_pss_params {
    int region_len;             /* -r  REGION_LEN                                   */
    unsigned long min_read_len; /* -l                                               */
    unsigned long max_read_len; /* -L                                               */
    int min_mq;                 /* -q                                               */
    const char *up_ctx;         /* -U                                               */
    const char *down_ctx;       /* -D                                               */
    int merged_only;            /* -m                                               */
} orc_pss_params;

/* status tallies returned by the run functions: how many lines ended in each state  */
enum { ORC_ST_OK = 0, ORC_ST_PARSE_SKIP = 1, ORC_ST_NO_CONTIG = 2, ORC_ST_FILTERED = 3,
       ORC_ST_KMER_FAIL = 4, ORC_ST_N = 5 };

/* fwd / rev: (region_len+2)*16 unsigned long each, row-major, caller-zeroed or
 * accumulated into.  Returns 0, or -1 when the SAM file cannot be opened.           */
int orc_pss_run(const orc_genome *g, const char *sam_path, const orc_pss_params *p,
                unsigned long *fwd, unsigned long *rev, unsigned long status[ORC_ST_N]);
/* same on one already-parsed alignment; returns 0 / 1 / -1 like process_aln          */
int orc_pss_process(const orc_genome *g, const orc_pss_params *p, orc_aln *a,
                    unsigned long *fwd, unsigned long *rev);

/* rates: region_len*12 doubles (pss-bam.c:504-529)                                   */
void orc_pss_rates(int region_len, const unsigned long *counts, double *rates);
int orc_pss_write_counts(const char *fasta_fn, const char *bam_fn, const char *out_prefix,
                         int region_len, const unsigned long *fwd, const unsigned long *rev);
int orc_pss_write_rates(const char *fasta_fn, const char *bam_fn, const char *out_prefix,
                        int region_len, const double *fwd_rates, const double *rev_rates);

/* ---- fragkon (fragkon.c:14-18 option globals) ------------------------------------- */
typedef struct orc_fk_params {
    int klen;                   /* -k                                               */
    int min_mq;                 /* -q                                               */
    unsigned long min_read_len; /* -l                                               */
    unsigned long max_read_len; /* -L                                               */
    int merged_only;            /* -m                                               */
} orc_fk_params;

/* k5 / k3: 4^klen unsigned int each (saturating, kmer.c:102-104), index = 2 bits per
 * base left-to-right, A0 C1 G2 T3 (kmer.c:184-214).  klen in [1,15].                */
int orc_fk_run(const orc_genome *g, const char *sam_path, const orc_fk_params *p,
               unsigned int *k5, unsigned int *k3, unsigned long status[ORC_ST_N]);
int orc_fk_process(const orc_genome *g, const orc_fk_params *p, const orc_aln *a,
                   unsigned int *k5, unsigned int *k3);
int orc_fk_write(FILE *out, const char *fasta_fn, const char *bam_fn, int klen,
                 const unsigned int *k5, const unsigned int *k3);

/* ---- genome-kmer-count (genome-kmer-count.c:56-79) ---------------------------------- */
/* counts[4^klen]: every k-mer start 0..len-k of every contig, same bins / saturation as fragkon.
 * Contigs shorter than k contribute nothing (the reference's loop bound underflows there). */
int orc_genome_kmer_count(const orc_genome *g, int klen, unsigned int *counts);

/* ---- SAM text (sam-parse.c:10-91) -------------------------------------------------- */
/* Parses one line into *a using `scratch` (>= 3*(strlen(line)+1) bytes).
 * Returns 0 ok / 1 "problem" exactly where line2saml does.                           */
int orc_parse_line(const char *line, orc_aln *a, char *scratch);

#ifdef __cplusplus
}
#endif
#endif
