/*
 * oracle/pss_oracle.c -- TEST INFRASTRUCTURE ONLY (see pss_oracle.h).
 *
 * CPU restatement of the reference per-read path.  It is written for clarity and
 * for faithfulness to the reference's observable behaviour, not for speed; the
 * product never routes through it.  Citations are file:line under /root/reference.
 *
 * Declared preconditions (inputs on which the reference itself has undefined
 * behaviour; the oracle defines them as "no tally" and the tests do not generate them
 * when comparing against oracle/_ref):
 *   P1  FASTA starts with '>' and every header line ends in '\n'
 *       (fasta-genome-io.c:105-150 walks off otherwise), ids <= 511 chars, unique,
 *       contigs <= 536870911 bases (fasta-genome-io.h:8-10).
 *   P2  SAM fields <= 2047 chars (sam-parse.h:10 buffers).
 *   P3  paired records whose |TLEN| exceeds strlen(SEQ) but whose CIGAR is "<|TLEN|>M"
 *       read stale bytes in the reference (pss-bam.c:401,411,485); here the missing
 *       read bases count as non-ACGT.
 *   P4  fragkon: alignment start < k/2 (or POS 0) indexes in front of the contig buffer
 *       in the reference (fragkon.c:129,137 compare an unsigned value with 0); here
 *       such a record is filtered (status 2).
 */
#include "pss_oracle.h"

#include <ctype.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define ORC_MAX_LINE 200000 /* sam-parse.h:8 MAX_LINE_LEN: fgets chunk size of both mains */

/* ================================================================================== */
/* genome                                                                              */
/* ================================================================================== */

static int ends_with_gz(const char *fn) /* fasta-genome-io.c:6-15 is_gz */
{
    size_t n = strlen(fn);
    return n >= 3 && fn[n - 3] == '.' && fn[n - 2] == 'g' && fn[n - 1] == 'z';
}

static int cmp_contig(const void *a, const void *b) /* fasta-genome-io.c:215-219 chr_cmp */
{
    return strcmp(((const orc_contig *)a)->id, ((const orc_contig *)b)->id);
}

/* Whole-file slurp (through zlib, which passes plain files through unchanged --
 * the reference picks gz vs plain by suffix, fasta-genome-io.c:32-48; for a well
 * formed input both roads give the same bytes). */
static unsigned char *slurp(const char *path, size_t *n_out)
{
    gzFile f;
    size_t cap = 1 << 20, n = 0;
    unsigned char *buf;
    if (ends_with_gz(path)) {
        f = gzopen(path, "rb");
    } else {
        FILE *probe = fopen(path, "rb");
        if (!probe) return NULL;
        fclose(probe);
        f = gzopen(path, "rb");
    }
    if (!f) return NULL;
    gzbuffer(f, 1 << 18);
    buf = (unsigned char *)malloc(cap);
    for (;;) {
        int got;
        if (cap - n < (1 << 19)) {
            cap *= 2;
            buf = (unsigned char *)realloc(buf, cap);
        }
        got = gzread(f, buf + n, (unsigned)(cap - n > (1u << 30) ? (1u << 30) : cap - n));
        if (got <= 0) break;
        n += (size_t)got;
    }
    gzclose(f);
    *n_out = n;
    return buf;
}

/* read_fasta / gzread_fasta, fasta-genome-io.c:105-200, for every record of the file:
 *   id   = bytes after '>' up to the first isspace()            (:111-115)
 *   rest of the header line is discarded                         (:116-118)
 *   body = every non-isspace() byte, toupper()ed, up to the next '>' or EOF (:120-131)
 * then init_genome's qsort by id (:236). */
orc_genome *orc_genome_load(const char *fasta_path)
{
    size_t n = 0, i = 0, cap_c = 16;
    unsigned char *txt = slurp(fasta_path, &n);
    orc_genome *g;
    if (!txt) return NULL;
    g = (orc_genome *)calloc(1, sizeof *g);
    g->contigs = (orc_contig *)calloc(cap_c, sizeof(orc_contig));
    while (i < n) {
        orc_contig c;
        size_t id0, id1, w = 0, body0;
        if (txt[i] != '>') { /* precondition P1 */
            orc_genome_free(g);
            free(txt);
            return NULL;
        }
        id0 = ++i;
        while (i < n && !isspace(txt[i])) i++;
        id1 = i;
        while (i < n && txt[i] != '\n') i++;
        if (i >= n) { /* header without newline: the reference spins forever (P1) */
            orc_genome_free(g);
            free(txt);
            return NULL;
        }
        body0 = i;
        while (i < n && txt[i] != '>') i++;
        c.id = (char *)malloc(id1 - id0 + 1);
        memcpy(c.id, txt + id0, id1 - id0);
        c.id[id1 - id0] = '\0';
        c.seq = (unsigned char *)malloc(i - body0 + 1);
        for (size_t j = body0; j < i; j++) {
            if (!isspace(txt[j])) c.seq[w++] = (unsigned char)toupper(txt[j]);
        }
        c.seq[w] = '\0';
        c.len = w;
        if (g->n == cap_c) {
            cap_c *= 2;
            g->contigs = (orc_contig *)realloc(g->contigs, cap_c * sizeof(orc_contig));
        }
        g->contigs[g->n++] = c;
    }
    free(txt);
    qsort(g->contigs, g->n, sizeof(orc_contig), cmp_contig);
    return g;
}

orc_genome *orc_genome_from_arrays(size_t n, const char *const *ids,
                                   const unsigned char *const *seqs, const size_t *lens)
{
    orc_genome *g = (orc_genome *)calloc(1, sizeof *g);
    g->contigs = (orc_contig *)calloc(n ? n : 1, sizeof(orc_contig));
    g->n = n;
    for (size_t i = 0; i < n; i++) {
        g->contigs[i].id = strdup(ids[i]);
        g->contigs[i].seq = (unsigned char *)malloc(lens[i] + 1);
        memcpy(g->contigs[i].seq, seqs[i], lens[i]);
        g->contigs[i].seq[lens[i]] = '\0';
        g->contigs[i].len = lens[i];
    }
    qsort(g->contigs, g->n, sizeof(orc_contig), cmp_contig);
    return g;
}

const orc_contig *orc_find_contig(const orc_genome *g, const char *id) /* find_seq :202-213 */
{
    orc_contig key;
    key.id = (char *)id;
    return (const orc_contig *)bsearch(&key, g->contigs, g->n, sizeof(orc_contig), cmp_contig);
}

void orc_genome_free(orc_genome *g)
{
    if (!g) return;
    for (size_t i = 0; i < g->n; i++) {
        free(g->contigs[i].id);
        free(g->contigs[i].seq);
    }
    free(g->contigs);
    free(g);
}

/* ================================================================================== */
/* SAM text                                                                            */
/* ================================================================================== */

/* line2saml, sam-parse.c:36-68.  The eleven mandatory fields are taken with scanf's
 * own rules (whitespace-delimited tokens; %u/%lu accept a sign; %i auto-detects the
 * base), which is what makes e.g. a blank inside a field shift everything; so the
 * restatement uses scanf too rather than approximating it.  A blank in a scanf
 * format matches any run of whitespace, exactly like the reference's tabs.
 * Success needs all eleven conversions and strlen(SEQ)==strlen(QUAL) (:50). */
int orc_parse_line(const char *line, orc_aln *a, char *scratch)
{
    size_t w = strlen(line) + 1;
    char *qname = scratch, *rname = scratch + w, *cigar = scratch + 2 * w;
    char *mrnm = scratch + 3 * w, *seq = scratch + 4 * w, *qual = scratch + 5 * w;
    unsigned int flag = 0, mapq = 0, mpos = 0;
    unsigned long pos = 0;
    int tlen = 0;
    int got = sscanf(line, "%s %u %s %lu %u %s %s %u %i %s %s", qname, &flag, rname, &pos, &mapq,
                     cigar, mrnm, &mpos, &tlen, seq, qual);
    if (got < 11) return 1;
    if (strlen(seq) != strlen(qual)) return 1;
    a->rname = rname;
    a->cigar = cigar;
    a->seq = seq;
    a->flag = flag;
    a->mapq = mapq;
    a->pos = pos;
    a->seq_len = (int)strlen(seq);
    a->isize = (flag & 1u) ? tlen : (int)strlen(seq); /* :66-68 unpaired => isize = strlen(SEQ) */
    return 0;
}

/* ================================================================================== */
/* shared per-read helpers                                                             */
/* ================================================================================== */

/* The 2-bit code both tools give a base: A0 C1 G2 T3, -1 for anything else.
 * pss-bam.c:205-251 (pair strings "AA".."TT" in this order) and kmer.c:190-208. */
static inline int base_code(int c)
{
    switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1;
    }
}

/* one byte of do_revcomp / do_rvcmp, pss-bam.c:60-79, fragkon.c:27-46:
 * A<->T, C<->G with lower case folded to the upper-case complement, every other byte
 * passes through untouched. */
static inline int comp_byte(int c)
{
    switch (c) {
    case 'A': case 'a': return 'T';
    case 'C': case 'c': return 'G';
    case 'G': case 'g': return 'C';
    case 'T': case 't': return 'A';
    default: return c;
    }
}

/* cigar_ok, pss-bam.c:113-123 / fragkon.c:68-78: CIGAR must be the decimal text of
 * `len` followed by a single 'M' and nothing else. */
static int cigar_is_single_match(int len, const char *cigar)
{
    char want[32];
    snprintf(want, sizeof want, "%dM", len);
    return strcmp(want, cigar) == 0;
}

/* flag bits, sam-parse.c:53-64 */
#define FL_PAIRED 0x1u
#define FL_PROPER 0x2u
#define FL_UNMAP 0x4u
#define FL_MUNMAP 0x8u
#define FL_REVERSE 0x10u
#define FL_READ1 0x40u
#define FL_READ2 0x80u
#define FL_SECONDARY 0x100u
#define FL_QCFAIL 0x200u
#define FL_DUP 0x400u
#define FL_SUPP 0x800u
#define FL_REJECT (FL_UNMAP | FL_SECONDARY | FL_QCFAIL | FL_DUP | FL_SUPP)

/* ================================================================================== */
/* pss-bam                                                                             */
/* ================================================================================== */

/* A view of "genome_seq"/"read_seq" as process_aln sees them after the optional
 * reverse complement (pss-bam.c:423-436): index 0..L+3 over the L aligned bases plus
 * two context bases each side; index 0..L-1 over the read. */
typedef struct {
    const unsigned char *ref; /* contig bytes                                        */
    size_t ref_len;
    long s;                   /* 0-based alignment start                            */
    int L;
    int rev;
    const char *read;
    int read_len;             /* strlen(SEQ); beyond it P3 applies                  */
} pss_view;

static inline int view_g(const pss_view *v, long j)
{
    if (!v->rev) return toupper(v->ref[v->s - 2 + j]);            /* :423-424 */
    return comp_byte(toupper(v->ref[v->s - 2 + (v->L + 3 - j)])); /* :433     */
}

static inline int view_r(const pss_view *v, long i)
{
    long k = v->rev ? (long)v->L - 1 - i : i;
    int c = (k >= 0 && k < v->read_len) ? toupper((unsigned char)v->read[k]) : 0; /* :425, P3 */
    return v->rev ? comp_byte(c) : c;                                               /* :436    */
}

/* add_ctx_counts, pss-bam.c:169-189: row 0 takes the context base two away from the
 * alignment, row 1 the adjacent one; only the AA/CC/GG/TT cells (0,5,10,15) move. */
static void tally_ctx(unsigned long *tab, int first_cb, int second_cb)
{
    int c0 = base_code(second_cb), c1 = base_code(first_cb);
    if (c0 >= 0) tab[0 * 16 + 5 * c0] += 1;
    if (c1 >= 0) tab[1 * 16 + 5 * c1] += 1;
}

/* add_fwd_counts, pss-bam.c:197-257: position i from the 5' end pairs read[i] with
 * genome_seq[2+i]; column = 4*code(read)+code(genome); non-ACGT on either side skips. */
static void tally_fwd(unsigned long *tab, const pss_view *v, int N)
{
    for (int i = 0; i < N; i++) {
        int rd = base_code(view_r(v, i)), rf = base_code(view_g(v, 2 + i));
        if (rd >= 0 && rf >= 0) tab[(i + 2) * 16 + 4 * rd + rf] += 1;
    }
}

/* add_rev_counts, pss-bam.c:266-326: position i from the 3' end pairs
 * read[L-1-i] with genome_seq[L+1-i]. */
static void tally_rev(unsigned long *tab, const pss_view *v, int N)
{
    for (int i = 0; i < N; i++) {
        int rd = base_code(view_r(v, (long)v->L - 1 - i));
        int rf = base_code(view_g(v, (long)v->L + 1 - i));
        if (rd >= 0 && rf >= 0) tab[(i + 2) * 16 + 4 * rd + rf] += 1;
    }
}

/* process_aln, pss-bam.c:390-496 */
int orc_pss_process(const orc_genome *g, const orc_pss_params *p, orc_aln *a, unsigned long *fwd,
                    unsigned long *rev)
{
    const orc_contig *ref = orc_find_contig(g, a->rname); /* :393-396 */
    int seq_len, N = p->region_len;
    long aln_start, aln_end;
    unsigned int fl = a->flag;
    pss_view v;
    if (!ref) return 1;

    seq_len = abs(a->isize);         /* :401 */
    aln_start = (long)a->pos - 1;    /* :403 */
    aln_end = aln_start + seq_len - 1;

    /* filters :407-420, same operand types as the reference (the second test compares a
     * long with a size_t, i.e. unsigned; the mapq test compares unsigned with int) */
    if (aln_start - 2 < 0) return -1;
    if ((unsigned long)(aln_end + 2) > (unsigned long)ref->len - 1) return -1;
    if (a->mapq < (unsigned int)p->min_mq) return -1;
    if (!((unsigned long)seq_len >= p->min_read_len && (unsigned long)seq_len <= p->max_read_len &&
          seq_len >= N)) /* read_len_ok :96-103 */
        return -1;
    if (!cigar_is_single_match(seq_len, a->cigar)) return -1;
    if (fl & FL_REJECT) return -1;
    if (p->merged_only && (fl & FL_PAIRED)) return -1;

    v.ref = ref->seq;
    v.ref_len = ref->len;
    v.s = aln_start;
    v.L = seq_len;
    v.rev = (fl & FL_REVERSE) != 0;
    v.read = a->seq;
    v.read_len = a->seq_len;

    {
        int up1 = view_g(&v, 1), up0 = view_g(&v, 0);                              /* :351-352 */
        int dn1 = view_g(&v, (long)seq_len + 2), dn0 = view_g(&v, (long)seq_len + 3); /* :353-354 */
        int up_ok = strchr(p->up_ctx, up1) != NULL;     /* :137 / :461 / :483 */
        int down_ok = strchr(p->down_ctx, dn1) != NULL; /* :138 / :472 / :489 */

        if (!(fl & FL_PAIRED)) { /* merged / single-end :428-447 */
            if (up_ok && down_ok) {
                tally_ctx(fwd, up1, up0);
                tally_ctx(rev, dn1, dn0);
                tally_fwd(fwd, &v, N);
                tally_rev(rev, &v, N);
                return 0;
            }
            return -1;
        }
        if ((fl & FL_PROPER) && !(fl & FL_MUNMAP)) { /* :450-494 */
            if ((fl & FL_READ1) && up_ok) { /* first mate: 5' side only */
                tally_ctx(fwd, up1, up0);
                tally_fwd(fwd, &v, N);
                return 0;
            }
            if ((fl & FL_READ2) && down_ok) { /* `else if`: also reached by 0xC0 records */
                tally_ctx(rev, dn1, dn0);
                tally_rev(rev, &v, N);
                return 0;
            }
        }
    }
    return -1;
}

/* the read loop of main(), pss-bam.c:764-783.  `samtools view` prints no header, so
 * leading '@' lines of a SAM text file are dropped here like the PATH shim does. */
typedef int (*line_fn)(void *ctx, orc_aln *a);

static int for_each_sam_line(const char *sam_path, line_fn fn, void *ctx,
                             unsigned long status[ORC_ST_N])
{
    FILE *f = fopen(sam_path, "r");
    char *line, *scratch;
    int in_header = 1;
    if (!f) return -1;
    line = (char *)malloc(ORC_MAX_LINE + 2);
    scratch = (char *)malloc(6 * (size_t)(ORC_MAX_LINE + 2));
    while (fgets(line, ORC_MAX_LINE + 1, f)) {
        orc_aln a;
        int st;
        if (in_header && line[0] == '@') continue;
        in_header = 0;
        if (orc_parse_line(line, &a, scratch)) {
            if (status) status[ORC_ST_PARSE_SKIP]++;
            continue;
        }
        st = fn(ctx, &a);
        if (status) status[st]++;
    }
    free(line);
    free(scratch);
    fclose(f);
    return 0;
}

typedef struct {
    const orc_genome *g;
    const orc_pss_params *p;
    unsigned long *fwd, *rev;
} pss_ctx;

static int pss_line(void *vctx, orc_aln *a)
{
    pss_ctx *c = (pss_ctx *)vctx;
    int st = orc_pss_process(c->g, c->p, a, c->fwd, c->rev);
    return st == 0 ? ORC_ST_OK : st == 1 ? ORC_ST_NO_CONTIG : ORC_ST_FILTERED;
}

int orc_pss_run(const orc_genome *g, const char *sam_path, const orc_pss_params *p,
                unsigned long *fwd, unsigned long *rev, unsigned long status[ORC_ST_N])
{
    pss_ctx c = {g, p, fwd, rev};
    return for_each_sam_line(sam_path, pss_line, &c, status);
}

/* find_sub_rates, pss-bam.c:504-529.  Column sums are formed in unsigned long and only
 * then converted; a row with any empty reference-base column keeps all-zero rates.
 * Quotient order: AC AG AT CA CG CT GA GC GT TA TC TG (count / n_of_reference_base). */
void orc_pss_rates(int region_len, const unsigned long *counts, double *rates)
{
    static const int cell[12] = {1, 2, 3, 4, 6, 7, 8, 9, 11, 12, 13, 14};
    for (int i = 0; i < region_len; i++) {
        const unsigned long *c = counts + (size_t)(i + 2) * 16;
        double n[4];
        for (int b = 0; b < 4; b++) n[b] = c[b] + c[4 + b] + c[8 + b] + c[12 + b];
        for (int j = 0; j < 12; j++) rates[i * 12 + j] = 0.0;
        if (n[0] == 0 || n[1] == 0 || n[2] == 0 || n[3] == 0) continue;
        for (int j = 0; j < 12; j++) rates[i * 12 + j] = c[cell[j]] / n[cell[j] & 3];
    }
}

/* print_counts, pss-bam.c:538-586 -- the byte-exact report (note the hard-coded
 * "v1.2.1:" with colon, the trailing tab on every row, and the reverse block listing
 * rows N-1..0 followed by the context rows labelled 1 and 2). */
int orc_pss_write_counts(const char *fasta_fn, const char *bam_fn, const char *out_prefix,
                         int region_len, const unsigned long *fwd, const unsigned long *rev)
{
    char fn[4096];
    FILE *fp;
    snprintf(fn, 2047, "%s.pss.counts.txt", out_prefix); /* MAX_FN_LEN buffer :541-542 */
    fp = fopen(fn, "w");
    if (!fp) return 1;
    fprintf(fp, "### pss-bam.c v1.2.1:\n### FASTA: %s\n### BAM: %s\n### OUT: %s\n", fasta_fn, bam_fn, fn);
    fputs("### Format of table:\n"
          "### Counts of how often a read base and genome base were seen at\n"
          "### each position in the aligned reads.\n"
          "### First base is what was seen in the read.\n"
          "### Second base is what was in the genome at that position.\n"
          "### POS AA AC AG AT CA CC CG CT GA GC GG GT TA TC TG TT\n"
          "### Forward read substitution counts and base context\n",
          fp);
    for (int i = -2; i < region_len; i++) {
        fprintf(fp, "%d\t", i);
        for (int j = 0; j < 16; j++) fprintf(fp, "%lu\t", fwd[(i + 2) * 16 + j]);
        fputc('\n', fp);
    }
    fputs("\n\n### Reverse read substitution counts and base context\n", fp);
    for (int i = region_len - 1; i >= 0; i--) {
        fprintf(fp, "%d\t", i);
        for (int j = 0; j < 16; j++) fprintf(fp, "%lu\t", rev[(i + 2) * 16 + j]);
        fputc('\n', fp);
    }
    for (int i = 1; i < 3; i++) {
        fprintf(fp, "%d\t", i);
        for (int j = 0; j < 16; j++) fprintf(fp, "%lu\t", rev[(2 - i) * 16 + j]);
        fputc('\n', fp);
    }
    fclose(fp);
    return 0;
}

/* print_rates, pss-bam.c:595-633 */
int orc_pss_write_rates(const char *fasta_fn, const char *bam_fn, const char *out_prefix,
                        int region_len, const double *fwd_rates, const double *rev_rates)
{
    char fn[4096];
    FILE *fp;
    snprintf(fn, 2047, "%s.pss.rates.txt", out_prefix);
    fp = fopen(fn, "w");
    if (!fp) return 1;
    fprintf(fp, "### pss-bam.c v%s\n### FASTA: %s\n### BAM: %s\n### OUT: %s\n", "1.2.1", fasta_fn, bam_fn, fn);
    fputs("### Format of table:\n"
          "### Substitution rates for all possible nucleotide substitutions at\n"
          "### each position in the aligned reads.\n"
          "### First base is what was seen in the read.\n"
          "### Second base is what was in the genome at that position.\n"
          "### POS AC AG AT CA CG CT GA GC GT TA TC TG\n"
          "### Forward read substitution rates\n",
          fp);
    for (int i = 0; i < region_len; i++) {
        fprintf(fp, "%d\t", i);
        for (int j = 0; j < 12; j++) fprintf(fp, "%.5e\t", fwd_rates[i * 12 + j]);
        fputc('\n', fp);
    }
    fputs("\n\n### Reverse read substitution rates\n", fp);
    for (int i = region_len - 1; i >= 0; i--) {
        fprintf(fp, "%d\t", i);
        for (int j = 0; j < 12; j++) fprintf(fp, "%.5e\t", rev_rates[i * 12 + j]);
        fputc('\n', fp);
    }
    fclose(fp);
    return 0;
}

/* ================================================================================== */
/* fragkon                                                                             */
/* ================================================================================== */

/* add_to_ksp, kmer.c:43-111: the k bases (case-folded) must all be ACGT; the bin is
 * the base-4 number read left to right (kmer2inx :184-214 for the first <=8 bases, the
 * pointer tree :67-98 for the rest -- together simply a 4^k-ary index); the count is an
 * unsigned int that sticks at UINT_MAX (:102-104).  `at(ctx,i)` yields base i. */
typedef int (*base_at)(const void *ctx, long i);

static int kmer_add(unsigned int *tab, int k, base_at at, const void *ctx)
{
    size_t bin = 0;
    for (int i = 0; i < k; i++) {
        int c = base_code(toupper(at(ctx, i)));
        if (c < 0) return -1;
        bin = (bin << 2) | (size_t)c;
    }
    if (tab[bin] < UINT_MAX) tab[bin] += 1;
    return 0;
}

typedef struct {
    const unsigned char *ref;
    size_t ref_len;
    long origin;   /* forward: first base of the window on the contig                */
    long sub0;     /* reverse: contig index of sub_ref[0]                            */
    long sub_len;  /* reverse: length handed to do_rvcmp (seq_len + KLEN)            */
    long rc_start; /* reverse: index into rvcmp_sub_ref where the k-mer starts       */
} fk_window;

static int fk_fwd_at(const void *vctx, long i)
{
    const fk_window *w = (const fk_window *)vctx;
    size_t j = (size_t)(w->origin + i);
    return j <= w->ref_len ? w->ref[j] : 0; /* index len is the NUL terminator */
}

/* rvcmp_sub_ref[t] = comp(sub_ref[sub_len-1-t]), fragkon.c:156-160; sub_ref was filled
 * by strncpy (:101-103), so anything at/after the contig's terminator reads as NUL. */
static int fk_rev_at(const void *vctx, long i)
{
    const fk_window *w = (const fk_window *)vctx;
    long t = w->rc_start + i;
    size_t j = (size_t)(w->sub0 + (w->sub_len - 1 - t));
    int c = j < w->ref_len ? w->ref[j] : 0;
    return comp_byte(c);
}

/* process_aln, fragkon.c:122-216 */
int orc_fk_process(const orc_genome *g, const orc_fk_params *p, const orc_aln *a, unsigned int *k5,
                   unsigned int *k3)
{
    const orc_contig *ref = orc_find_contig(g, a->rname); /* :124-127 */
    unsigned long aln_start, aln_end;
    unsigned int ok, ik, fl = a->flag;
    int K = p->klen, L = a->seq_len;
    fk_window w5, w3;
    if (!ref) return 1;

    aln_start = a->pos - 1;             /* :129, unsigned */
    aln_end = aln_start + (unsigned long)L - 1; /* :130 */
    ok = (unsigned int)K / 2;           /* :134 */
    ik = (unsigned int)K - ok;          /* :135 */

    /* :137-146.  The first clause of the reference (start - k/2 >= 0 on an unsigned)
     * is vacuous; P4 replaces it by the test it was meant to be. */
    if ((long)a->pos - 1 < (long)ok) return 2; /* P4 (covers POS 0 as well) */
    if (!(aln_end + (unsigned long)(K / 2) <= (unsigned long)ref->len - 1)) return 2;
    if (!(a->mapq >= (unsigned int)p->min_mq)) return 2;
    if (!((unsigned long)L >= p->min_read_len && (unsigned long)L <= p->max_read_len)) return 2; /* :52-58 */
    if (!cigar_is_single_match(L, a->cigar)) return 2;
    if (fl & FL_REJECT) return 2;

    w5.ref = w3.ref = ref->seq;
    w5.ref_len = w3.ref_len = ref->len;
    if (fl & FL_REVERSE) { /* :152-172, :192-204 */
        w5.sub0 = w3.sub0 = (long)aln_start - (long)ok;
        w5.sub_len = w3.sub_len = (long)L + K;
        w5.rc_start = 0;
        w3.rc_start = (long)ok + L - (long)ik;
        w5.origin = w3.origin = 0;
    } else { /* :176-177, :207-210 */
        w5.origin = (long)aln_start - (long)ok;
        w3.origin = (long)aln_start + L - (long)ik;
        w5.sub0 = w3.sub0 = w5.sub_len = w3.sub_len = w5.rc_start = w3.rc_start = 0;
    }
    {
        base_at at = (fl & FL_REVERSE) ? fk_rev_at : fk_fwd_at;
        if (!(fl & FL_PAIRED)) { /* :149-183; note: no -m test on this branch */
            int a5 = kmer_add(k5, K, at, &w5);
            int a3 = kmer_add(k3, K, at, &w3);
            return (a5 == 0 && a3 == 0) ? 0 : -1;
        }
        if (!p->merged_only && (fl & FL_PROPER) && !(fl & FL_MUNMAP)) { /* :187-213 */
            if (fl & FL_READ1) return kmer_add(k5, K, at, &w5);
            if (fl & FL_READ2) return kmer_add(k3, K, at, &w3);
        }
    }
    return 2;
}

typedef struct {
    const orc_genome *g;
    const orc_fk_params *p;
    unsigned int *k5, *k3;
} fk_ctx;

static int fk_line(void *vctx, orc_aln *a)
{
    fk_ctx *c = (fk_ctx *)vctx;
    int st = orc_fk_process(c->g, c->p, a, c->k5, c->k3);
    return st == 0 ? ORC_ST_OK : st == 1 ? ORC_ST_NO_CONTIG : st == 2 ? ORC_ST_FILTERED : ORC_ST_KMER_FAIL;
}

int orc_fk_run(const orc_genome *g, const char *sam_path, const orc_fk_params *p, unsigned int *k5,
               unsigned int *k3, unsigned long status[ORC_ST_N])
{
    fk_ctx c = {g, p, k5, k3};
    if (p->klen < 1 || p->klen > 15) return -2;
    return for_each_sam_line(sam_path, fk_line, &c, status);
}

/* header :367-368 + print_kmer_counts :231-249: every k-mer in ACGT-lexicographic
 * order, which is bin order. */
int orc_fk_write(FILE *out, const char *fasta_fn, const char *bam_fn, int klen, const unsigned int *k5,
                 const unsigned int *k3)
{
    size_t nb = (size_t)1 << (2 * klen);
    char kmer[64];
    fprintf(out, "### fragkon.c v0.3\n### %s\n### %s\n", fasta_fn, bam_fn);
    fprintf(out, "# KMER\t5' CONTEXT COUNTS\t3' CONTEXT COUNTS\n");
    kmer[klen] = '\0';
    for (size_t b = 0; b < nb; b++) {
        for (int i = 0; i < klen; i++) kmer[i] = "ACGT"[(b >> (2 * (klen - 1 - i))) & 3];
        fprintf(out, "%s\t%u\t%u\n", kmer, k5[b], k3[b]);
    }
    return 0;
}

/* ================================================================================== */
/* genome-kmer-count                                                                   */
/* ================================================================================== */

static int gkc_at(const void *ctx, long i) { return ((const unsigned char *)ctx)[i]; }

/* count_kmers, genome-kmer-count.c:69-79: add_to_ksp at every start i < len - k + 1 */
int orc_genome_kmer_count(const orc_genome *g, int klen, unsigned int *counts)
{
    if (klen < 1 || klen > 15) return -2;
    for (size_t c = 0; c < g->n; c++) {
        const orc_contig *s = &g->contigs[c];
        if (s->len < (size_t)klen) continue;
        for (size_t i = 0; i + (size_t)klen <= s->len; i++) (void)kmer_add(counts, klen, gkc_at, s->seq + i);
    }
    return 0;
}
