/*
 * pss-bam_amd/host/genome_load.c -- implementation of include/fasta-genome-io.h.
 *
 * Same contract as the reference loader (/root/reference/fasta-genome-io.c), new
 * machinery: the file is pulled through read()/gzread() in 4 MiB blocks and a 256-entry
 * class table drives one tight loop that drops white space, upper-cases, and splits
 * records at '>' -- instead of one fgetc()/gzgetc() + isspace() + toupper() per byte
 * (fasta-genome-io.c:105-148 / :157-198).  Contig storage grows geometrically per
 * contig; the reference's fixed 512 MiB staging buffer (:28) is only allocated when a
 * caller uses the record-at-a-time API (init_fasta_src / get_next_fa).
 *
 * Behaviour kept, byte for byte, for well-formed input:
 *   - id = bytes after '>' up to the first isspace() byte            (:111-115)
 *   - remainder of the header line ignored                            (:116-118)
 *   - body = all non-isspace() bytes, toupper()ed, until '>' or EOF   (:120-131)
 *   - a '>' ANYWHERE in a body starts the next record (the reference tests every byte)
 *   - contigs longer than MAX_SEQ_LEN are cut there with the reference's stderr note
 *     (:140-142); unlike the reference (which then mis-parses the tail as garbage
 *     records) the rest of that contig is skipped
 *   - Genome.seqs sorted with chr_cmp                                  (:236)
 * Declared preconditions (undefined behaviour in the reference, diagnosed here):
 *   file starts with '>' ; every header line ends in '\n' ; ids <= MAX_ID_LEN.
 */
#include "fasta-genome-io.h"

#include <errno.h>
#include <fcntl.h>
#include <unistd.h>

/* byte classes for the block parser */
enum { C_BASE = 0, C_SPACE = 1, C_GT = 2 };
static unsigned char g_class[256];
static unsigned char g_upper[256];
static int g_tables_ready = 0;

static void init_tables(void)
{
    if (g_tables_ready) return;
    for (int c = 0; c < 256; c++) {
        g_class[c] = isspace(c) ? C_SPACE : (c == '>' ? C_GT : C_BASE);
        g_upper[c] = (unsigned char)toupper(c);
    }
    g_tables_ready = 1;
}

int is_gz(const char *fn)
{
    size_t n = strlen(fn);
    if (n < 3) return 0; /* the reference indexes before the string here */
    return fn[n - 3] == '.' && fn[n - 2] == 'g' && fn[n - 1] == 'z';
}

FILE *fileOpen(const char *name, char access_mode[])
{
    FILE *f = fopen(name, access_mode);
    if (f == NULL) {
        fprintf(stderr, "%s\n", name);
        perror("Cannot open file");
    }
    return f;
}

int chr_cmp(const void *v1, const void *v2)
{
    const Seq *a = *(Seq *const *)v1, *b = *(Seq *const *)v2;
    return strcmp(a->id, b->id);
}

Seq *find_seq(Genome *genome, const char id[])
{
    Seq **hit;
    /* the reference copies the id into genome->dummy and bsearches with it (which is what
     * makes it non-reentrant); a key on the stack gives the same answer without the copy,
     * but callers may rely on dummy->id holding the last query, so keep that too */
    strncpy(genome->dummy->id, id, MAX_ID_LEN);
    genome->dummy->id[MAX_ID_LEN] = '\0';
    if (strlen(id) > MAX_ID_LEN) return NULL; /* no stored id can be that long */
    hit = (Seq **)bsearch(&genome->dummy, genome->seqs, genome->n_seqs, sizeof(Seq *), chr_cmp);
    return hit ? *hit : NULL;
}

/* ------------------------------------------------------------------------------------ */
/* block loader used by init_genome                                                      */
/* ------------------------------------------------------------------------------------ */

typedef struct {
    int gz;
    gzFile zf;
    int fd;
} blk_src;

static long blk_read(blk_src *s, unsigned char *buf, size_t cap)
{
    if (s->gz) return (long)gzread(s->zf, buf, (unsigned)cap);
    for (;;) {
        ssize_t n = read(s->fd, buf, cap);
        if (n < 0 && errno == EINTR) continue;
        return (long)n;
    }
}

typedef struct {
    Seq *cur;        /* contig being filled, NULL before the first '>' */
    size_t cap;      /* allocated bytes of cur->seq                    */
    int state;       /* 0 body, 1 id, 2 rest of header line, 3 skipping an over-long contig */
    size_t id_len;
    int bad;         /* format violation seen                           */
} parse_state;

static void finish_contig(Genome *g, parse_state *ps)
{
    Seq *s = ps->cur;
    if (!s) return;
    s->seq[s->len] = '\0';
    if (ps->cap > s->len + 1) {
        char *shr = (char *)realloc(s->seq, s->len + 1);
        if (shr) s->seq = shr;
    }
    if (g->n_seqs < MAX_GENOME_SEQS) g->seqs[g->n_seqs++] = s;
    else destroy_seq(s);
    ps->cur = NULL;
}

static int begin_contig(parse_state *ps)
{
    Seq *s = (Seq *)malloc(sizeof(Seq));
    if (!s) return -1;
    ps->cap = 1 << 16;
    s->seq = (char *)malloc(ps->cap);
    if (!s->seq) { free(s); return -1; }
    s->len = 0;
    s->id[0] = '\0';
    ps->cur = s;
    ps->state = 1;
    ps->id_len = 0;
    return 0;
}

static int feed(Genome *g, parse_state *ps, const unsigned char *p, size_t n)
{
    size_t i = 0;
    while (i < n) {
        if (ps->state == 0) { /* body: the hot loop */
            Seq *s = ps->cur;
            if (!s) { /* very first byte of the file */
                if (p[i] != '>') { ps->bad = 1; return -1; } /* precondition: starts with '>' */
                if (begin_contig(ps)) return -1;
                i++;
                continue;
            }
            while (i < n) {
                unsigned char c = p[i];
                unsigned char k = g_class[c];
                if (k == C_BASE) {
                    if (s->len + 2 > ps->cap) {
                        size_t want = ps->cap * 2;
                        char *nb;
                        if (want > (size_t)MAX_SEQ_LEN + 1) want = (size_t)MAX_SEQ_LEN + 1;
                        nb = (char *)realloc(s->seq, want);
                        if (!nb) return -1;
                        s->seq = nb;
                        ps->cap = want;
                    }
                    if (s->len == (size_t)MAX_SEQ_LEN) { /* fasta-genome-io.c:120-122,:140-142 */
                        fprintf(stderr, "%s is truncated to %d\n", s->id, MAX_SEQ_LEN);
                        ps->state = 3;
                        break;
                    }
                    s->seq[s->len++] = (char)g_upper[c];
                    i++;
                } else if (k == C_SPACE) {
                    i++;
                } else { /* '>' */
                    break;
                }
            }
            if (i < n && ps->state == 0) { /* stopped on '>' */
                finish_contig(g, ps);
                if (begin_contig(ps)) return -1;
                i++;
            }
        } else if (ps->state == 1) { /* id */
            while (i < n && !g_class[p[i]]) { /* C_BASE only: '>' inside an id is an ordinary byte */
                if (ps->id_len < MAX_ID_LEN) ps->cur->id[ps->id_len++] = (char)p[i];
                else ps->bad = 1; /* precondition: ids <= MAX_ID_LEN */
                i++;
            }
            if (i < n) {
                if (p[i] == '>' ) { /* isspace('>') is false: part of the id */
                    if (ps->id_len < MAX_ID_LEN) ps->cur->id[ps->id_len++] = '>';
                    i++;
                    continue;
                }
                ps->cur->id[ps->id_len] = '\0';
                ps->state = 2; /* p[i] is white space; a '\n' is consumed by state 2 */
            }
        } else if (ps->state == 2) { /* rest of the header line */
            const unsigned char *nl = (const unsigned char *)memchr(p + i, '\n', n - i);
            if (!nl) { i = n; break; }
            i = (size_t)(nl - p) + 1;
            ps->state = 0;
        } else { /* 3: drop the tail of an over-long contig */
            const unsigned char *gt = (const unsigned char *)memchr(p + i, '>', n - i);
            if (!gt) { i = n; break; }
            i = (size_t)(gt - p);
            ps->state = 0;
        }
    }
    return 0;
}

Genome *init_genome(const char fn[])
{
    Genome *genome;
    blk_src src;
    parse_state ps;
    unsigned char *buf;
    const size_t BLK = 4u << 20;
    long got;

    init_tables();
    if (fn == NULL) return NULL;
    memset(&src, 0, sizeof src);
    src.gz = is_gz(fn);
    if (src.gz) {
        src.zf = gzopen(fn, "rb");
        if (!src.zf) {
            fprintf(stderr, "%s\n", fn);
            perror("Cannot open file");
            return NULL;
        }
        gzbuffer(src.zf, 1u << 20);
    } else {
        src.fd = open(fn, O_RDONLY);
        if (src.fd < 0) {
            fprintf(stderr, "%s\n", fn);
            perror("Cannot open file");
            return NULL;
        }
#ifdef POSIX_FADV_SEQUENTIAL
        (void)posix_fadvise(src.fd, 0, 0, POSIX_FADV_SEQUENTIAL);
#endif
    }
    genome = (Genome *)malloc(sizeof(Genome));
    genome->seqs = (Seq **)malloc(sizeof(Seq *) * MAX_GENOME_SEQS);
    genome->dummy = (Seq *)calloc(1, sizeof(Seq));
    genome->n_seqs = 0;
    buf = (unsigned char *)malloc(BLK);
    memset(&ps, 0, sizeof ps);
    while ((got = blk_read(&src, buf, BLK)) > 0) {
        if (feed(genome, &ps, buf, (size_t)got)) break;
    }
    if (ps.cur && (ps.state == 1 || ps.state == 2)) {
        /* header line without '\n' at EOF: the reference never returns from this */
        ps.cur->id[ps.id_len < MAX_ID_LEN ? ps.id_len : MAX_ID_LEN] = '\0';
        ps.bad = 1;
    }
    finish_contig(genome, &ps);
    free(buf);
    if (src.gz) gzclose(src.zf); else close(src.fd);
    if (ps.bad) {
        fprintf(stderr, "%s: malformed FASTA (must start with '>', header lines end in newline, ids <= %d)\n",
                fn, MAX_ID_LEN);
        destroy_genome(genome);
        return NULL;
    }
    qsort(genome->seqs, genome->n_seqs, sizeof(Seq *), chr_cmp);
    return genome;
}

/* ------------------------------------------------------------------------------------ */
/* record-at-a-time API (kept for callers of the reference interface)                    */
/* ------------------------------------------------------------------------------------ */

Fa_Src *init_fasta_src(const char fn[])
{
    Fa_Src *fs;
    if (fn == NULL) return NULL;
    if (strlen(fn) > MAX_FN_LEN) return NULL;
    fs = (Fa_Src *)calloc(1, sizeof(Fa_Src));
    if (!fs) return NULL;
    strcpy(fs->fn, fn);
    fs->seq_buffer = (char *)malloc((size_t)MAX_SEQ_LEN + 1);
    if (!fs->seq_buffer) { free(fs); return NULL; }
    fs->seq_buffer[0] = '\0';
    fs->is_gz = is_gz(fn);
    if (fs->is_gz) {
        fs->fagz = gzopen(fs->fn, "r");
        if (fs->fagz == NULL) { free(fs->seq_buffer); free(fs); return NULL; }
    } else {
        fs->fafp = fileOpen(fs->fn, "r");
        if (fs->fafp == NULL) { free(fs->seq_buffer); free(fs); return NULL; }
    }
    return fs;
}

int close_fasta_src(Fa_Src *fs)
{
    if (!fs) return 0;
    if (fs->is_gz) gzclose(fs->fagz); else fclose(fs->fafp);
    free(fs->seq_buffer);
    free(fs);
    return 0;
}

/* one record from a byte getter; shared by the FILE* and gzFile front ends */
typedef int (*getc_fn)(void *h);
typedef void (*ungetc_fn)(int c, void *h);

static int read_record(getc_fn get, ungetc_fn unget, void *h, Seq *seq, char *seq_buffer)
{
    size_t i = 0;
    int c = get(h);
    init_tables();
    if (c == EOF) return -1;
    if (c != '>') return -2; /* precondition: a record starts with '>' */
    c = get(h);
    while (c != EOF && !isspace(c)) {
        if (i < MAX_ID_LEN) seq->id[i++] = (char)c;
        c = get(h);
    }
    seq->id[i] = '\0';
    while (c != EOF && c != '\n') c = get(h);
    i = 0;
    if (c != EOF) c = get(h);
    while (c != '>' && c != EOF && i < (size_t)MAX_SEQ_LEN) {
        if (!g_class[c & 0xFF]) seq_buffer[i++] = (char)g_upper[c & 0xFF];
        c = get(h);
    }
    seq_buffer[i] = '\0';
    if (i == (size_t)MAX_SEQ_LEN) {
        fprintf(stderr, "%s is truncated to %d\n", seq->id, MAX_SEQ_LEN);
        while (c != '>' && c != EOF) c = get(h); /* skip the tail of this contig */
    }
    if (c != EOF) unget(c, h);
    seq->seq = (char *)malloc(i + 1);
    if (!seq->seq) return -3;
    memcpy(seq->seq, seq_buffer, i + 1);
    seq->len = i;
    return 0;
}

static int f_get(void *h) { return getc_unlocked((FILE *)h); }
static void f_unget(int c, void *h) { ungetc(c, (FILE *)h); }
static int z_get(void *h) { return gzgetc((gzFile)h); }
static void z_unget(int c, void *h) { gzungetc(c, (gzFile)h); }

int read_fasta(FILE *fafp, Seq *seq, char *seq_buffer)
{
    return read_record(f_get, f_unget, fafp, seq, seq_buffer);
}

int gzread_fasta(gzFile gzfp, Seq *seq, char *seq_buffer)
{
    return read_record(z_get, z_unget, gzfp, seq, seq_buffer);
}

Seq *get_next_fa(Fa_Src *fa_source, Genome *genome)
{
    Seq *seq;
    int status;
    if (!fa_source || !genome || genome->n_seqs >= MAX_GENOME_SEQS) return NULL;
    seq = (Seq *)malloc(sizeof(Seq));
    if (!seq) return NULL;
    status = fa_source->is_gz ? gzread_fasta(fa_source->fagz, seq, fa_source->seq_buffer)
                              : read_fasta(fa_source->fafp, seq, fa_source->seq_buffer);
    if (status) {
        free(seq);
        return NULL;
    }
    fa_source->n++;
    genome->seqs[genome->n_seqs++] = seq;
    return seq;
}

int destroy_seq(Seq *seq)
{
    if (!seq) return 0;
    free(seq->seq);
    free(seq);
    return 0;
}

int destroy_genome(Genome *genome)
{
    if (!genome) return 0;
    for (size_t i = 0; i < genome->n_seqs; i++) destroy_seq(genome->seqs[i]);
    free(genome->seqs);
    free(genome->dummy);
    free(genome);
    return 0;
}
