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
 * Plain-text files are loaded by several threads (load_parallel below): the mapped file is
 * cut at line starts -- every line start is in "body" state, whatever came before -- each
 * piece is scanned for headers and base counts, the contigs are sized, and the pieces are
 * compacted straight into place.  Same bytes out as the one-thread parser, which remains
 * the path for .gz input, unmappable files and the over-long-contig corner.
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
#include "inflate_fast.h"

#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdatomic.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

/* A contig's bases.  Big ones ask for transparent huge pages (the box runs THP in "madvise" mode):
 * a 3.1 Gb genome is 800 000 first-touch faults in 4 KiB pages and 1 500 in 2 MiB pages -- faster
 * to fill by the parser threads, to pin for the upload, and to give back at exit.  free()-able, as
 * destroy_seq expects (fasta-genome-io.c:241-248). */
static char *contig_alloc(size_t bytes)
{
    const size_t huge = (size_t)2 << 20;
    if (bytes >= 4 * huge) {
        const size_t rounded = (bytes + huge - 1) & ~(huge - 1);
        void *p = NULL;
        if (posix_memalign(&p, huge, rounded) == 0) {
#ifdef MADV_HUGEPAGE
            (void)madvise(p, rounded, MADV_HUGEPAGE);
#endif
            return (char *)p;
        }
    }
    return (char *)malloc(bytes);
}

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

/* ------------------------------------------------------------------------------------ */
/* multi-threaded loader for plain-text files                                            */
/* ------------------------------------------------------------------------------------ */

typedef struct {
    size_t hdr;              /* offset of the '>' that opens this segment; (size_t)-1 = the
                                piece's leading bytes, which continue the previous contig   */
    size_t body_a, body_b;   /* body bytes of the segment within the file                   */
    size_t n_bases;          /* non-white-space bytes in it                                 */
    char *dst;               /* where they go (set when the contigs are sized)              */
    char id[MAX_ID_LEN + 1];
} fa_seg;

typedef struct {
    const unsigned char *data;
    size_t a, b;             /* this piece: [a, b), a is a line start                      */
    fa_seg *seg;
    size_t n_seg, cap_seg;
    int bad, open_header;    /* format violation; header line cut by the end of the file    */
} fa_piece;

typedef struct {
    fa_piece *pc;
    size_t n_pc;
    size_t next;             /* shared cursor, guarded by mu */
    pthread_mutex_t mu;
    int pass;                /* 0 scan, 1 copy */
} fa_job;

static fa_seg *piece_add_seg(fa_piece *pc)
{
    if (pc->n_seg == pc->cap_seg) {
        size_t cap = pc->cap_seg ? pc->cap_seg * 2 : 8;
        fa_seg *ns = (fa_seg *)realloc(pc->seg, cap * sizeof(fa_seg));
        if (!ns) return NULL;
        pc->seg = ns;
        pc->cap_seg = cap;
    }
    fa_seg *sg = &pc->seg[pc->n_seg++];
    memset(sg, 0, sizeof *sg);
    return sg;
}

/* pass 0: headers and base counts of one piece */
static void piece_scan(fa_piece *pc)
{
    const unsigned char *p = pc->data;
    size_t i = pc->a;
    fa_seg *sg = piece_add_seg(pc);
    if (!sg) { pc->bad = 1; return; }
    sg->hdr = (size_t)-1;
    sg->body_a = i;
    while (i < pc->b) {
        /* body: count bases up to the next '>' */
        size_t n = 0;
        while (i < pc->b) {
            const unsigned char k = g_class[p[i]];
            if (k == C_GT) break;
            n += k == C_BASE;
            i++;
        }
        sg->n_bases = n;
        sg->body_b = i;
        if (i == pc->b) break;
        /* header line: id up to the first white-space byte, then the rest of the line */
        sg = piece_add_seg(pc);
        if (!sg) { pc->bad = 1; return; }
        sg->hdr = i++;
        size_t id_len = 0;
        while (i < pc->b && !isspace(p[i])) {
            if (id_len < MAX_ID_LEN) sg->id[id_len++] = (char)p[i];
            else pc->bad = 1; /* precondition: ids <= MAX_ID_LEN */
            i++;
        }
        sg->id[id_len] = '\0';
        const unsigned char *nl = i < pc->b ? (const unsigned char *)memchr(p + i, '\n', pc->b - i) : NULL;
        if (!nl) { pc->open_header = 1; sg->body_a = sg->body_b = pc->b; return; }
        i = (size_t)(nl - p) + 1;
        sg->body_a = sg->body_b = i;
    }
}

/* pass 1: compact + upper-case every segment of one piece into place */
static void piece_copy(const fa_piece *pc)
{
    for (size_t k = 0; k < pc->n_seg; k++) {
        const fa_seg *sg = &pc->seg[k];
        char *d = sg->dst;
        if (!d) continue;
        for (size_t i = sg->body_a; i < sg->body_b; i++) {
            const unsigned char c = pc->data[i];
            if (g_class[c] == C_BASE) *d++ = (char)g_upper[c];
        }
    }
}

static void *fa_worker(void *arg)
{
    fa_job *job = (fa_job *)arg;
    for (;;) {
        pthread_mutex_lock(&job->mu);
        const size_t k = job->next++;
        pthread_mutex_unlock(&job->mu);
        if (k >= job->n_pc) break;
        if (job->pass == 0) piece_scan(&job->pc[k]);
        else piece_copy(&job->pc[k]);
    }
    return NULL;
}

static void fa_run(fa_job *job, int pass, int n_threads)
{
    pthread_t th[64];
    int started = 0;
    job->pass = pass;
    job->next = 0;
    for (int t = 0; t < n_threads - 1 && t < 64; t++)
        if (pthread_create(&th[started], NULL, fa_worker, job) == 0) started++;
    fa_worker(job);
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
}

static int fasta_threads(void)
{
    int n_threads = 16;
    const char *ev = getenv("PSSBAM_FASTA_THREADS");
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    if (ev) n_threads = atoi(ev);
    if (ncpu > 0 && n_threads > ncpu) n_threads = (int)ncpu;
    return n_threads;
}

static int parse_parallel(const unsigned char *data, size_t size, const char fn[], int n_threads, Genome **out);

/* Returns 1 = loaded into *out (NULL there = malformed input, already reported),
 * 0 = not applicable, use the one-thread parser. */
static int load_parallel(const char fn[], Genome **out)
{
    const int n_threads = fasta_threads();
    if (n_threads < 2) return 0;

    int fd = open(fn, O_RDONLY);
    struct stat st;
    if (fd < 0) return 0; /* the serial path reports it */
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < (off_t)(1 << 20)) { close(fd); return 0; }
    const size_t size = (size_t)st.st_size;
    const unsigned char *data = (const unsigned char *)mmap(NULL, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (data == (const unsigned char *)MAP_FAILED) return 0;
    (void)madvise((void *)data, size, MADV_SEQUENTIAL);
    const int rc = parse_parallel(data, size, fn, n_threads, out);
    munmap((void *)data, size);
    return rc;
}

/* the text of a whole FASTA file in memory -> contigs (same return convention) */
static int parse_parallel(const unsigned char *data, size_t size, const char fn[], int n_threads, Genome **out)
{
    int applicable = 1, bad = 0;
    Genome *genome = NULL;
    fa_job job;
    memset(&job, 0, sizeof job);
    pthread_mutex_init(&job.mu, NULL);
    /* pieces of 64 KiB .. 8 MiB (a few per thread), each starting right behind a '\n' */
    size_t want = size / (4u * (size_t)n_threads);
    if (want > (8u << 20)) want = 8u << 20;
    if (want < (64u << 10)) want = 64u << 10;
    job.pc = (fa_piece *)calloc(size / want + 2, sizeof(fa_piece));
    if (!job.pc) { applicable = 0; goto out; }
    for (size_t a = 0; a < size;) {
        size_t b = a + want;
        if (b >= size) b = size;
        else {
            const unsigned char *nl = (const unsigned char *)memchr(data + b, '\n', size - b);
            b = nl ? (size_t)(nl - data) + 1 : size;
        }
        job.pc[job.n_pc].data = data;
        job.pc[job.n_pc].a = a;
        job.pc[job.n_pc].b = b;
        job.n_pc++;
        a = b;
    }
    if (data[0] != '>') bad = 1; /* precondition: starts with '>' */
    if (!bad) fa_run(&job, 0, n_threads);

    /* size the contigs: segments in file order; a leading segment continues the open contig */
    genome = (Genome *)calloc(1, sizeof(Genome));
    if (!genome) { applicable = 0; goto out; }
    genome->seqs = (Seq **)malloc(sizeof(Seq *) * MAX_GENOME_SEQS);
    genome->dummy = (Seq *)calloc(1, sizeof(Seq));
    if (!genome->seqs || !genome->dummy) { applicable = 0; goto out; }
    genome->n_seqs = 0;
    for (int round = 0; round < 2 && !bad && applicable; round++) {
        /* round 0 adds up lengths, round 1 (after allocation) hands out destinations */
        Seq *cur = NULL;
        size_t idx = 0, fill = 0;
        for (size_t k = 0; k < job.n_pc && !bad; k++) {
            fa_piece *pc = &job.pc[k];
            if (round == 0 && (pc->bad || (pc->open_header))) { bad = 1; break; }
            for (size_t q = 0; q < pc->n_seg; q++) {
                fa_seg *sg = &pc->seg[q];
                if (sg->hdr != (size_t)-1) {
                    if (round == 0) {
                        if (genome->n_seqs >= MAX_GENOME_SEQS) { applicable = 0; break; } /* rare: let the serial parser decide */
                        cur = (Seq *)calloc(1, sizeof(Seq));
                        if (!cur) { applicable = 0; break; }
                        strcpy(cur->id, sg->id);
                        genome->seqs[genome->n_seqs++] = cur;
                    } else {
                        cur = genome->seqs[idx++];
                        fill = 0;
                    }
                }
                if (!cur) continue; /* the empty lead-in of the first piece (the file starts with '>') */
                if (round == 0) cur->len += sg->n_bases;
                else { sg->dst = cur->seq + fill; fill += sg->n_bases; }
            }
            if (!applicable) break;
        }
        if (round == 0 && !bad && applicable) {
            for (size_t c = 0; c < genome->n_seqs; c++) {
                Seq *sq = genome->seqs[c];
                if (sq->len > (size_t)MAX_SEQ_LEN) { applicable = 0; break; } /* truncation corner: serial parser */
                sq->seq = contig_alloc(sq->len + 1);
                if (!sq->seq) { applicable = 0; break; }
                sq->seq[sq->len] = '\0';
            }
        }
    }
    if (!bad && applicable) fa_run(&job, 1, n_threads);

out:
    for (size_t k = 0; k < job.n_pc; k++) free(job.pc[k].seg);
    free(job.pc);
    pthread_mutex_destroy(&job.mu);
    if (!applicable) {
        if (genome) destroy_genome(genome);
        return 0;
    }
    if (bad) {
        fprintf(stderr, "%s: malformed FASTA (must start with '>', header lines end in newline, ids <= %d)\n",
                fn, MAX_ID_LEN);
        destroy_genome(genome);
        *out = NULL;
        return 1;
    }
    qsort(genome->seqs, genome->n_seqs, sizeof(Seq *), chr_cmp);
    *out = genome;
    return 1;
}

/* ------------------------------------------------------------------------------------ */
/* .gz files written by bgzip (BGZF: what `samtools faidx` wants a compressed reference   */
/* to be): independent <= 64 KiB deflate blocks, inflated by all threads at once          */
/* ------------------------------------------------------------------------------------ */
typedef struct { size_t in_off; uint32_t in_len, isize, crc; size_t out_off; } fa_bgzf_block;

typedef struct {
    const unsigned char *data;
    unsigned char *text;
    const fa_bgzf_block *blk;
    size_t n_blk, next;
    pthread_mutex_t mu;
    atomic_int bad;   /* set by any worker, polled by all of them */
} fa_bgzf_job;

static void *fa_bgzf_worker(void *arg)
{
    fa_bgzf_job *job = (fa_bgzf_job *)arg;
    pss_inflater *st = (pss_inflater *)malloc(sizeof *st);
    if (!st) { atomic_store(&job->bad, 1); return NULL; }
    for (;;) {
        pthread_mutex_lock(&job->mu);
        const size_t k0 = job->next;
        job->next += 64;
        pthread_mutex_unlock(&job->mu);
        if (k0 >= job->n_blk || atomic_load(&job->bad)) break;
        for (size_t k = k0; k < k0 + 64 && k < job->n_blk; k++) {
            const fa_bgzf_block *b = &job->blk[k];
            if (!b->isize) continue;
            if (pss_inflate_raw(st, job->data + b->in_off, b->in_len, job->text + b->out_off, b->isize) != 0 ||
                pss_crc32(0, job->text + b->out_off, b->isize) != b->crc) { atomic_store(&job->bad, 1); break; }
        }
    }
    free(st);
    return NULL;
}

/* 1 = handled (*out set, NULL = malformed and reported), 0 = not a BGZF file / not applicable: gzread copes */
static int load_bgzf_parallel(const char fn[], Genome **out)
{
    const int n_threads = fasta_threads();
    if (n_threads < 2) return 0;
    int fd = open(fn, O_RDONLY);
    struct stat sb;
    if (fd < 0) return 0;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 28) { close(fd); return 0; }
    const size_t size = (size_t)sb.st_size;
    const unsigned char *data = (const unsigned char *)mmap(NULL, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (data == (const unsigned char *)MAP_FAILED) return 0;
    int rc = 0;
    fa_bgzf_block *blk = NULL;
    size_t n_blk = 0, cap = 0, total = 0, o = 0;
    unsigned char *text = NULL;
    /* every gzip member must be a BGZF block: FEXTRA with a 'B','C' subfield of two bytes (SAM spec 4.1) */
    while (o < size) {
        if (size - o < 18 || data[o] != 0x1f || data[o + 1] != 0x8b || data[o + 2] != 8 || !(data[o + 3] & 4)) goto done;
        const size_t xlen = (size_t)data[o + 10] | ((size_t)data[o + 11] << 8);
        if (size - o < 12 + xlen + 8) goto done;
        size_t bsize = 0, x = o + 12;
        while (x + 4 <= o + 12 + xlen) {
            const size_t sl = (size_t)data[x + 2] | ((size_t)data[x + 3] << 8);
            if (data[x] == 'B' && data[x + 1] == 'C' && sl == 2 && x + 6 <= o + 12 + xlen) bsize = ((size_t)data[x + 4] | ((size_t)data[x + 5] << 8)) + 1;
            x += 4 + sl;
        }
        if (!bsize || bsize < 12 + xlen + 8 || bsize > size - o || (data[o + 3] & ~4)) goto done; /* (other header flags: not what bgzip writes) */
        if (n_blk == cap) {
            cap = cap ? cap * 2 : 4096;
            fa_bgzf_block *nb = (fa_bgzf_block *)realloc(blk, cap * sizeof *nb);
            if (!nb) goto done;
            blk = nb;
        }
        fa_bgzf_block *b = &blk[n_blk++];
        b->in_off = o + 12 + xlen;
        b->in_len = (uint32_t)(bsize - 12 - xlen - 8);
        memcpy(&b->crc, data + o + bsize - 8, 4);
        memcpy(&b->isize, data + o + bsize - 4, 4);
        if (b->isize > 65536u) goto done;
        b->out_off = total;
        total += b->isize;
        o += bsize;
    }
    if (total < ((size_t)1 << 20)) goto done; /* small: the serial reader is as good */
    text = (unsigned char *)contig_alloc(total + 1);
    if (!text) goto done;
    {
        fa_bgzf_job job;
        memset(&job, 0, sizeof job);
        job.data = data;
        job.text = text;
        job.blk = blk;
        job.n_blk = n_blk;
        pthread_mutex_init(&job.mu, NULL);
        pthread_t th[64];
        int started = 0;
        for (int t = 0; t < n_threads - 1 && t < 64; t++)
            if (pthread_create(&th[started], NULL, fa_bgzf_worker, &job) == 0) started++;
        fa_bgzf_worker(&job);
        for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
        pthread_mutex_destroy(&job.mu);
        if (atomic_load(&job.bad)) {
            fprintf(stderr, "%s: damaged BGZF block (inflate / CRC-32 check failed)\n", fn);
            *out = NULL;
            rc = 1;
            goto done;
        }
    }
    rc = parse_parallel(text, total, fn, n_threads, out);
done:
    free(text); /* (contig_alloc memory is free()-able) */
    free(blk);
    munmap((void *)data, size);
    return rc;
}

/* Plain gzip: the stream is serial, but it need not be parsed a block at a time by the thread that inflates it --
 * the whole text goes into memory through zlib (the file's ISIZE trailer sizes the buffer when the file has one
 * member) and the multi-threaded parser takes it from there.  0 = not applicable (small, unreadable): gzread path. */
static int load_gzip_whole(const char fn[], Genome **out)
{
    const int n_threads = fasta_threads();
    if (n_threads < 2) return 0;
    struct stat sb;
    if (stat(fn, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 18) return 0;
    uint32_t isize = 0;
    {
        FILE *f = fopen(fn, "rb");
        if (!f) return 0;
        if (fseeko(f, -4, SEEK_END) != 0 || fread(&isize, 1, 4, f) != 4) isize = 0;
        fclose(f);
    }
    /* One member whose trailer says how much it inflates to (the usual `gzip ref.fa`): the block decoder of the BAM
     * reader (host/inflate_fast.c) takes the whole DEFLATE stream in one call, several times zlib's speed; anything
     * it does not like -- more members, > 4 GiB, a damaged stream -- goes through zlib below. */
    if (isize >= (1u << 20)) {
        int fd = open(fn, O_RDONLY);
        const size_t size = (size_t)sb.st_size;
        const unsigned char *data = fd >= 0 ? (const unsigned char *)mmap(NULL, size, PROT_READ, MAP_PRIVATE, fd, 0) : (const unsigned char *)MAP_FAILED;
        if (fd >= 0) close(fd);
        if (data != (const unsigned char *)MAP_FAILED) {
            size_t o = 10;
            int ok = size > 18 && data[0] == 0x1f && data[1] == 0x8b && data[2] == 8 && !(data[3] & 0xE0);
            const unsigned flg = ok ? data[3] : 0;
            if (ok && (flg & 4)) { ok = o + 2 <= size; if (ok) o += 2 + ((size_t)data[o] | ((size_t)data[o + 1] << 8)); }
            for (int fld = 0; ok && fld < 2; fld++)   /* FNAME, FCOMMENT: zero-terminated */
                if (flg & (fld ? 16u : 8u)) {
                    while (o < size && data[o]) o++;
                    o++;
                }
            if (ok && (flg & 2)) o += 2;
            ok = ok && o + 8 < size;
            unsigned char *text = ok ? (unsigned char *)contig_alloc((size_t)isize + 1) : NULL;
            pss_inflater *st = text ? (pss_inflater *)malloc(sizeof *st) : NULL;
            int rc = 0, done = 0;
            if (st) {
                uint32_t crc;
                memcpy(&crc, data + size - 8, 4);
                if (pss_inflate_raw(st, data + o, size - 8 - o, text, isize) == 0 && pss_crc32(0, text, isize) == crc) {
                    rc = parse_parallel(text, isize, fn, n_threads, out);
                    done = 1;
                }
            }
            free(st);
            free(text);
            munmap((void *)data, size);
            if (done) return rc;
        }
    }
    gzFile zf = gzopen(fn, "rb");
    if (!zf) return 0;
    gzbuffer(zf, 1u << 20);
    size_t cap = (size_t)isize + 1, len = 0;   /* (right for one member; FASTA text deflates 3.5-4.5x: room for several) */
    if (cap < 5 * (size_t)sb.st_size) cap = 5 * (size_t)sb.st_size;
    if (cap < ((size_t)64 << 20)) cap = (size_t)64 << 20;
    unsigned char *text = (unsigned char *)contig_alloc(cap);
    int rc = 0, bad = 0;
    if (!text) { gzclose(zf); return 0; }
    for (;;) {
        if (len == cap) {
            const size_t ncap = cap + cap / 2;
            unsigned char *nt = (unsigned char *)contig_alloc(ncap);
            if (!nt) { bad = 1; break; }
            memcpy(nt, text, len);
            free(text);
            text = nt;
            cap = ncap;
        }
        const size_t want = cap - len < ((size_t)256 << 20) ? cap - len : ((size_t)256 << 20);
        const int got = gzread(zf, text + len, (unsigned)want);
        if (got < 0) { bad = 1; break; }   /* damaged stream: the block-at-a-time path below deals with it as before */
        if (got == 0) break;
        len += (size_t)got;
    }
    gzclose(zf);
    if (!bad && len >= ((size_t)1 << 20)) rc = parse_parallel(text, len, fn, n_threads, out);
    free(text);
    return rc;
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
    if (!src.gz && load_parallel(fn, &genome)) return genome;
    if (src.gz && load_bgzf_parallel(fn, &genome)) return genome;
    if (src.gz && load_gzip_whole(fn, &genome)) return genome;
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
