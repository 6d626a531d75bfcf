/*
 * pss-bam_amd/host/sam_reader.c -- see sam_reader.h.
 *
 * Field rules are line2saml's (include/sam-parse.h, host/samline.c): eleven white-space
 * separated tokens, numeric fields with strtoul/strtol semantics, SEQ and QUAL of equal
 * length.  Encoding rules (text -> BAM, SAM spec 4.2), chosen so the device-side decode gives
 * back the same text-level facts the reference would act on:
 *   RNAME  -> id in a growing name table ('*' -> -1)
 *   POS    -> pos = POS-1, clamped into int32 (anything that large is filtered by both tools)
 *   MAPQ   -> one byte, clamped to 255 (SAM spec range)
 *   CIGAR  -> ops only when the text is canonical ("<n><op>..." without leading zeros);
 *             any other text becomes an empty CIGAR: the reference compares CIGAR as a
 *             string with "<L>M" (pss-bam.c:113-123), a non-canonical text never matches
 *   SEQ    -> 4-bit codes, case folded (process_aln upper-cases, pss-bam.c:425); characters
 *             outside "=ACMGRSVTWYHKDBN" become N; "*" -> l_seq 0
 *   QUAL   -> phred bytes; "*" -> 0xFF fill
 *   RG:Z:  -> kept as aux (the only tag the engine ever looks at); other tags dropped
 *
 * Machinery: the text of a batch (a mapped plain file, or a slab gunzipped by one thread) is cut
 * at line starts into one piece per thread; every thread tokenises and encodes its lines into a
 * private buffer; the pieces are then laid end to end in the batch buffer, in file order.
 * RNAMEs that no @SQ line announced are entered into the name table in a short serial pass, so
 * ids stay "order of first appearance".  Lines longer than MAX_LINE_LEN are cut the way the
 * reference's fgets(saml_buf, MAX_LINE_LEN + 1) cuts them (pss-bam.c:764).
 */
#include "sam_reader.h"

#include <ctype.h>
#include <errno.h>
#include <fcntl.h>
#include <limits.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "sam-parse.h"

#define SAM_MAX_THREADS 64
#define REF_UNKNOWN (-2) /* placeholder refID of a record whose RNAME is not in the table yet */

typedef struct {
    size_t rec_off;      /* record start within the piece's output */
    char *name;
} unknown_t;

typedef struct sam_piece {
    struct sam_reader *r;
    const char *a, *b;   /* text of this piece: whole lines */
    uint8_t *out;        /* private output, cap bytes */
    size_t cap, len;
    uint32_t *rec_len;   /* length of every record written */
    size_t n, n_cap;
    unknown_t *unk;
    size_t n_unk, unk_cap;
    uint64_t skipped;
    size_t base;         /* where the piece's output goes in the batch buffer */
    int failed;
    Saml *sp;            /* scratch (20 KB) */
    char *line;          /* scratch: one NUL-terminated line, MAX_LINE_LEN + 2 */
} sam_piece;

struct sam_reader {
    /* input: a mapped plain-text file, or a gz stream read slab by slab into tbuf */
    gzFile f;
    const char *map;
    size_t map_len, map_pos;
    char *tbuf;          /* gz path: text slab; [0, tlen) valid, the tail may be a partial line */
    size_t tcap, tlen, tpos;
    int in_header;
    char **names;
    int32_t n_names, names_cap;
    uint8_t *buf;
    size_t cap, len;
    uint32_t *offs;
    size_t offs_cap;
    uint64_t skipped;
    int eof;
    int n_threads;
    sam_piece piece[SAM_MAX_THREADS];
    char err[256];
};

static void init_seq_codes(void);

static void set_err(sam_reader *r, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(r->err, sizeof r->err, fmt, ap);
    va_end(ap);
}

int file_is_bam(const char *path)
{
    unsigned char hd[18];
    FILE *f = fopen(path, "rb");
    size_t n;
    if (!f) return -1;
    n = fread(hd, 1, sizeof hd, f);
    fclose(f);
    /* BGZF = gzip member with FEXTRA and a 'BC' subfield */
    return n == 18 && hd[0] == 0x1f && hd[1] == 0x8b && hd[2] == 8 && (hd[3] & 4) && hd[12] == 'B' && hd[13] == 'C';
}

/* read-only lookup for the parsing threads; REF_UNKNOWN when the name is not in the table */
static int32_t name_lookup(const sam_reader *r, const char *name, int32_t *hint)
{
    if (name[0] == '*' && name[1] == '\0') return -1;
    if (*hint >= 0 && *hint < r->n_names && strcmp(r->names[*hint], name) == 0) return *hint; /* sorted input */
    for (int32_t i = 0; i < r->n_names; i++)
        if (strcmp(r->names[i], name) == 0) return *hint = i;
    return REF_UNKNOWN;
}

static int32_t name_id(sam_reader *r, const char *name)
{
    if (name[0] == '*' && name[1] == '\0') return -1;
    for (int32_t i = r->n_names - 1; i >= 0; i--) /* recent names first: sorted input stays on one */
        if (strcmp(r->names[i], name) == 0) return i;
    if (r->n_names == r->names_cap) {
        r->names_cap = r->names_cap ? r->names_cap * 2 : 64;
        r->names = (char **)realloc(r->names, (size_t)r->names_cap * sizeof(char *));
    }
    r->names[r->n_names] = strdup(name);
    return r->n_names++;
}

sam_reader *sam_reader_open(const char *path, size_t batch_bytes, char *err, size_t errlen)
{
    sam_reader *r = (sam_reader *)calloc(1, sizeof *r);
    if (!r) return NULL;
    init_seq_codes(); /* idempotent; readers are opened from one thread */
    /* plain text in a regular file is mapped; everything else goes through zlib's gz layer
     * (which also passes plain text through) */
    {
        unsigned char magic[2] = {0, 0};
        struct stat st;
        int fd = open(path, O_RDONLY);
        if (fd < 0) {
            if (err) snprintf(err, errlen, "cannot open %s: %s", path, strerror(errno));
            free(r);
            return NULL;
        }
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0 && pread(fd, magic, 2, 0) == 2 &&
            !(magic[0] == 0x1f && magic[1] == 0x8b)) {
            void *m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
                r->map = (const char *)m;
                r->map_len = (size_t)st.st_size;
            }
        }
        close(fd);
    }
    if (!r->map) {
        r->f = gzopen(path, "rb");
        if (!r->f) {
            if (err) snprintf(err, errlen, "cannot open %s: %s", path, strerror(errno));
            free(r);
            return NULL;
        }
        gzbuffer(r->f, 1u << 20);
    }
    if (!batch_bytes && getenv("PSSBAM_BATCH_BYTES")) batch_bytes = (size_t)strtoull(getenv("PSSBAM_BATCH_BYTES"), NULL, 10);
    r->cap = batch_bytes ? batch_bytes : (size_t)256 << 20;
    if (r->cap < ((size_t)1 << 20)) r->cap = (size_t)1 << 20;
    if (r->cap > ((size_t)2 << 30)) r->cap = (size_t)2 << 30; /* record offsets are 32-bit */
    r->buf = (uint8_t *)malloc(r->cap + 4096);
    {
        long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
        const char *ev = getenv("PSSBAM_SAM_THREADS");
        int t = ev ? atoi(ev) : 16;
        if (ncpu > 0 && t > ncpu) t = (int)ncpu;
        r->n_threads = t < 1 ? 1 : (t > SAM_MAX_THREADS ? SAM_MAX_THREADS : t);
    }
    if (!r->map) {
        /* a text slab must hold at least one line and its encoding must fit the batch buffer:
         * a record is at most twice its text (4-byte CIGAR ops from two characters) */
        r->tcap = r->cap / 2 > (size_t)MAX_LINE_LEN + 2 ? r->cap / 2 : (size_t)MAX_LINE_LEN + 2;
        r->tbuf = (char *)malloc(r->tcap + 1);
    }
    r->in_header = 1;
    if (!r->buf || (!r->map && !r->tbuf)) {
        if (err) snprintf(err, errlen, "out of memory");
        sam_reader_close(r);
        return NULL;
    }
    return r;
}

int32_t sam_reader_n_ref(const sam_reader *r) { return r->n_names; }
const char *const *sam_reader_ref_names(const sam_reader *r) { return (const char *const *)r->names; }
uint64_t sam_reader_lines_skipped(const sam_reader *r) { return r->skipped; }
const char *sam_reader_error(const sam_reader *r) { return r->err; }

void sam_reader_close(sam_reader *r)
{
    if (!r) return;
    if (r->f) gzclose(r->f);
    if (r->map) munmap((void *)r->map, r->map_len);
    for (int32_t i = 0; i < r->n_names; i++) free(r->names[i]);
    free(r->names);
    free(r->buf);
    free(r->offs);
    free(r->tbuf);
    for (int t = 0; t < SAM_MAX_THREADS; t++) {
        sam_piece *pc = &r->piece[t];
        free(pc->out);
        free(pc->rec_len);
        for (size_t k = 0; k < pc->n_unk; k++) free(pc->unk[k].name);
        free(pc->unk);
        free(pc->sp);
        free(pc->line);
    }
    free(r);
}

/* @SQ SN:name -> reference table (header lines are otherwise ignored, like `samtools view`) */
static void take_header_line(sam_reader *r, const char *line)
{
    if (strncmp(line, "@SQ", 3) != 0) return;
    const char *p = strstr(line, "\tSN:");
    if (!p) return;
    p += 4;
    size_t n = strcspn(p, "\t\r\n");
    char tmp[MAX_FIELD_WIDTH + 1];
    if (n == 0 || n > MAX_FIELD_WIDTH) return;
    memcpy(tmp, p, n);
    tmp[n] = '\0';
    (void)name_id(r, tmp);
}

/* canonical CIGAR text -> ops; returns op count, or 0 for '*' / anything non-canonical */
static uint32_t parse_cigar(const char *c, uint32_t *ops, uint32_t max_ops)
{
    static const char opc[] = "MIDNSHP=X";
    uint32_t n = 0;
    if (c[0] == '*' && !c[1]) return 0;
    while (*c) {
        uint64_t v = 0;
        const char *d = c;
        if (!isdigit((unsigned char)*c)) return 0;
        if (*c == '0' && isdigit((unsigned char)c[1])) return 0; /* leading zero: not what %d prints */
        while (isdigit((unsigned char)*c)) {
            v = v * 10 + (uint64_t)(*c - '0');
            if (v >= (1u << 28)) return 0;
            c++;
        }
        (void)d;
        const char *o = *c ? strchr(opc, *c) : NULL;
        if (!o || n == max_ops) return 0;
        ops[n++] = ((uint32_t)v << 4) | (uint32_t)(o - opc);
        c++;
    }
    return n;
}

/* SEQ character -> BAM 4-bit code: "=ACMGRSVTWYHKDBN", case folded, anything else N (15) */
static uint8_t g_seq_code[256];
static void init_seq_codes(void)
{
    static const char tab[] = "=ACMGRSVTWYHKDBN";
    for (int c = 0; c < 256; c++) {
        const char *p = c ? strchr(tab, toupper(c)) : NULL;
        g_seq_code[c] = p ? (uint8_t)(p - tab) : 15;
    }
}
static inline uint8_t seq_code(char ch) { return g_seq_code[(unsigned char)ch]; }

/* encodes the parsed line at out (cap bytes); returns bytes written, 0 if it does not fit */
static size_t encode_record(int32_t ref_id, const Saml *sp, const char *raw_line, uint8_t *out, size_t cap)
{
    uint32_t ops[4096];
    const uint32_t n_ops = parse_cigar(sp->cigar, ops, 4096);
    const int star_seq = sp->seq[0] == '*' && sp->seq[1] == '\0';
    const uint32_t l_seq = star_seq ? 0u : (uint32_t)sp->seq_len;
    const size_t l_name = strlen(sp->qname) + 1 > 255 ? 255 : strlen(sp->qname) + 1;
    /* RG:Z: among the optional fields of the raw line (TAB separated, fields 12..) */
    const char *rg = NULL;
    size_t rg_len = 0;
    {
        const char *q = raw_line;
        int tabs = 0;
        while (*q && tabs < 11) { if (*q == '\t') tabs++; q++; }
        while (*q && *q != '\n') {
            size_t n = strcspn(q, "\t\r\n");
            if (!rg && n >= 5 && strncmp(q, "RG:Z:", 5) == 0) { rg = q + 5; rg_len = n - 5; }
            q += n;
            if (*q == '\t') q++; else break;
        }
    }
    const size_t body = 32 + l_name + 4u * n_ops + (l_seq + 1) / 2 + l_seq + (rg ? 3 + rg_len + 1 : 0);
    if (4 + body > cap) return 0;
    uint8_t *p = out;
#define PUT32(v) do { uint32_t _v = (uint32_t)(v); p[0] = (uint8_t)_v; p[1] = (uint8_t)(_v >> 8); p[2] = (uint8_t)(_v >> 16); p[3] = (uint8_t)(_v >> 24); p += 4; } while (0)
    PUT32(body);
    PUT32(ref_id);
    {
        long long pos0 = (long long)sp->pos - 1; /* POS is an unsigned long in Saml; 0 -> -1 */
        if (sp->pos > (unsigned long)INT_MAX) pos0 = INT_MAX; /* beyond any contig: filtered either way */
        PUT32((int32_t)pos0);
    }
    {
        uint32_t mq = sp->mapq > 255 ? 255 : sp->mapq;
        PUT32((uint32_t)l_name | (mq << 8) | (4680u << 16)); /* bin: unused by the engine */
    }
    PUT32(n_ops | ((sp->flag & 0xFFFFu) << 16));
    PUT32(l_seq);
    PUT32(0xFFFFFFFFu);
    PUT32(0xFFFFFFFFu);
    /* paired reads keep TLEN; for unpaired ones line2saml overwrote isize with strlen(SEQ)
     * (sam-parse.c:66-68) and the engine derives that itself */
    PUT32((sp->flag & 1u) ? sp->isize : 0);
    memcpy(p, sp->qname, l_name - 1);
    p[l_name - 1] = 0;
    p += l_name;
    for (uint32_t k = 0; k < n_ops; k++) PUT32(ops[k]);
    for (uint32_t j = 0; j < l_seq; j += 2) {
        uint8_t hi = seq_code(sp->seq[j]), lo = j + 1 < l_seq ? seq_code(sp->seq[j + 1]) : 0;
        *p++ = (uint8_t)((hi << 4) | lo);
    }
    if (sp->qual[0] == '*' && sp->qual[1] == '\0') memset(p, 0xFF, l_seq);
    else for (uint32_t j = 0; j < l_seq; j++) p[j] = (uint8_t)(sp->qual[j] - 33);
    p += l_seq;
    if (rg) {
        *p++ = 'R'; *p++ = 'G'; *p++ = 'Z';
        memcpy(p, rg, rg_len);
        p += rg_len;
        *p++ = 0;
    }
#undef PUT32
    return (size_t)(p - out);
}

/* one "line" as the reference's fgets(buf, MAX_LINE_LEN + 1) would deliver it: up to and including
 * the newline, or MAX_LINE_LEN characters of an over-long line */
static const char *next_line_end(const char *p, const char *end)
{
    const size_t room = (size_t)(end - p) < (size_t)MAX_LINE_LEN ? (size_t)(end - p) : (size_t)MAX_LINE_LEN;
    const char *nl = (const char *)memchr(p, '\n', room);
    return nl ? nl + 1 : p + room;
}

static void *piece_main(void *arg)
{
    sam_piece *pc = (sam_piece *)arg;
    sam_reader *r = pc->r;
    int32_t hint = -1;
    pc->len = pc->n = pc->n_unk = 0;
    pc->skipped = 0;
    pc->failed = 0;
    for (const char *p = pc->a; p < pc->b;) {
        const char *e = next_line_end(p, pc->b);
        const size_t n = (size_t)(e - p);
        memcpy(pc->line, p, n);
        pc->line[n] = '\0';
        p = e;
        if (line2saml(pc->line, pc->sp)) { pc->skipped++; continue; }
        const int32_t id = name_lookup(r, pc->sp->rname, &hint);
        const size_t w = encode_record(id, pc->sp, pc->line, pc->out + pc->len, pc->cap - pc->len);
        if (w == 0) { pc->failed = 1; return NULL; } /* cannot happen: cap >= 2 x text + slack */
        if (pc->n == pc->n_cap) {
            const size_t cap = pc->n_cap ? pc->n_cap * 2 : 4096;
            uint32_t *nl = (uint32_t *)realloc(pc->rec_len, cap * sizeof(uint32_t));
            if (!nl) { pc->failed = 1; return NULL; }
            pc->rec_len = nl;
            pc->n_cap = cap;
        }
        if (id == REF_UNKNOWN) {
            if (pc->n_unk == pc->unk_cap) {
                const size_t cap = pc->unk_cap ? pc->unk_cap * 2 : 64;
                unknown_t *nu = (unknown_t *)realloc(pc->unk, cap * sizeof(unknown_t));
                if (!nu) { pc->failed = 1; return NULL; }
                pc->unk = nu;
                pc->unk_cap = cap;
            }
            pc->unk[pc->n_unk].rec_off = pc->len;
            pc->unk[pc->n_unk].name = strdup(pc->sp->rname);
            pc->n_unk++;
        }
        pc->rec_len[pc->n++] = (uint32_t)w;
        pc->len += w;
    }
    return NULL;
}

static void *piece_copy(void *arg)
{
    sam_piece *pc = (sam_piece *)arg;
    if (pc->len) memcpy(pc->r->buf + pc->base, pc->out, pc->len);
    return NULL;
}

static void run_pieces(sam_reader *r, int n, void *(*fn)(void *))
{
    pthread_t th[SAM_MAX_THREADS];
    int started[SAM_MAX_THREADS];
    for (int t = 1; t < n; t++) started[t] = pthread_create(&th[t], NULL, fn, &r->piece[t]) == 0;
    fn(&r->piece[0]);
    for (int t = 1; t < n; t++) {
        if (started[t]) pthread_join(th[t], NULL);
        else fn(&r->piece[t]);
    }
}

/* text of the next batch: whole lines, at most cap/2 bytes (so that the encoding fits the batch
 * buffer); *done = nothing is left after it */
static int next_text(sam_reader *r, const char **pa, const char **pb)
{
    const size_t want = r->cap / 2;
    if (r->map) {
        const char *a = r->map + r->map_pos, *end = r->map + r->map_len, *b;
        if ((size_t)(end - a) <= want) b = end;
        else {
            b = a + want;
            while (b > a && b[-1] != '\n') b--; /* back to a line start */
            if (b == a) b = next_line_end(a, end); /* one line longer than the slab: take it, fgets-style cuts happen later */
        }
        r->map_pos += (size_t)(b - a);
        if (b == end) r->eof = 1;
        *pa = a;
        *pb = b;
        return 0;
    }
    /* gz: keep the partial last line of the previous slab, read on */
    if (r->tpos < r->tlen) memmove(r->tbuf, r->tbuf + r->tpos, r->tlen - r->tpos);
    r->tlen -= r->tpos;
    r->tpos = 0;
    while (!r->eof && r->tlen < r->tcap) {
        const int got = gzread(r->f, r->tbuf + r->tlen, (unsigned)((r->tcap - r->tlen) > (1u << 30) ? (1u << 30) : (r->tcap - r->tlen)));
        if (got < 0) { set_err(r, "read error in the SAM input"); return -1; }
        if (got == 0) { r->eof = 1; break; }
        r->tlen += (size_t)got;
    }
    size_t b = r->tlen;
    if (!r->eof) {
        while (b > 0 && r->tbuf[b - 1] != '\n') b--;
        if (b == 0) b = (size_t)(next_line_end(r->tbuf, r->tbuf + r->tlen) - r->tbuf);
    }
    r->tpos = b;
    *pa = r->tbuf;
    *pb = r->tbuf + b;
    return 0;
}

int64_t sam_reader_next(sam_reader *r, const uint8_t **records, const uint32_t **offsets, size_t *nbytes)
{
    for (;;) {
        if (r->eof && (r->map ? r->map_pos >= r->map_len : r->tpos >= r->tlen)) return 0;
        const char *a, *b;
        if (next_text(r, &a, &b)) return -1;
        /* the leading run of '@' lines is the header */
        while (r->in_header && a < b) {
            if (*a != '@') { r->in_header = 0; break; }
            const char *e = next_line_end(a, b);
            char tmp[4096];
            const size_t n = (size_t)(e - a) < sizeof tmp - 1 ? (size_t)(e - a) : sizeof tmp - 1;
            memcpy(tmp, a, n);
            tmp[n] = '\0';
            take_header_line(r, tmp);
            a = e;
        }
        if (a == b) continue;

        /* one piece per thread, cut at line starts */
        int np = r->n_threads;
        const size_t total = (size_t)(b - a);
        if (total < ((size_t)np << 16)) np = (int)(total >> 16) + 1;
        const char *cut = a;
        int used = 0;
        for (int t = 0; t < np && cut < b; t++) {
            const char *stop = t == np - 1 ? b : a + (total / (size_t)np) * (size_t)(t + 1);
            if (stop < cut) stop = cut;
            if (stop < b) {
                const char *nl = (const char *)memchr(stop, '\n', (size_t)(b - stop));
                stop = nl ? nl + 1 : b;
            }
            sam_piece *pc = &r->piece[used++];
            pc->r = r;
            pc->a = cut;
            pc->b = stop;
            const size_t need = 2 * (size_t)(stop - cut) + 4096;
            if (pc->cap < need) {
                free(pc->out);
                pc->out = (uint8_t *)malloc(need);
                pc->cap = pc->out ? need : 0;
            }
            if (!pc->sp) pc->sp = (Saml *)malloc(sizeof(Saml));
            if (!pc->line) pc->line = (char *)malloc(MAX_LINE_LEN + 2);
            if (!pc->out || !pc->sp || !pc->line) { set_err(r, "out of memory"); return -1; }
            cut = stop;
        }
        run_pieces(r, used, piece_main);

        /* names first met in this batch, in file order; then lay the pieces end to end */
        size_t n = 0, len = 0;
        for (int t = 0; t < used; t++) {
            sam_piece *pc = &r->piece[t];
            if (pc->failed) { set_err(r, "out of memory while encoding SAM text"); return -1; }
            for (size_t k = 0; k < pc->n_unk; k++) {
                const int32_t id = name_id(r, pc->unk[k].name);
                uint8_t *q = pc->out + pc->unk[k].rec_off + 4;
                q[0] = (uint8_t)id; q[1] = (uint8_t)(id >> 8); q[2] = (uint8_t)(id >> 16); q[3] = (uint8_t)(id >> 24);
                free(pc->unk[k].name);
            }
            pc->n_unk = 0;
            pc->base = len;
            len += pc->len;
            n += pc->n;
            r->skipped += pc->skipped;
        }
        if (len > r->cap) { set_err(r, "internal: encoded batch exceeds its buffer"); return -1; }
        if (n == 0) continue; /* nothing but rejected lines: next slab */
        if (n + 2 > r->offs_cap) {
            r->offs_cap = n + 2 + (n >> 2);
            uint32_t *no = (uint32_t *)realloc(r->offs, r->offs_cap * sizeof(uint32_t));
            if (!no) { set_err(r, "out of memory"); return -1; }
            r->offs = no;
        }
        run_pieces(r, used, piece_copy);
        size_t i = 0, o = 0;
        for (int t = 0; t < used; t++) {
            const sam_piece *pc = &r->piece[t];
            for (size_t k = 0; k < pc->n; k++) {
                r->offs[i++] = (uint32_t)o;
                o += pc->rec_len[k];
            }
        }
        r->offs[n] = (uint32_t)o;
        r->len = len;
        *records = r->buf;
        *offsets = r->offs;
        *nbytes = len;
        return (int64_t)n;
    }
}
