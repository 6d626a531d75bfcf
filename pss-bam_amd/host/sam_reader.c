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
 */
#include "sam_reader.h"

#include <ctype.h>
#include <errno.h>
#include <limits.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "sam-parse.h"

struct sam_reader {
    gzFile f;
    char *line;          /* MAX_LINE_LEN + 2 */
    int in_header;
    char **names;
    int32_t n_names, names_cap;
    uint8_t *buf;
    size_t cap, len;
    uint32_t *offs;
    size_t offs_cap;
    uint64_t skipped;
    int eof;
    int have_line;       /* a line that did not fit the previous batch is pending in `line` */
    char err[256];
};

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
    r->f = gzopen(path, "rb"); /* transparent for plain text */
    if (!r->f) {
        if (err) snprintf(err, errlen, "cannot open %s: %s", path, strerror(errno));
        free(r);
        return NULL;
    }
    gzbuffer(r->f, 1u << 20);
    if (!batch_bytes && getenv("PSSBAM_BATCH_BYTES")) batch_bytes = (size_t)strtoull(getenv("PSSBAM_BATCH_BYTES"), NULL, 10);
    r->cap = batch_bytes ? batch_bytes : (size_t)256 << 20;
    if (r->cap < ((size_t)1 << 20)) r->cap = (size_t)1 << 20;
    r->buf = (uint8_t *)malloc(r->cap + 4096);
    r->line = (char *)malloc(MAX_LINE_LEN + 2);
    r->in_header = 1;
    if (!r->buf || !r->line) {
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
    for (int32_t i = 0; i < r->n_names; i++) free(r->names[i]);
    free(r->names);
    free(r->buf);
    free(r->offs);
    free(r->line);
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

static uint8_t seq_code(char ch)
{
    static const char tab[] = "=ACMGRSVTWYHKDBN";
    const char *p = ch ? strchr(tab, toupper((unsigned char)ch)) : NULL;
    return p ? (uint8_t)(p - tab) : 15;
}

/* encodes the parsed line at out (cap bytes); returns bytes written, 0 if it does not fit */
static size_t encode_record(sam_reader *r, const Saml *sp, const char *raw_line, uint8_t *out, size_t cap)
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
    PUT32(name_id(r, sp->rname));
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

int64_t sam_reader_next(sam_reader *r, const uint8_t **records, const uint32_t **offsets, size_t *nbytes)
{
    static __thread Saml sp; /* 20 KB: keep it off the stack */
    size_t n = 0;
    r->len = 0;
    while (!r->eof || r->have_line) {
        if (!r->have_line) {
            /* same chunking as the reference's fgets(saml_buf, MAX_LINE_LEN + 1, ...) */
            if (!gzgets(r->f, r->line, MAX_LINE_LEN + 1)) { r->eof = 1; break; }
        }
        r->have_line = 0;
        if (r->in_header && r->line[0] == '@') { take_header_line(r, r->line); continue; }
        r->in_header = 0;
        if (line2saml(r->line, &sp)) { r->skipped++; continue; }
        if (n + 2 > r->offs_cap) {
            r->offs_cap = r->offs_cap ? r->offs_cap * 2 : (1u << 20);
            r->offs = (uint32_t *)realloc(r->offs, r->offs_cap * sizeof(uint32_t));
        }
        size_t w = encode_record(r, &sp, r->line, r->buf + r->len, r->cap - r->len);
        if (w == 0) {
            if (n == 0) { set_err(r, "a single alignment line does not fit the batch buffer"); return -1; }
            r->have_line = 1; /* re-parse it into the next batch */
            break;
        }
        r->offs[n++] = (uint32_t)r->len;
        r->len += w;
        if (r->len >= ((size_t)1 << 32) - (1u << 20)) break;
    }
    if (n == 0) return 0;
    r->offs[n] = (uint32_t)r->len;
    *records = r->buf;
    *offsets = r->offs;
    *nbytes = r->len;
    return (int64_t)n;
}
