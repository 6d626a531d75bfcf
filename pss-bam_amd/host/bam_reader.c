/*
 * pss-bam_amd/host/bam_reader.c -- multi-threaded BGZF inflate + BAM framing.
 *
 * Pipeline per batch:
 *   1. read() a slab of the compressed file, find the BGZF block boundaries by walking the
 *      18-byte headers (BSIZE) and pick up each block's ISIZE from its trailer;
 *   2. prefix-sum the ISIZEs: every block now knows where its payload goes in the batch buffer;
 *   3. worker threads pull block indices from a shared counter and inflate (raw deflate,
 *      zlib) directly into place, checking CRC32 and ISIZE;
 *   4. follow the block_size chain over the inflated bytes to index whole records; the
 *      trailing partial record is carried to the front of the next batch.
 * Format references: SAM/BAM specification sections 4.1 (BGZF) and 4.2 (BAM).
 */
#include "bam_reader.h"

#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>

#define BGZF_MAX_BLOCK 65536u
#define UPAD 4096u /* slack behind each batch buffer (device over-reads, page alignment) */

typedef struct {
    size_t coff;     /* offset of the block in the compressed slab */
    uint32_t clen;   /* whole block length (BSIZE + 1)             */
    uint32_t xlen;
    uint32_t isize;
    size_t uoff;     /* destination offset in the batch buffer     */
} blk_t;

struct bam_reader {
    int fd;
    int n_threads;
    /* compressed slab */
    uint8_t *cbuf;
    size_t ccap, clen, cpos; /* valid bytes [cpos, clen) */
    int file_eof;
    /* inflated batches: two buffers (one allocation).  The caller works on `cur` while a
     * background thread already inflates the following batch into the other one. */
    uint8_t *ubase;    /* the allocation: 2 * (ucap + UPAD) bytes                          */
    uint8_t *ubuf;     /* = buffer `cur`                                                  */
    int cur;
    size_t ucap;
    size_t ulen;       /* valid inflated bytes of buffer `cur`                            */
    size_t upos;       /* first byte of it not yet handed out                             */
    /* background fill of the other buffer */
    pthread_t bg_thread;
    int bg_running;
    int bg_rc;
    size_t bg_ulen;    /* valid bytes the fill left in the other buffer                   */
    const uint8_t *bg_carry;
    size_t bg_carry_len;
    int bg_pending;    /* a batch was handed out and its successor is being (was) prefetched */
    /* block table of the current batch */
    blk_t *blk;
    size_t n_blk, blk_cap;
    /* record index of the current batch */
    uint32_t *offs;
    size_t offs_cap;
    bam_header hdr;
    int header_done;
    char err[256];
    double inflate_s;
    /* worker coordination */
    uint8_t *ubuf_fill; /* buffer the inflate workers write into */
    size_t next_blk;
    pthread_mutex_t mu;
    int worker_failed;
};

static void set_err(bam_reader *r, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(r->err, sizeof r->err, fmt, ap);
    va_end(ap);
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

static uint32_t le16(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8); }
static uint32_t le32(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

/* refill the compressed slab: keep [cpos, clen), append from the file */
static int slab_fill(bam_reader *r)
{
    if (r->cpos > 0) {
        memmove(r->cbuf, r->cbuf + r->cpos, r->clen - r->cpos);
        r->clen -= r->cpos;
        r->cpos = 0;
    }
    while (!r->file_eof && r->clen < r->ccap) {
        ssize_t n = read(r->fd, r->cbuf + r->clen, r->ccap - r->clen);
        if (n < 0) {
            if (errno == EINTR) continue;
            set_err(r, "read failed: %s", strerror(errno));
            return -1;
        }
        if (n == 0) { r->file_eof = 1; break; }
        r->clen += (size_t)n;
    }
    return 0;
}

/* parses one BGZF header at p (avail bytes); 0 = need more bytes, -1 = not BGZF, else block length */
static long bgzf_block_len(const uint8_t *p, size_t avail, uint32_t *xlen_out)
{
    uint32_t xlen, o;
    if (avail < 18) return 0;
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return -1;
    xlen = le16(p + 10);
    if (avail < 12 + xlen) return 0;
    for (o = 12; o + 4 <= 12 + xlen;) {
        uint32_t slen = le16(p + o + 2);
        if (p[o] == 'B' && p[o + 1] == 'C' && slen == 2) {
            *xlen_out = xlen;
            return (long)le16(p + o + 4) + 1;
        }
        o += 4 + slen;
    }
    return -1;
}

static void *inflate_worker(void *arg)
{
    bam_reader *r = (bam_reader *)arg;
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) {
        pthread_mutex_lock(&r->mu);
        r->worker_failed = 1;
        pthread_mutex_unlock(&r->mu);
        return NULL;
    }
    for (;;) {
        size_t i;
        const blk_t *b;
        const uint8_t *src;
        pthread_mutex_lock(&r->mu);
        i = r->next_blk;
        r->next_blk += 8; /* a few blocks per grab keeps the lock cold */
        pthread_mutex_unlock(&r->mu);
        if (i >= r->n_blk) break;
        for (size_t k = i; k < i + 8 && k < r->n_blk; k++) {
            b = &r->blk[k];
            if (b->isize == 0) continue;
            src = r->cbuf + b->coff;
            inflateReset(&zs);
            zs.next_in = (Bytef *)(src + 12 + b->xlen);
            zs.avail_in = b->clen - 12 - b->xlen - 8;
            zs.next_out = r->ubuf_fill + b->uoff;
            zs.avail_out = b->isize;
            if (inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.avail_out != 0 ||
                (uint32_t)crc32(crc32(0L, Z_NULL, 0), r->ubuf_fill + b->uoff, b->isize) != le32(src + b->clen - 8)) {
                pthread_mutex_lock(&r->mu);
                r->worker_failed = 1;
                pthread_mutex_unlock(&r->mu);
            }
        }
    }
    inflateEnd(&zs);
    return NULL;
}

/* Fills batch buffer `dst` with [carry bytes | as many whole inflated BGZF blocks as fit];
 * *len_out = valid bytes.  The worker threads write into `dst` through r->ubuf_fill.
 * Returns 0 ok / -1 error. */
static int fill_buffer(bam_reader *r, uint8_t *dst, const uint8_t *carry, size_t carry_len, size_t *len_out)
{
    double t0;
    if (carry_len) memmove(dst, carry, carry_len);
    r->n_blk = 0;
    size_t uoff = carry_len;
    for (;;) {
        if (r->clen - r->cpos < BGZF_MAX_BLOCK && !r->file_eof) {
            /* the block table refers to slab offsets: stop here if blocks are already queued */
            if (r->n_blk) break;
            if (slab_fill(r)) return -1;
        }
        if (r->cpos >= r->clen) break; /* end of file */
        uint32_t xlen = 0;
        long bl = bgzf_block_len(r->cbuf + r->cpos, r->clen - r->cpos, &xlen);
        if (bl < 0) { set_err(r, "not a BGZF block at compressed offset (corrupt or not a BAM file)"); return -1; }
        if (bl == 0 || (size_t)bl > r->clen - r->cpos) {
            if (r->file_eof) { set_err(r, "truncated BGZF block at end of file"); return -1; }
            if (r->n_blk) break;
            if (slab_fill(r)) return -1;
            continue;
        }
        if ((size_t)bl < 12u + xlen + 8u) { set_err(r, "BGZF block shorter than its own header"); return -1; }
        uint32_t isize = le32(r->cbuf + r->cpos + bl - 4);
        if (isize > BGZF_MAX_BLOCK) { set_err(r, "BGZF ISIZE %u exceeds 64 KiB", isize); return -1; }
        if (uoff + isize > r->ucap) break; /* batch buffer full */
        if (r->n_blk == r->blk_cap) {
            r->blk_cap = r->blk_cap ? r->blk_cap * 2 : 8192;
            r->blk = (blk_t *)realloc(r->blk, r->blk_cap * sizeof(blk_t));
        }
        r->blk[r->n_blk++] = (blk_t){r->cpos, (uint32_t)bl, xlen, isize, uoff};
        uoff += isize;
        r->cpos += (size_t)bl;
    }
    *len_out = uoff;
    if (r->n_blk == 0) return 0;
    t0 = now_s();
    r->next_blk = 0;
    r->worker_failed = 0;
    r->ubuf_fill = dst;
    {
        int nt = r->n_threads;
        if ((size_t)nt > (r->n_blk + 7) / 8) nt = (int)((r->n_blk + 7) / 8);
        if (nt <= 1) {
            inflate_worker(r);
        } else {
            pthread_t th[64];
            int started = 0;
            for (int t = 0; t < nt && t < 64; t++)
                if (pthread_create(&th[started], NULL, inflate_worker, r) == 0) started++;
            if (started == 0) inflate_worker(r);
            for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
        }
    }
    r->inflate_s += now_s() - t0;
    if (r->worker_failed) { set_err(r, "BGZF inflate / CRC check failed"); return -1; }
    return 0;
}

/* synchronous refill of the current buffer: keeps its unconsumed tail, appends more blocks */
static int batch_fill(bam_reader *r)
{
    size_t len = 0;
    if (fill_buffer(r, r->ubuf, r->ubuf + r->upos, r->ulen - r->upos, &len)) return -1;
    r->ulen = len;
    r->upos = 0;
    return 0;
}

/* background: the batch after the current one goes into the other buffer */
static void *bg_fill_main(void *arg)
{
    bam_reader *r = (bam_reader *)arg;
    uint8_t *dst = r->ubase + (size_t)(r->cur ^ 1) * (r->ucap + UPAD);
    r->bg_rc = fill_buffer(r, dst, r->bg_carry, r->bg_carry_len, &r->bg_ulen);
    return NULL;
}

static void bg_start(bam_reader *r, const uint8_t *carry, size_t carry_len)
{
    r->bg_carry = carry;
    r->bg_carry_len = carry_len;
    r->bg_rc = 0;
    r->bg_running = pthread_create(&r->bg_thread, NULL, bg_fill_main, r) == 0;
    if (!r->bg_running) bg_fill_main(r); /* no thread: do it now */
}

/* waits for the background fill and makes its buffer the current one; 0 ok / -1 error */
static int bg_take(bam_reader *r)
{
    if (r->bg_running) {
        pthread_join(r->bg_thread, NULL);
        r->bg_running = 0;
    }
    if (r->bg_rc) return -1;
    r->cur ^= 1;
    r->ubuf = r->ubase + (size_t)r->cur * (r->ucap + UPAD);
    r->ulen = r->bg_ulen;
    r->upos = 0;
    return 0;
}

/* makes at least `need` inflated bytes available at upos (for the header); 0 ok, 1 EOF first, -1 error */
static int need_bytes(bam_reader *r, size_t need)
{
    while (r->ulen - r->upos < need) {
        size_t before = r->ulen - r->upos;
        if (need > r->ucap) { set_err(r, "BAM header larger than the batch buffer"); return -1; }
        if (batch_fill(r)) return -1;
        if (r->ulen - r->upos == before) return 1;
    }
    return 0;
}

static int parse_header(bam_reader *r)
{
    int rc;
    if ((rc = need_bytes(r, 12))) { if (rc > 0) set_err(r, "file too short for a BAM header"); return -1; }
    const uint8_t *p = r->ubuf + r->upos;
    if (memcmp(p, "BAM\1", 4) != 0) { set_err(r, "BAM magic not found"); return -1; }
    uint32_t l_text = le32(p + 4);
    if ((rc = need_bytes(r, 12 + (size_t)l_text))) { if (rc > 0) set_err(r, "truncated BAM header text"); return -1; }
    p = r->ubuf + r->upos;
    r->hdr.l_text = l_text;
    r->hdr.text = (char *)malloc((size_t)l_text + 1);
    memcpy(r->hdr.text, p + 8, l_text);
    r->hdr.text[l_text] = '\0';
    int32_t n_ref = (int32_t)le32(p + 8 + l_text);
    if (n_ref < 0) { set_err(r, "negative reference count"); return -1; }
    r->hdr.n_ref = n_ref;
    r->hdr.ref_name = (char **)calloc((size_t)n_ref + 1, sizeof(char *));
    r->hdr.ref_len = (uint32_t *)calloc((size_t)n_ref + 1, sizeof(uint32_t));
    size_t o = 12 + (size_t)l_text;
    for (int32_t i = 0; i < n_ref; i++) {
        if ((rc = need_bytes(r, o + 4))) { if (rc > 0) set_err(r, "truncated reference list"); return -1; }
        uint32_t l_name = le32(r->ubuf + r->upos + o);
        if (l_name == 0 || l_name > (1u << 20)) { set_err(r, "bad reference name length"); return -1; }
        if ((rc = need_bytes(r, o + 8 + l_name))) { if (rc > 0) set_err(r, "truncated reference list"); return -1; }
        p = r->ubuf + r->upos;
        r->hdr.ref_name[i] = (char *)malloc(l_name);
        memcpy(r->hdr.ref_name[i], p + o + 4, l_name);
        r->hdr.ref_name[i][l_name - 1] = '\0';
        r->hdr.ref_len[i] = le32(p + o + 4 + l_name);
        o += 8 + l_name;
    }
    r->upos += o;
    r->header_done = 1;
    return 0;
}

bam_reader *bam_reader_open(const char *path, int n_threads, size_t batch_bytes, char *err, size_t errlen)
{
    bam_reader *r = (bam_reader *)calloc(1, sizeof *r);
    if (!r) return NULL;
    r->fd = open(path, O_RDONLY);
    if (r->fd < 0) {
        if (err) snprintf(err, errlen, "cannot open %s: %s", path, strerror(errno));
        free(r);
        return NULL;
    }
#ifdef POSIX_FADV_SEQUENTIAL
    (void)posix_fadvise(r->fd, 0, 0, POSIX_FADV_SEQUENTIAL);
#endif
    if (n_threads <= 0) {
        long n = sysconf(_SC_NPROCESSORS_ONLN);
        n_threads = n > 32 ? 32 : (n < 1 ? 1 : (int)n);
    }
    r->n_threads = n_threads;
    if (!batch_bytes && getenv("PSSBAM_BATCH_BYTES")) batch_bytes = (size_t)strtoull(getenv("PSSBAM_BATCH_BYTES"), NULL, 10);
    r->ucap = batch_bytes ? batch_bytes : (size_t)256 << 20;
    if (r->ucap < 4 * BGZF_MAX_BLOCK) r->ucap = 4 * BGZF_MAX_BLOCK;
    r->ccap = r->ucap / 2 + 2 * BGZF_MAX_BLOCK; /* slab of compressed input per refill */
    r->cbuf = (uint8_t *)malloc(r->ccap);
    /* page-aligned so the caller can register it for DMA; slack for device over-reads */
    r->ucap = (r->ucap + 4095) & ~(size_t)4095;
    if (posix_memalign((void **)&r->ubase, 4096, 2 * (r->ucap + UPAD)) != 0) r->ubase = NULL;
    r->ubuf = r->ubase;
    r->cur = 0;
    pthread_mutex_init(&r->mu, NULL);
    if (!r->cbuf || !r->ubase) {
        if (err) snprintf(err, errlen, "out of memory");
        bam_reader_close(r);
        return NULL;
    }
    if (parse_header(r)) {
        if (err) snprintf(err, errlen, "%s: %s", path, r->err);
        bam_reader_close(r);
        return NULL;
    }
    return r;
}

const bam_header *bam_reader_header(const bam_reader *r) { return &r->hdr; }
const char *bam_reader_error(const bam_reader *r) { return r->err; }
double bam_reader_inflate_seconds(const bam_reader *r) { return r->inflate_s; }

void bam_reader_buffer(const bam_reader *r, void **base, size_t *bytes)
{
    *base = r->ubase;
    *bytes = 2 * (r->ucap + UPAD);
}

int64_t bam_reader_next(bam_reader *r, const uint8_t **records, const uint32_t **offsets, size_t *nbytes)
{
    /* the batch handed out by the previous call is finished with: switch to the one the
     * background thread has been inflating meanwhile */
    if (r->bg_running || r->bg_pending) {
        r->bg_pending = 0;
        if (bg_take(r)) return -1;
    }
    for (;;) {
        /* index whole records in [upos, ulen) */
        size_t o = r->upos, n = 0;
        const size_t limit = r->upos + (((size_t)1 << 32) - (1u << 20));
        while (o + 4 <= r->ulen && o < limit) {
            uint32_t bs = le32(r->ubuf + o);
            if (bs < 32) { set_err(r, "alignment record with block_size %u < 32", bs); return -1; }
            if (o + 4 + (size_t)bs > r->ulen) break;
            if (n + 2 > r->offs_cap) {
                r->offs_cap = r->offs_cap ? r->offs_cap * 2 : (1u << 20);
                r->offs = (uint32_t *)realloc(r->offs, r->offs_cap * sizeof(uint32_t));
            }
            r->offs[n++] = (uint32_t)(o - r->upos);
            o += 4 + (size_t)bs;
        }
        if (n > 0) {
            r->offs[n] = (uint32_t)(o - r->upos);
            *records = r->ubuf + r->upos;
            *offsets = r->offs;
            *nbytes = o - r->upos;
            r->upos = o;
            /* start inflating the following batch behind the caller's work on this one; the
             * unconsumed tail (a partial record) is carried over by the fill itself */
            bg_start(r, r->ubuf + o, r->ulen - o);
            r->bg_pending = 1;
            return (int64_t)n;
        }
        /* nothing whole in hand: inflate more, synchronously */
        size_t have = r->ulen - r->upos;
        if (have >= 4) {
            uint32_t bs = le32(r->ubuf + r->upos);
            if (4 + (size_t)bs > r->ucap) { set_err(r, "alignment record of %u bytes exceeds the batch buffer", bs); return -1; }
        }
        if (batch_fill(r)) return -1;
        if (r->ulen - r->upos == have) {
            if (have) { set_err(r, "truncated alignment record at end of file"); return -1; }
            return 0;
        }
    }
}

void bam_reader_close(bam_reader *r)
{
    if (!r) return;
    if (r->bg_running) pthread_join(r->bg_thread, NULL);
    if (r->fd >= 0) close(r->fd);
    free(r->cbuf);
    free(r->ubase);
    free(r->blk);
    free(r->offs);
    free(r->hdr.text);
    if (r->hdr.ref_name)
        for (int32_t i = 0; i < r->hdr.n_ref; i++) free(r->hdr.ref_name[i]);
    free(r->hdr.ref_name);
    free(r->hdr.ref_len);
    pthread_mutex_destroy(&r->mu);
    free(r);
}

/* ------------------------------------------------------------------------------------------ */
/* record -> SAM text (what `samtools view` prints); used by bin/bam2sam and by the tests       */
/* ------------------------------------------------------------------------------------------ */

typedef struct { char *p; size_t cap, n; int ovf; } sbuf;
static void sb_putc(sbuf *s, char c) { if (s->n + 1 < s->cap) s->p[s->n++] = c; else s->ovf = 1; }
static void sb_puts(sbuf *s, const char *t) { while (*t) sb_putc(s, *t++); }
static void sb_printf(sbuf *s, const char *fmt, ...)
{
    char tmp[64];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tmp, sizeof tmp, fmt, ap);
    va_end(ap);
    sb_puts(s, tmp);
}

static const uint8_t *aux_next(const uint8_t *p, const uint8_t *end)
{
    /* p at tag[2] type[1]; returns pointer past the field or NULL */
    if (end - p < 3) return NULL;
    uint8_t ty = p[2];
    p += 3;
    switch (ty) {
    case 'A': case 'c': case 'C': return end - p >= 1 ? p + 1 : NULL;
    case 's': case 'S': return end - p >= 2 ? p + 2 : NULL;
    case 'i': case 'I': case 'f': return end - p >= 4 ? p + 4 : NULL;
    case 'Z': case 'H': {
        const uint8_t *z = (const uint8_t *)memchr(p, 0, (size_t)(end - p));
        return z ? z + 1 : NULL;
    }
    case 'B': {
        if (end - p < 5) return NULL;
        uint8_t sub = p[0];
        uint32_t cnt = le32(p + 1);
        uint32_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
        uint64_t tot = 5 + (uint64_t)cnt * es;
        return (uint64_t)(end - p) >= tot ? p + tot : NULL;
    }
    default: return NULL;
    }
}

int bam_record_has_rg(const uint8_t *rec, uint32_t rec_len, const char *rg)
{
    if (rec_len < 36) return 0;
    uint32_t l_name = rec[12], n_cig = le16(rec + 16), l_seq = le32(rec + 20);
    uint64_t aux = 36ull + l_name + 4ull * n_cig + ((uint64_t)l_seq + 1) / 2 + l_seq;
    if (aux > rec_len) return 0;
    const uint8_t *p = rec + aux, *end = rec + rec_len;
    while (p && end - p >= 3) {
        const uint8_t *nx = aux_next(p, end);
        if (!nx) return 0;
        if (p[0] == 'R' && p[1] == 'G' && p[2] == 'Z') return strcmp((const char *)p + 3, rg) == 0;
        p = nx;
    }
    return 0;
}

long bam_record_to_sam(const uint8_t *rec, uint32_t rec_len, const bam_header *h, char *out, size_t cap)
{
    sbuf s = {out, cap, 0, 0};
    if (rec_len < 36) return -1;
    int32_t ref_id = (int32_t)le32(rec + 4), pos = (int32_t)le32(rec + 8);
    uint32_t l_name = rec[12], mapq = rec[13], n_cig = le16(rec + 16), flag = le16(rec + 18), l_seq = le32(rec + 20);
    int32_t nref = (int32_t)le32(rec + 24), npos = (int32_t)le32(rec + 28), tlen = (int32_t)le32(rec + 32);
    uint64_t cig = 36ull + l_name, seq = cig + 4ull * n_cig, qual = seq + ((uint64_t)l_seq + 1) / 2, aux = qual + l_seq;
    if (aux > rec_len || l_name == 0) return -1;
    sb_puts(&s, l_name > 1 ? (const char *)rec + 36 : "*");
    sb_printf(&s, "\t%u\t", flag);
    sb_puts(&s, (ref_id >= 0 && ref_id < h->n_ref) ? h->ref_name[ref_id] : "*");
    sb_printf(&s, "\t%d\t%u\t", pos + 1, mapq);
    if (n_cig == 0) sb_putc(&s, '*');
    for (uint32_t k = 0; k < n_cig; k++) {
        uint32_t c = le32(rec + cig + 4 * k);
        sb_printf(&s, "%u", c >> 4);
        sb_putc(&s, "MIDNSHP=X???????"[c & 15]);
    }
    sb_putc(&s, '\t');
    if (nref < 0) sb_putc(&s, '*');
    else if (nref == ref_id) sb_putc(&s, '=');
    else sb_puts(&s, nref < h->n_ref ? h->ref_name[nref] : "*");
    sb_printf(&s, "\t%d\t%d\t", npos + 1, tlen);
    if (l_seq == 0) sb_putc(&s, '*');
    for (uint32_t j = 0; j < l_seq; j++) {
        uint8_t b = rec[seq + (j >> 1)];
        sb_putc(&s, "=ACMGRSVTWYHKDBN"[(j & 1) ? (b & 15) : (b >> 4)]);
    }
    sb_putc(&s, '\t');
    if (l_seq == 0 || rec[qual] == 0xFF) sb_putc(&s, '*');
    else for (uint32_t j = 0; j < l_seq; j++) sb_putc(&s, (char)(rec[qual + j] + 33));
    /* optional fields */
    const uint8_t *p = rec + aux, *end = rec + rec_len;
    while (end - p >= 3) {
        const uint8_t *nx = aux_next(p, end);
        if (!nx) break;
        sb_putc(&s, '\t');
        sb_putc(&s, (char)p[0]);
        sb_putc(&s, (char)p[1]);
        sb_putc(&s, ':');
        const uint8_t *v = p + 3;
        switch (p[2]) {
        case 'A': sb_puts(&s, "A:"); sb_putc(&s, (char)v[0]); break;
        case 'c': sb_printf(&s, "i:%d", (int)(int8_t)v[0]); break;
        case 'C': sb_printf(&s, "i:%u", (unsigned)v[0]); break;
        case 's': sb_printf(&s, "i:%d", (int)(int16_t)le16(v)); break;
        case 'S': sb_printf(&s, "i:%u", le16(v)); break;
        case 'i': sb_printf(&s, "i:%d", (int32_t)le32(v)); break;
        case 'I': sb_printf(&s, "i:%u", le32(v)); break;
        case 'f': { float f; uint32_t w = le32(v); memcpy(&f, &w, 4); sb_printf(&s, "f:%g", f); break; }
        case 'Z': sb_puts(&s, "Z:"); sb_puts(&s, (const char *)v); break;
        case 'H': sb_puts(&s, "H:"); sb_puts(&s, (const char *)v); break;
        case 'B': {
            uint8_t sub = v[0];
            uint32_t cnt = le32(v + 1);
            const uint8_t *e = v + 5;
            sb_puts(&s, "B:");
            sb_putc(&s, (char)sub);
            for (uint32_t k = 0; k < cnt; k++) {
                sb_putc(&s, ',');
                switch (sub) {
                case 'c': sb_printf(&s, "%d", (int)(int8_t)e[k]); break;
                case 'C': sb_printf(&s, "%u", (unsigned)e[k]); break;
                case 's': sb_printf(&s, "%d", (int)(int16_t)le16(e + 2 * k)); break;
                case 'S': sb_printf(&s, "%u", le16(e + 2 * k)); break;
                case 'i': sb_printf(&s, "%d", (int32_t)le32(e + 4 * k)); break;
                case 'I': sb_printf(&s, "%u", le32(e + 4 * k)); break;
                default: { float f; uint32_t w = le32(e + 4 * k); memcpy(&f, &w, 4); sb_printf(&s, "%g", f); }
                }
            }
            break;
        }
        default: break;
        }
        p = nx;
    }
    sb_putc(&s, '\n');
    if (s.ovf) return -1;
    out[s.n] = '\0';
    return (long)s.n;
}
