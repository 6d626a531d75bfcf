/*
 * pss-bam_amd/host/inflate_fast.c -- see inflate_fast.h.
 *
 * Table entry (u32):  bits 0-4 codeword length | bits 5-7 kind | bits 8-15 literal byte or
 * extra-bit count | bits 16-31 base value (match length 3..258, distance 1..24577).
 * A codeword longer than the table index maps its prefix to K_LONG; such symbols are decoded
 * bit by bit from the canonical code description (counts per length + symbols in code order).
 */
#include "inflate_fast.h"

#include <stdlib.h>
#include <string.h>

enum { K_BAD = 0, K_LIT = 1, K_LEN = 2, K_EOB = 3, K_LONG = 4 };
#define E_LEN(e) ((e) & 31u)
#define E_KIND(e) (((e) >> 5) & 7u)
#define E_MAKE(len, kind, b8, base) ((uint32_t)(len) | ((uint32_t)(kind) << 5) | ((uint32_t)(b8) << 8) | ((uint32_t)(base) << 16))

enum { TT_LL, TT_DS, TT_CL };

static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static uint32_t sym_entry(int tt, unsigned sym, unsigned len)
{
    if (tt == TT_CL) return E_MAKE(len, K_LIT, sym, 0);
    if (tt == TT_DS) return sym < 30 ? E_MAKE(len, K_LEN, dist_extra[sym], dist_base[sym]) : E_MAKE(len, K_BAD, 0, 0);
    if (sym < 256) return E_MAKE(len, K_LIT, sym, 0);
    if (sym == 256) return E_MAKE(len, K_EOB, 0, 0);
    if (sym < 286) return E_MAKE(len, K_LEN, len_extra[sym - 257], len_base[sym - 257]);
    return E_MAKE(len, K_BAD, 0, 0);
}

static unsigned bit_reverse(unsigned v, unsigned n)
{
    unsigned r = 0;
    for (unsigned i = 0; i < n; i++) {
        r = (r << 1) | (v & 1u);
        v >>= 1;
    }
    return r;
}

/* canonical Huffman code from code lengths; 0 ok, -1 over-subscribed */
static int build_table(int tt, const uint8_t *lens, unsigned n_syms, unsigned tbits, uint32_t *table, uint16_t *count,
                       uint16_t *sorted)
{
    unsigned offs[16];
    memset(count, 0, 16 * sizeof(uint16_t));
    for (unsigned s = 0; s < n_syms; s++) count[lens[s]]++;
    count[0] = 0;
    int left = 1;
    for (unsigned l = 1; l <= 15; l++) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return -1;
    }
    offs[1] = 0;
    for (unsigned l = 1; l < 15; l++) offs[l + 1] = offs[l] + count[l];
    for (unsigned s = 0; s < n_syms; s++)
        if (lens[s]) sorted[offs[lens[s]]++] = (uint16_t)s;

    const unsigned size = 1u << tbits;
    for (unsigned i = 0; i < size; i++) table[i] = E_MAKE(0, K_BAD, 0, 0);
    unsigned code = 0, k = 0;
    for (unsigned l = 1; l <= 15; l++) {
        for (unsigned c = 0; c < count[l]; c++, k++, code++) {
            if (l <= tbits) {
                const uint32_t e = sym_entry(tt, sorted[k], l);
                for (unsigned i = bit_reverse(code, l); i < size; i += 1u << l) table[i] = e;
            } else {
                table[bit_reverse(code >> (l - tbits), tbits)] = E_MAKE(0, K_LONG, 0, 0);
            }
        }
        code <<= 1;
    }
    return 0;
}

/* bit-serial decode of one symbol from the canonical description; returns its table entry
 * (with the codeword length filled in) or a K_BAD entry */
static uint32_t decode_long(int tt, uint64_t bb, const uint16_t *count, const uint16_t *sorted)
{
    int code = 0, first = 0, index = 0;
    for (unsigned l = 1; l <= 15; l++) {
        code |= (int)((bb >> (l - 1)) & 1u);
        const int cnt = count[l];
        if (code - cnt < first) return sym_entry(tt, sorted[index + (code - first)], l);
        index += cnt;
        first += cnt;
        first <<= 1;
        code <<= 1;
    }
    return E_MAKE(0, K_BAD, 0, 0);
}

static inline uint64_t load64(const uint8_t *p)
{
    uint64_t v;
    memcpy(&v, p, 8);
    return v; /* little-endian hosts only (x86-64, like the BAM format itself) */
}

/* tops the bit buffer up to >= 56 bits while input remains */
#define REFILL()                                                          \
    do {                                                                  \
        if (in_end - in >= 8) {                                           \
            bb |= load64(in) << bl;                                       \
            in += (63u - bl) >> 3;                                        \
            bl |= 56u;                                                    \
        } else {                                                          \
            while (bl <= 56u && in < in_end) {                            \
                bb |= (uint64_t)*in++ << bl;                              \
                bl += 8u;                                                 \
            }                                                             \
        }                                                                 \
    } while (0)
#define DROP(n)                                                           \
    do {                                                                  \
        if ((n) > bl) return PSS_INF_TRUNCATED;                           \
        bb >>= (n);                                                       \
        bl -= (n);                                                        \
    } while (0)
#define BITS(n) ((unsigned)(bb & ((1ull << (n)) - 1ull)))

static int read_dynamic_header(pss_inflater *st, const uint8_t **pin, const uint8_t *in_end, uint64_t *pbb, unsigned *pbl)
{
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    const uint8_t *in = *pin;
    uint64_t bb = *pbb;
    unsigned bl = *pbl;
    uint8_t cl_lens[19];

    REFILL();
    const unsigned hlit = BITS(5) + 257;
    DROP(5);
    const unsigned hdist = BITS(5) + 1;
    DROP(5);
    const unsigned hclen = BITS(4) + 4;
    DROP(4);
    if (hlit > 286 || hdist > 30) return PSS_INF_BAD_BLOCK;
    memset(cl_lens, 0, sizeof cl_lens);
    for (unsigned i = 0; i < hclen; i++) {
        REFILL();
        cl_lens[order[i]] = (uint8_t)BITS(3);
        DROP(3);
    }
    if (build_table(TT_CL, cl_lens, 19, 7, st->cl, st->cl_count, st->cl_sorted)) return PSS_INF_BAD_CODES;

    const unsigned total = hlit + hdist;
    unsigned i = 0;
    while (i < total) {
        REFILL();
        const uint32_t e = st->cl[BITS(7)];
        if (E_KIND(e) != K_LIT) return PSS_INF_BAD_CODES; /* code-length codes are at most 7 bits: no long path */
        DROP(E_LEN(e));
        const unsigned sym = (e >> 8) & 0xFFu;
        if (sym < 16) {
            st->lens[i++] = (uint8_t)sym;
            continue;
        }
        unsigned rep, val = 0;
        if (sym == 16) {
            if (i == 0) return PSS_INF_BAD_CODES;
            val = st->lens[i - 1];
            rep = 3 + BITS(2);
            DROP(2);
        } else if (sym == 17) {
            rep = 3 + BITS(3);
            DROP(3);
        } else {
            rep = 11 + BITS(7);
            DROP(7);
        }
        if (i + rep > total) return PSS_INF_BAD_CODES;
        memset(st->lens + i, (int)val, rep);
        i += rep;
    }
    if (st->lens[256] == 0) return PSS_INF_BAD_CODES; /* no end-of-block code */
    if (build_table(TT_LL, st->lens, hlit, PSS_LL_BITS, st->ll, st->ll_count, st->ll_sorted)) return PSS_INF_BAD_CODES;
    if (build_table(TT_DS, st->lens + hlit, hdist, PSS_DS_BITS, st->ds, st->ds_count, st->ds_sorted)) return PSS_INF_BAD_CODES;
    *pin = in;
    *pbb = bb;
    *pbl = bl;
    return 0;
}

static void fixed_tables(pss_inflater *st)
{
    unsigned s = 0;
    for (; s < 144; s++) st->lens[s] = 8;
    for (; s < 256; s++) st->lens[s] = 9;
    for (; s < 280; s++) st->lens[s] = 7;
    for (; s < 288; s++) st->lens[s] = 8;
    for (s = 0; s < 32; s++) st->lens[288 + s] = 5;
    (void)build_table(TT_LL, st->lens, 288, PSS_LL_BITS, st->ll, st->ll_count, st->ll_sorted);
    (void)build_table(TT_DS, st->lens + 288, 32, PSS_DS_BITS, st->ds, st->ds_count, st->ds_sorted);
}

typedef struct {
    const uint8_t *in, *in_end;
    uint8_t *out, *out0, *out_end;
    uint64_t bb;
    unsigned bl;
} dstate;

/* Decodes literal/length + distance symbols of one Huffman block until its end-of-block code.
 * FAST = 1: no per-symbol bounds checks -- valid while >= 16 input bytes and >= 280 output
 * bytes remain (one iteration consumes <= 12 bytes and produces <= 3 literals or a 258-byte
 * match plus 15 bytes of copy slack); returns 1 when that margin is used up and the caller must
 * continue with FAST = 0, which checks everything.  0 = end of block, negative = error. */
static inline __attribute__((always_inline)) int run_codes(const pss_inflater *st, dstate *d, const int FAST)
{
    const uint8_t *in = d->in, *const in_end = d->in_end;
    uint8_t *out = d->out, *const out0 = d->out0, *const out_end = d->out_end;
    uint64_t bb = d->bb;
    unsigned bl = d->bl;
    const uint64_t ll_mask = (1u << PSS_LL_BITS) - 1u, ds_mask = (1u << PSS_DS_BITS) - 1u;
    int rc;
#define FINISH(code) do { rc = (code); goto done; } while (0)
#define XDROP(n) do { if (!FAST && (n) > bl) FINISH(PSS_INF_TRUNCATED); bb >>= (n); bl -= (n); } while (0)
#define XREFILL() do { if (FAST) { bb |= load64(in) << bl; in += (63u - bl) >> 3; bl |= 56u; } else REFILL(); } while (0)
#define PUT_LITERAL(e) do { XDROP(E_LEN(e)); if (!FAST && out == out_end) FINISH(PSS_INF_OVERRUN); *out++ = (uint8_t)((e) >> 8); } while (0)
    for (;;) {
        if (FAST && (in_end - in < 16 || out_end - out < 280)) FINISH(1);
        XREFILL();
        uint32_t e = st->ll[bb & ll_mask];
        /* up to three literals per refill: 3 x 15 bits <= 56 */
        if (E_KIND(e) == K_LIT) {
            PUT_LITERAL(e);
            e = st->ll[bb & ll_mask];
            if (E_KIND(e) == K_LIT) {
                PUT_LITERAL(e);
                e = st->ll[bb & ll_mask];
                if (E_KIND(e) == K_LIT) {
                    PUT_LITERAL(e);
                    continue;
                }
            }
            XREFILL();
            e = st->ll[bb & ll_mask];
        }
        if (E_KIND(e) == K_LONG) e = decode_long(TT_LL, bb, st->ll_count, st->ll_sorted);
        if (E_KIND(e) == K_LIT) {
            PUT_LITERAL(e);
            continue;
        }
        if (E_KIND(e) == K_EOB) {
            XDROP(E_LEN(e));
            FINISH(0);
        }
        if (E_KIND(e) != K_LEN) FINISH(PSS_INF_BAD_SYMBOL);
        /* match: length (<= 15 + 5 bits), then distance (<= 15 + 13 bits): 48 <= 56.  All fields are
         * cut out of the SAME bit buffer at running offsets and dropped once, which keeps the
         * dependency chain to  entry -> offset -> entry  instead of four buffer updates. */
        unsigned used = E_LEN(e);
        unsigned xb = (e >> 8) & 0xFFu;
        const size_t len = (e >> 16) + (unsigned)((bb >> used) & ((1ull << xb) - 1ull));
        used += xb;
        uint32_t dd = st->ds[(bb >> used) & ds_mask];
        if (E_KIND(dd) == K_LONG) dd = decode_long(TT_DS, bb >> used, st->ds_count, st->ds_sorted);
        if (E_KIND(dd) != K_LEN) FINISH(PSS_INF_BAD_SYMBOL);
        used += E_LEN(dd);
        xb = (dd >> 8) & 0xFFu;
        const size_t dist = (dd >> 16) + (unsigned)((bb >> used) & ((1ull << xb) - 1ull));
        used += xb;
        XDROP(used);
        if (dist > (size_t)(out - out0)) FINISH(PSS_INF_BAD_DISTANCE);
        if (!FAST && len > (size_t)(out_end - out)) FINISH(PSS_INF_OVERRUN);
        const uint8_t *src = out - dist;
        uint8_t *const stop = out + len;
        if (dist >= 16 && (FAST || (size_t)(out_end - out) >= len + 16)) {
            /* 16 bytes at a time (two words; source and destination chunks cannot overlap); may
             * scribble up to 15 bytes past `stop`, still inside the block's own output, which
             * later symbols overwrite.  Most matches are done after the first chunk or two. */
            memcpy(out, src, 16);
            if (len > 16) {
                memcpy(out + 16, src + 16, 16);
                if (len > 32) {
                    out += 32;
                    src += 32;
                    do {
                        memcpy(out, src, 16);
                        out += 16;
                        src += 16;
                    } while (out < stop);
                }
            }
            out = stop;
        } else if (dist >= 8 && (FAST || (size_t)(out_end - out) >= len + 8)) {
            do {
                memcpy(out, src, 8);
                out += 8;
                src += 8;
            } while (out < stop);
            out = stop;
        } else if (dist == 1) {
            memset(out, *src, len);
            out = stop;
        } else {
            while (out < stop) *out++ = *src++;
        }
    }
done:
#undef FINISH
#undef XDROP
#undef XREFILL
#undef PUT_LITERAL
    d->in = in;
    d->out = out;
    d->bb = bb;
    d->bl = bl;
    return rc;
}

int pss_inflate_raw(pss_inflater *st, const uint8_t *in, size_t in_len, uint8_t *out0, size_t out_len)
{
    const uint8_t *const in_end = in + in_len;
    uint8_t *out = out0;
    uint8_t *const out_end = out0 + out_len;
    uint64_t bb = 0;
    unsigned bl = 0;

    for (;;) {
        REFILL();
        const unsigned final = BITS(1);
        const unsigned type = (unsigned)(bb >> 1) & 3u;
        DROP(3);
        if (type == 0) { /* stored: skip to the byte boundary, LEN, ~LEN, bytes */
            DROP(bl & 7u);
            REFILL();
            if (bl < 32) return PSS_INF_TRUNCATED;
            const unsigned len = BITS(16), nlen = (unsigned)(bb >> 16) & 0xFFFFu;
            DROP(32);
            if ((len ^ nlen) != 0xFFFFu) return PSS_INF_BAD_BLOCK;
            if (len > (size_t)(out_end - out)) return PSS_INF_OVERRUN;
            /* the bit buffer holds whole bytes here: hand them back to the byte stream */
            in -= bl >> 3;
            bb = 0;
            bl = 0;
            if (len > (size_t)(in_end - in)) return PSS_INF_TRUNCATED;
            memcpy(out, in, len);
            out += len;
            in += len;
        } else if (type == 3) {
            return PSS_INF_BAD_BLOCK;
        } else {
            if (type == 1) fixed_tables(st);
            else {
                const int rc = read_dynamic_header(st, &in, in_end, &bb, &bl);
                if (rc) return rc;
            }
            dstate d = {in, in_end, out, out0, out_end, bb, bl};
            int rc = run_codes(st, &d, 1);
            if (rc == 1) rc = run_codes(st, &d, 0);
            if (rc) return rc;
            in = d.in;
            out = d.out;
            bb = d.bb;
            bl = d.bl;
        }
        if (final) break;
    }
    return out == out_end ? PSS_INF_OK : PSS_INF_SHORT;
}

/* ------------------------------------------------------------------------------------------ */
/* CRC-32                                                                                       */
/* ------------------------------------------------------------------------------------------ */

static uint32_t crc_tab[8][256];
static int crc_use_clmul = 0, crc_use_vclmul = 0;
static int have_clmul(void);
static int have_vclmul(void);

/* tables and CPU probe once, at load time: no lazy initialisation for threads to race on */
__attribute__((constructor)) static void crc_init(void)
{
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
        crc_tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
        for (int t = 1; t < 8; t++) crc_tab[t][i] = (crc_tab[t - 1][i] >> 8) ^ crc_tab[0][crc_tab[t - 1][i] & 0xFFu];
    crc_use_clmul = have_clmul();
    crc_use_vclmul = have_vclmul() && !getenv("PSSBAM_NO_AVX512");
}

static uint32_t crc_slice8(uint32_t c, const uint8_t *p, size_t n)
{
    while (n && ((uintptr_t)p & 7u)) {
        c = (c >> 8) ^ crc_tab[0][(c ^ *p++) & 0xFFu];
        n--;
    }
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        w ^= c;
        c = crc_tab[7][w & 0xFFu] ^ crc_tab[6][(w >> 8) & 0xFFu] ^ crc_tab[5][(w >> 16) & 0xFFu] ^ crc_tab[4][(w >> 24) & 0xFFu] ^
            crc_tab[3][(w >> 32) & 0xFFu] ^ crc_tab[2][(w >> 40) & 0xFFu] ^ crc_tab[1][(w >> 48) & 0xFFu] ^ crc_tab[0][w >> 56];
        p += 8;
        n -= 8;
    }
    while (n--) c = (c >> 8) ^ crc_tab[0][(c ^ *p++) & 0xFFu];
    return c;
}

#if defined(__x86_64__)
#include <cpuid.h>
#include <immintrin.h>

/* Folding with carry-less multiplication (Gopal et al., "Fast CRC Computation for Generic
 * Polynomials Using PCLMULQDQ Instruction", Intel 2009), bit-reflected CRC-32 0xEDB88320:
 * constants are x^(512+32), x^(512-32), x^(128+32), x^(128-32), x^64 mod P (reflected),
 * then Barrett reduction with P' and mu.  len >= 64, multiple of 16. */
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc_clmul(uint32_t crc, const uint8_t *p, size_t len)
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124ll);
    const __m128i poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    __m128i x0 = _mm_loadu_si128((const __m128i *)(p + 0)), x1 = _mm_loadu_si128((const __m128i *)(p + 16));
    __m128i x2 = _mm_loadu_si128((const __m128i *)(p + 32)), x3 = _mm_loadu_si128((const __m128i *)(p + 48));
    x0 = _mm_xor_si128(x0, _mm_cvtsi32_si128((int)crc));
    p += 64;
    len -= 64;
    while (len >= 64) { /* four lanes, each folded across 512 bits */
        __m128i h0 = _mm_clmulepi64_si128(x0, k1k2, 0x11), h1 = _mm_clmulepi64_si128(x1, k1k2, 0x11);
        __m128i h2 = _mm_clmulepi64_si128(x2, k1k2, 0x11), h3 = _mm_clmulepi64_si128(x3, k1k2, 0x11);
        x0 = _mm_clmulepi64_si128(x0, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x00);
        x2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x00);
        x0 = _mm_xor_si128(_mm_xor_si128(x0, h0), _mm_loadu_si128((const __m128i *)(p + 0)));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, h1), _mm_loadu_si128((const __m128i *)(p + 16)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, h2), _mm_loadu_si128((const __m128i *)(p + 32)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, h3), _mm_loadu_si128((const __m128i *)(p + 48)));
        p += 64;
        len -= 64;
    }
    /* four lanes -> one, folding across 128 bits */
    __m128i h = _mm_clmulepi64_si128(x0, k3k4, 0x11);
    x0 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x0, k3k4, 0x00), h), x1);
    h = _mm_clmulepi64_si128(x0, k3k4, 0x11);
    x0 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x0, k3k4, 0x00), h), x2);
    h = _mm_clmulepi64_si128(x0, k3k4, 0x11);
    x0 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x0, k3k4, 0x00), h), x3);
    while (len >= 16) {
        h = _mm_clmulepi64_si128(x0, k3k4, 0x11);
        x0 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x0, k3k4, 0x00), h), _mm_loadu_si128((const __m128i *)p));
        p += 16;
        len -= 16;
    }
    /* 128 -> 64 bits */
    const __m128i mask32 = _mm_setr_epi32(-1, 0, -1, 0);
    __m128i t = _mm_clmulepi64_si128(x0, k3k4, 0x10);
    x0 = _mm_xor_si128(_mm_srli_si128(x0, 8), t);
    t = _mm_srli_si128(x0, 4);
    x0 = _mm_and_si128(x0, mask32);
    x0 = _mm_xor_si128(_mm_clmulepi64_si128(x0, k5, 0x00), t);
    /* Barrett: 64 -> 32 bits */
    t = _mm_and_si128(x0, mask32);
    t = _mm_clmulepi64_si128(t, poly, 0x10);
    t = _mm_and_si128(t, mask32);
    t = _mm_clmulepi64_si128(t, poly, 0x00);
    x0 = _mm_xor_si128(x0, t);
    return (uint32_t)_mm_extract_epi32(x0, 1);
}

static int have_clmul(void)
{
    __builtin_cpu_init();
    return __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
}

/* The same folding 256 bytes at a time: four 512-bit accumulators = sixteen 128-bit lanes, each
 * folded across 2048 bits with x^(2048+32), x^(2048-32) mod P (same derivation as the constants
 * above: bit-reflected remainder shifted left by one).  When fewer than 256 bytes are left the
 * sixteen lanes ARE a 256-byte message congruent to everything consumed so far, so they are
 * handed, followed by the tail, to the 128-bit routine.  len >= 512. */
__attribute__((target("avx512f,avx512vl,vpclmulqdq,pclmul,sse4.1"))) static uint32_t crc_vclmul(uint32_t crc, const uint8_t *p, size_t len)
{
    const __m512i k = _mm512_broadcast_i32x4(_mm_set_epi64x(0x01322d1430ll, 0x011542778all));
    __m512i x0 = _mm512_loadu_si512(p), x1 = _mm512_loadu_si512(p + 64), x2 = _mm512_loadu_si512(p + 128), x3 = _mm512_loadu_si512(p + 192);
    x0 = _mm512_xor_si512(x0, _mm512_zextsi128_si512(_mm_cvtsi32_si128((int)crc)));
    p += 256;
    len -= 256;
    while (len >= 256) {
        const __m512i h0 = _mm512_clmulepi64_epi128(x0, k, 0x11), h1 = _mm512_clmulepi64_epi128(x1, k, 0x11);
        const __m512i h2 = _mm512_clmulepi64_epi128(x2, k, 0x11), h3 = _mm512_clmulepi64_epi128(x3, k, 0x11);
        x0 = _mm512_clmulepi64_epi128(x0, k, 0x00);
        x1 = _mm512_clmulepi64_epi128(x1, k, 0x00);
        x2 = _mm512_clmulepi64_epi128(x2, k, 0x00);
        x3 = _mm512_clmulepi64_epi128(x3, k, 0x00);
        x0 = _mm512_ternarylogic_epi64(x0, h0, _mm512_loadu_si512(p), 0x96);
        x1 = _mm512_ternarylogic_epi64(x1, h1, _mm512_loadu_si512(p + 64), 0x96);
        x2 = _mm512_ternarylogic_epi64(x2, h2, _mm512_loadu_si512(p + 128), 0x96);
        x3 = _mm512_ternarylogic_epi64(x3, h3, _mm512_loadu_si512(p + 192), 0x96);
        p += 256;
        len -= 256;
    }
    uint8_t tail[256 + 256] __attribute__((aligned(64)));
    _mm512_store_si512(tail, x0);
    _mm512_store_si512(tail + 64, x1);
    _mm512_store_si512(tail + 128, x2);
    _mm512_store_si512(tail + 192, x3);
    const size_t whole = len & ~(size_t)15;
    memcpy(tail + 256, p, whole);
    return crc_slice8(crc_clmul(0u, tail, 256 + whole), p + whole, len - whole);
}

static int have_vclmul(void)
{
    unsigned a, b, c, d;
    __builtin_cpu_init();
    if (!__builtin_cpu_supports("avx512f") || !__builtin_cpu_supports("avx512vl") || !have_clmul()) return 0; /* includes OS support for ZMM state */
    if (!__get_cpuid_count(7, 0, &a, &b, &c, &d)) return 0;
    return (c >> 10) & 1u; /* CPUID.7.0:ECX.VPCLMULQDQ */
}
#else
static int have_vclmul(void) { return 0; }
static uint32_t crc_vclmul(uint32_t crc, const uint8_t *p, size_t len) { (void)p; (void)len; return crc; }
static int have_clmul(void) { return 0; }
static uint32_t crc_clmul(uint32_t crc, const uint8_t *p, size_t len) { (void)p; (void)len; return crc; }
#endif

uint32_t pss_crc32(uint32_t crc, const uint8_t *buf, size_t len)
{
    uint32_t c = ~crc;
    if (len >= 512 && crc_use_vclmul) return ~crc_vclmul(c, buf, len);
    if (len >= 64 && crc_use_clmul) {
        const size_t body = len & ~(size_t)15;
        c = crc_clmul(c, buf, body);
        buf += body;
        len -= body;
    }
    return ~crc_slice8(c, buf, len);
}
