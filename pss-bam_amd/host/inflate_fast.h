/*
 * pss-bam_amd/host/inflate_fast.h -- raw DEFLATE (RFC 1951) decoder for BGZF payloads and a
 * CRC-32 (IEEE 802.3, the gzip one) for their trailers.
 *
 * BGZF blocks are small (<= 64 KiB out), self-contained and arrive with their exact output
 * size, which a general-purpose streaming inflate cannot exploit: this decoder works on whole
 * blocks with a 64-bit bit buffer, an 11-bit literal/length table whose entries carry the
 * decoded literal or the length base + extra-bit count, word-wide match copies, and no state
 * machine.  It must produce exactly `out_len` bytes and stop on the final block's end-of-block
 * code; anything else (including every malformed stream) is an error, never a stray access:
 * all reads stay inside [in, in + in_len), all writes inside [out, out + out_len).
 *
 * Part of the BAM feed: replaces zlib's inflate()/crc32() in bam_reader.c.
 * tests/test_host.py checks both against zlib on random, skewed, stored/fixed/dynamic and
 * deliberately corrupted streams.
 */
#ifndef PSSBAM_INFLATE_FAST_H
#define PSSBAM_INFLATE_FAST_H

#include <stddef.h>
#include <stdint.h>

#define PSS_LL_BITS 11
#define PSS_DS_BITS 9

typedef struct pss_inflater {
    uint32_t ll[1u << PSS_LL_BITS];   /* literal/length table, indexed by the next PSS_LL_BITS stream bits */
    uint32_t ds[1u << PSS_DS_BITS];   /* distance table */
    uint32_t cl[1u << 7];             /* code-length code table (dynamic block headers) */
    /* canonical-code data for the (rare) codes longer than the table index */
    uint16_t ll_count[16], ds_count[16], cl_count[16];
    uint16_t ll_sorted[288], ds_sorted[32], cl_sorted[19];
    uint8_t lens[288 + 32];
} pss_inflater;

/* 0 = ok: exactly out_len bytes written and the stream's final block ended.
 * negative = malformed / truncated / wrong size (see PSS_INF_* below). */
int pss_inflate_raw(pss_inflater *st, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len);

enum { PSS_INF_OK = 0, PSS_INF_TRUNCATED = -1, PSS_INF_BAD_BLOCK = -2, PSS_INF_BAD_CODES = -3, PSS_INF_BAD_SYMBOL = -4,
       PSS_INF_BAD_DISTANCE = -5, PSS_INF_OVERRUN = -6, PSS_INF_SHORT = -7 };

/* CRC-32 of buf[0..len) continuing from `crc` (0 to start), same values as zlib's crc32().
 * Uses carry-less multiplication when the CPU has it, slicing-by-8 tables otherwise. */
uint32_t pss_crc32(uint32_t crc, const uint8_t *buf, size_t len);

#endif
