/*
 * pss-bam_amd/csrc/synth_model.h -- the synthetic workload of SURVEY 8d as a pure function.
 *
 * Everything (reference base at (contig, pos); every field of alignment record i) is a
 * function of a seed and an index, built on splitmix64, so the SAME source compiled by
 * hipcc for the device and by the host compiler yields identical bytes: bench.py
 * generates 200 M reads straight into HBM, tests regenerate any sub-range on the host
 * and hand it to the oracle.  Workload/bench infrastructure -- not part of the product
 * path, and free of any reference code.
 *
 * Plain C subset (also included from .c files).
 */
#ifndef PSSBAM_SYNTH_MODEL_H
#define PSSBAM_SYNTH_MODEL_H

#include <stdint.h>

#if defined(__HIPCC__)
#define SYN_HD __host__ __device__ static inline
#else
#define SYN_HD static inline
#endif

#define SYN_MAX_CONTIGS 32

typedef struct synth_cfg {
    uint64_t seed;
    uint32_t n_contigs;
    uint32_t name_mode;       /* 0: single contig "chrS"; 1: chr1..chr22, chrX, chrY */
    uint64_t contig_len[SYN_MAX_CONTIGS];
    uint64_t n_reads;         /* size of the whole stream (positions depend on it)       */
    uint32_t len_min, len_max;/* read length uniform in [len_min, len_max]               */
    uint32_t sorted;          /* 1 coordinate-sorted stream, 0 pseudo-random order       */
    uint32_t cigar_mix;       /* 0: always <L>M; 1: C4 mix (S / I / D variants)          */
    uint32_t damage;          /* 1: aDNA C->T / G->A end damage                          */
    uint32_t sub_per_64k;     /* substitution probability * 65536 (1 % = 655)            */
    uint32_t dup_per_1k;      /* reads flagged 0x400, per 1000                           */
    uint32_t lowmq_per_1k;    /* reads with MAPQ < 20, per 1000                          */
    uint32_t n_run_len;       /* every 8192-bp block holds one run of this many N (41 = 0.5 %) */
    uint32_t pad_;
    /* derived by synth_cfg_finish(): */
    uint64_t usable_first[SYN_MAX_CONTIGS + 1]; /* prefix sums of usable start positions */
    uint64_t perm_mul, perm_add;                /* i -> (i*mul + add) mod n_reads, gcd(mul, n)=1 */
} synth_cfg;

SYN_HD uint64_t syn_mix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

SYN_HD uint64_t syn_hash3(uint64_t seed, uint64_t a, uint64_t b)
{
    return syn_mix(syn_mix(seed ^ (a * 0xD6E8FEB86659FD93ull)) ^ (b * 0xCA5A826395121157ull));
}

/* reference base (0..3 = ACGT, 4 = N) at (contig, pos) */
SYN_HD uint32_t syn_ref_code(const synth_cfg *c, uint32_t contig, uint64_t pos)
{
    if (c->n_run_len) {
        const uint64_t blk = pos >> 13;
        const uint64_t off = syn_hash3(c->seed ^ 0x4E4E4E4Eull, contig, blk) % (8192u - c->n_run_len);
        const uint64_t in = pos & 8191u;
        if (in >= off && in < off + c->n_run_len) return 4u;
    }
    const uint64_t h = syn_hash3(c->seed, contig, pos >> 5);
    return (uint32_t)(h >> (2u * (uint32_t)(pos & 31u))) & 3u;
}

/* soft-masking of the FASTA text (upper-cased again by any loader): ~45 % lower case, in runs */
SYN_HD int syn_ref_is_lower(const synth_cfg *c, uint32_t contig, uint64_t pos)
{
    return (syn_hash3(c->seed ^ 0x6C6F7765ull, contig, pos >> 9) % 100u) < 45u;
}

typedef struct synth_read {
    uint32_t contig;
    uint64_t s;          /* 0-based leftmost reference position                          */
    uint32_t L;          /* l_seq                                                        */
    uint32_t flag, mapq;
    uint32_t n_cigar;
    uint32_t cigar[3];   /* BAM encoding len<<4|op                                       */
    uint32_t rec_bytes;  /* 4 + block_size                                               */
} synth_read;

#define SYN_NAME_LEN 12u /* "s" + 10 digits + NUL */

/* The unsorted stream is a fixed permutation of the sorted one: slot i holds sorted read
 * syn_read_index(i).  (Tables are therefore identical for both orders.) */
SYN_HD uint64_t syn_read_index(const synth_cfg *c, uint64_t i)
{
    return c->sorted ? i : (i * c->perm_mul + c->perm_add) % c->n_reads;
}

/* fields of SORTED read i (callers map slots through syn_read_index first) */
SYN_HD void syn_read_fields(const synth_cfg *c, uint64_t i, synth_read *r)
{
    const uint64_t hi = syn_hash3(c->seed ^ 0x52454144ull, i, 0);
    const uint64_t hj = syn_hash3(c->seed ^ 0x52454144ull, i, 1);
    const uint32_t span = c->len_max - c->len_min + 1u;
    const uint64_t U = c->usable_first[c->n_contigs];
    uint64_t u;
    uint32_t k;
    r->L = c->len_min + (uint32_t)(hi % span);
    if (U >= c->n_reads) {
        /* one start per stride bucket -> monotone in i */
        const uint64_t stride = U / c->n_reads;
        u = i * stride + (hi >> 20) % stride;
    } else {
        /* more reads than positions: several reads per start, still monotone */
        u = (uint64_t)(((unsigned __int128)i * U) / c->n_reads);
    }
    if (u >= U) u = U - 1;
    for (k = 0; k + 1 < c->n_contigs && u >= c->usable_first[k + 1]; k++) {}
    r->contig = k;
    r->s = 2u + (u - c->usable_first[k]);
    r->flag = (hj & 1u) ? 0x10u : 0u;
    if ((uint32_t)((hj >> 8) % 1000u) < c->dup_per_1k) r->flag |= 0x400u;
    r->mapq = ((uint32_t)((hj >> 24) % 1000u) < c->lowmq_per_1k) ? (uint32_t)((hj >> 40) % 20u) : 37u;
    r->n_cigar = 1;
    r->cigar[0] = r->L << 4; /* <L>M */
    r->cigar[1] = r->cigar[2] = 0;
    if (c->cigar_mix && r->L >= 8u) {
        const uint32_t v = (uint32_t)((hj >> 44) % 1000u);
        const uint32_t a = 1u + (uint32_t)((hj >> 54) % (r->L / 2u));
        const uint32_t g = 1u + (uint32_t)((hj >> 60) % 3u);
        if (v < 700u) {
        } else if (v < 800u) { r->n_cigar = 2; r->cigar[0] = (a << 4) | 4u; r->cigar[1] = (r->L - a) << 4; }
        else if (v < 850u) { r->n_cigar = 2; r->cigar[0] = (r->L - a) << 4; r->cigar[1] = (a << 4) | 4u; }
        else if (v < 925u) { r->n_cigar = 3; r->cigar[0] = a << 4; r->cigar[1] = (g << 4) | 1u; r->cigar[2] = (r->L - a - g) << 4; }
        else { r->n_cigar = 3; r->cigar[0] = a << 4; r->cigar[1] = (g << 4) | 2u; r->cigar[2] = (r->L - a) << 4; }
    }
    r->rec_bytes = 4u + 32u + SYN_NAME_LEN + 4u * r->n_cigar + ((r->L + 1u) >> 1) + r->L;
}

/* base j of SEQ as a 2-bit code (4 = N): reference base + substitutions + end damage */
SYN_HD uint32_t syn_read_code(const synth_cfg *c, uint64_t i, const synth_read *r, uint32_t j)
{
    uint32_t b = syn_ref_code(c, r->contig, r->s + j);
    if (b > 3u) return 4u;
    {
        const uint64_t h = syn_hash3(c->seed ^ 0x53554253ull, i, j >> 2);
        const uint32_t piece = (uint32_t)(h >> (16u * (j & 3u))) & 0xFFFFu;
        if (piece < c->sub_per_64k) b = (b + 1u + piece % 3u) & 3u;
    }
    if (c->damage) {
        /* stored orientation: C->T decaying from the left end, G->A from the right end */
        const uint32_t dl = j, dr = r->L - 1u - j;
        if (dl < 8u && b == 1u) {
            const uint32_t p = (uint32_t)(syn_hash3(c->seed ^ 0x44414D31ull, i, dl) & 0xFFFFu);
            if (p < (19661u >> dl)) b = 3u; /* 0.3 * 0.5^dl */
        }
        if (dr < 8u && b == 2u) {
            const uint32_t p = (uint32_t)(syn_hash3(c->seed ^ 0x44414D32ull, i, dr) & 0xFFFFu);
            if (p < (19661u >> dr)) b = 0u;
        }
    }
    return b;
}

/* 2-bit code -> BAM 4-bit code (A1 C2 G4 T8, N15) */
SYN_HD uint32_t syn_nib(uint32_t code) { return code < 4u ? (1u << code) : 15u; }

SYN_HD uint32_t syn_reg2bin(uint64_t beg, uint64_t end)
{
    --end;
    if (beg >> 14 == end >> 14) return (uint32_t)(((1u << 15) - 1u) / 7u + (beg >> 14));
    if (beg >> 17 == end >> 17) return (uint32_t)(((1u << 12) - 1u) / 7u + (beg >> 17));
    if (beg >> 20 == end >> 20) return (uint32_t)(((1u << 9) - 1u) / 7u + (beg >> 20));
    if (beg >> 23 == end >> 23) return (uint32_t)(((1u << 6) - 1u) / 7u + (beg >> 23));
    if (beg >> 26 == end >> 26) return (uint32_t)(((1u << 3) - 1u) / 7u + (beg >> 26));
    return 0u;
}

SYN_HD void syn_put32(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}

/* writes BAM alignment record i (r->rec_bytes bytes) at out */
SYN_HD void syn_write_record(const synth_cfg *c, uint64_t i, const synth_read *r, uint8_t *out)
{
    uint32_t ref_span = 0, k, j;
    uint64_t d = i;
    uint8_t *p = out;
    for (k = 0; k < r->n_cigar; k++) {
        const uint32_t op = r->cigar[k] & 15u;
        if (op == 0u || op == 2u) ref_span += r->cigar[k] >> 4;
    }
    syn_put32(p, r->rec_bytes - 4u);
    syn_put32(p + 4, r->contig);
    syn_put32(p + 8, (uint32_t)r->s);
    syn_put32(p + 12, SYN_NAME_LEN | (r->mapq << 8) | (syn_reg2bin(r->s, r->s + (ref_span ? ref_span : 1u)) << 16));
    syn_put32(p + 16, r->n_cigar | (r->flag << 16));
    syn_put32(p + 20, r->L);
    syn_put32(p + 24, 0xFFFFFFFFu); /* next_refID -1 */
    syn_put32(p + 28, 0xFFFFFFFFu); /* next_pos -1   */
    syn_put32(p + 32, 0u);          /* tlen          */
    p += 36;
    p[0] = 's';
    for (k = 10; k >= 1; k--) { p[k] = (uint8_t)('0' + d % 10u); d /= 10u; }
    p[11] = 0;
    p += SYN_NAME_LEN;
    for (k = 0; k < r->n_cigar; k++, p += 4) syn_put32(p, r->cigar[k]);
    for (j = 0; j < r->L; j += 2) {
        const uint32_t hi = syn_nib(syn_read_code(c, i, r, j));
        const uint32_t lo = (j + 1u < r->L) ? syn_nib(syn_read_code(c, i, r, j + 1u)) : 0u;
        *p++ = (uint8_t)((hi << 4) | lo);
    }
    for (j = 0; j < r->L; j++) *p++ = 40; /* 'I' */
}

#endif /* PSSBAM_SYNTH_MODEL_H */
