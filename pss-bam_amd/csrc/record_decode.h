// pss-bam_amd/csrc/record_decode.h -- device-side BAM record decode + the two tools'
// filters, shared by every tally kernel.
//
// What is computed here is the *text-equivalent* reading of a binary BAM record: the
// reference never sees BAM, it sees what `samtools view` prints and line2saml parses
// (/root/reference/pss-bam.c:148-162, sam-parse.c:36-68).  The mapping (SURVEY 8a row
// a2) is:
//     RNAME  = name of refID ('*' when -1)        POS   = pos + 1
//     CIGAR  = "<len><op>..." ('*' when n_cigar_op == 0)
//     SEQ    = 4-bit codes through "=ACMGRSVTWYHKDBN" ('*' when l_seq == 0)
//     QUAL   = phred+33, or '*' when the first byte is 0xFF (or l_seq == 0)
//  => strlen(SEQ) = l_seq ? l_seq : 1; line2saml rejects the line (returns 1) when
//     strlen(QUAL) differs, i.e. when QUAL is '*' but l_seq >= 2 (sam-parse.c:50).
//
// No CPU fallback exists for any of this: these functions are __device__ only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pssbam {

// ---- FLAG bits (sam-parse.c:53-64) --------------------------------------------------
constexpr uint32_t FL_PAIRED = 0x1, FL_PROPER = 0x2, FL_UNMAP = 0x4, FL_MUNMAP = 0x8;
constexpr uint32_t FL_REVERSE = 0x10, FL_READ1 = 0x40, FL_READ2 = 0x80;
constexpr uint32_t FL_REJECT = 0x4 | 0x100 | 0x200 | 0x400 | 0x800;  // pss-bam.c:412-416, fragkon.c:142-146

// ---- kernel parameter block (lives in kernarg/constant space) ------------------------
struct TallyParams {
    const uint8_t *recs;          // record block
    const uint32_t *offs;         // n_recs + 1 offsets into recs
    uint32_t n_recs;
    const uint32_t *n_recs_dev;   // non-NULL: the record count is read from device memory (blocks indexed on the device)
    uint64_t recs_bytes;          // bytes of the record block (= offs[n_recs])
    uint32_t tally_mask;          // PSSBAM_TALLY_*
    const uint8_t *genome;        // all contigs, 1 stored byte/base (enc_byte), padded between
    // the same array at 4 bits/base for the tiled kernel's end windows (half the HBM lines):
    // position p = bits 4*(p%8) of dword p/8; nibble 0..3 = A C G T, 4 + (in -U) + 2*(in -D) otherwise
    const uint32_t *genome4;
    uint32_t acgt_ctx;            // bit 2v: base v in -U, bit 2v+1: in -D (v = 0..3 for A C G T)
    // BAM refID -> where its contig lies, resolved once per header by find_seq semantics:
    // ref_info[refID] = {gbase lo, gbase hi, contig length, found ? 1 : 0}; entry n_ref is the
    // one for RNAME "*" (refID -1; found only if a contig is literally named "*")
    const uint4 *ref_info;
    int32_t n_ref;
    // pss-bam options (pss-bam.c:12-18)
    int32_t N;
    uint32_t pss_min_mq;
    uint32_t pss_min_len, pss_max_len;  // clamped to u32 on the host (L is a u32)
    uint32_t pss_len_never;             // -l beyond any u32: no length can pass
    uint32_t pss_merged_only;
    uint32_t up_mask[8], down_mask[8];  // strchr(UP_CTX/DOWN_CTX, c) as 256-bit sets over STORED bytes
    // fragkon options (fragkon.c:14-18)
    int32_t K;
    uint32_t fk_min_mq;
    uint32_t fk_min_len, fk_max_len, fk_len_never;
    uint32_t fk_merged_only;
    // -R read group (NULL = keep all)
    const uint8_t *rg;
    uint32_t rg_len;
    // output: one block of u64 counters [fwd | rev | k5 | k3 | stats]
    unsigned long long *counters;
    uint32_t off_rev, off_k5, off_k3, off_stats;
    // tiled kernel geometry
    uint32_t reads_per_tile;      // T, multiple of 64
    uint32_t prefix_pieces;       // 16-byte pieces of each record the tiled kernel stages in LDS
    uint32_t row_base;            // tiled kernel: this launch tallies table rows row_base .. row_base+31
    uint32_t xcd_map;             // tiled kernel: XCD-contiguous workgroup -> tile mapping
    uint32_t ablate;              // diagnostics: phases to skip (results are wrong when non-zero)
    uint32_t *scratch;            // tiled kernel: per-workgroup partial tables (SCRATCH_WORDS each)
};

// stats slots, must match include/pssbam_hip.h.  The kernels count EVENTS only: every launch
// credits n_recs to RECORDS / PSS_OK / KMER_OK up front and each record that ends otherwise moves
// one unit from the OK slot(s) to its own slot (wrapping u64 arithmetic), so the common case
// costs no instruction at all.
enum { ST_RECORDS = 0, ST_RG_DROPPED, ST_PARSE_SKIP, ST_NO_CONTIG, ST_PSS_OK, ST_PSS_FILTERED,
       ST_KMER_OK, ST_KMER_FILTERED, ST_KMER_FAIL, ST_SLOW_PATH, ST_USED };

// ---- byte sources -------------------------------------------------------------------
// A record is read through one of these; both tolerate any alignment and never touch
// bytes outside [0, limit) of the record they were built for.
struct GlobalBytes {
    const uint8_t *p;
    __device__ __forceinline__ uint32_t u8(uint32_t o) const { return p[o]; }
    __device__ __forceinline__ uint32_t u16(uint32_t o) const { return p[o] | (uint32_t(p[o + 1]) << 8); }
    __device__ __forceinline__ uint32_t u32(uint32_t o) const {
        return p[o] | (uint32_t(p[o + 1]) << 8) | (uint32_t(p[o + 2]) << 16) | (uint32_t(p[o + 3]) << 24);
    }
};

// LDS window: dword-aligned reads + v_alignbyte.  Addressed as (16-byte aligned LDS base,
// byte offset) so the alignment comes from the integer offset and the pointer keeps its
// address space (ds_read, not flat).  The window carries >= 8 bytes of slack behind its last
// byte so the second dword of an unaligned read is always in bounds.
struct LdsBytes {
    const uint8_t *base;  // 16-byte aligned __shared__ storage
    uint32_t off;         // byte offset of the record inside it
    __device__ __forceinline__ uint32_t u8(uint32_t o) const { return base[off + o]; }
    __device__ __forceinline__ uint32_t u32(uint32_t o) const {
        const uint32_t a = off + o;
        const uint32_t *q = (const uint32_t *)(base + (a & ~3u));
        return __builtin_amdgcn_alignbyte(q[1], q[0], a & 3u);
    }
    __device__ __forceinline__ uint32_t u16(uint32_t o) const { return u32(o) & 0xFFFFu; }
};

// ---- decoded fixed part of one record -------------------------------------------------
struct RecHdr {
    int32_t ref_id, pos, tlen;
    uint32_t mapq, flag, n_cigar, l_seq, cigar0;
    uint32_t seq_off, qual_off, aux_off;  // byte offsets from the record's block_size word
    uint32_t rec_len;                     // 4 + block_size as given by the offset index
    bool well_formed;                     // variable-length parts fit in rec_len
};

template <class Src>
__device__ __forceinline__ RecHdr decode_hdr(const Src &src, uint32_t rec_len) {
    RecHdr h;
    h.rec_len = rec_len;
    h.well_formed = rec_len >= 36;
    if (!h.well_formed) {
        h.ref_id = -1; h.pos = -1; h.tlen = 0; h.mapq = 0; h.flag = FL_UNMAP; h.n_cigar = 0; h.l_seq = 0;
        h.cigar0 = 0; h.seq_off = h.qual_off = h.aux_off = rec_len;
        return h;
    }
    // SAM spec 4.2: block_size, refID, pos, l_read_name:8 mapq:8 bin:16, n_cigar_op:16 flag:16,
    //               l_seq, next_refID, next_pos, tlen
    h.ref_id = (int32_t)src.u32(4);
    h.pos = (int32_t)src.u32(8);
    const uint32_t w3 = src.u32(12);
    const uint32_t w4 = src.u32(16);
    h.l_seq = src.u32(20);
    h.tlen = (int32_t)src.u32(32);
    const uint32_t l_read_name = w3 & 0xFFu;
    h.mapq = (w3 >> 8) & 0xFFu;
    h.n_cigar = w4 & 0xFFFFu;
    h.flag = w4 >> 16;
    const uint64_t cig_off = 36ull + l_read_name;
    const uint64_t seq_off = cig_off + 4ull * h.n_cigar;
    const uint64_t qual_off = seq_off + ((uint64_t(h.l_seq) + 1) >> 1);
    const uint64_t aux_off = qual_off + h.l_seq;
    h.well_formed = aux_off <= rec_len;
    if (!h.well_formed) {
        h.n_cigar = 0; h.l_seq = 0; h.cigar0 = 0; h.flag |= FL_UNMAP;
        h.seq_off = h.qual_off = h.aux_off = rec_len;
        return h;
    }
    h.seq_off = (uint32_t)seq_off;
    h.qual_off = (uint32_t)qual_off;
    h.aux_off = (uint32_t)aux_off;
    h.cigar0 = h.n_cigar ? src.u32((uint32_t)cig_off) : 0u;
    return h;
}

// Same for a record staged in LDS: the 36 fixed bytes arrive as ten aligned dwords in one batch
// (one wait) and are funnel-shifted into place, instead of seven unaligned two-read fetches.
__device__ __forceinline__ RecHdr decode_hdr_lds(const LdsBytes &src, uint32_t rec_len) {
    const uint32_t sh = src.off & 3u;
    const uint32_t *q = (const uint32_t *)(src.base + (src.off & ~3u));
    uint32_t r[10], w[9];
#pragma unroll
    for (int k = 0; k < 10; k++) r[k] = q[k];
#pragma unroll
    for (int k = 0; k < 9; k++) w[k] = __builtin_amdgcn_alignbyte(r[k + 1], r[k], sh);
    RecHdr h;
    h.rec_len = rec_len;
    h.ref_id = (int32_t)w[1];
    h.pos = (int32_t)w[2];
    h.mapq = (w[3] >> 8) & 0xFFu;
    h.n_cigar = w[4] & 0xFFFFu;
    h.flag = w[4] >> 16;
    h.l_seq = w[5];
    h.tlen = (int32_t)w[8];
    const uint64_t cig_off = 36ull + (w[3] & 0xFFu);
    const uint64_t seq_off = cig_off + 4ull * h.n_cigar;
    const uint64_t qual_off = seq_off + ((uint64_t(h.l_seq) + 1) >> 1);
    const uint64_t aux_off = qual_off + h.l_seq;
    h.well_formed = rec_len >= 36u && aux_off <= rec_len;
    // a malformed record decodes as "unmapped, no bases"; offsets clamped so nothing is read
    // outside the record
    h.seq_off = h.well_formed ? (uint32_t)seq_off : rec_len;
    h.qual_off = h.well_formed ? (uint32_t)qual_off : rec_len;
    h.aux_off = h.well_formed ? (uint32_t)aux_off : rec_len;
    const uint32_t c0 = src.u32(h.well_formed ? (uint32_t)cig_off : 0u);
    h.cigar0 = (h.well_formed && h.n_cigar) ? c0 : 0u;
    if (!h.well_formed) { h.n_cigar = 0; h.l_seq = 0; h.flag |= FL_UNMAP; h.ref_id = -1; }
    return h;
}

// The same in 32-bit arithmetic (tally_compact: the short-read kernel is VALU-bound).  l_seq above
// 2^30 is declared malformed first -- a read of a gigabase is far outside the reference's own
// 2047-character fields (precondition P2) -- so no sum below can wrap.
__device__ __forceinline__ RecHdr decode_hdr_lds32(const LdsBytes &src, uint32_t rec_len) {
    const uint32_t sh = src.off & 3u;
    const uint32_t *q = (const uint32_t *)(src.base + (src.off & ~3u));
    uint32_t r[10], w[9];
#pragma unroll
    for (int k = 0; k < 10; k++) r[k] = q[k];
#pragma unroll
    for (int k = 0; k < 9; k++) w[k] = __builtin_amdgcn_alignbyte(r[k + 1], r[k], sh);
    RecHdr h;
    h.rec_len = rec_len;
    h.ref_id = (int32_t)w[1];
    h.pos = (int32_t)w[2];
    h.mapq = (w[3] >> 8) & 0xFFu;
    h.n_cigar = w[4] & 0xFFFFu;
    h.flag = w[4] >> 16;
    h.l_seq = w[5];
    h.tlen = (int32_t)w[8];
    const uint32_t cig_off = 36u + (w[3] & 0xFFu);
    const uint32_t seq_off = cig_off + 4u * h.n_cigar;
    const uint32_t lq = min(h.l_seq, 1u << 30);
    const uint32_t qual_off = seq_off + ((lq + 1u) >> 1);
    const uint32_t aux_off = qual_off + lq;
    h.well_formed = rec_len >= 36u && h.l_seq <= (1u << 30) && aux_off <= rec_len;
    h.seq_off = h.well_formed ? seq_off : rec_len;
    h.qual_off = h.well_formed ? qual_off : rec_len;
    h.aux_off = h.well_formed ? aux_off : rec_len;
    const uint32_t c0 = src.u32(h.well_formed ? cig_off : 0u);
    h.cigar0 = (h.well_formed && h.n_cigar) ? c0 : 0u;
    if (!h.well_formed) { h.n_cigar = 0; h.l_seq = 0; h.flag |= FL_UNMAP; h.ref_id = -1; }
    return h;
}

// `samtools view -r RG`: keep the record iff it carries RG:Z:<rg>.  Walks the aux
// fields (SAM spec 4.2.4); a field that runs past the record ends the walk.
template <class Src>
__device__ bool has_read_group(const Src &src, const RecHdr &h, const uint8_t *rg, uint32_t rg_len) {
    uint32_t o = h.aux_off;
    const uint32_t end = h.rec_len;
    while (o + 3 <= end) {
        const uint32_t t0 = src.u8(o), t1 = src.u8(o + 1), ty = src.u8(o + 2);
        o += 3;
        uint32_t sz;
        switch (ty) {
        case 'A': case 'c': case 'C': sz = 1; break;
        case 's': case 'S': sz = 2; break;
        case 'i': case 'I': case 'f': sz = 4; break;
        case 'Z': case 'H': {
            const bool is_rg = (t0 == 'R' && t1 == 'G' && ty == 'Z');
            uint32_t n = 0;
            bool same = is_rg;
            while (o + n < end) {
                const uint32_t c = src.u8(o + n);
                if (c == 0) break;
                if (same) same = (n < rg_len) && (rg[n] == c);
                n++;
            }
            if (o + n >= end) return false;            // unterminated string
            if (is_rg) return same && n == rg_len;      // first RG tag decides, like bam_aux_get
            sz = n + 1;
            break;
        }
        case 'B': {
            if (o + 5 > end) return false;
            const uint32_t sub = src.u8(o);
            const uint32_t cnt = src.u32(o + 1);
            const uint32_t es = (sub == 'c' || sub == 'C') ? 1u : (sub == 's' || sub == 'S') ? 2u : 4u;
            const uint64_t tot = 5ull + uint64_t(cnt) * es;
            if (tot > end - o) return false;
            sz = (uint32_t)tot;
            break;
        }
        default: return false;
        }
        if (sz > end - o) return false;
        o += sz;
    }
    return false;
}

// ---- base codes ----------------------------------------------------------------------
// Device-internal genome encoding: the byte permutation that swaps 'A'<->0, 'C'<->1,
// 'G'<->2, 'T'<->3 and leaves every other value where it is (applied once at upload,
// after the toupper() fold the reference applies at load and again in process_aln,
// fasta-genome-io.c:127, pss-bam.c:424).  A stored byte < 4 IS the base's 2-bit code
// (A0 C1 G2 T3: pss-bam.c:205-251 pair order, kmer.c:190-208); anything else is "not
// ACGT".  It is a bijection, so the -U/-D membership sets are simply permuted the same
// way on the host and nothing about the original byte is lost.
__host__ __device__ __forceinline__ uint32_t enc_byte(uint32_t b) {
    return b == 'A' ? 0u : b == 'C' ? 1u : b == 'G' ? 2u : b == 'T' ? 3u
         : b == 0u ? 'A' : b == 1u ? 'C' : b == 2u ? 'G' : b == 3u ? 'T' : b;
}
__device__ __forceinline__ uint32_t ref_code(uint32_t stored) { return stored < 4u ? stored : 4u; }
// BAM 4-bit code -> same scale: 1(A) 2(C) 4(G) 8(T); every other code prints as a
// non-ACGT letter ("=MRSVWYHKDBN") and never matches a pair string.
__device__ __forceinline__ uint32_t nib_code(uint32_t n) {
    // 2-bit entries for n = 1,2,4,8 -> 0,1,2,3 ; validity bitmap 0x0116 = bits 1,2,4,8
    return ((0x0116u >> n) & 1u) ? ((0x00030210u >> (2u * n)) & 3u) : 4u;
}
// one byte of do_revcomp (pss-bam.c:60-79) on a stored (encoded, upper-case) genome byte:
// ACGT complement, everything else unchanged
__device__ __forceinline__ uint32_t comp_stored(uint32_t stored) { return stored < 4u ? 3u - stored : stored; }
__device__ __forceinline__ bool in_set(const uint32_t (&m)[8], uint32_t b) { return (m[(b >> 5) & 7] >> (b & 31)) & 1u; }

template <class Src>
__device__ __forceinline__ uint32_t read_nibble(const Src &src, const RecHdr &h, uint32_t i) {
    // base i of SEQ; i >= l_seq (only reachable for out-of-contract records, P3) reads as 0
    if (i >= h.l_seq) return 0u;
    const uint32_t b = src.u8(h.seq_off + (i >> 1));
    return (i & 1u) ? (b & 0xFu) : (b >> 4);
}

// ---- what to do with one record ---------------------------------------------------------
// status of a record, as the reference's main loops would classify it
enum : uint32_t { RS_LIVE = 0, RS_RG_DROPPED = 1, RS_PARSE_SKIP = 2, RS_NO_CONTIG = 3 };

struct Plan {
    uint32_t status;         // RS_*
    bool live;               // record reached process_aln with a known contig
    uint64_t gbase;          // genome offset of the contig's first base
    int32_t s;               // 0-based alignment start (BAM pos)
    bool rev;                // FLAG 0x10
    uint32_t flag;
    // pss
    bool pss_cand;           // passed every pss filter that does not look at the genome
    bool pss_fwd, pss_rev;   // which table(s) this read is tallied into (set by plan_finish_pss)
    uint32_t L;              // pss effective length: |TLEN| when paired else strlen(SEQ)
    // fragkon
    bool fk5, fk3;           // which k-mer table(s) this read may add to
    uint32_t Lk;             // strlen(SEQ)
};

// Text-equivalence + contig lookup + the filters of the enabled tool(s), everything that can be
// decided from the record alone.  No genome access.  Straight-line predicated code in 32-bit
// arithmetic (the lanes of a wave hold different records; early returns would only serialise).
// reference-table providers for plan_head
struct RefsGlobal {  // straight from device memory
    const uint4 *t;
    __device__ __forceinline__ uint4 get(uint32_t i) const { return t[i]; }
};
struct RefsLdsCached {  // first `n_cached` entries (and the "*" entry, kept at index n_cached) in LDS
    const uint4 *lds;
    const uint4 *glob;
    uint32_t n_cached, n_ref;
    __device__ __forceinline__ uint4 get(uint32_t i) const {
        if (i < n_cached) return lds[i];
        if (i == n_ref) return lds[n_cached];
        return glob[i];
    }
};

template <bool DO_PSS, bool DO_KMER, bool MAY_HAVE_RG = true, class Src, class Refs>
__device__ __forceinline__ Plan plan_head(const TallyParams &P, const Src &src, const RecHdr &h, const Refs &refs) {
    Plan pl;
    pl.pss_fwd = pl.pss_rev = false;
    pl.flag = h.flag;

    bool rg_drop = false;
    if (MAY_HAVE_RG && P.rg) rg_drop = !has_read_group(src, h, P.rg, P.rg_len);  // uniform branch (kernel argument)

    // line2saml: strlen(SEQ) vs strlen(QUAL)  (sam-parse.c:50)
    const uint32_t l_text = h.l_seq ? h.l_seq : 1u;
    const uint32_t q0 = (h.well_formed && h.l_seq) ? src.u8(h.qual_off) : 0xFFu;
    const bool parse_skip = !h.well_formed || (q0 == 0xFFu && l_text != 1u);

    // find_seq(genome, RNAME)  (pss-bam.c:393-396, fragkon.c:124-127), pre-resolved per refID
    const bool rid_ok = (uint32_t)h.ref_id < (uint32_t)P.n_ref;
    const uint4 ri = refs.get(rid_ok ? (uint32_t)h.ref_id : (uint32_t)P.n_ref);
    const bool found = ri.w != 0u && (rid_ok || h.ref_id == -1);
    const uint32_t glen = ri.z;
    pl.gbase = ((uint64_t)ri.y << 32) | ri.x;
    const int32_t contig = found ? 0 : -1;
    pl.status = rg_drop ? RS_RG_DROPPED : parse_skip ? RS_PARSE_SKIP : contig < 0 ? RS_NO_CONTIG : RS_LIVE;
    const bool live = pl.status == RS_LIVE;
    pl.live = live;
    pl.s = h.pos;  // POS-1
    pl.rev = (h.flag & FL_REVERSE) != 0;
    const bool paired = (h.flag & FL_PAIRED) != 0;
    // cigar_ok: exactly "<len>M"; flags: none of 0x4 0x100 0x200 0x400 0x800; s >= 0 (the `s >= 2`
    // / `s >= k/2` tests below imply it); both tools share these
    const bool common = live && h.n_cigar == 1u && (h.cigar0 & 0xFu) == 0u && !(h.flag & FL_REJECT) && h.pos >= 0;
    const uint32_t op_len = h.cigar0 >> 4;
    const uint32_t s = (uint32_t)h.pos;
    const bool pair_ok = (h.flag & (FL_PROPER | FL_MUNMAP)) == FL_PROPER;

    pl.L = pl.Lk = l_text;
    pl.pss_cand = pl.fk5 = pl.fk3 = false;
    if (DO_PSS) {  // process_aln filters, pss-bam.c:401-420
        const uint32_t L = paired ? (uint32_t)(h.tlen < 0 ? -(int64_t)h.tlen : (int64_t)h.tlen) : l_text;
        pl.L = L;
        // s >= 2 && s + L + 2 <= glen, without overflow: L <= glen - 4 first (op_len == L < 2^28)
        bool ok = common && op_len == L && glen >= 4u && L <= glen - 4u && s - 2u <= glen - 4u - L;
        ok = ok && !(h.mapq < P.pss_min_mq);
        ok = ok && !P.pss_len_never && L >= P.pss_min_len && L <= P.pss_max_len && L >= (uint32_t)P.N;
        ok = ok && !(P.pss_merged_only && paired);
        // paired reads additionally need proper_pair && !munmap and a mate number (:450-452,:460,:471)
        ok = ok && (!paired || (pair_ok && (h.flag & (FL_READ1 | FL_READ2))));
        pl.pss_cand = ok;
    }
    if (DO_KMER) {  // process_aln filters, fragkon.c:129-146 (+ precondition P4: start >= k/2)
        const uint32_t L = l_text, okk = (uint32_t)P.K / 2u;
        // s >= k/2 && s + L + k/2 <= glen
        bool ok = common && op_len == L && glen >= 2u * okk && L <= glen - 2u * okk && s - okk <= glen - 2u * okk - L;
        ok = ok && h.mapq >= P.fk_min_mq;
        ok = ok && !P.fk_len_never && L >= P.fk_min_len && L <= P.fk_max_len;
        // unpaired: both ends (:149-183, no -m test); paired: needs !MERGED_ONLY && proper && !munmap,
        // read1 -> 5' only, else read2 -> 3' only (:187-213)
        const bool pok = ok && paired && !P.fk_merged_only && pair_ok;
        pl.fk5 = (ok && !paired) || (pok && (h.flag & FL_READ1));
        pl.fk3 = (ok && !paired) || (pok && !(h.flag & FL_READ1) && (h.flag & FL_READ2));
    }
    return pl;
}

// -U / -D membership providers over STORED genome bytes
struct CtxMasks {  // straight from the kernel arguments
    const TallyParams &P;
    __device__ __forceinline__ bool up(uint32_t b) const { return in_set(P.up_mask, b); }
    __device__ __forceinline__ bool down(uint32_t b) const { return in_set(P.down_mask, b); }
};
struct CtxLds {  // 256-byte LDS table: bit 0 = in UP_CTX, bit 1 = in DOWN_CTX
    const uint8_t *f;
    __device__ __forceinline__ bool up(uint32_t b) const { return f[b & 0xFFu] & 1u; }
    __device__ __forceinline__ bool down(uint32_t b) const { return (f[b & 0xFFu] >> 1) & 1u; }
};

// The -U / -D context test and the table choice (pss-bam.c:134-142, :428-494).
// left1 / right1 = STORED genome bytes at s-1 and s+L (first context base on each side of
// the alignment, reference orientation).
template <class Ctx>
__device__ __forceinline__ void plan_finish_pss(const Ctx &ctx, Plan &pl, uint32_t left1, uint32_t right1) {
    // first context base each side, in read orientation (reverse reads: revcomp'ed window)
    const uint32_t up = pl.rev ? comp_stored(right1) : left1;
    const uint32_t dn = pl.rev ? comp_stored(left1) : right1;
    const bool up_ok = ctx.up(up), dn_ok = ctx.down(dn);
    const bool paired = (pl.flag & FL_PAIRED) != 0;
    const bool r1 = (pl.flag & FL_READ1) != 0, r2 = (pl.flag & FL_READ2) != 0;
    // unpaired: both tables, both tests (:428-447); paired: read1 && up -> fwd only, else
    // read2 && down -> rev only (:450-494)
    pl.pss_fwd = pl.pss_cand && (paired ? (r1 && up_ok) : (up_ok && dn_ok));
    pl.pss_rev = pl.pss_cand && (paired ? (!(r1 && up_ok) && r2 && dn_ok) : (up_ok && dn_ok));
}

// The same decision from the packed reference's nibbles (tiled kernel): an "other" nibble carries
// its own -U / -D membership, complementing leaves it unchanged (do_revcomp copies non-ACGT bytes).
__device__ __forceinline__ void plan_finish_pss_packed(uint32_t acgt_ctx, Plan &pl, uint32_t left1, uint32_t right1) {
    const uint32_t up = pl.rev ? (right1 < 4u ? 3u - right1 : right1) : left1;
    const uint32_t dn = pl.rev ? (left1 < 4u ? 3u - left1 : left1) : right1;
    const bool up_ok = ((up < 4u ? acgt_ctx >> (2u * up) : up) & 1u) != 0;
    const bool dn_ok = ((dn < 4u ? acgt_ctx >> (2u * dn + 1u) : dn >> 1) & 1u) != 0;
    const bool paired = (pl.flag & FL_PAIRED) != 0;
    const bool r1 = (pl.flag & FL_READ1) != 0, r2 = (pl.flag & FL_READ2) != 0;
    pl.pss_fwd = pl.pss_cand && (paired ? (r1 && up_ok) : (up_ok && dn_ok));
    pl.pss_rev = pl.pss_cand && (paired ? (!(r1 && up_ok) && r2 && dn_ok) : (up_ok && dn_ok));
}

// Event mask of one fully planned record (bits = stats slots that are NOT the OK outcome).
// kmer_result: 0 ok / not enabled, 1 = an attempted k-mer add hit a non-ACGT base.
__device__ __forceinline__ uint32_t record_events(bool DO_PSS, bool DO_KMER, const Plan &pl, bool kmer_failed) {
    uint32_t ev = 0u;
    if (pl.status == RS_RG_DROPPED) ev |= 1u << ST_RG_DROPPED;
    if (pl.status == RS_PARSE_SKIP) ev |= 1u << ST_PARSE_SKIP;
    if (pl.status == RS_NO_CONTIG) ev |= 1u << ST_NO_CONTIG;
    if (DO_PSS && pl.live && !(pl.pss_fwd || pl.pss_rev)) ev |= 1u << ST_PSS_FILTERED;
    if (DO_KMER && pl.live && !(pl.fk5 || pl.fk3)) ev |= 1u << ST_KMER_FILTERED;
    if (DO_KMER && pl.live && kmer_failed) ev |= 1u << ST_KMER_FAIL;
    return ev;
}

// Books a record's events into an LDS array of signed deltas (flushed as wrapping u64 adds).
__device__ __forceinline__ void book_events(bool DO_PSS, bool DO_KMER, uint32_t ev, int32_t *lds_delta) {
    if (ev == 0u) return;  // the common case: nothing to do
    if (ev & (1u << ST_RG_DROPPED)) atomicAdd(&lds_delta[ST_RG_DROPPED], 1);
    if (ev & (1u << ST_PARSE_SKIP)) atomicAdd(&lds_delta[ST_PARSE_SKIP], 1);
    if (ev & (1u << ST_NO_CONTIG)) atomicAdd(&lds_delta[ST_NO_CONTIG], 1);
    const bool dead = (ev & ((1u << ST_RG_DROPPED) | (1u << ST_PARSE_SKIP) | (1u << ST_NO_CONTIG))) != 0;
    if (DO_PSS) {
        if (ev & (1u << ST_PSS_FILTERED)) atomicAdd(&lds_delta[ST_PSS_FILTERED], 1);
        if (dead || (ev & (1u << ST_PSS_FILTERED))) atomicAdd(&lds_delta[ST_PSS_OK], -1);
    }
    if (DO_KMER) {
        if (ev & (1u << ST_KMER_FILTERED)) atomicAdd(&lds_delta[ST_KMER_FILTERED], 1);
        if (ev & (1u << ST_KMER_FAIL)) atomicAdd(&lds_delta[ST_KMER_FAIL], 1);
        if (dead || (ev & ((1u << ST_KMER_FILTERED) | (1u << ST_KMER_FAIL)))) atomicAdd(&lds_delta[ST_KMER_OK], -1);
    }
}

// Flush of the per-workgroup deltas; block 0 also credits the launch's record count.
__device__ __forceinline__ void flush_events(bool DO_PSS, bool DO_KMER, const TallyParams &P, const int32_t *lds_delta) {
    for (uint32_t i = threadIdx.x; i < (uint32_t)ST_USED; i += blockDim.x) {
        long long d = lds_delta[i];
        if (blockIdx.x == 0) {
            if (i == ST_RECORDS || (DO_PSS && i == ST_PSS_OK) || (DO_KMER && i == ST_KMER_OK)) d += P.n_recs;
        }
        if (d) atomicAdd(&P.counters[P.off_stats + i], (unsigned long long)d);
    }
}

// both steps with the two context bytes fetched from global memory (lane-per-read kernels)
template <bool DO_PSS, bool DO_KMER, class Src>
__device__ Plan make_plan(const TallyParams &P, const Src &src, const RecHdr &h) {
    Plan pl = plan_head<DO_PSS, DO_KMER>(P, src, h, RefsGlobal{P.ref_info});
    if (DO_PSS) {
        uint32_t l1 = 0, r1 = 0;
        if (pl.pss_cand) {
            const uint8_t *G = P.genome + pl.gbase;
            l1 = G[(int64_t)pl.s - 1];
            r1 = G[(int64_t)pl.s + pl.L];
        }
        plan_finish_pss(CtxMasks{P}, pl, l1, r1);
    }
    return pl;
}

// Bin of the k bases G[w0 .. w0+k) (forward strand) or of their reverse complement.
// Returns false when any base is not ACGT (add_to_ksp returns -1, kmer.c:55-111).
// Windows (closed form of fragkon.c:152-181):
//   fwd 5' = G[s-ok, k)        fwd 3' = G[s+L-ik, k)
//   rev 5' = rc(G[s+L-ok, k))  rev 3' = rc(G[s-ok+(ik-ok), k))
__device__ __forceinline__ bool kmer_bin(const uint8_t *G, int64_t w0, int K, bool rc, uint32_t &bin) {
    uint32_t b = 0;
    bool ok = true;
    for (int t = 0; t < K; t++) {
        const uint32_t c = ref_code(rc ? G[w0 + (K - 1 - t)] : G[w0 + t]);
        ok = ok && (c < 4u);
        b = (b << 2) | ((rc ? 3u - c : c) & 3u);
    }
    bin = b;
    return ok;
}

__device__ __forceinline__ void kmer_windows(const Plan &pl, int K, int64_t &w5, int64_t &w3) {
    const int64_t ok = K / 2, ik = K - K / 2, s = pl.s;
    if (!pl.rev) { w5 = s - ok; w3 = s + (int64_t)pl.Lk - ik; }
    else { w5 = s + (int64_t)pl.Lk - ok; w3 = s - ok + (ik - ok); }
}

}  // namespace pssbam
