// pss-bam_amd/csrc/inflate_kernels.h -- BGZF payload inflate (RFC 1951) + CRC-32 on gfx950.
//
// What it replaces: the `samtools view` child of the reference decompresses the BAM on the host
// (/root/reference/pss-bam.c:148-162); round 1 moved that to host threads (host/inflate_fast.c),
// which leaves PCIe carrying INFLATED bytes.  With these kernels the link carries
// the compressed file and the blocks are inflated where the tally kernels read them.
//
// Mapping: ONE LANE PER BGZF BLOCK.  A BAM is hundreds of thousands of independent <= 64 KiB
// deflate streams, and Huffman decoding is a serial chain per stream; rather than fight the
// chain inside a block, every lane owns a whole block and walks it with plain SIMT code --
// 64 blocks per wave, the wave diverging between "literal" and "match" like any branchy kernel.
//   * per-lane decode tables in LDS, lane-interleaved (entry i of lane l at [i][l]): canonical
//     Huffman data only -- sorted symbols + one offset per code length (452 B per lane, 28.3 KiB per
//     wave);
//   * a symbol's code length comes from 15 register thresholds (canonical codes are ordered:
//     the length is the number of left-aligned upper bounds the next 15 stream bits reach), so
//     decoding a symbol is ~40 VALU + two LDS reads, with no loop and no per-length divergence;
//   * the dynamic block header is parsed twice (count, then place) instead of storing 320 code
//     lengths per lane;
//   * 64-bit bit buffer fed 16 stream bytes per request; far copies move 16/32 bytes at a time, distances < 8
//     are expanded from a periodic register pattern (the QUAL runs of a BAM are distance-1 matches: no
//     load-after-store chain);
//   * two data loops (inflate_block<DEFER>, below): the engine feed runs the one that requests a step's
//     loads by LDS-DMA and waits once per step, behind the decode work;
//   * every access is bounded: reads inside the compressed buffer (+ one dword of slack the caller
//     provides), writes inside [out_off, out_off + isize); a malformed stream sets the block's
//     status and stops that lane.
// A second kernel checks ISIZE/CRC-32 per block (wave per block, 1 KiB chunks per lane, chunk
// CRCs combined by multiplication with x^(8*bytes behind the chunk) mod P).
//
// Bound: latency (LDS + L2 round trips on a serial chain), hidden by block-level parallelism:
// 4 waves x 64 lanes x 256 CUs = 65 536 streams in flight.  Integer/bit work; no MFMA.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pssbam {

struct BgzfBlock {       // one BGZF block, as the host's header walk found it
    uint64_t in_off;     // offset of the raw deflate payload in the compressed buffer
    uint32_t in_len;     // payload bytes (block length - header - 8-byte trailer)
    uint32_t isize;      // ISIZE from the trailer: bytes this block inflates to
    uint64_t out_off;    // where they go in the output buffer
    uint32_t crc;        // CRC-32 from the trailer
    uint32_t status;     // out: 0 ok, else INF_*
};

enum : uint32_t { INF_OK = 0, INF_BAD_BLOCK = 1, INF_BAD_CODES = 2, INF_BAD_SYMBOL = 3, INF_BAD_DISTANCE = 4, INF_OVERRUN = 5,
                  INF_SHORT = 6, INF_TRUNCATED = 7, INF_BAD_CRC = 8 };

constexpr int INF_WAVE = 64;
// per-lane LDS, lane-interleaved, in three arrays of different element width:
//   u16[48]  : L_DELTA[16] litlen index base minus first code per code length; L_OFFS[16] scratch while a
//              table is built (next free slot per length); D_DELTA[16] the same for the distance code
//   u8[320]  : L_SYM[288] low 8 bits of the litlen symbols sorted by (code length, symbol);
//              D_SYM[32] distance symbols (also hosts the 19-symbol code-length code)
//   u32[9]   : bit 8 of the 288 litlen symbols
// = 452 B per lane, 28.3 KiB per wave: room for five waves per CU (the 16-bit symbol table of the first version
// allowed three).
constexpr int L_DELTA = 0, L_OFFS = 16, D_DELTA = 32, INF_N16 = 48;
constexpr int L_SYM = 0, D_SYM = 288, INF_N8 = 320;
constexpr int INF_N32 = 9;
constexpr uint32_t INF_LDS_BYTES = (INF_N16 * 2 + INF_N8 + INF_N32 * 4) * INF_WAVE;
constexpr int INF_WAVES_PER_CU = 4;   // one per SIMD: 4 do what 5 do, or better (2/3/4/5/6: 187/226/291/287 in place; 225/273/263/256 one-wait), and
                                      // a lone wave on its SIMD may use the whole register file
// Two data loops (inflate_block<DEFER>):
//   DEFER = false  "in place": a copy is loaded, waited for and stored where it is decoded; literals leave as byte
//                  stores, at most INF_RUN_INPLACE per step.
//   DEFER = true   "one wait per step": every step starts by REQUESTING what the next memory phase needs -- the source
//                  of the copy decoded in the previous step and the next 16 stream bytes -- then decodes (no memory
//                  operations), then waits ONCE, stores the copy and its own literals.
// Why the second one, and why its requests are LDS-DMA: in the compiled in-place loop every global_load is followed
// by s_waitcnt vmcnt(0) within a few instructions (the loaded registers are copied into loop-carried ones at once),
// so each load costs a full round trip AND drains the stores in front of it -- "prefetched" stream reads included
// (SQ_WAIT_ANY 68 % of the wave cycles, profiles/r02_inflate_variants.txt).  Writing the same schedule with ordinary
// loads does not help: the register allocator touches the loaded registers early and the wait moves there (measured:
// slower than in place).  global_load_lds_dwordx4 from inline asm lands the data in LDS without the compiler knowing
// a load is in flight; the one s_waitcnt vmcnt(0) of a step is written by hand, behind the decode phase.
// literals a lane may take per step.  DEFER: they travel in one register (<= 8), and the budget is a launch
// parameter -- 4 suits match-dominated streams, 6 literal-dominated ones (profiles/r02_inflate_variants.txt)
constexpr uint32_t INF_RUN_INPLACE = 4u, INF_RUN_DEFER_MAX = 8u;
// landing planes of the DEFER loop behind the decode tables: 16 bytes per lane each -- four for the source of a copy
// piece (two without PIECES), one for the stream
constexpr uint32_t INF_LAND_BYTES = 5u * 16u * INF_WAVE;
// PIECES (the engine feed's default; PSSBAM_INFLATE_PIECES=0 for A/B): BOUNDED WORK PER LANE AND STEP.  Stamps around the
// phases of a step (profiles/r03_inflate_step_stamps.txt) showed where a step's 11-15 thousand cycles went: 4.5-6.3 k in
// the decode phase, ~30 in the wait (what was requested HAD arrived) and 4.7-10 k behind it -- because a copy that was
// long (> 32 bytes: a BAM's SEQ and QUAL matches) or a run (QUAL: 150 bytes of period 1) was carried out IN PLACE, with
// its own load round trips and store loops, by the lanes that had one while the other 60 waited: with 64 lanes SOME lane
// has one at nearly every step, so every lane paid for it at every step.  Now a lane moves at most INF_PIECE bytes of a
// copy per step, through the landing planes like a short copy: a long copy or run spans several steps, during which ITS
// lane decodes nothing (its stream position and its literals wait) and the others go on.  Steps get shorter for all 64
// lanes; a lane with a 150-byte copy spends three of them on it.
constexpr uint32_t INF_PIECE = 64u;
constexpr uint32_t INF_LDS_BYTES_DEFER = INF_LDS_BYTES + INF_LAND_BYTES;
static_assert(INF_LDS_BYTES_DEFER * INF_WAVES_PER_CU <= 160u * 1024u, "the waves of the DEFER loop must fit the CU's LDS");

struct LaneLds {  // this lane's view of the three interleaved arrays
    // Interleaving is by DWORD: entry i of lane l sits in dword (i / per_dword) * 64 + l, so whatever
    // entries the 64 lanes ask for, lane l always reads bank l % 32 -- no bank conflicts beyond the
    // two-lanes-per-bank of a 64-wide wave (element-wise interleaving had the lanes of a wave collide
    // 4-ways on the byte table: 45 % of the LDS cycles were conflict cycles).
    uint8_t *b16;   // base of the u16 array + 4 * lane
    uint8_t *b8;    // base of the u8 array + 4 * lane
    uint32_t *b32;  // base of the u32 array + lane
    __device__ __forceinline__ uint16_t &at(int i) const { return *(uint16_t *)(b16 + (i >> 1) * (INF_WAVE * 4) + (i & 1) * 2); }
    __device__ __forceinline__ uint8_t &sym8(int i) const { return b8[(i >> 2) * (INF_WAVE * 4) + (i & 3)]; }
    __device__ __forceinline__ void clear_hi() const {
#pragma unroll
        for (int k = 0; k < INF_N32; k++) b32[k * INF_WAVE] = 0u;
    }
    // litlen symbol (9 bits) at sorted position i
    __device__ __forceinline__ void put_litlen(uint32_t i, uint32_t sym) const {
        sym8(L_SYM + (int)i) = (uint8_t)sym;
        if (sym >> 8) b32[(i >> 5) * INF_WAVE] |= 1u << (i & 31u);
    }
    __device__ __forceinline__ uint32_t get_litlen(uint32_t i) const {
        const uint32_t lo = sym8(L_SYM + (int)i), hi = b32[(i >> 5) * INF_WAVE];
        return lo | (((hi >> (i & 31u)) & 1u) << 8);
    }
};

struct BitReader {
    const uint32_t *p;     // next dword group to fetch
    const uint32_t *end;   // first dword that must not be read
    uint64_t buf;
    uint32_t cnt;          // valid bits in buf
    uint32_t rc;           // dwords left in the reservoir (r0 = the next two, r1 = the two behind them)
    uint64_t r0, r1;
    uint64_t n0, n1;       // the 16 bytes behind the reservoir, arrived (n_valid) ...
    uint32_t n_valid;
    uint4 ahead;           // ... and the 16 bytes behind those, in flight
    uint64_t consumed;     // bits handed out
    uint64_t limit;        // payload bits

    // The stream is read 16 bytes per request (a dword per request had every 128-byte line looked up --
    // and, with 80 000 streams thrashing L2, often fetched -- four times as often) through a queue of three
    // groups: the reservoir being consumed, one group that has arrived (n), one requested (ahead = the 16
    // bytes at p).  Beyond the end of the buffer the stream reads as zeros (overrun() tells).
    __device__ __forceinline__ void request() {
        if (p + 4 <= end) __builtin_memcpy(&ahead, p, 16);
        else {
            ahead.x = p < end ? p[0] : 0u;
            ahead.y = p + 1 < end ? p[1] : 0u;
            ahead.z = p + 2 < end ? p[2] : 0u;
            ahead.w = p + 3 < end ? p[3] : 0u;
        }
    }
    __device__ __forceinline__ void init() { buf = 0; cnt = 0; rc = 0; r0 = r1 = n0 = n1 = 0; n_valid = 0; consumed = 0; }
    __device__ __forceinline__ void take_ahead() {   // (the first use of `ahead` since request(): a wait)
        if (!n_valid) {
            n0 = (uint64_t)ahead.x | ((uint64_t)ahead.y << 32);
            n1 = (uint64_t)ahead.z | ((uint64_t)ahead.w << 32);
            n_valid = 1u;
            p += 4;
        }
    }
    __device__ __forceinline__ void top_up() {
        if (!n_valid) {
            take_ahead();
            request();
        }
    }
    // bits that can be handed out without touching memory
    __device__ __forceinline__ uint32_t avail() const { return cnt + 32u * rc + (n_valid ? 128u : 0u); }
    __device__ __forceinline__ void refill_nomem() {   // caller: avail() covers what it is about to take
        if (cnt <= 32u && (rc | n_valid)) {   // (nothing queued: the bits in buf are all there is until the next memory phase)
            if (rc == 0u) { r0 = n0; r1 = n1; rc = 4u; n_valid = 0u; }
            buf |= (r0 & 0xFFFFFFFFull) << cnt;
            cnt += 32u;
            r0 = (r0 >> 32) | (r1 << 32);
            r1 >>= 32;
            rc--;
        }
    }
    __device__ __forceinline__ void refill() {
        if (cnt <= 32u) {
            if (rc == 0u) top_up();   // (n_valid afterwards)
            refill_nomem();
        }
    }
    // DEFER loop: the address its stream request reads (never across the end of the buffer: the buffer's last 16
    // bytes instead) and the hand-over of what arrived
    __device__ __forceinline__ const uint32_t *request_addr() const { return p + 4 <= end ? p : end - 4; }
    __device__ __forceinline__ void take_group(uint4 g) {
        if (!n_valid) {
            n0 = (uint64_t)g.x | ((uint64_t)g.y << 32);
            n1 = (uint64_t)g.z | ((uint64_t)g.w << 32);
            if (p + 4 > end) {
                // the group straddles the end: g = [end - 4, end), whose LAST dwords are this group's first ones;
                // what lies beyond the buffer reads as zero
                const uint64_t k = (uint64_t)((p + 4) - end);   // dwords of the group that do not exist
                if (k >= 4u) n0 = n1 = 0;
                else if (k == 3u) { n0 = n1 >> 32; n1 = 0; }
                else if (k == 2u) { n0 = n1; n1 = 0; }
                else { n0 = (n0 >> 32) | (n1 << 32); n1 >>= 32; }
            }
            n_valid = 1u;
            p += 4;
        }
    }
    __device__ __forceinline__ uint32_t peek15() const { return __brev((uint32_t)buf) >> 17; }  // first-read bit = MSB
    __device__ __forceinline__ void drop(uint32_t n) { buf >>= n; cnt -= n; consumed += n; }
    __device__ __forceinline__ uint32_t take(uint32_t n) {  // n <= 16, caller has refilled
        const uint32_t v = (uint32_t)buf & ((1u << n) - 1u);
        drop(n);
        return v;
    }
    __device__ __forceinline__ bool overrun() const { return consumed > limit; }
};

// Canonical Huffman: builds upper[] (registers) and delta[] (LDS) from the per-length counts in
// LDS at cnt_at (which it turns into the next-free-slot table offs[]).  false = over-subscribed.
template <int MAXL>
__device__ __forceinline__ bool huff_finish(const LaneLds &t, int cnt_at, int delta_at, uint32_t (&upper)[15]) {
    uint32_t first = 0u, index = 0u;
    bool ok = true;
#pragma unroll
    for (int L = 1; L <= 15; L++) {
        if (L <= MAXL) {
            const uint32_t c = t.at(cnt_at + L);
            const uint32_t up = first + c;
            ok = ok && up <= (1u << L);
            upper[L - 1] = up << (15 - L);
            t.at(delta_at + L) = (uint16_t)(index - first);   // mod 2^16: added back to a 16-bit code value
            t.at(cnt_at + L) = (uint16_t)index;               // offs[L]
            index += c;
            first = up << 1;
        } else {
            upper[L - 1] = upper[MAXL - 1];
        }
    }
    return ok;
}

// The fifteen left-aligned upper bounds of a 15-length code as eight pairs of u16 (slot 15 = 0x8000: never reached), and
// the length of the code the next 15 stream bits begin with: 1 + the number of bounds they reach.  Compare-and-add through
// the scalar registers costs v_cmp + two wait states + v_addc per bound (hipcc: "s_nop 1" behind every v_cmp whose mask the
// next VALU instruction reads -- 150 cycles of a one-wave-per-SIMD kernel per symbol); here: eight packed 16-bit
// subtractions (the sign of a half = "below this bound"; 0x8000 - c wraps to "below" as it must), the sixteen sign bytes
// gathered four to a register, four popcounts -- 22 instructions, no scalar round trip.
struct Upper2 { uint32_t p[8]; };
__device__ __forceinline__ void pack_upper(const uint32_t (&upper)[15], Upper2 &u) {
#pragma unroll
    for (int j = 0; j < 8; j++) u.p[j] = upper[2 * j] | ((2 * j + 1 < 15 ? upper[2 * j + 1 < 15 ? 2 * j + 1 : 0] : 0x8000u) << 16);
}
__device__ __forceinline__ uint32_t code_len15(uint32_t c15, const Upper2 &u) {
    typedef short short2v __attribute__((ext_vector_type(2)));
    const uint32_t cc = c15 | (c15 << 16);
    uint32_t d[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const short2v x = __builtin_bit_cast(short2v, cc) - __builtin_bit_cast(short2v, u.p[j]);
        d[j] = __builtin_bit_cast(uint32_t, x);
    }
    uint32_t n = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) n += (uint32_t)__builtin_popcount(__builtin_amdgcn_perm(d[2 * j + 1], d[2 * j], 0x07050301u) & 0x80808080u);
    return 17u - n;
}

// symbol for the next code of the stream; < 0: the bits are no code of this table.
// WIDE: the literal/length table (9-bit symbols); else a u8 table at sym_at (distance / code-length code)
template <int MAXL, bool WIDE>
__device__ __forceinline__ int huff_decode(BitReader &br, const LaneLds &t, int delta_at, int sym_at, int n_sym,
                                           const uint32_t (&upper)[15]) {
    const uint32_t c15 = br.peek15();
    uint32_t L = 1u;
#pragma unroll
    for (int k = 0; k < MAXL; k++) L += c15 >= upper[k] ? 1u : 0u;
    if (L > (uint32_t)MAXL) return -1;
    const uint32_t idx = ((c15 >> (15u - L)) + t.at(delta_at + (int)L)) & 0xFFFFu;
    if (idx >= (uint32_t)n_sym) return -1;
    br.drop(L);
    return WIDE ? (int)t.get_litlen(idx) : (int)t.sym8(sym_at + (int)idx);
}

template <bool WIDE>
__device__ __forceinline__ int huff_decode15(BitReader &br, const LaneLds &t, int delta_at, int sym_at, int n_sym, const Upper2 &upper) {
    const uint32_t c15 = br.peek15();
    const uint32_t L = code_len15(c15, upper);
    if (L > 15u) return -1;
    const uint32_t idx = ((c15 >> (15u - L)) + t.at(delta_at + (int)L)) & 0xFFFFu;
    if (idx >= (uint32_t)n_sym) return -1;
    br.drop(L);
    return WIDE ? (int)t.get_litlen(idx) : (int)t.sym8(sym_at + (int)idx);
}

// The next TWO literal/length symbols, nothing dropped: b is decoded from the bits behind a's code before a's table
// reads have come back, so the two LDS round trips of a symbol (offset of its code length, then the symbol) are
// shared by the pair -- literal runs are what a BAM's quality strings inflate from.  sym < 0: no code of the table.
struct LitPair { int a, b; uint32_t la, lb; };
__device__ __forceinline__ LitPair litlen_peek2(const BitReader &br, const LaneLds &t, const Upper2 &upper) {
    const uint32_t ca = br.peek15();
    const uint32_t la = code_len15(ca, upper);
    const uint32_t la_c = min(la, 15u);
    const uint32_t cb = __brev((uint32_t)(br.buf >> la_c)) >> 17;   // (caller: at least 30 bits in buf)
    const uint32_t lb = code_len15(cb, upper);
    const uint32_t lb_c = min(lb, 15u);
    const uint32_t da = t.at(L_DELTA + (int)la_c), db = t.at(L_DELTA + (int)lb_c);
    const uint32_t ia = ((ca >> (15u - la_c)) + da) & 0xFFFFu, ib = ((cb >> (15u - lb_c)) + db) & 0xFFFFu;
    const bool va = la <= 15u && ia < 288u, vb = lb <= 15u && ib < 288u;
    uint32_t sa = t.get_litlen(va ? ia : 0u), sb = t.get_litlen(vb ? ib : 0u);
    // (both symbols are wanted HERE: left alone the compiler sinks b's table reads into the caller's "a is a literal"
    //  branch, and the pair pays four LDS round trips instead of two)
    asm volatile("" : "+v"(sa), "+v"(sb));
    return {va ? (int)sa : -1, vb ? (int)sb : -1, la_c, lb_c};
}

__device__ __forceinline__ uint64_t load_u64(const uint8_t *p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ void store_u64(uint8_t *p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
__device__ __forceinline__ uint4 load_u128(const uint8_t *p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
__device__ __forceinline__ void store_u128(uint8_t *p, uint4 v) { __builtin_memcpy(p, &v, 16); }
// The low n (<= 8) bytes of v in at most TWO store requests: a head piece and a tail piece of the same
// power-of-two width that overlap in the middle (the memory system here is paid per request).
__device__ __forceinline__ void store_tail(uint8_t *p, uint64_t v, uint32_t n) {
    if (n >= 8u) store_u64(p, v);
    else if (n >= 4u) {
        const uint32_t a = (uint32_t)v, b = (uint32_t)(v >> (8u * (n - 4u)));
        __builtin_memcpy(p, &a, 4);
        __builtin_memcpy(p + n - 4u, &b, 4);
    } else if (n >= 2u) {
        const uint16_t a = (uint16_t)v, b = (uint16_t)(v >> (8u * (n - 2u)));
        __builtin_memcpy(p, &a, 2);
        __builtin_memcpy(p + n - 2u, &b, 2);
    } else if (n) *p = (uint8_t)v;
}
// the low n (<= 16) bytes of (lo, hi), likewise
__device__ __forceinline__ void store_tail16(uint8_t *p, uint64_t lo, uint64_t hi, uint32_t n) {
    if (n >= 16u) { uint4 q; q.x = (uint32_t)lo; q.y = (uint32_t)(lo >> 32); q.z = (uint32_t)hi; q.w = (uint32_t)(hi >> 32); store_u128(p, q); }
    else if (n > 8u) {
        const uint32_t sh = 8u * (n - 8u);            // 8..56
        store_u64(p, lo);
        store_u64(p + n - 8u, (lo >> sh) | (hi << (64u - sh)));
    } else store_tail(p, lo, n);
}
// len bytes of the periodic sequence whose first `dist` (< 8) bytes are the low bytes of pat
__device__ __forceinline__ void store_run(uint8_t *dst, uint64_t pat, uint32_t dist, uint32_t len) {
    pat &= (1ull << (8u * dist)) - 1ull;
    for (uint32_t w = dist; w < 8u; w <<= 1) pat |= pat << (8u * w);
    if ((8u % dist) == 0u) {
        // period 1, 2 or 4 (the QUAL runs): the pattern repeats every 8 bytes, 16 per request
        uint4 q;
        q.x = q.z = (uint32_t)pat;
        q.y = q.w = (uint32_t)(pat >> 32);
        while (len >= 16u) { store_u128(dst, q); dst += 16; len -= 16u; }
        store_tail16(dst, pat, pat, len);
    } else {
        const uint32_t step = (8u / dist) * dist;   // whole periods per 8-byte store
        while (len >= 8u) { store_u64(dst, pat); dst += step; len -= step; }
        store_tail(dst, pat, len);
    }
}

// 16 bytes per lane, global -> LDS at lds_base + 16 * lane, without a register in between and without the compiler
// knowing (m0 is compiler-reserved and cannot be named as a clobber: saved and restored)
__device__ __forceinline__ void dma16_to_lds(uint32_t lds_base, const void *src) {
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_base);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(m0v) : "memory");
}
// the low n (1..64) bytes of four 16-byte planes, in at most five store requests
__device__ __forceinline__ void store_piece(uint8_t *p, uint4 a, uint4 b, uint4 c, uint4 d, uint32_t n) {
    uint4 last = a;
    if (n > 16u) { store_u128(p, a); p += 16; n -= 16u; last = b; }
    if (n > 16u) { store_u128(p, b); p += 16; n -= 16u; last = c; }
    if (n > 16u) { store_u128(p, c); p += 16; n -= 16u; last = d; }
    store_tail16(p, (uint64_t)last.x | ((uint64_t)last.y << 32), (uint64_t)last.z | ((uint64_t)last.w << 32), n);
}

// One BGZF block by one lane.  Returns INF_*.  land: this wave's landing planes, lit_run: literals per step (DEFER only).
// STAMP: a diagnostic build of the DEFER loop (tools/inflate_stamps.py): shader-clock stamps around the phases of every step,
// summed per lane into st[] -- [0] decode phase (requests .. wait), [1] the wait, [2] stores + classification + in-place
// copies, [3] steps, [4] lanes of the wave that were in the step, [5] literals, [6] steps that ended in an in-place copy.
// The stamp values go to a buffer of their own and feed nothing else.
struct StepStamps { uint64_t v[8]; };
template <bool DEFER, bool PIECES, bool STAMP = false>
__device__ uint32_t inflate_block(const uint8_t *comp, uint64_t comp_bytes, const BgzfBlock &b, uint8_t *outbuf, const LaneLds &t,
                                  uint8_t *land, uint32_t lane, uint32_t lit_run, StepStamps *st = nullptr) {
    uint8_t *out = outbuf + b.out_off;
    const uint32_t isize = b.isize;
    BitReader br;
    {
        const uint64_t a = b.in_off;
        br.p = (const uint32_t *)(comp + (a & ~3ull));
        br.end = (const uint32_t *)(comp + ((comp_bytes + 3ull) & ~3ull));
        br.init();
        br.limit = (uint64_t)b.in_len * 8ull;
        br.request();
        br.refill();
        const uint32_t skip = 8u * (uint32_t)(a & 3ull);
        br.buf >>= skip;
        br.cnt -= skip;
    }
    uint32_t pos = 0;
    uint32_t lu[15], du[15];   // code-length thresholds of the literal/length and distance codes
    Upper2 lu2, du2;           // ... packed, for the one-wait loop (code_len15)
    for (;;) {
        br.refill();
        const uint32_t bfinal = br.take(1), btype = br.take(2);
        if (btype == 0u) {
            // stored: skip to the byte boundary, LEN / NLEN, LEN raw bytes
            br.drop(br.cnt & 7u);
            br.refill();
            const uint32_t len = br.take(16);
            br.refill();
            const uint32_t nlen = br.take(16);
            if ((len ^ nlen) != 0xFFFFu) return INF_BAD_BLOCK;
            if (len > isize - pos) return INF_OVERRUN;
            if (br.consumed + (uint64_t)len * 8ull > br.limit) return INF_TRUNCATED;
            for (uint32_t i = 0; i < len; i++) {   // byte-aligned from here: a byte at a time through the bit buffer
                br.refill();
                out[pos++] = (uint8_t)br.take(8);
            }
        } else if (btype == 1u || btype == 2u) {
            if (btype == 1u) {
                // fixed code (RFC 1951 3.2.6): lengths 8 x144, 9 x112, 7 x24, 8 x8; 30 distance codes of 5 bits
#pragma unroll
                for (int L = 0; L < 16; L++) { t.at(L_OFFS + L) = 0; t.at(D_DELTA + L) = 0; }
                t.at(L_OFFS + 7) = 24; t.at(L_OFFS + 8) = 152; t.at(L_OFFS + 9) = 112;
                (void)huff_finish<15>(t, L_OFFS, L_DELTA, lu);
                pack_upper(lu, lu2);
                t.clear_hi();
                for (int s = 0; s < 288; s++) {
                    const int L = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
                    const uint32_t o = t.at(L_OFFS + L);
                    t.put_litlen(o, (uint32_t)s);
                    t.at(L_OFFS + L) = (uint16_t)(o + 1u);
                }
#pragma unroll
                for (int L = 0; L < 16; L++) t.at(L_OFFS + L) = 0;
                t.at(L_OFFS + 5) = 30;
                (void)huff_finish<15>(t, L_OFFS, D_DELTA, du);
                pack_upper(du, du2);
                for (int s = 0; s < 30; s++) t.sym8(D_SYM + s) = (uint8_t)s;
            } else {
                // dynamic code: HLIT, HDIST, HCLEN, the code-length code, then the two codes' lengths
                br.refill();
                const uint32_t nlen = br.take(5) + 257u, ndist = br.take(5) + 1u, ncode = br.take(4) + 4u;
                if (nlen > 286u || ndist > 30u) return INF_BAD_CODES;
                uint32_t cu[15];
                uint64_t cl_packed = 0;   // the code-length code's own lengths, 3 bits each, by symbol
                {
#pragma unroll
                    for (int L = 0; L < 16; L++) t.at(L_OFFS + L) = 0;
                    for (uint32_t i = 0; i < ncode; i++) {
                        br.refill();
                        const uint32_t l = br.take(3);
                        // position of code-length symbol: 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15
                        const uint32_t sym = (uint32_t)((0xF1E2D3C4B5A69780ull >> (4u * (i >= 3u ? i - 3u : 0u))) & 15ull);
                        const uint32_t s = i < 3u ? 16u + i : sym;
                        cl_packed |= (uint64_t)l << (3u * s);
                        if (l) t.at(L_OFFS + (int)l) = (uint16_t)(t.at(L_OFFS + (int)l) + 1u);
                    }
                    if (!huff_finish<7>(t, L_OFFS, D_DELTA, cu)) return INF_BAD_CODES;
                    for (uint32_t s = 0; s < 19u; s++) {
                        const uint32_t l = (uint32_t)(cl_packed >> (3u * s)) & 7u;
                        if (l) {
                            const uint32_t o = t.at(L_OFFS + (int)l);
                            t.sym8(D_SYM + (int)o) = (uint8_t)s;
                            t.at(L_OFFS + (int)l) = (uint16_t)(o + 1u);
                        }
                    }
                }
                // the nlen + ndist code lengths are walked twice from the same stream position: pass 0
                // counts them per length, pass 1 places the symbols (no per-lane array of 320 lengths)
                const BitReader mark = br;
                uint16_t lcount[16], dcount[16];   // pass 0 results (registers: static indices only below)
                uint32_t n_cl_sym = 0;   // symbols the code-length code actually has
                for (uint32_t s = 0; s < 19u; s++) n_cl_sym += ((cl_packed >> (3u * s)) & 7u) ? 1u : 0u;
                for (int pass = 0; pass < 2; pass++) {
                    if (pass == 1) {
                        // counts -> tables.  The code-length code's D_DELTA/D_SYM entries are still needed for
                        // pass 1, so the distance table is finished after the walk; litlen can be finished now
                        br = mark;
#pragma unroll
                        for (int L = 0; L < 16; L++) t.at(L_OFFS + L) = lcount[L];
                        if (!huff_finish<15>(t, L_OFFS, L_DELTA, lu)) return INF_BAD_CODES;
                        pack_upper(lu, lu2);
                        t.clear_hi();
                    } else {
#pragma unroll
                        for (int L = 0; L < 16; L++) { lcount[L] = 0; dcount[L] = 0; }
                    }
                    uint32_t idx = 0, prev = 0;
                    // distance symbols cannot be placed while D_SYM holds the code-length code: their
                    // lengths are remembered as 30 nibbles (two registers)
                    uint64_t dl_lo = 0, dl_hi = 0;
                    while (idx < nlen + ndist) {
                        br.refill();
                        const int sym = huff_decode<7, false>(br, t, D_DELTA, D_SYM, (int)n_cl_sym, cu);
                        if (sym < 0) return INF_BAD_CODES;
                        uint32_t len = (uint32_t)sym, rep = 1u;
                        if (sym >= 16) {
                            if (sym == 16) { if (idx == 0u) return INF_BAD_CODES; len = prev; rep = 3u + br.take(2); }
                            else if (sym == 17) { len = 0u; rep = 3u + br.take(3); }
                            else { len = 0u; rep = 11u + br.take(7); }
                        }
                        if (idx + rep > nlen + ndist) return INF_BAD_CODES;
                        prev = len;
                        for (uint32_t r = 0; r < rep; r++, idx++) {
                            if (!len) continue;
                            if (idx < nlen) {
                                if (pass == 0) {
#pragma unroll
                                    for (int L = 1; L < 16; L++) lcount[L] = (uint16_t)(lcount[L] + ((uint32_t)L == len ? 1u : 0u));
                                } else {
                                    const uint32_t o = t.at(L_OFFS + (int)len);
                                    t.put_litlen(o, idx);
                                    t.at(L_OFFS + (int)len) = (uint16_t)(o + 1u);
                                }
                            } else {
                                const uint32_t d = idx - nlen;
                                if (pass == 0) {
#pragma unroll
                                    for (int L = 1; L < 16; L++) dcount[L] = (uint16_t)(dcount[L] + ((uint32_t)L == len ? 1u : 0u));
                                } else {
                                    if (d < 16u) dl_lo |= (uint64_t)len << (4u * d);
                                    else dl_hi |= (uint64_t)len << (4u * (d - 16u));
                                }
                            }
                        }
                    }
                    if (br.overrun()) return INF_TRUNCATED;
                    if (pass == 1) {
                        // now the distance code: counts -> thresholds, then place its symbols
#pragma unroll
                        for (int L = 0; L < 16; L++) t.at(L_OFFS + L) = dcount[L];
                        if (!huff_finish<15>(t, L_OFFS, D_DELTA, du)) return INF_BAD_CODES;
                        pack_upper(du, du2);
                        for (uint32_t d = 0; d < ndist; d++) {
                            const uint32_t l = (uint32_t)((d < 16u ? dl_lo >> (4u * d) : dl_hi >> (4u * (d - 16u))) & 15ull);
                            if (l) {
                                const uint32_t o = t.at(L_OFFS + (int)l);
                                t.sym8(D_SYM + (int)o) = (uint8_t)d;
                                t.at(L_OFFS + (int)l) = (uint16_t)(o + 1u);
                            }
                        }
                    }
                }
            }
            // ---- the compressed data of this deflate block ------------------------------------
            if constexpr (DEFER) {
            // (0) requests, (1) decode, (2) wait + stores; see the head of the file.  This step's copy is only
            // classified in (2): short and not feeding on itself -> its source is what the next step requests,
            // anything else is copied in place.  Program order keeps the bytes right: everything in front of a
            // copy's source has been stored (issued) before the request is; at the end of a deflate block
            // nothing is pending (its last step has no copy).
            const uint32_t land0 = (uint32_t)(uintptr_t)land;   // LDS byte address of the planes
            constexpr uint32_t STREAM_PLANE = 64u * INF_WAVE;    // (planes 0-3: the copy piece)
            const uint8_t *lsrc = comp;     // what (0) reads: the pending piece's source, else the head of the buffer
            uint8_t *pdst = out;
            uint32_t plen = 0, pdist = 0;   // pending piece; pdist != 0: a run of period pdist (< 8) seeded by the 8 bytes in front of pdst
            uint32_t prem = 0, pcap = 0;    // PIECES: bytes of the copy behind the pending piece, and the most one piece may take
            bool pself = false;             // PIECES: the copy's distance is below a piece's size (it may feed on itself)
            for (;;) {
                uint64_t ts0 = 0, ts1 = 0, ts2 = 0;
                if constexpr (STAMP) ts0 = __builtin_amdgcn_s_memtime();
                dma16_to_lds(land0, lsrc);
                dma16_to_lds(land0 + 16u * INF_WAVE, lsrc + 16);
                if constexpr (PIECES) {
                    dma16_to_lds(land0 + 32u * INF_WAVE, lsrc + 32);
                    dma16_to_lds(land0 + 48u * INF_WAVE, lsrc + 48);
                }
                dma16_to_lds(land0 + STREAM_PLANE, br.request_addr());
                const bool busy = PIECES && prem != 0u;   // this lane is in the middle of a copy: it decodes nothing this step
                int sym = 512;   // 512: no token this step
                uint32_t run = 0;
                uint64_t lits = 0;
                while (!busy && run < lit_run && br.avail() >= 64u) {   // (two literals and a token take at most 30 + 33 bits)
                    br.refill_nomem();
                    const LitPair lp = litlen_peek2(br, t, lu2);
                    if (lp.a < 0) { sym = -1; break; }
                    br.drop(lp.la);
                    if (lp.a >= 256) { sym = lp.a; break; }
                    lits |= (uint64_t)(uint32_t)lp.a << (8u * run);
                    run++;
                    if (lp.b >= 256) { br.drop(lp.lb); sym = lp.b; break; }   // the token behind the literal: decoded already
                    if (lp.b >= 0 && run < INF_RUN_DEFER_MAX) {
                        br.drop(lp.lb);
                        lits |= (uint64_t)(uint32_t)lp.b << (8u * run);
                        run++;
                    }   // (lp.b < 0: the next round finds out)
                }
                if (sym < 0) return INF_BAD_SYMBOL;
                uint32_t len = 0, dist = 0;
                if (sym > 256 && sym != 512) {
                    if (sym > 285) return INF_BAD_SYMBOL;
                    br.refill_nomem();
                    // length: 257..264 -> 3..10; 265..284 -> ((4 + (s-265)%4) << e) + 3 with e = (s-261)/4 extra bits; 285 -> 258.
                    // Selects, not branches (a branch region costs a one-wave-per-SIMD kernel half a dozen scalar issue slots,
                    // and some lane takes each side at every step anyway); take(0) takes nothing.
                    {
                        const uint32_t s = (uint32_t)sym;
                        const bool mid = s >= 265u && s < 285u;
                        const uint32_t e = mid ? (s - 261u) >> 2 : 0u;
                        const uint32_t base = mid ? ((4u + ((s - 265u) & 3u)) << e) + 3u : s == 285u ? 258u : s - 254u;
                        len = base + br.take(e);
                    }
                    // (refill_nomem left >= 32 bits, >= 33 unless the buffer was EMPTY; the length took <= 5 of them, the distance
                    //  code takes <= 15 and its extra bits <= 13: a second refill only in that one case)
                    if (br.cnt < 28u) br.refill_nomem();
                    const int ds = huff_decode15<false>(br, t, D_DELTA, D_SYM, 30, du2);
                    if (ds < 0 || ds >= 30) return INF_BAD_DISTANCE;   // (>= 30: an unplaced slot of an incomplete code)
                    {
                        const uint32_t d = (uint32_t)ds;
                        const uint32_t e = d < 4u ? 0u : (d >> 1) - 1u;
                        dist = (d < 4u ? d + 1u : ((2u + (d & 1u)) << e) + 1u) + br.take(e);
                    }
                }
                // ---- (2): what (0) requested has had the decode phase to arrive
                if constexpr (STAMP) ts1 = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if constexpr (STAMP) {
                    ts2 = __builtin_amdgcn_s_memtime();
                    st->v[0] += ts1 - ts0;
                    st->v[1] += ts2 - ts1;
                    st->v[3] += 1;
                    st->v[4] += (uint64_t)__popcll(__ballot(1));
                    st->v[5] += run;
                }
                if (plen) {
                    uint4 pa = *(const uint4 *)(land + 16u * lane);
                    if constexpr (PIECES) {
                        // ONE store path for far pieces and for runs of period 1, 2, 4 (the QUAL runs: their 16-byte pattern stands
                        // in for all four planes) -- with 64 lanes SOME lane has each kind at nearly every step, and a wave pays
                        // for every path any of its lanes takes.  Whole 16-byte groups: the <= 15 bytes behind a piece's end are
                        // positions this lane fills LATER in program order (this step's literals, the next piece, the next token),
                        // and one lane's stores to one address keep their order; only at the block's end the exact form is used.
                        uint4 pb = *(const uint4 *)(land + 16u * INF_WAVE + 16u * lane), pc = *(const uint4 *)(land + 32u * INF_WAVE + 16u * lane),
                              pd = *(const uint4 *)(land + 48u * INF_WAVE + 16u * lane);
                        bool common = true;
                        const bool loose = (uint32_t)(pdst - out) + ((plen + 15u) & ~15u) <= isize;
                        if (pdist) {
                            uint64_t pat = (((uint64_t)pa.z | ((uint64_t)pa.w << 32)) >> (8u * (8u - pdist))) & ((1ull << (8u * pdist)) - 1ull);
                            if ((pdist & (pdist - 1u)) == 0u) {   // period 1, 2 or 4 (pdist < 8)
                                pat |= pdist < 2u ? pat << 8 : 0ull;
                                pat |= pdist < 4u ? pat << 16 : 0ull;
                                pat |= pat << 32;
                                pa.x = pa.z = (uint32_t)pat;
                                pa.y = pa.w = (uint32_t)(pat >> 32);
                                pb = pc = pd = pa;
                                // (✗ run pieces of up to 192 bytes -- the pattern needs no request, a 150-byte QUAL run would take one
                                //  step instead of three -- measured no gain: 289 / 183 / 132 GB/s against 292 / 187 / 134 with 64-byte
                                //  pieces; eight more conditional stores in every step cost what the steps saved)
                            } else {
                                store_run(pdst, pat, pdist, plen);   // (periods 3, 5, 6, 7: rare)
                                common = false;
                            }
                        }
                        if (common) {
                            if (loose) {
                                store_u128(pdst, pa);
                                if (plen > 16u) store_u128(pdst + 16, pb);
                                if (plen > 32u) store_u128(pdst + 32, pc);
                                if (plen > 48u) store_u128(pdst + 48, pd);
                            } else store_piece(pdst, pa, pb, pc, pd, plen);
                        }
                    } else if (pdist) store_run(pdst, ((uint64_t)pa.z | ((uint64_t)pa.w << 32)) >> (8u * (8u - pdist)), pdist, plen);
                    else if (plen > 16u) {
                        const uint4 pb = *(const uint4 *)(land + 16u * INF_WAVE + 16u * lane);
                        store_u128(pdst, pa);
                        store_tail16(pdst + 16, (uint64_t)pb.x | ((uint64_t)pb.y << 32), (uint64_t)pb.z | ((uint64_t)pb.w << 32), plen - 16u);
                    } else store_tail16(pdst, (uint64_t)pa.x | ((uint64_t)pa.y << 32), (uint64_t)pa.z | ((uint64_t)pa.w << 32), plen);
                    if (PIECES && prem) {   // the next piece of the same copy (a run's seed is re-read behind the bytes just stored)
                        pdst += plen;
                        if (pdist) lsrc = pdst - 16;
                        else if (pself) pcap = 2u * pcap <= INF_PIECE ? 2u * pcap : pcap;   // (the source stays: see below)
                        else lsrc += plen;
                        plen = min(prem, pcap);
                        prem -= plen;
                    } else {
                        plen = 0;
                        lsrc = comp;
                    }
                }
                br.take_group(*(const uint4 *)(land + STREAM_PLANE + 16u * lane));
                if (run) {
                    if (run > isize - pos) return INF_OVERRUN;
                    if (PIECES && pos + 8u <= isize) store_u64(out + pos, lits);   // (the bytes behind the literals: filled later, as above)
                    else store_tail(out + pos, lits, run);
                    pos += run;
                }
                if (sym == 256) {
                    if constexpr (STAMP) st->v[2] += __builtin_amdgcn_s_memtime() - ts2;
                    br.request();   // (the headers read the stream with ordinary loads again)
                    break;
                }
                if (len) {
                    if (dist > pos) return INF_BAD_DISTANCE;
                    if (len > isize - pos) return INF_OVERRUN;
                    uint8_t *dst = out + pos;
                    const uint8_t *src = dst - dist;
                    if (dist < 8u && pos >= 16u) {
                        lsrc = dst - 16;   // (the seed is the upper half of the first plane)
                        pdst = dst; pdist = dist;
                        if constexpr (PIECES) { pcap = INF_PIECE; plen = min(len, pcap); prem = len - plen; }
                        else plen = len;
                    } else if (PIECES && dist >= 8u) {
                        // pieces of <= 64 bytes, never longer than what lies between source and destination: a piece's source
                        // is wholly in front of its destination, i.e. in bytes stored (issued) before its request is.
                        // A copy that feeds on itself (dist < 64: the period-8..15 matches of binned quality strings used to be
                        // copied IN PLACE, 8 bytes per load round trip, with every other lane of the wave waiting -- 1.7 % of
                        // a lane's steps, i.e. two wave-steps in three) reads every piece from the SAME source: after pieces
                        // of dist, 2 dist, 4 dist ... bytes the destination is a whole number of periods ahead and twice as
                        // many valid bytes lie behind the source.
                        lsrc = src;
                        pdst = dst; pdist = 0u;
                        pself = dist < INF_PIECE;
                        pcap = min(INF_PIECE, dist); plen = min(len, pcap); prem = len - plen;
                    } else if (!PIECES && dist >= 8u && len <= 32u && dist >= len) {
                        // (the 32 bytes requested may run past dst by up to 24: not-yet-written bytes of this block, of
                        //  the next one or of the buffer's slack, none of which is stored)
                        lsrc = src;
                        pdst = dst; plen = len; pdist = 0u;
                    } else if (dist >= 64u) {
                        // long far matches, in place: 32 bytes per step, the loads of step i+1 issued before the stores of step i
                        uint32_t n = len;
                        uint4 a = load_u128(src), b = load_u128(src + 16);
                        while (n > 32u) {
                            const uint4 na = load_u128(src + 32), nb = load_u128(src + 48);   // (src + 64 <= dst)
                            store_u128(dst, a);
                            store_u128(dst + 16, b);
                            a = na; b = nb;
                            dst += 32; src += 32; n -= 32u;
                        }
                        if (n >= 16u) { store_u128(dst, a); dst += 16; n -= 16u; a = b; }
                        store_tail16(dst, (uint64_t)a.x | ((uint64_t)a.y << 32), (uint64_t)a.z | ((uint64_t)a.w << 32), n);
                    } else if (dist >= 8u) {
                        // 8 <= dist < 64 and the copy feeds on itself or is long: 8 bytes at a time, in place
                        uint32_t n = len;
                        while (n >= 8u) { store_u64(dst, load_u64(src)); dst += 8; src += 8; n -= 8u; }
                        if (n) store_tail(dst, load_u64(src), n);   // (src + n <= dst: the bytes used are old ones)
                    } else {
                        for (uint32_t i = 0; i < len; i++) dst[i] = dst[(int)i - (int)dist];   // a short period in the first bytes of a block
                    }
                    if constexpr (STAMP) st->v[6] += plen ? 0u : 1u;
                    pos += len;
                }
                if constexpr (STAMP) st->v[2] += __builtin_amdgcn_s_memtime() - ts2;
            }
            } else {
            for (;;) {
                // Literal run first, in its own inner loop: a literal costs a decode and a fire-and-forget
                // byte store, a match costs a load round trip -- and on a lock-stepped wave of 64 streams
                // SOME lane has a match at nearly every step.  Letting every lane run through up to
                // INF_RUN_INPLACE literals before the wave turns to the matches makes the round trip a cost
                // per (literal run + match), not per token.
                int sym;
                uint32_t run = 0;
                for (;;) {
                    br.refill();
                    sym = huff_decode<15, true>(br, t, L_DELTA, L_SYM, 288, lu);
                    if (sym < 0 || sym >= 256) break;
                    if (pos + run >= isize) return INF_OVERRUN;
                    out[pos + run] = (uint8_t)sym;
                    if (++run == INF_RUN_INPLACE) { sym = 512; break; }   // budget used up: give the matches their turn
                }
                pos += run;
                if (sym < 0) return INF_BAD_SYMBOL;
                if (sym == 512) continue;
                if (sym == 256) break;
                if (sym > 285) return INF_BAD_SYMBOL;
                // length: 257..264 -> 3..10; 265..284 -> ((4 + (s-265)%4) << e) + 3 with e = (s-261)/4 extra bits; 285 -> 258
                uint32_t len;
                {
                    const uint32_t s = (uint32_t)sym;
                    if (s < 265u) len = s - 254u;
                    else if (s == 285u) len = 258u;
                    else {
                        const uint32_t e = (s - 261u) >> 2;
                        len = ((4u + ((s - 265u) & 3u)) << e) + 3u + br.take(e);
                    }
                }
                br.refill();
                const int ds = huff_decode<15, false>(br, t, D_DELTA, D_SYM, 30, du);
                if (ds < 0 || ds >= 30) return INF_BAD_DISTANCE;   // (>= 30: an unplaced slot of an incomplete code)
                uint32_t dist;
                {
                    const uint32_t d = (uint32_t)ds;
                    if (d < 4u) dist = d + 1u;
                    else {
                        const uint32_t e = (d >> 1) - 1u;
                        dist = ((2u + (d & 1u)) << e) + 1u + br.take(e);
                    }
                }
                if (dist > pos) return INF_BAD_DISTANCE;
                if (len > isize - pos) return INF_OVERRUN;
                uint8_t *dst = out + pos;
                pos += len;
                // (every path below issues its loads first and its stores last: a store is never waited
                //  for, a load is waited for once; the over-reads stay inside the output buffer's slack)
                if (dist >= 64u) {
                    // far matches (the usual case in a BAM: the previous record)
                    const uint8_t *src = dst - dist;
                    {
                        // 32 bytes per step, and the loads of step i+1 are issued BEFORE the stores of step i (they
                        // cannot overlap them: src + 64 <= dst) -- the wait for a load then does not include the
                        // younger stores
                        uint4 a = load_u128(src), b = load_u128(src + 16);
                        while (len >= 64u) {
                            const uint4 na = load_u128(src + 32), nb = load_u128(src + 48);
                            store_u128(dst, a);
                            store_u128(dst + 16, b);
                            a = na; b = nb;
                            dst += 32; src += 32; len -= 32u;
                        }
                        if (len > 32u) {
                            const uint4 na = load_u128(src + 32), nb = load_u128(src + 48);   // (src + 64 <= dst)
                            store_u128(dst, a);
                            store_u128(dst + 16, b);
                            a = na; b = nb;
                            dst += 32; src += 32; len -= 32u;
                        }
                        // 1..32 bytes left, all of them in (a, b)
                        if (len >= 16u) { store_u128(dst, a); dst += 16; len -= 16u; a = b; }
                        store_tail16(dst, (uint64_t)a.x | ((uint64_t)a.y << 32), (uint64_t)a.z | ((uint64_t)a.w << 32), len);
                    }
                } else if (dist >= 8u) {
                    const uint8_t *src = dst - dist;
                    while (len >= 8u) { store_u64(dst, load_u64(src)); dst += 8; src += 8; len -= 8u; }
                    if (len) store_tail(dst, load_u64(src), len);   // (src + len <= dst: the bytes used are old ones)
                } else {
                    // periodic pattern of period `dist` in a register: no load-after-store chain
                    uint64_t pat;
                    if (pos - len >= 8u) pat = load_u64(dst - 8) >> (8u * (8u - dist));   // the last `dist` bytes written
                    else {
                        pat = 0;
                        for (uint32_t i = 0; i < dist; i++) pat |= (uint64_t)dst[(int)i - (int)dist] << (8u * i);
                    }
                    store_run(dst, pat, dist, len);
                }
            }
            }
            if (br.overrun()) return INF_TRUNCATED;
        } else {
            return INF_BAD_BLOCK;
        }
        if (bfinal) break;
    }
    if (br.overrun()) return INF_TRUNCATED;
    return pos == isize ? INF_OK : INF_SHORT;
}

// grid of single-wave workgroups, each wave takes 64 consecutive blocks at a time
template <bool DEFER, bool PIECES = false, bool STAMP = false>
__global__ void __launch_bounds__(INF_WAVE) bgzf_inflate_kernel(const uint8_t *comp, uint64_t comp_bytes, BgzfBlock *blocks,
                                                                uint32_t n_blocks, uint8_t *out, uint32_t lit_run, unsigned long long *dbg = nullptr,
                                                                uint32_t only_status = 0xFFFFFFFFu /* != ~0: only the blocks left in that state */) {
    extern __shared__ __attribute__((aligned(16))) uint8_t inf_lds[];
    const uint32_t lane = threadIdx.x;
    const LaneLds t{inf_lds + 4u * lane, inf_lds + INF_N16 * 2 * INF_WAVE + 4u * lane,
                    (uint32_t *)(inf_lds + (INF_N16 * 2 + INF_N8) * INF_WAVE) + lane};
    StepStamps stamps{};
    uint64_t t_kernel0 = 0;
    if constexpr (STAMP) t_kernel0 = __builtin_amdgcn_s_memtime();
    for (uint32_t g = blockIdx.x; g * INF_WAVE < n_blocks; g += gridDim.x) {
        const uint32_t i = g * INF_WAVE + lane;
        if (i < n_blocks) {
            const BgzfBlock b = blocks[i];
            if (only_status != 0xFFFFFFFFu && b.status != only_status) continue;
            uint32_t st = INF_OK;
            // (a descriptor the caller got wrong must not become a wild address)
            if (b.isize > 65536u || b.in_off > comp_bytes || b.in_len > comp_bytes - b.in_off) st = INF_BAD_BLOCK;
            else if (b.isize) st = inflate_block<DEFER, PIECES, STAMP>(comp, comp_bytes, b, out, t, inf_lds + INF_LDS_BYTES, lane, min(max(lit_run, 1u), INF_RUN_DEFER_MAX), &stamps);
            blocks[i].status = st;
        }
    }
    if constexpr (STAMP) {   // lane sums; [7] = lane-cycles inside the kernel (every lane counts the wave's whole life)
        stamps.v[7] = __builtin_amdgcn_s_memtime() - t_kernel0;
        if (dbg)
            for (int k = 0; k < 8; k++) atomicAdd(&dbg[k], (unsigned long long)stamps.v[k]);
    }
}

// ---------------------------------------------------------------------------------------
// CRC-32 (IEEE 802.3, reflected, as gzip) of every inflated block
// ---------------------------------------------------------------------------------------
constexpr uint32_t CRC_POLY = 0xEDB88320u;
constexpr uint32_t CRC_CHUNK = 1024;   // bytes per lane

// a(x) * b(x) mod P in the reflected representation (bit 31 = x^0)
__device__ __forceinline__ uint32_t gf2_mul(uint32_t a, uint32_t b) {
    uint32_t r = 0u;
#pragma unroll 8
    for (int i = 0; i < 32; i++) {
        r ^= (b & 0x80000000u) ? a : 0u;   // b's x^i term, i counted from bit 31
        b <<= 1;
        a = (a >> 1) ^ ((a & 1u) ? CRC_POLY : 0u);   // a *= x
    }
    return r;
}

// one wave per block: lane c owns the 1 KiB chunk that ends (63 - c) KiB before the block's end;
// chunk CRCs (no pre/post conditioning except the 0xFFFFFFFF start on the first data byte) are
// moved to the end of the block by multiplying with x^(8 * bytes behind the chunk) -- powers of
// x^(8 * 1024) from a 64-entry table -- and XORed together.
constexpr uint32_t CRC_REP = 32;   // copies of every table entry, one per LDS bank
constexpr uint32_t CRC_LDS_BYTES = 4u * 256u * CRC_REP * 4u;   // 128 KiB: one workgroup of 16 waves per CU
// REP = 32: the kernel on its own (16 waves per CU, conflict-free tables).  REP = 4: 16 KiB of tables, four waves -- slower
// by itself, but it fits a CU beside the four workgroups of the inflate kernel (133 KB of the 160), for a feed whose next
// inflate launch runs while this super-batch is checked (PSSBAM_FEED_INFLATE_STREAMS=2).
template <uint32_t REP>
__global__ void __launch_bounds__(REP >= 32u ? 1024 : 256) bgzf_crc_kernel(const uint8_t *out, BgzfBlock *blocks, uint32_t n_blocks,
                                                        const uint32_t *xpow_kib /* [64]: x^(8*1024*k) mod P */) {
    // slicing-by-4 tables, every entry replicated 32 times side by side: lane l reads copy l % 32, i.e. ALWAYS
    // bank l % 32 -- the random table indices of a wave's 64 lanes no longer collide (a single copy of the
    // tables had 67 % of the LDS cycles lost to bank conflicts: profiles/r02_inflate_prof_100M_5waves.txt)
    extern __shared__ uint32_t crc_tab[];   // [4][256][REP]
    for (uint32_t i = threadIdx.x; i < 4u * 256u * REP; i += blockDim.x) {
        const uint32_t t = i / (256u * REP), v = (i / REP) & 255u;
        uint32_t c = v;   // CRC register after byte v followed by t zero bytes
        for (uint32_t k = 0; k < 8u * (t + 1u); k++) c = (c >> 1) ^ ((c & 1u) ? CRC_POLY : 0u);
        crc_tab[i] = c;
    }
    __syncthreads();
    const uint32_t rep = threadIdx.x & (REP - 1u);
    const uint32_t *tab0 = crc_tab + rep, *tab1 = tab0 + 256u * REP, *tab2 = tab1 + 256u * REP, *tab3 = tab2 + 256u * REP;
#define TAB(t, v) (t)[(v) * REP]
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    for (uint32_t bi = blockIdx.x * waves + wave; bi < n_blocks; bi += gridDim.x * waves) {
        const BgzfBlock b = blocks[bi];
        if (b.status != INF_OK) continue;
        const uint32_t isize = b.isize;
        const uint32_t behind = (63u - lane) * CRC_CHUNK;                 // bytes after this lane's chunk
        const uint32_t e = isize > behind ? isize - behind : 0u;          // chunk = [s, e)
        const uint32_t s = e > CRC_CHUNK ? e - CRC_CHUNK : 0u;
        uint32_t crc = (s == 0u && e > 0u) ? 0xFFFFFFFFu : 0u;
        const uint8_t *p = out + b.out_off;
        // 16 bytes per load: the lanes of a wave read 1 KiB apart (one cache line each), so a dword at a
        // time would fetch every line 32 times over
        uint32_t i = s;
        for (; i < e && ((uintptr_t)(p + i) & 15u); i++) crc = TAB(tab0, (crc ^ p[i]) & 0xFFu) ^ (crc >> 8);
        // eight loads in flight per lane (a load at a time made the kernel a chain of memory round trips: 64 of them
        // per chunk).  (Requesting the NEXT eight before these are worked through -- two register sets taking turns,
        // the requests pinned in place -- changed nothing: 283.8 vs 292 GB/s inflate + CRC on another box, round 3.)
        for (; i + 128u <= e; i += 128u) {
            uint4 q[8];
#pragma unroll
            for (int j = 0; j < 8; j++) q[j] = *(const uint4 *)(p + i + 16u * (uint32_t)j);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t d[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t w = d[k] ^ crc;
                    crc = TAB(tab3, w & 0xFFu) ^ TAB(tab2, (w >> 8) & 0xFFu) ^ TAB(tab1, (w >> 16) & 0xFFu) ^ TAB(tab0, w >> 24);
                }
            }
        }
        for (; i + 16u <= e; i += 16u) {
            const uint4 q = *(const uint4 *)(p + i);
            const uint32_t d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t w = d[k] ^ crc;
                crc = TAB(tab3, w & 0xFFu) ^ TAB(tab2, (w >> 8) & 0xFFu) ^ TAB(tab1, (w >> 16) & 0xFFu) ^ TAB(tab0, w >> 24);
            }
        }
        for (; i < e; i++) crc = TAB(tab0, (crc ^ p[i]) & 0xFFu) ^ (crc >> 8);
        if (behind && e > 0u) crc = gf2_mul(crc, xpow_kib[63u - lane]);
        if (e == 0u) crc = 0u;
        for (int o = 32; o >= 1; o >>= 1) crc ^= (uint32_t)__shfl_xor((int)crc, o);
        if (lane == 0u) {
            const uint32_t final_crc = isize ? ~crc : 0u;
            if (final_crc != b.crc) blocks[bi].status = INF_BAD_CRC;
        }
    }
#undef TAB
}

// The same check with WHOLE LINES per request (round 3, profiles/r03_crc_pmc.txt: the kernel above fetches every 128-byte
// line from the fabric twice -- its lanes sit 1 KiB apart and ask for their line 16 bytes at a time -- and sits on the
// HBM ceiling at 2x the data).  Here the lanes of a wave read ADJACENT 16-byte pieces: lane l owns pieces l, l + 64,
// l + 128, ... of the block, one wave-instruction fetches 1 KiB of consecutive bytes.  CRC is linear, so a lane keeps
//     acc <- acc * x^(8*1024)  +  R(piece)        R = the register after the piece's 16 bytes from state 0
// (Horner over its pieces, which lie 1 KiB apart; the multiplication by the fixed x^8192 is four more table look-ups,
// tables Z0..Z3, worked out on the host), and at the end its sum moves to the end of the full pieces by x^(128 * pieces
// behind it) -- a 64-entry power table -- and the lanes' sums are XORed.  Bytes in front of the first 16-byte aligned
// piece and behind the last full one (<= 15 each) go through the byte table in lane 0.  Eight tables, sixteen copies of
// every entry side by side (lane l reads copy l % 16): 128 KiB, one 16-wave workgroup per CU as before.
constexpr uint32_t CRC2_REP = 16;
constexpr uint32_t CRC2_LDS_BYTES = 8u * 256u * CRC2_REP * 4u;
__global__ void __launch_bounds__(1024) bgzf_crc_lines_kernel(const uint8_t *out, BgzfBlock *blocks, uint32_t n_blocks,
                                                              const uint32_t *xpow16 /* [64]: x^(8*16*k) mod P */,
                                                              const uint32_t *ztab /* [4][256]: the register moved on by 1024 zero bytes */) {
    extern __shared__ uint32_t crc2_tab[];   // [8][256][CRC2_REP]
    for (uint32_t i = threadIdx.x; i < 8u * 256u * CRC2_REP; i += blockDim.x) {
        const uint32_t t = i / (256u * CRC2_REP), v = (i / CRC2_REP) & 255u;
        uint32_t c;
        if (t < 4u) {
            c = v;   // CRC register after byte v followed by t zero bytes
            for (uint32_t k = 0; k < 8u * (t + 1u); k++) c = (c >> 1) ^ ((c & 1u) ? CRC_POLY : 0u);
        } else c = ztab[(t - 4u) * 256u + v];
        crc2_tab[i] = c;
    }
    __syncthreads();
    const uint32_t rep = threadIdx.x & (CRC2_REP - 1u);
    const uint32_t *tab0 = crc2_tab + rep, *tab1 = tab0 + 256u * CRC2_REP, *tab2 = tab1 + 256u * CRC2_REP, *tab3 = tab2 + 256u * CRC2_REP;
    const uint32_t *z0 = tab3 + 256u * CRC2_REP, *z1 = z0 + 256u * CRC2_REP, *z2 = z1 + 256u * CRC2_REP, *z3 = z2 + 256u * CRC2_REP;
#define TAB2(t, v) (t)[(v) * CRC2_REP]
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    for (uint32_t bi = blockIdx.x * waves + wave; bi < n_blocks; bi += gridDim.x * waves) {
        const BgzfBlock b = blocks[bi];
        if (b.status != INF_OK) continue;
        const uint32_t T = b.isize;
        const uint8_t *p = out + b.out_off;
        const uint32_t head = min((16u - (uint32_t)((uintptr_t)p & 15u)) & 15u, T);   // bytes in front of the first aligned piece
        const uint32_t n_full = (T - head) >> 4, tail = (T - head) & 15u;
        const uint4 *pieces = (const uint4 *)(p + head);
        // lane 0 starts the block: 0xFFFFFFFF through the head bytes is the state its first piece starts from
        uint32_t s0 = 0u;
        if (lane == 0u) {
            s0 = 0xFFFFFFFFu;
            for (uint32_t i = 0; i < head; i++) s0 = TAB2(tab0, (s0 ^ p[i]) & 0xFFu) ^ (s0 >> 8);
        }
        uint32_t acc = 0u, n_mine = 0u;
        for (uint32_t i0 = lane; i0 < n_full; i0 += 8u * 64u) {
            uint4 q[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (i0 + 64u * (uint32_t)u < n_full) q[u] = pieces[i0 + 64u * (uint32_t)u];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (i0 + 64u * (uint32_t)u < n_full) {
                    const uint32_t d[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
                    uint32_t r = n_mine ? 0u : s0;   // (s0 is 0 in every lane but the block's first)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t w = d[k] ^ r;
                        r = TAB2(tab3, w & 0xFFu) ^ TAB2(tab2, (w >> 8) & 0xFFu) ^ TAB2(tab1, (w >> 16) & 0xFFu) ^ TAB2(tab0, w >> 24);
                    }
                    // acc * x^8192: the register moved on by the 1024 bytes between this lane's pieces
                    const uint32_t m = TAB2(z0, acc & 0xFFu) ^ TAB2(z1, (acc >> 8) & 0xFFu) ^ TAB2(z2, (acc >> 16) & 0xFFu) ^ TAB2(z3, acc >> 24);
                    acc = (n_mine ? m : 0u) ^ r;
                    n_mine++;
                }
            }
        }
        // to the end of the full pieces: 16 * (pieces behind this lane's last one) bytes, 0..63 pieces
        if (n_mine) {
            const uint32_t last = lane + 64u * (n_mine - 1u);
            const uint32_t behind = n_full - 1u - last;
            if (behind) acc = gf2_mul(acc, xpow16[behind]);
        }
        for (int o = 32; o >= 1; o >>= 1) acc ^= (uint32_t)__shfl_xor((int)acc, o);
        if (lane == 0u) {
            uint32_t crc = n_full ? acc : s0;   // (no full piece: the head was everything so far)
            const uint8_t *t8 = p + head + 16u * n_full;
            for (uint32_t i = 0; i < tail; i++) crc = TAB2(tab0, (crc ^ t8[i]) & 0xFFu) ^ (crc >> 8);
            const uint32_t final_crc = T ? ~crc : 0u;
            if (final_crc != b.crc) blocks[bi].status = INF_BAD_CRC;
        }
    }
#undef TAB2
}


// ---------------------------------------------------------------------------------------
// record index of an inflated super-batch, on the device
// ---------------------------------------------------------------------------------------
// The tally kernels want a u32 offset per alignment record, i.e. the block_size chain of the BAM
// stream -- a serial chain.  It is cut at the BGZF blocks: every block's lane looks for the first
// offset in ITS block from which a chain of plausible records runs to the block's end (offset 0 in
// files written by htslib, whose blocks start on record boundaries; somewhere in the first record's
// length in files written by htsjdk, whose records cross blocks), walks it, and reports where its
// last record ends.  A verification pass then checks that every block's chain ends exactly where the
// next block's begins; if so the pieces ARE the chain, and counts -> scan -> offsets follow in
// parallel.  If not (a false start, a block inside a record longer than a block whose neighbours
// disagree, ...) FEED_RAGGED is raised and the caller falls back to the host reader, whose indexer
// walks serially.  The partial record a super-batch ends with is copied in front of the next
// super-batch's data (bgzf_chain_carry), whose block 0 starts its walk there.
enum : uint32_t { FEED_BAD_BLOCK = 1u, FEED_RAGGED = 2u, FEED_BAD_RECORD = 4u, FEED_TRUNCATED = 8u };
constexpr uint64_t CHAIN_NONE = ~0ull;
constexpr uint32_t CHAIN_MAX_RECORD = 1u << 24;   // records above 16 MiB (the carry gap) go the host way

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// the layout arithmetic of SAM spec 4.2 + value ranges (host twin: plausible_record, bam_reader.c)
__device__ __forceinline__ bool plausible_record(const uint8_t *p, uint64_t avail, int32_t n_ref) {
    if (avail < 36u) return false;
    const uint32_t bs = load_u32_unaligned(p);
    const int32_t ref_id = (int32_t)load_u32_unaligned(p + 4), pos = (int32_t)load_u32_unaligned(p + 8);
    const uint32_t w3 = load_u32_unaligned(p + 12), w4 = load_u32_unaligned(p + 16), l_seq = load_u32_unaligned(p + 20);
    const int32_t next_ref = (int32_t)load_u32_unaligned(p + 24), next_pos = (int32_t)load_u32_unaligned(p + 28);
    const uint32_t l_name = w3 & 0xFFu, n_cig = w4 & 0xFFFFu;
    if (bs < 32u || bs > CHAIN_MAX_RECORD || l_name == 0u) return false;
    if (ref_id < -1 || ref_id >= n_ref || next_ref < -1 || next_ref >= n_ref || pos < -1 || next_pos < -1) return false;
    if (l_seq > bs) return false;
    if (32ull + l_name + 4ull * n_cig + ((uint64_t)l_seq + 1ull) / 2ull + l_seq > bs) return false;
    if (36ull + l_name <= avail && p[36u + l_name - 1u] != 0u) return false;   // the read name is NUL-terminated
    return true;
}

// per block: a = absolute offset of the first record starting in it (CHAIN_NONE: none), n = records
// starting in it, e = where the last of them ends, last = where the last of them starts
__global__ void __launch_bounds__(256) bgzf_chain_spec(const uint8_t *out, const BgzfBlock *blocks, uint32_t n_blocks, uint64_t data_end,
                                                       const uint64_t *first_start, int32_t n_ref, uint64_t *a, uint32_t *n,
                                                       uint64_t *e, uint64_t *last, uint32_t *flags, uint64_t *sb_words) {
    if (blockIdx.x == 0 && threadIdx.x == 0) sb_words[0] = sb_words[1] = 0ull;   // this super-batch's "links broken" / "repaired" words
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += gridDim.x * blockDim.x) {
        const BgzfBlock blk = blocks[b];
        uint64_t a_b = CHAIN_NONE, e_b = 0, last_b = 0;
        uint32_t n_b = 0;
        if (blk.status != INF_OK) atomicOr(flags, FEED_BAD_BLOCK);
        else {
            const uint64_t lo = blk.out_off, hi = lo + blk.isize;
            if (b == 0u) {
                // the chain's known position: the carried partial record (in the gap in front of the data), or
                // the first byte behind the BAM header
                uint64_t o = *first_start;
                a_b = o;
                while (o < hi && o + 4u <= data_end) {
                    const uint32_t bs = load_u32_unaligned(out + o);
                    if (bs < 32u || bs > CHAIN_MAX_RECORD) { atomicOr(flags, bs < 32u ? FEED_BAD_RECORD : FEED_RAGGED); break; }
                    last_b = o;
                    o += 4ull + bs;
                    n_b++;
                }
                e_b = o;
                if (!n_b) a_b = CHAIN_NONE;
            } else {
                // (the bytes behind the block's end are the next block's: the buffer is contiguous, so a record
                //  that starts in the last bytes of the block can be judged too; a candidate must also
                //  survive two records beyond the block -- with small blocks its own record says little)
                // Scan and walk are separate loops on purpose: the lanes of a wave find their candidates at
                // different offsets, and with the walk nested inside the scan every lane walked its block ALONE
                // while the others waited for the scan to come round to them (18 ms per super-batch of a file
                // whose records cross blocks; 1.5 ms with htslib's layout, where every candidate is offset 0).
                uint32_t tries = 0;
                uint64_t c = lo;
                while (tries < 4u) {
                    while (c < hi && !plausible_record(out + c, data_end - c, n_ref)) c++;
                    if (c >= hi) break;
                    tries++;
                    uint64_t o = c, l = c, end_in = c;
                    uint32_t k = 0, extra = 0;
                    bool good = true;
                    for (;;) {
                        if (o + 4u > data_end) break;   // the length word itself is cut off: the tail
                        const uint32_t bs = load_u32_unaligned(out + o);
                        // a record cut off by the end of the data cannot be judged by its fields: its length decides
                        const bool near_end = data_end - o < 36u + 256u;
                        if (!(plausible_record(out + o, data_end - o, n_ref) || (near_end && bs >= 32u && bs <= CHAIN_MAX_RECORD))) { good = false; break; }
                        if (o < hi) { k++; l = o; end_in = o + 4ull + bs; }
                        else if (++extra >= 2u) break;
                        o += 4ull + bs;
                        if (o >= data_end) break;
                    }
                    if (good && k) { a_b = c; n_b = k; e_b = end_in; last_b = l; break; }
                    c++;
                }
            }
        }
        a[b] = a_b;
        n[b] = n_b;
        e[b] = e_b;
        last[b] = last_b;
    }
}

// nexta[b] = the first record start at or after block b (suffix minimum of a[]; a[] grows with b), nexta[n] = NONE
__global__ void __launch_bounds__(1024) bgzf_chain_suffix(const uint64_t *a, uint32_t n, uint64_t *nexta, const uint64_t *only_if = nullptr) {
    if (only_if && !*only_if) return;   // (the pass behind bgzf_chain_repair: nothing was repaired)
    __shared__ uint64_t part[1024];
    const uint32_t t = threadIdx.x, per = (n + 1023u) / 1024u;
    const uint32_t lo = min(n, t * per), hi = min(n, lo + per);
    uint64_t m = CHAIN_NONE;
    for (uint32_t i = lo; i < hi; i++) m = min(m, a[i]);
    part[t] = m;
    __syncthreads();
    for (uint32_t d = 1u; d < 1024u; d <<= 1) {   // suffix minimum over the strips
        const uint64_t v = t + d < 1024u ? part[t + d] : CHAIN_NONE;
        __syncthreads();
        part[t] = min(part[t], v);
        __syncthreads();
    }
    uint64_t run = t + 1u < 1024u ? part[t + 1u] : CHAIN_NONE;   // everything behind this strip
    for (uint32_t i = hi; i > lo; i--) { run = min(run, a[i - 1u]); nexta[i - 1u] = run; }
    if (t == 0u) nexta[n] = CHAIN_NONE;
}

// checks that the per-block chains link up, settles the records of the last block against the end of
// the data (an incomplete last record is the tail), writes counts[] and the tail's start.
// Two passes (SECOND = false, true) around bgzf_chain_repair: the first notes broken links in the super-batch's own
// word (sb_words[0]) for the repair kernel, the second runs only if something was repaired (sb_words[1]) and raises
// what is still broken then.
template <bool SECOND>
__global__ void __launch_bounds__(256) bgzf_chain_verify(const uint8_t *out, const uint32_t *n, const uint64_t *e, const uint64_t *last,
                                                         const uint64_t *nexta, uint32_t n_blocks, uint64_t data_end, const uint64_t *first_start,
                                                         uint32_t *counts, uint64_t *tail_start, uint32_t *flags, uint64_t *sb_words) {
    if (SECOND && !sb_words[1]) return;
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += gridDim.x * blockDim.x) {
        uint32_t c = n[b];
        bool broken = false;
        if (c) {
            const uint64_t next = nexta[b + 1u];
            if (next != CHAIN_NONE) {
                if (e[b] != next) broken = true;
            } else if (e[b] == data_end) *tail_start = data_end;
            else if (e[b] > data_end) { c--; *tail_start = last[b]; }       // the last record runs past the data: the tail
            else if (data_end - e[b] < 4u) *tail_start = e[b];             // a cut-off length word: the tail
            else {
                // a record starts at e[b] that no block claimed (too little of it is there to be judged by its
                // fields): fine if it is the cut-off tail, a broken chain if it is whole
                const uint32_t bs = load_u32_unaligned(out + e[b]);
                if (bs >= 32u && bs <= CHAIN_MAX_RECORD && e[b] + 4ull + bs > data_end) *tail_start = e[b];
                else broken = true;
            }
        }
        counts[b] = c;
        if (b == 0u) {
            if (nexta[0] == CHAIN_NONE) *tail_start = *first_start;   // not one record starts in this super-batch: all of it is tail
            // block 0 anchors the chain at *first_start; when it starts no record itself (an empty BGZF block at the
            // head of the stream) the first candidate a LATER block found must be that very offset -- a candidate scan
            // that skipped an implausible record there would drop it silently, where the host reader diagnoses it
            else if (n[0] == 0u && nexta[0] != *first_start) broken = true;
        }
        if (broken) {
            if (SECOND) atomicOr(flags, FEED_RAGGED);
            else atomicOr((unsigned long long *)&sb_words[0], 1ull);
        }
    }
}

// Repairs broken links instead of giving the file up.  Every block guessed its first record start on its own; where a
// guess was wrong (bytes inside a record that look like a chain of records: a B-array of aux data, a long read's
// qualities) the chain coming from the LEFT is the true one -- block 0 is anchored, and a block whose start is true
// walks true.  So: find the first broken link, walk the records serially from the true end of the block on its left,
// rewriting the entries of the blocks passed (none / corrected), until the walk lands exactly on a block's own guess
// -- from there the guesses are true again -- and go on to the next broken link.  One wave; the search for broken
// links is 64 blocks wide, the walk is serial (a wrong block costs a walk over its records, ~0.5 ms; a file whose
// every block is wrong degenerates to the serial chain, as on the host).  Records are judged like the candidates
// were: one that is implausible, or longer than the carry gap, leaves FEED_RAGGED standing (the host reader words the
// diagnosis).  PSSBAM_FEED_REPAIR=0 launches the kernel with enabled = 0 (the round-2 behaviour: any broken link
// raises FEED_RAGGED).
__global__ void __launch_bounds__(64) bgzf_chain_repair(const uint8_t *out, const BgzfBlock *blocks, uint32_t n_blocks, uint64_t data_end,
                                                        const uint64_t *first_start, int32_t n_ref, uint64_t *a, uint32_t *n, uint64_t *e,
                                                        uint64_t *last, const uint64_t *nexta, uint32_t *flags, uint64_t *sb_words, int enabled) {
    if (!sb_words[0]) return;
    if (!enabled) {
        if (threadIdx.x == 0) atomicOr(flags, FEED_RAGGED);
        return;
    }
    const uint32_t lane = threadIdx.x;
    uint32_t from = 0;             // links left of block `from` hold
    bool failed = false, any = false;
    // the anchor: block 0 starts no record, and what the later blocks found does not begin at *first_start
    uint64_t pos = 0;
    uint32_t bi = 0;
    bool walking = false;
    if (n[0] == 0u && nexta[0] != CHAIN_NONE && nexta[0] != *first_start) { pos = *first_start; bi = 0; walking = true; }
    for (;;) {
        if (!walking) {
            // the next broken link at or behind `from`, 64 blocks at a time
            uint32_t found = 0xFFFFFFFFu;
            for (uint32_t base = from; base < n_blocks && found == 0xFFFFFFFFu; base += 64u) {
                const uint32_t b = base + lane;
                bool broken = false;
                if (b < n_blocks && n[b]) {
                    const uint64_t next = nexta[b + 1u], eb = e[b];
                    if (next != CHAIN_NONE) broken = eb != next;
                    else if (eb < data_end && data_end - eb >= 4u) {
                        const uint32_t bs = load_u32_unaligned(out + eb);
                        broken = !(bs >= 32u && bs <= CHAIN_MAX_RECORD && eb + 4ull + bs > data_end);
                    }
                }
                const unsigned long long m = __ballot(broken);
                if (m) found = base + (uint32_t)__ffsll((long long)m) - 1u;
            }
            if (found == 0xFFFFFFFFu) break;
            pos = e[found];
            bi = found + 1u;
            walking = true;
        }
        // (every lane walks the same addresses; lane 0 writes)
        any = true;
        while (bi < n_blocks) {
            const BgzfBlock blk = blocks[bi];
            const uint64_t lo = blk.out_off, hi = lo + blk.isize;
            if (pos >= hi || blk.isize == 0u) {   // no record starts in this block
                if (lane == 0u) { a[bi] = CHAIN_NONE; n[bi] = 0u; e[bi] = 0ull; last[bi] = 0ull; }
                bi++;
                continue;
            }
            if (n[bi] && a[bi] == pos) break;     // the block's own guess is the true start: in step again
            if (pos >= data_end) break;
            uint64_t first = pos, l = pos;
            uint32_t k = 0;
            while (pos < hi) {
                if (pos + 4u > data_end) break;   // a cut-off length word: the tail
                const uint32_t bs = load_u32_unaligned(out + pos);
                const bool near_end = data_end - pos < 36u + 256u;
                if (!(plausible_record(out + pos, data_end - pos, n_ref) || (near_end && bs >= 32u && bs <= CHAIN_MAX_RECORD))) { failed = true; break; }
                l = pos;
                pos += 4ull + bs;
                k++;
            }
            if (failed) break;
            if (lane == 0u) {
                a[bi] = k ? first : CHAIN_NONE;
                n[bi] = k;
                e[bi] = k ? pos : 0ull;
                last[bi] = k ? l : 0ull;
            }
            if (pos < hi) {   // stopped at the cut-off length word: nothing starts behind it
                for (uint32_t r = bi + 1u; r < n_blocks; r++)
                    if (lane == 0u) { a[r] = CHAIN_NONE; n[r] = 0u; e[r] = 0ull; last[r] = 0ull; }
                bi = n_blocks;
                break;
            }
            bi++;
        }
        if (failed) break;
        walking = false;
        from = bi;
        if (bi >= n_blocks) break;
    }
    if (lane == 0u) {
        if (failed) atomicOr(flags, FEED_RAGGED);
        else if (any) sb_words[1] = 1ull;
    }
}

// exclusive scan of up to a few hundred thousand counts by one workgroup; *total = their sum
__global__ void __launch_bounds__(1024) bgzf_index_scan(const uint32_t *counts, uint32_t n, uint32_t *base, uint32_t *total) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x, per = (n + 1023u) / 1024u;
    const uint32_t lo = min(n, t * per), hi = min(n, lo + per);
    uint32_t sum = 0u;
    for (uint32_t i = lo; i < hi; i++) sum += counts[i];
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1u; d < 1024u; d <<= 1) {
        const uint32_t v = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;   // exclusive
    for (uint32_t i = lo; i < hi; i++) { base[i] = run; run += counts[i]; }
    if (t == 1023u) *total = part[1023];
}

// offsets of the records of one tally sub-batch (blocks [0, n_blocks) of the arrays handed in), relative
// to the sub-batch's record base; the block that holds the sub-batch's last record also writes the
// end sentinel
__global__ void __launch_bounds__(256) bgzf_chain_write(const uint8_t *out, const uint64_t *a, const uint32_t *counts, const uint32_t *base,
                                                        uint32_t n_blocks, uint64_t sub_base, uint32_t *offs, const uint32_t *total) {
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += gridDim.x * blockDim.x) {
        const uint32_t c = counts[b];
        if (!c) continue;
        uint64_t o = a[b];
        uint32_t k = base[b];
        for (uint32_t i = 0; i < c; i++) {
            offs[k++] = (uint32_t)(o - sub_base);
            o += 4ull + load_u32_unaligned(out + o);
        }
        if (k == *total) offs[k] = (uint32_t)(o - sub_base);
    }
}

// The partial record a super-batch ends with travels to the next one through a small buffer of its own (the next
// super-batch's slot is not known -- maybe not even allocated -- when this one is flushed): _out copies the tail
// [*tail_start, data_end) there, _in puts it in front of the next slot's data, whose block 0 starts its walk there.
__global__ void __launch_bounds__(256) bgzf_chain_carry_out(const uint8_t *src_out, const uint64_t *tail_start, uint64_t data_end, uint8_t *carry,
                                                            uint64_t gap, uint64_t *tail_len_out, uint32_t *flags) {
    const uint64_t from = *tail_start, len = data_end > from ? data_end - from : 0ull;
    if (len > gap) {
        if (threadIdx.x == 0) { atomicOr(flags, FEED_RAGGED); *tail_len_out = 0; }
        return;
    }
    for (uint64_t i = threadIdx.x; i < len; i += blockDim.x) carry[i] = src_out[from + i];
    if (threadIdx.x == 0) *tail_len_out = len;
}
__global__ void __launch_bounds__(256) bgzf_chain_carry_in(const uint8_t *carry, const uint64_t *tail_len, uint8_t *dst_out, uint64_t gap,
                                                           uint64_t *first_start) {
    const uint64_t len = min(*tail_len, gap);
    for (uint64_t i = threadIdx.x; i < len; i += blockDim.x) dst_out[gap - len + i] = carry[i];
    if (threadIdx.x == 0) *first_start = gap - len;
}

// Several engines dealt alternating runs of ONE stream: the partial record engine A's run ends with goes to the engine
// that gets the next run through a page-locked host buffer both devices can address (_out on A's stream; _in on B's,
// behind an event).  A keeps nothing: the record is B's to complete and to tally.
__global__ void __launch_bounds__(256) bgzf_chain_handoff_out(const uint8_t *carry, uint64_t *tail_len, uint8_t *host_buf, uint64_t *host_len) {
    const uint64_t len = *tail_len;
    for (uint64_t i = threadIdx.x; i < len; i += blockDim.x) host_buf[i] = carry[i];
    __syncthreads();
    if (threadIdx.x == 0) { *host_len = len; *tail_len = 0; }
}
__global__ void __launch_bounds__(256) bgzf_chain_handoff_in(const uint8_t *host_buf, const uint64_t *host_len, uint8_t *carry, uint64_t gap, uint64_t *tail_len) {
    const uint64_t len = min(*host_len, gap);
    for (uint64_t i = threadIdx.x; i < len; i += blockDim.x) carry[i] = host_buf[i];
    if (threadIdx.x == 0) *tail_len = len;
}

// the caller declares that the next blocks do not continue the stream fed so far: a partial record
// left over at this point can never be completed
__global__ void bgzf_chain_break(uint64_t *tail_len, uint32_t *flags) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (*tail_len) atomicOr(flags, FEED_TRUNCATED);
        *tail_len = 0;
    }
}

}  // namespace pssbam
