// pss-bam_amd/csrc/inflate_wave.h -- BGZF inflate, ONE WAVE PER BLOCK, in two kernels.  AN EXPERIMENT (round 3), OFF BY
// DEFAULT: bit-exact (tests/test_gpu_inflate.py runs every inflate test through it), but slower than the lane-per-block
// kernel on every stream measured -- profiles/r03_inflate_wave.txt has the numbers and the reasons.  PSSBAM_INFLATE_WAVE=1
// routes pssbam_bgzf_inflate_host (tools/inflate_bench.py) through it; the engine feed does not use it.
//
// Why it was tried (profiles/r03_inflate_step_stamps.txt): the lane-per-block kernel (inflate_kernels.h) keeps a private
// 450-byte decode table per lane in LDS, which caps it at four waves per CU -- one per SIMD, every LDS and memory latency
// exposed, ~10 cycles per instruction -- and it has 65 536 streams in flight whose match sources (the lane's own output of
// a record ago) have long left the 32 MB of L2 when they are copied: 8-11 bytes cross the L2 <-> fabric boundary per byte
// inflated.  Here a block is decoded by the 64 lanes of ONE wave:
//
//   bgzf_tokens_kernel   Huffman decoding only.  The deflate block's tables are built once per WAVE (13 KiB of LDS with the
//                        token-start bitmap: twelve waves per CU), and the compressed bits are cut into 64 segments that
//                        the lanes decode SPECULATIVELY from guessed start offsets: a Huffman decoder that starts at a
//                        wrong bit falls into step with the true token sequence within a few dozen tokens (it is at a
//                        token boundary in the litlen state whenever the true decoder is), so lane i's chain runs into
//                        one that a later lane recorded in its own segment, and the true sequence is lane 0's chain up to
//                        where it meets lane 1's, lane 1's from there to where it meets lane 2's, ...  Four walks per
//                        segment: the own segment (A1), on to the meeting point (A2), count the output (B), emit (C):
//                        literals go straight to their place in the output, matches become 4-byte SEQUENCES (literal
//                        run, copy length, distance) in an arena, cut into pieces of <= 32 bytes.
//   bgzf_resolve_kernel  LZ77 only, in place in the output buffer: 64 sequences at a time are placed by prefix sums and
//                        copied by their lanes in rounds, as soon as their sources are final.
//
// What it showed: the walks cost ~2 000 cycles per token-iteration however the bits are fetched (global loads, LDS staging)
// and whatever decodes them (15-compare canonical, look-up tables) -- with 64 lanes SOME lane needs the rare path (a long
// code, a refill, an end of block) at nearly every iteration, so every iteration pays for every path -- and four walks of
// that per token eat what twelve waves per CU win over four.  See the profile file for what would have to change.
//
// A block this path cannot take (its sequences outgrow their share of the arena) gets INF_RETRY and is inflated by the
// lane-per-block kernel right behind, on the same stream.  What it replaces: `samtools view`, pss-bam.c:148-162.
#pragma once

#include "inflate_kernels.h"

namespace pssbam {

enum : uint32_t { INF_RETRY = 100u };   // transient: the lane-per-block kernel takes the block

// ---- sequences ---------------------------------------------------------------------------------------------------
// bits 0-7 literal run (0..255) in front of the copy, bits 8-15 copy length - 1 (1..256), bits 16-30 distance - 1, bit 31:
// has a copy.  A literal run above 255, or one that ends a lane's portion, goes out as copy-less sequences; a deflate match
// goes out as SEVERAL sequences -- pieces of <= WV_PIECE bytes, none longer than the distance, so that no piece reads
// what it writes and the pieces of a long match (a BAM's SEQ / QUAL copies) are resolved by as many lanes at once; only a
// run (distance < WV_RUN_DIST) stays whole, up to 256 bytes per sequence, and is expanded from its period.
constexpr uint32_t WV_PIECE = 32u, WV_RUN_DIST = 8u;
__device__ __forceinline__ uint32_t seq_pack(uint32_t lit, uint32_t mlen, uint32_t dist) {
    return lit | (mlen ? ((mlen - 1u) << 8) | ((dist - 1u) << 16) | 0x80000000u : 0u);
}
__device__ __forceinline__ uint32_t seq_piece_of(uint32_t dist) { return dist < WV_RUN_DIST ? 256u : min(WV_PIECE, dist); }
// entries a block of isize bytes may use (host and device agree): matches of 3-4 bytes throughout are beyond it -> INF_RETRY
__host__ __device__ inline uint64_t seq_cap_of(uint32_t isize) { return (uint64_t)isize / 4u + 1024u; }

constexpr uint32_t WV_WAVES = 1;            // waves per workgroup of the token kernel, each with its own block (16 KiB of LDS: ten per CU)
constexpr uint32_t WV_LL_BITS = 9, WV_D_BITS = 8, WV_CL_BITS = 7;   // direct look-up: a code of at most that many bits costs ONE LDS read
// bits per segment: an ODD number of dwords -- lane i's segment starts i * S / 32 dwords into the stage and the bitmap, and
// with an even stride the 64 lanes' reads would share a few of the 32 LDS banks (stride 32: all of them ONE bank)
constexpr uint32_t WV_SEG_MIN = 9u * 32u, WV_SEG_MAX = 33u * 32u;
constexpr uint32_t WV_MAP_WORDS = 64u * WV_SEG_MAX / 32u; // token-start bitmap of a chunk: 8.25 KiB

struct WaveLds {                    // per wave
    uint32_t map[WV_MAP_WORDS];     // bit (pos - chunk_base): a token starts there on the chain of the segment's owner
    // look-up by the next stream bits (first bit = bit 0).  Entry 0 = "longer code, or none": the canonical tables below decide.
    //   ll_lut: bits 0-3 code length, 4-5 kind (0 literal, 1 length symbol, 2 end of block), 6-8 extra bits, 9-17 literal / base length
    //   d_lut:  bits 0-3 code length, 4-7 extra bits, 8-22 base distance
    //   cl_lut: bits 0-2 code length, 3-7 symbol of the code-length code
    uint32_t ll_lut[1u << WV_LL_BITS];
    uint32_t d_lut[1u << WV_D_BITS];
    uint8_t cl_lut[1u << WV_CL_BITS];
    uint16_t ll_sym[288];           // literal/length symbols sorted by (code length, symbol)
    uint16_t d_sym[32];
    uint16_t ll_delta[16], d_delta[16];   // per length: sorted index of its first code minus that code (mod 2^16)
    uint32_t ll_upper[16], d_upper[16];   // per length L-1: one past its last code, left-aligned in 15 bits
    uint32_t cnt[16], offs[16];
    uint8_t lens[320];              // code lengths of the two alphabets while the tables are built
};

__device__ __forceinline__ uint64_t wv_peek(const uint8_t *payload, uint64_t bitpos) {   // >= 57 bits from bitpos on
    return load_u64(payload + (bitpos >> 3)) >> (bitpos & 7ull);
}

// A lane's view of the stream while it walks: bits straight from the compressed buffer (a block's payload is a few KB that
// 64 lanes read side by side: L1/L2 hits), eight bytes per refill, the NEXT eight always on their way already.  `buf` holds
// at least 48 valid bits after refill(): a whole token.  (Staging the chunk in LDS instead was measured: no faster, and its
// 8 KiB per wave cost more than half of the waves per CU.)
struct WvBits {
    const uint8_t *p;   // first byte not (wholly) in buf
    uint64_t buf, ahead;
    uint32_t cnt;
    __device__ __forceinline__ void start(const uint8_t *payload, uint64_t bitpos) {
        p = payload + (bitpos >> 3);
        const uint32_t sh = (uint32_t)(bitpos & 7ull);
        buf = load_u64(p) >> sh;
        cnt = 64u - sh;
        p += 8;
        ahead = load_u64(p);
    }
    __device__ __forceinline__ void refill() {
        if (cnt < 48u) {
            buf |= ahead << cnt;
            const uint32_t adv = (63u - cnt) >> 3;   // whole bytes of `ahead` that fit
            p += adv;
            cnt += 8u * adv;
            ahead = load_u64(p);
        }
    }
    __device__ __forceinline__ void drop(uint32_t n) { buf >>= n; cnt -= n; }
};

struct WvTok { uint32_t kind, val, len, dist, nbits; };   // kind: 0 literal (val), 1 match, 2 end of block, 3 no code
enum : uint32_t { TK_LIT = 0u, TK_MATCH = 1u, TK_EOB = 2u, TK_BAD = 3u };

// one canonical code from the low bits of v (first stream bit = bit 0): -> symbol, *L its length; < 0: no code
__device__ __forceinline__ int wv_code(uint64_t v, const uint32_t (&upper)[15], const uint16_t *delta, const uint16_t *sym, uint32_t n_sym, uint32_t *L) {
    const uint32_t c15 = __brev((uint32_t)v) >> 17;
    uint32_t l = 1u;
#pragma unroll
    for (int k = 0; k < 15; k++) l += c15 >= upper[k] ? 1u : 0u;
    *L = l;
    if (l > 15u) return -1;
    const uint32_t idx = ((c15 >> (15u - l)) + delta[l]) & 0xFFFFu;
    if (idx >= n_sym) return -1;
    return (int)sym[idx];
}

// the canonical way (15 thresholds, two dependent table reads per code): what the look-up tables are filled from, and the
// path of the few codes longer than their index.  Consumes the token.  (Inlined on purpose: a call would take the address
// of the caller's bit reader, which then lives in scratch memory -- every field access a round trip to memory -- and its stage
// pointer becomes a generic one: measured, the walks took 2-5 thousand cycles per token that way.)
__device__ __forceinline__ WvTok wv_token_slow(WvBits &br, const WaveLds &t, const uint32_t (&lu)[15], const uint32_t (&du)[15], uint32_t n_ll, uint32_t n_d) {
    WvTok k;
    uint32_t L;
    const int s = wv_code(br.buf, lu, t.ll_delta, t.ll_sym, n_ll, &L);
    k.val = (uint32_t)s;
    k.len = k.dist = 0u;
    k.nbits = L;
    if (s < 0 || s > 285) { k.kind = TK_BAD; return k; }
    if (s < 256) { k.kind = TK_LIT; br.drop(L); return k; }
    if (s == 256) { k.kind = TK_EOB; return k; }
    // length: 257..264 -> 3..10; 265..284 -> ((4 + (s-265)%4) << e) + 3 with e = (s-261)/4 extra bits; 285 -> 258
    const uint32_t u = (uint32_t)s;
    uint32_t used = L;
    if (u < 265u) k.len = u - 254u;
    else if (u == 285u) k.len = 258u;
    else {
        const uint32_t e = (u - 261u) >> 2;
        k.len = ((4u + ((u - 265u) & 3u)) << e) + 3u + ((uint32_t)(br.buf >> used) & ((1u << e) - 1u));
        used += e;
    }
    br.drop(used);
    uint32_t Ld;
    const int ds = wv_code(br.buf, du, t.d_delta, t.d_sym, n_d, &Ld);
    if (ds < 0 || ds >= 30) { k.kind = TK_BAD; return k; }
    uint32_t used2 = Ld;
    const uint32_t d = (uint32_t)ds;
    if (d < 4u) k.dist = d + 1u;
    else {
        const uint32_t e = (d >> 1) - 1u;
        k.dist = ((2u + (d & 1u)) << e) + 1u + ((uint32_t)(br.buf >> used2) & ((1u << e) - 1u));
        used2 += e;
    }
    br.drop(used2);
    k.kind = TK_MATCH;
    k.nbits = used + used2;
    return k;
}

// decodes AND consumes the token at the head of br (an end-of-block symbol or a bad code is left where it is)
__device__ __forceinline__ WvTok wv_token(WvBits &br, const WaveLds &t, const uint32_t (&lu)[15], const uint32_t (&du)[15], uint32_t n_ll, uint32_t n_d) {
    br.refill();
    const uint32_t e = t.ll_lut[(uint32_t)br.buf & ((1u << WV_LL_BITS) - 1u)];
    const uint32_t L = e & 15u;
    if (L == 0u) return wv_token_slow(br, t, lu, du, n_ll, n_d);
    WvTok k;
    k.kind = (e >> 4) & 3u;
    k.val = (e >> 9) & 0x1FFu;
    k.nbits = L;
    k.len = k.dist = 0u;
    if (k.kind == TK_LIT) { br.drop(L); return k; }
    if (k.kind != TK_MATCH) return k;
    const uint32_t xb = (e >> 6) & 7u;
    k.len = k.val + ((uint32_t)(br.buf >> L) & ((1u << xb) - 1u));
    br.drop(L + xb);
    const uint32_t de = t.d_lut[(uint32_t)br.buf & ((1u << WV_D_BITS) - 1u)];
    const uint32_t Ld = de & 15u;
    if (Ld == 0u) {   // a long distance code: the canonical tables (the length part is consumed already)
        uint32_t Lc;
        const int ds = wv_code(br.buf, du, t.d_delta, t.d_sym, n_d, &Lc);
        if (ds < 0 || ds >= 30) { k.kind = TK_BAD; return k; }
        const uint32_t d = (uint32_t)ds;
        uint32_t used2 = Lc;
        if (d < 4u) k.dist = d + 1u;
        else {
            const uint32_t ex = (d >> 1) - 1u;
            k.dist = ((2u + (d & 1u)) << ex) + 1u + ((uint32_t)(br.buf >> used2) & ((1u << ex) - 1u));
            used2 += ex;
        }
        br.drop(used2);
        k.nbits = L + xb + used2;
        return k;
    }
    const uint32_t xd = (de >> 4) & 15u;
    k.dist = (de >> 8) + ((uint32_t)(br.buf >> Ld) & ((1u << xd) - 1u));
    br.drop(Ld + xd);
    k.nbits = L + xb + Ld + xd;
    return k;
}

// fills the look-up tables from the canonical ones (every entry decodes its own index)
__device__ void wv_fill_luts(WaveLds &t, const uint32_t (&lu)[15], const uint32_t (&du)[15], uint32_t n_ll, uint32_t n_d, uint32_t lane) {
    for (uint32_t i = lane; i < (1u << WV_LL_BITS); i += 64u) {
        uint32_t L, e = 0u;
        const int s = wv_code((uint64_t)i, lu, t.ll_delta, t.ll_sym, n_ll, &L);
        if (s >= 0 && s <= 285 && L <= WV_LL_BITS) {
            const uint32_t u = (uint32_t)s;
            if (u < 256u) e = L | (TK_LIT << 4) | (u << 9);
            else if (u == 256u) e = L | (TK_EOB << 4);
            else {
                uint32_t xb = 0u, base;
                if (u < 265u) base = u - 254u;
                else if (u == 285u) base = 258u;
                else { xb = (u - 261u) >> 2; base = ((4u + ((u - 265u) & 3u)) << xb) + 3u; }
                e = L | (TK_MATCH << 4) | (xb << 6) | (base << 9);
            }
        }
        t.ll_lut[i] = e;
    }
    for (uint32_t i = lane; i < (1u << WV_D_BITS); i += 64u) {
        uint32_t L, e = 0u;
        const int s = wv_code((uint64_t)i, du, t.d_delta, t.d_sym, n_d, &L);
        if (s >= 0 && s < 30 && L <= WV_D_BITS) {
            const uint32_t d = (uint32_t)s;
            uint32_t xd = 0u, base = d + 1u;
            if (d >= 4u) { xd = (d >> 1) - 1u; base = ((2u + (d & 1u)) << xd) + 1u; }
            e = L | (xd << 4) | (base << 8);
        }
        t.d_lut[i] = e;
    }
}

// Canonical tables of one alphabet from lens[0..n) (LDS), by the whole wave: upper[] / delta[] for the decoder, the symbols
// sorted by (length, symbol).  -> number of symbols that have a code; 0xFFFFFFFF: over-subscribed.
__device__ uint32_t wv_build(WaveLds &t, const uint8_t *lens, uint32_t n, uint32_t *upper_out, uint16_t *delta_out, uint16_t *sym_out, uint32_t lane) {
    if (lane < 16u) t.cnt[lane] = 0u;
    for (uint32_t s = lane; s < n; s += 64u)
        if (lens[s]) atomicAdd(&t.cnt[lens[s]], 1u);
    // (the wave's LDS operations complete in order: no barrier inside one wave)
    uint32_t first = 0u, index = 0u;
    bool ok = true;
    uint32_t offs_run[16];
#pragma unroll
    for (int L = 1; L <= 15; L++) {
        const uint32_t c = t.cnt[L];
        const uint32_t up = first + c;
        ok = ok && up <= (1u << L);
        if (lane == 0u) {
            upper_out[L - 1] = up << (15 - L);
            delta_out[L] = (uint16_t)(index - first);
        }
        offs_run[L] = index;
        index += c;
        first = up << 1;
    }
    if (!ok) return 0xFFFFFFFFu;
    const uint64_t below = (1ull << lane) - 1ull;
    for (uint32_t base = 0; base < n; base += 64u) {
        const uint32_t s = base + lane;
        const uint32_t l = s < n ? lens[s] : 0u;
#pragma unroll
        for (int L = 1; L <= 15; L++) {
            const uint64_t m = __ballot(l == (uint32_t)L);
            if (l == (uint32_t)L) sym_out[offs_run[L] + (uint32_t)__popcll(m & below)] = (uint16_t)s;
            offs_run[L] += (uint32_t)__popcll(m);
        }
    }
    return index;
}

// ---- the token kernel ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64 * WV_WAVES) bgzf_tokens_kernel(const uint8_t *comp, uint64_t comp_bytes, BgzfBlock *blocks, uint32_t n_blocks,
                                                                  uint8_t *out, uint32_t *seq_arena, const uint64_t *seq_off, uint32_t *seq_count,
                                                                  unsigned long long *dbg = nullptr /* diagnostics: shader cycles per phase, summed over waves */) {
    __shared__ WaveLds lds[WV_WAVES];
    uint64_t tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
    // [0] block set-up + stored blocks, [1] code-length walk of a dynamic header, [2] table builds, [3] A1, [4] A2, [5] path, [6] B + scans, [7] C, [8] chunks, [9] blocks
#define WV_STAMP(k) do { if (dbg) { const uint64_t tn_ = __builtin_amdgcn_s_memtime(); tacc[k] += tn_ - tlast; tlast = tn_; } } while (0)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    WaveLds &t = lds[wave];
    for (uint32_t bi = blockIdx.x * WV_WAVES + wave; bi < n_blocks; bi += gridDim.x * WV_WAVES) {
        const BgzfBlock b = blocks[bi];
        uint32_t status = INF_OK;
        uint32_t n_seq_total = 0u;
        if (b.isize > 65536u || b.in_off > comp_bytes || b.in_len > comp_bytes - b.in_off) status = INF_BAD_BLOCK;
        else if (b.isize) {
            const uint8_t *payload = comp + b.in_off;
            const uint64_t limit = (uint64_t)b.in_len * 8ull;
            uint8_t *dst = out + b.out_off;
            uint32_t *seqs = seq_arena + seq_off[bi];
            const uint64_t cap = seq_cap_of(b.isize);
            uint64_t bitpos = 0;          // wave-uniform: where the true decoder stands
            uint32_t out_pos = 0u;        // bytes produced so far
            for (;;) {                    // deflate blocks
                if (bitpos + 3u > limit) { status = INF_TRUNCATED; break; }
                WvBits hb;   // (uniform: every lane reads the header the same way)
                hb.start(payload, bitpos);
                const uint32_t bfinal = (uint32_t)hb.buf & 1u, btype = ((uint32_t)hb.buf >> 1) & 3u;
                hb.drop(3u);
                bitpos += 3u;
                if (btype == 0u) {
                    // stored: to the byte boundary, LEN / NLEN, LEN raw bytes -- literals, as far as the resolve kernel is concerned
                    bitpos = (bitpos + 7ull) & ~7ull;
                    if (bitpos + 32u > limit) { status = INF_TRUNCATED; break; }
                    const uint64_t v = wv_peek(payload, bitpos);   // (in_len covers it: checked above)
                    const uint32_t len = (uint32_t)v & 0xFFFFu, nlen = ((uint32_t)v >> 16) & 0xFFFFu;
                    bitpos += 32u;
                    if ((len ^ nlen) != 0xFFFFu) { status = INF_BAD_BLOCK; break; }
                    if (len > b.isize - out_pos) { status = INF_OVERRUN; break; }
                    if (bitpos + (uint64_t)len * 8ull > limit) { status = INF_TRUNCATED; break; }
                    const uint8_t *src = payload + (bitpos >> 3);
                    for (uint32_t i = lane; i < len; i += 64u) dst[out_pos + i] = src[i];
                    const uint32_t n = (len + 254u) / 255u;   // copy-less sequences of <= 255 literals
                    if (n_seq_total + n > cap) { status = INF_RETRY; break; }
                    for (uint32_t i = lane; i < n; i += 64u) seqs[n_seq_total + i] = seq_pack(min(255u, len - 255u * i), 0u, 0u);
                    n_seq_total += n;
                    out_pos += len;
                    bitpos += (uint64_t)len * 8ull;
                    WV_STAMP(0);
                    if (bfinal) break;
                    continue;
                }
                if (btype == 3u) { status = INF_BAD_BLOCK; break; }
                // ---- the two codes of this deflate block, built once for the wave ---------------------------------
                uint32_t n_ll = 0u, n_d = 0u;
                if (btype == 1u) {
                    // fixed code (RFC 1951 3.2.6): lengths 8 x144, 9 x112, 7 x24, 8 x8; 30 distance codes of 5 bits
                    for (uint32_t s = lane; s < 320u; s += 64u)
                        t.lens[s] = (uint8_t)(s < 144u ? 8u : s < 256u ? 9u : s < 280u ? 7u : s < 288u ? 8u : s < 318u ? 5u : 0u);
                } else {
                    if (bitpos + 14u > limit) { status = INF_TRUNCATED; break; }
                    const uint32_t hlit = ((uint32_t)hb.buf & 31u) + 257u, hdist = (((uint32_t)hb.buf >> 5) & 31u) + 1u, hclen = (((uint32_t)hb.buf >> 10) & 15u) + 4u;
                    hb.drop(14u);
                    bitpos += 14u;
                    if (hlit > 286u || hdist > 30u) { status = INF_BAD_CODES; break; }
                    // the code-length code: 19 symbols, 3 bits each in the order 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15
                    if (lane < 19u) t.lens[lane] = 0u;
                    if (bitpos + 3ull * hclen > limit) { status = INF_TRUNCATED; break; }
                    if (lane < hclen) {
                        const uint32_t sym = lane < 3u ? 16u + lane : (uint32_t)((0xF1E2D3C4B5A69780ull >> (4u * (lane - 3u))) & 15ull);
                        t.lens[sym] = (uint8_t)((uint32_t)wv_peek(payload, bitpos + 3ull * lane) & 7u);
                    }
                    bitpos += 3ull * hclen;
                    WV_STAMP(0);
                    const uint32_t n_cl = wv_build(t, t.lens, 19u, t.d_upper, t.d_delta, t.d_sym, lane);
                    if (n_cl == 0xFFFFFFFFu) { status = INF_BAD_CODES; break; }
                    uint32_t cu[15];
#pragma unroll
                    for (int k = 0; k < 15; k++) cu[k] = t.d_upper[k];
                    for (uint32_t i = lane; i < (1u << WV_CL_BITS); i += 64u) {
                        uint32_t L;
                        const int sy = wv_code((uint64_t)i, cu, t.d_delta, t.d_sym, n_cl, &L);
                        t.cl_lut[i] = (uint8_t)(sy >= 0 && L <= WV_CL_BITS ? L | ((uint32_t)sy << 3) : 0u);
                    }
                    WV_STAMP(2);
                    // hlit + hdist code lengths, run-length coded in that code: one after the other, every lane the same
                    // (uniform control flow, broadcast LDS reads)
                    uint32_t idx = 0u, prev = 0u;
                    bool bad = false;
                    const uint32_t total = hlit + hdist;
                    hb.start(payload, bitpos);
                    while (idx < total) {
                        if (bitpos > limit) { bad = true; break; }
                        hb.refill();
                        const uint32_t ce = t.cl_lut[(uint32_t)hb.buf & ((1u << WV_CL_BITS) - 1u)];
                        const uint32_t L = ce & 7u, sym = ce >> 3;
                        if (L == 0u) { bad = true; break; }
                        uint32_t len = sym, rep = 1u, used = L;
                        if (sym == 16u) { if (idx == 0u) { bad = true; break; } len = prev; rep = 3u + ((uint32_t)(hb.buf >> used) & 3u); used += 2u; }
                        else if (sym == 17u) { len = 0u; rep = 3u + ((uint32_t)(hb.buf >> used) & 7u); used += 3u; }
                        else if (sym == 18u) { len = 0u; rep = 11u + ((uint32_t)(hb.buf >> used) & 127u); used += 7u; }
                        if (idx + rep > total) { bad = true; break; }
                        for (uint32_t r = lane; r < rep; r += 64u) {
                            const uint32_t at = idx + r;
                            t.lens[at < hlit ? at : 288u + (at - hlit)] = (uint8_t)len;   // (the distance lengths sit behind 288 slots)
                        }
                        prev = len;
                        idx += rep;
                        bitpos += used;
                        hb.drop(used);
                    }
                    WV_STAMP(1);
                    if (bad || bitpos > limit) { status = bad ? INF_BAD_CODES : INF_TRUNCATED; break; }
                    for (uint32_t s = hlit + lane; s < 288u; s += 64u) t.lens[s] = 0u;
                    for (uint32_t s = 288u + hdist + lane; s < 320u; s += 64u) t.lens[s] = 0u;
                }
                n_ll = wv_build(t, t.lens, 288u, t.ll_upper, t.ll_delta, t.ll_sym, lane);
                n_d = wv_build(t, t.lens + 288, 32u, t.d_upper, t.d_delta, t.d_sym, lane);
                if (n_ll == 0xFFFFFFFFu || n_d == 0xFFFFFFFFu) { status = INF_BAD_CODES; break; }
                uint32_t lu[15], du[15];
#pragma unroll
                for (int k = 0; k < 15; k++) { lu[k] = t.ll_upper[k]; du[k] = t.d_upper[k]; }
                wv_fill_luts(t, lu, du, n_ll, n_d, lane);
                WV_STAMP(2);

                // ---- the compressed data, a chunk of 64 segments at a time ---------------------------------------
                bool eob_seen = false;
                while (!eob_seen && status == INF_OK) {
                    if (bitpos >= limit) { status = INF_TRUNCATED; break; }
                    const uint64_t chunk0 = bitpos;
                    const uint64_t remain = limit - chunk0;
                    const uint32_t S = 32u * ((uint32_t)min<uint64_t>(WV_SEG_MAX / 32u, max<uint64_t>(WV_SEG_MIN / 32u, (remain + 2047ull) / 2048ull)) | 1u);
                    const uint64_t chunk_end = min<uint64_t>(limit, chunk0 + 64ull * S);
                    const uint64_t seg0 = chunk0 + (uint64_t)lane * S, seg1 = min<uint64_t>(chunk_end, seg0 + S);
                    const bool active = seg0 < chunk_end;
                    for (uint32_t w = lane; w < 64u * S / 32u; w += 64u) t.map[w] = 0u;
                    // -- walk A1: the lane's own segment, from its first bit (lane 0: a true token start); token starts recorded
                    uint32_t meet = 0u, into = 64u, end_kind = active ? 0u : 2u;   // relative to chunk0; see the path walk below
                    uint32_t pos = (uint32_t)(seg0 - chunk0);                      // (positions inside a chunk fit 17 bits)
                    const uint32_t seg_end = (uint32_t)(seg1 - chunk0), c_end = (uint32_t)(chunk_end - chunk0);
                    WvBits br;
                    br.start(payload, chunk0 + (active ? pos : 0u));
                    {
                        bool go = active;
                        while (__any(go)) {
                            if (go) {
                                atomicOr(&t.map[pos >> 5], 1u << (pos & 31u));
                                const WvTok k = wv_token(br, t, lu, du, n_ll, n_d);
                                if (k.kind == TK_BAD) { end_kind = 2u; meet = pos; go = false; }
                                else if (k.kind == TK_EOB) { end_kind = 1u; meet = pos; go = false; }
                                else {
                                    pos += k.nbits;
                                    if (pos >= seg_end) go = false;
                                }
                            }
                        }
                    }
                    WV_STAMP(3);
                    // -- walk A2: on into the following segments until a token start their owners recorded (or out of the chunk, or
                    //    two segments without meeting anybody: the chunk is then cut where this lane stands)
                    if (end_kind == 0u) {
                        bool go = true;
                        const uint32_t give_up = seg_end + 2u * S;
                        while (__any(go)) {
                            if (go) {
                                if (pos >= c_end) { meet = pos; into = 64u; go = false; }
                                else if ((t.map[pos >> 5] >> (pos & 31u)) & 1u) { meet = pos; into = pos / S; go = false; }
                                else if (pos >= give_up) { meet = pos; into = 65u; go = false; }   // 65: nobody met
                                else {
                                    const WvTok k = wv_token(br, t, lu, du, n_ll, n_d);
                                    if (k.kind == TK_BAD) { end_kind = 2u; meet = pos; go = false; }
                                    else if (k.kind == TK_EOB) { end_kind = 1u; meet = pos; go = false; }
                                    else pos += k.nbits;
                                }
                            }
                        }
                    }
                    WV_STAMP(4);
                    // -- the true sequence: lane 0's chain to where it meets lane j's, lane j's from there on, ...
                    bool on_path = false;
                    uint32_t from = 0u, to = 0u, next_rel = c_end;
                    {
                        uint32_t c = 0u, f = 0u;
                        for (uint32_t hop = 0; hop < 66u; hop++) {
                            const uint32_t m = (uint32_t)__shfl((int)meet, (int)c), it = (uint32_t)__shfl((int)into, (int)c), kind = (uint32_t)__shfl((int)end_kind, (int)c);
                            if (lane == c) { on_path = true; from = f; to = m; }
                            if (kind == 2u) { status = INF_BAD_SYMBOL; break; }
                            if (kind == 1u) { eob_seen = true; next_rel = m; break; }   // (m = the end-of-block symbol's first bit)
                            if (it >= 64u) { next_rel = m; break; }                      // left the chunk, or gave up: the next chunk starts here
                            f = m;
                            c = it;
                        }
                    }
                    if (status != INF_OK) break;
                    WV_STAMP(5);
                    if (dbg) tacc[8]++;
                    // -- walk B: what the lane's part of the true sequence produces
                    uint32_t n_out = 0u, n_sq = 0u;
                    {
                        uint32_t lit = 0u, p = from;
                        bool go = on_path && p < to;
                        br.start(payload, chunk0 + (go ? from : 0u));
                        while (__any(go)) {
                            if (go) {
                                const WvTok k = wv_token(br, t, lu, du, n_ll, n_d);
                                if (k.kind == TK_LIT) {
                                    lit++;
                                    n_out++;
                                    if (lit == 255u) { n_sq++; lit = 0u; }
                                } else {   // a match (end-of-block and bad codes end a chain: they are never inside [from, to))
                                    const uint32_t piece = seq_piece_of(k.dist);
                                    n_sq += (k.len + piece - 1u) / piece;
                                    lit = 0u;
                                    n_out += k.len;
                                }
                                p += k.nbits;
                                if (p >= to) go = false;
                            }
                        }
                        if (lit) n_sq++;
                    }
                    // exclusive prefix sums over the lanes (in path order = lane order: a later lane's part lies further on)
                    uint32_t o_inc = n_out, s_inc = n_sq;
                    for (int d = 1; d < 64; d <<= 1) {
                        const uint32_t a = (uint32_t)__shfl_up((int)o_inc, d), c2 = (uint32_t)__shfl_up((int)s_inc, d);
                        if ((int)lane >= d) { o_inc += a; s_inc += c2; }
                    }
                    const uint32_t o_tot = (uint32_t)__shfl((int)o_inc, 63), s_tot = (uint32_t)__shfl((int)s_inc, 63);
                    if (o_tot > b.isize - out_pos) { status = INF_OVERRUN; break; }
                    if ((uint64_t)n_seq_total + s_tot > cap) { status = INF_RETRY; break; }
                    WV_STAMP(6);
                    // -- walk C: emit.  Literals are collected eight to a store.
                    {
                        uint32_t o = out_pos + o_inc - n_out, sq = n_seq_total + s_inc - n_sq, lit = 0u, p = from;
                        bool go = on_path && p < to, bad_dist = false;
                        uint64_t acc = 0;      // literals not stored yet ...
                        uint32_t n_acc = 0u;   // ... their number, and `o` is where the NEXT output byte goes
                        br.start(payload, chunk0 + (go ? from : 0u));
                        while (__any(go)) {
                            if (go) {
                                const WvTok k = wv_token(br, t, lu, du, n_ll, n_d);
                                if (k.kind == TK_LIT) {
                                    acc |= (uint64_t)k.val << (8u * n_acc);
                                    o++;
                                    if (++n_acc == 8u) { store_u64(dst + o - 8u, acc); acc = 0; n_acc = 0u; }
                                    if (++lit == 255u) { seqs[sq++] = seq_pack(255u, 0u, 0u); lit = 0u; }
                                } else {
                                    if (n_acc) { store_tail(dst + o - n_acc, acc, n_acc); acc = 0; n_acc = 0u; }
                                    if (k.dist > o) bad_dist = true;
                                    const uint32_t piece = seq_piece_of(k.dist);
                                    for (uint32_t done = 0u; done < k.len; done += piece) {
                                        seqs[sq++] = seq_pack(lit, min(piece, k.len - done), k.dist);
                                        lit = 0u;
                                    }
                                    o += k.len;
                                }
                                p += k.nbits;
                                if (p >= to) go = false;
                            }
                        }
                        if (n_acc) store_tail(dst + o - n_acc, acc, n_acc);
                        if (lit) seqs[sq++] = seq_pack(lit, 0u, 0u);
                        if (__any(bad_dist)) { status = INF_BAD_DISTANCE; break; }
                    }
                    WV_STAMP(7);
                    out_pos += o_tot;
                    n_seq_total += s_tot;
                    bitpos = chunk0 + next_rel;
                    if (eob_seen) {   // step over the end-of-block symbol (its bits are in the stage)
                        br.start(payload, chunk0 + next_rel);
                        const WvTok k = wv_token(br, t, lu, du, n_ll, n_d);
                        bitpos += k.nbits;
                    }
                }
                if (status != INF_OK) break;
                if (bitpos > limit) { status = INF_TRUNCATED; break; }
                if (bfinal) break;
            }
            if (status == INF_OK && out_pos != b.isize) status = INF_SHORT;
        }
        if (lane == 0u) {
            blocks[bi].status = status;
            seq_count[bi] = status == INF_OK ? n_seq_total : 0u;
        }
        WV_STAMP(0);
        if (dbg) tacc[9]++;
    }
    if (dbg && lane == 0u)
        for (int k = 0; k < 10; k++) atomicAdd(&dbg[k], (unsigned long long)tacc[k]);
#undef WV_STAMP
}

// ---- the resolve kernel --------------------------------------------------------------------------------------------
// LZ77 only, one wave per block, IN PLACE in the output buffer: 64 sequences at a time are placed by prefix sums and their
// copies carried out in ROUNDS -- a copy runs once every byte of its source is final, i.e. lies in front of the first copy
// of the batch that is still to do (a record's matches copy from the record before: the rounds of a block are as many as
// its records, each a store -> load round trip through L2).  What hides that latency is the number of blocks in flight:
// no LDS, few registers, sixteen waves per CU.  (A version that kept the block in a 64 KiB LDS window -- two waves per CU,
// every LDS latency exposed -- took 230 us per block; profiles/r03_inflate_wave.txt.)  A wave's own stores are visible to
// its later loads once they have completed (s_waitcnt vmcnt(0)): the CU's L1 is coherent for the CU's own accesses.
constexpr uint32_t RS_WAVES = 4;
__global__ void __launch_bounds__(64 * RS_WAVES) bgzf_resolve_kernel(uint8_t *out, BgzfBlock *blocks, uint32_t n_blocks, const uint32_t *seq_arena,
                                                                   const uint64_t *seq_off, const uint32_t *seq_count, unsigned long long *dbg = nullptr) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint64_t tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = dbg ? __builtin_amdgcn_s_memtime() : 0ull;   // [1] scans, [2] rounds, [4] batches, [5] rounds, [6] blocks
#define RS_STAMP(k) do { if (dbg) { const uint64_t tn_ = __builtin_amdgcn_s_memtime(); tacc[k] += tn_ - tlast; tlast = tn_; } } while (0)
    for (uint32_t bi = blockIdx.x * RS_WAVES + wave; bi < n_blocks; bi += gridDim.x * RS_WAVES) {
        const BgzfBlock b = blocks[bi];
        if (b.status != INF_OK || !b.isize) continue;
        uint8_t *g = out + b.out_off;
        const uint32_t n = b.isize;
        const uint32_t *seqs = seq_arena + seq_off[bi];
        const uint32_t n_seq = seq_count[bi];
        uint32_t s_next = lane < n_seq ? seqs[lane] : 0u;
        uint32_t base = 0u;     // block offset where the next sequence's output starts
        bool bad = false;
        RS_STAMP(0);
        for (uint32_t s0 = 0; s0 < n_seq; s0 += 64u) {
            if (dbg) tacc[4]++;
            const uint32_t s = s_next;
            s_next = s0 + 64u + lane < n_seq ? seqs[s0 + 64u + lane] : 0u;   // the next batch, a batch ahead
            const uint32_t lit = s & 0xFFu, has = s >> 31, mlen = has ? ((s >> 8) & 0xFFu) + 1u : 0u, dist = ((s >> 16) & 0x7FFFu) + 1u;
            uint32_t inc = lit + mlen;
            const uint32_t mine = inc;
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)inc, d);
                if ((int)lane >= d) inc += up;
            }
            const uint32_t mdst = base + inc - mine + lit;   // where this lane's copy goes
            const uint32_t total = (uint32_t)__shfl((int)inc, 63);
            if (base + total > n) { bad = true; break; }
            bool todo = has != 0u;
            if (__any(todo && dist > mdst)) { bad = true; break; }   // a source in front of the block
            const uint32_t src = mdst - dist;
            RS_STAMP(1);
            while (true) {
                const uint64_t left = __ballot(todo);
                if (!left) break;
                if (dbg) tacc[5]++;
                const int first = __ffsll((long long)left) - 1;
                const uint32_t frontier = (uint32_t)__shfl((int)mdst, first);
                if (todo && ((int)lane == first || min(src + mlen, mdst) <= frontier)) {
                    if (dist >= WV_RUN_DIST) {
                        // a piece: <= 32 bytes, no longer than its distance (the token kernel cut it so): both loads before any store
                        const uint4 p0 = load_u128(g + src), p1 = load_u128(g + src + 16);   // (may read up to 31 bytes past the piece: never used)
                        if (mlen > 16u) {
                            store_u128(g + mdst, p0);
                            store_tail16(g + mdst + 16, (uint64_t)p1.x | ((uint64_t)p1.y << 32), (uint64_t)p1.z | ((uint64_t)p1.w << 32), mlen - 16u);
                        } else store_tail16(g + mdst, (uint64_t)p0.x | ((uint64_t)p0.y << 32), (uint64_t)p0.z | ((uint64_t)p0.w << 32), mlen);
                    } else if (mdst >= 8u) {
                        // a run: expanded from the period that ends in front of it
                        store_run(g + mdst, load_u64(g + mdst - 8) >> (8u * (8u - dist)), dist, mlen);
                    } else {
                        for (uint32_t k = 0; k < mlen; k++) g[mdst + k] = g[src + k];   // (a run in the block's first bytes)
                    }
                    todo = false;
                }
                // the round's stores have completed before the next round's loads are issued
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            base += total;
            RS_STAMP(2);
        }
        if (bad || base != n) {
            if (lane == 0u) blocks[bi].status = bad ? INF_BAD_DISTANCE : INF_SHORT;
        }
        if (dbg) tacc[6]++;
    }
    if (dbg && lane == 0u)
        for (int k = 0; k < 8; k++) atomicAdd(&dbg[10 + k], (unsigned long long)tacc[k]);
#undef RS_STAMP
}

}  // namespace pssbam
