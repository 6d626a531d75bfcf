// pss-bam_amd/csrc/tally_kernels.h -- the gfx950 tally kernels.
//
//  tally_simple : lane-per-read, records and reference bases gathered straight from
//                 global memory, counts into an LDS table (or global atomics when the
//                 table would not fit).  Any -r N, any k.  Cross-check of tally_tiled.
//  tally_tiled  : the production kernel.  One launch tallies 32 table rows (N <= 30: all of
//                 them; a larger -r takes one launch per 32 rows).  Persistent workgroups walk tiles
//                 of T = 128 consecutive reads:
//                   1. STAGE   LDS-DMA (global_load_lds_dwordx4) with a PER-LANE source address:
//                              lane q of the tile's piece list fetches 16-byte piece q % P of
//                              record q / P, and the hardware packs the pieces of one
//                              wave-instruction contiguously in LDS -- so record j's first
//                              P*16 bytes (header, name, CIGAR, SEQ, QUAL[0]: everything the
//                              path reads) land densely at stage + j*P*16.  The ~150 QUAL bytes
//                              of a 150-bp record are never requested: the staging buffer is
//                              half of a whole-record tile (4 workgroups per CU instead of 3)
//                              and half the DMA issue slots are saved -- HBM itself still moves
//                              whole 128-byte lines, so its traffic stays close to the record
//                              size -- and consecutive lanes read consecutive addresses (cheap
//                              for the texture addresser).  The next tile's DMA is issued as
//                              soon as CODES-A is done with the buffer.
//                   2. CODES   a lane pair per read (one lane per alignment end).  Part A:
//                              decode + filters from LDS, the end's reference window
//                              (32 bytes) and k-mer window gathered from the device genome,
//                              SEQ nibbles into registers.  Part B: one byte per window
//                              position = (cell << 1 | table) or a "no count" code, four
//                              positions per VALU instruction through v_perm_b32 used as a
//                              byte table (no memory lookups), written as the read's row of
//                              the code sheet.
//                   3. COLUMNS wave-per-read, lane = column of the code sheet: each lane
//                              owns one (end, position) and bumps ITS word of a
//                              [cell,table][row] LDS table.  Lanes of one wave-instruction
//                              never share a word and the two 32-lane halves hit disjoint
//                              banks, so the AA/CC/GG/TT skew of real data causes no
//                              serialisation at all.
//                 Counters leave LDS once, at kernel end, as plain stores into the workgroup's
//                 slot of a scratch buffer; reduce_partials sums the slots into the u64 block.
//
// Integer/byte work only: no MFMA anywhere (the bound is HBM bandwidth).
#pragma once

#include "record_decode.h"

namespace pssbam {

constexpr int TILED_THREADS = 256;
constexpr int TILED_WAVES = TILED_THREADS / 64;
constexpr int TILED_ROWS = 32;         // table rows (window positions per end) one pass covers; a larger
                                       // N+2 takes several passes over the block, 32 rows each (row_base)
constexpr int KMER_LDS_MAX_K = 4;      // 2 * 4^4 * 4 B = 2 KiB of LDS
constexpr uint32_t STAGE_SLACK = 64;   // readable bytes behind a staging buffer
constexpr uint32_t CODE_NONE = 32;     // sheet byte meaning "no count"; every code >= 32 is one
constexpr uint32_t TABLE_WORDS = 64 * 32;  // codes 0..31 = (cell << 1) | table, 32..63 = trash bin
// per-workgroup partial results of tally_tiled in global scratch: [table 1024 | k-mer bins 512 | stat deltas 16]
constexpr uint32_t SCRATCH_KMER = 1024, SCRATCH_DELTA = 1536, SCRATCH_WORDS = 1552;
constexpr uint32_t REF_LDS_ENTRIES = 64;   // BAM references whose contig info is cached in LDS (+1 for "*")

// ---------------------------------------------------------------------------------------
// per-read tally, lane-per-read form (tally_simple, and tile-overflow records)
// ---------------------------------------------------------------------------------------
struct LdsTableRowMajor {  // [table][row][16] u32 in LDS
    uint32_t *t;
    uint32_t rows;
    __device__ __forceinline__ void add(uint32_t table, uint32_t row, uint32_t cell) const {
        atomicAdd(&t[(table * rows + row) * 16u + cell], 1u);
    }
};
struct LdsTableColumnMajor {  // [(cell << 1) | table][32 rows] u32 in LDS (tiled kernel): rows row_base .. +31
    uint32_t *t;
    uint32_t row_base;
    __device__ __forceinline__ void add(uint32_t table, uint32_t row, uint32_t cell) const {
        const uint32_t r = row - row_base;
        if (r < 32u) atomicAdd(&t[(((cell << 1) | table) << 5) + r], 1u);
    }
};
struct GlobalTable {  // straight into the u64 counter block
    unsigned long long *c;
    uint32_t off_rev;
    __device__ __forceinline__ void add(uint32_t table, uint32_t row, uint32_t cell) const {
        atomicAdd(&c[(table ? off_rev : 0u) + row * 16u + cell], 1ull);
    }
};

// One end of one read into one table.  `left` selects the alignment's left end
// (reference s-2.., read bases 0..) or right end (reference ..s+L+1, read bases ..L-1);
// `comp` complements both bases (reverse-strand reads), which maps cell c to 15-c.
// Restates add_ctx_counts + add_fwd_counts / add_rev_counts, pss-bam.c:169-326.
template <class Src, class Tab>
__device__ void tally_end(const Tab &tab, uint32_t table, const Src &src, const RecHdr &h, const uint8_t *G,
                          int64_t s, uint32_t L, int N, bool left, bool comp) {
    const uint32_t c0 = ref_code(left ? G[s - 2] : G[s + L + 1]);  // second context base -> row 0
    const uint32_t c1 = ref_code(left ? G[s - 1] : G[s + L]);      // first context base  -> row 1
    if (c0 < 4u) tab.add(table, 0, comp ? 15u - 5u * c0 : 5u * c0);
    if (c1 < 4u) tab.add(table, 1, comp ? 15u - 5u * c1 : 5u * c1);
    for (int i = 0; i < N; i++) {
        const uint32_t ri = left ? (uint32_t)i : L - 1u - (uint32_t)i;
        const uint32_t rd = nib_code(read_nibble(src, h, ri));
        const uint32_t rf = ref_code(G[s + (int64_t)ri]);
        if (rd < 4u && rf < 4u) {
            const uint32_t cell = 4u * rd + rf;
            tab.add(table, (uint32_t)i + 2u, comp ? 15u - cell : cell);
        }
    }
}

template <class Src, class Tab>
__device__ __forceinline__ void tally_pss_record(const TallyParams &P, const Tab &tab, const Src &src,
                                                 const RecHdr &h, const Plan &pl) {
    const uint8_t *G = P.genome + pl.gbase;
    // forward-strand read: fwd table <- left end, rev table <- right end;
    // reverse-strand read: fwd table <- right end complemented, rev table <- left end complemented
    if (pl.pss_fwd) tally_end(tab, 0u, src, h, G, pl.s, pl.L, P.N, !pl.rev, pl.rev);
    if (pl.pss_rev) tally_end(tab, 1u, src, h, G, pl.s, pl.L, P.N, pl.rev, pl.rev);
}

// one k-mer add (5' when which == 0, 3' when which == 1); false = non-ACGT in the window
template <bool LDS_KMER>
__device__ __forceinline__ bool tally_one_kmer(const TallyParams &P, const Plan &pl, uint32_t which, uint32_t *lds_kmer) {
    const uint8_t *G = P.genome + pl.gbase;
    int64_t w5, w3;
    kmer_windows(pl, P.K, w5, w3);
    uint32_t bin;
    if (!kmer_bin(G, which ? w3 : w5, P.K, pl.rev, bin)) return false;
    if (LDS_KMER) atomicAdd(&lds_kmer[(which ? (1u << (2 * P.K)) : 0u) + bin], 1u);
    else atomicAdd(&P.counters[(which ? P.off_k3 : P.off_k5) + bin], 1ull);
    return true;
}

// both k-mer adds of one record; true = an attempted add failed (fragkon's status -1)
// (fragkon.c:164-181: both attempted, 0 only if both succeeded; :198-210 single add)
template <bool LDS_KMER>
__device__ __forceinline__ bool tally_kmer_record(const TallyParams &P, const Plan &pl, uint32_t *lds_kmer) {
    bool good = true;
    if (pl.fk5) good = tally_one_kmer<LDS_KMER>(P, pl, 0u, lds_kmer) && good;
    if (pl.fk3) good = tally_one_kmer<LDS_KMER>(P, pl, 1u, lds_kmer) && good;
    return !good;
}

// ---------------------------------------------------------------------------------------
// tally_simple
// ---------------------------------------------------------------------------------------
// dynamic LDS: [2*(N+2)*16 u32 table, if LDS_TABLE]
template <bool LDS_TABLE>
__global__ void __launch_bounds__(256) tally_simple(const TallyParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_lds[];
    __shared__ int32_t lds_delta[ST_USED];
    const uint32_t rows = (uint32_t)P.N + 2u;
    const uint32_t tab_words = LDS_TABLE ? 2u * rows * 16u : 0u;
    const bool do_pss = (P.tally_mask & 1u) != 0, do_kmer = (P.tally_mask & 2u) != 0;
    for (uint32_t i = threadIdx.x; i < tab_words; i += blockDim.x) dyn_lds[i] = 0u;
    if (threadIdx.x < ST_USED) lds_delta[threadIdx.x] = 0;
    __syncthreads();

    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < P.n_recs; r += stride) {
        const uint32_t o0 = P.offs[r], o1 = P.offs[r + 1];
        GlobalBytes src{P.recs + o0};
        const RecHdr h = decode_hdr(src, o1 - o0);
        Plan pl = make_plan<true, true>(P, src, h);
        if (!do_pss) pl.pss_fwd = pl.pss_rev = false;
        if (!do_kmer) pl.fk5 = pl.fk3 = false;
        if (pl.pss_fwd || pl.pss_rev) {
            if (LDS_TABLE) tally_pss_record(P, LdsTableRowMajor{dyn_lds, rows}, src, h, pl);
            else tally_pss_record(P, GlobalTable{P.counters, P.off_rev}, src, h, pl);
        }
        bool kfail = false;
        if (pl.fk5 || pl.fk3) kfail = tally_kmer_record<false>(P, pl, nullptr);
        book_events(do_pss, do_kmer, record_events(do_pss, do_kmer, pl, kfail), lds_delta);
    }
    __syncthreads();
    if (LDS_TABLE) {
        for (uint32_t i = threadIdx.x; i < tab_words; i += blockDim.x) {
            const uint32_t v = dyn_lds[i];
            if (v) {
                const uint32_t table = i / (rows * 16u), rest = i % (rows * 16u);
                atomicAdd(&P.counters[(table ? P.off_rev : 0u) + rest], (unsigned long long)v);
            }
        }
    }
    flush_events(do_pss, do_kmer, P, lds_delta);
}

// ---------------------------------------------------------------------------------------
// tally_tiled
// ---------------------------------------------------------------------------------------
// LDS objects.  Only the staging buffer is dynamic (extern) LDS; everything else is a
// separate static object:
//   stage  (dynamic) : T * P * 16 + STAGE_SLACK          first P pieces of every record of the tile
//   sheet  : TILED_MAX_T * 64                             code sheet [read][end*32 + position]
//   table  : 64 * 32 * 4                                  [(cell<<1)|table][row] u32; codes 32..63 = trash bin
//                                                         for "no count" codes, so the column pass has no branches
//   toffs  : 2 * (TILED_MAX_T + 4) * 4                    record offsets of this tile and the next
//   kmer   : 2 * 4^KMER_LDS_MAX_K * 4                     (LDS_KMER variants only)
//   refs   : (REF_LDS_ENTRIES + 1) * 16                   contig info of the first BAM references
__host__ __device__ inline uint32_t tiled_lds_bytes(uint32_t T, uint32_t pieces) { return T * pieces * 16u + STAGE_SLACK; }

// Four dwords at 4-byte alignment: gfx950 global loads only need dword alignment, so this
// compiles to ONE global_load_dwordx4 per lane.  A gather's cost in the texture addresser is per
// wave-instruction and per distinct line touched -- nine single-dword gathers of a 36-byte
// window cost three times what 2 x dwordx4 + 1 x dword do.
struct __attribute__((packed, aligned(4))) Quad { uint32_t v[4]; };
struct __attribute__((packed, aligned(4))) Tri { uint32_t v[3]; };

constexpr uint32_t TILED_MAX_T = 128;
static_assert(TILED_MAX_T * 2 == TILED_THREADS, "CODES maps one (read, end) pair to each thread");

// STAGE: piece q of the tile (q = record * P + piece) goes to stage + q*16.  Each wave-instruction
// moves 64 consecutive pieces (1 KiB of LDS); a lane's source is its record's 16-byte aligned
// start + 16 * piece.  Pieces that would start beyond the record block are not issued.
//
// The LDS-DMA instruction is issued through inline asm on purpose.  hipcc's waitcnt pass
// fences EVERY later LDS access behind vmcnt(0) once it has seen an LDS-DMA it cannot
// disambiguate (no alias-scope metadata reaches it from HIP source), which would serialise
// the transfer against the passes it is meant to hide behind.  The asm form is invisible to
// that pass; ordering is ours: the kernel waits with an explicit `s_waitcnt vmcnt(0)` + barrier
// before any lane reads `stage`, and nothing else writes `stage`.  (Compiler-counted vmcnt(N)
// waits for its own loads only get stricter with unseen operations in flight, never weaker:
// vmcnt retires in order.)
__device__ __forceinline__ void stage_tile_dma(const uint8_t *recs, uint64_t recs_limit, const uint32_t *tile_offs,
                                               uint32_t count, uint32_t pieces, uint8_t *stage, uint32_t tid) {
    const uint32_t n_pieces = count * pieces;
    const uint32_t lds0 = (uint32_t)(uintptr_t)stage;  // LDS byte address (low half of the generic pointer)
    // q / pieces by multiply-shift: exact for q < 2^13 and pieces <= 64 ((pieces-1) * q < 2^20)
    const uint32_t magic = ((1u << 20) + pieces - 1u) / pieces;
    for (uint32_t q0 = (tid & ~63u); q0 < n_pieces; q0 += TILED_THREADS) {
        const uint32_t q = q0 + (tid & 63u);
        const uint32_t jq = (q * magic) >> 20;
        const uint32_t j = min(jq, count - 1u), pc = q - jq * pieces;
        const uint64_t a = (uint64_t)(tile_offs[j] & ~15u) + 16u * pc;
        if (q < n_pieces && a + 16u <= recs_limit) {
            const uint8_t *src = recs + a;
            const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds0 + (q0 << 4));
            uint32_t keep;   // m0 is compiler-reserved and cannot be named as a clobber: save and restore it
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(m0v) : "memory");
        }
    }
}

// A record too large for the staging window: decoded and tallied straight from global memory
// by one lane.  Rare (a record of tens of KB); kept out of line so it costs the hot path
// nothing.  It reads the kernel arguments through a pointer to the kernarg segment (taken in
// the kernel): a reference to the kernel's by-value copy would force that whole struct into
// scratch memory.
template <bool DO_PSS, bool DO_KMER, bool LDS_KMER>
__device__ __attribute__((noinline)) uint32_t tally_overflow_record(const TallyParams *kernarg, uint32_t o0,
                                                                    uint32_t o1, uint32_t *table, uint32_t *lds_kmer) {
    const TallyParams &P = *kernarg;
    GlobalBytes gsrc{P.recs + o0};
    const RecHdr gh = decode_hdr(gsrc, o1 - o0);
    const Plan gpl = make_plan<DO_PSS, DO_KMER>(P, gsrc, gh);
    if (DO_PSS && (gpl.pss_fwd || gpl.pss_rev)) tally_pss_record(P, LdsTableColumnMajor{table, P.row_base}, gsrc, gh, gpl);
    bool kfail = false;
    if (DO_KMER && (gpl.fk5 || gpl.fk3)) kfail = tally_kmer_record<LDS_KMER>(P, gpl, lds_kmer);
    return record_events(DO_PSS, DO_KMER, gpl, kfail);
}

// Reference windows, one per alignment end, each with STATIC byte positions:
//   left  end (e = 0): 32 bytes from s-2      byte w <-> row w          (0,1 context; 2+i = position i)
//   right end (e = 1): 32 bytes up to s+L+1   byte w <-> row 31-w       (31 -> row 0, 30 -> row 1, 29-i -> 2+i)
// Read bases: left row 2+i <-> base i ; right row 2+i <-> base L-1-i, i.e. window byte w <-> base (L-30)+w.
// The kernel body takes its LDS regions as __restrict__ pointers: after inlining, every LDS
// access carries alias-scope metadata, which is what lets the compiler see that the code sheet,
// the count table and the offset buffer never alias the LDS-DMA destination (`stage`) -- without
// it every LDS access issued while a DMA transfer is in flight is fenced behind vmcnt(0) and
// the transfer cannot overlap the COLUMNS pass.
template <bool DO_PSS, bool DO_KMER, bool LDS_KMER, bool LATER_PASS>
__device__ __forceinline__ void tally_tiled_body(const TallyParams &P, const TallyParams *kernarg,
                                                 uint8_t *__restrict__ stage, uint8_t *__restrict__ sheet,
                                                 uint32_t *__restrict__ table,
                                                 uint32_t *__restrict__ toffs,
                                                 uint32_t *__restrict__ lds_kmer,
                                                 int32_t *__restrict__ lds_delta, uint4 *__restrict__ refs_lds) {
    const uint32_t T = P.reads_per_tile;   // <= TILED_MAX_T
    const uint32_t n_recs = P.n_recs_dev ? *P.n_recs_dev : P.n_recs;   // device-indexed blocks: the count lives in device memory
    const uint32_t pieces = P.prefix_pieces;  // 16-byte pieces staged per record
    const uint64_t recs_limit = (P.recs_bytes + 15ull) & ~15ull;  // the block is readable up to here
    const uint32_t ablate = P.ablate;      // diagnostics only (PSSBAM_ABLATE): 1 no COLUMNS, 2 no position loop, 4 no CODES, 128 no window gathers

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const int N = P.N;
    const uint32_t n_pos = (uint32_t)N + 2u;  // rows per table: 2 context + N positions
    // this launch tallies rows row_base .. row_base+31 (window positions shifted accordingly);
    // pass 0 also owns the status counters and the k-mer tally
    // (a compile-time 0 in the first-pass instantiation: the common N <= 30 case pays nothing)
    const uint32_t row_base = LATER_PASS ? P.row_base : 0u;
    const bool pass0 = !LATER_PASS;
    const uint32_t n_live = n_pos > row_base ? min(n_pos - row_base, 32u) : 0u;

    // ---- one-time set-up: zero the tables ---------------------------------------------------------
    for (uint32_t i = tid; i < TABLE_WORDS; i += TILED_THREADS) table[i] = 0u;
    if (LDS_KMER)
        for (uint32_t i = tid; i < 2u * (1u << (2 * P.K)); i += TILED_THREADS) lds_kmer[i] = 0u;
    if (tid < ST_USED) lds_delta[tid] = 0;
    // contig info of the first BAM references (all of them for a human-sized header) + the "*" entry
    const uint32_t n_ref_cached = min((uint32_t)P.n_ref, REF_LDS_ENTRIES);
    if (tid < n_ref_cached) refs_lds[tid] = P.ref_info[tid];
    if (tid == n_ref_cached) refs_lds[tid] = P.ref_info[P.n_ref];
    const uint32_t all_tiles = (n_recs + T - 1u) / T;
    // Workgroup -> tiles.  Workgroups are dealt to the 8 XCDs round-robin (blockIdx & 7); with
    // xcd_map every XCD walks its own contiguous eighth of the block, so neighbouring tiles -- which
    // share reference lines and the record line at their seam -- meet in the same L2.
    uint32_t tile0 = blockIdx.x, tstride = gridDim.x, n_tiles = all_tiles;
    if (P.xcd_map && (gridDim.x & 7u) == 0u && all_tiles >= 64u) {
        const uint32_t per = (all_tiles + 7u) >> 3, xcd = blockIdx.x & 7u;
        tile0 = xcd * per + (blockIdx.x >> 3);
        tstride = gridDim.x >> 3;
        n_tiles = min(all_tiles, (xcd + 1u) * per);
    }
    // software pipeline over this workgroup's tiles k0, k0+stride, ...:
    //   toffs[par]      offsets of the tile being processed, toffs[par^1] those of the next one
    //                   (written from VGPRs that were loaded one tile earlier)
    //   stage           pieces of the tile being processed; refilled for the next tile as soon as
    //                   CODES-A has read everything it needs
    uint32_t tile = tile0;
    uint32_t off_a = 0;
    const uint32_t TOFF = TILED_MAX_T + 4u;  // stride between the two offset buffers
    auto load_offsets = [&](uint32_t t) {
        const uint32_t r0 = t * T;
        if (tid <= T && r0 + tid <= n_recs) off_a = P.offs[r0 + tid];
    };
    auto tile_count = [&](uint32_t t) { return min(T, n_recs - t * T); };
    __syncthreads();  // LDS tables are set up
    if (tile < n_tiles) {
        load_offsets(tile);
        if (tid <= T) toffs[tid] = off_a;
        __syncthreads();
        stage_tile_dma(P.recs, recs_limit, toffs, tile_count(tile), pieces, stage, tid);
        if (tile + tstride < n_tiles) load_offsets(tile + tstride);
    }

    for (uint32_t it = 0; tile < n_tiles; tile += tstride, it++) {
        const uint32_t par = it & 1u;
        const uint32_t *cur_offs = toffs + par * TOFF;
        const uint32_t r0 = tile * T;
        const uint32_t count = min(T, n_recs - r0);
        const uint32_t next = tile + tstride;

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // DMA pieces + offset loads of this wave are in
        if (next < n_tiles && tid <= T) toffs[(par ^ 1u) * TOFF + tid] = off_a;  // next tile's offsets
        __syncthreads();  // everyone's DMA landed; previous COLUMNS pass is over

        // ---- CODES, part A: everything that reads `stage` ---------------------------------------
        // A lane pair per read (TILED_MAX_T * 2 == TILED_THREADS: one (read, end) per thread).
        // e = 0: left alignment end, 1: right end.
        const uint32_t j = tid >> 1, e = tid & 1u;
        const bool lane_on = tid < 2u * T && !(ablate & 4u);
        const bool in_tile = lane_on && j < count;
        uint32_t o0 = 0, o1 = 0;
        if (in_tile) { o0 = cur_offs[j]; o1 = cur_offs[j + 1]; }
        // record j's first `pieces` 16-byte pieces sit at stage + j*pieces*16, starting at its
        // 16-byte aligned address: byte x of the record is at offset (o0 & 15) + x
        const uint32_t avail = pieces * 16u - (o0 & 15u);       // record bytes present in LDS
        const bool hdr_ok = in_tile && o1 - o0 >= 36u && avail >= 48u;
        // lanes without a usable record decode a harmless dummy (offset 0, length 0 -> malformed ->
        // dead) so the lanes of a wave stay on one path
        LdsBytes src{stage, hdr_ok ? j * pieces * 16u + (o0 & 15u) : 0u};
        const RecHdr h = decode_hdr_lds(src, hdr_ok ? o1 - o0 : 0u);
        // everything the path reads ends at QUAL[0] (the -R filter walks the aux fields: whole record)
        const uint32_t needed = P.rg ? o1 - o0 : h.qual_off + 1u;
        const bool in_stage = hdr_ok && needed <= avail;
        Plan pl = plan_head<DO_PSS, DO_KMER>(P, src, h, RefsLdsCached{refs_lds, P.ref_info, n_ref_cached, (uint32_t)P.n_ref});
        if (!in_stage) { pl.status = RS_LIVE; pl.live = pl.pss_cand = pl.fk5 = pl.fk3 = false; }
        // this end's reference window, issued for every candidate before the -U/-D test so the
        // test costs no extra memory round trip
        const bool cand = DO_PSS && pl.pss_cand;
        // 32 window positions = 16 bytes of the 4-bit packed reference (+ up to 7 nibbles of
        // misalignment): five dwords, one dwordx4 + one dword gather
        uint32_t gq[5] = {0u, 0u, 0u, 0u, 0u};
        uint32_t gsh = 0u;
        if (cand) {
            const uint64_t ga = pl.gbase + (uint64_t)pl.s + (e ? (uint64_t)pl.L - 30ull - row_base : (uint64_t)row_base - 2ull);
            const uint32_t *pg = P.genome4 + (ga >> 3);
            if (!(ablate & 128u)) {
                const Quad q0 = *(const Quad *)pg;
#pragma unroll
                for (int k = 0; k < 4; k++) gq[k] = q0.v[k];
                gq[4] = pg[4];
            }
            gsh = 4u * (uint32_t)(ga & 7ull);
        }
        // the first context base of this end (position s-1 / s+L) decides -U / -D in every pass; only
        // the window of pass 0 holds it
        uint32_t cx = 0u;
        if (cand && !pass0) {
            const uint64_t pc = pl.gbase + (uint64_t)pl.s + (e ? (uint64_t)pl.L : (uint64_t)-1ll);
            cx = (P.genome4[pc >> 3] >> (4u * (uint32_t)(pc & 7ull))) & 15u;
        }
        // read bases of this end as a nibble stream aligned with the window bytes: stream nibble
        // b <-> read base n0 + b, n0 = -2 (left: bytes 0,1 are context, their nibbles are never
        // used) or L-30 (right).  20 bytes from SEQ as six aligned dwords, one batch.
        const int32_t n0 = e ? (int32_t)pl.L - 30 - (int32_t)row_base : (int32_t)row_base - 2;
        uint32_t rr[6];
#pragma unroll
        for (int k = 0; k < 6; k++) rr[k] = 0u;
        uint32_t ssh = 0u;
        if (cand) {
            const int32_t n0a = min(n0, (int32_t)h.l_seq);  // (past SEQ everything is blanked anyway: stay inside the record)
            const uint32_t sa = src.off + (uint32_t)((int32_t)h.seq_off + (n0a >> 1));  // arithmetic shift = floor
            const uint32_t *qs = (const uint32_t *)(stage + (sa & ~3u));
            ssh = sa & 3u;
#pragma unroll
            for (int k = 0; k < 6; k++) rr[k] = qs[k];
        }
        // fragkon window of this side of the alignment, fetched now so its latency overlaps the
        // other loads: the left-end lane owns the window at s-k/2.. (5' k-mer of a forward read, 3'
        // of a reverse read), the right-end lane the one at ..s+L+k/2 (fragkon.c:152-181)
        const uint32_t kwhich = e ^ (pl.rev ? 1u : 0u);  // 0 = 5' table, 1 = 3' table
        const bool kmer_try = DO_KMER && (kwhich ? pl.fk3 : pl.fk5) && !(ablate & 16u);
        uint32_t kw[3] = {0u, 0u, 0u};
        uint32_t ksh = 0u;
        if (kmer_try) {
            int64_t w5, w3;
            kmer_windows(pl, P.K, w5, w3);
            const uint64_t ka = pl.gbase + (uint64_t)(kwhich ? w3 : w5);
            const Tri kq = *(const Tri *)(P.genome4 + (ka >> 3));  // 15 window nibbles at any alignment (<= 22 of 24), one gather
#pragma unroll
            for (int k = 0; k < 3; k++) kw[k] = kq.v[k];
            ksh = 4u * (uint32_t)(ka & 7ull);
        }
        uint32_t ev_over = 0u;  // events of a record handled by the out-of-line path
        if (in_tile && !in_stage && e == 0u) {
            ev_over = tally_overflow_record<DO_PSS, DO_KMER, LDS_KMER>(kernarg, o0, o1, table, lds_kmer);
            if (pass0) atomicAdd(&lds_delta[ST_SLOW_PATH], 1);
        }
        // First use of the gathered registers happens HERE, before the next tile's DMA is issued:
        // vmcnt retires in order and hipcc's counted wait for these loads cannot see the
        // asm-issued DMA pieces, so a wait placed after the DMA issue would also wait for the
        // whole transfer and serialise it against CODES-B / COLUMNS.
        uint32_t W[4];  // nibble q of W[m] = window position 8m + q
#pragma unroll
        for (int m = 0; m < 4; m++) W[m] = __builtin_amdgcn_alignbit(gq[m + 1], gq[m], gsh);
#pragma unroll
        for (int k = 0; k < 2; k++) kw[k] = __builtin_amdgcn_alignbit(kw[k + 1], kw[k], ksh);
        // pin those uses here (the scheduler would otherwise sink them below the DMA issue)
#pragma unroll
        for (int m = 0; m < 4; m++) asm volatile("" : "+v"(W[m]));
        asm volatile("" : "+v"(cx));
        if (DO_KMER) {
#pragma unroll
            for (int k = 0; k < 2; k++) asm volatile("" : "+v"(kw[k]));
        }
        __syncthreads();

        // every wave is done with `stage`: the next tile's DMA starts now (its offsets were put
        // into LDS before the barrier at the top) and lands behind the rest of CODES and COLUMNS;
        // the offsets of the tile after that go into VGPRs
        if (next < n_tiles) {
            stage_tile_dma(P.recs, recs_limit, toffs + (par ^ 1u) * TOFF, tile_count(next), pieces, stage, tid);
            if (next + tstride < n_tiles) load_offsets(next + tstride);
        }

        // ---- CODES, part B: registers only ------------------------------------------------------------
        {
            // first context base next to the alignment: left window position 1 (s-1), right position 30 (s+L)
            const uint32_t own1 = !pass0 ? cx : e ? (W[3] >> 24) & 15u : (W[0] >> 4) & 15u;
            const uint32_t other1 = (uint32_t)__shfl_xor((int)own1, 1);
            if (DO_PSS) plan_finish_pss_packed(P.acgt_ctx, pl, e ? other1 : own1, e ? own1 : other1);
            uint32_t code_w[8];
#pragma unroll
            for (int k = 0; k < 8; k++) code_w[k] = CODE_NONE * 0x01010101u;
            // this lane's end feeds: left -> fwd table on forward reads, rev table on reverse reads
            const uint32_t tsel = e ^ (pl.rev ? 1u : 0u);
            if (cand && (tsel ? pl.pss_rev : pl.pss_fwd) && !(ablate & 2u)) {
                // Four window positions per VALU instruction, no memory lookups: v_perm_b32 with the
                // DATA as selector is an 8-entry byte table (selectors 0-7 pick a pool byte, 8-11
                // replicate the sign of pool byte 1/3/5/7, 12 gives 0x00, >= 13 gives 0xFF).
                //   read base : selector = BAM nibble ^ 4   -> A(1)->5  C(2)->6  G(4)->0  T(8)->12
                //               pool: [0]=G [5]=A [6]=C, T is the hardware's 0x00, every other nibble
                //               lands on 0xFF (pool filler, sign replicas of bytes 1,3,5,7, or >= 13);
                //               value = (3 - idx) << 3  (complemented: T must be 0), A carries 0x80 so
                //               that selector 10 (nibble 14) replicates a set sign bit
                //   reference : selector = the packed reference's nibble: 0..3 = A C G T, 4..7 = "other"
                //               pool: [0..3] = (3 - idx) << 1, [4..7] = 0xFF
                // OR of the two = (15 - cell) << 1, the reverse-strand code; forward-strand lanes XOR
                // 0x1E to get cell << 1.  Anything invalid is 0xFF and ends, after the final & 0x3F,
                // on code 33 or 63: rows >= 32 of the count table are the trash bin.
                // Both nibble streams are split into EVEN and ODD positions (byte i of E[m] / O[m] =
                // position 8m + 2i / 8m + 2i + 1); the code sheet row keeps that order, see COLUMNS.
                uint32_t S[5];
#pragma unroll
                for (int k = 0; k < 5; k++) S[k] = __builtin_amdgcn_alignbyte(rr[k + 1], rr[k], ssh);
                // SEQ byte i holds positions 2i (high nibble), 2i+1 (low nibble) when n0 is even;
                // when n0 is odd position 2i is the LOW nibble of byte i and 2i+1 the HIGH nibble of
                // byte i+1
                const bool odd = (n0 & 1) != 0;
                const uint32_t M = 0x0F0F0F0Fu;
                const uint32_t sx = pl.rev ? 0u : 0x1E1E1E1Eu;
                const uint32_t tsel4 = tsel * 0x01010101u;
                uint32_t RE[4], RO[4], GE[4], GO[4];
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const uint32_t S1 = __builtin_amdgcn_alignbyte(S[m + 1], S[m], 1);
                    const uint32_t A = odd ? S1 : S[m];
                    const uint32_t hiA = (A >> 4) & M, loS = S[m] & M;
                    const uint32_t E = odd ? loS : hiA, O = odd ? hiA : loS;
                    RE[m] = __builtin_amdgcn_perm(0xFF1098FFu, 0xFFFFFF08u, E ^ 0x04040404u);
                    RO[m] = __builtin_amdgcn_perm(0xFF1098FFu, 0xFFFFFF08u, O ^ 0x04040404u);
                    // the packed reference is little-endian in nibbles: even positions are the low ones
                    GE[m] = __builtin_amdgcn_perm(0xFFFFFFFFu, 0x00020406u, W[m] & M);
                    GO[m] = __builtin_amdgcn_perm(0xFFFFFFFFu, 0x00020406u, (W[m] >> 4) & M);
                }
                // bases at or beyond l_seq do not exist (precondition P3): blank them.  Rare (reads
                // shorter than the window), so the whole wave skips it when no lane needs it.
                const int32_t have_s = (int32_t)h.l_seq - n0;
                const uint32_t have = have_s <= 0 ? 0u : have_s >= 32 ? 32u : (uint32_t)have_s;
                if (__any(have < (e && pass0 ? 30u : 32u))) {
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        // bytes i of E[m] with 8m + 2i < have, of O[m] with 8m + 2i + 1 < have
                        const uint32_t left = have > 8u * m ? have - 8u * m : 0u;
                        const uint32_t ne = min((left + 1u) >> 1, 4u), no = min(left >> 1, 4u);
                        RE[m] |= ne >= 4u ? 0u : ~((1u << (8u * ne)) - 1u);
                        RO[m] |= no >= 4u ? 0u : ~((1u << (8u * no)) - 1u);
                    }
                }
                // context positions carry no read base: their cell is the diagonal one of their own
                // reference base (pss-bam.c:172-184).  Left: positions 0,1 = byte 0 of E[0], O[0];
                // right: positions 30,31 = byte 3 of E[3], O[3].
                {
                    const uint32_t ml = (e || !pass0) ? 0u : 0x000000FFu, mr = (e && pass0) ? 0xFF000000u : 0u;
                    RE[0] = (RE[0] & ~ml) | ((GE[0] << 2) & ml & 0x18181818u);
                    RO[0] = (RO[0] & ~ml) | ((GO[0] << 2) & ml & 0x18181818u);
                    RE[3] = (RE[3] & ~mr) | ((GE[3] << 2) & mr & 0x18181818u);
                    RO[3] = (RO[3] & ~mr) | ((GO[3] << 2) & mr & 0x18181818u);
                }
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    code_w[2 * m] = ((RE[m] | GE[m] | tsel4) ^ sx) & 0x3F3F3F3Fu;
                    code_w[2 * m + 1] = ((RO[m] | GO[m] | tsel4) ^ sx) & 0x3F3F3F3Fu;
                }
            }
            bool kmer_ok = true;
            if (kmer_try) {
                // bin = base-4 number of the k bases read left to right (kmer.c:184-214); for a
                // reverse-strand read the window is reverse-complemented (fragkon.c:156-160)
                uint32_t bin = 0u, bad = 0u;
#pragma unroll
                for (int t = 0; t < 16; t++) {   // K <= 15
                    if (t < P.K) {
                        const uint32_t c = (kw[t >> 3] >> (4 * (t & 7))) & 0xFu;
                        bad |= c & ~3u;
                        bin = pl.rev ? (bin | ((3u - (c & 3u)) << (2 * t))) : ((bin << 2) | (c & 3u));
                    }
                }
                kmer_ok = bad == 0u;
                if (kmer_ok && !(ablate & 8u)) {
                    if (LDS_KMER) atomicAdd(&lds_kmer[(kwhich ? (1u << (2 * P.K)) : 0u) + bin], 1u);
                    else atomicAdd(&P.counters[(kwhich ? P.off_k3 : P.off_k5) + bin], 1ull);
                }
            }
            if (lane_on) {  // code sheet row of read j: bytes [e*32, e*32+32)
                uint4 *dst = (uint4 *)(sheet + j * 64u + e * 32u);
                dst[0] = make_uint4(code_w[0], code_w[1], code_w[2], code_w[3]);
                dst[1] = make_uint4(code_w[4], code_w[5], code_w[6], code_w[7]);
            }
            bool kfail = false;
            if (DO_KMER) {
                // fragkon status of the read: -1 when any attempted add failed (either lane of the pair)
                const int bad = (kmer_try && !kmer_ok) ? 1 : 0;
                const int bad_other = __shfl_xor(bad, 1);  // every lane takes part in the exchange
                kfail = (bad | bad_other) != 0;
            }
            if (e == 0u && pass0) book_events(DO_PSS, DO_KMER, in_stage ? record_events(DO_PSS, DO_KMER, pl, kfail) : ev_over, lds_delta);
        }
        // (no barrier: wave w wrote the sheet rows of reads 32w .. 32w+31 -- j = tid >> 1 -- and its
        //  COLUMNS pass below reads exactly those rows)

        // ---- COLUMNS: wave-per-read, lane = (end, window byte) -----------------------------------
        if (DO_PSS && !(ablate & 1u)) {
            // sheet byte b of an end holds window position 8*(b/8) + 2*(b%4) + (b/4)%2 (even/odd split)
            const uint32_t e = lane >> 5, b = lane & 31u;
            const uint32_t wpos = (b & 24u) + 2u * (b & 3u) + ((b >> 2) & 1u);
            const uint32_t row = e ? 31u - wpos : wpos;
            const uint32_t j0 = min(count, wave * 32u), j1 = min(count, j0 + 32u);
            if (row < n_live) {  // (lanes of dead rows would only ever see CODE_NONE)
                uint32_t j = j0;
                for (; j + 8u <= j1; j += 8u) {
                    uint32_t c[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) c[u] = sheet[(j + u) * 64u + lane];
#pragma unroll
                    for (int u = 0; u < 8; u++) atomicAdd(&table[(c[u] << 5) + row], 1u);
                }
                for (; j < j1; j++) atomicAdd(&table[((uint32_t)sheet[j * 64u + lane] << 5) + row], 1u);
            }
        }
    }

    __syncthreads();
    // Partial results leave the workgroup as plain coalesced stores into its own scratch slot;
    // reduce_partials() sums the slots afterwards.  (Flushing with global atomics instead had
    // ~1000 workgroups queue on the same few hundred counters at the same moment: 6 % of the
    // kernel's time.)
    uint32_t *mine = P.scratch + (size_t)blockIdx.x * SCRATCH_WORDS;
    for (uint32_t i = tid; i < 32u * 32u; i += TILED_THREADS) mine[i] = DO_PSS ? table[i] : 0u;
    for (uint32_t i = tid; i < 512u; i += TILED_THREADS)
        mine[SCRATCH_KMER + i] = (LDS_KMER && i < 2u * (1u << (2 * P.K))) ? lds_kmer[i] : 0u;
    if (tid < 16u) mine[SCRATCH_DELTA + tid] = tid < (uint32_t)ST_USED ? (uint32_t)lds_delta[tid] : 0u;
}


// ---------------------------------------------------------------------------------------
// tally_compact: the tiled kernel for -r N <= 16 (the tool's default is 15)
// ---------------------------------------------------------------------------------------
// Short windows leave half of tally_tiled's work on dead positions: its 32-byte windows cost
// four v_perm groups per end and one COLUMNS wave-iteration per read whatever N is, and short
// records (30-80 bp ancient-DNA reads, 138 B) no longer hide that behind their record stream:
// BASELINE config 4 ran at 17 VALU wave-instructions per read, 66 % VALU-busy, HBM half idle
// (profiles/r02_C4_before.json).  This variant keeps the pipeline (LDS-DMA staged prefixes ->
// lane pair per read -> code sheet -> column tally) and changes the window geometry:
//   * a window holds the 16 POSITIONS of an end only (two v_perm groups): left end = read bases
//     0..15 / reference s..s+15, right end = read bases L-16..L-1 / reference s+L-16..s+L-1
//     (position i from the right = byte 15-i), so both ends use the same two groups and the left
//     end's nibble stream is always even-aligned;
//   * one dwordx4 gather per end fetches its 16 positions AND its two context bases (18 nibbles
//     at any nibble alignment fit 4 dwords);
//   * the context rows (0, 1) can only take the four diagonal cells: each (read, end) lane adds
//     them with two LDS atomics into a 16-fold replicated 16-word table (lane & 15 picks the
//     replica, so the AA/CC/GG/TT skew meets at most 4 lanes per word);
//   * the code sheet row is 32 bytes, and COLUMNS takes TWO reads per wave-iteration:
//     lane = (read parity, end, byte); the count table's 32 slots per code are (end, position),
//     so the 32 lanes of a half-wave still never share a word or a bank; the two ends' slots are
//     summed when the table leaves LDS.
// The DMA issue loop addresses through an SGPR base + 32-bit lane offset (no 64-bit VALU adds)
// and saves/restores m0 around the instruction (the compiler does not model the write).
constexpr uint32_t COMPACT_MAX_ROWS = 18;     // 2 context rows + 16 positions
constexpr uint32_t CTX_REP = 16;              // replicas of the context-row cells
constexpr uint32_t CTX_WORDS = 16 * CTX_REP;  // [table 2][row 2][base 4][replica 16]
constexpr uint32_t COMPACT_TABLE_WORDS = 33 * 32;  // codes 0..31 + one trash row (5 workgroups per CU instead of 4)

// piece q of the tile -> stage + q*16, source = record's 16-byte aligned start + 16 * piece;
// `full` tiles (count == T, every piece inside the block) skip the per-lane bounds tests
__device__ __forceinline__ void stage_tile_dma32(const uint8_t *recs, uint32_t recs_limit32, const uint32_t *tile_offs,
                                                 uint32_t count, uint32_t pieces, uint8_t *stage, uint32_t tid) {
    const uint32_t n_pieces = count * pieces;
    const uint32_t lds0 = (uint32_t)(uintptr_t)stage;
    const uint32_t magic = ((1u << 20) + pieces - 1u) / pieces;   // q / pieces, exact for q < 2^13, pieces <= 64
    for (uint32_t q0 = (tid & ~63u); q0 < n_pieces; q0 += TILED_THREADS) {
        const uint32_t q = q0 + (tid & 63u);
        const uint32_t jq = (q * magic) >> 20;
        const uint32_t j = min(jq, count - 1u), pc = q - jq * pieces;
        const uint32_t a = (tile_offs[j] & ~15u) + 16u * pc;     // < 4 GiB: the block is
        if (q < n_pieces && a <= recs_limit32) {                  // recs_limit32 = readable end - 16
            const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds0 + (q0 << 4));
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(a), "s"(recs), "s"(m0v) : "memory");
        }
    }
}

struct LdsTableCompact {  // overflow records (one lane, global-memory decode) in the compact layout
    uint32_t *t;          // [(cell << 1) | table][32 slots]: slot = end * 16 + (row - 2); both ends are summed at the end
    uint32_t *ctx;        // [table][row][base][replica]
    __device__ __forceinline__ void add(uint32_t table, uint32_t row, uint32_t cell) const {
        if (row >= 2u) { if (row < COMPACT_MAX_ROWS) atomicAdd(&t[(((cell << 1) | table) << 5) + (row - 2u)], 1u); }
        else atomicAdd(&ctx[((table * 2u + row) * 4u + cell / 5u) * CTX_REP], 1u);   // context cells are diagonal: 0,5,10,15
    }
};

template <bool DO_KMER, bool LDS_KMER>
__device__ __attribute__((noinline)) uint32_t tally_overflow_record_compact(const TallyParams *kernarg, uint32_t o0, uint32_t o1,
                                                                            uint32_t *table, uint32_t *ctx, uint32_t *lds_kmer) {
    const TallyParams &P = *kernarg;
    GlobalBytes gsrc{P.recs + o0};
    const RecHdr gh = decode_hdr(gsrc, o1 - o0);
    const Plan gpl = make_plan<true, DO_KMER>(P, gsrc, gh);
    if (gpl.pss_fwd || gpl.pss_rev) tally_pss_record(P, LdsTableCompact{table, ctx}, gsrc, gh, gpl);
    bool kfail = false;
    if (DO_KMER && (gpl.fk5 || gpl.fk3)) kfail = tally_kmer_record<LDS_KMER>(P, gpl, lds_kmer);
    return record_events(true, DO_KMER, gpl, kfail);
}

template <bool DO_KMER, bool LDS_KMER, int DECODE_REPS = 1, bool PLAN_ONCE = false>
__device__ __forceinline__ void tally_compact_body(const TallyParams &P, const TallyParams *kernarg,
                                                   uint8_t *__restrict__ stage, uint8_t *__restrict__ sheet,
                                                   uint32_t *__restrict__ table, uint32_t *__restrict__ toffs,
                                                   uint32_t *__restrict__ lds_kmer, int32_t *__restrict__ lds_delta,
                                                   uint4 *__restrict__ refs_lds, uint32_t *__restrict__ ctx_rep,
                                                   uint32_t *__restrict__ ovf_list, uint32_t *__restrict__ ovf_n) {
    const uint32_t T = P.reads_per_tile;
    const uint32_t n_recs = P.n_recs_dev ? *P.n_recs_dev : P.n_recs;   // device-indexed blocks: the count lives in device memory
    const uint32_t pieces = P.prefix_pieces;
    const uint32_t recs_limit32 = (uint32_t)(((P.recs_bytes + 15ull) & ~15ull) - 16ull);  // last piece start that is readable
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;

    // -U / -D membership by packed-reference nibble, complemented form in the upper half:
    // bit n = nibble n passes, bit 16 + n = the complement of nibble n passes
    uint32_t up_tab = 0u, dn_tab = 0u;
#pragma unroll
    for (uint32_t n = 0; n < 8u; n++) {
        const uint32_t c = n < 4u ? 3u - n : n;
        const uint32_t u = n < 4u ? (P.acgt_ctx >> (2u * n)) & 1u : n & 1u, uc = c < 4u ? (P.acgt_ctx >> (2u * c)) & 1u : c & 1u;
        const uint32_t d = n < 4u ? (P.acgt_ctx >> (2u * n + 1u)) & 1u : (n >> 1) & 1u, dc = c < 4u ? (P.acgt_ctx >> (2u * c + 1u)) & 1u : (c >> 1) & 1u;
        up_tab |= (u << n) | (uc << (16u + n));
        dn_tab |= (d << n) | (dc << (16u + n));
    }

    for (uint32_t i = tid; i < COMPACT_TABLE_WORDS; i += TILED_THREADS) table[i] = 0u;
    if (tid < CTX_WORDS) ctx_rep[tid] = 0u;
    if (LDS_KMER)
        for (uint32_t i = tid; i < 2u * (1u << (2 * P.K)); i += TILED_THREADS) lds_kmer[i] = 0u;
    if (tid < ST_USED) lds_delta[tid] = 0;
    if (tid == 0u) *ovf_n = 0u;
    const uint32_t n_ref_cached = min((uint32_t)P.n_ref, REF_LDS_ENTRIES);
    if (tid < n_ref_cached) refs_lds[tid] = P.ref_info[tid];
    if (tid == n_ref_cached) refs_lds[tid] = P.ref_info[P.n_ref];
    const uint32_t all_tiles = (n_recs + T - 1u) / T;
    uint32_t tile = blockIdx.x;
    const uint32_t tstride = gridDim.x, n_tiles = all_tiles;
    uint32_t off_a = 0;
    const uint32_t TOFF = TILED_MAX_T + 4u;
    auto load_offsets = [&](uint32_t t) {
        const uint32_t r0 = t * T;
        if (tid <= T && r0 + tid <= n_recs) off_a = P.offs[r0 + tid];
    };
    auto tile_count = [&](uint32_t t) { return min(T, n_recs - t * T); };
    __syncthreads();
    if (tile < n_tiles) {
        load_offsets(tile);
        if (tid <= T) toffs[tid] = off_a;
        __syncthreads();
        stage_tile_dma32(P.recs, recs_limit32, toffs, tile_count(tile), pieces, stage, tid);
        if (tile + tstride < n_tiles) load_offsets(tile + tstride);
    }

    for (uint32_t it = 0; tile < n_tiles; tile += tstride, it++) {
        const uint32_t par = it & 1u;
        const uint32_t *cur_offs = toffs + par * TOFF;
        const uint32_t r0 = tile * T;
        const uint32_t count = min(T, n_recs - r0);
        const uint32_t next = tile + tstride;

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (next < n_tiles && tid <= T) toffs[(par ^ 1u) * TOFF + tid] = off_a;
        __syncthreads();

        // ---- CODES, part A: everything that reads `stage` --------------------------------------
        const uint32_t j = tid >> 1, e = tid & 1u;
        const bool lane_on = tid < 2u * T;
        Plan pl;
        uint32_t l_seq = 0, soff = 0;   // SEQ length; LDS offset of the SEQ bytes of the staged record
        bool in_stage = false;
        if constexpr (PLAN_ONCE) {
            // A1: ONE lane per read (waves 0 and 1 of the four) decodes the header and applies the filters; the plan -- eight
            // words -- goes to LDS, where the code sheet will be written later (free between the previous tile's COLUMNS and
            // this tile's part B), and after one more barrier BOTH lanes of the read's pair pick it up.  The pair used to do
            // all of this twice, side by side: 17.6 % of the kernel's time on C4 (tools/ab_decode_twice.sh).
            uint4 *plan_lds = (uint4 *)sheet;
            if (tid < T) {
                const bool in_t = tid < count;
                uint32_t o0 = 0, o1 = 0;
                if (in_t) { o0 = cur_offs[tid]; o1 = cur_offs[tid + 1]; }
                const uint32_t avail = pieces * 16u - (o0 & 15u);
                const bool hdr_ok = in_t && o1 - o0 >= 36u && avail >= 48u;
                LdsBytes src{stage, hdr_ok ? tid * pieces * 16u + (o0 & 15u) : 0u};
                const RecHdr h = decode_hdr_lds32(src, hdr_ok ? o1 - o0 : 0u);
                const bool staged = hdr_ok && h.qual_off + 1u <= avail;
                Plan p1 = plan_head<true, DO_KMER, false>(P, src, h, RefsLdsCached{refs_lds, P.ref_info, n_ref_cached, (uint32_t)P.n_ref});
                if (!staged) { p1.status = RS_LIVE; p1.live = p1.pss_cand = p1.fk5 = p1.fk3 = false; }
                if (in_t && !staged) ovf_list[atomicAdd(ovf_n, 1u)] = tid;   // handled behind COLUMNS (below)
                const uint32_t bits = (p1.flag & 0xFFFFu) | (p1.status << 16) | (p1.live ? 1u << 18 : 0u) | (p1.pss_cand ? 1u << 19 : 0u) |
                                      (p1.fk5 ? 1u << 20 : 0u) | (p1.fk3 ? 1u << 21 : 0u) | (staged ? 1u << 22 : 0u);
                plan_lds[2u * tid] = make_uint4((uint32_t)p1.gbase, (uint32_t)(p1.gbase >> 32), (uint32_t)p1.s, p1.L);
                plan_lds[2u * tid + 1u] = make_uint4(h.l_seq, src.off + h.seq_off, bits, p1.Lk);
            }
            __syncthreads();
            const uint4 qa = plan_lds[2u * j], qb = plan_lds[2u * j + 1u];   // (j < 128: inside the sheet whatever T is)
            pl.gbase = (uint64_t)qa.x | ((uint64_t)qa.y << 32);
            pl.s = (int32_t)qa.z;
            pl.L = qa.w;
            l_seq = qb.x;
            soff = qb.y;
            pl.flag = qb.z & 0xFFFFu;
            pl.status = (qb.z >> 16) & 3u;
            pl.live = lane_on && ((qb.z >> 18) & 1u);
            pl.pss_cand = lane_on && ((qb.z >> 19) & 1u);
            pl.fk5 = lane_on && ((qb.z >> 20) & 1u);
            pl.fk3 = lane_on && ((qb.z >> 21) & 1u);
            in_stage = lane_on && ((qb.z >> 22) & 1u);
            pl.Lk = qb.w;
            pl.rev = (pl.flag & FL_REVERSE) != 0;
            pl.pss_fwd = pl.pss_rev = false;
        } else {
        const bool in_tile = lane_on && j < count;
        uint32_t o0 = 0, o1 = 0;
        if (in_tile) { o0 = cur_offs[j]; o1 = cur_offs[j + 1]; }
        const uint32_t avail = pieces * 16u - (o0 & 15u);
        const bool hdr_ok = in_tile && o1 - o0 >= 36u && avail >= 48u;
        LdsBytes src{stage, hdr_ok ? j * pieces * 16u + (o0 & 15u) : 0u};
        const RecHdr h = decode_hdr_lds32(src, hdr_ok ? o1 - o0 : 0u);
        const uint32_t needed = h.qual_off + 1u;   // (the -R filter, which walks the aux fields, stays with tally_tiled)
        in_stage = hdr_ok && needed <= avail;
        pl = plan_head<true, DO_KMER, false>(P, src, h, RefsLdsCached{refs_lds, P.ref_info, n_ref_cached, (uint32_t)P.n_ref});
        if constexpr (DECODE_REPS > 1) {
            // diagnostics (PSSBAM_COMPACT_DECODE_TWICE): header decode + filters a second time, from an offset the
            // compiler cannot tell from the first -- the time this adds is what the pair's shared decode costs per tile
            uint32_t off2 = src.off;
            asm volatile("" : "+v"(off2));
            const LdsBytes src2{stage, off2};
            const RecHdr h2 = decode_hdr_lds32(src2, hdr_ok ? o1 - o0 : 0u);
            const Plan p2 = plan_head<true, DO_KMER, false>(P, src2, h2, RefsLdsCached{refs_lds, P.ref_info, n_ref_cached, (uint32_t)P.n_ref});
            const uint32_t sink = (uint32_t)p2.s ^ p2.L ^ p2.flag ^ (uint32_t)p2.gbase ^ (uint32_t)(p2.gbase >> 32) ^ p2.status ^ (p2.pss_cand ? 1u : 0u) ^
                                  (p2.rev ? 2u : 0u) ^ (p2.fk5 ? 4u : 0u) ^ (p2.fk3 ? 8u : 0u) ^ h2.seq_off ^ h2.qual_off;
            asm volatile("" ::"v"(sink));
        }
        if (!in_stage) { pl.status = RS_LIVE; pl.live = pl.pss_cand = pl.fk5 = pl.fk3 = false; }
        l_seq = h.l_seq;
        soff = src.off + h.seq_off;
        // a record whose needed prefix is not staged is queued and handled after COLUMNS, where
        // almost nothing is live (the out-of-line call would otherwise sit in the register-hungry
        // middle of the tile)
        if (in_tile && !in_stage && e == 0u) ovf_list[atomicAdd(ovf_n, 1u)] = j;
        }
        const bool cand = pl.pss_cand;
        // this end's 16 positions + 2 context bases: 18 nibbles of the packed reference from
        //   left : s-2 .. s+15      (nibbles 0,1 = second, first context base; 2..17 = positions 0..15)
        //   right: s+L-16 .. s+L+1  (nibbles 0..15 = positions, byte w <-> position 15-w from the
        //                            right end; 16, 17 = first, second context base)
        uint32_t gq[4] = {0u, 0u, 0u, 0u};
        uint32_t gsh = 0u;
        if (cand) {
            const uint64_t ga = pl.gbase + (uint64_t)(int64_t)((int32_t)pl.s + (e ? (int32_t)pl.L - 16 : -2));
            const Quad q0 = *(const Quad *)(P.genome4 + (ga >> 3));
#pragma unroll
            for (int k = 0; k < 4; k++) gq[k] = q0.v[k];
            gsh = 4u * (uint32_t)(ga & 7ull);
        }
        // read bases as a nibble stream aligned with the window bytes: stream nibble b <-> read base
        // n0 + b, n0 = 0 (left) or L-16 (right; may be negative for L < 16: those positions are
        // beyond N <= L and never tallied).  9 bytes at any byte alignment = three aligned dwords.
        const int32_t n0 = e ? (int32_t)pl.L - 16 : 0;
        uint32_t rr[3] = {0u, 0u, 0u};
        uint32_t ssh = 0u;
        if (cand) {
            const int32_t n0a = min(n0, (int32_t)l_seq);
            const uint32_t sa = soff + (uint32_t)(n0a >> 1);
            const uint32_t *qs = (const uint32_t *)(stage + (sa & ~3u));
            ssh = sa & 3u;
#pragma unroll
            for (int k = 0; k < 3; k++) rr[k] = qs[k];
        }
        const uint32_t kwhich = e ^ (pl.rev ? 1u : 0u);
        const bool kmer_try = DO_KMER && (kwhich ? pl.fk3 : pl.fk5);
        uint32_t kw[3] = {0u, 0u, 0u};
        uint32_t ksh = 0u;
        if (kmer_try) {
            int64_t w5, w3;
            kmer_windows(pl, P.K, w5, w3);
            const uint64_t ka = pl.gbase + (uint64_t)(kwhich ? w3 : w5);
            const Tri kq = *(const Tri *)(P.genome4 + (ka >> 3));
#pragma unroll
            for (int k = 0; k < 3; k++) kw[k] = kq.v[k];
            ksh = 4u * (uint32_t)(ka & 7ull);
        }
        // consume the gathered registers before the next DMA is issued (vmcnt retires in order)
        uint32_t A[3];  // nibble q of A[m] = window nibble 8m + q
#pragma unroll
        for (int m = 0; m < 3; m++) A[m] = __builtin_amdgcn_alignbit(gq[m + 1], gq[m], gsh);
#pragma unroll
        for (int k = 0; k < 2; k++) kw[k] = __builtin_amdgcn_alignbit(kw[k + 1], kw[k], ksh);
#pragma unroll
        for (int m = 0; m < 3; m++) asm volatile("" : "+v"(A[m]));
        if (DO_KMER) {
#pragma unroll
            for (int k = 0; k < 2; k++) asm volatile("" : "+v"(kw[k]));
        }
        __syncthreads();

        if (next < n_tiles) {
            stage_tile_dma32(P.recs, recs_limit32, toffs + (par ^ 1u) * TOFF, tile_count(next), pieces, stage, tid);
            if (next + tstride < n_tiles) load_offsets(next + tstride);
        }

        // ---- CODES, part B: registers only -----------------------------------------------------
        {
            // context nibbles: left = nibbles 0 (second), 1 (first) of A[0]; right = nibbles 0 (first), 1 (second) of A[2]
            const uint32_t cx = e ? A[2] : A[0];
            const uint32_t c_lo = cx & 15u, c_hi = (cx >> 4) & 15u;
            const uint32_t own1 = e ? c_lo : c_hi, own2 = e ? c_hi : c_lo;   // first / second context base of this end
            const uint32_t other1 = (uint32_t)__shfl_xor((int)own1, 1);
            {   // pss-bam.c:134-142, :428-494 with the membership tables
                const uint32_t left1 = e ? other1 : own1, right1 = e ? own1 : other1;
                const uint32_t rsh = pl.rev ? 16u : 0u;
                const bool up_ok = ((up_tab >> ((pl.rev ? right1 : left1) + rsh)) & 1u) != 0;
                const bool dn_ok = ((dn_tab >> ((pl.rev ? left1 : right1) + rsh)) & 1u) != 0;
                const bool paired = (pl.flag & FL_PAIRED) != 0;
                const bool r1 = (pl.flag & FL_READ1) != 0, r2 = (pl.flag & FL_READ2) != 0;
                pl.pss_fwd = pl.pss_cand && (paired ? (r1 && up_ok) : (up_ok && dn_ok));
                pl.pss_rev = pl.pss_cand && (paired ? (!(r1 && up_ok) && r2 && dn_ok) : (up_ok && dn_ok));
            }
            // the 16 position nibbles: left = window nibbles 2..17, right = 0..15
            const uint32_t G0 = e ? A[0] : __builtin_amdgcn_alignbit(A[1], A[0], 8);
            const uint32_t G1 = e ? A[1] : __builtin_amdgcn_alignbit(A[2], A[1], 8);
            uint32_t code_w[4];
#pragma unroll
            for (int k = 0; k < 4; k++) code_w[k] = CODE_NONE * 0x01010101u;
            const uint32_t tsel = e ^ (pl.rev ? 1u : 0u);
            const bool tallied = cand && (tsel ? pl.pss_rev : pl.pss_fwd);
            if (tallied) {
                // (same v_perm byte tables as tally_tiled, two groups)
                uint32_t S[3];
                S[0] = __builtin_amdgcn_alignbyte(rr[1], rr[0], ssh);
                S[1] = __builtin_amdgcn_alignbyte(rr[2], rr[1], ssh);
                S[2] = __builtin_amdgcn_alignbyte(0u, rr[2], ssh);
                const bool odd = (n0 & 1) != 0;
                const uint32_t M = 0x0F0F0F0Fu;
                const uint32_t sx = pl.rev ? 0u : 0x1E1E1E1Eu;
                const uint32_t tsel4 = tsel * 0x01010101u;
                const uint32_t Gw[2] = {G0, G1};
                uint32_t RE[2], RO[2], GE[2], GO[2];
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    const uint32_t S1 = __builtin_amdgcn_alignbyte(S[m + 1], S[m], 1);
                    const uint32_t Ax = odd ? S1 : S[m];
                    const uint32_t hiA = (Ax >> 4) & M, loS = S[m] & M;
                    const uint32_t E = odd ? loS : hiA, O = odd ? hiA : loS;
                    RE[m] = __builtin_amdgcn_perm(0xFF1098FFu, 0xFFFFFF08u, E ^ 0x04040404u);
                    RO[m] = __builtin_amdgcn_perm(0xFF1098FFu, 0xFFFFFF08u, O ^ 0x04040404u);
                    GE[m] = __builtin_amdgcn_perm(0xFFFFFFFFu, 0x00020406u, Gw[m] & M);
                    GO[m] = __builtin_amdgcn_perm(0xFFFFFFFFu, 0x00020406u, (Gw[m] >> 4) & M);
                }
                // bases at or beyond l_seq do not exist (precondition P3): blank them
                const int32_t have_s = (int32_t)l_seq - n0;
                const uint32_t have = have_s <= 0 ? 0u : have_s >= 16 ? 16u : (uint32_t)have_s;
                if (__any(have < 16u)) {
#pragma unroll
                    for (int m = 0; m < 2; m++) {
                        const uint32_t left = have > 8u * m ? have - 8u * m : 0u;
                        const uint32_t ne = min((left + 1u) >> 1, 4u), no = min(left >> 1, 4u);
                        RE[m] |= ne >= 4u ? 0u : ~((1u << (8u * ne)) - 1u);
                        RO[m] |= no >= 4u ? 0u : ~((1u << (8u * no)) - 1u);
                    }
                }
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    code_w[2 * m] = ((RE[m] | GE[m] | tsel4) ^ sx) & 0x3F3F3F3Fu;
                    code_w[2 * m + 1] = ((RO[m] | GO[m] | tsel4) ^ sx) & 0x3F3F3F3Fu;
                }
                // context rows (pss-bam.c:169-189): row 1 <- first context base, row 0 <- second; the
                // cell is the diagonal one of the base (complemented for reverse-strand reads)
                const uint32_t rep = lane & (CTX_REP - 1u);
                const uint32_t b1 = pl.rev ? 3u - own1 : own1, b2 = pl.rev ? 3u - own2 : own2;
                if (own1 < 4u) atomicAdd(&ctx_rep[((tsel * 2u + 1u) * 4u + b1) * CTX_REP + rep], 1u);
                if (own2 < 4u) atomicAdd(&ctx_rep[((tsel * 2u + 0u) * 4u + b2) * CTX_REP + rep], 1u);
            }
            bool kmer_ok = true;
            if (kmer_try) {
                uint32_t bin = 0u, bad = 0u;
#pragma unroll
                for (int t = 0; t < 16; t++) {   // K <= 15
                    if (t < P.K) {
                        const uint32_t c = (kw[t >> 3] >> (4 * (t & 7))) & 0xFu;
                        bad |= c & ~3u;
                        bin = pl.rev ? (bin | ((3u - (c & 3u)) << (2 * t))) : ((bin << 2) | (c & 3u));
                    }
                }
                kmer_ok = bad == 0u;
                if (kmer_ok) {
                    if (LDS_KMER) atomicAdd(&lds_kmer[(kwhich ? (1u << (2 * P.K)) : 0u) + bin], 1u);
                    else atomicAdd(&P.counters[(kwhich ? P.off_k3 : P.off_k5) + bin], 1ull);
                }
            }
            if (lane_on)   // code sheet row of read j: 16 bytes per end, [E0 O0 E1 O1]
                *(uint4 *)(sheet + j * 32u + e * 16u) = make_uint4(code_w[0], code_w[1], code_w[2], code_w[3]);
            bool kfail = false;
            if (DO_KMER) {
                const int bad = (kmer_try && !kmer_ok) ? 1 : 0;
                const int bad_other = __shfl_xor(bad, 1);
                kfail = (bad | bad_other) != 0;
            }
            if (e == 0u && in_stage) book_events(true, DO_KMER, record_events(true, DO_KMER, pl, kfail), lds_delta);
        }
        // (no barrier here: wave w wrote the sheet rows of reads 32w .. 32w+31 -- j = tid >> 1 -- and
        //  its COLUMNS pass reads exactly those rows)

        // ---- COLUMNS: two reads per wave-iteration, lane = (read parity, end, sheet byte) ----------
        {
            // sheet byte b of an end holds window byte w = 8*(b/8) + 2*(b%4) + (b/4)%2; left: position w,
            // right: position 15-w; table slot = end*16 + position.  Every code >= 32 ("no count")
            // lands in the one trash row 32.
            const uint32_t ee = (lane >> 4) & 1u, b = lane & 15u;
            const uint32_t w = (b & 8u) + 2u * (b & 3u) + ((b >> 2) & 1u);
            const uint32_t posn = ee ? 15u - w : w;
            const uint32_t slot = ee * 16u + posn;
            const uint32_t j0 = min(count, wave * 32u), j1 = min(count, j0 + 32u);
            if (posn < (uint32_t)P.N) {
                const uint32_t rd = lane >> 5;   // which read of the pair
                uint32_t jj = j0;
                for (; jj + 16u <= j1; jj += 16u) {
                    uint32_t c[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) c[u] = sheet[(jj + 2u * u) * 32u + lane];
#pragma unroll
                    for (int u = 0; u < 8; u++) atomicAdd(&table[(min(c[u], 32u) << 5) + slot], 1u);
                }
                for (; jj < j1; jj += 2u)
                    if (jj + rd < j1) atomicAdd(&table[(min((uint32_t)sheet[jj * 32u + lane], 32u) << 5) + slot], 1u);
            }
        }
        // queued overflow records of this tile (rare: a record of tens of KB, or a block whose later
        // records are longer than the sampled prefix).  ovf_n was final at the barrier behind CODES-A.
        const uint32_t n_ovf = *ovf_n;
        if (n_ovf) {
            for (uint32_t i = tid; i < n_ovf; i += TILED_THREADS) {
                const uint32_t jo = ovf_list[i];
                const uint32_t ev = tally_overflow_record_compact<DO_KMER, LDS_KMER>(kernarg, cur_offs[jo], cur_offs[jo + 1u], table,
                                                                                    ctx_rep, lds_kmer);
                book_events(true, DO_KMER, ev, lds_delta);
                atomicAdd(&lds_delta[ST_SLOW_PATH], 1);
            }
            __syncthreads();   // everyone has read n_ovf / the list
            if (tid == 0u) *ovf_n = 0u;
        }
    }

    __syncthreads();
    // scratch slot in tally_tiled's format ([code][row]): rows 2.. = both ends' slots summed,
    // rows 0,1 = the context replicas summed (diagonal cells only)
    uint32_t *mine = P.scratch + (size_t)blockIdx.x * SCRATCH_WORDS;
    for (uint32_t i = tid; i < 32u * 32u; i += TILED_THREADS) {
        const uint32_t ct = i >> 5, row = i & 31u;
        uint32_t v = 0u;
        if (row >= 2u && row < COMPACT_MAX_ROWS) v = table[(ct << 5) + (row - 2u)] + table[(ct << 5) + 16u + (row - 2u)];
        else if (row < 2u) {
            const uint32_t t = ct & 1u, cell = ct >> 1;
            if (cell % 5u == 0u) {
                const uint32_t *rp = ctx_rep + ((t * 2u + row) * 4u + cell / 5u) * CTX_REP;
                for (uint32_t k = 0; k < CTX_REP; k++) v += rp[k];
            }
        }
        mine[i] = v;
    }
    for (uint32_t i = tid; i < 512u; i += TILED_THREADS)
        mine[SCRATCH_KMER + i] = (LDS_KMER && i < 2u * (1u << (2 * P.K))) ? lds_kmer[i] : 0u;
    if (tid < 16u) mine[SCRATCH_DELTA + tid] = tid < (uint32_t)ST_USED ? (uint32_t)lds_delta[tid] : 0u;
}

template <bool DO_KMER, bool LDS_KMER, bool PLAN_ONCE = false>
__global__ void __launch_bounds__(TILED_THREADS) tally_compact(const TallyParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t stage[];
    __shared__ __attribute__((aligned(16))) uint8_t sheet[TILED_MAX_T * 32u];
    __shared__ uint32_t table[COMPACT_TABLE_WORDS];
    __shared__ uint32_t toffs[2u * (TILED_MAX_T + 4u)];
    __shared__ uint32_t lds_kmer[LDS_KMER ? 2u * (1u << (2 * KMER_LDS_MAX_K)) : 1u];
    __shared__ int32_t lds_delta[ST_USED];
    __shared__ uint4 refs_lds[REF_LDS_ENTRIES + 1];
    __shared__ uint32_t ctx_rep[CTX_WORDS];
    __shared__ uint32_t ovf_list[TILED_MAX_T + 1];   // [TILED_MAX_T] = fill count
    const TallyParams *kernarg = (const TallyParams *)__builtin_amdgcn_kernarg_segment_ptr();
    tally_compact_body<DO_KMER, LDS_KMER, 1, PLAN_ONCE>(P, kernarg, stage, sheet, table, toffs, lds_kmer, lds_delta, refs_lds, ctx_rep, ovf_list,
                                                        ovf_list + TILED_MAX_T);
}

// diagnostics only (PSSBAM_COMPACT_DECODE_TWICE): the same kernel with the header decode + filters done twice per lane
__global__ void __launch_bounds__(TILED_THREADS) tally_compact_decode_twice(const TallyParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t stage[];
    __shared__ __attribute__((aligned(16))) uint8_t sheet[TILED_MAX_T * 32u];
    __shared__ uint32_t table[COMPACT_TABLE_WORDS];
    __shared__ uint32_t toffs[2u * (TILED_MAX_T + 4u)];
    __shared__ uint32_t lds_kmer[1u];
    __shared__ int32_t lds_delta[ST_USED];
    __shared__ uint4 refs_lds[REF_LDS_ENTRIES + 1];
    __shared__ uint32_t ctx_rep[CTX_WORDS];
    __shared__ uint32_t ovf_list[TILED_MAX_T + 1];
    const TallyParams *kernarg = (const TallyParams *)__builtin_amdgcn_kernarg_segment_ptr();
    tally_compact_body<false, false, 2>(P, kernarg, stage, sheet, table, toffs, lds_kmer, lds_delta, refs_lds, ctx_rep, ovf_list,
                                        ovf_list + TILED_MAX_T);
}

// Sums the per-workgroup partials of one tally_tiled launch into the u64 counter block.
// Thread (word w, group g) adds up slots g, g+REDUCE_GROUPS, ... of word w (loads coalesce across
// w) and contributes one atomic; launched on the same stream right behind the tally kernel.
constexpr uint32_t REDUCE_GROUPS = 32;
__global__ void __launch_bounds__(256) reduce_partials(const TallyParams P, uint32_t n_slots, uint32_t lds_kmer_on) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = gid % SCRATCH_WORDS, g = gid / SCRATCH_WORDS;
    if (g >= REDUCE_GROUPS) return;
    const bool do_pss = (P.tally_mask & 1u) != 0, do_kmer = (P.tally_mask & 2u) != 0;
    const uint32_t n_pos = (uint32_t)P.N + 2u;
    unsigned long long *dst = nullptr;
    bool is_delta = false;
    if (i < 1024u) {
        const uint32_t row = P.row_base + (i & 31u), ct = i >> 5, t = ct & 1u, cell = ct >> 1;
        if (do_pss && row < n_pos) dst = &P.counters[(t ? P.off_rev : 0u) + row * 16u + cell];
    } else if (i < SCRATCH_DELTA) {
        const uint32_t k = i - SCRATCH_KMER, nb = do_kmer ? 1u << (2 * P.K) : 0u;
        if (lds_kmer_on && k < 2u * nb) dst = &P.counters[k < nb ? P.off_k5 + k : P.off_k3 + (k - nb)];
    } else {
        const uint32_t k = i - SCRATCH_DELTA;  // status counters belong to pass 0
        if (k < (uint32_t)ST_USED && P.row_base == 0u) { dst = &P.counters[P.off_stats + k]; is_delta = true; }
    }
    if (!dst) return;
    long long sum = 0;
    const uint32_t *p = P.scratch + i;
#pragma unroll 8
    for (uint32_t b = g; b < n_slots; b += REDUCE_GROUPS) {
        const uint32_t v = p[(size_t)b * SCRATCH_WORDS];
        sum += is_delta ? (long long)(int32_t)v : (long long)v;
    }
    if (is_delta && g == 0u) {
        // every launch credits its record count to the OK slots; the deltas move records elsewhere
        const uint32_t k = i - SCRATCH_DELTA;
        if (k == ST_RECORDS || (do_pss && k == ST_PSS_OK) || (do_kmer && k == ST_KMER_OK)) sum += P.n_recs_dev ? *P.n_recs_dev : P.n_recs;
    }
    if (sum) atomicAdd(dst, (unsigned long long)sum);
}

template <bool DO_PSS, bool DO_KMER, bool LDS_KMER, bool LATER_PASS = false>
__global__ void __launch_bounds__(TILED_THREADS) tally_tiled(const TallyParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t stage[];
    __shared__ __attribute__((aligned(16))) uint8_t sheet[TILED_MAX_T * 64u];
    __shared__ uint32_t table[TABLE_WORDS];
    __shared__ uint32_t toffs[2u * (TILED_MAX_T + 4u)];
    __shared__ uint32_t lds_kmer[LDS_KMER ? 2u * (1u << (2 * KMER_LDS_MAX_K)) : 1u];
    __shared__ int32_t lds_delta[ST_USED];
    __shared__ uint4 refs_lds[REF_LDS_ENTRIES + 1];
    // the kernel's single argument, as it lies in the kernarg segment (for the out-of-line path)
    const TallyParams *kernarg = (const TallyParams *)__builtin_amdgcn_kernarg_segment_ptr();
    tally_tiled_body<DO_PSS, DO_KMER, LDS_KMER, LATER_PASS>(P, kernarg, stage, sheet, table, toffs, lds_kmer, lds_delta, refs_lds);
}

// genome-kmer-count (genome-kmer-count.c:69-79): every k-mer start of the device genome.  Each
// lane walks GKC_SPAN consecutive positions with a rolling 2-bit code; `run` = number of
// consecutive ACGT bases ending here, a window counts when run >= k.  Contig padding is stored
// as non-ACGT, so no window spans two contigs.  Bins: LDS histogram for k <= 6, else global.
constexpr uint32_t GKC_SPAN = 256;
template <bool LDS_BINS>
__global__ void __launch_bounds__(256) genome_kmer_kernel(const uint8_t *genome, uint64_t n, int K,
                                                          unsigned long long *bins) {
    __shared__ uint32_t lds_bins[LDS_BINS ? 4096 : 1];
    const uint32_t nb = 1u << (2 * K), mask = nb - 1u;
    if (LDS_BINS) {
        for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) lds_bins[i] = 0u;
        __syncthreads();
    }
    const uint64_t n_spans = (n + GKC_SPAN - 1) / GKC_SPAN;
    for (uint64_t sp = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; sp < n_spans; sp += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p0 = sp * GKC_SPAN, p1 = min(n, p0 + GKC_SPAN);
        // warm-up: the K-1 bases before the span (their windows belong to the previous span)
        uint32_t code = 0u, run = 0u;
        for (uint64_t p = p0 >= (uint64_t)(K - 1) ? p0 - (uint64_t)(K - 1) : 0; p < p0; p++) {
            const uint32_t c = genome[p];
            run = c < 4u ? run + 1u : 0u;
            code = ((code << 2) | (c & 3u)) & mask;
        }
        for (uint64_t p = p0; p < p1; p += 4) {  // spans start 4-byte aligned (GKC_SPAN % 4 == 0)
            const uint32_t w = *(const uint32_t *)(genome + p);
#pragma unroll
            for (int b = 0; b < 4; b++) {
                if (p + b < p1) {
                    const uint32_t c = (w >> (8 * b)) & 0xFFu;
                    run = c < 4u ? run + 1u : 0u;
                    code = ((code << 2) | (c & 3u)) & mask;
                    if (run >= (uint32_t)K) {  // window [p+b-K+1, p+b] is all ACGT
                        if (LDS_BINS) atomicAdd(&lds_bins[code], 1u);
                        else atomicAdd(&bins[code], 1ull);
                    }
                }
            }
        }
    }
    if (LDS_BINS) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x)
            if (lds_bins[i]) atomicAdd(&bins[i], (unsigned long long)lds_bins[i]);
    }
}

// The same census for k <= 8 from the 4-bit packed genome, with the whole histogram (or half of
// it per pass at k = 8) in LDS:
//   - a lane walks GKC4_SPAN consecutive positions = 128 packed bytes with a rolling 2-bit code,
//     eight dwordx4 loads per span;
//   - bins are REPLICATED rep times (lane & (rep-1) picks the copy, copies of a bin sit in adjacent
//     words = different banks): the AAAA / TTTT / poly-N skew of a real genome would otherwise
//     serialise the lanes of a wave on a few words;
//   - bin_lo / n_bins select the slice of the 4^k bins this pass owns (k = 8: two passes of 32 Ki
//     bins = 128 KiB of LDS each); one u64 global atomic per non-empty bin per workgroup at the end.
constexpr uint32_t GKC4_SPAN = 256;
__global__ void __launch_bounds__(512) genome_kmer_packed_kernel(const uint32_t *g4, uint64_t n_pos, int K, uint32_t bin_lo,
                                                                 uint32_t n_bins, uint32_t rep_log2,
                                                                 unsigned long long *bins) {
    extern __shared__ uint32_t lds_hist[];
    const uint32_t rep = 1u << rep_log2;
    for (uint32_t i = threadIdx.x; i < (n_bins << rep_log2); i += blockDim.x) lds_hist[i] = 0u;
    __syncthreads();
    const uint32_t mask = (1u << (2 * K)) - 1u;
    const uint32_t copy = threadIdx.x & (rep - 1u);
    const uint64_t n_spans = (n_pos + GKC4_SPAN - 1) / GKC4_SPAN;
    for (uint64_t sp = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; sp < n_spans; sp += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p0 = sp * GKC4_SPAN;
        const uint4 *src = (const uint4 *)(g4 + p0 / 8);
        uint32_t code = 0u, run = 0u;
        if (p0) {  // warm-up: the K-1 <= 7 positions before the span are in the previous dword
            const uint32_t w = g4[p0 / 8 - 1];
#pragma unroll
            for (int b = 1; b < 8; b++) {
                const uint32_t c = (w >> (4 * b)) & 15u;
                run = c < 4u ? run + 1u : 0u;
                code = (code << 2) | (c & 3u);
            }
            run = min(run, (uint32_t)(K - 1));  // windows starting before the span belong to the previous one
        }
#pragma unroll 2
        for (int q = 0; q < (int)(GKC4_SPAN / 32); q++) {
            if (p0 + 32ull * q >= n_pos) break;
            const uint4 v = src[q];
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int d = 0; d < 4; d++) {
#pragma unroll
                for (int b = 0; b < 8; b++) {
                    const uint32_t c = (w4[d] >> (4 * b)) & 15u;
                    run = c < 4u ? run + 1u : 0u;
                    code = ((code << 2) | (c & 3u)) & mask;
                    const uint32_t slot = code - bin_lo;
                    if (run >= (uint32_t)K && slot < n_bins && p0 + 32ull * q + 8u * d + b < n_pos)
                        atomicAdd(&lds_hist[(slot << rep_log2) + copy], 1u);
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_bins; i += blockDim.x) {
        uint32_t sum = 0u;
        for (uint32_t r = 0; r < rep; r++) sum += lds_hist[(i << rep_log2) + r];
        if (sum) atomicAdd(&bins[bin_lo + i], (unsigned long long)sum);
    }
}

// Upload-time genome transform: toupper() fold (init_genome stores upper case,
// fasta-genome-io.c:127; process_aln folds again, pss-bam.c:424) followed by the
// A/C/G/T <-> 0..3 byte swap of record_decode.h.  16 bytes per lane per step.
// a small table from page-locked host memory into device memory (engine.hip: the reference table)
__global__ void copy_table_kernel(uint4 *dst, const uint4 *src, uint32_t n) {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

__global__ void encode_genome_kernel(uint8_t *p, uint64_t n16) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint4 *q = (uint4 *)p;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        uint4 v = q[i];
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t o = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                uint32_t c = (w[k] >> (8 * b)) & 0xFFu;
                if (c >= 'a' && c <= 'z') c -= 32u;
                o |= enc_byte(c) << (8 * b);
            }
            w[k] = o;
        }
        q[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// The tiled kernel's 4-bit image of the stored genome (record_decode.h, TallyParams::genome4):
// one output dword = 8 consecutive positions, little-endian in nibbles.
struct CtxSets { uint32_t up[8], down[8]; };
__global__ void pack_genome4_kernel(const uint8_t *g, uint32_t *out, uint64_t n_out, CtxSets sets) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint2 *src = (const uint2 *)g;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += stride) {
        const uint2 v = src[i];
        uint32_t o = 0u;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint32_t c = ((b < 4 ? v.x : v.y) >> (8 * (b & 3))) & 0xFFu;
            const uint32_t nib = c < 4u ? c : 4u + (in_set(sets.up, c) ? 1u : 0u) + (in_set(sets.down, c) ? 2u : 0u);
            o |= nib << (4 * b);
        }
        out[i] = o;
    }
}

}  // namespace pssbam
