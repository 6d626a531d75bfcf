// pss-bam_amd/csrc/tally_kernels.h -- the gfx950 tally kernels.
//
//  tally_simple : lane-per-read, records and reference bases gathered straight from
//                 global memory, counts into an LDS table (or global atomics when the
//                 table would not fit).  Any -r N, any k.  Fallback + cross-check.
//  tally_tiled  : the production kernel for N <= 30.  Per tile of T consecutive reads:
//                   1. the tile's raw BAM bytes are streamed into LDS with 16 B/lane
//                      coalesced loads (the only bulk HBM traffic of the kernel);
//                   2. lane-per-read: decode + filters, gather the two reference end
//                      windows (s-2..s+N, s+L-N..s+L+2) into LDS, k-mer tally;
//                   3. wave-per-read, lane = table row: lanes 0..31 own the forward
//                      table's rows, lanes 32..63 the reverse table's rows; each lane
//                      forms its (read base, reference base) cell and bumps its own
//                      column of a [16][64] LDS table -- no two lanes of a wave ever
//                      touch the same word, so there is no intra-wave contention no
//                      matter how skewed the data (AA/CC/GG/TT dominate).
//                 Counters leave LDS once, at kernel end, as u64 global atomics.
//
// Integer/byte work only: no MFMA anywhere (the bound is HBM bandwidth).
#pragma once

#include "record_decode.h"

namespace pssbam {

constexpr int TILED_THREADS = 256;
constexpr int TILED_MAX_N = 30;        // 2*(N+2) rows must fit the 64 lanes of a wave
constexpr int WIN_DWORDS = 9;          // (N+2) + 3 alignment bytes <= 36
constexpr int KMER_LDS_MAX_K = 5;      // 2 * 4^5 * 4 B = 8 KiB of LDS

// ---------------------------------------------------------------------------------------
// per-read tally, lane-per-read form (used by tally_simple and for tile overflow records)
// ---------------------------------------------------------------------------------------
struct LdsTableRowMajor {  // [table][row][16] u32 in LDS
    uint32_t *t;
    uint32_t rows;
    __device__ __forceinline__ void add(uint32_t table, uint32_t row, uint32_t cell) const {
        atomicAdd(&t[(table * rows + row) * 16u + cell], 1u);
    }
};
struct LdsTableLaneMajor {  // [cell][lane] u32 in LDS, lane = table*32 + row (tiled kernel)
    uint32_t *t;
    __device__ __forceinline__ void add(uint32_t table, uint32_t row, uint32_t cell) const {
        atomicAdd(&t[cell * 64u + table * 32u + row], 1u);
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

// k-mer adds for one record; returns the stats bit (OK / FAIL) it earns.
template <bool LDS_KMER>
__device__ __forceinline__ uint32_t tally_kmer_record(const TallyParams &P, const Plan &pl, uint32_t *lds_kmer) {
    const uint8_t *G = P.genome + pl.gbase;
    int64_t w5, w3;
    kmer_windows(pl, P.K, w5, w3);
    const uint32_t nb = 1u << (2 * P.K);
    bool good = true;
    if (pl.fk5) {
        uint32_t bin;
        if (kmer_bin(G, w5, P.K, pl.rev, bin)) {
            if (LDS_KMER) atomicAdd(&lds_kmer[bin], 1u);
            else atomicAdd(&P.counters[P.off_k5 + bin], 1ull);
        } else good = false;
    }
    if (pl.fk3) {
        uint32_t bin;
        if (kmer_bin(G, w3, P.K, pl.rev, bin)) {
            if (LDS_KMER) atomicAdd(&lds_kmer[nb + bin], 1u);
            else atomicAdd(&P.counters[P.off_k3 + bin], 1ull);
        } else good = false;
    }
    return good ? (1u << ST_KMER_OK) : (1u << ST_KMER_FAIL);
}

__device__ __forceinline__ void flush_stats(const TallyParams &P, uint32_t *lds_stats) {
    for (uint32_t i = threadIdx.x; i < (uint32_t)ST_USED; i += blockDim.x)
        if (lds_stats[i]) atomicAdd(&P.counters[P.off_stats + i], (unsigned long long)lds_stats[i]);
}

// ---------------------------------------------------------------------------------------
// tally_simple
// ---------------------------------------------------------------------------------------
// dynamic LDS: [2*(N+2)*16 u32 table, if LDS_TABLE]
template <bool LDS_TABLE>
__global__ void __launch_bounds__(256) tally_simple(const TallyParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_lds[];
    __shared__ uint32_t lds_stats[ST_USED];
    const uint32_t rows = (uint32_t)P.N + 2u;
    const uint32_t tab_words = LDS_TABLE ? 2u * rows * 16u : 0u;
    for (uint32_t i = threadIdx.x; i < tab_words; i += blockDim.x) dyn_lds[i] = 0u;
    if (threadIdx.x < ST_USED) lds_stats[threadIdx.x] = 0u;
    __syncthreads();

    uint32_t my_stats[ST_USED];
#pragma unroll
    for (int i = 0; i < ST_USED; i++) my_stats[i] = 0u;

    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < P.n_recs; r += stride) {
        const uint32_t o0 = P.offs[r], o1 = P.offs[r + 1];
        GlobalBytes src{P.recs + o0};
        const RecHdr h = decode_hdr(src, o1 - o0);
        const Plan pl = make_plan(P, src, h);
        uint32_t m = pl.st_mask;
        if (pl.pss_fwd || pl.pss_rev) {
            if (LDS_TABLE) tally_pss_record(P, LdsTableRowMajor{dyn_lds, rows}, src, h, pl);
            else tally_pss_record(P, GlobalTable{P.counters, P.off_rev}, src, h, pl);
        }
        if (pl.fk5 || pl.fk3) m |= tally_kmer_record<false>(P, pl, nullptr);
#pragma unroll
        for (int i = 0; i < ST_USED; i++) my_stats[i] += (m >> i) & 1u;
    }
#pragma unroll
    for (int i = 0; i < ST_USED; i++)
        if (my_stats[i]) atomicAdd(&lds_stats[i], my_stats[i]);
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
    flush_stats(P, lds_stats);
}

// ---------------------------------------------------------------------------------------
// tally_tiled
// ---------------------------------------------------------------------------------------
// Per-read descriptor handed from phase 2 to phase 3 (16 bytes, read with one broadcast
// ds_read_b128 per wave).
struct __attribute__((aligned(16))) ReadDesc {
    uint32_t seq_off;  // offset of SEQ inside the staged tile
    uint32_t L;        // effective length
    uint32_t l_seq;    // bases really present
    uint32_t flags;    // bit0 fwd table, bit1 rev table, bit2 reverse strand, bits 8..9 left shift,
                       // bits 16..17 right shift
};

// dynamic LDS carve-up (all 16-byte aligned):
//   stage  : tile_bytes_cap + 16
//   desc   : T * 16
//   gwin   : T * 2 * WIN_DWORDS * 4
//   table  : 16 * 64 * 4
//   kmer   : 2 * 4^K * 4        (only when K <= KMER_LDS_MAX_K and the k-mer tally is on)
__host__ __device__ inline uint32_t tiled_lds_bytes(uint32_t T, uint32_t cap, bool kmer_lds, int K) {
    uint32_t b = ((cap + 16u + 15u) & ~15u) + T * 16u + T * 2u * WIN_DWORDS * 4u + 16u * 64u * 4u;
    if (kmer_lds) b += 2u * (1u << (2 * K)) * 4u;
    return b;
}

template <bool DO_PSS, bool DO_KMER, bool LDS_KMER>
__global__ void __launch_bounds__(TILED_THREADS) tally_tiled(const TallyParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    __shared__ uint32_t lds_stats[ST_USED];

    const uint32_t T = P.reads_per_tile;
    const uint32_t cap = P.tile_bytes_cap;
    uint8_t *stage = lds_raw;
    ReadDesc *desc = (ReadDesc *)(lds_raw + ((cap + 16u + 15u) & ~15u));
    uint32_t *gwin = (uint32_t *)(desc + T);
    uint32_t *table = gwin + T * 2u * WIN_DWORDS;
    uint32_t *lds_kmer = table + 16u * 64u;

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const int N = P.N;
    const uint32_t win_dw = ((uint32_t)N + 5u + 3u) >> 2;  // dwords covering (N+2) bytes at any shift

    for (uint32_t i = tid; i < 16u * 64u; i += TILED_THREADS) table[i] = 0u;
    if (LDS_KMER)
        for (uint32_t i = tid; i < 2u * (1u << (2 * P.K)); i += TILED_THREADS) lds_kmer[i] = 0u;
    if (tid < ST_USED) lds_stats[tid] = 0u;

    uint32_t my_stats[ST_USED];
#pragma unroll
    for (int i = 0; i < ST_USED; i++) my_stats[i] = 0u;

    const uint32_t n_tiles = (P.n_recs + T - 1u) / T;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t r0 = tile * T;
        const uint32_t r1 = min(r0 + T, P.n_recs);
        const uint32_t o_first = P.offs[r0], o_last = P.offs[r1];
        const uint32_t base16 = o_first & ~15u;
        const uint32_t want = o_last - base16;
        const uint32_t staged = min(want, cap);  // bytes [base16, base16+staged) are in LDS

        __syncthreads();  // previous tile fully consumed (also orders the table/kmer zeroing)
        // ---- phase 1: coalesced stream of the tile's record bytes into LDS ----------------
        {
            const uint4 *g = (const uint4 *)(P.recs + base16);
            uint4 *l = (uint4 *)stage;
            const uint32_t n16 = (staged + 15u) >> 4;
            uint32_t c = tid;
            for (; c + 3u * TILED_THREADS < n16; c += 4u * TILED_THREADS) {
                const uint4 v0 = g[c], v1 = g[c + TILED_THREADS], v2 = g[c + 2u * TILED_THREADS],
                            v3 = g[c + 3u * TILED_THREADS];
                l[c] = v0; l[c + TILED_THREADS] = v1; l[c + 2u * TILED_THREADS] = v2; l[c + 3u * TILED_THREADS] = v3;
            }
            for (; c < n16; c += TILED_THREADS) l[c] = g[c];
        }
        __syncthreads();

        // ---- phase 2: lane-per-read decode, filters, reference windows, k-mers -----------
        for (uint32_t j = tid; j < r1 - r0; j += TILED_THREADS) {
            const uint32_t r = r0 + j;
            const uint32_t o0 = P.offs[r], o1 = P.offs[r + 1];
            ReadDesc d;
            d.seq_off = 0; d.L = 0; d.l_seq = 0; d.flags = 0;
            uint32_t m;
            if (o1 - base16 <= staged) {
                LdsBytes src{stage + (o0 - base16)};
                const RecHdr h = decode_hdr(src, o1 - o0);
                const Plan pl = make_plan(P, src, h);
                m = pl.st_mask;
                if (DO_PSS && (pl.pss_fwd || pl.pss_rev)) {
                    const uint64_t ga = pl.gbase + (uint64_t)pl.s - 2u;              // left window start
                    const uint64_t gb = pl.gbase + (uint64_t)pl.s + pl.L - (uint32_t)N;  // right window start
                    const uint32_t *pa = (const uint32_t *)(P.genome + (ga & ~3ull));
                    const uint32_t *pb = (const uint32_t *)(P.genome + (gb & ~3ull));
                    uint32_t *wl = gwin + (j * 2u) * WIN_DWORDS, *wr = wl + WIN_DWORDS;
                    for (uint32_t k = 0; k < win_dw; k++) { wl[k] = pa[k]; wr[k] = pb[k]; }
                    d.seq_off = (o0 - base16) + h.seq_off;
                    d.L = pl.L;
                    d.l_seq = h.l_seq;
                    d.flags = (pl.pss_fwd ? 1u : 0u) | (pl.pss_rev ? 2u : 0u) | (pl.rev ? 4u : 0u) |
                              ((uint32_t)(ga & 3ull) << 8) | ((uint32_t)(gb & 3ull) << 16);
                }
                if (DO_KMER && (pl.fk5 || pl.fk3)) m |= tally_kmer_record<LDS_KMER>(P, pl, lds_kmer);
            } else {
                // record does not fit the staging window (huge record): whole thing from global
                GlobalBytes src{P.recs + o0};
                const RecHdr h = decode_hdr(src, o1 - o0);
                const Plan pl = make_plan(P, src, h);
                m = pl.st_mask;
                if (DO_PSS && (pl.pss_fwd || pl.pss_rev)) tally_pss_record(P, LdsTableLaneMajor{table}, src, h, pl);
                if (DO_KMER && (pl.fk5 || pl.fk3)) m |= tally_kmer_record<LDS_KMER>(P, pl, lds_kmer);
            }
            desc[j] = d;
#pragma unroll
            for (int i = 0; i < ST_USED; i++) my_stats[i] += (m >> i) & 1u;
        }
        __syncthreads();

        // ---- phase 3: wave-per-read, lane = table row --------------------------------------
        if (DO_PSS) {
            const uint32_t t = lane >> 5;       // 0 = forward table, 1 = reverse table
            const uint32_t row = lane & 31u;    // 0,1 context rows; 2+i = position i
            const bool row_live = row < (uint32_t)N + 2u;
            for (uint32_t j = wave; j < r1 - r0; j += TILED_THREADS / 64) {
                const ReadDesc d = desc[j];
                if (!((d.flags >> t) & 1u) || !row_live) continue;
                const bool rev = (d.flags & 4u) != 0;
                const bool left = (t == 0u) != rev;
                const uint8_t *w = (const uint8_t *)(gwin + (j * 2u + (left ? 0u : 1u)) * WIN_DWORDS);
                const uint32_t shift = left ? ((d.flags >> 8) & 3u) : ((d.flags >> 16) & 3u);
                // left window byte k <-> reference s-2+k; right window byte k <-> s+L-N+k
                const uint32_t rf = ref_code(w[shift + (left ? row : (uint32_t)N + 1u - row)]);
                uint32_t cell;
                bool ok = rf < 4u;
                if (row < 2u) {
                    cell = 5u * rf;
                } else {
                    const uint32_t ri = left ? row - 2u : d.L + 1u - row;
                    uint32_t nib = 0u;
                    if (ri < d.l_seq) {
                        const uint32_t b = stage[d.seq_off + (ri >> 1)];
                        nib = (ri & 1u) ? (b & 0xFu) : (b >> 4);
                    }
                    const uint32_t rd = nib_code(nib);
                    ok = ok && rd < 4u;
                    cell = 4u * rd + rf;
                }
                if (ok) atomicAdd(&table[(rev ? 15u - cell : cell) * 64u + lane], 1u);
            }
        }
    }

#pragma unroll
    for (int i = 0; i < ST_USED; i++)
        if (my_stats[i]) atomicAdd(&lds_stats[i], my_stats[i]);
    __syncthreads();
    if (DO_PSS) {
        for (uint32_t i = tid; i < 16u * 64u; i += TILED_THREADS) {
            const uint32_t v = table[i];
            const uint32_t cell = i >> 6, ln = i & 63u, t = ln >> 5, row = ln & 31u;
            if (v && row < (uint32_t)N + 2u)
                atomicAdd(&P.counters[(t ? P.off_rev : 0u) + row * 16u + cell], (unsigned long long)v);
        }
    }
    if (LDS_KMER) {
        const uint32_t nb = 1u << (2 * P.K);
        for (uint32_t i = tid; i < 2u * nb; i += TILED_THREADS) {
            const uint32_t v = lds_kmer[i];
            if (v) atomicAdd(&P.counters[(i < nb ? P.off_k5 + i : P.off_k3 + (i - nb))], (unsigned long long)v);
        }
    }
    flush_stats(P, lds_stats);
}

// upper-cases a-z in place: init_genome stores toupper()ed bases (fasta-genome-io.c:127)
// and process_aln folds again (pss-bam.c:424); callers handing raw arrays get the same.
__global__ void upcase_kernel(uint8_t *p, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint8_t c = p[i];
        if (c >= 'a' && c <= 'z') p[i] = c - 32;
    }
}

}  // namespace pssbam
