// pss-bam_amd/csrc/synth.hip -- libpssbam_synth.so: materialises the synthetic workload
// of synth_model.h on the device (bench.py: straight into HBM) and on the host (tests,
// CPU-baseline sample: BAM records, SAM text, FASTA).  Workload infrastructure only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include <zlib.h>

#include "synth_model.h"

// ---------------------------------------------------------------------------------------
// device kernels
// ---------------------------------------------------------------------------------------
__global__ void k_genome(const synth_cfg c, uint32_t contig, uint8_t *out, uint64_t len) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 4u;
    for (uint64_t p = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4u; p < len; p += stride) {
        uint32_t w = 0;
        for (uint32_t k = 0; k < 4 && p + k < len; k++) {
            const uint32_t b = syn_ref_code(&c, contig, p + k);
            w |= (uint32_t)(b < 4u ? "ACGT"[b] : 'N') << (8u * k);
        }
        if (p + 4 <= len) *(uint32_t *)(out + p) = w;  // out is 4-byte aligned (hipMalloc / torch)
        else for (uint32_t k = 0; p + k < len; k++) out[p + k] = (uint8_t)(w >> (8u * k));
    }
}

__global__ void k_records(const synth_cfg c, uint64_t slot0, uint64_t n, const uint32_t *offs, uint8_t *out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
        const uint64_t idx = syn_read_index(&c, slot0 + t);
        synth_read r;
        syn_read_fields(&c, idx, &r);
        syn_write_record(&c, idx, &r, out + offs[t]);
    }
}

__global__ void k_offsets_linear(uint32_t *offs, uint64_t n_plus_1, uint32_t rec_bytes) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_plus_1; t += stride)
        offs[t] = (uint32_t)(t * rec_bytes);
}

// ---------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------
template <class F>
static void parallel_for(uint64_t n, int threads, F f, uint64_t min_n = 4096) {
    if (threads <= 1 || n < min_n) { f(0, n); return; }
    std::vector<std::thread> th;
    const uint64_t per = (n + threads - 1) / threads;
    for (int t = 0; t < threads; t++) {
        const uint64_t a = std::min<uint64_t>(n, per * t), b = std::min<uint64_t>(n, a + per);
        if (a < b) th.emplace_back([=] { f(a, b); });
    }
    for (auto &x : th) x.join();
}

static void contig_name(const synth_cfg *c, uint32_t k, char buf[16]) {
    if (c->name_mode == 0) snprintf(buf, 16, k == 0 ? "chrS" : "chrS%u", k);
    else if (k < 22) snprintf(buf, 16, "chr%u", k + 1);
    else if (k == 22) snprintf(buf, 16, "chrX");
    else if (k == 23) snprintf(buf, 16, "chrY");
    else snprintf(buf, 16, "chrUn%u", k);
}

extern "C" {

int synth_cfg_finish(synth_cfg *c) {
    if (!c || c->n_contigs == 0 || c->n_contigs > SYN_MAX_CONTIGS || c->n_reads == 0 || c->len_min == 0 ||
        c->len_max < c->len_min)
        return -1;
    c->usable_first[0] = 0;
    for (uint32_t k = 0; k < c->n_contigs; k++) {
        if (c->contig_len[k] < (uint64_t)c->len_max + 5) return -1;
        c->usable_first[k + 1] = c->usable_first[k] + (c->contig_len[k] - c->len_max - 4);
    }
    uint64_t mul = 0x9E3779B1ull % c->n_reads;
    if (mul < 2) mul = 1;
    while (std::gcd(mul, c->n_reads) != 1) mul++;
    c->perm_mul = mul;
    c->perm_add = syn_mix(c->seed ^ 0x7065726Dull) % c->n_reads;
    return 0;
}

void synth_contig_name(const synth_cfg *c, uint32_t k, char *buf16) { contig_name(c, k, buf16); }

// contig bytes [p0, p0+n) in loaded form (upper case) or, with fasta_case, as FASTA text
// would hold them (soft-masked lower case)
int synth_genome_host(const synth_cfg *c, uint32_t contig, uint8_t *out, uint64_t p0, uint64_t n, int fasta_case,
                      int threads) {
    parallel_for(n, threads, [=](uint64_t a, uint64_t b) {
        for (uint64_t i = a; i < b; i++) {
            const uint32_t code = syn_ref_code(c, contig, p0 + i);
            uint8_t ch = code < 4u ? (uint8_t)"ACGT"[code] : (uint8_t)'N';
            if (fasta_case && syn_ref_is_lower(c, contig, p0 + i)) ch |= 0x20;
            out[i] = ch;
        }
    });
    return 0;
}

int synth_genome_device(const synth_cfg *c, uint32_t contig, uint8_t *d_out, uint64_t len, void *stream) {
    hipLaunchKernelGGL(k_genome, dim3(4096), dim3(256), 0, (hipStream_t)stream, *c, contig, d_out, len);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int synth_sizes_host(const synth_cfg *c, uint64_t slot0, uint64_t n, uint32_t *sizes, int threads) {
    parallel_for(n, threads, [=](uint64_t a, uint64_t b) {
        for (uint64_t t = a; t < b; t++) {
            synth_read r;
            syn_read_fields(c, syn_read_index(c, slot0 + t), &r);
            sizes[t] = r.rec_bytes;
        }
    });
    return 0;
}

int synth_records_host(const synth_cfg *c, uint64_t slot0, uint64_t n, const uint32_t *offs, uint8_t *out,
                       int threads) {
    parallel_for(n, threads, [=](uint64_t a, uint64_t b) {
        for (uint64_t t = a; t < b; t++) {
            const uint64_t idx = syn_read_index(c, slot0 + t);
            synth_read r;
            syn_read_fields(c, idx, &r);
            syn_write_record(c, idx, &r, out + offs[t]);
        }
    });
    return 0;
}

int synth_records_device(const synth_cfg *c, uint64_t slot0, uint64_t n, const uint32_t *d_offs, uint8_t *d_out,
                         void *stream) {
    if (!n) return 0;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 65536);
    hipLaunchKernelGGL(k_records, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *c, slot0, n, d_offs, d_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int synth_offsets_linear_device(uint32_t *d_offs, uint64_t n_plus_1, uint32_t rec_bytes, void *stream) {
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_plus_1 + 255) / 256, 16384);
    hipLaunchKernelGGL(k_offsets_linear, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_offs, n_plus_1, rec_bytes);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// SAM text of slots [slot0, slot0+n), written from the model's fields directly (NOT by
// decoding the BAM bytes): the independent text twin the reference binary is fed with.
int synth_sam_host(const synth_cfg *c, uint64_t slot0, uint64_t n, const char *path, int with_header) {
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    std::vector<char> iobuf(1 << 20);  // per call: the writers may run concurrently
    setvbuf(f, iobuf.data(), _IOFBF, iobuf.size());
    char nm[16];
    if (with_header) {
        fputs("@HD\tVN:1.6\tSO:unknown\n", f);
        for (uint32_t k = 0; k < c->n_contigs; k++) {
            contig_name(c, k, nm);
            fprintf(f, "@SQ\tSN:%s\tLN:%llu\n", nm, (unsigned long long)c->contig_len[k]);
        }
    }
    std::string seq, qual;
    for (uint64_t t = 0; t < n; t++) {
        const uint64_t idx = syn_read_index(c, slot0 + t);
        synth_read r;
        syn_read_fields(c, idx, &r);
        contig_name(c, r.contig, nm);
        char cig[64];
        int o = 0;
        for (uint32_t k = 0; k < r.n_cigar; k++)
            o += snprintf(cig + o, sizeof cig - o, "%u%c", r.cigar[k] >> 4, "MIDNSHP=X"[r.cigar[k] & 15u]);
        seq.resize(r.L);
        qual.assign(r.L, 'I');
        for (uint32_t j = 0; j < r.L; j++) {
            const uint32_t b = syn_read_code(c, idx, &r, j);
            seq[j] = b < 4u ? "ACGT"[b] : 'N';
        }
        fprintf(f, "s%010llu\t%u\t%s\t%llu\t%u\t%s\t*\t0\t0\t%s\t%s\n", (unsigned long long)idx, r.flag, nm,
                (unsigned long long)(r.s + 1), r.mapq, cig, seq.c_str(), qual.c_str());
    }
    return fclose(f) == 0 ? 0 : -1;
}

// FASTA text (soft-masked) of contigs [first, first+count)
int synth_fasta_host(const synth_cfg *c, const char *path, uint32_t first, uint32_t count, uint32_t width,
                     int threads) {
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    std::vector<char> iobuf(1 << 20);  // per call: the writers may run concurrently
    setvbuf(f, iobuf.data(), _IOFBF, iobuf.size());
    const uint64_t CH = 1u << 24;
    std::vector<uint8_t> buf(CH);
    std::vector<char> line((size_t)CH + CH / width + 2);
    char nm[16];
    for (uint32_t k = first; k < first + count && k < c->n_contigs; k++) {
        contig_name(c, k, nm);
        fprintf(f, ">%s synthetic length=%llu\n", nm, (unsigned long long)c->contig_len[k]);
        uint64_t col = 0;
        for (uint64_t p0 = 0; p0 < c->contig_len[k]; p0 += CH) {
            const uint64_t n = std::min<uint64_t>(CH, c->contig_len[k] - p0);
            synth_genome_host(c, k, buf.data(), p0, n, 1, threads);
            size_t w = 0;
            for (uint64_t i = 0; i < n; i++) {
                line[w++] = (char)buf[i];
                if (++col == width) { line[w++] = '\n'; col = 0; }
            }
            fwrite(line.data(), 1, w, f);
        }
        if (col) fputc('\n', f);
    }
    return fclose(f) == 0 ? 0 : -1;
}

// A complete BGZF-compressed BAM file holding slots [slot0, slot0+n) (header with the contig
// list, then the records), deflated by `threads` workers.  level 0..9 (1 = fast).  This is the
// on-disk form the front ends consume; used by the end-to-end measurements and CLI tests.
// layout 0: BGZF blocks the way htslib writes them -- the header in its own block(s), then whole
// records per block (a block is closed when the next record would not fit 0xff00 bytes; bam_write1
// calls bgzf_flush_try), so every block starts on a record boundary.  layout bit 0: the stream cut
// every 0xff00 bytes regardless of records (what htsjdk-style writers produce).
int synth_bam_file_host(const synth_cfg *c, uint64_t slot0, uint64_t n, const char *path, int level, int threads,
                        int layout) {
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    // header bytes
    std::string head;
    {
        std::string text = "@HD\tVN:1.6\tSO:" + std::string(c->sorted ? "coordinate" : "unsorted") + "\n";
        char nm[16];
        for (uint32_t k = 0; k < c->n_contigs; k++) {
            contig_name(c, k, nm);
            text += "@SQ\tSN:" + std::string(nm) + "\tLN:" + std::to_string(c->contig_len[k]) + "\n";
        }
        auto put32 = [&](uint32_t v) { for (int i = 0; i < 4; i++) head.push_back((char)(v >> (8 * i))); };
        head += "BAM\1";
        put32((uint32_t)text.size());
        head += text;
        put32(c->n_contigs);
        for (uint32_t k = 0; k < c->n_contigs; k++) {
            contig_name(c, k, nm);
            put32((uint32_t)strlen(nm) + 1);
            head.append(nm, strlen(nm) + 1);
            put32((uint32_t)c->contig_len[k]);
        }
    }
    const uint64_t CHUNK_READS = 1u << 20;  // records generated + compressed per round
    const size_t BLK = 0xFF00;
    std::vector<uint8_t> raw, carry(head.begin(), head.end());
    std::vector<uint32_t> sizes, offs;
    std::vector<size_t> cut;  // block boundaries within raw: block b = [cut[b], cut[b+1])
    int rc = 0;
    for (uint64_t a = 0; a < n || !carry.empty(); a += CHUNK_READS) {
        const uint64_t m = a < n ? std::min<uint64_t>(CHUNK_READS, n - a) : 0;
        sizes.resize(m);
        offs.resize(m + 1);
        if (m) synth_sizes_host(c, slot0 + a, m, sizes.data(), threads);
        uint64_t tot = 0;
        for (uint64_t i = 0; i < m; i++) { offs[i] = (uint32_t)tot; tot += sizes[i]; }
        offs[m] = (uint32_t)tot;
        raw.resize(carry.size() + tot);
        std::copy(carry.begin(), carry.end(), raw.begin());
        if (m) synth_records_host(c, slot0 + a, m, offs.data(), raw.data() + carry.size(), threads);
        if (m && (layout & 6)) {
            // QUAL of the named configurations is constant ("QUAL all I"), which DEFLATE turns into
            // one long match per read; these two modes give the inflate something closer to a sequencer's
            // output to chew on (the tally never looks past QUAL[0]): 2 = four quality bins (NovaSeq-style:
            // 37 / 25 / 11 / 2 with probability .80 / .12 / .06 / .02), 4 = 40 levels, skewed to the top
            uint8_t *base = raw.data() + carry.size();
            const uint32_t *of = offs.data();
            const bool bins = (layout & 2) != 0;
            const uint64_t seed = c->seed * 0x9E3779B97F4A7C15ull + (slot0 + a);
            parallel_for(m, threads, [=](uint64_t r0, uint64_t r1) {
                for (uint64_t i = r0; i < r1; i++) {
                    uint8_t *rec = base + of[i];
                    uint32_t l_seq, w3, w4;
                    memcpy(&w3, rec + 12, 4); memcpy(&w4, rec + 16, 4); memcpy(&l_seq, rec + 20, 4);
                    uint8_t *q = rec + 36 + (w3 & 0xFFu) + 4u * (w4 & 0xFFFFu) + (l_seq + 1u) / 2u;
                    uint64_t x = seed + i * 0xD1342543DE82EF95ull;
                    for (uint32_t k = 0; k < l_seq; k++) {
                        x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 29; x += 0x9E3779B97F4A7C15ull;   // a new word per base
                        const uint32_t u = (uint32_t)(x >> 40) & 0xFFFFu;
                        if (bins) q[k] = u < 52429u ? 37 : u < 60293u ? 25 : u < 64225u ? 11 : 2;
                        else {
                            const uint32_t v = (uint32_t)(x >> 20) & 0xFFFFu;
                            const uint32_t lo = u < v ? u : v;          // min of two uniforms: density falls linearly
                            q[k] = (uint8_t)(40u - lo * 39u / 65536u);  // 40 most likely ... 2 least
                        }
                    }
                }
            });
        }
        const bool last = a + m >= n;
        cut.clear();
        cut.push_back(0);
        if ((layout & 1) == 0) {
            // carry = the BAM header (first round only; flushed as its own blocks), never records:
            // in this layout every round ends on a record boundary
            const size_t lead = carry.size();
            for (size_t o = 0; o < lead;) { o = std::min(lead, o + BLK); cut.push_back(o); }
            size_t blk_start = lead;
            for (uint64_t i = 0; i < m; i++) {
                const size_t r0 = lead + offs[i], r1 = lead + offs[i + 1];
                if (r1 - blk_start > BLK && r0 > blk_start) { cut.push_back(r0); blk_start = r0; }  // flush before a record that does not fit
                while (r1 - blk_start > BLK) { blk_start += BLK; cut.push_back(blk_start); }        // a record larger than a block is split
            }
            if (raw.size() > blk_start) cut.push_back(raw.size());
        } else {
            // whole 0xff00-byte blocks now, remainder carried (flushed on the last round)
            const size_t n_whole = last ? (raw.size() + BLK - 1) / BLK : raw.size() / BLK;
            for (size_t b = 1; b <= n_whole; b++) cut.push_back(std::min(raw.size(), b * BLK));
        }
        const size_t n_blk = cut.size() - 1;
        std::vector<std::vector<uint8_t>> out(n_blk);
        parallel_for(n_blk, threads, [&](uint64_t b0, uint64_t b1) {
            z_stream zs;   // one deflate state per worker, reset per block (deflateInit2 clears ~260 KB each time)
            memset(&zs, 0, sizeof zs);
            deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
            for (uint64_t b = b0; b < b1; b++) {
                const size_t o = cut[b], len = cut[b + 1] - cut[b];
                deflateReset(&zs);
                std::vector<uint8_t> &dst = out[b];
                dst.resize(18 + deflateBound(&zs, len) + 8);
                zs.next_in = raw.data() + o;
                zs.avail_in = (uInt)len;
                zs.next_out = dst.data() + 18;
                zs.avail_out = (uInt)(dst.size() - 26);
                deflate(&zs, Z_FINISH);
                const size_t clen = zs.total_out;
                const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
                memcpy(dst.data(), hdr, 16);
                const uint32_t bsize = (uint32_t)(clen + 25), crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), raw.data() + o, (uInt)len);
                dst[16] = (uint8_t)bsize; dst[17] = (uint8_t)(bsize >> 8);
                uint8_t *t = dst.data() + 18 + clen;
                for (int i = 0; i < 4; i++) { t[i] = (uint8_t)(crc >> (8 * i)); t[4 + i] = (uint8_t)((uint32_t)len >> (8 * i)); }
                dst.resize(18 + clen + 8);
            }
            deflateEnd(&zs);
        }, 64);   // a block is ~64 KiB of deflate work: worth a thread from a few dozen on
        for (auto &d : out)
            if (fwrite(d.data(), 1, d.size(), f) != d.size()) rc = -1;
        carry.assign(raw.begin() + cut.back(), raw.end());
        if (last) break;
    }
    static const uint8_t eof_blk[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (fwrite(eof_blk, 1, 28, f) != 28) rc = -1;
    if (fclose(f) != 0) rc = -1;
    return rc;
}

}  // extern "C"
