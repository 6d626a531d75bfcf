/*
 * include/pssbam_hip.h -- C ABI of the MI355X (gfx950) tally engine.
 *
 * This is the drop-in boundary for pss-bam's per-read hot path.  The reference has no
 * FFI seam; the loop it replaces is
 *
 *     while (fgets(line)) { line2saml(line, sp); process_aln(fwd, rev, genome, sp); }
 *         /root/reference/pss-bam.c:764-783   (and fragkon.c:342-363 for the k-mer tool)
 *
 * i.e. "decode one alignment, filter it, tally it".  A caller now hands *blocks of raw
 * BAM alignment records* (exactly the bytes between two record boundaries of an
 * inflated BAM stream) to pssbam_engine_submit*, and collects the same two
 * unsigned long[(N+2)][16] tables (pss-bam.c:24-35, :755-756) and/or the two k-mer
 * tables (fragkon.c:335-336) from pssbam_engine_finish.
 *
 * Plain C types only.  Every function returns 0 on success and a negative PSSBAM_E*
 * code on failure; pssbam_last_error() then describes it.  The library never calls
 * exit() and never falls back to a CPU implementation: without a usable gfx950 device
 * pssbam_engine_create fails with PSSBAM_ENODEV.
 *
 * Threading: an engine is owned by one host thread at a time; different engines
 * (e.g. one per GPU) are independent.  All work is issued on one HIP stream per engine.
 */
#ifndef PSSBAM_HIP_H
#define PSSBAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSSBAM_ABI_VERSION 1
#define PSSBAM_MAX_KLEN 15     /* 4^15 64-bit bins per k-mer table = 8.6 GB of device memory */

/* error codes */
#define PSSBAM_OK 0
#define PSSBAM_EINVAL (-1)   /* bad argument / option outside the supported range        */
#define PSSBAM_ENODEV (-2)   /* no gfx950 device, or the HIP runtime refused to start    */
#define PSSBAM_EHIP (-3)     /* a HIP call failed (message has the HIP error string)     */
#define PSSBAM_ENOMEM (-4)
#define PSSBAM_ESTATE (-5)   /* call order violated (e.g. submit before set_genome)      */
#define PSSBAM_EFORMAT (-6)  /* malformed record block                                   */
#define PSSBAM_EBUSY (-7)    /* pssbam_engine_submit_bgzf before the genome: every feed slot is full; set the genome first */

/* which tallies one pass produces */
#define PSSBAM_TALLY_PSS 1u   /* substitution tables, pss-bam.c process_aln              */
#define PSSBAM_TALLY_KMER 2u  /* fragmentation-point k-mers, fragkon.c process_aln       */

/* kernel selection (diagnostics / tests; 0 lets the engine choose) */
#define PSSBAM_KERNEL_AUTO 0
#define PSSBAM_KERNEL_SIMPLE 1  /* lane-per-read, global gathers: the cross-check kernel  */
#define PSSBAM_KERNEL_TILED 2   /* LDS-staged record prefixes, lane=row tally; 32 table   */
                                /* rows per pass over the block (what AUTO picks)         */

/* pss-bam's option globals, /root/reference/pss-bam.c:12-18 (set by -r -l -L -q -U -D -m) */
typedef struct pssbam_pss_opts {
    int32_t region_len;        /* REGION_LEN, >= 0                                       */
    uint64_t min_read_len;     /* MIN_READ_LEN                                           */
    uint64_t max_read_len;     /* MAX_READ_LEN                                           */
    int32_t min_mq;            /* MIN_MQ (compared unsigned, as the reference does)      */
    const char *up_ctx;        /* UP_CTX: set of allowed first upstream bases            */
    const char *down_ctx;      /* DOWN_CTX                                               */
    int32_t merged_only;       /* MERGED_ONLY                                            */
} pssbam_pss_opts;

/* fragkon's option globals, /root/reference/fragkon.c:14-18 (set by -k -l -L -q -m) */
typedef struct pssbam_kmer_opts {
    int32_t klen;              /* KLEN, 1..PSSBAM_MAX_KLEN on the device                 */
    int32_t min_mq;
    uint64_t min_read_len;
    uint64_t max_read_len;
    int32_t merged_only;
} pssbam_kmer_opts;

typedef struct pssbam_config {
    uint32_t abi_version;      /* PSSBAM_ABI_VERSION                                     */
    uint32_t tally_mask;       /* PSSBAM_TALLY_*                                         */
    pssbam_pss_opts pss;       /* read when PSSBAM_TALLY_PSS is set                      */
    pssbam_kmer_opts kmer;     /* read when PSSBAM_TALLY_KMER is set                     */
    const char *read_group;    /* -R: keep only records with RG:Z:<this>; NULL = all
                                  (replaces `samtools view -r`, pss-bam.c:150-155)       */
    int32_t device;            /* HIP device ordinal, -1 = current device                */
    int32_t kernel;            /* PSSBAM_KERNEL_*                                        */
} pssbam_config;

typedef struct pssbam_engine pssbam_engine; /* opaque */
struct genome;                              /* Genome of fasta-genome-io.h               */

/* indices into the stats[] array of pssbam_engine_finish */
enum {
    PSSBAM_ST_RECORDS = 0,     /* records submitted                                      */
    PSSBAM_ST_RG_DROPPED = 1,  /* removed by the -R filter (never reach line2saml)       */
    PSSBAM_ST_PARSE_SKIP = 2,  /* line2saml would return 1 (SEQ/QUAL length mismatch)    */
    PSSBAM_ST_NO_CONTIG = 3,   /* find_seq fails: process_aln returns 1                  */
    PSSBAM_ST_PSS_OK = 4,      /* pss process_aln returns 0                              */
    PSSBAM_ST_PSS_FILTERED = 5,/* pss process_aln returns -1                             */
    PSSBAM_ST_KMER_OK = 6,     /* fragkon process_aln returns 0                          */
    PSSBAM_ST_KMER_FILTERED = 7,/* fragkon process_aln returns 2                         */
    PSSBAM_ST_KMER_FAIL = 8,   /* fragkon process_aln returns -1 (non-ACGT in a k-mer)   */
    PSSBAM_ST_SLOW_PATH = 9,   /* diagnostics: records the tiled kernel had to read from
                                  global memory (needed prefix larger than what it stages) */
    PSSBAM_ST_N = 16
};

const char *pssbam_last_error(void);      /* thread-local, never NULL                    */
int pssbam_device_count(void);            /* number of usable gfx950 devices, 0 if none  */
/* Optional: brings the HIP runtime and the device's context up (tens of ms) so that a caller
 * can overlap it with its own start-up work, e.g. from a helper thread while the FASTA loads
 * (nothing in the reference corresponds; pssbam_engine_create does it otherwise). */
int pssbam_warmup(int device);

int pssbam_engine_create(const pssbam_config *cfg, pssbam_engine **out);
void pssbam_engine_destroy(pssbam_engine *e);

/* Adopt an existing hipStream_t (e.g. torch's current stream) instead of the engine's
 * own.  Call before any submit. */
int pssbam_engine_set_stream(pssbam_engine *e, void *hip_stream);

/* Uploads the reference bases (1 byte per base, upper case as loaded, each contig
 * followed by zero padding) and remembers the sorted id table so contig lookup has
 * find_seq's strcmp semantics (fasta-genome-io.c:202-219).  The Genome stays owned by
 * the caller and may be destroyed afterwards. */
int pssbam_engine_set_genome(pssbam_engine *e, const struct genome *g);
/* The same without the wait: returns once the upload is enqueued (copies, case folding and 4-bit packing
 * run on a stream of their own; tally launches wait for them on the device), so one host thread can start
 * the uploads of several GPUs at once and the engine's stream keeps inflating meanwhile.  The Genome must
 * stay untouched until pssbam_engine_genome_wait, _sync or _finish has returned.  The reference loads the
 * genome, then loops (pss-bam.c:751-783); this is what lets the replacement overlap the two.
 * The one exception to "one thread per engine": after pssbam_engine_feed_open, set_genome_async may be called
 * from ANOTHER thread while the owning thread keeps submitting compressed blocks, provided set_references
 * follows on the owning thread after set_genome_async has returned. */
int pssbam_engine_set_genome_async(pssbam_engine *e, const struct genome *g);
int pssbam_engine_genome_wait(pssbam_engine *e);

/* Same from plain arrays; when seqs_on_device != 0 the seqs[i] are device pointers
 * (bench / generators) and are copied device-to-device. */
int pssbam_engine_set_genome_arrays(pssbam_engine *e, size_t n, const char *const *ids,
                                    const uint8_t *const *seqs, const uint64_t *lens,
                                    int seqs_on_device);

/* The BAM header's reference list, in refID order.  Each name is looked up in the
 * genome exactly like find_seq(genome, RNAME) would be for that record's text form. */
int pssbam_engine_set_references(pssbam_engine *e, int32_t n_ref, const char *const *names);

/* One block of whole BAM alignment records (each = le32 block_size + block_size
 * bytes), nbytes < 4 GiB.  offsets[i] is the byte offset of record i's block_size
 * word, offsets[n_records] == nbytes.  Host memory; the engine has copied what it
 * needs when the call returns (the copy and the kernel run asynchronously). */
int pssbam_engine_submit(pssbam_engine *e, const void *records, uint64_t nbytes,
                         const uint32_t *offsets, uint32_t n_records);

/* The same without the wait: returns as soon as the copy and the kernel are enqueued, so that one
 * host thread can keep the PCIe links of several GPUs busy at once (one engine per GPU).  The
 * caller's buffers must stay untouched until pssbam_engine_wait_copied(e, *ticket) has returned
 * (or pssbam_engine_copy_done says 1, or the engine has been synced).  *ticket == 0: nothing was
 * in flight (empty block).  Replaces nothing in the reference (its loop is synchronous,
 * pss-bam.c:764-783); pssbam_engine_submit == submit_async + wait_copied. */
int pssbam_engine_submit_async(pssbam_engine *e, const void *records, uint64_t nbytes,
                               const uint32_t *offsets, uint32_t n_records, uint64_t *ticket);
int pssbam_engine_wait_copied(pssbam_engine *e, uint64_t ticket);
int pssbam_engine_copy_done(pssbam_engine *e, uint64_t ticket);   /* 1 done, 0 in flight, < 0 error */

/* Same with both arrays already resident in device memory; nothing is copied and the
 * buffers must stay valid until pssbam_engine_sync / finish.  d_records must be 16-byte
 * aligned and readable up to nbytes rounded up to 16 (any hipMalloc / torch allocation is). */
int pssbam_engine_submit_device(pssbam_engine *e, const void *d_records, uint64_t nbytes,
                                const uint32_t *d_offsets, uint32_t n_records);

int pssbam_engine_sync(pssbam_engine *e);

/* Drains the stream and copies the accumulated tables out.  Any pointer may be NULL.
 *   fwd, rev : (region_len+2)*16 each; row 0/1 = 2nd/1st context base, row 2+i = position i
 *   k5, k3   : 4^klen each (64-bit; the fragkon front end clamps to UINT_MAX on print,
 *              kmer.c:102-104)
 * Tables keep accumulating across calls until pssbam_engine_reset. */
int pssbam_engine_finish(pssbam_engine *e, unsigned long *fwd, unsigned long *rev, uint64_t *k5,
                         uint64_t *k3, uint64_t stats[PSSBAM_ST_N]);
int pssbam_engine_reset(pssbam_engine *e);

/* The device-resident counter block [fwd | rev | k5 | k3 | stats] as one array of
 * n_u64 64-bit words, for a caller-side RCCL reduce across GPUs (sum, uint64). */
int pssbam_engine_counters_device(pssbam_engine *e, void **d_counters, size_t *n_u64);

/* Makes the engine accumulate into caller-owned device memory (n_u64 words, as reported
 * by pssbam_engine_counters_device, 8-byte aligned, zeroed by the caller) -- e.g. a
 * torch tensor that is then handed to torch.distributed / RCCL in place.  The current
 * counts are carried over.  NULL returns to the engine's own block. */
int pssbam_engine_bind_counters(pssbam_engine *e, void *d_counters, size_t n_u64);

/* genome-kmer-count (/root/reference/genome-kmer-count.c:56-79) on the uploaded genome: counts
 * every k-mer start of every contig (windows touching a non-ACGT base are not counted), k in
 * 1..PSSBAM_MAX_KLEN.  counts[4^k] in the same bin order as the fragkon tables.  Needs set_genome only. */
int pssbam_engine_genome_kmer_count(pssbam_engine *e, int klen, uint64_t *counts);

/* Node-level sum for one process driving several GPUs (one engine per device): adds the
 * counter blocks of engines[1..n-1] into engines[root] with ONE RCCL ncclReduce(sum,
 * uint64) per device inside a group call over xGMI (communicators from ncclCommInitAll,
 * librccl loaded on first use), after draining every engine's stream.  All engines must
 * have been created with identical options.  n == 1 is a no-op.  Blocks below 32 MiB (the pss
 * tables are 7 KB) are summed through the host instead -- n small copies and an add take
 * microseconds, a communicator over 8 GPUs takes seconds -- unless PSSBAM_REDUCE=rccl;
 * PSSBAM_REDUCE=host forces the host sum for any size. */
int pssbam_reduce_counters(pssbam_engine *const *engines, int n, int root);

/* Page-locks a host range (hipHostRegister) so pssbam_engine_submit's copies from it run as
 * true async DMA; the front ends register the BAM reader's batch buffer once. */
int pssbam_host_register(void *ptr, size_t bytes);
int pssbam_host_unregister(void *ptr);

/* HIP-event stopwatch on the engine's stream: begin records an event, end records a
 * second one, waits for it and returns the elapsed device time in milliseconds. */
int pssbam_engine_timer_begin(pssbam_engine *e);
int pssbam_engine_timer_end(pssbam_engine *e, float *ms);
/* Sum of the tally kernels' own durations (event pair around every launch) and their
 * number since the last call with reset != 0. */
int pssbam_engine_kernel_time(pssbam_engine *e, double *total_ms, uint64_t *n_launches, int reset);

/* Drains the engine and reports where its device time went: summed H2D copy durations (events on
 * the copy stream) and bytes, summed tally-kernel durations and launches.  Any pointer may be NULL. */
int pssbam_engine_phase_times(pssbam_engine *e, double *h2d_ms, uint64_t *h2d_bytes, double *kernel_ms,
                              uint64_t *n_launches);

/* ---- device-side BGZF inflate (second half) --------------------------------------
 * What it replaces: the `samtools view` child that decompresses the BAM for the reference
 * (/root/reference/pss-bam.c:148-162) -- here the compressed file crosses PCIe and every BGZF
 * block (SAM spec 4.1; raw DEFLATE, RFC 1951) is inflated on the GPU, one lane per block. */
typedef struct pssbam_bgzf_block {
    uint64_t in_off;   /* offset of the block's raw deflate payload in the compressed buffer   */
    uint32_t in_len;   /* payload bytes                                                        */
    uint32_t isize;    /* ISIZE: bytes the block inflates to (<= 65536)                        */
    uint64_t out_off;  /* where they go in the output buffer                                   */
    uint32_t crc;      /* CRC-32 of the inflated bytes, from the block trailer                 */
    uint32_t status;   /* out: 0 = ok, else which check failed (1..8, csrc/inflate_kernels.h)  */
} pssbam_bgzf_block;

/* Walks the BGZF headers of bytes[0..nbytes): fills blocks[] (out_off = running sum of ISIZE) up to
 * max_blocks (blocks == NULL: count only), stops at the first partial block.  Returns the number
 * of whole blocks or PSSBAM_EFORMAT; *consumed = bytes they cover, *inflated_bytes = sum of ISIZE. */
int64_t pssbam_bgzf_scan(const void *bytes, uint64_t nbytes, pssbam_bgzf_block *blocks, uint64_t max_blocks,
                         uint64_t *consumed, uint64_t *inflated_bytes);

/* Inflates n_blocks blocks on the current device, asynchronously on hip_stream: d_comp (4-byte
 * aligned, readable 4 bytes past comp_bytes) -> d_out at each block's out_off; d_blocks[i].status
 * tells how block i went (check_crc != 0 adds the ISIZE/CRC-32 kernel).  d_out must hold
 * out_off + isize bytes for every block: the table is the caller's, and only its in_off / in_len /
 * isize are checked on the device (a block that fails them gets status 1 and is not touched). */
int pssbam_bgzf_inflate_device(void *hip_stream, const void *d_comp, uint64_t comp_bytes, pssbam_bgzf_block *d_blocks,
                               uint32_t n_blocks, void *d_out, int check_crc);

/* The whole feed in one call: a batch of whole BGZF blocks (compressed bytes in host memory,
 * page-locked for full PCIe speed; blocks[] from pssbam_bgzf_scan with in_off relative to comp and
 * out_off starting at 0, inflating to < 4 GiB) is copied, inflated, CRC-checked, record-indexed and
 * tallied on the device, asynchronously.  Consecutive calls continue ONE record stream (records may
 * cross BGZF blocks and calls); first_record_offset = bytes of the stream's first block that come
 * before the first alignment record (the BAM header; only in the first call after create / reset).
 * comp must stay untouched until pssbam_engine_wait_bgzf_copied(e, *ticket).  Whether the blocks
 * were sound is known once the work has run: pssbam_engine_feed_status. */
int pssbam_engine_submit_bgzf(pssbam_engine *e, const void *comp, uint64_t comp_bytes, const pssbam_bgzf_block *blocks,
                              uint32_t n_blocks, uint32_t first_record_offset, uint64_t *ticket);
int pssbam_engine_wait_bgzf_copied(pssbam_engine *e, uint64_t ticket);
/* Optional, BEFORE set_genome: declares that compressed blocks will be fed ahead of the genome.  n_ref = the
 * reference count of the BAM header (the record chain is judged with it; set_references must bring the same
 * count later).  pssbam_engine_submit_bgzf is then legal at once: inflate, CRC-32 and record index run as the
 * blocks arrive -- they need no reference base -- and the tally launches of every super-batch follow when
 * set_genome(_async) + set_references have been called.  The inflated records wait in device memory meanwhile
 * (a ring of 4.4 GB slots that grows within what the device has free, genome_bytes_hint -- e.g. the FASTA's
 * size, 0 = unknown -- left alone); when every slot is full submit_bgzf returns PSSBAM_EBUSY and has taken
 * NOTHING of that chunk: set the genome, then submit the chunk again. */
int pssbam_engine_feed_open(pssbam_engine *e, int32_t n_ref, uint64_t genome_bytes_hint);
/* The blocks submitted next do NOT continue the stream fed so far (an engine that is dealt every n-th
 * run of a file): pending blocks are processed, a partial record left at this point raises
 * PSSBAM_FEED_TRUNCATED, and the next blocks start a new record chain at their first byte. */
int pssbam_engine_feed_break(pssbam_engine *e);
/* Several engines dealt alternating runs of ONE stream (one BAM over n GPUs): the blocks submitted to `to`
 * from now on continue the stream where the blocks submitted to `from` so far end.  `from`'s pending blocks
 * are processed; the partial record its run ends in (records cross BGZF blocks in files written by htsjdk)
 * travels to `to` -- device to device through page-locked host memory, no host wait -- and is completed,
 * indexed and tallied there, so the record chain is checked across engines as it is inside one.  The
 * reference has no counterpart (one process, one `samtools view` pipe: pss-bam.c:148-162, 764-783). */
int pssbam_engine_feed_handoff(pssbam_engine *from, pssbam_engine *to);
#define PSSBAM_FEED_BAD_BLOCK 1u   /* a block failed inflate / ISIZE / CRC-32                         */
#define PSSBAM_FEED_RAGGED 2u      /* the per-block record chains did not link up (or a record above 16 MiB):
                                      use the host reader for this file                               */
#define PSSBAM_FEED_BAD_RECORD 4u  /* an alignment record with block_size < 32                        */
#define PSSBAM_FEED_TRUNCATED 8u   /* the stream ended inside an alignment record (as of the last sync) */
int pssbam_engine_feed_status(pssbam_engine *e, uint32_t *flags, double *inflate_ms, uint64_t *inflated_bytes);
/* Optional, before pssbam_engine_submit_bgzf / _submit_device: a few whole alignment records in host
 * memory (e.g. the first ones of the file) from which the tiled kernels' staged record prefix is
 * sized -- otherwise the engine reads the first block back from the device to look at its records. */
int pssbam_engine_hint_records(pssbam_engine *e, const void *records, uint64_t nbytes);
/* Optional: allocates two of the feed's slots (2 x (2 GiB compressed + 4.5 GiB inflated) = ~13 GB) for
 * `device` ahead of time, e.g. from a helper thread while the FASTA loads; the first engine on that device
 * that feeds compressed blocks takes them.  pssbam_feed_release frees what no engine took. */
int pssbam_feed_reserve(int device);
int pssbam_feed_release(int device);

/* Test / tool convenience: host BGZF bytes in, inflated bytes out (out may be NULL), kernels timed
 * with HIP events (*kernel_ms = best of `repeats` runs of inflate + CRC). */
int pssbam_bgzf_inflate_host(int device, const void *bgzf, uint64_t nbytes, void *out, uint64_t out_cap, uint64_t *out_len,
                             uint32_t *n_blocks, uint32_t *first_bad_block, uint32_t *first_bad_status, double *kernel_ms,
                             int check_crc, int repeats);

/* Host helper: walks the block_size chain of an inflated BAM record stream.  Writes up
 * to max_records offsets (+ the end sentinel), returns the number of whole records
 * found (>= 0) or PSSBAM_EFORMAT; *consumed = bytes covered by those records. */
int64_t pssbam_index_records(const void *bytes, uint64_t nbytes, uint32_t *offsets,
                             uint64_t max_records, uint64_t *consumed);

#ifdef __cplusplus
}
#endif
#endif /* PSSBAM_HIP_H */
