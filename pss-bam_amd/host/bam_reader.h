/*
 * pss-bam_amd/host/bam_reader.h -- BGZF / BAM input for the front ends.
 *
 * Replaces the `samtools view` child process of the reference (pss-bam.c:148-162,
 * fragkon.c:84-93): BGZF blocks are inflated by a pool of host threads straight into one
 * large buffer, and the caller receives blocks of WHOLE raw alignment records plus their
 * offset index -- exactly what pssbam_engine_submit() wants.  No per-record work happens on
 * the host beyond following the block_size chain.
 */
#ifndef PSSBAM_BAM_READER_H
#define PSSBAM_BAM_READER_H

#include <stddef.h>
#include <stdint.h>

typedef struct bam_reader bam_reader;

typedef struct bam_header {
    char *text;            /* SAM header text (not NUL-counted in l_text) */
    uint32_t l_text;
    int32_t n_ref;
    char **ref_name;       /* n_ref names, refID order */
    uint32_t *ref_len;
} bam_header;

/* n_threads <= 0: one per online CPU (capped at 32).  batch_bytes: size of the inflated
 * batch buffers, two of them (0 = $PSSBAM_BATCH_BYTES or 256 MiB).  On failure returns NULL
 * and describes it in err. */
bam_reader *bam_reader_open(const char *path, int n_threads, size_t batch_bytes, char *err, size_t errlen);
/* The same with n_slots batch buffers in the ring instead of three (0 = $PSSBAM_SLOTS or 3, at most
 * 16): a caller that keeps several batches in flight -- asynchronous copies to several GPUs --
 * takes them with bam_reader_next_hold() and gives each back with bam_reader_release() once its
 * copy has completed; the reader keeps two slots for its own fill/index stages. */
bam_reader *bam_reader_open_slots(const char *path, int n_threads, size_t batch_bytes, int n_slots, char *err, size_t errlen);
int bam_reader_slots(const bam_reader *r);
size_t bam_reader_header_bytes(const bam_reader *r); /* inflated bytes in front of the first alignment record */
const bam_header *bam_reader_header(const bam_reader *r);

/* Next batch of whole alignment records.  *records points into the reader's own buffer
 * (valid until the next call), offsets[0..n] index it (offsets[n] == *nbytes).
 * Returns the record count, 0 at end of file, -1 on error (see bam_reader_error). */
int64_t bam_reader_next(bam_reader *r, const uint8_t **records, const uint32_t **offsets, size_t *nbytes);

/* Like bam_reader_next, but the batch stays valid until bam_reader_release(r, *slot_id); do not mix
 * the two styles on one reader. */
int64_t bam_reader_next_hold(bam_reader *r, const uint8_t **records, const uint32_t **offsets, size_t *nbytes, int *slot_id);
void bam_reader_release(bam_reader *r, int slot_id);

/* The batch buffer, so a caller can page-lock it for DMA (base, capacity). */
void bam_reader_buffer(const bam_reader *r, void **base, size_t *bytes);
const char *bam_reader_error(const bam_reader *r);
double bam_reader_inflate_seconds(const bam_reader *r); /* wall time spent inflating so far */
/* the reader thread's wall time so far: [0] block-table walk + carry copy, [1] inflate,
 * [2] record indexing, [3] waiting for the caller to give a batch slot back */
void bam_reader_phase_seconds(const bam_reader *r, double out[4]);
void bam_reader_close(bam_reader *r);

/* SAM text of one record (no trailing newline handling surprises: ends with '\n'); returns
 * bytes written, or -1 if it does not fit.  names = reference names for refID lookup. */
long bam_record_to_sam(const uint8_t *rec, uint32_t rec_len, const bam_header *h, char *out, size_t cap);
/* 1 iff the record carries RG:Z:<rg> (host twin of the device-side -R filter) */
int bam_record_has_rg(const uint8_t *rec, uint32_t rec_len, const char *rg);

#endif
