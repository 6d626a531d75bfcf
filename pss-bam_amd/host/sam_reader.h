/*
 * pss-bam_amd/host/sam_reader.h -- SAM *text* input (plain or gzip) for the front ends.
 *
 * The reference only ever sees SAM text (what `samtools view` prints, parsed by line2saml,
 * sam-parse.c:10-91).  Here each text line is turned into the BAM alignment record whose
 * text-equivalent reading on the GPU (csrc/record_decode.h) reproduces exactly the fields
 * line2saml would have produced -- so text and binary input share one device path.  Lines
 * line2saml rejects (fewer than eleven fields, SEQ/QUAL length mismatch) are dropped here and
 * counted.
 */
#ifndef PSSBAM_SAM_READER_H
#define PSSBAM_SAM_READER_H

#include <stddef.h>
#include <stdint.h>

typedef struct sam_reader sam_reader;

sam_reader *sam_reader_open(const char *path, size_t batch_bytes, char *err, size_t errlen);
/* reference names known so far: @SQ lines first, then RNAMEs met in records, in order of
 * first appearance; the table only grows, ids are stable */
int32_t sam_reader_n_ref(const sam_reader *r);
const char *const *sam_reader_ref_names(const sam_reader *r);
/* next batch of encoded records (same contract as bam_reader_next) */
int64_t sam_reader_next(sam_reader *r, const uint8_t **records, const uint32_t **offsets, size_t *nbytes);
uint64_t sam_reader_lines_skipped(const sam_reader *r); /* lines line2saml would have rejected */
const char *sam_reader_error(const sam_reader *r);
void sam_reader_close(sam_reader *r);

/* 1 if the file starts like a BGZF-compressed BAM, 0 if not (SAM text, possibly gzipped), -1 on I/O error */
int file_is_bam(const char *path);

#endif
