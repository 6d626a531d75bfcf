/*
 * include/sam-parse.h -- SAM record interface of the MI355X engine.
 *
 * Source-compatible with the reference's header of the same name
 * (/root/reference/sam-parse.h:1-61): identical macros, identical `struct saml`
 * member names / order / types (sizeof(Saml) == 20536, seq at offset 8224 on LP64)
 * and identical prototypes.  The parser behind it (pss-bam_amd/host/samline.c) is a
 * hand-written tokenizer with scanf-equivalent field rules instead of one sscanf.
 *
 * In the engine the per-read hot path does not go through this struct at all: BAM
 * records are decoded on the GPU (pss-bam_amd/csrc/record_decode.h).  `Saml` remains
 * for callers of the reference API and for the SAM-text input path.
 */
#ifndef PSSBAM_SAM_PARSE_H
#define PSSBAM_SAM_PARSE_H

#include <stdio.h>
#include <stdlib.h>
#include <ctype.h>
#include <string.h>
#include <limits.h>
#include <unistd.h>

/* reference sam-parse.h:8-14 */
#define MAX_LINE_LEN (200000)      /* longest SAM line handed to line2saml             */
#ifndef MAX_FN_LEN
#define MAX_FN_LEN (2047)
#endif
#define MAX_FIELD_WIDTH (2047)     /* longest text field kept in a Saml                */
#define MATCH (1)
#define MISMATCH (4)
#define GAP_OPEN (6)
#define GAP_EXT (1)

#ifdef __cplusplus
extern "C" {
#endif

/* One alignment line, reference sam-parse.h:20-56.  The twelve one-bit members are
 * FLAG bits 0x1 .. 0x800 in ascending order. */
typedef struct saml {
  char qname[MAX_FIELD_WIDTH + 1];
  unsigned int flag;
  unsigned int paired : 1;         /* 0x1   template has several segments              */
  unsigned int proper_pair : 1;    /* 0x2   every segment properly aligned             */
  unsigned int unmap : 1;          /* 0x4   this segment unmapped                      */
  unsigned int munmap : 1;         /* 0x8   next segment unmapped                      */
  unsigned int reverse : 1;        /* 0x10  SEQ is reverse-complemented                */
  unsigned int mreverse : 1;       /* 0x20  next segment reverse-complemented          */
  unsigned int read1 : 1;          /* 0x40  first segment of the template              */
  unsigned int read2 : 1;          /* 0x80  last segment of the template               */
  unsigned int secondary : 1;      /* 0x100                                            */
  unsigned int qc_failed : 1;      /* 0x200                                            */
  unsigned int duplicate : 1;      /* 0x400                                            */
  unsigned int supplementary : 1;  /* 0x800                                            */
  char rname[ MAX_FIELD_WIDTH + 1];
  unsigned long pos;               /* 1-based POS                                      */
  unsigned int mapq;
  char cigar[MAX_FIELD_WIDTH + 1];
  char mrnm[MAX_FIELD_WIDTH + 1];
  unsigned int mpos;
  int isize;                       /* TLEN; overwritten with strlen(seq) when !paired  */
  int seq_len;                     /* strlen(seq)                                      */
  char seq[MAX_FIELD_WIDTH + 1];
  char qual[MAX_FIELD_WIDTH + 1];
  char tags[MAX_FIELD_WIDTH + 1];  /* raw text of the optional fields, if any          */
  char BC[MAX_FIELD_WIDTH + 1];
  char RG[MAX_FIELD_WIDTH + 1];
  char opt_tags[MAX_FIELD_WIDTH + 1];
  int aln_seq_len;
  int NM;
  int AS;                          /* aligner score                                    */
  int XM;                          /* mismatches                                       */
  int XO;                          /* gap opens                                        */
  int XG;                          /* gap extensions                                   */
} Saml;

/* Fills *sp from one SAM text line.  0 = ok; 1 = fewer than eleven fields, or SEQ and
 * QUAL differ in length (callers skip such lines).  reference: sam-parse.c:10-91. */
int line2saml( const char* line, Saml* sp );

/* 1 iff the line starts with '@'.  reference: sam-parse.c:130-137. */
int is_header( const char* line );

/* Sum of the lengths of the M operations of a CIGAR string.  reference: :101-125. */
int aln_seq_len( const char* cigar );

/* 1 unless sp->AS is positive and below m*seq_len + b.  reference: :153-163. */
int good_score( Saml* sp, float m, float b );

#ifdef __cplusplus
}
#endif
#endif /* PSSBAM_SAM_PARSE_H */
