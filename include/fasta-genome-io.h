/*
 * include/fasta-genome-io.h -- reference-genome loader interface of the MI355X engine.
 *
 * Source-compatible with the reference's header of the same name
 * (/root/reference/fasta-genome-io.h:1-46): same macro names and values, same struct
 * tags / typedefs / member names, order and types (so sizeof and offsetof agree and
 * code written against the reference header compiles and links unchanged), same
 * function names, argument order and return conventions.  The implementation behind
 * it (pss-bam_amd/host/genome_load.c) is new: block I/O plus a table-driven
 * strip/upper-case pass instead of one fgetc()/gzgetc() per byte.
 *
 * Loaded form of a contig (reference behaviour, fasta-genome-io.c:105-148):
 *   id   : the bytes after '>' up to the first white-space character
 *   seq  : every non-white-space byte of the body, toupper()ed, NUL terminated
 *   len  : strlen(seq)
 * and Genome.seqs is sorted by strcmp(id) (fasta-genome-io.c:236) so find_seq can
 * bsearch it (fasta-genome-io.c:202-213).
 */
#ifndef PSSBAM_FASTA_GENOME_IO_H
#define PSSBAM_FASTA_GENOME_IO_H

#include <stdio.h>
#include <stdlib.h>
#include <ctype.h>
#include <string.h>
#include <limits.h>
#include <zlib.h>

/* limits, reference fasta-genome-io.h:7-10 */
#define MAX_FN_LEN (2047)          /* longest file name kept in Fa_Src.fn              */
#define MAX_ID_LEN (511)           /* longest contig id                                */
#define MAX_SEQ_LEN (536870911)    /* longest contig; longer ones are cut with a note  */
#define MAX_GENOME_SEQS (1000000)  /* most contigs a Genome holds                      */

#ifdef __cplusplus
extern "C" {
#endif

/* one contig; reference fasta-genome-io.h:13-17 (528 bytes on LP64) */
typedef struct seq {
  char id[MAX_ID_LEN + 1];
  char* seq;                       /* heap, owned by the Seq                           */
  size_t len;
} Seq;

/* the whole reference; reference fasta-genome-io.h:19-23 */
typedef struct genome {
  Seq** seqs;                      /* n_seqs entries, sorted by id                     */
  Seq* dummy;                      /* scratch key for find_seq (makes it non-reentrant) */
  size_t n_seqs;
} Genome;

/* an open FASTA stream; reference fasta-genome-io.h:25-32 */
typedef struct fa_src {
  char fn[MAX_FN_LEN+1];
  char* seq_buffer;                /* staging area for the contig being read           */
  int is_gz;                       /* chosen from the ".gz" suffix (is_gz)             */
  gzFile fagz;
  FILE* fafp;
  size_t n;                        /* records delivered so far                         */
} Fa_Src;

/* Loads every record of `fn` (plain or .gz) and sorts by id.  Heap result, release
 * with destroy_genome.  reference: fasta-genome-io.c:221-238. */
Genome* init_genome( const char fn[] );

/* Opens `fn`; NULL when fn is NULL or cannot be opened.  reference: :20-50. */
Fa_Src* init_fasta_src( const char fn[] );

/* Reads the next record, appends it to genome->seqs and returns it; NULL at end of
 * input.  reference: :60-83. */
Seq* get_next_fa( Fa_Src* fa_source, Genome* genome );

/* Record readers for the two stream kinds: 0 = a record was read, non-zero = end of
 * input.  seq_buffer must hold MAX_SEQ_LEN+1 bytes.  reference: :105-150, :157-200. */
int read_fasta( FILE* fp, Seq* seq, char* seq_buffer );
int gzread_fasta( gzFile gzfp, Seq* seq, char* seq_buffer );

/* Contig with exactly this id, or NULL.  reference: :202-213. */
Seq* find_seq( Genome* genome, const char id[] );

/* 1 iff the name ends in ".gz".  reference: :6-15. */
int is_gz( const char* fn );

/* fopen that reports failures on stderr (name, then perror) and returns NULL.
 * reference: :264-273. */
FILE* fileOpen( const char* name, char access_mode[] );

int close_fasta_src( Fa_Src* );                       /* reference: :85-95   */
int chr_cmp( const void *v1, const void *v2 );        /* Seq** comparator, by id; :215-219 */
int destroy_seq(Seq* seq);                            /* reference: :241-248 */
int destroy_genome(Genome* genome);                   /* reference: :250-261 */

#ifdef __cplusplus
}
#endif
#endif /* PSSBAM_FASTA_GENOME_IO_H */
