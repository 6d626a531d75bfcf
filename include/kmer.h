/*
 * include/kmer.h -- k-mer counter interface of the MI355X engine.
 *
 * Source-compatible with the reference's header of the same name
 * (/root/reference/kmer.h:1-33): same K_AR_SIZE, same `ktn` / `Kmers` layouts and
 * typedef names, same five prototypes with the same return conventions.
 *
 * Semantics kept (reference kmer.c:43-214): bases are case-folded; a k-mer is valid
 * iff all k bases are A/C/G/T; bin index = 2 bits per base read left to right with
 * A=0 C=1 G=2 T=3; the first min(k, K_AR_SIZE) bases index `ka`, later bases walk the
 * 4-ary node tree; counts are unsigned int and stick at UINT_MAX.
 *
 * On the GPU the same counter is a flat 4^k histogram (pss-bam_amd/csrc); this host
 * structure is what the fragkon front end fills from it for printing, and what
 * callers of the reference API keep using.
 */
#ifndef PSSBAM_KMER_H
#define PSSBAM_KMER_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <limits.h>

#define K_AR_SIZE (8)              /* bases resolved by the flat array part            */

#ifdef __cplusplus
extern "C" {
#endif

/* tree node for bases beyond K_AR_SIZE; reference kmer.h:9-15 */
typedef struct kmer_tree_node {
  struct kmer_tree_node* Ap;
  struct kmer_tree_node* Cp;
  struct kmer_tree_node* Gp;
  struct kmer_tree_node* Tp;
  unsigned int count;
} ktn;
typedef struct kmer_tree_node* ktnP;

/* reference kmer.h:18-24 */
typedef struct kmers {
  size_t k;                        /* k-mer length                                     */
  size_t k_ar_size ;               /* how many leading bases the array part resolves   */
  ktnP* ka;                        /* 4^K_AR_SIZE node pointers, NULL until first use  */
} Kmers;
typedef struct kmers* KSP;

KSP init_KSP( int k );                                        /* reference kmer.c:3-16    */
int add_to_ksp( const char* kmer, KSP ks );                   /* 0 counted, -1 invalid; :43-111 */
unsigned int kmer2count( const char* kmer, const KSP ks );    /* 0 when unseen/invalid; :121-166 */
int kmer2inx( const char* kmer,                               /* 1 = *inx set, 0 = invalid; :184-214 */
	      const size_t kmer_len,
	      size_t* inx );
int destroy_KSP(KSP ks);                                      /* reference kmer.c:220-231 */

#ifdef __cplusplus
}
#endif
#endif /* PSSBAM_KMER_H */
