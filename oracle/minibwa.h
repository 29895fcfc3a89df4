/* TEST INFRASTRUCTURE -- oracle only; never linked into the product library.
 *
 * minibwa: a from-scratch CPU restatement of the six lh3/bwa entry points the reference
 * hot path calls (bwa_index.hpp:120-122,160,177,214), plus the index-file readers they
 * need.  lh3/bwa is an un-vendored git submodule of the reference (submods/bwa is empty;
 * example/README.md:12 shows bwa 0.7.17-r1194-dirty), so this follows bwa's PUBLISHED
 * on-disk layout and Occ/LF arithmetic, and is pinned by:
 *   - the prebuilt index under /root/reference/example/index (block counts == L2 deltas,
 *     SA(k) for every k == naive suffix array of fwd+revcomp decoded from .pac), and
 *   - the reference's own call sites compiled against it (oracle/_ref).
 */
#ifndef UNC_MINIBWA_H
#define UNC_MINIBWA_H
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t bwtint_t;
typedef unsigned char ubyte_t;

#define BWTALGO_AUTO 0
#define OCC_INTV_SHIFT 7
#define OCC_INTERVAL (1ULL << OCC_INTV_SHIFT)

typedef struct {
    bwtint_t primary;   /* S^{-1}(0): row of the BWT matrix holding the sentinel */
    bwtint_t L2[5];     /* C(): cumulative base counts, L2[0] = 0, L2[4] = seq_len */
    bwtint_t seq_len;   /* length of fwd + revcomp */
    bwtint_t bwt_size;  /* number of u32 words in bwt[] */
    uint32_t *bwt;      /* interleaved Occ counts + 2-bit BWT, 64 B per 128 bases */
    uint32_t cnt_table[256];
    int sa_intv;
    bwtint_t n_sa;
    bwtint_t *sa;
} bwt_t;

typedef struct {
    int64_t offset;
    int32_t len;
    int32_t n_ambs;
    uint32_t gi;
    int32_t is_alt;
    char *name, *anno;
} bntann1_t;

typedef struct {
    int64_t offset;
    int32_t len;
    char amb;
} bntamb1_t;

typedef struct {
    int64_t l_pac;
    int32_t n_seqs;
    uint32_t seed;
    bntann1_t *anns;
    int32_t n_holes;
    bntamb1_t *ambs;
    FILE *fp_pac;
} bntseq_t;

bwt_t *bwt_restore_bwt(const char *fn);
void bwt_restore_sa(const char *fn, bwt_t *bwt);
void bwt_destroy(bwt_t *bwt);
bntseq_t *bns_restore(const char *prefix);
void bns_destroy(bntseq_t *bns);

bwtint_t bwt_occ(const bwt_t *bwt, bwtint_t k, ubyte_t c);
void bwt_2occ(const bwt_t *bwt, bwtint_t k, bwtint_t l, ubyte_t c, bwtint_t *ok, bwtint_t *ol);
bwtint_t bwt_sa(const bwt_t *bwt, bwtint_t k);
int bns_pos2rid(const bntseq_t *bns, int64_t pos_f);

/* Not available offline: the reference only calls these from `uncalled index`
 * (bwa_index.hpp:92-101) and load_pacseq (bwa_index.hpp:141-147). */
int bwa_idx_build(const char *fa, const char *prefix, int algo_type, int block_size);
void err_fread_noeof(void *ptr, size_t size, size_t nmemb, FILE *stream);

/* Work counters (per thread), SURVEY.md section 8(d): N_nbr = bwt_2occ calls,
 * N_sa = bwt_sa calls, N_lf = LF steps taken inside bwt_sa. */
typedef struct { uint64_t n_2occ, n_sa, n_lf; } minibwa_counters_t;
void minibwa_counters_get(minibwa_counters_t *out);
void minibwa_counters_reset(void);

#ifdef __cplusplus
}
#endif
#endif
