/* Host-side format builders under AddressSanitizer + UBSan (tests/test_sanitizers.py):
 * every layout the backend builds on the host (binned, two-phase, sliced-ELL, operator,
 * RCM, partition, synthetic generators incl. spd=1) on small inputs, leaks counted. */
#include "lsbench_hip.h"
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
int main(void) {
  const char *specs[] = {"powerlaw:n=30000,gamma=1.2,max=3000,seed=7", "powerlaw:n=5000,gamma=1.585350372615855,max=4096,seed=3",
                         "lap2d:nx=170,ny=150", "lap3d:nx=20,ny=17,nz=11", "powerlaw:n=3000,gamma=1.3,max=700,seed=11,spd=1",
                         "lap2d:nx=3,ny=2", "powerlaw:n=40,gamma=1.0,max=40,seed=1"};
  for (unsigned k = 0; k < sizeof specs / sizeof specs[0]; k++) {
    unsigned n;
    struct csr *A = lsbench_matrix_synth(specs[k], 0, 0, &n);
    if (!A) { printf("bad %s\n", specs[k]); return 1; }
    unsigned widths[] = {7, 300, 1024, 262144};
    for (int w = 0; w < 4; w++) {
      struct lsb_binned *B = lsb_csr_binize(A, widths[w]);
      unsigned long long s = 0;
      for (unsigned c = 0; c < B->nchunks; c++) s += B->chunk_begin[c + 1] - B->chunk_begin[c];
      if (s != B->nnz) { printf("binize mismatch\n"); return 1; }
      lsb_binned_free(B);
    }
    unsigned tilings[][2] = {{0, 0}, {64, 64}, {16384, 4096}};
    for (int t = 0; t < 3; t++) { /* every slot reached exactly once through grp_first / grp_mask / delta */
      struct lsb_pb *P = lsb_csr_pbize2(A, tilings[t][0], tilings[t][1]);
      unsigned char *hit = calloc(P->nnz + 1, 1);
      unsigned long long s = 0;
      for (unsigned it = 0; it < P->nitems; it++)
        for (unsigned e = P->item[3 * it + 1]; e < P->item[3 * it + 2]; e++) {
          const unsigned lane = e % 64;
          const unsigned long long le = lane == 63 ? ~0ull : (2ull << lane) - 1ull;
          const unsigned piece = P->grp_first[e / 64] + (unsigned)__builtin_popcountll(P->grp_mask[e / 64] & le);
          const unsigned slot = e + P->delta[piece];
          if (piece >= P->npieces || slot >= P->nnz || hit[slot]++ || P->roww[slot] >= P->rows ||
              P->colw[e] >= P->cols) { printf("pbize mismatch\n"); return 1; }
          s++;
        }
      if (s != P->nnz || P->bin_ptr[P->nbins] != P->nnz) { printf("pbize mismatch\n"); return 1; }
      free(hit);
      /* the bounds the kernels rely on, as the backend asserts them at upload (deep here); and the
       * two rules round 2's experiments broke must be caught: a product array without the two spare
       * slots of the pair loads, stores at the padded entry index (delta = 0) */
      char why[256];
      if (lsb_pb_check(P, P->nnz + 2, P->nnz + 2, 1, why, sizeof why)) { printf("pb_check: %s\n", why); return 1; }
      if (lsb_pb_check(P, P->nnz + 2, P->nnz + 2, 0, why, sizeof why)) { printf("pb_check shallow: %s\n", why); return 1; }
      if (!(P->bin_ptr[P->nbins] - 1u) % 2 == 0 && lsb_pb_check(P, P->nnz, P->nnz + 2, 0, why, sizeof why) != 3) {
        printf("pb_check accepted a product array without spare slots\n"); return 1; }
      if (P->nent > P->nnz) { /* padded chunks: storing at the entry index itself leaves [0, nnz) */
        unsigned keep = P->delta[P->npieces - 1];
        P->delta[P->npieces - 1] = 0;
        const int rc = lsb_pb_check(P, P->nnz + 2, P->nnz + 2, 0, why, sizeof why);
        P->delta[P->npieces - 1] = keep;
        /* (the last piece's entries sit at padded indices >= its slots; index + 0 reaches past nnz
         * unless nothing was padded in front of it) */
        unsigned last_e1 = P->item[3 * (P->nitems - 1) + 2];
        if (last_e1 > P->nnz && rc != 7) { printf("pb_check accepted stores at the padded entry index\n"); return 1; }
      }
      lsb_pb_free(P);
    }
    for (int pw = 1; pw <= (k >= 2 && k != 4 && k != 6 ? 3 : 2); pw++) { /* FSAI pattern: rows end in their diagonal, ascending, within the cap
                                        (the third power only where it stays small: the grids) */
      struct lsb_fsai_pattern *F = lsb_csr_fsai_pattern(A, pw, pw == 3 ? 7 : LSB_FSAI_CAP);
      for (unsigned i = 0; i < F->n; i++) {
        const unsigned a = F->offs[i], b = F->offs[i + 1];
        if (b <= a || b - a > F->cap || F->cols[b - 1] != i) { printf("fsai pattern row %u\n", i); return 1; }
        for (unsigned e = a + 1; e < b; e++)
          if (F->cols[e] <= F->cols[e - 1]) { printf("fsai pattern order\n"); return 1; }
      }
      lsb_fsai_pattern_free(F);
    }
    {
      struct lsb_sell *H16 = lsb_csr_sellize16(A, 0);
      if (H16) {
        struct lsb_sell_vc *Vc = lsb_sell16_value_slots(H16);
        struct lsb_sell_tmpls *Tm = lsb_sell16_templates(H16, Vc);
        /* the template kernel's bounds (deep), and three broken rules must be CAUGHT: a gather vector
         * one entry short, a far base one grid line too far, a mask index past the array */
        char why[256];
        if (lsb_tmpl_check(H16, Vc, Tm, 0, A->nrows, n, 1, why, sizeof why)) { printf("tmpl_check: %s\n", why); return 1; }
        if (lsb_tmpl_check(H16, Vc, NULL, 0, A->nrows, n, 1, why, sizeof why)) { printf("tmpl_check (no templates): %s\n", why); return 1; }
        if (Tm && Tm->covered) {
          long long reach = -1; /* one past the last column a constant slot's 128 gathers touch */
          for (unsigned s = 0; s < H16->nslice; s++)
            for (unsigned q = H16->sptr[s] / LSB_SELL_ROWS; q < H16->sptr[s + 1] / LSB_SELL_ROWS; q++)
              if (Vc->slots[4 * (size_t)q + 2] < 0 && (long long)s * LSB_SELL_ROWS + Vc->slots[4 * (size_t)q] + LSB_SELL_ROWS > reach)
                reach = (long long)s * LSB_SELL_ROWS + Vc->slots[4 * (size_t)q] + LSB_SELL_ROWS;
          if (reach > 0 && (reach > (long long)n || lsb_tmpl_check(H16, Vc, Tm, 0, A->nrows, (unsigned)reach - 1, 0, why, sizeof why) != 7)) {
            printf("tmpl_check accepted a gather vector one entry short of a constant slot's reach\n"); return 1; }
          for (unsigned s = 0; s < Tm->nslice; s++)
            if (Tm->tid[s] != 255) {
              struct lsb_sell_tmpl *t = &Tm->t[Tm->tid[s]];
              const int keep = t->base[0];
              t->base[0] -= 1;
              const int rc = lsb_tmpl_check(H16, Vc, Tm, 0, A->nrows, n, 0, why, sizeof why);
              t->base[0] = keep;
              if (rc == 0) { printf("tmpl_check accepted a template base that disagrees with its slice\n"); return 1; }
              break;
            }
          if (Tm->nmask) {
            const unsigned long long keep = Tm->nmask;
            Tm->nmask = 0;
            const int rc = lsb_tmpl_check(H16, Vc, Tm, 0, A->nrows, n, 0, why, sizeof why);
            Tm->nmask = keep;
            if (rc != 21) { printf("tmpl_check accepted a mask index past the mask array (rc %d)\n", rc); return 1; }
          }
        }
        lsb_sell_tmpls_free(Tm), lsb_sell_vc_free(Vc), lsb_sell_free(H16);
      }
    }
    struct lsb_sell *E = lsb_csr_sellize(A); lsb_sell_free(E);
    E = lsb_csr_sellize16(A, 0); lsb_sell_free(E);
    struct csr *S = lsb_csr_symmetrize_upper(A);
    unsigned *perm = malloc(sizeof(unsigned) * S->nrows);
    lsb_csr_rcm(S, perm);
    struct csr *Sp = lsb_csr_permute_sym(S, perm);
    unsigned b[5]; lsb_csr_partition_rows(S, 4, b);
    struct csr *R = lsb_csr_row_slice(S, b[1], b[2]);
    lsb_csr_free(R); lsb_csr_free(Sp); lsb_csr_free(S); free(perm);
    struct csr *part = lsbench_matrix_synth(specs[k], n / 3, n / 2, &n);
    lsb_csr_free(part);
    lsb_csr_free(A);
    printf("ok %s\n", specs[k]);
  }
  /* line padding of a 2-D grid and the z-column plans of the padded copy and of a 3-D grid (their builders and
   * checks walk every slice; a broken item must be CAUGHT) */
  {
    const char *grids[] = {"lap2d:nx=1000,ny=12", "lap3d:nx=128,ny=16,nz=9"};
    for (unsigned k = 0; k < 2; k++) {
      unsigned n;
      struct csr *A = lsbench_matrix_synth(grids[k], 0, 0, &n);
      struct csr *S = lsb_csr_copy_base0(A);
      unsigned nx = 0, nxp = 0;
      int *map = NULL;
      struct csr *P = lsb_csr_pad_lines(S, 128, &nx, &nxp, &map);
      if ((k == 0) != (P != NULL)) { printf("pad_lines: %s\n", grids[k]); return 1; }
      if (P) {
        if (nx != 1000 || nxp != 1024 || P->nrows != 12u * 1024u) { printf("pad_lines: %u -> %u, %u rows\n", nx, nxp, P->nrows); return 1; }
        for (unsigned r = 0; r < P->nrows; r++)
          if ((map[r] >= 0) != (r % nxp < nx)) { printf("pad_lines: map\n"); return 1; }
      }
      struct csr *G = P ? P : S;
      struct lsb_sell *H = lsb_csr_sellize16(G, 0);
      struct lsb_sell_vc *V = H ? lsb_sell16_value_slots(H) : NULL;
      struct lsb_sell_tmpls *T = V ? lsb_sell16_templates(H, V) : NULL;
      const unsigned period = k == 0 ? 8u : 16u; /* a padded line: 8 slices; a plane of the 3-D grid: 16 */
      for (unsigned kmax = 2; T && kmax <= 16; kmax += 7) {
        struct lsb_tmpl_cols *C = lsb_sell_tmpl_columns(T, period, kmax);
        struct lsb_tmpl_cols *Cr = lsb_sell_tmpl_columns_range(T, period, kmax, T->nslice / 4, T->nslice - 1);
        char why[256];
        if (C && lsb_tmpl_cols_check(T, C, why, sizeof why)) { printf("cols_check: %s\n", why); return 1; }
        if (Cr && lsb_tmpl_cols_check(T, Cr, why, sizeof why)) { printf("cols_check (range): %s\n", why); return 1; }
        if (C && C->nitem) {
          const unsigned keep = C->item[0];
          C->item[0] = T->nslice; /* a first slice past the layout */
          if (!lsb_tmpl_cols_check(T, C, why, sizeof why)) { printf("cols_check accepted an item past the layout\n"); return 1; }
          C->item[0] = keep;
        }
        lsb_tmpl_cols_free(C), lsb_tmpl_cols_free(Cr);
      }
      lsb_sell_tmpls_free(T), lsb_sell_vc_free(V), lsb_sell_free(H);
      free(map);
      if (P)
        lsb_csr_free(P);
      lsb_csr_free(S), lsb_csr_free(A);
      printf("ok %s (padding / z-columns)\n", grids[k]);
    }
  }
  return 0;
}
