/*
 * Shards of the solver object: upload of a row range in the layouts the
 * kernels want, choice and timing of the SpMV form,
 * creation / destruction of solvers, SpMV entry points.
 */
#define _GNU_SOURCE
#include "hip_solver.h"
#define NXCD_HOST 8u /* = NXCD of the kernels (hip_wg.h) */

unsigned pow2_ceil(unsigned v) {
  unsigned p = 1;
  while (p < v)
    p <<= 1;
  return p;
}

/* SpMV kernel choice: rows of a few dozen non-zeros at most stream through
 * LDS (adaptive); long-row matrices go wavefront-per-row. */
void choose_spmv(struct shard *s, const struct lsb_hip_opts *o) {
  const unsigned mean = s->n ? (unsigned)((s->nnz + s->n - 1) / s->n) : 1;
  int v = o->spmv_variant;
  /* A matrix of a few hundred thousand non-zeros is launch-latency bound: the
   * sub-wavefront kernel has a shorter dependent-load chain (offs -> cols ->
   * x) than the row-blocked one (rowblk -> offs -> cols -> x -> LDS -> offs)
   * and wins 3.2 vs 5.7 us per launch on tests/xn3b_A_18.txt. */
  if (v == LSB_SPMV_AUTO)
    v = s->nnz <= 500000ull ? LSB_SPMV_SUBWAVE : LSB_SPMV_ADAPTIVE;
  if (v == LSB_SPMV_PANEL && !s->pn)
    v = LSB_SPMV_ADAPTIVE; /* the operator did not qualify for panels */
  if (v == LSB_SPMV_BINNED && !s->bn)
    v = LSB_SPMV_ADAPTIVE;
  if (v == LSB_SPMV_TWOPHASE && !s->tp_bins)
    v = LSB_SPMV_ADAPTIVE;
  if (v == LSB_SPMV_SELL && !s->d_sptr)
    v = LSB_SPMV_ADAPTIVE; /* no sliced-ELL copy (32-bit offsets exceeded) */
  s->variant = v;
  unsigned L = pow2_ceil(mean ? mean : 1);
  if (L < 2)
    L = 2;
  if (L > 64)
    L = 64;
  s->lanes = L;
}


/* Column-panel form of the shard (lsb_csr_panelize) + its row blocks, one run
 * of blocks per panel so that a launch never crosses a panel. */
static void shard_build_panels(struct shard *s, const struct csr *view, unsigned width) {
  struct lsb_panel_csr *P = lsb_csr_panelize(view, width);
  const unsigned np = P->npanels;
  s->h_pblk = lsb_calloc(unsigned, (size_t)np + 1);
  size_t cap = (size_t)P->offs[P->npairs] / LSB_BLOCK_NNZ * 2 + 4 * (size_t)np + 16, nb = 0;
  unsigned *rball = (unsigned *)malloc((cap + 1) * sizeof(unsigned));
  unsigned char *lanes = (unsigned char *)malloc(cap + 1);
  for (unsigned p = 0; p < np; p++) {
    const unsigned b0 = P->pair_begin[p], cnt = P->pair_begin[p + 1] - b0;
    s->h_pblk[p] = (unsigned)nb;
    if (cnt == 0)
      continue;
    struct csr sub = {cnt, 0, P->offs + b0, NULL, NULL};
    unsigned *rb = NULL;
    const unsigned k = lsb_csr_row_blocks(&sub, LSB_BLOCK_NNZ, &rb);
    if (nb + k + 1 > cap)
      errx(EXIT_FAILURE, "hip_cdna4: panel row-block estimate too small");
    lsb_csr_block_lanes(&sub, rb, k, lanes + nb);
    for (unsigned i = 0; i <= k; i++)
      rball[nb + i] = rb[i] + b0; /* the last entry is the next panel's first */
    nb += k;
    free(rb);
  }
  s->h_pblk[np] = (unsigned)nb;
  rball[nb] = P->npairs;
  s->pn = np;
  s->pd_offs = (int *)dev_upload(P->offs, ((size_t)P->npairs + 1) * sizeof(int));
  s->pd_cols = (int *)dev_upload(P->cols, (size_t)P->offs[P->npairs] * sizeof(int));
  s->pd_vals = (double *)dev_upload(P->vals, (size_t)P->offs[P->npairs] * sizeof(double));
  s->pd_rowmap = (int *)dev_upload(P->pair_row, (size_t)P->npairs * sizeof(int));
  s->pd_rowblk = (int *)dev_upload(rball, (nb + 1) * sizeof(int));
  s->pd_blklanes = (unsigned char *)dev_upload(lanes, nb ? nb : 1);
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
  free(rball), free(lanes);
  lsb_panel_csr_free(P);
}

/* Binned form of the shard (lsb_csr_binize), uploaded as it is. */
static void shard_build_bins(struct shard *s, const struct csr *view, unsigned width) {
  struct lsb_binned *B = lsb_csr_binize(view, width);
  if (!B)
    return;
  s->bn = B->nbins, s->bcap = B->chunk_cap;
  s->h_binchunk = (unsigned *)malloc(((size_t)B->nbins + 1) * sizeof(unsigned));
  memcpy(s->h_binchunk, B->bin_chunk, ((size_t)B->nbins + 1) * sizeof(unsigned));
  s->bd_chunk = (unsigned *)dev_upload(B->chunk_begin, ((size_t)B->nchunks + 1) * sizeof(unsigned));
  s->bd_rows = (unsigned *)dev_upload(B->rows, (size_t)B->nnz * sizeof(unsigned));
  s->bd_cols = (unsigned *)dev_upload(B->cols, (size_t)B->nnz * sizeof(unsigned));
  s->bd_vals = (double *)dev_upload(B->vals, (size_t)B->nnz * sizeof(double));
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
  lsb_binned_free(B);
}

/* fp32 copy of a value array on the device; *exact &= "no value changed" */
static float *upload_f32(const double *v, size_t cnt, int *exact) {
  float *f = (float *)malloc((cnt ? cnt : 1) * sizeof(float));
  if (!f)
    errx(EXIT_FAILURE, "hip_cdna4: out of host memory for the fp32 matrix values");
  int same = 1;
#pragma omp parallel for reduction(& : same) schedule(static)
  for (long long i = 0; i < (long long)cnt; i++) {
    f[i] = (float)v[i];
    same &= (double)f[i] == v[i];
  }
  if (exact)
    *exact &= same;
  float *d = (float *)dev_upload(f, (cnt ? cnt : 1) * sizeof(float));
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
  free(f);
  return d;
}

/* Two-phase form of the shard (lsb_csr_pbize), uploaded as it is. */
static void shard_build_twophase(struct shard *s, const struct csr *view, unsigned n_glob) {
  struct lsb_pb *P = lsb_csr_pbize(view);
  if (!P)
    return;
  { /* the bounds k_pb_products / k_pb_reduce rely on, against the allocations made below */
    char why[256];
    if (lsb_pb_check(P, P->nnz + 2, P->nnz + 2, 0, why, sizeof why))
      errx(EXIT_FAILURE, "hip_cdna4: two-phase layout breaks a bound its kernels rely on: %s", why);
  }
  s->tp_items = P->nitems, s->tp_bins = P->nbins, s->tp_col_lo = P->ncols_lo, s->tp_xlen = n_glob;
  s->tp_cols = P->cols, s->tp_rows = P->rows;
  s->tp_item = (unsigned *)dev_upload(P->item, (size_t)P->nitems * 3 * sizeof(unsigned));
  s->tp_binptr = (unsigned *)dev_upload(P->bin_ptr, ((size_t)P->nbins + 1) * sizeof(unsigned));
  s->tp_first = (unsigned *)dev_upload(P->grp_first, (size_t)(P->nent / 64 + 1) * sizeof(unsigned));
  s->tp_mask = (unsigned long long *)dev_upload(P->grp_mask, (size_t)(P->nent / 64 + 1) * sizeof(unsigned long long));
  s->tp_delta = (unsigned *)dev_upload(P->delta, ((size_t)P->npieces + 1) * sizeof(unsigned));
  s->tp_colw = (unsigned short *)dev_upload(P->colw, (size_t)P->nent * sizeof(unsigned short));
  /* (+2: phase 2 loads slots in pairs, on clamped indices) */
  s->tp_roww = (unsigned short *)lsb_hip_malloc(((size_t)P->nnz + 2) * sizeof(unsigned short));
  LSB_CHK_HIP(hipMemcpy(s->tp_roww, P->roww, (size_t)P->nnz * sizeof(unsigned short), hipMemcpyHostToDevice));
  s->tp_vals = (double *)dev_upload(P->vals, (size_t)P->nent * sizeof(double));
  s->tp_prod = (double *)lsb_hip_malloc(((size_t)P->nnz + 2) * sizeof(double));
  LSB_CHK_HIP(hipMemsetAsync(s->tp_prod, 0, ((size_t)P->nnz + 2) * sizeof(double), g_stream));
  s->tp_binparts = (double *)lsb_hip_malloc((size_t)lsb_k_twophase_groups(P->nbins) * sizeof(double));
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
  lsb_pb_free(P);
}

/* count doubles out of the shard's vector slab (256-byte aligned), or a hipMalloc of their own */
double *shard_vec(struct shard *s, size_t count) {
  const size_t bytes = ((count ? count : 1) * sizeof(double) + 255u) & ~(size_t)255u;
  if (s->d_slab && s->slab_used + bytes <= s->slab_cap) {
    double *p = (double *)(s->d_slab + s->slab_used);
    s->slab_used += bytes;
    return p;
  }
  return (double *)lsb_hip_malloc(bytes);
}
void shard_vec_free(struct shard *s, void *p) {
  if (p && !(s->d_slab && (char *)p >= s->d_slab && (char *)p < s->d_slab + s->slab_cap))
    lsb_hip_free(p);
}

/* Upload rows [r0,r1) of the 0-based operator `S` (global column ids) as one
 * shard.  When `S` holds only the shard's rows, pass local=1. */
void shard_upload(struct shard *s, const struct csr *S, unsigned r0,
                         unsigned r1, int local, unsigned row_begin,
                         unsigned n_glob, const struct lsb_hip_opts *o) {
  const unsigned a = local ? 0 : r0, b = local ? S->nrows : r1;
  const unsigned n = b - a, j0 = S->offs[a], j1 = S->offs[b];
  const unsigned base = S->base;
  s->row_begin = row_begin, s->n = n, s->nnz = j1 - j0, s->n_glob = n_glob;
  int *offs = (int *)malloc(((size_t)n + 1) * sizeof(int));
  int *cols = (int *)malloc(((size_t)s->nnz + 1) * sizeof(int));
  unsigned lo = 0xFFFFFFFFu, hi = 0;
  for (unsigned i = 0; i <= n; i++)
    offs[i] = (int)(S->offs[a + i] - j0);
  for (unsigned j = j0; j < j1; j++) {
    const unsigned c = S->cols[j] - base;
    if (c >= n_glob)
      errx(EXIT_FAILURE, "column %u outside the %u-column operator", c, n_glob);
    cols[j - j0] = (int)c;
    if (c < lo)
      lo = c;
    if (c + 1 > hi)
      hi = c + 1;
  }
  if (s->nnz == 0)
    lo = hi = row_begin;
  s->col_lo = lo, s->col_hi = hi;
  if ((unsigned long long)s->nnz > 0x7FFFFFFFull || n_glob > 0x7FFFFFFFu)
    errx(EXIT_FAILURE, "shard too large for int32 device indices");
  s->d_offs = (int *)dev_upload(offs, ((size_t)n + 1) * sizeof(int));
  s->d_cols = (int *)dev_upload(cols, (size_t)s->nnz * sizeof(int));
  s->d_vals = (double *)dev_upload(S->vals + j0, (size_t)s->nnz * sizeof(double));
  s->mixed = o->precision == LSB_PREC_MIXED, s->exact32 = 1;
  if (s->mixed)
    s->d_vals32 = upload_f32(S->vals + j0, (size_t)s->nnz, &s->exact32);
  /* row blocks of the adaptive kernel, on the local offsets */
  struct csr view = {n, 0, (unsigned *)offs, NULL, NULL};
  unsigned *rb = NULL;
  s->nblk = lsb_csr_row_blocks(&view, LSB_BLOCK_NNZ, &rb);
  s->d_rowblk = (int *)dev_upload(rb, ((size_t)s->nblk + 1) * sizeof(int));
  unsigned char *lanes = (unsigned char *)malloc((size_t)s->nblk + 1);
  lsb_csr_block_lanes(&view, rb, s->nblk, lanes);
  s->d_blklanes = (unsigned char *)dev_upload(lanes, (size_t)s->nblk);
  /* Which row blocks touch columns owned by other shards?  Under row-range
   * partitioning of a banded operator they are a prefix and a suffix; the
   * blocks in between can start before the halo has arrived. */
  {
    const int row_end = (int)(row_begin + n);
    unsigned b1 = 0, b2 = s->nblk;
    int ok = 1;
    unsigned char *ext = (unsigned char *)calloc(s->nblk ? s->nblk : 1, 1);
    for (unsigned k = 0; k < s->nblk; k++)
      for (unsigned r = rb[k]; r < rb[k + 1] && !ext[k]; r++)
        if (offs[r + 1] > offs[r] &&
            (cols[offs[r]] < (int)row_begin || cols[offs[r + 1] - 1] >= row_end))
          ext[k] = 1;
    while (b1 < s->nblk && ext[b1])
      b1++;
    while (b2 > b1 && ext[b2 - 1])
      b2--;
    for (unsigned k = b1; k < b2; k++)
      ok &= !ext[k];
    free(ext);
    s->ov_ok = ok && b2 > b1, s->ov_b1 = b1, s->ov_b2 = b2;
  }
  /* Scattered rows (mean |col-row| in the millions, x far beyond L2): also
   * build the column-panel form; tune_spmv() keeps whichever is faster. */
  {
    struct csr gview = {n, 0, (unsigned *)offs, (unsigned *)cols, (double *)(S->vals + j0)};
    const char *e = getenv("LSBENCH_HIP_PANEL_COLS");
    /* window of x one bin / panel gathers from: 4 MiB = an XCD's whole L2.  Measured
     * on the 8 M-row power-law operator (binned form): 1 MiB 3.71 ms, 2 MiB 3.05,
     * 3 MiB 2.85, 4 MiB 2.80, 8 MiB 3.21 -- fewer passes over y win until the
     * window no longer fits */
    const unsigned width = e ? (unsigned)strtoul(e, NULL, 10) : 524288u;
    const int forced = o->spmv_variant == LSB_SPMV_PANEL;
    const int scattered = s->nnz > 4000000ull && (double)(hi - lo) * 8.0 > 16.0e6 &&
                          lsb_csr_mean_scatter(&gview, row_begin) > 1.0e6;
    if (width && forced) /* (no longer built on its own accord: superseded by the binned form) */
      shard_build_panels(s, &gview, width);
    if (width && (o->spmv_variant == LSB_SPMV_BINNED ||
                  (o->spmv_variant == LSB_SPMV_AUTO && scattered)))
      shard_build_bins(s, &gview, width);
    /* the two-phase form: 1.59 ms against the binned form's 2.79 ms on the 8 M-row
     * power-law operator; the timing pass decides per shard */
    if (o->spmv_variant == LSB_SPMV_TWOPHASE ||
        (o->spmv_variant == LSB_SPMV_AUTO && scattered))
      shard_build_twophase(s, &gview, n_glob);
  }
  /* Near-uniform row lengths (stencils, meshes): also keep a sliced-ELL copy;
   * tune_spmv() keeps whichever kernel is faster on this shard. */
  {
    struct csr gview = {n, 0, (unsigned *)offs, (unsigned *)cols, (double *)(S->vals + j0)};
    const int forced = o->spmv_variant == LSB_SPMV_SELL;
    const unsigned long long stored = (forced || s->nnz >= 4000000ull) ? lsb_csr_sell_stored(&gview) : 0;
    struct lsb_sell *E = NULL;
    if (stored && (forced || (o->spmv_variant == LSB_SPMV_AUTO && stored <= s->nnz + s->nnz / 8)))
      E = lsb_csr_sellize(&gview);
    if (E) {
      /* 3-D stencil?  the largest |col - row| a multiple of the slice height and
       * the slices whole planes: candidate period of the XCD dealing */
      {
        unsigned bw = 0;
        for (unsigned i = 0; i < n; i++)
          if (offs[i + 1] > offs[i]) {
            const long long g = (long long)row_begin + i;
            const long long a = g - cols[offs[i]], b = (long long)cols[offs[i + 1] - 1] - g;
            if (a > (long long)bw)
              bw = (unsigned)a;
            if (b > (long long)bw)
              bw = (unsigned)b;
          }
        if (bw >= 64 * LSB_SELL_ROWS && bw % LSB_SELL_ROWS == 0 && E->nslice >= 2 * (bw / LSB_SELL_ROWS))
          s->sell_period = bw / LSB_SELL_ROWS;
        /* ... and of the z-column walk, which only needs a slice for every XCD in a plane (a line-padded
         * 2-D grid: its "planes" are grid lines of a few dozen slices) */
        if (bw >= NXCD_HOST * LSB_SELL_ROWS && bw % LSB_SELL_ROWS == 0 && E->nslice >= 2 * (bw / LSB_SELL_ROWS))
          s->col_period = bw / LSB_SELL_ROWS;
      }
      s->nslice = E->nslice;
      s->sell32_bytes = (unsigned long long)E->stored * (s->mixed ? 8 : 12) + ((unsigned long long)E->nslice + 1) * 4;
      s->d_sptr = (unsigned *)dev_upload(E->sptr, ((size_t)E->nslice + 1) * sizeof(unsigned));
      s->d_scols = (int *)dev_upload(E->cols, ((size_t)E->stored + LSB_SELL_ROWS) * sizeof(int));
      s->d_svals = s->mixed ? (double *)upload_f32(E->vals, (size_t)E->stored + LSB_SELL_ROWS, NULL)
                            : (double *)dev_upload(E->vals, ((size_t)E->stored + LSB_SELL_ROWS) * sizeof(double));
      const int row_end = (int)(row_begin + n);
      unsigned s1 = 0, s2 = E->nslice;
      int ok = 1;
      unsigned char *ext = (unsigned char *)calloc(E->nslice ? E->nslice : 1, 1);
      for (unsigned k = 0; k < E->nslice; k++)
        for (unsigned r = k * LSB_SELL_ROWS; r < n && r < (k + 1) * LSB_SELL_ROWS && !ext[k]; r++)
          if (offs[r + 1] > offs[r] &&
              (cols[offs[r]] < (int)row_begin || cols[offs[r + 1] - 1] >= row_end))
            ext[k] = 1;
      while (s1 < E->nslice && ext[s1])
        s1++;
      while (s2 > s1 && ext[s2 - 1])
        s2--;
      for (unsigned k = s1; k < s2; k++)
        ok &= !ext[k];
      free(ext);
      s->ov_sok = ok && s2 > s1, s->ov_s1 = s1, s->ov_s2 = s2;
      LSB_CHK_HIP(hipStreamSynchronize(g_stream));
      lsb_sell_free(E);
      /* 10 instead of 12 bytes per entry where every slot of every slice is
       * one diagonal band (stencils, banded meshes) */
      struct lsb_sell *H = getenv("LSBENCH_HIP_NO_C16") ? NULL : lsb_csr_sellize16(&gview, row_begin);
      if (H && H->stored > s->nnz + s->nnz / 8) {
        lsb_sell_free(H);
        H = NULL;
      }
      if (H) {
        s->d_sptr16 = (unsigned *)dev_upload(H->sptr, ((size_t)H->nslice + 1) * sizeof(unsigned));
        s->d_scodes = (short *)dev_upload(H->codes, ((size_t)H->ncode_slots + 1) * LSB_SELL_ROWS * sizeof(short));
        /* slots whose 128 values are one number keep it once (lsb_sell16_value_slots): on a
         * constant-coefficient stencil that is every slot away from the grid's faces.  Taken
         * where it drops at least an eighth of the value slots. */
        { /* every slice the same number of slots?  then the kernel needs no look at sptr */
          unsigned ul = H->nslice ? (H->sptr[1] - H->sptr[0]) / LSB_SELL_ROWS : 0;
          for (unsigned k = 0; k < H->nslice && ul; k++)
            if ((H->sptr[k + 1] - H->sptr[k]) / LSB_SELL_ROWS != ul)
              ul = 0;
          s->sell_ulen = ul;
        }
        struct lsb_sell_vc *V = getenv("LSBENCH_HIP_NO_VCONST") ? NULL : lsb_sell16_value_slots(H);
        if (V && (unsigned long long)V->nval_slots * 8 > V->nslots * 7) {
          lsb_sell_vc_free(V);
          V = NULL;
        }
        if (V) {
          const size_t nv = ((size_t)V->nval_slots + 1) * LSB_SELL_ROWS;
          s->d_sbase = (int *)dev_upload(V->slots, 4 * ((size_t)V->nslots + 1) * sizeof(int));
          s->d_svconst = (double *)dev_upload(V->vconst, ((size_t)V->nslots + 1) * sizeof(double));
          s->d_svals16 = s->mixed ? (double *)upload_f32(V->vals, nv, NULL)
                                  : (double *)dev_upload(V->vals, nv * sizeof(double));
          s->sell_vslots = V->nval_slots, s->sell_slots = (unsigned)V->nslots;
          s->sell16_bytes = (unsigned long long)V->nslots * 24; /* slot record + the slot's constant */
          /* slices with identical constant records share a template (a structured grid has a
           * handful): a 16-byte record per slice instead of 24 per slot, and the three inner diagonals
           * from one gather (k_spmv_tmpl) */
          struct lsb_sell_tmpls *TT = getenv("LSBENCH_HIP_NO_TMPL") ? NULL : lsb_sell16_templates(H, V);
          { /* the bounds the constant-slot and template kernels rely on (unguarded 16-byte gathers,
             * value-slot / mask / template indices), against this shard's rows and gather vector */
            char why[256];
            if (lsb_tmpl_check(H, V, TT, row_begin, n, n_glob, 0, why, sizeof why))
              errx(EXIT_FAILURE, "hip_cdna4: sliced-ELL layout breaks a bound its kernels rely on: %s", why);
          }
          if (TT) {
            if (s->mixed) /* fp32 matrix values: the constants as the fp32 kernels see them */
              for (unsigned t = 0; t < TT->ntmpl; t++)
                for (int j = 0; j < TT->t[t].nslots; j++)
                  TT->t[t].cst[j] = (double)(float)TT->t[t].cst[j];
            { /* per slice ONE 16-byte record {template id, first kept value slot, first mask, 0}: a
               * single scalar load in the kernel */
              unsigned *rec = lsb_calloc(unsigned, 4 * ((size_t)TT->nslice + 1));
              for (unsigned k = 0; k < TT->nslice; k++)
                rec[4 * (size_t)k] = TT->tid[k], rec[4 * (size_t)k + 1] = TT->vbase[2 * (size_t)k],
                               rec[4 * (size_t)k + 2] = TT->vbase[2 * (size_t)k + 1];
              s->d_srec = (unsigned *)dev_upload(rec, 4 * ((size_t)TT->nslice + 1) * sizeof(unsigned));
              LSB_CHK_HIP(hipStreamSynchronize(g_stream));
              free(rec);
            }
            s->d_tmask = (unsigned long long *)dev_upload(TT->mask, 2 * ((size_t)TT->nmask + 1) * sizeof(unsigned long long));
            s->d_tmpl = (struct lsb_sell_tmpl *)dev_upload(TT->t, (size_t)TT->ntmpl * sizeof(struct lsb_sell_tmpl));
            s->tmpl_nfar = TT->nfar, s->tmpl_count = TT->ntmpl;
            s->tmpl_pure = TT->covered, s->tmpl_shaped = TT->shaped;
            /* what a launch streams with templates: 16 bytes per slice, the templates, and for the
             * slices without one their slot records, constants and two offsets */
            s->tmpl_bytes = 16ull * TT->nslice + (unsigned long long)TT->ntmpl * sizeof(struct lsb_sell_tmpl) +
                            16ull * TT->nmask +
                            TT->kept_read * LSB_SELL_ROWS * (s->mixed ? 4ull : 8ull); /* the values it still reads */
            for (unsigned k = 0; k < H->nslice; k++)
              if (TT->tid[k] == 255)
                s->tmpl_bytes += (unsigned long long)(H->sptr[k + 1] - H->sptr[k]) / LSB_SELL_ROWS * 24 + 8;
            /* 3-D stencil with planes of whole slices: the z-column plan (k_spmv_tmpl_col) */
            if (s->col_period && !getenv("LSBENCH_HIP_NO_COL")) {
              const char *ek = getenv("LSBENCH_HIP_COL_K");
              /* columns of up to 16 slices: every plane of x is then read 18 / 16 times (64 M-row 7-point
               * operator: 222 us against 233 us with columns of 8, profiles/r04_col.txt) */
              /* ... where that still leaves every wave several columns to walk: with a dozen thousand
               * waves resident a 10 M-row operator (78 k slices) has 1.2 columns of 16 per wave -- columns of 6 */
              /* (measured through the two-launch iteration, profiles/r04_px.txt: a 10 M-row 5-point grid of
               * whole-slice lines 120.9 / 128.3 / 130.0 us per iteration with columns of 4 / 6 / 12 -- one far
               * slot per side, nothing but streams: the shorter the column the more of them per wave; a
               * 50-plane slab of the 7-point grid 130.1 / 122.6 / 116.6 us -- two far slots: every plane
               * re-read costs two vectors and their +-line operands) */
              /* (final form of the iteration, config 3 padded: 111.0 / 110.0 / 108.6 / 115.8 / 116.7 us with columns
               * of 3 / 4 / 5 / 6 / 8) */
              unsigned kauto = TT->nfar >= 2 ? TT->nslice / 5120u : TT->nslice / 15000u;
              kauto = kauto < 4u ? 4u : kauto > 16u ? 16u : kauto;
              const unsigned kmax = ek ? (unsigned)atoi(ek) : kauto;
              /* [0]: every slice of the shard; [1]: the slices that need no halo, where the shard has
               * such a range (the interior launch of the split SpMV; the boundary launches go
               * through k_spmv_tmpl) */
              for (int w = 0; w < 2; w++) {
                if (w == 1 && !(s->d_colplan && s->ov_sok && (s->ov_s1 > 0 || s->ov_s2 < TT->nslice)))
                  break;
                struct lsb_tmpl_cols *CC = w == 0 ? lsb_sell_tmpl_columns(TT, s->col_period, kmax)
                                                  : lsb_sell_tmpl_columns_range(TT, s->col_period, kmax, s->ov_s1, s->ov_s2);
                if (!CC)
                  break;
                char why[256];
                if (lsb_tmpl_cols_check(TT, CC, why, sizeof why))
                  errx(EXIT_FAILURE, "hip_cdna4: z-column plan breaks a rule its kernel relies on: %s", why);
                unsigned *plan = lsb_calloc(unsigned, 16 + 4 * ((size_t)CC->nitem + 1));
                memcpy(plan, CC->xbeg, sizeof CC->xbeg);
                memcpy(plan + 16, CC->item, 4 * (size_t)CC->nitem * sizeof(unsigned));
                unsigned *d = (unsigned *)dev_upload(plan, (16 + 4 * ((size_t)CC->nitem + 1)) * sizeof(unsigned));
                LSB_CHK_HIP(hipStreamSynchronize(g_stream));
                free(plan);
                if (w == 0) {
                  s->d_colplan = d, s->col_items = CC->nitem, s->col_kmax = CC->kmax, s->col_centre0 = CC->centre0;
                  s->col_slices = CC->in_cols;
                  /* what a launch of the z-column walk streams besides x and y: a 16-byte item per
                   * column or single slice, the templates, a column's masks once, and for the single
                   * slices their slot records, constants and kept values */
                  s->col_bytes = 16ull * CC->nitem + (unsigned long long)TT->ntmpl * sizeof(struct lsb_sell_tmpl);
                  for (unsigned q = 0; q < CC->nitem; q++) {
                    const unsigned sl = CC->item[4 * (size_t)q], run = CC->item[4 * (size_t)q + 1] & ~LSB_TMPL_COL_LOCKSTEP;
                    if (run >= 2) {
                      const struct lsb_sell_tmpl *t = &TT->t[TT->tid[sl]];
                      for (int j = 0; j < t->nslots; j++)
                        s->col_bytes += t->kind[j] == 2 ? 16ull : 0ull;
                    } else {
                      const unsigned q0 = H->sptr[sl] / LSB_SELL_ROWS, len = (H->sptr[sl + 1] - H->sptr[sl]) / LSB_SELL_ROWS;
                      s->col_bytes += 24ull * len + 8;
                      for (unsigned j = 0; j < len; j++)
                        if (V->slots[4 * ((size_t)q0 + j) + 2] >= 0)
                          s->col_bytes += LSB_SELL_ROWS * (s->mixed ? 4ull : 8ull);
                    }
                  }
                } else {
                  s->d_colplan_in = d, s->col_items_in = CC->nitem;
                  s->col_centre0 &= CC->centre0;
                }
                lsb_tmpl_cols_free(CC);
              }
            }
            LSB_CHK_HIP(hipStreamSynchronize(g_stream));
            lsb_sell_tmpls_free(TT);
          }
        } else {
          s->d_sbase = (int *)dev_upload(H->sbase, 2 * ((size_t)H->stored / LSB_SELL_ROWS + 1) * sizeof(int));
          s->d_svals16 = s->mixed ? (double *)upload_f32(H->vals, (size_t)H->stored + LSB_SELL_ROWS, NULL)
                                  : (double *)dev_upload(H->vals, ((size_t)H->stored + LSB_SELL_ROWS) * sizeof(double));
          s->sell_vslots = s->sell_slots = (unsigned)(H->stored / LSB_SELL_ROWS);
          s->sell16_bytes = (unsigned long long)s->sell_slots * 8;
        }
        /* what one launch of the 16-bit form streams besides x and y: slot records (and
         * constants), the code arrays, the kept values, the slice offsets where it looks */
        s->sell16_bytes += (unsigned long long)H->ncode_slots * LSB_SELL_ROWS * sizeof(short) +
                           (unsigned long long)s->sell_vslots * LSB_SELL_ROWS * (s->mixed ? 4 : 8) +
                           (s->sell_ulen ? 0ull : ((unsigned long long)H->nslice + 1) * 4);

        LSB_CHK_HIP(hipStreamSynchronize(g_stream));
        lsb_sell_vc_free(V);
        lsb_sell_free(H);
      }
    }
  }
  precond_shard_blocks(s, offs, cols, S->vals + j0, o); /* block-Jacobi: dense diagonal blocks */
  LSB_CHK_HIP(hipStreamSynchronize(g_stream)); /* host staging is freed next */
  free(rb), free(offs), free(cols), free(lanes);

  { /* the vector slab: everything this configuration's iteration streams, in one allocation */
    size_t cnt = 3 * (size_t)n + n_glob;                       /* r, q, the diagonal; the gather vector */
    if (n_glob > n || o->nvirt > 1 || o->krylov == LSB_KRYLOV_PCG1)
      cnt += 2 * (size_t)n;                                    /* p, s of the single-reduction form */
    if (o->precond == LSB_PRECOND_CHEBYSHEV)
      cnt += 2 * (size_t)n_glob + (size_t)n;                   /* two gather vectors of z, the recurrence's d */
    else if (o->precond == LSB_PRECOND_BLOCKJACOBI)
      cnt += (size_t)n;                                        /* z */
    else if (o->precond == LSB_PRECOND_FSAI)
      cnt += 4 * (size_t)n;                                    /* z, t = G r, the second r / p buffers */
    /* ... where the vectors are of the Infinity Cache's scale or below (n <= 20 M rows): that is where
     * their relative placement decides which lines fight for the same sets.  Vectors that fit no cache
     * gain nothing from it and measured 5 % SLOWER out of one allocation (64 M-row 7-point operator:
     * 1098-1163 us per iteration against 1067-1096 from separate allocations, SpMV 300-305 against
     * 283-293 us, whatever padding was put between them; profiles/r04_placement.txt) */
    if (o->krylov != LSB_KRYLOV_GMRES && !getenv("LSBENCH_HIP_NO_SLAB") && (size_t)n * sizeof(double) <= ((size_t)160 << 20)) {
      s->slab_cap = cnt * sizeof(double) + 16 * 256;
      if (hipMalloc((void **)&s->d_slab, s->slab_cap) != hipSuccess) /* no room for one piece: pieces then */
        s->d_slab = NULL, s->slab_cap = 0, (void)hipGetLastError();
    }
  }
  s->d_r = shard_vec(s, n);
  s->d_q = shard_vec(s, n);
  s->d_pfull = shard_vec(s, n_glob);
  s->d_dinv = shard_vec(s, n);
  LSB_CHK_HIP(hipMemsetAsync(s->d_pfull, 0, (size_t)n_glob * sizeof(double), g_stream));
  s->d_parts_pq = (double *)lsb_hip_malloc(3 * LSB_MAX_PARTIALS * sizeof(double));
  /* two buffers: k_cg1_update reads the previous launch's partials while
   * writing its own */
  s->d_parts2 = (double *)lsb_hip_malloc(4 * LSB_MAX_PARTIALS * sizeof(double));
  s->d_st = (struct lsb_pcg_state *)lsb_hip_malloc(sizeof(struct lsb_pcg_state));
  LSB_CHK_HIP(hipMemsetAsync(s->d_st, 0, sizeof(struct lsb_pcg_state), g_stream));
  s->d_st_aux = (struct lsb_pcg_state *)lsb_hip_malloc(sizeof(struct lsb_pcg_state));
  LSB_CHK_HIP(hipMemsetAsync(s->d_st_aux, 0, sizeof(struct lsb_pcg_state), g_stream));
  choose_spmv(s, o);

  if (o->precond == LSB_PRECOND_JACOBI || o->precond == LSB_PRECOND_L1JACOBI ||
      o->precond == LSB_PRECOND_CHEBYSHEV) { /* Chebyshev is a polynomial in D^-1 S */
    int *d_nz = (int *)lsb_hip_malloc(sizeof(int)), nz = 0;
    LSB_CHK_HIP(hipMemsetAsync(d_nz, 0, sizeof(int), g_stream));
    if (o->precond == LSB_PRECOND_L1JACOBI)
      lsb_k_l1_setup(n, s->d_offs, s->d_vals, s->d_dinv, d_nz, g_stream);
    else
      lsb_k_jacobi_setup(n, row_begin, s->d_offs, s->d_cols, s->d_vals, s->d_dinv, d_nz,
                         g_stream);
    LSB_CHK_HIP(hipMemcpyAsync(&nz, d_nz, sizeof(int), hipMemcpyDeviceToHost, g_stream));
    LSB_CHK_HIP(hipStreamSynchronize(g_stream));
    lsb_hip_free(d_nz);
    if (nz)
      errx(EXIT_FAILURE, "hip_cdna4: %d rows have no non-zero %s; "
                         "Jacobi preconditioning needs one (cf. the stored-diagonal "
                         "assumption of src/cholmod-impl.h:13)", nz,
           o->precond == LSB_PRECOND_L1JACOBI ? "entry" : "diagonal entry");
  } else {
    /* dinv = 1: unpreconditioned CG through the same kernels */
    double *ones = (double *)malloc((size_t)(n ? n : 1) * sizeof(double));
    for (unsigned i = 0; i < n; i++)
      ones[i] = 1.0;
    LSB_CHK_HIP(hipMemcpy(s->d_dinv, ones, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    free(ones);
  }
  /* One value for every row (no preconditioner; the Laplacians' constant
   * diagonal)?  Then the fused sweeps take it as a kernel argument. */
  if (n && !getenv("LSBENCH_HIP_NO_UNIFORM_DINV")) {
    double *h = (double *)malloc((size_t)n * sizeof(double));
    LSB_CHK_HIP(hipMemcpy(h, s->d_dinv, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    int same = 1;
    for (unsigned i = 1; i < n && same; i++)
      same = h[i] == h[0];
    s->dinv_uniform = same, s->dinv_const = h[0];
    free(h);
  }
}

void shard_free(struct shard *s) {
  lsb_hip_free(s->d_offs), lsb_hip_free(s->d_cols), lsb_hip_free(s->d_vals), lsb_hip_free(s->d_vals32);
  lsb_hip_free(s->d_rowblk), lsb_hip_free(s->d_blklanes);
  shard_vec_free(s, s->d_dinv), shard_vec_free(s, s->d_r);
  shard_vec_free(s, s->d_q), shard_vec_free(s, s->d_pfull), lsb_hip_free(s->d_parts_pq);
  shard_vec_free(s, s->d_p1), shard_vec_free(s, s->d_s1);
  lsb_hip_free(s->d_parts2), lsb_hip_free(s->d_st), lsb_hip_free(s->d_st_aux);
  lsb_hip_free(s->pd_offs), lsb_hip_free(s->pd_cols), lsb_hip_free(s->pd_vals);
  lsb_hip_free(s->pd_rowmap), lsb_hip_free(s->pd_rowblk), lsb_hip_free(s->pd_blklanes);
  lsb_hip_free(s->d_sptr), lsb_hip_free(s->d_scols), lsb_hip_free(s->d_svals);
  lsb_hip_free(s->d_sptr16), lsb_hip_free(s->d_scodes), lsb_hip_free(s->d_sbase);
  lsb_hip_free(s->d_svals16), lsb_hip_free(s->d_svconst);
  lsb_hip_free(s->d_srec), lsb_hip_free(s->d_tmpl), lsb_hip_free(s->d_tmask), lsb_hip_free(s->d_colplan), lsb_hip_free(s->d_colplan_in);
  free(s->h_pblk);
  lsb_hip_free(s->bd_chunk), lsb_hip_free(s->bd_rows), lsb_hip_free(s->bd_cols);
  lsb_hip_free(s->bd_vals);
  free(s->h_binchunk);
  lsb_hip_free(s->tp_item), lsb_hip_free(s->tp_binptr), lsb_hip_free(s->tp_first);
  lsb_hip_free(s->tp_mask), lsb_hip_free(s->tp_delta);
  lsb_hip_free(s->tp_colw), lsb_hip_free(s->tp_roww), lsb_hip_free(s->tp_vals);
  lsb_hip_free(s->tp_prod), lsb_hip_free(s->tp_binparts);
  precond_free_shard(s);
  lsb_hip_free(s->d_slab); /* behind everything that may point into it */
  s->d_slab = NULL;
  free(s->recv), free(s->send);
}

void plan_exchange(struct shard *s, int me, int nall, const unsigned *hull) {
  s->recv = lsb_calloc(struct lsb_xfer, nall);
  s->send = lsb_calloc(struct lsb_xfer, nall);
  lsb_plan_exchange(me, nall, hull, s->recv, &s->nrecv, s->send, &s->nsend);
}

lsb_hip_solver *solver_alloc(int nshard, const struct lsb_hip_opts *o) {
  lsb_hip_solver *sv = lsb_calloc(lsb_hip_solver, 1);
  sv->nshard = nshard;
  sv->sh = lsb_calloc(struct shard, nshard);
  sv->o = *o;
  LSB_CHK_HIP(hipHostMalloc((void **)&sv->h_st, 2 * sizeof(struct lsb_pcg_state), 0));
  sv->env_no_fuse_p = getenv("LSBENCH_HIP_NO_FUSE_P") != NULL, sv->env_no_fuse_px = getenv("LSBENCH_HIP_NO_FUSE_PX") != NULL;
  return sv;
}

void solver_finish_setup(lsb_hip_solver *sv) {
  sv->d_scal_all = (double *)lsb_hip_malloc((size_t)sv->nshard * SCAL_STRIDE * sizeof(double));
  LSB_CHK_HIP(hipMemsetAsync(sv->d_scal_all, 0,
                             (size_t)sv->nshard * SCAL_STRIDE * sizeof(double), g_stream));
  for (int i = 0; i < sv->nshard; i++)
    sv->sh[i].d_scal = sv->d_scal_all + (size_t)i * SCAL_STRIDE;
  sv->d_tmp = (double *)lsb_hip_malloc((size_t)sv->n_here * sizeof(double));
  for (int i = 0; i < 4 * MAX_SAMPLES; i++)
    LSB_CHK_HIP(hipEventCreate(&sv->ev[i]));
  LSB_CHK_HIP(hipEventCreate(&sv->ev_t0));
  LSB_CHK_HIP(hipEventCreate(&sv->ev_t1));
  LSB_CHK_HIP(hipEventCreateWithFlags(&sv->ev_poll[0], hipEventDisableTiming));
  LSB_CHK_HIP(hipEventCreateWithFlags(&sv->ev_poll[1], hipEventDisableTiming));
  LSB_CHK_HIP(hipEventCreateWithFlags(&sv->ev_vec, hipEventDisableTiming));
  LSB_CHK_HIP(hipEventCreateWithFlags(&sv->ev_halo, hipEventDisableTiming));
  sv->have_events = 1;
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
  for (int i = 0; i < sv->nshard; i++)
    tune_spmv(sv, &sv->sh[i]);
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
  /* one constant Jacobi diagonal on every shard of every rank? */
  {
    const double c = sv->sh[0].dinv_const;
    int same = 1;
    for (int i = 0; i < sv->nshard; i++)
      same &= sv->sh[i].dinv_uniform && sv->sh[i].dinv_const == c;
    if (sv->dist) {
      const int P = lsb_hip_comm_size();
      unsigned mine[3], *all = lsb_calloc(unsigned, 3 * (size_t)P);
      mine[0] = (unsigned)same;
      memcpy(mine + 1, &c, sizeof c);
      lsb_hip_comm_allgather_u32(mine, 3, all);
      for (int q = 0; q < P; q++)
        same &= all[3 * q] && all[3 * q + 1] == all[1] && all[3 * q + 2] == all[2];
      free(all);
    }
    sv->cg1_implicit = same;
  }
  tune_blas1_nt(sv);
  precond_setup(sv);
  sv->overlap_on = -1;
  p2p_setup(sv);
  overlap_setup(sv);
  persist_setup(sv);
}

lsb_hip_solver *lsb_hip_solver_create(const struct csr *A,
                                      const struct lsb_hip_opts *o_in) {
  if (!lsb_initialized || !A || A->nrows == 0)
    return NULL;
  struct lsb_hip_opts o;
  if (o_in)
    o = *o_in;
  else
    lsb_hip_get_opts(&o);
  if (o.precision == LSB_PREC_MIXED && o.krylov == LSB_KRYLOV_GMRES) {
    warnx("hip_cdna4: mixed precision is an iterative refinement around CG; GMRES runs in fp64");
    o.precision = LSB_PREC_FP64;
  }
  /* the operator, 0-based, both triangles */
  struct csr *S = o.op_mode == LSB_OP_CHOLMOD_UPPER ? lsb_csr_symmetrize_upper(A)
                                                    : lsb_csr_copy_base0(A);
  const unsigned n_user = S->nrows;
  /* a constant-coefficient 2-D grid whose lines are not whole slices: pad the lines (lsb_csr_pad_lines) --
   * the +-nx diagonals become whole-slice offsets and the z-column forms apply along y.  b and x
   * travel through the same gather / scatter as a re-ordering's; the pad unknowns stay exactly 0.
   * LSBENCH_HIP_PAD_LINES = 0 never, 1 whatever the size; default: operators of >= 1 M rows. */
  int *padmap = NULL;
  {
    const char *e = getenv("LSBENCH_HIP_PAD_LINES");
    const int want = e ? atoi(e) : (S->nrows >= 1000000u ? 1 : 0);
    if (want && !o.reorder) {
      unsigned nx = 0, nxp = 0;
      struct csr *Sp = lsb_csr_pad_lines(S, LSB_SELL_ROWS, &nx, &nxp, &padmap);
      if (Sp) {
        if (o.verbose)
          fprintf(stderr, "hip_cdna4: grid lines of %u rows padded to %u (%u -> %u rows)\n", nx, nxp, S->nrows,
                  Sp->nrows);
        lsb_csr_free(S);
        S = Sp;
      }
    }
  }
  int P = o.nvirt > 1 ? o.nvirt : 1;
  if ((unsigned)P > S->nrows / 2)
    P = 1;
  lsb_hip_solver *sv = solver_alloc(P, &o);
  sv->n_user = n_user;
  if (padmap) { /* map[internal row] = the caller's row, -1 on pad rows */
    sv->d_perm = (int *)dev_upload(padmap, (size_t)S->nrows * sizeof(int));
    LSB_CHK_HIP(hipStreamSynchronize(g_stream));
    free(padmap);
    sv->d_bp = (double *)lsb_hip_malloc((size_t)S->nrows * sizeof(double));
    sv->d_xp = (double *)lsb_hip_malloc((size_t)S->nrows * sizeof(double));
    sv->padded = 1;
  }
  if (o.reorder) {
    /* Q = RCM(S); S <- Q S Q^T (src/cusparse.c:67-97) */
    unsigned *perm = (unsigned *)malloc((size_t)S->nrows * sizeof(unsigned));
    if (!perm || lsb_csr_rcm(S, perm))
      errx(EXIT_FAILURE, "hip_cdna4: out of memory computing the RCM ordering");
    struct csr *Sp = lsb_csr_permute_sym(S, perm);
    if (o.verbose)
      fprintf(stderr, "hip_cdna4: RCM bandwidth %u -> %u\n", lsb_csr_bandwidth(S),
              lsb_csr_bandwidth(Sp));
    lsb_csr_free(S);
    S = Sp;
    sv->d_perm = (int *)dev_upload(perm, (size_t)S->nrows * sizeof(int));
    LSB_CHK_HIP(hipStreamSynchronize(g_stream));
    free(perm);
    sv->d_bp = (double *)lsb_hip_malloc((size_t)S->nrows * sizeof(double));
    sv->d_xp = (double *)lsb_hip_malloc((size_t)S->nrows * sizeof(double));
  }
  sv->n_glob = sv->n_here = S->nrows, sv->row_first = 0;
  sv->dist = 0, sv->multi = P > 1;
  unsigned *bounds = lsb_calloc(unsigned, (size_t)P + 1);
  lsb_csr_partition_rows(S, (unsigned)P, bounds);
  unsigned *hull = lsb_calloc(unsigned, 4 * (size_t)P);
  for (int q = 0; q < P; q++) {
    shard_upload(&sv->sh[q], S, bounds[q], bounds[q + 1], 0, bounds[q], S->nrows, &o);
    hull[4 * q] = bounds[q], hull[4 * q + 1] = bounds[q + 1] - bounds[q];
    hull[4 * q + 2] = sv->sh[q].col_lo, hull[4 * q + 3] = sv->sh[q].col_hi;
  }
  for (int q = 0; q < P; q++) {
    plan_exchange(&sv->sh[q], q, P, hull);
    for (int k = 0; k < sv->sh[q].nrecv; k++)
      if (sv->sh[q].recv[k].count > sv->agree_halo)
        sv->agree_halo = (unsigned)sv->sh[q].recv[k].count;
  }
  free(hull), free(bounds);
  lsb_csr_free(S);
  solver_finish_setup(sv);
  return sv;
}

lsb_hip_solver *lsb_hip_solver_create_dist(const struct csr *A_rows,
                                           unsigned row_begin,
                                           unsigned n_global,
                                           const struct lsb_hip_opts *o_in) {
  if (!lsb_initialized || !A_rows)
    return NULL;
  struct lsb_hip_opts o;
  if (o_in)
    o = *o_in;
  else
    lsb_hip_get_opts(&o);
  if (o.precision == LSB_PREC_MIXED && o.krylov == LSB_KRYLOV_GMRES)
    o.precision = LSB_PREC_FP64; /* as in lsb_hip_solver_create */
  const int P = lsb_hip_comm_size(), me = lsb_hip_comm_rank();
  lsb_hip_solver *sv = solver_alloc(1, &o);
  sv->n_glob = n_global, sv->n_here = sv->n_user = A_rows->nrows, sv->row_first = row_begin;
  /* LSBENCH_HIP_DIST_ALONE=1: a communicator of ONE rank still runs the sharded
   * iteration -- exchange, all-reduce, single-reduction CG -- so that one GPU can time
   * what a rank's share costs with the communication launches in place
   * (tools/gpu_shard_floor.py) */
  const char *alone = getenv("LSBENCH_HIP_DIST_ALONE");
  sv->dist = sv->multi = P > 1 || (alone && atoi(alone) > 0);
  shard_upload(&sv->sh[0], A_rows, 0, 0, 1, row_begin, n_global, &o);
  unsigned mine[4] = {row_begin, A_rows->nrows, sv->sh[0].col_lo, sv->sh[0].col_hi};
  unsigned *hull = lsb_calloc(unsigned, 4 * (size_t)P);
  lsb_hip_comm_allgather_u32(mine, 4, hull);
  /* every rank must enqueue the SAME number of iterations between polls (the
   * collectives inside have to pair up), so the chunk size is derived from
   * numbers all ranks agree on: the largest shard */
  {
    unsigned nz = (unsigned)sv->sh[0].nnz, *allnz = lsb_calloc(unsigned, (size_t)P);
    lsb_hip_comm_allgather_u32(&nz, 1, allnz);
    for (int q = 0; q < P; q++) {
      if (allnz[q] > sv->agree_nnz)
        sv->agree_nnz = allnz[q];
      if (hull[4 * q + 1] > sv->agree_n)
        sv->agree_n = hull[4 * q + 1];
    }
    free(allnz);
  }
  /* sanity: the shards must tile [0, n_global) in rank order */
  unsigned expect = 0;
  int full = 1, equal = 1;
  for (int q = 0; q < P; q++) {
    if (hull[4 * q] != expect)
      errx(EXIT_FAILURE, "hip_cdna4: rank %d owns rows from %u, expected %u "
                         "(row ranges must tile the operator in rank order)",
           q, hull[4 * q], expect);
    expect += hull[4 * q + 1];
    full &= hull[4 * q + 2] == 0 && hull[4 * q + 3] == n_global;
    equal &= hull[4 * q + 1] == hull[1];
  }
  if (expect != n_global)
    errx(EXIT_FAILURE, "hip_cdna4: shards cover %u rows, operator has %u", expect, n_global);
  plan_exchange(&sv->sh[0], me, P, hull);
  if (P > 1 && full && equal) {
    /* every shard references every row: the north-star all-gather of x */
    sv->sh[0].nsend = 1, sv->sh[0].send[0].peer = -1;
    sv->sh[0].send[0].offset = row_begin, sv->sh[0].send[0].count = A_rows->nrows;
    sv->sh[0].nrecv = 0;
  }
  free(hull);
  {
    unsigned h = 0, *allh = lsb_calloc(unsigned, (size_t)P);
    for (int k = 0; k < sv->sh[0].nrecv; k++)
      if (sv->sh[0].recv[k].count > h)
        h = (unsigned)sv->sh[0].recv[k].count;
    lsb_hip_comm_allgather_u32(&h, 1, allh);
    for (int q = 0; q < P; q++)
      if (allh[q] > sv->agree_halo)
        sv->agree_halo = allh[q];
    free(allh);
  }
  solver_finish_setup(sv);
  return sv;
}

void lsb_hip_solver_destroy(lsb_hip_solver *sv) {
  if (!sv)
    return;
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
  drop_graphs(sv);
  if (sv->p2p) {
    if (sv->dist) /* no peer may still be storing into a mailbox that goes away */
      lsb_hip_comm_barrier();
    for (int i = 0; i < sv->nshard; i++)
      lsb_p2p_destroy(sv->p2p[i]);
    free(sv->p2p);
  }
  for (int i = 0; i < sv->nshard; i++)
    shard_free(&sv->sh[i]);
  if (sv->have_events) {
    for (int i = 0; i < 4 * MAX_SAMPLES; i++)
      LSB_CHK_HIP(hipEventDestroy(sv->ev[i]));
    LSB_CHK_HIP(hipEventDestroy(sv->ev_t0));
    LSB_CHK_HIP(hipEventDestroy(sv->ev_t1));
    LSB_CHK_HIP(hipEventDestroy(sv->ev_poll[0]));
    LSB_CHK_HIP(hipEventDestroy(sv->ev_poll[1]));
    LSB_CHK_HIP(hipEventDestroy(sv->ev_vec));
    LSB_CHK_HIP(hipEventDestroy(sv->ev_halo));
  }
  lsb_hip_free(sv->d_scal_all), lsb_hip_free(sv->d_tmp);
  lsb_hip_free(sv->d_vr), lsb_hip_free(sv->d_ve);
  lsb_hip_free(sv->ps.d_wgrow), lsb_hip_free(sv->ps.d_ug), lsb_hip_free(sv->ps.d_shared);
  lsb_hip_free(sv->d_perm), lsb_hip_free(sv->d_bp), lsb_hip_free(sv->d_xp);
  for (int i = 0; sv->gm && i < sv->nshard; i++) {
    lsb_hip_free(sv->gm[i].V), lsb_hip_free(sv->gm[i].parts);
    lsb_hip_free(sv->gm[i].ax), lsb_hip_free(sv->gm[i].st);
  }
  free(sv->gm);
  lsb_hip_free(sv->gm_red);
  if (sv->gm_hst)
    LSB_CHK_HIP(hipHostFree(sv->gm_hst));
  LSB_CHK_HIP(hipHostFree(sv->h_st));
  free(sv->sh), free(sv);
}

unsigned lsb_hip_solver_nrows_local(const lsb_hip_solver *s) { return s->n_user; } /* (n_here counts pad rows) */
int lsb_hip_solver_padded(const lsb_hip_solver *s) { return s->padded ? (int)(s->n_here - s->n_user) : 0; }
unsigned lsb_hip_solver_nrows_global(const lsb_hip_solver *s) { return s->n_glob; }
unsigned long long lsb_hip_solver_nnz_local(const lsb_hip_solver *s) {
  unsigned long long z = 0;
  for (int i = 0; i < s->nshard; i++)
    z += s->sh[i].nnz;
  return z;
}
unsigned lsb_hip_solver_nblocks(const lsb_hip_solver *s) { return s->sh[0].nblk; }
int lsb_hip_solver_spmv_variant(const lsb_hip_solver *s) { return s->sh[0].variant; }
unsigned lsb_hip_solver_spmv_flags(const lsb_hip_solver *s) { return s->sh[0].sp_flags; }
unsigned lsb_hip_solver_spmv_grid(const lsb_hip_solver *s) { return s->sh[0].sp_grid; }
unsigned lsb_hip_solver_spmv_period(const lsb_hip_solver *s) { return s->sh[0].sp_period; }
unsigned long long lsb_hip_solver_spmv_col_slices(const lsb_hip_solver *s) {
  return s->sh[0].d_colplan ? s->sh[0].col_slices : 0ull;
}
void lsb_hip_solver_sell_value_slots(const lsb_hip_solver *s, unsigned *kept, unsigned *total) {
  const struct shard *h = &s->sh[0];
  const int on = h->variant == LSB_SPMV_SELL && (h->sp_flags & LSB_SP_C16) && h->d_scodes;
  if (kept)
    *kept = on ? h->sell_vslots : 0;
  if (total)
    *total = on ? h->sell_slots : 0;
}
void lsb_hip_solver_comm_plan(const lsb_hip_solver *s, unsigned long long plan[8]) {
  const struct shard *h = &s->sh[0];
  memset(plan, 0, 8 * sizeof plan[0]);
  plan[0] = (unsigned long long)lsb_hip_comm_count();
  plan[6] = (unsigned long long)s->nshard;
  if (!s->multi)
    return;
  plan[5] = h->nsend == 1 && h->send[0].peer == -1;
  if (plan[5]) { /* in-place all-gather: count doubles per rank */
    const unsigned long long P = s->dist ? (unsigned long long)lsb_hip_comm_size() : (unsigned long long)s->nshard;
    plan[1] = plan[2] = P - 1;
    plan[3] = (P - 1) * h->send[0].count * 8ull, plan[4] = h->send[0].count * 8ull;
  } else {
    plan[1] = (unsigned long long)h->nrecv, plan[2] = (unsigned long long)h->nsend;
    for (int k = 0; k < h->nrecv; k++)
      plan[3] += h->recv[k].count * 8ull;
    for (int k = 0; k < h->nsend; k++)
      plan[4] += h->send[k].count * 8ull;
  }
  plan[7] = (unsigned long long)can_overlap(s);
}
int lsb_hip_solver_overlap(const lsb_hip_solver *s, double us[2]) {
  if (us)
    us[0] = s->overlap_us[0], us[1] = s->overlap_us[1];
  return can_overlap(s);
}
int lsb_hip_solver_fused_p(const lsb_hip_solver *s) { return lsb_fuse_p_kind(s); }
int lsb_hip_solver_blas1_nt(const lsb_hip_solver *s) { return s->nt_mask; }
unsigned long long lsb_hip_solver_spmv_layout_bytes(const lsb_hip_solver *s) {
  const struct shard *h = &s->sh[0];
  unsigned long long m = 12ull * h->nnz + 4ull * ((unsigned long long)h->n + 1); /* the CSR arrays */
  if (h->variant == LSB_SPMV_SELL)
    m = (h->sp_flags & LSB_SP_C16) && h->d_scodes
            ? ((h->sp_flags & LSB_SP_TMPL) && h->d_srec
                   ? ((h->sp_flags & LSB_SP_COL) && h->d_colplan ? h->col_bytes : h->tmpl_bytes)
                   : h->sell16_bytes)
            : h->sell32_bytes;
  else if (h->variant == LSB_SPMV_TWOPHASE || h->variant == LSB_SPMV_BINNED)
    return 0; /* more than one pass over intermediate data: no single-pass figure */
  return m + 16ull * h->n; /* + x read once, y written once */
}
int lsb_hip_solver_overlaps(const lsb_hip_solver *s) { return can_overlap(s); }
int lsb_hip_solver_comm(const lsb_hip_solver *s, double *p2p_us, double *rccl_us) {
  if (p2p_us)
    *p2p_us = s->p2p_us;
  if (rccl_us)
    *rccl_us = s->rccl_us;
  return !s->multi ? 0 : !s->p2p_on ? 1 : s->p2p_halo ? 3 : 2;
}


void sell_launch(struct shard *s, unsigned s0, unsigned ns, const double *xfull, double *y,
                        const double *xdot, double *partials, unsigned *np,
                        const struct lsb_pcg_state *st) {
  const unsigned f32 = s->mixed ? LSB_SP_F32 : 0u; /* the value arrays hold floats then */
  /* whole launches of a shard with a z-column plan (a Chebyshev epilogue and the split interior /
   * boundary launches go through k_spmv_tmpl) */
  const int col_all = s->d_colplan && s0 == 0 && ns == s->nslice;
  const int col_in = s->d_colplan_in && s0 == s->ov_s1 && ns == s->ov_s2 - s->ov_s1;
  if ((s->sp_flags & LSB_SP_C16) && (s->sp_flags & LSB_SP_TMPL) && (s->sp_flags & LSB_SP_COL) && (col_all || col_in) &&
      s->d_srec && s->d_scodes && !s->epi.zout)
    lsb_k_spmv_tmpl_col(s->sp_flags | f32, s->sp_grid, s->col_period, col_all ? s->d_colplan : s->d_colplan_in,
                        col_all ? s->col_items : s->col_items_in, s->col_centre0, s->n,
                        s->row_begin, s->n_glob, s->d_sptr16, s->d_tmask, s->d_tmpl, s->tmpl_nfar, s->d_sbase,
                        s->d_svals16, s->d_svconst, xfull, y, xdot, partials, np, st, &s->tail, g_stream);
  else if ((s->sp_flags & LSB_SP_C16) && (s->sp_flags & LSB_SP_TMPL) && s->d_srec && s->d_scodes)
    /* (a z-column flavour's grid was timed for the walk -- 3-4 workgroups per CU; the launches that go through
     * k_spmv_tmpl instead -- Chebyshev epilogue, boundary parts of a split SpMV -- take that kernel's own 6 per CU:
     * config 3 with Chebyshev(16) 941 -> 811 ms per solve) */
    lsb_k_spmv_tmpl(s->sp_flags | f32, (s->sp_flags & LSB_SP_COL) ? 1536u : s->sp_grid, s->sp_period, s->d_sptr16, s0, ns, s->n, s->row_begin, s->n_glob,
                    s->d_srec, s->d_tmask, s->d_tmpl, s->tmpl_nfar, s->d_sbase, s->d_svals16, s->d_svconst, xfull, y, xdot, partials, np, st,
                    &s->tail, &s->epi, g_stream);
  else if ((s->sp_flags & LSB_SP_C16) && s->d_scodes)
    lsb_k_spmv_sell(s->sp_flags | f32, s->sp_grid, s->sp_period, s->d_sptr16, s0, ns, s->n, s->row_begin, s->n_glob, s->d_scodes,
                    s->d_sbase, s->d_svals16, s->d_svconst, s->sell_ulen, xfull, y, xdot, partials, np, st, &s->tail, &s->epi,
                    g_stream);
  else
    lsb_k_spmv_sell((s->sp_flags & ~LSB_SP_C16) | f32, s->sp_grid, s->sp_period, s->d_sptr, s0, ns, s->n, s->row_begin, s->n_glob,
                    s->d_scols, NULL, s->d_svals, NULL, 0, xfull, y, xdot, partials, np, st, &s->tail, NULL, g_stream);
}

void spmv_shard(struct shard *s, const double *xfull, double *y,
                       const double *xdot, double *partials, unsigned *np,
                       const struct lsb_pcg_state *st) {
  if (s->variant == LSB_SPMV_PANEL) {
    /* y = 0, then one launch per column panel accumulates into it: inside a
     * launch every XCD gathers from the same 2 MiB slice of x, out of its L2 */
    LSB_CHK_HIP(hipMemsetAsync(y, 0, (size_t)s->n * sizeof(double), g_stream));
    for (unsigned p = 0; p < s->pn; p++) {
      const unsigned b0 = s->h_pblk[p], nb = s->h_pblk[p + 1] - b0;
      if (nb)
        lsb_k_spmv(LSB_SPMV_ADAPTIVE, s->n, s->pd_offs, s->pd_cols, s->pd_vals,
                   s->pd_rowblk + b0, s->pd_blklanes + b0, nb, s->lanes, s->sp_flags,
                   s->sp_grid, xfull, y, NULL, NULL, NULL, st, s->pd_rowmap, NULL, g_stream);
    }
    if (partials)
      lsb_k_dot(s->n, y, xdot, partials, np, g_stream);
    return;
  }
  if (s->variant == LSB_SPMV_SELL) {
    sell_launch(s, 0, s->nslice, xfull, y, xdot, partials, np, st);
    return;
  }
  if (s->variant == LSB_SPMV_TWOPHASE) {
    lsb_k_spmv_twophase(s->tp_items, s->tp_item, s->tp_vals, s->tp_colw, s->tp_first, s->tp_mask,
                        s->tp_delta, s->tp_roww, s->tp_col_lo, s->tp_cols, s->tp_rows, s->tp_bins,
                        s->tp_binptr, s->tp_prod, s->n, xfull, s->tp_xlen, y, xdot, partials, np,
                        s->tp_binparts, st, g_stream);
    return;
  }
  if (s->variant == LSB_SPMV_BINNED) {
    /* y = 0, then one launch per bin (= per L2-sized window of x) adds to it */
    LSB_CHK_HIP(hipMemsetAsync(y, 0, (size_t)s->n * sizeof(double), g_stream));
    for (unsigned b = 0; b < s->bn; b++)
      lsb_k_spmv_binned(s->sp_flags, s->bcap, s->bd_chunk, s->h_binchunk[b],
                        s->h_binchunk[b + 1] - s->h_binchunk[b], s->bd_rows, s->bd_cols, s->bd_vals,
                        xfull, y, st, g_stream);
    if (partials)
      lsb_k_dot(s->n, y, xdot, partials, np, g_stream);
    return;
  }
  const int f32 = s->mixed && (s->variant == LSB_SPMV_ADAPTIVE || s->variant == LSB_SPMV_SUBWAVE);
  lsb_k_spmv(s->variant, s->n, s->d_offs, s->d_cols, f32 ? (const double *)s->d_vals32 : s->d_vals,
             s->d_rowblk, s->d_blklanes, s->nblk, s->lanes, s->sp_flags | (f32 ? LSB_SP_F32 : 0u),
             s->sp_grid, xfull, y, xdot, partials, np, st, NULL, &s->tail, g_stream);
}

void spmv_shard_exact(struct shard *s, const double *xfull, double *y, const double *xdot,
                      double *partials, unsigned *np, const struct lsb_pcg_state *st) {
  if (!s->mixed) {
    spmv_shard(s, xfull, y, xdot, partials, np, st);
    return;
  }
  /* the CSR arrays always stay: row-blocked kernel (sub-wavefront for small operators) */
  const int v = s->nnz <= 500000ull ? LSB_SPMV_SUBWAVE : LSB_SPMV_ADAPTIVE;
  lsb_k_spmv(v, s->n, s->d_offs, s->d_cols, s->d_vals, s->d_rowblk, s->d_blklanes, s->nblk, s->lanes,
             LSB_SP_PREFETCH | LSB_SP_NT, 0, xfull, y, xdot, partials, np, st, NULL, NULL, g_stream);
}

/*
 * Pick the SpMV form and flavour for this shard by timing them (setup is
 * untimed, like the reference's csr_init): every form the shard has -- the
 * row-blocked CSR kernel {plain, prefetch, nontemporal, both}, the column-panel
 * form of scattered shards, the sliced-ELL copies {32-bit columns, 16-bit codes}
 * x {plain, nontemporal} x {8, 6 resident workgroups per CU} -- 3 launches each
 * after one warm-up, on the shard's own matrix with the dot product fused as
 * in the solve.  Which one wins depends on the operator: stencils and banded
 * meshes take the 16-bit sliced-ELL form (10 M-row 5-point: 104 us against
 * 152 us row-blocked), ragged rows have no sliced-ELL copy at all, and whether
 * nontemporal stream loads pay depends on how much of x's gather window
 * survives next to the matrix stream (64 M-row 7-point: plain loads).  The
 * copies that lose are freed.
 */
void tune_spmv(lsb_hip_solver *sv, struct shard *s) {
  const struct lsb_hip_opts *o = &sv->o;
  s->sp_flags = LSB_SP_PREFETCH | LSB_SP_NT;
  s->sp_grid = o->spmv_grid > 0 ? (unsigned)o->spmv_grid : LSB_MAX_PARTIALS;
  if (getenv("LSBENCH_HIP_FORCE_PERIOD")) /* tests: the plane-periodic dealing on small operators */
    s->sp_period = s->sell_period;
  if (o->spmv_tune >= 0) {
    s->sp_flags = (unsigned)o->spmv_tune & (31u | LSB_SP_TMPL | LSB_SP_DEFER | LSB_SP_COL); /* bit 2: 16-bit codes, where that copy exists;
                                                   bits 3, 4: binned form's gather flavour; bit 6: slice
                                                   templates, where the constant-slot layout has them */
    return;
  }
  if (s->variant == LSB_SPMV_SELL && !s->d_sptr)
    s->variant = LSB_SPMV_ADAPTIVE; /* the operator did not qualify for the copy */
  if ((s->variant != LSB_SPMV_ADAPTIVE && s->variant != LSB_SPMV_PANEL &&
       s->variant != LSB_SPMV_SELL && s->variant != LSB_SPMV_BINNED &&
       s->variant != LSB_SPMV_TWOPHASE) ||
      s->nnz < 4000000ull)
    return; /* small operators are launch-latency bound: nothing to tune */
  float best = 1e30f;
  unsigned bf = s->sp_flags, np;
  int bv = s->variant;
  const unsigned grid0 = s->sp_grid;
  unsigned bg = grid0, bp = 0;
  /* candidates {form, flags, grid}: the form asked for, or (auto) every form
   * this shard has; the sliced-ELL kernels have no prefetch flavour, they try
   * 6 instead of 8 resident workgroups per CU instead */
  struct {
    int v;
    unsigned f, g, p;
  } cand[32];
  memset(cand, 0, sizeof cand);
  int ncand = 0;
  const int any = o->spmv_variant == LSB_SPMV_AUTO;
  /* (the bound is checked BEFORE every write) */
#define CAND(V, F, G, P)                                                                       \
  do {                                                                                         \
    if (ncand >= (int)(sizeof cand / sizeof cand[0]))                                          \
      errx(EXIT_FAILURE, "hip_cdna4: SpMV timing pass: candidate table too small");            \
    cand[ncand].v = (V), cand[ncand].f = (F), cand[ncand].g = (G), cand[ncand].p = (P), ncand++; \
  } while (0)
  if (any || s->variant == LSB_SPMV_ADAPTIVE)
    for (unsigned f = 0; f < 4; f++)
      CAND(LSB_SPMV_ADAPTIVE, f, grid0, 0);
  if (s->pn && (any || s->variant == LSB_SPMV_PANEL))
    for (unsigned f = 0; f < 4; f++)
      CAND(LSB_SPMV_PANEL, f, grid0, 0);
  if (s->tp_bins && (any || s->variant == LSB_SPMV_TWOPHASE))
    CAND(LSB_SPMV_TWOPHASE, 0, grid0, 0);
  if (s->bn && (any || s->variant == LSB_SPMV_BINNED)) {
    /* stream loads {nontemporal, plain}; the gather of x stays a plain load: L1-
     * bypassing (sc1) gathers measured the same, nontemporal ones 1.7x slower
     * (flags 8 / 16, kept for experiments through opts.spmv_tune) */
    static const unsigned bf[] = {LSB_SP_NT, 0};
    for (unsigned k = 0; k < sizeof bf / sizeof bf[0]; k++)
      CAND(LSB_SPMV_BINNED, bf[k], grid0, 0);
  }
  const int periodic = s->sell_period != 0;
  if (s->d_sptr && (any || s->variant == LSB_SPMV_SELL))
    for (unsigned c16 = 0; c16 <= (s->d_scodes ? LSB_SP_C16 : 0u); c16 += LSB_SP_C16) {
      CAND(LSB_SPMV_SELL, c16 | LSB_SP_NT, grid0, 0);
      CAND(LSB_SPMV_SELL, c16, grid0, 0);
      if (o->spmv_grid <= 0)
        CAND(LSB_SPMV_SELL, c16 | LSB_SP_NT, 1536, 0);
      if (periodic) { /* every XCD an eighth of every plane */
        CAND(LSB_SPMV_SELL, c16 | LSB_SP_NT, grid0, s->sell_period);
        CAND(LSB_SPMV_SELL, c16, grid0, s->sell_period);
      }
      if (c16 && s->d_srec) { /* slice templates (no stream to load nontemporally: NT only marks the
                                 flavour as "solve-like" for the 3 % rule below) */
        const unsigned f = c16 | LSB_SP_NT | LSB_SP_TMPL;
        CAND(LSB_SPMV_SELL, f, grid0, 0);
        if (o->spmv_grid <= 0)
          CAND(LSB_SPMV_SELL, f, 1536, 0);
        if (periodic) {
          CAND(LSB_SPMV_SELL, f, grid0, s->sell_period);
          if (o->spmv_grid <= 0)
            CAND(LSB_SPMV_SELL, f, 1536, s->sell_period);
        }
        /* the same with y parked in LDS and stored one turn later (k_spmv_tmpl<.., DEFER>): pays where
         * the vectors come out of HBM, costs ~2 us where they sit in the Infinity Cache */
        if (s->tmpl_nfar >= 1 && s->nnz >= 16000000ull) {
          const unsigned fd = f | LSB_SP_DEFER, gd = o->spmv_grid <= 0 ? 1536u : grid0;
          CAND(LSB_SPMV_SELL, fd, gd, 0);
          if (periodic)
            CAND(LSB_SPMV_SELL, fd, gd, s->sell_period);
        }
        /* the z-column walk of a 3-D stencil (k_spmv_tmpl_col): one new plane per step */
        /* (fewer resident workgroups than the other flavours: 3-4 per CU measured 232-233 us on the
         * 64 M-row 7-point operator, 5 / 6 / 8 per CU 246 / 246 / 253, profiles/r04_col.txt) */
        if (s->d_colplan) {
          if (o->spmv_grid <= 0) {
            CAND(LSB_SPMV_SELL, f | LSB_SP_COL, 768, 0);
            CAND(LSB_SPMV_SELL, f | LSB_SP_COL, 1024, 0);
            CAND(LSB_SPMV_SELL, f | LSB_SP_COL, 1536, 0);
          } else
            CAND(LSB_SPMV_SELL, f | LSB_SP_COL, grid0, 0);
        }
      }
    }
#undef CAND
  for (int ci = 0; ci < ncand; ci++) {
    s->variant = cand[ci].v, s->sp_flags = cand[ci].f, s->sp_grid = cand[ci].g, s->sp_period = cand[ci].p;
    spmv_shard(s, s->d_pfull, s->d_q, s->d_pfull + s->row_begin, s->d_parts_pq, &np, NULL);
    LSB_CHK_HIP(hipEventRecord(sv->ev_t0, g_stream));
    const int reps = 5;
    for (int r = 0; r < reps; r++)
      spmv_shard(s, s->d_pfull, s->d_q, s->d_pfull + s->row_begin, s->d_parts_pq, &np, NULL);
    LSB_CHK_HIP(hipEventRecord(sv->ev_t1, g_stream));
    LSB_CHK_HIP(hipEventSynchronize(sv->ev_t1));
    float ms = 0.f;
    LSB_CHK_HIP(hipEventElapsedTime(&ms, sv->ev_t0, sv->ev_t1));
    ms *= 3.0f / reps; /* (the verbose line below quotes per-launch time as ms / 3) */
    if (o->verbose > 1)
      fprintf(stderr, "hip_cdna4: spmv tune form=%d flags=%u grid=%u period=%u: %.1f us\n", s->variant,
              s->sp_flags, s->sp_grid, s->sp_period, ms * 1e3f / 3);
    /* back-to-back launches flatter plain stream loads: inside a solve the sweeps'
     * vectors compete for the Infinity Cache with the matrix stream and the
     * nontemporal flavour runs 5-10 % faster than its timing here says (64 M-row
     * 7-point: 0.914 ms against 0.966-1.03 ms in the solve at equal times back to
     * back) -- a plain flavour has to win by 3 % to be taken */
    if (!(s->sp_flags & LSB_SP_NT))
      ms *= 1.03f;
    /* the z-column walk is also what the two-launch iteration runs on (k_pcg_col_px + k_pcg_col_r: 8 vector passes
     * instead of 11): where that form will apply -- one shard, Jacobi with a constant diagonal, classic PCG, fp64 --
     * a flavour without it has to beat the walk by 15 % as an SpMV to be worth the three launches */
    if ((s->sp_flags & LSB_SP_COL) && s->d_colplan && sv->nshard == 1 && !sv->dist && !sv->multi && s->dinv_uniform &&
        !s->mixed && o->precond == LSB_PRECOND_JACOBI && (o->krylov == LSB_KRYLOV_PCG || o->krylov == LSB_KRYLOV_AUTO) &&
        !sv->env_no_fuse_px)
      ms *= 0.85f;
    if (ms < best)
      best = ms, bf = s->sp_flags, bv = s->variant, bg = s->sp_grid, bp = s->sp_period;
  }
  s->sp_grid = bg;
  s->sp_period = bp;
  s->variant = bv;
  s->sp_flags = bf;
  /* the copies that lost are not kept */
  if (any && bv != LSB_SPMV_TWOPHASE && s->tp_bins) {
    lsb_hip_free(s->tp_item), lsb_hip_free(s->tp_binptr), lsb_hip_free(s->tp_first);
    lsb_hip_free(s->tp_mask), lsb_hip_free(s->tp_delta);
    lsb_hip_free(s->tp_colw), lsb_hip_free(s->tp_roww), lsb_hip_free(s->tp_vals);
    lsb_hip_free(s->tp_prod), lsb_hip_free(s->tp_binparts);
    s->tp_item = s->tp_binptr = s->tp_first = s->tp_delta = NULL, s->tp_mask = NULL;
    s->tp_colw = s->tp_roww = NULL;
    s->tp_vals = s->tp_prod = s->tp_binparts = NULL, s->tp_bins = 0;
  }
  if (any && bv != LSB_SPMV_BINNED && s->bn) {
    lsb_hip_free(s->bd_chunk), lsb_hip_free(s->bd_rows), lsb_hip_free(s->bd_cols);
    lsb_hip_free(s->bd_vals);
    s->bd_chunk = s->bd_rows = s->bd_cols = NULL, s->bd_vals = NULL, s->bn = 0;
  }
  if (!(bv == LSB_SPMV_SELL && (bf & LSB_SP_C16))) {
    lsb_hip_free(s->d_sptr16), lsb_hip_free(s->d_scodes), lsb_hip_free(s->d_sbase);
    lsb_hip_free(s->d_svals16), lsb_hip_free(s->d_svconst);
    s->d_sptr16 = NULL, s->d_scodes = NULL, s->d_sbase = NULL, s->d_svals16 = NULL, s->d_svconst = NULL;
  }
  if (!(bv == LSB_SPMV_SELL && (bf & LSB_SP_TMPL))) {
    lsb_hip_free(s->d_srec), lsb_hip_free(s->d_tmpl), lsb_hip_free(s->d_tmask);
    s->d_srec = NULL, s->d_tmpl = NULL, s->d_tmask = NULL;
  }
  if (!(bv == LSB_SPMV_SELL && (bf & LSB_SP_COL))) {
    lsb_hip_free(s->d_colplan), lsb_hip_free(s->d_colplan_in);
    s->d_colplan = s->d_colplan_in = NULL;
  }
  if (any && !(bv == LSB_SPMV_SELL && !(bf & LSB_SP_C16))) {
    lsb_hip_free(s->d_scols), lsb_hip_free(s->d_svals);
    s->d_scols = NULL, s->d_svals = NULL;
    if (bv != LSB_SPMV_SELL)
      lsb_hip_free(s->d_sptr), s->d_sptr = NULL;
  }
}

/* y = Op x for the rows of this process */
int lsb_hip_solver_spmv_dev(lsb_hip_solver *sv, const double *d_x, double *d_y) {
  if (!lsb_initialized)
    return 1;
  if (!sv || !d_x || !d_y)
    return 2;
  double *d_yout = NULL;
  if (sv->d_perm) { /* y = Q^T (Q S Q^T) Q x */
    lsb_k_perm_gather(sv->n_here, sv->d_perm, d_x, sv->d_bp, g_stream);
    d_x = sv->d_bp, d_yout = d_y, d_y = sv->d_xp;
  }
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    LSB_CHK_HIP(hipMemcpyAsync(s->d_pfull + s->row_begin, d_x + (s->row_begin - sv->row_first),
                               (size_t)s->n * sizeof(double), hipMemcpyDeviceToDevice,
                               g_stream));
  }
  if (sv->multi)
    exchange_p(sv, 0);
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    spmv_shard_exact(s, s->d_pfull, d_y + (s->row_begin - sv->row_first), NULL, NULL, NULL, NULL);
  }
  if (d_yout)
    lsb_k_perm_scatter(sv->n_here, sv->d_perm, d_y, d_yout, g_stream);
  drain_stream(sv, "lsb_hip_solver_spmv_dev");
  check_aux_status(sv, "lsb_hip_solver_spmv_dev");
  return 0;
}

int lsb_hip_solver_time_spmv(lsb_hip_solver *sv, int warm, int reps, double *ms_avg) {
  if (!lsb_initialized)
    return 1;
  if (!sv || reps < 1 || !ms_avg)
    return 2;
  struct shard *s = &sv->sh[0];
  unsigned np;
  for (int i = 0; i < warm; i++)
    spmv_shard(s, s->d_pfull, s->d_q, s->d_pfull + s->row_begin, s->d_parts_pq, &np, NULL);
  LSB_CHK_HIP(hipEventRecord(sv->ev_t0, g_stream));
  for (int i = 0; i < reps; i++)
    spmv_shard(s, s->d_pfull, s->d_q, s->d_pfull + s->row_begin, s->d_parts_pq, &np, NULL);
  LSB_CHK_HIP(hipEventRecord(sv->ev_t1, g_stream));
  LSB_CHK_HIP(hipEventSynchronize(sv->ev_t1));
  float ms = 0.f;
  LSB_CHK_HIP(hipEventElapsedTime(&ms, sv->ev_t0, sv->ev_t1));
  *ms_avg = (double)ms / reps;
  return 0;
}

int lsb_hip_solver_jacobi_sweep_dev(lsb_hip_solver *sv, double w, const double *d_b,
                                    double *d_x) {
  if (!lsb_initialized)
    return 1;
  if (!sv || !d_b || !d_x)
    return 2;
  if (sv->d_perm) { /* re-ordered / line-padded operator: the sweep in the solver's own numbering */
    lsb_k_perm_gather(sv->n_here, sv->d_perm, d_x, sv->d_xp, g_stream);
    lsb_k_perm_gather(sv->n_here, sv->d_perm, d_b, sv->d_bp, g_stream);
    for (int i = 0; i < sv->nshard; i++) {
      struct shard *s = &sv->sh[i];
      LSB_CHK_HIP(hipMemcpyAsync(s->d_pfull + s->row_begin, sv->d_xp + (s->row_begin - sv->row_first),
                                 (size_t)s->n * sizeof(double), hipMemcpyDeviceToDevice, g_stream));
    }
    if (sv->multi)
      exchange_p(sv, 0);
    for (int i = 0; i < sv->nshard; i++) {
      struct shard *s = &sv->sh[i];
      const size_t o = s->row_begin - sv->row_first;
      spmv_shard_exact(s, s->d_pfull, sv->d_tmp + o, NULL, NULL, NULL, NULL);
      lsb_k_jacobi_sweep(s->n, w, s->d_dinv, sv->d_bp + o, sv->d_tmp + o, sv->d_xp + o, g_stream);
    }
    lsb_k_perm_scatter(sv->n_here, sv->d_perm, sv->d_xp, d_x, g_stream);
    drain_stream(sv, "lsb_hip_solver_jacobi_sweep_dev");
    check_aux_status(sv, "lsb_hip_solver_jacobi_sweep_dev");
    return 0;
  }
  int rc = lsb_hip_solver_spmv_dev(sv, d_x, sv->d_tmp);
  if (rc)
    return rc;
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    const size_t o = s->row_begin - sv->row_first;
    lsb_k_jacobi_sweep(s->n, w, s->d_dinv, d_b + o, sv->d_tmp + o, d_x + o, g_stream);
  }
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
  return 0;
}
