/*
 * Matrix container I/O of the lsbench core, rebuilt for this library:
 * lsbench_matrix_read / _print / _free with the behaviour of the reference's
 * src/lsbench-csr.c:29-108, but a different mechanism (one fread + strto*
 * tokenizer, stable counting sort by row, per-row insertion/merge sort by
 * column) so that multi-GB files load at disk speed instead of
 * fscanf+qsort speed.
 *
 * Behaviour kept identical to the reference on every input it accepts:
 *   - line 1 "<nnz> <base>\n", then nnz records "<row> <col> <value>\n"
 *     (src/lsbench-csr.c:37,50); the byte after each record must be '\n', so a
 *     missing final newline is fatal (:51-52); base > 1 and nnz == 0 are
 *     fatal (:40-43); trailing extra lines are ignored;
 *   - records may come unsorted and duplicated: sorted by (row, col), equal
 *     (row, col) summed in file order (:54-63);
 *   - nrows = number of DISTINCT row ids present, rows renumbered densely,
 *     column ids kept verbatim, offs 0-based, cols keep the base (:66-86).
 *     (An absent row id therefore shifts the matrix; the reference does not
 *     notice, we warn on stderr.)
 *   - print format "%u %u %lf\n" with row+base and the stored col (:94-99).
 * "synth:<spec>" as a file name builds a synthetic operator instead
 * (lsb_synth.c); that is an addition.
 */
#define _GNU_SOURCE
#include "lsb_impl.h"
#include <ctype.h>
#include <errno.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

struct rec {
  unsigned r, c;
  double v;
};

static const char *skip_ws(const char *p, const char *end) {
  while (p < end && isspace((unsigned char)*p))
    p++;
  return p;
}

/* %u of fscanf: optional blanks, then an unsigned decimal (sign accepted). */
static int take_uint(const char **pp, const char *end, unsigned *out) {
  const char *p = skip_ws(*pp, end);
  if (p >= end)
    return 0;
  char *q;
  errno = 0;
  unsigned long v = strtoul(p, &q, 10);
  if (q == p)
    return 0;
  *out = (unsigned)v;
  *pp = q;
  return 1;
}

/* %lf of fscanf */
static int take_double(const char **pp, const char *end, double *out) {
  const char *p = skip_ws(*pp, end);
  if (p >= end)
    return 0;
  char *q;
  double v = strtod(p, &q);
  if (q == p)
    return 0;
  *out = v;
  *pp = q;
  return 1;
}

/* %c of fscanf followed by the reference's "(ch != '\n')" test */
static int take_newline(const char **pp, const char *end) {
  if (*pp >= end || **pp != '\n')
    return 0;
  (*pp)++;
  return 1;
}

static int rec_cmp(const void *pa, const void *pb) {
  const struct rec *a = (const struct rec *)pa, *b = (const struct rec *)pb;
  if (a->r != b->r)
    return a->r < b->r ? -1 : 1;
  if (a->c != b->c)
    return a->c < b->c ? -1 : 1;
  return 0;
}

/* stable sort of t[0..n) by column (rows are already grouped) */
static void sort_row_by_col(struct rec *t, unsigned n, struct rec *tmp) {
  if (n < 2)
    return;
  int sorted = 1;
  for (unsigned i = 1; i < n && sorted; i++)
    sorted = t[i - 1].c <= t[i].c;
  if (sorted)
    return;
  if (n <= 32) {
    for (unsigned i = 1; i < n; i++) {
      struct rec k = t[i];
      unsigned j = i;
      while (j > 0 && t[j - 1].c > k.c)
        t[j] = t[j - 1], j--;
      t[j] = k;
    }
    return;
  }
  /* bottom-up merge sort (stable) */
  struct rec *src = t, *dst = tmp;
  for (unsigned w = 1; w < n; w *= 2) {
    for (unsigned lo = 0; lo < n; lo += 2 * w) {
      unsigned mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
      unsigned a = lo, b = mid, k = lo;
      while (a < mid && b < hi)
        dst[k++] = (src[b].c < src[a].c) ? src[b++] : src[a++];
      while (a < mid)
        dst[k++] = src[a++];
      while (b < hi)
        dst[k++] = src[b++];
    }
    struct rec *sw = src;
    src = dst, dst = sw;
  }
  if (src != t)
    memcpy(t, src, (size_t)n * sizeof *t);
}

static struct csr *csr_from_file(const char *fname) {
  FILE *fp = fopen(fname, "rb");
  if (!fp)
    err(EXIT_FAILURE, "Unable to open file \"%s\" for reading", fname);
  fseek(fp, 0, SEEK_END);
  long fsz = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  char *buf = (char *)malloc((size_t)fsz + 1);
  if (!buf)
    err(EXIT_FAILURE, "Unable to allocate %ld bytes for \"%s\"", fsz, fname);
  size_t got = fread(buf, 1, (size_t)fsz, fp);
  fclose(fp);
  buf[got] = '\0';
  const char *p = buf, *end = buf + got;

  unsigned nnz, base;
  if (!take_uint(&p, end, &nnz) || !take_uint(&p, end, &base) ||
      !take_newline(&p, end))
    errx(EXIT_FAILURE, "Unable to read meta information about the matrix.");
  if (base > 1)
    errx(EXIT_FAILURE, "Base should be either 0 or 1, got: %u.", base);
  if (nnz == 0)
    errx(EXIT_FAILURE, "Number of nnz values in the file are zero.");

  struct rec *in = (struct rec *)malloc((size_t)nnz * sizeof *in);
  struct rec *t = (struct rec *)malloc((size_t)nnz * sizeof *t);
  if (!in || !t)
    errx(EXIT_FAILURE, "Unable to allocate memories for %u COO entries.", nnz);
  unsigned rmax = 0;
  for (unsigned i = 0; i < nnz; i++) {
    if (!take_uint(&p, end, &in[i].r) || !take_uint(&p, end, &in[i].c) ||
        !take_double(&p, end, &in[i].v) || !take_newline(&p, end))
      errx(EXIT_FAILURE, "Unable to read matrix entries.");
    if (in[i].r > rmax)
      rmax = in[i].r;
  }
  free(buf);

  /* group by row: stable counting sort when row ids are dense enough,
   * comparison sort otherwise (e.g. ids near 2^32) */
  if ((unsigned long long)rmax <= 8ull * nnz + 1024) {
    unsigned *cnt = lsb_calloc(unsigned, (size_t)rmax + 2);
    for (unsigned i = 0; i < nnz; i++)
      cnt[in[i].r + 1]++;
    for (unsigned r = 0; r <= rmax; r++)
      cnt[r + 1] += cnt[r];
    for (unsigned i = 0; i < nnz; i++)
      t[cnt[in[i].r]++] = in[i];
    free(cnt);
    /* then by column inside each row, stable => duplicates stay in file order */
    for (unsigned s = 0; s < nnz;) {
      unsigned e = s + 1;
      while (e < nnz && t[e].r == t[s].r)
        e++;
      sort_row_by_col(t + s, e - s, in + s);
      s = e;
    }
  } else {
    memcpy(t, in, (size_t)nnz * sizeof *t);
    qsort(t, nnz, sizeof *t, rec_cmp);
  }
  free(in);

  /* merge duplicates, count distinct rows */
  unsigned m = 0, nrows = 0, gaps = 0;
  for (unsigned s = 0; s < nnz;) {
    unsigned e = s + 1;
    t[m] = t[s];
    while (e < nnz && t[e].r == t[s].r && t[e].c == t[s].c)
      t[m].v += t[e].v, e++;
    if (m == 0 || t[m].r != t[m - 1].r) {
      if (t[m].r != nrows + base)
        gaps = 1;
      nrows++;
    }
    m++, s = e;
  }
  if (gaps)
    warnx("%s: row ids are not %u..%u without gaps; rows are renumbered "
          "densely exactly as the reference does (src/lsbench-csr.c:66-70) -- "
          "the matrix is probably not the one you meant",
          fname, base, base + nrows - 1);

  struct csr *A = lsb_calloc(struct csr, 1);
  A->nrows = nrows, A->base = base;
  A->offs = lsb_calloc(unsigned, (size_t)nrows + 1);
  A->cols = lsb_calloc(unsigned, m);
  A->vals = lsb_calloc(double, m);
  unsigned row = 0;
  for (unsigned i = 0; i < m; i++) {
    if (i > 0 && t[i].r != t[i - 1].r)
      A->offs[++row] = i;
    A->cols[i] = t[i].c;
    A->vals[i] = t[i].v;
  }
  A->offs[nrows] = m;
  free(t);
  return A;
}

/*
 * Binary cache of a parsed file, opt-in:
 * LSBENCH_MATRIX_CACHE=1 keeps "<file>.lsbcsr" next to the text file,
 * LSBENCH_MATRIX_CACHE=<dir> keeps it in <dir>.  Layout: 8-byte magic, the
 * text file's size and mtime (a stale cache is ignored and rewritten), nrows,
 * base, nnz, then offs / cols / vals exactly as in struct csr.  What it holds
 * is the OUTPUT of the parser above, so every loader rule (sorting, summed
 * duplicates, dense row renumbering, kept base) is applied once, by that code.
 */
struct cache_hdr {
  char magic[8];
  long long src_size, src_mtime_ns;
  unsigned nrows, base;
  unsigned long long nnz;
};
static const char CACHE_MAGIC[8] = {'L', 'S', 'B', 'C', 'S', 'R', '1', 0};

static int cache_path(const char *fname, char *out, size_t cap) {
  const char *e = getenv("LSBENCH_MATRIX_CACHE");
  if (!e || !*e || !strcmp(e, "0"))
    return 0;
  if (!strcmp(e, "1"))
    return snprintf(out, cap, "%s.lsbcsr", fname) < (int)cap;
  const char *b = strrchr(fname, '/');
  return snprintf(out, cap, "%s/%s.lsbcsr", e, b ? b + 1 : fname) < (int)cap;
}

static struct csr *cache_load(const char *cpath, const struct stat *src) {
  FILE *f = fopen(cpath, "rb");
  if (!f)
    return NULL;
  struct cache_hdr h;
  struct csr *A = NULL;
  if (fread(&h, sizeof h, 1, f) == 1 && !memcmp(h.magic, CACHE_MAGIC, 8) &&
      h.src_size == (long long)src->st_size &&
      h.src_mtime_ns == (long long)src->st_mtim.tv_sec * 1000000000ll + src->st_mtim.tv_nsec) {
    A = lsb_calloc(struct csr, 1);
    A->nrows = h.nrows, A->base = h.base;
    A->offs = (unsigned *)malloc(((size_t)h.nrows + 1) * sizeof(unsigned));
    A->cols = (unsigned *)malloc((size_t)(h.nnz ? h.nnz : 1) * sizeof(unsigned));
    A->vals = (double *)malloc((size_t)(h.nnz ? h.nnz : 1) * sizeof(double));
    if (!A->offs || !A->cols || !A->vals ||
        fread(A->offs, sizeof(unsigned), (size_t)h.nrows + 1, f) != (size_t)h.nrows + 1 ||
        fread(A->cols, sizeof(unsigned), (size_t)h.nnz, f) != (size_t)h.nnz ||
        fread(A->vals, sizeof(double), (size_t)h.nnz, f) != (size_t)h.nnz ||
        A->offs[h.nrows] != h.nnz || h.base > 1 || A->offs[0] != 0) {
      lsbench_matrix_free(A); /* truncated or foreign file: parse the text instead */
      A = NULL;
    }
    /* a damaged file must not reach the operator build: offsets non-decreasing,
     * columns at or above the base and strictly increasing inside a row (what
     * the parser guarantees, src/lsbench-csr.c:54-63) */
    for (unsigned i = 0; A && i < h.nrows; i++) {
      int bad = A->offs[i] > A->offs[i + 1] || A->offs[i + 1] > h.nnz;
      for (unsigned j = A->offs[i]; !bad && j < A->offs[i + 1]; j++)
        bad = A->cols[j] < h.base || (j > A->offs[i] && A->cols[j] <= A->cols[j - 1]);
      if (bad) {
        lsbench_matrix_free(A);
        A = NULL;
      }
    }
  }
  fclose(f);
  return A;
}

static void cache_store(const char *cpath, const struct stat *src, const struct csr *A) {
  char tmp[4200];
  if (snprintf(tmp, sizeof tmp, "%s.tmp%d", cpath, (int)getpid()) >= (int)sizeof tmp)
    return;
  FILE *f = fopen(tmp, "wb");
  if (!f)
    return; /* a cache that cannot be written is not an error */
  struct cache_hdr h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, CACHE_MAGIC, 8);
  h.src_size = (long long)src->st_size;
  h.src_mtime_ns = (long long)src->st_mtim.tv_sec * 1000000000ll + src->st_mtim.tv_nsec;
  h.nrows = A->nrows, h.base = A->base, h.nnz = A->offs[A->nrows];
  const int ok = fwrite(&h, sizeof h, 1, f) == 1 &&
                 fwrite(A->offs, sizeof(unsigned), (size_t)A->nrows + 1, f) == (size_t)A->nrows + 1 &&
                 fwrite(A->cols, sizeof(unsigned), (size_t)h.nnz, f) == (size_t)h.nnz &&
                 fwrite(A->vals, sizeof(double), (size_t)h.nnz, f) == (size_t)h.nnz;
  if (fclose(f) != 0 || !ok || rename(tmp, cpath) != 0)
    remove(tmp);
}

struct csr *lsbench_matrix_read(const char *fname) {
  char cpath[4096];
  struct stat sb;
  if (strncmp(fname, "synth:", 6) != 0 && cache_path(fname, cpath, sizeof cpath) &&
      stat(fname, &sb) == 0) {
    struct csr *C = cache_load(cpath, &sb);
    if (C)
      return C;
    C = csr_from_file(fname);
    if (C)
      cache_store(cpath, &sb, C);
    return C;
  }
  if (strncmp(fname, "synth:", 6) == 0) {
    unsigned n;
    struct csr *A = lsbench_matrix_synth(fname + 6, 0, 0, &n);
    if (!A)
      errx(EXIT_FAILURE, "Bad synthetic matrix spec \"%s\".", fname + 6);
    return A;
  }
  return csr_from_file(fname);
}

void lsbench_matrix_print(const struct csr *A) {
  for (unsigned i = 0; i < A->nrows; i++)
    for (unsigned j = A->offs[i]; j < A->offs[i + 1]; j++)
      printf("%u %u %lf\n", i + A->base, A->cols[j], A->vals[j]);
}

void lsbench_matrix_free(struct csr *A) {
  if (!A)
    return;
  free(A->offs);
  free(A->cols);
  free(A->vals);
  free(A);
}
