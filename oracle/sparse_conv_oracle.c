/*
 * ORACLE — test infrastructure only (never linked, imported or called by the product path).
 * CPU restatement of the sparse-convolution semantics the reference obtains from spconv.pytorch
 * (SubMConv3d / SparseConv3d / SparseConvTensor.dense) at
 * pcdet/models/backbones_3d/spconv_backbone.py:77-117,141-157 and
 * pcdet/models/backbones_2d/map_to_bev/height_compression.py:20-24.
 * The arithmetic lives in the third-party wheel spconv-cu113 v2.1.21 (+cumm), absent from
 * /root/reference (README.md:54, setup.py:48): this restates its published behaviour.
 * PARITY UNPINNED by the reference (it has no tests); tests/test_oracle_spconv.py pins this file against
 * torch.nn.functional.conv3d on the densified tensor instead.
 *
 * Deliberately a different algorithm from the HIP path: sorted key arrays + binary search, double
 * accumulation, scatter-form dgrad.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int64_t key; int row; } KeyRow;

static int cmp_keyrow(const void* a, const void* b) {
  const KeyRow* x = (const KeyRow*)a; const KeyRow* y = (const KeyRow*)b;
  if (x->key < y->key) return -1;
  if (x->key > y->key) return 1;
  return x->row - y->row;
}
static int cmp_i64(const void* a, const void* b) {
  int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}
static int64_t lin(int b, int z, int y, int x, const int* s) {
  return (((int64_t)b * s[0] + z) * s[1] + y) * (int64_t)s[2] + x;
}
static int find_row(const KeyRow* t, int n, int64_t key) {
  int lo = 0, hi = n - 1;
  int found = -1;
  while (lo <= hi) {
    int mid = (lo + hi) / 2;
    if (t[mid].key < key) lo = mid + 1;
    else { if (t[mid].key == key) found = mid; hi = mid - 1; }
  }
  return found < 0 ? -1 : t[found].row;   /* first (smallest row) among duplicates */
}

/* nbr (n,K) */
int oracle_subm_nbr(const int* coords, int n, const int* shape, const int* ks, int* nbr) {
  KeyRow* t = (KeyRow*)malloc(sizeof(KeyRow) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) {
    const int* c = coords + 4 * (size_t)i;
    t[i].key = lin(c[0], c[1], c[2], c[3], shape);
    t[i].row = i;
  }
  qsort(t, n, sizeof(KeyRow), cmp_keyrow);
  const int K = ks[0] * ks[1] * ks[2];
  for (int i = 0; i < n; ++i) {
    const int* c = coords + 4 * (size_t)i;
    int o = 0;
    for (int kz = 0; kz < ks[0]; ++kz)
      for (int ky = 0; ky < ks[1]; ++ky)
        for (int kx = 0; kx < ks[2]; ++kx, ++o) {
          int z = c[1] + kz - ks[0] / 2, y = c[2] + ky - ks[1] / 2, x = c[3] + kx - ks[2] / 2;
          int r = -1;
          if (z >= 0 && z < shape[0] && y >= 0 && y < shape[1] && x >= 0 && x < shape[2])
            r = find_row(t, n, lin(c[0], z, y, x, shape));
          nbr[(size_t)i * K + o] = r;
        }
  }
  free(t);
  return 0;
}

/* output active set of a strided conv, ascending linear index. returns n_out (may exceed max_out: then truncated) */
int oracle_spconv_out(const int* coords, int n, const int* ks, const int* st, const int* pd, const int* oshape,
                      int* out_coords, int max_out) {
  const int K = ks[0] * ks[1] * ks[2];
  int64_t* cand = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1) * K);
  size_t m = 0;
  for (int j = 0; j < n; ++j) {
    const int* c = coords + 4 * (size_t)j;
    for (int kz = 0; kz < ks[0]; ++kz)
      for (int ky = 0; ky < ks[1]; ++ky)
        for (int kx = 0; kx < ks[2]; ++kx) {
          int tz = c[1] + pd[0] - kz, ty = c[2] + pd[1] - ky, tx = c[3] + pd[2] - kx;
          if (tz < 0 || ty < 0 || tx < 0) continue;
          if (tz % st[0] || ty % st[1] || tx % st[2]) continue;
          int oz = tz / st[0], oy = ty / st[1], ox = tx / st[2];
          if (oz >= oshape[0] || oy >= oshape[1] || ox >= oshape[2]) continue;
          cand[m++] = lin(c[0], oz, oy, ox, oshape);
        }
  }
  qsort(cand, m, sizeof(int64_t), cmp_i64);
  int n_out = 0;
  for (size_t i = 0; i < m; ++i) {
    if (i > 0 && cand[i] == cand[i - 1]) continue;
    if (n_out < max_out) {
      int64_t l = cand[i];
      int x = (int)(l % oshape[2]); l /= oshape[2];
      int y = (int)(l % oshape[1]); l /= oshape[1];
      int z = (int)(l % oshape[0]); l /= oshape[0];
      int* o = out_coords + 4 * (size_t)n_out;
      o[0] = (int)l; o[1] = z; o[2] = y; o[3] = x;
    }
    ++n_out;
  }
  free(cand);
  return n_out;
}

/* nbr (n_out,K): for every output site and offset the input row, by direct lookup of the input site */
int oracle_spconv_nbr(const int* coords, int n, const int* ishape, const int* out_coords, int n_out, const int* ks,
                      const int* st, const int* pd, int* nbr) {
  KeyRow* t = (KeyRow*)malloc(sizeof(KeyRow) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) {
    const int* c = coords + 4 * (size_t)i;
    t[i].key = lin(c[0], c[1], c[2], c[3], ishape);
    t[i].row = i;
  }
  qsort(t, n, sizeof(KeyRow), cmp_keyrow);
  const int K = ks[0] * ks[1] * ks[2];
  for (int i = 0; i < n_out; ++i) {
    const int* c = out_coords + 4 * (size_t)i;
    int o = 0;
    for (int kz = 0; kz < ks[0]; ++kz)
      for (int ky = 0; ky < ks[1]; ++ky)
        for (int kx = 0; kx < ks[2]; ++kx, ++o) {
          int z = c[1] * st[0] - pd[0] + kz, y = c[2] * st[1] - pd[1] + ky, x = c[3] * st[2] - pd[2] + kx;
          int r = -1;
          if (z >= 0 && z < ishape[0] && y >= 0 && y < ishape[1] && x >= 0 && x < ishape[2])
            r = find_row(t, n, lin(c[0], z, y, x, ishape));
          nbr[(size_t)i * K + o] = r;
        }
  }
  free(t);
  return 0;
}

/* Y = sum_o X[nbr[:,o]] W[o], W (K,cin,cout); double accumulation */
void oracle_conv_fwd(const float* X, const float* W, const int* nbr, float* Y, int n_out, int K, int cin, int cout) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n_out; ++i) {
    double acc[512];
    for (int c = 0; c < cout; ++c) acc[c] = 0.0;
    for (int o = 0; o < K; ++o) {
      int j = nbr[(size_t)i * K + o];
      if (j < 0) continue;
      const float* x = X + (size_t)j * cin;
      const float* w = W + (size_t)o * cin * cout;
      for (int k = 0; k < cin; ++k) {
        double xv = x[k];
        for (int c = 0; c < cout; ++c) acc[c] += xv * (double)w[(size_t)k * cout + c];
      }
    }
    for (int c = 0; c < cout; ++c) Y[(size_t)i * cout + c] = (float)acc[c];
  }
}

/* dX (n_in,cin) by scattering dY through the same table: dX[nbr[i][o]] += dY[i] W[o]^T.
 * For a fixed offset the map i -> nbr[i][o] is injective, so the inner loop over i is race free. */
void oracle_conv_dgrad(const float* dY, const float* W, const int* nbr, float* dX, int n_in, int n_out, int K, int cin,
                       int cout) {
  double* acc = (double*)calloc((size_t)(n_in > 0 ? n_in : 1) * cin, sizeof(double));
  for (int o = 0; o < K; ++o) {
    const float* w = W + (size_t)o * cin * cout;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_out; ++i) {
      int j = nbr[(size_t)i * K + o];
      if (j < 0) continue;
      const float* g = dY + (size_t)i * cout;
      for (int k = 0; k < cin; ++k) {
        double s = 0.0;
        for (int c = 0; c < cout; ++c) s += (double)g[c] * (double)w[(size_t)k * cout + c];
        acc[(size_t)j * cin + k] += s;
      }
    }
  }
  for (size_t t = 0; t < (size_t)n_in * cin; ++t) dX[t] = (float)acc[t];
  free(acc);
}

/* dW (K,cin,cout); offsets are independent -> parallel over o */
void oracle_conv_wgrad(const float* X, const float* dY, const int* nbr, float* dW, int n_out, int K, int cin, int cout) {
  double* acc = (double*)calloc((size_t)K * cin * cout, sizeof(double));
#pragma omp parallel for schedule(dynamic, 1)
  for (int o = 0; o < K; ++o) {
    double* a = acc + (size_t)o * cin * cout;
    for (int i = 0; i < n_out; ++i) {
      int j = nbr[(size_t)i * K + o];
      if (j < 0) continue;
      const float* x = X + (size_t)j * cin;
      const float* g = dY + (size_t)i * cout;
      for (int k = 0; k < cin; ++k)
        for (int c = 0; c < cout; ++c) a[(size_t)k * cout + c] += (double)x[k] * (double)g[c];
    }
  }
  for (size_t t = 0; t < (size_t)K * cin * cout; ++t) dW[t] = (float)acc[t];
  free(acc);
}

/* dense(): out (B,C,D,H,W) zero + scatter */
void oracle_dense(const float* feat, const int* coords, float* out, int n, int B, int C, int D, int H, int W) {
  memset(out, 0, sizeof(float) * (size_t)B * C * D * H * W);
  for (int i = 0; i < n; ++i) {
    const int* c = coords + 4 * (size_t)i;
    for (int ch = 0; ch < C; ++ch)
      out[((((size_t)c[0] * C + ch) * D + c[1]) * H + c[2]) * W + c[3]] = feat[(size_t)i * C + ch];
  }
}
