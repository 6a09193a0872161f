/*
 * boda_oracle.c -- CPU restatement of Boda's conv_fwd / SGEMM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call this file,
 * and only as the checker / reported CPU baseline.  The product path (boda_amd/) never links it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against the
 * reference's own known-answer digests (test/good_tr/{sgemm-gen5,sgemm-gen600,conv-gen5,conv-debug,
 * conv-full-gen5,ops-prof-conv-3x3-cudnn-boda}/wisdom.wis, committed as data under tests/golden/).
 *
 * Each function cites the reference file:line (relative to the reference checkout) it restates.
 * Written from the reference's *behaviour*; no reference source text is reproduced here.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------------------------------------
 * deterministic test-data generators
 * ------------------------------------------------------------------------------------------- */

/* test/rtc/gen-util.h:1-9 : murmur3 fmix32 of the flat index (+ per-tensor constant), mapped to
 * U(-5,5) as (float)h * (10/U32_MAX) - 5.  The reference compiles this with --use_fast_math /
 * -cl-fast-relaxed-math, so the multiply-subtract is one fused op; fmaf() states that exactly. */
static inline float det_hash_rand(uint32_t rv) {
  uint32_t h = rv;
  h ^= h >> 16; h *= 0x85ebca6bu;
  h ^= h >> 13; h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return fmaf((float)h, 10.0f / 4294967296.0f /* (float)U32_MAX rounds to 2^32 */, -5.0f);
}
float bo_det_hash_rand(uint32_t rv) { return det_hash_rand(rv); }

/* per-tensor hash constants: test/rtc/gen_data_sgemm_a.cucl:18, gen_data_sgemm_b.cucl:15,
 * gen_data_Convolution_in.cucl:16, _filts.cucl:16, _biases.cucl:11 */
enum { BO_C_SGEMM = 12738732u, BO_C_CONV_IN = 234234567u, BO_C_CONV_FILTS = 8753985u, BO_C_CONV_BIASES = 39475612u };

/* test/rtc/gen_data_sgemm_a.cucl:8-20.  a has dims K:M (row-major, M fastest).  mode>=100 is
 * divided by 100 first (600 -> 6).  Note the un-chained 'if(2)' followed by 'if(3) else if...'. */
void bo_gen_data_sgemm_a(float *a, uint32_t K, uint32_t M, uint32_t mode, float vi) {
  uint32_t fin_mode = mode; if (fin_mode >= 100) fin_mode /= 100;
  uint64_t const n = (uint64_t)K * M;
  #pragma omp parallel for schedule(static)
  for (uint64_t i = 0; i < n; ++i) {
    uint32_t const k = (uint32_t)(i / M), m = (uint32_t)(i % M);
    float val = vi;
    if (fin_mode == 2) val += (float)m;
    if (fin_mode == 3) val += (float)k;
    else if (fin_mode == 4) { if (m == M / 2 && k == K / 2) val += 1.0f; }
    else if (fin_mode == 5) val += det_hash_rand((uint32_t)i + BO_C_SGEMM);
    else if (fin_mode == 6) val += (float)(m * 1000u + k);
    a[i] = val;
  }
}

/* test/rtc/gen_data_sgemm_b.cucl:7-20.  b has dims K:N.  mode>=100 -> identity pattern (n==k). */
void bo_gen_data_sgemm_b(float *b, uint32_t K, uint32_t N, uint32_t mode, float vi) {
  uint64_t const n_el = (uint64_t)K * N;
  #pragma omp parallel for schedule(static)
  for (uint64_t i = 0; i < n_el; ++i) {
    uint32_t const k = (uint32_t)(i / N), n = (uint32_t)(i % N);
    float val = vi;
    if (mode == 2) val += (float)n;
    if (mode == 3) val += (float)k;
    else if (mode == 4) { if (n == N / 2 && k == K / 2) val += 1.0f; }
    else if (mode == 5) val += det_hash_rand((uint32_t)i + BO_C_SGEMM);
    else if (mode >= 100) { if (n == k) val += 1.0f; }
    b[i] = val;
  }
}

/* test/rtc/gen_data_Convolution_{in,filts}.cucl:9-18 : 4-D tensor ?:?:y:x; modes 2/3/4 look at
 * the x / y coordinate, mode 5 hashes the flat index. */
static void gen_data_4d(float *t, uint64_t n, uint32_t Y, uint32_t X, uint32_t mode, float vi, uint32_t hc) {
  #pragma omp parallel for schedule(static)
  for (uint64_t i = 0; i < n; ++i) {
    uint32_t const x = (uint32_t)(i % X), y = (uint32_t)((i / X) % Y);
    float val = vi;
    if (mode == 2) val += (float)x;
    if (mode == 3) val += (float)y;
    else if (mode == 4) { if (x == X / 2 && y == Y / 2) val += 1.0f; }
    else if (mode == 5) val += det_hash_rand((uint32_t)i + hc);
    t[i] = val;
  }
}
void bo_gen_data_conv_in(float *in, uint32_t B, uint32_t C, uint32_t Y, uint32_t X, uint32_t mode, float vi) {
  gen_data_4d(in, (uint64_t)B * C * Y * X, Y, X, mode, vi, BO_C_CONV_IN);
}
void bo_gen_data_conv_filts(float *f, uint32_t OC, uint32_t IC, uint32_t Y, uint32_t X, uint32_t mode, float vi) {
  gen_data_4d(f, (uint64_t)OC * IC * Y * X, Y, X, mode, vi, BO_C_CONV_FILTS);
}
/* test/rtc/gen_data_Convolution_biases.cucl:8-13 : only mode 5 adds anything. */
void bo_gen_data_conv_biases(float *b, uint32_t OC, uint32_t mode, float vi) {
  for (uint32_t i = 0; i < OC; ++i) b[i] = vi + ((mode == 5) ? det_hash_rand(i + BO_C_CONV_BIASES) : 0.0f);
}

/* ---------------------------------------------------------------------------------------------
 * SGEMM:  c[M,N] = sum_k a[k,M] * b[k,N]      (a is stored pre-transposed, K:M)
 * test/rtc/sgemm.cucl:17-43 + src/cnn_codegen.cc:460-490 (per-thread fp32 accumulators, k ascending,
 * fused multiply-add under fast-math); same contract as src/culibs-wrap.cc:214-242.
 * fp32 fmaf chain in ascending k per output == what one GPU thread of the reference computes.
 * ------------------------------------------------------------------------------------------- */
void bo_sgemm(float const *a, float const *b, float *c, uint32_t M, uint32_t N, uint32_t K) {
  enum { MB = 8 };
  #pragma omp parallel for schedule(dynamic, 1)
  for (uint32_t m0 = 0; m0 < M; m0 += MB) {
    uint32_t const mb = (M - m0 < MB) ? (M - m0) : MB;
    for (uint32_t mi = 0; mi < mb; ++mi) memset(c + (size_t)(m0 + mi) * N, 0, sizeof(float) * N);
    for (uint32_t k = 0; k < K; ++k) {
      float const *bk = b + (size_t)k * N;
      for (uint32_t mi = 0; mi < mb; ++mi) {
        float const av = a[(size_t)k * M + m0 + mi];
        float *cm = c + (size_t)(m0 + mi) * N;
        for (uint32_t n = 0; n < N; ++n) cm[n] = fmaf(av, bk[n], cm[n]);
      }
    }
  }
}
/* same, fp64 accumulation (used to bound the fp32 ordering error, not a reference behaviour) */
void bo_sgemm_f64acc(float const *a, float const *b, float *c, uint32_t M, uint32_t N, uint32_t K) {
  #pragma omp parallel
  {
    double *acc = (double *)malloc(sizeof(double) * N);
    #pragma omp for schedule(dynamic, 4)
    for (uint32_t m = 0; m < M; ++m) {
      for (uint32_t n = 0; n < N; ++n) acc[n] = 0.0;
      for (uint32_t k = 0; k < K; ++k) {
        double const av = a[(size_t)k * M + m]; float const *bk = b + (size_t)k * N;
        for (uint32_t n = 0; n < N; ++n) acc[n] += av * (double)bk[n];
      }
      for (uint32_t n = 0; n < N; ++n) c[(size_t)m * N + n] = (float)acc[n];
    }
    free(acc);
  }
}

/* ---------------------------------------------------------------------------------------------
 * Convolution forward + bias (+ ReLU):  NCHW in/out, OIHW filts, cross-correlation, zero padding.
 *   in_y = out_y*stride_y + ky - pad_y   (test/rtc/conv.cucl:33-36)
 *   accumulate over (in_chan, ky, kx) ascending from 0 (conv.cucl:24-44),
 *   then out = acc + bias, then max(0,.) iff conv_has_relu (src/cnn_codegen.cc:35-42;
 *   src/cnn_op.cc:337 sets conv_has_relu=1 for every Convolution under ops-prof).
 *   Same semantics via cuDNN in src/culibs-wrap.cc:131-140,186-211.
 *   Output size (in + 2*pad - k)/stride + 1, floor: src/conv_util.cc:167-173 (the caller passes OH/OW).
 * ------------------------------------------------------------------------------------------- */
void bo_conv_fwd(float const *in, float const *filts, float const *biases, float *out,
                 uint32_t B, uint32_t C, uint32_t H, uint32_t W,
                 uint32_t OC, uint32_t KH, uint32_t KW,
                 uint32_t SY, uint32_t SX, uint32_t PY, uint32_t PX,
                 uint32_t OH, uint32_t OW, uint32_t relu) {
  int64_t const n_rows = (int64_t)B * OC;
  #pragma omp parallel
  {
    float *acc = (float *)malloc(sizeof(float) * (size_t)OH * OW);
    #pragma omp for schedule(dynamic, 1)
    for (int64_t r = 0; r < n_rows; ++r) {
      uint32_t const img = (uint32_t)(r / OC), oc = (uint32_t)(r % OC);
      for (size_t i = 0; i < (size_t)OH * OW; ++i) acc[i] = 0.0f;
      for (uint32_t ic = 0; ic < C; ++ic) {
        float const *inp = in + ((size_t)img * C + ic) * H * W;
        for (uint32_t ky = 0; ky < KH; ++ky) {
          for (uint32_t kx = 0; kx < KW; ++kx) {
            float const w = filts[(((size_t)oc * C + ic) * KH + ky) * KW + kx];
            for (uint32_t oy = 0; oy < OH; ++oy) {
              int32_t const iy = (int32_t)(oy * SY + ky) - (int32_t)PY;
              if (iy < 0 || iy >= (int32_t)H) continue; /* zero padding: fma(w,0,acc)==acc */
              /* valid ox range: 0 <= ox*SX + kx - PX < W */
              int32_t ox_lo = 0;
              if ((int32_t)kx < (int32_t)PX) ox_lo = ((int32_t)PX - (int32_t)kx + (int32_t)SX - 1) / (int32_t)SX;
              int32_t ox_hi = ((int32_t)W - 1 + (int32_t)PX - (int32_t)kx);
              ox_hi = (ox_hi < 0) ? -1 : ox_hi / (int32_t)SX;
              if (ox_hi >= (int32_t)OW) ox_hi = (int32_t)OW - 1;
              float const *irow = inp + (size_t)iy * W + kx - PX; /* index with ox*SX */
              float *arow = acc + (size_t)oy * OW;
              if (SX == 1) { for (int32_t ox = ox_lo; ox <= ox_hi; ++ox) arow[ox] = fmaf(w, irow[ox], arow[ox]); }
              else { for (int32_t ox = ox_lo; ox <= ox_hi; ++ox) arow[ox] = fmaf(w, irow[(size_t)ox * SX], arow[ox]); }
            }
          }
        }
      }
      float const bv = biases[oc];
      float *orow = out + (size_t)r * OH * OW;
      for (size_t i = 0; i < (size_t)OH * OW; ++i) {
        float v = acc[i] + bv;
        orow[i] = relu ? ((v > 0.0f) ? v : 0.0f) : v;
      }
    }
    free(acc);
  }
}

/* ---------------------------------------------------------------------------------------------
 * comparison: max relative difference with min significant magnitude 1
 * src/boda_base.cc:140-154 (min_sig_mag_rel_diff) and :160-175 (ssds_diff_t accumulation).
 * returns mrd; fills optional stats {mad, ssds, sds, num_diff, has_nan}
 * ------------------------------------------------------------------------------------------- */
double bo_min_sig_mag_rel_diff(double min_sig_mag, double v1, double v2) {
  double const a1 = fabs(v1), a2 = fabs(v2);
  double amax = (a1 > a2) ? a1 : a2; if (min_sig_mag > amax) amax = min_sig_mag;
  return fabs(v2 - v1) / amax;
}
double bo_ssds_diff(float const *o1, float const *o2, uint64_t n, double *stats5) {
  double mrd = 0, mad = 0, ssds = 0, sds = 0; uint64_t num_diff = 0; int has_nan = 0;
  #pragma omp parallel for reduction(max:mrd,mad) reduction(+:ssds,sds,num_diff) reduction(|:has_nan)
  for (uint64_t i = 0; i < n; ++i) {
    double const d = (double)o2[i] - (double)o1[i];
    if (isnan(d)) has_nan |= 1;
    if (o1[i] != o2[i]) ++num_diff;
    sds += d; ssds += d * d;
    double const ad = fabs(d); if (ad > mad) mad = ad;
    double const rd = bo_min_sig_mag_rel_diff(1.0, o1[i], o2[i]); if (rd > mrd) mrd = rd;
  }
  if (stats5) { stats5[0] = mad; stats5[1] = ssds; stats5[2] = sds; stats5[3] = (double)num_diff; stats5[4] = has_nan; }
  return mrd;
}

/* ---------------------------------------------------------------------------------------------
 * nda_digest_t sample plan + strided fp32 checksums   (src/boda_base.cc:214-272)
 *   strides = {1,2,3,5,7,11,13,17,19,23,29 (<= n)} U {dim strides} U {n}, ascending;
 *   per stride: floor_log2(stride+1) offsets drawn from boost::random::mt19937(seed) through
 *   boost::random::uniform_int_distribution<uint64_t>(0,stride-1); duplicate offsets skipped;
 *   checksum = sequential fp32 sum of v[offset::stride]  (the loop index is a uint32_t).
 * mt19937 is the standard MT (Matsumoto/Nishimura 1998, 32-bit, init_genrand seeding), which is
 * what boost::random::mt19937 implements; the uniform_int algorithm for a 32-bit engine and a
 * range < 2^32 is boost's "bucket" rejection method (boost/random/uniform_int_distribution.hpp,
 * generate_uniform_int, brange > range branch) -- third-party code not vendored in the reference
 * tree; pinned by reproducing all stored sample vectors of the golden digests.
 * ------------------------------------------------------------------------------------------- */
typedef struct { uint32_t mt[624]; int idx; } mt19937_t;
static void mt_seed(mt19937_t *g, uint32_t s) {
  g->mt[0] = s;
  for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}
static uint32_t mt_next(mt19937_t *g) {
  if (g->idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      uint32_t const y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->idx = 0;
  }
  uint32_t y = g->mt[g->idx++];
  y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
  return y;
}
static uint64_t boost_uniform_u64(mt19937_t *g, uint64_t range /* max-min */) {
  if (range == 0) return 0;                       /* no engine draw */
  uint64_t const brange = 0xffffffffull;
  if (range == brange) return mt_next(g);
  if (range > brange) { /* not reachable for tensors < 2^32 elements; compose two draws (boost's scheme) */
    for (;;) {
      uint64_t lo = mt_next(g);
      uint64_t hi = boost_uniform_u64(g, range / (brange + 1));
      if (hi > UINT64_MAX / (brange + 1)) continue;
      uint64_t r = hi * (brange + 1) + lo;
      if (r < lo || r > range) continue;
      return r;
    }
  }
  uint64_t bucket = brange / (range + 1);
  if (brange % (range + 1) == range) ++bucket;
  for (;;) { uint64_t const r = (uint64_t)mt_next(g) / bucket; if (r <= range) return r; }
}
static int cmp_u64(void const *a, void const *b) { uint64_t x = *(uint64_t const *)a, y = *(uint64_t const *)b; return (x > y) - (x < y); }

/* writes up to max_samps {stride, offset} pairs; returns count.  dim_strides: element strides of each dim. */
uint32_t bo_digest_plan(uint64_t n, uint32_t const *dim_strides, uint32_t ndims, uint64_t seed,
                        uint64_t *strides_out, uint64_t *offsets_out, uint32_t max_samps) {
  static uint32_t const primes[11] = {1, 2, 3, 5, 7, 11, 13, 17, 19, 23, 29};
  uint64_t ss[64]; uint32_t nss = 0;
  for (int i = 0; i < 11; ++i) if (primes[i] <= n) ss[nss++] = primes[i];
  for (uint32_t i = 0; i < ndims && nss < 63; ++i) ss[nss++] = dim_strides[i];
  ss[nss++] = n;
  qsort(ss, nss, sizeof(uint64_t), cmp_u64);
  uint32_t u = 0; for (uint32_t i = 0; i < nss; ++i) if (i == 0 || ss[i] != ss[i - 1]) ss[u++] = ss[i];
  nss = u;
  mt19937_t g; mt_seed(&g, (uint32_t)(seed & 0xffffffffull)); /* 32-bit engine seeded with truncated seed */
  uint32_t cnt = 0;
  for (uint32_t si = 0; si < nss; ++si) {
    uint64_t const stride = ss[si];
    uint32_t num_offsets = 0; { uint64_t v = stride + 1; while (v >>= 1) ++num_offsets; } /* floor_log2 */
    uint64_t seen[64]; uint32_t nseen = 0;
    for (uint32_t i = 0; i < num_offsets; ++i) {
      uint64_t const off = boost_uniform_u64(&g, stride - 1);
      int dup = 0; for (uint32_t j = 0; j < nseen; ++j) if (seen[j] == off) dup = 1;
      if (dup) continue;
      seen[nseen++] = off;
      if (cnt < max_samps) { strides_out[cnt] = stride; offsets_out[cnt] = off; }
      ++cnt;
    }
  }
  return cnt;
}
void bo_digest_f32(float const *v, uint64_t n, uint64_t const *strides, uint64_t const *offsets, uint32_t ns,
                   float *min_v, float *max_v, float *samps) {
  float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
  for (uint64_t i = 0; i < n; ++i) { if (v[i] < mn) mn = v[i]; if (v[i] > mx) mx = v[i]; }
  *min_v = mn; *max_v = mx;
  #pragma omp parallel for schedule(dynamic, 1)
  for (uint32_t s = 0; s < ns; ++s) {
    volatile float sv = 0.0f; /* volatile: keep strict sequential fp32 adds */
    float acc = 0.0f;
    for (uint64_t i = offsets[s]; i < n; i += strides[s]) acc += v[i];
    sv = acc; samps[s] = sv;
  }
}

int bo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---------------------------------------------------------------------------------------------
 * Non-conv forward ops of a full-net rtc_fwd (NiN / AlexNet): pooling, ReLU, LRN.
 * ------------------------------------------------------------------------------------------- */

/* Pooling, semantics of test/rtc/pool.cucl:12-40: padding pixels never take part, for max or for average; the reference visits
 * the window column by column (x outer, y inner), which fixes the summation order of an average; a max starts from -FLT_MAX; an
 * average divides by the number of in-plane taps (a window wholly in the padding gives -FLT_MAX / NaN, as there).  Restated here with the window clipped to the plane up front.
 * Output size uses the Caffe ceil convention (src/conv_util.cc:198-204) -- computed by the caller. */
void bo_pool_fwd(float const *in, float *out, uint32_t B, uint32_t C, uint32_t H, uint32_t W, uint32_t KH, uint32_t KW,
                 uint32_t SY, uint32_t SX, uint32_t PY, uint32_t PX, uint32_t OH, uint32_t OW, uint32_t avg_pool) {
  int64_t const planes = (int64_t)B * C;
  #pragma omp parallel for schedule(static)
  for (int64_t p = 0; p < planes; ++p) {
    float const *plane = in + (size_t)p * H * W; float *dst = out + (size_t)p * OH * OW;
    for (int64_t oy = 0; oy < OH; ++oy) {
      int64_t const y0 = oy * SY - (int64_t)PY, ya = y0 < 0 ? 0 : y0, yb = (y0 + KH > H) ? (ya > H ? ya : (int64_t)H) : y0 + KH; /* rows [ya, yb) */
      for (int64_t ox = 0; ox < OW; ++ox) {
        int64_t const x0 = ox * SX - (int64_t)PX, xa = x0 < 0 ? 0 : x0, xb = (x0 + KW > W) ? (xa > W ? xa : (int64_t)W) : x0 + KW; /* cols */
        float acc = avg_pool ? 0.0f : -3.402823466e+38f;
        for (int64_t x = xa; x < xb; ++x) for (int64_t y = ya; y < yb; ++y) {
          float const v = plane[y * W + x];
          if (avg_pool) acc += v; else if (v > acc) acc = v;
        }
        if (avg_pool) acc /= (float)((yb - ya) * (xb - xa));
        dst[oy * OW + ox] = acc;
      }
    }
  }
}
/* ReLU in place, semantics of test/rtc/relu.cucl:1-5: every value that is <= 0 (so -0.0 too) becomes +0.0, NaN is kept */
void bo_relu(float *inout, uint64_t n) { for (float *v = inout; v != inout + n; ++v) if (*v <= 0.0f) *v = 0.0f; }
/* LRN across channels, semantics of the reference's caffe-matching running-sum path (test/rtc/lrn.cucl:35-50): a window of
 * local_size channels slides along the channel axis of one pixel; the sum of squares is carried along -- the entering square is
 * added first, then the leaving one subtracted, an order that matters in float -- and once the window is centred on channel c
 *   out[c] = in[c] * powf(k + sumsq * (alpha / local_size), -beta).   Channels outside the tensor count as zero. */
void bo_lrn_fwd(float const *in, float *out, uint32_t B, uint32_t C, uint32_t H, uint32_t W, uint32_t local_size, float alpha, float beta, float k) {
  int64_t const half = local_size / 2, plane = (int64_t)H * W, nchan = C;
  float const per_elem = alpha / (float)local_size;
  #pragma omp parallel for schedule(static)
  for (int64_t pix = 0; pix < (int64_t)B * plane; ++pix) {
    int64_t const first = (pix / plane) * nchan * plane + pix % plane;     /* channel 0 of this pixel; channels are `plane` apart */
    float sumsq = 0.0f;
    for (int64_t c_in = 0; c_in < nchan + half; ++c_in) {                  /* c_in: channel entering the window */
      int64_t const c_gone = c_in - (int64_t)local_size, c_mid = c_in - half;
      float const entering = c_in < nchan ? in[first + c_in * plane] : 0.0f;
      float const leaving = c_gone >= 0 ? in[first + c_gone * plane] : 0.0f;
      sumsq += entering * entering;
      sumsq -= leaving * leaving;
      if (c_mid >= 0) out[first + c_mid * plane] = in[first + c_mid * plane] * powf(k + sumsq * per_elem, -beta);
    }
  }
}
