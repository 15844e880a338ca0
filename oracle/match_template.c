/*
 * oracle/match_template.c -- CPU restatement (TEST INFRASTRUCTURE, not product code)
 *
 * PARITY UNPINNED: the arithmetic restated here lives in OpenCV
 * (cv2.matchTemplate, version unpinned by the reference: requirements.txt:1-2,
 * README.md:31 "OpenCV 2.4.x or newer").  cv2 is not installed in this image
 * and the reference holds no golden vector for this path, so
 * this file restates the published algorithm and is checked only against its
 * own independent formulations (direct vs FFT vs brute-force definition).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (sushi_amd/) never does.
 *
 * What is restated
 *   reference call site : wav.py:185   cv2.matchTemplate(search, pattern, cv2.TM_SQDIFF_NORMED)
 *                         wav.py:186   result.argmin(axis=1)[0]
 *   algorithm           : OpenCV imgproc templmatch.cpp
 *       crossCorr()             -> corr[p] = sum_m T[m]*I[p+m], stored into the CV_32F result Mat
 *       common_matchTemplate()  -> integral (CV_64F) of I^2, meanStdDev(T), then per position
 *                                  num = wndSum2 - 2*corr + templSum2 ; num = MAX(num,0)
 *                                  t   = sqrt(MAX(wndSum2,0)) * templNorm   (t = 0 when wndSum2 is ~0)
 *                                  |num| < t ? num/t : (|num| < 1.125 t ? 1 : 1)     [SQDIFF_NORMED]
 *                                  result = (float)num
 *   A 1 x L image and a 1 x M template give a 1 x (L-M+1) float32 result.
 *
 * Numerics that matter for parity
 *   - cv2 writes the cross-correlation into the float32 result Mat and reads
 *     it back (`double num = rrow[j]`), so corr is rounded to float32 before
 *     the SQDIFF formula.  `corr_f32 != 0` reproduces that; `corr_f32 == 0`
 *     keeps corr in double (the mathematically exact variant).
 *   - cv2 computes corr by block DFT (double for CV_32F input, float for
 *     CV_8U input).  The DFT's rounding noise is not restated: corr here is
 *     the exactly-rounded value cv2 approximates (double accumulation for
 *     float input, exact int64 for uint8 input).
 *   - all window/template sums in double, as cv2's CV_64F integral does.
 */
#include <stdint.h>
#include <stddef.h>
#include <math.h>
#include <float.h>
#include <stdlib.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* templmatch.cpp common_matchTemplate(): template statistics through
 * meanStdDev, exactly in the order cv2 derives templSum2 / templNorm. */
static void templ_stats(double sum, double sqsum, int64_t M, double *templSum2, double *templNorm)
{
    double invArea = 1.0 / (double)M;
    double mean = sum * invArea;                 /* meanStdDev: mean              */
    double var = sqsum * invArea - mean * mean;  /* meanStdDev: sdv^2, clamped    */
    if (var < 0) var = 0;
    double sdv = sqrt(var);
    double norm = sdv * sdv;                     /* templNorm = templSdv[0]^2     */
    double sum2 = norm + mean * mean;            /* templSum2 = templNorm + mean^2 */
    /* numType != 1 : templNorm = templSum2 */
    norm = sum2;
    sum2 /= invArea;                             /* templSum2 /= invArea          */
    norm = sqrt(norm);
    norm /= sqrt(invArea);                       /* "care of accuracy here"       */
    *templSum2 = sum2;
    *templNorm = norm;
}

/* templmatch.cpp common_matchTemplate(): one output position, SQDIFF_NORMED. */
static inline float finish_sqdiff_normed(double corr, double wndSum2, double templSum2, double templNorm,
                                         int corr_f32)
{
    double num = corr_f32 ? (double)(float)corr : corr;  /* double num = rrow[j] */
    double t;
    num = wndSum2 - 2 * num + templSum2;
    if (num < 0) num = 0;                                /* num = MAX(num, 0.)   */
    {
        double diff2 = wndSum2 > 0 ? wndSum2 : 0;        /* wndMean2 == 0 here   */
        double lim = 10 * (double)FLT_EPSILON * wndSum2;
        if (lim > 0.5) lim = 0.5;
        if (diff2 <= lim)
            t = 0;                                       /* avoid rounding errors */
        else
            t = sqrt(diff2) * templNorm;
    }
    if (fabs(num) < t)
        num /= t;
    else if (fabs(num) < t * 1.125)
        num = num > 0 ? 1 : -1;
    else
        num = 1;                                         /* method == SQDIFF_NORMED */
    return (float)num;
}

/* 1 x L float32 image, 1 x M float32 template -> out[L-M+1] float32.
 * Direct O(P*M) evaluation, double accumulation.  Returns 0, or -1 on bad sizes
 * (cv2 raises cv2.error when the template is larger than the image). */
ORACLE_API int oracle_match_sqdiff_normed_f32(const float *img, int64_t L, const float *tmpl, int64_t M,
                                              float *out, int corr_f32)
{
    if (M <= 0 || L < M) return -1;
    int64_t P = L - M + 1;
    /* integral(img, sum, sqsum, CV_64F): running double prefix of squares */
    double *sq = (double *)malloc((size_t)(L + 1) * sizeof(double));
    if (!sq) return -2;
    sq[0] = 0;
    for (int64_t k = 0; k < L; k++) sq[k + 1] = sq[k] + (double)img[k] * (double)img[k];
    double ts = 0, ts2 = 0;
    for (int64_t m = 0; m < M; m++) { ts += tmpl[m]; ts2 += (double)tmpl[m] * (double)tmpl[m]; }
    double templSum2, templNorm;
    templ_stats(ts, ts2, M, &templSum2, &templNorm);
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < P; p++) {
        const float *w = img + p;
        double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        int64_t m = 0;
        for (; m + 4 <= M; m += 4) {
            c0 += (double)tmpl[m] * (double)w[m];
            c1 += (double)tmpl[m + 1] * (double)w[m + 1];
            c2 += (double)tmpl[m + 2] * (double)w[m + 2];
            c3 += (double)tmpl[m + 3] * (double)w[m + 3];
        }
        for (; m < M; m++) c0 += (double)tmpl[m] * (double)w[m];
        double corr = (c0 + c1) + (c2 + c3);
        out[p] = finish_sqdiff_normed(corr, sq[p + M] - sq[p], templSum2, templNorm, corr_f32);
    }
    free(sq);
    return 0;
}

/* Same for CV_8U inputs: every sum is an exact integer. */
ORACLE_API int oracle_match_sqdiff_normed_u8(const uint8_t *img, int64_t L, const uint8_t *tmpl, int64_t M,
                                             float *out, int corr_f32)
{
    if (M <= 0 || L < M) return -1;
    int64_t P = L - M + 1;
    double *sq = (double *)malloc((size_t)(L + 1) * sizeof(double));
    if (!sq) return -2;
    sq[0] = 0;
    for (int64_t k = 0; k < L; k++) sq[k + 1] = sq[k] + (double)((int)img[k] * (int)img[k]);
    int64_t ts = 0, ts2 = 0;
    for (int64_t m = 0; m < M; m++) { ts += tmpl[m]; ts2 += (int)tmpl[m] * (int)tmpl[m]; }
    double templSum2, templNorm;
    templ_stats((double)ts, (double)ts2, M, &templSum2, &templNorm);
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < P; p++) {
        const uint8_t *w = img + p;
        int64_t c = 0;
        for (int64_t m = 0; m < M; m++) c += (int)tmpl[m] * (int)w[m];
        out[p] = finish_sqdiff_normed((double)c, sq[p + M] - sq[p], templSum2, templNorm, corr_f32);
    }
    free(sq);
    return 0;
}

/* Post-processing only: the caller supplies corr[p] (e.g. from an FFT), this
 * applies common_matchTemplate().  Used by oracle.py's large-size path so that
 * the direct and FFT formulations share nothing but this epilogue. */
ORACLE_API int oracle_finish_sqdiff_normed(const double *corr, const double *sq_prefix, int64_t P, int64_t M,
                                           double templ_sum, double templ_sqsum, float *out, int corr_f32)
{
    if (M <= 0 || P <= 0) return -1;
    double templSum2, templNorm;
    templ_stats(templ_sum, templ_sqsum, M, &templSum2, &templNorm);
    for (int64_t p = 0; p < P; p++)
        out[p] = finish_sqdiff_normed(corr[p], sq_prefix[p + M] - sq_prefix[p], templSum2, templNorm, corr_f32);
    return 0;
}

/* wav.py:186  result.argmin(axis=1)[0] -- first index of the minimum (NumPy
 * argmin: lowest index on ties; no NaN can occur, see finish_sqdiff_normed). */
ORACLE_API int64_t oracle_argmin_f32(const float *v, int64_t n)
{
    int64_t best = 0;
    for (int64_t k = 1; k < n; k++)
        if (v[k] < v[best]) best = k;
    return best;
}

/* The definition itself, no prefix sums, no shared helper: R = sum (T-I)^2 / sqrt(sum T^2 * sum I^2)
 * in long double.  O(P*M); for cross-checking the restatement above on small cases. */
ORACLE_API int oracle_definition_sqdiff_normed_f32(const float *img, int64_t L, const float *tmpl, int64_t M,
                                                   double *out)
{
    if (M <= 0 || L < M) return -1;
    int64_t P = L - M + 1;
    for (int64_t p = 0; p < P; p++) {
        long double d2 = 0, t2 = 0, i2 = 0;
        for (int64_t m = 0; m < M; m++) {
            long double t = tmpl[m], i = img[p + m];
            d2 += (t - i) * (t - i);
            t2 += t * t;
            i2 += i * i;
        }
        long double den = sqrtl(t2 * i2);
        out[p] = den > 0 ? (double)(d2 / den) : 1.0;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * TM_CCOEFF_NORMED (BASELINE.json's prose names it; the reference itself calls TM_SQDIFF_NORMED).
 * templmatch.cpp common_matchTemplate() with numType == 1, isNormed:
 *     meanStdDev(templ, templMean, templSdv);  templNorm = templSdv^2
 *     if (templNorm < DBL_EPSILON) { result = Scalar::all(1); return; }
 *     templNorm = sqrt(templNorm) / sqrt(invArea)
 *     per position:  num = corr;  t = wndSum;  wndMean2 = t*t*invArea;  num -= t*templMean
 *                    wndSum2 = window sum of squares;  diff2 = MAX(wndSum2 - wndMean2, 0)
 *                    t = diff2 <= MIN(0.5, 10*FLT_EPSILON*wndSum2) ? 0 : sqrt(diff2)*templNorm
 *                    |num| < t ? num/t : (|num| < 1.125 t ? +-1 : 0)
 * and the caller takes the arg-MAX (first index).  Same status as the rest of this file: parity unpinned.
 * ------------------------------------------------------------------------------------------------ */
typedef struct { double mean, norm; int flat; } ccoeff_templ;

static ccoeff_templ ccoeff_templ_stats(double sum, double sqsum, int64_t M)
{
    ccoeff_templ t;
    double invArea = 1.0 / (double)M;
    double mean = sum * invArea;
    double var = sqsum * invArea - mean * mean;
    if (var < 0) var = 0;
    double sdv = sqrt(var);
    double norm = sdv * sdv;                      /* templNorm = templSdv[0]^2 */
    t.flat = norm < DBL_EPSILON;                  /* result = Scalar::all(1)    */
    norm = sqrt(norm);
    norm /= sqrt(invArea);
    t.mean = mean;
    t.norm = norm;
    return t;
}

static inline float finish_ccoeff_normed(double corr, double wndSum, double wndSqSum, const ccoeff_templ *ts,
                                         int64_t M, int corr_f32)
{
    if (ts->flat) return 1.0f;
    double invArea = 1.0 / (double)M;
    double num = corr_f32 ? (double)(float)corr : corr;  /* double num = rrow[j] */
    double t = wndSum;
    double wndMean2 = t * t;
    num -= t * ts->mean;
    wndMean2 *= invArea;
    double wndSum2 = wndSqSum;
    double diff2 = wndSum2 - wndMean2;
    if (diff2 < 0) diff2 = 0;
    double lim = 10 * (double)FLT_EPSILON * wndSum2;
    if (lim > 0.5) lim = 0.5;
    if (diff2 <= lim)
        t = 0;
    else
        t = sqrt(diff2) * ts->norm;
    if (fabs(num) < t)
        num /= t;
    else if (fabs(num) < t * 1.125)
        num = num > 0 ? 1 : -1;
    else
        num = 0;                                         /* method != SQDIFF_NORMED */
    return (float)num;
}

ORACLE_API int oracle_match_ccoeff_normed_f32(const float *img, int64_t L, const float *tmpl, int64_t M,
                                              float *out, int corr_f32)
{
    if (M <= 0 || L < M) return -1;
    int64_t P = L - M + 1;
    double *s1 = (double *)malloc((size_t)(L + 1) * sizeof(double));
    double *sq = (double *)malloc((size_t)(L + 1) * sizeof(double));
    if (!s1 || !sq) { free(s1); free(sq); return -2; }
    s1[0] = sq[0] = 0;
    for (int64_t k = 0; k < L; k++) { s1[k + 1] = s1[k] + (double)img[k]; sq[k + 1] = sq[k] + (double)img[k] * (double)img[k]; }
    double ts = 0, ts2 = 0;
    for (int64_t m = 0; m < M; m++) { ts += tmpl[m]; ts2 += (double)tmpl[m] * (double)tmpl[m]; }
    ccoeff_templ tst = ccoeff_templ_stats(ts, ts2, M);
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < P; p++) {
        const float *w = img + p;
        double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        int64_t m = 0;
        for (; m + 4 <= M; m += 4) {
            c0 += (double)tmpl[m] * (double)w[m];
            c1 += (double)tmpl[m + 1] * (double)w[m + 1];
            c2 += (double)tmpl[m + 2] * (double)w[m + 2];
            c3 += (double)tmpl[m + 3] * (double)w[m + 3];
        }
        for (; m < M; m++) c0 += (double)tmpl[m] * (double)w[m];
        double corr = (c0 + c1) + (c2 + c3);
        out[p] = finish_ccoeff_normed(corr, s1[p + M] - s1[p], sq[p + M] - sq[p], &tst, M, corr_f32);
    }
    free(s1); free(sq);
    return 0;
}

ORACLE_API int oracle_match_ccoeff_normed_u8(const uint8_t *img, int64_t L, const uint8_t *tmpl, int64_t M,
                                             float *out, int corr_f32)
{
    if (M <= 0 || L < M) return -1;
    int64_t P = L - M + 1;
    double *s1 = (double *)malloc((size_t)(L + 1) * sizeof(double));
    double *sq = (double *)malloc((size_t)(L + 1) * sizeof(double));
    if (!s1 || !sq) { free(s1); free(sq); return -2; }
    s1[0] = sq[0] = 0;
    for (int64_t k = 0; k < L; k++) { s1[k + 1] = s1[k] + (double)img[k]; sq[k + 1] = sq[k] + (double)((int)img[k] * (int)img[k]); }
    int64_t ts = 0, ts2 = 0;
    for (int64_t m = 0; m < M; m++) { ts += tmpl[m]; ts2 += (int)tmpl[m] * (int)tmpl[m]; }
    ccoeff_templ tst = ccoeff_templ_stats((double)ts, (double)ts2, M);
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < P; p++) {
        const uint8_t *w = img + p;
        int64_t c = 0;
        for (int64_t m = 0; m < M; m++) c += (int)tmpl[m] * (int)w[m];
        out[p] = finish_ccoeff_normed((double)c, s1[p + M] - s1[p], sq[p + M] - sq[p], &tst, M, corr_f32);
    }
    free(s1); free(sq);
    return 0;
}

/* Post-processing only (corr from an FFT): the large-size path of oracle.py. */
ORACLE_API int oracle_finish_ccoeff_normed(const double *corr, const double *s1_prefix, const double *sq_prefix,
                                           int64_t P, int64_t M, double templ_sum, double templ_sqsum, float *out,
                                           int corr_f32)
{
    if (M <= 0 || P <= 0) return -1;
    ccoeff_templ tst = ccoeff_templ_stats(templ_sum, templ_sqsum, M);
    for (int64_t p = 0; p < P; p++)
        out[p] = finish_ccoeff_normed(corr[p], s1_prefix[p + M] - s1_prefix[p], sq_prefix[p + M] - sq_prefix[p], &tst, M,
                                      corr_f32);
    return 0;
}

/* first index of the maximum (NumPy argmax) */
ORACLE_API int64_t oracle_argmax_f32(const float *v, int64_t n)
{
    int64_t best = 0;
    for (int64_t k = 1; k < n; k++)
        if (v[k] > v[best]) best = k;
    return best;
}

/* The definition: R = sum (T-Tm)(I-Im) / sqrt(sum (T-Tm)^2 * sum (I-Im)^2) in long double; 1 for a flat template
 * (cv2's early return), 0 for a flat window. */
ORACLE_API int oracle_definition_ccoeff_normed_f32(const float *img, int64_t L, const float *tmpl, int64_t M,
                                                   double *out)
{
    if (M <= 0 || L < M) return -1;
    int64_t P = L - M + 1;
    long double tm = 0;
    for (int64_t m = 0; m < M; m++) tm += tmpl[m];
    tm /= M;
    long double t2 = 0;
    for (int64_t m = 0; m < M; m++) t2 += (tmpl[m] - tm) * (tmpl[m] - tm);
    for (int64_t p = 0; p < P; p++) {
        long double im = 0;
        for (int64_t m = 0; m < M; m++) im += img[p + m];
        im /= M;
        long double c = 0, i2 = 0;
        for (int64_t m = 0; m < M; m++) {
            long double t = tmpl[m] - tm, i = img[p + m] - im;
            c += t * i;
            i2 += i * i;
        }
        if (t2 / M < DBL_EPSILON) { out[p] = 1.0; continue; }
        long double den = sqrtl(t2 * i2);
        out[p] = den > 0 ? (double)(c / den) : 0.0;
    }
    return 0;
}

ORACLE_API int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* the bench's CPU leg runs one process per core: each of them single-threaded */
ORACLE_API void oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}
