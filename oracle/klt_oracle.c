/*
 * klt_oracle.c -- CPU restatement of the reference's KLT front-end arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or
 * executed by the product (xrslam_amd/); only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it, and only as the checker.
 *
 * What it restates:
 *   a1  OpenCvImage::preprocess        xrslam-extra/src/xrslam/extra/opencv_image.cpp:156-161
 *         cv::CLAHE(6.0, 8x8)::apply + cv::buildOpticalFlowPyramid(win 21x21, maxLevel 3, derivs)
 *   a2  OpenCvImage::track_keypoints   opencv_image.cpp:75-154
 *         cv::calcOpticalFlowPyrLK forward, border/displacement gates, backward, 0.5 px check
 *   a10 OpenCvImage::detect_keypoints  opencv_image.cpp:38-73
 *         cv::GFTTDetector(max, 1e-3, 20, 3, harris=true, k=0.04) + sort + PoissonDiskFilter + border cull
 *   a9  PoissonDiskFilter<2>           xrslam/src/xrslam/utility/poisson_disk_filter.h:8-113
 *
 * The arithmetic itself lives in OpenCV, which is NOT under /root/reference
 * (system package, version unpinned: cmake/external/opencv.cmake:2).  The
 * published OpenCV 3.4/4.x algorithms are restated here (lkpyramid.cpp
 * LKTrackerInvoker / calcSharrDeriv, pyramids.cpp pyrDown 8U, clahe.cpp,
 * corner.cpp cornerHarris, featureselect.cpp goodFeaturesToTrack).
 *
 * Pinned against the reference's own known answers (tests/test_oracle_klt.py,
 * test_reference_known_answers and ..._at_frame_level):
 *   xrslam-test/test/src/test_feature_track.cpp:41,55,64  (164 / !NO_TRANSLATION / 161)
 * -- reproduced exactly, with the test's cv::undistort restated in oracle/undistort.py.
 *
 * Deliberate, documented deviations from a particular OpenCV *build* (whose
 * float summation order is SIMD/build specific anyway):
 *   - LK sums A11,A12,A22,b1,b2 are accumulated EXACTLY in int64 and converted
 *     to float once (OpenCV accumulates float partials in a build-specific
 *     order).  All later per-point float math follows lkpyramid.cpp verbatim.
 *   - Harris 3x3 sums of dx*dx, dx*dy, dy*dy are accumulated exactly in int32
 *     on the unscaled integer Sobel output and scaled once.
 *   This makes the oracle a deterministic function that a GPU implementation
 *   can match bit-for-bit, and is within ~1e-6 relative of any OpenCV build.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_LEVELS 8

typedef struct OrcLevel {
    int w, h;          /* level size (unpadded) */
    int pad;           /* border on every side */
    int istride;       /* bytes per padded image row */
    int dstride;       /* int16 elements per padded deriv row (2 per pixel) */
    uint8_t *img;      /* padded image buffer  (h+2pad) x istride */
    int16_t *deriv;    /* padded deriv buffer  (h+2pad) x dstride, interleaved dx,dy */
} OrcLevel;

typedef struct OrcPyramid {
    int nlevels;
    OrcLevel lv[ORC_MAX_LEVELS];
} OrcPyramid;

static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * (n - 1) - i;
    }
    return i;
}

static inline int cv_round_f(float v) { return (int)lrintf(v); }   /* cvRound: round-half-even */
static inline int cv_floor_f(float v) { return (int)floorf(v); }

/* Row / tile / point loops below are independent per iteration; they run on
 * g_threads OpenMP threads (default 1 = the reference-faithful single core; the
 * CPU baseline of bench.py also reports a multi-threaded figure because OpenCV's
 * parallel_for_ would spread the same loops).  Results do not depend on it. */
static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
/* Accumulation of the LK window sums (A11, A12, A22, b1, b2) -- a STUDY switch (tools/lk_accumulation_study.py), the
 * default 0 is what every test and the HIP kernel use:
 *   0  exact integers (deviation #1 of DESIGN.md section 5: order-independent, bit-reproducible)
 *   1  float, one pixel after the other in raster order -- OpenCV's scalar path (lkpyramid.cpp: iA11 += (float)(ixval*ixval))
 *   2  float in four lanes striding the row (pixels x, x+4, x+8, ... share a lane; the lanes are added at the end, the row
 *      tail goes to a scalar sum) -- the shape of OpenCV's 128-bit SIMD path
 * OpenCV's result depends on which of these its build took; the switch measures how far they can move a track. */
static int g_lk_accum = 0;
void orc_set_lk_accumulation(int mode) { g_lk_accum = mode; }
int orc_get_threads(void) { return g_threads; }

/* ------------------------------------------------------------------ CLAHE */
/* cv::CLAHE_Impl::apply, 8-bit, tiles must divide the image (EuRoC sizes do;
 * otherwise OpenCV pads with REFLECT_101, restated below as well). */
int orc_clahe(const uint8_t *src, int w, int h, int sstride, double clip_limit,
              int tiles_x, int tiles_y, uint8_t *dst, int dstride) {
    int tw, th;
    const uint8_t *lsrc = src;
    int lstride = sstride;
    uint8_t *ext = NULL;
    if (w % tiles_x == 0 && h % tiles_y == 0) {
        tw = w / tiles_x;
        th = h / tiles_y;
    } else {
        int ew = w + (tiles_x - (w % tiles_x)) % tiles_x;
        int eh = h + (tiles_y - (h % tiles_y)) % tiles_y;
        ext = (uint8_t *)malloc((size_t)ew * eh);
        for (int y = 0; y < eh; ++y)
            for (int x = 0; x < ew; ++x)
                ext[(size_t)y * ew + x] = src[(size_t)reflect101(y, h) * sstride + reflect101(x, w)];
        lsrc = ext;
        lstride = ew;
        tw = ew / tiles_x;
        th = eh / tiles_y;
    }
    const int hist_size = 256;
    const int tile_area = tw * th;
    const float lut_scale = (float)(hist_size - 1) / tile_area;
    int clip = 0;
    if (clip_limit > 0.0) {
        clip = (int)(clip_limit * tile_area / hist_size);
        if (clip < 1) clip = 1;
    }
    uint8_t *lut = (uint8_t *)malloc((size_t)tiles_x * tiles_y * hist_size);
#pragma omp parallel for collapse(2) num_threads(g_threads)
    for (int ty = 0; ty < tiles_y; ++ty) {
        for (int tx = 0; tx < tiles_x; ++tx) {
            int hist[256];
            memset(hist, 0, sizeof(hist));
            for (int y = 0; y < th; ++y) {
                const uint8_t *row = lsrc + (size_t)(ty * th + y) * lstride + tx * tw;
                for (int x = 0; x < tw; ++x) hist[row[x]]++;
            }
            if (clip > 0) {
                int clipped = 0;
                for (int i = 0; i < hist_size; ++i) {
                    if (hist[i] > clip) {
                        clipped += hist[i] - clip;
                        hist[i] = clip;
                    }
                }
                int redist = clipped / hist_size;
                int residual = clipped - redist * hist_size;
                for (int i = 0; i < hist_size; ++i) hist[i] += redist;
                if (residual != 0) {
                    int step = hist_size / residual;
                    if (step < 1) step = 1;
                    for (int i = 0; i < hist_size && residual > 0; i += step, residual--) hist[i]++;
                }
            }
            uint8_t *tl = lut + (size_t)(ty * tiles_x + tx) * hist_size;
            int sum = 0;
            for (int i = 0; i < hist_size; ++i) {
                sum += hist[i];
                int v = cv_round_f((float)sum * lut_scale);
                tl[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
    }
    /* interpolation body */
    const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
#pragma omp parallel for num_threads(g_threads)
    for (int y = 0; y < h; ++y) {
        float tyf = y * inv_th - 0.5f;
        int ty1 = cv_floor_f(tyf);
        int ty2 = ty1 + 1;
        float ya = tyf - ty1, ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > tiles_y - 1) ty2 = tiles_y - 1;
        const uint8_t *p1 = lut + (size_t)ty1 * tiles_x * hist_size;
        const uint8_t *p2 = lut + (size_t)ty2 * tiles_x * hist_size;
        for (int x = 0; x < w; ++x) {
            float txf = x * inv_tw - 0.5f;
            int tx1 = cv_floor_f(txf);
            int tx2 = tx1 + 1;
            float xa = txf - tx1, xa1 = 1.0f - xa;
            if (tx1 < 0) tx1 = 0;
            if (tx2 > tiles_x - 1) tx2 = tiles_x - 1;
            int v = src[(size_t)y * sstride + x];
            int i1 = tx1 * hist_size + v, i2 = tx2 * hist_size + v;
            float res = (p1[i1] * xa1 + p1[i2] * xa) * ya1 + (p2[i1] * xa1 + p2[i2] * xa) * ya;
            int r = cv_round_f(res);
            dst[(size_t)y * dstride + x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
    }
    free(lut);
    free(ext);
    return 0;
}

/* ---------------------------------------------------------------- pyramid */
static void level_alloc(OrcLevel *L, int w, int h, int pad) {
    L->w = w;
    L->h = h;
    L->pad = pad;
    L->istride = w + 2 * pad;
    L->dstride = 2 * (w + 2 * pad);
    L->img = (uint8_t *)calloc((size_t)(h + 2 * pad) * L->istride, 1);
    L->deriv = (int16_t *)calloc((size_t)(h + 2 * pad) * L->dstride, sizeof(int16_t));
}

static inline uint8_t *lv_img(const OrcLevel *L, int x, int y) {
    return L->img + (size_t)(y + L->pad) * L->istride + (x + L->pad);
}
static inline int16_t *lv_der(const OrcLevel *L, int x, int y) {
    return L->deriv + (size_t)(y + L->pad) * L->dstride + 2 * (x + L->pad);
}

static void level_fill_border(OrcLevel *L) {
    /* copyMakeBorder(..., BORDER_REFLECT_101) for the image, zeros for derivs */
#pragma omp parallel for num_threads(g_threads)
    for (int y = -L->pad; y < L->h + L->pad; ++y) {
        int sy = reflect101(y, L->h);
        for (int x = -L->pad; x < L->w + L->pad; ++x) {
            if (x >= 0 && x < L->w && y >= 0 && y < L->h) continue;
            *lv_img(L, x, y) = *lv_img(L, reflect101(x, L->w), sy);
        }
    }
}

/* cv::pyrDown, CV_8U, 5-tap [1 4 6 4 1], (sum+128)>>8, BORDER_REFLECT_101 */
static void pyr_down(const OrcLevel *S, OrcLevel *D) {
    int sw = S->w, sh = S->h;
#pragma omp parallel num_threads(g_threads)
    {
    int *rows = (int *)malloc(sizeof(int) * 5 * D->w);
#pragma omp for
    for (int y = 0; y < D->h; ++y) {
        for (int k = 0; k < 5; ++k) {
            int sy = reflect101(2 * y - 2 + k, sh);
            int *r = rows + k * D->w;
            for (int x = 0; x < D->w; ++x) {
                int x0 = reflect101(2 * x - 2, sw), x1 = reflect101(2 * x - 1, sw), x2 = reflect101(2 * x, sw);
                int x3 = reflect101(2 * x + 1, sw), x4 = reflect101(2 * x + 2, sw);
                const uint8_t *s = lv_img(S, 0, sy);
                r[x] = s[x0] + s[x4] + 4 * (s[x1] + s[x3]) + 6 * s[x2];
            }
        }
        for (int x = 0; x < D->w; ++x) {
            int v = rows[x] + rows[4 * D->w + x] + 4 * (rows[D->w + x] + rows[3 * D->w + x]) + 6 * rows[2 * D->w + x];
            *lv_img(D, x, y) = (uint8_t)((v + 128) >> 8);
        }
    }
    free(rows);
    }
}

/* cv::detail calcSharrDeriv (lkpyramid.cpp): 3/10/3 Scharr, unnormalised, int16,
 * image edges by reflect-101 on the level itself (not on the padded buffer). */
static void scharr_deriv(OrcLevel *L) {
    int w = L->w, h = L->h;
#pragma omp parallel for num_threads(g_threads)
    for (int y = 0; y < h; ++y) {
        const uint8_t *r0 = lv_img(L, 0, y > 0 ? y - 1 : (h > 1 ? 1 : 0));
        const uint8_t *r1 = lv_img(L, 0, y);
        const uint8_t *r2 = lv_img(L, 0, y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0));
        for (int x = 0; x < w; ++x) {
            int xm = x > 0 ? x - 1 : (w > 1 ? 1 : 0);
            int xp = x < w - 1 ? x + 1 : (w > 1 ? w - 2 : 0);
            int t0m = (r0[xm] + r2[xm]) * 3 + r1[xm] * 10;
            int t0p = (r0[xp] + r2[xp]) * 3 + r1[xp] * 10;
            int t1m = r2[xm] - r0[xm];
            int t1c = r2[x] - r0[x];
            int t1p = r2[xp] - r0[xp];
            int16_t *d = lv_der(L, x, y);
            d[0] = (int16_t)(t0p - t0m);
            d[1] = (int16_t)((t1p + t1m) * 3 + t1c * 10);
        }
    }
}

OrcPyramid *orc_pyr_create(int w, int h, int max_level, int pad) {
    OrcPyramid *P = (OrcPyramid *)calloc(1, sizeof(OrcPyramid));
    P->nlevels = max_level + 1;
    for (int l = 0; l <= max_level; ++l) {
        level_alloc(&P->lv[l], w, h, pad);
        w = (w + 1) / 2;
        h = (h + 1) / 2;
    }
    return P;
}

void orc_pyr_destroy(OrcPyramid *P) {
    if (!P) return;
    for (int l = 0; l < P->nlevels; ++l) {
        free(P->lv[l].img);
        free(P->lv[l].deriv);
    }
    free(P);
}

/* cv::buildOpticalFlowPyramid(img, pyr, Size(21,21), maxLevel, withDerivatives=true) */
void orc_pyr_build(OrcPyramid *P, const uint8_t *img, int stride) {
    OrcLevel *L0 = &P->lv[0];
    for (int y = 0; y < L0->h; ++y) memcpy(lv_img(L0, 0, y), img + (size_t)y * stride, L0->w);
    for (int l = 0; l < P->nlevels; ++l) {
        OrcLevel *L = &P->lv[l];
        if (l > 0) pyr_down(&P->lv[l - 1], L);
        level_fill_border(L);
        memset(L->deriv, 0, (size_t)(L->h + 2 * L->pad) * L->dstride * sizeof(int16_t));
        scharr_deriv(L);
    }
}

/* OpenCvImage::preprocess: CLAHE in place, then the pyramid.  `work` receives the CLAHE image. */
void orc_preprocess(OrcPyramid *P, const uint8_t *img, int stride, double clip, int tx, int ty, uint8_t *work) {
    orc_clahe(img, P->lv[0].w, P->lv[0].h, stride, clip, tx, ty, work, P->lv[0].w);
    orc_pyr_build(P, work, P->lv[0].w);
}

int orc_pyr_level_dims(const OrcPyramid *P, int l, int *w, int *h) {
    *w = P->lv[l].w;
    *h = P->lv[l].h;
    return P->nlevels;
}
/* copy out the unpadded level (for comparisons) */
void orc_pyr_get_level(const OrcPyramid *P, int l, uint8_t *img, int16_t *deriv) {
    const OrcLevel *L = &P->lv[l];
    for (int y = 0; y < L->h; ++y) {
        if (img) memcpy(img + (size_t)y * L->w, lv_img(L, 0, y), L->w);
        if (deriv) memcpy(deriv + (size_t)y * L->w * 2, lv_der(L, 0, y), sizeof(int16_t) * 2 * L->w);
    }
}
/* copy out the padded level incl. borders */
void orc_pyr_get_level_padded(const OrcPyramid *P, int l, uint8_t *img, int16_t *deriv) {
    const OrcLevel *L = &P->lv[l];
    size_t rows = (size_t)(L->h + 2 * L->pad);
    if (img) memcpy(img, L->img, rows * L->istride);
    if (deriv) memcpy(deriv, L->deriv, rows * L->dstride * sizeof(int16_t));
}

/* --------------------------------------------------------------------- LK */
#define W_BITS 14
static inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

typedef struct OrcLkStats {
    long long iters;       /* LK iterations executed (sum over points/levels) */
    long long templates;   /* templates extracted */
} OrcLkStats;

/* cv::calcOpticalFlowPyrLK with precomputed pyramids, winSize 21x21 (win argument),
 * criteria COUNT+EPS, OPTFLOW_USE_INITIAL_FLOW, minEigThreshold 1e-4. */
void orc_lk(const OrcPyramid *A, const OrcPyramid *B, const float *prev_pts, float *next_pts,
            uint8_t *status, int n, int win, int max_level, int max_count, double epsilon,
            OrcLkStats *stats) {
    if (max_count < 0) max_count = 0;
    if (max_count > 100) max_count = 100;
    if (epsilon < 0.) epsilon = 0.;
    if (epsilon > 10.) epsilon = 10.;
    epsilon *= epsilon;
    const float min_eig_threshold = 1e-4f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float half = (win - 1) * 0.5f;
    int levels1 = A->nlevels - 1;
    if (max_level > levels1) max_level = levels1;
    for (int i = 0; i < n; ++i) status[i] = 1;
    long long n_templates = 0, n_iters = 0;

    for (int level = max_level; level >= 0; --level) {
        const OrcLevel *I = &A->lv[level];
        const OrcLevel *J = &B->lv[level];
#pragma omp parallel num_threads(g_threads) reduction(+ : n_templates, n_iters)
        {
        int16_t *Ibuf = (int16_t *)malloc(sizeof(int16_t) * win * win * 3);
        int16_t *dIbuf = Ibuf + win * win;
#pragma omp for schedule(dynamic, 4)
        for (int pt = 0; pt < n; ++pt) {
            float px = prev_pts[2 * pt] * (float)(1. / (1 << level));
            float py = prev_pts[2 * pt + 1] * (float)(1. / (1 << level));
            float nx, ny;
            if (level == max_level) {
                nx = next_pts[2 * pt] * (float)(1. / (1 << level));
                ny = next_pts[2 * pt + 1] * (float)(1. / (1 << level));
            } else {
                nx = next_pts[2 * pt] * 2.f;
                ny = next_pts[2 * pt + 1] * 2.f;
            }
            next_pts[2 * pt] = nx;
            next_pts[2 * pt + 1] = ny;

            px -= half;
            py -= half;
            int ipx = cv_floor_f(px), ipy = cv_floor_f(py);
            if (ipx < -win || ipx >= I->w || ipy < -win || ipy >= I->h) {
                if (level == 0) status[pt] = 0;
                continue;
            }
            float a = px - ipx, b = py - ipy;
            int iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
            int iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
            int iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int64_t sA11 = 0, sA12 = 0, sA22 = 0;
            float fA[3] = {0.f, 0.f, 0.f}, qA[3][4] = {{0.f}};   /* study modes 1 / 2 */
            const int accum = g_lk_accum;
            for (int y = 0; y < win; ++y) {
                const uint8_t *s0 = lv_img(I, ipx, ipy + y);
                const uint8_t *s1 = lv_img(I, ipx, ipy + y + 1);
                const int16_t *d0 = lv_der(I, ipx, ipy + y);
                const int16_t *d1 = lv_der(I, ipx, ipy + y + 1);
                for (int x = 0; x < win; ++x) {
                    int ival = descale(s0[x] * iw00 + s0[x + 1] * iw01 + s1[x] * iw10 + s1[x + 1] * iw11, W_BITS - 5);
                    int ixval = descale(d0[2 * x] * iw00 + d0[2 * x + 2] * iw01 + d1[2 * x] * iw10 + d1[2 * x + 2] * iw11, W_BITS);
                    int iyval = descale(d0[2 * x + 1] * iw00 + d0[2 * x + 3] * iw01 + d1[2 * x + 1] * iw10 + d1[2 * x + 3] * iw11, W_BITS);
                    Ibuf[y * win + x] = (int16_t)ival;
                    dIbuf[2 * (y * win + x)] = (int16_t)ixval;
                    dIbuf[2 * (y * win + x) + 1] = (int16_t)iyval;
                    sA11 += (int64_t)ixval * ixval;
                    sA12 += (int64_t)ixval * iyval;
                    sA22 += (int64_t)iyval * iyval;
                    if (accum == 1 || (accum == 2 && x >= (win & ~3))) {
                        fA[0] += (float)(ixval * ixval);
                        fA[1] += (float)(ixval * iyval);
                        fA[2] += (float)(iyval * iyval);
                    } else if (accum == 2) {
                        qA[0][x & 3] += (float)ixval * (float)ixval;
                        qA[1][x & 3] += (float)ixval * (float)iyval;
                        qA[2][x & 3] += (float)iyval * (float)iyval;
                    }
                }
            }
            n_templates++;
            float A11 = (float)sA11 * FLT_SCALE;
            float A12 = (float)sA12 * FLT_SCALE;
            float A22 = (float)sA22 * FLT_SCALE;
            if (accum) {
                for (int q = 0; q < 3; ++q) fA[q] += qA[q][0] + qA[q][1] + qA[q][2] + qA[q][3];
                A11 = fA[0] * FLT_SCALE;
                A12 = fA[1] * FLT_SCALE;
                A22 = fA[2] * FLT_SCALE;
            }
            float D = A11 * A22 - A12 * A12;
            float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
            if (minEig < min_eig_threshold || D < 1.1920929e-07f) {
                if (level == 0) status[pt] = 0;
                continue;
            }
            D = 1.f / D;
            nx -= half;
            ny -= half;
            float pdx = 0.f, pdy = 0.f;
            for (int j = 0; j < max_count; ++j) {
                int inx = cv_floor_f(nx), iny = cv_floor_f(ny);
                if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) {
                    if (level == 0) status[pt] = 0;
                    break;
                }
                a = nx - inx;
                b = ny - iny;
                iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
                iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
                iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                int64_t sb1 = 0, sb2 = 0;
                float fb[2] = {0.f, 0.f}, qb[2][4] = {{0.f}};
                for (int y = 0; y < win; ++y) {
                    const uint8_t *j0 = lv_img(J, inx, iny + y);
                    const uint8_t *j1 = lv_img(J, inx, iny + y + 1);
                    for (int x = 0; x < win; ++x) {
                        int diff = descale(j0[x] * iw00 + j0[x + 1] * iw01 + j1[x] * iw10 + j1[x + 1] * iw11, W_BITS - 5) - Ibuf[y * win + x];
                        sb1 += (int64_t)diff * dIbuf[2 * (y * win + x)];
                        sb2 += (int64_t)diff * dIbuf[2 * (y * win + x) + 1];
                        if (accum == 1 || (accum == 2 && x >= (win & ~3))) {
                            fb[0] += (float)(diff * dIbuf[2 * (y * win + x)]);
                            fb[1] += (float)(diff * dIbuf[2 * (y * win + x) + 1]);
                        } else if (accum == 2) {
                            qb[0][x & 3] += (float)(diff * dIbuf[2 * (y * win + x)]);
                            qb[1][x & 3] += (float)(diff * dIbuf[2 * (y * win + x) + 1]);
                        }
                    }
                }
                n_iters++;
                float b1 = (float)sb1 * FLT_SCALE;
                float b2 = (float)sb2 * FLT_SCALE;
                if (accum) {
                    for (int q = 0; q < 2; ++q) fb[q] += qb[q][0] + qb[q][1] + qb[q][2] + qb[q][3];
                    b1 = fb[0] * FLT_SCALE;
                    b2 = fb[1] * FLT_SCALE;
                }
                float dx = (A12 * b2 - A22 * b1) * D;
                float dy = (A12 * b1 - A11 * b2) * D;
                nx += dx;
                ny += dy;
                next_pts[2 * pt] = nx + half;
                next_pts[2 * pt + 1] = ny + half;
                if ((double)dx * dx + (double)dy * dy <= epsilon) break;
                if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                    next_pts[2 * pt] -= dx * 0.5f;
                    next_pts[2 * pt + 1] -= dy * 0.5f;
                    break;
                }
                pdx = dx;
                pdy = dy;
            }
        }
        free(Ibuf);
        }
    }
    if (stats) {
        stats->templates += n_templates;
        stats->iters += n_iters;
    }
}

/* OpenCvImage::track_keypoints (opencv_image.cpp:75-154).
 * curr/next are double pixel positions; has_guess!=0 means next_inout holds the
 * prediction.  next_inout is only updated for points with status!=0. */
void orc_track_keypoints(const OrcPyramid *A, const OrcPyramid *B, const double *curr, double *next_inout,
                         int has_guess, uint8_t *status, int n, OrcLkStats *stats) {
    if (n <= 0) return;
    int cols = A->lv[0].w, rows = A->lv[0].h;
    float *cp = (float *)malloc(sizeof(float) * 2 * n * 3);
    float *np_ = cp + 2 * n, *rp = cp + 4 * n;
    uint8_t *st2 = (uint8_t *)malloc(n);
    for (int i = 0; i < 2 * n; ++i) {
        cp[i] = (float)curr[i];
        np_[i] = has_guess ? (float)next_inout[i] : cp[i];
    }
    orc_lk(A, B, cp, np_, status, n, 21, 3, 30, 0.01, stats);
    for (int i = 0; i < n; ++i) {
        float x = np_[2 * i], y = np_[2 * i + 1];
        if (x < 20 || x >= cols - 20 || y < 20 || y >= rows - 20) status[i] = 0;
        if (status[i]) {
            float dx = x - cp[2 * i], dy = y - cp[2 * i + 1];
            double nrm = sqrt((double)dx * (double)dx + (double)dy * (double)dy); /* Eigen vector<2>(p.x,p.y).norm() */
            if (nrm > rows / 4) status[i] = 0;
        }
    }
    for (int i = 0; i < 2 * n; ++i) rp[i] = cp[i];
    orc_lk(B, A, np_, rp, st2, n, 21, 3, 30, 0.01, stats);
    for (int i = 0; i < n; ++i) {
        if (status[i]) {
            float dx = cp[2 * i] - rp[2 * i], dy = cp[2 * i + 1] - rp[2 * i + 1];
            double nrm = sqrt((double)dx * (double)dx + (double)dy * (double)dy); /* cv::norm(Point2f) */
            if (!st2[i] || nrm > 0.5) status[i] = 0;
        }
    }
    for (int i = 0; i < n; ++i) {
        if (status[i]) {
            next_inout[2 * i] = np_[2 * i];
            next_inout[2 * i + 1] = np_[2 * i + 1];
        }
    }
    free(cp);
    free(st2);
}

/* ----------------------------------------------------------------- Harris */
/* cv::cornerHarris(src 8U, blockSize 3, ksize 3, k) with exact integer window sums. */
void orc_harris_response(const uint8_t *img, int w, int h, int stride, double k, float *resp) {
    int *dxy = (int *)malloc(sizeof(int) * 2 * (size_t)w * h);
#pragma omp parallel for num_threads(g_threads)
    for (int y = 0; y < h; ++y) {
        const uint8_t *r0 = img + (size_t)reflect101(y - 1, h) * stride;
        const uint8_t *r1 = img + (size_t)y * stride;
        const uint8_t *r2 = img + (size_t)reflect101(y + 1, h) * stride;
        for (int x = 0; x < w; ++x) {
            int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
            int dx = (r0[xp] - r0[xm]) + 2 * (r1[xp] - r1[xm]) + (r2[xp] - r2[xm]);
            int dy = (r2[xm] - r0[xm]) + 2 * (r2[x] - r0[x]) + (r2[xp] - r0[xp]);
            dxy[2 * ((size_t)y * w + x)] = dx;
            dxy[2 * ((size_t)y * w + x) + 1] = dy;
        }
    }
    double scale = (double)(1 << 2) * 3;   /* (1 << (aperture-1)) * block_size */
    scale *= 255.0;
    scale = 1.0 / scale;
    const float s2 = (float)(scale * scale);
#pragma omp parallel for num_threads(g_threads)
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            int sxx = 0, sxy = 0, syy = 0;
            for (int j = -1; j <= 1; ++j) {
                int yy = reflect101(y + j, h);
                for (int i = -1; i <= 1; ++i) {
                    int xx = reflect101(x + i, w);
                    int dx = dxy[2 * ((size_t)yy * w + xx)], dy = dxy[2 * ((size_t)yy * w + xx) + 1];
                    sxx += dx * dx;
                    sxy += dx * dy;
                    syy += dy * dy;
                }
            }
            float a = (float)sxx * s2, b = (float)sxy * s2, c = (float)syy * s2;
            float t1 = a * c;
            float t2 = b * b;
            float t3 = t1 - t2;
            float apc = a + c;
            resp[(size_t)y * w + x] = (float)((double)t3 - k * (double)apc * (double)apc);
        }
    }
    free(dxy);
}

typedef struct { float v; int idx; } Cand;
static int cand_cmp(const void *pa, const void *pb) {
    const Cand *a = (const Cand *)pa, *b = (const Cand *)pb;
    if (a->v > b->v) return -1;
    if (a->v < b->v) return 1;
    return (a->idx > b->idx) ? -1 : (a->idx < b->idx ? 1 : 0);   /* greaterThanPtr: higher address first */
}

/* cv::goodFeaturesToTrack(..., useHarris=true); returns corner count, xy ints as float, quality in `q` */
int orc_gftt(const uint8_t *img, int w, int h, int stride, int max_corners, double quality,
             double min_distance, double k, float *out_xy, float *out_q) {
    float *eig = (float *)malloc(sizeof(float) * (size_t)w * h);
    orc_harris_response(img, w, h, stride, k, eig);
    float maxv = eig[0];
#pragma omp parallel for num_threads(g_threads) reduction(max : maxv)
    for (long i = 1; i < (long)w * h; ++i)
        if (eig[i] > maxv) maxv = eig[i];
    float thr = (float)((double)maxv * quality);
#pragma omp parallel for num_threads(g_threads)
    for (long i = 0; i < (long)w * h; ++i)
        if (!(eig[i] > thr)) eig[i] = 0.f;
    /* local maxima of the 3x3 dilation, flagged per row, then collected in scan order */
    uint8_t *flag = (uint8_t *)calloc((size_t)w * h, 1);
#pragma omp parallel for num_threads(g_threads)
    for (int y = 1; y < h - 1; ++y) {
        for (int x = 1; x < w - 1; ++x) {
            float v = eig[(size_t)y * w + x];
            if (v == 0) continue;
            float m = v;
            for (int j = -1; j <= 1; ++j)
                for (int i = -1; i <= 1; ++i) {
                    float u = eig[(size_t)(y + j) * w + x + i];
                    if (u > m) m = u;
                }
            if (v == m) flag[(size_t)y * w + x] = 1;
        }
    }
    int nc = 0;
    for (size_t i = 0; i < (size_t)w * h; ++i) nc += flag[i];
    Cand *c = (Cand *)malloc(sizeof(Cand) * (size_t)(nc > 0 ? nc : 1));
    nc = 0;
    for (size_t i = 0; i < (size_t)w * h; ++i)
        if (flag[i]) {
            c[nc].v = eig[i];
            c[nc].idx = (int)i;
            nc++;
        }
    free(flag);
    qsort(c, nc, sizeof(Cand), cand_cmp);
    int ncorners = 0;
    if (min_distance >= 1) {
        const int cell = (int)lrint(min_distance);   /* cvRound(minDistance) */
        const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
        /* per-cell singly linked lists */
        int *head = (int *)malloc(sizeof(int) * gw * gh);
        int *nxt = (int *)malloc(sizeof(int) * (nc > 0 ? nc : 1));
        int *px = (int *)malloc(sizeof(int) * (nc > 0 ? nc : 1) * 2);
        for (int i = 0; i < gw * gh; ++i) head[i] = -1;
        double md2d = min_distance * min_distance;
        for (int i = 0; i < nc; ++i) {
            int y = c[i].idx / w, x = c[i].idx % w;
            int xc = x / cell, yc = y / cell;
            int x1 = xc - 1 < 0 ? 0 : xc - 1, y1 = yc - 1 < 0 ? 0 : yc - 1;
            int x2 = xc + 1 > gw - 1 ? gw - 1 : xc + 1, y2 = yc + 1 > gh - 1 ? gh - 1 : yc + 1;
            int good = 1;
            for (int yy = y1; yy <= y2 && good; ++yy)
                for (int xx = x1; xx <= x2 && good; ++xx)
                    for (int e = head[yy * gw + xx]; e >= 0; e = nxt[e]) {
                        float dx = (float)(x - px[2 * e]), dy = (float)(y - px[2 * e + 1]);
                        if ((double)(dx * dx + dy * dy) < md2d) {
                            good = 0;
                            break;
                        }
                    }
            if (good) {
                px[2 * ncorners] = x;
                px[2 * ncorners + 1] = y;
                nxt[ncorners] = head[yc * gw + xc];
                head[yc * gw + xc] = ncorners;
                out_xy[2 * ncorners] = (float)x;
                out_xy[2 * ncorners + 1] = (float)y;
                if (out_q) out_q[ncorners] = c[i].v;
                ++ncorners;
                if (max_corners > 0 && ncorners == max_corners) break;
            }
        }
        free(head);
        free(nxt);
        free(px);
    } else {
        for (int i = 0; i < nc; ++i) {
            out_xy[2 * ncorners] = (float)(c[i].idx % w);
            out_xy[2 * ncorners + 1] = (float)(c[i].idx / w);
            if (out_q) out_q[ncorners] = c[i].v;
            ++ncorners;
            if (max_corners > 0 && ncorners == max_corners) break;
        }
    }
    free(c);
    free(eig);
    return ncorners;
}

/* --------------------------------------------------------- Poisson filter */
/* PoissonDiskFilter<2> (utility/poisson_disk_filter.h:8-113), including its
 * iteration quirk (the first cell of the scan box is skipped, one cell past the
 * end is visited).  Cells map to the LAST point stored in them. */
typedef struct {
    double radius, r2, grid;
    int span;
    int gx0, gy0, gw, gh;   /* dense window over cell indices */
    int *cell;              /* -1 or point index */
    double *pts;
    int npts, cap;
} Pdf;

static void pdf_init(Pdf *f, double radius, double xmin, double ymin, double xmax, double ymax, int cap) {
    f->radius = radius;
    f->r2 = radius * radius;
    f->grid = radius / sqrt(2.0);
    f->span = (int)ceil(sqrt(2.0));
    f->gx0 = (int)floor(xmin / f->grid) - f->span - 2;
    f->gy0 = (int)floor(ymin / f->grid) - f->span - 2;
    f->gw = (int)floor(xmax / f->grid) + f->span + 3 - f->gx0;
    f->gh = (int)floor(ymax / f->grid) + f->span + 3 - f->gy0;
    f->cell = (int *)malloc(sizeof(int) * f->gw * f->gh);
    for (int i = 0; i < f->gw * f->gh; ++i) f->cell[i] = -1;
    f->pts = (double *)malloc(sizeof(double) * 2 * (cap > 0 ? cap : 1));
    f->npts = 0;
    f->cap = cap;
}
static void pdf_free(Pdf *f) {
    free(f->cell);
    free(f->pts);
}
static int *pdf_cell(Pdf *f, int ix, int iy) {
    ix -= f->gx0;
    iy -= f->gy0;
    if (ix < 0 || iy < 0 || ix >= f->gw || iy >= f->gh) return NULL;
    return &f->cell[iy * f->gw + ix];
}
static void pdf_preset(Pdf *f, double x, double y) {
    int ix = (int)floor(x / f->grid), iy = (int)floor(y / f->grid);
    int *c = pdf_cell(f, ix, iy);
    if (c) *c = f->npts;
    f->pts[2 * f->npts] = x;
    f->pts[2 * f->npts + 1] = y;
    f->npts++;
}
static int pdf_test(Pdf *f, double x, double y) {
    int ix = (int)floor(x / f->grid), iy = (int)floor(y / f->grid);
    int bx = ix - f->span, by = iy - f->span, ex = ix + f->span, ey = iy + f->span;
    int cx = bx, cy = by;
    while (cy <= ey) {
        cx++;
        if (cx > ex) {
            cx = bx;
            cy++;
        }
        int *c = pdf_cell(f, cx, cy);
        if (c && *c >= 0) {
            double dx = x - f->pts[2 * *c], dy = y - f->pts[2 * *c + 1];
            if (dx * dx + dy * dy < f->r2) return 0;
        }
    }
    return 1;
}

/* OpenCvImage::detect_keypoints (opencv_image.cpp:38-73).  `existing` (n_exist
 * double xy) are the already tracked points; new points are written to out_xy
 * (double), count returned.  KeyPoint::response is taken to be the corner
 * quality (OpenCV >= 4.5.x), so the reference's std::sort by response keeps the
 * goodFeaturesToTrack order. */
int orc_detect_keypoints(const uint8_t *img, int w, int h, int stride, const double *existing, int n_exist,
                         int max_points, double min_dist, double *out_xy) {
    float *xy = (float *)malloc(sizeof(float) * 2 * (size_t)(max_points > 0 ? max_points : w * h));
    int n = orc_gftt(img, w, h, stride, max_points, 1.0e-3, 20, 0.04, xy, NULL);
    int nout = 0;
    if (n > 0) {
        Pdf f;
        double xmin = 0, ymin = 0, xmax = w, ymax = h;
        for (int i = 0; i < n_exist; ++i) {
            if (existing[2 * i] < xmin) xmin = existing[2 * i];
            if (existing[2 * i] > xmax) xmax = existing[2 * i];
            if (existing[2 * i + 1] < ymin) ymin = existing[2 * i + 1];
            if (existing[2 * i + 1] > ymax) ymax = existing[2 * i + 1];
        }
        pdf_init(&f, min_dist, xmin, ymin, xmax, ymax, n_exist + n);
        for (int i = 0; i < n_exist; ++i) pdf_preset(&f, existing[2 * i], existing[2 * i + 1]);
        for (int i = 0; i < n; ++i) {
            double x = xy[2 * i], y = xy[2 * i + 1];
            if (pdf_test(&f, x, y)) {
                pdf_preset(&f, x, y);
                if (!(x < 20 || y < 20 || x >= w - 20 || y >= h - 20)) {
                    out_xy[2 * nout] = x;
                    out_xy[2 * nout + 1] = y;
                    nout++;
                }
            }
        }
        pdf_free(&f);
    }
    free(xy);
    return nout;
}

/* Stand-alone Poisson "permit/preset" pass used by Frame::track_keypoints
 * (map/frame.cpp:152-163): points visited in the given order; keep[i]=1 if
 * permitted (and then preset). */
void orc_poisson_select(const double *pts, int n, double radius, uint8_t *keep) {
    Pdf f;
    double xmin = 0, ymin = 0, xmax = 1, ymax = 1;
    for (int i = 0; i < n; ++i) {
        if (pts[2 * i] < xmin) xmin = pts[2 * i];
        if (pts[2 * i] > xmax) xmax = pts[2 * i];
        if (pts[2 * i + 1] < ymin) ymin = pts[2 * i + 1];
        if (pts[2 * i + 1] > ymax) ymax = pts[2 * i + 1];
    }
    pdf_init(&f, radius, xmin, ymin, xmax, ymax, n);
    for (int i = 0; i < n; ++i) {
        if (pdf_test(&f, pts[2 * i], pts[2 * i + 1])) {
            pdf_preset(&f, pts[2 * i], pts[2 * i + 1]);
            keep[i] = 1;
        } else {
            keep[i] = 0;
        }
    }
    pdf_free(&f);
}

/* cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) on a packed 1/32-pixel fixed-point map ([h][w][2]: word 0 = int16 sx |
 * int16 sy << 16, word 1 = ax | ay << 8), the per-frame half of cv::undistort as the reference's readers use it
 * (xrslam-pc/player/src/IO/euroc_dataset_reader.cpp:62-69); same arithmetic as oracle/undistort.py::_remap_fixed. */
void orc_remap_packed(const uint32_t *map2, int w, int h, const uint8_t *src, int sstride, uint8_t *dst, int dstride) {
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) {
            const uint32_t m0 = map2[2 * ((size_t)i * w + j)], m1 = map2[2 * ((size_t)i * w + j) + 1];
            const int sx = (int16_t)(m0 & 0xffffu), sy = (int16_t)(m0 >> 16), ax = (int)(m1 & 31u), ay = (int)((m1 >> 8) & 31u);
            long long acc = 0;
            for (int dy = 0; dy < 2; ++dy)
                for (int dx = 0; dx < 2; ++dx) {
                    const int y = sy + dy, x = sx + dx;
                    const int px = (x >= 0 && x < w && y >= 0 && y < h) ? src[(size_t)y * sstride + x] : 0;
                    const int wy = dy ? ay : 32 - ay, wx = dx ? ax : 32 - ax;
                    acc += (long long)px * (wy * wx * 32);
                }
            long long v = (acc + (1 << 14)) >> 15;
            dst[(size_t)i * dstride + j] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
}
