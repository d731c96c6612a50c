// klt_kernels.hip.h -- gfx950 device kernels of the KLT front-end.
//
// Replaces the OpenCV calls behind xrslam::extra::OpenCvImage
// (/root/reference/xrslam-extra/src/xrslam/extra/opencv_image.cpp):
//   k_clahe_lut / k_clahe_apply  <- cv::CLAHE::apply           (:157, :179-182)
//   k_pyrdown / k_scharr         <- cv::buildOpticalFlowPyramid (:159)
//   k_lk_track                   <- cv::calcOpticalFlowPyrLK x2 + gates (:94-135)
//   k_harris / k_harris_nms      <- cv::GFTTDetector::detect    (:44, :184-188)
//
// Arithmetic contract (shared with oracle/klt_oracle.c, which is the checker):
// every image quantity is integer-exact; LK normal-equation sums are exact
// int64 reductions converted to float once; all per-point float math is scalar
// IEEE (compiled with -ffp-contract=off) so status bits AND positions are
// bit-identical to the oracle.
//
// Execution model: wave64.  LK runs one wavefront per keypoint, the 21x21
// window is spread over the 64 lanes (7 pixels per lane, template held in
// registers), the search neighbourhood of the second image is staged in LDS
// once per pyramid level (whole dwords, coalesced), and the 2x2 normal-equation
// sums are reduced with cross-lane butterflies -- no atomics and no branch per
// pixel slot in the iteration loop, so the seven gathers of a step are in
// flight together.
#pragma once
#include <hip/hip_runtime.h>

#include "batch.hip.h"
#include <stdint.h>

namespace xrhip {

constexpr int KLT_LEVELS = 4;
constexpr int KLT_WIN = 21;
constexpr int KLT_PAD = 21;    // border rows above/below and logical border columns
constexpr int KLT_PADX = 32;   // physical left border in pixels (keeps x=0 32-byte aligned)

struct LevelView {
    const uint8_t *img;   // pointer to pixel (0,0) of the padded image
    const short2 *der;    // pointer to pixel (0,0) of the padded derivative image (dx,dy)
    int w, h;
    int istride;          // bytes (== pixels) per padded image row
    int pstride;          // short2 elements per padded derivative row
};

struct PyrView {
    LevelView lv[KLT_LEVELS];
};

// Every front-end kernel takes its arguments as Batch<Args> (batch.hip.h): up to XB argument sets, one per blockIdx.z.

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * (n - 1) - i;
    }
    return i;
}

// ---------------------------------------------------------------- CLAHE LUT
// One workgroup per tile: LDS histogram, clip + redistribute, inclusive scan,
// LUT.  (cv::CLAHE_CalcLut_Body restated; integer exact.)
constexpr int CL_THREADS = 1024;   // sixteen wavefronts count a tile's pixels (one batch of loads each), four of them finish the LUT
__device__ __forceinline__ void d_clahe_lut(const uint8_t *__restrict__ src, int sstride, int w, int h,
                                            int tw, int th, int tiles_x, int clip, float lut_scale,
                                            uint8_t *__restrict__ lut) {
    __shared__ int hist[CL_THREADS / 64][256];   // one histogram per wavefront: neighbouring pixels share grey levels, a single one serialises its atomics
    __shared__ int scan[2][256];
    const int tid = threadIdx.x, wv = tid >> 6;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    for (int i = tid; i < (CL_THREADS / 64) * 256; i += CL_THREADS) (&hist[0][0])[i] = 0;
    __syncthreads();
    const int area = tw * th;
    for (int i0 = tid; i0 < area; i0 += 8 * CL_THREADS) {   // eight pixel loads in flight per thread, then their (exact, integer) counts
        int px[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(i0 + u * CL_THREADS, area - 1);
            const int y = i / tw;
            const int x = i - y * tw;
            const int gx = reflect101(tx * tw + x, w), gy = reflect101(ty * th + y, h);
            px[u] = src[(size_t)gy * sstride + gx];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * CL_THREADS < area) atomicAdd(&hist[wv][px[u]], 1);
    }
    __syncthreads();
    const bool fin = tid < 256;   // the clip / redistribute / scan steps are 256 wide: the other wavefronts only keep the barriers
    const int t = tid & 255;
    int hv = 0;
    if (fin) {
#pragma unroll
        for (int q = 0; q < CL_THREADS / 64; ++q) hv += hist[q][t];
    }
    if (clip > 0) {
        int excess = hv > clip ? hv - clip : 0;
        if (hv > clip) hv = clip;
        if (fin) scan[0][t] = excess;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (fin && t < s) scan[0][t] += scan[0][t + s];
            __syncthreads();
        }
        int clipped = scan[0][0];
        __syncthreads();
        int redist = clipped / 256;
        int residual = clipped - redist * 256;
        hv += redist;
        if (residual != 0) {
            int step = 256 / residual;
            if (step < 1) step = 1;
            if (t % step == 0 && t / step < residual) hv++;
        }
    }
    // inclusive scan (Hillis-Steele, double buffered)
    int cur = 0;
    if (fin) scan[0][t] = hv;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        if (fin) {
            int v = scan[cur][t];
            if (t >= off) v += scan[cur][t - off];
            scan[cur ^ 1][t] = v;
        }
        cur ^= 1;
        __syncthreads();
    }
    if (!fin) return;
    int sum = scan[cur][t];
    int v = __float2int_rn((float)sum * lut_scale);
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    lut[(size_t)blockIdx.x * 256 + t] = (uint8_t)v;
}
struct ClaheLutArgs {
    const uint8_t *src;
    int sstride, w, h, tw, th, tiles_x, clip;
    float lut_scale;
    uint8_t *lut;
    int tiles;   // blocks of this entry
};
__global__ __launch_bounds__(CL_THREADS) void k_clahe_lut(Batch<ClaheLutArgs> b) {
    const ClaheLutArgs &a = b.e[blockIdx.z];
    if ((int)blockIdx.x >= a.tiles) return;
    d_clahe_lut(a.src, a.sstride, a.w, a.h, a.tw, a.th, a.tiles_x, a.clip, a.lut_scale, a.lut);
}

// -------------------------------------------------------------- CLAHE apply
// Writes pyramid level 0 INCLUDING its reflect-101 border in one pass.
// (cv::CLAHE_Interpolation_Body restated; float math as in the oracle.)
__global__ __launch_bounds__(256) void k_clahe_apply(const uint8_t *__restrict__ src, int sstride, int w, int h,
                                                     int tw, int th, int tiles_x, int tiles_y,
                                                     const uint8_t *__restrict__ lut, uint8_t *__restrict__ dst0,
                                                     int dstride) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x - KLT_PAD;
    const int y = blockIdx.y * blockDim.y + threadIdx.y - KLT_PAD;
    if (x >= w + KLT_PAD || y >= h + KLT_PAD) return;
    const int sx = reflect101(x, w), sy = reflect101(y, h);
    const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
    float tyf = sy * inv_th - 0.5f;
    int ty1 = (int)floorf(tyf);
    int ty2 = ty1 + 1;
    float ya = tyf - ty1, ya1 = 1.0f - ya;
    if (ty1 < 0) ty1 = 0;
    if (ty2 > tiles_y - 1) ty2 = tiles_y - 1;
    float txf = sx * inv_tw - 0.5f;
    int tx1 = (int)floorf(txf);
    int tx2 = tx1 + 1;
    float xa = txf - tx1, xa1 = 1.0f - xa;
    if (tx1 < 0) tx1 = 0;
    if (tx2 > tiles_x - 1) tx2 = tiles_x - 1;
    const int v = src[(size_t)sy * sstride + sx];
    const uint8_t *p1 = lut + (size_t)ty1 * tiles_x * 256;
    const uint8_t *p2 = lut + (size_t)ty2 * tiles_x * 256;
    const int i1 = tx1 * 256 + v, i2 = tx2 * 256 + v;
    float res = (p1[i1] * xa1 + p1[i2] * xa) * ya1 + (p2[i1] * xa1 + p2[i2] * xa) * ya;
    int r = __float2int_rn(res);
    r = r < 0 ? 0 : (r > 255 ? 255 : r);
    dst0[(ptrdiff_t)y * dstride + x] = (uint8_t)r;
}

// ------------------------------------------------------------------ pyrDown
// cv::pyrDown 8U ([1 4 6 4 1]^2, (sum+128)>>8).  The source level already
// carries a reflect-101 border >= 2, so taps read it directly.  Writes the
// destination level including its own reflect-101 border.
__global__ __launch_bounds__(256) void k_pyrdown(LevelView src, uint8_t *__restrict__ dst0, int dw, int dh,
                                                 int dstride) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x - KLT_PAD;
    const int y = blockIdx.y * blockDim.y + threadIdx.y - KLT_PAD;
    if (x >= dw + KLT_PAD || y >= dh + KLT_PAD) return;
    const int ox = reflect101(x, dw), oy = reflect101(y, dh);
    const uint8_t *s = src.img + (ptrdiff_t)(2 * oy - 2) * src.istride + (2 * ox - 2);
    int acc = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int wj = (j == 0 || j == 4) ? 1 : ((j == 2) ? 6 : 4);
        const uint8_t *r = s + (ptrdiff_t)j * src.istride;
        int row = r[0] + r[4] + 4 * (r[1] + r[3]) + 6 * r[2];
        acc += wj * row;
    }
    dst0[(ptrdiff_t)y * dstride + x] = (uint8_t)((acc + 128) >> 8);
}

// ------------------------------------------------------------------- Scharr
// cv::detail::calcSharrDeriv for all levels in one launch.  blk_off[l] is the
// first linear block index of level l (blocks are 64x4 pixel tiles).
struct ScharrArgs {
    LevelView lv[KLT_LEVELS];
    short2 *out[KLT_LEVELS];   // pixel (0,0) of each padded derivative buffer
    int blk_off[KLT_LEVELS + 1];
    int blk_w[KLT_LEVELS];     // blocks per row of each level
};

__global__ __launch_bounds__(256) void k_scharr(ScharrArgs a) {
    int b = blockIdx.x;
    int l = 0;
#pragma unroll
    for (int i = 1; i < KLT_LEVELS; ++i)
        if (b >= a.blk_off[i]) l = i;
    b -= a.blk_off[l];
    const LevelView L = a.lv[l];
    const int bx = b % a.blk_w[l], by = b / a.blk_w[l];
    const int x = bx * 64 + (threadIdx.x & 63);
    const int y = by * 4 + (threadIdx.x >> 6);
    if (x >= L.w || y >= L.h) return;
    const uint8_t *r0 = L.img + (ptrdiff_t)(y - 1) * L.istride + x;
    const uint8_t *r1 = r0 + L.istride;
    const uint8_t *r2 = r1 + L.istride;
    int t0m = (r0[-1] + r2[-1]) * 3 + r1[-1] * 10;
    int t0p = (r0[1] + r2[1]) * 3 + r1[1] * 10;
    int t1m = r2[-1] - r0[-1];
    int t1c = r2[0] - r0[0];
    int t1p = r2[1] - r0[1];
    short2 d;
    d.x = (short)(t0p - t0m);
    d.y = (short)((t1p + t1m) * 3 + t1c * 10);
    a.out[l][(ptrdiff_t)y * L.pstride + x] = d;
}

// ------------------------------------------------------- fused pyramid build
// The CLAHE plane, the three pyrDown levels and the Scharr derivatives of all four levels in TWO launches instead of five
// (k_clahe_apply + 3 x k_pyrdown + k_scharr):
//   k_pyr_a  level 0 (CLAHE interpolation) + level 1 + the level-0 derivatives; one workgroup per 32x32 tile of level 0
//   k_pyr_b  levels 2, 3 + the derivatives of levels 1, 2, 3 from the level-1 plane; one workgroup per 32x32 tile of level 1
// A workgroup builds its tile of every level it owns in LDS, together with the halo the level above and its own Scharr stencil
// read, so it depends on nothing another workgroup of the same launch writes:
//   k_pyr_a: level 1 16x16 <- level 0 35x35 (2 * 16 + 3; 1.2x the tile)
//   k_pyr_b: level 3 8 + 2 = 10 <- level 2 2 * 10 + 3 = 23 <- level 1 2 * 23 + 3 = 49 (read from the padded level-1 plane)
// A value at a coordinate outside the image is what the padded planes hold there: the level's own value at the reflect-101
// coordinate (k_clahe_apply / k_pyrdown write their borders that way), i.e. the SAME integer expressions evaluated at the
// reflected output coordinate -- bit-identical planes.  The 21-pixel border of each plane is written by the workgroup that owns
// the mirrored interior pixel.  Halo entries further than two pixels outside the image are never read by a valid output (a
// pyrDown tap reaches at most 2 * (w' - 1) + 2 <= w + 1); their reflected sources may fall outside the tile and are clamped.
// (First version: all four levels in one launch, 64x64 tiles with a 101x101 level-0 halo -- 2.5x redundant CLAHE work on 96 of
// the 256 CUs: 17.4 us against 23.5 us for the five kernels it replaced, profiles/r03_ab_variants.md.)
// Used when every level is at least 2 * KLT_PAD + 2 wide and high (one reflection covers the border); otherwise the five-launch
// path runs.
constexpr int PF_THREADS = 256;
__device__ __forceinline__ int pf_reflect(int i, int n) {   // one reflection, then clamped (far halo entries: see above)
    if (i < 0) i = -i;
    else if (i >= n) i = 2 * (n - 1) - i;
    return min(max(i, 0), n - 1);
}
// store a plane value at (x, y) and at its mirror images inside the KLT_PAD border
__device__ __forceinline__ void pf_store(uint8_t *img, int stride, int w, int h, int x, int y, uint8_t v) {
    img[(ptrdiff_t)y * stride + x] = v;
    const bool xl = x >= 1 && x <= KLT_PAD, xr = x <= w - 2 && x >= w - 1 - KLT_PAD;
    const bool yt = y >= 1 && y <= KLT_PAD, yb = y <= h - 2 && y >= h - 1 - KLT_PAD;
    if (!(xl || xr || yt || yb)) return;   // (nearly every pixel; whole wavefronts in the interior tiles)
    int xs[3], ys[3], nx = 1, ny = 1;
    xs[0] = x;
    ys[0] = y;
    if (xl) xs[nx++] = -x;
    if (xr) xs[nx++] = 2 * (w - 1) - x;
    if (yt) ys[ny++] = -y;
    if (yb) ys[ny++] = 2 * (h - 1) - y;
    for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i)
            if (i + j) img[(ptrdiff_t)ys[j] * stride + xs[i]] = v;
}
// Scharr derivative of one level from its LDS tile (calcSharrDeriv, as k_scharr) for the tile's valid output pixels:
// P = LDS row pitch, LO = halo below the tile origin, S = tile size, (X, Y) = tile origin
template <int P, int LO, int S>
__device__ __forceinline__ void pf_scharr(const uint8_t *T, int X, int Y, int w, int h, short2 *der, int stride, int tid) {
    for (int i = tid; i < S * S; i += PF_THREADS) {
        const int ly = i / S, lx = i - ly * S, x = X + lx, y = Y + ly;
        if (x >= w || y >= h) continue;
        const uint8_t *r1 = T + (ly + LO) * P + lx + LO, *r0 = r1 - P, *r2 = r1 + P;
        const int t0m = (r0[-1] + r2[-1]) * 3 + r1[-1] * 10;
        const int t0p = (r0[1] + r2[1]) * 3 + r1[1] * 10;
        const int t1m = r2[-1] - r0[-1], t1c = r2[0] - r0[0], t1p = r2[1] - r0[1];
        short2 d;
        d.x = (short)(t0p - t0m);
        d.y = (short)((t1p + t1m) * 3 + t1c * 10);
        der[(ptrdiff_t)y * stride + x] = d;
    }
}
// one pyrDown step LDS -> LDS (+ the plane): destination tile RD x RD (halo LOD, SD x SD outputs) at level origin (XD, YD) of a
// wd x hd level, from the source tile (extent RS, pitch PS) whose entry (0, 0) is source coordinate (2 XD - LOS, 2 YD - LOS)
template <int RD, int PD, int LOD, int SD, int RS, int PS, int LOS>
__device__ __forceinline__ void pf_down(const uint8_t *Ts, uint8_t *Td, int XD, int YD, int wd, int hd, uint8_t *img, int stride,
                                        int tid) {
    const int XS = 2 * XD, YS = 2 * YD;
    for (int i = tid; i < RD * RD; i += PF_THREADS) {
        const int ry = i / RD, rx = i - ry * RD;
        const int x = XD - LOD + rx, y = YD - LOD + ry;
        const int ox = pf_reflect(x, wd), oy = pf_reflect(y, hd);
        const int cx = min(max(2 * ox - 2 - (XS - LOS), 0), RS - 5), cy = min(max(2 * oy - 2 - (YS - LOS), 0), RS - 5);
        const uint8_t *s0 = Ts + cy * PS + cx;
        int acc = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int wj = (j == 0 || j == 4) ? 1 : ((j == 2) ? 6 : 4);
            const uint8_t *r = s0 + j * PS;
            const int row = r[0] + r[4] + 4 * (r[1] + r[3]) + 6 * r[2];
            acc += wj * row;
        }
        const uint8_t v = (uint8_t)((acc + 128) >> 8);
        Td[ry * PD + rx] = v;
        if (rx >= LOD && rx < LOD + SD && ry >= LOD && ry < LOD + SD && x < wd && y < hd) pf_store(img, stride, wd, hd, x, y, v);
    }
}

constexpr int PA_T = 32;                      // k_pyr_a: level-0 tile
constexpr int PA_R0 = 35, PA_LO0 = 2, PA_P0 = 36;   // level-0 LDS tile: [X0 - 2, X0 + 33)
constexpr int PA_R1 = 16, PA_P1 = 16;        // level-1 tile (no halo: nothing above it in this launch)
struct PyrAArgs {
    const uint8_t *raw;
    int rstride;
    int tw, th, tiles_x, tiles_y;   // CLAHE tile grid
    const uint8_t *lut;
    uint8_t *img0, *img1;           // pixel (0,0) of the padded planes
    short2 *der0;
    int w0, h0, s0, w1, h1, s1;
    int tiles_across;
    int blocks;   // tiles of this entry
};
__device__ __forceinline__ void d_pyr_a(const PyrAArgs &a) {
    __shared__ uint8_t T0[PA_R0 * PA_P0], T1[PA_R1 * PA_P1];
    __shared__ int cs[PA_R0], ci1[PA_R0], ci2[PA_R0], rs[PA_R0], rp1[PA_R0], rp2[PA_R0];
    __shared__ float cxa[PA_R0], cxa1[PA_R0], rya[PA_R0], rya1[PA_R0];
    const int tid = threadIdx.x;
    const int bx = blockIdx.x % a.tiles_across, by = blockIdx.x / a.tiles_across;
    const int X0 = PA_T * bx, Y0 = PA_T * by;
    // ---- level 0: CLAHE interpolation (cv::CLAHE_Interpolation_Body, the float expressions of k_clahe_apply).  What depends on the
    // column alone (source column, the two LUT columns, the horizontal weights) and on the row alone is formed once per tile column /
    // row; a pixel is then four LUT bytes and seven float operations.
    const int w = a.w0, h = a.h0;
    if (tid < PA_R0) {
        const float inv_tw = 1.0f / a.tw;
        const int sx = pf_reflect(X0 - PA_LO0 + tid, w);
        float txf = sx * inv_tw - 0.5f;
        int tx1 = (int)floorf(txf);
        int tx2 = tx1 + 1;
        const float xa = txf - tx1, xa1 = 1.0f - xa;
        if (tx1 < 0) tx1 = 0;
        if (tx2 > a.tiles_x - 1) tx2 = a.tiles_x - 1;
        cs[tid] = sx;
        ci1[tid] = tx1 * 256;
        ci2[tid] = tx2 * 256;
        cxa[tid] = xa;
        cxa1[tid] = xa1;
    } else if (tid >= 64 && tid < 64 + PA_R0) {
        const int r = tid - 64;
        const float inv_th = 1.0f / a.th;
        const int sy = pf_reflect(Y0 - PA_LO0 + r, h);
        float tyf = sy * inv_th - 0.5f;
        int ty1 = (int)floorf(tyf);
        int ty2 = ty1 + 1;
        const float ya = tyf - ty1, ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > a.tiles_y - 1) ty2 = a.tiles_y - 1;
        rs[r] = sy;
        rp1[r] = ty1 * a.tiles_x * 256;
        rp2[r] = ty2 * a.tiles_x * 256;
        rya[r] = ya;
        rya1[r] = ya1;
    }
    __syncthreads();
    {   // (the frame's pixels first, all of a thread's loads in flight together, then the interpolation)
        constexpr int N0 = (PA_R0 * PA_R0 + PF_THREADS - 1) / PF_THREADS;
        int pix[N0];
#pragma unroll
        for (int u = 0; u < N0; ++u) {
            const int i = min(tid + u * PF_THREADS, PA_R0 * PA_R0 - 1);
            const int ry = i / PA_R0, rx = i - ry * PA_R0;
            pix[u] = a.raw[(size_t)rs[ry] * a.rstride + cs[rx]];
        }
#pragma unroll
        for (int u = 0; u < N0; ++u) {
            const int i = tid + u * PF_THREADS;
            if (i >= PA_R0 * PA_R0) break;
            const int ry = i / PA_R0, rx = i - ry * PA_R0;
            const int v = pix[u];
            const uint8_t *p1 = a.lut + rp1[ry], *p2 = a.lut + rp2[ry];
            const int i1 = ci1[rx] + v, i2 = ci2[rx] + v;
            const float xa = cxa[rx], xa1 = cxa1[rx], ya = rya[ry], ya1 = rya1[ry];
            const float res = (p1[i1] * xa1 + p1[i2] * xa) * ya1 + (p2[i1] * xa1 + p2[i2] * xa) * ya;
            int r = __float2int_rn(res);
            r = r < 0 ? 0 : (r > 255 ? 255 : r);
            T0[ry * PA_P0 + rx] = (uint8_t)r;
            const int x = X0 - PA_LO0 + rx, y = Y0 - PA_LO0 + ry;
            if (rx >= PA_LO0 && rx < PA_LO0 + PA_T && ry >= PA_LO0 && ry < PA_LO0 + PA_T && x < w && y < h)
                pf_store(a.img0, a.s0, w, h, x, y, (uint8_t)r);
        }
    }
    __syncthreads();
    pf_down<PA_R1, PA_P1, 0, PA_T / 2, PA_R0, PA_P0, PA_LO0>(T0, T1, X0 >> 1, Y0 >> 1, a.w1, a.h1, a.img1, a.s1, tid);
    pf_scharr<PA_P0, PA_LO0, PA_T>(T0, X0, Y0, w, h, a.der0, a.s0, tid);
}
__global__ __launch_bounds__(PF_THREADS) void k_pyr_a(Batch<PyrAArgs> b) {
    const PyrAArgs &a = b.e[blockIdx.z];
    if ((int)blockIdx.x >= a.blocks) return;
    d_pyr_a(a);
}

constexpr int PB_T = 32;                                  // k_pyr_b: level-1 tile = 16x16 of level 2 = 8x8 of level 3
constexpr int PB_R1 = 49, PB_R2 = 23, PB_R3 = 10;        // LDS tile extents
constexpr int PB_LO1 = 10, PB_LO2 = 4, PB_LO3 = 1;       // halo below the tile origin
constexpr int PB_P1 = 52, PB_P2 = 24, PB_P3 = 12;        // LDS row pitches
struct PyrBArgs {
    uint8_t *img[KLT_LEVELS];       // pixel (0,0) of each padded plane (level 1: read, with its border; 2, 3: written)
    short2 *der[KLT_LEVELS];
    int w[KLT_LEVELS], h[KLT_LEVELS], istride[KLT_LEVELS];
    int tiles_across;
    int blocks;   // tiles of this entry
};
__device__ __forceinline__ void d_pyr_b(const PyrBArgs &a) {
    __shared__ uint8_t T1[PB_R1 * PB_P1], T2[PB_R2 * PB_P2], T3[PB_R3 * PB_P3];
    const int tid = threadIdx.x;
    const int bx = blockIdx.x % a.tiles_across, by = blockIdx.x / a.tiles_across;
    const int X1 = PB_T * bx, Y1 = PB_T * by;
    {   // the level-1 tile with its halo, straight from the padded plane (halo <= 10 + 6 < KLT_PAD); coordinates beyond the
        // plane's border (tiles past the image's last row / column: never read by a valid output) are clamped into it
        const int w = a.w[1], h = a.h[1];
        constexpr int N1 = (PB_R1 * PB_R1 + PF_THREADS - 1) / PF_THREADS;
        uint8_t px[N1];
#pragma unroll
        for (int u = 0; u < N1; ++u) {   // all of a thread's loads in flight together
            const int i = min(tid + u * PF_THREADS, PB_R1 * PB_R1 - 1);
            const int ry = i / PB_R1, rx = i - ry * PB_R1;
            const int x = min(X1 - PB_LO1 + rx, w + KLT_PAD - 1), y = min(Y1 - PB_LO1 + ry, h + KLT_PAD - 1);
            px[u] = a.img[1][(ptrdiff_t)y * a.istride[1] + x];
        }
#pragma unroll
        for (int u = 0; u < N1; ++u) {
            const int i = tid + u * PF_THREADS;
            if (i < PB_R1 * PB_R1) T1[(i / PB_R1) * PB_P1 + (i % PB_R1)] = px[u];
        }
    }
    __syncthreads();
    pf_down<PB_R2, PB_P2, PB_LO2, PB_T / 2, PB_R1, PB_P1, PB_LO1>(T1, T2, X1 >> 1, Y1 >> 1, a.w[2], a.h[2], a.img[2], a.istride[2], tid);
    pf_scharr<PB_P1, PB_LO1, PB_T>(T1, X1, Y1, a.w[1], a.h[1], a.der[1], a.istride[1], tid);
    __syncthreads();
    pf_down<PB_R3, PB_P3, PB_LO3, PB_T / 4, PB_R2, PB_P2, PB_LO2>(T2, T3, X1 >> 2, Y1 >> 2, a.w[3], a.h[3], a.img[3], a.istride[3], tid);
    pf_scharr<PB_P2, PB_LO2, PB_T / 2>(T2, X1 >> 1, Y1 >> 1, a.w[2], a.h[2], a.der[2], a.istride[2], tid);
    __syncthreads();
    pf_scharr<PB_P3, PB_LO3, PB_T / 4>(T3, X1 >> 2, Y1 >> 2, a.w[3], a.h[3], a.der[3], a.istride[3], tid);
}
__global__ __launch_bounds__(PF_THREADS) void k_pyr_b(Batch<PyrBArgs> b) {
    const PyrBArgs &a = b.e[blockIdx.z];
    if ((int)blockIdx.x >= a.blocks) return;
    d_pyr_b(a);
}

// ------------------------------------------------------------------- Harris
__device__ __forceinline__ int float_order_key(float f) {
    int b = __float_as_int(f);
    return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float float_from_key(int k) {
    return __int_as_float(k ^ ((k >> 31) & 0x7fffffff));
}

// cv::cornerHarris(8U, block 3, ksize 3, k) on the CLAHE image (pyramid level 0).
// 64x16 output tile per workgroup; the Sobel dx/dy of the tile and its one-pixel ring are staged in LDS as int16, the
// 3x3 window sums are exact int32.  Also reduces the global maximum.
//
// Round 4 (last change of the round), two things the kernel's 22 us were made of:
//  * the maximum went to ONE word through an atomicMax per WAVEFRONT -- 1440 same-address atomics that arrive together (all 360
//    workgroups are resident at once and take the same time) and retire at ~100 per microsecond: a ~14 us tail.  Now the four
//    wavefronts of a workgroup combine in LDS and the workgroup issues one (max is order-independent: same result);
//  * every Sobel element gathered its 8 neighbours as single bytes from global memory, five dependent round trips per thread.  Now the
//    pixels the tile needs (20 rows x 72 columns, whole dwords, coalesced) are fetched once into LDS and the Sobel reads them there.
//    Border semantics unchanged: the 3x3 box reflects the Sobel CENTRE (BORDER_REFLECT_101 of the covariance planes), the
//    Sobel itself reads the centre's neighbours from the padded plane (BORDER_REFLECT_101 of the image) -- a reflected centre and
//    its neighbours lie inside the fetched rectangle for every element a valid output reads; the others are clamped into it
//    (their values are never used).
constexpr int HR_TW = 72, HR_TH = 20;   // pixel columns X0-4 .. X0+67 (dword-aligned), rows Y0-2 .. Y0+17
__device__ __forceinline__ void d_harris(LevelView L, double kk, float s2, float *__restrict__ resp,
                                         int *__restrict__ max_key) {
    __shared__ short sdx[18][66];
    __shared__ short sdy[18][66];
    __shared__ uint32_t pix4[HR_TH * HR_TW / 4];
    __shared__ float wmax[4];
    const int X0 = blockIdx.x * 64, Y0 = blockIdx.y * 16;
    const int tid = threadIdx.x;
    const bool staged = L.w >= 4 && L.h >= 4;   // (smaller planes: more than one reflection of a centre is possible; gather as before)
    if (staged) {
        for (int i = tid; i < HR_TH * HR_TW / 4; i += 256) {
            const int r = i / (HR_TW / 4), c4 = i - r * (HR_TW / 4);
            // rows / dwords beyond the padded plane are never part of a valid output's footprint: clamp the address
            const int y = min(max(Y0 - 2 + r, -KLT_PAD), L.h + KLT_PAD - 1);
            const int x = min(max(X0 - 4 + 4 * c4, -KLT_PADX), L.istride - KLT_PADX - 4);
            pix4[i] = *reinterpret_cast<const uint32_t *>(L.img + (ptrdiff_t)y * L.istride + x);
        }
        __syncthreads();
        const uint8_t *pix = reinterpret_cast<const uint8_t *>(pix4);
        for (int i = tid; i < 18 * 66; i += 256) {
            const int cy = i / 66, cx = i - cy * 66;
            const int px = reflect101(X0 - 1 + cx, L.w), py = reflect101(Y0 - 1 + cy, L.h);
            const int lx = min(max(px - (X0 - 4), 1), HR_TW - 2), ly = min(max(py - (Y0 - 2), 1), HR_TH - 2);
            const uint8_t *r0 = pix + (ly - 1) * HR_TW + lx;
            const uint8_t *r1 = r0 + HR_TW;
            const uint8_t *r2 = r1 + HR_TW;
            const int dx = (r0[1] - r0[-1]) + 2 * (r1[1] - r1[-1]) + (r2[1] - r2[-1]);
            const int dy = (r2[-1] - r0[-1]) + 2 * (r2[0] - r0[0]) + (r2[1] - r0[1]);
            sdx[cy][cx] = (short)dx;
            sdy[cy][cx] = (short)dy;
        }
    } else {
        for (int i = tid; i < 18 * 66; i += 256) {
            int cy = i / 66, cx = i - cy * 66;
            int px = reflect101(X0 - 1 + cx, L.w), py = reflect101(Y0 - 1 + cy, L.h);
            const uint8_t *r0 = L.img + (ptrdiff_t)(py - 1) * L.istride + px;
            const uint8_t *r1 = r0 + L.istride;
            const uint8_t *r2 = r1 + L.istride;
            int dx = (r0[1] - r0[-1]) + 2 * (r1[1] - r1[-1]) + (r2[1] - r2[-1]);
            int dy = (r2[-1] - r0[-1]) + 2 * (r2[0] - r0[0]) + (r2[1] - r0[1]);
            sdx[cy][cx] = (short)dx;
            sdy[cy][cx] = (short)dy;
        }
    }
    __syncthreads();
    const int lx = tid & 63, ly0 = tid >> 6;
    float best = -3.402823466e+38f;
    for (int q = 0; q < 4; ++q) {
        const int ly = ly0 + 4 * q;
        const int x = X0 + lx, y = Y0 + ly;
        if (x < L.w && y < L.h) {
            int sxx = 0, sxy = 0, syy = 0;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    int dx = sdx[ly + j][lx + i], dy = sdy[ly + j][lx + i];
                    sxx += dx * dx;
                    sxy += dx * dy;
                    syy += dy * dy;
                }
            float a = (float)sxx * s2, b = (float)sxy * s2, c = (float)syy * s2;
            float t1 = a * c;
            float t2 = b * b;
            float t3 = t1 - t2;
            float apc = a + c;
            float r = (float)((double)t3 - kk * (double)apc * (double)apc);
            resp[(size_t)y * L.w + x] = r;
            best = fmaxf(best, r);
        }
    }
    // wavefront max, workgroup max through LDS, one atomic per workgroup
    for (int off = 32; off > 0; off >>= 1) best = fmaxf(best, __shfl_xor(best, off));
    if ((tid & 63) == 0) wmax[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) atomicMax(max_key, float_order_key(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}
struct HarrisArgs {
    LevelView L;
    double kk;
    float s2;
    float *resp;
    int *max_key;
    int gx, gy;   // tile grid of this entry
};
__global__ __launch_bounds__(256) void k_harris(Batch<HarrisArgs> b) {
    const HarrisArgs &a = b.e[blockIdx.z];
    if ((int)blockIdx.x >= a.gx || (int)blockIdx.y >= a.gy) return;
    d_harris(a.L, a.kk, a.s2, a.resp, a.max_key);
}

struct HarrisCand {
    float v;
    int idx;   // y*w + x
};

// Response bins of the candidate selection (k_harris_select below; counted here, where the candidates are found).  Bins are spaced
// LOGARITHMICALLY above the threshold (the bit pattern of a positive float is monotone in its value): the responses fall off like a
// power law, so linear bins put most of the ~10^4 candidates into the first few.  256 bins per binade: the 4096 bins span the 16
// binades above the threshold (quality 1e-3 leaves ten between threshold and maximum; anything higher shares the top bin).
constexpr int SEL_BINS = 4096;
__device__ __forceinline__ unsigned sel_thr_bits(float maxv, double quality) {
    return __float_as_uint(fmaxf((float)((double)maxv * quality), 1e-30f));
}
__device__ __forceinline__ int sel_bin(float v, unsigned thr_bits) {
    const unsigned b = __float_as_uint(fmaxf(v, 1e-30f));
    return b > thr_bits ? (int)min((unsigned)(SEL_BINS - 1), (b - thr_bits) >> 15) : 0;
}

// threshold (quality * max) + 3x3 non-maximum suppression over a 64x16 tile per
// workgroup.  Candidates are gathered in LDS and appended with ONE global atomic
// per workgroup (a per-pixel atomic on a single counter serialises at ~100
// atomics/us on this chip).  Order is arbitrary; the host orders candidates by
// the total order (v desc, idx desc).
__device__ __forceinline__ void d_harris_nms(const float *__restrict__ resp, int w, int h,
                                             const int *__restrict__ max_key, double quality,
                                             HarrisCand *__restrict__ cand, int *__restrict__ count,
                                             int capacity, unsigned *__restrict__ sel_hist) {
    __shared__ HarrisCand local[1024];
    __shared__ int nlocal;
    __shared__ int base;
    const int tid = threadIdx.x;
    if (tid == 0) nlocal = 0;
    __syncthreads();
    const float maxv = float_from_key(*max_key);
    const float thr = (float)((double)maxv * quality);
    // The tile's responses and their one-pixel ring go to LDS in one batch of coalesced loads (a pixel above the threshold used to
    // fetch its eight neighbours in a second, dependent round trip -- per row of the tile, four times per thread).  Only interior
    // pixels are tested, so every neighbour that is compared lies inside the plane; positions outside it are clamped (never compared).
    __shared__ float t[18][66];
    const int X0 = blockIdx.x * 64, Y0 = blockIdx.y * 16;
#pragma unroll
    for (int k = 0; k < (18 * 66 + 255) / 256; ++k) {
        const int i = min(tid + 256 * k, 18 * 66 - 1);
        const int cy = i / 66, cx = i - cy * 66;
        const int gx = min(max(X0 - 1 + cx, 0), w - 1), gy = min(max(Y0 - 1 + cy, 0), h - 1);
        t[cy][cx] = resp[(size_t)gy * w + gx];
    }
    __syncthreads();
    const int lx = tid & 63, ly0 = tid >> 6;
    for (int q = 0; q < 4; ++q) {
        const int ly = ly0 + 4 * q;
        const int x = X0 + lx;
        const int y = Y0 + ly;
        if (x < 1 || y < 1 || x >= w - 1 || y >= h - 1) continue;
        const float v = t[ly + 1][lx + 1];
        if (!(v > thr) || v == 0.f) continue;
        bool ok = v >= t[ly + 1][lx] && v >= t[ly + 1][lx + 2] && v >= t[ly][lx] && v >= t[ly][lx + 1] && v >= t[ly][lx + 2] &&
                  v >= t[ly + 2][lx] && v >= t[ly + 2][lx + 1] && v >= t[ly + 2][lx + 2];
        if (!ok) continue;
        int slot = atomicAdd(&nlocal, 1);
        local[slot].v = v;
        local[slot].idx = y * w + x;
    }
    __syncthreads();
    const int n = nlocal;
    if (n == 0) return;
    if (tid == 0) base = atomicAdd(count, n);
    __syncthreads();
    // each candidate is also counted in its response bin (round 6): the selection kernel -- one workgroup -- spent 3 of its 25 us reading
    // all candidates a first time for this histogram; here it is one fire-and-forget L2 atomic beside the store
    const unsigned thr_bits = sel_thr_bits(maxv, quality);
    for (int i = tid; i < n; i += 256) {
        int slot = base + i;
        if (slot < capacity) {
            cand[slot] = local[i];
            atomicAdd(&sel_hist[sel_bin(local[i].v, thr_bits)], 1u);
        }
    }
}
struct HarrisNmsArgs {
    const float *resp;
    int w, h;
    const int *max_key;
    double quality;
    HarrisCand *cand;
    int *count;
    int capacity;
    int gx, gy;
    unsigned *sel_hist;
};
__global__ __launch_bounds__(256) void k_harris_nms(Batch<HarrisNmsArgs> b) {
    const HarrisNmsArgs &a = b.e[blockIdx.z];
    if ((int)blockIdx.x >= a.gx || (int)blockIdx.y >= a.gy) return;
    d_harris_nms(a.resp, a.w, a.h, a.max_key, a.quality, a.cand, a.count, a.capacity, a.sel_hist);
}

// The host's greedy spacing pass visits candidates in (response desc, index desc) order and normally stops
// after a few hundred of the ~10^4 NMS survivors.  This kernel hands it a superset of the `keep` strongest, in
// visiting order: responses are binned logarithmically above the quality threshold (monotone in the response), the
// bin boundary that keeps >= keep candidates is found, everything at or above it is counting-sorted by bin and
// ranked inside its bin, and written -- together with a header and, last, a sequence number -- straight into
// pinned host memory.  One workgroup.
//
// Round 6 (in-kernel timers, profiles/r06_select_phases.md): the kernel was 3.2 us histogram + 1.1 boundary + 4.5 selection
// (a second read of all candidates) + 9.0 bitonic sort of 1024 + 2.3 fence, + a second fence behind the header.  Now the histogram
// is counted where the candidates are found (k_harris_nms: all workgroups), every thread reads its candidates once, beside the
// histogram, and keeps them in registers; the histogram's suffix sums give each bin its slot range (the candidates above the
// boundary are spread over ~10^3 bins, a handful each), a candidate's final slot is its bin's start + the number of its bin mates
// that precede it, and the header rides in front of the one fence.
constexpr int SEL_K = 896;     // the least `keep`: a few more than the ~400 the spacing pass usually visits for 150 corners
constexpr int SEL_SORT = 8192; // selected candidates the (dynamic) LDS holds: everything the top block can take is sorted
constexpr int SEL_HOLD = 16;   // candidates per thread kept in registers between the two passes (16384: a 752 x 480 frame has ~13000)
constexpr size_t SEL_LDS_BYTES = sizeof(HarrisCand) * (size_t)SEL_SORT;
struct SelectHeader {
    int n_candidates;   // all NMS survivors
    int n_top;          // candidates at or above the boundary bin (may exceed the capacity of the top block)
    int boundary_bin;
    int sorted;         // the top block is in visiting order (response desc, index desc)
    int seq;            // written last
};

__device__ __forceinline__ void d_harris_select(const HarrisCand *__restrict__ cand, int *__restrict__ count,
                                                int capacity, int *__restrict__ max_key, double quality,
                                                HarrisCand *top_out, int top_cap, SelectHeader *hdr, int seq, int keep,
                                                unsigned *__restrict__ sel_hist, HarrisCand *__restrict__ sel) {
    // keep: how many of the strongest candidates the host's spacing pass is handed at least (round 6: scaled with the number of corners
    // asked for -- with 896 for 300 / 600 corners the pass ran out on every frame and fell back to copying and heaping ALL ~10^4
    // candidates on the host, 0.3-0.4 ms per frame of the S2 / S3 streams)
    __shared__ __attribute__((aligned(16))) unsigned hist[SEL_BINS];   // each bin's next free slot
    __shared__ unsigned wsum[16];
    __shared__ int s_bin, s_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef XRHIP_KPROF
    long long kp[8];
    int kpn = 0;
#define SELPROF() do { __syncthreads(); if (tid == 0) kp[kpn] = wall_clock64(); ++kpn; } while (0)
#else
#define SELPROF() do { } while (0)
#endif
    SELPROF();
    // this thread's candidates: all loads in flight at once (with one, every trip waited out a full L2 round trip), and issued
    // beside the reads of the counter, the maximum and the histogram, not behind them: what lies beyond the count is ignored below
    HarrisCand held[SEL_HOLD];
#pragma unroll
    for (int u = 0; u < SEL_HOLD; ++u) held[u] = cand[min(tid + u * 1024, capacity - 1)];
    // the histogram k_harris_nms counted (four bins per thread), cleared for the context's next pass
    const uint4 h4 = *reinterpret_cast<const uint4 *>(&sel_hist[4 * tid]);
    *reinterpret_cast<uint4 *>(&sel_hist[4 * tid]) = make_uint4(0u, 0u, 0u, 0u);
    const int n_raw = *count;
    const int nc = min(n_raw, capacity);
    const unsigned thr_bits = sel_thr_bits(float_from_key(*max_key), quality);
    auto bin_of = [&](float v) __attribute__((always_inline)) -> int { return sel_bin(v, thr_bits); };
    auto before = [](const HarrisCand &a, const HarrisCand &b) __attribute__((always_inline)) -> bool {   // a is visited before b
        return (a.v > b.v) || (a.v == b.v && a.idx > b.idx);
    };
    const int last = max(nc - 1, 0);
    SELPROF();   // 1: (the histogram pass of rounds 3-5: now only the wait for the loads above)
    // S(b) = candidates in bins >= b, by a suffix scan over the workgroup; boundary = the largest bin with S(b) >= keep (0 when
    // there are fewer candidates: keep all); each bin's first slot = the candidates in the bins above it
    const unsigned own = (h4.x + h4.y) + (h4.z + h4.w);
    unsigned incl = own;   // over the lanes >= this one
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_down(incl, off);
        if (lane + off < 64) incl += o;
    }
    if (lane == 0) wsum[wave] = incl;
    if (tid == 0) {
        s_bin = 0;
        s_n = nc;
    }
    __syncthreads();
    unsigned higher = ((lane & 15) > wave) ? wsum[lane & 15] : 0u;   // the wavefronts above this one: 16 lanes, summed in four steps
    higher += __shfl_xor(higher, 1);
    higher += __shfl_xor(higher, 2);
    higher += __shfl_xor(higher, 4);
    higher += __shfl_xor(higher, 8);
    const unsigned S4 = higher + incl - own;   // candidates in the bins above this thread's four
    const unsigned S3 = S4 + h4.w, S2 = S3 + h4.z, S1 = S2 + h4.y, S0 = S1 + h4.x;
    {
        // S is monotone: exactly one bin has S(b) >= keep > S(b + 1) -- its owner alone writes (every thread below the boundary
        // raising a shared maximum was a thousand atomics on one word)
        const unsigned k = (unsigned)keep;
        const int mine = (S3 >= k && S4 < k) ? 3 : (S2 >= k && S3 < k) ? 2 : (S1 >= k && S2 < k) ? 1 : (S0 >= k && S1 < k) ? 0 : -1;
        if (mine >= 0 && 4 * tid + mine > 0) {
            s_bin = 4 * tid + mine;
            s_n = (int)(mine == 0 ? S0 : mine == 1 ? S1 : mine == 2 ? S2 : S3);
        }
    }
    *reinterpret_cast<uint4 *>(&hist[4 * tid]) = make_uint4(S1, S2, S3, S4);
    __syncthreads();
    const int bin = s_bin;
    const int n_top = s_n;
    SELPROF();   // 2: boundary
    int sorted = 0;
    if (n_top <= min(SEL_SORT, top_cap)) {
        // counting sort by bin: a slot from the bin's range per candidate (arbitrary order inside a bin) ...
#pragma unroll
        for (int u = 0; u < SEL_HOLD; ++u) {
            const int b = bin_of(held[u].v);
            if (tid + u * 1024 < nc && b >= bin) sel[atomicAdd(&hist[b], 1u)] = held[u];
        }
        for (int i0 = SEL_HOLD * 1024 + tid; i0 < nc; i0 += 8 * 1024) {
            HarrisCand c8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) c8[u] = cand[min(i0 + u * 1024, last)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = bin_of(c8[u].v);
                if (i0 + u * 1024 < nc && b >= bin) sel[atomicAdd(&hist[b], 1u)] = c8[u];
            }
        }
        __syncthreads();
        SELPROF();   // 3: selection
        // ... then the visiting order (response descending, then index descending: a strict total order, indices are unique) inside
        // each bin: bin b now occupies [hist[b + 1], hist[b]) -- every bin's counter has run up to the start of the bin below it
        // (written straight to its final slot of the top block: 8-byte stores, scattered only inside a bin's range)
        for (int i = tid; i < n_top; i += 1024) {
            const HarrisCand cd = sel[i];
            const int b = bin_of(cd.v);
            const int s0 = (b == SEL_BINS - 1) ? 0 : (int)hist[b + 1], s1 = (int)hist[b];
            int ahead = 0;
            for (int j = s0; j < s1; ++j) ahead += before(sel[j], cd) ? 1 : 0;
            host_store(reinterpret_cast<unsigned long long *>(top_out + s0 + ahead),
                       ((unsigned long long)(unsigned)cd.idx << 32) | (unsigned long long)__float_as_uint(cd.v));
        }
        sorted = 1;
    } else {
        SELPROF();   // (more candidates at the boundary than the top block takes: the host fetches the full list)
    }
    if (tid == 0) {
        host_store(&hdr->n_candidates, n_raw);
        host_store(&hdr->n_top, n_top);
        host_store(&hdr->boundary_bin, bin);
        host_store(&hdr->sorted, sorted);
        // this kernel is the last reader of the pass's running maximum and candidate counter: it leaves both reset for the context's
        // next detection (a host-side reset in front of k_harris was a copy command of its own per frame and per sequence)
        *max_key = (int)0x80000000;
        *count = 0;
    }
    SELPROF();   // 4: order + copy out
    host_stores_wait();
    __syncthreads();
    SELPROF();   // 5: fence
#ifdef XRHIP_KPROF
    if (tid == 0 && (seq & 63) == 0)
        printf("k_harris_select nc %d n_top %d: hist %lld boundary %lld select %lld order+out %lld fence %lld (x10 ns)\n", nc, n_top,
               kp[1] - kp[0], kp[2] - kp[1], kp[3] - kp[2], kp[4] - kp[3], kp[5] - kp[4]);
#endif
    if (tid == 0) host_store(&hdr->seq, seq);   // header and top block are in host memory
    (void)lane;
}
struct HarrisSelectArgs {
    const HarrisCand *cand;
    int *count;
    int capacity;
    int *max_key;
    double quality;
    HarrisCand *top_out;
    int top_cap;
    SelectHeader *hdr;
    int seq;
    int keep;
    unsigned *sel_hist;
};
__global__ __launch_bounds__(1024) void k_harris_select(Batch<HarrisSelectArgs> b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sel_lds[];   // SEL_LDS_BYTES
    const HarrisSelectArgs &a = b.e[blockIdx.z];
    d_harris_select(a.cand, a.count, a.capacity, a.max_key, a.quality, a.top_out, a.top_cap, a.hdr, a.seq, a.keep, a.sel_hist,
                    reinterpret_cast<HarrisCand *>(sel_lds));
}

// ----------------------------------------------------------------------- LK
// ------------------------------------------------------------------------------------------- undistortion
// cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) with the 1/32-pixel fixed-point map of cv::undistort /
// xrslam::extra::ImageUndistorter (host/undistort_map.hpp builds it once per camera): what the reference's dataset
// reader does to every frame on the host (xrslam-pc/player/src/IO/euroc_dataset_reader.cpp:62-69,
// tum_dataset_reader.cpp:67-76).  Integer arithmetic only -- 15-bit weights (32 - ay)(32 - ax) * 32 ..., rounding
// (acc + 2^14) >> 15 -- hence bit-exact against oracle/undistort.py.  One pixel per thread, four per lane row-wise would
// buy nothing: the launch is bound by its ~5 us of dispatch, the 13 bytes per pixel (8 map, ~4 source, 1 out) stream.
__global__ __launch_bounds__(256) void k_undistort(const uint8_t *__restrict__ src, int sstride, const uint2 *__restrict__ map,
                                                   uint8_t *__restrict__ dst, int dstride, int w, int h) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const uint2 m = map[(size_t)y * w + x];
    const int sx = (int)(short)(m.x & 0xffff), sy = (int)(short)(m.x >> 16), ax = (int)(m.y & 31), ay = (int)((m.y >> 8) & 31);
    const bool x0 = sx >= 0 && sx < w, x1 = sx + 1 >= 0 && sx + 1 < w, y0 = sy >= 0 && sy < h, y1 = sy + 1 >= 0 && sy + 1 < h;
    const uint8_t *r0 = src + (ptrdiff_t)min(max(sy, 0), h - 1) * sstride, *r1 = src + (ptrdiff_t)min(max(sy + 1, 0), h - 1) * sstride;
    const int cx0 = min(max(sx, 0), w - 1), cx1 = min(max(sx + 1, 0), w - 1);
    const int p00 = (x0 && y0) ? r0[cx0] : 0, p01 = (x1 && y0) ? r0[cx1] : 0, p10 = (x0 && y1) ? r1[cx0] : 0, p11 = (x1 && y1) ? r1[cx1] : 0;
    const int w00 = (32 - ay) * (32 - ax) * 32, w01 = (32 - ay) * ax * 32, w10 = ay * (32 - ax) * 32, w11 = ay * ax * 32;
    const int v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
    dst[(size_t)y * dstride + x] = (uint8_t)min(255, max(0, v));
}

// LK_WAVES wavefronts share one keypoint: the 441 pixels of the window are dealt to LK_THREADS lanes (pixel p = tid +
// LK_THREADS * slot) and every window sum is reduced inside the wavefront (DPP scans) and then across the wavefronts through
// a few bytes of LDS and ONE barrier.  The sums are exact integers, so the result does not depend on how the pixels are
// dealt: the same bits as one wavefront per point -- which is what the kernel was until the in-kernel cycle counters
// showed it bound by that one wavefront's instruction stream (40 % iterations, 25 % templates; profiles/r02_ab_variants.md).
#ifndef XRHIP_LK_WAVES
#define XRHIP_LK_WAVES 4
#endif
constexpr int LK_WAVES = XRHIP_LK_WAVES;
constexpr int LK_THREADS = 64 * LK_WAVES;
constexpr int LK_SLOTS = (21 * 21 + LK_THREADS - 1) / LK_THREADS;
constexpr int LK_W_BITS = 14;
// Search neighbourhood of the second image staged in LDS once per pyramid level: the 22x22 footprint of the window
// (21 + 1 for the bilinear taps) plus LK_TILE_R pixels on every side of the level's starting position, the left edge
// moved down to a 4-byte boundary (so rows are fetched as whole dwords, coalesced).  An iteration whose footprint
// leaves the tile reads global memory instead -- same bytes, same arithmetic.
constexpr bool LK_STAGE_J = true;
constexpr int LK_TILE_R = 5;
constexpr int LK_TILE_W = 36;               // 22 + 2 * LK_TILE_R + up to 3 pixels of alignment slack, in dwords: 9
constexpr int LK_TILE_W4 = LK_TILE_W / 4;
constexpr int LK_TILE_H = 22 + 2 * LK_TILE_R;
constexpr int LK_TILE_DWORDS = LK_TILE_H * LK_TILE_W4;
constexpr int LK_TILE_LOADS = (LK_TILE_DWORDS + LK_THREADS - 1) / LK_THREADS;   // dwords per lane

__device__ __forceinline__ int lk_descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
constexpr int lk_ceil_div(int a, int b) { return (a + b - 1) / b; }

// Exact wavefront-wide sum of a 64-bit integer, result in every lane.  DPP data movement (row_shr 1/2/4/8 scan
// inside each row of 16 lanes, row_bcast:15 / row_bcast:31 across the rows, v_readlane of lane 63) instead of six
// ds_bpermute round trips: the reductions sit on the serial path of every LK iteration.
// (Tried and dropped: staging a 34x34 search region of the target level in LDS per level -- the iteration's taps are
// L1 hits already, the staging pass cost more than it saved: 74 -> 102 us per launch.)
__device__ __forceinline__ long long dpp_shifted_i64(long long v, const int step) {
    int lo = (int)(v & 0xffffffffLL), hi = (int)(v >> 32);
    switch (step) {   // dpp_ctrl, row_mask and bank_mask must be compile-time constants
    case 0: lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, false); break;   // row_shr:1
    case 1: lo = __builtin_amdgcn_update_dpp(0, lo, 0x112, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x112, 0xf, 0xf, false); break;   // row_shr:2
    case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xf, 0xf, false); break;   // row_shr:4
    case 3: lo = __builtin_amdgcn_update_dpp(0, lo, 0x118, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x118, 0xf, 0xf, false); break;   // row_shr:8
    case 4: lo = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xa, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xa, 0xf, false); break;   // row_bcast:15 -> rows 1, 3
    default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xc, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xc, 0xf, false); break;  // row_bcast:31 -> rows 2, 3
    }
    return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {
    v += dpp_shifted_i64(v, 0);
    v += dpp_shifted_i64(v, 1);
    v += dpp_shifted_i64(v, 2);
    v += dpp_shifted_i64(v, 3);
    v += dpp_shifted_i64(v, 4);
    v += dpp_shifted_i64(v, 5);
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffLL), 63), hi = __builtin_amdgcn_readlane((int)(v >> 32), 63);
    return ((long long)hi << 32) | (unsigned int)lo;
}

// Bilinear taps: pixel (8 bit) or Scharr derivative (int16) times a 14-bit fixed-point weight (0 .. 2^14, the fourth
// one 2^14 minus the others, so -1 at worst).  Both factors fit 24 signed bits, the product is exact, and
// v_mul_i32_i24 issues at full rate where the general 32-bit v_mul_lo_u32 takes four passes.
// (Written as the instruction itself: the compiler rewrites __mul24 into a general multiply where it rescales the
// weights for the derivative taps.)
__device__ __forceinline__ int lk_tap(int sample, int weight) {
    int r;
    asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(sample), "v"(weight));
    return r;
}

// acc + sample * weight for the same operand ranges (v_mad_i32_i24: 24-bit signed factors, full 32-bit addend): one
// full-rate instruction per tap of the iteration loop, where mul + add3 took 1.5.
__device__ __forceinline__ int lk_tap_acc(int sample, int weight, int acc) {
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(sample), "v"(weight), "v"(acc));
    return r;
}

// Exact wavefront-wide sum of a 32-bit integer whose 64 terms may not fit 32 bits together (|v| < 2^31 per lane):
// the low 16 bits (unsigned) and the high 16 bits (signed) are summed separately -- 64 terms of at most 16 bits stay
// below 2^22 -- as 32-bit DPP scans (one v_add per stage instead of the add / add-with-carry pair and the two moves a
// 64-bit scan needs), and recombined in double precision, where every integer below 2^53 is exact.  Returns the sum
// as a double in every lane; (float) of it is the correctly rounded value, the same float (float)(int64 sum) is.
__device__ __forceinline__ int dpp_scan_add_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ double wave_sum_i32_exact(int v) {
    const int lo = dpp_scan_add_i32(v & 0xffff), hi = dpp_scan_add_i32(v >> 16);
    return (double)hi * 65536.0 + (double)lo;
}

// -DXRHIP_KPROF: shader-clock sums of the LK phases as lane 0 of a point's first wavefront sees them (xrhip_debug_lkprof):
//   0 level set-up (tile fetch issued, template gather, A sums)   1 A exchange, eigenvalue gate, tile -> LDS
//   2 iteration: weights + taps   3 iteration: the four DPP scans   4 iteration: exchange (store, barrier, loads, adds)
//   5 iteration: update + convergence tests   6 points   7 kernel entry -> exit per point
#ifdef XRHIP_KPROF
__device__ unsigned long long g_lk_prof[8];
#define LKPROF(slot)                                                 \
    do {                                                             \
        const long long lk_n_ = (long long)__builtin_readcyclecounter(); \
        lk_acc[slot] += lk_n_ - lk_t;                                \
        lk_t = lk_n_;                                                \
    } while (0)
#else
#define LKPROF(slot) \
    do {             \
    } while (0)
#endif
struct LkCounters {
    unsigned long long templates;
    unsigned long long iterations;
};

// One direction of cv::calcOpticalFlowPyrLK for ONE point, executed by the LK_WAVES wavefronts of a workgroup (all lanes carry
// identical scalar state; the window is lane-striped).  Returns the status bit; nx, ny are the OPTFLOW_USE_INITIAL_FLOW guess on
// entry and the tracked position on exit (level-0 pixels).
//
// Round 6: everything a level needs from global memory is requested for ALL levels before the first one starts.  In-kernel cycle
// counters (profiles/r06_lk_phases.md) had half of a point's 34 us in the set-up of its eight level visits -- per level the
// template gather and the search-tile fetch (one round trip to L2 / HBM each, ~2000 cycles), an exchange of the template's A sums
// (a barrier) and the tile's way into LDS (two more) -- against ~960 cycles per iteration.  The template of a level depends on the
// point alone; the search tile of a level is fetched around the INITIAL GUESS scaled to that level (rounds 3-5: around the position
// the coarser level ended at, known only then) -- an iteration whose footprint is not inside the tile reads global memory, as before,
// so where the tile sits changes nothing but speed.  One round trip, one exchange and one barrier per direction.  The arithmetic of
// every level is what it was: status bits and positions are bit-identical (tests/test_klt_gpu.py).
__device__ __forceinline__ int lk_one_way(const PyrView &A, const PyrView &B, float px0, float py0, float &nx_io,
                                          float &ny_io, const int (&wx)[LK_SLOTS], const int (&wy)[LK_SLOTS],
                                          const bool (&wvalid)[LK_SLOTS], unsigned &n_templates,
                                          unsigned &n_iters, uint32_t (*tile)[LK_TILE_DWORDS], double (*xch)[LK_WAVES][3 * KLT_LEVELS], int &xpar,
                                          long long (&lk_acc)[8], long long &lk_t) {
    // xch: [2][LK_WAVES][12] doubles of LDS through which the wavefronts of this point exchange their partial window sums
    // (every value an exact integer below 2^53, so their sum is exact in any order); xpar alternates between the two
    // halves, which is what lets one barrier per exchange suffice (a wavefront can only be writing half h for exchange
    // n + 1 after every wavefront passed the barrier of exchange n, i.e. after all reads of half h for exchange n - 1)
    const int wave_id = (int)threadIdx.x >> 6;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float half = (KLT_WIN - 1) * 0.5f;
    const double epsilon = 0.01 * 0.01;
    int status = 1;
    float outx = nx_io, outy = ny_io;
    // ---- (1) requests of all levels: search tiles around the scaled guess, then the template samples
    uint32_t tile_regs[KLT_LEVELS][LK_TILE_LOADS];
    int tx0[KLT_LEVELS], ty0[KLT_LEVELS];
    bool staged[KLT_LEVELS], tmpl[KLT_LEVELS];
    int tw[KLT_LEVELS][4];                                   // the template's bilinear weights
    // per pixel two 16-bit loads (the two 8-bit samples of a row are neighbours) and two 8-byte loads (two derivative pairs): half the
    // load instructions of one per tap -- the gather of 64 lanes x 16 loads per level was bound by the texture-address path
    typedef uint16_t __attribute__((aligned(1))) lk_u16u;
    typedef uint2 __attribute__((aligned(4))) lk_u2u;
    uint32_t raw_i[KLT_LEVELS][LK_SLOTS][2];                 // two 8-bit samples each: rows y, y + 1
    uint2 raw_d[KLT_LEVELS][LK_SLOTS][2];                    // two derivative pairs each: rows y, y + 1
#pragma unroll
    for (int level = 0; level < KLT_LEVELS; ++level) {
        const LevelView J = B.lv[level];
        const float lscale = (float)(1. / (1 << level));
        staged[level] = false;
        tx0[level] = ty0[level] = 0;
        if (LK_STAGE_J) {
            const float gx = nx_io * lscale, gy = ny_io * lscale;
            const int inx0 = (int)floorf(gx - half), iny0 = (int)floorf(gy - half);
            if (inx0 >= -KLT_WIN && inx0 < J.w && iny0 >= -KLT_WIN && iny0 < J.h) {
                int x0 = inx0 - LK_TILE_R;
                const int y0 = iny0 - LK_TILE_R;
                x0 -= (int)(reinterpret_cast<uintptr_t>(J.img + x0) & 3);   // row strides are multiples of 64 bytes
#pragma unroll
                for (int k = 0; k < LK_TILE_LOADS; ++k) {
                    const int idx = min((int)threadIdx.x + LK_THREADS * k, LK_TILE_DWORDS - 1);
                    const int r = idx / LK_TILE_W4, c4 = idx - r * LK_TILE_W4;
                    // rows / dwords outside the padded plane are never part of a valid footprint: clamp the address
                    const int y = min(max(y0 + r, -KLT_PAD), J.h + KLT_PAD - 1);
                    const int x = min(max(x0 + 4 * c4, -KLT_PADX), J.istride - KLT_PADX - 4);
                    tile_regs[level][k] = *reinterpret_cast<const uint32_t *>(J.img + (ptrdiff_t)y * J.istride + x);
                }
                tx0[level] = x0;
                ty0[level] = y0;
                staged[level] = true;
            }
        }
    }
#pragma unroll
    for (int level = 0; level < KLT_LEVELS; ++level) {
        const LevelView I = A.lv[level];
        const float lscale = (float)(1. / (1 << level));
        const float px = px0 * lscale - half, py = py0 * lscale - half;
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        tmpl[level] = !(ipx < -KLT_WIN || ipx >= I.w || ipy < -KLT_WIN || ipy >= I.h);
        const float a = px - ipx, b = py - ipy;
        tw[level][0] = __float2int_rn((1.f - a) * (1.f - b) * (1 << LK_W_BITS));
        tw[level][1] = __float2int_rn(a * (1.f - b) * (1 << LK_W_BITS));
        tw[level][2] = __float2int_rn((1.f - a) * b * (1 << LK_W_BITS));
        tw[level][3] = (1 << LK_W_BITS) - tw[level][0] - tw[level][1] - tw[level][2];
        // a template outside the plane is never used; its loads go to the plane's origin instead
        const int cx = tmpl[level] ? ipx : 0, cy = tmpl[level] ? ipy : 0;
#pragma unroll
        for (int sl = 0; sl < LK_SLOTS; ++sl) {
            const uint8_t *s0 = I.img + (ptrdiff_t)(cy + wy[sl]) * I.istride + (cx + wx[sl]);
            const uint8_t *s1 = s0 + I.istride;
            const short2 *d0 = I.der + (ptrdiff_t)(cy + wy[sl]) * I.pstride + (cx + wx[sl]);
            const short2 *d1 = d0 + I.pstride;
            raw_i[level][sl][0] = *reinterpret_cast<const lk_u16u *>(s0);
            raw_i[level][sl][1] = *reinterpret_cast<const lk_u16u *>(s1);
            raw_d[level][sl][0] = *reinterpret_cast<const lk_u2u *>(d0);
            raw_d[level][sl][1] = *reinterpret_cast<const lk_u2u *>(d1);
        }
    }
    // ---- (2) the search tiles go to LDS (every wavefront is past the last tile read of the previous direction: its last
    // iteration's exchange barrier came after the taps), the templates are formed, their A sums reduced inside the wavefront
#pragma unroll
    for (int level = 0; level < KLT_LEVELS; ++level)
        if (staged[level])
#pragma unroll
            for (int k = 0; k < LK_TILE_LOADS; ++k) {
                const int idx = (int)threadIdx.x + LK_THREADS * k;
                if (idx < LK_TILE_DWORDS) tile[level][idx] = tile_regs[level][k];
            }
    int Iv[KLT_LEVELS][LK_SLOTS], Ix[KLT_LEVELS][LK_SLOTS], Iy[KLT_LEVELS][LK_SLOTS];
    double part[KLT_LEVELS][3];
#pragma unroll
    for (int level = 0; level < KLT_LEVELS; ++level) {
        const int iw00 = tw[level][0], iw01 = tw[level][1], iw10 = tw[level][2], iw11 = tw[level][3];
        long long sA11 = 0, sA12 = 0, sA22 = 0;
#pragma unroll
        for (int sl = 0; sl < LK_SLOTS; ++sl) {
            const uint32_t r0 = raw_i[level][sl][0], r1 = raw_i[level][sl][1];
            const uint2 e0 = raw_d[level][sl][0], e1 = raw_d[level][sl][1];   // .x: (Ix, Iy) of the left pixel, .y: of the right one
            int ival = lk_descale(lk_tap((int)(r0 & 255u), iw00) + lk_tap((int)(r0 >> 8), iw01) + lk_tap((int)(r1 & 255u), iw10) + lk_tap((int)(r1 >> 8), iw11),
                                  LK_W_BITS - 5);
            int ixval = lk_descale(lk_tap((int)(short)(e0.x & 0xffffu), iw00) + lk_tap((int)(short)(e0.y & 0xffffu), iw01) +
                                       lk_tap((int)(short)(e1.x & 0xffffu), iw10) + lk_tap((int)(short)(e1.y & 0xffffu), iw11),
                                   LK_W_BITS);
            int iyval = lk_descale(lk_tap((int)e0.x >> 16, iw00) + lk_tap((int)e0.y >> 16, iw01) + lk_tap((int)e1.x >> 16, iw10) + lk_tap((int)e1.y >> 16, iw11),
                                   LK_W_BITS);
            // OpenCV stores these as int16 (Iptr/dIptr are short); a lane without a pixel in this slot carries zeros
            ival = wvalid[sl] ? (int)(short)ival : 0;
            ixval = wvalid[sl] ? (int)(short)ixval : 0;
            iyval = wvalid[sl] ? (int)(short)iyval : 0;
            Iv[level][sl] = ival;
            Ix[level][sl] = ixval;
            Iy[level][sl] = iyval;
            sA11 += (long long)(ixval * ixval);
            sA12 += (long long)(ixval * iyval);
            sA22 += (long long)(iyval * iyval);
        }
        if (LK_SLOTS <= 2) {
            // |Ix|, |Iy| <= 4080: two slots stay below 2^25 per lane and 64 lanes below 2^31 -- the wavefront's sums are exact in 32
            // bits (one fused DPP add per stage; the 64-bit scan is an add / add-with-carry pair and two moves)
            part[level][0] = (double)dpp_scan_add_i32((int)sA11);
            part[level][1] = (double)dpp_scan_add_i32((int)sA12);
            part[level][2] = (double)dpp_scan_add_i32((int)sA22);
        } else {
            part[level][0] = (double)wave_sum_i64(sA11);   // < 2^37: exact
            part[level][1] = (double)wave_sum_i64(sA12);
            part[level][2] = (double)wave_sum_i64(sA22);
        }
    }
    LKPROF(0);
    // ---- (3) ONE exchange for the A sums of all levels; its barrier also publishes the tiles
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int level = 0; level < KLT_LEVELS; ++level)
#pragma unroll
            for (int k = 0; k < 3; ++k) xch[xpar][wave_id][3 * level + k] = part[level][k];
    }
    __syncthreads();
    float A11[KLT_LEVELS], A12[KLT_LEVELS], A22[KLT_LEVELS], Dinv[KLT_LEVELS];
    bool gate[KLT_LEVELS];
#pragma unroll
    for (int level = 0; level < KLT_LEVELS; ++level) {
        double dA11 = 0.0, dA12 = 0.0, dA22 = 0.0;
#pragma unroll
        for (int w = 0; w < LK_WAVES; ++w) {
            dA11 += xch[xpar][w][3 * level];
            dA12 += xch[xpar][w][3 * level + 1];
            dA22 += xch[xpar][w][3 * level + 2];
        }
        // (float) of the exact sum held as a double is the correctly rounded value, the same float (float)(int64 sum) is
        A11[level] = (float)dA11 * FLT_SCALE;
        A12[level] = (float)dA12 * FLT_SCALE;
        A22[level] = (float)dA22 * FLT_SCALE;
        const float D = A11[level] * A22[level] - A12[level] * A12[level];
        const float minEig = (A22[level] + A11[level] - sqrtf((A11[level] - A22[level]) * (A11[level] - A22[level]) + 4.f * A12[level] * A12[level])) /
                             (float)(2 * KLT_WIN * KLT_WIN);
        gate[level] = !(minEig < 1e-4f || D < 1.1920929e-07f);
        Dinv[level] = 1.f / D;
    }
    xpar ^= 1;
    LKPROF(1);
    // ---- (4) the levels, coarse to fine
#pragma unroll
    for (int level = KLT_LEVELS - 1; level >= 0; --level) {
        const LevelView J = B.lv[level];
        const float lscale = (float)(1. / (1 << level));
        float nx, ny;
        if (level == KLT_LEVELS - 1) {
            nx = outx * lscale;
            ny = outy * lscale;
        } else {
            nx = outx * 2.f;
            ny = outy * 2.f;
        }
        outx = nx;
        outy = ny;
        if (!tmpl[level]) {
            if (level == 0) status = 0;
            continue;
        }
        n_templates++;
        if (!gate[level]) {
            if (level == 0) status = 0;
            continue;
        }
        const float D = Dinv[level];
        nx -= half;
        ny -= half;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < 30; ++j) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -KLT_WIN || inx >= J.w || iny < -KLT_WIN || iny >= J.h) {
                if (level == 0) status = 0;
                break;
            }
            const float a = nx - inx, b = ny - iny;
            const int iw00 = __float2int_rn((1.f - a) * (1.f - b) * (1 << LK_W_BITS));
            const int iw01 = __float2int_rn(a * (1.f - b) * (1 << LK_W_BITS));
            const int iw10 = __float2int_rn((1.f - a) * b * (1 << LK_W_BITS));
            const int iw11 = (1 << LK_W_BITS) - iw00 - iw01 - iw10;
            // per-lane partial sums stay in 32 bits: |diff| <= 8160 (8-bit pixels with 5 fractional bits), |Ix|, |Iy| <= 4080
            // (Scharr 3/10/3 of 8-bit pixels), seven slots: < 2.4e8
            int sb1 = 0, sb2 = 0;
            const int rx = inx - tx0[level], ry = iny - ty0[level];
            if (staged[level] && rx >= 0 && rx <= LK_TILE_W - 22 && ry >= 0 && ry <= LK_TILE_H - 22) {
                // LDS-qualified pointer: keeps these reads ds_read (the compiler otherwise sinks the last slot of this
                // branch and of the global-memory branch below into one block of flat loads -- a second round trip)
                typedef const __attribute__((address_space(3))) uint8_t *lds_bytes;
                const lds_bytes t = (lds_bytes)tile[level] + ry * LK_TILE_W + rx;
                // (a lane without a pixel in slot s holds Ix = Iy = 0 there: its products vanish, no branch needed)
#pragma unroll
                for (int sl = 0; sl < LK_SLOTS; ++sl) {
                    const lds_bytes j0 = t + wy[sl] * LK_TILE_W + wx[sl];
                    const lds_bytes j1 = j0 + LK_TILE_W;
                    int acc = lk_tap_acc(j0[0], iw00, 1 << (LK_W_BITS - 5 - 1));   // the rounding term of lk_descale rides along
                    acc = lk_tap_acc(j0[1], iw01, acc);
                    acc = lk_tap_acc(j1[0], iw10, acc);
                    acc = lk_tap_acc(j1[1], iw11, acc);
                    const int diff = (acc >> (LK_W_BITS - 5)) - Iv[level][sl];
                    sb1 = lk_tap_acc(diff, Ix[level][sl], sb1);
                    sb2 = lk_tap_acc(diff, Iy[level][sl], sb2);
                }
            } else {
#pragma unroll
                for (int sl = 0; sl < LK_SLOTS; ++sl) {
                    const uint8_t *j0 = J.img + (ptrdiff_t)(iny + wy[sl]) * J.istride + (inx + wx[sl]);
                    const uint8_t *j1 = j0 + J.istride;
                    int acc = lk_tap_acc(j0[0], iw00, 1 << (LK_W_BITS - 5 - 1));
                    acc = lk_tap_acc(j0[1], iw01, acc);
                    acc = lk_tap_acc(j1[0], iw10, acc);
                    acc = lk_tap_acc(j1[1], iw11, acc);
                    const int diff = (acc >> (LK_W_BITS - 5)) - Iv[level][sl];
                    sb1 = lk_tap_acc(diff, Ix[level][sl], sb1);
                    sb2 = lk_tap_acc(diff, Iy[level][sl], sb2);
                }
            }
            n_iters++;
            LKPROF(2);
            // The wavefront's sums of the low (unsigned) and high (signed) 16-bit halves, each below 2^22; the wavefronts exchange
            // THESE integers (one 16-byte store, four 16-byte loads, integer adds) and the halves are recombined in double precision
            // once.  Every partial sum is an exact integer: the same double, the same float as a 64-bit sum.
            int h4[4] = {dpp_scan_add_i32(sb1 & 0xffff), dpp_scan_add_i32(sb1 >> 16), dpp_scan_add_i32(sb2 & 0xffff), dpp_scan_add_i32(sb2 >> 16)};
            LKPROF(3);
            if (LK_WAVES > 1) {
                int4 *slots = reinterpret_cast<int4 *>(&xch[xpar][0][0]);   // one 96-byte slot per wavefront: its first 16 bytes
                if ((threadIdx.x & 63) == 0) slots[6 * wave_id] = make_int4(h4[0], h4[1], h4[2], h4[3]);
                __syncthreads();
                h4[0] = h4[1] = h4[2] = h4[3] = 0;
#pragma unroll
                for (int w = 0; w < LK_WAVES; ++w) {
                    const int4 v = slots[6 * w];
                    h4[0] += v.x;
                    h4[1] += v.y;
                    h4[2] += v.z;
                    h4[3] += v.w;
                }
                xpar ^= 1;
            }
            LKPROF(4);
            const double db1 = (double)h4[1] * 65536.0 + (double)h4[0], db2 = (double)h4[3] * 65536.0 + (double)h4[2];
            const float b1 = (float)db1 * FLT_SCALE;
            const float b2 = (float)db2 * FLT_SCALE;
            const float dx = (A12[level] * b2 - A22[level] * b1) * D;
            const float dy = (A12[level] * b1 - A11[level] * b2) * D;
            nx += dx;
            ny += dy;
            outx = nx + half;
            outy = ny + half;
            if ((double)dx * (double)dx + (double)dy * (double)dy <= epsilon) break;
            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                outx -= dx * 0.5f;
                outy -= dy * 0.5f;
                break;
            }
            pdx = dx;
            pdy = dy;
            LKPROF(5);
        }
    }
    nx_io = outx;
    ny_io = outy;
    return status;
}

__device__ __forceinline__ void lk_lane_layout(int lane, int (&wx)[LK_SLOTS], int (&wy)[LK_SLOTS],
                                               bool (&wvalid)[LK_SLOTS]) {
#pragma unroll
    for (int s = 0; s < LK_SLOTS; ++s) {
        int p = lane + LK_THREADS * s;
        wvalid[s] = p < KLT_WIN * KLT_WIN;
        if (!wvalid[s]) p = 0;
        wy[s] = p / KLT_WIN;
        wx[s] = p - wy[s] * KLT_WIN;
    }
}

// OpenCvImage::track_keypoints: forward LK, border/displacement gates, backward
// LK, 0.5 px round-trip check -- fused, one wavefront per keypoint.
// curr / next_io / status_out may live in pinned host memory (zero-copy): the points of a frame are a few KB, so
// the kernel reads and writes them over the host link itself.  done / done_target / host_seq implement the
// completion mailbox: every wavefront publishes its result (system-scope fence), bumps `done`, and the one that
// reaches done_target stores `seq` where the host is spinning.
// inl: the point (and guess), already converted to float by the host with the same rounding, delivered in the kernel-argument block
// (k_lk_track_inl) -- a workgroup's first instruction then no longer waits for a read of the pinned point list over the host link.
__device__ __forceinline__ void d_lk_track(const PyrView &A, const PyrView &B, const double2 *__restrict__ curr,
                                           double2 *__restrict__ next_io, int has_guess,
                                           uint8_t *__restrict__ status_out, int n,
                                           LkCounters *__restrict__ counters, unsigned *done, unsigned done_target,
                                           int *host_seq, int seq, const float4 *inl = nullptr) {
    const int pt = blockIdx.x;
    if (pt >= n) return;
    const int lane = threadIdx.x;
    int wx[LK_SLOTS], wy[LK_SLOTS];
    bool wvalid[LK_SLOTS];
    lk_lane_layout(lane, wx, wy, wvalid);
    float cx, cy, nx, ny;
    if (inl) {
        const float4 v = inl[pt];   // (current x, y, guess x, y): the guess equals the current position when there is none
        cx = v.x;
        cy = v.y;
        nx = v.z;
        ny = v.w;
    } else {
        const double2 c = curr[pt];
        cx = (float)c.x;
        cy = (float)c.y;   // to_opencv(): double -> float
        nx = cx;
        ny = cy;
        if (has_guess) {
            const double2 g = next_io[pt];
            nx = (float)g.x;
            ny = (float)g.y;
        }
    }
    unsigned n_templates = 0, n_iters = 0;
    __shared__ uint32_t tile[KLT_LEVELS][LK_TILE_DWORDS];
    __shared__ double xch[2][LK_WAVES][3 * KLT_LEVELS];
    int xpar = 0;
    long long lk_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long lk_t = (long long)__builtin_readcyclecounter();
    const long long lk_t0 = lk_t;
    int status = lk_one_way(A, B, cx, cy, nx, ny, wx, wy, wvalid, n_templates, n_iters, tile, xch, xpar, lk_acc, lk_t);
    const int cols = A.lv[0].w, rows = A.lv[0].h;
    if (nx < 20 || nx >= cols - 20 || ny < 20 || ny >= rows - 20) status = 0;
    if (status) {
        const float dx = nx - cx, dy = ny - cy;
        const double nrm = sqrt((double)dx * (double)dx + (double)dy * (double)dy);
        if (nrm > rows / 4) status = 0;
    }
    if (status) {
        float rx = cx, ry = cy;
        int st2 = lk_one_way(B, A, nx, ny, rx, ry, wx, wy, wvalid, n_templates, n_iters, tile, xch, xpar, lk_acc, lk_t);
        const float dx = cx - rx, dy = cy - ry;
        const double nrm = sqrt((double)dx * (double)dx + (double)dy * (double)dy);
        if (!st2 || nrm > 0.5) status = 0;
    }
#ifdef XRHIP_KPROF
    if (lane == 0) {
        for (int i = 0; i < 6; ++i) atomicAdd(&g_lk_prof[i], (unsigned long long)lk_acc[i]);
        atomicAdd(&g_lk_prof[6], 1ull);
        atomicAdd(&g_lk_prof[7], (unsigned long long)((long long)__builtin_readcyclecounter() - lk_t0));
    }
#else
    (void)lk_t0;
#endif
    if (lane == 0) {
        // the point's results go to the (pinned) point block; the point counts itself done once they are released, and the last one
        // to do so raises the sequence number
        host_store(&status_out[pt], (uint8_t)status);
        if (status) {
            host_store(&next_io[pt].x, (double)nx);
            host_store(&next_io[pt].y, (double)ny);
        }
        if (counters) {
            atomicAdd(&counters->templates, (unsigned long long)n_templates);
            atomicAdd(&counters->iterations, (unsigned long long)n_iters);
        }
        if (done) {
            // (Round 6: the system-scope stores above would do without these fences -- wait for the acknowledgements, count -- as in
            // k_harris_select and kp_preintegrate (host_mailbox.hip.h).  Measured in alternating traces of one call
            // (profiles/r06_fence.md) it makes no difference here: the points' fences overlap the other points' work.  They stay.)
            __threadfence_system();
            if (atomicAdd(done, 1u) + 1u == done_target) {
                __threadfence_system();
                host_store(host_seq, seq);
            }
        }
    }
}
struct LkTrackArgs {
    PyrView A, B;
    const double2 *curr;
    double2 *next_io;
    int has_guess;
    uint8_t *status_out;
    int n;   // points of this entry (blocks beyond them return at once)
    LkCounters *counters;
    unsigned *done;
    unsigned done_target;
    int *host_seq;
    int seq;
};
__global__ __launch_bounds__(LK_THREADS) void k_lk_track(Batch<LkTrackArgs> b) {
    const LkTrackArgs &a = b.e[blockIdx.z];
    d_lk_track(a.A, a.B, a.curr, a.next_io, a.has_guess, a.status_out, a.n, a.counters, a.done, a.done_target, a.host_seq, a.seq);
}
// One entry (a sequence that launches for itself) with up to LK_INLINE points: the points travel in the argument block (2.5 KB of the 4 KB).
constexpr int LK_INLINE = 160;
struct LkInlinePoints {
    float4 p[LK_INLINE];
};
__global__ __launch_bounds__(LK_THREADS) void k_lk_track_inl(LkTrackArgs a, LkInlinePoints pts) {
    d_lk_track(a.A, a.B, a.curr, a.next_io, a.has_guess, a.status_out, a.n, a.counters, a.done, a.done_target, a.host_seq, a.seq, pts.p);
}

// A host (pinned, device-mapped) or device frame into the dense plane the preprocessing reads: the upload of a group's frames is one
// launch (a copy-engine transfer per frame is a command of its own on the queue; on this stack it runs as a blit kernel anyway).
struct UploadArgs {
    const uint8_t *src;
    int sstride;
    uint8_t *dst;   // dense w x h
    int w, h;
};
__global__ __launch_bounds__(256) void k_upload(Batch<UploadArgs> b) {
    const UploadArgs &a = b.e[blockIdx.z];
    if (!a.src) return;
    const size_t total = (size_t)a.w * a.h;
    if (a.sstride == a.w && (total & 15) == 0 && ((reinterpret_cast<uintptr_t>(a.src) | reinterpret_cast<uintptr_t>(a.dst)) & 15) == 0) {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(a.src);
        uint4 *d4 = reinterpret_cast<uint4 *>(a.dst);
        const size_t n16 = total >> 4;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) d4[i] = s4[i];
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
            const size_t y = i / a.w, x = i - y * a.w;
            a.dst[i] = a.src[y * (size_t)a.sstride + x];
        }
    }
}

// plain calcOpticalFlowPyrLK (float in/out, every point) -- parity aid
__global__ __launch_bounds__(LK_THREADS) void k_lk_plain(PyrView A, PyrView B, const float2 *__restrict__ prev,
                                                 float2 *__restrict__ next_io, uint8_t *__restrict__ status_out,
                                                 int n) {
    const int pt = blockIdx.x;
    if (pt >= n) return;
    const int lane = threadIdx.x;
    int wx[LK_SLOTS], wy[LK_SLOTS];
    bool wvalid[LK_SLOTS];
    lk_lane_layout(lane, wx, wy, wvalid);
    const float2 p = prev[pt];
    float2 q = next_io[pt];
    unsigned a = 0, b = 0;
    __shared__ uint32_t tile[KLT_LEVELS][LK_TILE_DWORDS];
    __shared__ double xch[2][LK_WAVES][3 * KLT_LEVELS];
    int xpar = 0;
    long long lk_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long lk_t = 0;
    int status = lk_one_way(A, B, p.x, p.y, q.x, q.y, wx, wy, wvalid, a, b, tile, xch, xpar, lk_acc, lk_t);
    if (lane == 0) {
        status_out[pt] = (uint8_t)status;
        next_io[pt] = q;
    }
}

}   // namespace xrhip
