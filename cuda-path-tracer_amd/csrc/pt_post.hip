/*
 * pt_post.hip -- post-processing on the device: the (2k+1)^2 convolution filter (NORMAL and
 * BILATERAL) as an LDS-tiled stencil, and the element-wise image operations that the reference
 * runs as CPU loops between two host<->device round trips (src/post_processing.cu:105-278,
 * src/renderer.cu:277-297).  The image stays in HBM for the whole chain.
 *
 * Filter kernel: a 256-thread workgroup owns a 32x8 output tile and stages the tile plus its
 * k-pixel halo into LDS as three planes (R, G, B) -- consecutive lanes read consecutive dwords, so
 * the tap loop is free of bank conflicts -- with out-of-image texels marked invalid so that the
 * "crop" rule (taps outside the image contribute neither colour nor weight) holds at the borders.
 * Tap weights are wave-uniform and are read through the scalar path.  No MFMA: a stencil with a
 * per-pixel normalisation (and, for BILATERAL, a data-dependent weight) is not a dense contraction.
 *
 * NORMAL filters whose tap table is an outer product a[y] * b[x] (the gaussian and mean tables of
 * src/post_processing.cu:51-84 are) run as two one-dimensional passes instead: the window cropped
 * to the image is a rectangle, so both the weighted sum and the weight total factor exactly, and
 * the 65x65 bloom costs 130 multiply-adds per pixel and channel instead of 4225.  Only the order of
 * the floating-point additions differs from the two-dimensional loop.
 */
#include <cmath>
#include <cstring>
#include <vector>

#include "pt_internal.h"

namespace ptamd {

static constexpr int kTileW = 32, kTileH = 8, kPostBlock = 256;

__device__ __forceinline__ float lum3(float r, float g, float b) { return r * 0.2126f + g * 0.7152f + b * 0.0722f; }

/* src/post_processing.cu:31-35 */
__device__ __forceinline__ float gaussian_1d(float x, float sigma) {
    const float fac = 1.0f / (sigma * sigma * 2 * 3.1415926535f);
    const float ex = -((x * x) / (2 * sigma * sigma));
    return fac * powf(2.7182818284f, ex);
}

/* device_apply_filter (src/post_processing.cu:143-187) */
template <bool BILATERAL>
__global__ void __launch_bounds__(kPostBlock) filter_kernel(const float *__restrict__ in, float *__restrict__ out, int w, int h,
                                                            const float *__restrict__ taps, int ks) {
    extern __shared__ float lds[];
    const int tw = kTileW + 2 * ks, th = kTileH + 2 * ks;
    const int plane = tw * th;
    float *R = lds, *G = lds + plane, *B = lds + 2 * plane;
    const int x0 = blockIdx.x * kTileW - ks, y0 = blockIdx.y * kTileH - ks;
    for (int i = threadIdx.x; i < plane; i += kPostBlock) {
        const int ly = i / tw, lx = i - ly * tw;
        const int gx = x0 + lx, gy = y0 + ly;
        float r = 0.f, g = 0.f, b = 0.f;
        if (gx >= 0 && gx < w && gy >= 0 && gy < h) {
            const float *p = in + 3 * ((size_t)gy * w + gx);
            r = p[0]; g = p[1]; b = p[2];
        }
        R[i] = r; G[i] = g; B[i] = b;
    }
    __syncthreads();
    const int lx = threadIdx.x % kTileW, ly = threadIdx.x / kTileW;
    const int u = blockIdx.x * kTileW + lx, v = blockIdx.y * kTileH + ly;
    if (u >= w || v >= h) return;
    const int kw = 2 * ks + 1;
    /* rows / columns of the tap window that fall inside the image */
    const int ky_lo = max(-ks, -v), ky_hi = min(ks, h - 1 - v);
    const int kx_lo = max(-ks, -u), kx_hi = min(ks, w - 1 - u);
    float pr = 0.f, pg = 0.f, pb = 0.f, wsum = 0.f;
    const int c = (ly + ks) * tw + (lx + ks);
    const float cr = R[c], cg = G[c], cb = B[c];
    for (int y = -ks; y <= ks; y++) {
        if (y < ky_lo || y > ky_hi) continue;
        const float *trow = taps + (size_t)(ks + y) * kw + ks;
        const int rowbase = c + y * tw;
        for (int x = -ks; x <= ks; x++) {
            if (x < kx_lo || x > kx_hi) continue;
            const float gs = trow[x];
            const float nr = R[rowbase + x], ng = G[rowbase + x], nb = B[rowbase + x];
            float wgt = gs;
            if (BILATERAL) wgt = gaussian_1d(255 * lum3(nr - cr, ng - cg, nb - cb), 24.0f) * gs;
            pr = pr + wgt * nr;
            pg = pg + wgt * ng;
            pb = pb + wgt * nb;
            wsum += wgt;
        }
    }
    float *o = out + 3 * ((size_t)v * w + u);
    o[0] = pr / wsum; o[1] = pg / wsum; o[2] = pb / wsum;
}

/* separable NORMAL filter, pass along x: out(x, y) = sum over in-image dx of b[dx] * in(x + dx, y), not normalised.
 * One workgroup = 256 consecutive pixels of one row, staged with their halo as three planes. */
__global__ void __launch_bounds__(kPostBlock) sep_row_kernel(const float *__restrict__ in, float *__restrict__ out, int w, int h,
                                                             const float *__restrict__ b, int ks) {
    extern __shared__ float lds[];
    const int tw = kPostBlock + 2 * ks;
    float *R = lds, *G = lds + tw, *B = lds + 2 * tw;
    const int y = blockIdx.y, x0 = blockIdx.x * kPostBlock - ks;
    for (int i = threadIdx.x; i < tw; i += kPostBlock) {
        const int gx = x0 + i;
        float r = 0.f, g = 0.f, bl = 0.f;
        if (gx >= 0 && gx < w) {
            const float *p = in + 3 * ((size_t)y * w + gx);
            r = p[0]; g = p[1]; bl = p[2];
        }
        R[i] = r; G[i] = g; B[i] = bl;
    }
    __syncthreads();
    const int u = blockIdx.x * kPostBlock + threadIdx.x;
    if (u >= w) return;
    const int c = threadIdx.x + ks;
    float pr = 0.f, pg = 0.f, pb = 0.f;
    for (int dx = -ks; dx <= ks; dx++) {          /* out-of-image texels are staged as zeros */
        const float t = b[ks + dx];
        pr = pr + t * R[c + dx];
        pg = pg + t * G[c + dx];
        pb = pb + t * B[c + dx];
    }
    float *o = out + 3 * ((size_t)y * w + u);
    o[0] = pr; o[1] = pg; o[2] = pb;
}

/* pass along y over the row sums, and the normalisation by the weight of the cropped window:
 * (sum of a over in-image rows) * (sum of b over in-image columns) */
__global__ void __launch_bounds__(kPostBlock) sep_col_kernel(const float *__restrict__ in, float *__restrict__ out, int w, int h,
                                                             const float *__restrict__ a, const float *__restrict__ b, int ks) {
    extern __shared__ float lds[];
    const int th = kTileH + 2 * ks;
    const int plane = kTileW * th;
    float *R = lds, *G = lds + plane, *B = lds + 2 * plane;
    const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH - ks;
    for (int i = threadIdx.x; i < plane; i += kPostBlock) {
        const int ly = i / kTileW, lx = i - ly * kTileW;
        const int gx = x0 + lx, gy = y0 + ly;
        float r = 0.f, g = 0.f, bl = 0.f;
        if (gx < w && gy >= 0 && gy < h) {
            const float *p = in + 3 * ((size_t)gy * w + gx);
            r = p[0]; g = p[1]; bl = p[2];
        }
        R[i] = r; G[i] = g; B[i] = bl;
    }
    __syncthreads();
    const int lx = threadIdx.x % kTileW, ly = threadIdx.x / kTileW;
    const int u = x0 + lx, v = blockIdx.y * kTileH + ly;
    if (u >= w || v >= h) return;
    const int c = (ly + ks) * kTileW + lx;
    float pr = 0.f, pg = 0.f, pb = 0.f, wy = 0.f, wx = 0.f;
    for (int d = -ks; d <= ks; d++) {
        const float ta = a[ks + d], tb = b[ks + d];
        pr = pr + ta * R[c + d * kTileW];
        pg = pg + ta * G[c + d * kTileW];
        pb = pb + ta * B[c + d * kTileW];
        wy += (v + d >= 0 && v + d < h) ? ta : 0.f;
        wx += (u + d >= 0 && u + d < w) ? tb : 0.f;
    }
    const float wsum = wy * wx;
    float *o = out + 3 * ((size_t)v * w + u);
    o[0] = pr / wsum; o[1] = pg / wsum; o[2] = pb / wsum;
}

__global__ void __launch_bounds__(kPostBlock) normalize_kernel(float *buf, int64_t n, float div) {
    const int64_t i = (int64_t)blockIdx.x * kPostBlock + threadIdx.x;
    if (i < n) buf[i] = buf[i] / div;
}
__global__ void __launch_bounds__(kPostBlock) threshold_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t n_pixels, float thr) {
    const int64_t i = (int64_t)blockIdx.x * kPostBlock + threadIdx.x;
    if (i >= n_pixels) return;
    const float r = in[3 * i], g = in[3 * i + 1], b = in[3 * i + 2];
    const bool keep = !(lum3(r, g, b) < thr);
    out[3 * i] = keep ? r : 0.f; out[3 * i + 1] = keep ? g : 0.f; out[3 * i + 2] = keep ? b : 0.f;
}
__global__ void __launch_bounds__(kPostBlock) add_kernel(float *buf, const float *__restrict__ layer, int64_t n, float opacity) {
    const int64_t i = (int64_t)blockIdx.x * kPostBlock + threadIdx.x;
    if (i < n) buf[i] = buf[i] + opacity * layer[i];
}
/* ACESFilm, src/post_processing.cu:194-205 */
__device__ __forceinline__ float aces1(float x) {
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    const float r = (x * (a * x + b)) / (x * (c * x + d) + e);
    return fmaxf(0.0f, fminf(1.0f, r));
}
__global__ void __launch_bounds__(kPostBlock) aces_kernel(float *buf, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kPostBlock + threadIdx.x;
    if (i < n) buf[i] = aces1(buf[i]);
}
__global__ void __launch_bounds__(kPostBlock) gamma_kernel(float *buf, int64_t n, float inv_gamma) {
    const int64_t i = (int64_t)blockIdx.x * kPostBlock + threadIdx.x;
    if (i < n) buf[i] = powf(buf[i], inv_gamma);
}
/* fused steps of the fixed chain: normalise + threshold, and add + ACES + gamma */
__global__ void __launch_bounds__(kPostBlock) chain_pre_kernel(const float *__restrict__ sum, float *__restrict__ img, float *__restrict__ bright,
                                                               int64_t n_pixels, float div, float thr) {
    const int64_t i = (int64_t)blockIdx.x * kPostBlock + threadIdx.x;
    if (i >= n_pixels) return;
    const float r = sum[3 * i] / div, g = sum[3 * i + 1] / div, b = sum[3 * i + 2] / div;
    img[3 * i] = r; img[3 * i + 1] = g; img[3 * i + 2] = b;
    const bool keep = !(lum3(r, g, b) < thr);
    bright[3 * i] = keep ? r : 0.f; bright[3 * i + 1] = keep ? g : 0.f; bright[3 * i + 2] = keep ? b : 0.f;
}
__global__ void __launch_bounds__(kPostBlock) chain_mid_kernel(float *img, const float *__restrict__ bloom, int64_t n, float opacity, float inv_gamma) {
    const int64_t i = (int64_t)blockIdx.x * kPostBlock + threadIdx.x;
    if (i < n) {
        const float x = img[i] + opacity * bloom[i];
        img[i] = powf(aces1(x), inv_gamma);
    }
}

static inline unsigned grid1(int64_t n) { return (unsigned)((n + kPostBlock - 1) / kPostBlock); }

static int run_filter(const float *d_in, float *d_out, int w, int h, const float *d_taps, int ks, int type, hipStream_t stream) {
    const size_t lds = (size_t)3 * (kTileW + 2 * ks) * (kTileH + 2 * ks) * sizeof(float);
    if (lds > 160 * 1024) PT_FAIL("pt_apply_filter: kernel size too large for the LDS tile");
    dim3 grid((w + kTileW - 1) / kTileW, (h + kTileH - 1) / kTileH);
    if (type == 1) {
        if (lds > 64 * 1024) PT_HIP(hipFuncSetAttribute((const void *)filter_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(filter_kernel<true>, grid, dim3(kPostBlock), lds, stream, d_in, d_out, w, h, d_taps, ks);
    } else {
        if (lds > 64 * 1024) PT_HIP(hipFuncSetAttribute((const void *)filter_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(filter_kernel<false>, grid, dim3(kPostBlock), lds, stream, d_in, d_out, w, h, d_taps, ks);
    }
    PT_HIP(hipGetLastError());
    return 0;
}

/* a[y] * b[x] == taps[y][x] for every tap (to a few ulp)?  Then a and b (2*ks+1 floats each) are filled in. */
static bool factor_taps(const float *taps, int ks, std::vector<float> &a, std::vector<float> &b) {
    const int kw = 2 * ks + 1;
    const float c = taps[(size_t)ks * kw + ks];
    if (!(std::fabs(c) > 0.0f) || !std::isfinite(c)) return false;
    a.resize(kw); b.resize(kw);
    for (int i = 0; i < kw; i++) {
        a[i] = taps[(size_t)i * kw + ks];
        b[i] = taps[(size_t)ks * kw + i] / c;
    }
    for (int y = 0; y < kw; y++)
        for (int x = 0; x < kw; x++) {
            const float t = taps[(size_t)y * kw + x], p = a[y] * b[x];
            if (!std::isfinite(t) || std::fabs(t - p) > 2e-6f * std::fabs(t) + 1e-37f) return false;
        }
    return true;
}

/* the two passes; d_ab = a then b on the device, d_tmp = one image of scratch (may not alias in or out) */
static int run_separable(const float *d_in, float *d_tmp, float *d_out, int w, int h, const float *d_ab, int ks, hipStream_t stream) {
    const int kw = 2 * ks + 1;
    const size_t lds_row = (size_t)3 * (kPostBlock + 2 * ks) * sizeof(float);
    const size_t lds_col = (size_t)3 * kTileW * (kTileH + 2 * ks) * sizeof(float);
    if (lds_row > 64 * 1024 || lds_col > 64 * 1024) PT_FAIL("pt_apply_filter: kernel size too large for the LDS tile");
    hipLaunchKernelGGL(sep_row_kernel, dim3((w + kPostBlock - 1) / kPostBlock, h), dim3(kPostBlock), lds_row, stream, d_in, d_tmp, w, h,
                       d_ab + kw, ks);
    hipLaunchKernelGGL(sep_col_kernel, dim3((w + kTileW - 1) / kTileW, (h + kTileH - 1) / kTileH), dim3(kPostBlock), lds_col, stream,
                       (const float *)d_tmp, d_out, w, h, d_ab, d_ab + kw, ks);
    PT_HIP(hipGetLastError());
    return 0;
}

struct DevSel {
    int prev = -1;
    bool ok = true;
    explicit DevSel(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DevSel() { if (prev >= 0) (void)hipSetDevice(prev); }
};

/* src/post_processing.cu:40-46: the factor is evaluated in double (M_PI), the power in float */
static float gaussian_2d(float x, float y, float sigma) {
    float fac = 1.0f / (sigma * sigma * 2 * M_PI);
    float ex = -((x * x + y * y) / (2 * sigma * sigma));
    return fac * powf(M_E, ex);
}

} // namespace ptamd

using namespace ptamd;

extern "C" {

int pt_make_gaussian(unsigned size, float sigma, float *taps_out) {
    if (!taps_out) PT_FAIL("pt_make_gaussian: null argument");
    const int s = (int)size;
    for (int y = 0; y < 2 * s + 1; y++)
        for (int x = 0; x < 2 * s + 1; x++)
            taps_out[(size_t)y * (2 * size + 1) + x] = gaussian_2d(float(x - s), float(y - s), sigma);
    return 0;
}

int pt_make_mean(unsigned size, float *taps_out) {
    if (!taps_out) PT_FAIL("pt_make_mean: null argument");
    const size_t n = (size_t)(2 * size + 1) * (2 * size + 1);
    for (size_t i = 0; i < n; i++) taps_out[i] = 1.0f;
    return 0;
}

int pt_apply_filter(int device, const float *d_in, float *d_out, int w, int h, const float *taps, int ksize, int filter_type, void *stream_) {
    if (!d_in || !d_out || !taps) PT_FAIL("pt_apply_filter: null argument");
    if (w <= 0 || h <= 0) PT_FAIL("pt_apply_filter: empty image");
    if (ksize < 0) PT_FAIL("pt_apply_filter: negative kernel size");
    if (filter_type != 0 && filter_type != 1) PT_FAIL("pt_apply_filter: unknown filter type");
    if (d_in == d_out) PT_FAIL("pt_apply_filter: in-place filtering is not supported");
    DevSel g(device);
    hipStream_t stream = (hipStream_t)stream_;
    const size_t nt = (size_t)(2 * ksize + 1) * (2 * ksize + 1);
    std::vector<float> fa, fb;
    const size_t lds_col = (size_t)3 * kTileW * (kTileH + 2 * ksize) * sizeof(float);
    const bool separable = filter_type == 0 && lds_col <= 64 * 1024 && (size_t)h < 65536 && factor_taps(taps, ksize, fa, fb);
    float *d_taps = nullptr, *d_tmp = nullptr;
    PT_HIP(hipMalloc((void **)&d_taps, (nt < 2 ? 2 : nt) * sizeof(float)));     /* a 1 x 1 table still factors into two */
    int rc = 0;
    if (separable) {
        const size_t kw = 2 * (size_t)ksize + 1;
        if (hipMalloc((void **)&d_tmp, (size_t)w * h * 12) != hipSuccess) { rc = 1; set_error("pt_apply_filter: out of device memory"); }
        if (!rc && (hipMemcpyAsync(d_taps, fa.data(), kw * sizeof(float), hipMemcpyHostToDevice, stream) != hipSuccess ||
                    hipMemcpyAsync(d_taps + kw, fb.data(), kw * sizeof(float), hipMemcpyHostToDevice, stream) != hipSuccess)) {
            rc = 1; set_error("pt_apply_filter: tap upload failed");
        }
        if (!rc) rc = run_separable(d_in, d_tmp, d_out, w, h, d_taps, ksize, stream);
    } else {
        if (hipMemcpyAsync(d_taps, taps, nt * sizeof(float), hipMemcpyHostToDevice, stream) != hipSuccess) { rc = 1; set_error("pt_apply_filter: tap upload failed"); }
        if (!rc) rc = run_filter(d_in, d_out, w, h, d_taps, ksize, filter_type, stream);
    }
    /* the taps (host vectors included) and the scratch image must outlive the kernels */
    if (hipStreamSynchronize(stream) != hipSuccess && !rc) { rc = 1; set_error("pt_apply_filter: kernel failed"); }
    (void)hipFree(d_taps);
    if (d_tmp) (void)hipFree(d_tmp);
    return rc;
}

int pt_normalize_image(int device, float *d_buf, int64_t n_pixels, unsigned n_samples, void *stream) {
    if (!d_buf) PT_FAIL("pt_normalize_image: null argument");
    if (n_pixels <= 0) return 0;
    DevSel g(device);
    hipLaunchKernelGGL(normalize_kernel, dim3(grid1(3 * n_pixels)), dim3(kPostBlock), 0, (hipStream_t)stream, d_buf, 3 * n_pixels, (float)n_samples);
    PT_HIP(hipGetLastError());
    return 0;
}
int pt_apply_threshold(int device, const float *d_in, float *d_out, int64_t n_pixels, float threshold, void *stream) {
    if (!d_in || !d_out) PT_FAIL("pt_apply_threshold: null argument");
    if (n_pixels <= 0) return 0;
    DevSel g(device);
    hipLaunchKernelGGL(threshold_kernel, dim3(grid1(n_pixels)), dim3(kPostBlock), 0, (hipStream_t)stream, d_in, d_out, n_pixels, threshold);
    PT_HIP(hipGetLastError());
    return 0;
}
int pt_image_add(int device, float *d_buf, const float *d_layer, int64_t n_pixels, float opacity, void *stream) {
    if (!d_buf || !d_layer) PT_FAIL("pt_image_add: null argument");
    if (n_pixels <= 0) return 0;
    DevSel g(device);
    hipLaunchKernelGGL(add_kernel, dim3(grid1(3 * n_pixels)), dim3(kPostBlock), 0, (hipStream_t)stream, d_buf, d_layer, 3 * n_pixels, opacity);
    PT_HIP(hipGetLastError());
    return 0;
}
int pt_reduce_sum(int dst_device, float *d_dst, int n_src, const int *src_devices, const float *const *d_src, int64_t n_floats) {
    if (!d_dst || !src_devices || !d_src) PT_FAIL("pt_reduce_sum: null argument");
    if (n_src < 0 || n_floats < 0) PT_FAIL("pt_reduce_sum: negative count");
    for (int i = 0; i < n_src; i++)
        if (!d_src[i]) PT_FAIL("pt_reduce_sum: null source buffer");
    if (n_src == 0 || n_floats == 0) return 0;
    DevSel g(dst_device);
    if (!g.ok) PT_FAIL("pt_reduce_sum: cannot select device");
    const size_t bytes = (size_t)n_floats * sizeof(float);
    float *d_tmp = nullptr;
    PT_HIP(hipMalloc((void **)&d_tmp, bytes));
    int rc = 0;
    /* peer access from the destination to every other source GPU that allows it: copies between peers then go over the
     * GPUs' links; without it (or where the two GPUs cannot be peers) the runtime stages the copy through host memory.
     * "Already enabled" is fine; any other refusal only costs speed. */
    for (int i = 0; i < n_src; i++) {
        if (src_devices[i] == dst_device) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dst_device, src_devices[i]) == hipSuccess && can) {
            const hipError_t pe = hipDeviceEnablePeerAccess(src_devices[i], 0);
            if (pe != hipSuccess) (void)hipGetLastError();      /* hipErrorPeerAccessAlreadyEnabled, or not permitted: copy staged */
        } else {
            (void)hipGetLastError();
        }
    }
    for (int i = 0; i < n_src && !rc; i++) {
        /* the source devices have finished writing (the caller synchronised them) */
        hipError_t e = src_devices[i] == dst_device ? hipMemcpyAsync(d_tmp, d_src[i], bytes, hipMemcpyDeviceToDevice, nullptr)
                                                    : hipMemcpyPeerAsync(d_tmp, dst_device, d_src[i], src_devices[i], bytes, nullptr);
        if (e != hipSuccess) { rc = hip_fail(e, "peer copy", __FILE__, __LINE__); break; }
        hipLaunchKernelGGL(add_kernel, dim3(grid1(n_floats)), dim3(kPostBlock), 0, nullptr, d_dst, d_tmp, n_floats, 1.0f);
        e = hipGetLastError();
        if (e != hipSuccess) rc = hip_fail(e, "add_kernel", __FILE__, __LINE__);
    }
    hipError_t e = hipStreamSynchronize(nullptr);
    if (!rc && e != hipSuccess) rc = hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__);
    (void)hipFree(d_tmp);
    return rc;
}

int pt_apply_aces(int device, float *d_buf, int64_t n_pixels, void *stream) {
    if (!d_buf) PT_FAIL("pt_apply_aces: null argument");
    if (n_pixels <= 0) return 0;
    DevSel g(device);
    hipLaunchKernelGGL(aces_kernel, dim3(grid1(3 * n_pixels)), dim3(kPostBlock), 0, (hipStream_t)stream, d_buf, 3 * n_pixels);
    PT_HIP(hipGetLastError());
    return 0;
}
int pt_gamma_correct(int device, float *d_buf, int64_t n_pixels, float gamma, void *stream) {
    if (!d_buf) PT_FAIL("pt_gamma_correct: null argument");
    if (n_pixels <= 0) return 0;
    DevSel g(device);
    hipLaunchKernelGGL(gamma_kernel, dim3(grid1(3 * n_pixels)), dim3(kPostBlock), 0, (hipStream_t)stream, d_buf, 3 * n_pixels, 1.0f / gamma);
    PT_HIP(hipGetLastError());
    return 0;
}

/* src/renderer.cu:277-297: normalise, threshold 3.0, 65x65 gaussian (sigma 10) bloom added at 0.5,
 * ACES, gamma 2.2, 5x5 (sigma 0.7) bilateral */
int pt_post_chain(int device, const float *d_sum, float *d_out, int w, int h, unsigned n_samples, void *stream_) {
    if (!d_sum || !d_out) PT_FAIL("pt_post_chain: null argument");
    if (w <= 0 || h <= 0) PT_FAIL("pt_post_chain: empty image");
    DevSel g(device);
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t n_pixels = (int64_t)w * h;
    const size_t img_bytes = (size_t)n_pixels * 12;
    std::vector<float> bloom_taps(65 * 65), bil_taps(5 * 5);
    pt_make_gaussian(32, 10.0f, bloom_taps.data());
    pt_make_gaussian(2, 0.7f, bil_taps.data());
    std::vector<float> fa, fb;
    const bool separable = h < 65536 && factor_taps(bloom_taps.data(), 32, fa, fb);
    float *d_img = nullptr, *d_bright = nullptr, *d_bloom = nullptr, *d_taps = nullptr;
    int rc = 0;
    do {
        if (hipMalloc((void **)&d_img, img_bytes) != hipSuccess || hipMalloc((void **)&d_bright, img_bytes) != hipSuccess ||
            hipMalloc((void **)&d_bloom, img_bytes) != hipSuccess || hipMalloc((void **)&d_taps, (65 * 65 + 25) * sizeof(float)) != hipSuccess) {
            rc = 1; set_error("pt_post_chain: out of device memory"); break;
        }
        if (separable) {     /* a, b at the head of the 65 x 65 slot */
            memcpy(bloom_taps.data(), fa.data(), 65 * sizeof(float));
            memcpy(bloom_taps.data() + 65, fb.data(), 65 * sizeof(float));
        }
        if (hipMemcpyAsync(d_taps, bloom_taps.data(), 65 * 65 * 4, hipMemcpyHostToDevice, stream) != hipSuccess ||
            hipMemcpyAsync(d_taps + 65 * 65, bil_taps.data(), 25 * 4, hipMemcpyHostToDevice, stream) != hipSuccess) {
            rc = 1; set_error("pt_post_chain: tap upload failed"); break;
        }
        hipLaunchKernelGGL(chain_pre_kernel, dim3(grid1(n_pixels)), dim3(kPostBlock), 0, stream, d_sum, d_img, d_bright, n_pixels, (float)n_samples, 3.0f);
        const float *d_glow = d_bloom;
        if (separable) {
            /* row sums into d_bloom, column pass back into d_bright (the thresholded image is no longer needed) */
            if ((rc = run_separable(d_bright, d_bloom, d_bright, w, h, d_taps, 32, stream))) break;
            d_glow = d_bright;
        } else if ((rc = run_filter(d_bright, d_bloom, w, h, d_taps, 32, 0, stream))) break;
        hipLaunchKernelGGL(chain_mid_kernel, dim3(grid1(3 * n_pixels)), dim3(kPostBlock), 0, stream, d_img, d_glow, 3 * n_pixels, 0.5f, 1.0f / 2.2f);
        if ((rc = run_filter(d_img, d_out, w, h, d_taps + 65 * 65, 2, 1, stream))) break;
        if (hipGetLastError() != hipSuccess) { rc = 1; set_error("pt_post_chain: launch failed"); break; }
    } while (0);
    if (hipStreamSynchronize(stream) != hipSuccess && !rc) { rc = 1; set_error("pt_post_chain: kernel failed"); }
    if (d_img) (void)hipFree(d_img);
    if (d_bright) (void)hipFree(d_bright);
    if (d_bloom) (void)hipFree(d_bloom);
    if (d_taps) (void)hipFree(d_taps);
    return rc;
}

} // extern "C"
