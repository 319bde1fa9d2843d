// GPU image front end: uint8 RGB page -> PIL-identical bicubic resize -> rescale / normalise
// -> Qwen2-VL patch order, all in HBM.  The host sends the 3 bytes per pixel it decoded instead of 2 x 1176 floats
// per patch (1024x1024 page: 3 MB instead of 23 MB) and does no resampling.
//
//  resample_axis_kernel : one pass of Pillow's 8-bit resample (src/libImaging/Resample.c,
//                         ImagingResampleHorizontal_8bpc / Vertical_8bpc): per output coordinate a window of the
//                         input and integer weights with 22 fractional bits (tables built on the host by
//                         image_processing.resample_tables in the same double arithmetic), accumulate in int32 from
//                         1 << 21, shift, clamp to [0, 255].  Integer work: bit-identical to PIL.
//  normalize_patchify_kernel : x * (1/255) -> (x - mean) / std in fp32, one rounding per operation (no fma
//                         contraction: the HF processor does three separate numpy float32 operations,
//                         image_processing_pil_qwen2_vl.py:226-229), written in the patch order of :152-187
//                         [(gh/m, gw/m, m, m), (C, T, p, p)] with the frame repeated T times.  HBM-bound, tiny.
#include "kr_common.h"

namespace {

// AXIS 0: resample x (src [rows][in][3] -> dst [rows][out][3]); AXIS 1: resample y (src [in][cols][3] -> dst [out][cols][3])
template <int AXIS>
__global__ void __launch_bounds__(256) resample_axis_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                            int rows, int cols_in, int out, const int32_t* __restrict__ bounds,
                                                            const int32_t* __restrict__ coeffs, int ksize) {
    // AXIS 0: (r, o) = (row, output x), line length cols_in;  AXIS 1: rows = output rows, r = output y, o = column
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int width_out = AXIS == 0 ? out : cols_in;
    const int64_t total = (int64_t)(AXIS == 0 ? rows : out) * width_out;
    if (idx >= total) return;
    const int r = (int)(idx / width_out), o = (int)(idx - (int64_t)r * width_out);
    const int t = AXIS == 0 ? o : r;                    // the output coordinate along the resampled axis
    const int x0 = bounds[2 * t], n = bounds[2 * t + 1];
    const int32_t* k = coeffs + (int64_t)t * ksize;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int i = 0; i < n; ++i) {
        const uint8_t* p = AXIS == 0 ? src + ((int64_t)r * cols_in + (x0 + i)) * 3 : src + ((int64_t)(x0 + i) * cols_in + o) * 3;
        const int w = k[i];
        s0 += p[0] * w;
        s1 += p[1] * w;
        s2 += p[2] * w;
    }
    uint8_t* q = dst + idx * 3;
    q[0] = (uint8_t)min(max(s0 >> 22, 0), 255);
    q[1] = (uint8_t)min(max(s1 >> 22, 0), 255);
    q[2] = (uint8_t)min(max(s2 >> 22, 0), 255);
}

__global__ void __launch_bounds__(256) normalize_patchify_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, int rh,
                                                                 int rw, float m0, float m1, float m2, float d0, float d1,
                                                                 float d2, int patch, int merge, int temporal) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one thread per (y, x)
    if (idx >= (int64_t)rh * rw) return;
    const int y = (int)(idx / rw), x = (int)(idx - (int64_t)y * rw);
    const int gy = y / patch, py = y - gy * patch, gx = x / patch, px = x - gx * patch;
    const int gw = rw / patch;
    const int64_t n = ((int64_t)(gy / merge) * (gw / merge) + gx / merge) * (merge * merge) + (gy % merge) * merge + gx % merge;
    const int pp = patch * patch, row = 3 * temporal * pp;
    const uint8_t* p = img + idx * 3;
    const float mean[3] = {m0, m1, m2}, sd[3] = {d0, d1, d2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // three roundings, as numpy does them.  The library is built with -ffp-contract=fast and the backend would
        // fuse the multiply into the subtraction whatever the source says: the empty asm keeps the product opaque.
        float scaled = (float)p[c] * (1.0f / 255.0f);
        asm volatile("" : "+v"(scaled));
        float centred = scaled - mean[c];
        asm volatile("" : "+v"(centred));
        const float v = centred / sd[c];
        float* o = out + n * row + (int64_t)c * temporal * pp + py * patch + px;
        for (int t = 0; t < temporal; ++t) o[(int64_t)t * pp] = v;
    }
}

}  // namespace

extern "C" int kr_image_resize_bicubic_u8(const uint8_t* src, int h, int w, uint8_t* dst, int rh, int rw, uint8_t* tmp,
                                          const int32_t* h_bounds, const int32_t* h_coeffs, int h_ksize,
                                          const int32_t* v_bounds, const int32_t* v_coeffs, int v_ksize, kr_stream s) {
    KR_CHECK_ARG(src && dst && h > 0 && w > 0 && rh > 0 && rw > 0, "kr_image_resize_bicubic_u8: bad image");
    KR_CHECK_ARG(rw == w || (h_bounds && h_coeffs && h_ksize > 0), "kr_image_resize_bicubic_u8: horizontal tables missing");
    KR_CHECK_ARG(rh == h || (v_bounds && v_coeffs && v_ksize > 0), "kr_image_resize_bicubic_u8: vertical tables missing");
    KR_CHECK_ARG(!(rw != w && rh != h) || tmp, "kr_image_resize_bicubic_u8: both axes change: tmp [h][rw][3] needed");
    const uint8_t* cur = src;
    if (rw != w) {  // like PIL: an axis whose size stays is not resampled at all
        uint8_t* o = rh != h ? tmp : dst;
        const int64_t total = (int64_t)h * rw;
        resample_axis_kernel<0><<<(unsigned)((total + 255) / 256), 256, 0, kr_hs(s)>>>(cur, o, h, w, rw, h_bounds, h_coeffs, h_ksize);
        KR_CHECK_LAUNCH();
        cur = o;
    }
    if (rh != h) {
        const int64_t total = (int64_t)rh * rw;
        resample_axis_kernel<1><<<(unsigned)((total + 255) / 256), 256, 0, kr_hs(s)>>>(cur, dst, h, rw, rh, v_bounds, v_coeffs, v_ksize);
        KR_CHECK_LAUNCH();
        cur = dst;
    }
    if (cur == src) KR_CHECK_HIP(hipMemcpyAsync(dst, src, (size_t)h * w * 3, hipMemcpyDeviceToDevice, kr_hs(s)));
    return KR_OK;
}

extern "C" int kr_image_normalize_patchify(const uint8_t* img, int rh, int rw, const float* mean3, const float* std3,
                                           int patch, int merge, int temporal, float* out, kr_stream s) {
    KR_CHECK_ARG(img && out && mean3 && std3, "kr_image_normalize_patchify: null pointer");
    KR_CHECK_ARG(patch > 0 && merge > 0 && temporal > 0 && rh > 0 && rw > 0 && rh % (patch * merge) == 0 && rw % (patch * merge) == 0,
                 "kr_image_normalize_patchify: %dx%d is not a multiple of patch*merge = %d", rh, rw, patch * merge);
    const int64_t total = (int64_t)rh * rw;
    normalize_patchify_kernel<<<(unsigned)((total + 255) / 256), 256, 0, kr_hs(s)>>>(img, out, rh, rw, mean3[0], mean3[1], mean3[2],
                                                                                     std3[0], std3[1], std3[2], patch, merge, temporal);
    KR_CHECK_LAUNCH();
    return KR_OK;
}
