// Input pipeline of the tokenizer: the steps datasets/imagenetC.py and
// datasets/transforms_image.py run on the CPU with PIL / torchvision, restated as integer kernels.
//
//   * cvar_resample_u8       one pass (horizontal or vertical) of PIL's 8-bit separable resampler (Pillow
//                            src/libImaging/Resample.c, ImagingResampleHorizontal_8bpc / Vertical_8bpc): fixed-point
//                            coefficients with 22 fraction bits, accumulator seeded with 1 << 21, arithmetic shift, clip to
//                            [0, 255].  The coefficient tables are computed on the host exactly as precompute_coeffs /
//                            normalize_coeffs_8bpc do (controlvar_amd/preprocess.py) - so the result is bit-identical to
//                            Image.resize for every filter.
//   * cvar_crop_flip_normalize   F.crop / F.hflip / to_tensor / normalize(0.5, 0.5): uint8 HWC window -> fp32 CHW in [-1, 1]
//                            with the same two roundings ((x / 255) - 0.5) / 0.5 as torch.
//   * cvar_ignore_mask       datasets/imagenetC.py:152-185: background = all three channels == -1; scales with index >= 5
//                            take the nearest-neighbour downsample (torch's floorf(dst * in/out) source index) of the
//                            background mask for the control half, ones elsewhere.
#include "cvar_common.h"

__global__ void resample_u8_kernel(const unsigned char* __restrict__ src, int src_h, int src_w, int ch, int axis, int dst_extent,
                                   const int* __restrict__ bounds, const int* __restrict__ coeffs, int ksize,
                                   unsigned char* __restrict__ dst) {
    // axis 0: horizontal (dst is src_h x dst_extent); axis 1: vertical (dst is dst_extent x src_w)
    const int dst_h = axis == 0 ? src_h : dst_extent, dst_w = axis == 0 ? dst_extent : src_w;
    const long total = (long)dst_h * dst_w * ch;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % ch);
        const long pix = i / ch;
        const int x = (int)(pix % dst_w), y = (int)(pix / dst_w);
        const int o = axis == 0 ? x : y;
        const int lo = bounds[2 * o], n = bounds[2 * o + 1];
        const int* k = coeffs + (long)o * ksize;
        int ss = 1 << 21;                                              // 1 << (PRECISION_BITS - 1)
        if (axis == 0) {
            const unsigned char* row = src + ((long)y * src_w + lo) * ch + c;
            for (int t = 0; t < n; ++t) ss += (int)row[(long)t * ch] * k[t];
        } else {
            const unsigned char* col = src + ((long)lo * src_w + x) * ch + c;
            for (int t = 0; t < n; ++t) ss += (int)col[(long)t * src_w * ch] * k[t];
        }
        const int v = ss >> 22;                                        // arithmetic shift, then clip8
        dst[i] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

extern "C" int cvar_resample_u8(const void* src, int src_h, int src_w, int channels, int axis, int dst_extent,
                                const int* bounds, const int* coeffs, int ksize, void* dst, void* stream) {
    if (!src || !dst || !bounds || !coeffs || src_h <= 0 || src_w <= 0 || channels <= 0 || dst_extent <= 0 || ksize <= 0 || (axis != 0 && axis != 1))
        return CVAR_EINVAL;
    const long total = (long)(axis == 0 ? src_h : dst_extent) * (axis == 0 ? dst_extent : src_w) * channels;
    const int grid = (int)min((long)4096, (total + 255) / 256);
    hipLaunchKernelGGL(resample_u8_kernel, dim3(grid), dim3(256), 0, as_stream(stream), (const unsigned char*)src, src_h, src_w, channels, axis,
                       dst_extent, bounds, coeffs, ksize, (unsigned char*)dst);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

__global__ void crop_flip_normalize_kernel(const unsigned char* __restrict__ src, int src_w, int ch, int top, int left, int out_h, int out_w,
                                           int flip, float* __restrict__ dst) {
    const long total = (long)ch * out_h * out_w;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % out_w), y = (int)((i / out_w) % out_h), c = (int)(i / ((long)out_w * out_h));
        const int sx = left + (flip ? out_w - 1 - x : x), sy = top + y;
        const float v = (float)src[((long)sy * src_w + sx) * ch + c] / 255.0f;        // to_tensor: correctly rounded division
        dst[i] = (v - 0.5f) / 0.5f;                                                   // normalize(mean 0.5, std 0.5)
    }
}

extern "C" int cvar_crop_flip_normalize(const void* src, int src_h, int src_w, int channels, int top, int left, int out_h, int out_w,
                                        int flip, float* dst, void* stream) {
    if (!src || !dst || channels <= 0 || out_h <= 0 || out_w <= 0 || top < 0 || left < 0 || top + out_h > src_h || left + out_w > src_w)
        return CVAR_EINVAL;
    const long total = (long)channels * out_h * out_w;
    hipLaunchKernelGGL(crop_flip_normalize_kernel, dim3((int)min((long)4096, (total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       (const unsigned char*)src, src_w, channels, top, left, out_h, out_w, flip, dst);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

struct IgnoreParams {
    const float* cond; int B, H, W; int n_scales; int pn[16]; int first_masked; int image_first; float* out; int L;
};

__global__ void ignore_mask_kernel(const IgnoreParams p) {
    const long total = (long)p.B * p.L;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / p.L);
        int t = (int)(i % p.L);
        float v = 1.0f;
        for (int s = 0; s < p.n_scales; ++s) {
            const int n2 = p.pn[s] * p.pn[s];
            if (t < 2 * n2) {
                const bool control_half = p.image_first ? (t >= n2) : (t < n2);
                if (control_half && s >= p.first_masked) {
                    const int q = t % n2, oy = q / p.pn[s], ox = q % p.pn[s];
                    // torch nearest: src = min((int)floorf(dst * (float)in / out), in - 1)
                    const int sy = min((int)floorf((float)oy * ((float)p.H / (float)p.pn[s])), p.H - 1);
                    const int sx = min((int)floorf((float)ox * ((float)p.W / (float)p.pn[s])), p.W - 1);
                    const long plane = (long)p.H * p.W;
                    const float* px = p.cond + (long)b * 3 * plane + (long)sy * p.W + sx;
                    const bool background = (px[0] + px[plane] + px[2 * plane]) == -3.0f;   // cond.sum(dim=0) == -3, same summation order
                    v = background ? 0.0f : 1.0f;
                }
                break;
            }
            t -= 2 * n2;
        }
        p.out[i] = v;
    }
}

extern "C" int cvar_ignore_mask(const float* cond, int B, int H, int W, const int* patch_nums_host, int n_scales, int first_masked_scale,
                                int image_first, float* out, int L, void* stream) {
    if (!cond || !out || !patch_nums_host || B <= 0 || H <= 0 || W <= 0 || n_scales <= 0 || n_scales > 16) return CVAR_EINVAL;
    IgnoreParams p;
    p.cond = cond; p.B = B; p.H = H; p.W = W; p.n_scales = n_scales; p.first_masked = first_masked_scale; p.image_first = image_first;
    p.out = out; p.L = L;
    int sum = 0;
    for (int s = 0; s < 16; ++s) { p.pn[s] = s < n_scales ? patch_nums_host[s] : 0; sum += 2 * p.pn[s] * p.pn[s]; }
    if (sum != L) return CVAR_EINVAL;
    const long total = (long)B * L;
    hipLaunchKernelGGL(ignore_mask_kernel, dim3((int)min((long)2048, (total + 255) / 256)), dim3(256), 0, as_stream(stream), p);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}

// cvar_rle_paint: the raster half of process_anns (datasets/imagenetC.py:15-29).  Every kept annotation is a column-major
// run-length mask (COCO RLE: runs alternate 0 / 1 starting with 0); run_ends holds, per annotation, the exclusive prefix
// sums of its runs.  A pixel takes the colour of the LAST annotation that covers it (mask[m] = colour in file order), black
// otherwise.  The string codec and the centroid -> colour rule stay on the host (controlvar_amd/preprocess.py).
__global__ void rle_paint_kernel(const int* __restrict__ run_ends, const int* __restrict__ ann_off, const unsigned char* __restrict__ colours,
                                 int n_ann, int H, int W, unsigned char* __restrict__ out) {
    const long total = (long)H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i % W);
        const int pos = x * H + y;                       // column-major position of the pixel inside the RLE stream
        int hit = -1;
        for (int a = 0; a < n_ann; ++a) {
            const int* e = run_ends + ann_off[a];
            int lo = 0, hi = ann_off[a + 1] - ann_off[a];            // first run r with e[r] > pos
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (e[mid] > pos) hi = mid; else lo = mid + 1;
            }
            if (lo < ann_off[a + 1] - ann_off[a] && (lo & 1)) hit = a;   // odd runs are the ones
        }
        unsigned char* o = out + i * 3;
        if (hit >= 0) { o[0] = colours[3 * hit]; o[1] = colours[3 * hit + 1]; o[2] = colours[3 * hit + 2]; }
        else { o[0] = 0; o[1] = 0; o[2] = 0; }
    }
}

extern "C" int cvar_rle_paint(const int* run_ends, const int* ann_offsets, const void* colours, int n_ann, int H, int W, void* out, void* stream) {
    if (!out || H <= 0 || W <= 0 || n_ann < 0 || (n_ann > 0 && (!run_ends || !ann_offsets || !colours))) return CVAR_EINVAL;
    const long total = (long)H * W;
    hipLaunchKernelGGL(rle_paint_kernel, dim3((int)min((long)2048, (total + 255) / 256)), dim3(256), 0, as_stream(stream), run_ends, ann_offsets,
                       (const unsigned char*)colours, n_ann, H, W, (unsigned char*)out);
    CVAR_CHECK_LAUNCH();
    return CVAR_OK;
}
