// Evaluation transforms of the harness as HIP kernels over the WHOLE batch: Gaussian blur, Gaussian noise,
// brightness, rotation, horizontal flip, upper-left crop + resize back / pad back.
//
// Reference: wmar/augmentations/valuemetric.py:41-140, geometric.py:22-117 (modules that delegate to
// torchvision.transforms.functional), applied by generate.py:142-164 to every generated batch: ~90 (transform, parameter) pairs per
// image, each followed by images_to_codes + detect.  The arithmetic restated here is the PUBLISHED torchvision tensor algorithm of
// each call (torchvision is not installed offline; tests/test_augmentations_algorithms.py holds independent numpy restatements):
//   gaussian_blur(img, k)      separable weights exp(-x^2 / 2 s^2), s = 0.3 ((k - 1) / 2 - 1) + 0.8, normalised; reflect padding;
//                              depthwise 2-D correlation with w[i] w[j]
//   adjust_brightness(img, f)  f * img clamped to [0, 1]
//   rotate(img, angle)         quarter turns as an exact permutation (geometric.py:38-46), the rest as the inverse affine map about the
//                              image centre in pixel-centre coordinates, nearest sample (round half to even), zero fill
//   resize(antialias=True)     separable triangle filter widened by the scale factor, normalised weights
// One pass per transform: with `pm1` the kernel reads the decoder's [-1, 1] pixels, works in [0, 1], clamps and writes [-1, 1] again
// (the harness's `aug(imgs / 2 + 0.5).clamp(0, 1) * 2 - 1`, generate.py:146-150, bit for bit: x / 2 and c * 2 are exact).
#include "common.h"

namespace wmar {

enum { AUG_IDENTITY = 0, AUG_BLUR = 1, AUG_NOISE = 2, AUG_BRIGHTNESS = 3, AUG_ROTATE = 4, AUG_FLIP_H = 5, AUG_CROP_RESIZE = 6, AUG_CROP_PAD = 7 };
constexpr int AUG_MAX_K = 63;

struct AugArgs {
    const float* in;
    float* out;
    const float* noise;     // AUG_NOISE: standard normal draws, same shape as the images
    int planes, H, W;       // planes = B * C
    int pm1;
    int k;                  // blur: odd kernel size
    float w[AUG_MAX_K];     // blur: 1-D weights
    float f;                // brightness factor / noise standard deviation
    float cs, sn;           // rotation: cos / sin of the remainder angle
    int quarters;           // rotation: counter-clockwise quarter turns applied first (0..3)
    int has_rest;           // rotation: remainder != 0
    int nh, nw;             // crop: kept rows / columns
};

__device__ __forceinline__ float aug_in(const AugArgs& a, long long idx) {
    const float v = a.in[idx];
    return a.pm1 ? v * 0.5f + 0.5f : v;
}
__device__ __forceinline__ float aug_clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ void aug_out(const AugArgs& a, long long idx, float v, bool clamp) {
    if (clamp || a.pm1) v = aug_clamp01(v);
    a.out[idx] = a.pm1 ? v * 2.0f - 1.0f : v;
}
// pixel (y, x) of the plane after `q` counter-clockwise quarter turns (torch.rot90 on the last two dims), read from the original plane
__device__ __forceinline__ float aug_rot90_px(const AugArgs& a, long long plane, int q, int y, int x) {
    const int H = a.H, W = a.W;
    int sy, sx;
    if (q == 0) { sy = y; sx = x; }
    else if (q == 1) { sy = x; sx = W - 1 - y; }
    else if (q == 2) { sy = H - 1 - y; sx = W - 1 - x; }
    else { sy = H - 1 - x; sx = y; }
    return aug_in(a, plane * H * W + (long long)sy * W + sx);
}

// 16 x 16 output pixels per workgroup, the (16 + k - 1)^2 reflect-padded patch in LDS
__global__ __launch_bounds__(256) void k_aug_blur(AugArgs a) {
    extern __shared__ float tile[];
    const int k = a.k, p = k / 2, TW = 16 + k - 1;
    const long long plane = blockIdx.z;
    const int y0 = blockIdx.y * 16, x0 = blockIdx.x * 16;
    for (int t = threadIdx.x; t < TW * TW; t += 256) {
        int yy = y0 + t / TW - p, xx = x0 + t % TW - p;
        yy = yy < 0 ? -yy : (yy >= a.H ? 2 * (a.H - 1) - yy : yy);
        xx = xx < 0 ? -xx : (xx >= a.W ? 2 * (a.W - 1) - xx : xx);
        yy = yy < 0 ? 0 : (yy >= a.H ? a.H - 1 : yy);          // tiles past the image edge (ragged sizes): any in-bounds pixel
        xx = xx < 0 ? 0 : (xx >= a.W ? a.W - 1 : xx);
        tile[t] = aug_in(a, plane * a.H * a.W + (long long)yy * a.W + xx);
    }
    __syncthreads();
    const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
    const int y = y0 + ty, x = x0 + tx;
    if (y >= a.H || x >= a.W) return;
    float acc = 0.f;
    for (int i = 0; i < k; ++i) {
        const float wi = a.w[i];
        for (int j = 0; j < k; ++j) acc = fmaf(wi * a.w[j], tile[(ty + i) * TW + tx + j], acc);
    }
    aug_out(a, plane * a.H * a.W + (long long)y * a.W + x, acc, true);
}

template <int OP>
__global__ __launch_bounds__(256) void k_aug_point(AugArgs a) {
    const long long n = (long long)a.planes * a.H * a.W;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int x = (int)(idx % a.W), y = (int)((idx / a.W) % a.H);
    const long long plane = idx / ((long long)a.W * a.H);
    if (OP == AUG_IDENTITY) aug_out(a, idx, aug_in(a, idx), false);
    else if (OP == AUG_NOISE) aug_out(a, idx, aug_in(a, idx) + a.f * a.noise[idx], true);
    else if (OP == AUG_BRIGHTNESS) aug_out(a, idx, aug_in(a, idx) * a.f, true);
    else if (OP == AUG_FLIP_H) aug_out(a, idx, aug_in(a, plane * a.H * a.W + (long long)y * a.W + (a.W - 1 - x)), false);
    else if (OP == AUG_CROP_PAD) aug_out(a, idx, (y < a.nh && x < a.nw) ? aug_in(a, idx) : 0.f, false);
    else if (OP == AUG_ROTATE) {
        // the canvas after the quarter turns (square images keep their shape; the host rejects odd quarter turns of non-square ones)
        float v;
        if (!a.has_rest) v = aug_rot90_px(a, plane, a.quarters, y, x);
        else {
            const float dx = (float)x + 0.5f - 0.5f * (float)a.W, dy = (float)y + 0.5f - 0.5f * (float)a.H;
            const float sx = a.cs * dx - a.sn * dy, sy = a.sn * dx + a.cs * dy;          // inverse map of a counter-clockwise rotation
            const float fx = rintf(sx + 0.5f * (float)a.W - 0.5f), fy = rintf(sy + 0.5f * (float)a.H - 0.5f);
            v = (fx >= 0.f && fx < (float)a.W && fy >= 0.f && fy < (float)a.H) ? aug_rot90_px(a, plane, a.quarters, (int)fy, (int)fx) : 0.f;
        }
        aug_out(a, idx, v, false);
    } else if (OP == AUG_CROP_RESIZE) {
        // triangle filter of support max(scale, 1) around the source centre scale * (i + 0.5), weights normalised per axis
        const float scy = (float)a.nh / (float)a.H, scx = (float)a.nw / (float)a.W;
        const float spy = fmaxf(scy, 1.f), spx = fmaxf(scx, 1.f);
        const float cy = scy * ((float)y + 0.5f), cx = scx * ((float)x + 0.5f);
        const int ylo = max((int)(cy - spy + 0.5f), 0), yhi = min((int)(cy + spy + 0.5f), a.nh);
        const int xlo = max((int)(cx - spx + 0.5f), 0), xhi = min((int)(cx + spx + 0.5f), a.nw);
        float wys = 0.f, wxs = 0.f;
        for (int j = ylo; j < yhi; ++j) wys += fmaxf(0.f, 1.f - fabsf(((float)j - cy + 0.5f) / spy));
        for (int j = xlo; j < xhi; ++j) wxs += fmaxf(0.f, 1.f - fabsf(((float)j - cx + 0.5f) / spx));
        float acc = 0.f;
        for (int i = ylo; i < yhi; ++i) {
            const float wy = fmaxf(0.f, 1.f - fabsf(((float)i - cy + 0.5f) / spy)) / wys;
            float row = 0.f;
            for (int j = xlo; j < xhi; ++j) {
                const float wx = fmaxf(0.f, 1.f - fabsf(((float)j - cx + 0.5f) / spx)) / wxs;
                row = fmaf(wx, aug_in(a, plane * a.H * a.W + (long long)i * a.W + j), row);
            }
            acc = fmaf(wy, row, acc);
        }
        aug_out(a, idx, acc, false);
    }
}

}  // namespace wmar

using namespace wmar;

extern "C" int wmar_augment(int32_t op, const float* in_dev, float* out_dev, const float* noise_dev, int64_t B, int32_t C, int32_t H,
                            int32_t W, int32_t pm1, double p0, double p1, void* stream) {
    WMAR_REQUIRE(in_dev && out_dev && B >= 1 && C >= 1 && H >= 1 && W >= 1, "augment: bad argument");
    WMAR_REQUIRE(in_dev != out_dev || op == AUG_NOISE || op == AUG_BRIGHTNESS || op == AUG_IDENTITY, "augment: this transform cannot run in place");
    AugArgs a{};
    a.in = in_dev; a.out = out_dev; a.noise = noise_dev; a.planes = (int)(B * C); a.H = H; a.W = W; a.pm1 = pm1 ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)a.planes * H * W;
    const dim3 pgrid((unsigned)((n + 255) / 256));
    switch (op) {
        case AUG_IDENTITY: hipLaunchKernelGGL(k_aug_point<AUG_IDENTITY>, pgrid, dim3(256), 0, st, a); break;
        case AUG_BLUR: {
            const int k = (int)p0;
            WMAR_REQUIRE(k >= 1 && k <= AUG_MAX_K && (k & 1), "augment: blur kernel size %d (odd, 1..%d)", k, AUG_MAX_K);
            WMAR_REQUIRE(k / 2 < H && k / 2 < W, "augment: blur kernel %d does not fit a %d x %d image (reflect padding)", k, H, W);
            // gaussian_blur's weights: x = -(k-1)/2 .. (k-1)/2, exp(-x^2 / 2 sigma^2), normalised (float, as torchvision computes them in the image dtype)
            const float sigma = 0.3f * ((float)(k - 1) * 0.5f - 1.f) + 0.8f;
            float sum = 0.f;
            for (int i = 0; i < k; ++i) {
                const float x = -(float)(k - 1) * 0.5f + (float)i;
                a.w[i] = expf(-0.5f * (x / sigma) * (x / sigma));
                sum += a.w[i];
            }
            for (int i = 0; i < k; ++i) a.w[i] /= sum;
            a.k = k;
            const int TW = 16 + k - 1;
            hipLaunchKernelGGL(k_aug_blur, dim3((unsigned)((W + 15) / 16), (unsigned)((H + 15) / 16), (unsigned)a.planes), dim3(256),
                               (size_t)TW * TW * sizeof(float), st, a);
            break;
        }
        case AUG_NOISE:
            WMAR_REQUIRE(noise_dev, "augment: the noise transform needs its standard normal draws");
            a.f = (float)p0;
            hipLaunchKernelGGL(k_aug_point<AUG_NOISE>, pgrid, dim3(256), 0, st, a);
            break;
        case AUG_BRIGHTNESS:
            a.f = (float)p0;
            hipLaunchKernelGGL(k_aug_point<AUG_BRIGHTNESS>, pgrid, dim3(256), 0, st, a);
            break;
        case AUG_ROTATE: {
            // p0: counter-clockwise quarter turns (0..3), p1: remainder angle in degrees, [0, 90)
            a.quarters = ((int)p0 % 4 + 4) % 4;
            WMAR_REQUIRE(H == W || a.quarters % 2 == 0, "augment: odd quarter turns need a square image");
            const double rad = p1 * 3.14159265358979323846 / 180.0;
            a.has_rest = p1 != 0.0 ? 1 : 0;
            a.cs = (float)cos(rad); a.sn = (float)sin(rad);
            hipLaunchKernelGGL(k_aug_point<AUG_ROTATE>, pgrid, dim3(256), 0, st, a);
            break;
        }
        case AUG_FLIP_H: hipLaunchKernelGGL(k_aug_point<AUG_FLIP_H>, pgrid, dim3(256), 0, st, a); break;
        case AUG_CROP_RESIZE:
        case AUG_CROP_PAD:
            a.nh = (int)p0; a.nw = (int)p1;
            WMAR_REQUIRE(a.nh >= 1 && a.nh <= H && a.nw >= 1 && a.nw <= W, "augment: crop %d x %d of a %d x %d image", a.nh, a.nw, H, W);
            if (op == AUG_CROP_RESIZE) hipLaunchKernelGGL(k_aug_point<AUG_CROP_RESIZE>, pgrid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL(k_aug_point<AUG_CROP_PAD>, pgrid, dim3(256), 0, st, a);
            break;
        default: set_error("augment: unknown transform %d", op); return WMAR_EINVAL;
    }
    return launch_status("k_aug");
}
