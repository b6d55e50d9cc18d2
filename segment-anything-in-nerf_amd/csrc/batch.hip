// batch.hip -- the batch builder in front of the hot path, on the device (gfx950):
//   snf_pixel_indices  : PixelSampler / PatchPixelSampler.sample_method (nerfstudio/data/pixel_samplers.py:50-75,246-300)
//   snf_generate_rays  : RayGenerator.forward + the pinhole branch of Cameras._generate_rays_from_coords
//                        (model_components/ray_generators.py:44-63, cameras/cameras.py:284-311,576-722)
//   snf_gather_nearest : FeatureDataloader.__call__ (samnerf/data/feature_loader.py:49-56), also the image gather of
//                        PixelSampler.collate_image_dataset_batch and the patch-centre pick of samnerf/datamanager.py:106-110
// The reference does this with ~15 small torch launches on (partly) host-resident data; here the images, feature maps and
// cameras stay in HBM and a batch is three launches.  Index arithmetic is reproduced operation by operation (fp32 products
// rounded before the add / the truncation: FMA contraction is off in this file).
#include "common.hpp"

#pragma clang fp contract(off)

namespace snf {

// u [B,3] (patch == 1) or [B/p^2,3] (patch > 1) -> indices [B,3] int64 (camera, row, col)
__global__ __launch_bounds__(256) void k_pixel_indices(const float* __restrict__ u, int B, int p, int num_images, int H, int W,
                                                       long long* __restrict__ indices) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= B) return;
    const int pp = p * p, patch = r / pp, q = r - patch * pp;
    const int yy = q / p, xx = q - yy * p;
    const float* up = u + (size_t)patch * 3;
    float fc, fy, fx;
    if (p > 1) {
        fc = up[0] * (float)num_images;
        fy = up[1] * (float)(H - p) + (float)yy;
        fx = up[2] * (float)(W - p) + (float)xx;
    } else {
        fc = up[0] * (float)num_images;
        fy = up[1] * (float)H;
        fx = up[2] * (float)W;
    }
    indices[(size_t)r * 3 + 0] = (long long)floorf(fc);
    indices[(size_t)r * 3 + 1] = (long long)floorf(fy);
    indices[(size_t)r * 3 + 2] = (long long)floorf(fx);
}

__device__ __forceinline__ void cam_dir(float cxn, float cyn, const float* __restrict__ R, float (&d)[3]) {
    // sum(d[None, :] * rotation, -1): world_i = sum_j cam_j * R[i][j], cam = (cxn, cyn, -1); torch sums left to right
    const float v[3] = {cxn, cyn, -1.0f};
    float n2 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        d[i] = (v[0] * R[i * 4 + 0] + v[1] * R[i * 4 + 1]) + v[2] * R[i * 4 + 2];
    }
    n2 = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
    const float nrm = fmaxf(sqrtf(n2), 8.8817841970012523e-16f);  // normalize_with_norm: max(norm, 4 * eps(float64))
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = d[i] / nrm;
}

// indices [R,3] int64; c2w [N,3,4]; intr [N,4] = fx, fy, cx, cy
__global__ __launch_bounds__(256) void k_generate_rays(const long long* __restrict__ indices, int R, const float* __restrict__ c2w,
                                                       const float* __restrict__ intr, int N, float* __restrict__ origins,
                                                       float* __restrict__ directions, float* __restrict__ pixel_area,
                                                       long long* __restrict__ camera_indices) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    long long c = indices[(size_t)r * 3];
    const float y = (float)indices[(size_t)r * 3 + 1] + 0.5f, x = (float)indices[(size_t)r * 3 + 2] + 0.5f;
    const long long cc = c < 0 ? 0 : (c >= N ? N - 1 : c);
    const float* M = c2w + (size_t)cc * 12;
    const float fx = intr[cc * 4 + 0], fy = intr[cc * 4 + 1], cx = intr[cc * 4 + 2], cy = intr[cc * 4 + 3];
    const float ax = (x - cx) / fx, ay = -(y - cy) / fy;
    const float bx = (x - cx + 1.f) / fx, by = -(y - cy + 1.f) / fy;
    float d0[3], d1[3], d2[3];
    cam_dir(ax, ay, M, d0);
    cam_dir(bx, ay, M, d1);
    cam_dir(ax, by, M, d2);
    float e1 = 0.f, e2 = 0.f;
    {
        const float a0 = d0[0] - d1[0], a1 = d0[1] - d1[1], a2 = d0[2] - d1[2];
        e1 = sqrtf((a0 * a0 + a1 * a1) + a2 * a2);
        const float b0 = d0[0] - d2[0], b1 = d0[1] - d2[1], b2 = d0[2] - d2[2];
        e2 = sqrtf((b0 * b0 + b1 * b1) + b2 * b2);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        origins[(size_t)r * 3 + i] = M[i * 4 + 3];
        directions[(size_t)r * 3 + i] = d0[i];
    }
    pixel_area[r] = e1 * e2;
    camera_indices[r] = c;
}

// out[b, :] = feat[cam, long(row * fh/H), long(col * fw/W), :] for point b = points[(b * stride + offset)]
// (stride = p^2, offset = (p/2)*p + p/2 picks the patch centres; stride 1, offset 0 every ray).  One wave per point.
__global__ __launch_bounds__(256) void k_gather_nearest(const long long* __restrict__ points, int B, int stride, int offset,
                                                        const float* __restrict__ feat, int N, int fh, int fw, int C,
                                                        float sy, float sx, float* __restrict__ out) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const long long* pt = points + ((size_t)b * stride + offset) * 3;
    const long long c = pt[0];
    // (int64 tensor * python float) is computed in fp32, then .long() truncates
    long long yi = (long long)((float)pt[1] * sy), xi = (long long)((float)pt[2] * sx);
    if (yi < 0) yi += fh;  // torch advanced indexing wraps negatives; out-of-range is the caller's error (clamped here)
    if (xi < 0) xi += fw;
    yi = yi < 0 ? 0 : (yi >= fh ? fh - 1 : yi);
    xi = xi < 0 ? 0 : (xi >= fw ? fw - 1 : xi);
    const long long cc = c < 0 ? 0 : (c >= N ? N - 1 : c);
    const float* src = feat + (((size_t)cc * fh + yi) * fw + xi) * C;
    float* dst = out + (size_t)b * C;
    for (int k = lane; k < C; k += 64) dst[k] = src[k];
}

}  // namespace snf

using namespace snf;

extern "C" int snf_pixel_indices(const float* u, int B, int patch, int num_images, int H, int W, int64_t* indices,
                                 snf_stream_t stream) {
    SNF_REQUIRE(u && indices && B > 0 && patch >= 1 && num_images >= 1, "snf_pixel_indices: bad argument");
    SNF_REQUIRE(B % (patch * patch) == 0, "snf_pixel_indices: B=%d is not a multiple of patch^2=%d", B, patch * patch);
    SNF_REQUIRE(H > patch && W > patch, "snf_pixel_indices: image %dx%d smaller than the patch", H, W);
    hipLaunchKernelGGL(k_pixel_indices, dim3(ceil_div(B, 256)), dim3(256), 0, (hipStream_t)stream, u, B, patch, num_images, H, W,
                       (long long*)indices);
    SNF_LAUNCH_CHECK("snf_pixel_indices");
    return SNF_OK;
}

extern "C" int snf_generate_rays(const int64_t* indices, int R, const float* c2w, const float* intrinsics, int num_cameras,
                                 float* origins, float* directions, float* pixel_area, int64_t* camera_indices,
                                 snf_stream_t stream) {
    SNF_REQUIRE(indices && c2w && intrinsics && origins && directions && pixel_area && camera_indices && R > 0 && num_cameras > 0,
                "snf_generate_rays: bad argument");
    hipLaunchKernelGGL(k_generate_rays, dim3(ceil_div(R, 256)), dim3(256), 0, (hipStream_t)stream, (const long long*)indices, R,
                       c2w, intrinsics, num_cameras, origins, directions, pixel_area, (long long*)camera_indices);
    SNF_LAUNCH_CHECK("snf_generate_rays");
    return SNF_OK;
}

extern "C" int snf_gather_nearest(const int64_t* points, int B, int point_stride, int point_offset, const float* features,
                                  int num_images, int fh, int fw, int C, int H, int W, float* out, snf_stream_t stream) {
    SNF_REQUIRE(points && features && out && B > 0 && point_stride >= 1 && point_offset >= 0 && point_offset < point_stride,
                "snf_gather_nearest: bad argument");
    SNF_REQUIRE(num_images > 0 && fh > 0 && fw > 0 && C > 0 && H > 0 && W > 0, "snf_gather_nearest: bad shape");
    // img_scale = (fh / H, fw / W) are python floats; the product with the int64 index tensor is an fp32 tensor op
    const float sy = (float)((double)fh / (double)H), sx = (float)((double)fw / (double)W);
    hipLaunchKernelGGL(k_gather_nearest, dim3(ceil_div(B, 4)), dim3(256), 0, (hipStream_t)stream, (const long long*)points, B,
                       point_stride, point_offset, features, num_images, fh, fw, C, sy, sx, out);
    SNF_LAUNCH_CHECK("snf_gather_nearest");
    return SNF_OK;
}
