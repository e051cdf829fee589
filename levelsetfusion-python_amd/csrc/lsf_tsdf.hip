// TSDF generation from a depth image, nearest pixel (reference tsdf/generation.py:130-207 for the
// 2-D row slice, :356-437 for the volume; tsdf/common.py:34-47).  One thread per voxel, independent gathers from a
// 600 KB depth image that lives in L2 -- the kernel is bound by its 4 B/voxel output stream.
// Arithmetic follows the reference's dtypes under numpy >= 2 (oracle.tsdf_nearest): voxel centre evaluated in float64
// and rounded to float32, float32 extrinsics product, projection in the intrinsic matrix's dtype, depth * ratio and
// the signed distance in float64, result stored as float32.
#include "lsf_device.h"

using namespace lsf;

namespace {

struct TsdfParams {
    double fx, fy, cx, cy;
    double depth_unit_ratio, voxel_size, half_width;
    float e[12];  // first three rows of the 4x4 extrinsic matrix, row-major
    int off[3];
    int width, height, image_y;
    float default_value;
};

__device__ inline double tsdf_value(double sd, double half) {
    return sd < -half ? -1.0 : (sd > half ? 1.0 : sd / half);
}

template <typename P>
__device__ inline long long project(P f, float pc, P z, P c) {
    P v = ((f * (P)pc) / z + c) + (P)0.5;
    // int() of the reference truncates toward zero; saturate so that wild values stay out of range
    if (!(v > (P)-2147483000.0 && v < (P)2147483000.0)) return -1;
    return (long long)v;
}

template <int D, typename P>
__global__ __launch_bounds__(kBlock) void tsdf_nearest_kernel(const unsigned short* __restrict__ depth,
                                                              float* __restrict__ field, Grid g, TsdfParams p) {
    for_each_voxel(g, [&](int x, int y, int z) {
        const int i = vidx(g, x, y, z);
        float result = p.default_value;
        // 2-D: the field's y index is the depth axis and y_voxel = 0 (generation.py:176-180)
        const float xv = (float)((double)(x + p.off[0]) * p.voxel_size);
        const float yv = D == 3 ? (float)((double)(y + p.off[1]) * p.voxel_size) : 0.0f;
        const float zv = (float)((double)((D == 3 ? z : y) + p.off[2]) * p.voxel_size);
        const float pcx = ((p.e[0] * xv + p.e[1] * yv) + p.e[2] * zv) + p.e[3] * 1.0f;
        const float pcy = ((p.e[4] * xv + p.e[5] * yv) + p.e[6] * zv) + p.e[7] * 1.0f;
        const float pcz = ((p.e[8] * xv + p.e[9] * yv) + p.e[10] * zv) + p.e[11] * 1.0f;
        if (pcz > 0.0f) {
            const long long ix = project<P>((P)p.fx, pcx, (P)pcz, (P)p.cx);
            const long long iy = D == 3 ? project<P>((P)p.fy, pcy, (P)pcz, (P)p.cy) : (long long)p.image_y;
            if (ix >= 0 && ix < p.width && iy >= 0 && iy < p.height) {
                const double d = (double)depth[iy * p.width + ix] * p.depth_unit_ratio;
                if (d > 0.0) {
                    const double sd = d - (double)pcz;
                    result = sd < -p.half_width ? -1.0f : (sd > p.half_width ? 1.0f : (float)(sd / p.half_width));
                }
            }
        }
        field[i] = result;
    });
}

// ---------------------------------------------------------------------------------------------------------
// The two bilinear 2-D variants (tsdf/generation.py:18-75 "bilinear TSDF space", :78-128 "bilinear image space";
// utils/sampling.py:110-175).  image_y_coordinate is an integer, so the reference's y ratio is exactly 0 and both
// reduce to a blend of the taps (floor x, row) and (floor x + 1, row) along x:
//   image space: blend the raw depth values (out-of-image tap = 1 depth unit, utils/sampling.py:35-55), skip voxels
//                whose projection leaves the image and non-positive depths, then one TSDF value;
//   TSDF space:  a TSDF value per tap (out-of-image tap = 1.0), then blend.
// dtypes as numpy >= 2 evaluates the reference: ratio and 1 - ratio in the intrinsic matrix's dtype, products and sum
// in float64, result stored float32 (oracle.tsdf_bilinear).
template <typename P, int MODE>
__global__ __launch_bounds__(kBlock) void tsdf_bilinear_kernel(const unsigned short* __restrict__ depth,
                                                               float* __restrict__ field, Grid g, TsdfParams p) {
    for_each_voxel(g, [&](int x, int y, int z) {
        const int i = vidx(g, x, y, z);
        float result = p.default_value;
        const float xv = (float)((double)(x + p.off[0]) * p.voxel_size);
        const float zv = (float)((double)(y + p.off[2]) * p.voxel_size);
        const float pcx = ((p.e[0] * xv + p.e[1] * 0.0f) + p.e[2] * zv) + p.e[3] * 1.0f;
        const float pcz = ((p.e[8] * xv + p.e[9] * 0.0f) + p.e[10] * zv) + p.e[11] * 1.0f;
        if (pcz > 0.0f) {
            const P ix = ((P)p.fx * (P)pcx) / (P)pcz + (P)p.cx;
            const bool inside = ix >= (P)0 && ix < (P)p.width;
            if (MODE == 2 || inside) {
                const P fl = floor(ix);
                const P ratio = ix - fl, inverse = (P)1 - ratio;
                // saturate: a wild projection is out of the image anyway
                const long long bx = fl > (P)-2147483000.0 && fl < (P)2147483000.0 ? (long long)fl : -2;
                const unsigned short* row = depth + (long long)p.image_y * p.width;
                const bool in0 = bx >= 0 && bx < p.width, in1 = bx + 1 >= 0 && bx + 1 < p.width;
                const double d0 = in0 ? (double)row[bx] : 1.0, d1 = in1 ? (double)row[bx + 1] : 1.0;
                if (MODE == 1) {
                    const double d = (d0 * (double)inverse + d1 * (double)ratio) * p.depth_unit_ratio;
                    if (d > 0.0) result = (float)tsdf_value(d - (double)pcz, p.half_width);
                } else {
                    const double t0 = in0 ? tsdf_value(d0 * p.depth_unit_ratio - (double)pcz, p.half_width) : 1.0;
                    const double t1 = in1 ? tsdf_value(d1 * p.depth_unit_ratio - (double)pcz, p.half_width) : 1.0;
                    result = (float)(t0 * (double)inverse + t1 * (double)ratio);
                }
            }
        }
        field[i] = result;
    });
}

// ---------------------------------------------------------------------------------------------------------
// EWA filters (tsdf/ewa.py:59-184 3-D image space; :230-353, :358-481, :485-624 2-D image space / voxel space /
// voxel space inclusive; math_utils/elliptical_gaussians.py:27-62,145-158).  Per voxel: project a spherical Gaussian
// through the projection Jacobian into an image-space ellipse (+ unit pixel Gaussian), then average depth (image
// space) or per-pixel TSDF values (voxel space) over the pixels inside the ellipse.  float32 voxel / camera / image
// coordinates and Jacobian entries, everything else float64 -- as the reference evaluates it under numpy >= 2.
// Windows are a handful of pixels; the kernel is bound by its output stream like the nearest-pixel one.
struct EwaParams {
    double cov[9];     // covariance of the voxel sphere in camera space: R * (scale*voxel_size*I) * R^T
    double threshold;  // squared radius threshold F = 4 * scale * voxel_size
    float k[9];        // intrinsic matrix, row-major
    int method;        // 3 image space, 4 voxel space, 5 voxel space inclusive (FilteringMethod values)
};

template <int D>
__global__ __launch_bounds__(kBlock) void tsdf_ewa_kernel(const unsigned short* __restrict__ depth,
                                                          float* __restrict__ field, Grid g, TsdfParams p,
                                                          EwaParams q) {
    for_each_voxel(g, [&](int x, int y, int z) {
        const int i = vidx(g, x, y, z);
        // 2-D: field[y][x], depth axis = y index.  3-D: array axis 0 (z here) is world x, axis 2 (x here) is the
        // depth axis -- the reference's deliberate flip (tsdf/ewa.py:115-119)
        const float xv = (float)((double)((D == 3 ? z : x) + p.off[0]) * p.voxel_size);
        const float yv = D == 3 ? (float)((double)(y + p.off[1]) * p.voxel_size) : 0.0f;
        const float zv = (float)((double)((D == 3 ? x : y) + p.off[2]) * p.voxel_size);
        const float vc0 = ((p.e[0] * xv + p.e[1] * yv) + p.e[2] * zv) + p.e[3] * 1.0f;
        const float vc1 = ((p.e[4] * xv + p.e[5] * yv) + p.e[6] * zv) + p.e[7] * 1.0f;
        const float vc2 = ((p.e[8] * xv + p.e[9] * yv) + p.e[10] * zv) + p.e[11] * 1.0f;
        float result = p.default_value;
        if (vc2 > 0.05f) {  // near clipping distance, tsdf/ewa.py:29
            float vi0 = ((q.k[0] * vc0 + q.k[1] * vc1) + q.k[2] * vc2) / vc2;
            float vi1 = ((q.k[3] * vc0 + q.k[4] * vc1) + q.k[5] * vc2) / vc2;
            if (D == 2) vi1 = (float)p.image_y;
            bool ok = true;
            if (q.method == 5)
                ok = !(vi1 < -3.0f || vi1 >= (float)(p.height + 3) || vi0 < -3.0f || vi0 >= (float)(p.width + 3));
            if (ok) {
                const float ray = sqrtf((vc0 * vc0 + vc1 * vc1) + vc2 * vc2);
                const float z2 = vc2 * vc2;
                // rows 0 and 1 of the projection Jacobian (float32 entries), row 2 does not reach the 2x2 block
                const double j0[3] = {(double)(1.0f / vc2), 0.0, (double)(-vc0 / z2)};
                const double j1[3] = {0.0, (double)(1.0f / vc2), (double)(-vc1 / z2)};
                (void)ray;
                double t0[3], t1[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    t0[c] = (j0[0] * q.cov[c] + j0[1] * q.cov[3 + c]) + j0[2] * q.cov[6 + c];
                    t1[c] = (j1[0] * q.cov[c] + j1[1] * q.cov[3 + c]) + j1[2] * q.cov[6 + c];
                }
                const double m00 = (t0[0] * j0[0] + t0[1] * j0[1]) + t0[2] * j0[2];
                const double m01 = (t0[0] * j1[0] + t0[1] * j1[1]) + t0[2] * j1[2];
                const double m10 = (t1[0] * j0[0] + t1[1] * j0[1]) + t1[2] * j0[2];
                const double m11 = (t1[0] * j1[0] + t1[1] * j1[1]) + t1[2] * j1[2];
                const double s00 = q.k[0], s01 = q.k[1], s10 = q.k[3], s11 = q.k[4];
                // final covariance = S * M * S^T + I
                const double a00 = s00 * m00 + s01 * m10, a01 = s00 * m01 + s01 * m11;
                const double a10 = s10 * m00 + s11 * m10, a11 = s10 * m01 + s11 * m11;
                const double f00 = (a00 * s00 + a01 * s01) + 1.0, f01 = a00 * s10 + a01 * s11;
                const double f10 = a10 * s00 + a11 * s01, f11 = (a10 * s10 + a11 * s11) + 1.0;
                const double det = f00 * f11 - f01 * f10;
                const double q00 = f11 / det, q01 = -f01 / det, q10 = -f10 / det, q11 = f00 / det;
                // ellipse bounds (elliptical_gaussians.py:44-58)
                const double A = q00, B = q01 * 2.0, C = q11, F = q.threshold;
                double bx, by;
                if (fabs(B) < 10e-6) {
                    bx = sqrt(F / A);
                    by = sqrt(F / C);
                } else {
                    bx = sqrt(F / (C - B * B / (4.0 * A)));
                    by = sqrt(F / (A - B * B / (4.0 * C)));
                }
                const double vx = (double)vi0, vy = (double)vi1;
                int sx = (int)(vx - bx), ex = (int)ceil(vx + bx + 1.0);
                int sy = (int)(vy - by), ey = (int)ceil(vy + by + 1.0);
                if (!(ey <= 0 || sy >= p.height || ex <= 0 || sx >= p.width)) {
                    if (q.method != 5) {
                        sy = max(0, sy); ey = min(p.height, ey);
                        sx = max(0, sx); ex = min(p.width, ex);
                    }
                    double weights = 0.0, values = 0.0;
                    for (int ys = sy; ys < ey; ++ys)
                        for (int xs = sx; xs < ex; ++xs) {
                            const double p0 = (double)xs - vx, p1 = (double)ys - vy;
                            const double d2 = (p0 * q00 + p1 * q10) * p0 + (p0 * q01 + p1 * q11) * p1;
                            if (d2 > F) continue;
                            const double wgt = exp(-0.5 * d2);
                            if (ys < 0 || ys >= p.height || xs < 0 || xs >= p.width) {  // inclusive variant only
                                values += wgt * 1.0;
                                weights += wgt;
                                continue;
                            }
                            const double surface = (double)depth[ys * p.width + xs] * p.depth_unit_ratio;
                            if (surface <= 0.0) continue;
                            values += q.method == 3 ? wgt * surface
                                                    : wgt * tsdf_value(surface - (double)vc2, p.half_width);
                            weights += wgt;
                        }
                    if (q.method == 3) {
                        if (values > 0.0) result = (float)tsdf_value(values / weights - (double)vc2, p.half_width);
                    } else if (weights != 0.0) {
                        result = (float)(values / weights);
                    }
                }
            }
        }
        field[i] = result;
    });
}

TsdfParams convert_params(const lsf_tsdf_params* params) {
    TsdfParams p;
    p.fx = params->intrinsics[0]; p.fy = params->intrinsics[1]; p.cx = params->intrinsics[2]; p.cy = params->intrinsics[3];
    p.depth_unit_ratio = params->depth_unit_ratio;
    p.voxel_size = params->voxel_size;
    p.half_width = params->narrow_band_half_width;
    for (int k = 0; k < 12; ++k) p.e[k] = params->extrinsic[k];
    for (int k = 0; k < 3; ++k) p.off[k] = params->array_offset[k];
    p.width = params->image_width; p.height = params->image_height; p.image_y = params->image_y_coordinate;
    p.default_value = params->default_value;
    return p;
}

}  // namespace

extern "C" int lsf_tsdf_generate_ewa(const uint16_t* depth_image, float* field, const lsf_grid* grid,
                                     const lsf_tsdf_params* params, const lsf_ewa_params* ewa, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!depth_image || !field || !params || !ewa) return LSF_ERR_BAD_ARGUMENT;
    if (params->image_width <= 0 || params->image_height <= 0 || !(params->narrow_band_half_width > 0.0))
        return LSF_ERR_BAD_ARGUMENT;
    if (ewa->method < 3 || ewa->method > 5 || !(ewa->squared_radius_threshold > 0.0)) return LSF_ERR_BAD_ARGUMENT;
    if (grid->dims == 2 && (params->image_y_coordinate < 0 || params->image_y_coordinate >= params->image_height))
        return LSF_ERR_BAD_ARGUMENT;
    Grid g = make_grid(grid);
    Tiling t = make_tiling(g);
    if (t.total == 0) return 0;
    TsdfParams p = convert_params(params);
    EwaParams q;
    for (int k = 0; k < 9; ++k) { q.cov[k] = ewa->covariance_camera_space[k]; q.k[k] = ewa->intrinsic_matrix[k]; }
    q.threshold = ewa->squared_radius_threshold;
    q.method = ewa->method;
    const unsigned blocks = launch_blocks(t.total);
    const unsigned short* d = reinterpret_cast<const unsigned short*>(depth_image);
    if (grid->dims == 2)
        hipLaunchKernelGGL(tsdf_ewa_kernel<2>, dim3(blocks), dim3(kBlock), 0, as_stream(stream), d, field, g, p, q);
    else
        hipLaunchKernelGGL(tsdf_ewa_kernel<3>, dim3(blocks), dim3(kBlock), 0, as_stream(stream), d, field, g, p, q);
    return launch_status();
}

extern "C" int lsf_tsdf_generate_bilinear(const uint16_t* depth_image, float* field, const lsf_grid* grid,
                                          const lsf_tsdf_params* params, int32_t method, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!depth_image || !field || !params || grid->dims != 2 || (method != 1 && method != 2)) return LSF_ERR_BAD_ARGUMENT;
    if (params->image_width <= 0 || params->image_height <= 0 || !(params->narrow_band_half_width > 0.0) ||
        params->image_y_coordinate < 0 || params->image_y_coordinate >= params->image_height)
        return LSF_ERR_BAD_ARGUMENT;
    Grid g = make_grid(grid);
    Tiling t = make_tiling(g);
    if (t.total == 0) return 0;
    const TsdfParams p = convert_params(params);
    const unsigned blocks = launch_blocks(t.total);
    hipStream_t s = as_stream(stream);
    const unsigned short* d = reinterpret_cast<const unsigned short*>(depth_image);
    if (params->intrinsics_are_f32) {
        if (method == 1) hipLaunchKernelGGL((tsdf_bilinear_kernel<float, 1>), dim3(blocks), dim3(kBlock), 0, s, d, field, g, p);
        else hipLaunchKernelGGL((tsdf_bilinear_kernel<float, 2>), dim3(blocks), dim3(kBlock), 0, s, d, field, g, p);
    } else {
        if (method == 1) hipLaunchKernelGGL((tsdf_bilinear_kernel<double, 1>), dim3(blocks), dim3(kBlock), 0, s, d, field, g, p);
        else hipLaunchKernelGGL((tsdf_bilinear_kernel<double, 2>), dim3(blocks), dim3(kBlock), 0, s, d, field, g, p);
    }
    return launch_status();
}

extern "C" int lsf_tsdf_generate_nearest(const uint16_t* depth_image, float* field, const lsf_grid* grid,
                                         const lsf_tsdf_params* params, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!depth_image || !field || !params) return LSF_ERR_BAD_ARGUMENT;
    if (params->image_width <= 0 || params->image_height <= 0 || !(params->narrow_band_half_width > 0.0))
        return LSF_ERR_BAD_ARGUMENT;
    if (grid->dims == 2 && (params->image_y_coordinate < 0 || params->image_y_coordinate >= params->image_height))
        return LSF_ERR_BAD_ARGUMENT;
    Grid g = make_grid(grid);
    Tiling t = make_tiling(g);
    if (t.total == 0) return 0;
    TsdfParams p;
    p.fx = params->intrinsics[0]; p.fy = params->intrinsics[1]; p.cx = params->intrinsics[2]; p.cy = params->intrinsics[3];
    p.depth_unit_ratio = params->depth_unit_ratio;
    p.voxel_size = params->voxel_size;
    p.half_width = params->narrow_band_half_width;
    for (int k = 0; k < 12; ++k) p.e[k] = params->extrinsic[k];
    for (int k = 0; k < 3; ++k) p.off[k] = params->array_offset[k];
    p.width = params->image_width; p.height = params->image_height; p.image_y = params->image_y_coordinate;
    p.default_value = params->default_value;
    const unsigned blocks = launch_blocks(t.total);
    hipStream_t s = as_stream(stream);
    const unsigned short* d = reinterpret_cast<const unsigned short*>(depth_image);
    if (grid->dims == 2) {
        if (params->intrinsics_are_f32)
            hipLaunchKernelGGL((tsdf_nearest_kernel<2, float>), dim3(blocks), dim3(kBlock), 0, s, d, field, g, p);
        else
            hipLaunchKernelGGL((tsdf_nearest_kernel<2, double>), dim3(blocks), dim3(kBlock), 0, s, d, field, g, p);
    } else {
        if (params->intrinsics_are_f32)
            hipLaunchKernelGGL((tsdf_nearest_kernel<3, float>), dim3(blocks), dim3(kBlock), 0, s, d, field, g, p);
        else
            hipLaunchKernelGGL((tsdf_nearest_kernel<3, double>), dim3(blocks), dim3(kBlock), 0, s, d, field, g, p);
    }
    return launch_status();
}
