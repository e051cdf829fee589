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

}  // namespace

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
