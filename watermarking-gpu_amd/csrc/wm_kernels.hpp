// wm_kernels.hpp -- host-visible launch interface of the HIP kernels (internal to libwm_hip.so)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wmk {

constexpr int NGRAM = 44;  // 36 unique Rx entries + 8 rx entries (me_p3.hpp:8-21)

struct PlaneDesc {
    const void* p;
    long long pitch;    // elements
    long long fstride;  // elements between frames
    long long cstride;  // elements between channels
    int dtype;          // 0 f32, 1 u8
    int channels;
    int aligned;        // base/pitch/strides allow 4-pixel vector access
};

struct LaunchGeom {
    int rows, cols;
    int nstrips, nfull, nsegs, rps;  // nfull = strips lying fully inside the image (cols / 256)
    int nblk;  // strip-march blocks per frame (all strips)
    int nbb;   // blocks per frame of k_gram_border
};

struct EmbedScalars {
    float a;     // watermark strength (Watermark.cpp:170)
    float maxe;  // max|e| (ME) or 1
};

struct OpResult {
    int status;   // 0 OK, 1 unsolvable
    float value;  // a (embed) or correlation (detect)
};

void launch_gram(hipStream_t s, const LaunchGeom& lg, int frames, const PlaneDesc& x, double* pmain, double* pborder);
void launch_solve(hipStream_t s, const LaunchGeom& lg, int frames, const double* pmain, const double* pborder, float* coef,
                  int* status, double* gram_tot);
void launch_me_stats(hipStream_t s, const LaunchGeom& lg, int frames, const PlaneDesc& x, const float* W, int aligned_w,
                     const float* coef, const int* status, float* pmax, double* pss);
void launch_nvf_stats(hipStream_t s, const LaunchGeom& lg, int frames, const PlaneDesc& x, const float* W, int aligned_w,
                      int pad, double* pss);
void launch_embed_scalars(hipStream_t s, const LaunchGeom& lg, int frames, const float* pmax, const double* pss,
                          const int* status, float sF, EmbedScalars* scal, OpResult* res);
void launch_embed(hipStream_t s, const LaunchGeom& lg, int frames, int mask, int pad, const PlaneDesc& x, const float* W,
                  int aligned_w, const PlaneDesc& base, const PlaneDesc& out, const float* coef, const int* status,
                  const EmbedScalars* scal);
void launch_mask(hipStream_t s, const LaunchGeom& lg, int frames, int mask, int pad, const PlaneDesc& x, const float* coef,
                 const int* status, const EmbedScalars* scal, const PlaneDesc& mo, const PlaneDesc& eo);
void launch_detect(hipStream_t s, const LaunchGeom& lg, int frames, int mask, int pad, const PlaneDesc& x, const float* W,
                   int aligned_w, const float* coef, const int* status, double* pcorr);
void launch_corr_finalize(hipStream_t s, const LaunchGeom& lg, int frames, const double* pcorr, const int* status,
                          OpResult* res);
void launch_mask_result(hipStream_t s, int frames, const int* status, const float* coef, OpResult* res, float* coef_out);

}  // namespace wmk
