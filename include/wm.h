/*
 * wm.h -- C ABI of the MI355X-native watermark engine (libwm_hip.so).
 *
 * This is the drop-in boundary for the reference's `Watermark` class hot path
 * (kar-dim/Watermarking-GPU, Watermark_GPU/Watermark.hpp:26-72, Watermark.cpp:21-258):
 * NVF mask, 3x3 prediction-error (ME) mask, PSNR-scaled embed, correlation detector.
 * The reference has no FFI of its own (its L3 class calls ArrayFire/OpenCL directly);
 * each entry point below names the reference member it replaces.  The C++ surface that keeps
 * the reference's class/method names lives in include/Watermark.hpp and is a thin wrapper
 * over exactly these functions.
 *
 * Conventions
 *  - planes are row-major: x(r,c) = data[r*pitch + c]  (reference frames: main.cpp:355,379,405;
 *    W file: Watermark.cpp:62-75, W(r,c) = file[r*cols + c]); borders replicate (clamp-to-edge,
 *    nvf.hpp:9, me_p3.hpp:45, scaled_neighbors_p3.hpp:14).
 *  - all plane pointers are DEVICE pointers unless mem == WM_MEM_HOST (then the library
 *    stages through its own device buffers with hipMemcpy2DAsync on the slot's stream;
 *    pinned host memory from wm_host_alloc() makes those copies truly asynchronous).
 *  - a plane may describe a batch: `frames` planes `frame_stride` elements apart; one call
 *    then processes all frames with one launch per kernel (frames are independent units,
 *    main.cpp:326-331).
 *  - every call is an ENQUEUE on the slot's HIP stream; scalar results (a, correlation, status)
 *    are delivered to the caller's pointers by wm_sync(ctx, slot).  Passing slot = WM_SLOT_SYNC
 *    runs on slot 0 and synchronises before returning.
 *  - status codes: 0 OK; 1 WM_UNSOLVABLE (not an error: embed leaves out == base bit-exact and
 *    `a` untouched, detect yields 0.0f -- Watermark.cpp:164-165,246-247); < 0 errors, which the
 *    C++ wrapper turns into std::runtime_error like the reference (Watermark.cpp:24-25,65-66,70-71,
 *    111-113).
 *  - a ctx is thread-compatible, not thread-safe (the reference object is not re-entrant either:
 *    its const methods mutate the shared texture, Watermark.cpp:88-93,223).  Concurrency is by
 *    slot: each slot owns a stream and private scratch.
 */
#ifndef WM_H_
#define WM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WM_OK 0
#define WM_UNSOLVABLE 1
#define WM_ERR_BAD_P (-1)      /* p not in {3,5,7,9} (Watermark.cpp:24-25); ME needs p == 3 (main.cpp:89) */
#define WM_ERR_W_OPEN (-2)     /* W file cannot be opened (Watermark.cpp:65-66) */
#define WM_ERR_W_SIZE (-3)     /* W file size != rows*cols*4 (Watermark.cpp:70-71) */
#define WM_ERR_RUNTIME (-4)    /* HIP runtime / kernel failure (Watermark.cpp:111-113,133-135,194-196) */
#define WM_ERR_BAD_ARG (-5)    /* null pointer, shape mismatch, bad slot, unsupported layout */
#define WM_ERR_NO_DEVICE (-6)  /* no usable HIP device: the product path has no CPU fallback */
#define WM_ERR_ALLOC (-7)
#define WM_ERR_PSNR (-8)       /* psnr <= 0 (main.cpp:96) */
#define WM_ERR_BUSY (-9)       /* too many un-synced operations queued on one slot */

#define WM_SLOT_SYNC (-1)

typedef struct wm_ctx wm_ctx;

/* enum MASK_TYPE { ME, NVF }  (Watermark.hpp:10-14) -- same order and values */
typedef enum wm_mask_type { WM_MASK_ME = 0, WM_MASK_NVF = 1 } wm_mask_type;
typedef enum wm_dtype { WM_F32 = 0, WM_U8 = 1 } wm_dtype;
/* WM_MEM_SLOT_OUT (input planes only, `data` ignored): the device copy of what the last wm_embed on the same slot wrote
 * (grey output; same frames / dtype).  A streamed frame staged from host memory is then detected without crossing the host
 * link a second time: wm_embed(host in, host out, slot) ; wm_detect(SLOT_OUT plane, slot).  Valid until the slot's next embed. */
typedef enum wm_mem { WM_MEM_DEVICE = 0, WM_MEM_HOST = 1, WM_MEM_SLOT_OUT = 2 } wm_mem;

/* Stand-in for the af::array arguments of makeWatermark/detectWatermark (Watermark.hpp:69-70):
 * a non-owning view.  channels == 1 (grey) or 3 (planar RGB: [3][rows][pitch], main.cpp:169-190).
 * Any base address, pitch and width is accepted.  Speed: device planes of f32 elements whose base is 4-byte aligned, and of
 * u8 elements whose base, pitch and strides are multiples of 4 bytes, take the kernels' vector path (planes below 4 GiB);
 * a width that is not a multiple of 4 costs 10-17 % (one generic strip at the right edge), anything else re-lays every row
 * through LDS (~25 % slower).  WM_MEM_HOST planes are de-pitched into an aligned staging buffer. */
typedef struct wm_plane {
    void* data;
    int32_t rows, cols;
    int32_t channels;
    int32_t dtype;           /* wm_dtype: f32 in [0,255], or u8 (video Y plane, main.cpp:355-357) */
    int32_t mem;             /* wm_mem */
    int32_t frames;          /* >= 1 */
    int64_t pitch;           /* elements between rows (>= cols) */
    int64_t channel_stride;  /* elements between channel planes (ignored when channels == 1) */
    int64_t frame_stride;    /* elements between frames (ignored when frames == 1) */
} wm_plane;

/* Watermark::Watermark(rows, cols, randomMatrixPath, p, psnr, programs)  (Watermark.hpp:63, Watermark.cpp:21-27).
 * `w_rowmajor` is the host copy of the W file contents (rows*cols f32).  `device` replaces
 * settings.ini's opencl_device (main.cpp:73) as a HIP device ordinal. */
int wm_create(wm_ctx** out, int device, int rows, int cols, int p, float psnr, const float* w_rowmajor);
/* same, reading the raw f32 file itself: loadRandomMatrix (Watermark.cpp:62-75) */
int wm_create_from_file(wm_ctx** out, int device, int rows, int cols, int p, float psnr, const char* w_path);
/* same, with W generated ON the device from a seed: the counter-based N(0,1) generator of csrc/app/wm_genw.cpp (this build's
 * CommonRandomMatrix, CommonRandomMatrix/main.cpp:16-68) -- element (r,c) depends on (seed, r, c) only, so the matrix equals
 * the file `wm_genw rows cols seed file` writes (to the last ulp of the device's f64 log / cos) and every GPU of a node
 * fills its own copy without a file, an upload or a broadcast.  wm_w_device() + wm_memcpy_d2h() read it back. */
int wm_create_generated(wm_ctx** out, int device, int rows, int cols, int p, float psnr, uint32_t seed);
/* Watermark(const Watermark&) / operator= (Watermark.cpp:30-51): shares W, owns new scratch */
int wm_clone(const wm_ctx* src, wm_ctx** out);
/* Watermark::reinitialize(path, rows, cols)  (Watermark.cpp:78-85) */
int wm_reinit(wm_ctx* ctx, int rows, int cols, const float* w_rowmajor);
int wm_reinit_from_file(wm_ctx* ctx, int rows, int cols, const char* w_path);
void wm_destroy(wm_ctx* ctx);

/* number of slots (streams + scratch) and the largest `frames` a call may carry; default 2 x 1 */
int wm_configure(wm_ctx* ctx, int nslots, int max_frames);
/* One image per synchronous call (slot = WM_SLOT_SYNC, frames == 1: what makeWatermark / detectWatermark are in the
 * reference, Watermark.cpp:156-172,234-250) runs as ONE launch whose tiles stay in LDS when the shape allows it (p = 3,
 * cols >= 256 -- any width for f32 planes, a multiple of 4 for u8 planes --, rows/cols small enough for one 256 x <=128 tile
 * per CU: up to 3840x2160 on MI355X; aligned planes); everything else, and every batched / asynchronous call, takes the batched sweeps.  mode 0 switches
 * the fused kernels off, 1 (default; environment WM_FUSED=0 changes the default) on.  Results of the two paths agree to
 * the rounding of the partial sums' grouping (tests/test_gpu_fused.py). */
int wm_set_fused(wm_ctx* ctx, int mode);
/* Gram hand-over from wm_embed to a detector that reads its output (opt-in; default off).  The detector's first sweep --
 * the Gram matrix of the watermarked plane, Watermark.cpp:234-250 through computePredictionErrorMask -- reads a plane that
 * k_embed has just produced in registers.  With the hand-over on, a batched wm_embed (two frames or more) of grey f32 planes on the
 * aligned path (p = 3) also accumulates the lag sums of its output that stay inside each wavefront's tile, and wm_detect /
 * wm_gram on a WM_MEM_SLOT_OUT plane (the slot's last embed output, by contract unmodified since) then only adds the
 * products across tile seams, the border frame and the solve: one of the five sweeps of an embed + detect pair is not run.
 * The 44 sums are the same exact products in another f64 summation order (agreement ~1e-16 relative, tests/test_gpu_handover.py);
 * any other detector input, dtype or shape takes the ordinary Gram sweep.  Costs ~19 MB of device memory per slot at 4K.
 *
 * HAZARD.  The hand-over is only as good as the caller's promise that the output plane is UNMODIFIED between the embed and the
 * detector that names it as WM_MEM_SLOT_OUT.  The library ends a hand-over whenever it writes that plane itself (the slot's
 * next embed, an embed on another slot into the same buffer, wm_band_embed, wm_compute_mask outputs, wm_band_configure), but
 * it cannot see a write by the caller -- a kernel on another stream, a copy, the host through mapped memory.  After such a
 * write the detector would use the Gram matrix of the OLD plane with the pixels of the NEW one: a wrong correlation, no
 * error.  Environment WM_HANDOVER_VERIFY=1 (read when the context is created; a debug mode, it synchronises every call)
 * makes every handed-over wm_detect / wm_gram also run the ordinary Gram sweep over the plane as it is and compare the 44
 * totals to 1e-12 relative: a mismatch fails the call with WM_ERR_RUNTIME and a message naming frame and term. */
int wm_set_handover(wm_ctx* ctx, int on);
/* returns 1 if synchronous one-frame calls of this context take the fused kernels; workgroups / tile_rows describe the
 * tiling, fallbacks counts fused launches that timed out in a hand-off and were re-run on the sweeps (any may be NULL) */
int wm_fused_info(const wm_ctx* ctx, int* workgroups, int* tile_rows, unsigned long long* fallbacks);
/* Fused launches need the device to themselves; processes that share a device serialise them with an advisory lock on a
 * per-device file (named after the PCI address, in $WM_FUSED_LOCK_DIR, else /run/lock, else $TMPDIR or /tmp; opened
 * read-only and never through a symbolic link).  The lock is tried for 5 ms; a call that does not get it (a holder that is
 * stopped in a debugger must not hang everybody else) runs on the sweeps.  Returns how often that happened. */
unsigned long long wm_fused_lock_skips(const wm_ctx* ctx);
/* development aid: with WM_FUSED_STAMPS set in the environment when the context is created, every workgroup of a fused
 * launch records up to 16 time stamps (100 MHz clock) at its phase boundaries; copies up to `cap` of the last call's
 * [workgroups + 1][16] values of slot 0 to `out`, returns the count (0 when stamps are off) */
int wm_fused_stamps(wm_ctx* ctx, unsigned long long* out, int cap);
/* Self-test of the NVF quotient (nvf.hpp:50, `variance / (1 + variance)`): the kernels form it as reciprocal, product and one
 * residual correction (4 operations) instead of the IEEE division sequence.  The divisor is a function of the dividend, so
 * the inputs are a one-parameter family, and this entry checks it exhaustively: `variant` 2 (what the kernels use; 0 and 1 =
 * the 8- and 6-operation sequences with a refined reciprocal, 3 = product alone, which is NOT exact and shows the test can
 * fail) against the compiler's correctly rounded division for every f32 whose bit pattern lies in [bits_lo, bits_hi),
 * counting the values whose results differ in any bit.  The mask can only produce variances in [-0.5, 2^17): bits
 * [0, 0x48000000) and (0x80000000, 0xBF000000) -- 2.2e9 values, well under a second on the device.
 * Returns WM_OK; *mismatches = differing values, *first_bad = the smallest differing bit pattern (if any). */
int wm_selftest_nvf_quotient(int device, int variant, uint32_t bits_lo, uint32_t bits_hi, unsigned long long* mismatches, uint32_t* first_bad);
/* with WM_FUSED_STAMPS set: the 44 Gram sums (wm_gram's order) the last fused ME call of slot 0 folded; returns 44 or 0 */
int wm_fused_gram(wm_ctx* ctx, double* out44);
/* rows each wavefront marches per segment (tuning knob; 0 = automatic) */
int wm_set_rows_per_segment(wm_ctx* ctx, int rows_per_segment);

/* Watermark::makeWatermark(inputImage, outputImage, watermarkStrength, maskType)  (Watermark.cpp:156-172).
 * in_gray: the mask source ([rows,cols], 1 channel); base: what the watermark is added to (1 or 3
 * channels, same rows/cols/dtype family); out: same shape as base, may alias base.  If out also overlaps
 * in_gray (in-place video frames, main.cpp:356,380) the library snapshots in_gray first (one extra copy).
 * a_out[frames], status_out[frames] (either may be NULL) are written by wm_sync. */
int wm_embed(wm_ctx* ctx, int mask, const wm_plane* in_gray, const wm_plane* base, const wm_plane* out, float* a_out,
             int* status_out, int slot);
/* Watermark::detectWatermark(watermarkedImage, maskType)  (Watermark.cpp:234-250) */
int wm_detect(wm_ctx* ctx, int mask, const wm_plane* img, float* corr_out, int* status_out, int slot);

/* makeWatermark followed by detectWatermark on its result -- the pair the reference's sample protocol runs per image
 * (testForImage, main.cpp:165-220) -- as ONE call: the results of wm_embed(...) then wm_detect(out, ...), delivered
 * together (on the fused kernels bit for bit; on the sweeps the embed hands the lag sums of its output to the detector as
 * under wm_set_handover, so y and the strength are bit-identical and the score agrees to the rounding of the Gram sums'
 * grouping, <= 2e-7).  Grey output only (out->channels == 1); the detector reads the device copy of the plane the embed wrote
 * (WM_MEM_SLOT_OUT), so a host-staged frame crosses the host link once each way.  Synchronous one-image calls on the fused
 * kernels launch both operations back to back and wait once (one launch-to-completion round trip less than two calls;
 * environment WM_FUSED_PAIR=1: both halves in ONE launch, the same bits, measured 1.5-2 us slower at 4K -- DESIGN.md section 8);
 * everything else queues the two operations on the slot.  status_out[frames] (may be NULL): the embed's status. */
int wm_embed_detect(wm_ctx* ctx, int mask, const wm_plane* in_gray, const wm_plane* base, const wm_plane* out, float* a_out,
                    float* corr_out, int* status_out, int slot);

/* Building blocks exposed for parity tests (the reference keeps them private):
 * computeCustomMask / computePredictionErrorMask (Watermark.cpp:96-114,176-218).
 * mask_out / e_out: f32 device planes [rows,cols] (e_out may be NULL; ignored for NVF).
 * coef_out[8*frames] (may be NULL) receives the prediction coefficients at wm_sync. */
int wm_compute_mask(wm_ctx* ctx, int mask, const wm_plane* in_gray, const wm_plane* mask_out, const wm_plane* e_out,
                    float* coef_out, int* status_out, int slot);

/* Gram sums of the 3x3 neighbourhood (me kernel + af::sum folding, me_p3.hpp:8-21,61-82, Watermark.cpp:140-151):
 * gram_out[44*frames] doubles = the 36 upper-triangle Rx entries (i <= j, row-major) then the 8 rx entries.
 * Synchronous (parity-test building block). */
int wm_gram(wm_ctx* ctx, const wm_plane* img, double* gram_out, int slot);

/* ---- Intra-frame sharding: one image split into row bands over several GPUs -----------------
 * No counterpart in the reference (one image = one device there); for single images too large or too urgent for one
 * GPU.  A context created for `rows` x `cols` then holds a BAND: its owned rows plus p/2 + 1 halo rows (2 for p = 3) of real image data on
 * every side that is not an image border (W likewise: the band's rows of the watermark file).  wm_band_configure names
 * the owned rows [own_lo, own_hi) in plane coordinates and the row count of the whole image; own_lo == 0 /
 * own_hi == rows mark true image borders (replicate padding applies there only).  Sweeps then sum and store the
 * owned rows only, and the caller all-reduces the partial totals between the phases (one process per GPU,
 * torch.distributed / RCCL; watermarking-gpu_amd/bands.py):
 *   embed : wm_gram (44 sums) -> SUM -> wm_band_solve -> wm_band_stats ({max|e|, sum}) -> MAX, SUM -> wm_band_embed
 *   detect: halo rows of y from the neighbour bands -> wm_gram -> SUM -> wm_band_solve
 *           -> wm_band_detect_sums ({<e_u,e_w>, |e_u|^2, |e_w|^2}) -> SUM -> corr = (float)dot / (float)(sqrt(nw) * sqrt(nu))
 * All five calls are synchronous.  own_hi == 0 switches band mode off.  In band mode wm_embed / wm_detect see only the
 * band and are not meaningful. */
int wm_band_configure(wm_ctx* ctx, int own_lo, int own_hi, long long rows_global);
/* totals[44*frames]: the all-reduced wm_gram sums; solves c (Watermark.cpp:203) into the slot; status_out[frames] may be NULL */
int wm_band_solve(wm_ctx* ctx, const double* totals, int frames, int* status_out, int slot);
/* out[2*frames]: {max|e| (1 for NVF), sum (m W)^2 without the 1/max^2} over the owned rows; needs wm_band_solve first (ME) */
int wm_band_stats(wm_ctx* ctx, int mask, const wm_plane* in_gray, double* out, int slot);
/* max_sum[2*frames]: the all-reduced wm_band_stats values; writes the owned rows of `out` (device planes, out must not
 * overlap in_gray); a_out[frames] (may be NULL) receives the strength (Watermark.cpp:170) */
int wm_band_embed(wm_ctx* ctx, int mask, const wm_plane* in_gray, const wm_plane* base, const wm_plane* out,
                  const double* max_sum, float* a_out, int slot);
/* out[3*frames]: {<e_u,e_w>, ||e_u||^2, ||e_w||^2} over the owned rows; needs wm_band_solve first */
int wm_band_detect_sums(wm_ctx* ctx, int mask, const wm_plane* img, double* out, int slot);

/* The same phases with the exchange RESIDENT IN DEVICE MEMORY ("one RCCL all-reduce of 44 doubles per sweep"):
 * every call only enqueues on the slot's stream -- give the slot the stream the caller's collectives are ordered on with
 * wm_set_stream -- and hands its totals over in device memory the caller owns; the caller all-reduces / all-gathers them in
 * place with RCCL on that stream (watermarking-gpu_amd/bands.py with backend "nccl"; ncclAllReduce / ncclAllGather from C++).
 * Nothing synchronises with the host until the caller reads a result.
 *   embed : wm_band_gram_dev -> all-reduce SUM [44] -> wm_band_solve_dev -> wm_band_stats_dev -> all-gather [2] per rank
 *           -> wm_band_embed_dev (folds the gathered parts in rank order: the same bits on every rank)
 *   detect: halo rows of y (ncclSend / ncclRecv) -> wm_band_gram_dev -> all-reduce -> wm_band_solve_dev
 *           -> wm_band_detect_sums_dev -> all-reduce SUM [3] -> wm_band_corr_dev
 * totals_dev[44*frames], max_sum_dev[2*frames], gathered_max_sum_dev[nparts][2*frames], sums_dev[3*frames] doubles;
 * a_dev[frames] (may be NULL; NaN for an unsolvable frame, whose output is the base: Watermark.cpp:164-165), corr_dev[frames]
 * (0.0f when unsolvable: Watermark.cpp:246-247) floats -- all device pointers. */
int wm_band_gram_dev(wm_ctx* ctx, const wm_plane* img, double* totals_dev, int slot);
int wm_band_solve_dev(wm_ctx* ctx, const double* totals_dev, int frames, int slot);
int wm_band_stats_dev(wm_ctx* ctx, int mask, const wm_plane* in_gray, double* max_sum_dev, int slot);
int wm_band_embed_dev(wm_ctx* ctx, int mask, const wm_plane* in_gray, const wm_plane* base, const wm_plane* out,
                      const double* gathered_max_sum_dev, int nparts, float* a_dev, int slot);
int wm_band_detect_sums_dev(wm_ctx* ctx, int mask, const wm_plane* img, double* sums_dev, int slot);
int wm_band_corr_dev(wm_ctx* ctx, const double* sums_dev, int frames, float* corr_dev, int slot);

/* waits for everything queued on `slot`, then delivers the scalar results; returns WM_OK,
 * WM_UNSOLVABLE if any delivered frame was unsolvable, or < 0 */
int wm_sync(wm_ctx* ctx, int slot);

/* stream plumbing: run a slot on the caller's hipStream_t (NULL restores the slot's own stream; the legacy default stream
 * is named by HIP's handle for it, hipStreamLegacy) */
int wm_set_stream(wm_ctx* ctx, int slot, void* hip_stream);
void* wm_get_stream(wm_ctx* ctx, int slot);

/* device memory helpers so that host code above this ABI needs no HIP headers (include/Watermark.hpp is plain C++) */
void* wm_dev_alloc(int device, size_t bytes);
void wm_dev_free(void* p);
int wm_memcpy_h2d(void* dst_device, const void* src_host, size_t bytes);
int wm_memcpy_d2h(void* dst_host, const void* src_device, size_t bytes);
int wm_device_count(void);

/* pinned host memory for WM_MEM_HOST planes (the reference's CL_MEM_ALLOC_HOST_PTR buffer, main.cpp:273-275) */
void* wm_host_alloc(size_t bytes);
void wm_host_free(void* p);

/* properties */
int wm_rows(const wm_ctx* ctx);
int wm_cols(const wm_ctx* ctx);
int wm_p(const wm_ctx* ctx);
float wm_strength_factor(const wm_ctx* ctx); /* Watermark.cpp:22 */
int wm_device(const wm_ctx* ctx);
const float* wm_w_device(const wm_ctx* ctx); /* device copy of W, row-major */

/* Memory-system yardstick of the box, independent of the engine's sweeps (bench.py's `membench` leg): one kernel per launch over
 * `bytes` of device memory this call allocates -- kind 0: pure store (16 B per lane, non-temporal, the store form k_embed
 * uses), 1: pure copy (16 B loads + the same stores; `bytes` read and `bytes` written), 2: pure read (16 B loads folded into a
 * checksum); kinds 3, 4, 5: the same three in the grid shape that streams fastest on MI355X (65536 blocks of 256 threads, one
 * element per thread; kinds 0-2 use 2048 blocks with four elements in flight per thread, a grid like the sweeps' own) --
 * launched back to back for at least `seconds`, every launch timed by events attached to the dispatch.  Writes the
 * mean launch duration in microseconds and the launch count; GB/s = bytes moved per launch / that.  Two boxes that run
 * k_embed at different speeds with equal clocks and power either differ here too (a slow-writing memory system) or do not
 * (then the difference is the kernel's). */
int wm_membench(int device, int kind, size_t bytes, double seconds, double* mean_us, int* launches);

/* per-kernel timing with hipEvents recorded on the launch stream (bench.py's roofline leg) */
int wm_prof_enable(wm_ctx* ctx, int on);
int wm_prof_reset(wm_ctx* ctx);
int wm_prof_kernel_count(void);
const char* wm_prof_kernel_name(int kernel_id);
/* launches and total milliseconds recorded for one kernel since the last reset (syncs the device) */
int wm_prof_get(wm_ctx* ctx, int kernel_id, uint64_t* launches, double* total_ms);

const char* wm_strerror(int code);
const char* wm_last_error(const wm_ctx* ctx); /* detail of the last failing call ("" if none) */
const char* wm_version(void);

#ifdef __cplusplus
}
#endif
#endif /* WM_H_ */
