// vsf_internal.h -- structures shared by the host side of the C ABI and the gfx950 kernels.
#ifndef VSF_INTERNAL_H_
#define VSF_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/vsf.h"

// FAST march kernel: a wave owns a band of 248 keypoint columns (lanes 1..62 x 4 px; lanes 0 and 63 carry the raw halo
// pixels and score the ONE pixel next to the band -- lane 0's last, lane 63's first: their circles reach no further than
// the lane's own dword and its inner neighbour's -- which is all the NMS of the band's edge columns needs) x a strip of 32
// rows.  (Rounds 1-3: 240 columns, lanes 1 and 62 scored four pixels each for the sake of one.)
#define VSF_FAST_BAND_COLS 248
#define VSF_FAST_HALF_COLS 120   // a last band of at most this many columns is walked two strips per wave (30 lanes x 4)
#define VSF_FAST_STRIP_ROWS 32
#define VSF_FAST_RS_STRIDE (VSF_FAST_STRIP_ROWS + 2)  // u16 row-start table per unit (SR + 1 used)
#define VSF_BLUR_MMA_ROWS 26     // output rows per step of the matrix-core blur (32 loaded rows - 2 x 3 halo rows)
#define VSF_BLUR_MMA_STEPS 16    // double steps (52 rows) per unit: a workgroup walks this many blocks down its band pair
#define VSF_BLUR_MMA_STEPS_SMALL 2  // ... for batches below 32 images
// The blurred levels are stored in tiles of 4 rows x 32 bytes (= one 128-byte cache line), tiles in row-major order:
// the descriptor kernel gathers 39 x 39 windows from them and a window then touches ~24 lines instead of 39..78,
// while the blur kernel still writes whole 32-byte sectors (8 lanes x 4 bytes).
// Byte offset of pixel (x, y) inside a level of row pitch `pitch` (a multiple of 64; rows padded to a multiple of 8):
#define VSF_BLUR_TILE_OFFSET(pitch, x, y) \
  ((uint32_t)((y) >> 2) * (uint32_t)((pitch) * 4) + ((uint32_t)((x) >> 5) << 7) + ((uint32_t)((y) & 3) << 5) + (uint32_t)((x) & 31))
#define VSF_SORT_LDS_ROWS 4096   // matches of one pair sorted in LDS (8 B each); larger pairs sort in HBM scratch
#define VSF_IC_ITEMS 320         // ICAngles disc items per byte phase: 31 rows x 9 dwords = 279, padded to 5 x 64

// ---- pyramid level descriptor (device-resident table, read through scalar loads) ----
struct VsfLevel {
  int32_t w, h;            // level size (cv::ORB layer size)
  int32_t pitch;           // bytes per row in the pyramid buffers (multiple of 64)
  uint32_t offset;         // byte offset of row 0 inside one image's pyramid block
  float scale;             // layerScale[level]
  int32_t nfeatures;       // per-level budget n_l
  int32_t x_lo, x_hi;      // keypoints may sit at x_lo <= x < x_hi  (runByImageBorder / FAST rim)
  int32_t y_lo, y_hi;
  int32_t fast_a0;         // x_lo rounded down to a multiple of 4: column origin of band 0
  int32_t nbands, nstrips; // FAST units of this level: unit = strip * nbands + band
  int32_t unit0;           // index of this level's first unit (per image)
  uint32_t cand_offset;    // u32 index of this level's first candidate segment (per image)
  int32_t seg_cap;         // capacity of one unit's candidate segment
  int32_t kp_offset;       // index of this level's final-keypoint segment (per image)
  int32_t kp_cap;          // its capacity
  int32_t blur_vec_end;    // columns [0, blur_vec_end) round half-even (SSE2 path), the rest half-up
  uint32_t xtab, ytab;     // entry offsets of this level's resize tables in the host-side Geometry (level >= 1)
  int32_t resize_rows;     // largest strip height (16, 8, 4) the shared-row resize kernel may use for this level, or 0
  int32_t resize_any8;     // 1: an 8-row strip starting at ANY row stays inside 10 consecutive source rows
  uint32_t rscale_x[2], rscale_y[2];  // bit patterns of cv::resize's double scale_x / scale_y from level - 1 (host-computed)
  uint32_t blur_tcol;      // matrix-core blur: index of this level's first pass-1 operand (4 per 64-column band) in the table
};

// Resize coefficients of one output column / row (cv::resize INTER_LINEAR 8u: xofs/ialpha resp. yofs/ibeta with the
// out-of-range taps already clamped, weights kept); evaluated in place by k_pyramid.hip, tabulated on the host.
struct VsfTap {
  uint16_t i0, i1;  // source indices of the two taps
  int16_t c0, c1;   // 11-bit fixed-point weights
};

// Candidate keypoint, 32 bit: score << 24 | y << 12 | x   (level coordinates, x,y < 4096)
#define VSF_CAND_PACK(x, y, s) (((uint32_t)(s) << 24) | ((uint32_t)(y) << 12) | (uint32_t)(x))
#define VSF_CAND_X(c) ((int)((c)&0xFFFu))
#define VSF_CAND_Y(c) ((int)(((c) >> 12) & 0xFFFu))
#define VSF_CAND_SCORE(c) ((int)((c) >> 24))

// Final per-level keypoint record (level coordinates).
struct VsfLevelKp {
  uint32_t xy;  // y << 12 | x
  float response;
  float angle;     // degrees (k_describe.hip orb_angle_kernel)
  float ca, sb;    // cos / sin of the angle as computeOrbDescriptors takes them
};

struct VsfGeom {
  int nlevels;
  int width, height;
  uint32_t pyr_bytes;       // one image's pyramid block
  uint32_t cand_entries;    // one image's candidate buffer (u32 entries)
  int nunits;               // FAST cells (32-row strip x 240-column band) per image
  int nwork_full, nwork_half;  // FAST work items: waves covering one cell / two cells of a narrow last band
  int lvlkp_entries;        // one image's level-keypoint buffer (VsfLevelKp entries)
  uint64_t pyramid_pixels;
};

// Per-context launch choices (vsf_set_option / vsf_get_option; the defaults are what the measurements of NOTES.md
// section 6 settled on).  Nothing in the library reads the environment: a switch is a call on a context.
struct VsfTuning {
  int fast_both_max = 16;  // VSF_OPT_FAST_BOTH_MAX: largest batch (images) whose full and half-wave FAST cells share one launch
  int select_wide = 1;     // VSF_OPT_SELECT_WIDE: 1 = a frame or two takes the 1024-thread whole-level selection class
  int pipe_priority = 0;     // VSF_OPT_PIPE_PRIORITY: stream priority of the pipelined pyramid chain (0 normal, 1 lowest, -1 highest)
  int pipe_after_fast = 1;   // VSF_OPT_PIPE_AFTER_FAST: the pipelined pyramid of call k + 1 starts behind call k's FAST (1) or at once (0)
  int select_big_class = 1;  // VSF_OPT_SELECT_BIG_CLASS: 1 = the widest levels of a batch take the 9 216-entry class
  int jpeg_serial = 0;     // 1: every file through the one-wave-per-image decoder (set when the parallel decoder's LDS is refused)
  int pyramid_few = 16;    // VSF_OPT_PYRAMID_FEW: largest batch (images) that takes the slab kernel for every level
  int pyramid_chain = 8;   // VSF_OPT_PYRAMID_CHAIN: levels per slab launch (0: keep the per-level launches)
  int pyramid_rows = 6;    // VSF_OPT_PYRAMID_ROWS: rows of the chain's last level per slab
  int pyramid_tail_min = 0;  // VSF_OPT_PYRAMID_TAIL_MIN: smallest batch (images) whose one-band levels take the image-major kernel whatever
                             // share of the chip it fills (0: only batches that fill three quarters of a round of workgroups)
  int observe_copy_thread = 1;  // VSF_OPT_OBSERVE_COPY_THREAD: a helper thread takes the right image's staging copy while frames stream in
  int observe_thread = 0;  // VSF_OPT_OBSERVE_THREAD: 1 = an ObserveImage queue of depth >= 4 gets a launcher thread
  int lds_limit = 0;       // largest dynamic LDS a workgroup may ask for on this device (queried at vsf_create)
};

// First HIP error a launcher or a stream-plumbing helper met since the last check (host thread-local: a context is driven
// by one host thread at a time).  Launchers return void; they hand failures to vsf_note() and the entry point that called
// them turns the noted error into VSF_ERR_HIP (VSF_STICKY in vsf_api.hip) -- a failed event wait would otherwise silently
// remove an ordering edge between two kernels.
extern thread_local int vsf_tls_hip_error;
inline void vsf_note(hipError_t e) {
  if (e != hipSuccess && vsf_tls_hip_error == 0) vsf_tls_hip_error = (int)e;
}
// What vsf_comm.hip needs of a context (vsf_ctx is private to vsf_api.hip).
hipStream_t vsf_ctx_stream(const vsf_ctx* ctx);
int vsf_ctx_device(const vsf_ctx* ctx);
void vsf_ctx_set_last_error(vsf_ctx* ctx, int code);  // what vsf_last_hip_error returns
constexpr int VSF_RCCL_ERROR_BASE = 10000;             // vsf_last_hip_error = 10000 + ncclResult_t after a failed RCCL call
// An error a launcher noted belongs to the CONTEXT whose entry point was running.  Every public entry point holds one of
// these: on the way out -- early returns included -- whatever is still noted in the thread's slot moves into the context
// (vsf_ctx::pending_hip) and is returned by that context's next checking call (VSF_STICKY), never by another context that
// happens to be driven from the same host thread.
void vsf_ctx_absorb_noted_error(vsf_ctx* ctx);
// ... and on the way IN every entry point but the ObserveImage queue's own first sends what waits in that queue, so that
// nothing else ever runs beside the queue's launcher thread (vsf_observe.hip).
void vsf_ctx_enter(vsf_ctx* ctx);
struct VsfErrorScope {
  vsf_ctx* ctx;
  explicit VsfErrorScope(vsf_ctx* c, bool enter = true) : ctx(c) {
    if (ctx && enter) vsf_ctx_enter(ctx);
  }
  ~VsfErrorScope() {
    if (ctx && vsf_tls_hip_error != 0) vsf_ctx_absorb_noted_error(ctx);
  }
  VsfErrorScope(const VsfErrorScope&) = delete;
  VsfErrorScope& operator=(const VsfErrorScope&) = delete;
};
// Raises the dynamic-LDS limit of the kernels that need more than the default 64 KB (k_frontend / k_pyramid / k_jpeg);
// called once per context creation, checked.
hipError_t vsf_prepare_sort_kernels(int lds_limit);
hipError_t vsf_prepare_pyramid_kernels(int lds_limit);
hipError_t vsf_prepare_jpeg_kernels(int lds_limit);

// Kernel launchers (implemented in the k_*.hip files). All asynchronous on `s`.
struct VsfDev {
  const VsfLevel* levels;   // [nlevels]
  const uint32_t* units;    // [nwork_full + nwork_half]: level << 24 | band << 16 | (first) strip
  uint8_t* pyr;             // [max_images][pyr_bytes]   unblurred levels 1..L-1 (level 0 is the input)
  uint8_t* blur;            // [max_images][pyr_bytes]   blurred levels 0..L-1
  uint32_t* cand;           // [max_images][cand_entries]  per-unit candidate segments (unit-local raster order)
  uint16_t* rowstart;       // [max_images][nunits][VSF_FAST_RS_STRIDE]  start of each row inside its segment
  uint32_t* scratch;        // [max_images][6 * cand_entries]   selection arrays, rank tables, masks when LDS is too small
  VsfLevelKp* lvlkp;        // [max_images][lvlkp_entries]
  const uint2* ic_table;    // [4][VSF_IC_ITEMS] ICAngles byte weights (k_describe.hip)
  int32_t* lvl_count;       // [max_images][nlevels]
  int32_t* status;          // device status word (bit 0: capacity overflow)
  int status_stride;        // 0: that one word; 1: status[image] (a batch of the ObserveImage queue: one word per image)
  const VsfTuning* tune;    // the context's launch choices (host memory)
};

struct VsfImages {
  const uint8_t* base;      // level-0 images
  size_t image_stride;
  size_t row_stride;
  int n;
};

// Optional extra streams a launcher may fork independent kernel chains onto (joined back before it returns).
#define VSF_SIDE_STREAMS 1
struct VsfSideStream {
  hipStream_t stream[VSF_SIDE_STREAMS];
  hipEvent_t fork, join[VSF_SIDE_STREAMS];
  int n;
};
void vsf_launch_bayer_bg_gray(const uint8_t* d_src, int n, int w, int h, size_t src_image_stride, int src_pitch,
                              uint8_t* d_dst, size_t dst_image_stride, int dst_pitch, hipStream_t s);
void vsf_launch_pyramid(const VsfDev& d, const VsfGeom& g, const VsfLevel* h_levels, const VsfImages& im,
                        hipStream_t s, const VsfSideStream* side);
// threshold: FAST threshold; nms == 0 keeps every corner (standalone FAST only).
void vsf_launch_fast(const VsfDev& d, const VsfGeom& g, const VsfImages& im, int threshold, int nms, hipStream_t s,
                     int resident_waves_per_simd = 0, int n_cus = 0, uint32_t* d_cell_counters = nullptr);
void vsf_launch_select(const VsfDev& d, const VsfGeom& g, const VsfLevel* h_levels, const VsfImages& im,
                       hipStream_t s);
void vsf_launch_retain_best_test(uint2* d_data, uint32_t* d_tables, int n, int n_points, int use_lds, int mode,
                                 int* d_out_n, hipStream_t s);
// Matrix-core blur (k_blur.hip, round 3): units = level << 24 | band pair << 16 | first double step << 8 | double steps.
void vsf_launch_blur_mma(const VsfDev& d, const VsfGeom& g, const VsfImages& im, const uint32_t* d_units, int nunits,
                         const uint4* d_tcol, const uint4* d_tv, int bias, hipStream_t s);
void vsf_launch_describe(const VsfDev& d, const VsfGeom& g, const VsfImages& im, int max_keypoints,
                         vsf_keypoint* d_kp, uint8_t* d_desc, int32_t* d_counts, hipStream_t s);
void vsf_launch_fast_emit(const VsfDev& d, const VsfGeom& g, int n_images, int max_keypoints, vsf_keypoint* d_kp,
                          int32_t* d_counts, hipStream_t s);
void vsf_launch_knn2(const uint8_t* d_desc, const int32_t* d_counts, size_t set_stride, const int32_t* d_q_set,
                     const int32_t* d_t_set, int n_pairs, int max_rows, int32_t* d_idx2, int32_t* d_dist2,
                     hipStream_t s, int rows_hint = 0);  // rows_hint: expected rows per set (0: unknown), speed only
void vsf_launch_ratio_compact(const int32_t* d_counts, const int32_t* d_q_set, const int32_t* d_t_set, int n_pairs,
                              int max_rows, const int32_t* d_idx2, const int32_t* d_dist2, uint32_t ratio_num,
                              uint32_t ratio_shift, vsf_dmatch* d_matches, int32_t* d_nmatches, int32_t* d_status,
                              hipStream_t s);

// k_frontend.hip.  The fundamental matrix: h_F (host, 9 floats) travels by value in the kernel
// arguments; with h_F == NULL the kernel reads d_F (device-visible memory: vsf_observe's pinned per-call parameters).
struct VsfF9 {
  float v[9];
};
void vsf_launch_stereo_filter(const vsf_keypoint* d_kp, const uint8_t* d_desc, const vsf_dmatch* d_matches,
                              const int32_t* d_nmatches, int n_frames, int max_rows, const float* d_F, const float* h_F,
                              int order, const float* d_thr_override, float thr_in, float* d_residual, float* d_mean, float* d_thr,
                              vsf_keypoint* d_kp_out, uint8_t* d_desc_out, int32_t* d_counts_out, hipStream_t s);
void vsf_launch_stereo_residuals(const vsf_keypoint* d_kp, const vsf_dmatch* d_matches, const int32_t* d_nmatches,
                                 int n_frames, int max_rows, const float* d_F, const float* h_F, int order, float* d_residual,
                                 float* d_mean, hipStream_t s);
void vsf_launch_stereo_filter_only(const vsf_keypoint* d_kp, const uint8_t* d_desc, const vsf_dmatch* d_matches,
                                   const int32_t* d_nmatches, int n_frames, int max_rows, const float* d_residual,
                                   const float* d_thr, vsf_keypoint* d_kp_out, uint8_t* d_desc_out,
                                   int32_t* d_counts_out, hipStream_t s, const int32_t* d_out_sets = nullptr,
                                   int32_t* d_set_counts = nullptr);
// RemoveAmbigStereo of one frame in one launch (residuals, ordered mean, threshold hand-over through *d_thr_state, filter);
// false: the frame's capacity does not fit the kernel's LDS and nothing was launched.
bool vsf_launch_stereo_one_frame(const vsf_keypoint* d_kp, const uint8_t* d_desc, const vsf_dmatch* d_matches,
                                 const int32_t* d_nmatches, int max_rows, const float* h_F, int order, float* d_mean,
                                 float* d_thr, float* d_thr_state, vsf_keypoint* d_kp_out, uint8_t* d_desc_out,
                                 int32_t* d_counts_out, const int32_t* d_out_sets, int32_t* d_set_counts, hipStream_t s);
// k_points.hip (SURVEY 8(f) row f2 + the compact gather payload)
void vsf_launch_stereo_thresholds(const float* d_means, int n, float* d_state, float* d_thr, hipStream_t s);
void vsf_launch_fill_stereo_sets(int32_t* d_sets, int n_frames, hipStream_t s);  // [0..n): 2f + 1, [n..2n): 2f
void vsf_launch_vision_features(const vsf_keypoint* d_kp, const int32_t* d_counts, const uint64_t* d_pairs,
                                const int32_t* d_npairs, int n_frames, int max_rows, const vsf_calibration& c,
                                vsf_vision_feature* d_out, int32_t* d_nfeatures, int32_t* d_npoints, hipStream_t s);
void vsf_launch_pack_outputs(const vsf_vision_feature* d_features, const int32_t* d_nfeatures, int n_frames,
                             const uint64_t* d_pairs, const int32_t* d_npairs, int n_pairs, int max_rows,
                             uint8_t* d_payload, uint32_t cap_bytes, uint32_t* d_offsets, int32_t* d_status,
                             hipStream_t s);
void vsf_launch_sort_trim(const vsf_dmatch* d_matches, const int32_t* d_nmatches, int n_pairs, int max_rows,
                          float best_percent, const float* d_best_percent_of, void* d_scratch, uint64_t* d_pairs,
                          int32_t* d_npairs, hipStream_t s, bool force_serial = false, int lds_limit = 160 * 1024);
// The ObserveImage queue's output kernel (k_frontend.hip): one compact result per frame of a batch, written into pinned
// host memory.
#define VSF_OBSERVE_MAX_PAIRS 64
struct VsfObserveFrame {  // per frame of a batch; pinned host memory the kernels read directly
  int32_t left_set;  // descriptor set that holds the filtered left frame
  int32_t n_past;    // kept frames it is matched against (temporal factors)
  int32_t tp0;       // index of its first temporal pair in the batch's pair list (its right -> left pair is pair f)
  int32_t out_slot;  // result slot: out + out_slot * out_stride
};
struct VsfObserveArgs {
  int n_frames, max_rows;
  const int32_t* counts_raw;          // [2n] keypoints of the left / right images
  const int32_t* nmatches;            // [n] raw stereo matches
  const int32_t* counts_f;            // [2n] features of the filtered left / right frames
  const int32_t* npoints;             // [n] triangulated points
  const float* means;                 // [n]
  const float* thr;                   // [n] thresholds applied
  const vsf_vision_feature* features; // [n][max_rows]
  const vsf_keypoint* kp_f;           // [2n][max_rows] filtered keypoints (left, right per frame)
  const uint8_t* desc_sets;           // descriptor sets [..][max_rows][32]
  const uint64_t* pairs;              // [pairs][max_rows][2]
  const int32_t* npairs;              // [pairs]
  int32_t* status;                    // [2n] a status word per image of the batch: read into the header and cleared
  const VsfObserveFrame* frames;      // [n]
  uint8_t* out;                       // pinned host memory (device-visible)
  size_t out_stride;
  uint32_t out_cap;
};
void vsf_launch_observe_pack(const VsfObserveArgs& a, int max_pairs_per_frame, hipStream_t s);

// k_jpeg.hip (cv::imdecode(IMREAD_GRAYSCALE) for baseline JPEG)
#ifdef __cplusplus
#include <vector>
struct VsfJpegPlan {  // layout of one upload: [image descriptors | decode order | table sets | progressive scans | their
                      // Huffman tables | entropy-coded segments]
  size_t off_images = 0, off_index = 0, off_tables = 0, off_scans = 0, off_prog_huff = 0, off_stream = 0, total = 0;
  std::vector<uint8_t> head;                // everything in front of the segments
  std::vector<size_t> scan_begin;           // per file: where its entropy-coded segment starts
  std::vector<uint32_t> stream_off, stream_len;
  int n_par = 0;             // files without restart intervals: they take the self-synchronising parallel decode
  int n_prog = 0;            // progressive files: scan after scan into the coefficient buffer, one wave per file
  int n_prog_huff = 0;       // ... and the Huffman tables of their scans (expanded on the device)
  int max_luma_blocks = 0;   // luminance blocks of the (padded) image, largest over the batch
  int max_slots = 1;         // Huffman tables one file's scan uses, largest over the batch
};
vsf_status vsf_jpeg_plan(const uint8_t* const* jpeg, const size_t* nbytes, int n, int width, int height, bool force_serial,
                         VsfJpegPlan* plan);
void vsf_jpeg_fill(const VsfJpegPlan& plan, const uint8_t* const* jpeg, int n, uint8_t* dst);
#endif
// k_png.hip / vsf_png_host.cc (row f4: cv::imdecode(IMREAD_GRAYSCALE) for grayscale PNG)
#ifdef __cplusplus
struct VsfPngPlan {  // layout of one upload: [image descriptors | ends of the IDAT payloads | palette / gamma tables | zlib streams]
  size_t off_images = 0, off_pieces = 0, off_tables = 0, off_stream = 0, total = 0;
  std::vector<uint8_t> head;
  std::vector<uint32_t> piece_first, piece_off, piece_len;  // per file: its IDAT payloads (offset in the file, length)
  std::vector<uint32_t> stream_off, stream_len;
  size_t filtered_stride = 0;  // device scratch per image: the inflated scanlines
  bool any_rgb = false;        // a file of colour type 2 or 6 among them (a kernel of its own)
  bool any_general = false;    // a palette file or an interlaced gray one among them (another)
};
vsf_status vsf_png_plan(const uint8_t* const* png, const size_t* nbytes, int n, int width, int height, VsfPngPlan* plan);
void vsf_png_fill(const VsfPngPlan& plan, const uint8_t* const* png, int n, uint8_t* dst);
#endif
void vsf_launch_png_decode(const uint8_t* d_blob, size_t off_images, size_t off_pieces, size_t off_tables, size_t off_stream, int n, int width, int height,
                           uint8_t* d_filtered, size_t filtered_stride, int32_t* d_file_status, uint8_t* d_dst,
                           size_t dst_image_stride, int dst_pitch, int32_t* d_status, bool any_general, bool any_rgb, hipStream_t s);
size_t vsf_jpeg_clean_bytes(size_t stream_bytes, int n_par);
size_t vsf_jpeg_prog_huff_bytes(int n_tables);  // device scratch for the expanded tables of progressive scans
void vsf_launch_jpeg_decode(const uint8_t* d_blob, size_t off_images, size_t off_index, size_t off_tables, size_t off_scans,
                            size_t off_prog_huff, size_t off_stream, size_t total, int n_par, int n_prog, int n_prog_huff,
                            void* d_prog_huff, int n_ser, int max_luma_blocks, int max_slots, int width, int height,
                            uint8_t* d_clean,
                            int16_t* d_coef, size_t coef_stride, uint8_t* d_dst, size_t dst_image_stride, int dst_pitch,
                            int32_t* d_status, hipStream_t s, bool prog_serial, int32_t* d_prog_flags);
// prog_serial: progressive files scan after scan in one wave; d_prog_flags [n_prog]: scratch of the pipelined form

#endif  // VSF_INTERNAL_H_
